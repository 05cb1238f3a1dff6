// ABI bookkeeping entry points.
#include "common.h"

extern "C" int ctcasr_abi_version(void) { return CTCASR_ABI_VERSION; }

extern "C" const char *ctcasr_error_string(int code) {
    switch (code) {
        case CTCASR_OK: return "ok";
        case CTCASR_ERR_BAD_ARGUMENT: return "bad argument";
        case CTCASR_ERR_UNSUPPORTED: return "unsupported configuration";
        case CTCASR_ERR_WORKSPACE: return "workspace missing or too small";
        case CTCASR_ERR_LAUNCH: return "kernel launch failed";
        case CTCASR_ERR_TIMEOUT: return "in-kernel wait timed out";
        default: return "unknown error";
    }
}

