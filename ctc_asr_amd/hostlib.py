"""ctypes binding of ``libctcasr_host.so`` (``include/ctcasr_host.h``): host-only helpers that
are deliberately not part of the device hot-path ABI."""

import ctypes
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, 'libctcasr_host.so')

SIGNATURES = {
    'ctcasr_host_crc32c': (ctypes.c_uint32, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]),
}

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('{} is missing - build it with `python -m ctc_asr_amd.build`.'
                               .format(LIB_PATH))
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _lib = lib
    return _lib


def crc32c(data, crc=0):
    """CRC-32C of a bytes-like object / contiguous numpy array in host memory."""
    view = memoryview(data).cast('B')
    if len(view) == 0:
        return int(crc)
    buf = (ctypes.c_char * len(view)).from_buffer_copy(view) if view.readonly \
        else (ctypes.c_char * len(view)).from_buffer(view)
    return int(load().ctcasr_host_crc32c(ctypes.addressof(buf), len(view), int(crc)))
