// K10 beam search entry points (device kernel lands in a later commit of this round).
#include "common.h"

extern "C" size_t ctcasr_ctc_beam_workspace_bytes(int T, int B, int C, int beam_width) {
    (void)T; (void)B; (void)C; (void)beam_width;
    return 256;
}

extern "C" int ctcasr_ctc_beam_decode(const float *logits, const int32_t *seq_len, int T, int B,
                                      int C, int blank, int beam_width, int norm_mode,
                                      int32_t *out, int32_t *out_len, float *logp, void *workspace,
                                      size_t workspace_bytes, ctcasr_stream_t stream) {
    (void)logits; (void)seq_len; (void)T; (void)B; (void)C; (void)blank; (void)beam_width;
    (void)norm_mode; (void)out; (void)out_len; (void)logp; (void)workspace;
    (void)workspace_bytes; (void)stream;
    return CTCASR_ERR_UNSUPPORTED;
}
