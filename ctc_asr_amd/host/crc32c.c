/* Host-side helper library (libctcasr_host.so, plain C, no HIP): CRC-32C for TensorFlow's
 * tensor-bundle checkpoint files (ctc_asr_amd/tf_bundle.py).  Kept out of
 * libctcasr.so on purpose: that library's ABI is the device hot path only. */
#include "../../include/ctcasr_host.h"

#include <string.h>

/* CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8; `crc` chains calls: pass
 * 0 first, then the previous return value. */
uint32_t ctcasr_host_crc32c(const void *data, size_t size, uint32_t crc) {
    static uint32_t table[8][256];
    static int ready = 0;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0x82F63B78u : 0u);
            table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t)
                table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFFu];
        ready = 1;
    }
    const unsigned char *p = (const unsigned char *)data;
    uint32_t c = ~crc;
    while (size >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = table[7][lo & 0xFFu] ^ table[6][(lo >> 8) & 0xFFu] ^ table[5][(lo >> 16) & 0xFFu] ^
            table[4][lo >> 24] ^ table[3][hi & 0xFFu] ^ table[2][(hi >> 8) & 0xFFu] ^
            table[1][(hi >> 16) & 0xFFu] ^ table[0][hi >> 24];
        p += 8;
        size -= 8;
    }
    while (size--) c = (c >> 8) ^ table[0][(c ^ *p++) & 0xFFu];
    return ~c;
}
