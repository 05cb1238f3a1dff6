/* ctcasr_host.h — host-only helpers (libctcasr_host.so, plain C, no device code).  Not part of
 * the hot-path ABI (include/ctcasr.h): these serve the checkpoint import/export tools. */
#ifndef CTCASR_HOST_H_
#define CTCASR_HOST_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CRC-32C (Castagnoli) of a HOST buffer, chained through `crc` (0 to start): the checksum
 * TensorFlow's tensor-bundle files carry per tensor and per table block
 * (tensorflow/core/lib/hash/crc32c.h; files written by the reference's tf.estimator,
 * asr/train.py:31-55). */
uint32_t ctcasr_host_crc32c(const void *data, size_t size, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif /* CTCASR_HOST_H_ */
