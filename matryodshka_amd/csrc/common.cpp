// Error text + version of libmsi_hip.so.
#include <cstring>

#include "msi_common.h"

namespace msi {

char *error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace msi

extern "C" {

const char *msi_version(void) { return "msi_hip 0.4 (gfx950)"; }

int32_t msi_abi_version(void) { return MSI_ABI_VERSION; }

const char *msi_last_error_string(void) { return msi::error_buffer(); }

// CRC-32C (Castagnoli, reflected polynomial 0x82f63b78), table-driven, continuing from `crc` (0 for a new message):
// the checksum TensorFlow stores (masked) per tensor in a checkpoint's BundleEntryProto (tf_checkpoint.py verifies
// every tensor it restores with it).
uint32_t msi_crc32c_host(const void *data, size_t n, uint32_t crc) {
  static uint32_t table[8][256];
  static bool ready = false;   // (idempotent initialisation: a race writes the same values)
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82f63b78u : c >> 1;
      table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xffu];
    ready = true;
  }
  const unsigned char *p = static_cast<const unsigned char *>(data);
  uint32_t c = ~crc;
  while (n >= 8) {   // slicing-by-8
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = table[7][lo & 0xffu] ^ table[6][(lo >> 8) & 0xffu] ^ table[5][(lo >> 16) & 0xffu] ^ table[4][lo >> 24] ^
        table[3][hi & 0xffu] ^ table[2][(hi >> 8) & 0xffu] ^ table[1][(hi >> 16) & 0xffu] ^ table[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = table[0][(c ^ *p++) & 0xffu] ^ (c >> 8);
  return ~c;
}

}  // extern "C"
