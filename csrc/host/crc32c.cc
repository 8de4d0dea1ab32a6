// CRC32C (Castagnoli) + the TFRecord "masked" variant, slicing-by-8 software implementation.
// Used by the tfevents writer (component C24) and the checkpoint container (C25).
#include "dtf_host.h"

namespace {
uint32_t g_table[8][256];
bool g_init = false;

void init_tables() {
  const uint32_t poly = 0x82F63B78u;  // reflected 0x1EDC6F41
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ poly : (c >> 1);
    g_table[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = g_table[0][i];
    for (int t = 1; t < 8; ++t) {
      c = g_table[0][c & 0xff] ^ (c >> 8);
      g_table[t][i] = c;
    }
  }
  g_init = true;
}
struct Init { Init() { init_tables(); } } g_initializer;
}  // namespace

uint32_t dtf::crc32c_extend(uint32_t crc, const void* data, size_t n) {
  if (!g_init) init_tables();
  const uint8_t* p = static_cast<const uint8_t*>(data);
  uint32_t c = crc ^ 0xFFFFFFFFu;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7)) { c = g_table[0][(c ^ *p++) & 0xff] ^ (c >> 8); --n; }
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    v ^= c;
    c = g_table[7][v & 0xff] ^ g_table[6][(v >> 8) & 0xff] ^ g_table[5][(v >> 16) & 0xff] ^
        g_table[4][(v >> 24) & 0xff] ^ g_table[3][(v >> 32) & 0xff] ^ g_table[2][(v >> 40) & 0xff] ^
        g_table[1][(v >> 48) & 0xff] ^ g_table[0][(v >> 56) & 0xff];
    p += 8;
    n -= 8;
  }
  while (n--) c = g_table[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

uint32_t dtf::masked_crc32c(const void* data, size_t n) {
  const uint32_t crc = crc32c_extend(0, data, n);
  return ((crc >> 15) | (crc << 17)) + 0xa282ead8u;
}

DTF_HOST_API uint32_t dtf_crc32c(const void* data, uint64_t n) { return dtf::crc32c_extend(0, data, (size_t)n); }
DTF_HOST_API uint32_t dtf_masked_crc32c(const void* data, uint64_t n) { return dtf::masked_crc32c(data, (size_t)n); }
