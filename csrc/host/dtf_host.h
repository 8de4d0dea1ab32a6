// Host-side native runtime pieces (no GPU dependency): CRC32C, tfevents writer, rendezvous store,
// checkpoint container, bottleneck text codec.  C ABI consumed through ctypes.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>

#define DTF_HOST_API extern "C" __attribute__((visibility("default")))

namespace dtf {
uint32_t crc32c_extend(uint32_t crc, const void* data, size_t n);
uint32_t masked_crc32c(const void* data, size_t n);

// minimal protobuf wire-format encoder
struct PbWriter {
  std::string buf;
  void varint(uint64_t v) {
    while (v >= 0x80) { buf.push_back((char)((v & 0x7f) | 0x80)); v >>= 7; }
    buf.push_back((char)v);
  }
  void tag(int field, int wire) { varint(((uint64_t)field << 3) | (uint64_t)wire); }
  void f_varint(int field, uint64_t v) { tag(field, 0); varint(v); }
  void f_double(int field, double d) { tag(field, 1); buf.append(reinterpret_cast<const char*>(&d), 8); }
  void f_float(int field, float f) { tag(field, 5); buf.append(reinterpret_cast<const char*>(&f), 4); }
  void f_bytes(int field, const void* p, size_t n) { tag(field, 2); varint(n); buf.append(static_cast<const char*>(p), n); }
  void f_string(int field, const std::string& s) { f_bytes(field, s.data(), s.size()); }
  void f_packed_doubles(int field, const double* p, size_t n) { tag(field, 2); varint(n * 8); buf.append(reinterpret_cast<const char*>(p), n * 8); }
};

// Timed condition-variable waits.  Production builds wait on the STEADY clock (a wall-clock jump must not stretch or cut
// a timeout).  Under ThreadSanitizer the same wait goes through the system clock = pthread_cond_timedwait, which that
// runtime models; wait_for's pthread_cond_clockwait is invisible to it and yields false "double lock" reports.
#if defined(__SANITIZE_THREAD__)
#define DTF_TSAN_BUILD 1
#elif defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define DTF_TSAN_BUILD 1
#endif
#endif
template <class Pred>
inline bool cv_wait_ms(std::condition_variable& cv, std::unique_lock<std::mutex>& l, int64_t ms, Pred pred) {
#ifdef DTF_TSAN_BUILD
  return cv.wait_until(l, std::chrono::system_clock::now() + std::chrono::milliseconds(ms), pred);
#else
  return cv.wait_until(l, std::chrono::steady_clock::now() + std::chrono::milliseconds(ms), pred);
#endif
}
inline void cv_wait_ms(std::condition_variable& cv, std::unique_lock<std::mutex>& l, int64_t ms) {
#ifdef DTF_TSAN_BUILD
  cv.wait_until(l, std::chrono::system_clock::now() + std::chrono::milliseconds(ms));
#else
  cv.wait_until(l, std::chrono::steady_clock::now() + std::chrono::milliseconds(ms));
#endif
}
}  // namespace dtf
