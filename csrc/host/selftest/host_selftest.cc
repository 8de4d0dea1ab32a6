// Self-test of the host runtime (csrc/host) as ONE native executable, so it can be built with the sanitizers
// (-fsanitize=address,undefined and -fsanitize=thread, CPU only; tests/test_sanitizers.py) without having to inject a
// sanitizer runtime into the Python interpreter.  It mirrors tests/test_host_native.py: CRC32C vectors, the events
// writer (records framed with verified CRCs, producer + concurrent flusher), the rendezvous store under many client
// threads with connection churn and a stop with clients still attached, and the checkpoint container round trip.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../dtf_host.h"

extern "C" {
uint32_t dtf_crc32c(const void*, uint64_t);
uint32_t dtf_masked_crc32c(const void*, uint64_t);
void* dtf_events_open(const char*);
void dtf_events_begin(void*, int64_t, double);
void dtf_events_scalar(void*, const char*, float);
void dtf_events_histogram(void*, const char*, double, double, double, double, double, int, const double*, const double*);
void dtf_events_end(void*);
void dtf_events_flush(void*);
uint64_t dtf_events_num_records(void*);
void dtf_events_close(void*);
void* dtf_store_server_start(int);
int dtf_store_server_port(void*);
void dtf_store_server_stop(void*);
void* dtf_store_connect(const char*, int, int);
int64_t dtf_store_set(void*, const char*, const void*, uint64_t);
int64_t dtf_store_get(void*, const char*, void*, uint64_t, int64_t);
int64_t dtf_store_add(void*, const char*, int64_t);
int64_t dtf_store_check(void*, const char*);
int64_t dtf_store_delete(void*, const char*);
void dtf_store_close(void*);
void* dtf_ckpt_begin(const char*);
int dtf_ckpt_add(void*, const char*, int, int, const uint64_t*, const void*, uint64_t);
int dtf_ckpt_finish(void*);
void* dtf_ckpt_open(const char*);
int dtf_ckpt_count(void*);
int dtf_ckpt_entry(void*, int, char*, int, int*, int*, uint64_t*, uint64_t*);
int dtf_ckpt_read(void*, int, void*);
void dtf_ckpt_close(void*);
void* dtf_bundle_begin(const char*);
int dtf_bundle_add(void*, const char*, int, int, const int64_t*, const void*, uint64_t);
int dtf_bundle_finish(void*);
void* dtf_bundle_open(const char*);
int dtf_bundle_count(void*);
int dtf_bundle_entry(void*, int, char*, int, int*, int*, int64_t*, uint64_t*);
int dtf_bundle_read(void*, int, void*);
void dtf_bundle_close(void*);
int64_t dtf_parse_floats(const char*, uint64_t, float*, uint64_t);
int64_t dtf_format_floats(const float*, uint64_t, char*, uint64_t);
}

static int g_fail = 0;
#define EXPECT(c)                                                          \
  do {                                                                     \
    if (!(c)) {                                                            \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);       \
      ++g_fail;                                                            \
    }                                                                      \
  } while (0)

static void test_crc() {
  EXPECT(dtf_crc32c("123456789", 9) == 0xE3069283u);
  std::vector<unsigned char> z(32, 0), f(32, 0xff);
  EXPECT(dtf_crc32c(z.data(), 32) == 0x8A9136AAu);
  EXPECT(dtf_crc32c(f.data(), 32) == 0x62A8AB43u);
  // any split point gives the same value (the sliced tables vs. the byte loop)
  std::vector<unsigned char> buf(1031);
  for (size_t i = 0; i < buf.size(); ++i) buf[i] = (unsigned char)(i * 131 + 7);
  const uint32_t whole = dtf::crc32c_extend(0, buf.data(), buf.size());
  for (size_t cut : {size_t(1), size_t(3), size_t(8), size_t(517), size_t(1030)})
    EXPECT(dtf::crc32c_extend(dtf::crc32c_extend(0, buf.data(), cut), buf.data() + cut, buf.size() - cut) == whole);
}

static bool read_records(const char* path, size_t* count) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  *count = 0;
  for (;;) {
    unsigned char hdr[12];
    const size_t got = fread(hdr, 1, 12, f);
    if (got == 0) break;
    if (got != 12) { fclose(f); return false; }
    uint64_t len; uint32_t lcrc;
    memcpy(&len, hdr, 8); memcpy(&lcrc, hdr + 8, 4);
    if (dtf_masked_crc32c(hdr, 8) != lcrc || len > (1u << 24)) { fclose(f); return false; }
    std::vector<unsigned char> body(len + 4);
    if (fread(body.data(), 1, len + 4, f) != len + 4) { fclose(f); return false; }
    uint32_t dcrc; memcpy(&dcrc, body.data() + len, 4);
    if (dtf_masked_crc32c(body.data(), len) != dcrc) { fclose(f); return false; }
    ++*count;
  }
  fclose(f);
  return true;
}

static void test_events(const std::string& dir) {
  const std::string path = dir + "/events.out.tfevents.selftest";
  void* w = dtf_events_open(path.c_str());
  EXPECT(w != nullptr);
  if (!w) return;
  std::atomic<bool> done{false};
  std::thread flusher([&] { while (!done.load()) { dtf_events_flush(w); std::this_thread::yield(); } });
  const int n = 2000;
  double limits[4] = {-1.0, 0.0, 1.0, 1e300}, counts[4] = {1, 2, 3, 0};
  for (int i = 0; i < n; ++i) {
    dtf_events_begin(w, i, 1.0e9 + i);
    dtf_events_scalar(w, "loss", 1.0f / (1 + i));
    dtf_events_scalar(w, "accuracy", 0.5f);
    if (i % 100 == 0) dtf_events_histogram(w, "h", -1, 1, 6, 0.5, 3.5, 4, limits, counts);
    dtf_events_end(w);
  }
  done.store(true);
  flusher.join();
  dtf_events_flush(w);
  EXPECT(dtf_events_num_records(w) == (uint64_t)n + 1);  // + the file-version record
  dtf_events_close(w);
  size_t cnt = 0;
  EXPECT(read_records(path.c_str(), &cnt));
  EXPECT(cnt == (size_t)n + 1);
}

static void test_store() {
  void* srv = dtf_store_server_start(0);
  EXPECT(srv != nullptr);
  if (!srv) return;
  const int port = dtf_store_server_port(srv);
  const int T = 8, N = 300;
  std::vector<std::thread> th;
  std::atomic<int> bad{0};
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      void* c = dtf_store_connect("127.0.0.1", port, 5000);
      if (!c) { ++bad; return; }
      char key[32];
      snprintf(key, sizeof key, "k/%d", t);
      for (int i = 0; i < N; ++i) {
        if (dtf_store_add(c, "ctr", 1) <= 0) ++bad;
        const int v = t * 100000 + i;
        if (dtf_store_set(c, key, &v, sizeof v) < 0) ++bad;
        int back = -1;
        if (dtf_store_get(c, key, &back, sizeof back, 1000) != (int64_t)sizeof back || back != v) ++bad;
        if (i % 50 == 49) {  // connection churn: the server must forget the closed descriptor (and not close a reused one)
          dtf_store_close(c);
          c = dtf_store_connect("127.0.0.1", port, 5000);
          if (!c) { ++bad; return; }
        }
      }
      // a blocking get that another thread satisfies
      char peer[32];
      snprintf(peer, sizeof peer, "done/%d", (t + 1) % T);
      char mine[32];
      snprintf(mine, sizeof mine, "done/%d", t);
      const char one = 1;
      dtf_store_set(c, mine, &one, 1);
      char got = 0;
      if (dtf_store_get(c, peer, &got, 1, 10000) != 1 || got != 1) ++bad;
      dtf_store_close(c);
    });
  for (auto& x : th) x.join();
  EXPECT(bad.load() == 0);
  void* c = dtf_store_connect("127.0.0.1", port, 5000);
  EXPECT(c != nullptr);
  if (c) {
    EXPECT(dtf_store_add(c, "ctr", 0) == (int64_t)T * N);
    EXPECT(dtf_store_check(c, "k/0") == 1);
    EXPECT(dtf_store_delete(c, "k/0") >= 0);
    EXPECT(dtf_store_check(c, "k/0") == 0);
    char b;
    EXPECT(dtf_store_get(c, "never", &b, 1, 50) < 0);  // times out, does not hang
  }
  // stop with clients attached (one idle, one blocked in a get)
  void* blocked = dtf_store_connect("127.0.0.1", port, 5000);
  std::thread waiter([&] { char b; dtf_store_get(blocked, "never2", &b, 1, 20000); });
  std::this_thread::sleep_for(std::chrono::milliseconds(100));
  dtf_store_server_stop(srv);
  waiter.join();
  if (c) { EXPECT(dtf_store_add(c, "ctr", 1) < 0); dtf_store_close(c); }
  dtf_store_close(blocked);
}

static void test_ckpt(const std::string& dir) {
  const std::string path = dir + "/selftest.ckpt";
  std::vector<float> a(1000);
  for (size_t i = 0; i < a.size(); ++i) a[i] = std::sin((float)i);
  int64_t step = 1234;
  void* w = dtf_ckpt_begin(path.c_str());
  EXPECT(w != nullptr);
  uint64_t d2[2] = {10, 100}, d0[1] = {0};
  EXPECT(dtf_ckpt_add(w, "conv1/weights", 0, 2, d2, a.data(), a.size() * 4) == 0);
  EXPECT(dtf_ckpt_add(w, "Variable", 1, 0, d0, &step, 8) == 0);
  EXPECT(dtf_ckpt_finish(w) == 0);
  void* r = dtf_ckpt_open(path.c_str());
  EXPECT(r != nullptr);
  if (r) {
    EXPECT(dtf_ckpt_count(r) == 2);
    char name[64]; int dt, nd; uint64_t dims[8], nb;
    EXPECT(dtf_ckpt_entry(r, 0, name, sizeof name, &dt, &nd, dims, &nb) == 0);
    EXPECT(std::string(name) == "conv1/weights" && nd == 2 && dims[0] == 10 && dims[1] == 100 && nb == 4000);
    std::vector<float> b(1000);
    EXPECT(dtf_ckpt_read(r, 0, b.data()) == 0);
    EXPECT(memcmp(a.data(), b.data(), 4000) == 0);
    EXPECT(dtf_ckpt_read(r, 7, b.data()) != 0);
    dtf_ckpt_close(r);
  }
  // a truncated file is refused, not read past its end
  FILE* f = fopen(path.c_str(), "rb");
  std::vector<char> whole(1 << 16);
  const size_t n = fread(whole.data(), 1, whole.size(), f);
  fclose(f);
  const std::string cut = dir + "/cut.ckpt";
  f = fopen(cut.c_str(), "wb");
  fwrite(whole.data(), 1, n / 2, f);
  fclose(f);
  void* r2 = dtf_ckpt_open(cut.c_str());
  if (r2) {
    std::vector<float> b(1000);
    int bad = 0;
    for (int i = 0; i < dtf_ckpt_count(r2); ++i) bad += dtf_ckpt_read(r2, i, b.data()) != 0;
    EXPECT(dtf_ckpt_count(r2) == 0 || bad > 0);
    dtf_ckpt_close(r2);
  }
  // bottleneck text codec
  float vals[5] = {0.0f, 1.5f, -2.25e-3f, 3.4e38f, 1e-30f}, back[8];
  char text[256];
  const int64_t len = dtf_format_floats(vals, 5, text, sizeof text);
  EXPECT(len > 0);
  EXPECT(dtf_parse_floats(text, (uint64_t)len, back, 8) == 5);
  for (int i = 0; i < 5; ++i) EXPECT(back[i] == vals[i]);
  EXPECT(dtf_format_floats(vals, 5, text, 8) < 0);  // too small a buffer is an error, not an overflow
}

// TensorFlow tensor bundle: enough entries for several restart runs, a truncated index, a flipped data byte
static void test_bundle(const std::string& dir) {
  const std::string prefix = dir + "/selftest_model.ckpt";
  void* w = dtf_bundle_begin(prefix.c_str());
  EXPECT(w != nullptr);
  std::vector<std::vector<float>> vals(40);
  for (int i = 0; i < 40; ++i) {
    vals[i].assign(17 + i, (float)i);
    char name[64];
    snprintf(name, sizeof name, "tower/layer_%02d/weights", i);
    const int64_t d[2] = {1, (int64_t)vals[i].size()};
    EXPECT(dtf_bundle_add(w, name, 1, 2, d, vals[i].data(), vals[i].size() * 4) == 0);
  }
  const int32_t step = 77;
  EXPECT(dtf_bundle_add(w, "Variable", 3, 0, nullptr, &step, 4) == 0);
  EXPECT(dtf_bundle_finish(w) == 0);
  void* r = dtf_bundle_open(prefix.c_str());
  EXPECT(r != nullptr);
  if (r) {
    EXPECT(dtf_bundle_count(r) == 41);
    char name[128];
    int dt = 0, nd = 0;
    int64_t dims[8];
    uint64_t nb = 0;
    EXPECT(dtf_bundle_entry(r, 0, name, sizeof name, &dt, &nd, dims, &nb) == 0);
    EXPECT(std::string(name) == "Variable" && dt == 3 && nd == 0 && nb == 4);  // sorted: upper case first
    int32_t got = 0;
    EXPECT(dtf_bundle_read(r, 0, &got) == 0 && got == 77);
    EXPECT(dtf_bundle_entry(r, 40, name, sizeof name, &dt, &nd, dims, &nb) == 0);
    EXPECT(std::string(name) == "tower/layer_39/weights" && nd == 2 && dims[1] == 56 && nb == 56 * 4);
    std::vector<float> b(56);
    EXPECT(dtf_bundle_read(r, 40, b.data()) == 0 && b[55] == 39.0f);
    EXPECT(dtf_bundle_entry(r, 41, name, sizeof name, &dt, &nd, dims, &nb) != 0);
    dtf_bundle_close(r);
  }
  // a flipped byte in the data file is caught by the tensor's CRC
  {
    FILE* f = fopen((prefix + ".data-00000-of-00001").c_str(), "r+b");
    EXPECT(f != nullptr);
    if (f) { fseek(f, 2, SEEK_SET); fputc(0x7f, f); fclose(f); }
    void* r2 = dtf_bundle_open(prefix.c_str());
    EXPECT(r2 != nullptr);
    if (r2) {
      int bad = 0;
      std::vector<float> b(64);
      for (int i = 0; i < dtf_bundle_count(r2); ++i) bad += dtf_bundle_read(r2, i, b.data()) != 0;
      EXPECT(bad == 1);
      dtf_bundle_close(r2);
    }
  }
  // a truncated index is refused
  {
    FILE* f = fopen((prefix + ".index").c_str(), "rb");
    std::vector<char> all(1 << 16);
    const size_t n = f ? fread(all.data(), 1, all.size(), f) : 0;
    if (f) fclose(f);
    FILE* g = fopen((prefix + "_cut.index").c_str(), "wb");
    if (g) { fwrite(all.data(), 1, n / 2, g); fclose(g); }
    EXPECT(dtf_bundle_open((prefix + "_cut").c_str()) == nullptr);
  }
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  test_crc();
  test_events(dir);
  test_store();
  test_ckpt(dir);
  test_bundle(dir);
  if (g_fail) {
    fprintf(stderr, "host selftest: %d check(s) failed\n", g_fail);
    return 1;
  }
  printf("host selftest ok\n");
  return 0;
}
