// TensorBoard event-file writer (component C24): TFRecord framing + hand-encoded Event/Summary
// protobufs (the `tensorboard`/`tensorflow` packages are not available, and the reference's
// FileWriter is TF-native C++ anyway: demo1/train.py:151,157; retrain1/retrain.py:420-446).
//
// record  := uint64 length | uint32 masked_crc32c(length) | bytes data | uint32 masked_crc32c(data)
// Event   := { double wall_time = 1; int64 step = 2; string file_version = 3; bytes graph_def = 4;
//              Summary summary = 5; }
// Summary := { repeated Value value = 1 };  Value := { string tag = 1; float simple_value = 2;
//              HistogramProto histo = 5; }
// HistogramProto := { double min=1, max=2, num=3, sum=4, sum_squares=5;
//                     repeated double bucket_limit=6 [packed]; repeated double bucket=7 [packed]; }
//
// Records are appended to an in-memory queue and written by a background thread (the training
// loop never blocks on the filesystem; the reference synchronously writes every step, quirk Q5).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

#include "dtf_host.h"

namespace {
struct EventsWriter {
  FILE* f = nullptr;
  std::mutex mu;
  std::mutex io;  // serialises swap+write so concurrent drains keep record order
  std::condition_variable cv;
  std::string pending;      // framed records waiting for the writer thread
  std::string cur_summary;  // Summary being assembled between begin/end
  double cur_wall = 0;
  int64_t cur_step = 0;
  bool stop = false;
  std::atomic<uint64_t> records{0};
  std::thread th;

  void frame(const std::string& data) {
    char hdr[12];
    const uint64_t len = data.size();
    memcpy(hdr, &len, 8);
    const uint32_t c1 = dtf::masked_crc32c(hdr, 8);
    memcpy(hdr + 8, &c1, 4);
    const uint32_t c2 = dtf::masked_crc32c(data.data(), data.size());
    std::lock_guard<std::mutex> l(mu);
    pending.append(hdr, 12);
    pending.append(data);
    pending.append(reinterpret_cast<const char*>(&c2), 4);
    records++;
    if (pending.size() > (1u << 20)) cv.notify_one();
  }
  void drain(bool sync) {
    std::lock_guard<std::mutex> g(io);
    std::string out;
    {
      std::lock_guard<std::mutex> l(mu);
      out.swap(pending);
    }
    if (!out.empty()) fwrite(out.data(), 1, out.size(), f);
    if (sync) fflush(f);
  }
  void loop() {
    std::unique_lock<std::mutex> l(mu);
    while (!stop) {
      dtf::cv_wait_ms(cv, l, 2000);  // steady clock (system clock only under TSan, dtf_host.h)
      l.unlock();
      drain(true);
      l.lock();
    }
  }
};

double now_sec() {
  using namespace std::chrono;
  return duration_cast<duration<double>>(system_clock::now().time_since_epoch()).count();
}
}  // namespace

DTF_HOST_API void* dtf_events_open(const char* path) {
  FILE* f = fopen(path, "wb");
  if (!f) return nullptr;
  EventsWriter* w = new EventsWriter();
  w->f = f;
  dtf::PbWriter ev;
  ev.f_double(1, now_sec());
  ev.f_string(3, "brain.Event:2");
  w->frame(ev.buf);
  w->th = std::thread([w] { w->loop(); });
  return w;
}

DTF_HOST_API void dtf_events_begin(void* h, int64_t step, double wall_time) {
  EventsWriter* w = static_cast<EventsWriter*>(h);
  w->cur_summary.clear();
  w->cur_step = step;
  w->cur_wall = wall_time > 0 ? wall_time : now_sec();
}

DTF_HOST_API void dtf_events_scalar(void* h, const char* tag, float value) {
  EventsWriter* w = static_cast<EventsWriter*>(h);
  dtf::PbWriter v;
  v.f_string(1, tag);
  v.f_float(2, value);
  dtf::PbWriter s;
  s.f_bytes(1, v.buf.data(), v.buf.size());
  w->cur_summary += s.buf;
}

DTF_HOST_API void dtf_events_histogram(void* h, const char* tag, double mn, double mx, double num, double sum,
                                       double sum_squares, int nbuckets, const double* limits, const double* counts) {
  EventsWriter* w = static_cast<EventsWriter*>(h);
  dtf::PbWriter hp;
  hp.f_double(1, mn);
  hp.f_double(2, mx);
  hp.f_double(3, num);
  hp.f_double(4, sum);
  hp.f_double(5, sum_squares);
  hp.f_packed_doubles(6, limits, (size_t)nbuckets);
  hp.f_packed_doubles(7, counts, (size_t)nbuckets);
  dtf::PbWriter v;
  v.f_string(1, tag);
  v.f_bytes(5, hp.buf.data(), hp.buf.size());
  dtf::PbWriter s;
  s.f_bytes(1, v.buf.data(), v.buf.size());
  w->cur_summary += s.buf;
}

DTF_HOST_API void dtf_events_end(void* h) {
  EventsWriter* w = static_cast<EventsWriter*>(h);
  dtf::PbWriter ev;
  ev.f_double(1, w->cur_wall);
  ev.f_varint(2, (uint64_t)w->cur_step);
  ev.f_bytes(5, w->cur_summary.data(), w->cur_summary.size());
  w->frame(ev.buf);
  w->cur_summary.clear();
}

// opaque payload for the graph tab (we store a JSON description of the kernel sequence instead of a GraphDef)
DTF_HOST_API void dtf_events_graph(void* h, const void* data, uint64_t n) {
  EventsWriter* w = static_cast<EventsWriter*>(h);
  dtf::PbWriter ev;
  ev.f_double(1, now_sec());
  ev.f_bytes(4, data, (size_t)n);
  w->frame(ev.buf);
}

DTF_HOST_API void dtf_events_flush(void* h) { static_cast<EventsWriter*>(h)->drain(true); }

DTF_HOST_API uint64_t dtf_events_num_records(void* h) { return static_cast<EventsWriter*>(h)->records.load(); }

DTF_HOST_API void dtf_events_close(void* h) {
  EventsWriter* w = static_cast<EventsWriter*>(h);
  {
    std::lock_guard<std::mutex> l(w->mu);
    w->stop = true;
  }
  w->cv.notify_all();
  if (w->th.joinable()) w->th.join();
  w->drain(true);
  fclose(w->f);
  delete w;
}
