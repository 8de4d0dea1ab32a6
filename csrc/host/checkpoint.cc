// Native checkpoint container (component C25): a flat, CRC-protected tensor bundle.
//
// The reference delegates to TF's tensor-bundle writer (tf.train.Saver, demo1/train.py:144,165;
// Supervisor autosave demo2/train.py:166-172).  This container stores named tensors (the names are the
// reference's TF variable names, see utils/checkpoint.py) with per-tensor CRC32C, 64-byte aligned
// payloads (mmap/direct-read friendly) and an atomic tmp+fsync+rename commit so a crash during the
// timed autosave never leaves a torn "latest" checkpoint.
//
// layout: "DTFCKPT1" | u32 count | entries... ; entry = u32 name_len | name | u8 dtype | u8 ndim |
//         u64 dims[ndim] | u64 nbytes | u32 crc32c | pad to 64 | payload
#include <fcntl.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dtf_host.h"

namespace {
const char kMagic[8] = {'D', 'T', 'F', 'C', 'K', 'P', 'T', '1'};

struct Writer {
  FILE* f = nullptr;
  std::string path, tmp;
  uint32_t count = 0;
  bool failed = false;
  void put(const void* p, size_t n) {
    if (n && fwrite(p, 1, n, f) != n) failed = true;
  }
};

struct Entry {
  std::string name;
  uint8_t dtype = 0, ndim = 0;
  uint64_t dims[8] = {0};
  uint64_t nbytes = 0;
  uint32_t crc = 0;
  long offset = 0;
};

struct Reader {
  FILE* f = nullptr;
  std::vector<Entry> entries;
};
}  // namespace

DTF_HOST_API void* dtf_ckpt_begin(const char* path) {
  Writer* w = new Writer();
  w->path = path;
  w->tmp = w->path + ".tmp." + std::to_string((long)getpid());
  w->f = fopen(w->tmp.c_str(), "wb");
  if (!w->f) { delete w; return nullptr; }
  w->put(kMagic, 8);
  w->put(&w->count, 4);  // patched in finish
  return w;
}

DTF_HOST_API int dtf_ckpt_add(void* h, const char* name, int dtype, int ndim, const uint64_t* dims, const void* data,
                              uint64_t nbytes) {
  Writer* w = static_cast<Writer*>(h);
  if (ndim < 0 || ndim > 8) return -22;
  const uint32_t nl = (uint32_t)strlen(name);
  const uint8_t dt = (uint8_t)dtype, nd = (uint8_t)ndim;
  const uint32_t crc = dtf::crc32c_extend(0, data, (size_t)nbytes);
  w->put(&nl, 4);
  w->put(name, nl);
  w->put(&dt, 1);
  w->put(&nd, 1);
  w->put(dims, 8 * (size_t)ndim);
  w->put(&nbytes, 8);
  w->put(&crc, 4);
  const long pos = ftell(w->f);
  const long pad = (64 - (pos % 64)) % 64;
  static const char zeros[64] = {0};
  w->put(zeros, (size_t)pad);
  w->put(data, (size_t)nbytes);
  w->count++;
  return w->failed ? -5 : 0;
}

DTF_HOST_API int dtf_ckpt_finish(void* h) {
  Writer* w = static_cast<Writer*>(h);
  int rc = w->failed ? -5 : 0;
  if (fseek(w->f, 8, SEEK_SET) != 0) rc = -5;
  w->put(&w->count, 4);
  if (fflush(w->f) != 0) rc = -5;
  fsync(fileno(w->f));
  fclose(w->f);
  if (rc == 0 && !w->failed) {
    if (rename(w->tmp.c_str(), w->path.c_str()) != 0) rc = -5;
  } else {
    unlink(w->tmp.c_str());
    rc = -5;
  }
  delete w;
  return rc;
}

DTF_HOST_API void* dtf_ckpt_open(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return nullptr;
  char magic[8];
  uint32_t count = 0;
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kMagic, 8) != 0 || fread(&count, 4, 1, f) != 1) {
    fclose(f);
    return nullptr;
  }
  Reader* r = new Reader();
  r->f = f;
  for (uint32_t i = 0; i < count; ++i) {
    Entry e;
    uint32_t nl = 0;
    if (fread(&nl, 4, 1, f) != 1 || nl > (1u << 16)) goto bad;
    e.name.resize(nl);
    if (nl && fread(&e.name[0], 1, nl, f) != nl) goto bad;
    if (fread(&e.dtype, 1, 1, f) != 1 || fread(&e.ndim, 1, 1, f) != 1 || e.ndim > 8) goto bad;
    if (e.ndim && fread(e.dims, 8, e.ndim, f) != e.ndim) goto bad;
    if (fread(&e.nbytes, 8, 1, f) != 1 || fread(&e.crc, 4, 1, f) != 1) goto bad;
    {
      const long pos = ftell(f);
      e.offset = pos + (64 - (pos % 64)) % 64;
      if (fseek(f, e.offset + (long)e.nbytes, SEEK_SET) != 0) goto bad;
    }
    r->entries.push_back(e);
  }
  return r;
bad:
  fclose(f);
  delete r;
  return nullptr;
}

DTF_HOST_API int dtf_ckpt_count(void* h) { return (int)static_cast<Reader*>(h)->entries.size(); }

DTF_HOST_API int dtf_ckpt_entry(void* h, int i, char* name, int name_cap, int* dtype, int* ndim, uint64_t* dims,
                                uint64_t* nbytes) {
  Reader* r = static_cast<Reader*>(h);
  if (i < 0 || i >= (int)r->entries.size()) return -22;
  const Entry& e = r->entries[(size_t)i];
  snprintf(name, (size_t)name_cap, "%s", e.name.c_str());
  *dtype = e.dtype;
  *ndim = e.ndim;
  for (int d = 0; d < e.ndim; ++d) dims[d] = e.dims[d];
  *nbytes = e.nbytes;
  return 0;
}

// returns 0, -5 on I/O error, -74 on CRC mismatch (EBADMSG)
DTF_HOST_API int dtf_ckpt_read(void* h, int i, void* dst) {
  Reader* r = static_cast<Reader*>(h);
  if (i < 0 || i >= (int)r->entries.size()) return -22;
  const Entry& e = r->entries[(size_t)i];
  if (fseek(r->f, e.offset, SEEK_SET) != 0) return -5;
  if (e.nbytes && fread(dst, 1, (size_t)e.nbytes, r->f) != e.nbytes) return -5;
  if (dtf::crc32c_extend(0, dst, (size_t)e.nbytes) != e.crc) return -74;
  return 0;
}

DTF_HOST_API void dtf_ckpt_close(void* h) {
  Reader* r = static_cast<Reader*>(h);
  fclose(r->f);
  delete r;
}

// ---------------------------------------------------------------- bottleneck text codec (C32 compatibility)
// The reference caches each 2048-float bottleneck as comma-joined decimal text
// (retrain1/retrain.py:211-215,243-245).  Parsing 409 600 floats per training step in Python is the
// reference's real hot spot; these two helpers make the optional import/export path fast.
DTF_HOST_API int64_t dtf_parse_floats(const char* text, uint64_t len, float* out, uint64_t cap) {
  uint64_t n = 0;
  const char* p = text;
  const char* end = text + len;
  while (p < end) {
    while (p < end && (*p == ',' || *p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
    if (p >= end) break;
    char* q = nullptr;
    const float v = strtof(p, &q);
    if (q == p) return -1;  // invalid float -> caller recreates the bottleneck (retrain1/retrain.py:216-221)
    if (n < cap) out[n] = v;
    ++n;
    p = q;
  }
  return (int64_t)n;
}

DTF_HOST_API int64_t dtf_format_floats(const float* vals, uint64_t n, char* buf, uint64_t cap) {
  uint64_t pos = 0;
  for (uint64_t i = 0; i < n; ++i) {
    char tmp[32];
    const int k = snprintf(tmp, sizeof(tmp), i ? ",%.9g" : "%.9g", (double)vals[i]);
    if (pos + (uint64_t)k >= cap) return -1;
    memcpy(buf + pos, tmp, (size_t)k);
    pos += (uint64_t)k;
  }
  if (pos < cap) buf[pos] = '\0';
  return (int64_t)pos;
}
