// TensorFlow tensor-bundle checkpoints (component C25, SURVEY §5.4 stretch goal): the files `tf.train.Saver` writes and
// `saver.restore` reads (/root/reference/demo1/train.py:144,165; demo1/test.py:175-182; Supervisor autosave
// demo2/train.py:166-172):
//
//   <prefix>.index                 an SSTable (LevelDB table format): key "" -> BundleHeaderProto, then one key per variable
//                                  -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}
//   <prefix>.data-00000-of-00001   the tensors' raw little-endian bytes, back to back, in key order
//
// Written and parsed here without TensorFlow [ext: format from tensorflow/core/util/tensor_bundle and
// tensorflow/core/lib/io/{table_builder,format,block}.cc].  Table format: blocks of prefix-compressed entries
// (varint shared | varint non_shared | varint value_len | key suffix | value) + restart array + u32 restart count, each
// block followed by a 5-byte trailer (compression type 0, masked CRC32C of block + type); metaindex block, index block
// (last key of a data block -> BlockHandle), 48-byte footer ending in the magic 0xdb4775248b80fb57.
// The reader takes any valid uncompressed table (prefix compression, several blocks); the writer emits one restart
// per 16 entries and 256 KB blocks like TensorFlow's BundleWriter.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include <unistd.h>

#include "dtf_host.h"

namespace {
constexpr uint64_t kTableMagic = 0xdb4775248b80fb57ull;
constexpr size_t kBlockSize = 256 << 10;
constexpr int kRestartInterval = 16;

uint32_t mask_crc(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }

void put_varint(std::string& s, uint64_t v) {
  while (v >= 0x80) { s.push_back((char)((v & 0x7f) | 0x80)); v >>= 7; }
  s.push_back((char)v);
}
void put_fixed32(std::string& s, uint32_t v) { s.append(reinterpret_cast<const char*>(&v), 4); }
void put_fixed64(std::string& s, uint64_t v) { s.append(reinterpret_cast<const char*>(&v), 8); }

bool get_varint(const uint8_t*& p, const uint8_t* end, uint64_t& v) {
  v = 0;
  for (int shift = 0; shift <= 63 && p < end; shift += 7) {
    const uint8_t b = *p++;
    v |= (uint64_t)(b & 0x7f) << shift;
    if (!(b & 0x80)) return true;
  }
  return false;
}

struct BlockBuilder {
  std::string buf, last_key;
  std::vector<uint32_t> restarts{0};
  int counter = 0;
  bool empty() const { return buf.empty(); }
  void add(const std::string& key, const std::string& value) {
    size_t shared = 0;
    if (counter < kRestartInterval) {
      const size_t n = std::min(last_key.size(), key.size());
      while (shared < n && last_key[shared] == key[shared]) ++shared;
    } else {
      restarts.push_back((uint32_t)buf.size());
      counter = 0;
    }
    put_varint(buf, shared);
    put_varint(buf, key.size() - shared);
    put_varint(buf, value.size());
    buf.append(key, shared, std::string::npos);
    buf.append(value);
    last_key = key;
    ++counter;
  }
  std::string finish() {
    for (uint32_t r : restarts) put_fixed32(buf, r);
    put_fixed32(buf, (uint32_t)restarts.size());
    return buf;
  }
};

// appends block + trailer to `file`; returns the BlockHandle encoding (offset, size of the block without its trailer)
std::string write_block(std::string& file, const std::string& block) {
  std::string handle;
  put_varint(handle, file.size());
  put_varint(handle, block.size());
  file.append(block);
  const char type = 0;  // no compression
  uint32_t crc = dtf::crc32c_extend(0, block.data(), block.size());
  crc = dtf::crc32c_extend(crc, &type, 1);
  file.push_back(type);
  put_fixed32(file, mask_crc(crc));
  return handle;
}

struct BEntry {
  int dtype = 0;
  std::vector<int64_t> dims;
  int64_t offset = 0, size = 0;
  uint32_t crc = 0;
};

std::string encode_entry(const BEntry& e) {
  dtf::PbWriter shape;
  for (int64_t d : e.dims) {
    dtf::PbWriter dim;
    dim.f_varint(1, (uint64_t)d);
    shape.f_bytes(2, dim.buf.data(), dim.buf.size());
  }
  dtf::PbWriter w;
  w.f_varint(1, (uint64_t)e.dtype);
  w.f_bytes(2, shape.buf.data(), shape.buf.size());
  // shard_id 0 is the default and omitted
  if (e.offset) w.f_varint(4, (uint64_t)e.offset);
  w.f_varint(5, (uint64_t)e.size);
  w.tag(6, 5);
  w.buf.append(reinterpret_cast<const char*>(&e.crc), 4);
  return w.buf;
}

bool decode_entry(const uint8_t* p, const uint8_t* end, BEntry& e) {
  while (p < end) {
    uint64_t key;
    if (!get_varint(p, end, key)) return false;
    const int field = (int)(key >> 3), wire = (int)(key & 7);
    if (wire == 0) {
      uint64_t v;
      if (!get_varint(p, end, v)) return false;
      if (field == 1) e.dtype = (int)v;
      else if (field == 4) e.offset = (int64_t)v;
      else if (field == 5) e.size = (int64_t)v;
    } else if (wire == 2) {
      uint64_t n;
      if (!get_varint(p, end, n) || p + n > end) return false;
      if (field == 2) {  // TensorShapeProto
        const uint8_t* q = p;
        const uint8_t* qe = p + n;
        while (q < qe) {
          uint64_t k2, n2;
          if (!get_varint(q, qe, k2)) return false;
          if ((k2 & 7) == 2) {
            if (!get_varint(q, qe, n2) || q + n2 > qe) return false;
            if ((k2 >> 3) == 2) {  // Dim
              const uint8_t* r = q;
              const uint8_t* re = q + n2;
              int64_t size = 0;
              while (r < re) {
                uint64_t k3, v3;
                if (!get_varint(r, re, k3)) return false;
                if ((k3 & 7) == 0) { if (!get_varint(r, re, v3)) return false; if ((k3 >> 3) == 1) size = (int64_t)v3; }
                else if ((k3 & 7) == 2) { if (!get_varint(r, re, v3) || r + v3 > re) return false; r += v3; }
                else return false;
              }
              e.dims.push_back(size);
            }
            q += n2;
          } else if ((k2 & 7) == 0) {
            if (!get_varint(q, qe, n2)) return false;
          } else return false;
        }
      }
      p += n;
    } else if (wire == 5) {
      if (p + 4 > end) return false;
      if (field == 6) memcpy(&e.crc, p, 4);
      p += 4;
    } else if (wire == 1) {
      if (p + 8 > end) return false;
      p += 8;
    } else return false;
  }
  return true;
}

struct BundleWriter {
  std::string prefix;
  std::map<std::string, std::pair<BEntry, std::string>> items;  // sorted by key; value = (entry, bytes)
};

struct BundleReader {
  std::string prefix;
  std::vector<std::pair<std::string, BEntry>> entries;
  FILE* data = nullptr;
  int num_shards = 1;
};

bool read_file(const std::string& path, std::string& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize((size_t)n);
  const bool ok = n == 0 || fread(&out[0], 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok;
}

// entries of one table block (verifies the trailer CRC); false on corruption
bool parse_block(const std::string& file, uint64_t off, uint64_t size, std::vector<std::pair<std::string, std::string>>& out) {
  if (off + size + 5 > file.size() || size < 4) return false;
  const uint8_t* base = reinterpret_cast<const uint8_t*>(file.data()) + off;
  if (base[size] != 0) return false;  // compressed blocks are not written by TensorFlow's bundle writer
  uint32_t want;
  memcpy(&want, base + size + 1, 4);
  if (mask_crc(dtf::crc32c_extend(0, base, size + 1)) != want) return false;
  uint32_t nrestart;
  memcpy(&nrestart, base + size - 4, 4);
  if ((uint64_t)nrestart * 4 + 4 > size) return false;
  const uint8_t* p = base;
  const uint8_t* end = base + size - 4 - (uint64_t)nrestart * 4;
  std::string key;
  while (p < end) {
    uint64_t shared, non_shared, vlen;
    if (!get_varint(p, end, shared) || !get_varint(p, end, non_shared) || !get_varint(p, end, vlen)) return false;
    if (shared > key.size() || p + non_shared + vlen > end) return false;
    key.resize(shared);
    key.append(reinterpret_cast<const char*>(p), non_shared);
    p += non_shared;
    out.emplace_back(key, std::string(reinterpret_cast<const char*>(p), vlen));
    p += vlen;
  }
  return true;
}
}  // namespace

DTF_HOST_API void* dtf_bundle_begin(const char* prefix) {
  BundleWriter* w = new BundleWriter();
  w->prefix = prefix;
  return w;
}

// dtype: TensorFlow DataType (DT_FLOAT 1, DT_INT32 3, DT_INT64 9, ...); data is copied
DTF_HOST_API int dtf_bundle_add(void* h, const char* name, int dtype, int ndim, const int64_t* dims, const void* data,
                                uint64_t nbytes) {
  BundleWriter* w = static_cast<BundleWriter*>(h);
  if (!w || !name || !*name || ndim < 0 || ndim > 8 || (nbytes && !data)) return -22;
  BEntry e;
  e.dtype = dtype;
  e.dims.assign(dims, dims + ndim);
  e.size = (int64_t)nbytes;
  e.crc = mask_crc(dtf::crc32c_extend(0, data, nbytes));
  w->items[name] = std::make_pair(e, std::string(static_cast<const char*>(data), nbytes));
  return 0;
}

// writes <prefix>.data-00000-of-00001 and <prefix>.index (tmp + rename: never a torn checkpoint); frees the handle
DTF_HOST_API int dtf_bundle_finish(void* h) {
  BundleWriter* w = static_cast<BundleWriter*>(h);
  if (!w) return -22;
  int rc = 0;
  const std::string data_path = w->prefix + ".data-00000-of-00001", index_path = w->prefix + ".index";
  const std::string suffix = ".tmp." + std::to_string((long)getpid());
  {
    FILE* f = fopen((data_path + suffix).c_str(), "wb");
    if (!f) { delete w; return -1; }
    int64_t off = 0;
    for (auto& kv : w->items) {
      kv.second.first.offset = off;
      const std::string& bytes = kv.second.second;
      if (!bytes.empty() && fwrite(bytes.data(), 1, bytes.size(), f) != bytes.size()) rc = -1;
      off += (int64_t)bytes.size();
    }
    if (fflush(f) != 0 || fsync(fileno(f)) != 0) rc = -1;
    fclose(f);
  }
  std::string file;
  std::vector<std::pair<std::string, std::string>> index;  // last key of each data block -> handle
  BlockBuilder blk;
  auto flush_block = [&]() {
    if (blk.empty()) return;
    const std::string last = blk.last_key;
    index.emplace_back(last, write_block(file, blk.finish()));
    blk = BlockBuilder();
  };
  {
    dtf::PbWriter ver;
    ver.f_varint(1, 1);  // VersionDef.producer = kTensorBundleVersion
    dtf::PbWriter hdr;
    hdr.f_varint(1, 1);  // num_shards (endianness LITTLE = 0 is the default)
    hdr.f_bytes(3, ver.buf.data(), ver.buf.size());
    blk.add("", hdr.buf);
  }
  for (auto& kv : w->items) {
    blk.add(kv.first, encode_entry(kv.second.first));
    if (blk.buf.size() >= kBlockSize) flush_block();
  }
  flush_block();
  BlockBuilder meta;
  const std::string meta_handle = write_block(file, meta.finish());
  BlockBuilder idx;
  for (auto& kv : index) idx.add(kv.first, kv.second);
  const std::string idx_handle = write_block(file, idx.finish());
  std::string footer = meta_handle + idx_handle;
  footer.resize(40, '\0');
  put_fixed64(footer, kTableMagic);
  file.append(footer);
  {
    FILE* f = fopen((index_path + suffix).c_str(), "wb");
    if (!f) rc = -1;
    else {
      if (fwrite(file.data(), 1, file.size(), f) != file.size() || fflush(f) != 0 || fsync(fileno(f)) != 0) rc = -1;
      fclose(f);
    }
  }
  if (rc == 0 && (rename((data_path + suffix).c_str(), data_path.c_str()) != 0 ||
                  rename((index_path + suffix).c_str(), index_path.c_str()) != 0))
    rc = -1;
  delete w;
  return rc;
}

DTF_HOST_API void* dtf_bundle_open(const char* prefix) {
  std::string file;
  if (!read_file(std::string(prefix) + ".index", file) || file.size() < 48) return nullptr;
  uint64_t magic;
  memcpy(&magic, file.data() + file.size() - 8, 8);
  if (magic != kTableMagic) return nullptr;
  const uint8_t* p = reinterpret_cast<const uint8_t*>(file.data()) + file.size() - 48;
  const uint8_t* end = p + 40;
  uint64_t moff, msize, ioff, isize;
  if (!get_varint(p, end, moff) || !get_varint(p, end, msize) || !get_varint(p, end, ioff) || !get_varint(p, end, isize))
    return nullptr;
  std::vector<std::pair<std::string, std::string>> idx;
  if (!parse_block(file, ioff, isize, idx)) return nullptr;
  BundleReader* r = new BundleReader();
  r->prefix = prefix;
  for (auto& kv : idx) {
    const uint8_t* hp = reinterpret_cast<const uint8_t*>(kv.second.data());
    const uint8_t* he = hp + kv.second.size();
    uint64_t boff, bsize;
    std::vector<std::pair<std::string, std::string>> items;
    if (!get_varint(hp, he, boff) || !get_varint(hp, he, bsize) || !parse_block(file, boff, bsize, items)) { delete r; return nullptr; }
    for (auto& it : items) {
      if (it.first.empty()) {  // BundleHeaderProto: num_shards = field 1
        const uint8_t* q = reinterpret_cast<const uint8_t*>(it.second.data());
        const uint8_t* qe = q + it.second.size();
        uint64_t k, v;
        if (q < qe && get_varint(q, qe, k) && (k >> 3) == 1 && (k & 7) == 0 && get_varint(q, qe, v)) r->num_shards = (int)v;
        continue;
      }
      BEntry e;
      if (!decode_entry(reinterpret_cast<const uint8_t*>(it.second.data()),
                        reinterpret_cast<const uint8_t*>(it.second.data()) + it.second.size(), e)) { delete r; return nullptr; }
      r->entries.emplace_back(it.first, e);
    }
  }
  char shard[64];
  snprintf(shard, sizeof shard, ".data-00000-of-%05d", r->num_shards);
  r->data = fopen((r->prefix + shard).c_str(), "rb");
  if (!r->data || r->num_shards != 1) {  // (several shards: written by a distributed TF saver; not produced here)
    if (r->data) fclose(r->data);
    delete r;
    return nullptr;
  }
  return r;
}

DTF_HOST_API int dtf_bundle_count(void* h) { return h ? (int)static_cast<BundleReader*>(h)->entries.size() : -1; }

// name_cap bytes for the name (NUL-terminated); dims[8]; returns 0, or -22
DTF_HOST_API int dtf_bundle_entry(void* h, int i, char* name, int name_cap, int* dtype, int* ndim, int64_t* dims, uint64_t* nbytes) {
  BundleReader* r = static_cast<BundleReader*>(h);
  if (!r || i < 0 || i >= (int)r->entries.size()) return -22;
  const auto& e = r->entries[i];
  snprintf(name, (size_t)name_cap, "%s", e.first.c_str());
  *dtype = e.second.dtype;
  *ndim = (int)std::min<size_t>(8, e.second.dims.size());
  for (int d = 0; d < *ndim; ++d) dims[d] = e.second.dims[d];
  *nbytes = (uint64_t)e.second.size;
  return 0;
}

// reads tensor i into buf (nbytes of dtf_bundle_entry) and verifies its CRC32C; 0, -22, -5 (I/O) or -74 (bad CRC)
DTF_HOST_API int dtf_bundle_read(void* h, int i, void* buf) {
  BundleReader* r = static_cast<BundleReader*>(h);
  if (!r || i < 0 || i >= (int)r->entries.size()) return -22;
  const BEntry& e = r->entries[i].second;
  if (fseek(r->data, (long)e.offset, SEEK_SET) != 0) return -5;
  if (e.size && fread(buf, 1, (size_t)e.size, r->data) != (size_t)e.size) return -5;
  return mask_crc(dtf::crc32c_extend(0, buf, (size_t)e.size)) == e.crc ? 0 : -74;
}

DTF_HOST_API void dtf_bundle_close(void* h) {
  BundleReader* r = static_cast<BundleReader*>(h);
  if (!r) return;
  if (r->data) fclose(r->data);
  delete r;
}
