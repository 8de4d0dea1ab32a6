// MFMA (v_mfma_f32_32x32x16_bf16) GEMM / implicit-GEMM core for gfx950.
//
//   C[M,N] (+)= A[M,K] * B[N,K]^T      fp32 accumulate, bf16 MFMA operands
//
// One templated kernel; the operand *loaders* decide where a [rows x BK] K-slab comes from
// (row-major, column-major with an in-register transpose, NHWC conv gather, conv-weight views)
// and the *epilogues* decide what happens to the 32x32 accumulator fragments (split-K slabs,
// fp32 atomics, bias+ReLU bf16 store into a concat slice, bias+ReLU+2x2 max-pool, masking).
// fp32 sources are converted to bf16 while staging, so the fp32 master weights are read
// directly and no bf16 shadow copy of the parameters has to be maintained by the optimizer.
//
// Tiling: workgroup = WM x WN waves (64 lanes each); each wave owns TM x TN MFMA tiles of 32x32.
// LDS tiles are [rows][BK] bf16 with a (BK+8)*2-byte row stride: 16-byte aligned for ds_read_b128
// and conflict-free for the 16-lane groups of that instruction (row stride = 4 dwords mod 64 for
// BK = 128/256, 36 for BK = 64, 20 for BK = 32 -> rows hit distinct 4-bank slots).
//
// Latency structure (these GEMMs are small: one wave of workgroups, so kernel time == one
// workgroup's dependent chain): every global load of a K-slab is issued back to back from clamped
// addresses (no predicated loads: hipcc would branch around each and drain vmcnt per element,
// cdna_hip_programming.md §5 trap (c)), raw data stays in registers while the MFMAs of the previous
// slab run, and conversion/masking happens at LDS-store time.  BK is a template knob: a larger slab
// means fewer barrier + round-trip episodes on the chain.
//
// Fragment maps (cdna_hip_programming.md §3): lane l, r=l&31, h=l>>5:
//   A frag = A[row r][k = 8h..8h+7], B frag = B[k = 8h..8h+7][col r],
//   C/D reg i -> row (i&3) + 8*(i>>2) + 4h, col r.
#pragma once
#include "common.h"

struct ConvGeom {
  int H, W, Cin, cstride;  // gathered NHWC tensor: spatial dims, channels taken per tap, channel stride
  int Ho, Wo, KH, KW, stride, pad_h, pad_w;
};

struct GemmParams {
  const void* A;
  const void* B;
  void* C;
  int M, N, K;
  long lda, ldb, ldc;
  int kchunk;        // K elements per blockIdx.z slice (multiple of the kernel's BK)
  long slab_stride;  // elements between split-K slabs
  const float* bias;
  int relu;
  const void* aux_in;
  void* aux_out;
  ConvGeom g;
  int accumulate;  // epilogues that support it add to the existing output instead of overwriting
  int kseg;        // segmented-K loaders (f32_fc.hip): A / B are tables of operand blocks of kseg k-rows each; 0 = plain
  int ksegp;       // > 0: the k index space gives every block ksegp >= kseg places (kseg padded to the slab depth; the rows
                   // past kseg read as zero), so a slab never straddles two blocks; 0 = kseg
  // double-buffered B operand (the "wshard" data-parallel schedule keeps W_fc1 in two buffers that alternate by step parity,
  // engine.hip: MnistEngine::wshard): kernels that honour it read B_alt instead of B when *bsel is odd; null = plain
  const int* bsel;
  const void* B_alt;
};

// ------------------------------------------------------------------ raw element loads (convert later)
template <typename T>
struct Raw8;  // 8 consecutive elements
template <>
struct Raw8<uint16_t> {  // bf16 bits
  uint4 v;
  __device__ __forceinline__ void ld(const uint16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ uint4 bf() const { return v; }
};
template <>
struct Raw8<float> {
  float4 a, b;
  __device__ __forceinline__ void ld(const float* p) {
    a = reinterpret_cast<const float4*>(p)[0];
    b = reinterpret_cast<const float4*>(p)[1];
  }
  __device__ __forceinline__ uint4 bf() const {
    return make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
  }
};
template <typename T>
struct Raw2;  // two adjacent elements
template <>
struct Raw2<uint16_t> {
  uint32_t v;
  __device__ __forceinline__ void ld(const uint16_t* p) { v = *reinterpret_cast<const uint32_t*>(p); }
  __device__ __forceinline__ uint32_t e0() const { return v & 0xffffu; }
  __device__ __forceinline__ uint32_t e1() const { return v >> 16; }
};
template <>
struct Raw2<float> {
  float2 v;
  __device__ __forceinline__ void ld(const float* p) { v = *reinterpret_cast<const float2*>(p); }
  __device__ __forceinline__ uint32_t e0() const { return f2bf(v.x); }
  __device__ __forceinline__ uint32_t e1() const { return f2bf(v.y); }
};

// ------------------------------------------------------------------ loaders
// Every loader: init(p, r0, tid) / load(k0, kend, tid) -> raw registers / store(lds, tid) -> LDS [R][BK+8].
#define DTF_LDS_STRIDE (BK + 8)

// src[row*ld + k]   (k contiguous); needs ld % 8 == 0 (bf16) or % 4 == 0 (fp32), K % 8 == 0
template <typename T, int R, int NT, int BK, bool IS_A>
struct LoadRowK {
  static constexpr int CPR = BK / 8, NCHUNK = R * CPR, PER = (NCHUNK + NT - 1) / NT;
  Raw8<T> regs[PER];
  const T* src;
  long ld;
  int rows, r0, k0, kend;
  __device__ __forceinline__ void init(const GemmParams& p, int r0_, int) {
    src = reinterpret_cast<const T*>(IS_A ? p.A : p.B);
    ld = IS_A ? p.lda : p.ldb;
    rows = IS_A ? p.M : p.N;
    r0 = r0_;
  }
  __device__ __forceinline__ bool ok(int c) const {
    return (NCHUNK % NT == 0 || c < NCHUNK) && (r0 + c / CPR) < rows && (k0 + (c % CPR) * 8) < kend;
  }
  __device__ __forceinline__ void load(int k0_, int kend_, int tid) {
    k0 = k0_;
    kend = kend_;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + NT * i;
      regs[i].ld(src + (ok(c) ? (long)(r0 + c / CPR) * ld + k0 + (c % CPR) * 8 : 0L));
    }
  }
  __device__ __forceinline__ void store(uint16_t* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + NT * i;
      if (NCHUNK % NT == 0 || c < NCHUNK)
        *reinterpret_cast<uint4*>(lds + (c / CPR) * DTF_LDS_STRIDE + (c % CPR) * 8) = ok(c) ? regs[i].bf() : make_uint4(0, 0, 0, 0);
    }
  }
};

// src[k*ld + row]   (row contiguous): transposed in registers while staging.  rows, ld even.
template <typename T, int R, int NT, int BK, bool IS_A>
struct LoadColK {
  static constexpr int CPR = BK / 8, NTASK = (R / 2) * CPR, PER = (NTASK + NT - 1) / NT;
  Raw2<T> regs[PER][8];
  const T* src;
  long ld;
  int rows, r0, k0, kend;
  __device__ __forceinline__ void init(const GemmParams& p, int r0_, int) {
    src = reinterpret_cast<const T*>(IS_A ? p.A : p.B);
    ld = IS_A ? p.lda : p.ldb;
    rows = IS_A ? p.M : p.N;
    r0 = r0_;
  }
  __device__ __forceinline__ void load(int k0_, int kend_, int tid) {
    k0 = k0_;
    kend = kend_;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int t = tid + NT * i;
      const int rp = t % (R / 2), kc = t / (R / 2);
      const int row = r0 + 2 * rp;
      const bool okr = (NTASK % NT == 0 || t < NTASK) && row < rows;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + kc * 8 + j;
        regs[i][j].ld(src + ((okr && k < kend) ? (long)k * ld + row : 0L));
      }
    }
  }
  __device__ __forceinline__ void store(uint16_t* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int t = tid + NT * i;
      if (NTASK % NT == 0 || t < NTASK) {
        const int rp = t % (R / 2), kc = t / (R / 2);
        const bool okr = (r0 + 2 * rp) < rows;
        uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool ok = okr && (k0 + kc * 8 + j) < kend;
          lo[j >> 1] |= (ok ? regs[i][j].e0() : 0u) << ((j & 1) * 16);
          hi[j >> 1] |= (ok ? regs[i][j].e1() : 0u) << ((j & 1) * 16);
        }
        *reinterpret_cast<uint4*>(lds + (2 * rp) * DTF_LDS_STRIDE + kc * 8) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        *reinterpret_cast<uint4*>(lds + (2 * rp + 1) * DTF_LDS_STRIDE + kc * 8) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      }
    }
  }
};

// Implicit-GEMM activation gather: row m = output position, k = (tap, channel) of a bf16 NHWC tensor.
// ORDER 0: m = (b, ho, wo).  ORDER 1: m = 4*window + sub over 2x2 pooling windows, so the four
// members of a window land in one lane's consecutive accumulator registers (fused max-pool).
template <int R, int NT, int BK, int ORDER>
struct LoadConvA {
  static constexpr int CPR = BK / 8, NCHUNK = R * CPR, PER = (NCHUNK + NT - 1) / NT;
  static_assert(NT % CPR == 0, "every thread keeps one k-chunk column");
  uint4 regs[PER];
  unsigned okbits;
  const uint16_t* src;
  int img[PER], hi0[PER], wi0[PER];
  bool okr[PER];
  ConvGeom g;
  __device__ __forceinline__ void init(const GemmParams& p, int m0, int tid) {
    src = reinterpret_cast<const uint16_t*>(p.A);
    g = p.g;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + NT * i;
      const int m = m0 + c / CPR;
      okr[i] = (NCHUNK % NT == 0 || c < NCHUNK) && m < p.M;
      int b, ho, wo;
      if (ORDER == 0) {
        wo = m % g.Wo;
        const int t = m / g.Wo;
        ho = t % g.Ho;
        b = t / g.Ho;
      } else {
        const int win = m >> 2, sub = m & 3;
        const int Wp = g.Wo >> 1, Hp = g.Ho >> 1;
        const int px = win % Wp, t = win / Wp;
        const int py = t % Hp;
        b = t / Hp;
        ho = 2 * py + (sub >> 1);
        wo = 2 * px + (sub & 1);
      }
      img[i] = b;
      hi0[i] = ho * g.stride - g.pad_h;
      wi0[i] = wo * g.stride - g.pad_w;
    }
  }
  __device__ __forceinline__ void load(int k0, int kend, int tid) {
    const int kk = k0 + (tid % CPR) * 8;
    const int tap = kk / g.Cin, ci = kk - tap * g.Cin;
    const int dy = tap / g.KW, dx = tap - dy * g.KW;
    okbits = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int hi = hi0[i] + dy, wi = wi0[i] + dx;
      const bool ok = okr[i] && kk < kend && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
      okbits |= (ok ? 1u : 0u) << i;
      regs[i] = *reinterpret_cast<const uint4*>(src + (ok ? ((long)(img[i] * g.H + hi) * g.W + wi) * g.cstride + ci : 0L));
    }
  }
  __device__ __forceinline__ void store(uint16_t* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + NT * i;
      if (NCHUNK % NT == 0 || c < NCHUNK)
        *reinterpret_cast<uint4*>(lds + (c / CPR) * DTF_LDS_STRIDE + (c % CPR) * 8) =
            ((okbits >> i) & 1u) ? regs[i] : make_uint4(0, 0, 0, 0);
    }
  }
};

// Backward-data weight view of an fp32 HWIO filter W[tap][n=cin][kc=cout]:
//   B[n][k = (tap', cout)] = W[(KH*KW-1) - tap'][n][cout]      (taps flipped, k contiguous)
// g.Cin holds the channels per tap of the gathered tensor (= cout of the forward conv).
template <int R, int NT, int BK, bool FLIP = true>
struct LoadConvWBwd {
  static constexpr int CPR = BK / 8, NCHUNK = R * CPR, PER = (NCHUNK + NT - 1) / NT;
  Raw8<float> regs[PER];
  const float* src;
  int rows, r0, kc_per_tap, ntaps, k0, kend;
  __device__ __forceinline__ void init(const GemmParams& p, int r0_, int) {
    src = reinterpret_cast<const float*>(p.B);
    rows = p.N;
    r0 = r0_;
    kc_per_tap = p.g.Cin;
    ntaps = p.g.KH * p.g.KW;
  }
  __device__ __forceinline__ bool ok(int c) const {
    return (NCHUNK % NT == 0 || c < NCHUNK) && (r0 + c / CPR) < rows && (k0 + (c % CPR) * 8) < kend;
  }
  __device__ __forceinline__ void load(int k0_, int kend_, int tid) {
    k0 = k0_;
    kend = kend_;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + NT * i;
      const int row = c / CPR, kk = k0 + (c % CPR) * 8;
      const int tapp = kk / kc_per_tap, co = kk - tapp * kc_per_tap;
      const int tap = FLIP ? ntaps - 1 - tapp : tapp;
      regs[i].ld(src + (ok(c) ? ((long)tap * rows + (r0 + row)) * kc_per_tap + co : 0L));
    }
  }
  __device__ __forceinline__ void store(uint16_t* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + NT * i;
      if (NCHUNK % NT == 0 || c < NCHUNK)
        *reinterpret_cast<uint4*>(lds + (c / CPR) * DTF_LDS_STRIDE + (c % CPR) * 8) = ok(c) ? regs[i].bf() : make_uint4(0, 0, 0, 0);
    }
  }
};

// Backward-data gather for a general (stride 1 or 2, any padding) conv: row m = input position (b,h,w),
// k = (tap, cout) over the output-gradient tensor dz[B,Ho,Wo,Cout] (bf16, channel stride cstride):
//   A[m][(tap,co)] = dz[b, (h + pad_h - dy)/s, (w + pad_w - dx)/s, co]   when divisible and in range, else 0
// pairs with LoadConvWBwd<.., FLIP=false> (same tap order).  g.H/g.W = input dims, g.Ho/g.Wo = dz dims,
// g.Cin = channels per tap (= forward Cout).
template <int R, int NT, int BK>
struct LoadConvBwdA {
  static constexpr int CPR = BK / 8, NCHUNK = R * CPR, PER = (NCHUNK + NT - 1) / NT;
  static_assert(NT % CPR == 0, "every thread keeps one k-chunk column");
  uint4 regs[PER];
  unsigned okbits;
  const uint16_t* src;
  int img[PER], hb[PER], wb[PER];
  bool okr[PER];
  ConvGeom g;
  int sh;  // log2(stride)
  __device__ __forceinline__ void init(const GemmParams& p, int m0, int tid) {
    src = reinterpret_cast<const uint16_t*>(p.A);
    g = p.g;
    sh = g.stride - 1;  // stride in {1, 2}
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + NT * i;
      const int m = m0 + c / CPR;
      okr[i] = (NCHUNK % NT == 0 || c < NCHUNK) && m < p.M;
      const int w = m % g.W, t = m / g.W;
      img[i] = t / g.H;
      hb[i] = t % g.H + g.pad_h;
      wb[i] = w + g.pad_w;
    }
  }
  __device__ __forceinline__ void load(int k0, int kend, int tid) {
    const int kk = k0 + (tid % CPR) * 8;
    const int tap = kk / g.Cin, co = kk - tap * g.Cin;
    const int dy = tap / g.KW, dx = tap - dy * g.KW;
    const int smask = g.stride - 1;
    okbits = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int hz = hb[i] - dy, wz = wb[i] - dx;
      const int ho = hz >> sh, wo = wz >> sh;
      const bool ok = okr[i] && kk < kend && hz >= 0 && wz >= 0 && ((hz | wz) & smask) == 0 && ho < g.Ho && wo < g.Wo;
      okbits |= (ok ? 1u : 0u) << i;
      regs[i] = *reinterpret_cast<const uint4*>(src + (ok ? ((long)(img[i] * g.Ho + ho) * g.Wo + wo) * g.cstride + co : 0L));
    }
  }
  __device__ __forceinline__ void store(uint16_t* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + NT * i;
      if (NCHUNK % NT == 0 || c < NCHUNK)
        *reinterpret_cast<uint4*>(lds + (c / CPR) * DTF_LDS_STRIDE + (c % CPR) * 8) =
            ((okbits >> i) & 1u) ? regs[i] : make_uint4(0, 0, 0, 0);
    }
  }
};

// Filter-gradient activation gather (transposing): row r = (tap, cin), k = output position (b,y,x):
//   A[r][k] = X[b, y*stride + dy - pad, x*stride + dx - pad, cin]     (bf16 NHWC, Cin even)
template <int R, int NT, int BK>
struct LoadConvDwA {
  static constexpr int CPR = BK / 8, NTASK = (R / 2) * CPR, PER = (NTASK + NT - 1) / NT;
  uint32_t regs[PER][8];
  unsigned okbits[PER];
  const uint16_t* src;
  int dy[PER], dx[PER], ci[PER];
  bool okr[PER];
  ConvGeom g;
  int npos;
  __device__ __forceinline__ void init(const GemmParams& p, int r0, int tid) {
    src = reinterpret_cast<const uint16_t*>(p.A);
    g = p.g;
    npos = g.Ho * g.Wo;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int t = tid + NT * i;
      const int r = r0 + 2 * (t % (R / 2));
      okr[i] = (NTASK % NT == 0 || t < NTASK) && r < p.M;
      const int tap = r / g.Cin;
      ci[i] = r - tap * g.Cin;
      dy[i] = tap / g.KW - g.pad_h;
      dx[i] = tap % g.KW - g.pad_w;
    }
  }
  __device__ __forceinline__ void load(int k0, int kend, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int t = tid + NT * i;
      const int kc = t / (R / 2);
      const int pos = k0 + kc * 8;
      int b = pos / npos;
      const int rem = pos - b * npos;
      int y = rem / g.Wo, x = rem - y * g.Wo;
      okbits[i] = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int hi_ = y * g.stride + dy[i], wi_ = x * g.stride + dx[i];
        const bool ok = okr[i] && (pos + j) < kend && (unsigned)hi_ < (unsigned)g.H && (unsigned)wi_ < (unsigned)g.W;
        okbits[i] |= (ok ? 1u : 0u) << j;
        regs[i][j] = *reinterpret_cast<const uint32_t*>(src + (ok ? ((long)(b * g.H + hi_) * g.W + wi_) * g.cstride + ci[i] : 0L));
        if (++x == g.Wo) { x = 0; if (++y == g.Ho) { y = 0; ++b; } }
      }
    }
  }
  __device__ __forceinline__ void store(uint16_t* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int t = tid + NT * i;
      if (NTASK % NT == 0 || t < NTASK) {
        const int rp = t % (R / 2), kc = t / (R / 2);
        uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t v = ((okbits[i] >> j) & 1u) ? regs[i][j] : 0u;
          lo[j >> 1] |= (v & 0xffffu) << ((j & 1) * 16);
          hi[j >> 1] |= (v >> 16) << ((j & 1) * 16);
        }
        *reinterpret_cast<uint4*>(lds + (2 * rp) * DTF_LDS_STRIDE + kc * 8) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        *reinterpret_cast<uint4*>(lds + (2 * rp + 1) * DTF_LDS_STRIDE + kc * 8) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      }
    }
  }
};

// ------------------------------------------------------------------ epilogues
// run(p, acc, mbase, nbase, lane, z): acc[TM][TN] of f32x16; rows/cols per the C/D map above.
#define DTF_EP_FOREACH(body)                                                        \
  _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) {                              \
    const int row = mbase + tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);     \
    const int col = nbase + tn * 32 + (lane & 31);                                  \
    const float v = acc[tm][tn][i];                                                 \
    body                                                                            \
  }

// (the bias of a column is loaded once per 32-column tile, ahead of the element loop: inside it, under the range check, the
// compiler emits branch + load + s_waitcnt per element — 16 x TM x TN dependent round trips per lane, see
// scripts/dbg/isa_serial_loads.py)
struct EpStoreF32 {  // C[z][m][n] = acc (+bias) (relu)
  template <int TM, int TN>
  static __device__ __forceinline__ void run(const GemmParams& p, f32x16 (&acc)[TM][TN], int mbase, int nbase, int lane, int z) {
    float* C = reinterpret_cast<float*>(p.C) + (long)z * p.slab_stride;
    float bv[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = nbase + tn * 32 + (lane & 31);
      const float b = p.bias ? p.bias[col < p.N ? col : 0] : 0.f;
      bv[tn] = col < p.N ? b : 0.f;
    }
    DTF_EP_FOREACH(if (row < p.M && col < p.N) {
      float o = v + bv[tn];
      if (p.relu) o = fmaxf(o, 0.f);
      C[(long)row * p.ldc + col] = o;
    })
  }
};

struct EpAtomicF32 {  // C[m][n] += acc   (split-K / cross-workgroup accumulation)
  template <int TM, int TN>
  static __device__ __forceinline__ void run(const GemmParams& p, f32x16 (&acc)[TM][TN], int mbase, int nbase, int lane, int) {
    float* C = reinterpret_cast<float*>(p.C);
    DTF_EP_FOREACH(if (row < p.M && col < p.N) atomicAdd(&C[(long)row * p.ldc + col], v);)
  }
};

struct EpBiasReluBf16 {  // out[m*ldc + n] = bf16(relu(acc + bias[n]))   (conv into a concat slice)
  template <int TM, int TN>
  static __device__ __forceinline__ void run(const GemmParams& p, f32x16 (&acc)[TM][TN], int mbase, int nbase, int lane, int) {
    uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
    float bv[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = nbase + tn * 32 + (lane & 31);
      const float b = p.bias ? p.bias[col < p.N ? col : 0] : 0.f;
      bv[tn] = col < p.N ? b : 0.f;
    }
    DTF_EP_FOREACH(if (row < p.M && col < p.N) {
      float o = v + bv[tn];
      if (p.relu) o = fmaxf(o, 0.f);
      C[(long)row * p.ldc + col] = f2bf(o);
    })
  }
};

struct EpMaskMulF32 {  // C = acc * (mask_bf16 > 0): conv bwd-data through the ReLU of the pooled input
  template <int TM, int TN>
  static __device__ __forceinline__ void run(const GemmParams& p, f32x16 (&acc)[TM][TN], int mbase, int nbase, int lane, int) {
    float* C = reinterpret_cast<float*>(p.C);
    const uint16_t* mask = reinterpret_cast<const uint16_t*>(p.aux_in);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int col = nbase + tn * 32 + (lane & 31);
        uint16_t mk[16];  // (the 16 mask values of a tile are requested together: clamped address + select)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = mbase + tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
          mk[i] = mask[(row < p.M && col < p.N) ? (long)row * p.ldc + col : 0L];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = mbase + tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
          if (row < p.M && col < p.N)
            C[(long)row * p.ldc + col] = ((mk[i] & 0x7fffu) != 0 && (mk[i] & 0x8000u) == 0) ? acc[tm][tn][i] : 0.f;
        }
      }
  }
};

// Training-mode conv: raw bf16 output z[m][n] plus per-channel sum / sum of squares of the stored (bf16-rounded)
// values accumulated into aux_out = float[DTF_STAT_SLOTS][2][N] (pre-zeroed) — the batch-norm statistics come out of
// the conv.  Workgroups spread over DTF_STAT_SLOTS replicas of the accumulator: with 10^4 workgroups hitting the same
// 2N addresses the L2 atomic unit serialises them (measured 730 us for the 149x149 stem convs, ~10x their HBM time).
#define DTF_STAT_SLOTS 16
struct EpBf16Stats {
  template <int TM, int TN>
  static __device__ __forceinline__ void run(const GemmParams& p, f32x16 (&acc)[TM][TN], int mbase, int nbase, int lane, int) {
    uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
    float* stats = reinterpret_cast<float*>(p.aux_out) + (long)(blockIdx.x % DTF_STAT_SLOTS) * 2 * p.N;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = nbase + tn * 32 + (lane & 31);
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = mbase + tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
          if (row < p.M && col < p.N) {
            const uint32_t b = f2bf(acc[tm][tn][i]);
            C[(long)row * p.ldc + col] = (uint16_t)b;
            const float v = bf2f(b);
            s += v;
            ss += v * v;
          }
        }
      s += __shfl_xor(s, 32);
      ss += __shfl_xor(ss, 32);
      if (lane < 32 && col < p.N) {
        atomicAdd(&stats[col], s);
        atomicAdd(&stats[p.N + col], ss);
      }
    }
  }
};

struct EpBf16Acc {  // out[m*ldc + n] = bf16(acc (+ existing out when p.accumulate))   (gradient into a shared buffer)
  template <int TM, int TN>
  static __device__ __forceinline__ void run(const GemmParams& p, f32x16 (&acc)[TM][TN], int mbase, int nbase, int lane, int) {
    uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
    if (!p.accumulate) {  // (workgroup-uniform)
      DTF_EP_FOREACH(if (row < p.M && col < p.N) C[(long)row * p.ldc + col] = (uint16_t)f2bf(v);)
      return;
    }
    // The 16 old values of a 32x32 tile are requested together (clamped address + select), then added and stored: a load
    // under `if (row < M && col < N)` is a branch with its own s_waitcnt, i.e. 16 x TM x TN dependent round trips to L2
    // per lane (found in the ISA: every data-gradient kernel of the fine-tune step spent its epilogue that way).
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int col = nbase + tn * 32 + (lane & 31);
        uint16_t old[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = mbase + tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
          const bool ok = row < p.M && col < p.N;
          old[i] = C[ok ? (long)row * p.ldc + col : 0L];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = mbase + tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
          if (row < p.M && col < p.N) C[(long)row * p.ldc + col] = (uint16_t)f2bf(acc[tm][tn][i] + bf2f(old[i]));
        }
      }
  }
};

// bias + ReLU + 2x2/2 max-pool over rows in pool-window order (LoadConvA ORDER 1).
// out: bf16 [M/4][N] (== NHWC pooled), aux_out: uint8 argmax (0..3 = 2*sy+sx) for the backward.
struct EpBiasReluPool {
  template <int TM, int TN>
  static __device__ __forceinline__ void run(const GemmParams& p, f32x16 (&acc)[TM][TN], int mbase, int nbase, int lane, int) {
    uint16_t* out = reinterpret_cast<uint16_t*>(p.C);
    uint8_t* amax = reinterpret_cast<uint8_t*>(p.aux_out);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int col = nbase + tn * 32 + (lane & 31);
        const float bv = (col < p.N && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int row = mbase + tm * 32 + 8 * gq + 4 * (lane >> 5);
          float best = acc[tm][tn][4 * gq];
          int arg = 0;
#pragma unroll
          for (int j = 1; j < 4; ++j) {
            const float v = acc[tm][tn][4 * gq + j];
            if (v > best) { best = v; arg = j; }
          }
          if (row < p.M && col < p.N) {
            const long o = (long)(row >> 2) * p.ldc + col;
            out[o] = f2bf(fmaxf(best + bv, 0.f));
            if (amax) amax[o] = (uint8_t)arg;
          }
        }
      }
  }
};

// ------------------------------------------------------------------ the kernel
template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP>
__device__ __forceinline__ void dtf_gemm_body(const GemmParams& p, const int m0, const int n0, const int z,
                                              uint16_t* __restrict__ As, uint16_t* __restrict__ Bs) {
  // As: BM * DTF_LDS_STRIDE halves, Bs: BN * DTF_LDS_STRIDE halves of LDS, 16-byte aligned (the kernel owns the
  // allocation so horizontally fused kernels can share one buffer between different instantiations)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(TM >= 1 && TN >= 1 && BM % (WM * 32) == 0 && BN % (WN * 32) == 0 && BK % 16 == 0, "tile");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int kbeg = z * p.kchunk;
  const int kend = min(p.K, kbeg + p.kchunk);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  AL al;
  BL bl;
  al.init(p, m0, tid);
  bl.init(p, n0, tid);
  al.load(kbeg, kend, tid);
  bl.load(kbeg, kend, tid);
  const int r = lane & 31, h = lane >> 5;
  for (int k = kbeg; k < kend; k += BK) {
    __syncthreads();
    al.store(As, tid);
    bl.store(Bs, tid);
    __syncthreads();
    if (k + BK < kend) {
      al.load(k + BK, kend, tid);
      bl.load(k + BK, kend, tid);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        af[a] = *reinterpret_cast<const bf16x8*>(&As[(wm * (BM / WM) + a * 32 + r) * DTF_LDS_STRIDE + kk + 8 * h]);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        bfr[b] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * (BN / WN) + b * 32 + r) * DTF_LDS_STRIDE + kk + 8 * h]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
  }
  EP::template run<TM, TN>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane, z);
}


// C = sum over `nranks` operand pairs (A_r, B_r) of A_r * B_r, each pair with the same shapes / strides as p (K <= BK: one
// k-step per pair).  The pairs may live in peer GPUs' memory; pair r+1 is in flight while pair r is multiplied.
template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP>
__device__ __forceinline__ void dtf_gemm_body_ranks(const GemmParams& p, const void* const* __restrict__ Ar,
                                                    const void* const* __restrict__ Br, const int nranks, const int m0,
                                                    const int n0, uint16_t* __restrict__ As, uint16_t* __restrict__ Bs) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  AL al;
  BL bl;
  al.init(p, m0, tid);
  bl.init(p, n0, tid);
  al.src = reinterpret_cast<decltype(al.src)>(Ar[0]);
  bl.src = reinterpret_cast<decltype(bl.src)>(Br[0]);
  al.load(0, p.K, tid);
  bl.load(0, p.K, tid);
  const int r = lane & 31, h = lane >> 5;
  for (int rk = 0; rk < nranks; ++rk) {
    __syncthreads();
    al.store(As, tid);
    bl.store(Bs, tid);
    __syncthreads();
    if (rk + 1 < nranks) {
      al.src = reinterpret_cast<decltype(al.src)>(Ar[rk + 1]);
      bl.src = reinterpret_cast<decltype(bl.src)>(Br[rk + 1]);
      al.load(0, p.K, tid);
      bl.load(0, p.K, tid);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        af[a] = *reinterpret_cast<const bf16x8*>(&As[(wm * (BM / WM) + a * 32 + r) * DTF_LDS_STRIDE + kk + 8 * h]);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        bfr[b] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * (BN / WN) + b * 32 + r) * DTF_LDS_STRIDE + kk + 8 * h]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
  }
  EP::template run<TM, TN>(p, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), lane, 0);
}

// Weight-gradient GEMM with the batch as K (K <= 128, one slab):  C[m][n] = sum over `nranks` operand pairs of A_r^T B_r,
// A_r bf16 [K][lda] and B_r bf16 [K][ldb] row-major (features contiguous).  The tiles go into LDS AS THEY ARE -- a few
// coalesced 16-byte loads per lane -- and the MFMA fragments are read with the transposing LDS load
// (ds_read_b64_tr_b16), instead of transposing in registers on the way in: that loader needs 48 four-byte loads per
// lane and spent 5.4 of the fc1 filter-gradient workgroup's 10.8 us issuing and waiting for them (phase timestamps).
// 128 x 64 tile, 4 waves (32 rows each); LDS row pitches 320 / 192 bytes keep the 4-row read groups on distinct banks.
// Requires M, N, lda, ldb multiples of 8 and 16-byte aligned operands.  smem: 128 * (320 + 192) bytes.
constexpr int DTF_TN_PA = 320, DTF_TN_PB = 192, DTF_TN_LDS = 128 * (DTF_TN_PA + DTF_TN_PB);
template <class EP>
__device__ __forceinline__ void dtf_tn_body_tr(const GemmParams& p, const void* const* __restrict__ Ar,
                                               const void* const* __restrict__ Br, const int nranks, const int m0,
                                               const int n0, char* __restrict__ smem) {
  typedef short s16x4_t __attribute__((ext_vector_type(4)));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  char* As = smem;
  char* Bs = smem + 128 * DTF_TN_PA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = p.K, kpad = (K + 15) & ~15, nks = kpad >> 4;
  u32x4_t av[8], bv[4];
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};
  auto load = [&](const void* A, const void* B) {
    const uint16_t* a16 = reinterpret_cast<const uint16_t*>(A);
    const uint16_t* b16 = reinterpret_cast<const uint16_t*>(B);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i, row = c >> 4, cc = c & 15;
      const bool ok = row < K && m0 + cc * 8 < p.M;
      av[i] = *reinterpret_cast<const u32x4_t*>(a16 + (ok ? (long)row * p.lda + m0 + cc * 8 : 0L));
      av[i] = ok ? av[i] : zero4;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i, row = c >> 3, cc = c & 7;
      const bool ok = row < K && n0 + cc * 8 < p.N;
      bv[i] = *reinterpret_cast<const u32x4_t*>(b16 + (ok ? (long)row * p.ldb + n0 + cc * 8 : 0L));
      bv[i] = ok ? bv[i] : zero4;
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i, row = c >> 4, cc = c & 15;
      if (row < kpad) *reinterpret_cast<u32x4_t*>(As + row * DTF_TN_PA + cc * 16) = av[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i, row = c >> 3, cc = c & 7;
      if (row < kpad) *reinterpret_cast<u32x4_t*>(Bs + row * DTF_TN_PB + cc * 16) = bv[i];
    }
  };
  auto tr8 = [&](const char* lo_p, const int pitch) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lo_p));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lo_p + 4 * pitch));
    const s16x8_t v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v8);
  };
  // lane mapping of the transposing read (as in conv2_dw_body): 16 lanes cover 4 k-rows x 16 columns
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = (i16 & 3) * 4;
  const int mcol = 16 * (g & 1) + p4, krow = 8 * (g >> 1) + q;
  const char* abase = As + krow * DTF_TN_PA + (wave * 32 + mcol) * 2;
  const char* bbase = Bs + krow * DTF_TN_PB + mcol * 2;
  f32x16 acc[1][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][b][i] = 0.f;
  load(Ar[0], Br[0]);
  for (int rk = 0; rk < nranks; ++rk) {
    __syncthreads();
    store();
    __syncthreads();
    if (rk + 1 < nranks) load(Ar[rk + 1], Br[rk + 1]);
    bf16x8 a_c = tr8(abase, DTF_TN_PA), b0_c = tr8(bbase, DTF_TN_PB), b1_c = tr8(bbase + 64, DTF_TN_PB);
    for (int ks = 0; ks < nks; ++ks) {
      bf16x8 a_n = a_c, b0_n = b0_c, b1_n = b1_c;
      if (ks + 1 < nks) {
        a_n = tr8(abase + (ks + 1) * 16 * DTF_TN_PA, DTF_TN_PA);
        b0_n = tr8(bbase + (ks + 1) * 16 * DTF_TN_PB, DTF_TN_PB);
        b1_n = tr8(bbase + (ks + 1) * 16 * DTF_TN_PB + 64, DTF_TN_PB);
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_c, b0_c, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_c, b1_c, acc[0][1], 0, 0, 0);
      a_c = a_n; b0_c = b0_n; b1_c = b1_n;
    }
  }
  EP::template run<1, 2>(p, acc, m0 + wave * 32, n0, lane, 0);
}

template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP>
__global__ __launch_bounds__(64 * WM * WN) void dtf_gemm_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) uint16_t smem[(BM + BN) * DTF_LDS_STRIDE];
  dtf_gemm_body<BM, BN, BK, WM, WN, AL, BL, EP>(p, blockIdx.x * BM, blockIdx.y * BN, blockIdx.z, smem,
                                                smem + BM * DTF_LDS_STRIDE);
}

// Two independent problems of the same instantiation in ONE launch (horizontal fusion of small GEMMs:
// each extra kernel boundary costs ~4 us on this step).  Blocks [0, nblk0) run problem 0.
struct GemmPair {
  GemmParams p[2];
  int nblk0, mt0, mt1;  // blocks of problem 0; m-tiles per problem
};
template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP>
__global__ __launch_bounds__(64 * WM * WN) void dtf_gemm_pair_kernel(const GemmPair g) {
  const int which = (int)blockIdx.x >= g.nblk0;
  const int id = which ? blockIdx.x - g.nblk0 : blockIdx.x;
  const int mt = which ? g.mt1 : g.mt0;
  __shared__ __attribute__((aligned(16))) uint16_t smem[(BM + BN) * DTF_LDS_STRIDE];
  dtf_gemm_body<BM, BN, BK, WM, WN, AL, BL, EP>(g.p[which], (id % mt) * BM, (id / mt) * BN, 0, smem,
                                                smem + BM * DTF_LDS_STRIDE);
}

template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP>
static inline int dtf_launch_gemm(const GemmParams& p, int splits, hipStream_t st) {
  dim3 grid(dtf_cdiv(p.M, BM), dtf_cdiv(p.N, BN), splits);
  hipLaunchKernelGGL((dtf_gemm_kernel<BM, BN, BK, WM, WN, AL, BL, EP>), grid, dim3(64 * WM * WN), 0, st, p);
  DTF_CHECK_LAUNCH();
  return 0;
}

template <int BM, int BN, int BK, int WM, int WN, class AL, class BL, class EP>
static inline int dtf_launch_gemm_pair(const GemmParams& p0, const GemmParams& p1, hipStream_t st) {
  GemmPair g;
  g.p[0] = p0;
  g.p[1] = p1;
  g.mt0 = dtf_cdiv(p0.M, BM);
  g.mt1 = dtf_cdiv(p1.M, BM);
  g.nblk0 = g.mt0 * dtf_cdiv(p0.N, BN);
  const int total = g.nblk0 + g.mt1 * dtf_cdiv(p1.N, BN);
  hipLaunchKernelGGL((dtf_gemm_pair_kernel<BM, BN, BK, WM, WN, AL, BL, EP>), dim3(total), dim3(64 * WM * WN), 0, st, g);
  DTF_CHECK_LAUNCH();
  return 0;
}
