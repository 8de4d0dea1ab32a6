// Implicit-GEMM convolution core, second generation: asynchronous global->LDS loads (global_load_lds_dwordx4, no
// VGPR staging), a 3-stage LDS ring with ONE barrier per K-step, 64x64 wave tiles (4 MFMA 32x32x16 per 4 fragment
// reads), XOR-swizzled unpadded LDS rows, and an XCD-aware workgroup order.
//
//   C[M, N] = A_gather[M, K] * Wp[N, K]^T          bf16 operands, fp32 accumulate
//
// * A is gathered on the fly from a bf16 NHWC tensor (forward: im2col of the input; backward-data: the transposed
//   gather of the output gradient, stride 1 or 2).  Out-of-image taps, rows >= M and k >= K are given an out-of-range
//   buffer offset (the descriptor's range check returns zeros) instead of being predicated: every lane always issues
//   its load, so the per-wave vmcnt arithmetic is static.
// * Wp is a bf16 [N][K] (K contiguous) packing of the fp32 master weights, refreshed once per step by
//   `dtf_pack_conv_weights` (87 MB read / 87 MB written for all of Inception-v3: ~40 us).
// * LDS stage = A tile [256 rows][128 B] + B tile [BN rows][128 B]; a row is 8 chunks of 16 B and chunk c of row r is
//   stored at chunk position c ^ ((r >> 1) & 7).  ds_read_b128 is served in four 16-lane groups ({0-3,12-15,20-27},
//   ...; MI355X_MICROARCH.md §LDS) over a 256-B bank row = two 128-B tile rows: within a group the pairs
//   (row parity, (row >> 1) & 7) are all different, so the 16 lanes hit 16 different 16-B slots (c ^ (r & 7) would be
//   2-way conflicted: rows 0 and 24 share a group).  The swizzle is applied on the *global* side (the lane that fills
//   slot s of row R fetches logical chunk s ^ ((R >> 1) & 7)), because the LDS side of global_load_lds is fixed: lane l
//   writes 16 B at M0 + 16*l.
// * Pipeline: loads of K-step t+2 are issued right after the barrier of step t, into the stage consumed at step t-1;
//   `s_waitcnt vmcnt(LPS)` (LPS = loads per stage per wave) lets the newest stage stay in flight.
// * Workgroup order: id -> (xcd = id % 8, local = id / 8); the N tiles of one M tile are consecutive `local`s of the
//   same XCD, so the gathered A tile is fetched into one L2 and re-used from there.
#include "gemm_core.h"
#include "dtf_api.h"

namespace {
constexpr int ERR_SHAPE = -22;

struct IG2Params {
  GemmParams gp;  // A, C, M, N, K, ldc, bias, relu, aux_*, accumulate, g (geometry)
  const uint16_t* Wp;
  int mtiles, ntiles;
  unsigned magic_c, magic_kw;  // floor(2^32 / d) + 1 for d = channels per tap, KW
  // Stride-2 backward-data by parity class (template flag S2).  With stride 2 a tap (dy, dx) reaches an output only from
  // input pixels with dy = h + pad_h and dx = w + pad_w (mod 2), so in pixel order three quarters of a tile's MFMAs
  // multiply zeros.  Here the M rows are ordered class by class (class = 2 * parity(h + pad_h) + parity(w + pad_w)), every
  // class padded to whole tiles, and a tile runs only the K-steps that touch a tap of its class (smask).
  int tile0[4];                 // first m tile of class c (non-decreasing; mtiles = end of class 3)
  int rows[4];                  // rows of class c = B * nh * nw
  int nh[4], nw[4], h0[4], w0[4];  // class c holds the pixels (h0 + 2i, w0 + 2j), i < nh, j < nw
  unsigned long long smask[4];  // bit t: K-step t (64 k) touches a tap of class c
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MODE 0: forward gather (row = output position, tap shifts into the input)
// MODE 1: backward-data gather (row = input position, tap shifts into the output gradient; stride 1 or 2)
// Tile variants: BM x BN x stages.  256-row tiles keep one workgroup per CU (120-144 KB of LDS) and pay off when an M
// tile has many K-steps; 128-row tiles (72 / 64 KB) put two workgroups on a CU, whose prologues and epilogues then
// overlap — the better shape for the short K loops (K = 192..800) that dominate this network.
template <int TM, int TN>
__device__ __forceinline__ void ep_acc_s2(const IG2Params& q, int cls, f32x16 (&acc)[TM][TN], int mbase, int nbase, int lane);

template <int BM, int BN, int S, int MODE, class EP, bool S2 = false>
__global__ __launch_bounds__((BM / 64) * (BN / 64) * 64) void igemm2_kernel(const IG2Params q) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the buffer-resource builtins do not exist in the host pass)
  constexpr int BK = 64;
  constexpr int WN = BN / 64, NW = (BM / 64) * WN;
  static_assert(S == 2 || S == 3, "ring depth");
  static_assert(((BM / 8) % NW) == 0 && ((BN / 8) % NW) == 0 && (((BM / 8) / NW) % 2) == 0 && (((BN / 8) / NW) % 2) == 0,
                "every wave issues an even number of 8-row load instructions per operand (swizzle parity = i & 1)");
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PER = (BM / 8) / NW, B_PER = (BN / 8) / NW, LPS = A_PER + B_PER;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const GemmParams& p = q.gp;
  const ConvGeom g = p.g;

  const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
  const int nt = local % q.ntiles, mt = (local / q.ntiles) * 8 + xcd;
  if (mt >= q.mtiles) return;
  static_assert(!S2 || MODE == 1, "parity classes exist for the stride-2 data gradient only");
  int cls = 0;
  if (S2) cls = (mt >= q.tile0[1] ? 1 : 0) + (mt >= q.tile0[2] ? 1 : 0) + (mt >= q.tile0[3] ? 1 : 0);
  const int m0 = S2 ? (mt - q.tile0[cls]) * BM : mt * BM, n0 = nt * BN;  // (S2: row index inside the class)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane >> 3;
  // this lane's logical k offset inside a K-step (swizzle on the global side): LDS row R stores chunk c at position
  // c ^ ((R >> 1) & 7); the 8 rows of load instruction i start at a multiple of 8, so (R >> 1) & 7 = 4*(i & 1) + (lrow >> 1)
  const int kc8_even = ((lane & 7) ^ (lrow >> 1)) * 8, kc8_odd = ((lane & 7) ^ (4 + (lrow >> 1))) * 8;

  // ---- per-lane row descriptors of the A gather.  Issuing a gathered tile must cost few VALU instructions (they compete
  // with the MFMAs of the one wave per SIMD), so everything that does not change with the K-step is folded into, per
  // row, one 32-bit byte offset and one tap-validity bitmask; a load is then  off = (mask & tapbit) ? base + tapoff :
  // OOB, through a raw buffer descriptor whose range check turns OOB into zeros (padding taps, rows >= M, k >= K).
  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, 0x80000000, 0x00020000);
  const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(q.Wp), 0, 0x80000000, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int sh = g.stride - 1;  // stride 1 or 2
  const int cs_eff = MODE == 0 ? g.cstride : (g.cstride >> sh);
  unsigned abase_off[A_PER], amask[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int m = m0 + (wave * A_PER + i) * 8 + lrow;
    const bool okr = S2 ? m < q.rows[cls] : m < p.M;
    unsigned rowbits = 0, colbits = 0;
    if (MODE == 0) {
      const int wo = m % g.Wo, t = m / g.Wo;
      const int im = t / g.Ho;
      const int hi0 = (t % g.Ho) * g.stride - g.pad_h, wi0 = wo * g.stride - g.pad_w;
      abase_off[i] = (unsigned)(((im * g.H + hi0) * g.W + wi0) * g.cstride) * 2u;
      for (int d = 0; d < g.KH; ++d) rowbits |= ((unsigned)(hi0 + d) < (unsigned)g.H ? 1u : 0u) << d;
      for (int d = 0; d < g.KW; ++d) colbits |= ((unsigned)(wi0 + d) < (unsigned)g.W ? 1u : 0u) << d;
    } else {
      int w, hh, im;
      if (S2) {
        const int mm = okr ? m : 0, nwc = q.nw[cls], nhc = q.nh[cls], t = mm / nwc;
        w = q.w0[cls] + 2 * (mm - t * nwc);
        im = t / nhc;
        hh = q.h0[cls] + 2 * (t - im * nhc);
      } else {
        const int t = m / g.W;
        w = m % g.W;
        im = t / g.H;
        hh = t % g.H;
      }
      const int hb = hh + g.pad_h, wb = w + g.pad_w;
      abase_off[i] = (unsigned)(im * g.Ho * g.Wo * g.cstride + (hb * g.Wo + wb) * cs_eff) * 2u;
      for (int d = 0; d < g.KH; ++d) {
        const int hz = hb - d;
        rowbits |= ((hz >= 0 && (hz & sh) == 0 && (hz >> sh) < g.Ho) ? 1u : 0u) << d;
      }
      for (int d = 0; d < g.KW; ++d) {
        const int wz = wb - d;
        colbits |= ((wz >= 0 && (wz & sh) == 0 && (wz >> sh) < g.Wo) ? 1u : 0u) << d;
      }
    }
    unsigned mk = 0;
    for (int d = 0; d < g.KH; ++d) mk |= ((rowbits >> d) & 1u) ? (colbits << (d * g.KW)) : 0u;
    amask[i] = okr ? mk : 0u;
  }
  unsigned brow[B_PER];
#pragma unroll
  for (int j = 0; j < B_PER; ++j) {
    const int n = n0 + (wave * B_PER + j) * 8 + lrow;
    brow[j] = n < p.N ? (unsigned)n * (unsigned)p.K * 2u : OOB;
  }

  auto issue = [&](int step, int sb) {
    unsigned tapoff[2], tapbit[2], kb[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int kk = step * BK + (e ? kc8_odd : kc8_even);
      const bool kok = kk < p.K;
      const int tap = (int)__umulhi((unsigned)kk, q.magic_c);  // kk / Cin (exact for kk < 2^21)
      const int ci = kk - tap * g.Cin;
      const int dy = g.KW == 1 ? tap : (int)__umulhi((unsigned)tap, q.magic_kw);
      const int dx = tap - dy * g.KW;
      tapoff[e] = MODE == 0 ? (unsigned)((dy * g.W + dx) * g.cstride + ci) * 2u
                            : (unsigned)(ci - (dy * g.Wo + dx) * cs_eff) * 2u;
      tapbit[e] = kok ? (1u << (tap & 31)) : 0u;
      kb[e] = kok ? (unsigned)kk * 2u : OOB;  // brow + kb: at most one of the two may carry the OOB bit
    }
    unsigned char* abase = smem + sb * STAGE + wave * A_PER * 1024;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const unsigned off = (amask[i] & tapbit[i & 1]) ? abase_off[i] + tapoff[i & 1] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_ptr_t)(abase + i * 1024), 16, off, 0, 0, 0);
    }
    unsigned char* bbase = smem + sb * STAGE + A_BYTES + wave * B_PER * 1024;
#pragma unroll
    for (int j = 0; j < B_PER; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_ptr_t)(bbase + j * 1024), 16, (brow[j] | kb[j & 1]) & OOB ? OOB : brow[j] + kb[j & 1],
                                               0, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  const int nsteps = (p.K + BK - 1) / BK;
  // S2: the steps of this tile's class, in order; once they run out the prefetches ask for step nsteps (zeros, as below)
  unsigned long long pend = S2 ? q.smask[cls] : 0ull;
  const int nrun = S2 ? __builtin_popcountll(pend) : nsteps;
  auto next_step = [&](int linear) -> int {
    if (!S2) return linear;
    if (!pend) return nsteps;
    const int t = __builtin_ctzll(pend);
    pend &= pend - 1;
    return t;
  };
  issue(next_step(0), 0);
  if (S == 3) issue(next_step(1), 1);  // steps >= nsteps only fetch out-of-range zeros (k >= K)

  const int r = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int sw = (r >> 1) & 7;  // fragment rows start at multiples of 32: (R >> 1) & 7 == (r >> 1) & 7
  for (int t = 0; t < nrun; ++t) {
    wait_vmcnt<(S - 2) * LPS>();  // this wave's loads of stage t have landed (with S == 3 stage t+1 stays in flight)
    // Bare s_barrier, not __syncthreads(): the latter's fence makes hipcc emit s_waitcnt vmcnt(0), which would drain
    // the stage that is supposed to stay in flight.  After it everyone's stage-t data is in LDS and everyone is done
    // reading the stage consumed at step t-1 (its ds_reads were waited on before the MFMAs that used them).
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(next_step(t + S - 1), (t + S - 1) % S);
    const unsigned char* As = smem + (t % S) * STAGE;
    const unsigned char* Bs = As + A_BYTES;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
      const int ch = ((2 * k4 + h) ^ sw) * 16;
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const bf16x8*>(As + (wm * 64 + a * 32 + r) * 128 + ch);
#pragma unroll
      for (int b = 0; b < 2; ++b) bfr[b] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 64 + b * 32 + r) * 128 + ch);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
  }
  wait_vmcnt<0>();  // drain the out-of-range prefetches before the LDS is released
  if (S2)
    ep_acc_s2<2, 2>(q, cls, acc, m0 + wm * 64, n0 + wn * 64, lane);
  else
    EP::template run<2, 2>(p, acc, m0 + wm * 64, n0 + wn * 64, lane, 0);
#endif
}

// EpBf16Acc for rows in parity-class order: row -> pixel (im, h0 + 2i, w0 + 2j) of the class, then the same
// "request the 16 old values of a 32x32 tile together" accumulate as EpBf16Acc.
template <int TM, int TN>
__device__ __forceinline__ void ep_acc_s2(const IG2Params& q, int cls, f32x16 (&acc)[TM][TN], int mbase, int nbase, int lane) {
  const GemmParams& p = q.gp;
  uint16_t* C = reinterpret_cast<uint16_t*>(p.C);
  const int rows = q.rows[cls], nwc = q.nw[cls], nhc = q.nh[cls], h0 = q.h0[cls], w0 = q.w0[cls];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    long off[16];
    bool okr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ml = mbase + tm * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      okr[i] = ml < rows;
      const int mm = okr[i] ? ml : 0, t = mm / nwc, im = t / nhc;
      off[i] = ((long)(im * p.g.H + h0 + 2 * (t - im * nhc)) * p.g.W + (w0 + 2 * (mm - t * nwc))) * p.ldc;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int col = nbase + tn * 32 + (lane & 31);
      const bool cok = col < p.N;
      if (!p.accumulate) {  // (workgroup-uniform)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (okr[i] && cok) C[off[i] + col] = (uint16_t)f2bf(acc[tm][tn][i]);
      } else {
        uint16_t old[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) old[i] = C[(okr[i] && cok) ? off[i] + col : 0L];
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (okr[i] && cok) C[off[i] + col] = (uint16_t)f2bf(acc[tm][tn][i] + bf2f(old[i]));
      }
    }
  }
}

template <int BM, int BN, int S, int MODE, class EP, bool S2 = false>
int launch_one(IG2Params& q, hipStream_t st) {
  constexpr int NTHREADS = (BM / 64) * (BN / 64) * 64;
  constexpr int LDS = S * (BM * 128 + BN * 128);
  if (!S2) q.mtiles = dtf_cdiv(q.gp.M, BM);  // (S2: the caller has laid out the classes in tiles of BM rows)
  q.ntiles = dtf_cdiv(q.gp.N, BN);
  const int mt8 = (q.mtiles + 7) / 8 * 8;
  hipLaunchKernelGGL((igemm2_kernel<BM, BN, S, MODE, EP, S2>), dim3(mt8 * q.ntiles), dim3(NTHREADS), LDS, st, q);
  DTF_CHECK_LAUNCH();
  return 0;
}

// variant bit 0: 128-row tiles (3 stages at BN = 64, 2 at BN = 128: two workgroups per CU) instead of 256-row tiles;
// variant bit 1: 64-column tiles even when N > 64 (N = 160 or 192 fills 64-wide tiles far better than 128-wide ones)
// variant bit 2 (backward-data of stride-2 layers only): rows by parity class, see IG2Params / dtf_conv_bwd_data_v2
template <int BN, int MODE, class EP>
int launch_igemm2(IG2Params& q, int variant, hipStream_t st) {
  if (variant & 1) return launch_one<128, BN, (BN == 64 ? 3 : 2), MODE, EP>(q, st);
  return launch_one<256, BN, 3, MODE, EP>(q, st);
}
template <int MODE, class EP>
int launch_igemm2_n(IG2Params& q, int variant, hipStream_t st) {
  if (variant < 0 || variant > 3) return ERR_SHAPE;
  if (q.gp.N > 64 && !(variant & 2)) return launch_igemm2<128, MODE, EP>(q, variant, st);
  return launch_igemm2<64, MODE, EP>(q, variant, st);
}

template <int BM, int BN, int S, int MODE, class EP, bool S2 = false>
bool raise_one() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<BM, BN, S, MODE, EP, S2>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, S * (BM * 128 + BN * 128)) == hipSuccess;
}
template <int BN, int MODE, class EP>
bool raise_lds_limit() {
  return raise_one<256, BN, 3, MODE, EP>() && raise_one<128, BN, (BN == 64 ? 3 : 2), MODE, EP>();
}

bool g_ready = false;  // dtf_igemm2_init() has raised the dynamic-LDS limits

inline int fill(IG2Params& q, const void* A, const void* Wp, void* C, int M, int N, int K, long ldc, int H, int W, int Cg,
                int cstride, int Ho, int Wo, int KH, int KW, int stride, int pad_h, int pad_w) {
  if (!g_ready) return -2;  // dtf_igemm2_init() was not called
  q = IG2Params{};
  GemmParams& p = q.gp;
  p.A = A; p.C = C; p.M = M; p.N = N; p.K = K; p.ldc = ldc;
  p.g.H = H; p.g.W = W; p.g.Cin = Cg; p.g.cstride = cstride; p.g.Ho = Ho; p.g.Wo = Wo;
  p.g.KH = KH; p.g.KW = KW; p.g.stride = stride; p.g.pad_h = pad_h; p.g.pad_w = pad_w;
  q.Wp = reinterpret_cast<const uint16_t*>(Wp);
  q.magic_c = (unsigned)((1ull << 32) / (unsigned)Cg + 1ull);
  q.magic_kw = KW > 1 ? (unsigned)((1ull << 32) / (unsigned)KW + 1ull) : 0u;
  if ((long)K >= (1L << 21) || KH * KW > 32 || (long)N * K * 2 >= 0x7fffffffL) return ERR_SHAPE;
  if (M <= 0 || N <= 0 || K <= 0 || K % 8 || Cg % 8 || cstride % 8 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 ||
      KW <= 0 || (stride != 1 && stride != 2))
    return ERR_SHAPE;
  if ((H + 2 * pad_h - KH) / stride + 1 != Ho || (W + 2 * pad_w - KW) / stride + 1 != Wo) return ERR_SHAPE;
  if ((long)M * 1 > 0x7fffffffL) return ERR_SHAPE;
  return 0;
}
}  // namespace

// One-time setup (raises the dynamic-LDS limit of every instantiation).  Not capturable.
DTF_API int dtf_igemm2_init() {
  if (!g_ready) {
    if (!raise_lds_limit<64, 0, EpBf16Stats>() || !raise_lds_limit<128, 0, EpBf16Stats>() ||
        !raise_lds_limit<64, 0, EpBiasReluBf16>() || !raise_lds_limit<128, 0, EpBiasReluBf16>() ||
        !raise_lds_limit<64, 1, EpBf16Acc>() || !raise_lds_limit<128, 1, EpBf16Acc>() ||
        !raise_one<128, 64, 3, 1, EpBf16Acc, true>() || !raise_one<128, 128, 2, 1, EpBf16Acc, true>() ||
        !raise_one<256, 64, 3, 1, EpBf16Acc, true>() || !raise_one<256, 128, 3, 1, EpBf16Acc, true>())
      return -3;
    g_ready = true;
  }
  return 0;
}

// z[B,Ho,Wo,Cout] (bf16 dense) = conv(x, Wp) + BN statistics (see EpBf16Stats); Wp = bf16 [Cout][KH*KW*Cin]
DTF_API int dtf_conv_fwd_stats_v2(const void* x, const void* Wp, void* z, float* stats, int B, int H, int W, int Cin,
                                  int in_cstride, int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int Ho,
                                  int Wo, int variant, hipStream_t st) {
  IG2Params q;
  const int rc = fill(q, x, Wp, z, B * Ho * Wo, Cout, KH * KW * Cin, Cout, H, W, Cin, in_cstride, Ho, Wo, KH, KW, stride, pad_h, pad_w);
  if (rc) return rc;
  q.gp.aux_out = stats;
  return launch_igemm2_n<0, EpBf16Stats>(q, variant, st);
}

// Inference conv: out = bf16(relu(conv(x, Wp) + bias)) written at channel stride out_cstride (concat slice)
DTF_API int dtf_conv_bias_relu_v2(const void* x, const void* Wp, const float* bias, void* out, int B, int H, int W, int Cin,
                                  int in_cstride, int Cout, int out_cstride, int KH, int KW, int stride, int pad_h, int pad_w,
                                  int Ho, int Wo, int relu, int variant, hipStream_t st) {
  IG2Params q;
  const int rc = fill(q, x, Wp, out, B * Ho * Wo, Cout, KH * KW * Cin, out_cstride, H, W, Cin, in_cstride, Ho, Wo, KH, KW, stride, pad_h, pad_w);
  if (rc) return rc;
  q.gp.bias = bias;
  q.gp.relu = relu;
  return launch_igemm2_n<0, EpBiasReluBf16>(q, variant, st);
}

// dx[B,H,W,Cin] (bf16 at channel stride dx_cstride) (+)= conv^T(dz, Wq); Wq = bf16 [Cin][KH*KW*Cout]
DTF_API int dtf_conv_bwd_data_v2(const void* dz, const void* Wq, void* dx, int B, int H, int W, int Cin, int dx_cstride,
                                 int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int Ho, int Wo, int accumulate,
                                 int variant, hipStream_t st) {
  IG2Params q;
  const int rc = fill(q, dz, Wq, dx, B * H * W, Cin, KH * KW * Cout, dx_cstride, H, W, Cout, Cout, Ho, Wo, KH, KW, stride, pad_h, pad_w);
  if (rc) return rc;
  q.gp.accumulate = accumulate;
  if (variant & 4) {  // stride-2 layers by parity class (bits 0 and 1 pick the tile as for the other variants)
    const int S2_BM = (variant & 1) ? 128 : 256;
    const int K = KH * KW * Cout, nsteps = dtf_cdiv(K, 64);
    if (stride != 2 || variant > 7) return ERR_SHAPE;
    if (nsteps > 64) return launch_igemm2_n<1, EpBf16Acc>(q, variant & 3, st);  // (one 64-bit step mask per class: pixel order)
    int t0 = 0;
    for (int c = 0; c < 4; ++c) {
      const int ph = c >> 1, pw = c & 1;
      q.h0[c] = (ph + pad_h) & 1;
      q.w0[c] = (pw + pad_w) & 1;
      q.nh[c] = q.h0[c] < H ? (H - q.h0[c] + 1) / 2 : 0;
      q.nw[c] = q.w0[c] < W ? (W - q.w0[c] + 1) / 2 : 0;
      q.rows[c] = B * q.nh[c] * q.nw[c];
      q.tile0[c] = t0;
      t0 += dtf_cdiv(q.rows[c], S2_BM);
      unsigned long long mask = 0ull;
      for (int t = 0; t < nsteps; ++t) {
        const int k_hi = (t * 64 + 64 < K ? t * 64 + 64 : K) - 1;
        for (int tap = t * 64 / Cout; tap <= k_hi / Cout; ++tap)
          if (((tap / KW) & 1) == ph && ((tap % KW) & 1) == pw) mask |= 1ull << t;
      }
      q.smask[c] = mask;
    }
    q.mtiles = t0;
    const bool wide = q.gp.N > 64 && !(variant & 2);
    if (variant & 1)
      return wide ? launch_one<128, 128, 2, 1, EpBf16Acc, true>(q, st) : launch_one<128, 64, 3, 1, EpBf16Acc, true>(q, st);
    return wide ? launch_one<256, 128, 3, 1, EpBf16Acc, true>(q, st) : launch_one<256, 64, 3, 1, EpBf16Acc, true>(q, st);
  }
  return launch_igemm2_n<1, EpBf16Acc>(q, variant, st);
}

// ---------------------------------------------------------------------------------------------- weight packing
// Per conv: fp32 HWIO master W[tap][ci][co] -> Wp[co][tap][ci] (forward B operand) and Wq[ci][tap][co] (backward-data
// B operand), both bf16.  One launch covers every conv of the network: blockIdx.y = conv, descriptors in device memory.
struct PackDesc {
  long w_off, p_off, q_off;  // element offsets into the fp32 arena / the two bf16 pack buffers
  int taps, cin, cout, pad;
};

// One tile = (tap, 32 input channels, 64 output channels): read coalesced along co (the master's contiguous axis), written
// straight out for Wq (co contiguous) and through an LDS transpose for Wp (ci contiguous), so both stores are full
// 64-128-byte segments (the element-wise version scattered 2-byte stores at stride K: 0.64 TB/s).
__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const float* __restrict__ arena, const PackDesc* __restrict__ descs,
                                                                uint16_t* __restrict__ Wp, uint16_t* __restrict__ Wq) {
  __shared__ uint16_t t[32][66];
  const PackDesc d = descs[blockIdx.y];
  const float* w = arena + d.w_off;
  const int K = d.taps * d.cin, Kq = d.taps * d.cout;
  const int nci = (d.cin + 31) >> 5, nco = (d.cout + 63) >> 6;
  const int ntile = d.taps * nci * nco;
  const int tid = threadIdx.x;
  const int col = tid & 63, row0 = tid >> 6;         // load / Wq mapping: 4 rows per pass
  const int oc = tid >> 2, chunk = (tid & 3) * 8;    // Wp mapping: one co, 8 consecutive ci
  for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int cot = tile % nco, r = tile / nco;
    const int cit = r % nci, tap = r / nci;
    const int ci0 = cit * 32, co0 = cot * 64;
    __syncthreads();  // the previous tile has been read out
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = row0 + 4 * j;
      const int ci = ci0 + row, co = co0 + col;
      uint16_t v = 0;
      if (ci < d.cin && co < d.cout) {
        v = f2bf(w[((long)tap * d.cin + ci) * d.cout + co]);
        Wq[d.q_off + (long)ci * Kq + tap * d.cout + co] = v;
      }
      t[row][col] = v;
    }
    __syncthreads();
    const int co = co0 + oc;
    if (co < d.cout) {
      uint16_t* dst = Wp + d.p_off + (long)co * K + tap * d.cin + ci0 + chunk;
      if (ci0 + chunk + 8 <= d.cin && ((d.p_off + (long)co * K + tap * d.cin + ci0 + chunk) & 7) == 0) {
        uint32_t u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = (uint32_t)t[chunk + 2 * e][oc] | ((uint32_t)t[chunk + 2 * e + 1][oc] << 16);
        *reinterpret_cast<uint4*>(dst) = make_uint4(u[0], u[1], u[2], u[3]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (ci0 + chunk + e < d.cin) dst[e] = t[chunk + e][oc];
      }
    }
  }
}

DTF_API int dtf_pack_conv_weights(const float* arena, const void* descs, int nconv, void* Wp, void* Wq, hipStream_t st) {
  if (nconv <= 0) return ERR_SHAPE;
  hipLaunchKernelGGL(pack_conv_weights_kernel, dim3(96, nconv), dim3(256), 0, st, arena, (const PackDesc*)descs,
                     (uint16_t*)Wp, (uint16_t*)Wq);
  DTF_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------- filter gradient
// dW[(tap,ci)][co] (fp32 HWIO, pre-zeroed, split-K atomics) += sum_pos x[pos (+) tap][ci] * dz[pos][co]
//
// Both MFMA operands are K-major here (K = output positions) while memory is channel-major, so the tiles are staged in
// their natural [position][channel] layout by the same asynchronous global->LDS ring and transposed on the way out of
// LDS by ds_read_b64_tr_b16 (one transposing read delivers 4 consecutive k for this lane's m/n column).
//   workgroup = (split z, m tile, co tile) where m = tap * Cin + ci runs over the FLATTENED filter rows (HWIO is exactly
//   [m][co]): a layer with fewer input channels than the tile is wide packs several taps into one tile instead of
//   leaving it half empty (32 -> 32 channels, 3x3: 5 tiles of 64 rows instead of 9 quarter-full ones; every lane's 8
//   channels belong to one tap, so its tap shift is a per-lane constant and the K loop is unchanged);
//   4 waves as 2x2; wave tile = (BT/2)^2; K-step = 64 positions.
//   LDS row = BT channels (BT*2 bytes).  A transposing read is served per 32-lane half = 4 k-rows x 64 B; the 16-byte
//   chunk c of row r is stored at c ^ 4*(r & 3) (256-B rows) or c ^ 4*((r >> 1) & 1) (128-B rows, where row parity
//   already selects the half of the bank row), which puts those four 64-B pieces on four different quarters of the
//   256-B bank row.
namespace {
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

struct DW2Params {
  const uint16_t* x;
  const uint16_t* dz;
  float* dW;
  ConvGeom g;  // H, W, Cin, cstride (of x), Ho, Wo, KH, KW, stride, pad
  int Cout, K, kchunk, nmt, nct, Mtot;  // Mtot = taps * Cin filter rows, nmt tiles of them
  int rw, rh, qh;  // 64 positions = (qh*Ho + rh) rows + rw columns
};

template <int BT>
__global__ __launch_bounds__(256) void dw2_kernel(const DW2Params q) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int KP = 64, S = 3;
  constexpr int ROWB = BT * 2, CH = ROWB / 16, RPI = 64 / CH;
  constexpr int TILE_BYTES = KP * ROWB, STAGE = 2 * TILE_BYTES;
  constexpr int PER = KP / RPI / 4, LPS = 2 * PER;
  constexpr int WT = BT / 2, T = WT / 32;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const ConvGeom g = q.g;
  int id = blockIdx.x;
  const int ct = id % q.nct; id /= q.nct;
  const int mt = id % q.nmt;
  const int z = id / q.nmt;
  const int m0 = mt * BT, co0 = ct * BT;
  const int kbeg = z * q.kchunk;
  const int kend = min(q.K, kbeg + q.kchunk);
  if (kbeg >= kend) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane / CH, pch = lane % CH;
  // logical channel offset fetched by this lane (swizzle on the global side), see the bank note above
  const int cofs = (pch ^ (BT == 128 ? 4 * (lrow & 3) : 4 * ((lrow >> 1) & 1))) * 8;
  // this lane's 8 filter rows m0 + cofs .. + 7 = (tap, ci .. ci + 7): Cin % 8 == 0, so they never straddle two taps
  const int mrow = m0 + cofs;
  const bool xc_ok = mrow < q.Mtot, zc_ok = co0 + cofs < q.Cout;
  const int tap = xc_ok ? mrow / g.Cin : 0, ci_l = mrow - tap * g.Cin;
  const int dy = tap / g.KW - g.pad_h, dx = tap % g.KW - g.pad_w;
  const int npos = g.Ho * g.Wo;
  // Issuing a K-step must stay well below the MFMA time of the step (one wave per SIMD): the (image, row, column) of
  // each of this lane's positions is decoded with real divisions ONCE and then advanced by 64 positions per step with
  // carries only (64 = qw*Wo + rw, qw = qh*Ho + rh, host-computed); loads go through buffer descriptors whose range
  // check returns zeros for the out-of-range offset given to padding taps / tile overhang / positions >= kend.
  const auto rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(q.x), 0, 0x80000000, 0x00020000);
  const auto rsrcZ = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(q.dz), 0, 0x80000000, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  int pb[PER], ph[PER], pw[PER], ppos[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int pos = kbeg + (wave * PER + i) * RPI + lrow;
    ppos[i] = pos;
    pb[i] = pos / npos;
    const int rem = pos - pb[i] * npos;
    ph[i] = rem / g.Wo;
    pw[i] = rem - ph[i] * g.Wo;
  }
  const unsigned xcol = (unsigned)ci_l * 2u, zcol = (unsigned)(co0 + cofs) * 2u;

  // loads of the step the decoded positions currently stand at, then advance them by KP
  auto issue = [&](int sb) {
    unsigned char* xb = smem + sb * STAGE + wave * PER * 1024;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const bool ok = ppos[i] < kend;
      const int hi = ph[i] * g.stride + dy, wi = pw[i] * g.stride + dx;
      const bool okx = ok && xc_ok && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
      const unsigned xo = okx ? (unsigned)(((pb[i] * g.H + hi) * g.W + wi) * g.cstride) * 2u + xcol : OOB;
      const unsigned zo = (ok && zc_ok) ? (unsigned)ppos[i] * (unsigned)q.Cout * 2u + zcol : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr_t)(xb + i * 1024), 16, xo, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcZ, (lds_ptr_t)(xb + TILE_BYTES + i * 1024), 16, zo, 0, 0, 0);
      ppos[i] += KP;
      pw[i] += q.rw;
      const int c1 = pw[i] >= g.Wo ? 1 : 0;
      pw[i] -= c1 * g.Wo;
      ph[i] += q.rh + c1;
      const int c2 = ph[i] >= g.Ho ? 1 : 0;
      ph[i] -= c2 * g.Ho;
      pb[i] += q.qh + c2;
    }
  };

  f32x16 acc[T][T];
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  const int nsteps = (kend - kbeg + KP - 1) / KP;
  issue(0);
  issue(1);
  const int wm = wave >> 1, wn = wave & 1;
  // transposing-read lane roles: 16-lane group gq, row q and column quad p inside the group
  const int gq = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int krow = 8 * (gq >> 1) + qq;             // + 16*k16 + 4*s
  const int col0 = 16 * (gq & 1) + 4 * pp;         // + tile column base; in elements
  const int swz = BT == 128 ? 4 * qq : 4 * ((qq >> 1) & 1);  // every row this lane addresses has row & 3 == qq
  int aoff[T], boff[T];
#pragma unroll
  for (int a = 0; a < T; ++a) {
    const int bc = (wm * WT + a * 32 + col0) * 2;
    aoff[a] = krow * ROWB + (((bc >> 4) ^ swz) << 4) + (bc & 15);
    const int bn = (wn * WT + a * 32 + col0) * 2;
    boff[a] = TILE_BYTES + krow * ROWB + (((bn >> 4) ^ swz) << 4) + (bn & 15);
  }
  for (int t = 0; t < nsteps; ++t) {
    wait_vmcnt<LPS>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue((t + 2) % S);
    const unsigned char* st = smem + (t % S) * STAGE;
#pragma unroll
    for (int k16 = 0; k16 < KP / 16; ++k16) {
      bf16x8 af[T], bfr[T];
#pragma unroll
      for (int a = 0; a < T; ++a) {
        const unsigned char* pa = st + aoff[a] + k16 * 16 * ROWB;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(pa));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(pa + 4 * ROWB));
        const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        af[a] = __builtin_bit_cast(bf16x8, v);
        const unsigned char* pb = st + boff[a] + k16 * 16 * ROWB;
        const s16x4_t lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(pb));
        const s16x4_t hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(pb + 4 * ROWB));
        const s16x8_t v2 = {lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
        bfr[a] = __builtin_bit_cast(bf16x8, v2);
      }
#pragma unroll
      for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
  }
  wait_vmcnt<0>();
  // epilogue: fp32 atomics into dW[m = (tap, ci)][co]; lanes 0..31 cover 32 consecutive co (128 B)
  const int h = lane >> 5, r = lane & 31;
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b) {
      const int co = co0 + wn * WT + b * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = m0 + wm * WT + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
#ifdef DTF_DW2_PLAIN_STORES  // (timing experiment only: wrong sums, shows what the atomics cost)
        if (m < q.Mtot && co < q.Cout) q.dW[(long)m * q.Cout + co] = acc[a][b][i];
#else
        if (m < q.Mtot && co < q.Cout) atomicAdd(&q.dW[(long)m * q.Cout + co], acc[a][b][i]);
#endif
      }
    }
#endif
}

template <int BT>
bool raise_dw_lds_limit() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&dw2_kernel<BT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             3 * 2 * 64 * BT * 2) == hipSuccess;
}
}  // namespace

DTF_API int dtf_igemm2_dw_init() { return (raise_dw_lds_limit<64>() && raise_dw_lds_limit<128>()) ? 0 : -3; }

DTF_API int dtf_conv_dw_v2(const void* x, int in_cstride, const void* dz, float* dW, int B, int H, int W, int Cin, int Cout,
                           int KH, int KW, int stride, int pad_h, int pad_w, int Ho, int Wo, int splits, int variant,
                           hipStream_t st) {
  if (!g_ready) return -2;
  if (B <= 0 || Cin % 8 || Cout % 8 || in_cstride % 8 || splits < 1 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || stride < 1)
    return ERR_SHAPE;
  if ((H + 2 * pad_h - KH) / stride + 1 != Ho || (W + 2 * pad_w - KW) / stride + 1 != Wo) return ERR_SHAPE;
  DW2Params q{};
  q.x = (const uint16_t*)x; q.dz = (const uint16_t*)dz; q.dW = dW;
  q.g.H = H; q.g.W = W; q.g.Cin = Cin; q.g.cstride = in_cstride; q.g.Ho = Ho; q.g.Wo = Wo;
  q.g.KH = KH; q.g.KW = KW; q.g.stride = stride; q.g.pad_h = pad_h; q.g.pad_w = pad_w;
  q.Cout = Cout; q.K = B * Ho * Wo; q.Mtot = KH * KW * Cin;
  q.rw = 64 % Wo;
  q.rh = (64 / Wo) % Ho;
  q.qh = (64 / Wo) / Ho;
  if ((long)B * H * W * in_cstride * 2 >= 0x7fffffffL || (long)q.K * Cout * 2 >= 0x7fffffffL) return ERR_SHAPE;
  const int ksteps = dtf_cdiv(q.K, 64);
  q.kchunk = dtf_cdiv(ksteps, splits) * 64;
  const int nz = dtf_cdiv(q.K, q.kchunk);
  const bool big = Cin > 64 && Cout > 64 && !(variant & 1);  // variant 1: 64-wide tiles regardless
  const int BT = big ? 128 : 64;
  q.nmt = dtf_cdiv(q.Mtot, BT);
  q.nct = dtf_cdiv(Cout, BT);
  const long blocks = (long)nz * q.nmt * q.nct;
  if (blocks > 0x7fffffffL) return ERR_SHAPE;
  if (big)
    hipLaunchKernelGGL(dw2_kernel<128>, dim3((int)blocks), dim3(256), 3 * 2 * 64 * 128 * 2, st, q);
  else
    hipLaunchKernelGGL(dw2_kernel<64>, dim3((int)blocks), dim3(256), 3 * 2 * 64 * 64 * 2, st, q);
  DTF_CHECK_LAUNCH();
  return 0;
}
