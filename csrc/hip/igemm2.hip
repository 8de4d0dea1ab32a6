// Implicit-GEMM convolution core, second generation: asynchronous global->LDS loads (global_load_lds_dwordx4, no
// VGPR staging), a 3-stage LDS ring with ONE barrier per K-step, 64x64 wave tiles (4 MFMA 32x32x16 per 4 fragment
// reads), XOR-swizzled unpadded LDS rows, and an XCD-aware workgroup order.
//
//   C[M, N] = A_gather[M, K] * Wp[N, K]^T          bf16 operands, fp32 accumulate
//
// * A is gathered on the fly from a bf16 NHWC tensor (forward: im2col of the input; backward-data: the transposed
//   gather of the output gradient, stride 1 or 2).  Out-of-image taps, rows >= M and k >= K read a 16-byte zero page
//   instead of being predicated: every lane always issues its load, so the per-wave vmcnt arithmetic is static.
// * Wp is a bf16 [N][K] (K contiguous) packing of the fp32 master weights, refreshed once per step by
//   `dtf_pack_conv_weights` (87 MB read / 87 MB written for all of Inception-v3: ~40 us).
// * LDS stage = A tile [256 rows][128 B] + B tile [BN rows][128 B]; a row is 8 chunks of 16 B and chunk c of row r is
//   stored at chunk position c ^ (r & 7): the 8 lanes of a ds_read_b128 phase that read the same logical chunk of 8
//   consecutive rows hit 8 different chunk positions = all 32 banks once.  The swizzle is applied on the *global* side
//   (lane l of a load instruction fetches logical chunk (l & 7) ^ (l >> 3) of row l >> 3), because the LDS side of
//   global_load_lds is fixed: lane l writes 16 B at M0 + 16*l.
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
  const uint16_t* zeros;  // >= 16 bytes of zeros
  int mtiles, ntiles;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MODE 0: forward gather (row = output position, tap shifts into the input)
// MODE 1: backward-data gather (row = input position, tap shifts into the output gradient; stride 1 or 2)
template <int BN, int MODE, class EP>
__global__ __launch_bounds__(BN == 64 ? 256 : 512) void igemm2_kernel(const IG2Params q) {
  constexpr int BM = 256, BK = 64, S = 3;
  constexpr int NW = BN == 64 ? 4 : 8;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_PER = (BM / 8) / NW, B_PER = (BN / 8) / NW, LPS = A_PER + B_PER;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const GemmParams& p = q.gp;
  const ConvGeom g = p.g;

  const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
  const int nt = local % q.ntiles, mt = (local / q.ntiles) * 8 + xcd;
  if (mt >= q.mtiles) return;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane >> 3;
  const int kc8 = ((lane & 7) ^ lrow) * 8;  // this lane's logical k offset inside a K-step (swizzle on the global side)

  // ---- per-lane row descriptors of the A gather
  int img[A_PER], hb[A_PER], wb[A_PER];
  bool okr[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int m = m0 + (wave * A_PER + i) * 8 + lrow;
    okr[i] = m < p.M;
    if (MODE == 0) {
      const int wo = m % g.Wo, t = m / g.Wo;
      img[i] = t / g.Ho;
      hb[i] = (t % g.Ho) * g.stride - g.pad_h;
      wb[i] = wo * g.stride - g.pad_w;
    } else {
      const int w = m % g.W, t = m / g.W;
      img[i] = t / g.H;
      hb[i] = t % g.H + g.pad_h;
      wb[i] = w + g.pad_w;
    }
  }
  const uint16_t* __restrict__ Ag = reinterpret_cast<const uint16_t*>(p.A);
  long brow[B_PER];
  bool okb[B_PER];
#pragma unroll
  for (int j = 0; j < B_PER; ++j) {
    const int n = n0 + (wave * B_PER + j) * 8 + lrow;
    okb[j] = n < p.N;
    brow[j] = (long)n * p.K;
  }

  auto issue = [&](int step, int sb) {
    const int kk = step * BK + kc8;
    const bool kok = kk < p.K;
    const int tap = kk / g.Cin, ci = kk - tap * g.Cin;
    const int dy = tap / g.KW, dx = tap - dy * g.KW;
    unsigned char* abase = smem + sb * STAGE + wave * A_PER * 1024;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const uint16_t* src;
      if (MODE == 0) {
        const int hi = hb[i] + dy, wi = wb[i] + dx;
        const bool ok = okr[i] && kok && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
        src = ok ? Ag + ((long)(img[i] * g.H + hi) * g.W + wi) * g.cstride + ci : q.zeros;
      } else {
        const int sh = g.stride - 1;
        const int hz = hb[i] - dy, wz = wb[i] - dx;
        const int ho = hz >> sh, wo = wz >> sh;
        const bool ok = okr[i] && kok && hz >= 0 && wz >= 0 && ((hz | wz) & sh) == 0 && ho < g.Ho && wo < g.Wo;
        src = ok ? Ag + ((long)(img[i] * g.Ho + ho) * g.Wo + wo) * g.cstride + ci : q.zeros;
      }
      __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(abase + i * 1024), 16, 0, 0);
    }
    unsigned char* bbase = smem + sb * STAGE + A_BYTES + wave * B_PER * 1024;
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      const uint16_t* src = (okb[j] && kok) ? q.Wp + brow[j] + kk : q.zeros;
      __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(bbase + j * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

  const int nsteps = (p.K + BK - 1) / BK;
  issue(0, 0);
  issue(1, 1);  // steps >= nsteps only fetch the zero page (k >= K)

  const int r = lane & 31, h = lane >> 5;
  const int wm = BN == 64 ? wave : (wave >> 1), wn = BN == 64 ? 0 : (wave & 1);
  const int sw = r & 7;
  for (int t = 0; t < nsteps; ++t) {
    wait_vmcnt<LPS>();  // this wave's loads of stage t have landed (stage t+1 may still be in flight)
    // Bare s_barrier, not __syncthreads(): the latter's fence makes hipcc emit s_waitcnt vmcnt(0), which would drain
    // the stage that is supposed to stay in flight.  After it everyone's stage-t data is in LDS and everyone is done
    // reading the stage consumed at step t-1 (its ds_reads were waited on before the MFMAs that used them).
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(t + 2, (t + 2) % S);
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
  wait_vmcnt<0>();  // drain the zero-page prefetches before the LDS is released
  EP::template run<2, 2>(p, acc, m0 + wm * 64, n0 + wn * 64, lane, 0);
}

template <int BN, int MODE, class EP>
int launch_igemm2(IG2Params& q, hipStream_t st) {
  constexpr int NTHREADS = BN == 64 ? 256 : 512;
  constexpr int LDS = 3 * (256 * 128 + BN * 128);
  q.mtiles = dtf_cdiv(q.gp.M, 256);
  q.ntiles = dtf_cdiv(q.gp.N, BN);
  const int mt8 = (q.mtiles + 7) / 8 * 8;
  hipLaunchKernelGGL((igemm2_kernel<BN, MODE, EP>), dim3(mt8 * q.ntiles), dim3(NTHREADS), LDS, st, q);
  DTF_CHECK_LAUNCH();
  return 0;
}

template <int BN, int MODE, class EP>
bool raise_lds_limit() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm2_kernel<BN, MODE, EP>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (256 * 128 + BN * 128)) == hipSuccess;
}

uint16_t* g_zero_page = nullptr;

inline int fill(IG2Params& q, const void* A, const void* Wp, void* C, int M, int N, int K, long ldc, int H, int W, int Cg,
                int cstride, int Ho, int Wo, int KH, int KW, int stride, int pad_h, int pad_w) {
  if (!g_zero_page) return -2;  // dtf_igemm2_init() was not called
  q = IG2Params{};
  GemmParams& p = q.gp;
  p.A = A; p.C = C; p.M = M; p.N = N; p.K = K; p.ldc = ldc;
  p.g.H = H; p.g.W = W; p.g.Cin = Cg; p.g.cstride = cstride; p.g.Ho = Ho; p.g.Wo = Wo;
  p.g.KH = KH; p.g.KW = KW; p.g.stride = stride; p.g.pad_h = pad_h; p.g.pad_w = pad_w;
  q.Wp = reinterpret_cast<const uint16_t*>(Wp);
  q.zeros = g_zero_page;
  if (M <= 0 || N <= 0 || K <= 0 || K % 8 || Cg % 8 || cstride % 8 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || KH <= 0 ||
      KW <= 0 || (stride != 1 && stride != 2))
    return ERR_SHAPE;
  if ((H + 2 * pad_h - KH) / stride + 1 != Ho || (W + 2 * pad_w - KW) / stride + 1 != Wo) return ERR_SHAPE;
  if ((long)M * 1 > 0x7fffffffL) return ERR_SHAPE;
  return 0;
}
}  // namespace

// One-time setup (allocates the zero page, raises the dynamic-LDS limit of every instantiation).  Not capturable.
DTF_API int dtf_igemm2_init() {
  if (!g_zero_page) {
    if (hipMalloc(reinterpret_cast<void**>(&g_zero_page), 256) != hipSuccess) return -1;
    if (hipMemset(g_zero_page, 0, 256) != hipSuccess) return -1;
    if (!raise_lds_limit<64, 0, EpBf16Stats>() || !raise_lds_limit<128, 0, EpBf16Stats>() ||
        !raise_lds_limit<64, 0, EpBiasReluBf16>() || !raise_lds_limit<128, 0, EpBiasReluBf16>() ||
        !raise_lds_limit<64, 1, EpBf16Acc>() || !raise_lds_limit<128, 1, EpBf16Acc>())
      return -3;
  }
  return 0;
}

// z[B,Ho,Wo,Cout] (bf16 dense) = conv(x, Wp) + BN statistics (see EpBf16Stats); Wp = bf16 [Cout][KH*KW*Cin]
DTF_API int dtf_conv_fwd_stats_v2(const void* x, const void* Wp, void* z, float* stats, int B, int H, int W, int Cin,
                                  int in_cstride, int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int Ho,
                                  int Wo, hipStream_t st) {
  IG2Params q;
  const int rc = fill(q, x, Wp, z, B * Ho * Wo, Cout, KH * KW * Cin, Cout, H, W, Cin, in_cstride, Ho, Wo, KH, KW, stride, pad_h, pad_w);
  if (rc) return rc;
  q.gp.aux_out = stats;
  if (Cout > 64) return launch_igemm2<128, 0, EpBf16Stats>(q, st);
  return launch_igemm2<64, 0, EpBf16Stats>(q, st);
}

// Inference conv: out = bf16(relu(conv(x, Wp) + bias)) written at channel stride out_cstride (concat slice)
DTF_API int dtf_conv_bias_relu_v2(const void* x, const void* Wp, const float* bias, void* out, int B, int H, int W, int Cin,
                                  int in_cstride, int Cout, int out_cstride, int KH, int KW, int stride, int pad_h, int pad_w,
                                  int Ho, int Wo, int relu, hipStream_t st) {
  IG2Params q;
  const int rc = fill(q, x, Wp, out, B * Ho * Wo, Cout, KH * KW * Cin, out_cstride, H, W, Cin, in_cstride, Ho, Wo, KH, KW, stride, pad_h, pad_w);
  if (rc) return rc;
  q.gp.bias = bias;
  q.gp.relu = relu;
  if (Cout > 64) return launch_igemm2<128, 0, EpBiasReluBf16>(q, st);
  return launch_igemm2<64, 0, EpBiasReluBf16>(q, st);
}

// dx[B,H,W,Cin] (bf16 at channel stride dx_cstride) (+)= conv^T(dz, Wq); Wq = bf16 [Cin][KH*KW*Cout]
DTF_API int dtf_conv_bwd_data_v2(const void* dz, const void* Wq, void* dx, int B, int H, int W, int Cin, int dx_cstride,
                                 int Cout, int KH, int KW, int stride, int pad_h, int pad_w, int Ho, int Wo, int accumulate,
                                 hipStream_t st) {
  IG2Params q;
  const int rc = fill(q, dz, Wq, dx, B * H * W, Cin, KH * KW * Cout, dx_cstride, H, W, Cout, Cout, Ho, Wo, KH, KW, stride, pad_h, pad_w);
  if (rc) return rc;
  q.gp.accumulate = accumulate;
  if (Cin > 64) return launch_igemm2<128, 1, EpBf16Acc>(q, st);
  return launch_igemm2<64, 1, EpBf16Acc>(q, st);
}

// ---------------------------------------------------------------------------------------------- weight packing
// Per conv: fp32 HWIO master W[tap][ci][co] -> Wp[co][tap][ci] (forward B operand) and Wq[ci][tap][co] (backward-data
// B operand), both bf16.  One launch covers every conv of the network: blockIdx.y = conv, descriptors in device memory.
struct PackDesc {
  long w_off, p_off, q_off;  // element offsets into the fp32 arena / the two bf16 pack buffers
  int taps, cin, cout, pad;
};

__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const float* __restrict__ arena, const PackDesc* __restrict__ descs,
                                                                uint16_t* __restrict__ Wp, uint16_t* __restrict__ Wq) {
  const PackDesc d = descs[blockIdx.y];
  const long n = (long)d.taps * d.cin * d.cout;
  const float* w = arena + d.w_off;
  const int K = d.taps * d.cin, Kq = d.taps * d.cout;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int co = (int)(i % d.cout);
    const long t = i / d.cout;
    const int ci = (int)(t % d.cin), tap = (int)(t / d.cin);
    const uint16_t v = f2bf(w[i]);
    Wp[d.p_off + (long)co * K + tap * d.cin + ci] = v;
    Wq[d.q_off + (long)ci * Kq + tap * d.cout + co] = v;
  }
}

DTF_API int dtf_pack_conv_weights(const float* arena, const void* descs, int nconv, void* Wp, void* Wq, hipStream_t st) {
  if (nconv <= 0) return ERR_SHAPE;
  hipLaunchKernelGGL(pack_conv_weights_kernel, dim3(64, nconv), dim3(256), 0, st, arena, (const PackDesc*)descs,
                     (uint16_t*)Wp, (uint16_t*)Wq);
  DTF_CHECK_LAUNCH();
  return 0;
}
