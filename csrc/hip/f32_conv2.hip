// Exact-fp32 kernels for the MNIST conv2 layer (14x14x32 -> 14x14x64, 5x5 SAME, 2x2 pool behind it): forward, data
// gradient and filter gradient on v_mfma_f32_16x16x4_f32 — the reference's precision (tf.nn.conv2d on float32,
// /root/reference/demo1/train.py:40-41,84).
//
// At 64 FLOP/clk/SIMD these kernels are bound by the MFMA pipe (2 GFLOP each at batch 100 = 12.8 us on all 1024 SIMDs), so
// what matters is that every SIMD gets the same number of MFMAs and never waits:
//  * forward: one workgroup per 20 pooling windows of the whole batch (245 workgroups at batch 100), 8 waves = (output-
//    channel half, input-channel quarter): every wave issues the same 500 MFMAs on 10 accumulator tiles.
//  * data gradient: the same shape, one workgroup per 80 consecutive input positions of the batch (245 workgroups), 8 waves
//    that split K (8 output channels of every tap each): 500 MFMAs on 10 accumulator tiles per wave.
//    In both, the halo'd image sits in LDS, a wave streams its own disjoint slice of the filter straight from L2 into
//    registers one filter row ahead (no barrier inside the k loop), and the partial sums are combined through LDS at the end.
//  * filter gradient: one workgroup per (4 images, half of the output channels, filter row), 8 waves = (k group, input-
//    channel tile); per image the k groups take the 49 position groups round-robin with a per-image rotation, so over the
//    4 images every wave issues the same number of MFMAs; 250 workgroups at batch 100.
// A fragments that run over channels use the freedom in the k order (f32_gemm.h): one ds_read_b64 / b128 feeds 2 / 4 MFMAs.
#include "common.h"
#include "conv1_dw_body.h"
#include "adam_body.h"
#include "shard_body.h"
#include "dtf_api.h"
#include "f32_gemm.h"

// -DDTF_CONV_STAMPS (diagnostic builds only, scripts/dbg): thread 0 of every workgroup records (shader clock, 100 MHz
// real-time clock) pairs at phase boundaries into g_conv_stamps[workgroup][32]
#ifdef DTF_CONV_STAMPS
__device__ unsigned long long* g_conv_stamps;
DTF_API int dtf_dbg_set_conv_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), &p, sizeof(p)); }
#define CSTAMP_AT(bid, i) do { if (threadIdx.x == 0 && g_conv_stamps) { g_conv_stamps[(bid) * 32 + 2 * (i)] = __builtin_amdgcn_s_memtime(); \
  g_conv_stamps[(bid) * 32 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define CSTAMP(i) CSTAMP_AT(blockIdx.x, i)
// per-wave: loop-end clock and HW_ID (SIMD / CU placement) at g_conv_stamps[workgroup][32 .. 47] / [48 .. 63]
#define CSTAMP_WAVE() do { if ((threadIdx.x & 63) == 0 && g_conv_stamps) { \
  g_conv_stamps[gridDim.x * 32 + blockIdx.x * 32 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime(); \
  g_conv_stamps[gridDim.x * 32 + blockIdx.x * 32 + 16 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); } } while (0)
#else
#define CSTAMP(i) do { } while (0)
#define CSTAMP_AT(bid, i) do { } while (0)
#define CSTAMP_WAVE() do { } while (0)
#endif

namespace {
constexpr int ERR_SHAPE = -22;
constexpr int XP = 36;                  // floats per input halo pixel (32 ch + 4 pad)
// Row pitch of the halo'd input image in the forward kernel.  Its A fragments are ds_read_b64 (banks mod 64, conflicts
// within 32-lane halves); a half reads the 16 pixels of 4 pooling windows = 8 columns x 2 rows.  With 9 four-bank units
// per pixel the 8 columns cover 8 different units (9 is odd) and the second row must add 8 units: 672 / 4 = 168 = 8 mod 16
// (18 * 36 = 648 gives 2: two-way conflicts on 6 of 16 pixels).  Windows that wrap to the next window row still collide.
constexpr int XROW = 672;
constexpr int XIMG = 18 * XROW;         // 12 096 floats
constexpr int DP = 68;                  // floats per dY halo pixel (64 ch + 4 pad)
constexpr int DIMG = 18 * 18 * DP;      // 22 032 floats
}  // namespace

// ------------------------------------------------------------------ forward + bias + ReLU + 2x2 max-pool
// One workgroup = 20 consecutive pooling windows (80 output positions = 5 MFMA row tiles) of the batch's 49*B windows x all
// 64 output channels; 49*B / 20 = 245 workgroups at batch 100, i.e. 96 % of the CUs get one (one workgroup per image half:
// 200).  A range of 20 windows touches at most two images; both halo'd images are staged.  8 waves, two per SIMD, so one
// wave's LDS-read latency and filter prefetch run in the shadow of the other's MFMAs: wave = (output-channel half, input-
// channel quarter); the four K slices of a half are combined through LDS at the end.
// grid ceil(49 B / 20); 512 threads.  x fp32 [B,14,14,32]; W fp32 HWIO [800][64]; out fp32 [B,49,64]; amax uint8 [B,49,64]
namespace {
constexpr int FWD_WIN = 20, FWD_MT = FWD_WIN / 4;
constexpr int FWD_RED2 = 8 * FWD_MT * 2 * 256;  // [wave][tile][lane][4]
constexpr int FWD_LDS2 = (2 * XIMG > FWD_RED2 ? 2 * XIMG : FWD_RED2) * 4;
}  // namespace

__global__ __launch_bounds__(512) void conv2_fwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            uint8_t* __restrict__ amax, const int B) {
  extern __shared__ __attribute__((aligned(16))) float smem32[];
  float* img = smem32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int nh = wave >> 2, c0 = 8 * (wave & 3);  // output channels [32 nh, 32 nh + 32), input channels [c0, c0 + 8) of every tap
  const int n0 = 32 * nh;
  const int nwin = 49 * B;
  const int gw0 = blockIdx.x * FWD_WIN;
  const int b0 = gw0 / 49;
  const int glast = min(gw0 + FWD_WIN, nwin) - 1;
  const bool two = glast / 49 != b0;  // (workgroup-uniform)
  CSTAMP(0);

  // B fragments of tap t: element (j, nt) = W[t*32 + c0 + 2q + j][n0 + 16 nt + r]
  const float* wbase = W + (long)(c0 + 2 * q) * 64 + n0 + r;
  // A whole filter row (5 taps) is fetched one row ahead, pinned with a scheduling barrier: left to itself the compiler
  // sinks a one-tap-ahead prefetch to ~10 MFMAs before its use, and every tap then waits for L2 (45 instead of 32 cycles
  // per MFMA measured with two waves per SIMD).
  float bcur[5][4], bnxt[5][4];
#pragma unroll
  for (int dx = 0; dx < 5; ++dx)
#pragma unroll
    for (int e = 0; e < 4; ++e) bcur[dx][e] = wbase[(long)dx * 2048 + (e >> 1) * 64 + (e & 1) * 16];

  // ---- the halo'd image(s): 324 pixels x 8 float4 of payload each, halo -> zero
  f32x4 iv[2][6];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (s == 1 && !two) break;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int c = tid + 512 * i;
      const int pix = c >> 3, part = c & 7;
      const int hy = pix / 18, hx = pix - hy * 18;
      const int y = hy - 2, xx = hx - 2;
      const bool ok = c < 2592 && (unsigned)y < 14u && (unsigned)xx < 14u;
      iv[s][i] = *reinterpret_cast<const f32x4*>(x + (ok ? (((long)(b0 + s) * 14 + y) * 14 + xx) * 32 + part * 4 : 0L));
      if (!ok) iv[s][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (s == 1 && !two) break;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int c = tid + 512 * i;
      const int pix = c >> 3, hy = pix / 18, hx = pix - hy * 18;
      if (c < 2592) *reinterpret_cast<f32x4*>(img + s * XIMG + hy * XROW + hx * XP + (c & 7) * 4) = iv[s][i];
    }
  }
  __syncthreads();
  CSTAMP(1);

  // rows in pool-window order: m = 4*window + sub, so a lane's four accumulator registers are one pooling window
  int abase[FWD_MT];
#pragma unroll
  for (int mt = 0; mt < FWD_MT; ++mt) {
    int gw = gw0 + mt * 4 + (r >> 2);
    gw = gw < nwin ? gw : nwin - 1;  // padding windows of the last workgroup; their results are discarded
    const int slot = gw / 49 - b0, win = gw - (b0 + slot) * 49, sub = r & 3;
    const int py = win / 7, px = win - py * 7;
    abase[mt] = slot * XIMG + (2 * py + (sub >> 1)) * XROW + (2 * px + (sub & 1)) * XP + c0 + 2 * q;
  }
  f32x4 acc[FWD_MT][2];
#pragma unroll
  for (int mt = 0; mt < FWD_MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A fragments are read one tap ahead as well, so a wave's own instruction stream never waits for LDS between two taps
  // (with "oldest wave first" issue the younger wave of a SIMD only fills the older one's gaps, and runs alone at the end)
  float2 af[FWD_MT], afn[FWD_MT];
#pragma unroll
  for (int mt = 0; mt < FWD_MT; ++mt) af[mt] = *reinterpret_cast<const float2*>(img + abase[mt]);
#pragma unroll 1
  for (int dy = 0; dy < 5; ++dy) {
    const int dyn = dy < 4 ? dy + 1 : 4;  // (the last prefetch re-reads row 4)
#pragma unroll
    for (int dx = 0; dx < 5; ++dx)
#pragma unroll
      for (int e = 0; e < 4; ++e) bnxt[dx][e] = wbase[(long)(dyn * 5 + dx) * 2048 + (e >> 1) * 64 + (e & 1) * 16];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      // next tap in (dy, dx) order; after the last one the read is a harmless repeat of tap (4, 4)
      const int toffn = dx < 4 ? dy * XROW + (dx + 1) * XP : (dy < 4 ? dy + 1 : 4) * XROW + (dy < 4 ? 0 : 4) * XP;
#pragma unroll
      for (int mt = 0; mt < FWD_MT; ++mt) afn[mt] = *reinterpret_cast<const float2*>(img + abase[mt] + toffn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < FWD_MT; ++mt) {
        acc[mt][0] = mfma16f32(af[mt].x, bcur[dx][0], acc[mt][0]);
        acc[mt][1] = mfma16f32(af[mt].x, bcur[dx][1], acc[mt][1]);
      }
#pragma unroll
      for (int mt = 0; mt < FWD_MT; ++mt) {
        acc[mt][0] = mfma16f32(af[mt].y, bcur[dx][2], acc[mt][0]);
        acc[mt][1] = mfma16f32(af[mt].y, bcur[dx][3], acc[mt][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < FWD_MT; ++mt) af[mt] = afn[mt];
    }
#pragma unroll
    for (int dx = 0; dx < 5; ++dx)
#pragma unroll
      for (int e = 0; e < 4; ++e) bcur[dx][e] = bnxt[dx][e];
  }

  // ---- combine the four K slices of each channel half through LDS (the images are dead), then bias + ReLU + pool on
  // register quads; tile id = (nh * FWD_MT + mt) * 2 + nt, finalised by wave (id % 8)
  CSTAMP(2);
  CSTAMP_WAVE();
  __syncthreads();
  CSTAMP(3);
  float* red = smem32;
  constexpr int TPW = FWD_MT * 2;  // tiles per wave
#pragma unroll
  for (int mt = 0; mt < FWD_MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      *reinterpret_cast<f32x4*>(red + ((wave * TPW + mt * 2 + nt) * 64 + lane) * 4) = acc[mt][nt];
  __syncthreads();
  for (int id = wave; id < 2 * TPW; id += 8) {
    const int h = id / TPW, t = id - h * TPW;
    f32x4 s = *reinterpret_cast<const f32x4*>(red + (((4 * h) * TPW + t) * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(red + (((4 * h + w) * TPW + t) * 64 + lane) * 4);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    const int mt = t >> 1, nt = t & 1;
    const int gw = gw0 + mt * 4 + q, col = 32 * h + nt * 16 + r;
    float best = s[0];
    int arg = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j)
      if (s[j] > best) { best = s[j]; arg = j; }
    if (gw < nwin) {
      const long o = (long)gw * 64 + col;
      out[o] = fmaxf(best + bias[col], 0.f);
      if (amax) amax[o] = (uint8_t)arg;
    }
  }
  CSTAMP(4);
}

// ------------------------------------------------------------------ un-pool helpers (fp32 activations)
namespace {
// task = (window 0..48, 4-channel group): expands the pooled gradient into the four positions of its window
struct UnpoolTask32 {
  f32x4 g, pv;
  uint32_t av;
  __device__ __forceinline__ void load(const float* __restrict__ dpool, const float* __restrict__ pooled,
                                       const uint8_t* __restrict__ amax, long off) {
    g = *reinterpret_cast<const f32x4*>(dpool + off);
    pv = *reinterpret_cast<const f32x4*>(pooled + off);
    av = *reinterpret_cast<const uint32_t*>(amax + off);
  }
  // ReLU mask of the pooled activation, applied in place; returns nothing
  __device__ __forceinline__ void mask() {
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = pv[j] > 0.f ? g[j] : 0.f;
  }
  // writes the 4 positions of window (py, px) at dst + ((2py + sy + oy)*rowpix + 2px + sx + ox)*pitch
  __device__ __forceinline__ void store(float* dst, int py, int px, int rowpix, int pitch, int oy, int ox) const {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = ((av >> (8 * j)) & 0xffu) == (uint32_t)s ? g[j] : 0.f;
      *reinterpret_cast<f32x4*>(dst + ((2 * py + (s >> 1) + oy) * rowpix + 2 * px + (s & 1) + ox) * pitch) = o;
    }
  }
};
}  // namespace

// ------------------------------------------------------------------ data gradient + un-pool + ReLU mask
// Same shape as the forward: one workgroup = 80 consecutive input positions (5 MFMA row tiles) of the batch's 196*B x all 32
// input channels; 245 workgroups at batch 100.  All 8 waves split K: wave w takes output channels [8w, 8w + 8) of every tap
// (two k-steps per tap, 500 MFMAs on 10 accumulator tiles, like the forward); the eight partial sums are combined through
// LDS.  80 positions touch at most 7 image rows of at most two images: only the halo rows those need (<= 15 rows of
// 18 x 68 floats) are un-pooled into LDS.  The bias gradient is produced by the filter-gradient launch, which stages dY
// per image anyway.
// grid ceil(196 B / 80); 512 threads.  dpool2 fp32 [B,49,64]; pool2 fp32 [B,49,64]; amax2 u8 [B,49,64]; W fp32 HWIO
// [25][32][64]; pool1 fp32 [B,196,32] (ReLU mask of the layer below); dpool1 fp32 [B,196,32]
namespace {
constexpr int BD_POS = 80, BD_MT = BD_POS / 16;
constexpr int BD_ROWS = 15;
// Row pitch of the un-pooled dY halo image: the A fragments are ds_read_b64 of 16 consecutive positions (17 four-bank units
// per pixel: consecutive pixels take consecutive units); a tile that runs over a row end continues 4 halo pixels further,
// which the row pitch must turn into "the next unit": (pitch / 4) mod 16 = 14 * 17 mod 16 = 14 -> 1272 (18 * 68 = 1224 gives 2).
constexpr int BD_ROWP = 1272;
constexpr int BD_IMG = BD_ROWS * BD_ROWP;        // 19 080 floats
constexpr int BD_RED = 8 * BD_MT * 2 * 256;      // [wave][tile][lane][4]: 20 480 floats
constexpr int BD_LDS = (BD_IMG > BD_RED ? BD_IMG : BD_RED) * 4;
}  // namespace

__device__ __forceinline__ void conv2_bwd_data_f32_body(float* __restrict__ smem32, const int bid,
                                                        const float* __restrict__ dpool2, const float* __restrict__ pool2,
                                                        const uint8_t* __restrict__ amax2, const float* __restrict__ W,
                                                        const float* __restrict__ pool1, float* __restrict__ dpool1,
                                                        const int B) {
  float* dimg = smem32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int k0 = 8 * wave;  // this wave's output channels [k0, k0 + 8) of every tap
  const int npos = 196 * B;
  const int P0 = bid * BD_POS;
  const int b0 = P0 / 196;
  const int plast = min(P0 + BD_POS, npos) - 1;
  const bool two = plast / 196 != b0;  // (workgroup-uniform, like everything up to the task decode)
  CSTAMP(0);
  // slot 0 = image b0, output rows [ylo0, yhi0]; slot 1 = image b0 + 1, output rows [0, yhi1].  Output row y reads halo
  // rows y .. y + 4 (halo row = image row + 2): slot 0 stages halo rows [ylo0, yhi0 + 4] at LDS rows 0.., slot 1 halo rows
  // [0, yhi1 + 4] behind them.
  const int ylo0 = (P0 - b0 * 196) / 14;
  const int yhi0 = two ? 13 : (plast - b0 * 196) / 14;
  const int yhi1 = two ? (plast - (b0 + 1) * 196) / 14 : -5;
  const int nrow0 = yhi0 - ylo0 + 5, nrow1 = yhi1 + 5;
  // window rows whose image rows fall into the staged halo rows
  const int py0lo = max(ylo0 - 2, 0) >> 1, py0hi = min(yhi0 + 2, 13) >> 1;
  const int nwr0 = py0hi - py0lo + 1;
  const int nwr1 = two ? (min(yhi1 + 2, 13) >> 1) + 1 : 0;
  const int ntask = (nwr0 + nwr1) * 112;  // (window row, px 0..6, 4-channel group 0..15); <= 6 window rows in all

  // B fragments of (flipped) tap t': elements j = W[24 - t'][16 nt + r][k0 + 2q + j]
  const float* wbase = W + (long)r * 64 + k0 + 2 * q;
  // (a whole filter row one row ahead, pinned: see conv2_fwd_f32_kernel)
  float2 bcur[5][2], bnxt[5][2];
#pragma unroll
  for (int dx = 0; dx < 5; ++dx)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bcur[dx][nt] = *reinterpret_cast<const float2*>(wbase + (long)(24 - dx) * 2048 + nt * 1024);

  UnpoolTask32 ut[2];
  int trow[2], tcol[2], tslot[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int task = tid + 512 * i;
    const int t = task < ntask ? task : 0;
    const int wr = t / 112, rem = t - wr * 112;
    const int slot = wr >= nwr0 ? 1 : 0;
    const int py = slot ? wr - nwr0 : py0lo + wr;
    const int px = rem >> 4;
    tslot[i] = slot; trow[i] = py; tcol[i] = px;
    ut[i].load(dpool2, pool2, amax2, ((long)(b0 + slot) * 49 + py * 7 + px) * 64 + (rem & 15) * 4);
  }
  CSTAMP(5);
  {
    const int n4 = (nrow0 + nrow1) * (BD_ROWP / 4);
    for (int c = tid; c < n4; c += 512) *reinterpret_cast<f32x4*>(dimg + c * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  CSTAMP(6);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int task = tid + 512 * i;
    if (task < ntask) {
      ut[i].mask();
      const int slot = tslot[i];
      const int hlo = slot ? 0 : ylo0, hhi = slot ? yhi1 + 4 : yhi0 + 4, rbase = slot ? nrow0 : -ylo0;
      float* dst = dimg + ((task - (task / 112) * 112) & 15) * 4;
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        const int hy = 2 * trow[i] + (sp >> 1) + 2;
        if (hy >= hlo && hy <= hhi) {
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = ((ut[i].av >> (8 * j)) & 0xffu) == (uint32_t)sp ? ut[i].g[j] : 0.f;
          *reinterpret_cast<f32x4*>(dst + (rbase + hy) * BD_ROWP + (2 * tcol[i] + (sp & 1) + 2) * DP) = o;
        }
      }
    }
  }

  CSTAMP(7);
  int abase[BD_MT];
#pragma unroll
  for (int mt = 0; mt < BD_MT; ++mt) {
    const int p = min(P0 + mt * 16 + r, plast);  // (padding rows of the last workgroup repeat its last position)
    const int slot = p >= (b0 + 1) * 196 ? 1 : 0;
    const int pos = p - (b0 + slot) * 196;
    const int y = pos / 14, xx = pos - y * 14;
    abase[mt] = ((slot ? nrow0 : -ylo0) + y) * BD_ROWP + xx * DP + k0 + 2 * q;
  }
  // ReLU mask of the layer below for the (at most two) tiles this wave finalises, fetched ahead of the k loop
  float pm[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = wave + 8 * u;
    const int mt = t >> 1, nt = t & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // (clamped address + select: a conditional load becomes a branch with its own s_waitcnt, and the eight of them a
      // chain of eight dependent round trips to L2 in the prologue: 12.4k cycles before the k loop, measured)
      const int p = P0 + mt * 16 + 4 * q + i;
      const bool ok = t < 2 * BD_MT && p < npos;
      const float v = pool1[ok ? (long)p * 32 + nt * 16 + r : 0L];
      pm[u][i] = ok ? v : 0.f;
    }
  }
  f32x4 acc[BD_MT][2];
#pragma unroll
  for (int mt = 0; mt < BD_MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  CSTAMP(1);

  // (A fragments one tap ahead: see conv2_fwd_f32_kernel)
  float2 af[BD_MT], afn[BD_MT];
#pragma unroll
  for (int mt = 0; mt < BD_MT; ++mt) af[mt] = *reinterpret_cast<const float2*>(dimg + abase[mt]);
#pragma unroll 1
  for (int dy = 0; dy < 5; ++dy) {
    const int dyn = dy < 4 ? dy + 1 : 4;
#pragma unroll
    for (int dx = 0; dx < 5; ++dx)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        bnxt[dx][nt] = *reinterpret_cast<const float2*>(wbase + (long)(24 - (dyn * 5 + dx)) * 2048 + nt * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      const int toffn = dx < 4 ? dy * BD_ROWP + (dx + 1) * DP : (dy < 4 ? dy + 1 : 4) * BD_ROWP + (dy < 4 ? 0 : 4) * DP;
#pragma unroll
      for (int mt = 0; mt < BD_MT; ++mt) afn[mt] = *reinterpret_cast<const float2*>(dimg + abase[mt] + toffn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < BD_MT; ++mt) {
        acc[mt][0] = mfma16f32(af[mt].x, bcur[dx][0].x, acc[mt][0]);
        acc[mt][1] = mfma16f32(af[mt].x, bcur[dx][1].x, acc[mt][1]);
      }
#pragma unroll
      for (int mt = 0; mt < BD_MT; ++mt) {
        acc[mt][0] = mfma16f32(af[mt].y, bcur[dx][0].y, acc[mt][0]);
        acc[mt][1] = mfma16f32(af[mt].y, bcur[dx][1].y, acc[mt][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < BD_MT; ++mt) af[mt] = afn[mt];
    }
#pragma unroll
    for (int dx = 0; dx < 5; ++dx)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bcur[dx][nt] = bnxt[dx][nt];
  }

  // ---- combine the eight K slices through LDS (the staged rows are dead); tile id = mt * 2 + nt, finalised by wave id % 8
  CSTAMP(2);
  CSTAMP_WAVE();
  __syncthreads();
  CSTAMP(3);
  float* red = smem32;
  constexpr int TPW = BD_MT * 2;
#pragma unroll
  for (int mt = 0; mt < BD_MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      *reinterpret_cast<f32x4*>(red + ((wave * TPW + mt * 2 + nt) * 64 + lane) * 4) = acc[mt][nt];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = wave + 8 * u;
    if (t < TPW) {
      f32x4 s = *reinterpret_cast<const f32x4*>(red + (t * 64 + lane) * 4);
#pragma unroll
      for (int w = 1; w < 8; ++w) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((w * TPW + t) * 64 + lane) * 4);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
      }
      const int mt = t >> 1, nt = t & 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = P0 + mt * 16 + 4 * q + i;
        if (p < npos) dpool1[(long)p * 32 + nt * 16 + r] = pm[u][i] > 0.f ? s[i] : 0.f;
      }
    }
  }
  CSTAMP(4);
}

// The data-gradient workgroups plus, optionally, trailing optimizer workgroups (W_fc1's Adam update: its gradient is final
// since the fc backward launch).  80 KB of LDS per workgroup = two per CU, so the 245 MFMA-bound workgroups leave 267 resident
// slots; the bandwidth-bound update runs in them instead of next to the (longer) filter-gradient launch.
struct ConvBwdData32 {
  const float* dpool2; const float* pool2; const uint8_t* amax2; const float* W; const float* pool1; float* dpool1;
  int B, nbd;
  OptimParams adam;  // adam.n == 0: no optimizer workgroups
  int nadam;
  ShardPush push;    // "shard" data parallelism (shard_body.h): this rank's slice of the fc1 filter gradient leaves for every
  int npush;         // rank's landing buffer from npush trailing workgroups (link-bound: they only occupy resident slots)
};

__global__ __launch_bounds__(512) void conv2_bwd_data_f32_kernel(const ConvBwdData32 a) {
  extern __shared__ __attribute__((aligned(16))) float smem32[];
  const int id = blockIdx.x;
  if (id < a.nbd)
    conv2_bwd_data_f32_body(smem32, id, a.dpool2, a.pool2, a.amax2, a.W, a.pool1, a.dpool1, a.B);
  else if (id < a.nbd + a.nadam) {
    if (a.adam.wsh_world > 0)  // (launch-uniform) "wshard": a share of this rank's rows, stored into every rank's next buffer
      adam_tf_body<false, true>(a.adam, id - a.nbd, a.nadam, 512);
    else
      adam_tf_body(a.adam, id - a.nbd, a.nadam, 512);
  } else
    shard_push_body(a.push, id - a.nbd - a.nadam, a.npush, 512);
}

// ------------------------------------------------------------------ filter gradient
// block -> (image quad g, output-channel half c, filter row dy); 512 threads.
// pool1 fp32 [B,196,32] (the conv input); dW fp32 [25][32][64] accumulated with atomics.
namespace {
// (Round 3 measured a conflict-free layout here -- 48-float pixels and parity-swizzled dY halves for ds_read_b32's 32 banks:
// bank-conflict share 40.7 -> 15.3 %, launch 35.1 -> 36.5 us, step +1.4 us on the same box.  The 36-float pitch stays.)
constexpr int DW_XP = XP;
constexpr int DW_XROWS = 14 * 18 * DW_XP; // the 14 halo rows a filter row touches: 9072 floats
constexpr int DW_DYP = XP;                // dY of one channel half, position-major, 36-float pitch
constexpr int DW_DY = 196 * DW_DYP;       // 7056 floats
constexpr int DW_RED = 8 * 10 * 256;     // [wave][tile][lane][4]: 20 480 floats
constexpr int DW_LDS = (DW_XROWS + DW_DY > DW_RED ? DW_XROWS + DW_DY : DW_RED) * 4;

__device__ __forceinline__ void conv2_dw_f32_body(float* __restrict__ smem, const int id, const float* __restrict__ pool1,
                                                  const float* __restrict__ dpool2, const float* __restrict__ pool2,
                                                  const uint8_t* __restrict__ amax2, float* __restrict__ dW,
                                                  float* __restrict__ db2, int B) {
  // 512 threads: wave = (k group kq = wave & 3, input-channel tile mt = wave >> 2); two waves per SIMD, so one wave's LDS
  // reads and address arithmetic run in the shadow of the other's MFMAs (with four waves the loop ran at 51 cycles per MFMA)
  float* ximg = smem;
  float* dyi = smem + DW_XROWS;
  const int dy = id % 5, c = (id / 5) & 1, g = id / 10;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int kq = wave & 3, mt = wave >> 2;
  const int b0 = 4 * g;
  const int nimg = min(4, B - b0);

  f32x4 xv[4];
  UnpoolTask32 ut;
  f32x4 gsum = {0.f, 0.f, 0.f, 0.f};  // bias gradient of this thread's channel group (filter row 0's workgroups only)
  const bool bias_role = db2 && dy == 0;
  auto load = [&](const int b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = tid + 512 * i;  // 2016 chunks: (staged row 0..13, halo col 0..17, part 0..7)
      const int pix = ch >> 3, part = ch & 7;
      const int sy = pix / 18, hx = pix - sy * 18;
      const int y = sy + dy - 2, xx = hx - 2;
      const bool ok = ch < 2016 && (unsigned)y < 14u && (unsigned)xx < 14u;
      xv[i] = *reinterpret_cast<const f32x4*>(pool1 + (ok ? (((long)b * 14 + y) * 14 + xx) * 32 + part * 4 : 0L));
      if (!ok) xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int t = tid < 392 ? tid : 0;  // 392 tasks: window = task >> 3, channel group (of this half) = task & 7
    ut.load(dpool2, pool2, amax2, (long)b * 3136 + (t >> 3) * 64 + c * 32 + (t & 7) * 4);
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = tid + 512 * i;
      if (ch < 2016) *reinterpret_cast<f32x4*>(ximg + (ch >> 3) * DW_XP + (ch & 7) * 4) = xv[i];
    }
    if (tid < 392) {
      ut.mask();
      const int win = tid >> 3, py = win / 7, px = win - py * 7;
      ut.store(dyi + (tid & 7) * 4, py, px, 14, DW_DYP, 0, 0);
      if (bias_role) { gsum[0] += ut.g[0]; gsum[1] += ut.g[1]; gsum[2] += ut.g[2]; gsum[3] += ut.g[3]; }
    }
  };

  f32x4 acc[5][2];  // [dx][cout tile] of input-channel tile mt
#pragma unroll
  for (int dx = 0; dx < 5; ++dx)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[dx][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  CSTAMP(0);
  if (nimg > 0) load(b0);
  for (int im = 0; im < nimg; ++im) {
    __syncthreads();  // the previous image has been consumed
    store();
    __syncthreads();
    if (im == 0) CSTAMP(1);
    if (im == 1) CSTAMP(2);
    if (im + 1 < nimg) load(b0 + im + 1);
    // k-step s covers positions 4s .. 4s+3 (lane quarter q <-> position 4s + q); k group (s + im) & 3 takes step s, so
    // over the four images every wave issues the same number of MFMAs
    for (int s = (kq - im) & 3; s < 49; s += 4) {
      const int pos = 4 * s + q;
      const int y = pos / 14, xx = pos - y * 14;
      const float* xa = ximg + (y * 18 + xx) * DW_XP + mt * 16 + r;  // tap (dy, dx): staged row y, halo column xx + dx
      const float* da = dyi + pos * DW_DYP + r;
      const float b0f = da[0], b1f = da[16];
      float a[5];
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) a[dx] = xa[dx * DW_XP];
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        acc[dx][0] = mfma16f32(a[dx], b0f, acc[dx][0]);
        acc[dx][1] = mfma16f32(a[dx], b1f, acc[dx][1]);
      }
    }
  }
  // bias gradient (workgroup-uniform branch): threads with equal (tid & 7) hold the same 4 channels of this half; the
  // waves' sums meet in LDS so the workgroup issues ONE 32-lane atomic (per-wave atomics on these two cache lines from
  // every workgroup queue up at the L2 and hold back the filter-gradient atomics behind them)
  if (bias_role) {
    __syncthreads();  // the staged image is dead
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = gsum[j];
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane < 8) smem[wave * 32 + lane * 4 + j] = v;
    }
    __syncthreads();
    if (tid < 32) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += smem[w * 32 + tid];
      atomicAdd(&db2[c * 32 + tid], v);
    }
  }
  // combine the four k groups' partial sums, then one atomic per element and workgroup
  CSTAMP(3);
  __syncthreads();
  CSTAMP(4);
  float* red = smem;
#pragma unroll
  for (int dx = 0; dx < 5; ++dx)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      *reinterpret_cast<f32x4*>(red + ((wave * 10 + dx * 2 + nt) * 64 + lane) * 4) = acc[dx][nt];
  __syncthreads();
  for (int t = wave; t < 20; t += 8) {  // output tile t = (dx, cin tile, cout tile)
    const int dx = t >> 2, mtt = (t >> 1) & 1, nt = t & 1;
    f32x4 s = *reinterpret_cast<const f32x4*>(red + (((mtt * 4) * 10 + dx * 2 + nt) * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(red + (((mtt * 4 + w) * 10 + dx * 2 + nt) * 64 + lane) * 4);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cin = mtt * 16 + 4 * q + i, cout = c * 32 + nt * 16 + r;
#ifdef DTF_DW_PLAIN_STORES  // (timing experiment only: wrong sums, shows what the atomics cost)
      dW[((long)(dy * 5 + dx) * 32 + cin) * 64 + cout] = s[i];
#else
      atomicAdd(&dW[((long)(dy * 5 + dx) * 32 + cin) * 64 + cout], s[i]);
#endif
    }
  }
  CSTAMP(5);
}

// conv2 dW (workgroups [0, n2)), conv1 dW/db (the next n1) and optionally trailing optimizer workgroups (see ConvDwDuo in
// conv2_lds.hip) in one launch.
struct ConvDwDuo32 {
  const float* pool1; const float* dpool2; const float* pool2; const uint8_t* amax2; float* dW2; float* db2;
  int B, n2;
  const float* images; const int* perm; const int* step_ptr; int spe;
  const float* dpool1; const uint8_t* amax1; float* dW1; float* db1;
  int n1;
  OptimParams adam;  // adam.n == 0: no optimizer workgroups
  int nadam;
};
constexpr int DUO32_LDS = DW_LDS > CONV1_DW_LDS_FLOATS * 4 ? DW_LDS : CONV1_DW_LDS_FLOATS * 4;
}  // namespace

__global__ __launch_bounds__(512) void conv_dw_duo_f32_kernel(const ConvDwDuo32 a) {
  extern __shared__ __attribute__((aligned(16))) float smem32[];
  const int id = blockIdx.x;
  if (id < a.n2)
    conv2_dw_f32_body(smem32, id, a.pool1, a.dpool2, a.pool2, a.amax2, a.dW2, a.db2, a.B);
  else if (id >= a.n2 + a.n1) {
    CSTAMP(0);
    if (a.adam.wsh_world > 0)  // (launch-uniform) "wshard": this rank's slice only, stored into every rank's next-parity buffer
      adam_tf_body<false, true>(a.adam, id - a.n2 - a.n1, a.nadam, 512);
    else
      adam_tf_body(a.adam, id - a.n2 - a.n1, a.nadam, 512);
    CSTAMP(5);
  } else {
    CSTAMP(0);
    conv1_dw_body(smem32, id - a.n2, a.n1, a.images, a.perm, a.step_ptr, a.spe, a.B, a.dpool1, a.amax1, a.dW1, a.db1);
    CSTAMP(5);
  }
}

// "wshard" fields of a hosted optimizer range: the slice's place in every rank's two parameter buffers
static bool wsh_fields_ok(const OptimParams* adam) {
  if (!adam || adam->wsh_world == 0) return true;
  if (adam->wsh_world < 1 || adam->wsh_world > 8 || adam->wsh_rank < 0 || adam->wsh_rank >= adam->wsh_world) return false;
  for (int par = 0; par < 2; ++par)
    for (int r = 0; r < adam->wsh_world; ++r)
      if (!adam->Wsh[par][r] || (reinterpret_cast<uintptr_t>(adam->Wsh[par][r]) & 15)) return false;
  return true;
}

static bool g_f32_attr_set = false;
DTF_API int dtf_conv2_f32_init() {
  if (g_f32_attr_set) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_fwd_f32_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS2);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_bwd_data_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          BD_LDS);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dw_duo_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          DUO32_LDS);
  if (e != hipSuccess) return (int)e;
  g_f32_attr_set = true;
  return 0;
}

DTF_API int dtf_conv2_fwd_f32(const float* x, const float* W, const float* bias, float* out, void* amax, int B, hipStream_t st) {
  if (B <= 0) return ERR_SHAPE;
  if (dtf_f32_emulation()) return dtf_conv2_fwd_x3(x, W, bias, out, amax, B, st);
  if (!g_f32_attr_set) {
    const int rc = dtf_conv2_f32_init();
    if (rc) return rc;
  }
  hipLaunchKernelGGL(conv2_fwd_f32_kernel, dim3(dtf_cdiv(49L * B, FWD_WIN)), dim3(512), FWD_LDS2, st, x, W, bias, out,
                     (uint8_t*)amax, B);
  DTF_CHECK_LAUNCH();
  return 0;
}

DTF_API int dtf_conv2_bwd_data_f32(const float* dpool2, const float* pool2, const void* amax2, const float* W,
                                   const float* pool1, float* dpool1, int B, const OptimParams* adam, hipStream_t st) {
  return dtf_conv2_bwd_data_f32_push(dpool2, pool2, amax2, W, pool1, dpool1, B, adam, nullptr, st);
}

DTF_API int dtf_conv2_bwd_data_f32_push(const float* dpool2, const float* pool2, const void* amax2, const float* W,
                                        const float* pool1, float* dpool1, int B, const OptimParams* adam,
                                        const ShardPush* push, hipStream_t st) {
  if (B <= 0) return ERR_SHAPE;
  if (adam && (adam->n <= 0 || adam->n % 4 || adam->zero_from % 4 || !adam->step)) return ERR_SHAPE;
  if (!wsh_fields_ok(adam)) return ERR_SHAPE;
  if (push && (push->n4 <= 0 || !push->src || push->world < 1 || push->world > 8 || push->rank < 0 || push->rank >= push->world))
    return ERR_SHAPE;
  // (the bf16x6 kernel holds 122 KB of LDS: one workgroup per CU, no resident slots for hosted roles)
  if (!adam && !push && dtf_f32_emulation()) return dtf_conv2_bwd_data_x3(dpool2, pool2, amax2, W, pool1, dpool1, B, st);
  if (!g_f32_attr_set) {
    const int rc = dtf_conv2_f32_init();
    if (rc) return rc;
  }
  ConvBwdData32 a;
  a.dpool2 = dpool2; a.pool2 = pool2; a.amax2 = (const uint8_t*)amax2; a.W = W; a.pool1 = pool1; a.dpool1 = dpool1;
  a.B = B; a.nbd = dtf_cdiv(196L * B, BD_POS);
  a.adam = OptimParams{};
  a.nadam = 0;
  if (adam) {
    a.adam = *adam;
    a.adam.ticket = nullptr;  // (the step is advanced by the optimizer launch that covers the rest of the arena)
    a.nadam = 512 - a.nbd > 64 ? 512 - a.nbd : 64;
  }
  a.push = ShardPush{};
  a.npush = 0;
  if (push) {
    for (int p = 0; p < push->world; ++p)
      if (!push->dst[p]) return ERR_SHAPE;
    a.push = *push;
    // 2048 float4 per workgroup and pass; never more workgroups than the resident slots the data-gradient role leaves
    const long want = (push->n4 + 2047) / 2048;
    const int room = 512 - a.nbd - a.nadam > 32 ? 512 - a.nbd - a.nadam : 32;
    a.npush = (int)(want < room ? want : room);
  }
  hipLaunchKernelGGL(conv2_bwd_data_f32_kernel, dim3(a.nbd + a.nadam + a.npush), dim3(512), BD_LDS, st, a);
  DTF_CHECK_LAUNCH();
  return 0;
}

// Both filter gradients of the conv stack in one launch: conv2's from (pool1, dpool2, pool2, amax2), conv1's from
// (images[perm], dpool1, amax1) when dW1 != null; conv2's bias gradient when db2 != null.  dW2 / db2 / dW1 / db1 are
// accumulated with atomics (pre-zeroed).
DTF_API int dtf_conv_dw_duo_f32(const float* dpool2, const float* pool2, const void* amax2, const float* pool1, float* dW2,
                                float* db2, int B, const float* images, const int* perm, const int* step_ptr, int spe,
                                const float* dpool1, const void* amax1, float* dW1, float* db1, const OptimParams* adam,
                                hipStream_t st) {
  if (B <= 0 || (perm && spe <= 0)) return ERR_SHAPE;
  if (adam && (adam->n <= 0 || adam->n % 4 || adam->zero_from % 4 || !adam->step)) return ERR_SHAPE;
  if (!wsh_fields_ok(adam)) return ERR_SHAPE;
  if (!g_f32_attr_set) {
    const int rc = dtf_conv2_f32_init();
    if (rc) return rc;
  }
  ConvDwDuo32 a;
  a.pool1 = pool1; a.dpool2 = dpool2; a.pool2 = pool2; a.amax2 = (const uint8_t*)amax2; a.dW2 = dW2; a.db2 = db2;
  a.B = B; a.n2 = dtf_cdiv(B, 4) * 10;
  a.images = images; a.perm = perm; a.step_ptr = step_ptr; a.spe = spe;
  a.dpool1 = dpool1; a.amax1 = (const uint8_t*)amax1; a.dW1 = dW1; a.db1 = db1;
  a.n1 = dW1 ? (B < 128 ? B : 128) : 0;
  a.adam = OptimParams{};
  a.nadam = 0;
  if (adam) {
    a.adam = *adam;
    a.adam.ticket = nullptr;  // (the step is advanced by the optimizer launch that covers the rest of the arena)
    // every workgroup of the launch reserves the same 80 KB of LDS, so two fit on a CU: the optimizer workgroups fill the
    // slots the two filter-gradient roles leave in one resident wave of 512 (more would queue behind them as a tail)
    a.nadam = 512 - (a.n2 + a.n1) > 64 ? 512 - (a.n2 + a.n1) : 64;
  }
  hipLaunchKernelGGL(conv_dw_duo_f32_kernel, dim3(a.n2 + a.n1 + a.nadam), dim3(512), DUO32_LDS, st, a);
  DTF_CHECK_LAUNCH();
  return 0;
}
