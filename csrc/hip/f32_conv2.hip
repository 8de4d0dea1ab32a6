// Exact-fp32 kernels for the MNIST conv2 layer (14x14x32 -> 14x14x64, 5x5 SAME, 2x2 pool behind it): forward, data
// gradient and filter gradient on v_mfma_f32_16x16x4_f32 — the reference's precision (tf.nn.conv2d on float32,
// /root/reference/demo1/train.py:40-41,84).
//
// At 64 FLOP/clk/SIMD these kernels are bound by the MFMA pipe (2 GFLOP each at batch 100 = 12.8 us on all 1024 SIMDs), so
// what matters is that every SIMD gets the same number of MFMAs and never waits:
//  * forward: one workgroup per 20 pooling windows of the whole batch (245 workgroups at batch 100), 8 waves = (output-
//    channel half, input-channel quarter): every wave issues the same 500 MFMAs on 10 accumulator tiles.
//  * data gradient: one workgroup per (image, half of the input channels), the four waves split K (output-channel quarters
//    of every tap): 1300 MFMAs each on 13 accumulator tiles.
//    In both, the halo'd image sits in LDS, a wave streams its own disjoint slice of the filter straight from L2 into
//    registers one tap ahead (no barrier inside the k loop), and the partial sums are combined through LDS at the end.
//  * filter gradient: one workgroup per (4 images, half of the output channels, filter row); per image the waves take
//    the 49 position groups round-robin with a per-image rotation, so over the 4 images each wave issues exactly 49 x 20
//    MFMAs; 250 workgroups at batch 100.
// A fragments that run over channels use the freedom in the k order (f32_gemm.h): one ds_read_b64 / b128 feeds 2 / 4 MFMAs.
#include "common.h"
#include "conv1_dw_body.h"
#include "adam_body.h"
#include "f32_gemm.h"

// -DDTF_CONV_STAMPS (diagnostic builds only, scripts/dbg): thread 0 of every workgroup records (shader clock, 100 MHz
// real-time clock) pairs at phase boundaries into g_conv_stamps[workgroup][32]
#ifdef DTF_CONV_STAMPS
__device__ unsigned long long* g_conv_stamps;
DTF_API int dtf_dbg_set_conv_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), &p, sizeof(p)); }
#define CSTAMP(i) do { if (threadIdx.x == 0 && g_conv_stamps) { g_conv_stamps[blockIdx.x * 32 + 2 * (i)] = __builtin_amdgcn_s_memtime(); \
  g_conv_stamps[blockIdx.x * 32 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
// per-wave: loop-end clock and HW_ID (SIMD / CU placement) at g_conv_stamps[workgroup][32 .. 47] / [48 .. 63]
#define CSTAMP_WAVE() do { if ((threadIdx.x & 63) == 0 && g_conv_stamps) { \
  g_conv_stamps[gridDim.x * 32 + blockIdx.x * 32 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime(); \
  g_conv_stamps[gridDim.x * 32 + blockIdx.x * 32 + 16 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg(4 | (31 << 11)); } } while (0)
#else
#define CSTAMP(i) do { } while (0)
#define CSTAMP_WAVE() do { } while (0)
#endif

namespace {
constexpr int ERR_SHAPE = -22;
constexpr int XP = 36;                  // floats per input halo pixel (32 ch + 4 pad)
constexpr int XIMG = 18 * 18 * XP;      // 11 664 floats
constexpr int DP = 68;                  // floats per dY halo pixel (64 ch + 4 pad)
constexpr int DIMG = 18 * 18 * DP;      // 22 032 floats
constexpr int BWD_RED = 4 * 13 * 256;
constexpr int BWD_LDS = ((DIMG > BWD_RED ? DIMG : BWD_RED) + 64) * 4;
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
      if (c < 2592) *reinterpret_cast<f32x4*>(img + s * XIMG + (c >> 3) * XP + (c & 7) * 4) = iv[s][i];
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
    abase[mt] = slot * XIMG + ((2 * py + (sub >> 1)) * 18 + 2 * px + (sub & 1)) * XP + c0 + 2 * q;
  }
  f32x4 acc[FWD_MT][2];
#pragma unroll
  for (int mt = 0; mt < FWD_MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

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
      const int toff = (dy * 18 + dx) * XP;
      float2 af[FWD_MT];
#pragma unroll
      for (int mt = 0; mt < FWD_MT; ++mt) af[mt] = *reinterpret_cast<const float2*>(img + abase[mt] + toff);
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
    }
    __builtin_amdgcn_sched_barrier(0);
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

// ------------------------------------------------------------------ data gradient + un-pool + ReLU mask + bias gradient
// grid (B, 2): input channels [16 y, 16 y + 16).  dpool2 fp32 [B,49,64]; pool2 fp32 [B,49,64]; amax2 u8 [B,49,64];
// W fp32 HWIO [25][32][64]; pool1 fp32 [B,196,32] (ReLU mask of the layer below); dpool1 fp32 [B,196,32]; db2[64] (atomics)
__global__ __launch_bounds__(256) void conv2_bwd_data_f32_kernel(const float* __restrict__ dpool2, const float* __restrict__ pool2,
                                                                 const uint8_t* __restrict__ amax2, const float* __restrict__ W,
                                                                 const float* __restrict__ pool1, float* __restrict__ dpool1,
                                                                 float* __restrict__ db2) {
  extern __shared__ __attribute__((aligned(16))) float smem32[];
  float* dimg = smem32;
  float* dbs = smem32 + (DIMG > BWD_RED ? DIMG : BWD_RED);
  const int b = blockIdx.x, n0 = blockIdx.y * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int k0 = 16 * wave;  // this wave's output channels [k0, k0 + 16) of every tap

  // B fragment of (flipped) tap t': elements j = W[24 - t'][n0 + r][k0 + 4q + j]
  const float* wbase = W + (long)(n0 + r) * 64 + k0 + 4 * q;
  // (a whole filter row one row ahead, pinned: see conv2_fwd_f32_kernel)
  f32x4 bcur[5], bnxt[5];
#pragma unroll
  for (int dx = 0; dx < 5; ++dx) bcur[dx] = *reinterpret_cast<const f32x4*>(wbase + (long)(24 - dx) * 2048);

  UnpoolTask32 ut[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int task = tid + 256 * i;  // 784 tasks: window = task >> 4, channel group = task & 15 (= tid & 15 for every i)
    ut[i].load(dpool2, pool2, amax2, (long)b * 3136 + (task < 784 ? task * 4 : 0));
  }
  if (tid < 64) dbs[tid] = 0.f;
  for (int c = tid; c < DIMG / 4; c += 256) *reinterpret_cast<f32x4*>(dimg + c * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  f32x4 gsum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int task = tid + 256 * i;
    if (task < 784) {
      ut[i].mask();
      const int win = task >> 4, py = win / 7, px = win - py * 7;
      ut[i].store(dimg + (task & 15) * 4, py, px, 18, DP, 2, 2);
      gsum[0] += ut[i].g[0]; gsum[1] += ut[i].g[1]; gsum[2] += ut[i].g[2]; gsum[3] += ut[i].g[3];
    }
  }
  // bias gradient (one of the image's two workgroups): lanes l, l^16, l^32 share a channel group
  if (db2 && blockIdx.y == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = gsum[j];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lane < 16) atomicAdd(&dbs[lane * 4 + j], v);
    }
  }
  __syncthreads();
  if (db2 && blockIdx.y == 0 && tid < 64) atomicAdd(&db2[tid], dbs[tid]);

  int abase[13];
#pragma unroll
  for (int mt = 0; mt < 13; ++mt) {
    int m = mt * 16 + r;
    m = m < 196 ? m : 195;
    const int y = m / 14, xx = m - y * 14;
    abase[mt] = (y * 18 + xx) * DP + k0 + 4 * q;
  }
  f32x4 acc[13];
#pragma unroll
  for (int mt = 0; mt < 13; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int dy = 0; dy < 5; ++dy) {
    const int dyn = dy < 4 ? dy + 1 : 4;
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) bnxt[dx] = *reinterpret_cast<const f32x4*>(wbase + (long)(24 - (dyn * 5 + dx)) * 2048);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) {
      const int toff = (dy * 18 + dx) * DP;
      f32x4 af[13];
#pragma unroll
      for (int mt = 0; mt < 13; ++mt) af[mt] = *reinterpret_cast<const f32x4*>(dimg + abase[mt] + toff);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 13; ++mt) acc[mt] = mfma16f32(af[mt][j], bcur[dx][j], acc[mt]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) bcur[dx] = bnxt[dx];
  }

  __syncthreads();
  float* red = smem32;
#pragma unroll
  for (int mt = 0; mt < 13; ++mt) *reinterpret_cast<f32x4*>(red + ((wave * 13 + mt) * 64 + lane) * 4) = acc[mt];
  __syncthreads();
  for (int mt = wave; mt < 13; mt += 4) {
    f32x4 s = *reinterpret_cast<const f32x4*>(red + (mt * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((w * 13 + mt) * 64 + lane) * 4);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = mt * 16 + 4 * q + i;
      if (row < 196) {
        const long o = ((long)b * 196 + row) * 32 + n0 + r;
        dpool1[o] = pool1[o] > 0.f ? s[i] : 0.f;
      }
    }
  }
}

// ------------------------------------------------------------------ filter gradient
// block -> (image quad g, output-channel half c, filter row dy); 256 threads.
// pool1 fp32 [B,196,32] (the conv input); dW fp32 [25][32][64] accumulated with atomics.
namespace {
constexpr int DW_XROWS = 14 * 18 * XP;   // the 14 halo rows a filter row touches: 9072 floats
constexpr int DW_DY = 196 * XP;          // dY of one channel half, position-major, 36-float pitch: 7056 floats
constexpr int DW_RED = 4 * 20 * 256;     // 20 480 floats
constexpr int DW_LDS = (DW_XROWS + DW_DY > DW_RED ? DW_XROWS + DW_DY : DW_RED) * 4;

__device__ __forceinline__ void conv2_dw_f32_body(float* __restrict__ smem, const int id, const float* __restrict__ pool1,
                                                  const float* __restrict__ dpool2, const float* __restrict__ pool2,
                                                  const uint8_t* __restrict__ amax2, float* __restrict__ dW, int B) {
  float* ximg = smem;
  float* dyi = smem + DW_XROWS;
  const int dy = id % 5, c = (id / 5) & 1, g = id / 10;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int b0 = 4 * g;
  const int nimg = min(4, B - b0);

  f32x4 xv[8];
  UnpoolTask32 ut[2];
  auto load = [&](const int b) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ch = tid + 256 * i;  // 2016 chunks: (staged row 0..13, halo col 0..17, part 0..7)
      const int pix = ch >> 3, part = ch & 7;
      const int sy = pix / 18, hx = pix - sy * 18;
      const int y = sy + dy - 2, xx = hx - 2;
      const bool ok = ch < 2016 && (unsigned)y < 14u && (unsigned)xx < 14u;
      xv[i] = *reinterpret_cast<const f32x4*>(pool1 + (ok ? (((long)b * 14 + y) * 14 + xx) * 32 + part * 4 : 0L));
      if (!ok) xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int task = tid + 256 * i;  // 392 tasks: window = task >> 3, channel group (of this half) = task & 7
      const int t = task < 392 ? task : 0;
      ut[i].load(dpool2, pool2, amax2, (long)b * 3136 + (t >> 3) * 64 + c * 32 + (t & 7) * 4);
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ch = tid + 256 * i;
      if (ch < 2016) *reinterpret_cast<f32x4*>(ximg + (ch >> 3) * XP + (ch & 7) * 4) = xv[i];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int task = tid + 256 * i;
      if (task < 392) {
        ut[i].mask();
        const int win = task >> 3, py = win / 7, px = win - py * 7;
        ut[i].store(dyi + (task & 7) * 4, py, px, 14, XP, 0, 0);
      }
    }
  };

  f32x4 acc[5][2][2];  // [dx][cin tile][cout tile]
#pragma unroll
  for (int dx = 0; dx < 5; ++dx)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[dx][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nimg > 0) load(b0);
  for (int im = 0; im < nimg; ++im) {
    __syncthreads();  // the previous image has been consumed
    store();
    __syncthreads();
    if (im + 1 < nimg) load(b0 + im + 1);
    // k-step s covers positions 4s .. 4s+3 (lane quarter q <-> position 4s + q); wave (s + im) & 3 takes step s
    for (int s = (wave - im) & 3; s < 49; s += 4) {
      const int pos = 4 * s + q;
      const int y = pos / 14, xx = pos - y * 14;
      const float* xa = ximg + (y * 18 + xx) * XP + r;  // tap (dy, dx): staged row y, halo column xx + dx
      const float* da = dyi + pos * XP + r;
      const float b0f = da[0], b1f = da[16];
      float a[5][2];
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        a[dx][0] = xa[dx * XP];
        a[dx][1] = xa[dx * XP + 16];
      }
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        acc[dx][0][0] = mfma16f32(a[dx][0], b0f, acc[dx][0][0]);
        acc[dx][0][1] = mfma16f32(a[dx][0], b1f, acc[dx][0][1]);
        acc[dx][1][0] = mfma16f32(a[dx][1], b0f, acc[dx][1][0]);
        acc[dx][1][1] = mfma16f32(a[dx][1], b1f, acc[dx][1][1]);
      }
    }
  }
  // combine the four waves' partial sums, then one atomic per element and workgroup
  __syncthreads();
  float* red = smem;
#pragma unroll
  for (int dx = 0; dx < 5; ++dx)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        *reinterpret_cast<f32x4*>(red + ((wave * 20 + (dx * 2 + mt) * 2 + nt) * 64 + lane) * 4) = acc[dx][mt][nt];
  __syncthreads();
  for (int t = wave; t < 20; t += 4) {
    f32x4 s = *reinterpret_cast<const f32x4*>(red + (t * 64 + lane) * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((w * 20 + t) * 64 + lane) * 4);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    const int dx = t >> 2, mt = (t >> 1) & 1, nt = t & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cin = mt * 16 + 4 * q + i, cout = c * 32 + nt * 16 + r;
      atomicAdd(&dW[((long)(dy * 5 + dx) * 32 + cin) * 64 + cout], s[i]);
    }
  }
}

// conv2 dW (workgroups [0, n2)), conv1 dW/db (the next n1) and optionally trailing optimizer workgroups (see ConvDwDuo in
// conv2_lds.hip) in one launch.
struct ConvDwDuo32 {
  const float* pool1; const float* dpool2; const float* pool2; const uint8_t* amax2; float* dW2;
  int B, n2;
  const float* images; const int* perm; const int* step_ptr; int spe;
  const float* dpool1; const uint8_t* amax1; float* dW1; float* db1;
  int n1;
  OptimParams adam;  // adam.n == 0: no optimizer workgroups
  int nadam;
};
constexpr int DUO32_LDS = DW_LDS > CONV1_DW_LDS_FLOATS * 4 ? DW_LDS : CONV1_DW_LDS_FLOATS * 4;
}  // namespace

__global__ __launch_bounds__(256) void conv_dw_duo_f32_kernel(const ConvDwDuo32 a) {
  extern __shared__ __attribute__((aligned(16))) float smem32[];
  const int id = blockIdx.x;
  if (id < a.n2)
    conv2_dw_f32_body(smem32, id, a.pool1, a.dpool2, a.pool2, a.amax2, a.dW2, a.B);
  else if (id >= a.n2 + a.n1)
    adam_tf_body(a.adam, id - a.n2 - a.n1, a.nadam, 256);
  else
    conv1_dw_body(smem32, id - a.n2, a.n1, a.images, a.perm, a.step_ptr, a.spe, a.B, a.dpool1, a.amax1, a.dW1, a.db1);
}

static bool g_f32_attr_set = false;
DTF_API int dtf_conv2_f32_init() {
  if (g_f32_attr_set) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_fwd_f32_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, FWD_LDS2);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv2_bwd_data_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          BWD_LDS);
  if (e != hipSuccess) return (int)e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dw_duo_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          DUO32_LDS);
  if (e != hipSuccess) return (int)e;
  g_f32_attr_set = true;
  return 0;
}

DTF_API int dtf_conv2_fwd_f32(const float* x, const float* W, const float* bias, float* out, void* amax, int B, hipStream_t st) {
  if (B <= 0) return ERR_SHAPE;
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
                                   const float* pool1, float* dpool1, float* db2, int B, hipStream_t st) {
  if (B <= 0) return ERR_SHAPE;
  if (!g_f32_attr_set) {
    const int rc = dtf_conv2_f32_init();
    if (rc) return rc;
  }
  hipLaunchKernelGGL(conv2_bwd_data_f32_kernel, dim3(B, 2), dim3(256), BWD_LDS, st, dpool2, pool2, (const uint8_t*)amax2, W,
                     pool1, dpool1, db2);
  DTF_CHECK_LAUNCH();
  return 0;
}

// Both filter gradients of the conv stack in one launch: conv2's from (pool1, dpool2, pool2, amax2), conv1's from
// (images[perm], dpool1, amax1) when dW1 != null.  dW2 / dW1 / db1 are accumulated with atomics (pre-zeroed).
DTF_API int dtf_conv_dw_duo_f32(const float* dpool2, const float* pool2, const void* amax2, const float* pool1, float* dW2,
                                int B, const float* images, const int* perm, const int* step_ptr, int spe,
                                const float* dpool1, const void* amax1, float* dW1, float* db1, const OptimParams* adam,
                                hipStream_t st) {
  if (B <= 0 || (perm && spe <= 0)) return ERR_SHAPE;
  if (adam && (adam->n <= 0 || adam->n % 4 || adam->zero_from % 4 || !adam->step)) return ERR_SHAPE;
  if (!g_f32_attr_set) {
    const int rc = dtf_conv2_f32_init();
    if (rc) return rc;
  }
  ConvDwDuo32 a;
  a.pool1 = pool1; a.dpool2 = dpool2; a.pool2 = pool2; a.amax2 = (const uint8_t*)amax2; a.dW2 = dW2;
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
  hipLaunchKernelGGL(conv_dw_duo_f32_kernel, dim3(a.n2 + a.n1 + a.nadam), dim3(256), DUO32_LDS, st, a);
  DTF_CHECK_LAUNCH();
  return 0;
}
