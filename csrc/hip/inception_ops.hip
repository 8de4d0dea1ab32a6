// Non-GEMM kernels of the Inception-v3 path:
//   * max-pool 3x3/2 VALID, avg-pool 3x3/1 SAME (TF semantics: padding excluded from the divisor),
//     global 8x8 average pool -> fp32 bottleneck; all NHWC bf16, 8 channels (16 B) per lane, outputs
//     written at a channel stride/offset so a pool branch lands directly in its concat slice;
//   * fused JPEG-decoded-image preprocessing: bilinear resize (TF1 ResizeBilinear, align_corners=False,
//     src = dst * scale) + (x-128)/128 normalisation + optional distortion (random scale/crop/flip/
//     brightness of retrain1/retrain.py:137-165 composed into one sampling transform) for a whole
//     batch of variable-sized uint8 HWC images -> bf16 NHWC [B,299,299,8] (3 channels + zero pad).
#include "common.h"

namespace {
constexpr int ERR_SHAPE = -22;

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = bf2f(w[j] & 0xffffu);
    f[2 * j + 1] = bf2f(w[j] >> 16);
  }
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}
}  // namespace

// MODE 0: max 3x3 stride s VALID;  MODE 1: avg 3x3 stride 1 SAME (exclude padding)
template <int MODE>
__global__ __launch_bounds__(256) void pool3x3_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out, int B,
                                                      int H, int W, int C, int in_cs, int Ho, int Wo, int out_cs,
                                                      int stride) {
  const int c8n = C >> 3;
  const long total = (long)B * Ho * Wo * c8n;
  for (long gi = (long)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (long)gridDim.x * 256) {
    const int c0 = (int)(gi % c8n) * 8;
    long t = gi / c8n;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const int b = (int)(t / Ho);
    // all nine taps are requested before the first is used (clamped address + select; with the loads under `if (ok)` the
    // compiler emits nine branches with a load and an s_waitcnt each: nine dependent round trips per output, found in the ISA)
    const int h0 = MODE == 0 ? ho * stride : ho - 1, w0 = MODE == 0 ? wo * stride : wo - 1;
    uint4 v[9];
    bool ok[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int hi = h0 + t9 / 3, wi = w0 + t9 % 3;
      ok[t9] = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      v[t9] = *reinterpret_cast<const uint4*>(x + (ok[t9] ? (((long)b * H + hi) * W + wi) * in_cs + c0 : 0L));
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = MODE == 0 ? -3.0e38f : 0.f;
    int cnt = 0;
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      float f[8];
      unpack8(v[t9], f);
      cnt += ok[t9] ? 1 : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float fj = ok[t9] ? f[j] : (MODE == 0 ? -3.0e38f : 0.f);
        acc[j] = MODE == 0 ? fmaxf(acc[j], fj) : acc[j] + fj;
      }
    }
    if (MODE == 1) {
      const float inv = 1.f / (float)cnt;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] *= inv;
    }
    *reinterpret_cast<uint4*>(out + (((long)b * Ho + ho) * Wo + wo) * out_cs + c0) = pack8(acc);
  }
}

DTF_API int dtf_maxpool3x3(const void* x, void* out, int B, int H, int W, int C, int in_cs, int out_cs, int stride,
                           hipStream_t st) {
  if (B <= 0 || C % 8 || in_cs % 8 || out_cs % 8 || H < 3 || W < 3 || stride < 1) return ERR_SHAPE;
  const int Ho = (H - 3) / stride + 1, Wo = (W - 3) / stride + 1;
  const long total = (long)B * Ho * Wo * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pool3x3_kernel<0>, dim3((int)blocks), dim3(256), 0, st, (const uint16_t*)x, (uint16_t*)out, B, H, W,
                     C, in_cs, Ho, Wo, out_cs, stride);
  DTF_CHECK_LAUNCH();
  return 0;
}

DTF_API int dtf_avgpool3x3_same(const void* x, void* out, int B, int H, int W, int C, int in_cs, int out_cs,
                                hipStream_t st) {
  if (B <= 0 || C % 8 || in_cs % 8 || out_cs % 8) return ERR_SHAPE;
  const long total = (long)B * H * W * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pool3x3_kernel<1>, dim3((int)blocks), dim3(256), 0, st, (const uint16_t*)x, (uint16_t*)out, B, H, W,
                     C, in_cs, H, W, out_cs, 1);
  DTF_CHECK_LAUNCH();
  return 0;
}

// out[b][c] (fp32) = mean over the HxW positions of x[b,:,:,c]   (pool_3: 8x8x2048 -> the 2048-d bottleneck)
__global__ __launch_bounds__(256) void global_avgpool_kernel(const uint16_t* __restrict__ x, float* __restrict__ out,
                                                             int B, int HW, int C) {
  const int c8n = C >> 3;
  const long total = (long)B * c8n;
  const long gi = (long)blockIdx.x * 256 + threadIdx.x;
  if (gi >= total) return;
  const int c0 = (int)(gi % c8n) * 8;
  const int b = (int)(gi / c8n);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int p = 0; p < HW; ++p) {
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(x + ((long)b * HW + p) * C + c0), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += f[j];
  }
  const float inv = 1.f / (float)HW;
  float4* o = reinterpret_cast<float4*>(out + (long)b * C + c0);
  o[0] = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
  o[1] = make_float4(acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv);
}

DTF_API int dtf_global_avgpool(const void* x, float* out, int B, int HW, int C, hipStream_t st) {
  if (B <= 0 || C % 8 || HW <= 0) return ERR_SHAPE;
  const long total = (long)B * (C / 8);
  hipLaunchKernelGGL(global_avgpool_kernel, dim3(dtf_cdiv(total, 256)), dim3(256), 0, st, (const uint16_t*)x, out, B, HW, C);
  DTF_CHECK_LAUNCH();
  return 0;
}

// Per-image descriptor of the batched preprocessing kernel.
struct ImageDesc {
  long offset;       // byte offset of the uint8 HWC (3-channel) pixels in the packed batch buffer
  int H, W;          // source size
  float scale_y, scale_x;  // source px per destination px (H/299 for the plain path)
  float off_y, off_x;      // crop offset in destination px of the (virtual) pre-crop image
  int flip;          // mirror left-right
  float brightness;  // multiplicative factor applied before normalisation (1 = none)
};

// dst[b, y, x, 0..2] = ((bilinear(src_b, (y+off_y)*scale_y, (x'+off_x)*scale_x) * brightness) - 128) / 128, ch 3..7 = 0
// F32: the same pixels as fp32 NHWC-8 (the input of the reference-precision forward, models/inception_v3.py Fp32Backbone).
template <bool F32>
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ pixels, const ImageDesc* __restrict__ desc,
                                                         void* __restrict__ out_, int B, int S) {
  const long total = (long)B * S * S;
  for (long gi = (long)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (long)gridDim.x * 256) {
    const int x = (int)(gi % S);
    long t = gi / S;
    const int y = (int)(t % S);
    const int b = (int)(t / S);
    const ImageDesc d = desc[b];
    const int xs = d.flip ? (S - 1 - x) : x;
    const float fy = ((float)y + d.off_y) * d.scale_y, fx = ((float)xs + d.off_x) * d.scale_x;
    const int y0 = min((int)fy, d.H - 1), x0 = min((int)fx, d.W - 1);
    const int y1 = min(y0 + 1, d.H - 1), x1 = min(x0 + 1, d.W - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const uint8_t* src = pixels + d.offset;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float p00 = src[((long)y0 * d.W + x0) * 3 + c], p01 = src[((long)y0 * d.W + x1) * 3 + c];
      const float p10 = src[((long)y1 * d.W + x0) * 3 + c], p11 = src[((long)y1 * d.W + x1) * 3 + c];
      const float top = p00 + (p01 - p00) * wx, bot = p10 + (p11 - p10) * wx;
      v[c] = ((top + (bot - top) * wy) * d.brightness - 128.f) * (1.f / 128.f);
    }
    if constexpr (F32) {
      float4* o = reinterpret_cast<float4*>(reinterpret_cast<float*>(out_) + gi * 8);
      o[0] = make_float4(v[0], v[1], v[2], 0.f);
      o[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(out_) + gi * 8) =
          make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], 0.f), 0u, 0u);
    }
  }
}

template <bool F32>
static int launch_preprocess(const void* pixels, const void* desc, void* out, int B, int S, hipStream_t st) {
  if (B <= 0 || S <= 0) return ERR_SHAPE;
  const long total = (long)B * S * S;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(preprocess_kernel<F32>, dim3((int)blocks), dim3(256), 0, st, (const uint8_t*)pixels,
                     (const ImageDesc*)desc, out, B, S);
  DTF_CHECK_LAUNCH();
  return 0;
}

DTF_API int dtf_preprocess_images(const void* pixels, const void* desc, void* out, int B, int S, hipStream_t st) {
  return launch_preprocess<false>(pixels, desc, out, B, S, st);
}

// out: fp32 [B, S, S, 8]
DTF_API int dtf_preprocess_images_f32(const void* pixels, const void* desc, void* out, int B, int S, hipStream_t st) {
  return launch_preprocess<true>(pixels, desc, out, B, S, st);
}

// ---------------------------------------------------------------------------------------------- batch gather
// dst[b] = src[rows[b]] for rows of `row16` 16-byte units (the fine-tune step's batch: 1.43 MB bf16 images picked by the
// device-side class-balanced sampler out of the HBM-resident training set, /root/reference/retrain1/retrain.py:322-340 does
// the same pick on the host, one file at a time), and labels_out[b] = labels[rows[b]].
// grid = (workgroups per row, B): every workgroup copies a strided share of one row, 4 x 16 B in flight per lane.
__global__ __launch_bounds__(256) void gather_rows_kernel(const f32x4* __restrict__ src, const int* __restrict__ rows,
                                                          f32x4* __restrict__ dst, long row16, const int* __restrict__ labels,
                                                          int* __restrict__ labels_out) {
  const int b = blockIdx.y;
  const long r = rows[b];
  if (labels && blockIdx.x == 0 && threadIdx.x == 0) labels_out[b] = labels[r];
  const f32x4* __restrict__ s = src + r * row16;
  f32x4* __restrict__ d = dst + (long)b * row16;
  constexpr int U = 4;
  for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < row16; base += (long)gridDim.x * 256 * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + u * 256;
      v[u] = s[i < row16 ? i : base];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + u * 256;
      if (i < row16) d[i] = v[u];
    }
  }
}

DTF_API int dtf_gather_rows(const void* src, const int* rows, void* dst, long row_bytes, int B, const int* labels,
                            int* labels_out, hipStream_t st) {
  if (!src || !rows || !dst || B <= 0 || row_bytes <= 0 || row_bytes % 16 || (labels && !labels_out)) return -22;
  const long row16 = row_bytes / 16;
  long per_row = (row16 + 256 * 4 * 4 - 1) / (256 * 4 * 4);  // ~4 passes per workgroup
  if (per_row < 1) per_row = 1;
  if (per_row > 64) per_row = 64;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)per_row, (unsigned)B), dim3(256), 0, st,
                     static_cast<const f32x4*>(src), rows, static_cast<f32x4*>(dst), row16, labels, labels_out);
  DTF_CHECK_LAUNCH();
  return 0;
}
