// Fused single-pass tensor statistics for summaries (component C18 / kernel K19).
// One read of the tensor yields sum, sum of squares, min, max and the TensorFlow-default exponential
// histogram (limits +-1e-12 * 1.1^k, see utils/summary.py) — everything `variable_summaries`
// (retrain1/retrain.py:249-258) and `tf.summary.histogram` (:277,279) need, instead of the reference's
// five separate reductions + histogram op per variable per step.
#include <hip/hip_runtime.h>
#include "common.h"
#include "dtf_api.h"

namespace {

constexpr int MAX_POS = 800;  // 774 positive limits in the default table

__device__ __forceinline__ unsigned order_key(float f) {  // monotone float -> uint map
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// index of the bucket holding x, with buckets [limit[i-1], limit[i]) as in TF's histogram (upper_bound)
__device__ __forceinline__ int bucket_of(float x, const double* __restrict__ pos, int npos) {
    const double ax = fabs((double)x);
    const int zero_idx = npos;  // limits = -pos[npos-1..0], 0, pos[0..npos-1], DBL_MAX
    if (ax < pos[0]) return (x >= 0.f) ? zero_idx + 1 : zero_idx;
    int k = (int)floorf((log2f(fabsf(x)) + 39.8631371f) * 7.27254f) + 1;  // ~ log_{1.1}(ax / 1e-12) + 1
    k = max(0, min(k, npos));
    if (x >= 0.f) {  // smallest k with pos[k] > ax
        while (k > 0 && pos[k - 1] > ax) --k;
        while (k < npos && pos[k] <= ax) ++k;
        return zero_idx + 1 + k;
    }
    // negative: smallest kk with pos[kk] >= ax; bucket = zero_idx - kk
    while (k > 0 && pos[k - 1] >= ax) --k;
    while (k < npos && pos[k] < ax) ++k;
    return zero_idx - k;
}

__global__ void __launch_bounds__(256) tensor_stats_kernel(const float* __restrict__ x, long n,
                                                           const double* __restrict__ pos, int npos,
                                                           double* __restrict__ sums, unsigned* __restrict__ counts) {
    extern __shared__ unsigned hist[];  // 2*npos + 2 counters
    __shared__ double red[4][4];
    const int nb = 2 * npos + 2;
    for (int i = threadIdx.x; i < nb; i += 256) hist[i] = 0u;
    __syncthreads();
    double s = 0.0, ss = 0.0;
    float mn = INFINITY, mx = -INFINITY;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float v = x[i];
        if (v != v) continue;  // NaNs are not counted
        s += v;
        ss += (double)v * v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
        atomicAdd(&hist[bucket_of(v, pos, npos)], 1u);
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        ss += __shfl_xor(ss, o);
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w][0] = s; red[w][1] = ss; red[w][2] = mn; red[w][3] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int j = 1; j < 4; ++j) {
            s += red[j][0]; ss += red[j][1];
            mn = fminf(mn, (float)red[j][2]); mx = fmaxf(mx, (float)red[j][3]);
        }
        atomicAdd(&sums[0], s);
        atomicAdd(&sums[1], ss);
        atomicMin(&counts[nb], order_key(mn));
        atomicMax(&counts[nb + 1], order_key(mx));
    }
    for (int i = threadIdx.x; i < nb; i += 256) {
        const unsigned c = hist[i];
        if (c) atomicAdd(&counts[i], c);
    }
}

__global__ void stats_init_kernel(double* sums, unsigned* counts, int nb) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nb) counts[i] = 0u;
    if (i == 0) { sums[0] = 0.0; sums[1] = 0.0; counts[nb] = 0xFFFFFFFFu; counts[nb + 1] = 0u; }
}

}  // namespace

// sums: double[2] (sum, sum of squares); counts: uint32[2*npos + 4] = histogram, then order-keyed min and max.
DTF_API int dtf_tensor_stats(const float* x, long n, const double* pos_limits, int npos, double* sums, unsigned* counts,
                             hipStream_t st) {
    if (npos <= 0 || npos > MAX_POS || n < 0) return 1;
    const int nb = 2 * npos + 2;
    stats_init_kernel<<<(nb + 255) / 256, 256, 0, st>>>(sums, counts, nb);
    if (n > 0) {
        long blocks = (n + 256 * 8 - 1) / (256 * 8);
        if (blocks > 1024) blocks = 1024;
        if (blocks < 1) blocks = 1;
        tensor_stats_kernel<<<(int)blocks, 256, nb * sizeof(unsigned), st>>>(x, n, pos_limits, npos, sums, counts);
    }
    DTF_CHECK_LAUNCH();
    return 0;
}
