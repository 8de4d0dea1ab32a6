// C ABI of libdtf_hip.so (mirrored by distributed_tensorflow_amd/native.py).
#pragma once
#include "common.h"

struct HeadParams;
struct OptimParams;
struct ShardPush;

DTF_API int dtf_fc_fwd(const void* X, int x_is_f32, const float* W, float* out, int M, int N, int K, int splits,
                       const float* bias, int relu, hipStream_t st);
DTF_API int dtf_fc_bwd_data(const void* dY, const float* W, float* dX, int M, int N, int K, int splits, hipStream_t st);
DTF_API int dtf_fc_dw(const void* X, int x_is_f32, const void* dY, float* dW, int M, int N, int K, int splits, hipStream_t st);
DTF_API int dtf_fc_bwd_trio(const void* X0, const void* dY0, float* dW0, int M0, int N0, const void* X1, const void* dY1,
                            float* dW1, int M1, int N1, int K, const void* dYq, const float* Wq, float* dXq, int Mq, int Nq,
                            int Kq, int splits, const void* const* gA, const void* const* gB, int granks, hipStream_t st);
DTF_API int dtf_ipc_sync(const void* comm, hipStream_t st);
DTF_API int dtf_ipc_gather_copy(const void* comm, void* dst, long bytes, int blocks_per_peer, hipStream_t st);
DTF_API int dtf_ipc_publish(const void* comm, const float* src, long n, int blocks, hipStream_t st);
DTF_API int dtf_ipc_adam_oneshot(const void* comm, const OptimParams* op, hipStream_t st);
DTF_API int dtf_ipc_sync_gather(const void* comm, void* dst, long bytes, int blocks_per_peer, hipStream_t st);
DTF_API int dtf_fc_dw_pair(const void* X0, const void* dY0, float* dW0, int M0, int N0, const void* X1, const void* dY1,
                           float* dW1, int M1, int N1, int K, hipStream_t st);
DTF_API int dtf_conv_bias_relu_pool(const void* x, const float* Wt, const float* bias, void* out, void* argmax, int B,
                                    int H, int W, int Cin, int Cout, int KH, int KW, hipStream_t st);
DTF_API int dtf_conv_bwd_data_mask(const void* dY, const float* Wt, const void* mask, float* dX, int B, int H, int W,
                                   int Cin, int Cout, int KH, int KW, hipStream_t st);
DTF_API int dtf_conv_dw(const void* x, const void* dY, float* dW, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                        int splits, hipStream_t st);
DTF_API int dtf_conv_bias_relu(const void* x, const void* Wp, const float* bias, void* out, int B, int H, int W, int Cin,
                               int in_cstride, int Cout, int out_cstride, int KH, int KW, int stride, int pad_h, int pad_w,
                               int Ho, int Wo, int relu, hipStream_t st);
DTF_API int dtf_conv1_fwd(const float* images, const int* perm, const int* step_ptr, int spe, int B, const float* Wt,
                          const float* bias, void* out, void* amax, hipStream_t st);
DTF_API int dtf_conv1_dw(const float* images, const int* perm, const int* step_ptr, int spe, int B, const float* dpool1,
                         const void* amax, float* dW, float* db, hipStream_t st);
DTF_API int dtf_unpool(const float* dpool, const void* pooled, const void* amax, void* dY, float* db, int B, int Hp, int Wp,
                       int C, hipStream_t st);
DTF_API int dtf_head(const HeadParams* hp, int blocks, hipStream_t st);
DTF_API int dtf_adam_tf(const OptimParams* p, hipStream_t st);
DTF_API int dtf_sgd(const OptimParams* p, hipStream_t st);
DTF_API int dtf_conv2_lds_init();
DTF_API int dtf_conv2_bwd_lds_init();
DTF_API int dtf_conv1_fwd_pack(const float* images, const int* perm, const int* step_ptr, int spe, int B, const float* Wt,
                               const float* bias, void* out, void* amax, const float* Wc2, void* Wc2f, void* Wc2b,
                               hipStream_t st);
DTF_API int dtf_conv2_fwd_lds_packed(const void* x, const void* Wf, const float* bias, void* out, void* amax, int B,
                                     hipStream_t st);
DTF_API int dtf_conv2_bwd_data_lds_packed(const float* dpool2, const void* pool2, const void* amax2, const void* Wb,
                                          const void* pool1, float* dpool1, float* db2, int B, hipStream_t st);
DTF_API int dtf_conv2_bwd_data_lds(const float* dpool2, const void* pool2, const void* amax2, const float* W,
                                   const void* pool1, float* dpool1, float* db2, int B, hipStream_t st);
DTF_API int dtf_conv_dw_duo(const float* dpool2, const void* pool2, const void* amax2, const void* pool1, float* dW2, int B,
                            int ipw, const float* images, const int* perm, const int* step_ptr, int spe,
                            const float* dpool1, const void* amax1, float* dW1, float* db1, const struct OptimParams* adam,
                            hipStream_t st);
DTF_API int dtf_conv2_dw_lds(const float* dpool2, const void* pool2, const void* amax2, const void* pool1, float* dW,
                             int B, int ipw, hipStream_t st);
DTF_API int dtf_conv2_bwd_lds(const float* dpool2, const void* pool2, const void* amax2, const float* W, const void* pool1,
                              float* dpool1, float* dW, float* db2, int B, int ipw, hipStream_t st);
DTF_API int dtf_tensor_stats(const float* x, long n, const double* pos_limits, int npos, double* sums, unsigned* counts,
                             hipStream_t st);

// ---- exact-fp32 (reference precision) path: f32_fc.hip, f32_conv2.hip, mnist_kernels.hip
DTF_API int dtf_fc_fwd_f32(const float* X, const float* W, float* out, int M, int N, int K, int splits, const float* bias,
                           int relu, hipStream_t st);
DTF_API int dtf_fc_bwd_data_f32(const float* dY, const float* W, float* dX, int M, int N, int K, int splits, hipStream_t st);
DTF_API int dtf_fc_dw_f32(const float* X, int ldx, const float* dY, int ldy, float* dW, int M, int N, int K, int splits,
                          hipStream_t st);
DTF_API int dtf_fc_bwd_trio_f32_bias(const float* X0, const float* dY0, float* dW0, int M0, int N0, const float* X1,
                                     const float* dY1, int ldy1, float* dW1, int M1, int N1, int K, const float* dYq,
                                     const float* Wq, float* dXq, int Mq, int Nq, int Kq, int splits, const void* const* gA,
                                     const void* const* gB, int granks, float* db0, float* db1, float* metrics,
                                     int metrics_ring, const int* step_ptr, hipStream_t st);
DTF_API int dtf_fc_bwd_trio_f32_shard_bias(const float* X1, const float* dY1, int ldy1, float* dW1, int M1, int N1, int K,
                                           const float* dYq, const float* Wq, float* dXq, int Mq, int Nq, int Kq, int splits,
                                           const void* const* gA, const void* const* gB, int granks, int ranks_per_set,
                                           int lda0, float* dW0_slice, int M0_slice, int N0, const float* Wq_alt,
                                           const int* wq_sel, float* db0, float* db1, float* metrics, int metrics_ring,
                                           const int* step_ptr, hipStream_t st);
DTF_API int dtf_fc_bwd_trio_f32(const float* X0, const float* dY0, float* dW0, int M0, int N0, const float* X1,
                                const float* dY1, int ldy1, float* dW1, int M1, int N1, int K, const float* dYq,
                                const float* Wq, float* dXq, int Mq, int Nq, int Kq, int splits, const void* const* gA,
                                const void* const* gB, int granks, hipStream_t st);
DTF_API int dtf_conv2_f32_init();
DTF_API int dtf_conv1_fwd_f32(const float* images, const int* perm, const int* step_ptr, int spe, int B, const float* Wt,
                              const float* bias, float* out, void* amax, hipStream_t st);
DTF_API int dtf_conv2_fwd_f32(const float* x, const float* W, const float* bias, float* out, void* amax, int B, hipStream_t st);
DTF_API int dtf_conv2_bwd_data_f32(const float* dpool2, const float* pool2, const void* amax2, const float* W,
                                   const float* pool1, float* dpool1, int B, const struct OptimParams* adam, hipStream_t st);
DTF_API int dtf_conv_dw_duo_f32(const float* dpool2, const float* pool2, const void* amax2, const float* pool1, float* dW2,
                                float* db2, int B, const float* images, const int* perm, const int* step_ptr, int spe,
                                const float* dpool1, const void* amax1, float* dW1, float* db1, const struct OptimParams* adam,
                                hipStream_t st);

// ---- fp32 on the bf16 matrix cores ("bf16x6", bf16x3.h): f32x3_conv2.hip
DTF_API void dtf_f32_emulation_set(int on);
DTF_API int dtf_f32_emulation();
DTF_API int dtf_conv2_pack3(const float* W, void* Wf3, void* Wb3, hipStream_t st);
DTF_API int dtf_conv2_fwd_x3_packed(const float* x, const void* Wf3, const float* bias, float* out, void* amax, int B,
                                    hipStream_t st);
DTF_API int dtf_conv2_bwd_data_x3_packed(const float* dpool2, const float* pool2, const void* amax2, const void* Wb3,
                                         const float* pool1, float* dpool1, int B, hipStream_t st);
DTF_API int dtf_conv2_fwd_x3(const float* x, const float* W, const float* bias, float* out, void* amax, int B, hipStream_t st);
DTF_API int dtf_conv2_bwd_data_x3(const float* dpool2, const float* pool2, const void* amax2, const float* W,
                                  const float* pool1, float* dpool1, int B, hipStream_t st);
DTF_API int dtf_conv1_fwd_f32_pack3(const float* images, const int* perm, const int* step_ptr, int spe, int B, const float* Wt,
                                    const float* bias, float* out, void* amax, const float* Wc2, void* Wf3, void* Wb3,
                                    hipStream_t st);

// ---- "shard" data parallelism (shard_body.h): per-rank work independent of the number of ranks
DTF_API int dtf_conv2_bwd_data_f32_push(const float* dpool2, const float* pool2, const void* amax2, const float* W,
                                        const float* pool1, float* dpool1, int B, const struct OptimParams* adam,
                                        const struct ShardPush* push, hipStream_t st);
DTF_API int dtf_fc_bwd_trio_f32_shard(const float* X1, const float* dY1, int ldy1, float* dW1, int M1, int N1, int K,
                                      const float* dYq, const float* Wq, float* dXq, int Mq, int Nq, int Kq, int splits,
                                      const void* const* gA, const void* const* gB, int granks, int ranks_per_set, int lda0,
                                      float* dW0_slice, int M0_slice, int N0, const float* Wq_alt, const int* wq_sel,
                                      hipStream_t st);
DTF_API int dtf_fc_fwd_f32_sel(const float* X, const float* W0, const float* W1, const int* sel, float* out, int M, int N,
                               int K, int splits, hipStream_t st);
DTF_API int dtf_ipc_wshard_bufs(const void* comm, long lo_floats, long n_floats, float** out);
// ---- "early push" operand exchange (ipc-gather2)
DTF_API int dtf_ipc_sync_gather_pushed(const void* comm, void* dst, long bytes, int rows, int pitch4, long tail_off4,
                                       long tail_n4, int blocks_per_peer, hipStream_t st);
DTF_API int dtf_ipc_push_desc(const void* comm, float* src, long n_floats, struct ShardPush* push);
DTF_API int dtf_fc_fwd_f32_push(const float* X, const float* W, float* out, int M, int N, int K, int splits,
                                const struct ShardPush* push, hipStream_t st);
DTF_API int dtf_ipc_reset_errors(const void* comm);
DTF_API int dtf_ipc_sync_gather_slice(const void* comm, void* dst, long bytes, int rows, int pitch4, int lo4, int w4,
                                      long tail_off4, long tail_n4, int blocks_per_peer, hipStream_t st);
DTF_API int dtf_ipc_shard_desc(const void* comm, long lo_floats, long n_floats, float* src, int zero_src,
                               struct ShardPush* push);
DTF_API const float* dtf_ipc_landing(const void* comm);
