// Plain-old-data parameter blocks shared by the kernels, the step executor and the ctypes mirror
// in distributed_tensorflow_amd/native.py (keep field order/types in sync with that file).
#pragma once
#include <stdint.h>

struct HeadParams {
  const float* slabs;     // [nslab][B][1024] raw fc1 partial sums (no bias)
  int nslab;
  long slab_stride;
  const float* b_fc1;
  const float* W_fc2;     // [1024][10]
  const float* b_fc2;
  const int* labels_all;  // int32 class ids of the whole split
  const int* perm;        // or null (labels_all[b])
  const int* step_ptr;
  int spe;
  int B;
  float keep_prob;
  float grad_scale;       // 1 / local batch
  unsigned long long seed;
  int double_softmax;
  int training;
  uint16_t* h_drop;       // bf16 [B][1024]           (training)
  uint16_t* dh;           // bf16 [B][1024] d/d(fc1 pre-activation)   (training)
  uint16_t* dz;           // bf16 [B][10]   dlogits   (training; feeds the fc2 dW GEMM)
  float* db_fc2;          // [10]   atomics
  float* db_fc1;          // [1024] atomics
  float* metrics;         // [2]: loss sum, correct count (atomics)
  int metrics_ring;       // >0: metrics += 2*(step % ring)
  float* probs;           // optional fp32 [B][10]
  int* pred;              // optional int32 [B]
  float* zero_buf;        // optional: buffer zeroed as a side job (float4 granularity)
  long zero_n4;
  int f32;                // h_drop / dh are fp32 [B][1024] and dz fp32 [B][16] (zero-padded rows) instead of bf16
};

struct OptimParams {
  float* P;
  float* G;
  float* M;
  float* V;
  long n;          // elements (multiple of 4)
  long zero_from;  // gradients at [zero_from, n) are reset to 0 (atomically accumulated segments)
  float lr, beta1, beta2, eps, grad_scale;
  int* step;             // device: number of updates applied so far
  unsigned int* ticket;  // device: arrival counter (0 between launches)
  float* metrics_ring;   // optional: slot (t % ring) zeroed for the next step's head kernel
  int ring;
  const float* Gsrc;     // optional: read the gradients from here instead of G (an all-reduce result buffer);
                         // the re-zeroing of [zero_from, n) still applies to G
  const float* Gsrcs[8];  // fused one-shot all-reduce (dtf_ipc_adam_oneshot): gradient = sum of the first nsrc of these
  int nsrc;
  unsigned* skip;         // optional device words [skip flag, skipped-step count]: flag != 0 = this step's gradient exchange
                          // had an expired wait somewhere (ipc_allreduce.hip): leave P/M/V and the step counter alone, only
                          // discard the accumulated gradients; the launch that owns the ticket clears the flag
  // "wshard" data-parallel schedule (engine.hip: MnistEngine::wshard; wsh_world > 0): the range is this rank's slice of a parameter tensor that
  // lives in two IPC-mapped buffers per rank, alternating by step parity.  The update reads the parameters from this rank's
  // buffer of parity (*step & 1) and stores the new values into EVERY rank's buffer of the other parity; P is unused.
  float* Wsh[2][8];       // [parity][rank]: the slice's place in that rank's buffer, as mapped into this process
  int wsh_world, wsh_rank;
};
