// fp32 products on the bf16 matrix cores ("bf16x6"): an fp32 value is the exact sum of three bf16 planes
//   a = a0 + a1 + a2 (+ eps, |eps| <= 2^-26 |a|),   a0 = rne(a), a1 = rne(a - a0), a2 = rne(a - a0 - a1)
// (both residuals are exact in fp32: each has at most 16 significant bits left), and a product is taken as
//   a b ~= a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0
// — every bf16 x bf16 product is exact in the fp32 accumulator of v_mfma_f32_16x16x32_bf16, the dropped terms (a1 b2,
// a2 b1, a2 b2) are below 2^-24 |a b|.  Six bf16 MFMAs at 16x the rate of v_mfma_f32_16x16x4_f32 = 6/16 of its time; the
// split costs ~5 VALU operations per element and is done ONCE where an operand is staged (LDS image, packed filter), never
// per fragment read.  The results meet the unchanged fp32 tolerances of tests/test_f32_gpu.py (float64 oracle).
#pragma once
#include "common.h"

// one MFMA k-step: lanes (r = lane & 15, q = lane >> 4) hold k = 8q .. 8q + 7 of row r (A) / column r (B);
// the accumulator's element i is row 4q + i, column r (the layout of the 16x16x4 fp32 MFMA)
__device__ __forceinline__ f32x4 mfma16bf(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

struct Bf3Pair { uint32_t p0, p1, p2; };  // (lo, hi) of two values, per plane

// (two values per v_cvt_pk_bf16_f32 and per v_pk_add_f32: 10 VALU operations per pair; converting the values one by one and
// OR-ing the halves together, as pack_bf2 compiles, takes 15)
typedef __bf16 bf16x2_x3 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf3_pack2(const f32x2 v) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_x3));
}
__device__ __forceinline__ Bf3Pair bf3_split2(float lo, float hi) {
  Bf3Pair o;
  f32x2 r = {lo, hi};
  o.p0 = bf3_pack2(r);
  r = r - f32x2{__uint_as_float(o.p0 << 16), __uint_as_float(o.p0 & 0xffff0000u)};
  o.p1 = bf3_pack2(r);
  r = r - f32x2{__uint_as_float(o.p1 << 16), __uint_as_float(o.p1 & 0xffff0000u)};
  o.p2 = bf3_pack2(r);
  return o;
}

// four consecutive values -> 8 bytes per plane
struct Bf3Quad { uint2 p0, p1, p2; };
__device__ __forceinline__ Bf3Quad bf3_split4(f32x4 v) {
  const Bf3Pair a = bf3_split2(v[0], v[1]), b = bf3_split2(v[2], v[3]);
  Bf3Quad o;
  o.p0 = make_uint2(a.p0, b.p0);
  o.p1 = make_uint2(a.p1, b.p1);
  o.p2 = make_uint2(a.p2, b.p2);
  return o;
}

// the six products of one (A, B) fragment pair, small terms first
#define DTF_BF3_PRODUCTS(acc, a, b)          \
  do {                                       \
    acc = mfma16bf((a)[2], (b)[0], acc);     \
    acc = mfma16bf((a)[0], (b)[2], acc);     \
    acc = mfma16bf((a)[1], (b)[1], acc);     \
    acc = mfma16bf((a)[1], (b)[0], acc);     \
    acc = mfma16bf((a)[0], (b)[1], acc);     \
    acc = mfma16bf((a)[0], (b)[0], acc);     \
  } while (0)
