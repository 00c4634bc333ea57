// Three-way bf16 split of fp32 operands for the bf16 matrix cores (shared by gemm_split.hip and bwd_fused.hip).
#pragma once
#include "common.h"

namespace gcmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Round-to-nearest three-way split, two values at a time (v_cvt_pk_bf16_f32 packs the pair):
// x = p1 + p2 + p3 + r3 with |p2| <= 2^-8 |x|, |p3| <= 2^-16 |x|, |r3| <= 2^-24 |x| and signed,
// unbiased residuals.  The six products kept below drop p2*q3 + p3*q2 + p3*q3 <= 2^-23 |x||y|.
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
  f32x2 v = {x0, x1};
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 f1 = {__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xFFFF0000u)};
  const f32x2 r1 = v - f1;  // exact
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
  const f32x2 f2 = {__uint_as_float(p2 << 16), __uint_as_float(p2 & 0xFFFF0000u)};
  const f32x2 r2 = r1 - f2;  // exact
  p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

// one value: the three bf16 pieces in the low halves of p1..p3
__device__ __forceinline__ void split3(float x, unsigned& p1, unsigned& p2, unsigned& p3) {
  unsigned q1, q2, q3;
  split3_pair(x, 0.f, q1, q2, q3);
  p1 = q1 << 16;  // callers take the piece from the HIGH half
  p2 = q2 << 16;
  p3 = q3 << 16;
}

// ---- bfloat16 activation storage (gcmi_model_desc.storage == 1): raw 16-bit patterns in HBM, fp32 everywhere else
typedef unsigned short bf16_t;

// two floats -> one dword of two bf16 (round to nearest even, v_cvt_pk_bf16_f32; a NaN stays a NaN): low half = a
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
// the value a float has after one trip through bf16 storage
__device__ __forceinline__ float round_bf16(float a) { return bf16_lo(pack_bf16x2(a, 0.f)); }
// 8 (4) bf16 of a 16-byte (8-byte) piece -> floats
__device__ __forceinline__ void widen8(const uint4 r, float (&v)[8]) {
  v[0] = bf16_lo(r.x); v[1] = bf16_hi(r.x); v[2] = bf16_lo(r.y); v[3] = bf16_hi(r.y);
  v[4] = bf16_lo(r.z); v[5] = bf16_hi(r.z); v[6] = bf16_lo(r.w); v[7] = bf16_hi(r.w);
}
__device__ __forceinline__ float4 widen4(const uint2 r) {
  return make_float4(bf16_lo(r.x), bf16_hi(r.x), bf16_lo(r.y), bf16_hi(r.y));
}
__device__ __forceinline__ uint2 narrow4(float a, float b, float c, float d) {
  return make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}
// two floats that ARE bf16 values (low 16 bits zero) -> their packed pair, exactly (one v_perm_b32)
__device__ __forceinline__ unsigned pack_exact_bf16x2(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}

struct Frag3 {
  u32x4 p[3];
};

__device__ __forceinline__ Frag3 split_frag(const float (&v)[8]) {
  Frag3 f;
  unsigned q[3][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split3_pair(v[2 * i], v[2 * i + 1], q[0][i], q[1][i], q[2][i]);
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    f.p[s].x = q[s][0];
    f.p[s].y = q[s][1];
    f.p[s].z = q[s][2];
    f.p[s].w = q[s][3];
  }
  return f;
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) {
  return __builtin_bit_cast(bf16x8, v);
}

// a by-value table entry picked with selects (a dynamic index into a kernarg struct would go through scratch)
template <typename T, int N>
__device__ __forceinline__ T pick_n(const T (&a)[N], int s) {
  T v = a[0];
#pragma unroll
  for (int k = 1; k < N; ++k) v = (s == k) ? a[k] : v;
  return v;
}

}  // namespace gcmi
