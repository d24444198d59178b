// The exact three-way bf16 split  x = hi + mid + lo  (round-to-nearest residuals) shared by the operand-split kernel
// (conv.hip) and the producers that write the planes themselves (train.hip: frozen-BatchNorm forward / backward).
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32, a in the low half
}
__device__ __forceinline__ float lo_f32(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f32(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
// x = hi + mid + lo per element; returns the three planes' 8-byte pieces (4 consecutive k each)
__device__ __forceinline__ void split3(float4 v, uint2 &h, uint2 &m, uint2 &l) {
  h.x = pack_bf16(v.x, v.y);
  h.y = pack_bf16(v.z, v.w);
  const float rx = v.x - lo_f32(h.x), ry = v.y - hi_f32(h.x), rz = v.z - lo_f32(h.y), rw = v.w - hi_f32(h.y);
  m.x = pack_bf16(rx, ry);
  m.y = pack_bf16(rz, rw);
  l.x = pack_bf16(rx - lo_f32(m.x), ry - hi_f32(m.x));
  l.y = pack_bf16(rz - lo_f32(m.y), rw - hi_f32(m.y));
}
