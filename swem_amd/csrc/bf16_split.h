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

// ---- fp16 pair ("f16x3" conv arithmetic, round 4) ------------------------------------------------------------------
// x = hi + mid with both terms fp16, round-to-nearest residual: 11 + 11 significant bits and the residual's sign, i.e. x to
// within 2^-24 |x| -- half an fp32 ulp -- wherever mid is a NORMAL fp16 number (|x| >= 2^-2); below that mid is subnormal
// (v_cvt_f16_f32 and the f16 MFMA both keep subnormals: tools/f16_probe.hip) and the absolute error is bounded by half the
// fp16 subnormal spacing, 2^-25.  |x| must stay below 65520 (the fp16 range; the reference's own mixed-precision mode,
// basic_trainer.py:83-86, runs these activations in fp16 -- its fp32 inference has no such limit): beyond it hi is inf, the
// products are NaN, and a consumer's ReLU epilogue turns that NaN into 0.  Every producer of a pair therefore tests what it
// splits (f16_oor) and reports SWEM_FAULT_RANGE through the caller's fault word (range_fault): include/swem_hip.h.
// Plane-count code of this format in the C ABI: SWEM_PLANES_F16 (= 4): two planes, hi then mid.
#ifndef SWEM_PLANES_F16
#define SWEM_PLANES_F16 4
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_f16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));   // round to nearest even, a in the low half
}
__device__ __forceinline__ float lo_f16(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
__device__ __forceinline__ float hi_f16(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
__device__ __forceinline__ void split2h(float4 v, uint2 &h, uint2 &m) {
  h.x = pack_f16(v.x, v.y);
  h.y = pack_f16(v.z, v.w);
  m.x = pack_f16(v.x - lo_f16(h.x), v.y - hi_f16(h.x));
  m.y = pack_f16(v.z - lo_f16(h.y), v.w - hi_f16(h.y));
}
// 1 if any of the four values is beyond the fp16 range: |x| >= 65520 (rounds to inf), inf or NaN -- an unsigned compare of
// the magnitude bits against 65520.0f = 0x477ff000
__device__ __forceinline__ unsigned f16_oor(float4 v) {
  const unsigned a = max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu),
                         max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu));
  return a >= 0x477ff000u ? 1u : 0u;
}
__device__ __forceinline__ unsigned f32_nan(float4 v) {   // 1 if any of the four is a NaN
  const unsigned a = max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu),
                         max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu));
  return a > 0x7f800000u ? 1u : 0u;
}
#ifndef SWEM_FAULT_RANGE
#define SWEM_FAULT_RANGE 4
#endif
// Report `bad` (per lane, accumulated over everything the lane split) into the caller's sticky fault word: one wave-level
// test in the normal case; in the fault case ONE lane per wave looks at the word and sets the bit if it is not set yet.
// 1-D blocks whose size is a multiple of 64 (lane = threadIdx.x & 63).
__device__ __forceinline__ void range_fault(unsigned *fault, unsigned bad) {
  if (!fault) return;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(bad != 0u);
  if (m && (threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m)) {
    if (!(__hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (unsigned)SWEM_FAULT_RANGE))
      __hip_atomic_fetch_or(fault, (unsigned)SWEM_FAULT_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// the split a producer was asked for: npl = 2 / 3 bf16 planes (l valid for 3) or SWEM_PLANES_F16
__device__ __forceinline__ void split_as(int npl, float4 v, uint2 &h, uint2 &m, uint2 &l) {
  if (npl == SWEM_PLANES_F16) {
    split2h(v, h, m);
    l = make_uint2(0u, 0u);
  } else {
    split3(v, h, m, l);
  }
}
// ... and `bad` |= "an fp16 pair was asked for and v does not fit" (bf16 planes have the fp32 range: never a fault)
__device__ __forceinline__ void split_as(int npl, float4 v, uint2 &h, uint2 &m, uint2 &l, unsigned &bad) {
  split_as(npl, v, h, m, l);
  if (npl == SWEM_PLANES_F16) bad |= f16_oor(v);
}
