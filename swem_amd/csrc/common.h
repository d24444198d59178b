// Shared helpers for libswem_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

#include "../../include/swem_hip.h"

void swem_set_error(const char *fmt, ...);
// em.hip: one bank of key bases into rows [out_off, out_off+L) of a packed-keys image [NK][C/4][out_rows][4],
// l2-normalised (normalize = 1) or as they are (0)
int swem_norm_bases_into(void *stream, const float *kappa, float *kn, int NK, int C, int L, int out_rows,
                         int out_off, int normalize);
// conv.hip: batched GEMM y[b] = x[b] . w[b]^T on pre-split bf16 planes (the pre-split convolution kernel as a 1x1 layer over
// an M x 1 image per batch item): x = plane 0 of [K/8][B*M][8] planes `ps` elements apart (bs = M*K), w = plane 0 of batch
// item 0's filter planes [K/8][Ncols][8], planes Ncols*K apart, batch items w_bs elements apart; M a multiple of 128.
// plan / ws as swem_conv2d_nhwc_bf16x3 (math field 3 = two planes, 1 = three; SWEM_PLAN_F16: x and w are fp16 pairs).
// y_planes (may be NULL): y's own planes [Ncols/8][B*M][8], y_nplanes (2, 3 or SWEM_PLANES_F16) of them written, B*M*Ncols
// elements apart (Ncols % 8 == 0).  out_scale multiplies every output (the caller's operand scaling, undone exactly).
// fault: the caller's sticky fault word (SWEM_FAULT_RANGE when y does not fit an fp16 pair asked for), or NULL.
int swem_gemm_bf16x3_batched(void *stream, const void *x, int K, long long bs, long long ps, int B, int M, const void *w,
                             long long w_bs, float *y, int Ncols, int plan, void *ws, size_t ws_bytes, void *y_planes,
                             int y_nplanes, float out_scale, void *fault);
// raise the dynamic-LDS limit of a kernel once per device (needed above 64 KiB).  The "done" record is one atomic bit per
// device ordinal: concurrent first launches from two host threads both set the attribute (idempotent) and OR their bit.
#define SWEM_ALLOW_LDS(kernel, bytes)                                                                   \
  do {                                                                                                  \
    static std::atomic<unsigned long long> done_{0ull};                                                 \
    int dev_ = 0;                                                                                       \
    (void)hipGetDevice(&dev_);                                                                          \
    const unsigned long long bit_ = 1ull << (dev_ & 63);                                                \
    if (!(done_.load(std::memory_order_acquire) & bit_)) {                                              \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),                       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));    \
      if (e_ != hipSuccess) {                                                                           \
        swem_set_error("hipFuncSetAttribute(%s): %s", #kernel, hipGetErrorString(e_));                  \
        return SWEM_E_HIP;                                                                              \
      }                                                                                                 \
      done_.fetch_or(bit_, std::memory_order_release);                                                  \
    }                                                                                                   \
  } while (0)

#define SWEM_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      swem_set_error(__VA_ARGS__);     \
      return (code);                   \
    }                                  \
  } while (0)

#define SWEM_CHECK_LAUNCH(name)                                    \
  do {                                                             \
    hipError_t e_ = hipGetLastError();                             \
    if (e_ != hipSuccess) {                                        \
      swem_set_error("%s: %s", (name), hipGetErrorString(e_));     \
      return SWEM_E_HIP;                                           \
    }                                                              \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SWEM_L2_EPS 1e-6f

// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31];
// D[i][j] lives in lane (j + 32*hi), register r with i = (r & 3) + 8 * (r >> 2) + 4 * hi.
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// Four k-steps from one float4 per operand.  Lane half h holds k = 4*(2j+h) + e, e = 0..3: the k order is a
// permutation of 0..7 that A and B share, so the sum is the same set of products.
__device__ __forceinline__ f32x16 mfma32x4(float4 a, float4 b, f32x16 c) {
  c = mfma32(a.x, b.x, c);
  c = mfma32(a.y, b.y, c);
  c = mfma32(a.z, b.z, c);
  c = mfma32(a.w, b.w, c);
  return c;
}
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// exp(x * inv_tau) for the softmax kernels as ONE transcendental: v_exp_f32 on x * (log2(e)/tau).
// The reference computes exp((a - m) / tau); the two differ by ~2 ulp (2e-7 relative), far inside the 1e-4 per-step
// tolerance, and the ~30-instruction libm expf would dominate the EM / matching epilogues (64-128 values per lane).
__device__ __forceinline__ float exp_scaled(float x, float log2e_over_tau) {
  return __builtin_amdgcn_exp2f(x * log2e_over_tau);
}
#define SWEM_LOG2E 1.4426950408889634f

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- debug stamps (-DSWEM_EM_STAMPS builds only; see em.hip) ---------------------------------------------------------
#ifdef SWEM_EM_STAMPS
extern long long *g_swem_stamps;
extern int g_swem_stamp_slot;
#define STAMP_ARG , long long *stamps
#define STAMP_PASS , (g_swem_stamps ? g_swem_stamps + 16 * (g_swem_stamp_slot++) : nullptr)
#ifndef SWEM_STAMP_BLOCK
#define SWEM_STAMP_BLOCK 0   /* which block of a launch writes the stamps (-DSWEM_STAMP_BLOCK=n) */
#endif
#define STAMP(i)                                                                                       \
  do {                                                                                                 \
    if (stamps && blockIdx.x == SWEM_STAMP_BLOCK && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { \
      stamps[2 * (i)] = (long long)__builtin_amdgcn_s_memtime();                                       \
      stamps[2 * (i) + 1] = (long long)__builtin_amdgcn_s_memrealtime();                               \
    }                                                                                                  \
  } while (0)
// accumulated cycles of a code region (diagnosis of the k-loops: time inside the counted waits / the barrier):
//   STAMP_ACC_DECL(t); ... STAMP_T0(); <region> STAMP_ACC(t); ... STAMP_ACC_OUT(slot, t);
#define STAMP_ACC_DECL(v) long long v = 0, v##_t0_ = 0
#define STAMP_T0(v) v##_t0_ = (long long)__builtin_amdgcn_s_memtime()
#define STAMP_ACC(v) v += (long long)__builtin_amdgcn_s_memtime() - v##_t0_
#define STAMP_ACC_OUT(i, v)                                                                             \
  do {                                                                                                 \
    if (stamps && blockIdx.x == SWEM_STAMP_BLOCK && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { \
      stamps[2 * (i)] = v;                                                                             \
      stamps[2 * (i) + 1] = -1;                                                                        \
    }                                                                                                  \
  } while (0)
#else
#define STAMP_ARG
#define STAMP_PASS
#define STAMP(i)
#define STAMP_ACC_DECL(v)
#define STAMP_T0(v)
#define STAMP_ACC(v)
#define STAMP_ACC_OUT(i, v)
#endif
