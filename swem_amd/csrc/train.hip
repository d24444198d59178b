// Training step of the SWEM path on gfx950: loss (BootstrappedCE + mask IoU) forward / backward and the AdamW update.
// C ABI: include/swem_hip_train.h.  Everything here is HBM-bound single-pass work; reductions that feed a scalar are
// accumulated in fp64 in a fixed order (deterministic, and closer to the exact sum than any fp32 order).
#include <math.h>

#include "../../include/swem_hip_train.h"
#include "common.h"
#include "bf16_split.h"

namespace {

constexpr int LOSS_MAXC = 8;  // objects + background per clip (reference trains with MAX_NUM_OBJS = 2)

// order-preserving map float -> uint (ascending)
__device__ __forceinline__ unsigned fkey(float x) {
  unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ double block_sum(double v, double *sh) {
  const int tid = threadIdx.x;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  __syncthreads();
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
  if (tid == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;  // valid in thread 0
}

// bce_losses.py:28-36 / losses/__init__.py:50-55: softmax over the valid channels, per-pixel CE, partial IoU sums
__global__ __launch_bounds__(256) void loss_pixel_fwd_kernel(const float *__restrict__ logits,
                                                             const long long *__restrict__ label,
                                                             const float *__restrict__ valid, float *__restrict__ prob,
                                                             float *__restrict__ raw, float *__restrict__ part, int N1,
                                                             long long HW, int nblk, long long label_bs) {
  __shared__ double sh[4];
  const int b = blockIdx.y;
  const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool in = px < HW;
  float lg[LOSS_MAXC];
  bool ok[LOSS_MAXC];
  int rank[LOSS_MAXC];  // index among the valid channels
  int nv = 0;
#pragma unroll
  for (int c = 0; c < LOSS_MAXC; ++c) {
    ok[c] = c < N1 && (!valid || valid[b * N1 + c] > 0.5f);
    rank[c] = nv;
    nv += ok[c] ? 1 : 0;
    lg[c] = (ok[c] && in) ? logits[((long long)b * N1 + c) * HW + px] : -INFINITY;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < LOSS_MAXC; ++c) mx = fmaxf(mx, lg[c]);
  float se = 0.f;
#pragma unroll
  for (int c = 0; c < LOSS_MAXC; ++c) se += ok[c] ? expf(lg[c] - mx) : 0.f;
  const float lse = logf(se);
  const int tgt = in ? (int)label[(long long)b * label_bs + px] : -1;
  float ce = 0.f;
#pragma unroll
  for (int c = 0; c < LOSS_MAXC; ++c) {
    if (c >= N1) continue;
    const float lsm = lg[c] - mx - lse;  // log_softmax, as F.cross_entropy
    const float p = ok[c] ? expf(lsm) : 0.f;
    if (in) prob[((long long)b * N1 + c) * HW + px] = p;
    const bool hit = ok[c] && rank[c] == tgt;
    if (hit) ce = -lsm;
    const float oh = hit ? 1.f : 0.f;
    const double s_in = block_sum((in && ok[c]) ? (double)fminf(p, oh) : 0.0, sh);
    const double s_un = block_sum((in && ok[c]) ? (double)fmaxf(p, oh) : 0.0, sh);
    if (threadIdx.x == 0) {
      float *dst = part + (((long long)b * N1 + c) * nblk + blockIdx.x) * 2;
      dst[0] = (float)s_in;
      dst[1] = (float)s_un;
    }
  }
  if (in) raw[(long long)b * HW + px] = ce;
}

__global__ __launch_bounds__(256) void loss_iou_reduce_kernel(const float *__restrict__ part, float *__restrict__ iou,
                                                              int nblk) {
  __shared__ double sh[4];
  const int bc = blockIdx.x;
  double a = 0.0, u = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) {
    a += part[((long long)bc * nblk + i) * 2];
    u += part[((long long)bc * nblk + i) * 2 + 1];
  }
  a = block_sum(a, sh);
  u = block_sum(u, sh);
  if (threadIdx.x == 0) {
    iou[bc * 2] = (float)a;
    iou[bc * 2 + 1] = (float)(u + 1e-6);  // bce_losses.py:121
  }
}

// torch.topk(raw_loss, k) per row (bce_losses.py:49-50) as a radix select of the k-th largest value: 4 passes over
// the row with an 8-bit histogram each, then one pass for the count / sum above the threshold.  One block per row.
__global__ __launch_bounds__(1024) void loss_row_select_kernel(const float *__restrict__ raw,
                                                               float *__restrict__ rowstat, long long HW,
                                                               long long k_arg, const long long *__restrict__ k_dev) {
  const long long k = k_dev ? *k_dev : k_arg;   // device-resident when the step is replayed from a HIP graph
  __shared__ unsigned hist[256];
  __shared__ unsigned sel_digit, sel_remain;
  __shared__ double sh[16];
  const float *row = raw + (long long)blockIdx.x * HW;
  const int tid = threadIdx.x;
  unsigned prefix = 0, remain = (unsigned)k;
  if (k > 0) {
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
      // (run-length merged: cross-entropy values share their sign / exponent byte almost everywhere, so the first passes would
      // send a thousand threads' increments to two or three LDS words, one at a time -- 230 us for a 147k-pixel row, the step's
      // slowest single-block launch; a thread counts a run of equal digits in a register and adds it once.  Counts are integers:
      // the histogram is the same whatever the order.)
      unsigned run_d = 256u, run_n = 0u;
      for (long long i = tid; i < HW; i += 1024) {
        const unsigned key = fkey(row[i]);
        if ((key & himask) != prefix) continue;
        const unsigned dgt = (key >> shift) & 255u;
        if (dgt == run_d) {
          ++run_n;
        } else {
          if (run_n) atomicAdd(&hist[run_d], run_n);
          run_d = dgt;
          run_n = 1u;
        }
      }
      if (run_n) atomicAdd(&hist[run_d], run_n);
      __syncthreads();
      if (tid == 0) {
        unsigned acc = 0;
        int d = 255;
        for (; d > 0; --d) {
          if (acc + hist[d] >= remain) break;
          acc += hist[d];
        }
        sel_digit = (unsigned)d;
        sel_remain = remain - acc;
      }
      __syncthreads();
      prefix |= sel_digit << shift;
      remain = sel_remain;
      __syncthreads();
    }
  }
  // prefix = key of the k-th largest value
  double sum = 0.0, cnt = 0.0, eq = 0.0;
  float thr = 0.f;
  for (long long i = tid; i < HW; i += 1024) {
    const float x = row[i];
    const unsigned key = fkey(x);
    if (k == 0 || key > prefix) {
      sum += x;
      cnt += 1.0;
    } else if (key == prefix) {
      eq += 1.0;
      thr = x;
    }
  }
  sum = block_sum(sum, sh);
  cnt = block_sum(cnt, sh);
  eq = block_sum(eq, sh);
  // every thread that saw the threshold holds the same value; publish one
  __shared__ float thr_sh;
  if (tid == 0) thr_sh = 0.f;
  __syncthreads();
  if (thr != 0.f) thr_sh = thr;
  __syncthreads();
  if (tid == 0) {
    float *o = rowstat + blockIdx.x * 4;
    o[0] = k > 0 ? thr_sh : 0.f;
    o[1] = (float)cnt;
    o[2] = (float)sum;
    o[3] = (float)eq;
  }
}

// losses/__init__.py:57-61: means over rows / planes.  One small block.
__global__ void loss_reduce_kernel(const float *__restrict__ rowstat, const float *__restrict__ iou,
                                   const float *__restrict__ valid, float *__restrict__ losses, int B, int N1, int T,
                                   long long HW, long long k_arg, const long long *__restrict__ k_dev,
                                   float aux_ratio) {
  if (threadIdx.x != 0) return;
  const long long k = k_dev ? *k_dev : k_arg;
  const double keff = k > 0 ? (double)k : (double)HW;
  double main = 0.0;
  for (int r = 0; r < T * B; ++r) {
    const float *s = rowstat + r * 4;
    main += ((double)s[2] + (keff - (double)s[1]) * (double)s[0]) / keff;
  }
  main /= (double)(T * B);
  double aux = 0.0;
  for (int b = 0; b < B; ++b) {
    int nv = 0;
    double acc = 0.0;
    for (int c = 0; c < N1; ++c) {
      if (valid && !(valid[b * N1 + c] > 0.5f)) continue;
      ++nv;
      for (int t = 0; t < T; ++t) {
        const float *q = iou + (((long long)t * B + b) * N1 + c) * 2;
        acc += (double)(q[0] / q[1]);
      }
    }
    aux += 1.0 - acc / (double)(T * nv);
  }
  aux /= (double)B;
  losses[0] = (float)(main + (double)aux_ratio * aux);
  losses[1] = (float)main;
  losses[2] = (float)aux;
}

// d total / d logits: CE term w * (p - onehot) on the selected pixels, IoU term through the softmax Jacobian.
// Ties follow ATen: torch.min/max split the gradient in half on equal inputs; pixels equal to the top-k threshold
// share the remaining k - #above slots evenly (torch.topk picks an arbitrary subset of them).
__global__ __launch_bounds__(256) void loss_pixel_bwd_kernel(const float *__restrict__ prob,
                                                             const float *__restrict__ raw,
                                                             const long long *__restrict__ label,
                                                             const float *__restrict__ valid,
                                                             const float *__restrict__ rowstat,
                                                             const float *__restrict__ iou, float *__restrict__ dlogits,
                                                             int B, int N1, int T, long long HW, long long k_arg,
                                                             float aux_ratio, const float *__restrict__ gout,
                                                             long long label_bs, const long long *__restrict__ k_dev) {
  const long long k = k_dev ? *k_dev : k_arg;
  const float gscale = gout ? gout[0] : 1.f;
  const int b = blockIdx.y;
  const long long px = (long long)blockIdx.x * 256 + threadIdx.x;
  if (px >= HW) return;
  const float *rs = rowstat + b * 4;
  const float keff = k > 0 ? (float)k : (float)HW;
  const float x = raw[(long long)b * HW + px];
  float w;
  if (k == 0 || x > rs[0]) w = 1.f;
  else if (x == rs[0]) w = (keff - rs[1]) / rs[3];
  else w = 0.f;
  w /= keff * (float)(B * T);
  const int tgt = (int)label[(long long)b * label_bs + px];
  float p[LOSS_MAXC], dp[LOSS_MAXC], oh[LOSS_MAXC];
  bool ok[LOSS_MAXC];
  int nv = 0;
#pragma unroll
  for (int c = 0; c < LOSS_MAXC; ++c) {
    ok[c] = c < N1 && (!valid || valid[b * N1 + c] > 0.5f);
    oh[c] = (ok[c] && nv == tgt) ? 1.f : 0.f;
    nv += ok[c] ? 1 : 0;
    p[c] = ok[c] ? prob[((long long)b * N1 + c) * HW + px] : 0.f;
  }
  const float s = -aux_ratio / (float)(B * T * nv);
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < LOSS_MAXC; ++c) {
    dp[c] = 0.f;
    if (!ok[c]) continue;
    const float I = iou[(b * N1 + c) * 2], U = iou[(b * N1 + c) * 2 + 1];
    // min(p, t): t = 1 -> p (tie at p == 1); t = 0 -> 0 (tie at p == 0).  max(p, t): the complement.
    const float di = oh[c] > 0.5f ? (p[c] < 1.f ? 1.f : 0.5f) : (p[c] > 0.f ? 0.f : 0.5f);
    const float du = oh[c] > 0.5f ? (p[c] < 1.f ? 0.f : 0.5f) : (p[c] > 0.f ? 1.f : 0.5f);
    dp[c] = s * (di * U - I * du) / (U * U);
    dot += p[c] * dp[c];
  }
#pragma unroll
  for (int c = 0; c < LOSS_MAXC; ++c) {
    if (c >= N1) continue;
    const float g = ok[c] ? w * (p[c] - oh[c]) + p[c] * (dp[c] - dot) : 0.f;
    dlogits[((long long)b * N1 + c) * HW + px] = g * gscale;
  }
}

// gate (optional): `ngate` device floats read at the top of every block; any non-zero one and the whole launch leaves p, m, v
// untouched -- the found_inf gate of a loss scaler, fed by the step's fault flags (swem_fault_flags_f32) after their all-reduce,
// so that every rank skips the same steps without a host round trip.  applied (optional): += 1 by one thread of a launch that
// did update (the host reconciles its step count with it when it next looks).
__global__ __launch_bounds__(256) void adamw_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                    float *__restrict__ m, float *__restrict__ v, long long n, float decay,
                                                    float b1, float b2, float eps, float step_size, float sqrt_bc2,
                                                    const float *__restrict__ gate, int ngate, int *__restrict__ applied) {
  if (gate) {
    for (int i = 0; i < ngate; ++i)
      if (gate[i] != 0.f) return;
  }
  if (applied && blockIdx.x == 0 && threadIdx.x == 0) *applied += 1;
  const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= n) return;
  if (i0 + 4 <= n) {
    float4 pp = *reinterpret_cast<float4 *>(p + i0), gg = *reinterpret_cast<const float4 *>(g + i0);
    float4 mm = *reinterpret_cast<float4 *>(m + i0), vv = *reinterpret_cast<float4 *>(v + i0);
    float *P = &pp.x, *G = &gg.x, *M = &mm.x, *V = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      P[e] *= decay;
      M[e] = b1 * M[e] + (1.f - b1) * G[e];
      V[e] = b2 * V[e] + (1.f - b2) * G[e] * G[e];
      P[e] -= step_size * (M[e] / (sqrtf(V[e]) / sqrt_bc2 + eps));
    }
    *reinterpret_cast<float4 *>(p + i0) = pp;
    *reinterpret_cast<float4 *>(m + i0) = mm;
    *reinterpret_cast<float4 *>(v + i0) = vv;
  } else {
    for (long long i = i0; i < n; ++i) {
      float pv = p[i] * decay, gv = g[i];
      float mv = b1 * m[i] + (1.f - b1) * gv, vv = b2 * v[i] + (1.f - b2) * gv * gv;
      p[i] = pv - step_size * (mv / (sqrtf(vv) / sqrt_bc2 + eps));
      m[i] = mv;
      v[i] = vv;
    }
  }
}

}  // namespace

extern "C" size_t swem_vos_loss_workspace(int B, int N1, long long HW) {
  if (B <= 0 || N1 <= 0 || HW <= 0) return 0;
  return (size_t)B * N1 * cdiv(HW, 256) * 2 * sizeof(float);
}

extern "C" int swem_vos_loss_frame_fwd_f32(void *stream, const float *logits, const long long *label,
                                           long long label_bs, const float *valid, float *prob, float *raw, float *rowstat, float *iou,
                                           int B, int N1, long long HW, long long k, const long long *k_dev, void *ws,
                                           size_t ws_bytes) {
  SWEM_REQUIRE(logits && label && prob && raw && rowstat && iou, SWEM_E_ARG, "vos_loss_fwd: null pointer");
  SWEM_REQUIRE(B > 0 && N1 >= 2 && N1 <= LOSS_MAXC && HW > 0 && HW < (1ll << 31) && k >= 0 && k <= HW, SWEM_E_SHAPE,
               "vos_loss_fwd: B=%d N1=%d (2..%d) HW=%lld k=%lld", B, N1, LOSS_MAXC, HW, k);
  const size_t need = swem_vos_loss_workspace(B, N1, HW);
  SWEM_REQUIRE(ws && ws_bytes >= need, SWEM_E_WORKSPACE, "vos_loss_fwd: workspace %zu < %zu bytes", ws_bytes, need);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nblk = cdiv(HW, 256);
  float *part = static_cast<float *>(ws);
  hipLaunchKernelGGL(loss_pixel_fwd_kernel, dim3(nblk, B), dim3(256), 0, st, logits, label, valid, prob, raw, part, N1,
                     HW, nblk, label_bs);
  SWEM_CHECK_LAUNCH("loss_pixel_fwd_kernel");
  hipLaunchKernelGGL(loss_iou_reduce_kernel, dim3(B * N1), dim3(256), 0, st, part, iou, nblk);
  SWEM_CHECK_LAUNCH("loss_iou_reduce_kernel");
  hipLaunchKernelGGL(loss_row_select_kernel, dim3(B), dim3(1024), 0, st, raw, rowstat, HW, k, k_dev);
  SWEM_CHECK_LAUNCH("loss_row_select_kernel");
  return SWEM_OK;
}

extern "C" int swem_vos_loss_reduce_f32(void *stream, const float *rowstat, const float *iou, const float *valid,
                                        float *losses, int B, int N1, int T, long long HW, long long k,
                                        const long long *k_dev, float aux_ratio) {
  SWEM_REQUIRE(rowstat && iou && losses, SWEM_E_ARG, "vos_loss_reduce: null pointer");
  SWEM_REQUIRE(B > 0 && N1 >= 2 && N1 <= LOSS_MAXC && T > 0 && HW > 0, SWEM_E_SHAPE, "vos_loss_reduce: bad shape");
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), rowstat, iou, valid,
                     losses, B, N1, T, HW, k, k_dev, aux_ratio);
  SWEM_CHECK_LAUNCH("loss_reduce_kernel");
  return SWEM_OK;
}

extern "C" int swem_vos_loss_frame_bwd_f32(void *stream, const float *prob, const float *raw, const long long *label,
                                           long long label_bs, const float *valid, const float *rowstat, const float *iou, float *dlogits,
                                           int B, int N1, int T, long long HW, long long k, const long long *k_dev,
                                           float aux_ratio, const float *gout) {
  SWEM_REQUIRE(prob && raw && label && rowstat && iou && dlogits, SWEM_E_ARG, "vos_loss_bwd: null pointer");
  SWEM_REQUIRE(B > 0 && N1 >= 2 && N1 <= LOSS_MAXC && T > 0 && HW > 0 && HW < (1ll << 31), SWEM_E_SHAPE,
               "vos_loss_bwd: bad shape");
  hipLaunchKernelGGL(loss_pixel_bwd_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, static_cast<hipStream_t>(stream), prob,
                     raw, label, valid, rowstat, iou, dlogits, B, N1, T, HW, k, aux_ratio, gout, label_bs, k_dev);
  SWEM_CHECK_LAUNCH("loss_pixel_bwd_kernel");
  return SWEM_OK;
}

extern "C" int swem_adamw_gated_f32(void *stream, float *p, const float *g, float *m, float *v, long long n, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, int step, const float *gate,
                                    int ngate, int *applied) {
  SWEM_REQUIRE(p && g && m && v, SWEM_E_ARG, "adamw: null pointer");
  SWEM_REQUIRE(n > 0 && step >= 1, SWEM_E_SHAPE, "adamw: n=%lld step=%d", n, step);
  SWEM_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, SWEM_E_ARG,
               "adamw: buffers must be 16-byte aligned");
  SWEM_REQUIRE(ngate >= 0 && (gate || ngate == 0), SWEM_E_ARG, "adamw: %d gate flags without a gate pointer", ngate);
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adamw_kernel, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g,
                     m, v, n, (float)(1.0 - (double)lr * weight_decay), beta1, beta2, eps, (float)(lr / bc1),
                     (float)sqrt(bc2), ngate > 0 ? gate : nullptr, ngate, applied);
  SWEM_CHECK_LAUNCH("adamw_kernel");
  return SWEM_OK;
}

extern "C" int swem_adamw_f32(void *stream, float *p, const float *g, float *m, float *v, long long n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step) {
  return swem_adamw_gated_f32(stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, nullptr, 0, nullptr);
}

namespace {
__global__ void fault_flags_kernel(const unsigned *__restrict__ fault, float *__restrict__ out) {
  const unsigned w = *fault;
  out[0] = (w & SWEM_FAULT_RANGE) ? 1.f : 0.f;
  out[1] = (w & ~(unsigned)SWEM_FAULT_RANGE) ? 1.f : 0.f;
}
}  // namespace

extern "C" int swem_fault_flags_f32(void *stream, const unsigned *fault, float *flags2) {
  SWEM_REQUIRE(fault && flags2, SWEM_E_ARG, "fault_flags: null pointer");
  hipLaunchKernelGGL(fault_flags_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), fault, flags2);
  SWEM_CHECK_LAUNCH("fault_flags_kernel");
  return SWEM_OK;
}

// =====================================================================================================================
// Backward kernels of the pointwise stages (forward: pointwise.hip).  Gather form throughout: every output element is
// written by exactly one thread, sums run in a fixed order (deterministic, no atomics).
// =====================================================================================================================
namespace {

__device__ __forceinline__ float4 ld4t(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4t(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
inline dim3 grid1t(long long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

// same source-index rule as pointwise.hip (ATen area_pixel_compute_source_index, align_corners=False)
struct LerpT {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ LerpT lerp_coord_t(int dst, float scale, int in) {
  float src = scale * (dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  int i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  float l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
  LerpT r;
  r.i0 = i0;
  r.i1 = i0 + (i0 < in - 1 ? 1 : 0);
  r.l1 = l1;
  r.l0 = 1.f - l1;
  return r;
}
// weight with which destination coordinate d reads source coordinate s (0 if it does not)
__device__ __forceinline__ float lerp_weight(int d, float scale, int in, int s) {
  const LerpT l = lerp_coord_t(d, scale, in);
  return (l.i0 == s ? l.l0 : 0.f) + (l.i1 == s ? l.l1 : 0.f);
}
// destination range that can read source coordinate s
__device__ __forceinline__ void lerp_span(int s, float scale, int out, int &d0, int &d1) {
  const float inv = 1.f / scale;
  d0 = max(0, (int)floorf(((float)s - 1.f + 0.5f) * inv - 0.5f) - 1);
  d1 = min(out - 1, (int)ceilf(((float)s + 1.f + 0.5f) * inv - 0.5f) + 1);
}

// ---- frozen BatchNorm + residual + ReLU as its own stage (training keeps the raw conv output for the BN gradients)
// Optional second output of the two BatchNorm stages: the bf16 planes of the value just written, [plane][C/8][M][8]
// (swem_split_bf16x3_f32's layout and rounding), so that the convolution that consumes it -- forward: the next layer,
// backward: this layer's data / weight gradient -- finds its operand split already (one launch and one read of the
// fp32 map less per BatchNorm call).  A thread owns 4 consecutive channels of one row = half a 16-byte cell.
__device__ __forceinline__ void store_planes4(unsigned short *__restrict__ planes, long long M, int C, long long m, int c,
                                              float4 v, bool f16 = false) {
  const long long cell = ((long long)(c >> 3) * M + m) * 8 + (c & 4);
  const long long plane = M * C;
  if (f16) {   // the fp16 (hi, mid) pair of the f16x3 arithmetic (SWEM_PLANES_F16; the caller tests the range)
    uint2 h, md;
    split2h(v, h, md);
    *reinterpret_cast<uint2 *>(planes + cell) = h;
    *reinterpret_cast<uint2 *>(planes + plane + cell) = md;
    return;
  }
  uint2 h, md, l;
  split3(v, h, md, l);
  *reinterpret_cast<uint2 *>(planes + cell) = h;
  *reinterpret_cast<uint2 *>(planes + plane + cell) = md;
  *reinterpret_cast<uint2 *>(planes + 2 * plane + cell) = l;
}
__global__ __launch_bounds__(256) void bn_act_kernel(const float *__restrict__ c, const float *__restrict__ alpha,
                                                     const float *__restrict__ shift, const float *__restrict__ res,
                                                     float *__restrict__ y, long long M, int C, int relu,
                                                     unsigned short *__restrict__ planes, int planes_f16, unsigned *fault) {
  const int cq = C / 4;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * cq) {
    if (planes && planes_f16) range_fault(fault, 0u);   // (the ballot inside wants the whole wave)
    return;
  }
  const int c4 = (int)(i % cq);
  const float4 v = ld4t(c + i * 4), a = ld4t(alpha + c4 * 4), s = ld4t(shift + c4 * 4);
  float4 o = make_float4(v.x * a.x + s.x, v.y * a.y + s.y, v.z * a.z + s.z, v.w * a.w + s.w);
  if (res) {
    const float4 rv = ld4t(res + i * 4);
    o = make_float4(o.x + rv.x, o.y + rv.y, o.z + rv.z, o.w + rv.w);
  }
  const unsigned nan_in = f32_nan(o);   // (before the ReLU turns a NaN into a clean 0: ADVICE r05)
  if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
  st4t(y + i * 4, o);
  if (planes) {
    store_planes4(planes, M, C, i / cq, c4 * 4, o, planes_f16 != 0);
    if (planes_f16) range_fault(fault, f16_oor(o) | nan_in);
  }
}
// Backward of bn_act in one pass: dz = dy * (y > 0) (also the residual's gradient), dc = dz * alpha (the conv output's
// gradient), and per block the column partials s1 = sum dz, s2 = sum dz * c for the BatchNorm parameters.
// Block = 16 channel quads x 16 row lanes over BN_ROWS rows.
// Rows per block: enough blocks to fill the chip (a 192x192x64 stem map has ONE 64-channel column block), at most 512
// rows, a multiple of the 16 row lanes.
static inline int bn_rows(long long M, int C) {
  const long long cb = cdiv(C / 4, 16);
  long long rows = (M * cb + 1023) / 1024;
  rows = (rows + 15) / 16 * 16;
  return (int)(rows < 32 ? 32 : (rows > 512 ? 512 : rows));
}
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                         const float *__restrict__ c, const float *__restrict__ alpha,
                                                         float *__restrict__ dz, float *__restrict__ dc,
                                                         float *__restrict__ part, long long M, int C, int relu,
                                                         int BN_ROWS, unsigned short *__restrict__ planes,
                                                         float *__restrict__ amax_parts) {
  // amax_parts (optional): one float per block, the largest |dc| the block wrote -- the first pass of the scaled fp16 split of
  // this gradient map (swem_split_f16x2_scaled_f32 with nparts = the block count), for free
  __shared__ float4 sh1[256], sh2[256];
  unsigned amax = 0;
  const int cq = C / 4;
  const int c4 = blockIdx.x * 16 + (threadIdx.x & 15);
  const int rsub = threadIdx.x >> 4;
  const long long m0 = (long long)blockIdx.y * BN_ROWS, m1 = min(M, m0 + BN_ROWS);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (c4 < cq) {
    const float4 a = ld4t(alpha + c4 * 4);
#pragma unroll 2
    for (long long m = m0 + rsub; m < m1; m += 16) {
      const long long i = m * C + c4 * 4;
      float4 g = ld4t(dy + i);
      if (relu) {
        const float4 o = ld4t(y + i);
        g = make_float4(o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f);
      }
      if (dz) st4t(dz + i, g);
      const float4 gc = make_float4(g.x * a.x, g.y * a.y, g.z * a.z, g.w * a.w);
      st4t(dc + i, gc);
      if (planes) store_planes4(planes, M, C, m, c4 * 4, gc);
      amax = max(max(amax, __float_as_uint(gc.x) & 0x7fffffffu), max(__float_as_uint(gc.y) & 0x7fffffffu, __float_as_uint(gc.z) & 0x7fffffffu));
      amax = max(amax, __float_as_uint(gc.w) & 0x7fffffffu);
      if (part) {
        const float4 v = ld4t(c + i);
        s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
        s2.x += g.x * v.x; s2.y += g.y * v.y; s2.z += g.z * v.z; s2.w += g.w * v.w;
      }
    }
  }
  if (amax_parts) {   // (uniform branch: every thread of the block takes it)
    __shared__ unsigned sha[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o));
    if ((threadIdx.x & 63) == 0) sha[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0)
      amax_parts[blockIdx.y * gridDim.x + blockIdx.x] = __uint_as_float(max(max(sha[0], sha[1]), max(sha[2], sha[3])));
  }
  if (!part) return;
  sh1[threadIdx.x] = s1;
  sh2[threadIdx.x] = s2;
  __syncthreads();
  if (rsub == 0 && c4 < cq) {
    for (int j = 1; j < 16; ++j) {
      const float4 t1 = sh1[threadIdx.x + 16 * j], t2 = sh2[threadIdx.x + 16 * j];
      s1.x += t1.x; s1.y += t1.y; s1.z += t1.z; s1.w += t1.w;
      s2.x += t2.x; s2.y += t2.y; s2.z += t2.z; s2.w += t2.w;
    }
    float *dst = part + ((long long)blockIdx.y * 2) * C + c4 * 4;
    st4t(dst, s1);
    st4t(dst + C, s2);
  }
}
// dgamma += invstd * (s2 - mean * s1); dbeta += s1, with s1 / s2 summed over the row blocks in order
// (block = 16 channels x 16 row lanes; lane r sums rows r, r+16, ... in order, the lanes are combined in order)
__global__ __launch_bounds__(256) void bn_param_grad_kernel(const float *__restrict__ part, int nrow,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ invstd, float *__restrict__ dgamma,
                                                            float *__restrict__ dbeta, int C) {
  __shared__ float sh[2][16][16];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s1 = 0.f, s2 = 0.f;
  if (c < C)
    for (int j = rl; j < nrow; j += 16) {
      s1 += part[((long long)j * 2) * C + c];
      s2 += part[((long long)j * 2 + 1) * C + c];
    }
  sh[0][rl][cl] = s1;
  sh[1][rl][cl] = s2;
  __syncthreads();
  if (rl != 0 || c >= C) return;
  s1 = sh[0][0][cl];
  s2 = sh[1][0][cl];
  for (int j = 1; j < 16; ++j) {
    s1 += sh[0][j][cl];
    s2 += sh[1][j][cl];
  }
  if (dgamma) dgamma[c] += invstd[c] * (s2 - mean[c] * s1);
  if (dbeta) dbeta[c] += s1;
}

__global__ void bn_fold_kernel(const float *__restrict__ gamma, const float *__restrict__ beta,
                               const float *__restrict__ mean, const float *__restrict__ var, float eps,
                               float *__restrict__ alpha, float *__restrict__ shift, float *__restrict__ invstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.f / sqrtf(var[c] + eps);
  const float a = gamma[c] / sqrtf(var[c] + eps);
  alpha[c] = a;
  shift[c] = beta[c] - mean[c] * a;
  invstd[c] = is;
}

// ---- y = f * sigmoid(a) on two [M][C] maps (training runs layer_f / layer_a as two convolutions)
__global__ __launch_bounds__(256) void glu_kernel(const float *__restrict__ f, const float *__restrict__ a,
                                                  float *__restrict__ y, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 vf = ld4t(f + i * 4), va = ld4t(a + i * 4);
  st4t(y + i * 4, make_float4(vf.x * sigmoidf_(va.x), vf.y * sigmoidf_(va.y), vf.z * sigmoidf_(va.z),
                              vf.w * sigmoidf_(va.w)));
}
__global__ __launch_bounds__(256) void glu_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ f,
                                                      const float *__restrict__ a, float *__restrict__ df,
                                                      float *__restrict__ da, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 g = ld4t(dy + i * 4), vf = ld4t(f + i * 4), va = ld4t(a + i * 4);
  const float s0 = sigmoidf_(va.x), s1 = sigmoidf_(va.y), s2 = sigmoidf_(va.z), s3 = sigmoidf_(va.w);
  st4t(df + i * 4, make_float4(g.x * s0, g.y * s1, g.z * s2, g.w * s3));
  st4t(da + i * 4, make_float4(g.x * vf.x * s0 * (1.f - s0), g.y * vf.y * s1 * (1.f - s1), g.z * vf.z * s2 * (1.f - s2),
                               g.w * vf.w * s3 * (1.f - s3)));
}
// y = a + b (identity shortcut of a residual block when the kernel cannot fuse it)
__global__ __launch_bounds__(256) void add_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                  float *__restrict__ y, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 va = ld4t(a + i * 4), vb = ld4t(b + i * 4);
  st4t(y + i * 4, make_float4(va.x + vb.x, va.y + vb.y, va.z + vb.z, va.w + vb.w));
}

// ---- max_pool2d(3, 2, 1) backward: the gradient goes to the first maximum of each window (ATen scan order ky, kx)
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                          float *__restrict__ dx, int B, int H, int W, int C, int Ho,
                                                          int Wo) {
  const int cq = C / 4;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * H * W * cq) return;
  const int c4 = (int)(i % cq);
  long long t = i / cq;
  const int ix = (int)(t % W);
  t /= W;
  const int iy = (int)(t % H);
  const int b = (int)(t / H);
  const float4 me = ld4t(x + i * 4);
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  const float mv[4] = {me.x, me.y, me.z, me.w};
  // windows (oy, ox) containing (iy, ix): oy*2-1 <= iy <= oy*2+1
  for (int oy = max(0, (iy - 1 + 1) / 2); oy <= min(Ho - 1, (iy + 1) / 2); ++oy)
    for (int ox = max(0, (ix - 1 + 1) / 2); ox <= min(Wo - 1, (ix + 1) / 2); ++ox) {
      // position of this pixel in the window's scan order, and whether an earlier position holds a value >= ours
      bool win[4] = {true, true, true, true};
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = oy * 2 - 1 + ky;
        if ((unsigned)yy >= (unsigned)H) continue;
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = ox * 2 - 1 + kx;
          if ((unsigned)xx >= (unsigned)W) continue;
          if (yy == iy && xx == ix) continue;
          const float4 o = ld4t(x + (((long long)b * H + yy) * W + xx) * C + c4 * 4);
          const float ov[4] = {o.x, o.y, o.z, o.w};
          const bool earlier = yy < iy || (yy == iy && xx < ix);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (earlier ? ov[e] >= mv[e] : ov[e] > mv[e]) win[e] = false;
        }
      }
      const float4 d = ld4t(dy + (((long long)b * Ho + oy) * Wo + ox) * C + c4 * 4);
      const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (win[e]) g[e] += dv[e];
    }
  st4t(dx + i * 4, make_float4(g[0], g[1], g[2], g[3]));
}

// The same gradient when the forward OUTPUT y is at hand (round 6): a pixel takes a window's gradient iff it equals the window's
// maximum y and no EARLIER position of the window (ATen scan order ky, kx) does -- one 16-byte load of y and one of dy per window
// (<= 4 windows per pixel), the earlier positions looked at only where the pixel itself attains the maximum (one position in nine).
// maxpool_bwd_kernel above re-derives every window's maximum from its nine inputs: up to 36 loads per pixel, 126 us per launch on
// the training shape.  Same result bit for bit (finite inputs).
__global__ __launch_bounds__(256) void maxpool_bwd_y_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                            const float *__restrict__ dy, float *__restrict__ dx, int B, int H,
                                                            int W, int C, int Ho, int Wo) {
  const int cq = C / 4;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * H * W * cq) return;
  const int c4 = (int)(i % cq);
  long long t = i / cq;
  const int ix = (int)(t % W);
  t /= W;
  const int iy = (int)(t % H);
  const int b = (int)(t / H);
  const float4 me = ld4t(x + i * 4);
  const float mv[4] = {me.x, me.y, me.z, me.w};
  float g[4] = {0.f, 0.f, 0.f, 0.f};
  for (int oy = max(0, iy / 2); oy <= min(Ho - 1, (iy + 1) / 2); ++oy)
    for (int ox = max(0, ix / 2); ox <= min(Wo - 1, (ix + 1) / 2); ++ox) {
      const long long o = (((long long)b * Ho + oy) * Wo + ox) * C + c4 * 4;
      const float4 m = ld4t(y + o);
      const float mm[4] = {m.x, m.y, m.z, m.w};
      bool win[4];
      bool any = false;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        win[e] = mv[e] == mm[e];
        any |= win[e];
      }
      if (!any) continue;
      // an earlier position of this window that attains the maximum too takes the gradient instead
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = oy * 2 - 1 + ky;
        if ((unsigned)yy >= (unsigned)H || yy > iy) continue;
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = ox * 2 - 1 + kx;
          if ((unsigned)xx >= (unsigned)W) continue;
          if (!(yy < iy || xx < ix)) continue;      // (scan order: rows first)
          const float4 ov4 = ld4t(x + (((long long)b * H + yy) * W + xx) * C + c4 * 4);
          const float ov[4] = {ov4.x, ov4.y, ov4.z, ov4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ov[e] == mm[e]) win[e] = false;
        }
      }
      const float4 d = ld4t(dy + o);
      const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (win[e]) g[e] += dv[e];
    }
  st4t(dx + i * 4, make_float4(g[0], g[1], g[2], g[3]));
}

// ---- adjoint of the bilinear upsampling inside upsample_add: dlow[b][ly][lx][c] = sum_o w(o->l) dy[b][oy][ox][c]
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dlow, int B,
                                                           int Hl, int Wl, int Ho, int Wo, int C) {
  const int cq = C / 4;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * Hl * Wl * cq) return;
  const int c4 = (int)(i % cq);
  long long t = i / cq;
  const int lx = (int)(t % Wl);
  t /= Wl;
  const int ly = (int)(t % Hl);
  const int b = (int)(t / Hl);
  const float sy = (float)Hl / (float)Ho, sx = (float)Wl / (float)Wo;
  int y0, y1, x0, x1;
  lerp_span(ly, sy, Ho, y0, y1);
  lerp_span(lx, sx, Wo, x0, x1);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int oy = y0; oy <= y1; ++oy) {
    const float wy = lerp_weight(oy, sy, Hl, ly);
    if (wy == 0.f) continue;
    for (int ox = x0; ox <= x1; ++ox) {
      const float w = wy * lerp_weight(ox, sx, Wl, lx);
      if (w == 0.f) continue;
      const float4 d = ld4t(dy + (((long long)b * Ho + oy) * Wo + ox) * C + c4 * 4);
      acc.x += w * d.x; acc.y += w * d.y; acc.z += w * d.z; acc.w += w * d.w;
    }
  }
  st4t(dlow + i * 4, acc);
}
// the same adjoint on NCHW planes (decode head: 1/4-scale logit -> output size)
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float *__restrict__ dy, float *__restrict__ dx,
                                                                  int planes, int Hi, int Wi, int Ho, int Wo) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)planes * Hi * Wi) return;
  const int lx = (int)(i % Wi);
  long long t = i / Wi;
  const int ly = (int)(t % Hi);
  const long long pl = t / Hi;
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  int y0, y1, x0, x1;
  lerp_span(ly, sy, Ho, y0, y1);
  lerp_span(lx, sx, Wo, x0, x1);
  const float *src = dy + pl * Ho * Wo;
  float acc = 0.f;
  for (int oy = y0; oy <= y1; ++oy) {
    const float wy = lerp_weight(oy, sy, Hi, ly);
    if (wy == 0.f) continue;
    for (int ox = x0; ox <= x1; ++ox) acc += wy * lerp_weight(ox, sx, Wi, lx) * src[(long long)oy * Wo + ox];
  }
  dx[i] = acc;
}

// ---- decode head backward (swem.py:92-116): softmax -> aggregate (clamp, logit) -> valid -> sigmoid, per output pixel
// d_up[b][n][pix] = gradient at the bilinearly upsampled single-channel logit (resize_bilinear_bwd brings it to 1/4)
__global__ __launch_bounds__(256) void decode_head_bwd_kernel(const float *__restrict__ logit4,
                                                              const float *__restrict__ valid,
                                                              const float *__restrict__ dlogits,
                                                              const float *__restrict__ dprob, float *__restrict__ dup,
                                                              int B, int N, int h4, int w4, int Ho, int Wo) {
  constexpr int MAXN = 7;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long HW = (long long)Ho * Wo;
  if (i >= B * HW) return;
  const int b = (int)(i / HW);
  const long long pix = i - b * HW;
  const int oy = (int)(pix / Wo), ox = (int)(pix - (long long)oy * Wo);
  const LerpT ly = lerp_coord_t(oy, (float)h4 / (float)Ho, h4), lx = lerp_coord_t(ox, (float)w4 / (float)Wo, w4);
  float sg[MAXN], p[MAXN], lg[MAXN + 1];
  float bg = 1.f, mx = -INFINITY;
  for (int n = 0; n < N; ++n) {
    const float *src = logit4 + ((long long)b * N + n) * h4 * w4;
    const float r0 = lx.l0 * src[ly.i0 * w4 + lx.i0] + lx.l1 * src[ly.i0 * w4 + lx.i1];
    const float r1 = lx.l0 * src[ly.i1 * w4 + lx.i0] + lx.l1 * src[ly.i1 * w4 + lx.i1];
    sg[n] = sigmoidf_(ly.l0 * r0 + ly.l1 * r1);
    p[n] = valid ? sg[n] * valid[(long long)b * (N + 1) + n + 1] : sg[n];
    bg *= 1.f - p[n];
    const float pc = fminf(fmaxf(p[n], 1e-7f), 1.f - 1e-7f);
    lg[n + 1] = logf(pc / (1.f - pc));
    mx = fmaxf(mx, lg[n + 1]);
  }
  const float bgc = fminf(fmaxf(bg, 1e-7f), 1.f - 1e-7f);
  lg[0] = logf(bgc / (1.f - bgc));
  mx = fmaxf(mx, lg[0]);
  // total gradient at the aggregated logits: the loss's plus the softmax Jacobian applied to d prob
  float dl[MAXN + 1];
  float sum = 0.f;
  for (int n = 0; n <= N; ++n) sum += expf(lg[n] - mx);
  float dot = 0.f;
  for (int n = 0; n <= N; ++n) {
    const float q = expf(lg[n] - mx) / sum;
    const float dq = dprob ? dprob[((long long)b * (N + 1) + n) * HW + pix] : 0.f;
    dl[n] = q;           // stash the probability
    dot += q * dq;
  }
  for (int n = 0; n <= N; ++n) {
    const float q = dl[n];
    const float dq = dprob ? dprob[((long long)b * (N + 1) + n) * HW + pix] : 0.f;
    dl[n] = (dlogits ? dlogits[((long long)b * (N + 1) + n) * HW + pix] : 0.f) + q * (dq - dot);
  }
  // background: lg0 = logit(clamp(prod(1 - p)));  clamp passes the gradient inside [1e-7, 1 - 1e-7] (inclusive)
  const float dbg = (bg >= 1e-7f && bg <= 1.f - 1e-7f) ? dl[0] / (bgc * (1.f - bgc)) : 0.f;
  for (int n = 0; n < N; ++n) {
    const float pc = fminf(fmaxf(p[n], 1e-7f), 1.f - 1e-7f);
    float dp = (p[n] >= 1e-7f && p[n] <= 1.f - 1e-7f) ? dl[n + 1] / (pc * (1.f - pc)) : 0.f;
    float others = 1.f;
    for (int j = 0; j < N; ++j)
      if (j != n) others *= 1.f - p[j];
    dp -= dbg * others;
    const float v = valid ? valid[(long long)b * (N + 1) + n + 1] : 1.f;
    dup[((long long)b * N + n) * HW + pix] = dp * v * sg[n] * (1.f - sg[n]);
  }
}

// ---- prediction head backward: logit = conv3x3(relu(x)) -> 1 channel
// dx[pix][c] = (x > 0) * sum_taps dlogit[pix - tap] * w[tap][c]
__global__ __launch_bounds__(256) void pred_head_bwd_x_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ dlogit, float *__restrict__ dx,
                                                              int B, int H, int W, int C) {
  const int cq = C / 4;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)B * H * W * cq) return;
  const int c4 = (int)(i % cq);
  long long t = i / cq;
  const int ix = (int)(t % W);
  t /= W;
  const int iy = (int)(t % H);
  const int b = (int)(t / H);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ky = 0; ky < 3; ++ky) {
    const int oy = iy + 1 - ky;
    if ((unsigned)oy >= (unsigned)H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ox = ix + 1 - kx;
      if ((unsigned)ox >= (unsigned)W) continue;
      const float d = dlogit[((long long)b * H + oy) * W + ox];
      const float4 q = ld4t(w + (ky * 3 + kx) * C + c4 * 4);
      acc.x += d * q.x; acc.y += d * q.y; acc.z += d * q.z; acc.w += d * q.w;
    }
  }
  const float4 v = ld4t(x + i * 4);
  st4t(dx + i * 4, make_float4(v.x > 0.f ? acc.x : 0.f, v.y > 0.f ? acc.y : 0.f, v.z > 0.f ? acc.z : 0.f,
                               v.w > 0.f ? acc.w : 0.f));
}
// dw[tap][c] partial sums over a chunk of pixels; part [chunks][9][C].
// Round 6: a block walks the INPUT pixels of its chunk -- thread = (pixel lane, channel quad), 256 / (C/4) pixel lanes -- reads each
// relu(x) quad ONCE and feeds all nine taps from the nine neighbouring dlogit values (wave-uniform scalars); the lanes' partial sums
// are added in lane order through the LDS.  (Rounds 1-5: thread = (tap, quad) walking all 256 OUTPUT pixels of a chunk with 64-bit
// divisions per pixel: x read nine times, 72 blocks on the training shape, 288 us per launch -- the slowest launch of the step.)
constexpr int PH_CHUNK = 64;
__global__ __launch_bounds__(256) void pred_head_bwd_w_kernel(const float *__restrict__ x,
                                                              const float *__restrict__ dlogit,
                                                              float *__restrict__ part, int B, int H, int W, int C) {
  extern __shared__ float phw_s[];   // [lanes][9][C]
  const int cq = C / 4;
  const int lanes = 256 / cq;        // (host: cq divides 256)
  const int c4 = threadIdx.x % cq, pl = threadIdx.x / cq;
  const int npix = B * H * W;
  const int p0 = blockIdx.x * PH_CHUNK, p1 = min(npix, p0 + PH_CHUNK);
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int pix = p0 + pl; pix < p1; pix += lanes) {
    const int ix = pix % W, t2 = pix / W;
    const int iy = t2 % H, b = t2 / H;
    float4 v = ld4t(x + (long long)pix * C + c4 * 4);
    v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    const float *dl = dlogit + (long long)b * H * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int oy = iy + 1 - ky;          // the output pixel whose tap (ky, kx) reads this input pixel
      if ((unsigned)oy >= (unsigned)H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ox = ix + 1 - kx;
        if ((unsigned)ox >= (unsigned)W) continue;
        const float d = dl[oy * W + ox];
        float4 &a = acc[ky * 3 + kx];
        a.x += d * v.x; a.y += d * v.y; a.z += d * v.z; a.w += d * v.w;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) st4t(phw_s + ((long long)pl * 9 + t) * C + c4 * 4, acc[t]);
  __syncthreads();
  for (int item = threadIdx.x; item < 9 * C; item += 256) {
    float s = phw_s[item];
    for (int l = 1; l < lanes; ++l) s += phw_s[(long long)l * 9 * C + item];
    part[(long long)blockIdx.x * 9 * C + item] = s;
  }
}
// dw (OIHW [1][C][3][3]) += sum of the chunk partials; db += sum dlogit
__global__ __launch_bounds__(256) void pred_head_bwd_reduce_kernel(const float *__restrict__ part,
                                                                   const float *__restrict__ dlogit,
                                                                   float *__restrict__ dw, float *__restrict__ db,
                                                                   int nchunk, long long npix, int C) {
  __shared__ double sh[4];
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x < (unsigned)((9 * C + 255) / 256)) {
    if (item < 9 * C) {
      const int tap = item / C, c = item - tap * C;
      float s = 0.f;
      for (int j = 0; j < nchunk; ++j) s += part[((long long)j * 9 + tap) * C + c];
      dw[(long long)c * 9 + tap] += s;
    }
    return;
  }
  double s = 0.0;   // last block: the bias
  for (long long i = threadIdx.x; i < npix; i += 256) s += dlogit[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) db[0] += (float)s;
}

// ---- value-encoder input packing backward: channels [img(3), m, other = 1 - m - m_bg] of the padded 8
__global__ __launch_bounds__(256) void prep_value_bwd_kernel(const float *__restrict__ dxin, float *__restrict__ dmasks,
                                                             int B, int N, long long HW, int single_obj) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * HW) return;
  const int b = (int)(i / HW);
  const long long p = i - b * HW;
  float dbg = 0.f;
  for (int n = 0; n < N; ++n) {
    const float *g = dxin + (((long long)b * N + n) * HW + p) * 8;
    const float dm = g[3], dother = single_obj ? 0.f : g[4];
    dmasks[((long long)b * (N + 1) + n + 1) * HW + p] = dm - dother;
    dbg -= dother;
  }
  dmasks[((long long)b * (N + 1)) * HW + p] = dbg;
}

}  // namespace

#define STT static_cast<hipStream_t>(stream)

extern "C" int swem_bn_act_planes_f32(void *stream, const float *c, const float *alpha, const float *shift, const float *res,
                                      float *y, long long M, int C, int relu, void *planes, int nplanes, void *fault) {
  SWEM_REQUIRE(c && alpha && shift && y && C % 4 == 0 && M > 0, SWEM_E_ARG, "bn_act: bad argument");
  SWEM_REQUIRE(!planes || C % 8 == 0, SWEM_E_SHAPE, "bn_act: the operand planes need C %% 8 == 0");
  SWEM_REQUIRE(!planes || nplanes == 3 || nplanes == SWEM_PLANES_F16, SWEM_E_ARG, "bn_act: three bf16 planes or SWEM_PLANES_F16");
  hipLaunchKernelGGL(bn_act_kernel, grid1t(M * (C / 4)), dim3(256), 0, STT, c, alpha, shift, res, y, M, C, relu,
                     static_cast<unsigned short *>(planes), nplanes == SWEM_PLANES_F16 ? 1 : 0, static_cast<unsigned *>(fault));
  SWEM_CHECK_LAUNCH("bn_act_kernel");
  return SWEM_OK;
}
extern "C" int swem_bn_act_f32(void *stream, const float *c, const float *alpha, const float *shift, const float *res,
                               float *y, long long M, int C, int relu, void *planes) {
  return swem_bn_act_planes_f32(stream, c, alpha, shift, res, y, M, C, relu, planes, 3, nullptr);
}
extern "C" size_t swem_bn_act_bwd_workspace(long long M, int C) {
  if (M <= 0 || C <= 0) return 0;
  return (size_t)cdiv(M, bn_rows(M, C)) * 2 * C * sizeof(float);
}
extern "C" int swem_bn_act_bwd_amax_parts(long long M, int C) {
  if (M <= 0 || C <= 0) return 0;
  return cdiv(C / 4, 16) * cdiv(M, bn_rows(M, C));
}
static int bn_act_bwd_impl(void *stream, const float *dy, const float *y, const float *c, const float *alpha,
                           const float *mean, const float *invstd, float *dz, float *dc, float *dgamma,
                           float *dbeta, long long M, int C, int relu, void *planes, void *ws,
                           size_t ws_bytes, float *amax_parts);
extern "C" int swem_bn_act_bwd_f32(void *stream, const float *dy, const float *y, const float *c, const float *alpha,
                                   const float *mean, const float *invstd, float *dz, float *dc, float *dgamma,
                                   float *dbeta, long long M, int C, int relu, void *planes, void *ws,
                                   size_t ws_bytes) {
  return bn_act_bwd_impl(stream, dy, y, c, alpha, mean, invstd, dz, dc, dgamma, dbeta, M, C, relu, planes, ws, ws_bytes, nullptr);
}
extern "C" int swem_bn_act_bwd_amax_f32(void *stream, const float *dy, const float *y, const float *c, const float *alpha,
                                        const float *mean, const float *invstd, float *dz, float *dc, float *dgamma,
                                        float *dbeta, long long M, int C, int relu, float *amax_parts, void *ws,
                                        size_t ws_bytes) {
  SWEM_REQUIRE(amax_parts, SWEM_E_ARG, "bn_act_bwd_amax: null amax_parts");
  return bn_act_bwd_impl(stream, dy, y, c, alpha, mean, invstd, dz, dc, dgamma, dbeta, M, C, relu, nullptr, ws, ws_bytes, amax_parts);
}
static int bn_act_bwd_impl(void *stream, const float *dy, const float *y, const float *c, const float *alpha,
                           const float *mean, const float *invstd, float *dz, float *dc, float *dgamma,
                           float *dbeta, long long M, int C, int relu, void *planes, void *ws,
                           size_t ws_bytes, float *amax_parts) {
  SWEM_REQUIRE(dy && alpha && dc && (y || !relu) && C % 4 == 0 && M > 0, SWEM_E_ARG, "bn_act_bwd: bad argument");
  SWEM_REQUIRE(!planes || C % 8 == 0, SWEM_E_SHAPE, "bn_act_bwd: the bf16 planes need C %% 8 == 0");
  const bool params = dgamma || dbeta;
  SWEM_REQUIRE(!params || (c && mean && invstd), SWEM_E_ARG, "bn_act_bwd: parameter gradients need c, mean, invstd");
  const int rows = bn_rows(M, C), nrow = cdiv(M, rows);
  float *part = nullptr;
  if (params) {
    const size_t need = swem_bn_act_bwd_workspace(M, C);
    SWEM_REQUIRE(ws && ws_bytes >= need, SWEM_E_WORKSPACE, "bn_act_bwd: workspace %zu < %zu bytes", ws_bytes, need);
    part = static_cast<float *>(ws);
  }
  hipLaunchKernelGGL(bn_act_bwd_kernel, dim3(cdiv(C / 4, 16), nrow), dim3(256), 0, STT, dy, y, c, alpha, dz, dc, part, M,
                     C, relu, rows, static_cast<unsigned short *>(planes), amax_parts);
  SWEM_CHECK_LAUNCH("bn_act_bwd_kernel");
  if (params) {
    hipLaunchKernelGGL(bn_param_grad_kernel, dim3(cdiv(C, 16)), dim3(256), 0, STT, part, nrow, mean, invstd, dgamma,
                       dbeta, C);
    SWEM_CHECK_LAUNCH("bn_param_grad_kernel");
  }
  return SWEM_OK;
}
extern "C" int swem_bn_fold_f32(void *stream, const float *gamma, const float *beta, const float *mean,
                                const float *var, float eps, float *alpha, float *shift, float *invstd, int C) {
  SWEM_REQUIRE(gamma && beta && mean && var && alpha && shift && invstd && C > 0, SWEM_E_ARG, "bn_fold: bad argument");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, STT, gamma, beta, mean, var, eps, alpha, shift,
                     invstd, C);
  SWEM_CHECK_LAUNCH("bn_fold_kernel");
  return SWEM_OK;
}
extern "C" int swem_glu_f32(void *stream, const float *f, const float *a, float *y, long long n) {
  SWEM_REQUIRE(f && a && y && n > 0 && n % 4 == 0, SWEM_E_ARG, "glu: bad argument");
  hipLaunchKernelGGL(glu_kernel, grid1t(n / 4), dim3(256), 0, STT, f, a, y, n / 4);
  SWEM_CHECK_LAUNCH("glu_kernel");
  return SWEM_OK;
}
extern "C" int swem_glu_bwd_f32(void *stream, const float *dy, const float *f, const float *a, float *df, float *da,
                                long long n) {
  SWEM_REQUIRE(dy && f && a && df && da && n > 0 && n % 4 == 0, SWEM_E_ARG, "glu_bwd: bad argument");
  hipLaunchKernelGGL(glu_bwd_kernel, grid1t(n / 4), dim3(256), 0, STT, dy, f, a, df, da, n / 4);
  SWEM_CHECK_LAUNCH("glu_bwd_kernel");
  return SWEM_OK;
}
extern "C" int swem_add_f32(void *stream, const float *a, const float *b, float *y, long long n) {
  SWEM_REQUIRE(a && b && y && n > 0 && n % 4 == 0, SWEM_E_ARG, "add: bad argument");
  hipLaunchKernelGGL(add_kernel, grid1t(n / 4), dim3(256), 0, STT, a, b, y, n / 4);
  SWEM_CHECK_LAUNCH("add_kernel");
  return SWEM_OK;
}
extern "C" int swem_maxpool3x3s2_bwd_f32(void *stream, const float *x, const float *dy, float *dx, int B, int H, int W,
                                         int C) {
  SWEM_REQUIRE(x && dy && dx && C % 4 == 0, SWEM_E_ARG, "maxpool_bwd: bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd_kernel, grid1t((long long)B * H * W * (C / 4)), dim3(256), 0, STT, x, dy, dx, B, H, W,
                     C, Ho, Wo);
  SWEM_CHECK_LAUNCH("maxpool_bwd_kernel");
  return SWEM_OK;
}
extern "C" int swem_maxpool3x3s2_bwd_y_f32(void *stream, const float *x, const float *y, const float *dy, float *dx, int B, int H,
                                           int W, int C) {
  SWEM_REQUIRE(x && y && dy && dx && C % 4 == 0, SWEM_E_ARG, "maxpool_bwd_y: bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_bwd_y_kernel, grid1t((long long)B * H * W * (C / 4)), dim3(256), 0, STT, x, y, dy, dx, B, H, W, C,
                     Ho, Wo);
  SWEM_CHECK_LAUNCH("maxpool_bwd_y_kernel");
  return SWEM_OK;
}
extern "C" int swem_upsample_bwd_nhwc_f32(void *stream, const float *dy, float *dlow, int B, int Hl, int Wl, int Ho,
                                          int Wo, int C) {
  SWEM_REQUIRE(dy && dlow && C % 4 == 0 && Ho >= Hl && Wo >= Wl, SWEM_E_ARG, "upsample_bwd: bad argument");
  hipLaunchKernelGGL(upsample_bwd_kernel, grid1t((long long)B * Hl * Wl * (C / 4)), dim3(256), 0, STT, dy, dlow, B, Hl,
                     Wl, Ho, Wo, C);
  SWEM_CHECK_LAUNCH("upsample_bwd_kernel");
  return SWEM_OK;
}
extern "C" int swem_resize_bilinear_bwd_f32(void *stream, const float *dy, float *dx, int planes, int Hi, int Wi,
                                            int Ho, int Wo) {
  SWEM_REQUIRE(dy && dx && planes > 0 && Ho >= Hi && Wo >= Wi, SWEM_E_ARG,
               "resize_bilinear_bwd: bad argument (upsampling only)");
  hipLaunchKernelGGL(resize_bilinear_bwd_kernel, grid1t((long long)planes * Hi * Wi), dim3(256), 0, STT, dy, dx, planes,
                     Hi, Wi, Ho, Wo);
  SWEM_CHECK_LAUNCH("resize_bilinear_bwd_kernel");
  return SWEM_OK;
}
extern "C" int swem_decode_head_bwd_f32(void *stream, const float *logit4, const float *valid, const float *dlogits,
                                        const float *dprob, float *dlogit4, int B, int N, int h4, int w4, int Ho,
                                        int Wo, void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(logit4 && dlogit4 && (dlogits || dprob) && N > 0 && N <= 7, SWEM_E_ARG, "decode_head_bwd: bad argument");
  const size_t need = (size_t)B * N * Ho * Wo * sizeof(float);
  SWEM_REQUIRE(ws && ws_bytes >= need, SWEM_E_WORKSPACE, "decode_head_bwd: workspace %zu < %zu bytes", ws_bytes, need);
  float *dup = static_cast<float *>(ws);
  hipLaunchKernelGGL(decode_head_bwd_kernel, grid1t((long long)B * Ho * Wo), dim3(256), 0, STT, logit4, valid, dlogits,
                     dprob, dup, B, N, h4, w4, Ho, Wo);
  SWEM_CHECK_LAUNCH("decode_head_bwd_kernel");
  hipLaunchKernelGGL(resize_bilinear_bwd_kernel, grid1t((long long)B * N * h4 * w4), dim3(256), 0, STT, dup, dlogit4,
                     B * N, h4, w4, Ho, Wo);
  SWEM_CHECK_LAUNCH("resize_bilinear_bwd_kernel");
  return SWEM_OK;
}
extern "C" size_t swem_pred_head_bwd_workspace(int B, int H, int W, int C) {
  return (size_t)cdiv((long long)B * H * W, PH_CHUNK) * 9 * C * sizeof(float);
}
extern "C" int swem_pred_head_bwd_f32(void *stream, const float *x, const float *w, const float *dlogit, float *dx,
                                      float *dw, float *db, int B, int H, int W, int C, void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(x && w && dlogit && dx && dw && db && C % 4 == 0, SWEM_E_ARG, "pred_head_bwd: bad argument");
  const size_t need = swem_pred_head_bwd_workspace(B, H, W, C);
  SWEM_REQUIRE(ws && ws_bytes >= need, SWEM_E_WORKSPACE, "pred_head_bwd: workspace %zu < %zu bytes", ws_bytes, need);
  const long long npix = (long long)B * H * W;
  const int nchunk = cdiv(npix, PH_CHUNK);
  float *part = static_cast<float *>(ws);
  hipLaunchKernelGGL(pred_head_bwd_x_kernel, grid1t(npix * (C / 4)), dim3(256), 0, STT, x, w, dlogit, dx, B, H, W, C);
  SWEM_CHECK_LAUNCH("pred_head_bwd_x_kernel");
  SWEM_REQUIRE(C / 4 <= 256 && 256 % (C / 4) == 0, SWEM_E_SHAPE, "pred_head_bwd: C / 4 must divide 256 (C = %d)", C);
  hipLaunchKernelGGL(pred_head_bwd_w_kernel, dim3(nchunk), dim3(256), (size_t)(256 / (C / 4)) * 9 * C * sizeof(float), STT, x, dlogit,
                     part, B, H, W, C);
  SWEM_CHECK_LAUNCH("pred_head_bwd_w_kernel");
  hipLaunchKernelGGL(pred_head_bwd_reduce_kernel, dim3(cdiv(9 * C, 256) + 1), dim3(256), 0, STT, part, dlogit, dw, db,
                     nchunk, npix, C);
  SWEM_CHECK_LAUNCH("pred_head_bwd_reduce_kernel");
  return SWEM_OK;
}
extern "C" int swem_prep_value_input_bwd_f32(void *stream, const float *dxin, float *dmasks, int B, int N, int H, int W,
                                             int single_obj) {
  SWEM_REQUIRE(dxin && dmasks && N > 0, SWEM_E_ARG, "prep_value_input_bwd: bad argument");
  hipLaunchKernelGGL(prep_value_bwd_kernel, grid1t((long long)B * H * W), dim3(256), 0, STT, dxin, dmasks, B, N,
                     (long long)H * W, single_obj);
  SWEM_CHECK_LAUNCH("prep_value_bwd_kernel");
  return SWEM_OK;
}
