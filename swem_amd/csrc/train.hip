// Training step of the SWEM path on gfx950: loss (BootstrappedCE + mask IoU) forward / backward and the AdamW update.
// C ABI: include/swem_hip_train.h.  Everything here is HBM-bound single-pass work; reductions that feed a scalar are
// accumulated in fp64 in a fixed order (deterministic, and closer to the exact sum than any fp32 order).
#include <math.h>

#include "../../include/swem_hip_train.h"
#include "common.h"

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
                                                               long long k) {
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
      for (long long i = tid; i < HW; i += 1024) {
        const unsigned key = fkey(row[i]);
        if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
      }
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
                                   long long HW, long long k, float aux_ratio) {
  if (threadIdx.x != 0) return;
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
                                                             int B, int N1, int T, long long HW, long long k,
                                                             float aux_ratio, const float *__restrict__ gout,
                                                             long long label_bs) {
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

__global__ __launch_bounds__(256) void adamw_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                    float *__restrict__ m, float *__restrict__ v, long long n, float decay,
                                                    float b1, float b2, float eps, float step_size, float sqrt_bc2) {
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
                                           int B, int N1, long long HW, long long k, void *ws, size_t ws_bytes) {
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
  hipLaunchKernelGGL(loss_row_select_kernel, dim3(B), dim3(1024), 0, st, raw, rowstat, HW, k);
  SWEM_CHECK_LAUNCH("loss_row_select_kernel");
  return SWEM_OK;
}

extern "C" int swem_vos_loss_reduce_f32(void *stream, const float *rowstat, const float *iou, const float *valid,
                                        float *losses, int B, int N1, int T, long long HW, long long k,
                                        float aux_ratio) {
  SWEM_REQUIRE(rowstat && iou && losses, SWEM_E_ARG, "vos_loss_reduce: null pointer");
  SWEM_REQUIRE(B > 0 && N1 >= 2 && N1 <= LOSS_MAXC && T > 0 && HW > 0, SWEM_E_SHAPE, "vos_loss_reduce: bad shape");
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), rowstat, iou, valid,
                     losses, B, N1, T, HW, k, aux_ratio);
  SWEM_CHECK_LAUNCH("loss_reduce_kernel");
  return SWEM_OK;
}

extern "C" int swem_vos_loss_frame_bwd_f32(void *stream, const float *prob, const float *raw, const long long *label,
                                           long long label_bs, const float *valid, const float *rowstat, const float *iou, float *dlogits,
                                           int B, int N1, int T, long long HW, long long k, float aux_ratio,
                                           const float *gout) {
  SWEM_REQUIRE(prob && raw && label && rowstat && iou && dlogits, SWEM_E_ARG, "vos_loss_bwd: null pointer");
  SWEM_REQUIRE(B > 0 && N1 >= 2 && N1 <= LOSS_MAXC && T > 0 && HW > 0 && HW < (1ll << 31), SWEM_E_SHAPE,
               "vos_loss_bwd: bad shape");
  hipLaunchKernelGGL(loss_pixel_bwd_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, static_cast<hipStream_t>(stream), prob,
                     raw, label, valid, rowstat, iou, dlogits, B, N1, T, HW, k, aux_ratio, gout, label_bs);
  SWEM_CHECK_LAUNCH("loss_pixel_bwd_kernel");
  return SWEM_OK;
}

extern "C" int swem_adamw_f32(void *stream, float *p, const float *g, float *m, float *v, long long n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step) {
  SWEM_REQUIRE(p && g && m && v, SWEM_E_ARG, "adamw: null pointer");
  SWEM_REQUIRE(n > 0 && step >= 1, SWEM_E_SHAPE, "adamw: n=%lld step=%d", n, step);
  SWEM_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0, SWEM_E_ARG,
               "adamw: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adamw_kernel, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, static_cast<hipStream_t>(stream), p, g,
                     m, v, n, (float)(1.0 - (double)lr * weight_decay), beta1, beta2, eps, (float)(lr / bc1),
                     (float)sqrt(bc2));
  SWEM_CHECK_LAUNCH("adamw_kernel");
  return SWEM_OK;
}
