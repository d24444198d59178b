// Matching (reference methods/SWEM/modules.py:198-208, 232-289) as ONE kernel per frame plus a small
// memory-packing step.  Nothing of size (bases x pixels) ever reaches HBM.
//
// prep   : the two banks ('first', 'update') of each object are l2-normalised over C and packed as
//          mkn [N][Ltot][C] (row = class*Lm + bank*L + l, the order of get_mem + flatten, modules.py:266,304),
//          and the value bases as mvp [N][V][Ltot] (modules.py:272).
// match  : block = (object, 32-pixel tile), 4 waves, the whole exp-affinity tile e[Ltot][32] in LDS (132 KiB at
//          Ltot = 1024, row stride 33 floats so column AND row walks are conflict free):
//   1. stage the raw query keys, l2-normalise each pixel in LDS (modules.py:282);
//   2. affinity  mkn . q  on v_mfma_f32_32x32x2_f32 with the pixel on the lane and the base in the accumulator
//      registers, so the joint {bg,fg} max and the exp-sum are in-register + one LDS exchange (modules.py:247-250);
//   3. readout  mvp . e  (second MFMA chain, B operand read from the LDS tile), divided by the exp-sum in the
//      epilogue (modules.py:265-266, 272-274) and stored NHWC;
//   4. top-l features: each wave takes 8 pixels; per (pixel, class) the Lm exp values are sorted across the wave
//      (64-lane bitonic network on J = Lm/64 registers, pairwise top-64 merges), a wave scan gives the prefix sums,
//      feat = c_bg / (c_bg + c_fg) and its complement are stored as 2*topl channels (modules.py:198-208).
#include "common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// mvp[n][v][cls*Lm + off + l] = nu[n][cls][v][l]
__global__ void pack_values_kernel(const float *__restrict__ nu, float *__restrict__ mvp, int N, int V, int L, int Lm,
                                   int off) {
  const int lq = L / 4;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * 2 * V * lq) return;
  int l4 = (int)(i % lq);
  long long t = i / lq;
  int v = (int)(t % V);
  t /= V;
  int cls = (int)(t & 1);
  int n = (int)(t >> 1);
  float4 val = ld4(nu + i * 4);
  *reinterpret_cast<float4 *>(mvp + ((long long)n * V + v) * (2 * Lm) + cls * Lm + off + l4 * 4) = val;
}

__device__ __forceinline__ float sort64_desc(float v, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      float o = __shfl_xor(v, j);
      bool desc = (lane & k) == 0;
      bool lower = (lane & j) == 0;
      v = (lower == desc) ? fmaxf(v, o) : fminf(v, o);
    }
  }
  return v;
}
// a, b descending across the wave -> the 64 largest of their union, descending
__device__ __forceinline__ float merge_top64(float a, float b, int lane) {
  float v = fmaxf(a, __shfl(b, 63 - lane));
#pragma unroll
  for (int j = 32; j > 0; j >>= 1) {
    float o = __shfl_xor(v, j);
    v = ((lane & j) == 0) ? fmaxf(v, o) : fminf(v, o);
  }
  return v;
}

template <int J>  // Lm = 64*J bases per class; Ltot = 128*J
__global__ __launch_bounds__(256) void match_kernel(const float *__restrict__ qk, const float *__restrict__ mkn,
                                                    const float *__restrict__ mvp, float *__restrict__ mem_out,
                                                    float *__restrict__ S, int C, int V, int P, int topl, float tau) {
  constexpr int Ltot = 128 * J, Lm = 64 * J;
  constexpr int ES = 33;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int QS = C + 4;
  float *et = sm;                 // [Ltot][33]
  float *qs = et + Ltot * ES;     // [32][C+4]
  float *red = qs + 32 * QS;      // [2][4][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n = blockIdx.y, p0 = blockIdx.x * 32;
  const int cq = C / 4;
  // 1. stage + normalise the query tile
  for (int idx = tid; idx < 32 * cq; idx += 256) {
    int row = idx / cq, c4 = idx - row * cq;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p0 + row < P) v = ld4(qk + (long long)(p0 + row) * C + c4 * 4);
    *reinterpret_cast<float4 *>(qs + row * QS + c4 * 4) = v;
  }
  __syncthreads();
  for (int rr = 0; rr < 8; ++rr) {
    int row = wave * 8 + rr;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
      float v = qs[row * QS + c];
      s += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float den = sqrtf(s) + SWEM_L2_EPS;
    for (int c = lane; c < C; c += 64) qs[row * QS + c] /= den;
  }
  __syncthreads();
  // 2. affinity: this wave owns bases [wave*32J, +32J)
  f32x16 acc[J];
#pragma unroll
  for (int t = 0; t < J; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  {
    const float *krow = mkn + ((long long)n * Ltot + wave * 32 * J + r) * C + 4 * h;
    const float *qrow = qs + r * QS + 4 * h;
    for (int j = 0; j < C / 8; ++j) {
      float4 b4 = *reinterpret_cast<const float4 *>(qrow + 8 * j);
#pragma unroll
      for (int t = 0; t < J; ++t) acc[t] = mfma32x4(ld4(krow + (long long)t * 32 * C + 8 * j), b4, acc[t]);
    }
  }
  float m = -__builtin_huge_valf();
#pragma unroll
  for (int t = 0; t < J; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) m = fmaxf(m, acc[t][e]);
  m = fmaxf(m, __shfl_xor(m, 32));
  if (h == 0) red[wave * 32 + r] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[r], red[32 + r]), fmaxf(red[64 + r], red[96 + r]));
  float se = 0.f;
#pragma unroll
  for (int t = 0; t < J; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = expf((acc[t][e] - m) / tau);
      se += v;
      et[(wave * 32 * J + 32 * t + acc_row(e, h)) * ES + r] = v;
    }
  se += __shfl_xor(se, 32);
  if (h == 0) red[128 + wave * 32 + r] = se;
  __syncthreads();
  const float esum = (red[128 + r] + red[128 + 32 + r]) + (red[128 + 64 + r] + red[128 + 96 + r]);
  // 3. readout: 128 value rows per pass, one 32x32 tile per wave
  const bool pin = p0 + r < P;
  for (int vc = 0; vc < V; vc += 128) {
    f32x16 o;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
    const float *vrow = mvp + ((long long)n * V + vc + wave * 32 + r) * Ltot + 4 * h;
    const float *ecol = et + (4 * h) * ES + r;
    for (int j = 0; j < Ltot / 8; ++j) {
      float4 a4 = ld4(vrow + 8 * j);
      const float *ep = ecol + (8 * j) * ES;
      float4 b4 = make_float4(ep[0], ep[ES], ep[2 * ES], ep[3 * ES]);
      o = mfma32x4(a4, b4, o);
    }
    if (pin) {
      float *dst = mem_out + ((long long)n * P + p0 + r) * V + vc + wave * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4 *>(dst + 8 * g) =
            make_float4(o[4 * g] / esum, o[4 * g + 1] / esum, o[4 * g + 2] / esum, o[4 * g + 3] / esum);
    }
  }
  // 4. top-l prefix features: wave -> pixels [wave*8, +8)
  for (int pi = 0; pi < 8; ++pi) {
    const int pix = wave * 8 + pi;
    float cum[2];
#pragma unroll
    for (int cls = 0; cls < 2; ++cls) {
      float v[J];
#pragma unroll
      for (int j = 0; j < J; ++j) v[j] = sort64_desc(et[(cls * Lm + lane + 64 * j) * ES + pix], lane);
#pragma unroll
      for (int w = 1; w < J; w <<= 1)
#pragma unroll
        for (int j = 0; j + w < J; j += 2 * w) v[j] = merge_top64(v[j], v[j + w], lane);
      float c = v[0];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        float t = __shfl_up(c, d);
        if (lane >= d) c += t;
      }
      cum[cls] = c;
    }
    if (p0 + pix < P && lane < topl) {
      float f = cum[0] / (cum[0] + cum[1]);
      float *dst = S + ((long long)n * P + p0 + pix) * (2 * topl);
      dst[lane] = f;
      dst[topl + lane] = 1.f - f;
    }
  }
}

struct MatchWs {
  size_t mkn, mvp, total;
};
MatchWs match_ws(int N, int C, int V, int L, int nbanks) {
  MatchWs w;
  const size_t Ltot = (size_t)2 * nbanks * L;
  w.mkn = 0;
  w.mvp = align_up((size_t)N * Ltot * C * 4, 256);
  w.total = w.mvp + align_up((size_t)N * V * Ltot * 4, 256);
  return w;
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" size_t swem_match_workspace(int N, int C, int V, int P, int L, int nbanks) {
  (void)P;
  return match_ws(N, C, V, L, nbanks).total;
}

extern "C" int swem_match_f32(void *stream, const float *qk, const float *kappa_first, const float *nu_first,
                              const float *kappa_update, const float *nu_update, float *mem_out, float *S, int N,
                              int C, int V, int P, int L, int topl, float tau, void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(qk && kappa_first && nu_first && mem_out && S, SWEM_E_ARG, "match: null pointer");
  SWEM_REQUIRE((kappa_update == nullptr) == (nu_update == nullptr), SWEM_E_ARG, "match: update bank half given");
  const int nbanks = kappa_update ? 2 : 1;
  const int Lm = nbanks * L, Ltot = 2 * Lm;
  SWEM_REQUIRE(Lm == 64 || Lm == 128 || Lm == 256 || Lm == 512, SWEM_E_SHAPE,
               "match: bases per class must be 64, 128, 256 or 512 (got %d)", Lm);
  SWEM_REQUIRE(C % 8 == 0 && V % 128 == 0 && L % 4 == 0, SWEM_E_SHAPE, "match: need C %% 8 == 0, V %% 128 == 0");
  SWEM_REQUIRE(topl >= 1 && topl <= 64 && topl <= Lm, SWEM_E_SHAPE, "match: topl must be in [1, 64] (got %d)", topl);
  SWEM_REQUIRE(tau > 0.f, SWEM_E_ARG, "match: tau must be positive");
  const size_t lds = ((size_t)Ltot * 33 + 32 * (C + 4) + 256) * sizeof(float);
  SWEM_REQUIRE(lds <= 160 * 1024, SWEM_E_SHAPE, "match: tile needs %zu bytes of LDS (> 160 KiB); reduce C", lds);
  MatchWs w = match_ws(N, C, V, L, nbanks);
  SWEM_REQUIRE(ws && ws_bytes >= w.total, SWEM_E_WORKSPACE, "match: workspace %zu < %zu", ws_bytes, w.total);
  float *mkn = (float *)((char *)ws + w.mkn), *mvp = (float *)((char *)ws + w.mvp);
  int rc;
  if ((rc = swem_norm_bases_into(stream, kappa_first, mkn, 2 * N, C, L, Lm, 0))) return rc;
  if (nbanks == 2 && (rc = swem_norm_bases_into(stream, kappa_update, mkn, 2 * N, C, L, Lm, L))) return rc;
  {
    long long work = (long long)N * 2 * V * (L / 4);
    hipLaunchKernelGGL(pack_values_kernel, dim3(cdiv(work, 256)), dim3(256), 0, ST, nu_first, mvp, N, V, L, Lm, 0);
    if (nbanks == 2)
      hipLaunchKernelGGL(pack_values_kernel, dim3(cdiv(work, 256)), dim3(256), 0, ST, nu_update, mvp, N, V, L, Lm, L);
    SWEM_CHECK_LAUNCH("pack_values");
  }
  dim3 grid(cdiv(P, 32), N);
#define MATCH(J_)                                                                                                   \
  do {                                                                                                              \
    SWEM_ALLOW_LDS((match_kernel<J_>), 160 * 1024);                                                                 \
    hipLaunchKernelGGL((match_kernel<J_>), grid, dim3(256), lds, ST, qk, mkn, mvp, mem_out, S, C, V, P, topl, tau); \
  } while (0)
  if (Lm == 64) MATCH(1);
  else if (Lm == 128) MATCH(2);
  else if (Lm == 256) MATCH(4);
  else MATCH(8);
#undef MATCH
  SWEM_CHECK_LAUNCH("match_kernel");
  return SWEM_OK;
}
