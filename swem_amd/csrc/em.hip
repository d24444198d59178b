// Sequential weighted EM (reference methods/SWEM/modules.py:93-168) on the gfx950 fp32 matrix cores.
//
// Data layout (device, fp32; NK = 2*N, class minor; Pp = P rounded up to 32):
//   x  [P][C]       raw key, one row per pixel          (reference x_t)
//   xT [C][Pp]      transposed copy, zero padded        (reference x)
//   kn [NK][L][C]   l2-normalised bases, row per base   (l2norm(kappa, dim=-2), modules.py:115)
//   zT [NK][L][Pp]  responsibilities, row per base, pad columns zero  (Pp = P rounded up to 32)
// With the K dimension contiguous in every operand, each lane loads 16 bytes and feeds four
// v_mfma_f32_32x32x2_f32 steps (common.h: mfma32x4).
//
// Kernels per EM iteration (4 small launches):
//   em_ew      : one GEMM  s = x_t . kn  per 32-pixel tile serves BOTH the W step of the previous iteration
//                (cosine = s / (|x|+eps), joint {bg,fg} max, exp-sums, weights = mask * (1 - p)) and the E step
//                (row softmax of s/tau, times weights).  The pixel sits on the MFMA lane, the base index in the
//                accumulator registers, so the row reductions are in-register + one cross-half shuffle + one LDS
//                exchange between the 4 waves (2 classes x 2 halves of L).
//   M GEMM     : xT . z (or vT . z for the value update) for both classes of an object side by side, run by the
//                implicit-GEMM conv kernel as a batched 1x1 "conv" (LDS-staged, split over P, deterministic).
//   em_zsum    : zita = zita_ + sum_p z (one wave per base row).
//   em_finalize: fixed-order slab reduction (deterministic) and the prior blend (zita_*kappa_ + S)/zita; the key-base
//                variant also emits the next iteration's normalised transposed bases (block-local column norms).
#include "../../include/swem_hip_train.h"
#include "common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }

// kn[nk][c/4][l][c%4] = kappa[nk][c][l] / (||kappa[nk][:][l]|| + eps).  Block: 32 bases x all channels.
__global__ __launch_bounds__(256) void em_norm_bases_kernel(const float *__restrict__ kappa, float *__restrict__ kn,
                                                            int C, int L, int out_rows, int out_off) {
  extern __shared__ float sm[];  // tile[C][33], part[8][32], nrm[32]
  float *tile = sm, *part = sm + C * 33, *nrm = part + 256;
  const int nk = blockIdx.y, l0 = blockIdx.x * 32;
  const int l = threadIdx.x & 31, g = threadIdx.x >> 5;
  const float *src = kappa + (long long)nk * C * L + l0 + l;
  float ss = 0.f;
  for (int c = g; c < C; c += 8) {
    float v = (l0 + l < L) ? src[(long long)c * L] : 0.f;
    tile[c * 33 + l] = v;
    ss += v * v;
  }
  part[g * 32 + l] = ss;
  __syncthreads();
  if (threadIdx.x < 32) {
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += part[i * 32 + threadIdx.x];
    nrm[threadIdx.x] = sqrtf(s) + SWEM_L2_EPS;
  }
  __syncthreads();
  // kn is channel-group major, [nk][C/4][row][4]: the E/W and affinity kernels put one base row on every lane, and with
  // row-major [row][C] every lane of a load touched its own cache line (32 lines per instruction, the blocks' GEMMs were
  // bound by that: 15 of em_ew's 19 us); here the 32 rows' 16-byte chunks of one k-step are contiguous.
  for (int idx = threadIdx.x; idx < 32 * C; idx += 256) {
    const int e = idx & 3, ll = (idx >> 2) & 31, c4 = idx >> 7;
    if (l0 + ll < L)
      kn[(((long long)nk * (C / 4) + c4) * out_rows + out_off + l0 + ll) * 4 + e] = tile[(c4 * 4 + e) * 33 + ll] / nrm[ll];
  }
}

// L = 32 * LT * WPC bases per class: WPC waves per class, LT 32-base tiles per wave.  The block is 2*WPC waves; more waves
// per block shorten every wave's MFMA chain and epilogue (there are only P/32 x N blocks: 102 at config B, well under the
// 256 CUs, so the kernel's time is one block's latency).
template <int LT, int WPC>
__global__ __launch_bounds__(128 * WPC) void em_ew_kernel(const float *__restrict__ x, const float *__restrict__ kn,
                                                    const float *__restrict__ masks, const float *__restrict__ w_in,
                                                    float *__restrict__ w_out, float *__restrict__ zT, int C, int P,
                                                    int Pp, int L, float tau, int do_w, int do_e) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int XS = C + 4;
  float *xs = sm;             // [32][C+4]
  constexpr int NW = 2 * WPC;
  float *xn = xs + 32 * XS;   // [32]
  float *red = xn + 32;       // [3][NW][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int cls = wave / WPC, lh = wave % WPC;
  const int n = blockIdx.y, p0 = blockIdx.x * 32;
  const int nk = n * 2 + cls;
  const int cq = C / 4;
  for (int idx = tid; idx < 32 * cq; idx += 64 * NW) {
    int row = idx / cq, c4 = idx - row * cq;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p0 + row < P) v = ld4(x + (long long)(p0 + row) * C + c4 * 4);
    *reinterpret_cast<float4 *>(xs + row * XS + c4 * 4) = v;
  }
  __syncthreads();
  for (int rr = 0; rr < 32 / NW; ++rr) {
    int row = wave * (32 / NW) + rr;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
      float v = xs[row * XS + c];
      s += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) xn[row] = sqrtf(s) + SWEM_L2_EPS;
  }

  f32x16 acc[LT];
#pragma unroll
  for (int t = 0; t < LT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const int lbase = lh * 32 * LT;
  const float *krow = kn + (((long long)nk * (C / 4) + h) * L + lbase + r) * 4;  // chunk (2j + h) of row lbase + r
  const float *xrow = xs + r * XS + 4 * h;
  {
    // The base rows stream from L2 (every lane its own row: 32 cache lines per load instruction) with one wave per SIMD,
    // so nothing but the wave's own prefetch hides the ~1 us round trip: a register ring keeps PF k-steps in flight
    // (with one step ahead the 16-step loop was latency bound: 24 us for 7 us of MFMA work).
    constexpr int PF = 4;
    float4 ring[PF][LT];
    const int steps = C / 8;
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
      for (int t = 0; t < LT; ++t) ring[d][t] = ld4(krow + ((long long)2 * min(d, steps - 1) * L + t * 32) * 4);
    for (int j0 = 0; j0 < steps; j0 += PF) {
#pragma unroll
      for (int d = 0; d < PF; ++d) {
        const int j = j0 + d;
        if (j < steps) {
          float4 b4 = *reinterpret_cast<const float4 *>(xrow + 8 * j);
          float4 a4[LT];
#pragma unroll
          for (int t = 0; t < LT; ++t) a4[t] = ring[d][t];
          const int jn = min(j + PF, steps - 1);
#pragma unroll
          for (int t = 0; t < LT; ++t) ring[d][t] = ld4(krow + ((long long)2 * jn * L + t * 32) * 4);
#pragma unroll
          for (int t = 0; t < LT; ++t) acc[t] = mfma32x4(a4[t], b4, acc[t]);
        }
      }
    }
  }
  __syncthreads();  // xn visible

  const int p = p0 + r;
  const bool pin = p < P;
  const float k2 = SWEM_LOG2E / tau;
  float wgt;
  if (do_w) {
    // W step (modules.py:98-108): cosine, joint max over L and {bg,fg}, exp sums, 1 - p literally
    const float rden = 1.0f / xn[r];  // cosine = s / (|x| + eps): one reciprocal per pixel instead of a division per base
    float m = -__builtin_huge_valf();
#pragma unroll
    for (int t = 0; t < LT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) m = fmaxf(m, acc[t][e] * rden);
    m = fmaxf(m, __shfl_xor(m, 32));
    if (h == 0) red[wave * 32 + r] = m;
    __syncthreads();
    m = red[r];
#pragma unroll
    for (int q = 1; q < NW; ++q) m = fmaxf(m, red[q * 32 + r]);
    float se = 0.f;
#pragma unroll
    for (int t = 0; t < LT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) se += exp_scaled(acc[t][e] * rden - m, k2);
    se += __shfl_xor(se, 32);
    if (h == 0) red[NW * 32 + wave * 32 + r] = se;
    __syncthreads();
    float s_bg = 0.f, s_fg = 0.f;
#pragma unroll
    for (int q = 0; q < WPC; ++q) {
      s_bg += red[NW * 32 + q * 32 + r];
      s_fg += red[NW * 32 + (WPC + q) * 32 + r];
    }
    const float prop = (cls ? s_fg : s_bg) / (s_bg + s_fg);
    const float mk = pin ? masks[(long long)nk * P + p] : 0.f;
    wgt = mk * (1.f - prop);
    if (lh == 0 && h == 0 && pin && w_out) w_out[(long long)nk * P + p] = wgt;
  } else {
    wgt = pin ? w_in[(long long)nk * P + p] : 0.f;
  }
  if (!do_e) return;
  // E step (modules.py:116-119)
  float m = -__builtin_huge_valf();
#pragma unroll
  for (int t = 0; t < LT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) m = fmaxf(m, acc[t][e]);
  m = fmaxf(m, __shfl_xor(m, 32));
  if (h == 0) red[2 * NW * 32 + wave * 32 + r] = m;
  __syncthreads();
  m = red[2 * NW * 32 + cls * WPC * 32 + r];
#pragma unroll
  for (int q = 1; q < WPC; ++q) m = fmaxf(m, red[2 * NW * 32 + (cls * WPC + q) * 32 + r]);
  float se = 0.f;
#pragma unroll
  for (int t = 0; t < LT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = exp_scaled(acc[t][e] - m, k2);
      acc[t][e] = v;
      se += v;
    }
  se += __shfl_xor(se, 32);
  __syncthreads();  // everyone has read the maxima before the slots are reused for the sums
  if (h == 0) red[2 * NW * 32 + wave * 32 + r] = se;
  __syncthreads();
  se = 0.f;
#pragma unroll
  for (int q = 0; q < WPC; ++q) se += red[2 * NW * 32 + (cls * WPC + q) * 32 + r];
  const float zscale = wgt / se;  // softmax normalisation and the pixel weight in one factor
  if (p < Pp) {
    float *dst = zT + ((long long)nk * L + lbase) * Pp + p;
#pragma unroll
    for (int t = 0; t < LT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int l = 32 * t + acc_row(e, h);
        dst[(long long)l * Pp] = pin ? acc[t][e] * zscale : 0.f;
      }
  }
}

// zita[nk][l] = zita_prev[nk][l] + sum_p zT[nk][l][p]: one wave per base row, 16-byte coalesced reads, fixed order
__global__ __launch_bounds__(256) void em_zsum_kernel(const float *__restrict__ zT, const float *__restrict__ zita_prev,
                                                      float *__restrict__ zt, float *__restrict__ zita_out, int rows,
                                                      int Pp) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float *zr = zT + (long long)row * Pp;
  float s = 0.f;
  for (int k = lane * 4; k < Pp; k += 256) {
    float4 v = ld4(zr + k);
    s += (v.x + v.y) + (v.z + v.w);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    float z = zita_prev[row] + s;
    zt[row] = z;
    if (zita_out) zita_out[row] = z;
  }
}

// out[nk][row][l] = (zita_prev[l] * prev[row][l] + S[n][row][cls*L + l]) / zita[l],  nk = 2n + cls
// (S = A . z for both classes of an object, produced by the GEMM with the classes side by side).  Block: 32 bases x 32 rows.
__global__ __launch_bounds__(256) void em_finalize_kernel(const float *__restrict__ S, const float *__restrict__ prev,
                                                          const float *__restrict__ zita_prev,
                                                          const float *__restrict__ zt, float *__restrict__ out, int NK,
                                                          int R, int L) {
  const int nk = blockIdx.y, l = blockIdx.x * 32 + (threadIdx.x & 31);
  const int row0 = blockIdx.z * 32 + (threadIdx.x >> 5);
  const float zp = zita_prev[(long long)nk * L + l], z = zt[(long long)nk * L + l];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + 8 * i;
    if (row >= R) break;
    const long long o = ((long long)nk * R + row) * L + l;
    const float s = S[((long long)(nk >> 1) * R + row) * (2 * L) + (nk & 1) * L + l];
    out[o] = (zp * prev[o] + s) / z;
  }
}

// Same blend for the key bases (R = C rows), fused with the l2-normalised transposed copy the next E/W step reads:
// block = 32 bases x all C rows, so the column norms are block-local.
__global__ __launch_bounds__(1024) void em_finalize_norm_kernel(const float *__restrict__ S,
                                                                const float *__restrict__ prev,
                                                                const float *__restrict__ zita_prev,
                                                                float *__restrict__ zt, float *__restrict__ out,
                                                                float *__restrict__ kn_out, int NK, int R, int L,
                                                                const float *__restrict__ zT, float *__restrict__ zita_out,
                                                                int Pp) {
  extern __shared__ float sm[];  // tile[R][33], red[32][32], nrm[32]
  float *tile = sm, *red = sm + R * 33, *nrm = red + 1024;
  const int nk = blockIdx.y, l0 = blockIdx.x * 32;
  const int l = threadIdx.x & 31, g = threadIdx.x >> 5;  // 32 row groups: enough loads in flight to hide the slab reads
  if (zT) {
    // zita = zita_prev + sum_p z for this block's 32 bases (em_zsum_kernel's order: one wave per row, the same partial
    // sums), fused here to save a launch per EM iteration; published through zt for the value update
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;  // 16 waves, 2 rows each
    for (int rr = w; rr < 32; rr += 16) {
      const float *zr = zT + ((long long)nk * L + l0 + rr) * Pp;
      float s = 0.f;
      for (int k = lane * 4; k < Pp; k += 256) {
        float4 v = ld4(zr + k);
        s += (v.x + v.y) + (v.z + v.w);
      }
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (lane == 0) {
        const float zv = zita_prev[(long long)nk * L + l0 + rr] + s;
        nrm[rr] = zv;
        zt[(long long)nk * L + l0 + rr] = zv;
        if (zita_out) zita_out[(long long)nk * L + l0 + rr] = zv;
      }
    }
    __syncthreads();
  }
  const float zp = zita_prev[(long long)nk * L + l0 + l], z = zT ? nrm[l] : zt[(long long)nk * L + l0 + l];
  __syncthreads();  // nrm is reused for the column norms below
  for (int row = g; row < R; row += 32) {
    const long long o = ((long long)nk * R + row) * L + l0 + l;
    const float s = S[((long long)(nk >> 1) * R + row) * (2 * L) + (nk & 1) * L + l0 + l];
    const float v = (zp * prev[o] + s) / z;
    out[o] = v;
    tile[row * 33 + l] = v;
  }
  __syncthreads();
  if (g < 8) {  // column norms from the tile, in the association of em_norm_bases_kernel (bit-identical kn)
    float ss = 0.f;
    for (int c = g; c < R; c += 8) {
      const float v = tile[c * 33 + l];
      ss += v * v;
    }
    red[g * 32 + l] = ss;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += red[i * 32 + threadIdx.x];
    nrm[threadIdx.x] = sqrtf(s) + SWEM_L2_EPS;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 32 * R; idx += 1024) {  // [nk][R/4][l][4], as em_norm_bases_kernel
    const int e = idx & 3, ll = (idx >> 2) & 31, c4 = idx >> 7;
    kn_out[(((long long)nk * (R / 4) + c4) * L + l0 + ll) * 4 + e] = tile[(c4 * 4 + e) * 33 + ll] / nrm[ll];
  }
}

struct MWs {
  size_t S, conv, zt, total;
};
// workspace of one M step: S [N][R][2L], the GEMM's split-K scratch, zita scratch
MWs mstep_ws(int NK, int R, int P, int L) {
  MWs w;
  const int N = NK / 2, Pp = swem_em_pad(P);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes, 256);
    return at;
  };
  w.S = take((size_t)N * R * 2 * L * sizeof(float));
  w.conv = take(swem_conv2d_workspace(N, R, 1, Pp, 2 * L, 1, 1, 1, 0, 0, 0));
  w.zt = take((size_t)NK * L * sizeof(float));
  w.total = o;
  return w;
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" int swem_em_pad(int P) { return (P + 31) / 32 * 32; }

// kn rows of bank `kappa` land at row out_off + l of an [NK][out_rows][C] image (matching concatenates banks)
int swem_norm_bases_into(void *stream, const float *kappa, float *kn, int NK, int C, int L, int out_rows,
                         int out_off) {
  SWEM_REQUIRE(kappa && kn && NK > 0 && C > 0 && L > 0, SWEM_E_ARG, "em_norm_bases: bad argument");
  SWEM_REQUIRE(C <= 1024, SWEM_E_SHAPE, "em_norm_bases: C > 1024");
  size_t lds = ((size_t)C * 33 + 256 + 32) * sizeof(float);
  hipLaunchKernelGGL(em_norm_bases_kernel, dim3(cdiv(L, 32), NK), dim3(256), lds, ST, kappa, kn, C, L, out_rows,
                     out_off);
  SWEM_CHECK_LAUNCH("em_norm_bases");
  return SWEM_OK;
}

extern "C" int swem_em_norm_bases_f32(void *stream, const float *kappa, float *kn, int NK, int C, int L) {
  return swem_norm_bases_into(stream, kappa, kn, NK, C, L, L, 0);
}

extern "C" int swem_em_ew_f32(void *stream, const float *x, const float *kn, const float *masks, const float *w_in,
                              float *w_out, float *zT, int N, int C, int P, int L, float tau, int do_w, int do_e) {
  SWEM_REQUIRE(x && kn && N > 0 && P > 0, SWEM_E_ARG, "em_ew: bad argument");
  SWEM_REQUIRE(C % 8 == 0 && C <= 512, SWEM_E_SHAPE, "em_ew: C must be a multiple of 8 and <= 512 (got %d)", C);
  SWEM_REQUIRE(L == 64 || L == 128 || L == 256, SWEM_E_SHAPE, "em_ew: L must be 64, 128 or 256 (got %d)", L);
  SWEM_REQUIRE(!do_w || masks, SWEM_E_ARG, "em_ew: W step needs masks");
  SWEM_REQUIRE(do_w || !do_e || w_in, SWEM_E_ARG, "em_ew: E step without W step needs w_in");
  SWEM_REQUIRE(!do_e || zT, SWEM_E_ARG, "em_ew: E step needs zT");
  SWEM_REQUIRE(tau > 0.f, SWEM_E_ARG, "em_ew: tau must be positive");
  const int Pp = swem_em_pad(P);
  dim3 grid(cdiv(P, 32), N);
  // 64 bases per class: 2 waves per class; 128: 4 waves x 1 tile; 256: 4 waves x 2 tiles (8-wave blocks)
#define EW(LT_, WPC_)                                                                                               \
  hipLaunchKernelGGL((em_ew_kernel<LT_, WPC_>), grid, dim3(128 * WPC_),                                              \
                     ((size_t)32 * (C + 4) + 32 + 3 * 2 * WPC_ * 32) * sizeof(float), ST, x, kn, masks, w_in, w_out, zT, \
                     C, P, Pp, L, tau, do_w, do_e)
  if (L == 64) EW(1, 2);
  else if (L == 128) EW(1, 4);
  else EW(2, 4);
#undef EW
  SWEM_CHECK_LAUNCH("em_ew");
  return SWEM_OK;
}

extern "C" size_t swem_em_mstep_workspace(int NK, int R, int P, int L) { return mstep_ws(NK, R, P, L).total; }

namespace {
// zsum_mode: 0 = separate em_zsum launch (the step-level entry point), 1 = fused into the finalize+norm kernel (key bases
// inside memorize), 2 = zt already holds zita (the value update reuses the last key step's)
int mstep_impl(void *stream, const float *A, int a_batch_div, const float *zT, const float *prev, const float *zita_prev,
               float *out, float *zita_out, float *kn_out, int NK, int R, int P, int L, float *S, void *conv_ws,
               size_t conv_bytes, float *zt, int zsum_mode) {
  const int Pp = swem_em_pad(P), N = NK / 2;
  // S[n] = A[n] . [z_bg | z_fg]^T : a batched GEMM on the conv kernel: an R x 1 "image" with Pp channels per object
  // (A shared by all objects when a_batch_div == 0), 2L 1x1 filters per object = its two classes' rows of zT
  int rc = swem_conv2d_nhwc_f32(stream, A, Pp, a_batch_div ? (long long)R * Pp : 0, nullptr, 0, 0, nullptr, 0, 0, N, R, 1,
                                zT, (long long)2 * L * Pp, nullptr, nullptr, nullptr, 0, S, 2 * L, 1, 1, 1, 0, 0, 0, conv_ws,
                                conv_bytes);
  if (rc) return rc;
  if (zsum_mode == 0) {
    hipLaunchKernelGGL(em_zsum_kernel, dim3(cdiv(NK * L, 4)), dim3(256), 0, ST, zT, zita_prev, zt, zita_out, NK * L, Pp);
    SWEM_CHECK_LAUNCH("em_zsum");
  }
  if (kn_out) {
    size_t lds = ((size_t)R * 33 + 1024 + 32) * sizeof(float);
    hipLaunchKernelGGL(em_finalize_norm_kernel, dim3(L / 32, NK), dim3(1024), lds, ST, S, prev, zita_prev, zt, out,
                       kn_out, NK, R, L, zsum_mode == 1 ? zT : nullptr, zita_out, Pp);
  } else {
    hipLaunchKernelGGL(em_finalize_kernel, dim3(L / 32, NK, cdiv(R, 32)), dim3(256), 0, ST, S, prev, zita_prev, zt, out,
                       NK, R, L);
  }
  SWEM_CHECK_LAUNCH("em_finalize");
  return SWEM_OK;
}
}  // namespace

extern "C" int swem_em_mstep_f32(void *stream, const float *A, int a_batch_div, const float *zT, const float *prev,
                                 const float *zita_prev, float *out, float *zita_out, float *kn_out, int NK, int R,
                                 int P, int L, void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(A && zT && prev && zita_prev && out, SWEM_E_ARG, "em_mstep: null pointer");
  SWEM_REQUIRE(NK % 2 == 0 && L % 32 == 0 && R % 128 == 0, SWEM_E_SHAPE,
               "em_mstep: need NK even, L %% 32 == 0 and R %% 128 == 0 (got %d, %d, %d)", NK, L, R);
  SWEM_REQUIRE(a_batch_div == 0 || a_batch_div == 2, SWEM_E_ARG, "em_mstep: a_batch_div must be 0 (shared A) or 2");
  SWEM_REQUIRE(!kn_out || R <= 1024, SWEM_E_SHAPE, "em_mstep: kn_out needs R <= 1024");
  MWs w = mstep_ws(NK, R, P, L);
  SWEM_REQUIRE(ws && ws_bytes >= w.total, SWEM_E_WORKSPACE, "em_mstep: workspace %zu < %zu", ws_bytes, w.total);
  char *base = static_cast<char *>(ws);
  return mstep_impl(stream, A, a_batch_div, zT, prev, zita_prev, out, zita_out, kn_out, NK, R, P, L,
                    reinterpret_cast<float *>(base + w.S), base + w.conv, w.zt - w.conv,
                    reinterpret_cast<float *>(base + w.zt), 0);
}

namespace {
struct MemWs {
  size_t xT, vT, kn, zT, wb, ztb, part, total;
};
MemWs memorize_ws(int N, int C, int V, int P, int L) {
  const int Pp = swem_em_pad(P), NK = 2 * N;
  MemWs w;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes, 256);
    return at;
  };
  w.xT = take((size_t)C * Pp * 4);
  w.vT = take((size_t)N * V * Pp * 4);
  w.kn = take((size_t)NK * L * C * 4);
  w.zT = take((size_t)NK * L * Pp * 4);
  w.wb = take((size_t)NK * P * 4);
  w.ztb = take((size_t)NK * L * 4);
  size_t p1 = swem_em_mstep_workspace(NK, C, P, L), p2 = swem_em_mstep_workspace(NK, V, P, L);
  w.part = take(p1 > p2 ? p1 : p2);
  w.total = o;
  return w;
}
}  // namespace

extern "C" size_t swem_memorize_workspace(int N, int C, int V, int P, int L) {
  return memorize_ws(N, C, V, P, L).total;
}

namespace {
int memorize_impl(void *stream, const float *x, const float *v, const float *masks, const float *kappa_prev,
                  const float *nu_prev, const float *zita_prev, float *kappa_out, float *nu_out, float *zita_out, int N,
                  int C, int V, int P, int L, int T, float tau, void *ws, size_t ws_bytes, float *zT_ext);
}
extern "C" int swem_memorize_f32(void *stream, const float *x, const float *v, const float *masks,
                                 const float *kappa_prev, const float *nu_prev, const float *zita_prev,
                                 float *kappa_out, float *nu_out, float *zita_out, int N, int C, int V, int P, int L,
                                 int T, float tau, void *ws, size_t ws_bytes) {
  return memorize_impl(stream, x, v, masks, kappa_prev, nu_prev, zita_prev, kappa_out, nu_out, zita_out, N, C, V, P, L, T,
                       tau, ws, ws_bytes, nullptr);
}
// training: the same, and the last E step's responsibilities zT [2N][L][Pp] are kept for the value update's backward
extern "C" int swem_memorize_train_f32(void *stream, const float *x, const float *v, const float *masks,
                                       const float *kappa_prev, const float *nu_prev, const float *zita_prev,
                                       float *kappa_out, float *nu_out, float *zita_out, float *zT_out, int N, int C,
                                       int V, int P, int L, int T, float tau, void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(zT_out, SWEM_E_ARG, "memorize_train: zT_out is null");
  return memorize_impl(stream, x, v, masks, kappa_prev, nu_prev, zita_prev, kappa_out, nu_out, zita_out, N, C, V, P, L, T,
                       tau, ws, ws_bytes, zT_out);
}
namespace {
int memorize_impl(void *stream, const float *x, const float *v, const float *masks, const float *kappa_prev,
                  const float *nu_prev, const float *zita_prev, float *kappa_out, float *nu_out, float *zita_out, int N,
                  int C, int V, int P, int L, int T, float tau, void *ws, size_t ws_bytes, float *zT_ext) {
  SWEM_REQUIRE(x && v && masks && kappa_prev && nu_prev && zita_prev && kappa_out && nu_out && zita_out, SWEM_E_ARG,
               "memorize: null pointer");
  SWEM_REQUIRE(T >= 1, SWEM_E_ARG, "memorize: T < 1");
  SWEM_REQUIRE(kappa_out != kappa_prev && nu_out != nu_prev && zita_out != zita_prev, SWEM_E_ARG,
               "memorize: outputs must not alias the prior bases (the prior is read by every iteration)");
  MemWs w = memorize_ws(N, C, V, P, L);
  SWEM_REQUIRE(ws && ws_bytes >= w.total, SWEM_E_WORKSPACE, "memorize: workspace %zu < %zu", ws_bytes, w.total);
  char *base = static_cast<char *>(ws);
  float *xT = (float *)(base + w.xT), *vT = (float *)(base + w.vT), *kn = (float *)(base + w.kn);
  float *zT = zT_ext ? zT_ext : (float *)(base + w.zT), *wb = (float *)(base + w.wb);
  // the two M steps' scratch (S, split-K partials, zita) carved from the shared slot
  MWs wk = mstep_ws(2 * N, C, P, L), wv = mstep_ws(2 * N, V, P, L);
  char *part = base + w.part;
  float *Sk = (float *)(part + wk.S), *Sv = (float *)(part + wv.S);
  void *convk = part + wk.conv, *convv = part + wv.conv;
  const size_t convk_bytes = wk.zt - wk.conv, convv_bytes = wv.zt - wv.conv;
  float *ztb = (float *)(base + w.ztb);
  const int Pp = swem_em_pad(P), NK = 2 * N;
  int rc;
  if ((rc = swem_transpose_f32(stream, x, xT, 1, P, C, Pp))) return rc;
  if ((rc = swem_transpose_f32(stream, v, vT, N, P, V, Pp))) return rc;
  if ((rc = swem_em_norm_bases_f32(stream, kappa_prev, kn, NK, C, L))) return rc;
  for (int it = 0; it < T; ++it) {
    // W step of iteration it-1 (modules.py:161-162) and E step of iteration it share one GEMM
    if ((rc = swem_em_ew_f32(stream, x, kn, masks, masks, wb, zT, N, C, P, L, tau, it > 0, 1))) return rc;
    // key bases: GEMM + one kernel for zita, the prior blend and the next iteration's normalised bases (the last
    // iteration's kn lands in the same scratch and is simply not used)
    if ((rc = mstep_impl(stream, xT, 0, zT, kappa_prev, zita_prev, kappa_out, zita_out, kn, NK, C, P, L, Sk, convk,
                         convk_bytes, ztb, 1)))
      return rc;
  }
  // value bases from the last z (modules.py:164-165); zita is the one just written (ztb)
  return mstep_impl(stream, vT, 2, zT, nu_prev, zita_prev, nu_out, nullptr, nullptr, NK, V, P, L, Sv, convv, convv_bytes,
                    ztb, 2);
}
}  // namespace

// ------------------------------------------------------------------------------------------------ backward (training)
// nu = (zita_prev * nu_prev + v . z) / zita  (modules.py:164-165) is the only part of swem() that carries gradient
// (E/M/W run under no_grad): d v = z . (dnu / zita),  d nu_prev = dnu * zita_prev / zita.
namespace {
// Gp[n][v][cls*L + l] = dnu[nk][v][l] / zita[nk][l];  dnu_prev[nk][v][l] = Gp * zita_prev[nk][l]
__global__ void nu_bwd_prep_kernel(const float *__restrict__ dnu, const float *__restrict__ zita,
                                   const float *__restrict__ zita_prev, float *__restrict__ Gp,
                                   float *__restrict__ dnu_prev, int N, int V, int L) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * 2 * V * L) return;
  const int l = (int)(i % L);
  long long t = i / L;
  const int v = (int)(t % V);
  t /= V;  // nk
  const int cls = (int)(t & 1), n = (int)(t >> 1);
  const float g = dnu[i] / zita[t * L + l];
  Gp[((long long)n * V + v) * (2 * L) + cls * L + l] = g;
  if (dnu_prev) dnu_prev[i] = g * zita_prev[t * L + l];
}
struct NuBwdWs {
  size_t Gp, zTt, dvp, conv, total;
};
NuBwdWs nu_bwd_ws(int N, int V, int P, int L) {
  NuBwdWs w;
  const int Pm = swem_match_pad(P);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes, 256);
    return at;
  };
  w.Gp = take((size_t)N * V * 2 * L * 4);
  w.zTt = take((size_t)N * Pm * 2 * L * 4);
  w.dvp = take((size_t)N * Pm * V * 4);
  w.conv = take(swem_conv2d_workspace(N, Pm, 1, 2 * L, V, 1, 1, 1, 0, 0, 0));
  w.total = o;
  return w;
}
}  // namespace

extern "C" size_t swem_nu_update_bwd_workspace(int N, int V, int P, int L) { return nu_bwd_ws(N, V, P, L).total; }

extern "C" int swem_nu_update_bwd_f32(void *stream, const float *zT, const float *zita_prev, const float *zita,
                                      const float *dnu, float *dv, float *dnu_prev, int N, int V, int P, int L, void *ws,
                                      size_t ws_bytes) {
  SWEM_REQUIRE(zT && zita_prev && zita && dnu && dv, SWEM_E_ARG, "nu_update_bwd: null pointer");
  SWEM_REQUIRE(N > 0 && V % 4 == 0 && L % 32 == 0, SWEM_E_SHAPE, "nu_update_bwd: need V %% 4 == 0 and L %% 32 == 0");
  NuBwdWs w = nu_bwd_ws(N, V, P, L);
  SWEM_REQUIRE(ws && ws_bytes >= w.total, SWEM_E_WORKSPACE, "nu_update_bwd: workspace %zu < %zu", ws_bytes, w.total);
  char *base = static_cast<char *>(ws);
  float *Gp = (float *)(base + w.Gp), *zTt = (float *)(base + w.zTt), *dvp = (float *)(base + w.dvp);
  const int Pp = swem_em_pad(P), Pm = swem_match_pad(P);
  hipLaunchKernelGGL(nu_bwd_prep_kernel, dim3(cdiv((long long)N * 2 * V * L, 256)), dim3(256), 0, ST, dnu, zita,
                     zita_prev, Gp, dnu_prev, N, V, L);
  SWEM_CHECK_LAUNCH("nu_bwd_prep");
  // z pixel-major per object: zT[n] is [2L][Pp] -> [Pp][2L], rows padded with zeros to the GEMM tile (Pm)
  if (hipMemsetAsync(zTt, 0, (size_t)N * Pm * 2 * L * 4, ST) != hipSuccess) {
    swem_set_error("nu_update_bwd: memset failed");
    return SWEM_E_HIP;
  }
  int rc;
  for (int n = 0; n < N; ++n)
    if ((rc = swem_transpose_f32(stream, zT + (long long)n * 2 * L * Pp, zTt + (long long)n * Pm * 2 * L, 1, 2 * L, Pp,
                                 2 * L)))
      return rc;
  // dv[n] = z[n] . Gp[n]^T : batched GEMM on the conv kernel (a Pm x 1 image with 2L channels, V filters per object)
  if ((rc = swem_conv2d_nhwc_f32(stream, zTt, 2 * L, (long long)Pm * 2 * L, nullptr, 0, 0, nullptr, 0, 0, N, Pm, 1, Gp,
                                 (long long)V * 2 * L, nullptr, nullptr, nullptr, 0, dvp, V, 1, 1, 1, 0, 0, 0,
                                 base + w.conv, w.total - w.conv)))
    return rc;
  if (hipMemcpy2DAsync(dv, (size_t)P * V * 4, dvp, (size_t)Pm * V * 4, (size_t)P * V * 4, N, hipMemcpyDeviceToDevice,
                       ST) != hipSuccess) {
    swem_set_error("nu_update_bwd: copy failed");
    return SWEM_E_HIP;
  }
  return SWEM_OK;
}
