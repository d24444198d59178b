// Sequential weighted EM (reference methods/SWEM/modules.py:93-168) on the gfx950 fp32 matrix cores.
//
// Data layout (device, fp32; NK = 2*N, class minor).  Every operand is read as it lies in memory: no transposed copies.
//   x  [P][C]            raw key, one row per pixel          (reference x_t; NHWC feature map as the encoder wrote it)
//   v  [N][P][V]         value map per object, pixel-major   (NHWC)
//   kn [NK][C/4][R][4]   l2-normalised bases, channel-group major; a bank occupies rows [off, off+L) of R >= L rows
//                        (R = L inside memorize; R = 2L, off = L when it is the 'update' half of matching's packed banks)
//   z  [N][Pz][2L]       responsibilities, one row per PIXEL: z[n][p][cls*L + l]; Pz = swem_em_pad(P) rows allocated,
//                        rows [0, ceil16(P)) written (pad pixels as zeros)
//
// Three launches per EM iteration (a dependent kernel boundary costs ~1.5 us on this chip, a grid-wide barrier inside a
// persistent kernel 4-7 us and a split-K seam with a last-arriver combine 5-13 us: MI355X_MICROARCH.md, price list --
// so the iteration is three well-filled kernels, not one cooperative one):
//   em_ew16    : block = (object, 16-pixel tile), 8 waves = 2 classes x 4 base quarters.  One GEMM  s = kn . x_t  on
//                v_mfma_f32_16x16x4_f32 serves BOTH the W step of the previous iteration (cosine = s / (|x|+eps), joint
//                {bg,fg} max, exp-sums, weights = mask * (1 - p)) and the E step (softmax of s/tau over the class's bases,
//                times weights).  The pixel sits on the MFMA column lane, four bases of a tile in the accumulator
//                registers: row reductions are in-register + two shuffles + ONE LDS exchange for the maxima and one for the
//                sums.  The pixel's key (all C channels) lives in registers; every base row is loaded exactly once per
//                block, all loads issued before the first MFMA.  z leaves as 16-byte stores, 64 contiguous bytes per pixel
//                and tile.  204 blocks at config B (P = 1620, N = 2) instead of the 102 a 32-pixel tile gives.
//   em_mstep   : S = X^T . z over a CHUNK of pixels (split-P): block = (object-class, 32 bases) x (128 rows of X) x chunk,
//                8 waves = 4 row tiles x 2 halves of the chunk, v_mfma_f32_32x32x2_f32.  X = x (key bases) or v[n]
//                (value bases, last iteration only: both in ONE launch).  Operands are fetched by raw buffer loads
//                (pixels >= P read as zeros by the buffer's range check on the vector offset: no masks), issued ahead of
//                the MFMA chain.  Partial sums go to Spart[chunk][nk][row][L]; the column
//                sums of z (zita's increment) ride along.
//   em_finalize: fixed-order sum over the chunks (deterministic), prior blend (zita_*prev + S)/zita, and for the key
//                bases the l2-normalised kn the next E/W step reads (block = 32 bases x all C rows: norms are block-local).
//                On the last iteration it can also write the bases straight into matching's packed banks.
#include "../../include/swem_hip_train.h"
#include "common.h"

// Debug build only (-DSWEM_EM_STAMPS): in-kernel clock stamps of block 0 / wave 0 (common.h), written to a buffer set by
// swem_debug_set_stamps (tools/em_stamps.py, tools/conv_stamps.py).  The product build contains none of it.
#ifdef SWEM_EM_STAMPS
long long *g_swem_stamps = nullptr;
int g_swem_stamp_slot = 0;
extern "C" void swem_debug_set_stamps(void *p) {
  g_swem_stamps = static_cast<long long *>(p);
  g_swem_stamp_slot = 0;
}
#endif

namespace {

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// kn[nk][c/4][off + l][c%4] = kappa[nk][c][l] / (||kappa[nk][:][l]|| + eps).  Block: 32 bases x all channels.
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
  // channel-group major: the E/W and affinity kernels put one base row on every lane, so the 16 (32) rows' 16-byte chunks
  // of one k-step are contiguous
  for (int idx = threadIdx.x; idx < 32 * C; idx += 256) {
    const int e = idx & 3, ll = (idx >> 2) & 31, c4 = idx >> 7;
    if (l0 + ll < L)
      kn[(((long long)nk * (C / 4) + c4) * out_rows + out_off + l0 + ll) * 4 + e] = tile[(c4 * 4 + e) * 33 + ll] / nrm[ll];
  }
}

// ---------------------------------------------------------------------------------------------------- E / W step
// L = 64 * LT bases per class (LT 16-base MFMA tiles per wave), C = 16 * CM channels.
// v_mfma_f32_16x16x4_f32: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; D[i][j] lives in lane
// (j + 16 g), register r, with i = 4 g + r.  A = base rows, B = pixels: the pixel is on the lane, 4 bases in the registers.
// One 16-byte load feeds four k-steps: lane group g takes channels 16 m + 4 g + e (e = 0..3) of chunk m -- a permutation
// of the channel order that A and B share, so the sum runs over the same products.
template <int LT, int CM>
__global__ __launch_bounds__(512) void em_ew16_kernel(const float *__restrict__ x, const float *__restrict__ kn,
                                                      int kn_rows, int kn_off, const float *__restrict__ masks,
                                                      const float *__restrict__ w_in, float *__restrict__ w_out,
                                                      float *__restrict__ z, int P, int Pz, float tau, int do_w,
                                                      int do_e STAMP_ARG) {
  constexpr int C = 16 * CM, L = 64 * LT;
  STAMP(0);
  __shared__ float red[3][8][16];  // per-wave maxima / W exp-sums / E exp-sums per pixel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int cls = wave >> 2, q = wave & 3;
  const int n = blockIdx.y, p = blockIdx.x * 16 + li;
  const int nk = 2 * n + cls;
  const bool pin = p < P;
  // base rows of this wave: [q*16*LT, (q+1)*16*LT) of class cls; lane (li, g) loads row li of every tile, chunk 4m + g
  const float *kb = kn + (((long long)nk * (C / 4) + g) * kn_rows + kn_off + q * 16 * LT + li) * 4;
  // issue order = arrival order (vmcnt counts in order): the pixel's key first, then the base rows chunk by chunk, so the
  // MFMAs of chunk m wait for chunk m only and the rest of the 256 KB streams in behind them
  // (buffer loads: pad pixels p >= P read as zeros by the range check -- no select in front of the MFMAs)
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, P * C * 4, 0x00020000);
  float4 xf[CM];
#pragma unroll
  for (int m = 0; m < CM; ++m) {
    u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)((p * C + 16 * m + 4 * g) * 4), 0, 0);
    xf[m] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
  }
  // the pixel's mask (W step) or incoming weight, needed only in the epilogue: fetched now, behind nothing
  __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>((do_w ? masks : w_in) + (long long)nk * P), 0, P * 4, 0x00020000);
  const float mk = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rm, (unsigned)(p * 4), 0, 0));
  float4 a[CM][LT];
#pragma unroll
  for (int m = 0; m < CM; ++m)
#pragma unroll
    for (int t = 0; t < LT; ++t) a[m][t] = ld4(kb + ((long long)4 * m * kn_rows + 16 * t) * 4);
  // keep every load above the MFMA chain: left alone, the scheduler sinks each load to just before its use to save
  // registers and the wave then pays one L2 round trip per chunk (seen in the ISA: vmcnt(1) in front of every MFMA group)
  __builtin_amdgcn_sched_barrier(0);
  STAMP(1);
  float ss = 0.f;
#pragma unroll
  for (int m = 0; m < CM; ++m) ss += (xf[m].x * xf[m].x + xf[m].y * xf[m].y) + (xf[m].z * xf[m].z + xf[m].w * xf[m].w);
  ss += __shfl_xor(ss, 16);
  ss += __shfl_xor(ss, 32);
  const float xnorm = sqrtf(ss) + SWEM_L2_EPS;

  f32x4 acc[LT];
#pragma unroll
  for (int t = 0; t < LT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < CM; ++m) {
#pragma unroll
    for (int t = 0; t < LT; ++t) acc[t] = mfma16(a[m][t].x, xf[m].x, acc[t]);
#pragma unroll
    for (int t = 0; t < LT; ++t) acc[t] = mfma16(a[m][t].y, xf[m].y, acc[t]);
#pragma unroll
    for (int t = 0; t < LT; ++t) acc[t] = mfma16(a[m][t].z, xf[m].z, acc[t]);
#pragma unroll
    for (int t = 0; t < LT; ++t) acc[t] = mfma16(a[m][t].w, xf[m].w, acc[t]);
  }

  // maxima of the raw logits over this wave's bases; the W step's cosine is s / (|x| + eps) with a positive per-pixel
  // factor, so its joint maximum is that factor times the larger class maximum (rounding is monotonic): one exchange
  float ml = -__builtin_huge_valf();
#pragma unroll
  for (int t = 0; t < LT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) ml = fmaxf(ml, acc[t][e]);
  ml = fmaxf(ml, __shfl_xor(ml, 16));
  ml = fmaxf(ml, __shfl_xor(ml, 32));
  if (g == 0) red[0][wave][li] = ml;
  STAMP(2);
  __syncthreads();
  STAMP(3);
  const float m_bg = fmaxf(fmaxf(red[0][0][li], red[0][1][li]), fmaxf(red[0][2][li], red[0][3][li]));
  const float m_fg = fmaxf(fmaxf(red[0][4][li], red[0][5][li]), fmaxf(red[0][6][li], red[0][7][li]));
  const float k2 = SWEM_LOG2E / tau;
  const float rden = 1.0f / xnorm;  // one reciprocal per pixel instead of a division per base
  if (do_w) {
    // W step (modules.py:98-108): exp-sums of the cosines against the joint maximum
    const float mw = fmaxf(m_bg, m_fg) * rden;
    float sw = 0.f;
#pragma unroll
    for (int t = 0; t < LT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) sw += exp_scaled(acc[t][e] * rden - mw, k2);
    sw += __shfl_xor(sw, 16);
    sw += __shfl_xor(sw, 32);
    if (g == 0) red[1][wave][li] = sw;
  }
  if (do_e) {
    // E step (modules.py:116-119): exp against the class's own row maximum
    const float me = cls ? m_fg : m_bg;
    float se = 0.f;
#pragma unroll
    for (int t = 0; t < LT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = exp_scaled(acc[t][e] - me, k2);
        acc[t][e] = v;
        se += v;
      }
    se += __shfl_xor(se, 16);
    se += __shfl_xor(se, 32);
    if (g == 0) red[2][wave][li] = se;
  }
  STAMP(4);
  __syncthreads();
  STAMP(5);
  float wgt;
  if (do_w) {
    const float s_bg = (red[1][0][li] + red[1][1][li]) + (red[1][2][li] + red[1][3][li]);
    const float s_fg = (red[1][4][li] + red[1][5][li]) + (red[1][6][li] + red[1][7][li]);
    const float prop = (cls ? s_fg : s_bg) / (s_bg + s_fg);
    wgt = mk * (1.f - prop);  // 1 - p literally, as the reference
    if (w_out && q == 0 && g == 0 && pin) w_out[(long long)nk * P + p] = wgt;
  } else {
    wgt = mk;
  }
  if (!do_e) return;
  const float se = (red[2][4 * cls][li] + red[2][4 * cls + 1][li]) + (red[2][4 * cls + 2][li] + red[2][4 * cls + 3][li]);
  const float zscale = pin ? wgt / se : 0.f;  // softmax normalisation and the pixel weight in one factor
  float *dst = z + ((long long)n * Pz + p) * (2 * L) + cls * L + q * 16 * LT + 4 * g;
#pragma unroll
  for (int t = 0; t < LT; ++t)
    *reinterpret_cast<float4 *>(dst + 16 * t) =
        make_float4(acc[t][0] * zscale, acc[t][1] * zscale, acc[t][2] * zscale, acc[t][3] * zscale);
  STAMP(6);
#ifdef SWEM_EM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(7);
#endif
}

// ---------------------------------------------------------------------------------------------------- M step (split-P)
struct MStepP {
  const float *x;  // [P][C] key rows (row space [0, Ck)), or NULL when Ck == 0
  const float *v;  // [N][P][V] value rows (row space [Ck, Ck + Vv)), or NULL
  const float *z;  // [N][Pz][2L]
  float *Spart;    // [nch][NK][Rtot][L]
  float *zpart;    // [nch][NK][L]
  int Ck, C, V, P, Pz, L, NK, Rtot;
};

// Block = (nk, 32 bases) x (128 rows of the row space) x chunk of 4*STEPS pixels; wave = (row tile ct, chunk half kh).
template <int STEPS>
__global__ __launch_bounds__(512) void em_mstep_kernel(MStepP p STAMP_ARG) {
  STAMP(0);
  __shared__ float red[4][16][64];
  __shared__ float zred[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, kk = lane >> 5;
  const int ct = wave & 3, kh = wave >> 2;
  const int tiles = p.L / 32;
  const int nk = blockIdx.x / tiles, l0 = (blockIdx.x - nk * tiles) * 32;
  const int n = nk >> 1, cls = nk & 1;
  const int row0 = blockIdx.y * 128;
  const int chunk = blockIdx.z;
  const int pw0 = chunk * 4 * STEPS + kh * 2 * STEPS;  // first pixel of this wave's half chunk
  // wave-uniform descriptors: the row operand (x or v[n]) and z[n]; offsets beyond P rows read as zeros
  const bool key = row0 < p.Ck;
  const float *src = key ? p.x : p.v + (long long)n * p.P * p.V;
  const int stride = key ? p.C : p.V;
  const int col = (key ? row0 : row0 - p.Ck) + 32 * ct;
  __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, (int)((long long)p.P * stride * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.z + (long long)n * p.Pz * 2 * p.L), 0, (int)((long long)p.P * 2 * p.L * 4), 0x00020000);
  // the pixel offset travels in the VECTOR offset: the hardware range-checks voffset + immediate against num_records (the
  // scalar offset is added after the check), so pixels >= P come back as zeros without touching memory
  const unsigned va = (unsigned)(((pw0 + kk) * stride + col + i) * 4);
  const unsigned vb = (unsigned)(((pw0 + kk) * 2 * p.L + cls * p.L + l0 + i) * 4);
  const unsigned sa = (unsigned)stride * 8u, sb = (unsigned)p.L * 16u;  // bytes per step (two pixels)
  float av[STEPS], bv[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    av[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, va + s * sa, 0, 0));
    bv[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, vb + s * sb, 0, 0));
  }
  __builtin_amdgcn_sched_barrier(0);  // all loads in flight before the MFMA chain (see em_ew16_kernel)
  STAMP(1);
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) acc = mfma32(av[s], bv[s], acc);
  // the column sums of z after the MFMA chain (in front of it the adds wait for the LAST load and the chain with them)
  // (the empty asm makes the sum's start depend on the accumulator: instruction selection otherwise emits the adds first)
  float zs = 0.f;
  asm volatile("" : "+v"(zs) : "v"(acc[0]));
#pragma unroll
  for (int s = 0; s < STEPS; ++s) zs += bv[s];
  zs += __shfl_xor(zs, 32);
  if (kh == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[ct][e][lane] = acc[e];
  }
  if (ct == 0 && lane < 32) zred[kh][lane] = zs;
  STAMP(2);
  __syncthreads();
  STAMP(3);
  if (kh == 1) return;
  float *dst = p.Spart + (((long long)chunk * p.NK + nk) * p.Rtot + row0 + 32 * ct) * p.L + l0 + i;
#pragma unroll
  for (int e = 0; e < 16; ++e) dst[(long long)acc_row(e, kk) * p.L] = acc[e] + red[ct][e][lane];
  if (ct == 0 && lane < 32 && blockIdx.y == 0)
    p.zpart[((long long)chunk * p.NK + nk) * p.L + l0 + lane] = zred[0][lane] + zred[1][lane];
  STAMP(4);
#ifdef SWEM_EM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(5);
#endif
}

// ---------------------------------------------------------------------------------------------------- finalize
struct FinP {
  const float *Spart, *zpart;
  const float *kappa_prev, *nu_prev, *zita_prev;
  const float *zita_in;  // when set, zita is read from here instead of zita_prev + sum of zpart
  float *kappa_out, *nu_out, *zita_out;
  float *kn_out;          // optional: normalised key bases [NK][C/4][kn_rows][4] at row offset kn_off
  float *mvp_out;         // optional: value bases packed for matching, mvp[n][v][cls*mvp_lm + mvp_off + l]
  int kn_rows, kn_off, mvp_lm, mvp_off;
  int Ck, C, V, L, NK, Rtot, nch;
};

// Block = (32 bases, nk, 128-row group of the row space); 1024 threads = 32 bases x 32 row lanes.
constexpr int FIN_CH = 8;  // chunk partials fetched together (config B: all 8)
__global__ __launch_bounds__(1024) void em_finalize_kernel(FinP p STAMP_ARG) {
  STAMP(0);
  __shared__ float tile[128 * 33];
  __shared__ float part[8][32];
  __shared__ float nrm[32];
  const int nk = blockIdx.y, l0 = blockIdx.x * 32;
  const int l = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int row0 = blockIdx.z * 128;
  const bool key = row0 < p.Ck;
  const float zp = p.zita_prev[(long long)nk * p.L + l0 + l];
  const int R = key ? p.C : p.V;
  const int rbase = key ? row0 : row0 - p.Ck;
  const float *prev = key ? p.kappa_prev : p.nu_prev;
  float *out = key ? p.kappa_out : p.nu_out;
  const int n = nk >> 1, cls = nk & 1;
  float pv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    pv[q] = rbase + g + 32 * q < R ? prev[((long long)nk * R + rbase + g + 32 * q) * p.L + l0 + l] : 0.f;
  // ONE round trip for everything this thread reads: the prior above, and per chunk the column-sum partial of its base
  // (every thread sums zita itself -- the same addresses in all 32 row lanes, cache hits -- instead of a first phase behind
  // a barrier) and the S partials of its four rows.  Every load of a chunk group is in flight before the first add (a
  // dependent chain of L2 / Infinity-Cache round trips was most of this kernel's time); sums run in chunk order.
  const long long cs = (long long)p.NK * p.Rtot * p.L;
  float sums[4] = {0.f, 0.f, 0.f, 0.f}, zsum = 0.f;
  for (int c0 = 0; c0 < p.nch; c0 += FIN_CH) {
    float t[4][FIN_CH], tz[FIN_CH];
#pragma unroll
    for (int ch = 0; ch < FIN_CH; ++ch)
      tz[ch] = (c0 + ch < p.nch && !p.zita_in) ? p.zpart[((long long)(c0 + ch) * p.NK + nk) * p.L + l0 + l] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float *sp = p.Spart + ((long long)nk * p.Rtot + row0 + g + 32 * q) * p.L + l0 + l;
#pragma unroll
      for (int ch = 0; ch < FIN_CH; ++ch) t[q][ch] = (c0 + ch < p.nch && rbase + g + 32 * q < R) ? sp[(c0 + ch) * cs] : 0.f;
    }
#pragma unroll
    for (int ch = 0; ch < FIN_CH; ++ch) zsum += tz[ch];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int ch = 0; ch < FIN_CH; ++ch) sums[q] += t[q][ch];
  }
  STAMP(1);
  const float zt = p.zita_in ? p.zita_in[(long long)nk * p.L + l0 + l] : zp + zsum;
  if (p.zita_out && blockIdx.z == 0 && g == 0) p.zita_out[(long long)nk * p.L + l0 + l] = zt;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rr = g + 32 * q;
    const int row = rbase + rr;
    float v = 0.f;
    if (row < R) {
      const long long o = ((long long)nk * R + row) * p.L + l0 + l;
      v = (zp * pv[q] + sums[q]) / zt;
      out[o] = v;
      if (!key && p.mvp_out)
        p.mvp_out[((long long)n * p.V + row) * (2 * p.mvp_lm) + cls * p.mvp_lm + p.mvp_off + l0 + l] = v;
    }
    if (key) tile[rr * 33 + l] = v;
  }
  STAMP(2);
  if (!key || !p.kn_out) return;
  __syncthreads();
  STAMP(3);
  if (g < 8) {  // column norms from the tile, in the association of em_norm_bases_kernel (bit-identical kn)
    float ss = 0.f;
    for (int c = g; c < p.C; c += 8) {
      const float v = tile[c * 33 + l];
      ss += v * v;
    }
    part[g][l] = ss;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += part[i][threadIdx.x];
    nrm[threadIdx.x] = sqrtf(s) + SWEM_L2_EPS;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 32 * p.C; idx += 1024) {
    const int e = idx & 3, ll = (idx >> 2) & 31, c4 = idx >> 7;
    p.kn_out[(((long long)nk * (p.C / 4) + c4) * p.kn_rows + p.kn_off + l0 + ll) * 4 + e] = tile[(c4 * 4 + e) * 33 + ll] / nrm[ll];
  }
  STAMP(4);
#ifdef SWEM_EM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(5);
#endif
}

struct MWs {
  size_t Spart, zpart, total;
  int nch, steps;
};
// workspace of one M step over a row space of Rtot rows: chunk partials of S and of the column sums of z
MWs mstep_ws(int NK, int Rtot, int P, int L) {
  MWs w;
  // 208-pixel chunks (52 two-pixel steps per half-chunk wave) unless that leaves most of the chip idle.  The choice
  // depends on the key rows' grid only, so that every M step of a memorize (keys alone, keys + values) and the step-level
  // entry point cut P the same way: the same partial sums in the same order whatever the row space
  w.steps = 52;
  long long blocks = (long long)NK * (L / 32) * cdiv(P, 4 * w.steps);
  if (blocks < 160) w.steps = 26;
  w.nch = cdiv(P, 4 * w.steps);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes, 256);
    return at;
  };
  w.Spart = take((size_t)w.nch * NK * Rtot * L * sizeof(float));
  w.zpart = take((size_t)w.nch * NK * L * sizeof(float));
  w.total = o;
  return w;
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" int swem_em_pad(int P) { return (P + 127) / 128 * 128; }

// kn rows of bank `kappa` land at row out_off + l of an [NK][C/4][out_rows][4] image (matching concatenates banks)
int swem_norm_bases_into(void *stream, const float *kappa, float *kn, int NK, int C, int L, int out_rows,
                         int out_off) {
  SWEM_REQUIRE(kappa && kn && NK > 0 && C > 0 && L > 0, SWEM_E_ARG, "em_norm_bases: bad argument");
  SWEM_REQUIRE(C <= 1024 && C % 4 == 0, SWEM_E_SHAPE, "em_norm_bases: C must be a multiple of 4, at most 1024");
  size_t lds = ((size_t)C * 33 + 256 + 32) * sizeof(float);
  hipLaunchKernelGGL(em_norm_bases_kernel, dim3(cdiv(L, 32), NK), dim3(256), lds, ST, kappa, kn, C, L, out_rows,
                     out_off);
  SWEM_CHECK_LAUNCH("em_norm_bases");
  return SWEM_OK;
}

extern "C" int swem_em_norm_bases_f32(void *stream, const float *kappa, float *kn, int NK, int C, int L) {
  return swem_norm_bases_into(stream, kappa, kn, NK, C, L, L, 0);
}

namespace {
int ew_launch(void *stream, const float *x, const float *kn, int kn_rows, int kn_off, const float *masks,
              const float *w_in, float *w_out, float *z, int N, int C, int P, int L, float tau, int do_w, int do_e) {
  const int Pz = swem_em_pad(P);
  dim3 grid(cdiv(P, 16), N);
#define EW(LT_, CM_)                                                                                                  \
  hipLaunchKernelGGL((em_ew16_kernel<LT_, CM_>), grid, dim3(512), 0, ST, x, kn, kn_rows, kn_off, masks, w_in, w_out, z, \
                     P, Pz, tau, do_w, do_e STAMP_PASS)
  if (C == 128) {
    if (L == 64) EW(1, 8);
    else if (L == 128) EW(2, 8);
    else EW(4, 8);
  } else {
    if (L == 64) EW(1, 4);
    else if (L == 128) EW(2, 4);
    else EW(4, 4);
  }
#undef EW
  SWEM_CHECK_LAUNCH("em_ew");
  return SWEM_OK;
}
}  // namespace

extern "C" int swem_em_ew_f32(void *stream, const float *x, const float *kn, const float *masks, const float *w_in,
                              float *w_out, float *z, int N, int C, int P, int L, float tau, int do_w, int do_e) {
  SWEM_REQUIRE(x && kn && N > 0 && P > 0, SWEM_E_ARG, "em_ew: bad argument");
  SWEM_REQUIRE(C == 64 || C == 128, SWEM_E_SHAPE, "em_ew: the key dimension must be 64 or 128 (got %d)", C);
  SWEM_REQUIRE(L == 64 || L == 128 || L == 256, SWEM_E_SHAPE, "em_ew: L must be 64, 128 or 256 (got %d)", L);
  SWEM_REQUIRE(!do_w || masks, SWEM_E_ARG, "em_ew: W step needs masks");
  SWEM_REQUIRE(do_w || !do_e || w_in, SWEM_E_ARG, "em_ew: E step without W step needs w_in");
  SWEM_REQUIRE(!do_e || z, SWEM_E_ARG, "em_ew: E step needs z");
  SWEM_REQUIRE(tau > 0.f, SWEM_E_ARG, "em_ew: tau must be positive");
  return ew_launch(stream, x, kn, L, 0, masks, w_in, w_out, z, N, C, P, L, tau, do_w, do_e);
}

namespace {
// One M step over the row space [keys (Ck rows of x) | values (Vv rows of v)] and the finalize of both.
int mstep_impl(void *stream, const float *x, const float *v, const float *z, const float *kappa_prev,
               const float *nu_prev, const float *zita_prev, float *kappa_out, float *nu_out, float *zita_out,
               float *kn_out, int kn_rows, int kn_off, float *mvp_out, int mvp_lm, int mvp_off, int NK, int Ck, int Vv,
               int C, int V, int P, int L, char *ws) {
  const int Rtot = Ck + Vv;
  MWs w = mstep_ws(NK, Rtot, P, L);
  MStepP mp;
  mp.x = x;
  mp.v = v;
  mp.z = z;
  mp.Spart = reinterpret_cast<float *>(ws + w.Spart);
  mp.zpart = reinterpret_cast<float *>(ws + w.zpart);
  mp.Ck = Ck, mp.C = C, mp.V = V, mp.P = P, mp.Pz = swem_em_pad(P), mp.L = L, mp.NK = NK, mp.Rtot = Rtot;
  dim3 grid(NK * (L / 32), Rtot / 128, w.nch);
  if (w.steps == 52) hipLaunchKernelGGL((em_mstep_kernel<52>), grid, dim3(512), 0, ST, mp STAMP_PASS);
  else hipLaunchKernelGGL((em_mstep_kernel<26>), grid, dim3(512), 0, ST, mp STAMP_PASS);
  SWEM_CHECK_LAUNCH("em_mstep");
  FinP fp;
  fp.Spart = mp.Spart, fp.zpart = mp.zpart;
  fp.kappa_prev = kappa_prev, fp.nu_prev = nu_prev, fp.zita_prev = zita_prev, fp.zita_in = nullptr;
  fp.kappa_out = kappa_out, fp.nu_out = nu_out, fp.zita_out = zita_out;
  fp.kn_out = kn_out, fp.mvp_out = mvp_out;
  fp.kn_rows = kn_rows, fp.kn_off = kn_off, fp.mvp_lm = mvp_lm, fp.mvp_off = mvp_off;
  fp.Ck = Ck, fp.C = C, fp.V = V, fp.L = L, fp.NK = NK, fp.Rtot = Rtot, fp.nch = w.nch;
  hipLaunchKernelGGL(em_finalize_kernel, dim3(L / 32, NK, Rtot / 128), dim3(1024), 0, ST, fp STAMP_PASS);
  SWEM_CHECK_LAUNCH("em_finalize");
  return SWEM_OK;
}
}  // namespace

extern "C" size_t swem_em_mstep_workspace(int NK, int R, int P, int L) { return mstep_ws(NK, R, P, L).total; }

extern "C" int swem_em_mstep_f32(void *stream, const float *A, int a_per_object, const float *z, const float *prev,
                                 const float *zita_prev, float *out, float *zita_out, float *kn_out, int NK, int R,
                                 int P, int L, void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(A && z && prev && zita_prev && out, SWEM_E_ARG, "em_mstep: null pointer");
  SWEM_REQUIRE(NK % 2 == 0 && L % 32 == 0 && R % 128 == 0, SWEM_E_SHAPE,
               "em_mstep: need NK even, L %% 32 == 0 and R %% 128 == 0 (got %d, %d, %d)", NK, L, R);
  SWEM_REQUIRE(!kn_out || (R == 128 && !a_per_object), SWEM_E_SHAPE, "em_mstep: kn_out needs the key rows (R == 128, shared A)");
  MWs w = mstep_ws(NK, R, P, L);
  SWEM_REQUIRE(ws && ws_bytes >= w.total, SWEM_E_WORKSPACE, "em_mstep: workspace %zu < %zu", ws_bytes, w.total);
  if (a_per_object)  // value rows: A = v [N][P][R]
    return mstep_impl(stream, nullptr, A, z, nullptr, prev, zita_prev, nullptr, out, zita_out, nullptr, 0, 0, nullptr, 0, 0,
                      NK, 0, R, 0, R, P, L, static_cast<char *>(ws));
  return mstep_impl(stream, A, nullptr, z, prev, nullptr, zita_prev, out, nullptr, zita_out, kn_out, L, 0, nullptr, 0, 0, NK, R,
                    0, R, 0, P, L, static_cast<char *>(ws));
}

namespace {
struct MemWs {
  size_t kn, z, part, total;
};
MemWs memorize_ws(int N, int C, int V, int P, int L) {
  const int Pz = swem_em_pad(P), NK = 2 * N;
  MemWs w;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes, 256);
    return at;
  };
  w.kn = take((size_t)NK * L * C * 4);
  w.z = take((size_t)N * Pz * 2 * L * 4);
  size_t p1 = mstep_ws(NK, C, P, L).total, p2 = mstep_ws(NK, C + V, P, L).total;
  w.part = take(p1 > p2 ? p1 : p2);
  w.total = o;
  return w;
}

int memorize_impl(void *stream, const float *x, const float *v, const float *masks, const float *kappa_prev,
                  const float *nu_prev, const float *zita_prev, float *kappa_out, float *nu_out, float *zita_out, int N,
                  int C, int V, int P, int L, int T, float tau, void *ws, size_t ws_bytes, float *z_ext,
                  const float *kn_prior, int knp_rows, int knp_off, float *kn_out, int kno_rows, int kno_off,
                  float *mvp_out, int mvp_lm, int mvp_off) {
  SWEM_REQUIRE(x && v && masks && kappa_prev && nu_prev && zita_prev && kappa_out && nu_out && zita_out, SWEM_E_ARG,
               "memorize: null pointer");
  SWEM_REQUIRE(T >= 1, SWEM_E_ARG, "memorize: T < 1");
  SWEM_REQUIRE(kappa_out != kappa_prev && nu_out != nu_prev && zita_out != zita_prev, SWEM_E_ARG,
               "memorize: outputs must not alias the prior bases (the prior is read by every iteration)");
  SWEM_REQUIRE(C == 64 || C == 128, SWEM_E_SHAPE, "memorize: the key dimension must be 64 or 128 (got %d)", C);
  SWEM_REQUIRE(L == 64 || L == 128 || L == 256, SWEM_E_SHAPE, "memorize: L must be 64, 128 or 256 (got %d)", L);
  SWEM_REQUIRE(V % 128 == 0, SWEM_E_SHAPE, "memorize: the value dimension must be a multiple of 128 (got %d)", V);
  SWEM_REQUIRE(C == 128, SWEM_E_SHAPE, "memorize: the M step takes 128 key channels (got %d)", C);
  SWEM_REQUIRE(tau > 0.f, SWEM_E_ARG, "memorize: tau must be positive");
  MemWs w = memorize_ws(N, C, V, P, L);
  SWEM_REQUIRE(ws && ws_bytes >= w.total, SWEM_E_WORKSPACE, "memorize: workspace %zu < %zu", ws_bytes, w.total);
  char *base = static_cast<char *>(ws);
  float *kn = (float *)(base + w.kn);
  float *z = z_ext ? z_ext : (float *)(base + w.z);
  const int NK = 2 * N, Pz = swem_em_pad(P);
  int rc;
  if (z_ext && hipMemsetAsync(z_ext, 0, (size_t)N * Pz * 2 * L * 4, ST) != hipSuccess) {
    swem_set_error("memorize: memset failed");   // (training keeps z: rows [P, Pz) must read as zeros in the backward GEMM)
    return SWEM_E_HIP;
  }
  const float *kcur = kn_prior;
  int krows = knp_rows, koff = knp_off;
  if (!kcur) {
    if ((rc = swem_em_norm_bases_f32(stream, kappa_prev, kn, NK, C, L))) return rc;
    kcur = kn, krows = L, koff = 0;
  }
  for (int it = 0; it < T; ++it) {
    const bool last = it == T - 1;
    // W step of iteration it-1 (modules.py:161-162) and E step of iteration it share one GEMM
    if ((rc = ew_launch(stream, x, kcur, krows, koff, masks, masks, nullptr, z, N, C, P, L, tau, it > 0, 1))) return rc;
    // key bases every iteration; the value bases (modules.py:164-165) from the LAST z, in the same two launches
    if ((rc = mstep_impl(stream, x, last ? v : nullptr, z, kappa_prev, nu_prev, zita_prev, kappa_out, nu_out, zita_out,
                         last ? kn_out : kn, last ? kno_rows : L, last ? kno_off : 0, last ? mvp_out : nullptr, mvp_lm,
                         mvp_off, NK, C, last ? V : 0, C, V, P, L, base + w.part)))
      return rc;
    kcur = kn, krows = L, koff = 0;
  }
  return SWEM_OK;
}
}  // namespace

extern "C" size_t swem_memorize_workspace(int N, int C, int V, int P, int L) {
  return memorize_ws(N, C, V, P, L).total;
}

extern "C" int swem_memorize_f32(void *stream, const float *x, const float *v, const float *masks,
                                 const float *kappa_prev, const float *nu_prev, const float *zita_prev,
                                 float *kappa_out, float *nu_out, float *zita_out, int N, int C, int V, int P, int L,
                                 int T, float tau, void *ws, size_t ws_bytes) {
  return memorize_impl(stream, x, v, masks, kappa_prev, nu_prev, zita_prev, kappa_out, nu_out, zita_out, N, C, V, P, L, T,
                       tau, ws, ws_bytes, nullptr, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0);
}

// The same with matching's packed banks kept current (swem_match_packed_f32): the prior's normalised form is READ from
// the pack's 'update' half when `prior_packed` (it was written there by the previous frame's call), and the new bases'
// normalised keys / packed values are WRITTEN to bank `bank` (0 = 'first', 1 = 'update') of the pack.
extern "C" int swem_memorize_packed_f32(void *stream, const float *x, const float *v, const float *masks,
                                        const float *kappa_prev, const float *nu_prev, const float *zita_prev,
                                        float *kappa_out, float *nu_out, float *zita_out, float *mkn, float *mvp,
                                        int prior_packed, int bank, int N, int C, int V, int P, int L, int T, float tau,
                                        void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(mkn && mvp, SWEM_E_ARG, "memorize_packed: null pack");
  SWEM_REQUIRE(bank == 0 || bank == 1, SWEM_E_ARG, "memorize_packed: bank must be 0 or 1");
  return memorize_impl(stream, x, v, masks, kappa_prev, nu_prev, zita_prev, kappa_out, nu_out, zita_out, N, C, V, P, L, T,
                       tau, ws, ws_bytes, nullptr, prior_packed ? mkn : nullptr, 2 * L, L, mkn, 2 * L, bank * L, mvp,
                       2 * L, bank * L);
}

// training: the same as swem_memorize_f32, and the last E step's responsibilities z [N][Pz][2L] (Pz = swem_em_pad(P), rows
// >= P zero) are kept for the value update's backward
extern "C" int swem_memorize_train_f32(void *stream, const float *x, const float *v, const float *masks,
                                       const float *kappa_prev, const float *nu_prev, const float *zita_prev,
                                       float *kappa_out, float *nu_out, float *zita_out, float *z_out, int N, int C,
                                       int V, int P, int L, int T, float tau, void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(z_out, SWEM_E_ARG, "memorize_train: z_out is null");
  return memorize_impl(stream, x, v, masks, kappa_prev, nu_prev, zita_prev, kappa_out, nu_out, zita_out, N, C, V, P, L, T,
                       tau, ws, ws_bytes, z_out, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0);
}

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
  size_t Gp, dvp, conv, total;
};
NuBwdWs nu_bwd_ws(int N, int V, int P, int L) {
  NuBwdWs w;
  const int Pm = swem_em_pad(P);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes, 256);
    return at;
  };
  w.Gp = take((size_t)N * V * 2 * L * 4);
  w.dvp = take((size_t)N * Pm * V * 4);
  w.conv = take(swem_conv2d_workspace(N, Pm, 1, 2 * L, V, 1, 1, 1, 0, 0, 0));
  w.total = o;
  return w;
}
}  // namespace

extern "C" size_t swem_nu_update_bwd_workspace(int N, int V, int P, int L) { return nu_bwd_ws(N, V, P, L).total; }

extern "C" int swem_nu_update_bwd_f32(void *stream, const float *z, const float *zita_prev, const float *zita,
                                      const float *dnu, float *dv, float *dnu_prev, int N, int V, int P, int L, void *ws,
                                      size_t ws_bytes) {
  SWEM_REQUIRE(z && zita_prev && zita && dnu && dv, SWEM_E_ARG, "nu_update_bwd: null pointer");
  SWEM_REQUIRE(N > 0 && V % 4 == 0 && L % 32 == 0, SWEM_E_SHAPE, "nu_update_bwd: need V %% 4 == 0 and L %% 32 == 0");
  NuBwdWs w = nu_bwd_ws(N, V, P, L);
  SWEM_REQUIRE(ws && ws_bytes >= w.total, SWEM_E_WORKSPACE, "nu_update_bwd: workspace %zu < %zu", ws_bytes, w.total);
  char *base = static_cast<char *>(ws);
  float *Gp = (float *)(base + w.Gp), *dvp = (float *)(base + w.dvp);
  const int Pm = swem_em_pad(P);
  hipLaunchKernelGGL(nu_bwd_prep_kernel, dim3(cdiv((long long)N * 2 * V * L, 256)), dim3(256), 0, ST, dnu, zita,
                     zita_prev, Gp, dnu_prev, N, V, L);
  SWEM_CHECK_LAUNCH("nu_bwd_prep");
  // dv[n] = z[n] . Gp[n]^T : batched GEMM on the conv kernel; z is already pixel-major with Pm zero-padded rows per object
  // (a Pm x 1 image with 2L channels, V filters per object)
  int rc;
  if ((rc = swem_conv2d_nhwc_f32(stream, z, 2 * L, (long long)Pm * 2 * L, nullptr, 0, 0, nullptr, 0, 0, N, Pm, 1, Gp,
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
