// Sequential weighted EM (reference methods/SWEM/modules.py:93-168) on the gfx950 fp32 matrix cores.
//
// Data layout (device, fp32; NK = 2*N, class minor).  Every operand is read as it lies in memory: no transposed copies.
//   x  [P][C]            raw key, one row per pixel          (reference x_t; NHWC feature map as the encoder wrote it)
//   v  [N][P][V]         value map per object, pixel-major   (NHWC)
//   kp [NK][C/4+1][R][4] "packed keys": the key bases channel-group major, plus one group holding every base row's squared
//                        norm as C/32 partial sums; a bank occupies rows [off, off+L) of R >= L rows (R = L inside
//                        memorize; R = 2L, off = L when it is the 'update' half of matching's packed banks)
//   z  [N][Pz][2L]       responsibilities, one row per PIXEL: z[n][p][cls*L + l]; Pz = swem_em_pad(P) rows allocated,
//                        rows [0, ceil16(P)) written (pad pixels as zeros)
//
// TWO launches per EM iteration (a dependent kernel boundary costs ~1.5-2 us on this chip, a grid-wide barrier inside a
// persistent kernel 4-7 us and a split-K seam with a last-arriver combine 5-13 us: MI355X_MICROARCH.md, price list --
// so the iteration is two well-filled kernels, not one cooperative one):
//   em_ew16    : block = (object, 16-pixel tile), 8 waves = 2 classes x 4 base quarters.  One GEMM  s = l2norm(kappa) . x_t
//                on v_mfma_f32_16x16x4_f32 serves BOTH the W step of the previous iteration (cosine = s / (|x|+eps), joint
//                {bg,fg} max, exp-sums, weights = mask * (1 - p)) and the E step (softmax of s/tau over the class's bases,
//                times weights).  The pixel sits on the MFMA column lane, four bases of a tile in the accumulator
//                registers: row reductions are in-register + two shuffles + ONE LDS exchange for the maxima and one for the
//                sums.  The pixel's key (all C channels) lives in registers; every base row is loaded exactly once per
//                block, all loads issued before the first MFMA, and scaled by 1 / (|row| + eps) (l2norm, modules.py:7-9)
//                from the norms kept in the pack.  z leaves as 16-byte stores, 64 contiguous bytes per pixel and tile.
//                204 blocks at config B (P = 1620, N = 2).
//   em_mstep   : the whole M step: block = (object-class, 16 bases) x (32 rows of X) over ALL pixels, X = x (key bases) or
//                v[n] (value bases, last iteration only: both in ONE launch); 8 waves take the 4-pixel k-steps round robin,
//                the block sums them in wave order (deterministic), adds the column sums of z to zita, blends with the
//                prior (zita_prev * prev + S) / zita and writes the bases in the reference's layout AND as packed keys
//                with their norm partials (on the last iteration straight into matching's packed banks).
//                256 blocks for the key rows at config B: the only cut of this small GEMM that fills the chip without
//                splitting the pixel sum across blocks -- which is what made round 1's third kernel (and a 4 MB round
//                trip of partial sums) necessary.
#include "../../include/swem_hip_train.h"
#include "common.h"
#include "bf16_split.h"

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

// normalize = 1 (l2norm, modules.py:7-9):  kn[nk][c/4][off + l][c%4] = kappa[nk][c][l] / (||kappa[nk][:][l]|| + eps), C/4 groups.
// normalize = 0 ("packed keys", what the E/W and affinity kernels read): the plain re-layout, C/4 + 1 groups -- the extra
// group holds the squared norm of every base row as C/32 partial sums (one per 32 channels, balanced tree in channel
// order: exactly what the M step's blocks write, so a pack built here equals one the M step kept, bit for bit).
// Block: 32 bases x all channels.
__global__ __launch_bounds__(256) void em_norm_bases_kernel(const float *__restrict__ kappa, float *__restrict__ kn,
                                                            int C, int L, int out_rows, int out_off, int normalize) {
  extern __shared__ float sm[];  // tile[C][33], part[8][32], nrm[32]
  float *tile = sm, *part = sm + C * 33, *nrm = part + 256;
  const int nk = blockIdx.y, l0 = blockIdx.x * 32;
  const int l = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int KG = normalize ? C / 4 : C / 4 + 1;
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
    nrm[threadIdx.x] = normalize ? sqrtf(s) + SWEM_L2_EPS : 1.0f;
  }
  if (!normalize && g < 4 && l0 + l < L) {
    float q[32];
    float r = 0.f;
    if (32 * g < C) {
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const float t = tile[(32 * g + j) * 33 + l];
        q[j] = __fmul_rn(t, t);
      }
#pragma unroll
      for (int w = 1; w < 32; w <<= 1)
#pragma unroll
        for (int j = 0; j < 32; j += 2 * w) q[j] = __fadd_rn(q[j], q[j + w]);
      r = q[0];
    }
    kn[(((long long)nk * KG + C / 4) * out_rows + out_off + l0 + l) * 4 + g] = r;
  }
  __syncthreads();
  // channel-group major: the E/W and affinity kernels put one base row on every lane, so the 16 (32) rows' 16-byte chunks
  // of one k-step are contiguous
  for (int idx = threadIdx.x; idx < 32 * C; idx += 256) {
    const int e = idx & 3, ll = (idx >> 2) & 31, c4 = idx >> 7;
    if (l0 + ll < L)
      kn[(((long long)nk * KG + c4) * out_rows + out_off + l0 + ll) * 4 + e] = tile[(c4 * 4 + e) * 33 + ll] / nrm[ll];
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
                                                      int do_e, int xg STAMP_ARG) {
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
  const float *kb = kn + (((long long)nk * (C / 4 + 1) + g) * kn_rows + kn_off + q * 16 * LT + li) * 4;
  // issue order = arrival order (vmcnt counts in order): the pixel's key first, then the base rows chunk by chunk, so the
  // MFMAs of chunk m wait for chunk m only and the rest of the 256 KB streams in behind them
  // (buffer loads: pad pixels p >= P read as zeros by the range check -- no select in front of the MFMAs)
  // (xg objects share one key map: object n of a batch of clips reads clip n / xg's -- round 6, sequences in lock step)
  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x + (long long)(n / xg) * P * C), 0, P * C * 4,
                                                                  0x00020000);
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
  float4 nq[LT];   // the rows' squared norms (partial sums), group C/4 of the pack
#pragma unroll
  for (int t = 0; t < LT; ++t) nq[t] = ld4(kb + ((long long)(C / 4 - g) * kn_rows + 16 * t) * 4);
  // Every base row is requested before the first MFMA.  (Measured alternatives: left alone, the scheduler sinks each load
  // to just before its use -- one L2 round trip per chunk, vmcnt(1) in front of every MFMA group; streaming the rows 4
  // chunks ahead of the MFMAs that consume them, to start the chain while the CU's memory pipe -- 64 bytes a clock, ~2 us
  // for the 8 waves' 256 KB -- is still delivering, made the kernel 2 us SLOWER: the in-order issue of a wave's loads
  // between its MFMA groups stalls the chain more than the early start gains.  EW_AHEAD = CM keeps everything up front.)
  constexpr int EW_AHEAD = CM;
  float4 a[CM][LT];
#pragma unroll
  for (int m = 0; m < EW_AHEAD; ++m)
#pragma unroll
    for (int t = 0; t < LT; ++t) a[m][t] = ld4(kb + ((long long)4 * m * kn_rows + 16 * t) * 4);
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
  // l2norm of the bases (modules.py:7-9, :95, :114).  The rows arrive raw with their squared norms beside them (packed
  // keys: the M step writes both, no kernel in between normalises); every lane scales ITS row's fragments on the way into
  // the MFMA chain -- vector work that hides behind the other wave's MFMAs (measured: no change in kernel time).
  // Scaling the GEMM's OUTPUT instead (16 multiplies per lane, not 128) doubled the EM's rounding noise: bases grow to
  // norm ~24 at config sizes and the fp32 MFMA chain on un-normalised rows lands 2x further from float64 after three
  // iterations than the reference's own fp32 arithmetic (tools/em_noise.py); with the rows scaled first it lands as close.
  float rn[LT];
#pragma unroll
  for (int t = 0; t < LT; ++t) rn[t] = 1.0f / (sqrtf((nq[t].x + nq[t].y) + (nq[t].z + nq[t].w)) + SWEM_L2_EPS);
#pragma unroll
  for (int m = 0; m < CM; ++m) {
#pragma unroll
    for (int t = 0; t < LT; ++t) a[m][t].x *= rn[t], a[m][t].y *= rn[t], a[m][t].z *= rn[t], a[m][t].w *= rn[t];
    if (m + EW_AHEAD < CM) {
#pragma unroll
      for (int t = 0; t < LT; ++t) a[m + EW_AHEAD][t] = ld4(kb + ((long long)4 * (m + EW_AHEAD) * kn_rows + 16 * t) * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < LT; ++t) acc[t] = mfma16(a[m][t].x, xf[m].x, acc[t]);
#pragma unroll
    for (int t = 0; t < LT; ++t) acc[t] = mfma16(a[m][t].y, xf[m].y, acc[t]);
#pragma unroll
    for (int t = 0; t < LT; ++t) acc[t] = mfma16(a[m][t].z, xf[m].z, acc[t]);
#pragma unroll
    for (int t = 0; t < LT; ++t) acc[t] = mfma16(a[m][t].w, xf[m].w, acc[t]);
    __builtin_amdgcn_sched_barrier(0);
  }
  // (the accumulators now hold s = x . l2norm(kappa): the rows were scaled on the way in, above)
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

// ---------------------------------------------------------------------------------------------------- M step
struct MStepP {
  const float *x;  // [P][C] key rows (row space [0, Ck)), or NULL when Ck == 0
  const float *v;  // [N][P][V] value rows (row space [Ck, Ck + Vv)), or NULL
  const float *z;  // [N][Pz][2L]
  const float *kappa_prev, *nu_prev, *zita_prev;
  float *kappa_out, *nu_out, *zita_out;
  float *kp_out;   // optional: the new key bases packed [NK][C/4][kp_rows][4] at row offset kp_off (what E/W / affinity read)
  float *mvp_out;  // optional: value bases packed for matching, mvp[n][v][cls*mvp_lm + mvp_off + l]
  unsigned short *mvq_out;  // optional: the same as the fp16 pair (hi, mid: bf16_split.h, split2h) per object, [n][2][2*mvp_lm/8][V][8]
  unsigned *fault;          // optional: the caller's sticky fault word (SWEM_FAULT_RANGE: a value base beyond the fp16 range)
  int kp_rows, kp_off, mvp_lm, mvp_off;
  int Ck, C, V, P, Pz, L, NK, nrt, total;  // nrt = 32-row tiles of the row space, total = NK * (L/16) * nrt blocks
  int rpg;                                 // row tiles per group of the tile order (a divisor of nrt)
  int xg;                                  // objects per key map (x is [N / xg][P][C]: object n reads clip n / xg's keys)
};

// One launch = the whole M step (modules.py:122-127, 164-165), no partial sums in memory and no second kernel:
// block = (object-class, 16 bases) x (32 rows of the row space) over ALL pixels; the 8 waves take the 4-pixel k-steps
// round robin (wave w: steps w, w + 8, ...), two v_mfma_f32_16x16x4_f32 tiles each (A = z^T: base on the row lane, B =
// the 32 rows), and the block sums its 8 waves in wave order through the LDS (deterministic), adds the column sums of z to
// zita and blends with the prior: out = (zita_prev * prev + S) / zita.  512 output tiles of 16 x 16 at config B = two per
// CU: the only cut of this GEMM that fills 256 CUs without splitting the pixel sum across blocks.  Operands come by raw
// buffer loads (dwords: z is pixel-major, a lane needs ONE base of FOUR pixels per step; pixels >= P read as zeros by the
// range check on the vector offset), three groups of MG steps in flight ahead of the MFMA chain.
// Blocks that share z columns sit on one XCD (blockIdx -> tile order below): an XCD's L2 holds its 1/8 of z plus x.
template <int MG>  // k-steps per load group (3 dwords each)
struct MGroup {
  float za[MG], xa[MG], xb[MG];
};
template <int MG>
__device__ __forceinline__ void mgroup_load(MGroup<MG> &q, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, unsigned &oa,
                                            unsigned &ob, unsigned sa, unsigned sb) {
#pragma unroll
  for (int s = 0; s < MG; ++s) {   // running offsets: one address register per operand, not one per load in flight
    q.za[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, ob, 0, 0));
    q.xa[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, oa, 0, 0));
    q.xb[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, oa + 64u, 0, 0));
    oa += sa;
    ob += sb;
  }
}
// two accumulators per tile (even / odd steps of the group): independent MFMA chains back to back
template <int MG>
__device__ __forceinline__ void mgroup_mfma(const MGroup<MG> &q, f32x4 (&acc)[4], float &zs) {
#pragma unroll
  for (int s = 0; s < MG; ++s) {
    acc[2 * (s & 1)] = mfma16(q.za[s], q.xa[s], acc[2 * (s & 1)]);
    acc[2 * (s & 1) + 1] = mfma16(q.za[s], q.xb[s], acc[2 * (s & 1) + 1]);
  }
  // the column sums of z behind the chain (in front of it the adds would wait for the group's last load)
  asm volatile("" : "+v"(zs) : "v"(acc[0][0]));
#pragma unroll
  for (int s = 0; s < MG; ++s) zs += q.za[s];
}

// LOOP = false: at most three load groups per wave (P <= 96 MG pixels), straight-line code; WPE = waves per SIMD the
// register budget is cut for (blocks per CU = WPE / 2)
template <bool LOOP, int MG, int WPE>
__global__ __launch_bounds__(512, WPE) void em_mstep_kernel(MStepP p STAMP_ARG) {
  STAMP(0);
  __shared__ float red[8][2][4][64];
  __shared__ float zred[8][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  // consecutive blockIdx go round the 8 XCDs: XCD k takes the k-th eighth of the tile order (base tile major)
  const int per = gridDim.x >> 3;
  const int u = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (u >= p.total) return;
  // tile order: row tiles in groups of p.rpg, then base tile, then row tile inside the group -- an XCD's consecutive
  // blocks share few z columns AND few row columns, so that both fit its 4 MB L2 (sized by mstep_impl)
  const int tiles = p.L / 16;
  const int per_rg = p.NK * tiles * p.rpg;
  const int rg = u / per_rg, ur = u - rg * per_rg;
  const int bt = ur / p.rpg, rt = rg * p.rpg + (ur - bt * p.rpg);
  const int nk = bt / tiles, l0 = (bt - nk * tiles) * 16;
  const int n = nk >> 1, cls = nk & 1;
  const int row0 = rt * 32;
  const bool key = row0 < p.Ck;
  const float *src = key ? p.x + (long long)(n / p.xg) * p.P * p.C : p.v + (long long)n * p.P * p.V;
  const int stride = key ? p.C : p.V;
  const int col = key ? row0 : row0 - p.Ck;
  __amdgpu_buffer_rsrc_t ra =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, (int)((long long)p.P * stride * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.z + (long long)n * p.Pz * 2 * p.L), 0, (int)((long long)p.P * 2 * p.L * 4), 0x00020000);
  // the epilogue's operands first: output element (base l0 + oi, row col + oj) of this thread.  Key rows: a wave = 2 bases x
  // 32 rows (the squared-norm partials are a reduction over the rows: five shuffles).  Value rows: a wave = 16 bases x 4
  // rows -- the 16 bases of a row are 64 contiguous bytes of nu / mvp, one cache line segment per four lanes' worth (with
  // the key mapping every lane of a store would touch its own line: 1.7 us of a value block's 2.7 us epilogue)
  const int oj = key ? (tid & 31) : (tid >> 4), oi = key ? (tid >> 5) : (tid & 15);
  const int R = key ? p.C : p.V;
  const float zp = p.zita_prev[(long long)nk * p.L + l0 + oi];
  const float pv = (key ? p.kappa_prev : p.nu_prev)[((long long)nk * R + col + oj) * p.L + l0 + oi];
  // pixel of (step t of this wave, lane group g) = 4 * (wave + 8 t) + g; it travels in the VECTOR offset (range-checked)
  const unsigned va = (unsigned)(((4 * wave + g) * stride + col + li) * 4);
  const unsigned vb = (unsigned)(((4 * wave + g) * 2 * p.L + cls * p.L + l0 + li) * 4);
  const unsigned sa = (unsigned)stride * 128u, sb = (unsigned)p.L * 256u;  // bytes per step of one wave (32 pixels)
  const int T = ((p.P + 3) / 4 + 7) / 8, ng = (T + MG - 1) / MG;
  MGroup<MG> qa, qb, qc;
  unsigned oa = va, ob = vb;   // groups are loaded in step order, whichever registers they land in
  mgroup_load(qa, ra, rb, oa, ob, sa, sb);
  mgroup_load(qb, ra, rb, oa, ob, sa, sb);   // (a group past the last pixel reads zeros without touching memory)
  __builtin_amdgcn_sched_barrier(0);  // two groups in flight before the MFMA chain (see em_ew16_kernel)
  STAMP(1);
  f32x4 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  float zs = 0.f;
  // the third group is requested behind the first one's MFMAs: by then the memory pipe has delivered most of the first two
  // (it moves 64 bytes a clock per CU: the 8 waves' two groups are ~1.5 us of its time), and the wait counter (63 at most)
  // still resolves every group exactly
#define MPHASE(stmt) stmt; __builtin_amdgcn_sched_barrier(0)   /* phases stay in program order (register pressure) */
  if constexpr (!LOOP) {   // up to 1632 pixels (config B: 1620): straight-line code, the waits are exact per group
    MPHASE(mgroup_mfma(qa, acc, zs));
    MPHASE(mgroup_load(qc, ra, rb, oa, ob, sa, sb));
    if (ng > 1) { MPHASE(mgroup_mfma(qb, acc, zs)); }
    if (ng > 2) { MPHASE(mgroup_mfma(qc, acc, zs)); }
  } else {
    for (int gi = 0; gi < ng; gi += 3) {
      MPHASE(mgroup_mfma(qa, acc, zs));
      MPHASE(mgroup_load(qc, ra, rb, oa, ob, sa, sb));
      MPHASE(mgroup_mfma(qb, acc, zs));
      MPHASE(mgroup_load(qa, ra, rb, oa, ob, sa, sb));
      MPHASE(mgroup_mfma(qc, acc, zs));
      MPHASE(mgroup_load(qb, ra, rb, oa, ob, sa, sb));
    }
  }
#undef MPHASE
  zs += __shfl_xor(zs, 16);
  zs += __shfl_xor(zs, 32);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[wave][0][e][lane] = acc[0][e] + acc[2][e];
    red[wave][1][e][lane] = acc[1][e] + acc[3][e];
  }
  if (lane < 16) zred[wave][lane] = zs;
  STAMP(2);
  __syncthreads();
  STAMP(3);
  // D[i = base][j = row] of a tile lives in lane j + 16 (i >> 2), register i & 3
  const int tt = oj >> 4, sl = (oj & 15) + 16 * (oi >> 2), sr = oi & 3;
  float S = 0.f, zsum = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    S += red[w][tt][sr][sl];
    zsum += zred[w][oi];
  }
  const float zt = zp + zsum;
  const float val = (zp * pv + S) / zt;
  const int row = col + oj;
  if (key) {
    p.kappa_out[((long long)nk * p.C + row) * p.L + l0 + oi] = val;
    if (p.kp_out) {
      const int KG = p.C / 4 + 1;
      p.kp_out[(((long long)nk * KG + (row >> 2)) * p.kp_rows + p.kp_off + l0 + oi) * 4 + (row & 3)] = val;
      // the row tile's share of the bases' squared norms (group C/4 of the pack): balanced tree over the 32 rows, the order
      // em_norm_bases_kernel reproduces
      float q = __fmul_rn(val, val);
#pragma unroll
      for (int w = 1; w < 32; w <<= 1) q = __fadd_rn(q, __shfl_xor(q, w));
      float *nq = p.kp_out + (((long long)nk * KG + p.C / 4) * p.kp_rows + p.kp_off + l0 + oi) * 4;
      if (oj == 0) nq[rt] = q;
      if (rt == 0 && p.C == 64 && oj >= 2 && oj < 4) nq[oj] = 0.f;
    }
    if (rt == 0 && oj == 0 && p.zita_out) p.zita_out[(long long)nk * p.L + l0 + oi] = zt;
  } else {
    p.nu_out[((long long)nk * p.V + row) * p.L + l0 + oi] = val;
    if (p.mvp_out) p.mvp_out[((long long)n * p.V + row) * (2 * p.mvp_lm) + cls * p.mvp_lm + p.mvp_off + l0 + oi] = val;
    if (p.Ck == 0 && rt == 0 && oj == 0 && p.zita_out) p.zita_out[(long long)nk * p.L + l0 + oi] = zt;
    if (p.mvq_out) {
      // the readout GEMM's pre-split filter planes (match.hip): 8 consecutive bases of one value row are one 16-byte run, so
      // the tile turns through the LDS (the reduction buffer is free again) and 64 threads store 16 bytes per plane
      __syncthreads();
      float *vt = &red[0][0][0][0];   // [16 bases][33]
      vt[oi * 33 + oj] = val;
      __syncthreads();
      if (tid < 64) {
        const int grp = tid >> 5, vr = tid & 31;
        float4 a = make_float4(vt[(8 * grp) * 33 + vr], vt[(8 * grp + 1) * 33 + vr], vt[(8 * grp + 2) * 33 + vr],
                               vt[(8 * grp + 3) * 33 + vr]);
        float4 b = make_float4(vt[(8 * grp + 4) * 33 + vr], vt[(8 * grp + 5) * 33 + vr], vt[(8 * grp + 6) * 33 + vr],
                               vt[(8 * grp + 7) * 33 + vr]);
        uint2 h0, m0, h1, m1;
        split2h(a, h0, m0);
        split2h(b, h1, m1);
        range_fault(p.fault, f16_oor(a) | f16_oor(b));   // (tid < 64: wave 0, all lanes)
        const int kg = (cls * p.mvp_lm + p.mvp_off + l0) / 8 + grp, ngrp = 2 * p.mvp_lm / 8;
        unsigned short *base = p.mvq_out + (long long)n * 2 * ngrp * p.V * 8;
        *reinterpret_cast<uint4 *>(base + ((long long)kg * p.V + col + vr) * 8) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4 *>(base + ((long long)(ngrp + kg) * p.V + col + vr) * 8) = make_uint4(m0.x, m0.y, m1.x, m1.y);
      }
    }
  }
  STAMP(4);
#ifdef SWEM_EM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(5);
#endif
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" int swem_em_pad(int P) { return (P + 127) / 128 * 128; }

// rows of bank `kappa` land at row out_off + l of an [NK][C/4][out_rows][4] image (matching concatenates banks);
// normalize = 0: the plain re-layout ("packed keys": their readers normalise the rows themselves)
int swem_norm_bases_into(void *stream, const float *kappa, float *kn, int NK, int C, int L, int out_rows, int out_off,
                         int normalize) {
  SWEM_REQUIRE(kappa && kn && NK > 0 && C > 0 && L > 0, SWEM_E_ARG, "em_norm_bases: bad argument");
  SWEM_REQUIRE(C <= 1024 && C % 4 == 0, SWEM_E_SHAPE, "em_norm_bases: C must be a multiple of 4, at most 1024");
  size_t lds = ((size_t)C * 33 + 256 + 32) * sizeof(float);
  hipLaunchKernelGGL(em_norm_bases_kernel, dim3(cdiv(L, 32), NK), dim3(256), lds, ST, kappa, kn, C, L, out_rows,
                     out_off, normalize);
  SWEM_CHECK_LAUNCH("em_norm_bases");
  return SWEM_OK;
}

extern "C" int swem_em_norm_bases_f32(void *stream, const float *kappa, float *kn, int NK, int C, int L) {
  return swem_norm_bases_into(stream, kappa, kn, NK, C, L, L, 0, 1);
}

extern "C" int swem_em_pack_bases_f32(void *stream, const float *kappa, float *kp, int NK, int C, int L) {
  return swem_norm_bases_into(stream, kappa, kp, NK, C, L, L, 0, 0);
}

namespace {
int ew_launch(void *stream, const float *x, const float *kn, int kn_rows, int kn_off, const float *masks,
              const float *w_in, float *w_out, float *z, int N, int C, int P, int L, float tau, int do_w, int do_e,
              int xg = 1 << 30) {
  const int Pz = swem_em_pad(P);
  dim3 grid(cdiv(P, 16), N);
#define EW(LT_, CM_)                                                                                                  \
  hipLaunchKernelGGL((em_ew16_kernel<LT_, CM_>), grid, dim3(512), 0, ST, x, kn, kn_rows, kn_off, masks, w_in, w_out, z, \
                     P, Pz, tau, do_w, do_e, xg STAMP_PASS)
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
// One M step over the row space [keys (Ck rows of x) | values (Vv rows of v)]: one launch.
int mstep_impl(void *stream, const float *x, const float *v, const float *z, const float *kappa_prev,
               const float *nu_prev, const float *zita_prev, float *kappa_out, float *nu_out, float *zita_out,
               float *kp_out, int kp_rows, int kp_off, float *mvp_out, int mvp_lm, int mvp_off, int NK, int Ck, int Vv,
               int C, int V, int P, int L, unsigned short *mvq_out = nullptr, unsigned *fault = nullptr, int xg = 1 << 30) {
  MStepP mp;
  mp.fault = fault;
  mp.xg = xg;
  mp.x = x, mp.v = v, mp.z = z;
  mp.kappa_prev = kappa_prev, mp.nu_prev = nu_prev, mp.zita_prev = zita_prev;
  mp.kappa_out = kappa_out, mp.nu_out = nu_out, mp.zita_out = zita_out;
  mp.kp_out = kp_out, mp.mvp_out = mvp_out, mp.mvq_out = mvq_out;
  mp.kp_rows = kp_rows, mp.kp_off = kp_off, mp.mvp_lm = mvp_lm, mp.mvp_off = mvp_off;
  mp.Ck = Ck, mp.C = C, mp.V = V, mp.P = P, mp.Pz = swem_em_pad(P), mp.L = L, mp.NK = NK;
  mp.nrt = (Ck + Vv) / 32;
  mp.total = NK * (L / 16) * mp.nrt;
  // An XCD runs total/8 consecutive tiles = (that / rpg) base tiles x rpg row tiles and reads 64 bytes per pixel of z for
  // each of the former, 128 of x / v for each of the latter: least for rpg ~ sqrt(total / 16).  (Keys + values at config
  // B: 1280 tiles; with all 20 row tiles in one group an XCD touches 5 MB -- more than its L2 -- with 10 it is 3.7 MB.)
  mp.rpg = mp.nrt;
  {
    const float want = sqrtf((float)mp.total / 16.f);
    float best = 1e30f;
    for (int d = 1; d <= mp.nrt; ++d)
      if (mp.nrt % d == 0 && fabsf((float)d - want) < best) best = fabsf((float)d - want), mp.rpg = d;
  }
  const dim3 grid((mp.total + 7) / 8 * 8);
  // (measured alternatives: smaller load groups cut for three or four resident blocks per CU -- <true, 6, 6>, <true, 4, 8> --
  // spill; the two-block form below is what the register file allows with the operands prefetched into registers)
  if (P <= 4 * 8 * 3 * 17) hipLaunchKernelGGL((em_mstep_kernel<false, 17, 4>), grid, dim3(512), 0, ST, mp STAMP_PASS);
  else hipLaunchKernelGGL((em_mstep_kernel<true, 17, 2>), grid, dim3(512), 0, ST, mp STAMP_PASS);
  SWEM_CHECK_LAUNCH("em_mstep");
  return SWEM_OK;
}
}  // namespace

// (the M step needs no scratch any more; the query stays in the ABI and reports zero)
extern "C" size_t swem_em_mstep_workspace(int NK, int R, int P, int L) {
  (void)NK, (void)R, (void)P, (void)L;
  return 0;
}

extern "C" int swem_em_mstep_f32(void *stream, const float *A, int a_per_object, const float *z, const float *prev,
                                 const float *zita_prev, float *out, float *zita_out, float *kp_out, int NK, int R,
                                 int P, int L, void *ws, size_t ws_bytes) {
  (void)ws, (void)ws_bytes;
  SWEM_REQUIRE(A && z && prev && zita_prev && out, SWEM_E_ARG, "em_mstep: null pointer");
  SWEM_REQUIRE(NK % 2 == 0 && L % 16 == 0 && R % 32 == 0, SWEM_E_SHAPE,
               "em_mstep: need NK even, L %% 16 == 0 and R %% 32 == 0 (got %d, %d, %d)", NK, L, R);
  SWEM_REQUIRE(!kp_out || !a_per_object, SWEM_E_SHAPE, "em_mstep: kp_out goes with the key rows (shared A)");
  if (a_per_object)  // value rows: A = v [N][P][R]
    return mstep_impl(stream, nullptr, A, z, nullptr, prev, zita_prev, nullptr, out, zita_out, nullptr, 0, 0, nullptr, 0, 0,
                      NK, 0, R, 0, R, P, L);
  return mstep_impl(stream, A, nullptr, z, prev, nullptr, zita_prev, out, nullptr, zita_out, kp_out, L, 0, nullptr, 0, 0, NK, R,
                    0, R, 0, P, L);
}

namespace {
struct MemWs {
  size_t kn, z, total;
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
  w.kn = take((size_t)NK * L * (C + 4) * 4);   // packed keys: C/4 + 1 groups
  w.z = take((size_t)N * Pz * 2 * L * 4);
  w.total = o;
  return w;
}

int memorize_impl(void *stream, const float *x, const float *v, const float *masks, const float *kappa_prev,
                  const float *nu_prev, const float *zita_prev, float *kappa_out, float *nu_out, float *zita_out, int N,
                  int C, int V, int P, int L, int T, float tau, void *ws, size_t ws_bytes, float *z_ext,
                  const float *kn_prior, int knp_rows, int knp_off, float *kn_out, int kno_rows, int kno_off,
                  float *mvp_out, int mvp_lm, int mvp_off, unsigned short *mvq_out = nullptr, bool keys_only = false,
                  unsigned *fault = nullptr, int clips = 1) {
  // clips > 1 (round 6): the N objects are those of `clips` clips, N / clips each, and x holds one key map per clip [clips][P][C];
  // everything else is per object already.  The launches are the single-clip ones with `clips` times the objects: per object the
  // same blocks on the same data.
  // keys_only: everything that does not read the value map -- all T (E, W, key M) steps; the last E step's z stays in z_ext
  // for the value update (swem_memorize_packed_values_f32), which is ONE more M-step launch over the value rows
  SWEM_REQUIRE(x && (v || keys_only) && masks && kappa_prev && (nu_prev || keys_only) && zita_prev && kappa_out &&
                   (nu_out || keys_only) && zita_out, SWEM_E_ARG, "memorize: null pointer");
  SWEM_REQUIRE(!keys_only || z_ext, SWEM_E_ARG, "memorize (keys): the responsibilities need a buffer of their own");
  SWEM_REQUIRE(T >= 1, SWEM_E_ARG, "memorize: T < 1");
  SWEM_REQUIRE(clips >= 1 && N % clips == 0, SWEM_E_SHAPE, "memorize: %d objects do not divide into %d clips", N, clips);
  const int xg = N / clips;
  SWEM_REQUIRE(kappa_out != kappa_prev && (keys_only || nu_out != nu_prev) && zita_out != zita_prev, SWEM_E_ARG,
               "memorize: outputs must not alias the prior bases (the prior is read by every iteration)");
  SWEM_REQUIRE(C == 64 || C == 128, SWEM_E_SHAPE, "memorize: the key dimension must be 64 or 128 (got %d)", C);
  SWEM_REQUIRE(L == 64 || L == 128 || L == 256, SWEM_E_SHAPE, "memorize: L must be 64, 128 or 256 (got %d)", L);
  SWEM_REQUIRE(V % 32 == 0, SWEM_E_SHAPE, "memorize: the value dimension must be a multiple of 32 (got %d)", V);
  SWEM_REQUIRE(tau > 0.f, SWEM_E_ARG, "memorize: tau must be positive");
  MemWs w = memorize_ws(N, C, V, P, L);
  SWEM_REQUIRE(ws && ws_bytes >= w.total, SWEM_E_WORKSPACE, "memorize: workspace %zu < %zu", ws_bytes, w.total);
  char *base = static_cast<char *>(ws);
  float *kn = (float *)(base + w.kn);
  float *z = z_ext ? z_ext : (float *)(base + w.z);
  const int NK = 2 * N, Pz = swem_em_pad(P);
  int rc;
  if (z_ext && !keys_only && hipMemsetAsync(z_ext, 0, (size_t)N * Pz * 2 * L * 4, ST) != hipSuccess) {
    swem_set_error("memorize: memset failed");   // (training keeps z: rows [P, Pz) must read as zeros in the backward GEMM)
    return SWEM_E_HIP;
  }
  const float *kcur = kn_prior;
  int krows = knp_rows, koff = knp_off;
  if (!kcur) {
    if ((rc = swem_em_pack_bases_f32(stream, kappa_prev, kn, NK, C, L))) return rc;
    kcur = kn, krows = L, koff = 0;
  }
  for (int it = 0; it < T; ++it) {
    const bool last = it == T - 1;
    // W step of iteration it-1 (modules.py:161-162) and E step of iteration it share one GEMM
    if ((rc = ew_launch(stream, x, kcur, krows, koff, masks, masks, nullptr, z, N, C, P, L, tau, it > 0, 1, xg))) return rc;
    // key bases every iteration; the value bases (modules.py:164-165) from the LAST z, in the same two launches
    const bool vals = last && !keys_only;
    if ((rc = mstep_impl(stream, x, vals ? v : nullptr, z, kappa_prev, nu_prev, zita_prev, kappa_out, nu_out, zita_out,
                         last ? kn_out : kn, last ? kno_rows : L, last ? kno_off : 0, vals ? mvp_out : nullptr, mvp_lm,
                         mvp_off, NK, C, vals ? V : 0, C, V, P, L, vals ? mvq_out : nullptr, fault, xg)))
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
                                        void *mvq, int prior_packed, int bank, int N, int C, int V, int P, int L, int T,
                                        float tau, void *ws, size_t ws_bytes, void *fault) {
  SWEM_REQUIRE(mkn && mvp, SWEM_E_ARG, "memorize_packed: null pack");
  SWEM_REQUIRE(bank == 0 || bank == 1, SWEM_E_ARG, "memorize_packed: bank must be 0 or 1");
  return memorize_impl(stream, x, v, masks, kappa_prev, nu_prev, zita_prev, kappa_out, nu_out, zita_out, N, C, V, P, L, T,
                       tau, ws, ws_bytes, nullptr, prior_packed ? mkn : nullptr, 2 * L, L, mkn, 2 * L, bank * L, mvp,
                       2 * L, bank * L, static_cast<unsigned short *>(mvq), false, static_cast<unsigned *>(fault));
}

// The same for the objects of SEVERAL clips in one call (round 6: sequences in lock step): N objects in total, N / clips per
// clip, x = one key map per clip [clips][P][C]; v, masks, bases and the pack are per object as before (the clips' objects
// back to back).  Per object the launches run the same blocks on the same data as `clips` single-clip calls: identical results.
extern "C" int swem_memorize_packed_clips_f32(void *stream, const float *x, const float *v, const float *masks,
                                              const float *kappa_prev, const float *nu_prev, const float *zita_prev,
                                              float *kappa_out, float *nu_out, float *zita_out, float *mkn, float *mvp,
                                              void *mvq, int prior_packed, int bank, int N, int clips, int C, int V, int P, int L,
                                              int T, float tau, void *ws, size_t ws_bytes, void *fault) {
  SWEM_REQUIRE(mkn && mvp, SWEM_E_ARG, "memorize_packed_clips: null pack");
  SWEM_REQUIRE(bank == 0 || bank == 1, SWEM_E_ARG, "memorize_packed_clips: bank must be 0 or 1");
  return memorize_impl(stream, x, v, masks, kappa_prev, nu_prev, zita_prev, kappa_out, nu_out, zita_out, N, C, V, P, L, T,
                       tau, ws, ws_bytes, nullptr, prior_packed ? mkn : nullptr, 2 * L, L, mkn, 2 * L, bank * L, mvp,
                       2 * L, bank * L, static_cast<unsigned short *>(mvq), false, static_cast<unsigned *>(fault), clips);
}

// The packed memorize in two calls, so that a caller can run the part that does not need the value map -- every E, W and key
// M step: 2T - 1 of the 2T launches -- BESIDE the value encoder that produces it (evaluator.frame_chain on a side stream):
// `keys` leaves the last E step's responsibilities in z [N][swem_em_pad(P)][2L]; `values` is the value update
// nu = (zita_prev nu_prev + v . z) / zita (modules.py:164-165) from them, written to bank `bank` of the pack like the one-call
// form does.  Together they launch the same blocks on the same data as swem_memorize_packed_f32: identical results.
extern "C" int swem_memorize_packed_keys_f32(void *stream, const float *x, const float *masks, const float *kappa_prev,
                                             const float *zita_prev, float *kappa_out, float *zita_out, float *mkn, float *z,
                                             int prior_packed, int bank, int N, int C, int P, int L, int T, float tau,
                                             void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(mkn && z, SWEM_E_ARG, "memorize_packed_keys: null pack / z");
  SWEM_REQUIRE(bank == 0 || bank == 1, SWEM_E_ARG, "memorize_packed_keys: bank must be 0 or 1");
  return memorize_impl(stream, x, nullptr, masks, kappa_prev, nullptr, zita_prev, kappa_out, nullptr, zita_out, N, C, 32, P, L,
                       T, tau, ws, ws_bytes, z, prior_packed ? mkn : nullptr, 2 * L, L, mkn, 2 * L, bank * L, nullptr, 2 * L,
                       bank * L, nullptr, true);
}
extern "C" int swem_memorize_packed_values_f32(void *stream, const float *v, const float *z, const float *nu_prev,
                                               const float *zita_prev, float *nu_out, float *mvp, void *mvq, int bank, int N,
                                               int V, int P, int L, void *fault) {
  SWEM_REQUIRE(v && z && nu_prev && zita_prev && nu_out && mvp, SWEM_E_ARG, "memorize_packed_values: null pointer");
  SWEM_REQUIRE(bank == 0 || bank == 1, SWEM_E_ARG, "memorize_packed_values: bank must be 0 or 1");
  SWEM_REQUIRE(V % 32 == 0 && (L == 64 || L == 128 || L == 256), SWEM_E_SHAPE, "memorize_packed_values: V %% 32, L in {64,128,256}");
  return mstep_impl(stream, nullptr, v, z, nullptr, nu_prev, zita_prev, nullptr, nu_out, nullptr, nullptr, 0, 0, mvp, 2 * L,
                    bank * L, 2 * N, 0, V, 0, V, P, L, static_cast<unsigned short *>(mvq), static_cast<unsigned *>(fault));
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
