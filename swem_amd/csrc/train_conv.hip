// Weight gradient of the implicit-GEMM convolution and the column reductions of the backward pass (gfx950).
//
//   dW[n][k] = sum_m dY[m][n] * A[m][k]        n = filter, k = (ky, kx, ci), m = output pixel (the reduction axis)
//
// A "TN" GEMM: both operands are read pixel-major ([m][channel], channels contiguous), so a 32-pixel slab of each is
// staged in LDS as it lies in memory and the MFMA operands are read down the pixel axis: lane (i, h) of
// v_mfma_f32_32x32x2_f32 takes dY[m = 2kk + h][n = i] and A[m = 2kk + h][k = i] -- consecutive lanes read consecutive
// floats of one LDS row (conflict free).  Block = 2x2 waves, wave tile (32 WT) x (32 WT); one block owns one
// (filter tile, tap, channel tile) and a slice of the pixels (grid.z); the slices are summed in a fixed order by the
// reduce kernel, which also writes the reference's OIHW layout and accumulates into the gradient buffer.
#include "../../include/swem_hip_train.h"
#include "common.h"
#include "lds_dma.h"
#include "bf16_split.h"

namespace {

struct WgradP {
  const float *dy;
  const float *x;
  long long bs;       // batch stride of x in elements (0 = one map shared by every batch item)
  int cs, c_off;      // channels of this source, its offset inside the concatenated Cin
  int B, H, W, Ho, Wo, Cout, Cin, KH, KW, stride, pad, relu_in;
  int M, m_per_split, ctiles, K;
  float *partial;     // [zsplit][Cout][K]
};

__device__ __forceinline__ float4 ld4g(const float *p) { return *reinterpret_cast<const float4 *>(p); }

template <int WT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradP p) {
  constexpr int BT = 64 * WT;        // tile edge (filters and channels)
  constexpr int LD = BT + 4;         // LDS row stride in floats (keeps 16-byte alignment, shifts the second half-wave)
  constexpr int NL = BT / 32;        // float4 loads per thread per operand per 32-pixel slab
  __shared__ __attribute__((aligned(16))) float Ds[32 * LD];
  __shared__ __attribute__((aligned(16))) float Xs[32 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * BT;
  const int tap = blockIdx.y / p.ctiles, ci0 = (blockIdx.y - tap * p.ctiles) * BT;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  const int m_begin = blockIdx.z * p.m_per_split, m_end = min(p.M, m_begin + p.m_per_split);
  const int HoWo = p.Ho * p.Wo;
  // one slab row (pixel) per 8 threads, NL consecutive float4 each: the pixel -> (b, oy, ox) decode (two integer
  // divisions; vector instructions do not overlap the fp32 MFMA) is done once per thread and slab
  const int lrow = tid >> 3, lc4 = (tid & 7) * NL;
  float4 gd[NL], gx[NL];
  auto gload = [&](int mb) {
    const int m = mb + lrow;
    const bool mok = m < m_end;
    const int b = m / HoWo, rem = m - b * HoWo;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
    const bool pok = mok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const float *drow = p.dy + (long long)m * p.Cout + n0;
    const float *xrow = p.x + (long long)b * p.bs + ((long long)iy * p.W + ix) * p.cs + ci0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int cc = (lc4 + i) * 4;
      gd[i] = (mok && n0 + cc < p.Cout) ? ld4g(drow + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 v = (pok && ci0 + cc < p.cs) ? ld4g(xrow + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.relu_in) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      gx[i] = v;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      *reinterpret_cast<float4 *>(&Ds[lrow * LD + (lc4 + i) * 4]) = gd[i];
      *reinterpret_cast<float4 *>(&Xs[lrow * LD + (lc4 + i) * 4]) = gx[i];
    }
  };
  f32x16 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  gload(m_begin);
  for (int mb = m_begin; mb < m_end; mb += 32) {
    __syncthreads();   // the previous slab has been consumed
    lstore();
    __syncthreads();
    if (mb + 32 < m_end) gload(mb + 32);
    const float *Db = Ds + h * LD + wn * 32 * WT + r;
    const float *Xb = Xs + h * LD + wc * 32 * WT + r;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[WT], b[WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) a[i] = Db[2 * kk * LD + 32 * i];
#pragma unroll
      for (int j = 0; j < WT; ++j) b[j] = Xb[2 * kk * LD + 32 * j];
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    }
  }
  float *dst = p.partial + (long long)blockIdx.z * p.Cout * p.K + (long long)tap * p.Cin + p.c_off;
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int c = ci0 + wc * 32 * WT + 32 * j + r;
      if (c >= p.cs) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + wn * 32 * WT + 32 * i + acc_row(e, h);
        if (n < p.Cout) dst[(long long)n * p.K + c] = acc[i][j][e];
      }
    }
}

// ------------------------------------------------------------------------------------------------------------
// The same weight gradient on the bf16 matrix pipe, from the PRE-SPLIT planes the forward / data-gradient convolutions
// already made (swem_split_bf16x3_f32: [plane][C/8][pixel][8 bf16]).  NPL = 3: exact three-way split of both operands,
// six products, fp32 accumulate (fp32-level error, as the forward's bf16x6 mode); NPL = 1: plane 0 only, one product
// (config.AMP).
//
// The reduction axis is the PIXEL, which is the strided axis of both operands -- the bf16 MFMA wants 8 consecutive k per
// lane.  LDS keeps 32-pixel x 64-channel images as they lie in memory ([pixel][64 bf16] = 128-byte rows) and the
// fragments are read with ds_read_b64_tr_b16, the hardware transpose (4 pixels x 16 channels per 16-lane group).  The
// images are filled by buffer_load ... lds: one transfer = 8 pixel rows x 8 sixteen-byte chunks, lane -> LDS slot is
// fixed (lane-linear), so the conflict-avoiding swizzle -- chunk ^= 4 for rows with bit 1 set; the four rows one
// transposed read touches then sit in four different 64-byte bank groups -- is applied on the GLOBAL side: the lane
// that owns slot (row, c') fetches chunk c' ^ swz(row).  Each chunk is one (pixel, 8-channel group) cell of the
// channel-group-major planes, so a transfer reads eight 128-byte lines; taps that fall outside the image and rows
// past the slice use an out-of-range offset and land as zeros.  Wave w fills rows 8w..8w+7 of every image of a slab:
// one pixel decode per lane, advanced by 32 pixels per slab with carries instead of divisions.
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef short wg_s16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned WG_OOB = 0xfffffff0u;

struct WgradBP {
  const unsigned short *dy3;  // [3][Cout/8][M][8]
  const unsigned short *x3;   // [3][cs/8][npix][8] of the current source
  long long dps, xps;         // plane strides in elements
  long long xbs_pix;          // batch stride of x in pixels (0 = one map shared by every batch item)
  int xnpix;                  // pixels per plane of x
  int cs, c_off;
  int B, H, W, Ho, Wo, Cout, Cin, KH, KW, stride, pad;
  int M, m_per_split, ctiles, K;
  float *partial;             // [zsplit][Cout][K]
};

__device__ __forceinline__ wg_s16x4 lds_tr16(unsigned addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_s16x4 __attribute__((address_space(3))) *)(size_t)addr);
}
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x16 wg_mfma(wg_s16x8 a, wg_s16x8 b, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wg_f16x8, a), __builtin_bit_cast(wg_f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, a), __builtin_bit_cast(wg_bf16x8, b), c, 0,
                                                   0, 0);
}

// KS = pixels per slab: 32 (every wave fills rows 8w..8w+7 of all images) or 16 (waves 0,1 fill the dY images, waves 2,3
// the x images; half the LDS per stage, so the six-product mode keeps three blocks per CU instead of one).
// F16 (round 5, NPL = 2): the planes are fp16 (hi, mid) pairs -- dY's scaled by a power of two (swem_split_f16x2_scaled_f32) --
// three products hi.mid + mid.hi + hi.hi on the f16 MFMA: the f16x3 arithmetic of the forward pass (fp32-level error) at half
// the matrix work of the six-product mode.
// NST (round 6): stages of the slab ring.  Rounds 2-5 ran two stages with a full vmcnt(0) + barrier per slab: a slab is 12-24
// MFMAs per wave (400-800 cycles), the L2 -> LDS round trip it must hide 2-4k -- only the CU's other resident blocks hid it.  With
// NST stages the transfers of slab s + NST - 1 are requested at the top of slab s and the hand-over waits for slab s + 1 only
// (counted vmcnt, as conv_igemm_bf3s_kernel does): NST - 2 slabs stay in flight across every barrier.  The slabs are consumed
// in the same order, so the sums are bit-identical to the two-stage form.
template <int WT, int NPL, int KS, bool F16 = false, int NST = 2>
__global__ __launch_bounds__(256) void conv_wgrad_bf_kernel(WgradBP p) {
  static_assert(KS == 16 || KS == 32, "slab of 16 or 32 pixels");
  static_assert(!F16 || NPL == 2, "fp16 planes come as a (hi, mid) pair");
  static_assert(NST >= 2 && NST <= 4, "two to four stages");
  constexpr int NDMA = (KS == 32 ? 2 : 1) * WT * NPL;   // 16-byte-per-lane transfers a wave requests per slab
  static_assert(2 * NDMA <= 63, "vmcnt is a 6-bit counter");
  constexpr int NPB = F16 ? 2 : 3;              // planes behind the base pointers
  constexpr int BT = 64 * WT;
  constexpr int IMG = KS * 128;                 // bytes of one KS-pixel x 64-channel image
  constexpr int OPB = WT * NPL * IMG;           // one operand of one stage: [image][plane]
  constexpr int STAGE = 2 * OPB;                // dY images, then x images
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * BT;
  const int tap = blockIdx.y / p.ctiles, ci0 = (blockIdx.y - tap * p.ctiles) * BT;
  const int ky = tap / p.KW, kx = tap - ky * p.KW;
  const int m_begin = blockIdx.z * p.m_per_split, m_end = min(p.M, m_begin + p.m_per_split);

  // ---- transfers: this lane's slot is (row 8 wave + lane/8, chunk' lane%8) of every image
  const int lrow = (KS == 32 ? 8 * wave : 8 * (wave & 1)) + (lane >> 3);
  const bool move_d = KS == 32 || wave < 2, move_x = KS == 32 || wave >= 2;   // which operand this wave transfers
  const int ch = (lane & 7) ^ (((lrow >> 1) & 1) << 2);   // the chunk (8-channel group) that lands in this slot
  int m = m_begin + lrow;
  int b, oy, ox;
  {
    const int HoWo = p.Ho * p.Wo;
    b = m / HoWo;
    const int rem = m - b * HoWo;
    oy = rem / p.Wo;
    ox = rem - oy * p.Wo;
  }
  const unsigned dgroup = (unsigned)p.M * 16u, xgroup = (unsigned)p.xnpix * 16u;   // bytes per 8-channel group
  bool dok[WT], xok[WT];
#pragma unroll
  for (int i = 0; i < WT; ++i) {
    dok[i] = n0 / 8 + 8 * i + ch < p.Cout / 8;
    xok[i] = ci0 / 8 + 8 * i + ch < p.cs / 8;
  }
  const i32x4 rsd = raw_rsrc(p.dy3, (unsigned)(NPB * p.dps * 2));
  const i32x4 rsx = raw_rsrc(p.x3, (unsigned)(NPB * p.xps * 2));
  const unsigned dplane = (unsigned)(p.dps * 2), xplane = (unsigned)(p.xps * 2);
  const unsigned d_base = (unsigned)(n0 / 8) * dgroup, x_base = (unsigned)(ci0 / 8) * xgroup;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const unsigned lds_w = lds0 + (unsigned)(KS == 32 ? wave : (wave & 1)) * 1024u;   // this wave's 8 rows of an image
  auto issue = [&](int stage) {
    const bool mok = m < m_end;
    const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
    const bool pok = mok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const unsigned dv = (unsigned)ch * dgroup + (unsigned)m * 16u;
    const unsigned xv = (unsigned)ch * xgroup + (unsigned)((long long)b * p.xbs_pix + (long long)iy * p.W + ix) * 16u;
    const unsigned sd = lds_w + (unsigned)stage * STAGE, sx = sd + OPB;
    if (move_d) {
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          dma16(rsd, sd + (i * NPL + pl) * IMG, (mok && dok[i]) ? dv : WG_OOB, d_base + i * 8 * dgroup + pl * dplane);
    }
    if (move_x) {
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
          dma16(rsx, sx + (i * NPL + pl) * IMG, (pok && xok[i]) ? xv : WG_OOB, x_base + i * 8 * xgroup + pl * xplane);
    }
  };
  auto advance = [&]() {
    m += KS;
    ox += KS;
    while (ox >= p.Wo) {
      ox -= p.Wo;
      ++oy;
    }
    while (oy >= p.Ho) {
      oy -= p.Ho;
      ++b;
    }
  };

  // ---- fragment reads: 16-lane group g = lane/16 -> (k half h = g/2, channel half gg = g%2); lane 4q + pp of the group
  // supplies pixel row 8h + 4t + q, bytes 8 pp .. 8 pp + 7 of the group's 32-byte run (read t = 0, 1; 16 pixels per k-step)
  const int gg = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
  const unsigned swz = (unsigned)((q >> 1) & 1) << 2;
  unsigned fr[2];   // byte offset inside an image for sub-tile (32 channels) 0 / 1
#pragma unroll
  for (int j = 0; j < 2; ++j)
    fr[j] = (unsigned)(8 * h + q) * 128u + ((((unsigned)(4 * j) ^ swz) + 2u * gg + (pp >> 1)) * 16u) + 8u * (pp & 1);
  // this wave's sub-tiles: filters 32 WT wn + 32 i, channels 32 WT wc + 32 j
  const unsigned dimg = lds0 + (WT == 2 ? wn : 0) * (NPL * IMG), ximg = lds0 + OPB + (WT == 2 ? wc : 0) * (NPL * IMG);

  f32x16 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // wait until at most `c` of this wave's slabs (NDMA transfers each) are still in flight
  auto wait_slabs = [&](int c) __attribute__((always_inline)) {
    if (c <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (c == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
  };
  const int nslab = (m_end - m_begin + KS - 1) / KS;
  issue(0);
  int issued = 1;
#pragma unroll
  for (int d = 1; d < NST - 1; ++d)
    if (d < nslab) {
      advance();
      issue(d);
      ++issued;
    }
  wait_slabs(issued - 1);   // slab 0 has landed; the younger ones may stay in flight
  __builtin_amdgcn_s_barrier();
  int st = 0;
  for (int sl = 0; sl < nslab; ++sl) {
    if (sl + NST - 1 < nslab) {   // slab sl + NST - 1 into the stage the previous hand-over released
      advance();
      issue(st == 0 ? NST - 1 : st - 1);
    }
    const unsigned so = (unsigned)st * STAGE;
#pragma unroll
    for (int ks = 0; ks < KS / 16; ++ks) {
      wg_s16x8 a[NPL][WT], bb[NPL][WT];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
        for (int i = 0; i < WT; ++i) {
          const unsigned ad = dimg + so + pl * IMG + ks * 2048 + fr[WT == 2 ? i : wn];
          a[pl][i] = __builtin_shufflevector(lds_tr16(ad), lds_tr16(ad + 512), 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int j = 0; j < WT; ++j) {
          const unsigned ad = ximg + so + pl * IMG + ks * 2048 + fr[WT == 2 ? j : wc];
          bb[pl][j] = __builtin_shufflevector(lds_tr16(ad), lds_tr16(ad + 512), 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) {
          f32x16 c = acc[i][j];
          if constexpr (NPL == 3) {
            c = wg_mfma<F16>(a[0][i], bb[2][j], c);
            c = wg_mfma<F16>(a[2][i], bb[0][j], c);
            c = wg_mfma<F16>(a[1][i], bb[1][j], c);
          }
          if constexpr (NPL >= 2) {
            c = wg_mfma<F16>(a[0][i], bb[1][j], c);
            c = wg_mfma<F16>(a[1][i], bb[0][j], c);
          }
          c = wg_mfma<F16>(a[0][i], bb[0][j], c);
          acc[i][j] = c;
        }
    }
    // hand-over: slab sl + 1 has landed (this wave's share; the slabs behind it, up to sl + NST - 1, may stay in flight), this
    // wave's fragment reads of stage st are done; the barrier publishes the one and frees the other
    const int last = min(sl + NST - 1, nslab - 1);   // youngest slab requested so far
    wait_slabs(last - (sl + 1));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    st = st == NST - 1 ? 0 : st + 1;
  }
  float *dst = p.partial + (long long)blockIdx.z * p.Cout * p.K + (long long)tap * p.Cin + p.c_off;
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int c = ci0 + wc * 32 * WT + 32 * j + r;
      if (c >= p.cs) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + wn * 32 * WT + 32 * i + acc_row(e, h);
        if (n < p.Cout) dst[(long long)n * p.K + c] = acc[i][j][e];
      }
    }
}

// sum the pixel slices in order, write OIHW (the reference's parameter layout), optionally accumulate.  Block = one
// filter x 64 input channels: the partials are read along ci (coalesced) for every tap, transposed through LDS, and
// stored as one contiguous run of 64 * taps floats of dw.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float *__restrict__ partial,
                                                                float *__restrict__ dw, int Cout, int Cin, int taps,
                                                                int Cin_store, int zsplit, int accumulate, int cpb,
                                                                const float *__restrict__ factor) {
  // (factor: one float in device memory the sums are multiplied by -- the inverse of dY's power-of-two plane scale; NULL = 1)
  const float fac = factor ? *factor : 1.f;
  extern __shared__ float red_s[];   // [cpb][taps]; cpb = 64 or 16 input channels per block (16: enough blocks for small layers)
  const int n = blockIdx.y, c0 = blockIdx.x * cpb;
  const int nc = min(cpb, Cin_store - c0);
  const long long K = (long long)taps * Cin, zs = (long long)Cout * K;
  const float *src = partial + (long long)n * K + c0;
  for (int idx = threadIdx.x; idx < cpb * taps; idx += 256) {
    const int tap = idx / cpb, cl = idx - tap * cpb;
    float s = 0.f;
    if (cl < nc)
      for (int z = 0; z < zsplit; ++z) s += src[z * zs + (long long)tap * Cin + cl];
    red_s[cl * taps + tap] = s * fac;
  }
  __syncthreads();
  float *o = dw + ((long long)n * Cin_store + c0) * taps;
  for (int idx = threadIdx.x; idx < nc * taps; idx += 256) o[idx] = accumulate ? o[idx] + red_s[idx] : red_s[idx];
}
static inline void launch_wgrad_reduce(hipStream_t st, const float *partial, float *dw, int Cout, int Cin, int taps,
                                       int cin_store, int zsplit, int accumulate, const float *factor = nullptr) {
  const int cpb = (long long)cdiv(cin_store, 64) * Cout < 1024 ? 16 : 64;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(cdiv(cin_store, cpb), Cout), dim3(256), cpb * taps * sizeof(float), st,
                     partial, dw, Cout, Cin, taps, cin_store, zsplit, accumulate, cpb, factor);
}

// per-column sums of a [M][C] matrix (optionally of the product with a second one): stage 1 = one partial row per
// block row, stage 2 = fixed-order sum.  Serves the bias / frozen-BatchNorm parameter gradients.
static inline int cs_rows(long long M, int C) {  // rows per stage-1 block: enough blocks to fill the chip, <= 512
  const long long cb = cdiv(C / 4, 16);
  long long rows = (M * cb + 1023) / 1024;
  rows = (rows + 15) / 16 * 16;
  return (int)(rows < 32 ? 32 : (rows > 512 ? 512 : rows));
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                             float *__restrict__ part, long long M, int C, int CS_ROWS) {
  __shared__ float4 sh1[256], sh2[256];
  const int cq = C / 4;
  const int c4 = blockIdx.x * 16 + (threadIdx.x & 15);
  const int rsub = threadIdx.x >> 4;  // 16 row lanes
  const long long m0 = (long long)blockIdx.y * CS_ROWS, m1 = min(M, m0 + CS_ROWS);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (c4 < cq)
    for (long long m = m0 + rsub; m < m1; m += 16) {
      const float4 va = ld4g(a + m * C + c4 * 4);
      s1.x += va.x; s1.y += va.y; s1.z += va.z; s1.w += va.w;
      if (b) {
        const float4 vb = ld4g(b + m * C + c4 * 4);
        s2.x += va.x * vb.x; s2.y += va.y * vb.y; s2.z += va.z * vb.z; s2.w += va.w * vb.w;
      }
    }
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
    *reinterpret_cast<float4 *>(dst) = s1;
    *reinterpret_cast<float4 *>(dst + C) = s2;
  }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ part, float *__restrict__ out1,
                                                           float *__restrict__ out2, int nrow, int C,
                                                           int accumulate) {
  // block = 16 columns x 16 row lanes; lane r sums rows r, r+16, ... in order, the lanes are combined in order
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
  if (out1) out1[c] = accumulate ? out1[c] + s1 : s1;
  if (out2) out2[c] = accumulate ? out2[c] + s2 : s2;
}

// y[i] (+)= sum_b x[b][i]: gradient of a tensor that was broadcast over the objects of a frame
__global__ __launch_bounds__(256) void sum_batch_kernel(const float *__restrict__ x, float *__restrict__ y, int B,
                                                        long long n4, int accumulate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 s = ld4g(x + i * 4);
  for (int b = 1; b < B; ++b) {
    const float4 v = ld4g(x + ((long long)b * n4 + i) * 4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (accumulate) {
    const float4 o = ld4g(y + i * 4);
    s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
  }
  *reinterpret_cast<float4 *>(y + i * 4) = s;
}

// Clip-batched training (train.py, round 6): a map that the N objects of EACH of G clips share -- the key encoder's 1/16 features
// in the value encoder's fuser, the query values in the fusion layer, the decoder's skip features -- is laid out once per object
// (y[(g N + j)][i] = x[g][i]) so that the clips' objects go through the convolutions as ONE batch; the gradient sums the N copies.
__global__ __launch_bounds__(256) void expand_groups_kernel(const float *__restrict__ x, float *__restrict__ y, int N, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int g = blockIdx.y;
  const float4 v = ld4g(x + ((long long)g * n4 + i) * 4);
  for (int j = 0; j < N; ++j) *reinterpret_cast<float4 *>(y + (((long long)g * N + j) * n4 + i) * 4) = v;
}
__global__ __launch_bounds__(256) void sum_groups_kernel(const float *__restrict__ x, float *__restrict__ y, int N, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int g = blockIdx.y;
  float4 s = ld4g(x + ((long long)g * N * n4 + i) * 4);
  for (int j = 1; j < N; ++j) {               // (fixed order: deterministic, and the order sum_batch_kernel uses for one clip)
    const float4 v = ld4g(x + (((long long)g * N + j) * n4 + i) * 4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  *reinterpret_cast<float4 *>(y + ((long long)g * n4 + i) * 4) = s;
}

struct WgradPlan {
  int wt, zsplit, m_per_split;
};
// Tile and pixel slices (measured with tools/wgrad_bench.py --tune on the training shapes): the 128x128 tile where the
// grid is large enough without many slices (>= 128 tiles, or >= 4096 pixels to cut), else 64x64; slices so that
// tiles x slices is about two blocks per CU, at most 16 (every slice is a full partial gradient: written by the GEMM
// and read by the reduce) and at least 128 pixels each.
static WgradPlan wgrad_pick(long long M, int Cout, int KH, int KW, int c0, int c1, int c2, int plan, int min_tiles2 = 128) {
  WgradPlan pl;
  auto tiles_of = [&](int bt) {
    return (long long)cdiv(Cout, bt) * KH * KW * (cdiv(c0, bt) + (c1 ? cdiv(c1, bt) : 0) + (c2 ? cdiv(c2, bt) : 0));
  };
  const bool big = Cout >= 128 && c0 >= 128 && (c1 == 0 || c1 >= 128) && (c2 == 0 || c2 >= 128);
  pl.wt = (plan & 15) ? (plan & 15) : ((big && (tiles_of(128) >= min_tiles2 || M >= 4096)) ? 2 : 1);
  if (pl.wt != 1 && pl.wt != 2) pl.wt = 1;
  const long long tiles = tiles_of(64 * pl.wt);
  const int zforce = (plan >> 4) & 255;
  // (~512 blocks per launch: measured in the four-lane step, round 6 -- 128 / 256 / 512 / 1024 blocks: 89.1 / 94.4 / 95.5 / 92.1 clips/s;
  // fewer slices mean less partial-sum traffic for the reduce but the other lanes do not fill the gap)
  long long zs = zforce > 0 ? zforce : (512 + tiles - 1) / (tiles > 0 ? tiles : 1);
  const long long zmax = (M + 127) / 128;
  // pixel slices: enough blocks to fill the chip (~512); at most 16 -- or 64 where the partial sums stay small (<= 8 MB: the
  // 64-channel and 1x1 layers, whose 4-36 tiles left three quarters of the CUs idle at 16 slices: tools/wgrad_bench.py --tune,
  // round 5: 57.6 -> 33.9 us on 3x96x96 1x1 64->256)
  const long long by_bytes = (long long)(8.0 * 1024 * 1024 / ((double)Cout * KH * KW * (c0 + c1 + c2) * 4.0));
  const long long zcap = by_bytes > 64 ? 64 : (by_bytes < 16 ? 16 : by_bytes);
  if (zs > zcap && zforce == 0) zs = zcap;
  if (zs > zmax) zs = zmax;
  if (zs < 1) zs = 1;
  long long per = ((M + zs - 1) / zs + 31) / 32 * 32;
  pl.m_per_split = (int)per;
  pl.zsplit = (int)((M + per - 1) / per);
  return pl;
}
WgradPlan wgrad_plan(long long M, int Cout, int KH, int KW, int c0, int c1, int c2) {
  return wgrad_pick(M, Cout, KH, KW, c0, c1, c2, 0);
}

}  // namespace

extern "C" size_t swem_conv2d_wgrad_workspace(int B, int H, int W, int c0, int c1, int c2, int Cout, int KH, int KW,
                                              int stride, int pad) {
  if (stride <= 0) return 0;
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  const long long M = (long long)B * Ho * Wo;
  WgradPlan pl = wgrad_plan(M, Cout, KH, KW, c0, c1, c2);
  return (size_t)pl.zsplit * Cout * KH * KW * (c0 + c1 + c2) * sizeof(float);
}

extern "C" int swem_conv2d_wgrad_f32(void *stream, const float *dy, const float *x0, int c0, long long bs0,
                                     const float *x1, int c1, long long bs1, const float *x2, int c2, long long bs2,
                                     int B, int H, int W, int Cout, int KH, int KW, int stride, int pad, int relu_in,
                                     float *dw, int cin_store, int accumulate, void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(dy && x0 && dw, SWEM_E_ARG, "conv2d_wgrad: null pointer");
  SWEM_REQUIRE((c1 == 0 || x1) && (c2 == 0 || x2) && c0 > 0 && c0 % 4 == 0 && c1 % 4 == 0 && c2 % 4 == 0 &&
                   Cout % 4 == 0 && Cout > 0,
               SWEM_E_SHAPE, "conv2d_wgrad: channel counts must be multiples of 4");
  SWEM_REQUIRE(B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, SWEM_E_SHAPE,
               "conv2d_wgrad: bad geometry");
  const int Cin = c0 + c1 + c2;
  SWEM_REQUIRE(cin_store > 0 && cin_store <= Cin, SWEM_E_SHAPE, "conv2d_wgrad: cin_store out of range");
  WgradP p;
  p.dy = dy;
  p.B = B; p.H = H; p.W = W;
  p.Ho = (H + 2 * pad - KH) / stride + 1;
  p.Wo = (W + 2 * pad - KW) / stride + 1;
  SWEM_REQUIRE(p.Ho > 0 && p.Wo > 0, SWEM_E_SHAPE, "conv2d_wgrad: empty output");
  const long long M = (long long)B * p.Ho * p.Wo;
  SWEM_REQUIRE(M < (1ll << 31), SWEM_E_SHAPE, "conv2d_wgrad: too large");
  p.M = (int)M;
  p.Cout = Cout; p.Cin = Cin; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.relu_in = relu_in;
  p.K = KH * KW * Cin;
  WgradPlan pl = wgrad_plan(M, Cout, KH, KW, c0, c1, c2);
  const size_t need = (size_t)pl.zsplit * Cout * p.K * sizeof(float);
  SWEM_REQUIRE(ws && ws_bytes >= need, SWEM_E_WORKSPACE, "conv2d_wgrad: workspace %zu < %zu bytes", ws_bytes, need);
  p.partial = static_cast<float *>(ws);
  p.m_per_split = pl.m_per_split;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float *xs[3] = {x0, x1, x2};
  const int cs[3] = {c0, c1, c2};
  const long long bss[3] = {bs0, bs1, bs2};
  const int bt = 64 * pl.wt;
  int off = 0;
  for (int s = 0; s < 3; ++s) {
    if (cs[s] == 0) continue;
    p.x = xs[s]; p.cs = cs[s]; p.bs = bss[s]; p.c_off = off;
    p.ctiles = cdiv(cs[s], bt);
    dim3 grid(cdiv(Cout, bt), KH * KW * p.ctiles, pl.zsplit);
    if (pl.wt == 2) hipLaunchKernelGGL(conv_wgrad_kernel<2>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(256), 0, st, p);
    SWEM_CHECK_LAUNCH("conv_wgrad_kernel");
    off += cs[s];
  }
  launch_wgrad_reduce(st, p.partial, dw, Cout, Cin, KH * KW, cin_store, pl.zsplit, accumulate);
  SWEM_CHECK_LAUNCH("conv_wgrad_reduce_kernel");
  return SWEM_OK;
}

// (three stages by default where they cost no resident block: 3 x stage <= 48 KB -> three blocks per CU as before)
#define WGRAD_DEFAULT_NST(stage_bytes) ((3 * (stage_bytes) <= 48 * 1024) ? 3 : 2)

// ---- bf16-pipe weight gradient (pre-split planes)
// (the six-product mode takes the 128x128 tile from 64 tiles on: with the 16-pixel slab it keeps three blocks per CU)
static WgradPlan wgrad_bf_plan(long long M, int Cout, int KH, int KW, int c0, int c1, int c2, int plan, int math) {
  return wgrad_pick(M, Cout, KH, KW, c0, c1, c2, plan, math == 1 ? 64 : 128);
}

template <int WT, int NPL, int KS, bool F16, int NST>
static int launch_wgrad_bf_n(const WgradBP &p, dim3 grid, hipStream_t st) {
  constexpr size_t lds = (size_t)NST * 2 * WT * NPL * KS * 128;
  SWEM_ALLOW_LDS((conv_wgrad_bf_kernel<WT, NPL, KS, F16, NST>), lds);
  hipLaunchKernelGGL((conv_wgrad_bf_kernel<WT, NPL, KS, F16, NST>), grid, dim3(256), lds, st, p);
  return SWEM_OK;
}
// nst: plan bits 13-14 (0 = the default below; 2, 3: forced -- tools/wgrad_bench.py --nst).  Default (round 6, measured on the
// training shapes, profiles/r06_wgrad_nst.txt): three stages for the 64x64 tile -- the small layers, whose few blocks per CU hide
// nothing: 33.4 -> 27.9 us on 3x96x96 1x1 64->256, 28.7 -> 23.4 on 3x48x48 1x1 128->512 -- and two for the 128x128 tile, whose three
// resident blocks per CU already cover the round trip (three stages there: 1-4 % slower, the LDS is the co-limiter).
template <int WT, int NPL, int KS, bool F16 = false>
static int launch_wgrad_bf(const WgradBP &p, dim3 grid, hipStream_t st, int nst = 0) {
  constexpr size_t stage = 2 * WT * NPL * KS * 128;
  if (nst == 0) nst = WT == 1 ? WGRAD_DEFAULT_NST(stage) : 2;
  if (nst >= 3 && 3 * stage <= 160 * 1024) return launch_wgrad_bf_n<WT, NPL, KS, F16, 3>(p, grid, st);
  return launch_wgrad_bf_n<WT, NPL, KS, F16, 2>(p, grid, st);
}

extern "C" size_t swem_conv2d_wgrad_bf16x3_workspace(int B, int H, int W, int c0, int c1, int c2, int Cout, int KH,
                                                     int KW, int stride, int pad, int plan) {
  if (stride <= 0) return 0;
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  const long long M = (long long)B * Ho * Wo;
  const int z1 = wgrad_bf_plan(M, Cout, KH, KW, c0, c1, c2, plan, 1).zsplit;   // either math mode may follow
  const int z2 = wgrad_bf_plan(M, Cout, KH, KW, c0, c1, c2, plan, 2).zsplit;
  return (size_t)(z1 > z2 ? z1 : z2) * Cout * KH * KW * (c0 + c1 + c2) * sizeof(float);
}

// math: 1 = bf16x6, 2 = plain bf16, 3 = f16x3 (fp16 pairs; `factor` = device float the result is multiplied by, or NULL)
static int wgrad_planes_impl(void *stream, const unsigned short *dy3, long long dy_ps,
                             const unsigned short *x0, int c0, long long bs0, long long ps0,
                             const unsigned short *x1, int c1, long long bs1, long long ps1,
                             const unsigned short *x2, int c2, long long bs2, long long ps2, int B, int H,
                             int W, int Cout, int KH, int KW, int stride, int pad, int math, float *dw,
                             int cin_store, int accumulate, int plan, void *ws, size_t ws_bytes, const float *factor) {
  SWEM_REQUIRE(dy3 && x0 && dw, SWEM_E_ARG, "conv2d_wgrad_bf16x3: null pointer");
  SWEM_REQUIRE((c1 == 0 || x1) && (c2 == 0 || x2) && c0 > 0 && c0 % 8 == 0 && c1 % 8 == 0 && c2 % 8 == 0 &&
                   Cout % 8 == 0 && Cout > 0,
               SWEM_E_SHAPE, "conv2d_wgrad_bf16x3: channel counts must be multiples of 8");
  SWEM_REQUIRE(B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, SWEM_E_SHAPE,
               "conv2d_wgrad_bf16x3: bad geometry");
  SWEM_REQUIRE(math >= 1 && math <= 3, SWEM_E_ARG, "conv2d_wgrad_bf16x3: math must be 1 (bf16x6) or 2 (bf16)");
  const int Cin = c0 + c1 + c2;
  SWEM_REQUIRE(cin_store > 0 && cin_store <= Cin, SWEM_E_SHAPE, "conv2d_wgrad_bf16x3: cin_store out of range");
  WgradBP p;
  p.dy3 = dy3;
  p.dps = dy_ps;
  p.B = B; p.H = H; p.W = W;
  p.Ho = (H + 2 * pad - KH) / stride + 1;
  p.Wo = (W + 2 * pad - KW) / stride + 1;
  SWEM_REQUIRE(p.Ho > 0 && p.Wo > 0, SWEM_E_SHAPE, "conv2d_wgrad_bf16x3: empty output");
  const long long M = (long long)B * p.Ho * p.Wo;
  SWEM_REQUIRE(M * 128 < (1ll << 32) && dy_ps == M * Cout, SWEM_E_SHAPE,
               "conv2d_wgrad_bf16x3: dY planes must be [Cout/8][B*Ho*Wo][8] and below 4 GiB per 64 channels");
  p.M = (int)M;
  p.Cout = Cout; p.Cin = Cin; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
  p.K = KH * KW * Cin;
  WgradPlan pl = wgrad_bf_plan(M, Cout, KH, KW, c0, c1, c2, plan, math);
  const size_t need = (size_t)pl.zsplit * Cout * p.K * sizeof(float);
  SWEM_REQUIRE(ws && ws_bytes >= need, SWEM_E_WORKSPACE, "conv2d_wgrad_bf16x3: workspace %zu < %zu bytes", ws_bytes,
               need);
  p.partial = static_cast<float *>(ws);
  p.m_per_split = pl.m_per_split;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned short *xs[3] = {x0, x1, x2};
  const int cs[3] = {c0, c1, c2};
  const long long bss[3] = {bs0, bs1, bs2}, pss[3] = {ps0, ps1, ps2};
  const int bt = 64 * pl.wt;
  int off = 0;
  for (int s = 0; s < 3; ++s) {
    if (cs[s] == 0) continue;
    SWEM_REQUIRE(pss[s] % cs[s] == 0 && bss[s] % cs[s] == 0 && pss[s] / cs[s] * 128 < (1ll << 32), SWEM_E_SHAPE,
                 "conv2d_wgrad_bf16x3: source %d plane / batch stride", s);
    p.x3 = xs[s]; p.cs = cs[s]; p.c_off = off; p.xps = pss[s];
    p.xnpix = (int)(pss[s] / cs[s]);
    p.xbs_pix = bss[s] / cs[s];
    p.ctiles = cdiv(cs[s], bt);
    dim3 grid(cdiv(Cout, bt), KH * KW * p.ctiles, pl.zsplit);
    int rc;
    // slab: 16 pixels for the six-product 128x128 tile (48 KB of LDS instead of 96: three blocks per CU), else 32;
    // plan bit 12 flips the choice (tools/wgrad_bench.py)
    const bool ks16 = ((pl.wt == 2 && (math == 1 || math == 3)) != (((plan >> 12) & 1) != 0));   // (f16x3: measured like bf16x6)
    const int nst = (plan >> 13) & 3;
    if (math == 3) {
      if (pl.wt == 2) rc = ks16 ? launch_wgrad_bf<2, 2, 16, true>(p, grid, st, nst) : launch_wgrad_bf<2, 2, 32, true>(p, grid, st, nst);
      else rc = ks16 ? launch_wgrad_bf<1, 2, 16, true>(p, grid, st, nst) : launch_wgrad_bf<1, 2, 32, true>(p, grid, st, nst);
    } else if (pl.wt == 2) {
      if (math == 1) rc = ks16 ? launch_wgrad_bf<2, 3, 16>(p, grid, st, nst) : launch_wgrad_bf<2, 3, 32>(p, grid, st, nst);
      else rc = ks16 ? launch_wgrad_bf<2, 1, 16>(p, grid, st, nst) : launch_wgrad_bf<2, 1, 32>(p, grid, st, nst);
    } else {
      if (math == 1) rc = ks16 ? launch_wgrad_bf<1, 3, 16>(p, grid, st, nst) : launch_wgrad_bf<1, 3, 32>(p, grid, st, nst);
      else rc = ks16 ? launch_wgrad_bf<1, 1, 16>(p, grid, st, nst) : launch_wgrad_bf<1, 1, 32>(p, grid, st, nst);
    }
    if (rc != SWEM_OK) return rc;
    SWEM_CHECK_LAUNCH("conv_wgrad_bf_kernel");
    off += cs[s];
  }
  launch_wgrad_reduce(st, p.partial, dw, Cout, Cin, KH * KW, cin_store, pl.zsplit, accumulate, factor);
  SWEM_CHECK_LAUNCH("conv_wgrad_reduce_kernel");
  return SWEM_OK;
}

extern "C" int swem_conv2d_wgrad_bf16x3(void *stream, const unsigned short *dy3, long long dy_ps,
                                        const unsigned short *x0, int c0, long long bs0, long long ps0,
                                        const unsigned short *x1, int c1, long long bs1, long long ps1,
                                        const unsigned short *x2, int c2, long long bs2, long long ps2, int B, int H,
                                        int W, int Cout, int KH, int KW, int stride, int pad, int math, float *dw,
                                        int cin_store, int accumulate, int plan, void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(math == 1 || math == 2, SWEM_E_ARG, "conv2d_wgrad_bf16x3: math must be 1 (bf16x6) or 2 (bf16)");
  return wgrad_planes_impl(stream, dy3, dy_ps, x0, c0, bs0, ps0, x1, c1, bs1, ps1, x2, c2, bs2, ps2, B, H, W, Cout, KH, KW, stride,
                           pad, math, dw, cin_store, accumulate, plan, ws, ws_bytes, nullptr);
}

extern "C" int swem_conv2d_wgrad_f16x3(void *stream, const void *dy2, long long dy_ps, const void *x0, int c0, long long bs0,
                                       long long ps0, const void *x1, int c1, long long bs1, long long ps1, const void *x2,
                                       int c2, long long bs2, long long ps2, int B, int H, int W, int Cout, int KH, int KW,
                                       int stride, int pad, const float *dy_inv_scale, float *dw, int cin_store,
                                       int accumulate, int plan, void *ws, size_t ws_bytes) {
  return wgrad_planes_impl(stream, static_cast<const unsigned short *>(dy2), dy_ps, static_cast<const unsigned short *>(x0), c0,
                           bs0, ps0, static_cast<const unsigned short *>(x1), c1, bs1, ps1,
                           static_cast<const unsigned short *>(x2), c2, bs2, ps2, B, H, W, Cout, KH, KW, stride, pad, 3, dw,
                           cin_store, accumulate, plan, ws, ws_bytes, dy_inv_scale);
}

// ---- the fp16 (hi, mid) pair of a GRADIENT map, scaled by a power of two chosen from its largest magnitude
// Gradients span too many binades for an unscaled fp16 pair (the pair carries 22-23 bits only where |x| >= 2^-2, and nothing above
// 65504).  Pass 1: SWEM_AMAX_PARTS block maxima of |x| (no atomics, nothing to zero: the scratch may be uninitialised memory).
// Pass 2: every block reduces the maxima itself, scales by 2^s with s = 13 - floor(log2 amax) -- the largest element lands in
// [2^13, 2^14), every element within 2^-15 of it keeps >= 22 bits, smaller ones an absolute error <= 2^-39 of the maximum -- and
// splits; block 0 stores 2^-s for the consumers (the data-gradient convolution's epilogue scale, the weight gradient's reduce).
// Exact (powers of two), deterministic, graph-safe (no host decision).  A non-finite maximum sets SWEM_FAULT_RANGE (scale 1).
namespace {
__global__ __launch_bounds__(256) void amax_partials_kernel(const float4 *__restrict__ x, long long n4, float *__restrict__ part) {
  unsigned m = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = x[i];
    m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu, __float_as_uint(v.z) & 0x7fffffffu));
    m = max(m, __float_as_uint(v.w) & 0x7fffffffu);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  __shared__ unsigned sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = __uint_as_float(max(max(sh[0], sh[1]), max(sh[2], sh[3])));
}
__device__ __forceinline__ int grad_scale_exp(const float *part, int nparts, unsigned *fault) {
  // (block-wide: every thread returns s)
  unsigned m = 0u;
  for (int i = threadIdx.x; i < nparts; i += 256) m = max(m, __float_as_uint(part[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  __shared__ unsigned sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  m = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
  if (m >= 0x7f800000u) {   // inf / NaN in the map: the planes carry it on (as the fp32 arithmetic would); reported
    if (fault && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0)
      __hip_atomic_fetch_or(fault, (unsigned)SWEM_FAULT_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0;
  }
  if (m == 0u) return 0;
  const int s = 140 - (int)(m >> 23);   // 13 - (biased exponent - 127)
  return s < -126 ? -126 : (s > 126 ? 126 : s);
}
__global__ __launch_bounds__(256) void split_f16x2_scaled_kernel(const float *__restrict__ x, unsigned short *__restrict__ out,
                                                                 long long npix, int C, const float *__restrict__ part,
                                                                 int nparts, float *__restrict__ inv_out, unsigned *fault) {
  const int s = grad_scale_exp(part, nparts, fault);
  const float scale = __uint_as_float((unsigned)(s + 127) << 23);
  if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) *inv_out = __uint_as_float((unsigned)(127 - s) << 23);
  const long long pix = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int cg = blockIdx.y * 8 + (threadIdx.x & 7);
  if (pix >= npix || cg >= C / 8) return;
  const float *src = x + pix * C + cg * 8;
  float4 v0 = *reinterpret_cast<const float4 *>(src), v1 = *reinterpret_cast<const float4 *>(src + 4);
  v0 = make_float4(v0.x * scale, v0.y * scale, v0.z * scale, v0.w * scale);
  v1 = make_float4(v1.x * scale, v1.y * scale, v1.z * scale, v1.w * scale);
  uint2 h0, m0, h1, m1;
  split2h(v0, h0, m0);
  split2h(v1, h1, m1);
  const long long plane = npix * C, i = (long long)cg * npix + pix;
  *reinterpret_cast<uint4 *>(out + i * 8) = make_uint4(h0.x, h0.y, h1.x, h1.y);
  *reinterpret_cast<uint4 *>(out + plane + i * 8) = make_uint4(m0.x, m0.y, m1.x, m1.y);
}
__global__ __launch_bounds__(256) void vec_scale_kernel(const float *__restrict__ in, const float *__restrict__ factor,
                                                        float *__restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (in ? in[i] : 1.f) * *factor;
}
}  // namespace

extern "C" int swem_split_f16x2_scaled_f32(void *stream, const float *x, void *out, long long npix, int C, float *scratch,
                                           int nparts, void *fault) {
  SWEM_REQUIRE(x && out && scratch && npix > 0 && C > 0 && C % 8 == 0 && nparts >= 0, SWEM_E_ARG,
               "split_f16x2_scaled: need C %% 8 == 0");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nparts == 0) {   // no producer-made maxima: the first pass here
    nparts = SWEM_AMAX_PARTS;
    hipLaunchKernelGGL(amax_partials_kernel, dim3(SWEM_AMAX_PARTS), dim3(256), 0, st, reinterpret_cast<const float4 *>(x),
                       npix * C / 4, scratch + 1);
    SWEM_CHECK_LAUNCH("amax_partials_kernel");
  }
  hipLaunchKernelGGL(split_f16x2_scaled_kernel, dim3((unsigned)cdiv(npix, 32), (unsigned)cdiv(C / 8, 8)), dim3(256), 0, st, x,
                     static_cast<unsigned short *>(out), npix, C, scratch + 1, nparts, scratch, static_cast<unsigned *>(fault));
  SWEM_CHECK_LAUNCH("split_f16x2_scaled_kernel");
  return SWEM_OK;
}

// filters [Cout][K] -> the fp16 pair planes [2][K/8][Cout][8] of the f16x3 arithmetic, every filter (output column) scaled by its
// own power of two so that its largest weight lies in [2^13, 2^14) (swem_hip.h, "f16x3"), and the epilogue scale that undoes it:
// scale_out[n] = (scale_in ? scale_in[n] : 1) * 2^-e[n].  One launch per pack (the training step re-packs every filter every step).
namespace {
__global__ __launch_bounds__(256) void pack_filters_f16x2_kernel(const float *__restrict__ w, unsigned short *__restrict__ out,
                                                                 int Cout, int K, const float *__restrict__ scale_in,
                                                                 float *__restrict__ scale_out) {
  const int n = blockIdx.x;
  const float *row = w + (long long)n * K;
  unsigned m = 0;
  for (int k = threadIdx.x * 4; k < K; k += 1024) {
    const float4 v = *reinterpret_cast<const float4 *>(row + k);
    m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu, __float_as_uint(v.z) & 0x7fffffffu));
    m = max(m, __float_as_uint(v.w) & 0x7fffffffu);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  __shared__ unsigned sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  m = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
  int s = (m == 0u || m >= 0x7f800000u) ? 0 : 140 - (int)(m >> 23);
  s = s < -100 ? -100 : (s > 100 ? 100 : s);
  const float scale = __uint_as_float((unsigned)(s + 127) << 23);
  if (threadIdx.x == 0) scale_out[n] = (scale_in ? scale_in[n] : 1.f) * __uint_as_float((unsigned)(127 - s) << 23);
  const long long plane = (long long)Cout * K;
  for (int k = threadIdx.x * 8; k < K; k += 2048) {
    float4 v0 = *reinterpret_cast<const float4 *>(row + k), v1 = *reinterpret_cast<const float4 *>(row + k + 4);
    v0 = make_float4(v0.x * scale, v0.y * scale, v0.z * scale, v0.w * scale);
    v1 = make_float4(v1.x * scale, v1.y * scale, v1.z * scale, v1.w * scale);
    uint2 h0, m0, h1, m1;
    split2h(v0, h0, m0);
    split2h(v1, h1, m1);
    const long long i = (long long)(k >> 3) * Cout + n;
    *reinterpret_cast<uint4 *>(out + i * 8) = make_uint4(h0.x, h0.y, h1.x, h1.y);
    *reinterpret_cast<uint4 *>(out + plane + i * 8) = make_uint4(m0.x, m0.y, m1.x, m1.y);
  }
}
}  // namespace

extern "C" int swem_pack_filters_f16x2_f32(void *stream, const float *w, void *out, int Cout, int K, const float *scale_in,
                                           float *scale_out) {
  SWEM_REQUIRE(w && out && scale_out && Cout > 0 && K > 0 && K % 8 == 0, SWEM_E_ARG, "pack_filters_f16x2: need K %% 8 == 0");
  hipLaunchKernelGGL(pack_filters_f16x2_kernel, dim3(Cout), dim3(256), 0, static_cast<hipStream_t>(stream), w,
                     static_cast<unsigned short *>(out), Cout, K, scale_in, scale_out);
  SWEM_CHECK_LAUNCH("pack_filters_f16x2_kernel");
  return SWEM_OK;
}

extern "C" int swem_vec_scale_f32(void *stream, const float *in, const float *factor, float *out, int n) {
  SWEM_REQUIRE(factor && out && n > 0, SWEM_E_ARG, "vec_scale: bad argument");
  hipLaunchKernelGGL(vec_scale_kernel, dim3(cdiv(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), in, factor, out, n);
  SWEM_CHECK_LAUNCH("vec_scale_kernel");
  return SWEM_OK;
}

extern "C" size_t swem_colsum_workspace(long long M, int C) {
  if (M <= 0 || C <= 0) return 0;
  return (size_t)cdiv(M, cs_rows(M, C)) * 2 * C * sizeof(float);
}

extern "C" int swem_colsum_f32(void *stream, const float *a, const float *b, float *out1, float *out2, long long M,
                               int C, int accumulate, void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(a && (out1 || out2) && M > 0 && C > 0 && C % 4 == 0, SWEM_E_ARG, "colsum: bad argument");
  SWEM_REQUIRE(!out2 || b, SWEM_E_ARG, "colsum: the product sum needs the second matrix");
  const size_t need = swem_colsum_workspace(M, C);
  SWEM_REQUIRE(ws && ws_bytes >= need, SWEM_E_WORKSPACE, "colsum: workspace %zu < %zu bytes", ws_bytes, need);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rows = cs_rows(M, C), nrow = cdiv(M, rows);
  float *part = static_cast<float *>(ws);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(C / 4, 16), nrow), dim3(256), 0, st, a, b, part, M, C, rows);
  SWEM_CHECK_LAUNCH("colsum_partial_kernel");
  hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 16)), dim3(256), 0, st, part, out1, out2, nrow, C, accumulate);
  SWEM_CHECK_LAUNCH("colsum_final_kernel");
  return SWEM_OK;
}

extern "C" int swem_expand_groups_f32(void *stream, const float *x, float *y, int G, int N, long long n) {
  SWEM_REQUIRE(x && y && G > 0 && G < 65536 && N > 0 && n > 0 && n % 4 == 0, SWEM_E_ARG, "expand_groups: bad argument");
  hipLaunchKernelGGL(expand_groups_kernel, dim3(cdiv(n / 4, 256), G), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, N, n / 4);
  SWEM_CHECK_LAUNCH("expand_groups_kernel");
  return SWEM_OK;
}

extern "C" int swem_sum_groups_f32(void *stream, const float *x, float *y, int G, int N, long long n) {
  SWEM_REQUIRE(x && y && G > 0 && G < 65536 && N > 0 && n > 0 && n % 4 == 0, SWEM_E_ARG, "sum_groups: bad argument");
  hipLaunchKernelGGL(sum_groups_kernel, dim3(cdiv(n / 4, 256), G), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, N, n / 4);
  SWEM_CHECK_LAUNCH("sum_groups_kernel");
  return SWEM_OK;
}

extern "C" int swem_sum_batch_f32(void *stream, const float *x, float *y, int B, long long n, int accumulate) {
  SWEM_REQUIRE(x && y && B > 0 && n > 0 && n % 4 == 0, SWEM_E_ARG, "sum_batch: bad argument");
  hipLaunchKernelGGL(sum_batch_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, B,
                     n / 4, accumulate);
  SWEM_CHECK_LAUNCH("sum_batch_kernel");
  return SWEM_OK;
}
