// Implicit-GEMM convolution on the gfx950 fp32 matrix cores.
//
//   GEMM view:  Y[m][n] = sum_k A[m][k] * Wt[n][k]
//     m = output pixel (b, oy, ox)            M = B*Ho*Wo
//     n = output filter                       Ncols = Cout (2*Cout for the GLU pair)
//     k = (ky, kx, ci), ci fastest            K = KH*KW*Cin
//   A is never materialised: each thread gathers 16-byte channel groups of the NHWC input
//   (up to three concatenated sources, optional broadcast over the batch, optional input ReLU).
//
// Block = 256 threads = 4 waves in a 2x2 grid; wave tile = (32*WM) x (32*WN) outputs built from
// v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain).  K is walked in 32-wide blocks that
// are register-staged (global -> VGPR while the previous block is multiplied, then VGPR -> LDS)
// into a double-buffered LDS image  [k/4][row][4 floats], row stride padded by one 16-byte slot:
//   - the 8 lanes that write one row hit 8 different slots (stride BM+1 is odd),
//   - a ds_read_b128 of 16 consecutive rows is conflict free,
//   - one float4 per operand feeds 4 MFMAs (lane half h owns k = 4*(2j+h)..+3).
// The epilogue applies scale/shift (bias + folded frozen BatchNorm), the residual, ReLU or the GLU gate
// and stores NHWC: 32 consecutive channels per half wave = one 128-byte segment.
//
// Layers whose grid would not fill the 256 CUs split K over blockIdx.z; the partial sums go to a
// workspace and a second kernel reduces them in a fixed order (deterministic) and runs the epilogue.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "lds_dma.h"
#include "bf16_split.h"
#include <type_traits>

namespace {

constexpr int BK = 32;
constexpr int KQ = BK / 4;  // float4 slots per row per k-block

struct ConvP {
  const float *x[3];
  int c[3];
  long long bs[3];
  int B, H, W, Ho, Wo, Cin, K, M;
  const float *w, *scale, *shift, *res;
  float uscale;  // the scale of every output column where `scale` is NULL (1; matching's readout undoes its operand scaling here)
  long long res_bs, w_bs;  // w_bs: elements between the filter banks of consecutive batch items (0 = shared)
  const unsigned short *wsplit;  // optional [3][Ncols][K] bf16: the filters pre-split into hi/mid/lo planes
  const unsigned short *xs[3];   // pre-split activations (conv_igemm_bf3s_kernel): plane 0 of each source
  long long ps[3];               // ... and the element stride between a source's three planes
  int bsp[3], npx[3];            // pre-split sources: batch stride in pixels, pixels per plane (host-side divisions)
  float *y;
  int Cout, Ncols, KH, KW, stride, pad, flags;
  int nkb, kb_per_split;
  float *partial;
  int nplanes;  // 3 = bf16x6 (hi/mid/lo planes, six products), 2 = three products on (hi, mid), 1 = plain bf16 (hi plane only)
  int f16;      // nplanes == 2 only: the operand planes are fp16 pairs (bf16_split.h, split2h) -- the "f16x3" arithmetic
  int xpn;      // XCD partition of the N tiles (1, 2, 4, 8): see tile_coords
  int mt0;      // first M tile of this launch (the tail launch of a tail-split layer starts further down)
  int part_m0;  // first output row held by `partial` (rows are stored relative to it)
  // Optional: the output's bf16 planes, written by the epilogue itself for the convolutions that will consume this tensor
  // pre-split (the layout swem_split_bf16x3_f32 produces, [plane][Cout/8][M][8]): variant 0 = the planes of y, variant 1 =
  // the planes of relu(y) (a consumer that applies its input ReLU while splitting).  NULL = not wanted.
  // Optional: the residual as operand PLANES instead of an fp32 map (round 4): a block output that only convolutions and the
  // next block's residual add consume is written as planes only, and the add reads it back from them -- hi + mid of the fp16
  // pair (exact in fp32: the value to 22-23 significant bits) or hi + mid + lo of three bf16 planes (exactly the value).
  const unsigned short *res_pl;   // plane 0 of the residual, [C/8][res_npx][8]; NULL = the fp32 map `res`
  long long res_ps;               // elements between its planes
  int res_npx, res_npl;           // pixels per plane; SWEM_PLANES_F16 or 3
  unsigned short *ysp[2];
  int ysp_npl[2];     // planes to write per variant: 2 (hi, mid: all consumers run bf16x3) or 3
  long long ysp_ps;   // elements between the planes (M * Cout)
  // Stream-K (conv_igemm_bf3s_kernel, plan bits 24-27 == 1): sk_workers persistent blocks share the tiles x k-blocks
  // iteration space in equal contiguous ranges; see the kernel
  // n / HoWo and n / Wo for 0 <= n < 2^31 as umulhi(n, mul) >> sh (fast_div; a runtime integer division is ~40 dependent
  // vector instructions, and the row set-up of every block does four of them)
  unsigned fd_howo_mul, fd_howo_sh, fd_wo_mul, fd_wo_sh;
  int sk_workers, sk_mtiles, sk_ntiles;
  float *sk_ws;         // one block tile of fp32 partial sums per worker
  unsigned *sk_flags;   // one word per worker, zero at launch: "my partial tile is in sk_ws"
  int spin_limit;       // polls a block may spend waiting for another block's partial tile (2^24 ~ seconds; SWEM_SPIN_LIMIT)
  unsigned *fault;      // optional sticky fault word (the caller's, one per device: include/swem_hip.h): a bounded wait that expires
                        // or an output that does not fit the fp16 pair it is written as ORs SWEM_FAULT_* into it -- the tile is
                        // then wrong, and the host can know (never cleared on the device)
};

// One v_max_f32 per element.  fmaxf() costs two (hipcc first canonicalises the operand with v_max x,x), and in the fp32
// MFMA k-loop every vector instruction is time taken from the matrix pipe.  NaN inputs map to 0 (IEEE maxNum),
// like ATen's clamp-based ReLU on this path's finite activations.
__device__ __forceinline__ float relu1(float x) {
  float y;
  asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
  return y;
}
// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2), so the
// linear id is remapped to give every XCD one contiguous range of tiles, with the N tiles of one M tile adjacent:
// the tiles that share activation rows (same M tile, neighbouring M tiles' 3x3 halo) then hit the same L2.
// With xpn > 1 the N tiles are first cut into xpn groups and the order runs group by group (M tiles outer, the group's
// N tiles inner): 8 / xpn XCDs share one group and split its M tiles, so an XCD fetches 1/xpn of the filters and
// xpn/8 of the activations instead of all filters and 1/8 of the activations -- the cheaper cut where the filters
// outweigh the activations (the 30x54 layers: 60 MB of filter planes against 20 MB of activation planes).
// Placement only changes speed, never results.
__device__ __forceinline__ void tile_coords(int &mt, int &nt, int xpn = 1) {
  const int nwg = gridDim.x * gridDim.y, id = blockIdx.x + blockIdx.y * gridDim.x;
  const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, local = id >> 3;
  const int nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  const int ng = (int)gridDim.y / xpn, G = (int)gridDim.x * ng;   // N tiles per group, tiles per group
  const int jn = nid / G, rem = nid - jn * G;
  mt = rem / ng;
  nt = jn * ng + rem - mt * ng;
}

__device__ __forceinline__ float4 relu4(float4 v) { return make_float4(relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w)); }

// zero a loaded vector with a bit mask: the value is used unconditionally, so the load itself stays unconditional
__device__ __forceinline__ float4 mask4(float4 v, bool keep) {
  const unsigned m = keep ? 0xffffffffu : 0u;
  return make_float4(__uint_as_float(__float_as_uint(v.x) & m), __uint_as_float(__float_as_uint(v.y) & m),
                     __uint_as_float(__float_as_uint(v.z) & m), __uint_as_float(__float_as_uint(v.w) & m));
}

// floor(n / d) for 0 <= n < 2^31 by a host-made multiplier: mul = floor(2^(31+l) / d) + 1, sh = l - 1 with l = ceil(log2 d)
// (checked exhaustively against // for d < 3000 and on random d < 2^22); d = 1 is mul = 0.
static inline void fast_div_make(unsigned d, unsigned &mul, unsigned &sh) {
  if (d <= 1) {
    mul = 0; sh = 0;
    return;
  }
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  mul = (unsigned)(((1ull << (31 + l)) / d) + 1);
  sh = l - 1;
}
__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned sh) {
  return mul ? (int)(__umulhi((unsigned)n, mul) >> sh) : n;
}

// Source pixel coordinate of output coordinate o (carried as o0) under tap k.
//   forward:  o0 = o*stride - pad,  i = o0 + k
//   data gradient (SWEM_CONV_DGRAD: x is dY, the output is dX): o0 = o + pad, t = o0 - k must be a multiple of the
//   stride, i = t / stride  (stride 1 or 2).  Evaluated only when the tap changes, never inside the k-loop.
__device__ __forceinline__ bool tap_coord(const ConvP &p, int o0, int k, int lim, int &i) {
  if (p.flags & SWEM_CONV_DGRAD) {
    const int t = o0 - k;
    i = t >> (p.stride - 1);
    return t >= 0 && (t & (p.stride - 1)) == 0 && i < lim;
  }
  i = o0 + k;
  return (unsigned)i < (unsigned)lim;
}
__device__ __forceinline__ int tap_origin(const ConvP &p, int o) {
  return (p.flags & SWEM_CONV_DGRAD) ? o + p.pad : o * p.stride - p.pad;
}

// ---- output planes from the epilogue (fused operand split) ----------------------------------------------------------
// A wave stages one 32-pixel x 32-channel tile of final outputs in its own slice of the (now idle) operand LDS as dwords
// (hi | mid << 16), re-reads it as 16-byte runs of 8 channels and stores them into the [C/8][M][8] planes; a second pass
// does the lo plane when three are wanted.  The split is bf16_split.h's (bit-identical to swem_split_bf16x3_f32).
constexpr int PL_STRIDE = 36;                     // dwords per staged row: 16-byte aligned, conflict-free b128 reads
constexpr int PL_BYTES = 32 * PL_STRIDE * 4;      // LDS bytes per wave
__device__ __forceinline__ unsigned *planes_lds() {
  extern __shared__ __attribute__((aligned(16))) char smem_pl[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  return reinterpret_cast<unsigned *>(smem_pl + wave * PL_BYTES);
}
__device__ __forceinline__ void split_bf16_3(float v, unsigned &hm, unsigned &lo) {
  const unsigned h = pack_bf16(v, 0.f) & 0xffffu;
  const float r1 = v - lo_f32(h);
  const unsigned m = pack_bf16(r1, 0.f) & 0xffffu;
  const float r2 = r1 - lo_f32(m);
  hm = h | (m << 16);
  lo = pack_bf16(r2, 0.f) & 0xffffu;
}
// Flush the staged tile (rows m_base + [0,32), channels n_base + [0,32)); pass 0 writes planes 0 and 1 from (hi | mid << 16)
// dwords, pass 1 writes plane 2 from dwords holding lo in their low half.
__device__ __forceinline__ void planes_flush(const ConvP &p, const unsigned *lds, unsigned short *dst, int m_base, int n_base,
                                             int pass) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's staging writes have landed
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int id = lane + 64 * c, k = id >> 5, ml = id & 31;   // 8-channel group k of row ml: consecutive lanes = consecutive pixels
    const int m = m_base + ml, n = n_base + 8 * k;
    const uint4 a = *reinterpret_cast<const uint4 *>(lds + ml * PL_STRIDE + 8 * k);
    const uint4 b = *reinterpret_cast<const uint4 *>(lds + ml * PL_STRIDE + 8 * k + 4);
    if (m < p.M && n < p.Cout) {
      unsigned short *q = dst + ((long long)(n >> 3) * p.M + m) * 8;
      uint4 lo4 = make_uint4(__builtin_amdgcn_perm(a.y, a.x, 0x05040100), __builtin_amdgcn_perm(a.w, a.z, 0x05040100),
                             __builtin_amdgcn_perm(b.y, b.x, 0x05040100), __builtin_amdgcn_perm(b.w, b.z, 0x05040100));
      if (pass == 0) {
        uint4 hi4 = make_uint4(__builtin_amdgcn_perm(a.y, a.x, 0x07060302), __builtin_amdgcn_perm(a.w, a.z, 0x07060302),
                               __builtin_amdgcn_perm(b.y, b.x, 0x07060302), __builtin_amdgcn_perm(b.w, b.z, 0x07060302));
        *reinterpret_cast<uint4 *>(q) = lo4;                    // plane 0: the low halves (hi terms)
        *reinterpret_cast<uint4 *>(q + p.ysp_ps) = hi4;         // plane 1: the high halves (mid terms)
      } else {
        *reinterpret_cast<uint4 *>(q + 2 * p.ysp_ps) = lo4;     // plane 2
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);   // the reads are done before the next tile is staged
}

// Four consecutive channels n..n+3 of residual pixel `pix` from its planes (see ConvP::res_pl): 8 bytes per plane.
__device__ __forceinline__ float4 residual_from_planes(const ConvP &p, long long pix, int n) {
  const unsigned short *q = p.res_pl + ((long long)(n >> 3) * p.res_npx + pix) * 8 + (n & 7);
  const uint2 h = *reinterpret_cast<const uint2 *>(q), m = *reinterpret_cast<const uint2 *>(q + p.res_ps);
  if (p.res_npl == SWEM_PLANES_F16)
    return make_float4(lo_f16(h.x) + lo_f16(m.x), hi_f16(h.x) + hi_f16(m.x), lo_f16(h.y) + lo_f16(m.y), hi_f16(h.y) + hi_f16(m.y));
  const uint2 l = *reinterpret_cast<const uint2 *>(q + 2 * p.res_ps);
  return make_float4((lo_f32(h.x) + lo_f32(m.x)) + lo_f32(l.x), (hi_f32(h.x) + hi_f32(m.x)) + hi_f32(l.x),
                     (lo_f32(h.y) + lo_f32(m.y)) + lo_f32(l.y), (hi_f32(h.y) + hi_f32(m.y)) + hi_f32(l.y));
}

// One 32 x 32 tile of SCALED accumulators (acc * scale + shift), staged by the wave in its LDS slice in row order, goes
// out in ROW layout: lane (row id / 8, channel group id % 8) holds four consecutive channels of one pixel, adds the residual
// (or applies the mask) from a 16-byte load, applies the ReLU, stores 16 bytes of y -- an instruction covers 8 rows x 128
// bytes, against 4 rows x 64 bytes (or 2 x 128) of a dword store from the accumulator layout: a quarter of the store
// instructions (the epilogue was store-issue bound: 9.8k cycles of a 128x128 block, 4.6k of a 64x64 one) -- and writes the
// bf16 planes of the final values (an even lane and its odd neighbour hold the 8 channels of one 16-byte run).
__device__ __forceinline__ void out_tile32(const ConvP &p, const unsigned *lds, int m_base, int n_base) {
  const int lane = threadIdx.x & 63;
  const bool relu_out = p.flags & SWEM_CONV_RELU_OUT;
  const int HoWo = p.Ho * p.Wo;
  unsigned bad = 0;   // a value this lane wrote into an fp16 pair was beyond the fp16 range (SWEM_FAULT_RANGE)
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's staging writes have landed
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int id = lane + 64 * c, row = id >> 3, cg = id & 7;
    const int m = m_base + row, n = n_base + 4 * cg;
    const uint4 raw = *reinterpret_cast<const uint4 *>(lds + row * PL_STRIDE + 4 * cg);
    float4 v = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w));
    const bool in = m < p.M && n < p.Cout;
    if (in && (p.res || p.res_pl)) {
      const int b = fast_div(m, p.fd_howo_mul, p.fd_howo_sh);
      // (res_bs: fp32 elements between batch items, 0 = one image for the whole batch; the planes' pixel index follows from it)
      const float4 rv = p.res_pl ? residual_from_planes(p, (long long)b * (p.res_bs / p.Cout) + (m - b * HoWo), n)
                                 : *reinterpret_cast<const float4 *>(p.res + (long long)b * p.res_bs + (long long)(m - b * HoWo) * p.Cout + n);
      if (p.flags & SWEM_CONV_MASK_POS) {
        v = make_float4(rv.x > 0.f ? v.x : 0.f, rv.y > 0.f ? v.y : 0.f, rv.z > 0.f ? v.z : 0.f, rv.w > 0.f ? v.w : 0.f);
      } else {
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
    }
    // (a NaN must not leave through a ReLU as a clean 0 in an fp16 pair without a trace -- fmaxf(NaN, 0) = 0, and the planes of
    // relu(y) may be the only output: the range test below looks at the value BEFORE any ReLU too)
    const unsigned nan_in = f32_nan(v);
    if (relu_out) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    if (in && p.y) *reinterpret_cast<float4 *>(p.y + (long long)m * p.Cout + n) = v;   // (y = NULL: planes-only output)
#pragma unroll
    for (int var = 0; var < 2; ++var) {
      if (!p.ysp[var]) continue;
      const float4 q = var ? relu4(v) : v;
      uint2 h, mm, l;
      unsigned oor = 0;
      split_as(p.ysp_npl[var], q, h, mm, l, oor);
      if (p.ysp_npl[var] == SWEM_PLANES_F16) oor |= nan_in;
      if (in) bad |= oor;
      const uint2 h2 = make_uint2(__shfl_down(h.x, 1), __shfl_down(h.y, 1));
      const uint2 m2 = make_uint2(__shfl_down(mm.x, 1), __shfl_down(mm.y, 1));
      const uint2 l2 = make_uint2(__shfl_down(l.x, 1), __shfl_down(l.y, 1));
      if (in && (cg & 1) == 0) {
        unsigned short *d = p.ysp[var] + ((long long)(n >> 3) * p.M + m) * 8;
        *reinterpret_cast<uint4 *>(d) = make_uint4(h.x, h.y, h2.x, h2.y);
        *reinterpret_cast<uint4 *>(d + p.ysp_ps) = make_uint4(mm.x, mm.y, m2.x, m2.y);
        if (p.ysp_npl[var] == 3) *reinterpret_cast<uint4 *>(d + 2 * p.ysp_ps) = make_uint4(l.x, l.y, l2.x, l2.y);
      }
    }
  }
  if (p.ysp[0] || p.ysp[1]) range_fault(p.fault, bad);
  __builtin_amdgcn_s_waitcnt(0xc07f);   // the reads are done before the next tile is staged
}

// Epilogue shared by both kernels: raw split-K partials, or scale/shift (+residual, ReLU) / GLU gate, NHWC stores.
template <int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvP &p, f32x16 (&acc)[WM][WN], int m0, int n0, int wm, int wn,
                                              int r, int h) {
  const int mrow0 = m0 + wm * 32 * WM;
  const int ncol0 = n0 + wn * 32 * WN;
  if (p.partial) {  // split-K: raw partial sums, [z][M][Ncols]
    float *dst = p.partial + (long long)blockIdx.z * (p.M - p.part_m0) * p.Ncols - (long long)p.part_m0 * p.Ncols;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int jn = 0; jn < WN; ++jn) {
        int n = ncol0 + 32 * jn + r;
        if (n >= p.Ncols) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int m = mrow0 + 32 * i + acc_row(e, h);
          if (m < p.M) dst[(long long)m * p.Ncols + n] = acc[i][jn][e];
        }
      }
    return;
  }
  if constexpr (WN == 2) {
    if (p.flags & SWEM_CONV_GLU) {
      // packed columns [group][f|a][32]: this wave's two N tiles are the f and a banks of one group; the gated values go
      // through the wave's LDS slice and out in row layout like every other output (16-byte stores, bf16 planes)
      const int nf = ncol0 + r, na = ncol0 + 32 + r;
      const bool nin = na < p.Ncols;
      const float scf = (nin && p.scale) ? p.scale[nf] : p.uscale, sca = (nin && p.scale) ? p.scale[na] : p.uscale;
      const float shf = (nin && p.shift) ? p.shift[nf] : 0.f, sha = (nin && p.shift) ? p.shift[na] : 0.f;
      __syncthreads();   // every wave is done with the operand stages
      unsigned *lds = planes_lds();
#pragma unroll
      for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float f = acc[i][0][e] * scf + shf, a = acc[i][1][e] * sca + sha;
          lds[acc_row(e, h) * PL_STRIDE + r] = __float_as_uint(f * sigmoidf_(a));
        }
        out_tile32(p, lds, mrow0 + 32 * i, nin ? (ncol0 >> 1) : p.Cout);
      }
      return;
    }
  }
  __syncthreads();   // every wave is done with the operand stages: the LDS is free for the output staging
  unsigned *lds = planes_lds();
#pragma unroll
  for (int jn = 0; jn < WN; ++jn) {
    const int n = ncol0 + 32 * jn + r;
    const bool nin = n < p.Ncols;
    const float sc = (nin && p.scale) ? p.scale[n] : p.uscale, sh = (nin && p.shift) ? p.shift[n] : 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) lds[acc_row(e, h) * PL_STRIDE + r] = __float_as_uint(acc[i][jn][e] * sc + sh);
      out_tile32(p, lds, mrow0 + 32 * i, ncol0 + 32 * jn);
    }
  }
}

template <int WM, int WN, bool DB>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int SA = BM + 1, SB = BN + 1;  // row strides in float4 slots
  constexpr int RA = BM / 32, RB = BN / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *As = reinterpret_cast<float4 *>(smem);  // [2][KQ][SA]
  constexpr int NBUF = DB ? 2 : 1;
  float4 *Bs = As + NBUF * KQ * SA;               // [NBUF][KQ][SB]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int tm, tn;
  tile_coords(tm, tn);
  tm += p.mt0;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-thread gather coordinates (fixed over the K loop) ----
  const int kq = tid & 7, rbase = tid >> 3;
  int iy0[RA], ix0[RA], bidx[RA];
  bool rowok[RA];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = m0 + rbase + 32 * i;
    rowok[i] = m < p.M;
    int b = m / HoWo;
    int rem = m - b * HoWo;
    int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    iy0[i] = tap_origin(p, oy);
    ix0[i] = tap_origin(p, ox);
    bidx[i] = b;
  }
  const float *wrow[RB];
  bool colok[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    int n = n0 + rbase + 32 * i;
    colok[i] = n < p.Ncols;
    wrow[i] = p.w + (long long)(m0 / (p.Ho * p.Wo)) * p.w_bs + (long long)(colok[i] ? n : 0) * p.K;
  }
  const bool relu_in = p.flags & SWEM_CONV_RELU_IN;

  float4 ga[RA], gb[RB];
  unsigned okbits = 0;  // validity of the staged vectors: bit i = A row i, bit 8+i = B row i (applied at the LDS store)
  // Loads are unconditional (out-of-image taps read a clamped in-bounds address and are zeroed by a select) so the
  // compiler issues all of them back to back with ONE wait before the LDS stores; the input ReLU is applied at the
  // store, not at the load (a use right behind each load would serialise the L2 round trips).
  auto gload = [&](int kb) {
    const int k = kb * BK + kq * 4;
    const bool kok = k < p.K;
    const int kc = kok ? k : 0;
    unsigned bits = 0;
    int tap = kc / p.Cin;
    int ci = kc - tap * p.Cin;
    int ky = tap / p.KW, kx = tap - ky * p.KW;
    // channel -> (source, local channel) by selects: no runtime-indexed arrays (they would go to scratch)
    const float *src = p.x[0];
    int cs = p.c[0];
    long long sbs = p.bs[0];
    if (ci >= p.c[0]) {
      ci -= p.c[0];
      src = p.x[1];
      cs = p.c[1];
      sbs = p.bs[1];
      if (ci >= p.c[1]) {
        ci -= p.c[1];
        src = p.x[2];
        cs = p.c[2];
        sbs = p.bs[2];
      }
    }
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int iy = iy0[i] + ky, ix = ix0[i] + kx;
      bool ok = kok && rowok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      long long off = ok ? bidx[i] * sbs + ((long long)iy * p.W + ix) * cs + ci : 0;
      ga[i] = *reinterpret_cast<const float4 *>(src + off);
      bits |= ok ? (1u << i) : 0u;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      gb[i] = *reinterpret_cast<const float4 *>(wrow[i] + kc);
      bits |= (kok && colok[i]) ? (1u << (8 + i)) : 0u;
    }
    okbits = bits;
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      float4 v = mask4(ga[i], (okbits >> i) & 1u);
      As[(buf * KQ + kq) * SA + rbase + 32 * i] = relu_in ? relu4(v) : v;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) Bs[(buf * KQ + kq) * SB + rbase + 32 * i] = mask4(gb[i], (okbits >> (8 + i)) & 1u);
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int kb_begin = blockIdx.z * p.kb_per_split;
  const int kb_end = min(p.nkb, kb_begin + p.kb_per_split);

  gload(kb_begin);
  lstore(0);
  __syncthreads();
  int buf = 0;
  for (int kb = kb_begin; kb < kb_end; ++kb) {
    const bool more = kb + 1 < kb_end;
    if (more) gload(kb + 1);
    const float4 *Ab = As + buf * KQ * SA + wm * 32 * WM + r + h * SA;
    const float4 *Bb = Bs + buf * KQ * SB + wn * 32 * WN + r + h * SB;
    // operand fragments are fetched one k-group ahead of the MFMAs that use them
    float4 a[2][WM], b[2][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) a[0][i] = Ab[32 * i];
#pragma unroll
    for (int i = 0; i < WN; ++i) b[0][i] = Bb[32 * i];
#pragma unroll
    for (int j = 0; j < KQ / 2; ++j) {
      if (j + 1 < KQ / 2) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a[(j + 1) & 1][i] = Ab[2 * (j + 1) * SA + 32 * i];
#pragma unroll
        for (int i = 0; i < WN; ++i) b[(j + 1) & 1][i] = Bb[2 * (j + 1) * SB + 32 * i];
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int jn = 0; jn < WN; ++jn) acc[i][jn] = mfma32x4(a[j & 1][i], b[j & 1][jn], acc[i][jn]);
      // keep the LDS reads of group j+1 ahead of the MFMAs of group j (the scheduler otherwise sinks them behind and
      // reuses the fragment registers, exposing the LDS latency: 124 vs 155 TFLOP/s in tools/mfma_lds.hip)
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (DB) {
      if (more) lstore(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    } else {
      // single LDS image: half the LDS, twice the resident blocks; other blocks' MFMAs cover this block's refill
      __syncthreads();
      if (more) lstore(0);
      __syncthreads();
    }
  }

  conv_epilogue<WM, WN>(p, acc, m0, n0, wm, wn, r, h);
}

// ------------------------------------------------------------------------------------------------------------
// Fast variant for the common case (every source width, hence Cin, a multiple of 32).
//
// Measured on gfx950 (profiles/r01_conv_pmc.txt): while v_mfma_f32_32x32x2_f32 runs, NO vector-ALU instruction
// co-executes (SQ_VALU_MFMA_COEXEC_CYCLES = 0: the fp32 MFMA runs at the fp32 vector rate on the same datapath), so
// every VALU instruction in the k-loop is time taken from the matrix pipe.  This kernel therefore keeps the loop free
// of vector arithmetic:
//   - a 32-wide k-block lies inside one tap of one source, so its position (source, tap, channel) is block-uniform
//     and advances with scalar instructions;
//   - operands are fetched with raw buffer loads: per-lane byte offsets are recomputed only when the tap or source
//     changes (every Cin/32 k-blocks), the channel offset travels in the scalar offset, and out-of-image taps /
//     out-of-range rows use an out-of-range offset, which the buffer unit returns as zeros (no masks, no branches);
//   - the only vector op left is the optional input ReLU on the staged values.
struct KPos {
  int src, ci0, ky, kx;
};
__device__ __forceinline__ void kpos_init(KPos &q, const ConvP &p, int kb) {
  int k = kb * BK;
  int tap = k / p.Cin;
  int ci = k - tap * p.Cin;
  q.ky = tap / p.KW;
  q.kx = tap - q.ky * p.KW;
  q.src = 0;
  if (ci >= p.c[0]) {
    ci -= p.c[0];
    q.src = 1;
    if (ci >= p.c[1]) {
      ci -= p.c[1];
      q.src = 2;
    }
  }
  q.ci0 = ci;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 bufload4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
constexpr unsigned OOB = 0xfffffff0u;  // >= any num_records: the load returns 0 and touches no memory

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_igemm_pipe_kernel(ConvP p) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int SA = BM + 1, SB = BN + 1;
  constexpr int RA = BM / 32, RB = BN / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *As = reinterpret_cast<float4 *>(smem);  // [2][KQ][SA]
  float4 *Bs = As + 2 * KQ * SA;                  // [2][KQ][SB]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int tm, tn;
  tile_coords(tm, tn);
  tm += p.mt0;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kq = tid & 7, rbase = tid >> 3;

  // buffer descriptors (wave-uniform: built from kernel arguments and block-uniform scalars only)
  const long long HWin = (long long)p.H * p.W;
  auto src_rsrc = [&](int sidx) __attribute__((always_inline)) {
    const float *base = sidx == 0 ? p.x[0] : (sidx == 1 ? p.x[1] : p.x[2]);
    const int cs = sidx == 0 ? p.c[0] : (sidx == 1 ? p.c[1] : p.c[2]);
    const long long bs = sidx == 0 ? p.bs[0] : (sidx == 1 ? p.bs[1] : p.bs[2]);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0,
                                             (int)((bs ? (long long)p.B * bs : HWin * cs) * 4), 0x00020000);
  };
  __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.w + (long long)(m0 / (p.Ho * p.Wo)) * p.w_bs), 0, (int)((long long)p.Ncols * p.K * 4),
      0x00020000);

  int iy0[RA], ix0[RA], bidx[RA];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = m0 + rbase + 32 * i;
    int b = m / HoWo;
    int rem = m - b * HoWo;
    int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    iy0[i] = tap_origin(p, oy);
    ix0[i] = tap_origin(p, ox);
    bidx[i] = m < p.M ? b : -1;
  }
  unsigned wvoff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    int n = n0 + rbase + 32 * i;
    wvoff[i] = n < p.Ncols ? (unsigned)(((long long)n * p.K + kq * 4) * 4) : OOB;
  }
  const bool relu_in = p.flags & SWEM_CONV_RELU_IN;

  // per-lane byte offsets of the A rows for the current (source, tap); recomputed only when that changes
  unsigned avoff[RA];
  auto set_tap = [&](const KPos &q) {
    const int cs = q.src == 0 ? p.c[0] : (q.src == 1 ? p.c[1] : p.c[2]);
    const long long bs = q.src == 0 ? p.bs[0] : (q.src == 1 ? p.bs[1] : p.bs[2]);
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int iy, ix;
      const bool oky = tap_coord(p, iy0[i], q.ky, p.H, iy), okx = tap_coord(p, ix0[i], q.kx, p.W, ix);
      const bool ok = bidx[i] >= 0 && oky && okx;
      const long long e = bidx[i] * bs + ((long long)iy * p.W + ix) * cs + kq * 4;
      avoff[i] = ok ? (unsigned)(e * 4) : OOB;
    }
  };

  float4 ga[RA], gb[RB];
  __amdgpu_buffer_rsrc_t rsa;
  auto gload = [&](const KPos &q, int kb) {
    const unsigned soff = (unsigned)q.ci0 * 4u;
#pragma unroll
    for (int i = 0; i < RA; ++i) ga[i] = bufload4(rsa, avoff[i], soff);
    const unsigned wsoff = (unsigned)kb * (BK * 4u);
#pragma unroll
    for (int i = 0; i < RB; ++i) gb[i] = bufload4(rsw, wvoff[i], wsoff);
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < RA; ++i) As[(buf * KQ + kq) * SA + rbase + 32 * i] = relu_in ? relu4(ga[i]) : ga[i];
#pragma unroll
    for (int i = 0; i < RB; ++i) Bs[(buf * KQ + kq) * SB + rbase + 32 * i] = gb[i];
  };
  // advance to the next k-block (scalar); refresh the lane offsets when the tap or the source changes
  auto advance = [&](KPos &q) {
    q.ci0 += BK;
    const int cs = q.src == 0 ? p.c[0] : (q.src == 1 ? p.c[1] : p.c[2]);
    if (q.ci0 >= cs) {
      q.ci0 = 0;
      ++q.src;
      const int cn = q.src == 1 ? p.c[1] : (q.src == 2 ? p.c[2] : 0);
      if (cn == 0) {
        q.src = 0;
        if (++q.kx == p.KW) {
          q.kx = 0;
          ++q.ky;
        }
      }
      set_tap(q);
      rsa = src_rsrc(q.src);
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int kb_begin = blockIdx.z * p.kb_per_split;
  const int kb_end = min(p.nkb, kb_begin + p.kb_per_split);
  KPos q;
  kpos_init(q, p, kb_begin);
  set_tap(q);
  rsa = src_rsrc(q.src);
  gload(q, kb_begin);
  lstore(0);
  __syncthreads();
  int buf = 0;
  for (int kb = kb_begin; kb < kb_end; ++kb) {
    // the gathers for the next k-block are issued unconditionally (the last iteration re-reads its own block into the
    // idle buffer): a conditional definition of the staging registers costs register copies + early waits here
    const bool more = kb + 1 < kb_end;
    if (more) advance(q);
    gload(q, more ? kb + 1 : kb);
    const float4 *Ab = As + buf * KQ * SA + wm * 32 * WM + r + h * SA;
    const float4 *Bb = Bs + buf * KQ * SB + wn * 32 * WN + r + h * SB;
    float4 a[2][WM], b[2][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) a[0][i] = Ab[32 * i];
#pragma unroll
    for (int i = 0; i < WN; ++i) b[0][i] = Bb[32 * i];
#pragma unroll
    for (int j = 0; j < KQ / 2; ++j) {
      if (j + 1 < KQ / 2) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a[(j + 1) & 1][i] = Ab[2 * (j + 1) * SA + 32 * i];
#pragma unroll
        for (int i = 0; i < WN; ++i) b[(j + 1) & 1][i] = Bb[2 * (j + 1) * SB + 32 * i];
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int jn = 0; jn < WN; ++jn) acc[i][jn] = mfma32x4(a[j & 1][i], b[j & 1][jn], acc[i][jn]);
      __builtin_amdgcn_sched_barrier(0);
    }
    lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  conv_epilogue<WM, WN>(p, acc, m0, n0, wm, wn, r, h);
}

// ------------------------------------------------------------------------------------------------------------
// fp32-accurate convolution on the bf16 matrix cores ("bf16x6"): every fp32 operand is split exactly into three bf16
// terms  x = hi + mid + lo  (8 + 8 + 8 significant bits, round-to-nearest residuals) while it is staged into LDS, and
// the product is rebuilt from the six term pairs above 2^-24:  hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid,
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (each bf16 x bf16 product is exact).  The dropped pairs are <= 2^-24
// relative, i.e. the result carries fp32-level error, at 16/6 = 2.7x the fp32-MFMA rate -- and, unlike the fp32 MFMA,
// the bf16 MFMA co-executes with the vector ALU, so the splitting arithmetic of one wave hides behind another's MFMAs.
// Same tiling, loads, k-block bookkeeping and epilogue as conv_igemm_pipe_kernel; LDS holds three bf16 planes per
// operand,  [plane][k/8][row][8 bf16], one 16-byte MFMA fragment per (row, k/8), plane stride padded by 16 bytes.
__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 bufload2(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
  return make_uint2(v.x, v.y);
}

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_igemm_bf3_kernel(ConvP p) {
  constexpr bool PS = false;  // (pre-split filters in this register-staged variant were measured slower: more loads)
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int RA = BM / 32, RB = BN / 32;
  constexpr int SA = BM + 1, SB = BN + 1;          // k/8-group strides in 16-byte slots (+1: conflict-free stores)
  constexpr int PA = 4 * SA, PB = 4 * SB;         // plane strides
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4 *As = reinterpret_cast<uint4 *>(smem);  // [2 buffers][3 planes][PA]
  uint4 *Bs = As + 2 * 3 * PA;                  // [2][3][PB]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int tm, tn;
  tile_coords(tm, tn);
  tm += p.mt0;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kq = tid & 7, rbase = tid >> 3;

  const long long HWin = (long long)p.H * p.W;
  auto src_rsrc = [&](int sidx) __attribute__((always_inline)) {
    const float *base = sidx == 0 ? p.x[0] : (sidx == 1 ? p.x[1] : p.x[2]);
    const int cs = sidx == 0 ? p.c[0] : (sidx == 1 ? p.c[1] : p.c[2]);
    const long long bs = sidx == 0 ? p.bs[0] : (sidx == 1 ? p.bs[1] : p.bs[2]);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0,
                                             (int)((bs ? (long long)p.B * bs : HWin * cs) * 4), 0x00020000);
  };
  __amdgpu_buffer_rsrc_t rsw =
      PS ? __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(p.wsplit), 0,
                                             (int)((long long)3 * p.Ncols * p.K * 2), 0x00020000)
         : __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w + (long long)(m0 / (p.Ho * p.Wo)) * p.w_bs), 0,
                                             (int)((long long)p.Ncols * p.K * 4), 0x00020000);
  const unsigned wplane = (unsigned)((long long)p.Ncols * p.K * 2);  // bytes per pre-split plane

  int iy0[RA], ix0[RA], bidx[RA];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    int m = m0 + rbase + 32 * i;
    int b = m / HoWo;
    int rem = m - b * HoWo;
    int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    iy0[i] = tap_origin(p, oy);
    ix0[i] = tap_origin(p, ox);
    bidx[i] = m < p.M ? b : -1;
  }
  unsigned wvoff[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    int n = n0 + rbase + 32 * i;
    wvoff[i] = n < p.Ncols ? (unsigned)(((long long)n * p.K + kq * 4) * (PS ? 2 : 4)) : OOB;
  }
  const bool relu_in = p.flags & SWEM_CONV_RELU_IN;
  unsigned avoff[RA];
  auto set_tap = [&](const KPos &q) {
    const int cs = q.src == 0 ? p.c[0] : (q.src == 1 ? p.c[1] : p.c[2]);
    const long long bs = q.src == 0 ? p.bs[0] : (q.src == 1 ? p.bs[1] : p.bs[2]);
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      int iy, ix;
      const bool oky = tap_coord(p, iy0[i], q.ky, p.H, iy), okx = tap_coord(p, ix0[i], q.kx, p.W, ix);
      const bool ok = bidx[i] >= 0 && oky && okx;
      const long long e = bidx[i] * bs + ((long long)iy * p.W + ix) * cs + kq * 4;
      avoff[i] = ok ? (unsigned)(e * 4) : OOB;
    }
  };
  float4 ga[RA], gb[PS ? 1 : RB];
  uint2 gs[PS ? 3 : 1][RB];
  __amdgpu_buffer_rsrc_t rsa;
  auto gload = [&](const KPos &q, int kb) {
    const unsigned soff = (unsigned)q.ci0 * 4u;
#pragma unroll
    for (int i = 0; i < RA; ++i) ga[i] = bufload4(rsa, avoff[i], soff);
    if constexpr (PS) {
      const unsigned wsoff = (unsigned)kb * (BK * 2u);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < RB; ++i) gs[pl][i] = bufload2(rsw, wvoff[i], wsoff + pl * wplane);
    } else {
      const unsigned wsoff = (unsigned)kb * (BK * 4u);
#pragma unroll
      for (int i = 0; i < RB; ++i) gb[i] = bufload4(rsw, wvoff[i], wsoff);
    }
  };
  // split + store: 8 bytes per plane at [plane][k8 = kq/2][row] + (kq & 1) * 8
  auto lstore = [&](int buf) {
    char *abase = reinterpret_cast<char *>(As + buf * 3 * PA) + ((kq >> 1) * SA + rbase) * 16 + (kq & 1) * 8;
    char *bbase = reinterpret_cast<char *>(Bs + buf * 3 * PB) + ((kq >> 1) * SB + rbase) * 16 + (kq & 1) * 8;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      uint2 hh, mm, ll;
      split3(relu_in ? relu4(ga[i]) : ga[i], hh, mm, ll);
      *reinterpret_cast<uint2 *>(abase + 32 * i * 16) = hh;
      *reinterpret_cast<uint2 *>(abase + PA * 16 + 32 * i * 16) = mm;
      *reinterpret_cast<uint2 *>(abase + 2 * PA * 16 + 32 * i * 16) = ll;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      uint2 hh, mm, ll;
      if constexpr (PS) {
        hh = gs[0][i];
        mm = gs[1][i];
        ll = gs[2][i];
      } else {
        split3(gb[i], hh, mm, ll);
      }
      *reinterpret_cast<uint2 *>(bbase + 32 * i * 16) = hh;
      *reinterpret_cast<uint2 *>(bbase + PB * 16 + 32 * i * 16) = mm;
      *reinterpret_cast<uint2 *>(bbase + 2 * PB * 16 + 32 * i * 16) = ll;
    }
  };
  auto advance = [&](KPos &q) {
    q.ci0 += BK;
    const int cs = q.src == 0 ? p.c[0] : (q.src == 1 ? p.c[1] : p.c[2]);
    if (q.ci0 >= cs) {
      q.ci0 = 0;
      ++q.src;
      const int cn = q.src == 1 ? p.c[1] : (q.src == 2 ? p.c[2] : 0);
      if (cn == 0) {
        q.src = 0;
        if (++q.kx == p.KW) {
          q.kx = 0;
          ++q.ky;
        }
      }
      set_tap(q);
      rsa = src_rsrc(q.src);
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int kb_begin = blockIdx.z * p.kb_per_split;
  const int kb_end = min(p.nkb, kb_begin + p.kb_per_split);
  KPos q;
  kpos_init(q, p, kb_begin);
  set_tap(q);
  rsa = src_rsrc(q.src);
  gload(q, kb_begin);
  lstore(0);
  __syncthreads();
  int buf = 0;
  for (int kb = kb_begin; kb < kb_end; ++kb) {
    const bool more = kb + 1 < kb_end;
    if (more) advance(q);
    gload(q, more ? kb + 1 : kb);
    const uint4 *Ab = As + buf * 3 * PA + wm * 32 * WM + r;
    const uint4 *Bb = Bs + buf * 3 * PB + wn * 32 * WN + r;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {  // two 16-wide k-steps; lane half h holds k = 8h..8h+7 of the step
      const int k8 = 2 * s2 + h;
      uint4 a[3][WM], b[3][WN];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < WM; ++i) a[pl][i] = Ab[pl * PA + k8 * SA + 32 * i];
#pragma unroll
        for (int i = 0; i < WN; ++i) b[pl][i] = Bb[pl * PB + k8 * SB + 32 * i];
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int jn = 0; jn < WN; ++jn) {
          f32x16 c = acc[i][jn];
          c = mfma_bf16(a[0][i], b[2][jn], c);  // small terms first
          c = mfma_bf16(a[2][i], b[0][jn], c);
          c = mfma_bf16(a[1][i], b[1][jn], c);
          c = mfma_bf16(a[0][i], b[1][jn], c);
          c = mfma_bf16(a[1][i], b[0][jn], c);
          c = mfma_bf16(a[0][i], b[0][jn], c);
          acc[i][jn] = c;
        }
    }
    lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  conv_epilogue<WM, WN>(p, acc, m0, n0, wm, wn, r, h);
}

// ------------------------------------------------------------------------------------------------------------
// bf16x6 with PRE-SPLIT operands: activations and filters arrive as three bf16 planes each (swem_split_bf16x3_f32 /
// weight packing), so one input element is split once per tensor instead of once per (tap, N tile) use.  The k-loop
// then holds no vector arithmetic and no register staging at all: every 16-byte MFMA fragment row-run is moved
// HBM/L2 -> LDS by the buffer unit itself (buffer_load_dwordx4 ... lds: 64 lanes x 16 B = 64 consecutive rows of one
// [plane][k/8] image; padding taps use an out-of-range offset and land as zeros) while the waves read fragments and
// issue MFMAs.  The planes are CHANNEL-GROUP MAJOR, [plane][C/8][pixel][8 bf16] (filters [plane][K/8][Cout][8]), so the
// 64 lanes of one transfer read 64 consecutive 16-byte chunks = full cache lines (with NHWC planes each lane would
// touch its own line: measured 80-100 instead of 115-125 TFLOP/s for the register-staged kernel).
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma_bf16_16(uint4 a, uint4 b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// F16: the planes hold fp16 pairs (bf16_split.h, split2h) and the products run on the f16 MFMA of the same shape -- the
// "f16x3" arithmetic: three products on 23-bit operands, fp32-level error at the bf16x3 kernel's cost.
template <bool F16>
__device__ __forceinline__ f32x4v mm16(uint4 a, uint4 b, f32x4v c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return mfma_bf16_16(a, b, c);
}
template <bool F16>
__device__ __forceinline__ f32x16 mm32(uint4 a, uint4 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return mfma_bf16(a, b, c);
}
// Epilogue for 16x16 accumulator tiles (v_mfma_f32_16x16x32_bf16): lane l holds column l % 16 and rows 4 (l / 16) + e.
// Same semantics as conv_epilogue; a wave owns TM2 x TN2 tiles = 16 TM2 rows x 16 TN2 columns.
template <int TM2, int TN2>
__device__ __forceinline__ void conv_epilogue16(const ConvP &p, f32x4v (&acc)[TM2][TN2], int mrow0, int ncol0, int lane) {
  const int col = lane & 15, rg = lane >> 4;
  if (p.partial) {
    float *dst = p.partial + (long long)blockIdx.z * (p.M - p.part_m0) * p.Ncols - (long long)p.part_m0 * p.Ncols;
#pragma unroll
    for (int i = 0; i < TM2; ++i)
#pragma unroll
      for (int j = 0; j < TN2; ++j) {
        const int n = ncol0 + 16 * j + col;
        if (n >= p.Ncols) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = mrow0 + 16 * i + 4 * rg + e;
          if (m < p.M) dst[(long long)m * p.Ncols + n] = acc[i][j][e];
        }
      }
    return;
  }
  if constexpr (TN2 == 4) {
    if (p.flags & SWEM_CONV_GLU) {
      // packed columns [group][f|a][32]: tiles 0,1 are f, tiles 2,3 the gates of the same 32 channels; the gated values go
      // through the wave's LDS slice and out in row layout (16-byte stores, bf16 planes)
      static_assert(TM2 % 2 == 0, "the output staging works on 32 x 32 sub-tiles");
      const bool nin = ncol0 + 64 <= p.Ncols;
      float scf[2], sca[2], shf[2], sha[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nf = ncol0 + 16 * j + col, na = nf + 32;
        scf[j] = (nin && p.scale) ? p.scale[nf] : p.uscale; sca[j] = (nin && p.scale) ? p.scale[na] : p.uscale;
        shf[j] = (nin && p.shift) ? p.shift[nf] : 0.f; sha[j] = (nin && p.shift) ? p.shift[na] : 0.f;
      }
      __syncthreads();   // every wave is done with the operand stages
      unsigned *lds = planes_lds();
#pragma unroll
      for (int i0 = 0; i0 < TM2; i0 += 2) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              lds[(16 * ii + 4 * rg + e) * PL_STRIDE + 16 * j + col] = __float_as_uint(
                  (acc[i0 + ii][j][e] * scf[j] + shf[j]) * sigmoidf_(acc[i0 + ii][j + 2][e] * sca[j] + sha[j]));
        out_tile32(p, lds, mrow0 + 16 * i0, nin ? (ncol0 >> 1) : p.Cout);
      }
      return;
    }
  }
  // 32 x 32 sub-tiles (2 x 2 accumulator tiles) through the wave's LDS slice, out in row layout (out_tile32)
  static_assert(TM2 % 2 == 0 && TN2 % 2 == 0, "the output staging works on 32 x 32 sub-tiles");
  __syncthreads();   // every wave is done with the operand stages
  unsigned *lds = planes_lds();
#pragma unroll
  for (int j0 = 0; j0 < TN2; j0 += 2) {
    float sc[2], sh[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int n = ncol0 + 16 * (j0 + jj) + col;
      const bool nin = n < p.Ncols;
      sc[jj] = (nin && p.scale) ? p.scale[n] : p.uscale;
      sh[jj] = (nin && p.shift) ? p.shift[n] : 0.f;
    }
#pragma unroll
    for (int i0 = 0; i0 < TM2; i0 += 2) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            lds[(16 * ii + 4 * rg + e) * PL_STRIDE + 16 * jj + col] = __float_as_uint(acc[i0 + ii][j0 + jj][e] * sc[jj] + sh[jj]);
      out_tile32(p, lds, mrow0 + 16 * i0, ncol0 + 16 * j0);
    }
  }
}

// M16: the products run on v_mfma_f32_16x16x32_bf16 (one k-block = one MFMA k-step) instead of 32x32x16: the same cycles
// per FLOP, but the chip holds a higher clock on this shape under a dense bf16 load (MI355X_MICROARCH.md, DVFS item 7).
// NPL = 3: the six-product bf16x6 arithmetic on the hi/mid/lo planes; NPL = 1: plain bf16 (hi plane only, one product) --
// the mixed-precision mode of the training step (config.AMP), never the default.
// KG = k/8 groups per k-block: 4 (32 k, the default) or 2 (16 k: half the LDS per stage -- the 128x128 tile then fits three
// blocks of four waves per CU instead of one block; a k-block is half of a (channel block, tap) cell of the K order).
// Stream-K instantiations re-read the launch parameters in every segment from the kernel-argument segment through a pointer
// the optimiser cannot see through: left alone it hoists ~60 invariant scalar loads (descriptors, strides, epilogue
// pointers) out of the segment loop, keeps them all live across the k-loop and spills 70 of them -- 30 more vector registers
// and one resident block per CU instead of two.  (ConvP is the kernel's first argument: offset 0 of the segment.)
template <bool SK>
__device__ __forceinline__ const ConvP &segment_params(const ConvP &p) {
  if constexpr (SK) {
    typedef const __attribute__((address_space(4))) ConvP CP;
    CP *pp = (CP *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(pp));
    return *(const ConvP *)pp;
  } else {
    return p;
  }
}

// PF ("prefetched fragments", round 3): the wave keeps the MFMA fragments of k-block kb in registers and reads those of
// kb+1 from the LDS BETWEEN the MFMA groups of kb, into a second register set -- its matrix stream no longer stops for its
// own transfer requests, fragment reads and the stage hand-over, and a stage is free for the next transfer one iteration
// earlier (its fragments are in registers), so a ring of NST stages keeps NST-1 k-blocks in flight under the MFMAs instead of
// NST-2 + a hand-over in front of them.  16x16x32 MFMA, one or two planes.
template <int WM, int WN, int NST, int NW, bool M16, int NPL, int KG = 4, bool PF = false, bool SK = false, bool F16 = false>
// (second launch bound = resident blocks per CU the register allocation must allow: the 16-k-block tiles (KG == 2) exist to run
// THREE blocks of four waves per CU -- 168 registers; left at 2 the three-plane instantiation came out at 169 once the epilogue
// grew by the plane-residual path, and the exact-split leg lost 4 %)
__global__ __launch_bounds__(64 * NW, NW == 4 ? (KG == 2 ? 3 : 2) : 1) void conv_igemm_bf3s_kernel(ConvP p STAMP_ARG) {
  STAMP(0);
  static_assert(!F16 || NPL == 2, "fp16 planes come as a (hi, mid) pair");
  static_assert(KG == 4 || (KG == 2 && !M16 && NW == 4), "16-k blocks: four waves, 32x32x16 MFMA");
  static_assert(!PF || (M16 && NPL <= 2 && KG == 4), "prefetched fragments: 16x16x32 MFMA, at most two planes");
  // block tile 64WM x 64WN, NW waves as an (NW/2) x 2 grid, each owning TM x TN 32x32 accumulator tiles
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int TM = 4 * WM / NW, TN = WN;
  static_assert(TM >= 1 && TM * (NW / 2) * 32 == BM, "wave grid must tile the block");
  // slot stride per k/8 group: the 16x16x32 fragment read mixes two k/8 groups inside one 16-lane bank group, which is
  // conflict free when the stride is a multiple of 16 slots; the 32x32x16 read does not care (kept as it was)
  constexpr int SA = BM + (M16 ? 0 : 1), SB = BN + (M16 ? 0 : 1);
  constexpr int PA = KG * SA, PB = KG * SB;
  constexpr int NA = NPL * KG * WM, NB = NPL * KG * WN;  // 64-row fragment runs per k-block
  // NST LDS stages: the transfers run NST-1 k-blocks ahead of the MFMAs
  constexpr int NDMA = (NA + NB) / NW;             // transfers per wave per k-block
  static_assert(NW == KG || WM == WN, "split roles: half the waves move the A image, half the B image, the same count each");
  static_assert((NA + NB) % NW == 0, "every wave must issue the same number of transfers (counted vmcnt)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4 *As = reinterpret_cast<uint4 *>(smem);  // [NST][3][PA]
  uint4 *Bs = As + NST * NPL * PA;              // [NST][NPL][PB]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  // ---- Stream-K.  A grid of T tiles on S resident-block slots runs ceil(T / S) rounds however few tiles the last one holds
  // (2x120x216x256 -> 256: 810 tiles of 128x128 on 512 slots = 1.58 -> 2 rounds; most other layers: ~410 blocks = one round
  // with a fifth of the slots empty).  Here the launch is sk_workers PERSISTENT blocks (every slot one), and worker v takes the
  // v-th equal share of the (tile, k-block) iterations, tiles in the XCD-aware order: the tail of a tile another worker began
  // (-> its partial sums go to the workspace), whole tiles, and the head of a last tile (-> this worker adds the partial
  // sums of the workers that follow, in their order: deterministic, and runs the epilogue).  The producer of a partial tile
  // works on it FIRST and never waits for anyone; the owner needs it LAST -- no worker waits on a worker that could be
  // waiting, and a worker that is not yet resident (other streams' kernels hold slots) is dispatched as earlier ones finish.
  // Hand-over across CUs / XCDs (per-XCD L2s are not coherent): the partial sums are stored and loaded past the caches
  // (sc0 sc1), the producer drains its stores (vmcnt(0) + block barrier) before its flag, flags are agent-scope atomics.
  constexpr bool sk = SK;   // (its own instantiation: the segment loop costs the plain kernel ~30 registers)
  long long sk_it = 0, sk_end = 0, sk_total = 0;
  int sk_v = 0;
  if (sk) {
    const int W_ = p.sk_workers, q8 = W_ >> 3, r8 = W_ & 7, xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    sk_v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + local;   // an XCD's workers share a tile range
    sk_total = (long long)p.sk_mtiles * p.sk_ntiles * p.nkb;
    sk_it = sk_total * sk_v / W_;
    sk_end = sk_total * (sk_v + 1) / W_;
  }
  const ConvP &p_arg = p;
  for (;;) {   // segments of this worker (without stream-K: the one tile of this block)
  const ConvP &p = segment_params<SK>(p_arg);
  int tm, tn, seg_k0 = 0, seg_k1 = p.nkb, sk_tile = 0;
  if (sk) {
    if (sk_it >= sk_end) break;
    sk_tile = (int)(sk_it / p.nkb);
    seg_k0 = (int)(sk_it - (long long)sk_tile * p.nkb);
    seg_k1 = (int)min((long long)p.nkb, seg_k0 + (sk_end - sk_it));
    sk_it += seg_k1 - seg_k0;
    const int ng = p.sk_ntiles / p.xpn, G = p.sk_mtiles * ng;   // tile order of tile_coords
    const int jn = sk_tile / G, rem = sk_tile - jn * G;
    tm = rem / ng;
    tn = jn * ng + rem - tm * ng;
    __syncthreads();   // every wave has left the previous segment (its epilogue stages planes in the operand LDS)
  } else {
    tile_coords(tm, tn, p.xpn);
    tm += p.mt0;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  auto src_rsrc = [&](int sidx) __attribute__((always_inline)) {
    const unsigned short *base = sidx == 0 ? p.xs[0] : (sidx == 1 ? p.xs[1] : p.xs[2]);
    const long long ps = sidx == 0 ? p.ps[0] : (sidx == 1 ? p.ps[1] : p.ps[2]);
    return raw_rsrc(base, (unsigned)(3 * ps * 2));
  };
  // (per-batch filter planes -- matching's value readout, a batched GEMM: tiles never straddle two batch items)
  const i32x4 rsw = raw_rsrc(p.wsplit + (p.w_bs ? (long long)(m0 / (p.Ho * p.Wo)) * p.w_bs : 0ll),
                             (unsigned)((long long)3 * p.Ncols * p.K * 2));
  const unsigned wplane = (unsigned)((long long)p.Ncols * p.K * 2);
  const unsigned wgroup = (unsigned)p.Ncols * 16u;  // bytes per k/8 group of the filter planes

  // rows served by this lane: run j covers rows j*64 + lane of the A (pixels) and B (filters) tiles
  int iy0[WM], ix0[WM], bidx[WM];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int j = 0; j < WM; ++j) {
    int m = m0 + j * 64 + lane;
    int b = fast_div(m, p.fd_howo_mul, p.fd_howo_sh);
    int rem = m - b * HoWo;
    int oy = fast_div(rem, p.fd_wo_mul, p.fd_wo_sh), ox = rem - oy * p.Wo;
    iy0[j] = tap_origin(p, oy);
    ix0[j] = tap_origin(p, ox);
    bidx[j] = m < p.M ? b : -1;
  }
  unsigned bvoff[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    int n = n0 + j * 64 + lane;
    bvoff[j] = n < p.Ncols ? (unsigned)n * 16u : OOB;
  }
#ifdef SWEM_PROLOGUE_STAMPS
  STAMP(6);
#endif
  // Each wave moves ONE k/8 group (g = wave & 3) of every plane: the slot -> (operand, plane, row run) map is then a
  // compile-time constant and the wave only adds its group's offsets, kept in scalars that advance by one add per k-block.
  // (With eight waves, waves 0-3 move the A image and 4-7 the B image.)  Scalar instructions share the wave's issue
  // slot with the MFMAs: the first version of this loop spent 24 % of its wave cycles on ~190 of them per k-block.
  const int g = wave % KG, half = wave / KG;   // (NW == KG: every wave moves both images of its group)
  unsigned avoff[WM];
  unsigned aplane = 0, agroup = 0;  // byte strides between the planes / the 8-channel groups of the current source
  unsigned a_base = 0;              // scalar offset of (current k-block, group g) in plane 0 of the current source
  unsigned w_base = 0;              // ... of the filter planes
  i32x4 rsa;
  // K order of this kernel: k = (ci / 32, ky, kx, ci % 32) over the concatenated input channels ("channel-block major";
  // the filter planes are packed in that order).  With the usual tap-major order (ky, kx, ci) a tap is one pass over all
  // input channels, so a block re-reads its pixels' neighbourhood KH*KW times, each time long after the last: with ~13
  // blocks per XCD the footprint of a pass exceeds the 4 MiB L2 and the activations are fetched once per tap (measured
  // on 2x30x54x1280 -> 512: 8 x filters + 9 x activations).  Here the KH*KW taps of one 32-channel block are
  // consecutive k-blocks and the re-reads hit the L2 right away.  The tap changes every k-block, so its per-lane cost
  // must be a couple of vector instructions: the lane keeps the pixel index of tap (0, 0) and a validity bit per tap;
  // the tap's offset is that index plus a wave-uniform delta (forward: +(ky W + kx); data gradient: -((ky >> s) W +
  // (kx >> s)) with s = stride - 1, because a valid tap has i = (o0 - k) / stride = (o0 >> s) - (k >> s)).
  const int dsh = (p.flags & SWEM_CONV_DGRAD) ? p.stride - 1 : 0;
  int pix0[WM];
  unsigned long long tmask[WM];
  // (round 5) the forward convolution's valid taps along an axis are the contiguous range  k in [max(0, -o0), min(K, lim - o0))
  // -- two bit ranges per row instead of KH + KW coordinate tests with their data-gradient branches (stamps: 2.8k of the 128x128
  // tile's 8.0k prologue cycles, 0.65k of the 64x64 tile's 4.0k); the generic loops serve the data gradient
  const bool fwd_taps = !(p.flags & SWEM_CONV_DGRAD) && p.KW < 32 && p.KH < 32;
#pragma unroll
  for (int j = 0; j < WM; ++j) {
    // bit (ky * KW + kx) = tap row ky valid AND tap column kx valid: KH + KW coordinate tests, not KH * KW
    unsigned long long mk = 0;
    if (fwd_taps) {
      auto range_bits = [](int o0, int K, int lim) __attribute__((always_inline)) {
        const int lo = o0 < 0 ? -o0 : 0, hi = lim - o0 < K ? lim - o0 : K;
        return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
      };
      const unsigned colv = range_bits(ix0[j], p.KW, p.W), rowv = range_bits(iy0[j], p.KH, p.H);
      for (int ky = 0; ky < p.KH; ++ky) mk |= ((rowv >> ky) & 1u) ? (unsigned long long)colv << (ky * p.KW) : 0ull;
    } else {
      unsigned colv = 0;
      for (int kx = 0; kx < p.KW; ++kx) {
        int ix;
        colv |= (tap_coord(p, ix0[j], kx, p.W, ix) ? 1u : 0u) << kx;
      }
      for (int ky = 0; ky < p.KH; ++ky) {
        int iy;
        if (tap_coord(p, iy0[j], ky, p.H, iy)) mk |= (unsigned long long)colv << (ky * p.KW);
      }
    }
    tmask[j] = bidx[j] >= 0 ? mk : 0ull;
  }
#ifdef SWEM_PROLOGUE_STAMPS
  STAMP(7);
#endif
  auto set_src = [&](const KPos &q) __attribute__((always_inline)) {
    const long long ps = q.src == 0 ? p.ps[0] : (q.src == 1 ? p.ps[1] : p.ps[2]);
    const int npx = q.src == 0 ? p.npx[0] : (q.src == 1 ? p.npx[1] : p.npx[2]);
    const int bsp = q.src == 0 ? p.bsp[0] : (q.src == 1 ? p.bsp[1] : p.bsp[2]);
    aplane = (unsigned)(ps * 2);
    agroup = (unsigned)npx * 16u;       // pixels in the plane * 16 bytes
    a_base = (unsigned)(q.ci0 / 8 + g) * agroup;
    rsa = src_rsrc(q.src);
#pragma unroll
    for (int j = 0; j < WM; ++j)
      pix0[j] = (bidx[j] < 0 ? 0 : bidx[j]) * bsp + (iy0[j] >> dsh) * p.W + (ix0[j] >> dsh);
  };
  auto set_tap = [&](const KPos &q) __attribute__((always_inline)) {
    const int t = q.ky * p.KW + q.kx;
    const int d = (q.ky >> dsh) * p.W + (q.kx >> dsh);
    const int delta = (p.flags & SWEM_CONV_DGRAD) ? -d : d;
#pragma unroll
    for (int j = 0; j < WM; ++j) avoff[j] = ((tmask[j] >> t) & 1ull) ? (unsigned)(pix0[j] + delta) * 16u : OOB;
  };
  int khalf = 0;   // KG == 2: which 16-channel half of the current (channel block, tap) cell
  auto advance = [&](KPos &q) __attribute__((always_inline)) {
    w_base += KG * wgroup;
    if (KG == 2) {
      khalf ^= 1;
      if (khalf) {                 // second half of the same cell: the same tap, the next two channel groups
        a_base += 2 * agroup;
        return;
      }
      a_base -= 2 * agroup;
    }
    if (++q.kx == p.KW) {
      q.kx = 0;
      if (++q.ky == p.KH) {
        q.ky = 0;
        q.ci0 += BK;
        a_base += 4 * agroup;
        const int cs = q.src == 0 ? p.c[0] : (q.src == 1 ? p.c[1] : p.c[2]);
        if (q.ci0 >= cs) {
          q.ci0 = 0;
          ++q.src;
          set_src(q);
        }
      }
    }
    set_tap(q);
  };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const unsigned lds_a = lds0 + (unsigned)g * (SA * 16), lds_b = lds0 + NST * NPL * PA * 16 + (unsigned)g * (SB * 16);
  auto issue = [&](int stage) __attribute__((always_inline)) {
    const unsigned sa = lds_a + (unsigned)stage * (NPL * PA * 16), sb = lds_b + (unsigned)stage * (NPL * PB * 16);
    if (NW == KG || half == 0) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < WM; ++j) dma16(rsa, sa + (pl * PA + j * 64) * 16, avoff[j], a_base + pl * aplane);
    }
    if (NW == KG || half == 1) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int j = 0; j < WN; ++j) dma16(rsw, sb + (pl * PB + j * 64) * 16, bvoff[j], w_base + pl * wplane);
    }
  };

  f32x16 acc[M16 ? 1 : TM][M16 ? 1 : TN];
  f32x4v acc16[M16 ? 2 * TM : 1][M16 ? 2 * TN : 1];
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc16[i][j][e] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  }

  const int kb_begin32 = sk ? seg_k0 : blockIdx.z * p.kb_per_split;
  const int kb_begin = kb_begin32 * (4 / KG);                       // in this kernel's k-blocks (16 or 32 k)
  // (the LAST split of a fused K-split tile -- its reducer -- takes whatever the others leave: the host may give them a
  // smaller share each, so that their partial tiles are in memory before the reducer is done with its own k-blocks)
  const int kb_end = (sk ? seg_k1 : ((int)blockIdx.z == (int)gridDim.z - 1 && p.sk_flags && p.partial)
                                        ? p.nkb : min(p.nkb, kb_begin32 + p.kb_per_split)) * (4 / KG);
  // Ring of NST stages.  In iteration kb the transfers of block kb+NST-1 are issued into the stage that was read in
  // iteration kb-1, the MFMAs run on stage kb%NST, then a COUNTED wait (all but the newest (NST-2)*NDMA transfers of
  // this wave, i.e. everything up to block kb+1) and a raw s_barrier publish stage (kb+1)%NST.  __syncthreads() would
  // drain vmcnt(0) and serialise the ~1-2 us L2/MALL -> LDS latency with every 0.3 us of MFMA work.
  KPos q;
  {   // k-block kb = (channel block, tap)
    const int taps = p.KH * p.KW;
    int cb = 0;
    q.ky = q.kx = 0;
    if (kb_begin32 != 0) {   // (a launch without K-split starts at k-block 0: no divisions)
      cb = kb_begin32 / taps;
      const int t = kb_begin32 - cb * taps;
      q.ky = t / p.KW;
      q.kx = t - q.ky * p.KW;
    }
    q.src = 0;
    int ci = cb * BK;
    if (ci >= p.c[0]) {
      ci -= p.c[0];
      q.src = 1;
      if (ci >= p.c[1]) {
        ci -= p.c[1];
        q.src = 2;
      }
    }
    q.ci0 = ci;
    set_src(q);
  }
  w_base = (unsigned)(kb_begin32 * 4 + g) * wgroup;
  set_tap(q);
  STAMP(1);
  // wait until at most `c` of this wave's k-block transfers (NDMA instructions each) are still in flight
  auto wait_blocks = [&](int c) __attribute__((always_inline)) {
    static_assert(3 * NDMA <= 63, "vmcnt is a 6-bit counter");
    if (c <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (c == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else if (c == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NDMA) : "memory");
  };
  // The transfer of block kb+NST-1 is issued at the top of iteration kb, behind the previous iteration's MFMAs (NST-1 stages in
  // flight).  (Issuing it right behind the hand-over instead -- all NST stages in flight -- measured SLOWER: 308 against 330
  // TFLOP/s on 2x120x216 256->256, 296 against 338 on 2x30x54 1280->512: the eight waves' address work and transfer requests
  // right behind the barrier hold up all their MFMA chains at once.  The experiment's diff: profiles/r06_experiments/.)
  if constexpr (PF) {
    // ---------------------------------------------------------------------------------------------------------------
    // Prologue: all NST stages requested; block 0's fragments into register set 0; block 1 landed and published.
    issue(0);
    int issued = 1;
#pragma unroll
    for (int d = 1; d < NST; ++d)
      if (kb_begin + d < kb_end) {
        advance(q);
        issue(d);
        ++issued;
      }
    wait_blocks(issued - 1);
    __builtin_amdgcn_s_barrier();
    STAMP(2);
    const int r16 = lane & 15, kgl = lane >> 4;   // tile row / column, k/8 group of this lane
    const uint4 *Ab0 = As + kgl * SA + wm * 32 * TM + r16;
    const uint4 *Bb0 = Bs + kgl * SB + wn * 32 * TN + r16;
    uint4 fa[2][NPL][2 * TM], fb[2][NPL][2 * TN];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int i = 0; i < 2 * TM; ++i) fa[0][pl][i] = Ab0[pl * PA + 16 * i];
#pragma unroll
      for (int i = 0; i < 2 * TN; ++i) fb[0][pl][i] = Bb0[pl * PB + 16 * i];
    }
    wait_blocks(issued - 2);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the builtin, so that hipcc's own wait insertion knows the reads are done
    __builtin_amdgcn_s_barrier();
    int st = 0;   // stage of block kb: free from the top of iteration kb on (its fragments are in registers)
    STAMP_ACC_DECL(t_vm);
    STAMP_ACC_DECL(t_bar);
    auto body = [&](auto CUR, int kb) __attribute__((always_inline)) {
      constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
      if (kb + NST < kb_end) {   // block kb+NST into the stage block kb occupied
        advance(q);
        issue(st);
      }
      const int stn = st == NST - 1 ? 0 : st + 1;   // stage of block kb+1: landed and published by the previous hand-over
      const bool more = kb + 1 < kb_end;
      const uint4 *Ab = Ab0 + stn * NPL * PA, *Bb = Bb0 + stn * NPL * PB;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2 * TM; ++i) {
        if (more) {   // this group's share of block kb+1's fragments: row tile i, and the column tiles dealt to it
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            fa[nxt][pl][i] = Ab[pl * PA + 16 * i];
#pragma unroll
            for (int j = 0; j < 2 * TN; ++j)
              if (j * (2 * TM) / (2 * TN) == i) fb[nxt][pl][j] = Bb[pl * PB + 16 * j];
          }
        }
#pragma unroll
        for (int jn = 0; jn < 2 * TN; ++jn) {
          f32x4v c = acc16[i][jn];
          if constexpr (NPL >= 2) {   // "bf16x3": the three products above 2^-16, smallest first
            c = mm16<F16>(fa[cur][0][i], fb[cur][1][jn], c);
            c = mm16<F16>(fa[cur][1][i], fb[cur][0][jn], c);
          }
          c = mm16<F16>(fa[cur][0][i], fb[cur][0][jn], c);
          acc16[i][jn] = c;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) {
        // hand-over: this wave's share of block kb+2 has landed (the younger blocks may stay in flight), its reads of block
        // kb+1's stage are done; the barrier publishes the one and frees the other
        const int last = min(kb + NST, kb_end - 1);   // youngest block requested so far
        STAMP_T0(t_vm);
        if constexpr (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else wait_blocks(last - (kb + 2));
        STAMP_ACC(t_vm);
        STAMP_T0(t_bar);
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the builtin, so that hipcc's own wait insertion knows the reads are done
        __builtin_amdgcn_s_barrier();
        STAMP_ACC(t_bar);
      }
      __builtin_amdgcn_sched_barrier(0);
      st = stn;
    };
    for (int kb = kb_begin; kb < kb_end; kb += 2) {
      body(std::integral_constant<int, 0>{}, kb);
      if (kb + 1 < kb_end) body(std::integral_constant<int, 1>{}, kb + 1);
    }
#ifndef SWEM_PROLOGUE_STAMPS
    STAMP_ACC_OUT(6, t_vm);
    STAMP_ACC_OUT(7, t_bar);
#endif
  } else {
    constexpr int NPRO = NST - 1;   // blocks issued before the loop
    issue(0);
    int issued = 1;   // k-blocks handed to the DMA so far (relative to kb_begin)
#pragma unroll
    for (int d = 1; d < NPRO; ++d)
      if (kb_begin + d < kb_end) {
        advance(q);
        issue(d);
        ++issued;
      }
    wait_blocks(issued - 1);   // block 0 has landed; the younger ones may stay in flight
    __builtin_amdgcn_s_barrier();
    STAMP(2);
    int st = 0;
    STAMP_ACC_DECL(t_vm);
    STAMP_ACC_DECL(t_bar);
    for (int kb = kb_begin; kb < kb_end; ++kb) {
      if (kb + NST - 1 < kb_end) {
        advance(q);
        issue(st == 0 ? NST - 1 : st - 1);  // stage (kb+NST-1) % NST
      }
      // Where the stage hand-over (counted wait for block kb+1 + s_barrier) sits inside the MFMA sequence: the MFMAs only
      // touch registers once the fragments are read, so any of them may run before or after it.  In front of it they would
      // hide the transfer this wave has just issued; behind it the waves of the block are decoupled while they compute (the
      // barrier does not wait for the slowest wave's MFMAs) and the next transfer starts earlier.  Measured on the config-B
      // layers (tools/conv_bench.py, tuned plans; none / half / all of the MFMAs in front): 330 / 334 / 310 TFLOP/s on
      // 2x120x216 256->256, 338 / 317 / 302 on 2x30x54 1280->512, 329 / 314 / 296 on 2x60x108 512->256 -- everything BEHIND the
      // hand-over wins: the other resident block's MFMAs hide the transfer, not this one's.  (The compiler's own scheduling had
      // arrived at nearly this order by sinking the MFMAs below the asm waits; it is now pinned by scheduling barriers.  The
      // two-waves-per-SIMD stagger and static-priority experiments of round 5 -- within +-1 % or 2-6 % slower -- are archived
      // with this one: profiles/r06_experiments/.)
      auto hand_over = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        // block kb+1 must have landed (this wave's share); the blocks behind it (up to kb+NST-1) may stay in flight
        const int last = min(kb + NST - 1, kb_end - 1);   // youngest block issued so far
        STAMP_T0(t_vm);
        wait_blocks(last - (kb + 1));
        STAMP_ACC(t_vm);
        STAMP_T0(t_bar);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads of stage st are done
        __builtin_amdgcn_s_barrier();
        STAMP_ACC(t_bar);
        __builtin_amdgcn_sched_barrier(0);
      };
      if constexpr (M16) {
        const int r16 = lane & 15, kg = lane >> 4;   // tile row / column, k/8 group of this lane
        const uint4 *Ab = As + st * NPL * PA + kg * SA + wm * 32 * TM + r16;
        const uint4 *Bb = Bs + st * NPL * PB + kg * SB + wn * 32 * TN + r16;
        uint4 a[NPL][2 * TM], b[NPL][2 * TN];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
          for (int i = 0; i < 2 * TM; ++i) a[pl][i] = Ab[pl * PA + 16 * i];
#pragma unroll
          for (int i = 0; i < 2 * TN; ++i) b[pl][i] = Bb[pl * PB + 16 * i];
        }
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) {
          if (i == 0) hand_over();
#pragma unroll
          for (int jn = 0; jn < 2 * TN; ++jn) {
            f32x4v c = acc16[i][jn];
            if constexpr (NPL == 3) {
              c = mm16<F16>(a[0][i], b[2][jn], c);
              c = mm16<F16>(a[2][i], b[0][jn], c);
              c = mm16<F16>(a[1][i], b[1][jn], c);
            }
            if constexpr (NPL >= 2) {   // NPL == 2: "bf16x3", the three products above 2^-16
              c = mm16<F16>(a[0][i], b[1][jn], c);
              c = mm16<F16>(a[1][i], b[0][jn], c);
            }
            c = mm16<F16>(a[0][i], b[0][jn], c);
            acc16[i][jn] = c;
          }
        }
      } else {
        const uint4 *Ab = As + st * NPL * PA + wm * 32 * TM + r;
        const uint4 *Bb = Bs + st * NPL * PB + wn * 32 * TN + r;
        uint4 a[KG / 2][NPL][TM], b[KG / 2][NPL][TN];
#pragma unroll
        for (int s2 = 0; s2 < KG / 2; ++s2) {
          const int k8 = 2 * s2 + h;
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[s2][pl][i] = Ab[pl * PA + k8 * SA + 32 * i];
#pragma unroll
            for (int i = 0; i < TN; ++i) b[s2][pl][i] = Bb[pl * PB + k8 * SB + 32 * i];
          }
        }
#pragma unroll
        for (int s2 = 0; s2 < KG / 2; ++s2) {
          if (s2 == 0) hand_over();
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
              f32x16 c = acc[i][jn];
              if constexpr (NPL == 3) {
                c = mm32<F16>(a[s2][0][i], b[s2][2][jn], c);
                c = mm32<F16>(a[s2][2][i], b[s2][0][jn], c);
                c = mm32<F16>(a[s2][1][i], b[s2][1][jn], c);
              }
              if constexpr (NPL >= 2) {
                c = mm32<F16>(a[s2][0][i], b[s2][1][jn], c);
                c = mm32<F16>(a[s2][1][i], b[s2][0][jn], c);
              }
              c = mm32<F16>(a[s2][0][i], b[s2][0][jn], c);
              acc[i][jn] = c;
            }
        }
      }
      st = st == NST - 1 ? 0 : st + 1;
    }
#ifndef SWEM_PROLOGUE_STAMPS
    STAMP_ACC_OUT(6, t_vm);
    STAMP_ACC_OUT(7, t_bar);
#endif
  }
  STAMP(3);
  if (sk && (seg_k0 > 0 || seg_k1 < p.nkb)) {
    // the accumulators as 16-byte chunks per lane, chunk c of every lane contiguous (coalesced): the layout of a partial tile
    // (chunk c = accumulator tile c of the 16x16 layout, or quarter c % 4 of 32x32 tile c / 4: compile-time register indices)
    constexpr int NCH = 4 * TM * TN;   // float4 chunks per lane (16 accumulators per 32x32 of the wave tile)
    constexpr unsigned TILE_BYTES = BM * BN * 4;
    if (seg_k0 > 0) {   // a tile another worker began: hand the partial sums over, then go on
      const __amdgpu_buffer_rsrc_t rp =
          __builtin_amdgcn_make_buffer_rsrc(p.sk_ws + (long long)sk_v * (BM * BN), 0, TILE_BYTES, 0x00020000);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        u32x4 v;
        if constexpr (M16) {
          const f32x4v t = acc16[c / (2 * TN)][c % (2 * TN)];
          v.x = __float_as_uint(t[0]); v.y = __float_as_uint(t[1]); v.z = __float_as_uint(t[2]); v.w = __float_as_uint(t[3]);
        } else {
          const f32x16 t = acc[(c / 4) / TN][(c / 4) % TN];
          v.x = __float_as_uint(t[4 * (c % 4)]); v.y = __float_as_uint(t[4 * (c % 4) + 1]);
          v.z = __float_as_uint(t[4 * (c % 4) + 2]); v.w = __float_as_uint(t[4 * (c % 4) + 3]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(v, rp, (unsigned)((c * 64 * NW + tid) * 16), 0, 0x11);   // sc0 sc1
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(p.sk_flags + sk_v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    // the head of a tile: add the partial sums of the workers that hold the rest of it, in their order
    const long long tile_end = (long long)(sk_tile + 1) * p.nkb;
    for (int u = sk_v + 1; u < p.sk_workers; ++u) {
      if (sk_total * u / p.sk_workers >= tile_end) break;
      if (tid == 0) {
        int spins = 0;
        unsigned seen = __hip_atomic_load(p.sk_flags + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (seen == 0u && spins < p.spin_limit) {
          __builtin_amdgcn_s_sleep(16);
          ++spins;
          seen = __hip_atomic_load(p.sk_flags + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (seen == 0u && p.fault)   // the producer never arrived: what follows adds a tile that was not written
          __hip_atomic_fetch_or(p.fault, (unsigned)SWEM_FAULT_STREAMK_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (tid == 0) __hip_atomic_store(p.sk_flags + u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed once: leave it zero
      const __amdgpu_buffer_rsrc_t rp =
          __builtin_amdgcn_make_buffer_rsrc(p.sk_ws + (long long)u * (BM * BN), 0, TILE_BYTES, 0x00020000);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rp, (unsigned)((c * 64 * NW + tid) * 16), 0, 0x11);
        if constexpr (M16) {
          f32x4v t = acc16[c / (2 * TN)][c % (2 * TN)];
          t[0] += __uint_as_float(v.x); t[1] += __uint_as_float(v.y); t[2] += __uint_as_float(v.z); t[3] += __uint_as_float(v.w);
          acc16[c / (2 * TN)][c % (2 * TN)] = t;
        } else {
          f32x16 t = acc[(c / 4) / TN][(c / 4) % TN];
          t[4 * (c % 4)] += __uint_as_float(v.x); t[4 * (c % 4) + 1] += __uint_as_float(v.y);
          t[4 * (c % 4) + 2] += __uint_as_float(v.z); t[4 * (c % 4) + 3] += __uint_as_float(v.w);
          acc[(c / 4) / TN][(c / 4) % TN] = t;
        }
      }
    }
  }
  // ---- K-split without a reduce launch (round 3: p.sk_flags set, gridDim.z splits).  The LAST split (z = gridDim.z - 1) of a
  // tile is its reducer: the others store their partial tile past the caches (the stream-K layout: 16-byte chunks per lane),
  // drain, and count themselves on the tile's counter; the reducer keeps its own sums in registers, waits until the counter
  // says the others' tiles are in memory, adds them in z order (own + p0 + p1 + ...: a fixed order -- deterministic) and runs
  // the epilogue.  The reducer only ever waits for blocks dispatched BEFORE it (z is the slowest grid dimension), which are
  // resident or finished and wait for nobody: no deadlock whatever the grid size.  Against "the last arriver reduces"
  // (this round's first form) a tile's own sums never travel: 2 (nsp - 1) instead of 2 nsp tile moves -- the splits of a
  // layer finish together, so the stores and loads of the partials are a burst the matrix pipe cannot hide (13 of the 54 us
  // of `2x30x54 k3 512->512` at nsp = 4).  Replaces conv_splitk_epilogue_kernel for this kernel.
  bool fused_last = false;
  if constexpr (!SK) {
    if (p.partial && p.sk_flags) {
      constexpr int NCH = 4 * TM * TN;
      constexpr unsigned TILE_BYTES = BM * BN * 4;
      const int nsp = (int)gridDim.z, tile_lin = (tm - p.mt0) * (int)gridDim.y + tn;
      float *slots = p.partial + (long long)tile_lin * nsp * (BM * BN);
      if ((int)blockIdx.z != nsp - 1) {
        const __amdgpu_buffer_rsrc_t rp =
            __builtin_amdgcn_make_buffer_rsrc(slots + (long long)blockIdx.z * (BM * BN), 0, TILE_BYTES, 0x00020000);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          u32x4 v;
          if constexpr (M16) {
            const f32x4v t = acc16[c / (2 * TN)][c % (2 * TN)];
            v.x = __float_as_uint(t[0]); v.y = __float_as_uint(t[1]); v.z = __float_as_uint(t[2]); v.w = __float_as_uint(t[3]);
          } else {
            const f32x16 t = acc[(c / 4) / TN][(c / 4) % TN];
            v.x = __float_as_uint(t[4 * (c % 4)]); v.y = __float_as_uint(t[4 * (c % 4) + 1]);
            v.z = __float_as_uint(t[4 * (c % 4) + 2]); v.w = __float_as_uint(t[4 * (c % 4) + 3]);
          }
          __builtin_amdgcn_raw_buffer_store_b128(v, rp, (unsigned)((c * 64 * NW + tid) * 16), 0, 0x11);   // sc0 sc1
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.sk_flags + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;   // not the reducer: done
      }
      fused_last = true;
      if (tid == 0) {
        // (a read-modify-write of zero: executed where the other splits' increments are, never served from a cache)
        // (>= and a bounded spin: a producer that is never dispatched -- XCDs advance their dispatch independently, so with
        // other streams' grids in flight the "dispatched before me" order is not a guarantee -- or counters left
        // non-zero by an aborted launch end in a wrong tile AND a set fault word, never in a hung queue)
        int spin = 0;
        unsigned seen = __hip_atomic_fetch_add(p.sk_flags + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (seen < (unsigned)(nsp - 1) && spin < p.spin_limit) {
          __builtin_amdgcn_s_sleep(8);
          ++spin;
          seen = __hip_atomic_fetch_add(p.sk_flags + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (seen < (unsigned)(nsp - 1) && p.fault)
          __hip_atomic_fetch_or(p.fault, (unsigned)SWEM_FAULT_KSPLIT_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.sk_flags + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // leave it zero
      }
      __syncthreads();
      for (int zz = 0; zz < nsp - 1; ++zz) {
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(slots + (long long)zz * (BM * BN), 0, TILE_BYTES, 0x00020000);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rp, (unsigned)((c * 64 * NW + tid) * 16), 0, 0x11);
          if constexpr (M16) {
            f32x4v t = acc16[c / (2 * TN)][c % (2 * TN)];
            t[0] += __uint_as_float(v.x); t[1] += __uint_as_float(v.y); t[2] += __uint_as_float(v.z); t[3] += __uint_as_float(v.w);
            acc16[c / (2 * TN)][c % (2 * TN)] = t;
          } else {
            f32x16 t = acc[(c / 4) / TN][(c / 4) % TN];
            t[4 * (c % 4)] += __uint_as_float(v.x); t[4 * (c % 4) + 1] += __uint_as_float(v.y);
            t[4 * (c % 4) + 2] += __uint_as_float(v.z); t[4 * (c % 4) + 3] += __uint_as_float(v.w);
            acc[(c / 4) / TN][(c / 4) % TN] = t;
          }
        }
      }
    }
  }
  if (fused_last) {   // the epilogue proper (the parameters' `partial` field only selects the raw-partials path)
    ConvP q = p;
    q.partial = nullptr;
    if constexpr (M16) conv_epilogue16<2 * TM, 2 * TN>(q, acc16, m0 + wm * 32 * TM, n0 + wn * 32 * TN, lane);
    else conv_epilogue<TM, TN>(q, acc, m0, n0, wm, wn, r, h);
  } else {
    if constexpr (M16) conv_epilogue16<2 * TM, 2 * TN>(p, acc16, m0 + wm * 32 * TM, n0 + wn * 32 * TN, lane);
    else conv_epilogue<TM, TN>(p, acc, m0, n0, wm, wn, r, h);
  }
  STAMP(4);
  if (!sk) break;
  }   // segments
#ifdef SWEM_EM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(5);
#endif
}

// Epilogue of conv_t256_kernel (round 6).  conv_epilogue16 inlines one copy of out_tile32 -- residual / mask / ReLU / fp32 map / two
// plane variants, all runtime flags -- per 32 x 32 sub-tile of the wave's accumulators, because the accumulators are registers and
// a register array only takes compile-time indices: with this kernel's 128 accumulator registers that was 8 (+ 4 for the GLU form)
// unrolled copies, 294 KB of code for conv_t256_kernel<true, 0>, 68 spilled registers and 72 scratch instructions in the epilogue
// of the one-block-per-CU kernel whose measured weakness IS its epilogue.  Here the accumulators leave the
// registers first: a pass stages up to four scaled sub-tiles (static register indices, ~70 instructions each) into the wave's
// slice of the idle operand LDS (18 KB per wave), and ONE out_tile32 in a loop that is not unrolled takes them out -- the LDS
// address is the only thing that depends on the loop counter.  Same values, same stores, same order per sub-tile.
constexpr int T256_EPI_SUB = 4;                                  // sub-tiles staged per pass
constexpr int T256_EPI_BYTES = T256_EPI_SUB * PL_BYTES;          // LDS bytes per wave (x 8 waves = 144 KB <= the block's 144 KB)
template <int TM2, int TN2>
__device__ __forceinline__ void t256_epilogue(const ConvP &p, f32x4v (&acc)[TM2][TN2], int mrow0, int ncol0, int lane, char *smem) {
  static_assert(TM2 % 2 == 0 && TN2 % 2 == 0, "the output staging works on 32 x 32 sub-tiles");
  const int col = lane & 15, rg = lane >> 4;
  if (p.partial) {   // (a K-split launch whose partial tiles a separate kernel reduces: raw sums, as conv_epilogue16)
    float *dst = p.partial + (long long)blockIdx.z * (p.M - p.part_m0) * p.Ncols - (long long)p.part_m0 * p.Ncols;
#pragma unroll
    for (int i = 0; i < TM2; ++i)
#pragma unroll
      for (int j = 0; j < TN2; ++j) {
        const int n = ncol0 + 16 * j + col;
        if (n >= p.Ncols) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = mrow0 + 16 * i + 4 * rg + e;
          if (m < p.M) dst[(long long)m * p.Ncols + n] = acc[i][j][e];
        }
      }
    return;
  }
  constexpr int NI = TM2 / 2, NJ = TN2 / 2;
  constexpr int SUBW = 32 * PL_STRIDE;                           // dwords per staged sub-tile
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned *lds = reinterpret_cast<unsigned *>(smem + wave * T256_EPI_BYTES);
  const bool glu = TN2 == 4 && (p.flags & SWEM_CONV_GLU);
  // per-column scale / shift of this lane's columns (GLU: tiles 0, 1 are f, tiles 2, 3 the gates of the same 32 channels)
  float sc[TN2], sh[TN2];
  const bool gin = ncol0 + 64 <= p.Ncols;
#pragma unroll
  for (int j = 0; j < TN2; ++j) {
    const int n = ncol0 + 16 * j + col;
    const bool nin = glu ? gin : n < p.Ncols;
    sc[j] = (nin && p.scale) ? p.scale[n] : p.uscale;
    sh[j] = (nin && p.shift) ? p.shift[n] : 0.f;
  }
  __syncthreads();   // every wave is done with the operand stages: the LDS is free for the output staging
  const int nsub = glu ? NI : NI * NJ;                            // sub-tile s: rows 32 (s % NI), column pair s / NI
  // The passes are spelled out (the accumulators a pass stages are dead after it: only the later passes' registers stay live
  // across the output loop); the output loop inside a pass is NOT unrolled.
  constexpr int NPASS = (NI * NJ + T256_EPI_SUB - 1) / T256_EPI_SUB;
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    const int s0 = pass * T256_EPI_SUB;
    // ---- stage sub-tiles s0 .. s0 + 3 (compile-time register indices)
#pragma unroll
    for (int kk = 0; kk < T256_EPI_SUB; ++kk) {
      const int s = s0 + kk;
      if (s >= NI * NJ || s >= nsub) continue;
      const int i0 = 2 * (s % NI), j0 = 2 * (s / NI);
      unsigned *dst = lds + kk * SUBW;
      if (glu) {
        if constexpr (TN2 == 4) {
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                dst[(16 * ii + 4 * rg + e) * PL_STRIDE + 16 * j + col] = __float_as_uint(
                    (acc[i0 + ii][j][e] * sc[j] + sh[j]) * sigmoidf_(acc[i0 + ii][j + 2][e] * sc[j + 2] + sh[j + 2]));
        }
      } else {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              dst[(16 * ii + 4 * rg + e) * PL_STRIDE + 16 * jj + col] =
                  __float_as_uint(acc[i0 + ii][j0 + jj][e] * sc[j0 + jj] + sh[j0 + jj]);
      }
    }
    // ---- ... and take them out: one out_tile32 per pass, the sub-tile is a runtime index
    const int cnt = min(T256_EPI_SUB, nsub - s0);      // (<= 0 for the second pass of the GLU form: NI sub-tiles in all)
#pragma nounroll
    for (int k = 0; k < cnt; ++k) {
      const int s = s0 + k;
      const int si = s % NI, sj = s / NI;
      out_tile32(p, lds + k * SUBW, mrow0 + 32 * si, glu ? (gin ? (ncol0 >> 1) : p.Cout) : ncol0 + 32 * sj);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// "t256" (round 5): the f16x3 convolution on a 256 x 256 tile, eight waves, ONE block per CU.
// The kernel above is the "128x128 tile, one hand-over per k-block" structure, whose ceiling on this chip is ~900 TFLOP/s of
// MFMA work (cdna_hip_programming.md, "The step-3 structure's ~900 TF ceiling"; this library's f16x3 layers run at 3 x 290-370
// = 870-1100): every k-block a wave requests the next transfers, reads 12 fragments, waits, meets the block at a barrier and
// only then starts 24 MFMAs -- what hides that sequence is the CU's OTHER resident block.  The structure that gets past it keeps
// ONE block per CU busy by itself: a wave tile of 128 x 64 (4 MFMAs per fragment read instead of 2), the transfers of k-block
// kb + 1 requested a full k-block (96 MFMAs per wave, ~3k cycles per SIMD) before anything waits for them, and the fragments of
// the next quarter of the wave tile read from the LDS while the MFMAs of the current quarter run (two fragment register sets).
//   per k-block (32 k = one filter tap x 32 channels, hi + mid planes; stage = kb & 1, 64 KB per stage) eight steps over the
//   eighths (32 rows x 32 columns) of the wave tile in snake order -- consecutive eighths share their A pair or their B pair:
//     steps 0, 1: request the transfers of k-block kb + 1 into the other stage (eight 1 KB transfers per wave);
//     every step: read the ONE new fragment pair of the next step (four ds_read_b128), then the twelve MFMAs of this step
//                 on fragments read one step earlier (two A and two B register pairs, double-buffered: 64 registers);
//     step 7:     vmcnt(0) (requested six steps ago), lgkmcnt(0), ONE s_barrier -- it publishes k-block kb + 1's stage and
//                 releases this one for the transfers of kb + 2 -- read the first pairs of k-block kb + 1; MFMAs of step 7
// Same k order, same three products per (A, B) fragment pair, same epilogue (conv_epilogue16) as the kernel above; the
// accumulation order over k is the same, so the results are bit-identical to it.  LDS image per operand and stage:
// [plane][k/8 group][256 rows][16 bytes] (fragment reads conflict-free as above: the group stride is a multiple of 16 slots).
// Each wave transfers ONE 64-row run (run = wave & 3, plane = wave >> 2) of every k/8 group of both operands: one pixel decode
// and one tap mask per lane.  128 accumulator + 64 fragment registers per lane, two waves per SIMD.
// Preconditions (host): fp16 pairs (SWEM_PLAN_F16), no per-batch filters, plan tile 4 x 4, no tail split / stream-K.
// TMW = 0: the layout above (wave grid 2 x 4, BM = 256).  TMW = 8, 10, 12, 14: wave grid 1 x 8 -- every wave owns ALL BM = 16 TMW
// rows x 32 columns -- so that the tile HEIGHT can be chosen per layer: with one block per CU a launch is whole rounds of 256
// tiles, and 203 tiles (2x120x216 rows / 256) leave a fifth of the chip idle where 232 tiles of 224 rows fill it (plan bits
// 20-23 = TMW / 2).  Its k-block: the B pair once, then TMW / 2 steps of one A pair (read one step ahead) x twelve MFMAs.
template <bool F16, int TMW = 0>
__global__ __launch_bounds__(512, 1) void conv_t256_kernel(ConvP p STAMP_ARG) {
  STAMP(0);
  static_assert(TMW == 0 || (TMW >= 4 && TMW <= 14 && TMW % 2 == 0), "1 x 8 layout: an even number of 16-row tiles, below 256 rows");
  // F16 = false: the bf16x6 arithmetic (three bf16 planes per operand, six products: all 24 operand bits) -- 128-row tiles only
  // (TMW = 8): (128 + 256) rows x 64 bytes x 3 planes = 72 KB per stage.
  static_assert(F16 || TMW == 8, "three planes fit the LDS with 128-row tiles only");
  constexpr int NPL = F16 ? 2 : 3, KG = 4, BM = TMW ? 16 * TMW : 256, BN = 256, NW = 8;
  constexpr int NTM = TMW ? TMW : 8, NTN = TMW ? 2 : 4;   // accumulator tiles of a wave
  constexpr int AROWS = F16 ? 256 : BM;        // rows of the A image (fp16 pairs: 256 allocated, BM of them used)
  constexpr int PAA = KG * AROWS, PA = KG * 256;   // 16-byte slots per plane of the A / B image
  constexpr unsigned OPA = NPL * PAA * 16, OPB = NPL * PA * 16;   // bytes of the A / B image of one stage
  constexpr unsigned STAGE = OPA + OPB;        // A image, then B image (fp16 pairs: 64 KB)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // wave tile: rows wr * 128 + [0, 128), columns wc * 64 + [0, 64)  (TMW: all rows, columns wave * 32 + [0, 32))
  const int wr = TMW ? 0 : wave >> 2, wc = TMW ? wave : wave & 3;
  const int r16 = lane & 15, kgl = lane >> 4;
  int tm, tn;
  tile_coords(tm, tn, p.xpn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int dj = wave & 3, dpl = wave >> 2;    // the 64-row run and the plane this wave transfers
  const bool a_run = dj * 64 < BM;             // (a tile lower than 256 rows has fewer runs; rows past BM of the last one are
                                               // the next tile's pixels: transferred, never read)

  auto src_rsrc = [&](int sidx) __attribute__((always_inline)) {
    const unsigned short *base = sidx == 0 ? p.xs[0] : (sidx == 1 ? p.xs[1] : p.xs[2]);
    const long long ps = sidx == 0 ? p.ps[0] : (sidx == 1 ? p.ps[1] : p.ps[2]);
    return raw_rsrc(base, (unsigned)(3 * ps * 2));
  };
  const i32x4 rsw = raw_rsrc(p.wsplit, (unsigned)((long long)3 * p.Ncols * p.K * 2));
  const unsigned wplane = (unsigned)((long long)p.Ncols * p.K * 2);
  const unsigned wgroup = (unsigned)p.Ncols * 16u;
  // ---- the pixel this lane transfers (A image, row dj * 64 + lane) and its valid taps; the filter it transfers (B image)
  int iy0, ix0, bidx;
  {
    const int m = m0 + dj * 64 + lane;
    const int b = fast_div(m, p.fd_howo_mul, p.fd_howo_sh);
    const int rem = m - b * (p.Ho * p.Wo);
    const int oy = fast_div(rem, p.fd_wo_mul, p.fd_wo_sh), ox = rem - oy * p.Wo;
    iy0 = tap_origin(p, oy);
    ix0 = tap_origin(p, ox);
    bidx = m < p.M ? b : -1;
  }
  unsigned bvoff;
  {
    const int n = n0 + dj * 64 + lane;
    bvoff = n < p.Ncols ? (unsigned)n * 16u : OOB;
  }
  const int dsh = (p.flags & SWEM_CONV_DGRAD) ? p.stride - 1 : 0;
  unsigned long long tmask = 0;
  {
    unsigned colv = 0;
    for (int kx = 0; kx < p.KW; ++kx) {
      int ix;
      colv |= (tap_coord(p, ix0, kx, p.W, ix) ? 1u : 0u) << kx;
    }
    for (int ky = 0; ky < p.KH; ++ky) {
      int iy;
      if (tap_coord(p, iy0, ky, p.H, iy)) tmask |= (unsigned long long)colv << (ky * p.KW);
    }
    if (bidx < 0) tmask = 0ull;
  }
  unsigned avoff = OOB, aplane = 0, agroup = 0, a_base = 0, w_base = 0;
  int pix0 = 0;
  i32x4 rsa;
  auto set_src = [&](const KPos &q) __attribute__((always_inline)) {
    const long long ps = q.src == 0 ? p.ps[0] : (q.src == 1 ? p.ps[1] : p.ps[2]);
    const int npx = q.src == 0 ? p.npx[0] : (q.src == 1 ? p.npx[1] : p.npx[2]);
    const int bsp = q.src == 0 ? p.bsp[0] : (q.src == 1 ? p.bsp[1] : p.bsp[2]);
    aplane = (unsigned)(ps * 2);
    agroup = (unsigned)npx * 16u;
    a_base = (unsigned)(q.ci0 / 8) * agroup + (F16 ? (unsigned)dpl * aplane : 0u);
    rsa = src_rsrc(q.src);
    pix0 = (bidx < 0 ? 0 : bidx) * bsp + (iy0 >> dsh) * p.W + (ix0 >> dsh);
  };
  auto set_tap = [&](const KPos &q) __attribute__((always_inline)) {
    const int t = q.ky * p.KW + q.kx;
    const int d = (q.ky >> dsh) * p.W + (q.kx >> dsh);
    const int delta = (p.flags & SWEM_CONV_DGRAD) ? -d : d;
    avoff = ((tmask >> t) & 1ull) ? (unsigned)(pix0 + delta) * 16u : OOB;
  };
  auto advance = [&](KPos &q) __attribute__((always_inline)) {
    w_base += KG * wgroup;
    if (++q.kx == p.KW) {
      q.kx = 0;
      if (++q.ky == p.KH) {
        q.ky = 0;
        q.ci0 += BK;
        a_base += 4 * agroup;
        const int cs = q.src == 0 ? p.c[0] : (q.src == 1 ? p.c[1] : p.c[2]);
        if (q.ci0 >= cs) {
          q.ci0 = 0;
          ++q.src;
          set_src(q);
        }
      }
    }
    set_tap(q);
  };
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  // fp16 pairs: wave (dj, dpl) moves the four k/8 groups of plane dpl of its run.  Three planes: the twelve (plane, group) cells of
  // a run are shared by the two waves (dj, 0 / 1): cell index = 2 i + (wave >> 2), i < 6.
  const unsigned lds_a = lds0 + (unsigned)((F16 ? dpl * PAA : 0) + dj * 64) * 16u;
  const unsigned lds_b = lds0 + OPA + (unsigned)((F16 ? dpl * PA : 0) + dj * 64) * 16u;
  auto issue_a = [&](int stage) __attribute__((always_inline)) {
    const unsigned sa = lds_a + (unsigned)stage * STAGE;
    if (a_run) {
      if constexpr (F16) {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) dma16(rsa, sa + (unsigned)kg * (AROWS * 16), avoff, a_base + (unsigned)kg * agroup);
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int cell = 2 * i + dpl, pl = cell >> 2, kg = cell & 3;
          dma16(rsa, sa + (unsigned)(pl * PAA + kg * AROWS) * 16u, avoff, a_base + (unsigned)kg * agroup + (unsigned)pl * aplane);
        }
      }
    }
  };
  auto issue_b = [&](int stage) __attribute__((always_inline)) {
    const unsigned sb = lds_b + (unsigned)stage * STAGE;
    if constexpr (F16) {
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) dma16(rsw, sb + (unsigned)kg * (BN * 16), bvoff, w_base + (unsigned)kg * wgroup);
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int cell = 2 * i + dpl, pl = cell >> 2, kg = cell & 3;
        dma16(rsw, sb + (unsigned)(pl * PA + kg * BN) * 16u, bvoff, w_base + (unsigned)kg * wgroup + (unsigned)pl * wplane);
      }
    }
  };

  f32x4v acc16[NTM][NTN];
#pragma unroll
  for (int i = 0; i < NTM; ++i)
#pragma unroll
    for (int j = 0; j < NTN; ++j) acc16[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  const int kb_begin = blockIdx.z * p.kb_per_split;
  const int kb_end = ((int)blockIdx.z == (int)gridDim.z - 1 && p.sk_flags && p.partial) ? p.nkb : min(p.nkb, kb_begin + p.kb_per_split);
  KPos q;
  {
    const int taps = p.KH * p.KW;
    const int cb = kb_begin / taps, t = kb_begin - cb * taps;
    q.ky = t / p.KW;
    q.kx = t - q.ky * p.KW;
    q.src = 0;
    int ci = cb * BK;
    if (ci >= p.c[0]) {
      ci -= p.c[0];
      q.src = 1;
      if (ci >= p.c[1]) {
        ci -= p.c[1];
        q.src = 2;
      }
    }
    q.ci0 = ci;
    set_src(q);
  }
  w_base = (unsigned)(kb_begin * 4) * wgroup + (F16 ? (unsigned)dpl * wplane : 0u);
  set_tap(q);

  // fragment addresses: the lane's slot inside a plane of the A / B image (tile 0 of the wave, k/8 group kgl)
  const uint4 *As = reinterpret_cast<const uint4 *>(smem) + kgl * AROWS + wr * 128 + r16;
  const uint4 *Bs = reinterpret_cast<const uint4 *>(smem + OPA) + kgl * BN + (TMW ? wc * 32 : wc * 64) + r16;
  constexpr int SSL = STAGE / 16;              // slots per stage
  // eighth (AM, BN) of the wave tile: rows AM * 32 + [0, 32) (two A tiles), columns BN * 32 + [0, 32) (two B tiles)
  auto load_a = [&](int stage, int am, uint4 (&f)[NPL][2]) __attribute__((always_inline)) {
    const uint4 *a = As + stage * SSL + am * 32;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i) f[pl][i] = a[pl * PAA + 16 * i];
  };
  auto load_b = [&](int stage, int bn, uint4 (&f)[NPL][2]) __attribute__((always_inline)) {
    const uint4 *b = Bs + stage * SSL + bn * 32;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int j = 0; j < 2; ++j) f[pl][j] = b[pl * PA + 16 * j];
  };
#define T256_MFMA(AM, BNQ, FA, FB)                                                                          \
  {                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)            \
        acc16[AM * 2 + i][BNQ * 2 + j] = mm16<F16>(FA[0][i], FB[1][j], acc16[AM * 2 + i][BNQ * 2 + j]);     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)            \
        acc16[AM * 2 + i][BNQ * 2 + j] = mm16<F16>(FA[1][i], FB[0][j], acc16[AM * 2 + i][BNQ * 2 + j]);     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)            \
        acc16[AM * 2 + i][BNQ * 2 + j] = mm16<F16>(FA[0][i], FB[0][j], acc16[AM * 2 + i][BNQ * 2 + j]);     \
    __builtin_amdgcn_s_setprio(0);                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  }
  // One k-block = eight steps over the wave tile's eighths in snake order, so that consecutive steps share the A pair or the B
  // pair: every step reads ONE new pair (four ds_read_b128) for the next step while its twelve MFMAs run.  BX holds the B pair
  // of eighth column 0 on entry; the next k-block's first B pair lands in BY (the roles swap from k-block to k-block).
#define T256_KBLOCK(BX, BY)                                                                                 \
  {                                                                                                         \
    const bool more = kb + 1 < kb_end;                                                                      \
    if (more) {                                                                                             \
      advance(q);                                                                                           \
      issue_a(st ^ 1);                                                                                      \
    }                                                                                                       \
    load_b(st, 1, BY);                                                                                      \
    T256_MFMA(0, 0, a0, BX);                                                                                \
    if (more) issue_b(st ^ 1);                                                                              \
    load_a(st, 1, a1);                                                                                      \
    T256_MFMA(0, 1, a0, BY);                                                                                \
    load_b(st, 0, BX);                                                                                      \
    T256_MFMA(1, 1, a1, BY);                                                                                \
    load_a(st, 2, a0);                                                                                      \
    T256_MFMA(1, 0, a1, BX);                                                                                \
    load_b(st, 1, BY);                                                                                      \
    T256_MFMA(2, 0, a0, BX);                                                                                \
    load_a(st, 3, a1);                                                                                      \
    T256_MFMA(2, 1, a0, BY);                                                                                \
    load_b(st, 0, BX);                                                                                      \
    T256_MFMA(3, 1, a1, BY);                                                                                \
    /* hand-over: k-block kb + 1 has landed (requested seven steps ago), this wave's reads of stage st are done */ \
    STAMP_T0(t_vm);                                                                                         \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
    STAMP_ACC(t_vm);                                                                                        \
    STAMP_T0(t_bar);                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    __builtin_amdgcn_s_barrier();                                                                           \
    STAMP_ACC(t_bar);                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (more) {                                                                                             \
      load_a(st ^ 1, 0, a0);                                                                                \
      load_b(st ^ 1, 0, BY);                                                                                \
    }                                                                                                       \
    T256_MFMA(3, 0, a1, BX);                                                                                \
    st ^= 1;                                                                                                \
  }
  STAMP(1);
  issue_a(0);
  issue_b(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  STAMP(2);
  STAMP_ACC_DECL(t_vm);
  STAMP_ACC_DECL(t_bar);
  int st = 0;
  if constexpr (TMW == 0) {
    uint4 a0[NPL][2], a1[NPL][2], b0[NPL][2], b1[NPL][2];
    load_a(0, 0, a0);
    load_b(0, 0, b0);
    for (int kb = kb_begin; kb < kb_end; ++kb) {
      T256_KBLOCK(b0, b1);
      if (++kb >= kb_end) break;
      T256_KBLOCK(b1, b0);
    }
  } else {
    // 1 x 8 layout: fa[(c + PA_) & 1] holds the A pair of step c (rows 32 c + [0, 32)), fb[PB_] this k-block's B pair.  The
    // buffer parities are compile-time constants (registers, not scratch): the k-loop is unrolled by two k-blocks.
    constexpr int NCH = TMW / 2;
    // (A half-k-block stagger of the two waves that share a SIMD -- waves 4-7 behind one extra barrier, two meetings per k-block
    // -- measured 1-5 % SLOWER on every layer, profiles/r05_kernel_experiments.txt section 8: the ~750 cycles per k-block outside
    // the MFMAs are not an idle SIMD at the hand-over, and the stagger halves the transfers' lead.  Diff: profiles/r06_experiments/.)
    uint4 fa[2][NPL][2], fb[2][NPL][2];
    load_a(0, 0, fa[0]);
    load_b(0, 0, fb[0]);
    int kb = kb_begin;
    auto kblock = [&](auto pa_, auto pb_) __attribute__((always_inline)) {
      constexpr int PA_ = decltype(pa_)::value, PB_ = decltype(pb_)::value;
      const bool more = kb + 1 < kb_end;
      if (more) {
        advance(q);
        issue_a(st ^ 1);
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c == 1 && more) issue_b(st ^ 1);
        if (c + 1 < NCH) {
          load_a(st, c + 1, fa[(c + 1 + PA_) & 1]);
        } else {
          // hand-over: k-block kb + 1 has landed (requested NCH - 1 steps ago), this wave's reads of stage st are done
          STAMP_T0(t_vm);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          STAMP_ACC(t_vm);
          STAMP_T0(t_bar);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          STAMP_ACC(t_bar);
          __builtin_amdgcn_sched_barrier(0);
          if (more) {
            load_a(st ^ 1, 0, fa[(NCH + PA_) & 1]);
            load_b(st ^ 1, 0, fb[PB_ ^ 1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        // products in the order of conv_igemm_bf3s_kernel (per accumulator): fp16 pairs a0 b1, a1 b0, a0 b0; three planes
        // a0 b2, a2 b0, a1 b1, a0 b1, a1 b0, a0 b0
        constexpr int NPROD = F16 ? 3 : 6;
        constexpr int PAI[6] = {0, 2, 1, 0, 1, 0}, PBI[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int prod = 0; prod < NPROD; ++prod)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc16[2 * c + i][j] = mm16<F16>(fa[(c + PA_) & 1][PAI[prod + 6 - NPROD]][i], fb[PB_][PBI[prod + 6 - NPROD]][j],
                                              acc16[2 * c + i][j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      }
      st ^= 1;
    };
    for (; kb < kb_end; ++kb) {
      kblock(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      if (++kb >= kb_end) break;
      kblock(std::integral_constant<int, NCH & 1>{}, std::integral_constant<int, 1>{});
    }
  }
#undef T256_KBLOCK
#undef T256_MFMA
  STAMP(3);
  STAMP_ACC_OUT(6, t_vm);
  STAMP_ACC_OUT(7, t_bar);
  // ---- K-split reduced inside the launch: conv_igemm_bf3s_kernel's scheme (the last split of a tile is its reducer)
  bool fused_last = false;
  if (p.partial && p.sk_flags) {
    constexpr int NCH = NTM * NTN;   // float4 chunks per lane: the wave's accumulator tiles
    constexpr unsigned TILE_BYTES = BM * BN * 4;
    const int nsp = (int)gridDim.z, tile_lin = tm * (int)gridDim.y + tn;
    float *slots = p.partial + (long long)tile_lin * nsp * (BM * BN);
    if ((int)blockIdx.z != nsp - 1) {
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(slots + (long long)blockIdx.z * (BM * BN), 0, TILE_BYTES, 0x00020000);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const f32x4v t = acc16[c / NTN][c % NTN];
        u32x4 v;
        v.x = __float_as_uint(t[0]); v.y = __float_as_uint(t[1]); v.z = __float_as_uint(t[2]); v.w = __float_as_uint(t[3]);
        __builtin_amdgcn_raw_buffer_store_b128(v, rp, (unsigned)((c * 64 * NW + tid) * 16), 0, 0x11);   // sc0 sc1
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(p.sk_flags + tile_lin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    fused_last = true;
    if (tid == 0) {
      int spin = 0;
      unsigned seen = __hip_atomic_fetch_add(p.sk_flags + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (seen < (unsigned)(nsp - 1) && spin < p.spin_limit) {
        __builtin_amdgcn_s_sleep(8);
        ++spin;
        seen = __hip_atomic_fetch_add(p.sk_flags + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (seen < (unsigned)(nsp - 1) && p.fault)
        __hip_atomic_fetch_or(p.fault, (unsigned)SWEM_FAULT_KSPLIT_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.sk_flags + tile_lin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int zz = 0; zz < nsp - 1; ++zz) {
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(slots + (long long)zz * (BM * BN), 0, TILE_BYTES, 0x00020000);
      // (sixteen 16-byte loads in flight per lane, not all NCH: beside 128 accumulator registers more would spill; fewer would
      // make the reducer's tail a chain of load latencies)
      constexpr int RCH = 16;
#pragma unroll
      for (int c0 = 0; c0 < NCH; c0 += RCH) {
        u32x4 v[RCH];
#pragma unroll
        for (int cc = 0; cc < RCH; ++cc)
          if (c0 + cc < NCH) v[cc] = __builtin_amdgcn_raw_buffer_load_b128(rp, (unsigned)(((c0 + cc) * 64 * NW + tid) * 16), 0, 0x11);
#pragma unroll
        for (int cc = 0; cc < RCH; ++cc)
          if (c0 + cc < NCH) {
            f32x4v t = acc16[(c0 + cc) / NTN][(c0 + cc) % NTN];
            t[0] += __uint_as_float(v[cc].x); t[1] += __uint_as_float(v[cc].y); t[2] += __uint_as_float(v[cc].z); t[3] += __uint_as_float(v[cc].w);
            acc16[(c0 + cc) / NTN][(c0 + cc) % NTN] = t;
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (fused_last) {
    ConvP q2 = p;
    q2.partial = nullptr;
    t256_epilogue<NTM, NTN>(q2, acc16, m0 + wr * 128, n0 + (TMW ? wc * 32 : wc * 64), lane, smem);
  } else {
    t256_epilogue<NTM, NTN>(p, acc16, m0 + wr * 128, n0 + (TMW ? wc * 32 : wc * 64), lane, smem);
  }
  STAMP(4);
}

#ifdef SWEM_CONV_T256_ONLY
// Second translation unit of this file (swem_amd/build.py compiles conv.hip twice, side by side: the instantiations of
// conv_igemm_bf3s_kernel take five minutes, these five one): only conv_t256_kernel and its launcher.  ConvP lives in this file's
// anonymous namespace, so it crosses the boundary as an opaque pointer (same source, same layout).
}  // namespace
int swem_conv_t256_launch(int trows, const void *convp, unsigned gx, unsigned gy, unsigned gz, void *stream) {
  const ConvP &q = *static_cast<const ConvP *>(convp);
  const dim3 grid(gx, gy, gz);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // two stages x (A, B) x two planes x four k/8 groups x 256 rows = 128 KB for the k-loop; the epilogue stages four 32 x 32 sub-tiles
  // per wave (t256_epilogue): 144 KB
  constexpr size_t lds = 8 * T256_EPI_BYTES;
  static_assert(lds >= 2 * 2 * 2 * 4 * 256 * 16 && lds <= 160 * 1024, "LDS of conv_t256_kernel");
  if (!q.f16) {   // bf16x6 (three planes): 128-row tiles, (128 + 256) rows x 3 planes x 4 groups x 16 bytes per stage
    constexpr size_t lds3 = 2 * 3 * 4 * (128 + 256) * 16;
    static_assert(lds3 >= 8 * T256_EPI_BYTES, "the bf16x6 form's operand stages hold the epilogue's staging too");
    SWEM_ALLOW_LDS((conv_t256_kernel<false, 8>), lds3);
    hipLaunchKernelGGL((conv_t256_kernel<false, 8>), grid, dim3(512), lds3, st, q STAMP_PASS);
    return SWEM_OK;
  }
#define T256_LAUNCH(TMW_)                                                                         \
  {                                                                                               \
    SWEM_ALLOW_LDS((conv_t256_kernel<true, TMW_>), lds);                                          \
    hipLaunchKernelGGL((conv_t256_kernel<true, TMW_>), grid, dim3(512), lds, st, q STAMP_PASS);   \
  }
  switch (trows) {
    case 128: T256_LAUNCH(8); break;
    case 160: T256_LAUNCH(10); break;
    case 192: T256_LAUNCH(12); break;
    case 224: T256_LAUNCH(14); break;
    default: T256_LAUNCH(0)
  }
#undef T256_LAUNCH
  return SWEM_OK;
}
#else
}  // namespace
int swem_conv_t256_launch(int trows, const void *convp, unsigned gx, unsigned gy, unsigned gz, void *stream);   // (the other unit)
namespace {

// Reduce split-K partials in z order and apply the same epilogue.  One thread per 4 output channels.
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(ConvP p, int nsplit) {
  const bool glu = p.flags & SWEM_CONV_GLU;
  const int cq = p.Cout / 4;
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)(p.M - p.part_m0) * cq) return;
  const int mrel = (int)(idx / cq);
  const int m = p.part_m0 + mrel;
  const int co = (int)(idx - (long long)mrel * cq) * 4;
  const long long MN = (long long)(p.M - p.part_m0) * p.Ncols;
  auto sum4 = [&](int col) {
    const float *src = p.partial + (long long)mrel * p.Ncols + col;
    float4 s = *reinterpret_cast<const float4 *>(src);
    for (int z = 1; z < nsplit; ++z) {
      float4 t = *reinterpret_cast<const float4 *>(src + z * MN);
      s.x += t.x;
      s.y += t.y;
      s.z += t.z;
      s.w += t.w;
    }
    return s;
  };
  auto affine4 = [&](float4 s, int col) {
    float4 sc = p.scale ? *reinterpret_cast<const float4 *>(p.scale + col) : make_float4(p.uscale, p.uscale, p.uscale, p.uscale);
    float4 sh = p.shift ? *reinterpret_cast<const float4 *>(p.shift + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    return make_float4(s.x * sc.x + sh.x, s.y * sc.y + sh.y, s.z * sc.z + sh.z, s.w * sc.w + sh.w);
  };
  float4 v;
  if (glu) {
    const int g = co >> 5, rr = co & 31;
    float4 f = affine4(sum4(g * 64 + rr), g * 64 + rr);
    float4 a = affine4(sum4(g * 64 + 32 + rr), g * 64 + 32 + rr);
    v = make_float4(f.x * sigmoidf_(a.x), f.y * sigmoidf_(a.y), f.z * sigmoidf_(a.z), f.w * sigmoidf_(a.w));
  } else {
    v = affine4(sum4(co), co);
    if (p.res || p.res_pl) {
      const int HoWo = p.Ho * p.Wo;
      int b = m / HoWo;
      float4 rv = p.res_pl ? residual_from_planes(p, (long long)b * (p.res_bs / p.Cout) + (m - b * HoWo), co)
                           : *reinterpret_cast<const float4 *>(p.res + (long long)b * p.res_bs +
                                                               (long long)(m - b * HoWo) * p.Cout + co);
      if (p.flags & SWEM_CONV_MASK_POS) {
        v = make_float4(rv.x > 0.f ? v.x : 0.f, rv.y > 0.f ? v.y : 0.f, rv.z > 0.f ? v.z : 0.f, rv.w > 0.f ? v.w : 0.f);
      } else {
        v.x += rv.x;
        v.y += rv.y;
        v.z += rv.z;
        v.w += rv.w;
      }
    }
  }
  const unsigned nan_in = f32_nan(v);   // (before any ReLU: out_tile32)
  if (!glu && (p.flags & SWEM_CONV_RELU_OUT)) v = relu4(v);
  if (p.y) *reinterpret_cast<float4 *>(p.y + (long long)m * p.Cout + co) = v;
  if (!(p.ysp[0] || p.ysp[1])) return;
  // output planes (fused operand split): an even lane and its odd neighbour hold the 8 channels of one 16-byte run
  // (Cout / 4 is even -- Cout % 8 == 0 is the planes' precondition -- so a pair never straddles two pixels)
  unsigned bad = 0;
#pragma unroll
  for (int var = 0; var < 2; ++var) {
    if (!p.ysp[var]) continue;
    const float4 q = var ? relu4(v) : v;
    uint2 h, mm, l;
    split_as(p.ysp_npl[var], q, h, mm, l, bad);
    if (p.ysp_npl[var] == SWEM_PLANES_F16) bad |= nan_in;
    const uint2 h2 = make_uint2(__shfl_down(h.x, 1), __shfl_down(h.y, 1));
    const uint2 m2 = make_uint2(__shfl_down(mm.x, 1), __shfl_down(mm.y, 1));
    const uint2 l2 = make_uint2(__shfl_down(l.x, 1), __shfl_down(l.y, 1));
    if ((co & 4) == 0) {
      unsigned short *d = p.ysp[var] + ((long long)(co >> 3) * p.M + m) * 8;
      *reinterpret_cast<uint4 *>(d) = make_uint4(h.x, h.y, h2.x, h2.y);
      *reinterpret_cast<uint4 *>(d + p.ysp_ps) = make_uint4(mm.x, mm.y, m2.x, m2.y);
      if (p.ysp_npl[var] == 3) *reinterpret_cast<uint4 *>(d + 2 * p.ysp_ps) = make_uint4(l.x, l.y, l2.x, l2.y);
    }
  }
  range_fault(p.fault, bad);
}

struct Plan {
  int wm, wn, nsplit, kb_per_split;
};
constexpr int SWEM_KSPLIT_SKEW_DEFAULT = 0;   // percent (see the fused K-split launch)
constexpr int SK_MAX_WORKERS = 1024;   // stream-K: resident-block slots of the chip (256 CUs x at most 4 blocks)

// Tuning hook (tools/conv_bench.py): SWEM_CONV_PLAN="wm,wn,nsplit" forces one plan for every launch.
bool forced_plan(Plan &pl, int nkb, bool glu) {
  static int state = 0, fwm = 0, fwn = 0, fns = 0;
  if (state == 0) {
    const char *e = getenv("SWEM_CONV_PLAN");
    state = (e && sscanf(e, "%d,%d,%d", &fwm, &fwn, &fns) == 3) ? 1 : -1;
  }
  if (state < 0) return false;
  if (glu && fwn != 2) return false;
  int ns = fns < 1 ? 1 : (fns > nkb ? nkb : fns);
  int per = cdiv(nkb, ns);
  pl = Plan{fwm, fwn, cdiv(nkb, per), per};
  return true;
}

Plan make_plan(int M, int Ncols, int nkb, bool glu) {
  {
    Plan f;
    if (forced_plan(f, nkb, glu)) return f;
  }
  // Candidate wave tiles, largest first.  Take the largest whose grid gives every CU about two blocks
  // (the LDS image allows two resident blocks); if none does, take the smallest and split K until it does.
  static const int cand[3][2] = {{2, 2}, {1, 2}, {1, 1}};
  const long long target = 2 * 256;
  Plan best{1, glu ? 2 : 1, 1, nkb};
  for (int c = 0; c < 3; ++c) {
    const int wm = cand[c][0], wn = cand[c][1];
    if (glu && wn != 2) continue;
    if (!glu && Ncols < 64 * wn && wn > 1) continue;
    best = Plan{wm, wn, 1, nkb};
    if ((long long)cdiv(M, 64 * wm) * cdiv(Ncols, 64 * wn) >= target) return best;
  }
  const long long blocks = (long long)cdiv(M, 64 * best.wm) * cdiv(Ncols, 64 * best.wn);
  int ns = (int)((target + blocks - 1) / blocks);
  const int maxsplit = nkb / 4;  // keep at least 4 k-blocks (128 k) per split
  if (ns > maxsplit) ns = maxsplit;
  if (ns > 16) ns = 16;
  if (ns < 1) ns = 1;
  const int per = cdiv(nkb, ns);
  best.nsplit = cdiv(nkb, per);
  best.kb_per_split = per;
  return best;
}

template <int WM, int WN>
int launch_pipe(const ConvP &p, dim3 grid, hipStream_t st) {
  constexpr size_t lds = 2 * KQ * (64 * WM + 1 + 64 * WN + 1) * sizeof(float4);
  SWEM_ALLOW_LDS((conv_igemm_pipe_kernel<WM, WN>), lds);
  hipLaunchKernelGGL((conv_igemm_pipe_kernel<WM, WN>), grid, dim3(256), lds, st, p);
  return SWEM_OK;
}

template <int WM, int WN>
int launch_bf3(const ConvP &p, dim3 grid, hipStream_t st) {
  constexpr size_t lds = 2 * 3 * 4 * (64 * WM + 1 + 64 * WN + 1) * 16;
  SWEM_ALLOW_LDS((conv_igemm_bf3_kernel<WM, WN>), lds);
  hipLaunchKernelGGL((conv_igemm_bf3_kernel<WM, WN>), grid, dim3(256), lds, st, p);
  return SWEM_OK;
}

// dynamic LDS of a launch: the operand stages, or -- where a small tile's stages are smaller -- the plane staging of the
// epilogue (one PL_BYTES slice per wave), when output planes are wanted
static inline size_t lds_with_planes(const ConvP &p, size_t lds, int nwaves) {
  (void)p;
  const size_t need = (size_t)nwaves * PL_BYTES;   // (round 3: every epilogue stages its outputs, planes wanted or not)
  return lds < need ? need : lds;
}

// resident blocks per CU of one instantiation at its dynamic-LDS size (stream-K sizes its persistent grid by it): the
// runtime's occupancy figure, never more than the LDS allows (the API has been seen one block per CU high near SGPR-count
// edges, MI355X_MICROARCH.md: an over-sized persistent grid only costs a second, short round here -- no worker ever waits
// on a worker that has not produced its partial sums first -- but the balance is lost)
template <typename K>
int blocks_per_cu(K kernel, int threads, size_t lds) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kernel), threads, lds) != hipSuccess || nb < 1)
    nb = 1;
  const int by_lds = (int)((160 * 1024) / (lds ? lds : 1));
  if (by_lds >= 1 && nb > by_lds) nb = by_lds;
  const int by_waves = 32 / (threads / 64);
  return nb > by_waves ? by_waves : nb;
}

// occ != nullptr: do not launch, report the resident blocks per CU of the instantiation the launch would use
// (p.sk_workers != 0 selects the stream-K instantiation: a query passes -1)
template <int WM, int WN, int NST, int NW, bool M16, int KG, bool PF, int NPL, bool SK, bool F16 = false>
int launch_bf3s_one(const ConvP &p, dim3 grid, hipStream_t st, int *occ) {
  constexpr size_t lds = NST * NPL * KG * (64 * WM + 1 + 64 * WN + 1) * 16;   // NST stages x NPL planes
  const size_t dyn = lds_with_planes(p, lds, NW);
  SWEM_ALLOW_LDS((conv_igemm_bf3s_kernel<WM, WN, NST, NW, M16, NPL, KG, PF, SK, F16>), lds);
  if (occ) {
    // (asked anew on every query: the answer depends on the instantiation, its dynamic-LDS size and the current device, and a
    // cached copy would be unsynchronised mutable state in a library whose contract is "re-entrant across streams" -- the
    // query costs microseconds and only stream-K launches make it)
    *occ = blocks_per_cu(conv_igemm_bf3s_kernel<WM, WN, NST, NW, M16, NPL, KG, PF, SK, F16>, 64 * NW, dyn);
    return SWEM_OK;
  }
  hipLaunchKernelGGL((conv_igemm_bf3s_kernel<WM, WN, NST, NW, M16, NPL, KG, PF, SK, F16>), grid, dim3(64 * NW), dyn, st, p STAMP_PASS);
  return SWEM_OK;
}
template <int WM, int WN, int NST, int NW, bool M16 = false, int KG = 4, bool PF = false>
int launch_bf3s_n(const ConvP &p, dim3 grid, hipStream_t st, int *occ = nullptr) {
  const bool sk = p.sk_workers != 0;
  // stream-K instantiations exist for the two-plane ("bf16x3") and one-plane kernels of the eight-wave 16x16x32 tile and of
  // the prefetched-fragment tiles (what the tuner is offered); everything else runs the plain grid
  constexpr bool SKOK = M16 && KG == 4 && NST == 2 && (NW == 8 || PF);
  if (sk && !(SKOK && p.nplanes <= 2)) {
    if (occ) {
      *occ = 0;   // no stream-K form of this variant: the caller launches the plain grid
      return SWEM_OK;
    }
    swem_set_error("conv2d_bf16x3: this kernel variant has no stream-K form");
    return SWEM_E_ARG;
  }
  // (SWEM_ISA_SUBSET: tests/test_abi.py compiles this file to assembly to check the LDS-DMA kernels' use of M0; a fifth of the
  // instantiations -- every tile, the two-plane kernels of both operand formats, no stream-K -- keeps that check to a minute)
#ifndef SWEM_ISA_SUBSET
  if (p.nplanes == 1) {   // plain bf16 (one plane, one product): a third of the LDS, the same tiles
    if constexpr (SKOK) {
      if (sk) return launch_bf3s_one<WM, WN, NST, NW, M16, KG, PF, 1, true>(p, grid, st, occ);
    }
    return launch_bf3s_one<WM, WN, NST, NW, M16, KG, PF, 1, false>(p, grid, st, occ);
  }
#endif
  if (p.nplanes == 2 && p.f16) {   // "f16x3": the same kernel on fp16 (hi, mid) planes and the f16 MFMA
    if (sk) {
      if (occ) {
        *occ = 0;
        return SWEM_OK;
      }
      swem_set_error("conv2d_bf16x3: the f16x3 arithmetic has no stream-K form");
      return SWEM_E_ARG;
    }
    return launch_bf3s_one<WM, WN, NST, NW, M16, KG, PF, 2, false, true>(p, grid, st, occ);
  }
  if (p.nplanes == 2) {   // "bf16x3": hi and mid planes, three products (hi.hi + hi.mid + mid.hi): two thirds of the LDS
#ifndef SWEM_ISA_SUBSET
    if constexpr (SKOK) {
      if (sk) return launch_bf3s_one<WM, WN, NST, NW, M16, KG, PF, 2, true>(p, grid, st, occ);
    }
#endif
    return launch_bf3s_one<WM, WN, NST, NW, M16, KG, PF, 2, false>(p, grid, st, occ);
  }
#ifndef SWEM_ISA_SUBSET
  if constexpr (NST <= 3 && !PF) {
    return launch_bf3s_one<WM, WN, NST, NW, M16, KG, PF, 3, false>(p, grid, st, occ);
  }
#endif
  swem_set_error("conv2d_bf16x3: four-stage rings and prefetched fragments need at most two planes");
  return SWEM_E_ARG;
}

// variant (plan bits 20-23): 0 = the tile's default; 1 = three LDS stages instead of two (or two instead of three);
// 2 = eight waves on the 128x128 tile (two stages), 3 = eight waves, three stages; 4 / 6 = variants 0 / 2 on the
// 16x16x32 MFMA shape
template <int WM, int WN>
int launch_bf3s(const ConvP &p, dim3 grid, hipStream_t st, int variant, int *occ = nullptr) {
  // deeper rings (round 2): with two planes a stage is 33 KB for the 128x128 tile, so FOUR stages fit; the k-loop of every
  // tile was bound by the L2 / Infinity-Cache -> LDS latency of the one or two k-blocks in flight (in-kernel stamps,
  // tools/conv_stamps.py: 2600 cycles per k-block against 768 of MFMA on the 128x128 tile, 830 against 192 on 64x64)
#ifndef SWEM_ISA_SUBSET
  if (p.nplanes <= 2) {
    if constexpr (WM == 2 && WN == 2) {
      if (variant == 12) return launch_bf3s_n<2, 2, 4, 8>(p, grid, st, occ);
      if (variant == 13) return launch_bf3s_n<2, 2, 4, 8, true>(p, grid, st, occ);
    }
    if constexpr (WM == 2 && WN == 2) {
      // prefetched fragments (PF): 5 = four waves of 64x64, two stages (two blocks per CU); 15 = four waves, three stages (one
      // block per CU, one wave per SIMD); 7 = eight waves of 32x64, four stages (one block per CU)
      if (variant == 5) return launch_bf3s_n<2, 2, 2, 4, true, 4, true>(p, grid, st, occ);
      if (variant == 15) return launch_bf3s_n<2, 2, 3, 4, true, 4, true>(p, grid, st, occ);
      if (variant == 7) return launch_bf3s_n<2, 2, 4, 8, true, 4, true>(p, grid, st, occ);
    }
    if (variant == 10) return launch_bf3s_n<WM, WN, 4, 4>(p, grid, st, occ);
    if (variant == 11) return launch_bf3s_n<WM, WN, 4, 4, true>(p, grid, st, occ);
  }
#endif
  if constexpr (WM == 2 && WN == 2) {
#ifndef SWEM_ISA_SUBSET
    if (variant == 14) return launch_bf3s_n<2, 2, 3, 8, true>(p, grid, st, occ);
    if (variant == 2) return launch_bf3s_n<2, 2, 2, 8>(p, grid, st, occ);
    if (variant == 3) return launch_bf3s_n<2, 2, 3, 8>(p, grid, st, occ);
    if (variant == 9) return launch_bf3s_n<2, 2, 3, 4, false, 2>(p, grid, st, occ);   // ... three stages: two blocks per CU
#endif
    if (variant == 6) return launch_bf3s_n<2, 2, 2, 8, true>(p, grid, st, occ);
    if (variant == 8) return launch_bf3s_n<2, 2, 2, 4, false, 2>(p, grid, st, occ);   // 16-k blocks: three blocks per CU
  }
  if (variant == 4)
    return (WM * WN == 1) ? launch_bf3s_n<WM, WN, 3, 4, true>(p, grid, st, occ) : launch_bf3s_n<WM, WN, 2, 4, true>(p, grid, st, occ);
  const bool three = (WM * WN == 1) != (variant == 1);
  return three ? launch_bf3s_n<WM, WN, 3, 4>(p, grid, st, occ) : launch_bf3s_n<WM, WN, 2, 4>(p, grid, st, occ);
}

template <int WM, int WN, bool DB>
int launch(const ConvP &p, dim3 grid, hipStream_t st) {
  constexpr size_t lds = (DB ? 2 : 1) * KQ * (64 * WM + 1 + 64 * WN + 1) * sizeof(float4);
  SWEM_ALLOW_LDS((conv_igemm_kernel<WM, WN, DB>), lds);
  hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, DB>), grid, dim3(256), lds_with_planes(p, lds, 4), st, p);
  return SWEM_OK;
}

}  // namespace

// plan hint: 0 = heuristic; else wm | wn << 4 | nsplit << 8 | math << 16 | variant << 20 (swem_hip.h)
Plan resolve_plan(int plan, int M, int Ncols, int nkb, bool glu) {
  const bool presplit_math = ((plan >> 16) & 3) != 0;   // the 128x64 tile (wm 2, wn 1) exists for the pre-split kernels only
  const bool f16x3 = ((plan >> 16) & 7) == 7;           // ... and the 256-column tiles (wm 4, wn 4: conv_t256_kernel) for f16x3,
  const bool bf16x6_128 = ((plan >> 16) & 7) == 1 && ((plan >> 20) & 15) == 4;   // or for bf16x6 with 128-row tiles (bits 20-23 = 4)
  plan &= 0xffff;
  if (plan > 0) {
    int wm = plan & 15, wn = (plan >> 4) & 15, ns = plan >> 8;
    bool ok = (wm == 1 || wm == 2) && (wn == 1 || wn == 2) && !(wm == 2 && wn == 1 && !presplit_math) && !(glu && wn != 2);
    ok = ok || (wm == 4 && wn == 4 && (f16x3 || bf16x6_128));
    if (ok) {
      ns = ns < 1 ? 1 : (ns > nkb ? nkb : ns);
      int per = cdiv(nkb, ns);
      return Plan{wm, wn, cdiv(nkb, per), per};
    }
  }
  return make_plan(M, Ncols, nkb, glu);
}

// Tail split (plan bits 24-27 = K-split factor ts of the LAST, partly filled round of tiles): a grid of T tiles on S
// resident-block slots runs ceil(T/S) rounds however few tiles the last one holds (2x120x216x256->256: 1620 tiles on 512
// slots = 3.16 -> 4 rounds).  The whole rounds are launched as they are; the M-tile rows of the remainder are launched a
// second time split ts ways over K (so they fill the chip) and reduced by the split-K epilogue over those rows only.
struct TailSplit {
  int main_mt, nsplit, kb_per_split;
};
// rows of a conv_t256_kernel tile (plan tile 4 x 4): plan bits 20-23 = TMW / 2 (the 1 x 8 wave layout, 16 TMW rows), 0 = 256
static inline int t256_rows(int plan) {
  const int v = (plan >> 20) & 15;
  return (v >= 4 && v <= 7) ? 32 * v : 256;
}
static inline int tile_rows(int plan, const Plan &pl) { return pl.wm == 4 ? t256_rows(plan) : 64 * pl.wm; }
static TailSplit tail_split(int plan, const Plan &pl, int M, int Ncols, int nkb) {
  TailSplit t{cdiv(M, 64 * pl.wm), 1, nkb};
  const int ts = (plan >> 24) & 15;
  if (ts <= 1 || pl.nsplit != 1 || nkb < 2 * ts) return t;
  const int bpc = (pl.wm == 2 && pl.wn == 2) ? 1 : 2;  // resident blocks per CU (LDS: 98 KB for the 128x128 tile, 74 KB else)
  const long long slots = (long long)swem_device_cus() * bpc;
  const int mt = cdiv(M, 64 * pl.wm), nt = cdiv(Ncols, 64 * pl.wn);
  const long long tiles = (long long)mt * nt, full = tiles / slots * slots;
  if (full == 0 || full == tiles) return t;
  t.main_mt = (int)(full / nt);
  if (t.main_mt >= mt) {
    t.main_mt = mt;
    return t;
  }
  t.kb_per_split = cdiv(nkb, ts);
  t.nsplit = cdiv(nkb, t.kb_per_split);
  return t;
}

// output size: forward floor((H + 2p - K)/s) + 1; data gradient: the forward INPUT size (H-1)*s + K - 2p + e, where
// e = rows/cols of the forward input the strided filter never reached (flag bits EH / EW)
static inline void conv_out_dims(int H, int W, int KH, int KW, int stride, int pad, int flags, int &Ho, int &Wo) {
  if (flags & SWEM_CONV_DGRAD) {
    Ho = (H - 1) * stride + KH - 2 * pad + ((flags & SWEM_CONV_DGRAD_EH) ? 1 : 0);
    Wo = (W - 1) * stride + KW - 2 * pad + ((flags & SWEM_CONV_DGRAD_EW) ? 1 : 0);
  } else {
    Ho = (H + 2 * pad - KH) / stride + 1;
    Wo = (W + 2 * pad - KW) / stride + 1;
  }
}

extern "C" size_t swem_conv2d_workspace(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                        int flags, int plan) {
  if (stride <= 0) return 0;
  int Ho, Wo;
  conv_out_dims(H, W, KH, KW, stride, pad, flags, Ho, Wo);
  long long M = (long long)B * Ho * Wo;
  int Ncols = (flags & SWEM_CONV_GLU) ? 2 * Cout : Cout;
  int nkb = cdiv((long long)KH * KW * Cin, BK);
  Plan pl = resolve_plan(plan, (int)M, Ncols, nkb, flags & SWEM_CONV_GLU);
  if (((plan >> 24) & 15) == 1)   // stream-K: a partial tile and a flag per worker (workers <= resident slots)
    return (size_t)SK_MAX_WORKERS * ((size_t)64 * pl.wm * 64 * pl.wn * sizeof(float) + sizeof(unsigned));
  if (pl.nsplit <= 1) {
    TailSplit t = tail_split(plan, pl, (int)M, Ncols, nkb);
    return t.nsplit > 1 ? (size_t)t.nsplit * (M - (long long)t.main_mt * 64 * pl.wm) * Ncols * sizeof(float) : 0;
  }
  // (the pre-split kernel keeps padded partial TILES and a counter per tile; the fp32 kernels [z][M][Ncols]: the larger)
  const size_t mt = cdiv(M, tile_rows(plan, pl)), nt = cdiv(Ncols, 64 * pl.wn), tile = (size_t)tile_rows(plan, pl) * 64 * pl.wn * sizeof(float);
  return (size_t)pl.nsplit * mt * nt * tile + mt * nt * sizeof(unsigned);
}

namespace {
struct PlaneOut {
  void *planes[2];
  int npl[2];
  unsigned *counters = nullptr;   // optional: caller-owned tile counters, ALL ZERO between calls (swem_conv2d_nhwc_bf16x3_planes_ctr)
  size_t ncounters = 0;
  const void *res_planes = nullptr;   // optional: the residual as planes (swem_conv2d_nhwc_bf16x3_planes_res)
  long long res_ps = 0, res_npx = 0;
  int res_npl = 0;
  unsigned *fault = nullptr;          // optional: the caller's sticky fault word (SWEM_FAULT_*)
};
// validate the optional output planes and put them into the launch parameters (after p.M / p.Cout are set)
int set_planes(ConvP &p, const PlaneOut *po, bool glu, const char *who) {
  p.ysp[0] = p.ysp[1] = nullptr;
  p.ysp_npl[0] = p.ysp_npl[1] = 3;
  p.ysp_ps = (long long)p.M * p.Cout;
  if (!po || (!po->planes[0] && !po->planes[1])) return SWEM_OK;
  (void)glu;   // the gated output goes through the same row-layout stores as any other
  SWEM_REQUIRE(p.Cout % 8 == 0, SWEM_E_SHAPE, "%s: output planes need Cout %% 8 == 0", who);
  for (int v = 0; v < 2; ++v) {
    SWEM_REQUIRE(!po->planes[v] || po->npl[v] == 2 || po->npl[v] == 3 || po->npl[v] == SWEM_PLANES_F16, SWEM_E_ARG,
                 "%s: 2 or 3 bf16 output planes, or SWEM_PLANES_F16 (an fp16 pair)", who);
    p.ysp[v] = static_cast<unsigned short *>(po->planes[v]);
    p.ysp_npl[v] = po->npl[v];
  }
  return SWEM_OK;
}
int conv2d_f32_impl(void *stream, const float *x0, int c0, long long bs0, const float *x1, int c1, long long bs1,
                    const float *x2, int c2, long long bs2, int B, int H, int W, const float *w, long long w_bs,
                    const float *scale, const float *shift, const float *res, long long res_bs, float *y, int Cout, int KH,
                    int KW, int stride, int pad, int flags, int plan, void *ws, size_t ws_bytes, const PlaneOut *po);
}  // namespace

extern "C" int swem_conv2d_nhwc_f32(void *stream, const float *x0, int c0, long long bs0, const float *x1, int c1,
                                    long long bs1, const float *x2, int c2, long long bs2, int B, int H, int W,
                                    const float *w, long long w_bs, const float *scale, const float *shift,
                                    const float *res, long long res_bs, float *y, int Cout, int KH, int KW, int stride,
                                    int pad, int flags, int plan, void *ws, size_t ws_bytes) {
  return conv2d_f32_impl(stream, x0, c0, bs0, x1, c1, bs1, x2, c2, bs2, B, H, W, w, w_bs, scale, shift, res, res_bs, y, Cout,
                         KH, KW, stride, pad, flags, plan, ws, ws_bytes, nullptr);
}

extern "C" int swem_conv2d_nhwc_f32_planes(void *stream, const float *x0, int c0, long long bs0, const float *x1, int c1,
                                           long long bs1, const float *x2, int c2, long long bs2, int B, int H, int W,
                                           const float *w, long long w_bs, const float *scale, const float *shift,
                                           const float *res, long long res_bs, float *y, int Cout, int KH, int KW,
                                           int stride, int pad, int flags, int plan, void *ws, size_t ws_bytes,
                                           void *planes, int nplanes, void *planes_relu, int nplanes_relu, void *fault) {
  PlaneOut po{{planes, planes_relu}, {nplanes, nplanes_relu}};
  po.fault = static_cast<unsigned *>(fault);
  return conv2d_f32_impl(stream, x0, c0, bs0, x1, c1, bs1, x2, c2, bs2, B, H, W, w, w_bs, scale, shift, res, res_bs, y, Cout,
                         KH, KW, stride, pad, flags, plan, ws, ws_bytes, &po);
}

namespace {
int conv2d_f32_impl(void *stream, const float *x0, int c0, long long bs0, const float *x1, int c1, long long bs1,
                    const float *x2, int c2, long long bs2, int B, int H, int W,
                    const float *w, long long w_bs, const float *scale, const float *shift, const float *res,
                    long long res_bs, float *y, int Cout, int KH, int KW, int stride, int pad, int flags, int plan, void *ws,
                    size_t ws_bytes, const PlaneOut *po) {
  SWEM_REQUIRE(x0 && w && (y || (po && (po->planes[0] || po->planes[1]))), SWEM_E_ARG,
               "conv2d: null pointer (y may be NULL only when output planes are given)");
  if (!x1) c1 = 0;
  if (!x2) c2 = 0;
  SWEM_REQUIRE(!(x2 && !x1), SWEM_E_ARG, "conv2d: source 2 without source 1");
  SWEM_REQUIRE(c0 > 0 && c0 % 4 == 0 && c1 % 4 == 0 && c2 % 4 == 0, SWEM_E_SHAPE,
               "conv2d: every source needs a channel count that is a multiple of 4 (got %d,%d,%d)", c0, c1, c2);
  SWEM_REQUIRE(B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, SWEM_E_SHAPE,
               "conv2d: bad geometry");
  SWEM_REQUIRE(Cout > 0 && Cout % 4 == 0, SWEM_E_SHAPE, "conv2d: Cout must be a multiple of 4 (got %d)", Cout);
  const bool glu = flags & SWEM_CONV_GLU;
  SWEM_REQUIRE(!glu || (Cout % 32 == 0 && !res), SWEM_E_SHAPE, "conv2d: GLU needs Cout %% 32 == 0 and no residual");
  ConvP p;
  p.sk_workers = 0; p.sk_mtiles = p.sk_ntiles = 0; p.sk_ws = nullptr; p.sk_flags = nullptr; p.fault = nullptr; p.spin_limit = 1 << 24; p.res_pl = nullptr; p.res_ps = 0; p.res_npx = 0; p.res_npl = 0;
  p.x[0] = x0; p.x[1] = x1 ? x1 : x0; p.x[2] = x2 ? x2 : x0;
  p.c[0] = c0; p.c[1] = c1; p.c[2] = c2;
  p.bs[0] = bs0; p.bs[1] = bs1; p.bs[2] = bs2;
  p.B = B; p.H = H; p.W = W;
  conv_out_dims(H, W, KH, KW, stride, pad, flags, p.Ho, p.Wo);
  SWEM_REQUIRE(p.Ho > 0 && p.Wo > 0, SWEM_E_SHAPE, "conv2d: empty output");
  SWEM_REQUIRE(!(flags & SWEM_CONV_DGRAD) || ((stride == 1 || stride == 2) && c0 % 32 == 0 && c1 % 32 == 0 && c2 % 32 == 0),
               SWEM_E_SHAPE, "conv2d: the data-gradient mode needs stride 1 or 2 and sources that are multiples of 32 channels");
  SWEM_REQUIRE(!(flags & SWEM_CONV_MASK_POS) || res, SWEM_E_ARG, "conv2d: SWEM_CONV_MASK_POS needs the mask in res");
  p.Cin = c0 + c1 + c2;
  p.K = KH * KW * p.Cin;
  long long M = (long long)B * p.Ho * p.Wo;
  SWEM_REQUIRE(M < (1ll << 31) && M * (glu ? 2 * Cout : Cout) < (1ll << 40), SWEM_E_SHAPE, "conv2d: too large");
  p.M = (int)M;
  p.w = w; p.scale = scale; p.uscale = 1.f; p.shift = shift; p.res = res; p.res_bs = res_bs; p.w_bs = w_bs; p.y = y;
  p.wsplit = nullptr;
  p.Cout = Cout; p.Ncols = glu ? 2 * Cout : Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.flags = flags;
  p.nkb = cdiv(p.K, BK);
  fast_div_make((unsigned)(p.Ho * p.Wo), p.fd_howo_mul, p.fd_howo_sh);
  fast_div_make((unsigned)p.Wo, p.fd_wo_mul, p.fd_wo_sh);
  Plan pl = resolve_plan(plan, p.M, p.Ncols, p.nkb, glu);
  if (pl.wm == 4) pl = make_plan(p.M, p.Ncols, p.nkb, glu);   // (the 256-column tiles belong to the pre-split kernel: heuristic tile here)
  if (pl.wm == 2 && pl.wn == 1) pl.wm = 1;   // (a 128x64 plan on inputs that are not pre-split: the kernels here have no such tile)
  p.kb_per_split = pl.kb_per_split;
  p.partial = nullptr;
  p.mt0 = 0; p.part_m0 = 0; p.nplanes = 3; p.f16 = 0; p.xpn = 1;
  if (int prc = set_planes(p, po, glu, "conv2d")) return prc;
  p.fault = po ? po->fault : nullptr;
  if (pl.nsplit > 1) {
    size_t need = (size_t)pl.nsplit * M * p.Ncols * sizeof(float);
    SWEM_REQUIRE(ws && ws_bytes >= need, SWEM_E_WORKSPACE, "conv2d: workspace %zu < %zu bytes", ws_bytes, need);
    p.partial = static_cast<float *>(ws);
  }
  SWEM_REQUIRE(w_bs == 0 || (p.Ho * p.Wo) % (64 * pl.wm) == 0, SWEM_E_SHAPE,
               "conv2d: per-batch filters need Ho*Wo (%d) to be a multiple of the %d-row tile", p.Ho * p.Wo, 64 * pl.wm);
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid(cdiv(p.M, 64 * pl.wm), cdiv(p.Ncols, 64 * pl.wn), pl.nsplit);
  int rc;
  static int single = -1;
  if (single < 0) {
    const char *e = getenv("SWEM_CONV_SINGLE");
    single = e ? atoi(e) : 0;
  }
  static int nopipe = -1;
  if (nopipe < 0) {
    const char *e = getenv("SWEM_CONV_NOPIPE");
    nopipe = e ? atoi(e) : 0;
  }
  const bool pipe_ok = !nopipe && c0 % 32 == 0 && c1 % 32 == 0 && c2 % 32 == 0;
  static int math_env = -1;
  if (math_env < 0) {
    const char *e = getenv("SWEM_CONV_MATH");  // tuning/debug override: "bf16x6" or "f32"
    math_env = !e ? 0 : (strcmp(e, "bf16x6") == 0 ? 1 : 2);
  }
  const bool emulate = pipe_ok && (math_env == 1 || (math_env == 0 && ((plan >> 16) & 1)));
  if (emulate && pl.wm == 2 && pl.wn == 2) rc = launch_bf3<2, 2>(p, grid, st);
  else if (emulate && pl.wm == 1 && pl.wn == 2) rc = launch_bf3<1, 2>(p, grid, st);
  else if (emulate && pl.wm == 1 && pl.wn == 1) rc = launch_bf3<1, 1>(p, grid, st);
  else if (pipe_ok && pl.wm == 2 && pl.wn == 2) rc = launch_pipe<2, 2>(p, grid, st);
  else if (pipe_ok && pl.wm == 1 && pl.wn == 2) rc = launch_pipe<1, 2>(p, grid, st);
  else if (pipe_ok && pl.wm == 1 && pl.wn == 1) rc = launch_pipe<1, 1>(p, grid, st);
  else if (pl.wm == 2 && pl.wn == 2) rc = single ? launch<2, 2, false>(p, grid, st) : launch<2, 2, true>(p, grid, st);
  else if (pl.wm == 1 && pl.wn == 2) rc = single ? launch<1, 2, false>(p, grid, st) : launch<1, 2, true>(p, grid, st);
  else if (pl.wm == 1 && pl.wn == 1) rc = single ? launch<1, 1, false>(p, grid, st) : launch<1, 1, true>(p, grid, st);
  else {
    swem_set_error("conv2d: unsupported tile plan %dx%d", pl.wm, pl.wn);
    return SWEM_E_ARG;
  }
  if (rc) return rc;
  SWEM_CHECK_LAUNCH("conv_igemm_kernel");
  if (pl.nsplit > 1) {
    long long work = M * (Cout / 4);
    hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3(cdiv(work, 256)), dim3(256), 0, st, p, pl.nsplit);
    SWEM_CHECK_LAUNCH("conv_splitk_epilogue_kernel");
  }
  return SWEM_OK;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
namespace {
// x [npix][C] fp32 -> three bf16 planes [C/8][npix][8] with x = hi + mid + lo (optionally of relu(x)).
// Block = 32 pixels x 8 channel groups; thread (pixel p, group g) = (tid / 8, tid % 8): the 8 threads of a pixel read 256
// contiguous bytes of its row (full cache lines; with the pixel on the fast thread index every thread read 32 bytes of
// its own line), and per channel group 8 consecutive pixels store one 128-byte run of every plane.
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float *__restrict__ x, unsigned short *__restrict__ out,
                                                           long long npix, int C, int relu) {
  const long long pix = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int cg = blockIdx.y * 8 + (threadIdx.x & 7);
  if (pix >= npix || cg >= C / 8) return;
  const float *src = x + pix * C + cg * 8;
  float4 v0 = *reinterpret_cast<const float4 *>(src), v1 = *reinterpret_cast<const float4 *>(src + 4);
  if (relu) {
    v0 = relu4(v0);
    v1 = relu4(v1);
  }
  uint2 h0, m0, l0, h1, m1, l1;
  split3(v0, h0, m0, l0);
  split3(v1, h1, m1, l1);
  const long long plane = npix * C, i = (long long)cg * npix + pix;
  *reinterpret_cast<uint4 *>(out + i * 8) = make_uint4(h0.x, h0.y, h1.x, h1.y);
  *reinterpret_cast<uint4 *>(out + plane + i * 8) = make_uint4(m0.x, m0.y, m1.x, m1.y);
  *reinterpret_cast<uint4 *>(out + 2 * plane + i * 8) = make_uint4(l0.x, l0.y, l1.x, l1.y);
}
}  // namespace

namespace {
// ... -> two fp16 planes [C/8][npix][8] with x = hi + mid (bf16_split.h, split2h): the operand format of the f16x3 arithmetic
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float *__restrict__ x, unsigned short *__restrict__ out,
                                                          long long npix, int C, int relu, unsigned *fault) {
  const long long pix = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int cg = blockIdx.y * 8 + (threadIdx.x & 7);
  if (pix >= npix || cg >= C / 8) return;
  const float *src = x + pix * C + cg * 8;
  float4 v0 = *reinterpret_cast<const float4 *>(src), v1 = *reinterpret_cast<const float4 *>(src + 4);
  // (a NaN in the map is a fault whatever follows: relu1 would turn it into a clean 0 before the range test sees it)
  const unsigned nan_in = relu ? (f32_nan(v0) | f32_nan(v1)) : 0u;
  if (relu) {
    v0 = relu4(v0);
    v1 = relu4(v1);
  }
  uint2 h0, m0, h1, m1;
  split2h(v0, h0, m0);
  split2h(v1, h1, m1);
  const long long plane = npix * C, i = (long long)cg * npix + pix;
  *reinterpret_cast<uint4 *>(out + i * 8) = make_uint4(h0.x, h0.y, h1.x, h1.y);
  *reinterpret_cast<uint4 *>(out + plane + i * 8) = make_uint4(m0.x, m0.y, m1.x, m1.y);
  range_fault(fault, f16_oor(v0) | f16_oor(v1) | nan_in);   // (behind the input ReLU: a large NEGATIVE value under a ReLU is a plain 0)
}
}  // namespace

extern "C" int swem_split_f16x2_f32(void *stream, const float *x, void *out, long long npix, int C, int relu, void *fault) {
  SWEM_REQUIRE(x && out && npix > 0 && C > 0 && C % 8 == 0, SWEM_E_ARG, "split_f16x2: need C %% 8 == 0");
  hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)cdiv(npix, 32), (unsigned)cdiv(C / 8, 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, static_cast<unsigned short *>(out), npix, C, relu,
                     static_cast<unsigned *>(fault));
  SWEM_CHECK_LAUNCH("split_f16x2");
  return SWEM_OK;
}

extern "C" int swem_split_bf16x3_f32(void *stream, const float *x, void *out, long long npix, int C, int relu) {
  SWEM_REQUIRE(x && out && npix > 0 && C > 0 && C % 8 == 0, SWEM_E_ARG, "split_bf16x3: need C %% 8 == 0");
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)cdiv(npix, 32), (unsigned)cdiv(C / 8, 8)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, static_cast<unsigned short *>(out), npix, C, relu);
  SWEM_CHECK_LAUNCH("split_bf16x3");
  return SWEM_OK;
}

namespace {
int conv2d_bf16x3_impl(void *stream, const void *x0, int c0, long long bs0, long long ps0, const void *x1, int c1,
                       long long bs1, long long ps1, const void *x2, int c2, long long bs2, long long ps2, int B, int H, int W,
                       const void *w_bf16x3, const float *scale, const float *shift, const float *res, long long res_bs,
                       float *y, int Cout, int KH, int KW, int stride, int pad, int flags, int plan, void *ws, size_t ws_bytes,
                       const PlaneOut *po, long long w_bs = 0, float uscale = 1.f);
}
extern "C" int swem_conv2d_nhwc_bf16x3(void *stream, const void *x0, int c0, long long bs0, long long ps0,
                                       const void *x1, int c1, long long bs1, long long ps1, const void *x2, int c2,
                                       long long bs2, long long ps2, int B, int H, int W, const void *w_bf16x3,
                                       const float *scale, const float *shift, const float *res, long long res_bs,
                                       float *y, int Cout, int KH, int KW, int stride, int pad, int flags, int plan,
                                       void *ws, size_t ws_bytes) {
  return conv2d_bf16x3_impl(stream, x0, c0, bs0, ps0, x1, c1, bs1, ps1, x2, c2, bs2, ps2, B, H, W, w_bf16x3, scale, shift, res,
                            res_bs, y, Cout, KH, KW, stride, pad, flags, plan, ws, ws_bytes, nullptr);
}
extern "C" int swem_conv2d_nhwc_bf16x3_planes(void *stream, const void *x0, int c0, long long bs0, long long ps0,
                                              const void *x1, int c1, long long bs1, long long ps1, const void *x2, int c2,
                                              long long bs2, long long ps2, int B, int H, int W, const void *w_bf16x3,
                                              const float *scale, const float *shift, const float *res, long long res_bs,
                                              float *y, int Cout, int KH, int KW, int stride, int pad, int flags, int plan,
                                              void *ws, size_t ws_bytes, void *planes, int nplanes, void *planes_relu,
                                              int nplanes_relu) {
  PlaneOut po{{planes, planes_relu}, {nplanes, nplanes_relu}};
  return conv2d_bf16x3_impl(stream, x0, c0, bs0, ps0, x1, c1, bs1, ps1, x2, c2, bs2, ps2, B, H, W, w_bf16x3, scale, shift, res,
                            res_bs, y, Cout, KH, KW, stride, pad, flags, plan, ws, ws_bytes, &po);
}
// ... with caller-owned tile counters: `counters` (ncounters words) must be ALL ZERO when the call is enqueued and must not be
// used by another stream at the same time; the kernels leave it all zero again.  K-split and stream-K launches then need no
// memset launch for their counters / flags (4.7 us each on this chip, as many as K-split layers per frame).
extern "C" int swem_conv2d_nhwc_bf16x3_planes_ctr(void *stream, const void *x0, int c0, long long bs0, long long ps0,
                                                  const void *x1, int c1, long long bs1, long long ps1, const void *x2, int c2,
                                                  long long bs2, long long ps2, int B, int H, int W, const void *w_bf16x3,
                                                  const float *scale, const float *shift, const float *res, long long res_bs,
                                                  float *y, int Cout, int KH, int KW, int stride, int pad, int flags, int plan,
                                                  void *ws, size_t ws_bytes, void *planes, int nplanes, void *planes_relu,
                                                  int nplanes_relu, void *counters, size_t ncounters, void *fault) {
  PlaneOut po{{planes, planes_relu}, {nplanes, nplanes_relu}, static_cast<unsigned *>(counters), ncounters};
  po.fault = static_cast<unsigned *>(fault);
  return conv2d_bf16x3_impl(stream, x0, c0, bs0, ps0, x1, c1, bs1, ps1, x2, c2, bs2, ps2, B, H, W, w_bf16x3, scale, shift, res,
                            res_bs, y, Cout, KH, KW, stride, pad, flags, plan, ws, ws_bytes, &po);
}
// ... and with the RESIDUAL given as operand planes (see include/swem_hip.h)
extern "C" int swem_conv2d_nhwc_bf16x3_planes_res(void *stream, const void *x0, int c0, long long bs0, long long ps0,
                                                  const void *x1, int c1, long long bs1, long long ps1, const void *x2, int c2,
                                                  long long bs2, long long ps2, int B, int H, int W, const void *w_bf16x3,
                                                  const float *scale, const float *shift, const void *res_planes,
                                                  long long res_ps, long long res_npx, int res_nplanes, long long res_bs,
                                                  float *y, int Cout, int KH, int KW, int stride, int pad, int flags, int plan,
                                                  void *ws, size_t ws_bytes, void *planes, int nplanes, void *planes_relu,
                                                  int nplanes_relu, void *counters, size_t ncounters, void *fault) {
  PlaneOut po{{planes, planes_relu}, {nplanes, nplanes_relu}, static_cast<unsigned *>(counters), ncounters};
  po.fault = static_cast<unsigned *>(fault);
  po.res_planes = res_planes; po.res_ps = res_ps; po.res_npx = res_npx; po.res_npl = res_nplanes;
  SWEM_REQUIRE(res_planes, SWEM_E_ARG, "conv2d_bf16x3_planes_res: null residual planes");
  return conv2d_bf16x3_impl(stream, x0, c0, bs0, ps0, x1, c1, bs1, ps1, x2, c2, bs2, ps2, B, H, W, w_bf16x3, scale, shift, nullptr,
                            res_bs, y, Cout, KH, KW, stride, pad, flags, plan, ws, ws_bytes, &po);
}
// batched GEMM on pre-split planes (common.h): y[b] = x[b] . w[b]^T with per-batch filter planes, w_bs bf16 elements apart
int swem_gemm_bf16x3_batched(void *stream, const void *x, int K, long long bs, long long ps, int B, int M, const void *w,
                             long long w_bs, float *y, int Ncols, int plan, void *ws, size_t ws_bytes, void *y_planes,
                             int y_nplanes, float out_scale, void *fault) {
  PlaneOut po{{y_planes, nullptr}, {y_nplanes, 3}};
  po.fault = static_cast<unsigned *>(fault);
  return conv2d_bf16x3_impl(stream, x, K, bs, ps, nullptr, 0, 0, 0, nullptr, 0, 0, 0, B, M, 1, w, nullptr, nullptr, nullptr, 0, y,
                            Ncols, 1, 1, 1, 0, 0, plan, ws, ws_bytes, y_planes ? &po : nullptr, w_bs, out_scale);
}
namespace {
int conv2d_bf16x3_impl(void *stream, const void *x0, int c0, long long bs0, long long ps0, const void *x1, int c1,
                       long long bs1, long long ps1, const void *x2, int c2, long long bs2, long long ps2, int B, int H, int W,
                       const void *w_bf16x3, const float *scale, const float *shift, const float *res, long long res_bs,
                       float *y, int Cout, int KH, int KW, int stride, int pad, int flags, int plan, void *ws, size_t ws_bytes,
                       const PlaneOut *po, long long w_bs, float uscale) {
  SWEM_REQUIRE(x0 && w_bf16x3 && (y || (po && (po->planes[0] || po->planes[1]))), SWEM_E_ARG,
               "conv2d_bf16x3: null pointer (y may be NULL only when output planes are given)");
  if (!x1) c1 = 0;
  if (!x2) c2 = 0;
  SWEM_REQUIRE(!(x2 && !x1), SWEM_E_ARG, "conv2d_bf16x3: source 2 without source 1");
  SWEM_REQUIRE(c0 > 0 && c0 % 32 == 0 && c1 % 32 == 0 && c2 % 32 == 0, SWEM_E_SHAPE,
               "conv2d_bf16x3: every source needs a channel count that is a multiple of 32 (got %d,%d,%d)", c0, c1, c2);
  SWEM_REQUIRE(B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, SWEM_E_SHAPE,
               "conv2d_bf16x3: bad geometry");
  SWEM_REQUIRE(Cout > 0 && Cout % 4 == 0 && !(flags & SWEM_CONV_RELU_IN), SWEM_E_SHAPE,
               "conv2d_bf16x3: Cout %% 4 != 0, or RELU_IN (apply the ReLU when splitting the input)");
  const bool glu = flags & SWEM_CONV_GLU;
  SWEM_REQUIRE(!glu || (Cout % 32 == 0 && !res), SWEM_E_SHAPE, "conv2d_bf16x3: GLU needs Cout %% 32 == 0, no residual");
  ConvP p;
  p.sk_workers = 0; p.sk_mtiles = p.sk_ntiles = 0; p.sk_ws = nullptr; p.sk_flags = nullptr; p.fault = nullptr; p.spin_limit = 1 << 24; p.res_pl = nullptr; p.res_ps = 0; p.res_npx = 0; p.res_npl = 0;
  p.x[0] = p.x[1] = p.x[2] = nullptr;
  p.xs[0] = static_cast<const unsigned short *>(x0);
  p.xs[1] = static_cast<const unsigned short *>(x1 ? x1 : x0);
  p.xs[2] = static_cast<const unsigned short *>(x2 ? x2 : x0);
  p.ps[0] = ps0; p.ps[1] = x1 ? ps1 : ps0; p.ps[2] = x2 ? ps2 : ps0;
  p.c[0] = c0; p.c[1] = c1; p.c[2] = c2;
  p.bs[0] = bs0; p.bs[1] = bs1; p.bs[2] = bs2;
  for (int i = 0; i < 3; ++i) {
    p.bsp[i] = p.c[i] > 0 ? (int)(p.bs[i] / p.c[i]) : 0;
    p.npx[i] = p.c[i] > 0 ? (int)(p.ps[i] / p.c[i]) : 0;
  }
  SWEM_REQUIRE(ps0 * 6 < (1ll << 31) && p.ps[1] * 6 < (1ll << 31) && p.ps[2] * 6 < (1ll << 31), SWEM_E_SHAPE,
               "conv2d_bf16x3: a source exceeds the 2 GiB buffer-descriptor range");
  p.B = B; p.H = H; p.W = W;
  conv_out_dims(H, W, KH, KW, stride, pad, flags, p.Ho, p.Wo);
  SWEM_REQUIRE(p.Ho > 0 && p.Wo > 0, SWEM_E_SHAPE, "conv2d_bf16x3: empty output");
  SWEM_REQUIRE(!(flags & SWEM_CONV_DGRAD) || stride == 1 || stride == 2, SWEM_E_SHAPE,
               "conv2d_bf16x3: the data-gradient mode needs stride 1 or 2");
  p.Cin = c0 + c1 + c2;
  p.K = KH * KW * p.Cin;
  long long M = (long long)B * p.Ho * p.Wo;
  SWEM_REQUIRE(M < (1ll << 31), SWEM_E_SHAPE, "conv2d_bf16x3: too large");
  p.M = (int)M;
  p.w = nullptr; p.wsplit = static_cast<const unsigned short *>(w_bf16x3);
  p.scale = scale; p.uscale = uscale; p.shift = shift; p.res = res; p.res_bs = res_bs; p.w_bs = w_bs; p.y = y;
  p.Cout = Cout; p.Ncols = glu ? 2 * Cout : Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.flags = flags;
  p.nkb = cdiv(p.K, BK);
  fast_div_make((unsigned)(p.Ho * p.Wo), p.fd_howo_mul, p.fd_howo_sh);
  fast_div_make((unsigned)p.Wo, p.fd_wo_mul, p.fd_wo_sh);
  Plan pl = resolve_plan(plan, p.M, p.Ncols, p.nkb, glu);
  SWEM_REQUIRE(w_bs == 0 || (p.Ho * p.Wo) % 128 == 0, SWEM_E_SHAPE,
               "conv2d_bf16x3: per-batch filters need Ho*Wo (%d) to be a multiple of the 128-row tile", p.Ho * p.Wo);
  p.kb_per_split = pl.kb_per_split;
  p.partial = nullptr;
  p.mt0 = 0; p.part_m0 = 0; p.nplanes = 3; p.f16 = 0; p.xpn = 1;
  if (int prc = set_planes(p, po, glu, "conv2d_bf16x3")) return prc;
  // caller-owned counters: tile counters / stream-K flags (all zero between calls); the sticky fault word is the caller's too
  if (po && po->res_planes) {
    SWEM_REQUIRE(!res && !glu && !(flags & SWEM_CONV_MASK_POS) && Cout % 8 == 0, SWEM_E_ARG,
                 "conv2d_bf16x3: a plane residual replaces `res` (no GLU, no mask; Cout %% 8 == 0)");
    SWEM_REQUIRE(po->res_npl == SWEM_PLANES_F16 || po->res_npl == 3, SWEM_E_ARG,
                 "conv2d_bf16x3: a plane residual is an fp16 pair or three bf16 planes (two bf16 planes carry 16 bits only)");
    SWEM_REQUIRE(po->res_npx > 0 && po->res_ps >= po->res_npx * (long long)Cout && res_bs % Cout == 0, SWEM_E_ARG,
                 "conv2d_bf16x3: plane residual geometry");
    p.res_pl = static_cast<const unsigned short *>(po->res_planes);
    p.res_ps = po->res_ps; p.res_npx = (int)po->res_npx; p.res_npl = po->res_npl;
  }
  const size_t nctr = (po && po->counters) ? po->ncounters : 0;
  p.fault = po ? po->fault : nullptr;
  {
    static int limit = -1;   // (tests shorten the wait to see the fault path: tests/test_gpu_ops.py)
    if (limit < 0) {
      const char *e = getenv("SWEM_SPIN_LIMIT");
      limit = e ? atoi(e) : (1 << 24);
      if (limit < 0) limit = 0;
    }
    p.spin_limit = limit;
  }
  if (pl.nsplit > 1) {
    size_t need = (size_t)pl.nsplit * M * p.Ncols * sizeof(float);
    SWEM_REQUIRE(ws && ws_bytes >= need, SWEM_E_WORKSPACE, "conv2d_bf16x3: workspace %zu < %zu bytes", ws_bytes, need);
    p.partial = static_cast<float *>(ws);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  SWEM_REQUIRE(KH * KW <= 64, SWEM_E_SHAPE, "conv2d_bf16x3: at most 64 filter taps (one validity bit per tap and pixel)");
  const int variant = (plan >> 20) & 15;
  // math 2 = plain bf16 (mixed-precision training), 3 = "bf16x3" (hi + mid planes, three products: 16 significant bits per
  // operand, ~2^-16 relative error per product), else bf16x6
  p.nplanes = ((plan >> 16) & 3) == 2 ? 1 : (((plan >> 16) & 3) == 3 ? 2 : 3);
  // plan bit 18 (SWEM_PLAN_F16) with math 3: "f16x3" -- the planes of every source and of the filters are fp16 pairs
  // (swem_split_f16x2_f32; the filters scaled per output column by a power of two that the caller folds into `scale`)
  p.f16 = (p.nplanes == 2 && ((plan >> 18) & 1)) ? 1 : 0;
  SWEM_REQUIRE(!((plan >> 18) & 1) || p.nplanes == 2, SWEM_E_ARG, "conv2d_bf16x3: plan bit 18 (fp16 planes) needs math mode 3");
  SWEM_REQUIRE(pl.wm != 4 || ((p.f16 || (p.nplanes == 3 && variant == 4 && !glu)) && w_bs == 0 && ((plan >> 24) & 15) == 0), SWEM_E_ARG,
               "conv2d_bf16x3: the 256-column tiles run f16x3 plans (or bf16x6 with 128-row tiles, no GLU) without per-batch filters, tail split or stream-K");
  const int trows = tile_rows(plan, pl);
  SWEM_REQUIRE(pl.wm != 4 || (trows == 256 && variant == 0) || (trows >= 128 && trows < 256 && trows % 32 == 0 && !glu), SWEM_E_ARG,
               "conv2d_bf16x3: 256-column tile heights are 128, 160, 192, 224 (plan bits 20-23 = rows / 32; no GLU) or 256 (bits 0)");
  const int mtiles = cdiv(p.M, trows), ntiles = cdiv(p.Ncols, 64 * pl.wn);
  // XCD partition of the N tiles: plan bits 28-29 force 2 / 4 / 8 groups; 0 = the cut with the least fetch traffic by
  // the model  groups * activations + (8 / groups) * filters  (each XCD reads its groups' filters and its share of the
  // M tiles' activations once: measured 533 -> 166 MB on 2x30x54x1280 -> 512 together with the channel-block K order)
  p.xpn = 1 << ((plan >> 28) & 3);
  if (((plan >> 28) & 3) == 0) {
    static int forced = -1;
    if (forced < 0) {
      const char *e = getenv("SWEM_CONV_XPN");
      forced = e ? atoi(e) : 0;
    }
    if (forced > 0) {
      p.xpn = forced;
    } else {
      const double abytes = (double)B * H * W * p.Cin, wbytes = (double)p.Ncols * p.K;
      double best = abytes + 8.0 * wbytes;
      for (int pn = 2; pn <= 8; pn *= 2) {
        if (ntiles % pn) continue;
        const double cost = pn * abytes + (8.0 / pn) * wbytes;
        if (cost < 0.9 * best) {
          best = cost;
          p.xpn = pn;
        }
      }
    }
  }
  if (p.xpn != 1 && p.xpn != 2 && p.xpn != 4 && p.xpn != 8) p.xpn = 1;
  if (ntiles % p.xpn) p.xpn = 1;
  auto run = [&](const ConvP &q, dim3 grid, int *occ = nullptr) -> int {
    if (pl.wm == 4) {   // the 256 x 256 tile: its own kernel (f16x3 only), one block per CU
      if (occ) {
        *occ = 1;
        return SWEM_OK;
      }
      return swem_conv_t256_launch(trows, &q, grid.x, grid.y, grid.z, st);
    }
    if (pl.wm == 2 && pl.wn == 2) return launch_bf3s<2, 2>(q, grid, st, variant, occ);
    if (pl.wm == 1 && pl.wn == 2) return launch_bf3s<1, 2>(q, grid, st, variant, occ);
    if (pl.wm == 2 && pl.wn == 1) return launch_bf3s<2, 1>(q, grid, st, variant, occ);   // 128x64: the 64-channel layers
    return launch_bf3s<1, 1>(q, grid, st, variant, occ);
  };
  int rc;
  if (((plan >> 24) & 15) == 1 && w_bs == 0) {
    // stream-K (see the kernel): persistent workers, one per resident-block slot, equal shares of the tiles x k-blocks
    int nb = 1;
    p.sk_workers = -1;                           // (query the stream-K instantiation)
    if ((rc = run(p, dim3(1, 1, 1), &nb))) return rc;
    p.sk_workers = 0;
    const long long iters = (long long)mtiles * ntiles * p.nkb;
    long long W = (long long)nb * swem_device_cus();
    if (W > iters / 4) W = iters / 4;            // at least four k-blocks per worker
    if (W > SK_MAX_WORKERS) W = SK_MAX_WORKERS;
    if (W >= 2) {
      const size_t tile = (size_t)64 * pl.wm * 64 * pl.wn * sizeof(float);
      const size_t need = (size_t)W * tile + (size_t)W * sizeof(unsigned);
      SWEM_REQUIRE(ws && ws_bytes >= need, SWEM_E_WORKSPACE, "conv2d_bf16x3: workspace %zu < %zu bytes (stream-K)", ws_bytes, need);
      p.kb_per_split = p.nkb;
      p.partial = nullptr;
      p.sk_workers = (int)W; p.sk_mtiles = mtiles; p.sk_ntiles = ntiles;
      p.sk_ws = static_cast<float *>(ws);
      if (nctr >= (size_t)W) {
        p.sk_flags = po->counters;     // zero on entry, reset by the workers that consume them: no memset launch
      } else {
        p.sk_flags = reinterpret_cast<unsigned *>(static_cast<char *>(ws) + (size_t)W * tile);
        if (hipMemsetAsync(p.sk_flags, 0, (size_t)W * sizeof(unsigned), st) != hipSuccess) {
          swem_set_error("conv2d_bf16x3: hipMemsetAsync of the stream-K flags failed");
          return SWEM_E_HIP;
        }
      }
      static int dbg = -1;
      if (dbg < 0) dbg = getenv("SWEM_SK_DEBUG") ? 1 : 0;
      if (dbg) fprintf(stderr, "stream-K: %d x %d tiles x %d k-blocks on %lld workers (%d blocks per CU)\n", mtiles, ntiles, p.nkb, W, nb);
      if ((rc = run(p, dim3((unsigned)W, 1, 1)))) return rc;
      SWEM_CHECK_LAUNCH("conv_igemm_bf3s_kernel");
      return SWEM_OK;
    }
  }
  const TailSplit tl = tail_split(plan, pl, p.M, p.Ncols, p.nkb);
  if (tl.nsplit > 1) {
    if ((rc = run(p, dim3(tl.main_mt, ntiles, 1)))) return rc;      // the whole rounds
    ConvP t = p;                                                    // the last round's tile rows, split over K
    t.mt0 = tl.main_mt;
    t.part_m0 = tl.main_mt * 64 * pl.wm;
    t.kb_per_split = tl.kb_per_split;
    const size_t need = (size_t)tl.nsplit * (M - t.part_m0) * p.Ncols * sizeof(float);
    SWEM_REQUIRE(ws && ws_bytes >= need, SWEM_E_WORKSPACE, "conv2d_bf16x3: workspace %zu < %zu bytes", ws_bytes, need);
    t.partial = static_cast<float *>(ws);
    if ((rc = run(t, dim3(mtiles - tl.main_mt, ntiles, tl.nsplit)))) return rc;
    SWEM_CHECK_LAUNCH("conv_igemm_bf3s_kernel");
    const long long work = (M - t.part_m0) * (Cout / 4);
    hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3(cdiv(work, 256)), dim3(256), 0, st, t, tl.nsplit);
    SWEM_CHECK_LAUNCH("conv_splitk_epilogue_kernel");
    return SWEM_OK;
  }
  if (pl.nsplit > 1) {
    // K-split reduced by the last split (z = nsplit - 1) of every tile (see the kernel): partial TILES (padded), one counter per tile
    const size_t tile = (size_t)trows * 64 * pl.wn * sizeof(float);
    const size_t ntile = (size_t)mtiles * ntiles;
    const size_t need = (size_t)pl.nsplit * ntile * tile + ntile * sizeof(unsigned);
    SWEM_REQUIRE(ws && ws_bytes >= need, SWEM_E_WORKSPACE, "conv2d_bf16x3: workspace %zu < %zu bytes (fused K-split)", ws_bytes, need);
    p.partial = static_cast<float *>(ws);
    if (nctr >= ntile) {
      p.sk_flags = po->counters;       // zero on entry; the reducing split of a tile resets its counter: no memset launch
    } else {
      p.sk_flags = reinterpret_cast<unsigned *>(static_cast<char *>(ws) + (size_t)pl.nsplit * ntile * tile);
      if (hipMemsetAsync(p.sk_flags, 0, ntile * sizeof(unsigned), st) != hipSuccess) {
        swem_set_error("conv2d_bf16x3: hipMemsetAsync of the K-split counters failed");
        return SWEM_E_HIP;
      }
    }
    {
      // Staggered shares (round 4): equal shares make all splits of a tile finish together -- the producers' stores and the
      // reducer's loads are then a burst nothing hides (13 of 54 us on 2x30x54 k3 512->512 at nsp = 4, HISTORY.md).  The
      // producers get `skew` percent less than an equal share each, the reducer the rest: their tiles land while it still
      // multiplies.  SWEM_KSPLIT_SKEW overrides the default (0 = equal shares).
      static int skew = -1;
      if (skew < 0) {
        const char *e = getenv("SWEM_KSPLIT_SKEW");
        skew = e ? atoi(e) : SWEM_KSPLIT_SKEW_DEFAULT;
        if (skew < 0 || skew > 50) skew = 0;
      }
      int per = (int)((long long)p.nkb * (100 - skew) / (100LL * pl.nsplit));
      if (per < 1) per = 1;
      if ((long long)per * (pl.nsplit - 1) >= p.nkb) per = pl.kb_per_split;   // (too few k-blocks to skew)
      if (skew > 0 && per <= pl.kb_per_split) p.kb_per_split = per;
    }
    if ((rc = run(p, dim3(mtiles, ntiles, pl.nsplit)))) return rc;
    SWEM_CHECK_LAUNCH("conv_igemm_bf3s_kernel");
    return SWEM_OK;
  }
  if ((rc = run(p, dim3(mtiles, ntiles, pl.nsplit)))) return rc;
  SWEM_CHECK_LAUNCH("conv_igemm_bf3s_kernel");
  if (pl.nsplit > 1) {
    long long work = M * (Cout / 4);
    hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3(cdiv(work, 256)), dim3(256), 0, st, p, pl.nsplit);
    SWEM_CHECK_LAUNCH("conv_splitk_epilogue_kernel");
  }
  return SWEM_OK;
}
}  // namespace
#endif   // SWEM_CONV_T256_ONLY
