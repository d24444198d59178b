// Matching (reference methods/SWEM/modules.py:198-208, 232-289): three high-occupancy kernels + one batched GEMM.
//
// prep     : the two banks ('first', 'update') of each object are l2-normalised over C and packed as
//            mkn [N][Ltot][C] (row = class*Lm + bank*L + l, the order of get_mem + flatten, modules.py:266,304),
//            and the value bases as mvp [N][V][Ltot] (modules.py:272).
// affinity : block = (object, 32-pixel tile): l2-normalise the query pixels in LDS (modules.py:282), affinity
//            mkn . q on v_mfma_f32_32x32x2_f32 with the pixel on the MFMA lane and the base in the accumulator registers,
//            so the joint {bg,fg} max and exp-sum are in-register + one LDS exchange (modules.py:247-250, 265-266);
//            writes the probabilities pixel-major  pT [N][Pm][Ltot]  (13 MB at config B: L2 / Infinity-Cache resident).
// top-l    : one wave per (object, pixel): 64-lane bitonic sort of each class's Lm values, pairwise top-64 merges,
//            wave scan, feat = c_bg/(c_bg+c_fg) and its complement -> S [N][P][2*topl] (modules.py:198-208).
// readout  : mem_out[n] = pT[n] . mvp[n]^T (modules.py:272-274) is a batched GEMM run by the implicit-GEMM conv kernel
//            (1x1 "conv" over a Pm x 1 image with Ltot channels and per-object filters), output NHWC [N][Pm][V].
// (The first version kept the whole exp tile in 132 KiB of LDS inside one kernel; at one block per CU and one wave
//  per SIMD every global and LDS latency was exposed: 209 us per frame.  This split runs in a fraction of that.)
#include "../../include/swem_hip_train.h"
#include "common.h"
#include "bf16_split.h"

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

constexpr float MATCH_P_SCALE = 16384.f;   // the readout's probability planes hold p * 2^14 (see match_core)
// the readout GEMM's pre-split filters: mvq[n][plane][(cls*Lm + off + l)/8][v][l%8] = fp16 hi / mid of nu[n][cls][v][l]
__global__ void pack_value_planes_kernel(const float *__restrict__ nu, unsigned short *__restrict__ mvq, int N, int V, int L,
                                         int Lm, int off, unsigned *fault) {
  const int l8n = L / 8;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * 2 * V * l8n) return;
  const int v = (int)(i % V);     // value row fastest: consecutive threads store consecutive 16-byte runs
  long long t = i / V;
  const int l8 = (int)(t % l8n);
  t /= l8n;
  const int cls = (int)(t & 1), n = (int)(t >> 1);
  const float *src = nu + (((long long)n * 2 + cls) * V + v) * L + l8 * 8;
  uint2 h0, m0, h1, m1;
  const float4 s0 = ld4(src), s1 = ld4(src + 4);
  split2h(s0, h0, m0);
  split2h(s1, h1, m1);
  range_fault(fault, f16_oor(s0) | f16_oor(s1));
  const int ngrp = 2 * Lm / 8, kg = (cls * Lm + off) / 8 + l8;
  unsigned short *base = mvq + (long long)n * 2 * ngrp * V * 8;
  *reinterpret_cast<uint4 *>(base + ((long long)kg * V + v) * 8) = make_uint4(h0.x, h0.y, h1.x, h1.y);
  *reinterpret_cast<uint4 *>(base + ((long long)(ngrp + kg) * V + v) * 8) = make_uint4(m0.x, m0.y, m1.x, m1.y);
}

// Wave-wide bitonic sorting of NON-NEGATIVE floats (the exp values of matching): on their bit patterns an unsigned integer
// compare orders them like the float compare, without the canonicalising v_max x,x hipcc puts in front of fmaxf/fminf.
// The partner lane (lane ^ j) comes by DPP for j = 1, 2 (quad_perm), 4 (row_shl / row_shr on alternating banks) and 8
// (row_ror:8) -- vector-ALU moves, no LDS crossbar -- and by ds_bpermute only across the 16-lane rows (j = 16, 32):
// 18 of a 64-lane sort's 21 stages and 4 of a merge's 6 stay off the LDS.  (Round 1 issued one ds_bpermute + ~6 vector
// instructions per compare-exchange: the top-l kernel was bound by vector issue, ~2500 instructions per pixel.)
template <int J>
__device__ __forceinline__ unsigned partner(unsigned v) {
  if constexpr (J == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // [2,3,0,1]
  else if constexpr (J == 4) {
    int t = __builtin_amdgcn_update_dpp((int)v, (int)v, 0x104, 0xF, 0x5, false);   // banks 0, 2: lane i <- lane i + 4
    return (unsigned)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);   // banks 1, 3: lane i <- lane i - 4
  } else if constexpr (J == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);   // row_ror:8
  else return (unsigned)__shfl_xor((int)v, J);
}
template <int K, int J>
__device__ __forceinline__ unsigned cmpex(unsigned v, int lane) {
  const unsigned o = partner<J>(v);
  const bool keep_max = ((lane & K) == 0) == ((lane & J) == 0);   // K = 64 never matches a lane bit: descending overall
  const unsigned mx = v > o ? v : o, mn = v > o ? o : v;
  return keep_max ? mx : mn;
}
__device__ __forceinline__ float sort64_desc(float vf, int lane) {
  unsigned v = __float_as_uint(vf);
  v = cmpex<2, 1>(v, lane);
  v = cmpex<4, 2>(v, lane);
  v = cmpex<4, 1>(v, lane);
  v = cmpex<8, 4>(v, lane);
  v = cmpex<8, 2>(v, lane);
  v = cmpex<8, 1>(v, lane);
  v = cmpex<16, 8>(v, lane);
  v = cmpex<16, 4>(v, lane);
  v = cmpex<16, 2>(v, lane);
  v = cmpex<16, 1>(v, lane);
  v = cmpex<32, 16>(v, lane);
  v = cmpex<32, 8>(v, lane);
  v = cmpex<32, 4>(v, lane);
  v = cmpex<32, 2>(v, lane);
  v = cmpex<32, 1>(v, lane);
  v = cmpex<64, 32>(v, lane);
  v = cmpex<64, 16>(v, lane);
  v = cmpex<64, 8>(v, lane);
  v = cmpex<64, 4>(v, lane);
  v = cmpex<64, 2>(v, lane);
  v = cmpex<64, 1>(v, lane);
  return __uint_as_float(v);
}
// a, b descending across the wave -> the 64 largest of their union, descending
__device__ __forceinline__ float merge_top64(float a, float b, int lane) {
  const unsigned ua = __float_as_uint(a), ub = (unsigned)__shfl((int)__float_as_uint(b), 63 - lane);
  unsigned v = ua > ub ? ua : ub;
  v = cmpex<64, 32>(v, lane);
  v = cmpex<64, 16>(v, lane);
  v = cmpex<64, 8>(v, lane);
  v = cmpex<64, 4>(v, lane);
  v = cmpex<64, 2>(v, lane);
  v = cmpex<64, 1>(v, lane);
  return __uint_as_float(v);
}

// The same networks on G independent rows of TWc 64-lane chunks at once, STAGE BY STAGE over all chunks: a compare-exchange
// reads the register the previous stage of the same chunk wrote through DPP, which costs two wait states when the two are
// adjacent (574 s_nop in the affinity kernel's 6.2k instructions); with the other chunks' stage in between none is needed.
// Result: the 64 largest of row g, descending across the wave, in v[g * TWc].  Bit-identical to sort64_desc / merge_top64.
template <int K, int J, int N>
__device__ __forceinline__ void cmpex_all(unsigned (&v)[N], int lane) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = cmpex<K, J>(v[i], lane);
}
template <int G, int TWc>
__device__ __forceinline__ void top64_rows(float (&vf)[G * TWc], int lane) {
  constexpr int N = G * TWc;
  unsigned v[N];
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __float_as_uint(vf[i]);
  cmpex_all<2, 1>(v, lane);
  cmpex_all<4, 2>(v, lane); cmpex_all<4, 1>(v, lane);
  cmpex_all<8, 4>(v, lane); cmpex_all<8, 2>(v, lane); cmpex_all<8, 1>(v, lane);
  cmpex_all<16, 8>(v, lane); cmpex_all<16, 4>(v, lane); cmpex_all<16, 2>(v, lane); cmpex_all<16, 1>(v, lane);
  cmpex_all<32, 16>(v, lane); cmpex_all<32, 8>(v, lane); cmpex_all<32, 4>(v, lane); cmpex_all<32, 2>(v, lane);
  cmpex_all<32, 1>(v, lane);
  cmpex_all<64, 32>(v, lane); cmpex_all<64, 16>(v, lane); cmpex_all<64, 8>(v, lane); cmpex_all<64, 4>(v, lane);
  cmpex_all<64, 2>(v, lane); cmpex_all<64, 1>(v, lane);
#pragma unroll
  for (int w = 1; w < TWc; w <<= 1) {
    // the pairs (j, j + w) of every row: first the cross-over max of all pairs, then the six merge stages of all pairs
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j + w < TWc; j += 2 * w) {
        const unsigned ub = (unsigned)__shfl((int)v[g * TWc + j + w], 63 - lane), ua = v[g * TWc + j];
        v[g * TWc + j] = ua > ub ? ua : ub;
      }
#define SWEM_MERGE_STAGE(J_)                                                                     \
    _Pragma("unroll") for (int g = 0; g < G; ++g)                                                \
      _Pragma("unroll") for (int j = 0; j + w < TWc; j += 2 * w) v[g * TWc + j] = cmpex<64, J_>(v[g * TWc + j], lane)
    SWEM_MERGE_STAGE(32); SWEM_MERGE_STAGE(16); SWEM_MERGE_STAGE(8); SWEM_MERGE_STAGE(4); SWEM_MERGE_STAGE(2); SWEM_MERGE_STAGE(1);
#undef SWEM_MERGE_STAGE
  }
#pragma unroll
  for (int g = 0; g < G; ++g) vf[g * TWc] = __uint_as_float(v[g * TWc]);
}

// K1: affinity + joint softmax, pT[n][p][l] = exp((aff - max)/tau) / sum  (pixel-major, one row of Ltot probabilities per
// pixel; rows >= P are 0).  16-pixel tiles (the shape of em.hip's E/W kernel): block = (object, 16-pixel tile), 8 waves; wave w owns the
// 16*TW bases [w*16*TW, +16*TW) of the concatenated bank order (class w / 4).  v_mfma_f32_16x16x4_f32 with the base rows as
// A and the pixels as B: the pixel is on the lane, four bases of a tile in the accumulator registers.  The query pixel's
// key stays in registers (normalised there: modules.py:282); every base row is loaded once per block, all loads of a pass
// issued before its MFMAs.  Twice the blocks of the 32-pixel kernel (204 at config B) and half the MFMA chain per wave.
typedef unsigned u32x4m __attribute__((ext_vector_type(4)));
typedef float f32x4m __attribute__((ext_vector_type(4)));
template <int TW, int CM>  // TW 16-base tiles per wave (Ltot = 128 * TW), C = 16 * CM
__global__ __launch_bounds__(512) void match_affinity16_kernel(const float *__restrict__ qk, const float *__restrict__ mkn,
                                                               float *__restrict__ pT, unsigned short *__restrict__ pq,
                                                               float *__restrict__ S, unsigned short *__restrict__ sq, int sq_npl,
                                                               int topl, int P, int Pm, float tau, int xg) {
  constexpr int C = 16 * CM, Ltot = 128 * TW, Lm = Ltot / 2;
  constexpr int TP = TW > 4 ? 4 : TW, NPASS = TW / TP;   // tiles per pass: at most 4 (32 x 16 bytes of base rows in flight)
  __shared__ float red[2][8][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int n = blockIdx.y, p = blockIdx.x * 16 + li;
  const int cls = wave >> 2, wq = wave & 3;
  // (xg objects share one query key map: object n of a batch of clips reads clip n / xg's -- round 6)
  __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(qk + (long long)(n / xg) * P * C), 0, P * C * 4,
                                                                  0x00020000);
  float4 xf[CM];
#pragma unroll
  for (int m = 0; m < CM; ++m) {
    u32x4m t = __builtin_amdgcn_raw_buffer_load_b128(rq, (unsigned)((p * C + 16 * m + 4 * g) * 4), 0, 0);
    xf[m] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
  }
  const float *kb = mkn + (((long long)(2 * n + cls) * (C / 4 + 1) + g) * Lm + wq * 16 * TW + li) * 4;
  float4 a[CM][TP];
#pragma unroll
  for (int m = 0; m < CM; ++m)
#pragma unroll
    for (int t = 0; t < TP; ++t) a[m][t] = ld4(kb + ((long long)4 * m * Lm + 16 * t) * 4);
  float4 nq[TW];   // group C/4 of the pack: the rows' squared norms as partial sums
#pragma unroll
  for (int t = 0; t < TW; ++t) nq[t] = ld4(kb + ((long long)(C / 4 - g) * Lm + 16 * t) * 4);
  __builtin_amdgcn_sched_barrier(0);   // every load above the MFMA chain (em.hip, em_ew16_kernel)
  // l2norm of the query pixel (modules.py:282): q / (|q| + eps), on the registers
  float ss = 0.f;
#pragma unroll
  for (int m = 0; m < CM; ++m) ss += (xf[m].x * xf[m].x + xf[m].y * xf[m].y) + (xf[m].z * xf[m].z + xf[m].w * xf[m].w);
  ss += __shfl_xor(ss, 16);
  ss += __shfl_xor(ss, 32);
  const float den = sqrtf(ss) + SWEM_L2_EPS;
#pragma unroll
  for (int m = 0; m < CM; ++m) {
    xf[m].x /= den;
    xf[m].y /= den;
    xf[m].z /= den;
    xf[m].w /= den;
  }
  f32x4m acc[TW];
#pragma unroll
  for (int t = 0; t < TW; ++t) acc[t] = f32x4m{0.f, 0.f, 0.f, 0.f};
  // l2norm of the bases (modules.py:283): every lane scales its rows' fragments by 1 / (|row| + eps), from the squared
  // norms kept beside the packed keys, on the way into the MFMA chain (em.hip, em_ew16_kernel, says why not on the output)
  float rn[TW];
#pragma unroll
  for (int t = 0; t < TW; ++t) rn[t] = 1.0f / (sqrtf((nq[t].x + nq[t].y) + (nq[t].z + nq[t].w)) + SWEM_L2_EPS);
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
    for (int m = 0; m < CM; ++m) {
#pragma unroll
      for (int t = 0; t < TP; ++t) {
        const float r = rn[ps * TP + t];
        a[m][t].x *= r, a[m][t].y *= r, a[m][t].z *= r, a[m][t].w *= r;
      }
#pragma unroll
      for (int t = 0; t < TP; ++t)
        acc[ps * TP + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][t].x, xf[m].x, acc[ps * TP + t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TP; ++t)
        acc[ps * TP + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][t].y, xf[m].y, acc[ps * TP + t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TP; ++t)
        acc[ps * TP + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][t].z, xf[m].z, acc[ps * TP + t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TP; ++t)
        acc[ps * TP + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][t].w, xf[m].w, acc[ps * TP + t], 0, 0, 0);
      if (ps + 1 < NPASS) {   // the next pass's rows of this chunk take the registers just consumed
#pragma unroll
        for (int t = 0; t < TP; ++t) a[m][t] = ld4(kb + ((long long)4 * m * Lm + 16 * ((ps + 1) * TP + t)) * 4);
      }
    }
  }
  // joint {bg, fg} softmax over all Ltot bases (modules.py:248-250, 265-266)
  float ml = -__builtin_huge_valf();
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) ml = fmaxf(ml, acc[t][e]);
  ml = fmaxf(ml, __shfl_xor(ml, 16));
  ml = fmaxf(ml, __shfl_xor(ml, 32));
  if (g == 0) red[0][wave][li] = ml;
  __syncthreads();
  float mx = red[0][0][li];
#pragma unroll
  for (int q = 1; q < 8; ++q) mx = fmaxf(mx, red[0][q][li]);
  const float k2 = SWEM_LOG2E / tau;
  float se = 0.f;
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = exp_scaled(acc[t][e] - mx, k2);
      acc[t][e] = v;
      se += v;
    }
  se += __shfl_xor(se, 16);
  se += __shfl_xor(se, 32);
  if (g == 0) red[1][wave][li] = se;
  __syncthreads();
  const float esum = ((red[1][0][li] + red[1][1][li]) + (red[1][2][li] + red[1][3][li])) +
                     ((red[1][4][li] + red[1][5][li]) + (red[1][6][li] + red[1][7][li]));
  const float inv = p < P ? 1.0f / esum : 0.f;   // rows of pad pixels are written as zeros
#pragma unroll
  for (int t = 0; t < TW; ++t) acc[t][0] *= inv, acc[t][1] *= inv, acc[t][2] *= inv, acc[t][3] *= inv;
  if (pT) {
    float *dst = pT + ((long long)n * Pm + p) * Ltot + wave * 16 * TW + 4 * g;
#pragma unroll
    for (int t = 0; t < TW; ++t)
      *reinterpret_cast<float4 *>(dst + 16 * t) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  }
  if (S) {
    // Top-l prefix features (modules.py:198-208) of the tile's 16 pixels without the round trip of the probabilities
    // through memory: class by class the tile turns through the LDS (pixel on the lane -> pixel per wave), wave w sorts
    // pixels 2w and 2w+1 exactly as match_topl_kernel does (the same network on the same values: bit-identical features).
    __shared__ __attribute__((aligned(16))) float pl[16][Lm + 4];
    float cum[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      __syncthreads();   // the previous class has been read
      if (cls == c) {
        float *d = &pl[li][wq * 16 * TW + 4 * g];
#pragma unroll
        for (int t = 0; t < TW; ++t) *reinterpret_cast<float4 *>(d + 16 * t) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
      }
      __syncthreads();
      {
        float v[2 * TW];   // both pixels of the wave: 2 TW independent chunks through the networks together
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
          for (int j = 0; j < TW; ++j) v[sp * TW + j] = pl[2 * wave + sp][lane + 64 * j];
        top64_rows<2, TW>(v, lane);
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          float cs = v[sp * TW];
#pragma unroll
          for (int dd = 1; dd < 64; dd <<= 1) {
            const float tt = __shfl_up(cs, dd);
            if (lane >= dd) cs += tt;
          }
          cum[sp][c] = cs;
        }
      }
    }
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      const int pp = blockIdx.x * 16 + 2 * wave + sp;
      if (pp < P && lane < topl) {
        const float f = cum[sp][0] / (cum[sp][0] + cum[sp][1]);
        float *dst = S + ((long long)n * P + pp) * (2 * topl);
        dst[lane] = f;
        dst[topl + lane] = 1.f - f;
        if (sq) {
          // S's own bf16 planes [2 topl / 8][N * P][8] for the pre-split convolution that consumes it (modules.py:288-289):
          // channel c of pixel gp is element ((c / 8) * npix + gp) * 8 + c % 8 of a plane
          uint2 h, m, lo;
          split_as(sq_npl, make_float4(f, 1.f - f, 0.f, 0.f), h, m, lo);
          const long long npix = (long long)gridDim.y * P, gp = (long long)n * P + pp, plane = npix * 2 * topl;
          const long long i0 = ((long long)(lane >> 3) * npix + gp) * 8 + (lane & 7);
          const long long i1 = ((long long)((topl + lane) >> 3) * npix + gp) * 8 + ((topl + lane) & 7);
          sq[i0] = (unsigned short)h.x, sq[i1] = (unsigned short)(h.x >> 16);
          sq[plane + i0] = (unsigned short)m.x, sq[plane + i1] = (unsigned short)(m.x >> 16);
          if (sq_npl == 3) sq[2 * plane + i0] = (unsigned short)lo.x, sq[2 * plane + i1] = (unsigned short)(lo.x >> 16);
        }
      }
    }
  }
  if (pq) {
    // the same probabilities (times 2^14) as the readout GEMM's pre-split operand: fp16 planes hi / mid, [plane][Ltot/8][N*Pm][8]
    // (conv.hip's activation layout).  The lane's four bases are half of an 8-channel group: 8 bytes per plane and tile,
    // the 16 pixels x 2 halves of a group one contiguous 256-byte run.
    const long long npix = (long long)gridDim.y * Pm, gp = (long long)n * Pm + p;
    unsigned short *d0 = pq + (((long long)(wave * 2 * TW + (g >> 1)) * npix + gp) * 8 + 4 * (g & 1));
    const long long plane = npix * Ltot;
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      uint2 h, m;
      split2h(make_float4(acc[t][0] * MATCH_P_SCALE, acc[t][1] * MATCH_P_SCALE, acc[t][2] * MATCH_P_SCALE, acc[t][3] * MATCH_P_SCALE), h, m);
      *reinterpret_cast<uint2 *>(d0 + (long long)2 * t * npix * 8) = h;
      *reinterpret_cast<uint2 *>(d0 + plane + (long long)2 * t * npix * 8) = m;
    }
  }
}

// K3: top-l prefix features (modules.py:198-208).  One wave per (object, pixel): the Lm probabilities of each class are
// sorted across the wave (bitonic network on J registers per lane + pairwise top-64 merges), a wave scan gives the
// prefix sums and feat = c_bg / (c_bg + c_fg) (invariant to the common 1/sum factor of the row).
template <int J>
__global__ __launch_bounds__(256) void match_topl_kernel(const float *__restrict__ pT, float *__restrict__ S, int N,
                                                         int P, int Pm, int topl) {
  constexpr int Ltot = 128 * J, Lm = 64 * J;
  const int lane = threadIdx.x & 63;
  const long long idx = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= (long long)N * P) return;
  const int n = (int)(idx / P), pix = (int)(idx - (long long)n * P);
  const float *row = pT + ((long long)n * Pm + pix) * Ltot;
  float cum[2];
#pragma unroll
  for (int cls = 0; cls < 2; ++cls) {
    float v[J];
#pragma unroll
    for (int j = 0; j < J; ++j) v[j] = row[cls * Lm + lane + 64 * j];
    top64_rows<1, J>(v, lane);
    float c = v[0];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      float t = __shfl_up(c, d);
      if (lane >= d) c += t;
    }
    cum[cls] = c;
  }
  if (lane < topl) {
    float f = cum[0] / (cum[0] + cum[1]);
    float *dst = S + idx * (2 * topl);
    dst[lane] = f;
    dst[topl + lane] = 1.f - f;
  }
}

// Lm bases per class -> (tiles per wave J, waves NW) of the affinity kernel: blocks of 8 waves from 256 bases per class on
// (there are only Pm/32 x N blocks, so a block's latency is the kernel's time)
// S != NULL: the top-l features come out of the same launch (pT may then be NULL: nothing else reads the probabilities)
static int launch_affinity(hipStream_t st, const float *qk, const float *mkn, float *pT, unsigned short *pq, float *S,
                           unsigned short *sq, int sq_npl, int topl, int N, int C, int P, int Pm, int Lm, float tau, int xg = 1 << 30) {
  if (C == 128 || C == 64) {
    // the grid covers all Pm rows of pT: the readout GEMM's last row tile reads rows [P, Pm), which pad tiles write as zeros
    dim3 grid16(Pm / 16, N);
#define AFF16(TW_, CM_)                                                                                              \
  hipLaunchKernelGGL((match_affinity16_kernel<TW_, CM_>), grid16, dim3(512), 0, st, qk, mkn, pT, pq, S, sq, sq_npl, topl, P, Pm, tau, xg)
    if (C == 128) {
      if (Lm == 64) AFF16(1, 8);
      else if (Lm == 128) AFF16(2, 8);
      else if (Lm == 256) AFF16(4, 8);
      else if (Lm == 512) AFF16(8, 8);
      else {
        swem_set_error("match: bases per class must be 64, 128, 256 or 512 (got %d)", Lm);
        return SWEM_E_SHAPE;
      }
    } else {
      if (Lm == 64) AFF16(1, 4);
      else if (Lm == 128) AFF16(2, 4);
      else if (Lm == 256) AFF16(4, 4);
      else if (Lm == 512) AFF16(8, 4);
      else {
        swem_set_error("match: bases per class must be 64, 128, 256 or 512 (got %d)", Lm);
        return SWEM_E_SHAPE;
      }
    }
#undef AFF16
    return SWEM_OK;
  }
  swem_set_error("match: the key dimension must be 64 or 128 (got %d)", C);
  return SWEM_E_SHAPE;
}

struct MatchWs {
  size_t mkn, mvp, pT, pq, conv, total;
};
MatchWs match_ws(int N, int C, int V, int P, int L, int nbanks, int plan) {
  MatchWs w;
  const size_t Ltot = (size_t)2 * nbanks * L;
  const int Pm = swem_match_pad(P);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes, 256);
    return at;
  };
  w.mkn = take((size_t)N * Ltot * (C + 4) * 4);   // packed keys: C/4 + 1 groups
  w.mvp = take((size_t)N * V * Ltot * 4);
  w.pT = take((size_t)N * Pm * Ltot * 4);
  w.pq = take((size_t)N * Pm * Ltot * 4);   // the probabilities again as two bf16 planes (pre-split readout)
  w.conv = take(swem_conv2d_workspace(N, Pm, 1, (int)Ltot, V, 1, 1, 1, 0, 0, plan));
  w.total = o;
  return w;
}

// ------------------------------------------------------------------------------------------------ backward (training)
// dnu[n][cls][v][l] = dmvp[n][v][cls*Lm + off + l]   (inverse of pack_values_kernel)
__global__ void unpack_values_kernel(const float *__restrict__ dmvp, float *__restrict__ dnu, int N, int V, int L,
                                     int Lm, int off) {
  const int lq = L / 4;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)N * 2 * V * lq) return;
  int l4 = (int)(i % lq);
  long long t = i / lq;
  int v = (int)(t % V);
  t /= V;
  int cls = (int)(t & 1);
  int n = (int)(t >> 1);
  *reinterpret_cast<float4 *>(dnu + i * 4) = ld4(dmvp + ((long long)n * V + v) * (2 * Lm) + cls * Lm + off + l4 * 4);
}

// Per (object, pixel), one wave: gradient at the affinities from (a) the readout path (dP = dmem . mv, computed by a
// GEMM before this kernel) and (b) the top-l prefix features: the class's values are sorted as in the forward kernel,
// d f_i -> d c_i -> suffix sums R_j = sum_{i >= j} d c_i, and every element takes R at its rank (found by a binary
// search in the sorted top-64 list; elements below the list get none).  Then the joint-softmax Jacobian:
// da_l = p_l * (g_l - sum_j p_j g_j) / tau.  The result overwrites dP.
template <int J>
__global__ __launch_bounds__(256) void match_bwd_pixel_kernel(const float *__restrict__ pT, float *__restrict__ dP,
                                                              const float *__restrict__ dS, int N, int P, int Pm,
                                                              int topl, float inv_tau) {
  constexpr int Ltot = 128 * J, Lm = 64 * J;
  __shared__ float sv[4][2][64], sr[4][2][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long idx = (long long)blockIdx.x * 4 + w;
  if (idx >= (long long)N * P) return;
  const int n = (int)(idx / P), pix = (int)(idx - (long long)n * P);
  const float *row = pT + ((long long)n * Pm + pix) * Ltot;
  float *drow = dP + ((long long)n * Pm + pix) * Ltot;
  float pv[2][J], cum[2];
#pragma unroll
  for (int cls = 0; cls < 2; ++cls) {
    float v[J];
#pragma unroll
    for (int j = 0; j < J; ++j) pv[cls][j] = v[j] = row[cls * Lm + lane + 64 * j];
    top64_rows<1, J>(v, lane);
    float c = v[0];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      float t = __shfl_up(c, d);
      if (lane >= d) c += t;
    }
    cum[cls] = c;
    sv[w][cls][lane] = v[0];
  }
  const float df = (dS && lane < topl) ? dS[idx * (2 * topl) + lane] - dS[idx * (2 * topl) + topl + lane] : 0.f;
  const float den = cum[0] + cum[1];
  float rb = lane < topl ? df * cum[1] / (den * den) : 0.f;
  float rf = lane < topl ? -df * cum[0] / (den * den) : 0.f;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float tb = __shfl_down(rb, d), tf = __shfl_down(rf, d);
    if (lane + d < 64) {
      rb += tb;
      rf += tf;
    }
  }
  sr[w][0][lane] = rb;
  sr[w][1][lane] = rf;
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed before it reads them back
  float g[2][J];
  float dot = 0.f;
#pragma unroll
  for (int cls = 0; cls < 2; ++cls)
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float x = pv[cls][j];
      int lo = 0, hi = 64;
#pragma unroll
      for (int it = 0; it < 7; ++it) {   // [0, 64) halves to empty in 7 steps; entries [0, lo) are > x
        if (lo >= hi) break;
        const int mid = (lo + hi) >> 1;
        if (sv[w][cls][mid] > x) lo = mid + 1;
        else hi = mid;
      }
      const float gs = lo < topl ? sr[w][cls][lo] : 0.f;
      const float gg = gs + drow[cls * Lm + lane + 64 * j];
      g[cls][j] = gg;
      dot += x * gg;
    }
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
#pragma unroll
  for (int cls = 0; cls < 2; ++cls)
#pragma unroll
    for (int j = 0; j < J; ++j) drow[cls * Lm + lane + 64 * j] = pv[cls][j] * (g[cls][j] - dot) * inv_tau;
}

// mknT[c][n*Ltot + cls*Lm + l] = mkn[2n + cls][c/4][l][c%4]: the filters of the d qn GEMM (reduction axis contiguous)
__global__ void kn_cmajor_kernel(const float *__restrict__ mkn, float *__restrict__ mknT, int N, int C, int Lm) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long per = (long long)2 * N * Lm;
  if (i >= per * C) return;
  const int c = (int)(i / per);
  const long long rowg = i - (long long)c * per;          // n*Ltot + cls*Lm + l = nk*Lm + l
  const int nk = (int)(rowg / Lm), l = (int)(rowg - (long long)nk * Lm);
  mknT[i] = mkn[(((long long)nk * (C / 4) + c / 4) * Lm + l) * 4 + (c & 3)];
}

// l2norm backward (modules.py:7-9,282): qn = q / (|q| + eps);  dq = g / (|q| + eps) - q (q.g) / (|q| (|q| + eps)^2)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float *__restrict__ q, const float *__restrict__ g,
                                                         float *__restrict__ dq, int P, int C) {
  const int lane = threadIdx.x & 63;
  const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= P) return;
  float s = 0.f, d = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float a = q[pix * C + c], b = g[pix * C + c];
    s += a * a;
    d += a * b;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    d += __shfl_xor(d, o);
  }
  const float nrm = sqrtf(s), den = nrm + SWEM_L2_EPS;
  const float k = nrm > 0.f ? d / (nrm * den * den) : 0.f;
  for (int c = lane; c < C; c += 64) dq[pix * C + c] = g[pix * C + c] / den - q[pix * C + c] * k;
}

struct MatchBwdWs {
  size_t fwd, mknn, mvpT, dP, mknT, dmvp, dqn, conv, wgrad, total;
};
MatchBwdWs match_bwd_ws(int N, int C, int V, int P, int L, int nbanks) {
  MatchBwdWs w;
  const size_t Ltot = (size_t)2 * nbanks * L;
  const int Pm = swem_match_pad(P);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    size_t at = o;
    o = align_up(o + bytes, 256);
    return at;
  };
  w.fwd = take(match_ws(N, C, V, P, L, nbanks, 0).total);
  w.mknn = take((size_t)N * Ltot * C * 4);
  w.mvpT = take((size_t)N * Ltot * V * 4);
  w.dP = take((size_t)N * Pm * Ltot * 4);
  w.mknT = take((size_t)2 * C * N * Ltot * 4);   // + room for one group's packed filters when N > 3
  w.dmvp = take((size_t)N * V * Ltot * 4);
  w.dqn = take((size_t)Pm * C * 4);
  size_t c1 = swem_conv2d_workspace(N, Pm, 1, V, (int)Ltot, 1, 1, 1, 0, 0, 0);
  size_t c2 = swem_conv2d_workspace(1, Pm, 1, (int)((N < 3 ? N : 3) * Ltot), C, 1, 1, 1, 0, 0, 0);
  w.conv = take(c1 > c2 ? c1 : c2);
  w.wgrad = take(swem_conv2d_wgrad_workspace(1, Pm, 1, (int)Ltot, 0, 0, V, 1, 1, 1, 0));
  w.total = o;
  return w;
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" int swem_match_pad(int P) { return (P + 127) / 128 * 128; }

extern "C" size_t swem_match_workspace(int N, int C, int V, int P, int L, int nbanks, int readout_plan) {
  return match_ws(N, C, V, P, L, nbanks, readout_plan).total;
}

namespace {
// affinity + top-l features + readout on PACKED banks: mkn [2N][C/4+1][Lm][4], mvp [N][V][2Lm]
int match_core(void *stream, const float *qk, const float *mkn, const float *mvp, const unsigned short *mvq, float *pT,
               unsigned short *pq, float *mem_out, float *S, int N, int C, int V, int P, int Lm, int topl, float tau,
               int readout_plan, void *conv_ws, size_t conv_bytes, void *mem_planes = nullptr, int mem_npl = 3,
               void *s_planes = nullptr, int s_npl = 3, void *fault = nullptr, int xg = 1 << 30) {
  const int Pm = swem_match_pad(P), Ltot = 2 * Lm;
  int rc;
  dim3 gridt(cdiv((long long)N * P, 4));
  // readout on pre-split planes ("bf16x3": hi + mid planes, three bf16 products -- the arithmetic of the convolutions that
  // consume mem_out) when the plan asks for it and the caller keeps the value planes
  const bool presplit = mvq && pq && ((readout_plan >> 16) & 3) == 3;
  SWEM_REQUIRE(presplit || (!mem_planes && !s_planes), SWEM_E_ARG,
               "match: output planes come with the pre-split readout only (value planes + readout plan math 3)");
  if (presplit) {
    // one launch: affinity + softmax, the probabilities as bf16 planes for the readout, and the top-l features
    if ((rc = launch_affinity(ST, qk, mkn, nullptr, pq, S, static_cast<unsigned short *>(s_planes), s_npl, topl, N, C, P, Pm, Lm, tau, xg))) return rc;
    SWEM_CHECK_LAUNCH("match_affinity (fused top-l)");
  } else {
    if ((rc = launch_affinity(ST, qk, mkn, pT, nullptr, nullptr, nullptr, 3, 0, N, C, P, Pm, Lm, tau, xg))) return rc;
#define TOPL(J_) hipLaunchKernelGGL((match_topl_kernel<J_>), gridt, dim3(256), 0, ST, pT, S, N, P, Pm, topl)
    if (Lm == 64) TOPL(1);
    else if (Lm == 128) TOPL(2);
    else if (Lm == 256) TOPL(4);
    else TOPL(8);
#undef TOPL
    SWEM_CHECK_LAUNCH("match_affinity / match_topl");
  }
  // value readout (modules.py:272-273) = batched GEMM  mem_out[n] = pT[n] . mvp[n]^T  on the conv kernel:
  // "image" of Pm x 1 pixels with Ltot channels, 1x1 filters = the V value rows of object n (w_bs = V*Ltot)
  // The pre-split readout runs the f16x3 arithmetic whatever the plan's math field says beyond "pre-split" (round 4): the
  // pack's value planes are fp16 pairs (em.hip, pack_value_planes_kernel) and the affinity kernel wrote the probabilities as
  // the fp16 pair of p * 2^14 (p <= 1: most of a row is far below 2^-2, where an unscaled `mid` would be subnormal); the
  // epilogue multiplies by 2^-14 (exact).  1e-7 from the fp32 readout, where the bf16 (hi, mid) planes of round 3 gave 3e-6.
  if (presplit) {
    // (the f16x3 kernels have no stream-K form: a plan tuned for the bf16x3 readout of round 3 with plan bits 24-27 == 1 runs
    // the plain grid of its tile instead of failing at launch -- ADVICE r04; tail-split factors 2..15 stay)
    int plan = (readout_plan & ~(3 << 16)) | (3 << 16) | SWEM_PLAN_F16;
    if (((plan >> 24) & 15) == 1) plan &= ~(15 << 24);
    // (the probability planes hold p * 2^14 <= 2^14 and the top-l features lie in [0, 1]: only mem_out's own planes -- the
    // readout's epilogue -- can leave the fp16 range)
    return swem_gemm_bf16x3_batched(stream, pq, Ltot, (long long)Pm * Ltot, (long long)N * Pm * Ltot, N, Pm, mvq,
                                    (long long)2 * V * Ltot, mem_out, V, plan, conv_ws, conv_bytes, mem_planes, mem_npl,
                                    1.f / MATCH_P_SCALE, fault);
  }
  return swem_conv2d_nhwc_f32(stream, pT, Ltot, (long long)Pm * Ltot, nullptr, 0, 0, nullptr, 0, 0, N, Pm, 1, mvp,
                              (long long)V * Ltot, nullptr, nullptr, nullptr, 0, mem_out, V, 1, 1, 1, 0, 0,
                              readout_plan, conv_ws, conv_bytes);
}

int match_check(int C, int V, int L, int Lm, int topl, float tau) {
  SWEM_REQUIRE(Lm == 64 || Lm == 128 || Lm == 256 || Lm == 512, SWEM_E_SHAPE,
               "match: bases per class must be 64, 128, 256 or 512 (got %d)", Lm);
  SWEM_REQUIRE((C == 64 || C == 128) && V % 4 == 0 && L % 4 == 0, SWEM_E_SHAPE,
               "match: need C = 64 or 128 and V %% 4 == 0 (got C = %d, V = %d)", C, V);
  SWEM_REQUIRE(topl >= 1 && topl <= 64 && topl <= Lm, SWEM_E_SHAPE, "match: topl must be in [1, 64] (got %d)", topl);
  SWEM_REQUIRE(tau > 0.f, SWEM_E_ARG, "match: tau must be positive");
  return SWEM_OK;
}
}  // namespace

// one bank's bases into the packed form matching reads (modules.py:295-306 `get_mem` + the l2norm of :283)
extern "C" int swem_match_pack_bank_f32(void *stream, const float *kappa, const float *nu, float *mkn, float *mvp,
                                        void *mvq, int bank, int nbanks, int N, int C, int V, int L, void *fault) {
  SWEM_REQUIRE(kappa && nu && mkn && mvp, SWEM_E_ARG, "match_pack_bank: null pointer");
  SWEM_REQUIRE(nbanks >= 1 && nbanks <= 2 && bank >= 0 && bank < nbanks, SWEM_E_ARG, "match_pack_bank: bad bank index");
  const int Lm = nbanks * L;
  int rc;
  if ((rc = swem_norm_bases_into(stream, kappa, mkn, 2 * N, C, L, Lm, bank * L, 0))) return rc;
  const long long work = (long long)N * 2 * V * (L / 4);
  hipLaunchKernelGGL(pack_values_kernel, dim3(cdiv(work, 256)), dim3(256), 0, ST, nu, mvp, N, V, L, Lm, bank * L);
  if (mvq) {
    SWEM_REQUIRE(L % 8 == 0, SWEM_E_SHAPE, "match_pack_bank: value planes need L %% 8 == 0");
    hipLaunchKernelGGL(pack_value_planes_kernel, dim3(cdiv(work / 2, 256)), dim3(256), 0, ST, nu,
                       static_cast<unsigned short *>(mvq), N, V, L, Lm, bank * L, static_cast<unsigned *>(fault));
  }
  SWEM_CHECK_LAUNCH("pack_values");
  return SWEM_OK;
}

extern "C" int swem_match_f32(void *stream, const float *qk, const float *kappa_first, const float *nu_first,
                              const float *kappa_update, const float *nu_update, float *mem_out, float *S, int N,
                              int C, int V, int P, int L, int topl, float tau, int readout_plan, void *ws,
                              size_t ws_bytes) {
  SWEM_REQUIRE(qk && kappa_first && nu_first && mem_out && S, SWEM_E_ARG, "match: null pointer");
  SWEM_REQUIRE((kappa_update == nullptr) == (nu_update == nullptr), SWEM_E_ARG, "match: update bank half given");
  const int nbanks = kappa_update ? 2 : 1;
  const int Lm = nbanks * L;
  int rc;
  if ((rc = match_check(C, V, L, Lm, topl, tau))) return rc;
  MatchWs w = match_ws(N, C, V, P, L, nbanks, readout_plan);
  SWEM_REQUIRE(ws && ws_bytes >= w.total, SWEM_E_WORKSPACE, "match: workspace %zu < %zu", ws_bytes, w.total);
  char *base = static_cast<char *>(ws);
  float *mkn = (float *)(base + w.mkn), *mvp = (float *)(base + w.mvp), *pT = (float *)(base + w.pT);
  if ((rc = swem_match_pack_bank_f32(stream, kappa_first, nu_first, mkn, mvp, nullptr, 0, nbanks, N, C, V, L, nullptr))) return rc;
  if (nbanks == 2 &&
      (rc = swem_match_pack_bank_f32(stream, kappa_update, nu_update, mkn, mvp, nullptr, 1, nbanks, N, C, V, L, nullptr)))
    return rc;
  return match_core(stream, qk, mkn, mvp, nullptr, pT, nullptr, mem_out, S, N, C, V, P, Lm, topl, tau, readout_plan,
                    base + w.conv, w.total - w.conv);
}

// The same on banks the caller keeps packed (swem_match_pack_bank_f32 / swem_memorize_packed_f32): no per-frame
// normalisation and repacking of 2 x 2 banks.  Both banks: mkn [2N][C/4+1][2L][4], mvp [N][V][4L].
extern "C" size_t swem_match_packed_workspace(int N, int C, int V, int P, int L, int readout_plan) {
  MatchWs w = match_ws(N, C, V, P, L, 2, readout_plan);
  return w.total - w.pT;
}

namespace {
int match_packed_impl(void *stream, const float *qk, const float *mkn, const float *mvp, const void *mvq, float *mem_out, float *S,
                      int N, int C, int V, int P, int L, int topl, float tau, int readout_plan, void *ws, size_t ws_bytes,
                      void *mem_planes, int mem_npl, void *s_planes, int s_npl, void *fault, int clips = 1) {
  SWEM_REQUIRE(qk && mkn && mvp && mem_out && S, SWEM_E_ARG, "match_packed: null pointer");
  SWEM_REQUIRE(clips >= 1 && N % clips == 0, SWEM_E_SHAPE, "match_packed: %d objects do not divide into %d clips", N, clips);
  int rc;
  if ((rc = match_check(C, V, L, 2 * L, topl, tau))) return rc;
  SWEM_REQUIRE((!mem_planes || ((mem_npl == 2 || mem_npl == 3 || mem_npl == SWEM_PLANES_F16) && V % 8 == 0)) &&
                   (!s_planes || ((s_npl == 2 || s_npl == 3 || s_npl == SWEM_PLANES_F16) && topl % 4 == 0)),
               SWEM_E_ARG, "match_packed: output planes: 2 or 3 per tensor, V %% 8 == 0, topl %% 4 == 0");
  MatchWs w = match_ws(N, C, V, P, L, 2, readout_plan);
  SWEM_REQUIRE(ws && ws_bytes >= w.total - w.pT, SWEM_E_WORKSPACE, "match_packed: workspace %zu < %zu", ws_bytes,
               w.total - w.pT);
  char *base = static_cast<char *>(ws) - w.pT;     // the workspace starts at the probability slot
  return match_core(stream, qk, mkn, mvp, static_cast<const unsigned short *>(mvq), (float *)(base + w.pT),
                    (unsigned short *)(base + w.pq), mem_out, S, N, C, V, P, 2 * L, topl, tau, readout_plan, base + w.conv,
                    w.total - w.conv, mem_planes, mem_npl, s_planes, s_npl, fault, N / clips);
}
}  // namespace

// The same for the objects of SEVERAL clips in one call (round 6: sequences in lock step): N objects in total, N / clips per
// clip, qk = one query key map per clip [clips][P][C]; the packs and the outputs are per object as before.  Plane pointers may
// be NULL (then as swem_match_packed_f32).  Per object the same blocks on the same data as `clips` single-clip calls.
extern "C" int swem_match_packed_clips_f32(void *stream, const float *qk, const float *mkn, const float *mvp, const void *mvq,
                                           float *mem_out, float *S, int N, int clips, int C, int V, int P, int L, int topl, float tau,
                                           int readout_plan, void *ws, size_t ws_bytes, void *mem_planes, int mem_nplanes,
                                           void *s_planes, int s_nplanes, void *fault) {
  return match_packed_impl(stream, qk, mkn, mvp, mvq, mem_out, S, N, C, V, P, L, topl, tau, readout_plan, ws, ws_bytes, mem_planes,
                           mem_planes ? mem_nplanes : 3, s_planes, s_planes ? s_nplanes : 3, fault, clips);
}

extern "C" int swem_match_packed_f32(void *stream, const float *qk, const float *mkn, const float *mvp, const void *mvq,
                                     float *mem_out, float *S, int N, int C, int V, int P, int L, int topl, float tau,
                                     int readout_plan, void *ws, size_t ws_bytes) {
  return match_packed_impl(stream, qk, mkn, mvp, mvq, mem_out, S, N, C, V, P, L, topl, tau, readout_plan, ws, ws_bytes, nullptr, 3,
                           nullptr, 3, nullptr);
}

extern "C" int swem_match_packed_f32_planes(void *stream, const float *qk, const float *mkn, const float *mvp, const void *mvq,
                                            float *mem_out, float *S, int N, int C, int V, int P, int L, int topl, float tau,
                                            int readout_plan, void *ws, size_t ws_bytes, void *mem_planes, int mem_nplanes,
                                            void *s_planes, int s_nplanes, void *fault) {
  return match_packed_impl(stream, qk, mkn, mvp, mvq, mem_out, S, N, C, V, P, L, topl, tau, readout_plan, ws, ws_bytes, mem_planes,
                           mem_nplanes, s_planes, s_nplanes, fault);
}

// backward of swem_match_f32 for one clip: d mem_out [N][Pm][V] and dS [N][P][2*topl] (either may be NULL) ->
// dqk [P][C] (summed over the objects) and the value bases' gradients dnu_* [N][2][V][L]; the key bases carry none
// (the EM runs under no_grad, modules.py:93,112,122).
extern "C" size_t swem_match_bwd_workspace(int N, int C, int V, int P, int L, int nbanks) {
  return match_bwd_ws(N, C, V, P, L, nbanks).total;
}

extern "C" int swem_match_bwd_f32(void *stream, const float *qk, const float *kappa_first, const float *nu_first,
                                  const float *kappa_update, const float *nu_update, const float *dmem,
                                  const float *dS, float *dqk, float *dnu_first, float *dnu_update, int N, int C, int V,
                                  int P, int L, int topl, float tau, void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(qk && kappa_first && nu_first && dmem && dqk && dnu_first, SWEM_E_ARG, "match_bwd: null pointer");
  SWEM_REQUIRE((kappa_update == nullptr) == (nu_update == nullptr) && (nu_update == nullptr) == (dnu_update == nullptr),
               SWEM_E_ARG, "match_bwd: update bank half given");
  SWEM_REQUIRE(N >= 1 && N <= 7, SWEM_E_SHAPE, "match_bwd: 1..7 objects per clip (got %d)", N);
  const int nbanks = kappa_update ? 2 : 1;
  const int Lm = nbanks * L, Ltot = 2 * Lm, Pm = swem_match_pad(P);
  MatchBwdWs w = match_bwd_ws(N, C, V, P, L, nbanks);
  SWEM_REQUIRE(ws && ws_bytes >= w.total, SWEM_E_WORKSPACE, "match_bwd: workspace %zu < %zu", ws_bytes, w.total);
  char *base = static_cast<char *>(ws);
  // forward intermediates again: mkn, mvp, pT (the readout GEMM's output lands in the dP slot and is overwritten)
  MatchWs fw = match_ws(N, C, V, P, L, nbanks, 0);
  char *fb = base + w.fwd;
  float *mkn = (float *)(fb + fw.mkn), *mvp = (float *)(fb + fw.mvp), *pT = (float *)(fb + fw.pT);
  float *mvpT = (float *)(base + w.mvpT), *dP = (float *)(base + w.dP), *mknT = (float *)(base + w.mknT);
  float *dmvp = (float *)(base + w.dmvp), *dqn = (float *)(base + w.dqn);
  int rc;
  // packed keys twice: raw for the affinity kernel (it normalises, exactly as in the forward), normalised for the d qn GEMM
  float *mknn = (float *)(base + w.mknn);
  if ((rc = swem_norm_bases_into(stream, kappa_first, mkn, 2 * N, C, L, Lm, 0, 0))) return rc;
  if (nbanks == 2 && (rc = swem_norm_bases_into(stream, kappa_update, mkn, 2 * N, C, L, Lm, L, 0))) return rc;
  if ((rc = swem_norm_bases_into(stream, kappa_first, mknn, 2 * N, C, L, Lm, 0, 1))) return rc;
  if (nbanks == 2 && (rc = swem_norm_bases_into(stream, kappa_update, mknn, 2 * N, C, L, Lm, L, 1))) return rc;
  const long long work = (long long)N * 2 * V * (L / 4);
  hipLaunchKernelGGL(pack_values_kernel, dim3(cdiv(work, 256)), dim3(256), 0, ST, nu_first, mvp, N, V, L, Lm, 0);
  if (nbanks == 2)
    hipLaunchKernelGGL(pack_values_kernel, dim3(cdiv(work, 256)), dim3(256), 0, ST, nu_update, mvp, N, V, L, Lm, L);
  dim3 gridt(cdiv((long long)N * P, 4));
  if ((rc = launch_affinity(ST, qk, mkn, pT, nullptr, nullptr, nullptr, 3, 0, N, C, P, Pm, Lm, tau))) return rc;
  SWEM_CHECK_LAUNCH("match_bwd (forward recompute)");
  // (1) dP[n] = dmem[n] . mvp[n]   (batched GEMM on the conv kernel; filters = mvp[n]^T [Ltot][V])
  if ((rc = swem_transpose_f32(stream, mvp, mvpT, N, V, Ltot, V))) return rc;
  if ((rc = swem_conv2d_nhwc_f32(stream, dmem, V, (long long)Pm * V, nullptr, 0, 0, nullptr, 0, 0, N, Pm, 1, mvpT,
                                 (long long)Ltot * V, nullptr, nullptr, nullptr, 0, dP, Ltot, 1, 1, 1, 0, 0, 0,
                                 base + w.conv, w.wgrad - w.conv)))
    return rc;
  // (2) dmvp[n] = dmem[n]^T . pT[n]  (the weight-gradient GEMM), unpacked into the banks' layout
  for (int n = 0; n < N; ++n)
    if ((rc = swem_conv2d_wgrad_f32(stream, dmem + (long long)n * Pm * V, pT + (long long)n * Pm * Ltot, Ltot, 0, nullptr,
                                    0, 0, nullptr, 0, 0, 1, Pm, 1, V, 1, 1, 1, 0, 0, dmvp + (long long)n * V * Ltot,
                                    Ltot, 0, base + w.wgrad, w.total - w.wgrad)))
      return rc;
  hipLaunchKernelGGL(unpack_values_kernel, dim3(cdiv(work, 256)), dim3(256), 0, ST, dmvp, dnu_first, N, V, L, Lm, 0);
  if (nbanks == 2)
    hipLaunchKernelGGL(unpack_values_kernel, dim3(cdiv(work, 256)), dim3(256), 0, ST, dmvp, dnu_update, N, V, L, Lm, L);
  SWEM_CHECK_LAUNCH("unpack_values");
  // (3) per pixel: top-l feature gradient + softmax Jacobian -> da (in place of dP)
#define PIX(J_)                                                                                                 \
  hipLaunchKernelGGL((match_bwd_pixel_kernel<J_>), gridt, dim3(256), 0, ST, pT, dP, dS, N, P, Pm, topl, 1.0f / tau)
  if (Lm == 64) PIX(1);
  else if (Lm == 128) PIX(2);
  else if (Lm == 256) PIX(4);
  else PIX(8);
#undef PIX
  SWEM_CHECK_LAUNCH("match_bwd_pixel");
  // (4) d qn = sum_n da[n] . mkn[n]: one GEMM with the objects as concatenated sources (filters [C][N*Ltot])
  hipLaunchKernelGGL(kn_cmajor_kernel, dim3(cdiv((long long)C * N * Ltot, 256)), dim3(256), 0, ST, mknn, mknT, N, C, Lm);
  SWEM_CHECK_LAUNCH("kn_cmajor");
  // (the conv kernel concatenates up to three sources: objects go in groups of three, later groups add onto the result)
  for (int n0 = 0; n0 < N; n0 += 3) {
    const int g = N - n0 < 3 ? N - n0 : 3;
    const float *d0 = dP + (long long)n0 * Pm * Ltot, *d1 = g > 1 ? d0 + (long long)Pm * Ltot : nullptr,
                *d2 = g > 2 ? d0 + 2ll * Pm * Ltot : nullptr;
    // filters of this group: columns [n0*Ltot, (n0+g)*Ltot) of every row of mknT -> packed [C][g*Ltot]
    float *wg = mknT;
    if (N > 3) {
      wg = mknT + (long long)C * N * Ltot;   // second half of the slot (sized for it below)
      if (hipMemcpy2DAsync(wg, (size_t)g * Ltot * 4, mknT + (long long)n0 * Ltot, (size_t)N * Ltot * 4, (size_t)g * Ltot * 4, C,
                           hipMemcpyDeviceToDevice, ST) != hipSuccess) {
        swem_set_error("match_bwd: copy failed");
        return SWEM_E_HIP;
      }
    }
    if ((rc = swem_conv2d_nhwc_f32(stream, d0, Ltot, 0, d1, g > 1 ? Ltot : 0, 0, d2, g > 2 ? Ltot : 0, 0, 1, Pm, 1, wg, 0,
                                   nullptr, nullptr, n0 ? dqn : nullptr, 0, dqn, C, 1, 1, 1, 0, 0, 0, base + w.conv,
                                   w.wgrad - w.conv)))
      return rc;
  }
  // (5) through the query's l2norm
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv(P, 4)), dim3(256), 0, ST, qk, dqn, dqk, P, C);
  SWEM_CHECK_LAUNCH("l2norm_bwd");
  return SWEM_OK;
}
