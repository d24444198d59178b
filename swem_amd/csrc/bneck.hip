// A whole ResNet bottleneck block (mod_resnet.py:77-113; torchvision v1.5 Bottleneck, the key encoder's layer1) in ONE launch:
//
//     y = relu( bn3(conv1x1_{Cm->C}( relu(bn2(conv3x3_{Cm->Cm}( relu(bn1(conv1x1_{C->Cm}(x))) ))) )) + x ),      C = 4 Cm
//
// for the identity blocks of a stage (stride 1, no down-sampling branch), in the f16x3 arithmetic of the pre-split convolution
// kernel (conv.hip): every operand an fp16 (hi, mid) pair, three products hi.mid + mid.hi + hi.hi on v_mfma_f32_16x16x32_f16,
// fp32 accumulation, frozen BatchNorm as per-filter scale / shift.  As three launches the block moves its input, its output
// and two Cm-channel intermediates through memory four times over (1,059 MB for ten 480p frames) and runs at 0.3x of a plain
// copy of its input planes to its output planes (profiles/r05_bottleneck_fusion_bound.json: 337 us against 101 us); here the
// Cm-channel intermediates never leave the LDS.
//
// Block = one 8 x 16 tile of output pixels of one image, 8 waves, one block per CU.  Three GEMM phases:
//   1  y1 = relu(bn1(x . w1)) on the tile's 10 x 18 HALO patch (180 pixels, 192 LDS rows), K = C in 32-channel k-blocks: the
//      x planes and w1 go L2 -> LDS by buffer_load ... lds through a two-stage ring exactly as in conv.hip (patch pixels outside
//      the image use an out-of-range offset and land as zeros); y1 is split into its fp16 pair and kept in LDS, ZERO where the
//      patch pixel lies outside the image (conv2's zero padding pads y1, not x);
//   2  y2 = relu(bn2(conv3x3(y1))): K = 9 Cm, k-block = (32-channel block, tap); the A fragments are read from the y1 patch at
//      the tap's row offset, w2 streams through a four-stage ring; y2 (128 pixels x Cm) stays in LDS as an fp16 pair;
//   3  y = relu(bn3(y2 . w3) + x): w3 (64 KB as a pair) is loaded whole -- its second plane already during phase 2 --, K = Cm;
//      the epilogue stages 32 x 32 sub-tiles per wave through the LDS and leaves in ROW layout (conv.hip, out_tile32): 16-byte
//      stores of y (optional) and of y's own fp16 pair planes, the residual read back from x's planes (hi + mid: the value to
//      22-23 significant bits, exactly what the unfused path adds: swem_conv2d_nhwc_bf16x3_planes_res).
// The k order of every GEMM and the order of the three products are the conv kernel's (M16 form), so the block agrees with the
// three-launch path to fp32 rounding of identical operations.
#include "common.h"
#include "lds_dma.h"
#include "bf16_split.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr unsigned BN_OOB = 0xfffffff0u;   // >= any num_records: the transfer lands as zeros and touches no memory

struct BneckP {
  const unsigned short *x;    // fp16 pair planes of the input, [2][C/8][npix][8]
  long long x_ps;             // elements between its planes
  int npix;                   // pixels per plane of x (x_ps / C: the planes may belong to a larger batch)
  int ynpix;                  // ... of the output planes
  const unsigned short *w1, *w2, *w3;   // filter planes (fp16 pairs, per-column power-of-two scaling folded into s*)
  const float *s1, *b1, *s2, *b2, *s3, *b3;
  float *y;                   // optional fp32 NHWC output
  unsigned short *yp;         // optional fp16 pair planes of the output, [2][C/8][npix][8]
  long long y_ps;
  int B, H, W, tiles_x, tiles_y;
  unsigned *fault;
};

__device__ __forceinline__ f32x4v mm(uint4 a, uint4 b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float relu_(float x) {   // one v_max_f32 (conv.hip, relu1)
  float y;
  asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
  return y;
}
__device__ __forceinline__ void wait_vm(int n) {   // s_waitcnt vmcnt(n): the immediates the rings need
  switch (n) {
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // (0, or anything unforeseen: wait for everything)
  }
}
// fp16 (hi, mid) halves of one value as raw 16-bit patterns
__device__ __forceinline__ void pair16(float v, unsigned short &h, unsigned short &m) {
  const _Float16 hh = (_Float16)v;
  const _Float16 mm_ = (_Float16)(v - (float)hh);
  h = __builtin_bit_cast(unsigned short, hh);
  m = __builtin_bit_cast(unsigned short, mm_);
}

constexpr int TH = 8, TW = 16, PW = TW + 2, PP = (TH + 2) * (TW + 2), PR = 192;   // tile, patch width / pixels / LDS rows
constexpr int PL_STRIDE_B = 36;                    // dwords per staged epilogue row (conv.hip)
constexpr int PL_BYTES_B = 32 * PL_STRIDE_B * 4;   // epilogue staging bytes per wave

// LDS map (Cm = 64: exactly the CU's 160 KB).  One block per CU, so nothing but the block's own ring depth hides the L2 -> LDS
// latency (conv.hip leans on its second resident block for that): every phase uses ALL the LDS that is dead at the time.
//   [0, 64K)      R   phase 1: ring stages 0-1        phase 2-3: w3, whole (requested at the start of phase 2)
//   [64K, 112K)   Y1  phase 1: ring stages 2-3 (part) phase 2: the y1 patch               phase 3: epilogue staging
//   [112K, 144K)  Y2  phase 1: ring stage 3 (part)    phase 2: w2 ring stages 0-3         phase 3: y2
//   [144K, 160K)  D                                   phase 2: w2 ring stages 4-5
template <int CM>
struct BneckLds {
  static constexpr int C = 4 * CM;
  static constexpr int A1 = 2 * 4 * PR * 16;          // phase-1 x stage
  static constexpr int B1 = 2 * 4 * CM * 16;          // phase-1 w1 stage
  static constexpr int ST1 = A1 + B1;
  static constexpr int NS1 = 4;                       // phase-1 ring: three k-blocks in flight
  static constexpr int ST2 = 2 * 4 * CM * 16;         // phase-2 w2 stage
  static constexpr int NS2 = 6;                       // ... five k-blocks in flight
  static constexpr int W3 = 2 * (CM / 8) * C * 16;    // w3, whole
  static constexpr int R = W3;
  static constexpr int Y1 = R;                        // y1 patch [pl][CM/8][PR] x 16 B
  static constexpr int Y1B = 2 * (CM / 8) * PR * 16;
  static constexpr int Y2 = Y1 + Y1B;                 // y2 [pl][CM/8][128] x 16 B
  static constexpr int Y2B = 2 * (CM / 8) * (TH * TW) * 16;
  static constexpr int RING2 = Y2;                    // the w2 ring lives where y2 will be written, plus the spare behind it
  static constexpr int TOTAL = RING2 + NS2 * ST2;
  static_assert(NS1 * ST1 <= TOTAL, "the phase-1 ring spans the regions that are dead during phase 1");
  static_assert(NS2 * ST2 >= Y2B && TOTAL <= 160 * 1024, "LDS budget");
  static_assert(8 * PL_BYTES_B <= Y1B, "the epilogue staging reuses the y1 region");
};

template <int CM>
__global__ __launch_bounds__(512, 1) void bneck_kernel(BneckP p STAMP_ARG) {
  STAMP(0);
  static_assert(CM == 64, "layer1 geometry: Cm = 64, C = 256");
  using L = BneckLds<CM>;
  constexpr int C = 4 * CM;
  constexpr int NK1 = C / 32;             // k-blocks of phase 1
  constexpr int NK2 = 9 * CM / 32;        // ... of phase 2 (18)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kgl = lane >> 4;
  int t = blockIdx.x;
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y;
  const int b = t / p.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int img0 = b * p.H * p.W;   // (pixel indices fit 28 bits: npix * 16 bytes is a 32-bit buffer offset)

  const i32x4 rsx = raw_rsrc(p.x, (unsigned)(2 * p.x_ps * 2));
  const unsigned xplane = (unsigned)(p.x_ps * 2), xgroup = (unsigned)p.npix * 16u;
  const int dpl = wave >> 2, dkg = wave & 3;   // the (plane, k/8 group) this wave transfers in phases 1 and 2

  // ------------------------------------------------------------------ phase 1: y1 = relu(bn1(x . w1)) on the halo patch
  unsigned avoff[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int pp = j * 64 + lane;
    const int py = (pp * 57) >> 10, px = pp - py * PW;      // pp / 18 for pp < 192
    const int gy = y0 - 1 + py, gx = x0 - 1 + px;
    const bool in = pp < PP && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    avoff[j] = in ? (unsigned)(img0 + gy * p.W + gx) * 16u : BN_OOB;
  }
  const unsigned bvoff = (unsigned)lane * 16u;   // filter column = lane (CM = 64 columns)
  const i32x4 rsw1 = raw_rsrc(p.w1, (unsigned)(2 * C * CM * 2));
  constexpr unsigned w1plane = C * CM * 2, wgroup = CM * 16;
  auto issue1 = [&](int kb, int stage) __attribute__((always_inline)) {
    const unsigned sa = lds0 + (unsigned)stage * L::ST1 + (unsigned)(dpl * 4 + dkg) * (PR * 16);
#pragma unroll
    for (int j = 0; j < 3; ++j) dma16(rsx, sa + j * 1024, avoff[j], (unsigned)(kb * 4 + dkg) * xgroup + dpl * xplane);
    const unsigned sb = lds0 + (unsigned)stage * L::ST1 + L::A1 + (unsigned)(dpl * 4 + dkg) * (CM * 16);
    dma16(rsw1, sb, bvoff, (unsigned)(kb * 4 + dkg) * wgroup + dpl * w1plane);
  };
  const int wm1 = wave >> 1, wn1 = wave & 1;   // 4 x 2 wave grid: 3 row tiles x 2 column tiles each
  f32x4v acc1[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  // ring of NS1 stages, NS1 - 1 k-blocks in flight (4 transfers per wave and k-block): conv.hip's counted-wait hand-over
#pragma unroll
  for (int d = 0; d < L::NS1 - 1; ++d) issue1(d, d);
  wait_vm(4 * (L::NS1 - 2));        // k-block 0 has landed; the younger ones may stay in flight
  __builtin_amdgcn_s_barrier();
  STAMP(1);
  {
    int st = 0;
    for (int kb = 0; kb < NK1; ++kb) {
      if (kb + L::NS1 - 1 < NK1) issue1(kb + L::NS1 - 1, st == 0 ? L::NS1 - 1 : st - 1);
      const char *As = smem + st * L::ST1, *Bs = As + L::A1;
      uint4 a[2][3], bb[2][2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
          a[pl][i] = *reinterpret_cast<const uint4 *>(As + ((pl * 4 + kgl) * PR + (3 * wm1 + i) * 16 + r16) * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bb[pl][j] = *reinterpret_cast<const uint4 *>(Bs + ((pl * 4 + kgl) * CM + (2 * wn1 + j) * 16 + r16) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int last = kb + L::NS1 - 1 < NK1 ? kb + L::NS1 - 1 : NK1 - 1;   // youngest k-block requested so far
      wait_vm(4 * (last - (kb + 1)));                       // k-block kb + 1 has landed (this wave's share)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's fragment reads of stage st are done
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4v c = acc1[i][j];
          c = mm(a[0][i], bb[1][j], c);
          c = mm(a[1][i], bb[0][j], c);
          c = mm(a[0][i], bb[0][j], c);
          acc1[i][j] = c;
        }
      st = st == L::NS1 - 1 ? 0 : st + 1;
    }
  }
  STAMP(2);
  // w3 (both planes) goes into R during phase 2: wave w moves k/8 group w of each plane, all C columns (4 runs of 64)
  const i32x4 rsw3 = raw_rsrc(p.w3, (unsigned)(2 * CM * C * 2));
  constexpr unsigned w3plane = CM * C * 2, w3group = C * 16;
  auto issue3 = [&](int pl) __attribute__((always_inline)) {   // wave w: k/8 group w of plane pl, all C columns (4 runs of 64)
#pragma unroll
    for (int j = 0; j < C / 64; ++j)
      dma16(rsw3, lds0 + (unsigned)((pl * (CM / 8) + wave) * C + j * 64) * 16u, (unsigned)(j * 64 + lane) * 16u,
            (unsigned)wave * w3group + pl * w3plane);
  };
  unsigned bad = 0;   // a value that does not fit the fp16 pair it is written as (SWEM_FAULT_RANGE)
  // (the y1 region overlaps ring stages 2-3: the loop's last hand-over -- every wave's fragment reads done -- lies behind us)
  {
    // y1 -> LDS as the fp16 pair, [pl][channel / 8][patch row][channel % 8]; zero outside the image
    char *Y1 = smem + L::Y1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = (2 * wn1 + j) * 16 + r16;
      const float sc = p.s1[n], sh = p.b1[n];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int pp = (3 * wm1 + i) * 16 + 4 * kgl + e;
          const int py = (pp * 57) >> 10, px = pp - py * PW;
          const int gy = y0 - 1 + py, gx = x0 - 1 + px;
          const bool in = pp < PP && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
          v[e] = in ? relu_(acc1[i][j][e] * sc + sh) : 0.f;
          unsigned short h, m;
          pair16(v[e], h, m);
          char *dst = Y1 + (((n >> 3)) * PR + pp) * 16 + (n & 7) * 2;
          *reinterpret_cast<unsigned short *>(dst) = h;
          *reinterpret_cast<unsigned short *>(dst + (CM / 8) * PR * 16) = m;
        }
        bad |= f16_oor(make_float4(v[0], v[1], v[2], v[3]));
      }
    }
  }
  // ------------------------------------------------------------------ phase 2: y2 = relu(bn2(conv3x3(y1)))
  const i32x4 rsw2 = raw_rsrc(p.w2, (unsigned)(2 * 9 * CM * CM * 2));
  constexpr unsigned w2plane = 9 * CM * CM * 2;
  auto issue2 = [&](int kb, int stage) __attribute__((always_inline)) {
    dma16(rsw2, lds0 + L::RING2 + (unsigned)stage * L::ST2 + (unsigned)(dpl * 4 + dkg) * (CM * 16), bvoff,
          (unsigned)(kb * 4 + dkg) * wgroup + dpl * w2plane);
  };
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();     // every wave has written its part of y1
  issue3(0);
  issue3(1);
#pragma unroll
  for (int d = 0; d < L::NS2 - 1; ++d) issue2(d, d);
  wait_vm(L::NS2 - 2);              // k-block 0 has landed (and all of w3, requested before it); the younger ones may stay in flight
  __builtin_amdgcn_s_barrier();
  STAMP(3);
  const int wm2 = wave >> 1, wn2 = wave & 1;   // 4 x 2 wave grid: 2 output rows x 2 column tiles each
  f32x4v acc2[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc2[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  {
    int st = 0, cb = 0, ky = 0, kx = 0;
    for (int kb = 0; kb < NK2; ++kb) {
      if (kb + L::NS2 - 1 < NK2) issue2(kb + L::NS2 - 1, st == 0 ? L::NS2 - 1 : st - 1);
      const char *Ay = smem + L::Y1, *Bs = smem + L::RING2 + st * L::ST2;
      uint4 a[2][2], bb[2][2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[pl][i] = *reinterpret_cast<const uint4 *>(
              Ay + ((pl * (CM / 8) + cb * 4 + kgl) * PR + (2 * wm2 + i + ky) * PW + r16 + kx) * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bb[pl][j] = *reinterpret_cast<const uint4 *>(Bs + ((pl * 4 + kgl) * CM + (2 * wn2 + j) * 16 + r16) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      // k-block kb + 1 must have landed; the younger ones (up to kb + NS2 - 1) may stay in flight
      const int last = kb + L::NS2 - 1 < NK2 ? kb + L::NS2 - 1 : NK2 - 1;
      wait_vm(last - (kb + 1));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4v c = acc2[i][j];
          c = mm(a[0][i], bb[1][j], c);
          c = mm(a[1][i], bb[0][j], c);
          c = mm(a[0][i], bb[0][j], c);
          acc2[i][j] = c;
        }
      st = st == L::NS2 - 1 ? 0 : st + 1;
      if (++kx == 3) {
        kx = 0;
        if (++ky == 3) {
          ky = 0;
          ++cb;
        }
      }
    }
  }
  STAMP(4);
  // (y2 is written over the w2 ring: the loop's last hand-over -- every wave's fragment reads done -- lies behind us)
  {
    char *Y2 = smem + L::Y2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = (2 * wn2 + j) * 16 + r16;
      const float sc = p.s2[n], sh = p.b2[n];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = (2 * wm2 + i) * 16 + 4 * kgl + e;
          v[e] = relu_(acc2[i][j][e] * sc + sh);
          unsigned short h, mm_;
          pair16(v[e], h, mm_);
          char *dst = Y2 + ((n >> 3) * (TH * TW) + m) * 16 + (n & 7) * 2;
          *reinterpret_cast<unsigned short *>(dst) = h;
          *reinterpret_cast<unsigned short *>(dst + (CM / 8) * (TH * TW) * 16) = mm_;
        }
        bad |= f16_oor(make_float4(v[0], v[1], v[2], v[3]));
      }
    }
  }
  // ------------------------------------------------------------------ phase 3: y = relu(bn3(y2 . w3) + x)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();     // w3 and y2 are complete
  const int wm3 = wave >> 2, wn3 = wave & 3;   // 2 x 4 wave grid: 4 row tiles x 4 column tiles each
  // The epilogue works on four 32 x 32 sub-tiles per wave, s = 0..3 -> (column pair j0 = 2 (s / 2), row pair i0 = 2 (s % 2)); lane
  // (row id / 8, channel group id % 8) of pass c handles four channels of one pixel.  The identity (x's fp16 pair at the output
  // pixel) is fetched one sub-tile AHEAD -- the first one before the MFMAs below -- so that its L2 latency is never exposed.
  auto geom = [&](int s_, int c, long long &pix, int &n, bool &in) __attribute__((always_inline)) {
    const int i0 = 2 * (s_ & 1), j0 = 2 * (s_ >> 1);
    const int id = lane + 64 * c, row = id >> 3, cg = id & 7;
    const int m = (4 * wm3 + i0) * 16 + row;           // pixel of the tile: output row m / 16, column m % 16
    const int gy = y0 + (m >> 4), gx = x0 + (m & 15);
    in = gy < p.H && gx < p.W;
    pix = (long long)img0 + (long long)gy * p.W + gx;
    n = (4 * wn3 + j0) * 16 + 4 * cg;
  };
  uint2 rh[2][4], rm[2][4];
  auto res_fetch = [&](int buf, int s_) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      long long pix;
      int n;
      bool in;
      geom(s_, c, pix, n, in);
      rh[buf][c] = rm[buf][c] = make_uint2(0u, 0u);
      if (in) {
        const unsigned short *q = p.x + ((long long)(n >> 3) * p.npix + pix) * 8 + (n & 7);
        rh[buf][c] = *reinterpret_cast<const uint2 *>(q);
        rm[buf][c] = *reinterpret_cast<const uint2 *>(q + p.x_ps);
      }
    }
  };
  res_fetch(0, 0);
  f32x4v acc3[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc3[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cb = 0; cb < CM / 32; ++cb) {
    const char *Ay = smem + L::Y2, *Bw = smem;
    uint4 a[2][4], bb[2][4];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        a[pl][i] = *reinterpret_cast<const uint4 *>(Ay + ((pl * (CM / 8) + cb * 4 + kgl) * (TH * TW) + (4 * wm3 + i) * 16 + r16) * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bb[pl][j] = *reinterpret_cast<const uint4 *>(Bw + ((pl * (CM / 8) + cb * 4 + kgl) * C + (4 * wn3 + j) * 16 + r16) * 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4v c = acc3[i][j];
        c = mm(a[0][i], bb[1][j], c);
        c = mm(a[1][i], bb[0][j], c);
        c = mm(a[0][i], bb[0][j], c);
        acc3[i][j] = c;
      }
  }
  STAMP(5);
  // epilogue: 32 x 32 sub-tiles through this wave's slice of the (idle) y1 region, out in row layout
  unsigned *stg = reinterpret_cast<unsigned *>(smem + L::Y1 + wave * PL_BYTES_B);
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) {
    const int i0 = 2 * (s_ & 1), j0 = 2 * (s_ >> 1);
    if (s_ + 1 < 4) res_fetch((s_ + 1) & 1, s_ + 1);
    float sc[2], sh[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int n = (4 * wn3 + j0 + jj) * 16 + r16;
      sc[jj] = p.s3[n];
      sh[jj] = p.b3[n];
    }
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          stg[(16 * ii + 4 * kgl + e) * PL_STRIDE_B + 16 * jj + r16] = __float_as_uint(acc3[i0 + ii][j0 + jj][e] * sc[jj] + sh[jj]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      long long pix;
      int n;
      bool in;
      geom(s_, c, pix, n, in);
      const int id = lane + 64 * c, row = id >> 3, cg = id & 7;
      const uint4 raw = *reinterpret_cast<const uint4 *>(stg + row * PL_STRIDE_B + 4 * cg);
      float4 v = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w));
      {   // the identity: hi + mid of the input's fp16 pair (zeros where the pixel hangs over the image: never stored)
        const uint2 h = rh[s_ & 1][c], mi = rm[s_ & 1][c];
        v.x += lo_f16(h.x) + lo_f16(mi.x);
        v.y += hi_f16(h.x) + hi_f16(mi.x);
        v.z += lo_f16(h.y) + lo_f16(mi.y);
        v.w += hi_f16(h.y) + hi_f16(mi.y);
      }
      const unsigned nan_in = f32_nan(v);   // (before the ReLU turns a NaN into a clean 0: ADVICE r05)
      v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      if (in && p.y) *reinterpret_cast<float4 *>(p.y + pix * C + n) = v;
      if (p.yp) {
        uint2 h, mi;
        split2h(v, h, mi);
        if (in) bad |= f16_oor(v) | nan_in;
        const uint2 h2 = make_uint2(__shfl_down(h.x, 1), __shfl_down(h.y, 1));
        const uint2 m2 = make_uint2(__shfl_down(mi.x, 1), __shfl_down(mi.y, 1));
        if (in && (cg & 1) == 0) {
          unsigned short *d = p.yp + ((long long)(n >> 3) * p.ynpix + pix) * 8;
          *reinterpret_cast<uint4 *>(d) = make_uint4(h.x, h.y, h2.x, h2.y);
          *reinterpret_cast<uint4 *>(d + p.y_ps) = make_uint4(mi.x, mi.y, m2.x, m2.y);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next sub-tile is staged
  }
  range_fault(p.fault, bad);
  STAMP(6);
#ifdef SWEM_EM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(7);
#endif
}

}  // namespace

// include/swem_hip.h
extern "C" int swem_bottleneck_f16x3(void *stream, const void *x_planes, long long x_ps, int B, int H, int W, int C,
                                     const void *w1, const float *s1, const float *b1, const void *w2, const float *s2,
                                     const float *b2, const void *w3, const float *s3, const float *b3, float *y,
                                     void *y_planes, long long y_ps, void *fault) {
  SWEM_REQUIRE(x_planes && w1 && w2 && w3 && s1 && b1 && s2 && b2 && s3 && b3 && (y || y_planes), SWEM_E_ARG,
               "bottleneck_f16x3: null pointer (scale and shift of every convolution are required; y or y_planes)");
  SWEM_REQUIRE(C == 256, SWEM_E_SHAPE, "bottleneck_f16x3: C must be 256 (Cm = 64: the layer1 geometry), got %d", C);
  SWEM_REQUIRE(B > 0 && H > 0 && W > 0, SWEM_E_SHAPE, "bottleneck_f16x3: bad geometry");
  const long long npix = (long long)B * H * W;
  SWEM_REQUIRE(x_ps >= npix * C && x_ps % C == 0 && (!y_planes || (y_ps >= npix * C && y_ps % C == 0)) && x_ps * 4 < (1ll << 32),
               SWEM_E_SHAPE, "bottleneck_f16x3: plane strides must be multiples of C that hold B*H*W pixels, within the 4 GiB "
               "buffer-descriptor range");
  BneckP p;
  p.x = static_cast<const unsigned short *>(x_planes);
  p.x_ps = x_ps;
  p.npix = (int)(x_ps / C);
  p.ynpix = y_planes ? (int)(y_ps / C) : 0;
  p.w1 = static_cast<const unsigned short *>(w1);
  p.w2 = static_cast<const unsigned short *>(w2);
  p.w3 = static_cast<const unsigned short *>(w3);
  p.s1 = s1, p.b1 = b1, p.s2 = s2, p.b2 = b2, p.s3 = s3, p.b3 = b3;
  p.y = y;
  p.yp = static_cast<unsigned short *>(y_planes);
  p.y_ps = y_ps;
  p.B = B, p.H = H, p.W = W;
  p.tiles_x = cdiv(W, TW), p.tiles_y = cdiv(H, TH);
  p.fault = static_cast<unsigned *>(fault);
  constexpr size_t lds = BneckLds<64>::TOTAL;
  SWEM_ALLOW_LDS((bneck_kernel<64>), lds);
  hipLaunchKernelGGL((bneck_kernel<64>), dim3((unsigned)(B * p.tiles_x * p.tiles_y)), dim3(512), lds,
                     static_cast<hipStream_t>(stream), p STAMP_PASS);
  SWEM_CHECK_LAUNCH("bneck_kernel");
  return SWEM_OK;
}
