// Memory-bound kernels of the SWEM frame pipeline: input packing, pooling, resampling, CBAM gates,
// the single-channel prediction head and the mask aggregation head.  All activations are NHWC fp32
// with C % 4 == 0 so every lane moves 16 bytes; index arithmetic mirrors ATen's so index maps match.
#include "../../include/swem_hip_train.h"
#include "common.h"
#include "bf16_split.h"

namespace {

// ATen area_pixel_compute_source_index (align_corners=False, not cubic) + guard_index_and_lambda.
struct Lerp {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lerp lerp_coord(int dst, float scale, int in) {
  float src = scale * (dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  int i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  float l1 = src - (float)i0;
  l1 = fminf(fmaxf(l1, 0.f), 1.f);
  Lerp r;
  r.i0 = i0;
  r.i1 = i0 + (i0 < in - 1 ? 1 : 0);
  r.l1 = l1;
  r.l0 = 1.f - l1;
  return r;
}
// ATen nearest_neighbor_compute_source_index (legacy 'nearest')
__device__ __forceinline__ int nearest_coord(int dst, float scale, int in) {
  int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float4 f4max(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4lerp2(float l0, float4 a, float l1, float4 b) {
  // (the contraction spelled out: left to the compiler, two kernels may fuse different halves of l0 a + l1 b and differ in
  // the last bit -- the planes-writing variants must reproduce the plain kernels exactly)
  return make_float4(fmaf(l1, b.x, __fmul_rn(l0, a.x)), fmaf(l1, b.y, __fmul_rn(l0, a.y)), fmaf(l1, b.z, __fmul_rn(l0, a.z)),
                     fmaf(l1, b.w, __fmul_rn(l0, a.w)));
}

__global__ void prep_key_input_kernel(const float *__restrict__ f, float3 mean, float3 stdv, float *__restrict__ out,
                                      int B, long long HW) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * HW) return;
  long long b = i / HW, p = i - b * HW;
  const float *src = f + b * 3 * HW + p;
  st4(out + i * 4, make_float4((src[0] - mean.x) / stdv.x, (src[HW] - mean.y) / stdv.y,
                               (src[2 * HW] - mean.z) / stdv.z, 0.f));
}

__global__ void prep_value_input_kernel(const float *__restrict__ f, const float *__restrict__ masks, float3 mean,
                                        float3 stdv, float *__restrict__ out, int B, int N, long long HW,
                                        int single_obj) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * N * HW) return;
  long long bn = i / HW, p = i - bn * HW;
  int b = (int)(bn / N), n = (int)(bn - (long long)b * N);
  const float *src = f + (long long)b * 3 * HW + p;
  const float *mb = masks + (long long)b * (N + 1) * HW + p;
  float m = mb[(long long)(n + 1) * HW];
  float other = single_obj ? 0.f : (1.f - m - mb[0]);
  st4(out + i * 8, make_float4((src[0] - mean.x) / stdv.x, (src[HW] - mean.y) / stdv.y,
                               (src[2 * HW] - mean.z) / stdv.z, m));
  st4(out + i * 8 + 4, make_float4(other, 0.f, 0.f, 0.f));
}

// ---- space-to-depth stems --------------------------------------------------------------------------------------
// conv1 of both encoders is 7x7 / stride 2 / pad 3 on 3 (key) or 5 (value) input channels: K = 147 / 245, far too thin for
// the matrix cores (the generic fp32 kernel runs it at 45-50 TFLOP/s: 5 % of a frame).  Folding the stride into the
// channels -- a 2x2 pixel block becomes one "pixel" of 4 x 8 channels -- turns it into a 4x4 / stride 1 convolution on 32
// channels (K = 512), which the pre-split bf16 kernel takes: out(o) reads input rows 2o-3 .. 2o+3 = blocks o-2 .. o+1, tap
// dy = (ky+1) >> 1 of phase py = (ky+1) & 1 (the (dy, py) = (0, 0) tap has no filter row: zero).  The blocks are stored
// shifted by one, behind a zero row and column, so that the asymmetric padding (2 before, 1 after) becomes the symmetric
// pad = 1 the kernels know.  These kernels write that image (fp32, for the fp32 conv kernels) AND its bf16 planes: one
// thread per ORIGINAL pixel = one 8-channel group of its block, 16 contiguous bytes per plane.
// value input (swem.py:48-53, networks.py:115-117): channels (r, g, b, mask, other objects, 0, 0, 0); key input
// (networks.py:161): (r, g, b, 0, ...); masks == NULL selects the key form.
__global__ void prep_input_s2d_kernel(const float *__restrict__ f, const float *__restrict__ masks, float3 mean, float3 stdv,
                                      float *__restrict__ out, unsigned short *__restrict__ planes, int nplanes, int B,
                                      int N, int H, int W, int single_obj, unsigned *fault) {
  const int Hb = H / 2 + 1, Wb = W / 2 + 1;
  const long long npix = (long long)B * N * Hb * Wb;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (image, block row, block col, phase)
  if (i >= npix * 4) return;
  const int ph = (int)(i & 3);
  const long long blk = i >> 2;
  const int bx = (int)(blk % Wb);
  long long t = blk / Wb;
  const int by = (int)(t % Hb);
  const long long bn = t / Hb;
  float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
  if (by > 0 && bx > 0) {
    const int iy = 2 * (by - 1) + (ph >> 1), ix = 2 * (bx - 1) + (ph & 1);
    const long long HW = (long long)H * W, pp = (long long)iy * W + ix;
    const int b = (int)(bn / N), n = (int)(bn - (long long)b * N);
    const float *src = f + (long long)b * 3 * HW + pp;
    v0 = make_float4((src[0] - mean.x) / stdv.x, (src[HW] - mean.y) / stdv.y, (src[2 * HW] - mean.z) / stdv.z, 0.f);
    if (masks) {
      const float *mb = masks + (long long)b * (N + 1) * HW + pp;
      const float m = mb[(long long)(n + 1) * HW];
      v0.w = m;
      v1.x = single_obj ? 0.f : (1.f - m - mb[0]);
    }
  }
  if (out) {
    st4(out + blk * 32 + ph * 8, v0);
    st4(out + blk * 32 + ph * 8 + 4, v1);
  }
  if (planes) {
    uint2 h0, m0, l0, h1, m1, l1;
    unsigned bad = 0;
    split_as(nplanes, v0, h0, m0, l0, bad);
    split_as(nplanes, v1, h1, m1, l1, bad);
    range_fault(fault, bad);   // (a frame far outside [0, 1], or a NaN in it)
    const long long plane = npix * 32, o = ((long long)ph * npix + blk) * 8;
    *reinterpret_cast<uint4 *>(planes + o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
    *reinterpret_cast<uint4 *>(planes + plane + o) = make_uint4(m0.x, m0.y, m1.x, m1.y);
    if (nplanes == 3) *reinterpret_cast<uint4 *>(planes + 2 * plane + o) = make_uint4(l0.x, l0.y, l1.x, l1.y);
  }
}

__global__ void maxpool_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int H, int W, int C, int Ho,
                               int Wo) {
  const int cq = C / 4;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * Ho * Wo * cq) return;
  int c4 = (int)(i % cq);
  long long t = i / cq;
  int ox = (int)(t % Wo);
  t /= Wo;
  int oy = (int)(t % Ho);
  int b = (int)(t / Ho);
  const float ninf = -__builtin_huge_valf();
  float4 m = make_float4(ninf, ninf, ninf, ninf);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    int iy = oy * 2 - 1 + ky;
    if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      int ix = ox * 2 - 1 + kx;
      if ((unsigned)ix >= (unsigned)W) continue;
      m = f4max(m, ld4(x + (((long long)b * H + iy) * W + ix) * C + c4 * 4));
    }
  }
  st4(y + i * 4, m);
}

// bf16 planes (hi, mid[, lo]) of 8 consecutive channels of one pixel, with and / or without a ReLU, in the operand-split
// layout [plane][C/8][npix][8] (split_bf16x3_kernel, conv.hip): i = channel group * npix + pixel, plane = npix * C.
__device__ __forceinline__ void planes8_out(const float4 (&v)[2], unsigned short *pl0, int npl0, unsigned short *pl1, int npl1,
                                            long long plane, long long i, unsigned *fault) {
  unsigned bad = 0;   // a value that does not fit the fp16 pair it is written as (SWEM_FAULT_RANGE, include/swem_hip.h)
#pragma unroll
  for (int var = 0; var < 2; ++var) {
    unsigned short *out = var ? pl1 : pl0;
    const int npl = var ? npl1 : npl0;
    if (!out) continue;
    uint2 h0, m0, l0, h1, m1, l1;
    split_as(npl, var ? make_float4(fmaxf(v[0].x, 0.f), fmaxf(v[0].y, 0.f), fmaxf(v[0].z, 0.f), fmaxf(v[0].w, 0.f)) : v[0], h0, m0, l0, bad);
    split_as(npl, var ? make_float4(fmaxf(v[1].x, 0.f), fmaxf(v[1].y, 0.f), fmaxf(v[1].z, 0.f), fmaxf(v[1].w, 0.f)) : v[1], h1, m1, l1, bad);
    // (the ReLU variant: fmaxf(NaN, 0) = 0 would put a clean 0 into the fp16 pair -- the NaN is reported from the value before it,
    // as split_f16x2_kernel does)
    if (var && npl == SWEM_PLANES_F16) bad |= f32_nan(v[0]) | f32_nan(v[1]);
    *reinterpret_cast<uint4 *>(out + i * 8) = make_uint4(h0.x, h0.y, h1.x, h1.y);
    *reinterpret_cast<uint4 *>(out + plane + i * 8) = make_uint4(m0.x, m0.y, m1.x, m1.y);
    if (npl == 3) *reinterpret_cast<uint4 *>(out + 2 * plane + i * 8) = make_uint4(l0.x, l0.y, l1.x, l1.y);
  }
  range_fault(fault, bad);
}

// maxpool_kernel plus the result's bf16 planes for the convolution that consumes it (the first block of layer1 reads the
// pooled stem twice: conv1 and the downsample branch).  Block = 32 pixels x 8 channel groups, thread = (pixel, 8 channels).
__global__ __launch_bounds__(256) void maxpool_planes_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                             unsigned short *__restrict__ pl0, int npl0,
                                                             unsigned short *__restrict__ pl1, int npl1, int B, int H, int W,
                                                             int C, int Ho, int Wo, unsigned *fault) {
  const long long npix = (long long)B * Ho * Wo;
  const long long pix = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int cg = blockIdx.y * 8 + (threadIdx.x & 7);
  if (pix >= npix || cg >= C / 8) return;
  const int ox = (int)(pix % Wo);
  long long t = pix / Wo;
  const int oy = (int)(t % Ho), b = (int)(t / Ho);
  const float ninf = -__builtin_huge_valf();
  float4 v[2] = {make_float4(ninf, ninf, ninf, ninf), make_float4(ninf, ninf, ninf, ninf)};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 - 1 + ky;
    if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 - 1 + kx;
      if ((unsigned)ix >= (unsigned)W) continue;
      const float *src = x + (((long long)b * H + iy) * W + ix) * C + cg * 8;
      v[0] = f4max(v[0], ld4(src));
      v[1] = f4max(v[1], ld4(src + 4));
    }
  }
  st4(y + pix * C + cg * 8, v[0]);
  st4(y + pix * C + cg * 8 + 4, v[1]);
  planes8_out(v, pl0, npl0, pl1, npl1, npix * C, (long long)cg * npix + pix, fault);
}

__global__ void upsample_add_kernel(const float *__restrict__ skip, long long skip_bs, const float *__restrict__ low,
                                    float *__restrict__ y, int B, int Hl, int Wl, int Ho, int Wo, int C, int group) {
  const int cq = C / 4;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * Ho * Wo * cq) return;
  int c4 = (int)(i % cq);
  long long t = i / cq;
  int ox = (int)(t % Wo);
  t /= Wo;
  int oy = (int)(t % Ho);
  int b = (int)(t / Ho);
  Lerp ly = lerp_coord(oy, (float)Hl / (float)Ho, Hl), lx = lerp_coord(ox, (float)Wl / (float)Wo, Wl);
  const float *base = low + (long long)b * Hl * Wl * C + c4 * 4;
  float4 r0 = f4lerp2(lx.l0, ld4(base + ((long long)ly.i0 * Wl + lx.i0) * C), lx.l1,
                      ld4(base + ((long long)ly.i0 * Wl + lx.i1) * C));
  float4 r1 = f4lerp2(lx.l0, ld4(base + ((long long)ly.i1 * Wl + lx.i0) * C), lx.l1,
                      ld4(base + ((long long)ly.i1 * Wl + lx.i1) * C));
  float4 up = f4lerp2(ly.l0, r0, ly.l1, r1);
  float4 s = ld4(skip + (long long)(b / group) * skip_bs + ((long long)oy * Wo + ox) * C + c4 * 4);
  st4(y + i * 4, f4add(s, up));
}

// The same, and the result's bf16 planes (hi, mid[, lo]) for the convolutions that consume it pre-split -- with and / or
// without their input ReLU (networks.py:26-27: ResBlock.conv1 sees relu(x), the downsample conv x).  Block = 32 pixels x 8
// channel groups, thread = (pixel, 8 channels): the 8 threads of a pixel read / write 256 contiguous bytes of its row, and
// per channel group 8 consecutive pixels store one 128-byte run of every plane (split_bf16x3_kernel's mapping, conv.hip).
__global__ __launch_bounds__(256) void upsample_add_planes_kernel(const float *__restrict__ skip, long long skip_bs,
                                                                  const float *__restrict__ low, float *__restrict__ y,
                                                                  unsigned short *__restrict__ pl0, int npl0,
                                                                  unsigned short *__restrict__ pl1, int npl1, int B, int Hl,
                                                                  int Wl, int Ho, int Wo, int C, unsigned *fault, int group) {
  const long long npix = (long long)B * Ho * Wo;
  const long long pix = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int cg = blockIdx.y * 8 + (threadIdx.x & 7);
  if (pix >= npix || cg >= C / 8) return;
  const int ox = (int)(pix % Wo);
  long long t = pix / Wo;
  const int oy = (int)(t % Ho), b = (int)(t / Ho);
  Lerp ly = lerp_coord(oy, (float)Hl / (float)Ho, Hl), lx = lerp_coord(ox, (float)Wl / (float)Wo, Wl);
  const float *base = low + (long long)b * Hl * Wl * C + cg * 8;
  const float *sk = skip + (long long)(b / group) * skip_bs + ((long long)oy * Wo + ox) * C + cg * 8;
  float4 v[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float4 r0 = f4lerp2(lx.l0, ld4(base + ((long long)ly.i0 * Wl + lx.i0) * C + 4 * h), lx.l1,
                        ld4(base + ((long long)ly.i0 * Wl + lx.i1) * C + 4 * h));
    float4 r1 = f4lerp2(lx.l0, ld4(base + ((long long)ly.i1 * Wl + lx.i0) * C + 4 * h), lx.l1,
                        ld4(base + ((long long)ly.i1 * Wl + lx.i1) * C + 4 * h));
    v[h] = f4add(ld4(sk + 4 * h), f4lerp2(ly.l0, r0, ly.l1, r1));
    st4(y + pix * C + cg * 8 + 4 * h, v[h]);
  }
  planes8_out(v, pl0, npl0, pl1, npl1, npix * C, (long long)cg * npix + pix, fault);
}

__global__ void resize_planes_kernel(const float *__restrict__ x, float *__restrict__ y, int planes, int Hi, int Wi,
                                     int Ho, int Wo, int mode) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)planes * Ho * Wo) return;
  int ox = (int)(i % Wo);
  long long t = i / Wo;
  int oy = (int)(t % Ho);
  long long pl = t / Ho;
  const float *src = x + pl * Hi * Wi;
  const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;
  if (mode == 0) {
    int iy = (Ho == Hi) ? oy : nearest_coord(oy, sh, Hi);
    int ix = (Wo == Wi) ? ox : nearest_coord(ox, sw, Wi);
    y[i] = src[(long long)iy * Wi + ix];
  } else {
    Lerp ly = lerp_coord(oy, sh, Hi), lx = lerp_coord(ox, sw, Wi);
    float r0 = lx.l0 * src[(long long)ly.i0 * Wi + lx.i0] + lx.l1 * src[(long long)ly.i0 * Wi + lx.i1];
    float r1 = lx.l0 * src[(long long)ly.i1 * Wi + lx.i0] + lx.l1 * src[(long long)ly.i1 * Wi + lx.i1];
    y[i] = ly.l0 * r0 + ly.l1 * r1;
  }
}

// ATen upsample_bicubic2d (align_corners=False, A = -0.75): unclamped source coordinate, 4x4 taps clamped to the image
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__global__ void resize_bicubic_kernel(const float *__restrict__ x, float *__restrict__ y, int planes, int Hi, int Wi,
                                      int Ho, int Wo) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)planes * Ho * Wo) return;
  int ox = (int)(i % Wo);
  long long t = i / Wo;
  int oy = (int)(t % Ho);
  long long pl = t / Ho;
  const float *src = x + pl * Hi * Wi;
  const float A = -0.75f;
  const float sy = (float)Hi / (float)Ho * (oy + 0.5f) - 0.5f, sx = (float)Wi / (float)Wo * (ox + 0.5f) - 0.5f;
  const float fy = floorf(sy), fx = floorf(sx);
  const int iy = (int)fy, ix = (int)fx;
  const float ty = sy - fy, tx = sx - fx;
  const float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
  const float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int yy = min(max(iy - 1 + a, 0), Hi - 1);
    float row = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int xx = min(max(ix - 1 + b, 0), Wi - 1);
      row += wx[b] * src[(long long)yy * Wi + xx];
    }
    acc += wy[a] * row;
  }
  y[i] = acc;
}
// horizontal flip of planes (torch.flip(dims=[-1]), swem_evaluator.py:46-49)
__global__ void flip_w_kernel(const float *__restrict__ x, float *__restrict__ y, long long rows, int W) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * W) return;
  long long r = i / W;
  int c = (int)(i - r * W);
  y[i] = x[r * W + (W - 1 - c)];
}
// y = alpha * a + beta * b   (b may be NULL: y = alpha * a); TTA score averaging, swem_evaluator.py:49-53
__global__ void lincomb_kernel(const float *__restrict__ a, float alpha, const float *__restrict__ b, float beta,
                               float *__restrict__ y, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  y[i] = b ? alpha * a[i] + beta * b[i] : alpha * a[i];
}
// swem_evaluator.py:124-130: objects that first appear at this frame: zero the predicted scores where a new object is
// annotated, then append the new objects' masks as extra channels
__global__ void inject_objects_kernel(const float *__restrict__ prob, const float *__restrict__ newm,
                                      float *__restrict__ out, int B, int N1, int Nn1, long long HW) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * HW) return;
  int b = (int)(i / HW);
  long long pix = i - b * HW;
  const float *nm = newm + (long long)b * Nn1 * HW + pix;
  float cover = 0.f;
  for (int n = 1; n < Nn1; ++n) cover += nm[(long long)n * HW];
  const int Nout = N1 + Nn1 - 1;
  float *o = out + (long long)b * Nout * HW + pix;
  const float *pp = prob + (long long)b * N1 * HW + pix;
  for (int n = 0; n < N1; ++n) o[(long long)n * HW] = cover > 0.f ? 0.f : pp[(long long)n * HW];
  for (int n = 1; n < Nn1; ++n) o[(long long)(N1 + n - 1) * HW] = nm[(long long)n * HW];
}

template <typename HT>
__global__ void mask_prep_kernel(const HT *__restrict__ hard, int Hh, int Wh, const float *__restrict__ soft, int Hs,
                                 int Ws, float *__restrict__ out, int B, int N, int h, int w) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int P = h * w;
  if (i >= (long long)B * N * P) return;
  int p = (int)(i % P);
  long long bn = i / P;
  int b = (int)(bn / N), n = (int)(bn - (long long)b * N);
  int oy = p / w, ox = p - oy * w;
  const HT *hp = hard + ((long long)b * (N + 1) + n + 1) * Hh * Wh;
  int iy = (Hh == h) ? oy : nearest_coord(oy, (float)Hh / (float)h, Hh);
  int ix = (Wh == w) ? ox : nearest_coord(ox, (float)Wh / (float)w, Wh);
  float mh = (float)hp[(long long)iy * Wh + ix];
  const float *sp = soft + ((long long)b * (N + 1) + n + 1) * Hs * Ws;
  Lerp ly = lerp_coord(oy, (float)Hs / (float)h, Hs), lx = lerp_coord(ox, (float)Ws / (float)w, Ws);
  float r0 = lx.l0 * sp[(long long)ly.i0 * Ws + lx.i0] + lx.l1 * sp[(long long)ly.i0 * Ws + lx.i1];
  float r1 = lx.l0 * sp[(long long)ly.i1 * Ws + lx.i0] + lx.l1 * sp[(long long)ly.i1 * Ws + lx.i1];
  float ms = ly.l0 * r0 + ly.l1 * r1;
  out[(bn * 2 + 0) * P + p] = (1.f - mh) * (1.f - ms);
  out[(bn * 2 + 1) * P + p] = mh * ms;
}

// ---------------------------------------------------------------- CBAM
constexpr int CBAM_CHUNKS = 32;
// stage 1: per (b, chunk): partial sum and max over the chunk's pixels for every channel.  A work item = (4 channels,
// one of nsub interleaved pixel subsets of the chunk): nsub x C/4 items per block, so a thread's chain of dependent loads is
// per / nsub pixels long (6-7 at 1/16 of 480p: the kernel is latency, not bandwidth: 6.6 MB); the subsets are combined in
// a fixed order through the LDS (deterministic sums).
__global__ __launch_bounds__(1024) void cbam_pool_partial_kernel(const float *__restrict__ x, float *__restrict__ part, int P,
                                                                 int C, int nsub) {
  extern __shared__ float4 psm[];   // [nsub][2][C/4]
  const int cq = C / 4;
  const int b = blockIdx.y, ch = blockIdx.x;
  const int per = (P + CBAM_CHUNKS - 1) / CBAM_CHUNKS;
  const int p0 = ch * per, p1 = min(P, p0 + per);
  const float ninf = -__builtin_huge_valf();
  for (int it = threadIdx.x; it < cq * nsub; it += blockDim.x) {
    const int c4 = it % cq, sub = it / cq;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), m = make_float4(ninf, ninf, ninf, ninf);
#pragma unroll 4
    for (int p = p0 + sub; p < p1; p += nsub) {
      float4 v = ld4(x + ((long long)b * P + p) * C + c4 * 4);
      s = f4add(s, v);
      m = f4max(m, v);
    }
    psm[(sub * 2) * cq + c4] = s;
    psm[(sub * 2 + 1) * cq + c4] = m;
  }
  __syncthreads();
  for (int c4 = threadIdx.x; c4 < cq; c4 += blockDim.x) {
    float4 s = psm[c4], m = psm[cq + c4];
    for (int sub = 1; sub < nsub; ++sub) {
      s = f4add(s, psm[(sub * 2) * cq + c4]);
      m = f4max(m, psm[(sub * 2 + 1) * cq + c4]);
    }
    float *dst = part + (((long long)b * CBAM_CHUNKS + ch) * 2) * C + c4 * 4;
    st4(dst, s);
    st4(dst + C, m);
  }
}
static void launch_cbam_pool(hipStream_t st, const float *x, float *part, int B, int P, int C) {
  const int nsub = C <= 1024 ? 8 : C <= 2048 ? 4 : 2;    // nsub x 2 x C floats of LDS: at most 64 KB (C <= 4096)
  hipLaunchKernelGGL(cbam_pool_partial_kernel, dim3(CBAM_CHUNKS, B), dim3(1024), (size_t)nsub * 2 * C * sizeof(float), st, x,
                     part, P, C, nsub);
}
// stage 2 + MLP (attentions.py:26-50): one block per batch item
__global__ void cbam_mlp_kernel(const float *__restrict__ part, const float *__restrict__ w1,
                                const float *__restrict__ b1, const float *__restrict__ w2,
                                const float *__restrict__ b2, float *__restrict__ cscale, int P, int C, int hid) {
  extern __shared__ float sm[];  // avg[C], max[C], hidden[2][hid]
  float *avg = sm, *mx = sm + C, *hd = sm + 2 * C;
  const int b = blockIdx.x;
  for (int it = threadIdx.x; it < 2 * C; it += blockDim.x) {   // item = (channel, sum | max): independent loads, unrolled
    const int c = it % C, kind = it / C;
    const float *src = part + ((long long)b * CBAM_CHUNKS * 2 + kind) * C + c;
    float r = kind ? -__builtin_huge_valf() : 0.f;
#pragma unroll 8
    for (int ch = 0; ch < CBAM_CHUNKS; ++ch) {
      const float v = src[(long long)ch * 2 * C];
      r = kind ? fmaxf(r, v) : r + v;
    }
    if (kind) mx[c] = r;
    else avg[c] = r / (float)P;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int j = wave; j < 2 * hid; j += nw) {  // one wave per (pool type, hidden unit)
    const float *v = j < hid ? avg : mx;
    const float *wr = w1 + (long long)(j % hid) * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += wr[c] * v[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) hd[j] = fmaxf(s + b1[j % hid], 0.f);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float sa = b2[c], sm_ = b2[c];
#pragma unroll 8
    for (int j = 0; j < hid; ++j) {
      float wv = w2[(long long)c * hid + j];
      sa += wv * hd[j];
      sm_ += wv * hd[hid + j];
    }
    cscale[(long long)b * C + c] = sigmoidf_(sa + sm_);
  }
}
// The same for C <= 512, hid <= 32 (hid % 4 == 0), 1024 threads: the kernel is three dependent rounds of global loads on one
// block per image (17 us of pure latency at 1/16 of 480p); the weights do not depend on the data, so every load of all
// three rounds is issued up front and the rounds run from registers.  Same summation order as cbam_mlp_kernel.
__global__ __launch_bounds__(1024) void cbam_mlp_fast_kernel(const float *__restrict__ part, const float *__restrict__ w1,
                                                             const float *__restrict__ b1, const float *__restrict__ w2,
                                                             const float *__restrict__ b2, float *__restrict__ cscale, int P,
                                                             int C, int hid) {
  extern __shared__ float sm[];  // avg[C], max[C], hidden[2][hid]
  float *avg = sm, *mx = sm + C, *hd = sm + 2 * C;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool pin = t < 2 * C;
  const int pc = pin ? t % C : 0, kind = pin ? t / C : 0;
  const float *src = part + ((long long)b * CBAM_CHUNKS * 2 + kind) * C + pc;
  float pv[CBAM_CHUNKS];
#pragma unroll
  for (int ch = 0; ch < CBAM_CHUNKS; ++ch) pv[ch] = pin ? src[(long long)ch * 2 * C] : 0.f;
  float w1r[4][8], b1r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = wave + 16 * r;
    const bool jin = j < 2 * hid;
    const float *wr = w1 + (long long)(jin ? j % hid : 0) * C;
    b1r[r] = jin ? b1[j % hid] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) w1r[r][k] = (jin && lane + 64 * k < C) ? wr[lane + 64 * k] : 0.f;
  }
  const bool cin = t < C;
  float4 w2r[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
    w2r[q] = (cin && 4 * q < hid) ? ld4(w2 + (long long)t * hid + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float b2r = cin ? b2[t] : 0.f;
  if (pin) {
    float r = kind ? -__builtin_huge_valf() : 0.f;
#pragma unroll
    for (int ch = 0; ch < CBAM_CHUNKS; ++ch) r = kind ? fmaxf(r, pv[ch]) : r + pv[ch];
    if (kind) mx[pc] = r;
    else avg[pc] = r / (float)P;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int j = wave + 16 * r;
    if (j >= 2 * hid) continue;
    const float *v = j < hid ? avg : mx;
    float sacc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (lane + 64 * k < C) sacc += w1r[r][k] * v[lane + 64 * k];
    for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
    if (lane == 0) hd[j] = fmaxf(sacc + b1r[r], 0.f);
  }
  __syncthreads();
  if (cin) {
    float sa = b2r, sm_ = b2r;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float wv[4] = {w2r[q].x, w2r[q].y, w2r[q].z, w2r[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = 4 * q + e;
        if (j < hid) {
          sa += wv[e] * hd[j];
          sm_ += wv[e] * hd[hid + j];
        }
      }
    }
    cscale[(long long)b * C + t] = sigmoidf_(sa + sm_);
  }
}
// channel max / mean of x*cscale per pixel: one wave per pixel (attentions.py:53-55)
__global__ void cbam_spatial_pool_kernel(const float *__restrict__ x, const float *__restrict__ cscale,
                                         float *__restrict__ comp, int B, int P, int C) {
  const int lane = threadIdx.x & 63;
  long long pix = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pix >= (long long)B * P) return;
  int b = (int)(pix / P);
  float m = -__builtin_huge_valf(), s = 0.f;
  for (int c4 = lane; c4 < C / 4; c4 += 64) {
    float4 v = ld4(x + pix * C + c4 * 4), g = ld4(cscale + (long long)b * C + c4 * 4);
    float a0 = v.x * g.x, a1 = v.y * g.y, a2 = v.z * g.z, a3 = v.w * g.w;
    m = fmaxf(fmaxf(m, fmaxf(a0, a1)), fmaxf(a2, a3));
    s += (a0 + a1) + (a2 + a3);
  }
  for (int o = 32; o > 0; o >>= 1) {
    m = fmaxf(m, __shfl_xor(m, o));
    s += __shfl_xor(s, o);
  }
  if (lane == 0) {
    comp[pix * 2] = m;
    comp[pix * 2 + 1] = s / (float)C;
  }
}
__global__ void cbam_sgate_kernel(const float *__restrict__ comp, const float *__restrict__ w7,
                                  const float *__restrict__ b7, float *__restrict__ sg, int B, int H, int W) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * H * W) return;
  int ox = (int)(i % W);
  long long t = i / W;
  int oy = (int)(t % H);
  int b = (int)(t / H);
  float acc = b7[0];
  for (int ky = 0; ky < 7; ++ky) {
    int iy = oy - 3 + ky;
    if ((unsigned)iy >= (unsigned)H) continue;
    for (int kx = 0; kx < 7; ++kx) {
      int ix = ox - 3 + kx;
      if ((unsigned)ix >= (unsigned)W) continue;
      const float *cp = comp + (((long long)b * H + iy) * W + ix) * 2;
      acc += w7[ky * 7 + kx] * cp[0] + w7[49 + ky * 7 + kx] * cp[1];
    }
  }
  sg[i] = sigmoidf_(acc);
}
__global__ void cbam_apply_kernel(const float *__restrict__ x, const float *__restrict__ cscale,
                                  const float *__restrict__ sg, float *__restrict__ y, int B, int P, int C) {
  const int cq = C / 4;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * P * cq) return;
  int c4 = (int)(i % cq);
  long long pix = i / cq;
  int b = (int)(pix / P);
  float4 v = ld4(x + i * 4), g = ld4(cscale + (long long)b * C + c4 * 4);
  float s = sg[pix];
  st4(y + i * 4, make_float4(v.x + v.x * g.x * s, v.y + v.y * g.y * s, v.z + v.z * g.z * s, v.w + v.w * g.w * s));
}

// cbam_apply_kernel plus the result's bf16 planes (the value encoder's output feeds the fuser's first convolution).
__global__ __launch_bounds__(256) void cbam_apply_planes_kernel(const float *__restrict__ x, const float *__restrict__ cscale,
                                                                const float *__restrict__ sg, float *__restrict__ y,
                                                                unsigned short *__restrict__ pl0, int npl0,
                                                                unsigned short *__restrict__ pl1, int npl1, int B, int P, int C,
                                                                unsigned *fault) {
  const long long npix = (long long)B * P;
  const long long pix = (long long)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int cg = blockIdx.y * 8 + (threadIdx.x & 7);
  if (pix >= npix || cg >= C / 8) return;
  const int b = (int)(pix / P);
  const float s = sg[pix];
  float4 v[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float4 u = ld4(x + pix * C + cg * 8 + 4 * h), g = ld4(cscale + (long long)b * C + cg * 8 + 4 * h);
    v[h] = make_float4(u.x + u.x * g.x * s, u.y + u.y * g.y * s, u.z + u.z * g.z * s, u.w + u.w * g.w * s);
    st4(y + pix * C + cg * 8 + 4 * h, v[h]);
  }
  planes8_out(v, pl0, npl0, pl1, npl1, npix * C, (long long)cg * npix + pix, fault);
}

// ---------------------------------------------------------------- CBAM backward (training; attentions.py:22-84)
// y = x + u * s,  u = x * g (channel gate g = sigmoid(mlp(avg) + mlp(max))),  s = sigmoid(conv7([max_c u, mean_c u]))
__global__ void cbam_stats_kernel(const float *__restrict__ part, float *__restrict__ avg, float *__restrict__ mx,
                                  int P, int C) {
  const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f, m = -__builtin_huge_valf();
  for (int ch = 0; ch < CBAM_CHUNKS; ++ch) {
    const float *src = part + (((long long)b * CBAM_CHUNKS + ch) * 2) * C + c;
    s += src[0];
    m = fmaxf(m, src[C]);
  }
  avg[(long long)b * C + c] = s / (float)P;
  mx[(long long)b * C + c] = m;
}
// first pixel attaining the channel maximum (torch's max-pool gradient goes to one position).
// Block = 64 channels of one batch item: 16 pixel lanes x 16 channel quads (a pixel lane walks every 16th pixel with 16-byte loads,
// the lanes' first hits are combined through the LDS).  (Rounds 1-5: one thread per channel walking all P pixels, 4-16 blocks in
// all: 147 us for 2 x 576 x 512 -- a quarter of the CBAM backward.)
__global__ __launch_bounds__(256) void cbam_argmax_pix_kernel(const float *__restrict__ x, const float *__restrict__ mx,
                                                              int *__restrict__ amax, int P, int C) {
  __shared__ int sh[16][64];
  const int b = blockIdx.y, q = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + q * 4;
  int idx[4] = {P, P, P, P};
  if (c < C) {
    const float4 m = ld4(mx + (long long)b * C + c);
    const float mv[4] = {m.x, m.y, m.z, m.w};
    for (int p = pl; p < P; p += 16) {
      const float4 v = ld4(x + ((long long)b * P + p) * C + c);
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (vv[e] == mv[e] && p < idx[e]) idx[e] = p;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) sh[pl][q * 4 + e] = idx[e];
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
    int best = P;
    for (int l = 0; l < 16; ++l) best = min(best, sh[l][threadIdx.x]);
    amax[(long long)b * C + blockIdx.x * 64 + threadIdx.x] = best < P ? best : 0;
  }
}
// da[p] = (sum_c dy * x * g) * s * (1 - s): gradient at the spatial gate's pre-activation.  One wave per pixel.
__global__ void cbam_bwd_pix1_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                     const float *__restrict__ cscale, const float *__restrict__ sg,
                                     float *__restrict__ da, int B, int P, int C) {
  const int lane = threadIdx.x & 63;
  const long long pix = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pix >= (long long)B * P) return;
  const int b = (int)(pix / P);
  float s = 0.f;
  for (int c4 = lane; c4 < C / 4; c4 += 64) {
    const float4 v = ld4(x + pix * C + c4 * 4), g = ld4(cscale + (long long)b * C + c4 * 4), d = ld4(dy + pix * C + c4 * 4);
    s += d.x * v.x * g.x + d.y * v.y * g.y + d.z * v.z * g.z + d.w * v.w * g.w;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    const float q = sg[pix];
    da[pix] = s * q * (1.f - q);
  }
}
// transposed 7x7 conv: dcomp[p][k] = sum_taps w7[k][ky][kx] * da[p + 3 - tap]
__global__ void cbam_bwd_sconv_kernel(const float *__restrict__ da, const float *__restrict__ w7,
                                      float *__restrict__ dcomp, int B, int H, int W) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * H * W) return;
  const int ix = (int)(i % W);
  long long t = i / W;
  const int iy = (int)(t % H);
  const int b = (int)(t / H);
  float a0 = 0.f, a1 = 0.f;
  for (int ky = 0; ky < 7; ++ky) {
    const int oy = iy + 3 - ky;
    if ((unsigned)oy >= (unsigned)H) continue;
    for (int kx = 0; kx < 7; ++kx) {
      const int ox = ix + 3 - kx;
      if ((unsigned)ox >= (unsigned)W) continue;
      const float d = da[((long long)b * H + oy) * W + ox];
      a0 += w7[ky * 7 + kx] * d;
      a1 += w7[49 + ky * 7 + kx] * d;
    }
  }
  dcomp[i * 2] = a0;
  dcomp[i * 2 + 1] = a1;
}
// dw7[k][tap] += sum_p da[p] * comp[p + tap - 3][k];  db7 += sum da.  One block per (k, tap); block 98 = bias.
__global__ __launch_bounds__(256) void cbam_bwd_w7_kernel(const float *__restrict__ da, const float *__restrict__ comp,
                                                          float *__restrict__ dw7, float *__restrict__ db7, int B,
                                                          int H, int W) {
  __shared__ float sh[256];
  const int item = blockIdx.x;
  const int k = item / 49, tap = item - k * 49, ky = tap / 7, kx = tap - ky * 7;
  float s = 0.f;
  for (long long i = threadIdx.x; i < (long long)B * H * W; i += 256) {
    if (item == 98) {
      s += da[i];
      continue;
    }
    const int ox = (int)(i % W);
    long long t = i / W;
    const int oy = (int)(t % H);
    const int b = (int)(t / H);
    const int iy = oy - 3 + ky, ix = ox - 3 + kx;
    if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
    s += da[i] * comp[(((long long)b * H + iy) * W + ix) * 2 + k];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (item == 98) db7[0] += sh[0];
    else dw7[item] += sh[0];
  }
}
// du = dy * s + dcomp1 / C + [c == argmax_c u] * dcomp0;  dxp = dy + du * g.  One wave per pixel.
__global__ void cbam_bwd_pix2_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                     const float *__restrict__ cscale, const float *__restrict__ sg,
                                     const float *__restrict__ comp, const float *__restrict__ dcomp,
                                     float *__restrict__ du, float *__restrict__ dxp, int B, int P, int C) {
  const int lane = threadIdx.x & 63;
  const long long pix = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pix >= (long long)B * P) return;
  const int b = (int)(pix / P);
  const float m = comp[pix * 2], d0 = dcomp[pix * 2], d1 = dcomp[pix * 2 + 1] / (float)C, q = sg[pix];
  int first = 1 << 30;  // first channel whose gated value equals the maximum
  for (int c4 = lane; c4 < C / 4; c4 += 64) {
    const float4 v = ld4(x + pix * C + c4 * 4), g = ld4(cscale + (long long)b * C + c4 * 4);
    const float u[4] = {v.x * g.x, v.y * g.y, v.z * g.z, v.w * g.w};
#pragma unroll
    for (int e = 3; e >= 0; --e)
      if (u[e] == m) first = min(first, c4 * 4 + e);
  }
  for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
  for (int c4 = lane; c4 < C / 4; c4 += 64) {
    const float4 g = ld4(cscale + (long long)b * C + c4 * 4), d = ld4(dy + pix * C + c4 * 4);
    float r[4] = {d.x * q + d1, d.y * q + d1, d.z * q + d1, d.w * q + d1};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c4 * 4 + e == first) r[e] += d0;
    st4(du + pix * C + c4 * 4, make_float4(r[0], r[1], r[2], r[3]));
    st4(dxp + pix * C + c4 * 4, make_float4(d.x + r[0] * g.x, d.y + r[1] * g.y, d.z + r[2] * g.z, d.w + r[3] * g.w));
  }
}
// channel-gate MLP backward, one block, batch items in sequence (the parameter gradients accumulate)
__global__ __launch_bounds__(1024) void cbam_bwd_mlp_kernel(const float *__restrict__ avg, const float *__restrict__ mx,
                                                           const float *__restrict__ cscale,
                                                           const float *__restrict__ dg, const float *__restrict__ w1,
                                                           const float *__restrict__ b1, const float *__restrict__ w2,
                                                           float *__restrict__ dw1, float *__restrict__ db1,
                                                           float *__restrict__ dw2, float *__restrict__ db2,
                                                           float *__restrict__ davg, float *__restrict__ dmx, int B,
                                                           int C, int hid) {
  extern __shared__ float sm[];  // dA[C], h[2][hid], dh[2][hid], partial davg / dmax [2][4][C]
  float *dA = sm, *hd = sm + C, *dh = hd + 2 * hid, *pa = dh + 2 * hid;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int b = 0; b < B; ++b) {
    const float *av = avg + (long long)b * C, *mv = mx + (long long)b * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float g = cscale[(long long)b * C + c];
      dA[c] = dg[(long long)b * C + c] * g * (1.f - g);
    }
    for (int j = wave; j < 2 * hid; j += nw) {
      const float *v = j < hid ? av : mv;
      const float *wr = w1 + (long long)(j % hid) * C;
      float s = 0.f;
      for (int c = lane; c < C; c += 64) s += wr[c] * v[c];
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (lane == 0) hd[j] = fmaxf(s + b1[j % hid], 0.f);
    }
    __syncthreads();
    for (int j = wave; j < hid; j += nw) {  // dh = (h > 0) * W2^T dA  (the same W2 for both pools)
      float s = 0.f;
      for (int c = lane; c < C; c += 64) s += w2[(long long)c * hid + j] * dA[c];
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (lane == 0) {
        dh[j] = hd[j] > 0.f ? s : 0.f;
        dh[hid + j] = hd[hid + j] > 0.f ? s : 0.f;
        db1[j] += dh[j] + dh[hid + j];
      }
    }
    __syncthreads();
    // (round 6: work items (channel, j lane) over 1024 threads -- four lanes take every fourth hidden unit; rounds 1-5 walked all
    // hid units per channel on 256 threads, a chain of dependent read-modify-writes: 132 us per launch)
    for (int item = threadIdx.x; item < 4 * C; item += blockDim.x) {
      const int c = item % C, jl = item / C;
      float da_ = 0.f, dm_ = 0.f;
      const float a_c = dA[c], av_c = av[c], mv_c = mv[c];
      for (int j = jl; j < hid; j += 4) {
        dw2[(long long)c * hid + j] += a_c * (hd[j] + hd[hid + j]);
        dw1[(long long)j * C + c] += dh[j] * av_c + dh[hid + j] * mv_c;
        const float wv = w1[(long long)j * C + c];
        da_ += wv * dh[j];
        dm_ += wv * dh[hid + j];
      }
      pa[jl * C + c] = da_;
      pa[(4 + jl) * C + c] = dm_;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      db2[c] += 2.f * dA[c];
      davg[(long long)b * C + c] = ((pa[c] + pa[C + c]) + pa[2 * C + c]) + pa[3 * C + c];
      dmx[(long long)b * C + c] = ((pa[4 * C + c] + pa[5 * C + c]) + pa[6 * C + c]) + pa[7 * C + c];
    }
    __syncthreads();
  }
}
// dx = dxp + davg / P + [p == argmax_p x] * dmax
__global__ void cbam_bwd_pix3_kernel(const float *__restrict__ dxp, const float *__restrict__ davg,
                                     const float *__restrict__ dmx, const int *__restrict__ amax,
                                     float *__restrict__ dx, int B, int P, int C) {
  const int cq = C / 4;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * P * cq) return;
  const int c4 = (int)(i % cq);
  const long long pix = i / cq;
  const int b = (int)(pix / P), p = (int)(pix - (long long)b * P);
  const float4 v = ld4(dxp + i * 4), a = ld4(davg + (long long)b * C + c4 * 4), m = ld4(dmx + (long long)b * C + c4 * 4);
  const int4 am = *reinterpret_cast<const int4 *>(amax + (long long)b * C + c4 * 4);
  const float ip = 1.f / (float)P;
  st4(dx + i * 4, make_float4(v.x + a.x * ip + (am.x == p ? m.x : 0.f), v.y + a.y * ip + (am.y == p ? m.y : 0.f),
                              v.z + a.z * ip + (am.z == p ? m.z : 0.f), v.w + a.w * ip + (am.w == p ? m.w : 0.f)));
}
// dg[b][c] = sum_p du[b][p][c] * x[b][p][c].  Block = 64 channels of one batch item, 16 pixel lanes x 16 channel quads; the lanes'
// partial sums are added in lane order (deterministic).  (Rounds 1-5: one thread per 4 channels walking all P pixels, 4 blocks of one
// wave: 149 us for 2 x 576 x 512.)
__global__ __launch_bounds__(256) void cbam_bwd_dg_kernel(const float *__restrict__ du, const float *__restrict__ x,
                                                          float *__restrict__ dg, int P, int C) {
  __shared__ float sh[16][64];
  const int b = blockIdx.y, q = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + q * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C)
    for (int p = pl; p < P; p += 16) {
      const float4 a = ld4(du + ((long long)b * P + p) * C + c), v = ld4(x + ((long long)b * P + p) * C + c);
      s.x += a.x * v.x; s.y += a.y * v.y; s.z += a.z * v.z; s.w += a.w * v.w;
    }
  sh[pl][q * 4] = s.x; sh[pl][q * 4 + 1] = s.y; sh[pl][q * 4 + 2] = s.z; sh[pl][q * 4 + 3] = s.w;
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
    float t = sh[0][threadIdx.x];
    for (int l = 1; l < 16; ++l) t += sh[l][threadIdx.x];
    dg[(long long)b * C + blockIdx.x * 64 + threadIdx.x] = t;
  }
}

// ---------------------------------------------------------------- decoder heads
// conv3x3(relu(x)) -> 1 channel (networks.py:213).  As a GEMM it is [pixels x C] . [C x 9 taps]: every INPUT pixel's nine
// tap products d[p][tap] = relu(x[p]) . w[tap] are one row of a v_mfma_f32_16x16x4_f32 tile (A = 16 consecutive pixels of a
// row, B = the nine tap filters padded to 16 columns; exact fp32 FMA chains), and an output pixel is the sum of nine of them
// from its 3x3 neighbourhood.  Block = 6 x 30 output pixels: its 8 x 32 input pixels (halo included) are 16 segments of 16
// pixels, four per wave; the tap products go through the LDS, then 180 threads add their nine neighbours.  Each input pixel
// is read ONCE per block as 16-byte loads (round 2 read it nine times, one wave per output pixel with a 64-lane shuffle
// reduction each: 42 us for 2x120x216x256, 68 us beside three other sequences -- 11 TB/s of L2 traffic).
typedef float f32x4p __attribute__((ext_vector_type(4)));
constexpr int PH_TH = 6, PH_TW = 30;   // output tile; input tile (PH_TH + 2) x (PH_TW + 2) = 8 x 32
template <int CQ>   // C / 16: channel groups of 16 (four k-steps of four channels each)
__global__ __launch_bounds__(256) void pred_head_mfma_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ bias, float *__restrict__ logit, int B,
                                                              int H, int W, int tiles_x, int tiles_y) {
  __shared__ float t[(PH_TH + 2) * (PH_TW + 2)][12];   // tap products of the input tile (9 used; 48-byte rows)
  const int C = 16 * CQ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y, b = bid / tiles_y;
  const int oy0 = ty * PH_TH, ox0 = tx * PH_TW;
  // B operand: lane (column li = tap, k-slot g) holds w[tap][16 q + 4 g + e] for step (q, e); taps 9..15 are zero columns
  float4 wf[CQ];
#pragma unroll
  for (int q = 0; q < CQ; ++q) wf[q] = li < 9 ? ld4(w + li * C + 16 * q + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  // the next segment's 16 loads are in flight under the current segment's 64 dependent MFMAs (two register sets)
  float4 xv[2][CQ];
  auto load_seg = [&](int sidx, float4 (&dst)[CQ]) __attribute__((always_inline)) {
    const int seg = wave * 4 + sidx;             // 16 segments: row seg / 2 of the input tile, half seg % 2
    const int iy = oy0 - 1 + (seg >> 1), ix = ox0 - 1 + 16 * (seg & 1) + li;
    const bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    const float *xp = x + (((long long)b * H + (in ? iy : 0)) * W + (in ? ix : 0)) * C + 4 * g;
#pragma unroll
    for (int q = 0; q < CQ; ++q) dst[q] = in ? ld4(xp + 16 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  load_seg(0, xv[0]);
#pragma unroll
  for (int sidx = 0; sidx < 4; ++sidx) {
    const int seg = wave * 4 + sidx;
    if (sidx + 1 < 4) load_seg(sidx + 1, xv[(sidx + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    f32x4p acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};   // two chains: the MFMA's 40-cycle dependent latency
#pragma unroll
    for (int q = 0; q < CQ; ++q) {
      const float4 v = xv[sidx & 1][q];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fmaxf(v.x, 0.f), wf[q].x, acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fmaxf(v.y, 0.f), wf[q].y, acc1, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fmaxf(v.z, 0.f), wf[q].z, acc, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fmaxf(v.w, 0.f), wf[q].w, acc1, 0, 0, 0);
    }
    acc += acc1;
    __builtin_amdgcn_sched_barrier(0);
    // D[pixel i][tap j] lives in lane j + 16 (i >> 2), register i & 3
    if (li < 9) {
#pragma unroll
      for (int e = 0; e < 4; ++e) t[(seg >> 1) * (PH_TW + 2) + 16 * (seg & 1) + 4 * g + e][li] = acc[e];
    }
  }
  __syncthreads();
  if (tid < PH_TH * PH_TW) {
    const int r = tid / PH_TW, c = tid - r * PH_TW;
    const int oy = oy0 + r, ox = ox0 + c;
    if (oy < H && ox < W) {
      float s = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) s += t[(r + ky) * (PH_TW + 2) + c + kx][ky * 3 + kx];
      logit[((long long)b * H + oy) * W + ox] = s + bias[0];
    }
  }
}

// generic channel counts (C % 4 == 0): one wave per output pixel, lanes over channel groups
__global__ void pred_head_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                 const float *__restrict__ bias, float *__restrict__ logit, int B, int H, int W,
                                 int C) {
  const int lane = threadIdx.x & 63;
  long long pix = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (pix >= (long long)B * H * W) return;
  int ox = (int)(pix % W);
  long long t = pix / W;
  int oy = (int)(t % H);
  int b = (int)(t / H);
  float s = 0.f;
  for (int ky = 0; ky < 3; ++ky) {
    int iy = oy - 1 + ky;
    if ((unsigned)iy >= (unsigned)H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      int ix = ox - 1 + kx;
      if ((unsigned)ix >= (unsigned)W) continue;
      const float *xp = x + (((long long)b * H + iy) * W + ix) * C;
      const float *wp = w + (ky * 3 + kx) * C;
      for (int c4 = lane; c4 < C / 4; c4 += 64) {
        float4 v = ld4(xp + c4 * 4), q = ld4(wp + c4 * 4);
        s += fmaxf(v.x, 0.f) * q.x + fmaxf(v.y, 0.f) * q.y + fmaxf(v.z, 0.f) * q.z + fmaxf(v.w, 0.f) * q.w;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) logit[pix] = s + bias[0];
}

__global__ void decode_head_kernel(const float *__restrict__ logit4, const float *__restrict__ valid,
                                   float *__restrict__ logits, float *__restrict__ prob,
                                   long long *__restrict__ amax, int B, int N, int h4, int w4, int Ho, int Wo) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long HW = (long long)Ho * Wo;
  if (i >= B * HW) return;
  int b = (int)(i / HW);
  long long pix = i - b * HW;
  int oy = (int)(pix / Wo), ox = (int)(pix - (long long)oy * Wo);
  Lerp ly = lerp_coord(oy, (float)h4 / (float)Ho, h4), lx = lerp_coord(ox, (float)w4 / (float)Wo, w4);
  float *lg = logits + (long long)b * (N + 1) * HW + pix;
  float *pr = prob + (long long)b * (N + 1) * HW + pix;
  float bg = 1.f, mx = -__builtin_huge_valf();
  for (int n = 0; n < N; ++n) {
    const float *src = logit4 + ((long long)b * N + n) * h4 * w4;
    float r0 = lx.l0 * src[ly.i0 * w4 + lx.i0] + lx.l1 * src[ly.i0 * w4 + lx.i1];
    float r1 = lx.l0 * src[ly.i1 * w4 + lx.i0] + lx.l1 * src[ly.i1 * w4 + lx.i1];
    float p = sigmoidf_(ly.l0 * r0 + ly.l1 * r1);
    if (valid) p *= valid[(long long)b * (N + 1) + n + 1];
    bg *= 1.f - p;
    float pc = fminf(fmaxf(p, 1e-7f), 1.f - 1e-7f);
    float l = logf(pc / (1.f - pc));
    lg[(long long)(n + 1) * HW] = l;
    mx = fmaxf(mx, l);
  }
  {
    float pc = fminf(fmaxf(bg, 1e-7f), 1.f - 1e-7f);
    float l = logf(pc / (1.f - pc));
    lg[0] = l;
    mx = fmaxf(mx, l);
  }
  float sum = 0.f;
  for (int n = 0; n <= N; ++n) sum += expf(lg[(long long)n * HW] - mx);
  float best = -1.f;
  int bi = 0;
  for (int n = 0; n <= N; ++n) {
    float v = expf(lg[(long long)n * HW] - mx) / sum;
    pr[(long long)n * HW] = v;
    if (v > best) {
      best = v;
      bi = n;
    }
  }
  if (amax) amax[i] = bi;
}

__global__ void argmax_onehot_kernel(const float *__restrict__ prob, long long *__restrict__ amax,
                                     long long *__restrict__ onehot, int B, int N1, long long HW) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * HW) return;
  int b = (int)(i / HW);
  long long pix = i - b * HW;
  const float *src = prob + (long long)b * N1 * HW + pix;
  float best = src[0];
  int bi = 0;
  for (int n = 1; n < N1; ++n) {
    float v = src[(long long)n * HW];
    if (v > best) {
      best = v;
      bi = n;
    }
  }
  if (amax) amax[i] = bi;
  if (onehot)
    for (int n = 0; n < N1; ++n) onehot[((long long)b * N1 + n) * HW + pix] = (n == bi) ? 1 : 0;
}

__global__ void transpose_kernel(const float *__restrict__ in, float *__restrict__ out, int R, int Cc, int ld) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const float *src = in + (long long)b * R * Cc;
  float *dst = out + (long long)b * Cc * ld;
  int c = blockIdx.x * 32 + threadIdx.x;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    int r = blockIdx.y * 32 + j;
    tile[j][threadIdx.x] = (r < R && c < Cc) ? src[(long long)r * Cc + c] : 0.f;
  }
  __syncthreads();
  int r = blockIdx.y * 32 + threadIdx.x;  // output column (may be padding: r in [R, ld))
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    int cc = blockIdx.x * 32 + j;
    if (cc < Cc && r < ld) dst[(long long)cc * ld + r] = tile[threadIdx.x][j];
  }
}

// channel concat of two NHWC sources (either may be shared by all batch items: bs = 0)
__global__ void concat2_kernel(const float *__restrict__ x0, int c0, long long bs0, const float *__restrict__ x1,
                               int c1, long long bs1, float *__restrict__ y, int B, long long P) {
  const int cq = (c0 + c1) / 4;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * P * cq) return;
  int c = (int)(i % cq) * 4;
  long long t = i / cq;
  long long p = t % P;
  int b = (int)(t / P);
  float4 v = c < c0 ? ld4(x0 + b * bs0 + p * c0 + c) : ld4(x1 + b * bs1 + p * c1 + (c - c0));
  st4(y + i * 4, v);
}

// int64 index maps -> uint8 before they leave the device (basic_evaluator.py:176: .cpu().numpy().astype(np.uint8)):
// 8x less PCIe traffic; 16 maps per thread, one 16-byte store
__global__ void pack_u8_kernel(const long long *__restrict__ x, unsigned char *__restrict__ y, long long n) {
  const long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i0 >= n) return;
  if (i0 + 16 <= n) {
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned v = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) v |= (unsigned)(x[i0 + q * 4 + e] & 0xff) << (8 * e);
      w[q] = v;
    }
    *reinterpret_cast<uint4 *>(y + i0) = make_uint4(w[0], w[1], w[2], w[3]);
  } else {
    for (long long i = i0; i < n; ++i) y[i] = (unsigned char)(x[i] & 0xff);
  }
}

inline dim3 grid1(long long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" int swem_prep_key_input_f32(void *stream, const float *frames, const float *mean3, const float *std3,
                                       float *out, int B, int H, int W) {
  SWEM_REQUIRE(frames && mean3 && std3 && out, SWEM_E_ARG, "prep_key_input: null pointer");
  // mean/std are HOST pointers to 3 floats each (module buffers, networks.py:157-158)
  float3 m = make_float3(mean3[0], mean3[1], mean3[2]), s = make_float3(std3[0], std3[1], std3[2]);
  long long n = (long long)B * H * W;
  hipLaunchKernelGGL(prep_key_input_kernel, grid1(n), dim3(256), 0, ST, frames, m, s, out, B, (long long)H * W);
  SWEM_CHECK_LAUNCH("prep_key_input");
  return SWEM_OK;
}

extern "C" int swem_prep_value_input_f32(void *stream, const float *frame, const float *masks, const float *mean3,
                                         const float *std3, float *out, int B, int N, int H, int W,
                                         int single_obj) {
  SWEM_REQUIRE(frame && masks && mean3 && std3 && out && N > 0, SWEM_E_ARG, "prep_value_input: bad argument");
  float3 m = make_float3(mean3[0], mean3[1], mean3[2]), s = make_float3(std3[0], std3[1], std3[2]);
  long long n = (long long)B * N * H * W;
  hipLaunchKernelGGL(prep_value_input_kernel, grid1(n), dim3(256), 0, ST, frame, masks, m, s, out, B, N,
                     (long long)H * W, single_obj);
  SWEM_CHECK_LAUNCH("prep_value_input");
  return SWEM_OK;
}

extern "C" int swem_prep_input_s2d_f32(void *stream, const float *frame, const float *masks, const float *mean3,
                                       const float *std3, float *out, void *planes, int nplanes, int B, int N, int H, int W,
                                       int single_obj, void *fault) {
  SWEM_REQUIRE(frame && mean3 && std3 && (out || planes) && B > 0 && N > 0, SWEM_E_ARG, "prep_input_s2d: bad argument");
  SWEM_REQUIRE(H % 2 == 0 && W % 2 == 0, SWEM_E_SHAPE, "prep_input_s2d: the frame size must be even (got %dx%d)", H, W);
  SWEM_REQUIRE(!planes || nplanes == 2 || nplanes == 3 || nplanes == SWEM_PLANES_F16, SWEM_E_ARG, "prep_input_s2d: 2 or 3 planes, or SWEM_PLANES_F16");
  float3 m = make_float3(mean3[0], mean3[1], mean3[2]), s = make_float3(std3[0], std3[1], std3[2]);
  const long long n = (long long)B * N * (H / 2 + 1) * (W / 2 + 1) * 4;
  hipLaunchKernelGGL(prep_input_s2d_kernel, grid1(n), dim3(256), 0, ST, frame, masks, m, s, out,
                     static_cast<unsigned short *>(planes), nplanes, B, N, H, W, single_obj, static_cast<unsigned *>(fault));
  SWEM_CHECK_LAUNCH("prep_input_s2d");
  return SWEM_OK;
}

extern "C" int swem_maxpool3x3s2_nhwc_f32(void *stream, const float *x, float *y, int B, int H, int W, int C) {
  SWEM_REQUIRE(x && y && C % 4 == 0, SWEM_E_SHAPE, "maxpool: C %% 4 != 0");
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_kernel, grid1((long long)B * Ho * Wo * (C / 4)), dim3(256), 0, ST, x, y, B, H, W, C, Ho,
                     Wo);
  SWEM_CHECK_LAUNCH("maxpool");
  return SWEM_OK;
}

extern "C" int swem_maxpool3x3s2_nhwc_f32_planes(void *stream, const float *x, float *y, int B, int H, int W, int C,
                                                 void *planes, int nplanes, void *planes_relu, int nplanes_relu, void *fault) {
  SWEM_REQUIRE(x && y && C % 8 == 0, SWEM_E_SHAPE, "maxpool_planes: need C %% 8 == 0");
  SWEM_REQUIRE((!planes || (nplanes >= 2 && nplanes <= 4)) && (!planes_relu || (nplanes_relu >= 2 && nplanes_relu <= 4)),
               SWEM_E_ARG, "maxpool_planes: 2 or 3 planes per variant");
  int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long npix = (long long)B * Ho * Wo;
  hipLaunchKernelGGL(maxpool_planes_kernel, dim3((unsigned)cdiv(npix, 32), (unsigned)cdiv(C / 8, 8)), dim3(256), 0, ST, x, y,
                     static_cast<unsigned short *>(planes), nplanes, static_cast<unsigned short *>(planes_relu), nplanes_relu,
                     B, H, W, C, Ho, Wo, static_cast<unsigned *>(fault));
  SWEM_CHECK_LAUNCH("maxpool_planes");
  return SWEM_OK;
}

// group: `group` consecutive batch items share one skip image (item b reads skip image b / group: the N objects of a clip share
// the clip's skip feature, swem.py:94-95, with several clips in the batch) -- no per-object copy of the skip maps
static int upsample_add_launch(void *stream, const float *skip, long long skip_bs, int group, const float *low, float *y, int B, int Hl,
                               int Wl, int Ho, int Wo, int C) {
  SWEM_REQUIRE(skip && low && y && C % 4 == 0 && group >= 1, SWEM_E_SHAPE, "upsample_add: bad argument");
  hipLaunchKernelGGL(upsample_add_kernel, grid1((long long)B * Ho * Wo * (C / 4)), dim3(256), 0, ST, skip, skip_bs,
                     low, y, B, Hl, Wl, Ho, Wo, C, group);
  SWEM_CHECK_LAUNCH("upsample_add");
  return SWEM_OK;
}
static int upsample_add_planes_launch(void *stream, const float *skip, long long skip_bs, int group, const float *low, float *y, int B,
                                      int Hl, int Wl, int Ho, int Wo, int C, void *planes, int nplanes, void *planes_relu,
                                      int nplanes_relu, void *fault) {
  SWEM_REQUIRE(skip && low && y && C % 8 == 0 && group >= 1, SWEM_E_SHAPE, "upsample_add_planes: need C %% 8 == 0");
  SWEM_REQUIRE((!planes || (nplanes >= 2 && nplanes <= 4)) && (!planes_relu || (nplanes_relu >= 2 && nplanes_relu <= 4)),
               SWEM_E_ARG, "upsample_add_planes: 2 or 3 planes per variant");
  const long long npix = (long long)B * Ho * Wo;
  hipLaunchKernelGGL(upsample_add_planes_kernel, dim3((unsigned)cdiv(npix, 32), (unsigned)cdiv(C / 8, 8)), dim3(256), 0, ST,
                     skip, skip_bs, low, y, static_cast<unsigned short *>(planes), nplanes,
                     static_cast<unsigned short *>(planes_relu), nplanes_relu, B, Hl, Wl, Ho, Wo, C, static_cast<unsigned *>(fault), group);
  SWEM_CHECK_LAUNCH("upsample_add_planes");
  return SWEM_OK;
}

extern "C" int swem_upsample_add_nhwc_f32(void *stream, const float *skip, long long skip_bs, const float *low,
                                          float *y, int B, int Hl, int Wl, int Ho, int Wo, int C) {
  return upsample_add_launch(stream, skip, skip_bs, 1, low, y, B, Hl, Wl, Ho, Wo, C);
}

extern "C" int swem_upsample_add_nhwc_f32_planes(void *stream, const float *skip, long long skip_bs, const float *low,
                                                 float *y, int B, int Hl, int Wl, int Ho, int Wo, int C, void *planes,
                                                 int nplanes, void *planes_relu, int nplanes_relu, void *fault) {
  return upsample_add_planes_launch(stream, skip, skip_bs, 1, low, y, B, Hl, Wl, Ho, Wo, C, planes, nplanes, planes_relu, nplanes_relu,
                                    fault);
}

extern "C" int swem_upsample_add_grouped_nhwc_f32(void *stream, const float *skip, long long skip_bs, int group, const float *low,
                                                  float *y, int B, int Hl, int Wl, int Ho, int Wo, int C) {
  return upsample_add_launch(stream, skip, skip_bs, group, low, y, B, Hl, Wl, Ho, Wo, C);
}

extern "C" int swem_upsample_add_grouped_nhwc_f32_planes(void *stream, const float *skip, long long skip_bs, int group,
                                                         const float *low, float *y, int B, int Hl, int Wl, int Ho, int Wo, int C,
                                                         void *planes, int nplanes, void *planes_relu, int nplanes_relu, void *fault) {
  return upsample_add_planes_launch(stream, skip, skip_bs, group, low, y, B, Hl, Wl, Ho, Wo, C, planes, nplanes, planes_relu,
                                    nplanes_relu, fault);
}

extern "C" int swem_resize_planes_f32(void *stream, const float *x, float *y, int planes, int Hi, int Wi, int Ho,
                                      int Wo, int mode) {
  SWEM_REQUIRE(x && y && mode >= 0 && mode <= 3, SWEM_E_ARG, "resize_planes: bad argument");
  if (mode == 2)
    hipLaunchKernelGGL(resize_bicubic_kernel, grid1((long long)planes * Ho * Wo), dim3(256), 0, ST, x, y, planes, Hi, Wi,
                       Ho, Wo);
  else if (mode == 3) {
    SWEM_REQUIRE(Hi == Ho && Wi == Wo, SWEM_E_SHAPE, "resize_planes: flip keeps the size");
    hipLaunchKernelGGL(flip_w_kernel, grid1((long long)planes * Ho * Wo), dim3(256), 0, ST, x, y,
                       (long long)planes * Ho, Wo);
  } else
    hipLaunchKernelGGL(resize_planes_kernel, grid1((long long)planes * Ho * Wo), dim3(256), 0, ST, x, y, planes, Hi, Wi,
                       Ho, Wo, mode);
  SWEM_CHECK_LAUNCH("resize_planes");
  return SWEM_OK;
}

extern "C" int swem_mask_prep_f32(void *stream, const void *hard, int hard_is_i64, int Hh, int Wh, const float *soft,
                                  int Hs, int Ws, float *out, int B, int N, int h, int w) {
  SWEM_REQUIRE(hard && soft && out && N > 0, SWEM_E_ARG, "mask_prep: bad argument");
  long long n = (long long)B * N * h * w;
  if (hard_is_i64)
    hipLaunchKernelGGL(mask_prep_kernel<long long>, grid1(n), dim3(256), 0, ST, static_cast<const long long *>(hard),
                       Hh, Wh, soft, Hs, Ws, out, B, N, h, w);
  else
    hipLaunchKernelGGL(mask_prep_kernel<float>, grid1(n), dim3(256), 0, ST, static_cast<const float *>(hard), Hh, Wh,
                       soft, Hs, Ws, out, B, N, h, w);
  SWEM_CHECK_LAUNCH("mask_prep");
  return SWEM_OK;
}

extern "C" size_t swem_cbam_workspace(int B, int H, int W, int C) {
  return ((size_t)B * CBAM_CHUNKS * 2 * C + (size_t)B * H * W * 3) * sizeof(float);
}

static int cbam_impl(void *stream, const float *x, const float *w1, const float *b1, const float *w2, const float *b2,
                     const float *w7, const float *b7, float *cscale, float *y, int B, int H, int W, int C, int hid, void *ws,
                     size_t ws_bytes, void *planes, int nplanes, void *planes_relu, int nplanes_relu, void *fault) {
  SWEM_REQUIRE(x && w1 && b1 && w2 && b2 && w7 && b7 && cscale && y, SWEM_E_ARG, "cbam: null pointer");
  SWEM_REQUIRE(C % 4 == 0 && C <= 4096 && hid > 0 && hid <= 256, SWEM_E_SHAPE, "cbam: unsupported C/hid");
  SWEM_REQUIRE(ws && ws_bytes >= swem_cbam_workspace(B, H, W, C), SWEM_E_WORKSPACE, "cbam: workspace too small");
  const int P = H * W;
  float *part = static_cast<float *>(ws);
  float *comp = part + (size_t)B * CBAM_CHUNKS * 2 * C;
  float *sg = comp + (size_t)B * P * 2;
  launch_cbam_pool(ST, x, part, B, P, C);
  SWEM_CHECK_LAUNCH("cbam_pool_partial");
  // one block per batch item, 16 waves: the 2*hid hidden units are a wave each (4 waves took 48 us for this tiny MLP)
  if (C <= 512 && hid <= 32 && hid % 4 == 0) {
    hipLaunchKernelGGL(cbam_mlp_fast_kernel, dim3(B), dim3(1024), (2 * C + 2 * hid) * sizeof(float), ST, part, w1, b1, w2, b2,
                       cscale, P, C, hid);
  } else {
    hipLaunchKernelGGL(cbam_mlp_kernel, dim3(B), dim3(1024), (2 * C + 2 * hid) * sizeof(float), ST, part, w1, b1, w2, b2,
                       cscale, P, C, hid);
  }
  SWEM_CHECK_LAUNCH("cbam_mlp");
  hipLaunchKernelGGL(cbam_spatial_pool_kernel, grid1((long long)B * P * 64), dim3(256), 0, ST, x, cscale, comp, B, P,
                     C);
  SWEM_CHECK_LAUNCH("cbam_spatial_pool");
  hipLaunchKernelGGL(cbam_sgate_kernel, grid1((long long)B * P), dim3(256), 0, ST, comp, w7, b7, sg, B, H, W);
  SWEM_CHECK_LAUNCH("cbam_sgate");
  if (planes || planes_relu) {
    SWEM_REQUIRE(C % 8 == 0, SWEM_E_SHAPE, "cbam_planes: need C %% 8 == 0");
    SWEM_REQUIRE((!planes || (nplanes >= 2 && nplanes <= 4)) && (!planes_relu || (nplanes_relu >= 2 && nplanes_relu <= 4)),
                 SWEM_E_ARG, "cbam_planes: 2 or 3 planes per variant");
    hipLaunchKernelGGL(cbam_apply_planes_kernel, dim3((unsigned)cdiv((long long)B * P, 32), (unsigned)cdiv(C / 8, 8)),
                       dim3(256), 0, ST, x, cscale, sg, y, static_cast<unsigned short *>(planes), nplanes,
                       static_cast<unsigned short *>(planes_relu), nplanes_relu, B, P, C, static_cast<unsigned *>(fault));
  } else {
    hipLaunchKernelGGL(cbam_apply_kernel, grid1((long long)B * P * (C / 4)), dim3(256), 0, ST, x, cscale, sg, y, B, P, C);
  }
  SWEM_CHECK_LAUNCH("cbam_apply");
  return SWEM_OK;
}

extern "C" int swem_cbam_f32(void *stream, const float *x, const float *w1, const float *b1, const float *w2,
                             const float *b2, const float *w7, const float *b7, float *cscale, float *y, int B, int H,
                             int W, int C, int hid, void *ws, size_t ws_bytes) {
  return cbam_impl(stream, x, w1, b1, w2, b2, w7, b7, cscale, y, B, H, W, C, hid, ws, ws_bytes, nullptr, 3, nullptr, 3, nullptr);
}

extern "C" int swem_cbam_f32_planes(void *stream, const float *x, const float *w1, const float *b1, const float *w2,
                                    const float *b2, const float *w7, const float *b7, float *cscale, float *y, int B, int H,
                                    int W, int C, int hid, void *ws, size_t ws_bytes, void *planes, int nplanes,
                                    void *planes_relu, int nplanes_relu, void *fault) {
  return cbam_impl(stream, x, w1, b1, w2, b2, w7, b7, cscale, y, B, H, W, C, hid, ws, ws_bytes, planes, nplanes, planes_relu,
                   nplanes_relu, fault);
}

// backward of y = x + CBAM(x) (swem_cbam_f32): dx; the six parameter gradients are ACCUMULATED.
extern "C" size_t swem_cbam_bwd_workspace(int B, int H, int W, int C) {
  const size_t P = (size_t)H * W;
  // part, comp, sg, da, dcomp | cscale, avg, max, dg, davg, dmax, amax | du, dxp
  return ((size_t)B * CBAM_CHUNKS * 2 * C + (size_t)B * P * (2 + 1 + 1 + 2) + (size_t)B * C * 7 + 2 * (size_t)B * P * C) *
         sizeof(float);
}
extern "C" int swem_cbam_bwd_f32(void *stream, const float *x, const float *w1, const float *b1, const float *w2,
                                 const float *b2, const float *w7, const float *b7, const float *dy, float *dx,
                                 float *dw1, float *db1, float *dw2, float *db2, float *dw7, float *db7, int B, int H,
                                 int W, int C, int hid, void *ws, size_t ws_bytes) {
  SWEM_REQUIRE(x && w1 && b1 && w2 && b2 && w7 && b7 && dy && dx && dw1 && db1 && dw2 && db2 && dw7 && db7, SWEM_E_ARG,
               "cbam_bwd: null pointer");
  SWEM_REQUIRE(C % 4 == 0 && C <= 4096 && hid > 0 && hid <= 256, SWEM_E_SHAPE, "cbam_bwd: unsupported C/hid");
  SWEM_REQUIRE(ws && ws_bytes >= swem_cbam_bwd_workspace(B, H, W, C), SWEM_E_WORKSPACE, "cbam_bwd: workspace too small");
  const int P = H * W;
  float *part = static_cast<float *>(ws);
  float *comp = part + (size_t)B * CBAM_CHUNKS * 2 * C;
  float *sg = comp + (size_t)B * P * 2;
  float *da = sg + (size_t)B * P;
  float *dcomp = da + (size_t)B * P;
  float *cscale = dcomp + (size_t)B * P * 2;
  float *avg = cscale + (size_t)B * C, *mx = avg + (size_t)B * C, *dg = mx + (size_t)B * C;
  float *davg = dg + (size_t)B * C, *dmx = davg + (size_t)B * C;
  int *amax = reinterpret_cast<int *>(dmx + (size_t)B * C);
  float *du = reinterpret_cast<float *>(amax + (size_t)B * C);
  float *dxp = du + (size_t)B * P * C;
  // forward intermediates again (cheap next to keeping them alive across the whole clip)
  launch_cbam_pool(ST, x, part, B, P, C);
  hipLaunchKernelGGL(cbam_mlp_kernel, dim3(B), dim3(256), (2 * C + 2 * hid) * sizeof(float), ST, part, w1, b1, w2, b2,
                     cscale, P, C, hid);
  hipLaunchKernelGGL(cbam_stats_kernel, dim3(cdiv(C, 256), B), dim3(256), 0, ST, part, avg, mx, P, C);
  hipLaunchKernelGGL(cbam_argmax_pix_kernel, dim3(cdiv(C, 64), B), dim3(256), 0, ST, x, mx, amax, P, C);
  hipLaunchKernelGGL(cbam_spatial_pool_kernel, grid1((long long)B * P * 64), dim3(256), 0, ST, x, cscale, comp, B, P,
                     C);
  hipLaunchKernelGGL(cbam_sgate_kernel, grid1((long long)B * P), dim3(256), 0, ST, comp, w7, b7, sg, B, H, W);
  SWEM_CHECK_LAUNCH("cbam_bwd (forward recompute)");
  hipLaunchKernelGGL(cbam_bwd_pix1_kernel, grid1((long long)B * P * 64), dim3(256), 0, ST, x, dy, cscale, sg, da, B, P,
                     C);
  hipLaunchKernelGGL(cbam_bwd_sconv_kernel, grid1((long long)B * P), dim3(256), 0, ST, da, w7, dcomp, B, H, W);
  hipLaunchKernelGGL(cbam_bwd_w7_kernel, dim3(99), dim3(256), 0, ST, da, comp, dw7, db7, B, H, W);
  hipLaunchKernelGGL(cbam_bwd_pix2_kernel, grid1((long long)B * P * 64), dim3(256), 0, ST, x, dy, cscale, sg, comp,
                     dcomp, du, dxp, B, P, C);
  hipLaunchKernelGGL(cbam_bwd_dg_kernel, dim3(cdiv(C, 64), B), dim3(256), 0, ST, du, x, dg, P, C);
  SWEM_REQUIRE((size_t)(9 * C + 4 * hid) * sizeof(float) <= 64 * 1024, SWEM_E_SHAPE, "cbam_bwd: C = %d, hid = %d need more than 64 KB of LDS", C, hid);
  hipLaunchKernelGGL(cbam_bwd_mlp_kernel, dim3(1), dim3(1024), (9 * C + 4 * hid) * sizeof(float), ST, avg, mx, cscale, dg, w1,
                     b1, w2, dw1, db1, dw2, db2, davg, dmx, B, C, hid);
  hipLaunchKernelGGL(cbam_bwd_pix3_kernel, grid1((long long)B * P * (C / 4)), dim3(256), 0, ST, dxp, davg, dmx, amax, dx,
                     B, P, C);
  SWEM_CHECK_LAUNCH("cbam_bwd");
  return SWEM_OK;
}

extern "C" int swem_pred_head_f32(void *stream, const float *x, const float *w, const float *bias, float *logit,
                                  int B, int H, int W, int C) {
  SWEM_REQUIRE(x && w && bias && logit && C % 4 == 0, SWEM_E_SHAPE, "pred_head: bad argument");
  if (C == 256) {   // the decoder's head (networks.py:206): the matrix-core form
    const int tiles_x = (W + PH_TW - 1) / PH_TW, tiles_y = (H + PH_TH - 1) / PH_TH;
    hipLaunchKernelGGL(pred_head_mfma_kernel<16>, dim3((unsigned)(tiles_x * tiles_y * B)), dim3(256), 0, ST, x, w, bias, logit,
                       B, H, W, tiles_x, tiles_y);
  } else {
    hipLaunchKernelGGL(pred_head_kernel, grid1((long long)B * H * W * 64), dim3(256), 0, ST, x, w, bias, logit, B, H, W, C);
  }
  SWEM_CHECK_LAUNCH("pred_head");
  return SWEM_OK;
}

extern "C" int swem_decode_head_f32(void *stream, const float *logit4, const float *valid, float *logits, float *prob,
                                    long long *argmax, int B, int N, int h4, int w4, int Ho, int Wo) {
  SWEM_REQUIRE(logit4 && logits && prob && N > 0, SWEM_E_ARG, "decode_head: bad argument");
  hipLaunchKernelGGL(decode_head_kernel, grid1((long long)B * Ho * Wo), dim3(256), 0, ST, logit4, valid, logits, prob,
                     argmax, B, N, h4, w4, Ho, Wo);
  SWEM_CHECK_LAUNCH("decode_head");
  return SWEM_OK;
}

extern "C" int swem_argmax_onehot_i64(void *stream, const float *prob, long long *argmax, long long *onehot, int B,
                                      int N1, long long HW) {
  SWEM_REQUIRE(prob && N1 > 0, SWEM_E_ARG, "argmax_onehot: bad argument");
  hipLaunchKernelGGL(argmax_onehot_kernel, grid1((long long)B * HW), dim3(256), 0, ST, prob, argmax, onehot, B, N1,
                     HW);
  SWEM_CHECK_LAUNCH("argmax_onehot");
  return SWEM_OK;
}

extern "C" int swem_concat2_nhwc_f32(void *stream, const float *x0, int c0, long long bs0, const float *x1, int c1,
                                     long long bs1, float *y, int B, long long P) {
  SWEM_REQUIRE(x0 && x1 && y && c0 % 4 == 0 && c1 % 4 == 0 && c0 > 0 && c1 > 0, SWEM_E_SHAPE,
               "concat2: channel counts must be positive multiples of 4");
  hipLaunchKernelGGL(concat2_kernel, grid1((long long)B * P * ((c0 + c1) / 4)), dim3(256), 0, ST, x0, c0, bs0, x1, c1,
                     bs1, y, B, P);
  SWEM_CHECK_LAUNCH("concat2");
  return SWEM_OK;
}

extern "C" int swem_lincomb_f32(void *stream, const float *a, float alpha, const float *b, float beta, float *y,
                                long long n) {
  SWEM_REQUIRE(a && y && n >= 0, SWEM_E_ARG, "lincomb: bad argument");
  hipLaunchKernelGGL(lincomb_kernel, grid1(n), dim3(256), 0, ST, a, alpha, b, beta, y, n);
  SWEM_CHECK_LAUNCH("lincomb");
  return SWEM_OK;
}

extern "C" int swem_inject_objects_f32(void *stream, const float *prob, const float *new_masks, float *out, int B,
                                       int N1, int Nn1, long long HW) {
  SWEM_REQUIRE(prob && new_masks && out && N1 >= 1 && Nn1 >= 2, SWEM_E_ARG, "inject_objects: bad argument");
  hipLaunchKernelGGL(inject_objects_kernel, grid1((long long)B * HW), dim3(256), 0, ST, prob, new_masks, out, B, N1, Nn1,
                     HW);
  SWEM_CHECK_LAUNCH("inject_objects");
  return SWEM_OK;
}

extern "C" int swem_pack_u8_i64(void *stream, const long long *x, unsigned char *y, long long n) {
  SWEM_REQUIRE(x && y && n > 0 && ((uintptr_t)y % 16) == 0, SWEM_E_ARG, "pack_u8: bad argument (y must be 16-byte aligned)");
  hipLaunchKernelGGL(pack_u8_kernel, grid1((n + 15) / 16), dim3(256), 0, ST, x, y, n);
  SWEM_CHECK_LAUNCH("pack_u8");
  return SWEM_OK;
}

extern "C" int swem_transpose_f32(void *stream, const float *in, float *out, int batch, int R, int Cc, int ld) {
  SWEM_REQUIRE(in && out && ld >= R, SWEM_E_SHAPE, "transpose: ld < R");
  dim3 grid(cdiv(Cc, 32), cdiv(ld, 32), batch);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, ST, in, out, R, Cc, ld);
  SWEM_CHECK_LAUNCH("transpose");
  return SWEM_OK;
}
