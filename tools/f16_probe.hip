// Probe for the f16x3 conv arithmetic (hipcc --offload-arch=gfx950 -O3 tools/f16_probe.hip -o /tmp/f16_probe):
//  1. does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL inputs (the `mid` plane of values below 0.25 is subnormal)?
//  2. does v_cvt_f16_f32 produce them (MODE.fp_denorm for 16-bit)?
//  3. sustained rate and clock of the f16 against the bf16 MFMA (same cycles per instruction; the clock the chip holds may differ).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void denorm_kernel(float *out, float a_val, float b_val) {
  // every A element = fp16(a_val), every B element = fp16(b_val): C[i][j] = 32 * a * b
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}

template <int MODE>   // 0 bf16 16x16x32, 1 f16 16x16x32, 2 bf16 32x32x16, 3 f16 32x32x16
__global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, unsigned long long *clk) {
  f32x4 acc4[8];
  f32x16 acc16[4];
  for (int i = 0; i < 8; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc16[i][e] = 0.f;
  f16x8 ha, hb;
  bf16x8 ba, bb;
  for (int i = 0; i < 8; ++i) {
    ha[i] = (_Float16)(threadIdx.x * 0.001f + 1.0f + i); hb[i] = (_Float16)(0.5f - threadIdx.x * 0.002f);
    ba[i] = (__bf16)(threadIdx.x * 0.001f + 1.0f + i); bb[i] = (__bf16)(0.5f - threadIdx.x * 0.002f);
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc4[i], 0, 0, 0);
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4[i], 0, 0, 0);
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc16[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc16[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc16[i], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc4[i][0] + acc4[i][3];
  for (int i = 0; i < 4; ++i) s += acc16[i][0] + acc16[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
void rate(const char *name, int blocks, int iters) {
  float *out; unsigned long long *clk, h[2];
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double per = MODE < 2 ? 2.0 * 16 * 16 * 32 * 8 : 2.0 * 32 * 32 * 16 * 4;
    printf("%-16s blocks %4d iters %d: %.3f ms  %.0f TFLOP/s  clock %.0f MHz\n", name, blocks, iters, ms,
           per * iters * blocks * 4 / ms / 1e9, (double)h[0] / (double)h[1] * 100.0);
  }
  hipFree(out); hipFree(clk);
}

int main() {
  float *out, h[3];
  hipMalloc(&out, 12);
  const float cases[][2] = {{1.0f, 1.0f}, {1e-6f, 1024.f}, {5.96046448e-8f, 1.0f}, {3e-5f, 3e-5f}, {6.0e-5f, 2.0f}};
  for (auto &c : cases) {
    hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, out, c[0], c[1]);
    hipMemcpy(h, out, 12, hipMemcpyDeviceToHost);
    printf("a = %.9g (fp16 -> %.9g)  b = %.9g (fp16 -> %.9g):  mfma = %.9g   expected 32ab = %.9g\n", c[0], h[1], c[1], h[2], h[0],
           32.0 * (double)h[1] * (double)h[2]);
  }
  rate<0>("bf16 16x16x32", 1024, 40000);
  rate<1>("f16  16x16x32", 1024, 40000);
  rate<2>("bf16 32x32x16", 1024, 40000);
  rate<3>("f16  32x32x16", 1024, 40000);
  return 0;
}
