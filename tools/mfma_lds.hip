// Calibration 2: MFMA fed from LDS exactly like the conv inner loop (4 ds_read_b128 per 16 MFMAs, 2x2 tiles per wave).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 m4(float4 a, float4 b, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
  return c;
}
template <bool PREFETCH>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  extern __shared__ float4 sm[];  // [8][129] x2 (A,B)
  const int SA = 129;
  for (int i = threadIdx.x; i < 2 * 8 * SA; i += 256) sm[i] = make_float4(i * 1e-3f, 1.f, 0.5f, -i * 1e-3f);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const float4 *Ab = sm + (wave >> 1) * 64 + r + h * SA, *Bb = sm + 8 * SA + (wave & 1) * 64 + r + h * SA;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (PREFETCH) {
      float4 a[2][2], b[2][2];
      a[0][0] = Ab[0]; a[0][1] = Ab[32]; b[0][0] = Bb[0]; b[0][1] = Bb[32];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < 3) { a[(j+1)&1][0] = Ab[2*(j+1)*SA]; a[(j+1)&1][1] = Ab[2*(j+1)*SA+32]; b[(j+1)&1][0] = Bb[2*(j+1)*SA]; b[(j+1)&1][1] = Bb[2*(j+1)*SA+32]; }
        acc[0][0] = m4(a[j&1][0], b[j&1][0], acc[0][0]); acc[0][1] = m4(a[j&1][0], b[j&1][1], acc[0][1]);
        acc[1][0] = m4(a[j&1][1], b[j&1][0], acc[1][0]); acc[1][1] = m4(a[j&1][1], b[j&1][1], acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 a0 = Ab[2*j*SA], a1 = Ab[2*j*SA+32], b0 = Bb[2*j*SA], b1 = Bb[2*j*SA+32];
        acc[0][0] = m4(a0, b0, acc[0][0]); acc[0][1] = m4(a0, b1, acc[0][1]);
        acc[1][0] = m4(a1, b0, acc[1][0]); acc[1][1] = m4(a1, b1, acc[1][1]);
      }
    }
    asm volatile("" ::: "memory");
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <bool PF>
void run(int blocks, int iters, size_t lds) {
  float *out; hipMalloc(&out, blocks * 256 * 4);
  hipFuncSetAttribute((const void*)k<PF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PF>, dim3(blocks), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * 32 * 32 * 2 * 64.0 * iters * blocks * 4;
    printf("prefetch %d blocks %4d lds %zu iters %d: %.3f ms  %.1f TFLOP/s\n", (int)PF, blocks, lds, iters, ms, fl / ms / 1e9);
  }
}
int main() {
  run<false>(256, 2000, 150000);  // 1 block/CU
  run<false>(512, 2000, 66048);   // 2 blocks/CU
  run<false>(1024, 1000, 33024);  // 4 blocks/CU
  run<true>(256, 2000, 150000);
  run<true>(512, 2000, 66048);
  run<true>(1024, 1000, 33024);
  return 0;
}
