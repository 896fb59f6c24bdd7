// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 in the bf16x3 pattern (16 accumulators, 3 products each),
// operands in registers (variant 0) or re-read from LDS every step (variant 1).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int LDSREAD>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ bf16x8 sm[2048];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += 256) { bf16x8 v; for (int e = 0; e < 8; ++e) v[e] = (__bf16)(0.001f * (i + e)); sm[i] = v; }
  __syncthreads();
  f32x4 acc[4][4];
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
  bf16x8 ah[4], al[4], bh[4], bl[4];
  for (int n = 0; n < 4; ++n) { ah[n] = sm[tid + n * 256]; al[n] = sm[(tid + n * 256 + 7) & 2047]; bh[n] = sm[(tid * 3 + n) & 2047]; bl[n] = sm[(tid * 5 + n) & 2047]; }
  for (int it = 0; it < iters; ++it) {
    if (LDSREAD) {
      for (int n = 0; n < 4; ++n) { bh[n] = sm[(tid + it * 64 + n * 16) & 2047]; bl[n] = sm[(tid + it * 64 + n * 16 + 1024) & 2047]; }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[n], bh[m], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[n], bl[m], acc[m][n], 0, 0, 0);
        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[n], bh[m], acc[m][n], 0, 0, 0);
      }
  }
  f32x4 s = {0, 0, 0, 0};
  for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n];
  out[blockIdx.x * 256 + tid] = s.x + s.y + s.z + s.w;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs : {256, 512, 1024, 2048}) for (int v = 0; v < 2; ++v) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (v) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, out, iters); else hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)wgs * 4 * iters * 48;
    printf("wgs %4d ldsread %d: %.3f ms  %.1f TFLOP/s dense-bf16 (%.1f effective bf16x3)  cycles/MFMA/SIMD@2.4GHz %.1f\n", wgs, v, ms, mf * 16384 / ms / 1e9,
           mf * 16384 / 3 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / 1024));
  }
  return 0;
}
