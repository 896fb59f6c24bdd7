// Reproducer attempt for the round-2 nondeterminism of enc_first_bwd_kernel: the two sums that differed from launch to
// launch were the LOW lanes of `v_pk_fma_f32 vD, vA, vB, vD op_sel:[0,1,0]` (both lanes multiply by the HIGH half of src1, src0 straight from a
// ds_read2_b32), only when other kernels shared the CU.  This program issues exactly that instruction in a loop, checks each lane against a
// scalar v_fma_f32 of the same operands (bit-exact in exact arithmetic), alone and with a matrix-core / memory hog on a second stream.
//   hipcc --offload-arch=gfx950 -O2 tools/pk_fma_opsel_repro.hip -o /tmp/pk_repro && /tmp/pk_repro
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void victim(const float* __restrict__ x, unsigned* bad, int n, int iters) {
  __shared__ float xs[1026];
  for (int i = threadIdx.x; i < 1026; i += 256) xs[i] = x[(blockIdx.x * 1031u + i) % n];
  __syncthreads();
  f2 acc = {0.f, 0.f};
  float ref_lo = 0.f, ref_hi = 0.f;
  for (int it = 0; it < iters; ++it) {
    const int p = (threadIdx.x >> 2) + 64 * (it & 15);
    const f2 xcxp = {xs[p + 1], xs[p + 2]};                                   // ds_read2_b32, as in the kernel
    const f2 r = {xs[p] * 0.37f, xs[p] * 1.7f};
    const f2 gy = r * (f2){0.9f + 1e-3f * it, 1.1f};                            // a v_pk_mul_f32 right before the use, as in the kernel
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(xcxp), "v"(gy));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref_lo) : "v"(xcxp.x), "v"(gy.y));   // (plain C here is re-vectorised into the same v_pk_fma)
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ref_hi) : "v"(xcxp.y), "v"(gy.y));
  }
  if (__float_as_uint(acc.x) != __float_as_uint(ref_lo)) atomicAdd(bad, 1u);
  if (__float_as_uint(acc.y) != __float_as_uint(ref_hi)) atomicAdd(bad + 1, 1u);
}

__global__ __launch_bounds__(256) void mfma_hog(float* out, int iters) {   // keeps the matrix pipe and some VALU of every CU busy
  f4 acc = {0, 0, 0, 0};
  const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a + i, b, acc, 0, 0, 0);
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  const int n = 1 << 24;
  float *x, *hog_out, *big_a, *big_b;
  unsigned *bad, h[2];
  hipMalloc(&x, n * 4); hipMalloc(&hog_out, 4096 * 256 * 4); hipMalloc(&bad, 8); hipMalloc(&big_a, 1u << 30); hipMalloc(&big_b, 1u << 30);
  float* hx = new float[n];
  for (int i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 8) / 16777216.f - 0.5f;
  hipMemcpy(x, hx, n * 4, hipMemcpyHostToDevice);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  const char* modes[3] = {"alone", "with an MFMA hog on a second stream", "with a 1 GiB device copy on a second stream"};
  for (int m = 0; m < 3; ++m) {
    hipMemset(bad, 0, 8);
    for (int rep = 0; rep < 100; ++rep) {
      if (m == 1) hipLaunchKernelGGL(mfma_hog, dim3(1024), dim3(256), 0, s2, hog_out, 40000);
      if (m == 2) hipMemcpyAsync(big_b, big_a, 1u << 30, hipMemcpyDeviceToDevice, s2);
      hipLaunchKernelGGL(victim, dim3(4096), dim3(256), 0, s1, x, bad, n, 4096);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    printf("%-46s: threads whose LOW lane differs from the scalar fma: %u, HIGH lane: %u (100 launches x 1M threads)\n", modes[m], h[0], h[1]);
  }
  return 0;
}
