// Does HBM throughput depend on the bytes each lane requests per load instruction?  Grid-stride streams of 3 reads : 1 write (the
// fused-backward pattern) and 1 : 1 with 4-, 8- and 16-byte accesses per lane.  (Round 3: the fp16 gradient chain halved the bytes of two
// of the four streams of the <= 32-channel backward kernels with 8-byte instead of 16-byte lane accesses and bought 3 %.)
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/lane_width.hip -o /tmp/lane_width && /tmp/lane_width
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int NR>
__global__ __launch_bounds__(256) void stream(const T* a, const T* b, const T* c, T* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    T v = a[i];
    if (NR > 1) v += b[i];
    if (NR > 2) v += c[i];
    y[i] = v;
  }
}
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <typename T, int NR>
static void run(const char* name, void* a, void* b, void* c, void* y, size_t bytes, int wgs) {
  const size_t n = bytes / sizeof(T);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream<T, NR>), dim3(wgs), dim3(256), 0, 0, (const T*)a, (const T*)b, (const T*)c, (T*)y, n);
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((stream<T, NR>), dim3(wgs), dim3(256), 0, 0, (const T*)a, (const T*)b, (const T*)c, (T*)y, n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s %4d workgroups: %6.2f TB/s\n", name, wgs, (NR + 1) * (double)bytes * 10 / (ms * 1e-3) / 1e12);
}
int main() {
  const size_t bytes = (size_t)1 << 30;
  void *a, *b, *c, *y;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&y, bytes);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(c, 0, bytes);
  for (int wgs : {2048, 8192}) {
    run<float, 1>("1R:1W  4 B per lane", a, b, c, y, bytes, wgs);
    run<f2, 1>("1R:1W  8 B per lane", a, b, c, y, bytes, wgs);
    run<f4, 1>("1R:1W 16 B per lane", a, b, c, y, bytes, wgs);
    run<float, 3>("3R:1W  4 B per lane", a, b, c, y, bytes, wgs);
    run<f2, 3>("3R:1W  8 B per lane", a, b, c, y, bytes, wgs);
    run<f4, 3>("3R:1W 16 B per lane", a, b, c, y, bytes, wgs);
  }
  return 0;
}
