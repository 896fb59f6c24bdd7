// What does a TUNED plain stream reach on this box?  (`lane_width.hip` is one load per lane per trip, no loads in
// flight, non-persistent grids -- not evidence about the ceiling.)  Persistent grid = 256 CUs x k workgroups, U independent 16-B loads in
// flight per lane per trip (U = 4 / 8), 1R:1W and 3R:1W, >= 2 GiB moved per launch, grid-stride and BLOCKED (each workgroup streams one
// contiguous region: the tile assignment the per-workgroup statistics rows of round 4 need) chunk orders, plain and non-temporal accesses.
//   hipcc --offload-arch=gfx950 -O3 -w tools/ubench/stream_tuned.hip -o /tmp/stream_tuned && /tmp/stream_tuned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NR, int U, bool NT, bool BLOCKED>
__global__ __launch_bounds__(256) void stream(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c, f4* __restrict__ y, size_t nchunk) {
  // chunk = 256 lanes x U x 16 B; lane l's j-th access of a chunk is element j*256 + l: every wave instruction is one contiguous 1 KB run
  size_t first, step, last;
  if (BLOCKED) {
    const size_t per = (nchunk + gridDim.x - 1) / gridDim.x;
    first = (size_t)blockIdx.x * per; last = first + per < nchunk ? first + per : nchunk; step = 1;
  } else {
    first = blockIdx.x; last = nchunk; step = gridDim.x;
  }
  for (size_t ch = first; ch < last; ch += step) {
    const size_t base = ch * (256 * U) + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) v[j] = NT ? __builtin_nontemporal_load(a + base + j * 256) : a[base + j * 256];
    if (NR > 1) {
      f4 w[U];
#pragma unroll
      for (int j = 0; j < U; ++j) w[j] = NT ? __builtin_nontemporal_load(b + base + j * 256) : b[base + j * 256];
#pragma unroll
      for (int j = 0; j < U; ++j) v[j] += w[j];
    }
    if (NR > 2) {
      f4 w[U];
#pragma unroll
      for (int j = 0; j < U; ++j) w[j] = NT ? __builtin_nontemporal_load(c + base + j * 256) : c[base + j * 256];
#pragma unroll
      for (int j = 0; j < U; ++j) v[j] += w[j];
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      if (NT) __builtin_nontemporal_store(v[j], y + base + j * 256); else y[base + j * 256] = v[j];
    }
  }
}

template <int NR, int U, bool NT, bool BLOCKED>
static double run(void* a, void* b, void* c, void* y, size_t bytes, int wgs) {
  const size_t nchunk = bytes / (256 * U * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream<NR, U, NT, BLOCKED>), dim3(wgs), dim3(256), 0, 0, (const f4*)a, (const f4*)b, (const f4*)c, (f4*)y, nchunk);
  hipEventRecord(e0);
  const int reps = 8;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((stream<NR, U, NT, BLOCKED>), dim3(wgs), dim3(256), 0, 0, (const f4*)a, (const f4*)b, (const f4*)c, (f4*)y, nchunk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return (NR + 1) * (double)bytes * reps / (ms * 1e-3) / 1e12;
}

template <int NR, int U>
static void sweep(const char* name, void* a, void* b, void* c, void* y, size_t bytes) {
  for (int k : {1, 2, 4, 8}) {
    const int wgs = 256 * k;
    printf("%s U=%d  %4d wgs (k=%d):  grid-stride %5.2f  blocked %5.2f  grid-stride nt %5.2f  blocked nt %5.2f  TB/s\n", name, U, wgs, k,
           run<NR, U, false, false>(a, b, c, y, bytes, wgs), run<NR, U, false, true>(a, b, c, y, bytes, wgs),
           run<NR, U, true, false>(a, b, c, y, bytes, wgs), run<NR, U, true, true>(a, b, c, y, bytes, wgs));
    fflush(stdout);
  }
}

int main() {
  const size_t bytes = (size_t)1 << 30;   // per buffer: 1R:1W moves 2 GiB per launch, 3R:1W 4 GiB
  void *a, *b, *c, *y;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&y, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 1, bytes); hipMemset(c, 1, bytes);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s, %d CUs, buffers of %zu MiB\n", p.name, p.multiProcessorCount, bytes >> 20);
  sweep<1, 4>("1R:1W", a, b, c, y, bytes);
  sweep<1, 8>("1R:1W", a, b, c, y, bytes);
  sweep<3, 4>("3R:1W", a, b, c, y, bytes);
  sweep<3, 8>("3R:1W", a, b, c, y, bytes);
  return 0;
}
