// gfx950 (MI355X): packed-fp32 VALU instructions whose LOW lane reads the HIGH half of a source pair (op_sel) return wrong LOW-lane results
// while ANOTHER wave on the same CU executes v_mfma_f32_16x16x32_bf16.  Stand-alone reproducer of the round-2 nondeterminism of
// enc_first_bwd_kernel: two sums of that kernel were the LOW lanes of `v_pk_fma_f32 vD, vA, vB, vD op_sel:[0,1,0]` and
// differed from launch to launch only when the split-precision (bf16 MFMA) kernels of the other encoder streams shared its CUs.
//   hipcc --offload-arch=gfx950 -O2 -w tools/pk_fma_opsel_repro.hip -o /tmp/pk_repro && /tmp/pk_repro
// "victim" kernels issue ONE packed instruction form in a loop and check both lanes against scalar v_fma/v_mul/v_add of the selected
// halves (bit-exact by construction); "hog" kernels of one instruction class each run beside them on a second stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// OP: 0 fma (d = a*b + c), 1 mul (d = a*b), 2 add (d = a + b).  S0,S1,S2 / H0,H1,H2: half of each source the LOW / HIGH lane reads
// (= the instruction's op_sel / op_sel_hi bits; the assembler string MODS must say the same).
#define VICTIM(NAME, OP, MODS, S0, S1, S2, H0, H1, H2)                                                                              \
  __global__ __launch_bounds__(256) void NAME(const float* __restrict__ x, unsigned* bad, int n, int iters) {                      \
    __shared__ float xs[1026];                                                                                                      \
    for (int i = threadIdx.x; i < 1026; i += 256) xs[i] = x[(blockIdx.x * 1031u + i) % n];                                          \
    __syncthreads();                                                                                                                \
    unsigned nl = 0, nh = 0;                                                                                                        \
    f2 c = {0.25f, -0.5f};                                                                                                          \
    for (int it = 0; it < iters; ++it) {                                                                                            \
      const int p = (threadIdx.x >> 2) + 64 * (it & 15);                                                                            \
      const f2 a = {xs[p + 1], xs[p + 2]};                                                                                          \
      const f2 b = (f2){xs[p] * 0.37f, xs[p] * 1.7f} * (f2){0.9f + 1e-3f * it, 1.1f};                                               \
      f2 d = c;                                                                                                                     \
      float rl = S2 ? c.y : c.x, rh = H2 ? c.y : c.x;                                                                               \
      if (OP == 0) {                                                                                                                \
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 " MODS : "+v"(d) : "v"(a), "v"(b));                                               \
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(rl) : "v"(S0 ? a.y : a.x), "v"(S1 ? b.y : b.x));                             \
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(rh) : "v"(H0 ? a.y : a.x), "v"(H1 ? b.y : b.x));                             \
      } else if (OP == 1) {                                                                                                         \
        asm volatile("v_pk_mul_f32 %0, %1, %2 " MODS : "=v"(d) : "v"(a), "v"(b));                                                   \
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(rl) : "v"(S0 ? a.y : a.x), "v"(S1 ? b.y : b.x));                                 \
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(rh) : "v"(H0 ? a.y : a.x), "v"(H1 ? b.y : b.x));                                 \
      } else {                                                                                                                      \
        asm volatile("v_pk_add_f32 %0, %1, %2 " MODS : "=v"(d) : "v"(a), "v"(b));                                                   \
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(rl) : "v"(S0 ? a.y : a.x), "v"(S1 ? b.y : b.x));                                 \
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(rh) : "v"(H0 ? a.y : a.x), "v"(H1 ? b.y : b.x));                                 \
      }                                                                                                                             \
      nl += __float_as_uint(d.x) != __float_as_uint(rl);                                                                            \
      nh += __float_as_uint(d.y) != __float_as_uint(rh);                                                                            \
      c = (f2){rl * 0.5f, rh * 0.5f};   /* continue from the reference: one wrong result is counted once */                        \
    }                                                                                                                               \
    if (nl) atomicAdd(bad, nl);                                                                                                     \
    if (nh) atomicAdd(bad + 1, nh);                                                                                                 \
  }
VICTIM(v_fma_plain, 0, "", 0, 0, 0, 1, 1, 1)
VICTIM(v_fma_s1_hi, 0, "op_sel:[0,1,0]", 0, 1, 0, 1, 1, 1)                       // the form in enc_first_bwd_kernel: both lanes read b.hi
VICTIM(v_fma_s0_hi, 0, "op_sel:[1,0,0]", 1, 0, 0, 1, 1, 1)
VICTIM(v_fma_s2_swap, 0, "op_sel:[0,0,1] op_sel_hi:[1,1,0]", 0, 0, 1, 1, 1, 0)
VICTIM(v_fma_s1_lo, 0, "op_sel_hi:[1,0,1]", 0, 0, 0, 1, 0, 1)                    // both lanes read b.lo (HIGH lane reads a LOW half)
VICTIM(v_mul_s0_swap, 1, "op_sel:[1,0] op_sel_hi:[0,0]", 1, 0, 0, 0, 0, 0)       // the commonest form in this library (580 x)
VICTIM(v_mul_s1_hi, 1, "op_sel:[0,1]", 0, 1, 0, 1, 1, 0)
VICTIM(v_add_s1_hi, 2, "op_sel:[0,1]", 0, 1, 0, 1, 1, 0)

template <int KIND>   // 0: fp32 MFMA 16x16x4; 1: bf16 MFMA 16x16x32 (new on gfx950); 2: ds_read_b128 traffic; 3: packed fp32 VALU
__global__ __launch_bounds__(256) void hog(float* out, int iters) {
  __shared__ f4 sm[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = (f4){1.f * i, 2.f, 3.f, 4.f};
  __syncthreads();
  f4 acc = {0, 0, 0, 0};
  const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  const bf8 ab = {(__bf16)a, (__bf16)b, (__bf16)1.f, (__bf16)2.f, (__bf16)a, (__bf16)b, (__bf16)3.f, (__bf16)4.f};
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a + i, b, acc, 0, 0, 0);
    if (KIND == 1) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, acc, 0, 0, 0);
    if (KIND == 2) acc += sm[(threadIdx.x + i * 17) & 1023];
    if (KIND == 3) acc = acc * (f4){1.0001f, 0.9999f, 1.0002f, 0.9998f} + (f4){a, b, a, b};
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
  const int n = 1 << 24;
  float *x, *hog_out;
  unsigned *bad, h[2];
  hipMalloc(&x, n * 4); hipMalloc(&hog_out, 4096 * 256 * 4); hipMalloc(&bad, 8);
  float* hx = new float[n];
  for (int i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) >> 8) / 16777216.f - 0.5f;
  hipMemcpy(x, hx, n * 4, hipMemcpyHostToDevice);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  typedef void (*vk)(const float*, unsigned*, int, int);
  const vk victims[8] = {v_fma_plain, v_fma_s1_hi, v_fma_s0_hi, v_fma_s2_swap, v_fma_s1_lo, v_mul_s0_swap, v_mul_s1_hi, v_add_s1_hi};
  const char* vn[8] = {"v_pk_fma_f32 (no op_sel)", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0]",
                       "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,0]", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[0,1]"};
  const char* hn[5] = {"alone", "fp32 MFMA 16x16x4", "bf16 MFMA 16x16x32", "ds_read_b128", "packed fp32 VALU"};
  for (int v = 0; v < 8; ++v)
    for (int m = 0; m < 5; ++m) {
      hipMemset(bad, 0, 8);
      for (int rep = 0; rep < 20; ++rep) {
        if (m == 1) hipLaunchKernelGGL(hog<0>, dim3(1024), dim3(256), 0, s2, hog_out, 20000);
        if (m == 2) hipLaunchKernelGGL(hog<1>, dim3(1024), dim3(256), 0, s2, hog_out, 20000);
        if (m == 3) hipLaunchKernelGGL(hog<2>, dim3(1024), dim3(256), 0, s2, hog_out, 20000);
        if (m == 4) hipLaunchKernelGGL(hog<3>, dim3(1024), dim3(256), 0, s2, hog_out, 20000);
        hipLaunchKernelGGL(victims[v], dim3(4096), dim3(256), 0, s1, x, bad, n, 2048);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
      printf("%-48s beside %-20s: wrong LOW-lane results %10u, wrong HIGH-lane results %10u  (of 4.3e10 each)\n", vn[v], hn[m], h[0], h[1]);
    }
  return 0;
}
