// Shared device helpers for the wav2sleep gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/w2s.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Element index of A[row][k] (row = output channel of the GEMM, k in [0, K), K % 32 == 0, rows % 16 == 0) in the
// fragment-major bf16 weight planes of the split-precision conv: [row/16][k/32][lane = ((k%32)/8)*16 + row%16][k%8], i.e. the
// 16 B a lane feeds to v_mfma_f32_16x16x32_bf16 are contiguous and a wave's fetch of one fragment is one 1 KB run.
__host__ __device__ inline size_t w2s_frag_index(int row, int k, int K) {
  return (((size_t)(row >> 4) * (K >> 5) + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (row & 15)) * 8 + (k & 7);
}

#define W2S_CHECK_LAUNCH()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return W2S_ELAUNCH; \
  } while (0)

// D[16x16] += A[16x4] * B[4x16] on the fp32 matrix core.  Lane l supplies A[l&15][l>>4], B[l>>4][l&15];
// afterwards lane l holds D[4*(l>>4)+reg][l&15] (cdna_hip_programming.md section 3).
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Phi(x) - 1/2 = erf(x / sqrt 2) / 2 as a clamped rational minimax x*P(x^2)/Q(x^2): the Eigen/XLA float erf (|err| <= 4e-7 abs, no
// branches, one v_rcp) with the 1/sqrt 2 argument scaling folded into the coefficients (z^2 = x^2/2: exact powers of two) and the
// factor 1/2 into P -- GELU(x) = x*(1/2 + t) and GELU'(x) = 1/2 + t + x*pdf(x) then need no extra multiplies.  ocml's erff has 3x the
// VALU instructions, which made every conv VALU-bound (profiles/r01 PMC); the folding removes another ~25 % (r02: the >= 64-channel
// layers are bound by VALU issue, every wave instruction counts).
#define W2S_HE_P0 -1.5059950899190544e-12f
#define W2S_HE_P1 3.0611993495632817e-10f
#define W2S_HE_P2 -4.6426510635910745e-08f
#define W2S_HE_P3 -2.5157562504318776e-06f
#define W2S_HE_P4 -6.49646099191159e-05f
#define W2S_HE_P5 -0.0005223043845035136f
#define W2S_HE_P6 -0.00569080701097846f
#define W2S_HE_Q0 -9.103794695874967e-07f
#define W2S_HE_Q1 -2.6671756131690927e-05f
#define W2S_HE_Q2 -0.0004207067540846765f
#define W2S_HE_Q3 -0.0036866646260023117f
#define W2S_HE_Q4 -0.014264739118516445f
#define W2S_HE_CLAMP 5.65685424949238f   /* 4 sqrt 2: erf saturates to fp32 beyond */
__device__ __forceinline__ float half_erf_fast(float x) {
  x = __builtin_amdgcn_fmed3f(x, -W2S_HE_CLAMP, W2S_HE_CLAMP);
  const float x2 = x * x;
  float p = fmaf(x2, W2S_HE_P0, W2S_HE_P1);
  p = fmaf(x2, p, W2S_HE_P2);
  p = fmaf(x2, p, W2S_HE_P3);
  p = fmaf(x2, p, W2S_HE_P4);
  p = fmaf(x2, p, W2S_HE_P5);
  p = fmaf(x2, p, W2S_HE_P6);
  float q = fmaf(x2, W2S_HE_Q0, W2S_HE_Q1);
  q = fmaf(x2, q, W2S_HE_Q2);
  q = fmaf(x2, q, W2S_HE_Q3);
  q = fmaf(x2, q, W2S_HE_Q4);
  return (x * p) * __builtin_amdgcn_rcpf(q);
}
// exact-form (erf) GELU and its derivative -- models/utils.py:61-74 nn.GELU(approximate='none')
__device__ __forceinline__ float gelu_f(float x) { return x * (0.5f + half_erf_fast(x)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f + half_erf_fast(x);
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);  // exp(-x^2/2)
  return cdf + x * pdf;
}
// 4-wide forms written on the vector type: every polynomial step is one float4 fma (two v_pk_fma_f32), and the four
// divisions of the rational share ONE v_rcp_f32 (quarter-rate) through r = 1/(q0 q1 q2 q3); |q| is in [0.014, 2.4], so
// the product cannot over/underflow.
__device__ __forceinline__ f32x4 splat4(float c) { return (f32x4){c, c, c, c}; }
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x4 half_erf4(f32x4 x) {
  x.x = __builtin_amdgcn_fmed3f(x.x, -W2S_HE_CLAMP, W2S_HE_CLAMP); x.y = __builtin_amdgcn_fmed3f(x.y, -W2S_HE_CLAMP, W2S_HE_CLAMP);
  x.z = __builtin_amdgcn_fmed3f(x.z, -W2S_HE_CLAMP, W2S_HE_CLAMP); x.w = __builtin_amdgcn_fmed3f(x.w, -W2S_HE_CLAMP, W2S_HE_CLAMP);
  const f32x4 x2 = x * x;
  f32x4 p = fma4(x2, splat4(W2S_HE_P0), splat4(W2S_HE_P1));
  p = fma4(x2, p, splat4(W2S_HE_P2));
  p = fma4(x2, p, splat4(W2S_HE_P3));
  p = fma4(x2, p, splat4(W2S_HE_P4));
  p = fma4(x2, p, splat4(W2S_HE_P5));
  p = fma4(x2, p, splat4(W2S_HE_P6));
  f32x4 q = fma4(x2, splat4(W2S_HE_Q0), splat4(W2S_HE_Q1));
  q = fma4(x2, q, splat4(W2S_HE_Q2));
  q = fma4(x2, q, splat4(W2S_HE_Q3));
  q = fma4(x2, q, splat4(W2S_HE_Q4));
  // 1/q for the four lanes from one reciprocal
  const float q01 = q.x * q.y, q23 = q.z * q.w;
  const float r = __builtin_amdgcn_rcpf(q01 * q23);
  const float r01 = r * q23, r23 = r * q01;
  const f32x4 inv = {r01 * q.y, r01 * q.x, r23 * q.w, r23 * q.z};
  return (x * p) * inv;
}
__device__ __forceinline__ f32x4 gelu4(f32x4 v) { return v * (half_erf4(v) + 0.5f); }
__device__ __forceinline__ f32x4 gelu_grad4(f32x4 v) {
  const f32x4 cdf = half_erf4(v) + 0.5f;
  const f32x4 t = v * v * -0.72134752044448170368f;  // exp(-x^2/2) = 2^t
  const f32x4 pdf = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y), __builtin_amdgcn_exp2f(t.z), __builtin_amdgcn_exp2f(t.w)};
  return fma4(v * 0.39894228040143267794f, pdf, cdf);
}
// GELU and GELU' of the same argument from ONE erf evaluation (the fused backward needs both for the tile's centre rows)
__device__ __forceinline__ void gelu_both4(f32x4 v, f32x4& h, f32x4& gp) {
  const f32x4 cdf = half_erf4(v) + 0.5f;
  const f32x4 t = v * v * -0.72134752044448170368f;
  const f32x4 pdf = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y), __builtin_amdgcn_exp2f(t.z), __builtin_amdgcn_exp2f(t.w)};
  h = v * cdf;
  gp = fma4(v * 0.39894228040143267794f, pdf, cdf);
}
__device__ __forceinline__ float sanitize_f(float x) { return isinf(x) ? 0.0f : x; }

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// wave-uniform base + 32-bit BYTE offset per lane: selects the scalar-base addressing mode (no 64-bit VALU address arithmetic)
__device__ __forceinline__ f32x4 ld4o(const float* base, unsigned elem_off) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + (elem_off << 2));
}
__device__ __forceinline__ void st4o(float* base, unsigned elem_off, f32x4 v) {
  *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + (elem_off << 2)) = v;
}

// ---- fp16 storage of the encoder's gradient chain (W2S_GRAD_FP16) --------------------------------------------
// The gradient tensors between two fused-backward kernels of the <= 32-channel blocks (d/d xhat1, d/d xhat2, d/d pre: half of those
// kernels' bytes) are stored as fp16 with ONE power-of-two scale per tensor; everything is accumulated and applied in fp32.  A tensor
// carries a two-float header: hdr[0] = scale (stored = true * scale), hdr[1] = max |true value| as float bits, accumulated by its
// producer with an integer atomicMax (order-independent => runs stay bit-reproducible).  A producer derives its output scale from the
// maximum of the tensor(s) it READS (final when it starts): stored values then sit near 2^8 with 2^7 of headroom below fp16's 65504 (a
// layer's gain is bounded by rstd <= 10 times the weights' row sums) and keep all 11 significant bits down to 2^-22 of the maximum.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h16x4 ld4h(const void* base, unsigned elem_off) {
  return *reinterpret_cast<const h16x4*>(reinterpret_cast<const char*>(base) + (elem_off << 1));
}
__device__ __forceinline__ void st4h(void* base, unsigned elem_off, h16x4 v) {
  *reinterpret_cast<h16x4*>(reinterpret_cast<char*>(base) + (elem_off << 1)) = v;
}
__device__ __forceinline__ f32x4 h2f4(h16x4 h) { return __builtin_convertvector(h, f32x4); }
__device__ __forceinline__ h16x4 f2h4(f32x4 v) { return __builtin_convertvector(v, h16x4); }   // round to nearest even
__device__ __forceinline__ float w2s_gscale_for(float ref) {
  if (!(ref > 0.f) || !(ref < 3.0e38f)) return 1.f;
  int e;
  (void)frexpf(ref, &e);   // ref = m * 2^e, m in [0.5, 1)
  return ldexpf(1.f, min(9 - e, 120));
}
__device__ __forceinline__ float amax4(float a, f32x4 v) {
  return fmaxf(fmaxf(a, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
// end of a producer kernel: wave maximum, one integer atomic per wave; workgroup 0 publishes the scale
__device__ __forceinline__ void w2s_amax_commit(float* hdr, float amax, float scale) {
  for (int m = 1; m < 64; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(hdr + 1), __float_as_uint(amax));
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) hdr[0] = scale;
}

// counter-based RNG for dropout: splitmix64 of (seed, element index) -> uniform [0,1).  The same
// (seed, index) regenerates the same mask in the backward pass; nothing is stored.
__device__ __forceinline__ float w2s_uniform(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float w2s_dropscale(uint64_t seed, uint64_t idx, float p) {
  return (w2s_uniform(seed, idx) >= p) ? 1.0f / (1.0f - p) : 0.0f;
}

// ------------------------------------------------------------------------------------------------------------------
// Statistics finalisation inside the PRODUCER kernel (replaces a w2s_stats_finalize launch between every two encoder layers:
// 168 launches and ~1.3 ms per train step on the critical path).  Every workgroup that has written its per-tile partial
// sums for sample b takes a ticket from counter[b]; the workgroup that draws the last ticket re-reads all of that sample's
// partials and reduces them in a FIXED order in fp64 -- the result does not depend on which workgroup happens to be last, so
// runs stay bit-reproducible.  Release/acquire: partial stores -> __threadfence -> ticket (device-scope atomic); last
// workgroup: ticket -> __threadfence (L1 invalidate) -> loads.  The last workgroup also re-arms the counter.
// part: [B][ntiles][2][C];  kind 0: out = (mean, rstd = 1/sqrt(biased var + eps));  kind 1: out = (sum1/count, sum2/count).
// ------------------------------------------------------------------------------------------------------------------
// per-tile partial sums that another workgroup of the SAME launch may read (w2s_stat_finish): agent-scope relaxed accesses
__device__ __forceinline__ void w2s_part_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float w2s_part_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
struct StatFin { float* out; int* cnt; double inv_count; float eps; int kind; };
__device__ __forceinline__ void w2s_stat_finish(const StatFin& F, const float* part, int b, int ntiles, int C, int expected) {
  __shared__ double fin_red[256];
  __shared__ double fin_sum[256];
  __shared__ int fin_last;
  if (!F.out) return;  // uniform
  const int tid = threadIdx.x;
  // The partials were written with agent-scope write-through stores (w2s_part_store); waiting for their acknowledgement is all the
  // release this needs.  A __threadfence() here would write back / invalidate the whole XCD L2 once per tile (measured: 8x slower step).
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (tid == 0) fin_last = (__hip_atomic_fetch_add(&F.cnt[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expected - 1) ? 1 : 0;
  __syncthreads();
  if (!fin_last) return;
  // one partial row = 2C floats ([2][C]); thread = (4-float slot q of the row, tile lane ln); agent-scope loads (they bypass the
  // non-coherent L1 / remote-XCD L2 lines), four rows = 16 loads in flight per thread
  const int qn = (2 * C) >> 2, nl = 256 / qn;   // C = 16: 32 tile lanes ... C = 128: 4
  const int q = tid % qn, ln = tid / qn;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  {
    const float* p = part + ((size_t)b * ntiles) * (2 * C) + q * 4;
    int t = ln;
    for (; t + 3 * nl < ntiles; t += 4 * nl) {
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[u][e] = w2s_part_load(p + (size_t)(t + u * nl) * (2 * C) + e);
      s0 += ((double)v[0][0] + (double)v[1][0]) + ((double)v[2][0] + (double)v[3][0]);
      s1 += ((double)v[0][1] + (double)v[1][1]) + ((double)v[2][1] + (double)v[3][1]);
      s2 += ((double)v[0][2] + (double)v[1][2]) + ((double)v[2][2] + (double)v[3][2]);
      s3 += ((double)v[0][3] + (double)v[1][3]) + ((double)v[2][3] + (double)v[3][3]);
    }
    for (; t < ntiles; t += nl) {
      const float* r = p + (size_t)t * (2 * C);
      s0 += (double)w2s_part_load(r); s1 += (double)w2s_part_load(r + 1); s2 += (double)w2s_part_load(r + 2); s3 += (double)w2s_part_load(r + 3);
    }
  }
  // fixed-order tree over the tile lanes, one float4 component at a time (fin_red holds 256 doubles)
  double tot[4] = {s0, s1, s2, s3};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    __syncthreads();
    fin_red[tid] = tot[e];
    __syncthreads();
    for (int half = nl >> 1; half > 0; half >>= 1) {
      if (ln < half) fin_red[tid] += fin_red[tid + half * qn];
      __syncthreads();
    }
    if (ln == 0) fin_sum[q * 4 + e] = fin_red[tid];   // fin_sum[k*C + c]
  }
  __syncthreads();
  if (tid < C) {
    const double a1 = fin_sum[tid], a2 = fin_sum[C + tid];
    float o0, o1;
    if (F.kind == 0) {
      const double mean = a1 * F.inv_count;
      double var = a2 * F.inv_count - mean * mean;
      if (var < 0.0) var = 0.0;
      o0 = (float)mean;
      o1 = (float)(1.0 / sqrt(var + (double)F.eps));
    } else {
      o0 = (float)(a1 * F.inv_count);
      o1 = (float)(a2 * F.inv_count);
    }
    F.out[((size_t)b * C + tid) * 2] = o0;
    F.out[((size_t)b * C + tid) * 2 + 1] = o1;
  }
  if (tid == 0) __hip_atomic_store(&F.cnt[b], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();  // fin_red / fin_last are reused by the next tile of a persistent workgroup
}

// sum over the 16 lanes that share (lane >> 4)  [row of the MFMA output fragment]
// DPP lane permutes inside a 16-lane row (no LDS traffic, unlike the ds_bpermute behind __shfl_xor): pair swap, quad-pair
// swap, then the 8-lane and 16-lane mirrors -- after the quad steps every lane of a quad holds the quad sum, so a mirror
// partner always contributes the other half.  Fixed order => deterministic.
__device__ __forceinline__ float dpp_f(float v, const int ctrl) {
  switch (ctrl) {  // the control must be an immediate
    case 0: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    case 1: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    case 2: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false)); // row_mirror
  }
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f(v, 0);
  v += dpp_f(v, 1);
  v += dpp_f(v, 2);
  v += dpp_f(v, 3);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
}
