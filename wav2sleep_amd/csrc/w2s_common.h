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
#ifndef W2S_ERF_LOWDEG
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
#define W2S_HE_NP 7
#define W2S_HE_NQ 5
#else
// -DW2S_ERF_LOWDEG: P of 5 and Q of 4 terms (7 instead of 10 polynomial steps), |err| <= 1.9e-6 abs in fp32 (fitted for this file: Lawson-
// weighted least squares on [0, 4 sqrt 2], /tmp-style script in DESIGN.md) -- below the 2^-17 relative rounding of the bf16 hi/lo split
// every GELU output goes through on its way into the matrix cores
#define W2S_HE_P0 -1.0954269113668116e-07f
#define W2S_HE_P1 4.9546486891069972e-05f
#define W2S_HE_P2 0.0042260996717269385f
#define W2S_HE_P3 0.028607305310687833f
#define W2S_HE_P4 0.3989514081808952f
#define W2S_HE_Q0 0.0014177404916614384f
#define W2S_HE_Q1 0.025206600271118533f
#define W2S_HE_Q2 0.23847808392940692f
#define W2S_HE_Q3 1.0f
#define W2S_HE_NP 5
#define W2S_HE_NQ 4
#endif
// The backward's erf (GELU' and the GELU recomputed beside it -- the weight-gradient operand h of the fused backward kernels sees an
// activation that differs from the forward's by up to ~2e-6 |x| --: gelu_grad4 / gelu_both4 / gelu_grad_f -- gradients only, never a logit):
// P of 5 and Q of 4 terms, |err| <= 1.9e-6 abs (7 instead of 10 polynomial steps).  The same fit flipped arg-max labels when the FORWARD
// used it (near-tied logits at the default initialisation, docs/lab_notes_r4.md section 7); in the backward it moves the gradients by
// ~1e-6 relative against a 2e-3 bar (measured 4.3e-4, dominated by the split-precision products).  -DW2S_BWD_ERF_LOWDEG=0: the forward's erf.
#ifndef W2S_BWD_ERF_LOWDEG
#define W2S_BWD_ERF_LOWDEG 1
#endif
#define W2S_HL_P0 -1.0954269113668116e-07f
#define W2S_HL_P1 4.9546486891069972e-05f
#define W2S_HL_P2 0.0042260996717269385f
#define W2S_HL_P3 0.028607305310687833f
#define W2S_HL_P4 0.3989514081808952f
#define W2S_HL_Q0 0.0014177404916614384f
#define W2S_HL_Q1 0.025206600271118533f
#define W2S_HL_Q2 0.23847808392940692f
#define W2S_HL_Q3 1.0f
#define W2S_HE_CLAMP 5.65685424949238f   /* 4 sqrt 2: erf saturates to fp32 beyond */
// The backward's fit overshoots 1/2 beyond x = 4.69 (up to 0.5000017 at the forward's clamp): clamped at 4.6 instead, where it reads
// 0.4999995 (true value 0.4999979; beyond, the true value approaches 1/2 from below), so |half_erf_bwd| < 1/2 everywhere -- cdf stays in
// (0, 1) and GELU' keeps its sign for strongly negative inputs -- at no extra instruction and inside the same 1.9e-6 error bound
#define W2S_HL_CLAMP 4.6f
// -DW2S_ERF_IDENTITY: timing-only build (tools/altlib.sh; numerics WRONG on purpose): the rational erf and the exp2 of GELU' collapse
// to one multiply each -- how fast is the skeleton of a kernel without its transcendental work?  
__device__ __forceinline__ float half_erf_fast(float x) {
#ifdef W2S_ERF_IDENTITY
  return x * 0.125f;
#endif
  x = __builtin_amdgcn_fmed3f(x, -W2S_HE_CLAMP, W2S_HE_CLAMP);
  const float x2 = x * x;
  float p = fmaf(x2, W2S_HE_P0, W2S_HE_P1);
  p = fmaf(x2, p, W2S_HE_P2);
  p = fmaf(x2, p, W2S_HE_P3);
  p = fmaf(x2, p, W2S_HE_P4);
#if W2S_HE_NP == 7
  p = fmaf(x2, p, W2S_HE_P5);
  p = fmaf(x2, p, W2S_HE_P6);
#endif
  float q = fmaf(x2, W2S_HE_Q0, W2S_HE_Q1);
  q = fmaf(x2, q, W2S_HE_Q2);
  q = fmaf(x2, q, W2S_HE_Q3);
#if W2S_HE_NQ == 5
  q = fmaf(x2, q, W2S_HE_Q4);
#endif
  return (x * p) * __builtin_amdgcn_rcpf(q);
}
// exact-form (erf) GELU and its derivative -- models/utils.py:61-74 nn.GELU(approximate='none')
__device__ __forceinline__ float gelu_f(float x) { return x * (0.5f + half_erf_fast(x)); }
__device__ __forceinline__ float half_erf_bwd(float x) {
#if !W2S_BWD_ERF_LOWDEG || defined(W2S_ERF_IDENTITY)
  return half_erf_fast(x);
#else
  x = __builtin_amdgcn_fmed3f(x, -W2S_HL_CLAMP, W2S_HL_CLAMP);
  const float x2 = x * x;
  float p = fmaf(x2, W2S_HL_P0, W2S_HL_P1);
  p = fmaf(x2, p, W2S_HL_P2);
  p = fmaf(x2, p, W2S_HL_P3);
  p = fmaf(x2, p, W2S_HL_P4);
  float q = fmaf(x2, W2S_HL_Q0, W2S_HL_Q1);
  q = fmaf(x2, q, W2S_HL_Q2);
  q = fmaf(x2, q, W2S_HL_Q3);
  return (x * p) * __builtin_amdgcn_rcpf(q);
#endif
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f + half_erf_bwd(x);
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);  // exp(-x^2/2)
  return cdf + x * pdf;
}
// 4-wide forms written on the vector type: every polynomial step is one float4 fma (two v_pk_fma_f32), and the four
// divisions of the rational share ONE v_rcp_f32 (quarter-rate) through r = 1/(q0 q1 q2 q3); |q| is in [0.014, 2.4], so
// the product cannot over/underflow.
__device__ __forceinline__ f32x4 splat4(float c) { return (f32x4){c, c, c, c}; }
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
#ifdef W2S_ERF_IDENTITY
#define W2S_EXP2(t) ((t) * 0.03125f)
#else
#define W2S_EXP2(t) __builtin_amdgcn_exp2f(t)
#endif
__device__ __forceinline__ f32x4 half_erf4(f32x4 x) {
#ifdef W2S_ERF_IDENTITY
  return x * 0.125f;
#endif
  x.x = __builtin_amdgcn_fmed3f(x.x, -W2S_HE_CLAMP, W2S_HE_CLAMP); x.y = __builtin_amdgcn_fmed3f(x.y, -W2S_HE_CLAMP, W2S_HE_CLAMP);
  x.z = __builtin_amdgcn_fmed3f(x.z, -W2S_HE_CLAMP, W2S_HE_CLAMP); x.w = __builtin_amdgcn_fmed3f(x.w, -W2S_HE_CLAMP, W2S_HE_CLAMP);
  const f32x4 x2 = x * x;
  f32x4 p = fma4(x2, splat4(W2S_HE_P0), splat4(W2S_HE_P1));
  p = fma4(x2, p, splat4(W2S_HE_P2));
  p = fma4(x2, p, splat4(W2S_HE_P3));
  p = fma4(x2, p, splat4(W2S_HE_P4));
#if W2S_HE_NP == 7
  p = fma4(x2, p, splat4(W2S_HE_P5));
  p = fma4(x2, p, splat4(W2S_HE_P6));
#endif
  f32x4 q = fma4(x2, splat4(W2S_HE_Q0), splat4(W2S_HE_Q1));
  q = fma4(x2, q, splat4(W2S_HE_Q2));
  q = fma4(x2, q, splat4(W2S_HE_Q3));
#if W2S_HE_NQ == 5
  q = fma4(x2, q, splat4(W2S_HE_Q4));
#endif
  // 1/q for the four lanes from one reciprocal
  const float q01 = q.x * q.y, q23 = q.z * q.w;
  const float r = __builtin_amdgcn_rcpf(q01 * q23);
  const float r01 = r * q23, r23 = r * q01;
  const f32x4 inv = {r01 * q.y, r01 * q.x, r23 * q.w, r23 * q.z};
  return (x * p) * inv;
}
__device__ __forceinline__ f32x4 half_erf4_bwd(f32x4 x) {
#if !W2S_BWD_ERF_LOWDEG || defined(W2S_ERF_IDENTITY)
  return half_erf4(x);
#else
  x.x = __builtin_amdgcn_fmed3f(x.x, -W2S_HL_CLAMP, W2S_HL_CLAMP); x.y = __builtin_amdgcn_fmed3f(x.y, -W2S_HL_CLAMP, W2S_HL_CLAMP);
  x.z = __builtin_amdgcn_fmed3f(x.z, -W2S_HL_CLAMP, W2S_HL_CLAMP); x.w = __builtin_amdgcn_fmed3f(x.w, -W2S_HL_CLAMP, W2S_HL_CLAMP);
  const f32x4 x2 = x * x;
  f32x4 p = fma4(x2, splat4(W2S_HL_P0), splat4(W2S_HL_P1));
  p = fma4(x2, p, splat4(W2S_HL_P2));
  p = fma4(x2, p, splat4(W2S_HL_P3));
  p = fma4(x2, p, splat4(W2S_HL_P4));
  f32x4 q = fma4(x2, splat4(W2S_HL_Q0), splat4(W2S_HL_Q1));
  q = fma4(x2, q, splat4(W2S_HL_Q2));
  q = fma4(x2, q, splat4(W2S_HL_Q3));
  const float q01 = q.x * q.y, q23 = q.z * q.w;
  const float r = __builtin_amdgcn_rcpf(q01 * q23);
  const float r01 = r * q23, r23 = r * q01;
  const f32x4 inv = {r01 * q.y, r01 * q.x, r23 * q.w, r23 * q.z};
  return (x * p) * inv;
#endif
}
__device__ __forceinline__ f32x4 gelu4(f32x4 v) { return v * (half_erf4(v) + 0.5f); }
__device__ __forceinline__ f32x4 gelu_grad4(f32x4 v) {
  const f32x4 cdf = half_erf4_bwd(v) + 0.5f;
  const f32x4 t = v * v * -0.72134752044448170368f;  // exp(-x^2/2) = 2^t
  const f32x4 pdf = {W2S_EXP2(t.x), W2S_EXP2(t.y), W2S_EXP2(t.z), W2S_EXP2(t.w)};
  return fma4(v * 0.39894228040143267794f, pdf, cdf);
}
// GELU and GELU' of the same argument from ONE erf evaluation (the fused backward needs both for the tile's centre rows)
__device__ __forceinline__ void gelu_both4(f32x4 v, f32x4& h, f32x4& gp) {
  const f32x4 cdf = half_erf4_bwd(v) + 0.5f;
  const f32x4 t = v * v * -0.72134752044448170368f;
  const f32x4 pdf = {W2S_EXP2(t.x), W2S_EXP2(t.y), W2S_EXP2(t.z), W2S_EXP2(t.w)};
  h = v * cdf;
  gp = fma4(v * 0.39894228040143267794f, pdf, cdf);
}
// x = hi + lo as two bf16 (round 5: written on the bits -- hipcc's own lowering of `(__bf16)(x - (float)(__bf16)x)` converted half of the
// elements one by one: 14 instructions per four elements, this is 10): hi pair = ONE v_cvt_pk_bf16_f32, its two values back as fp32 are a
// shift and a mask of that word, lo pair = ONE v_cvt_pk_bf16_f32 of the two remainders
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float x, float y, unsigned& hi2, unsigned& lo2) {
  const bf16x2 h = {(__bf16)x, (__bf16)y};
  const unsigned u = __builtin_bit_cast(unsigned, h);
  const float xh = __builtin_bit_cast(float, u << 16), yh = __builtin_bit_cast(float, u & 0xffff0000u);
  const bf16x2 l = {(__bf16)(x - xh), (__bf16)(y - yh)};
  hi2 = u;
  lo2 = __builtin_bit_cast(unsigned, l);
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_store4(__bf16* hi, __bf16* lo, int off, f32x4 t) {
  unsigned h0, h1, l0, l1;
  split_pair(t.x, t.y, h0, l0);
  split_pair(t.z, t.w, h1, l1);
  *reinterpret_cast<u32x2*>(hi + off) = (u32x2){h0, h1};
  *reinterpret_cast<u32x2*>(lo + off) = (u32x2){l0, l1};
}
__device__ __forceinline__ void zero_store4(__bf16* pl, int off) { *reinterpret_cast<u32x2*>(pl + off) = (u32x2){0u, 0u}; }
__device__ __forceinline__ float sanitize_f(float x) { return isinf(x) ? 0.0f : x; }

// The on-load transforms (conv_cl.inl pro_apply) with the per-channel coefficients formed ONCE (per tile / per staging pass) so that an element costs fused
// multiply-adds only (round 5; the textbook forms above are 2 + 3 (+1) dependent operations per element):
//   IN(+GELU):    n  = x r + (-m r)
//   IN backward:  gy = r (g - s1 - n s2) = r g + (-r^2 s2) y + r (r s2 m - s1)
//   ... + GELU':  n  = r y + (-m r),  gy = (r g) GELU'(n) + n (-r s2) + (-r s1)
struct ProCoef { f32x4 a, b, c, d; };
__device__ __forceinline__ ProCoef pro_coef(int pro, f32x4 mean, f32x4 rstd, f32x4 s1, f32x4 s2) {
  ProCoef k{rstd, -(mean * rstd), {0, 0, 0, 0}, {0, 0, 0, 0}};
  if (pro == W2S_PRO_INBWD) { k.b = -(rstd * rstd * s2); k.c = rstd * (rstd * s2 * mean - s1); }
  else if (pro == W2S_PRO_INBWD_GP) { k.c = -(rstd * s2); k.d = -(rstd * s1); }
  else if (pro >= W2S_PRO_AFFINE) { k.a = mean; k.b = rstd; k.c = s1; k.d = s2; }   // (scale, shift) and, for the backward modes, (c, d) (generic path)
  return k;
}
__device__ __forceinline__ f32x4 pro_apply_k(int pro, f32x4 v, f32x4 v2, const ProCoef& k) {
  switch (pro) {
    case W2S_PRO_SANITIZE:
      v.x = sanitize_f(v.x); v.y = sanitize_f(v.y); v.z = sanitize_f(v.z); v.w = sanitize_f(v.w);
      return v;
    case W2S_PRO_GELU:
      return gelu4(v);
    case W2S_PRO_IN_GELU:
    case W2S_PRO_FIRST:
      return gelu4(fma4(v, k.a, k.b));
    case W2S_PRO_INBWD:
      return fma4(k.a, v, fma4(k.b, v2, k.c));
    case W2S_PRO_INBWD_GP: {
      const f32x4 n = fma4(v2, k.a, k.b);
      return fma4(v * k.a, gelu_grad4(n), fma4(n, k.c, k.d));
    }
    case W2S_PRO_AFFINE:       // generic path: act(x * scale + shift), activation = pro - W2S_PRO_AFFINE (get_activation, models/utils.py:61-74)
      return fma4(v, k.a, k.b);
    case W2S_PRO_AFFINE + 1:
      return __builtin_elementwise_max(fma4(v, k.a, k.b), splat4(0.f));
    case W2S_PRO_AFFINE + 2: {
      const f32x4 n = fma4(v, k.a, k.b);
      return __builtin_elementwise_max(n, n * 0.01f);
    }
    case W2S_PRO_AFFINE + 3:
      return gelu4(fma4(v, k.a, k.b));
    case W2S_PRO_AFFINE + 4: {
      const f32x4 n = fma4(v, k.a, k.b);
      return (f32x4){n.x / (1.0f + __expf(-n.x)), n.y / (1.0f + __expf(-n.y)), n.z / (1.0f + __expf(-n.z)), n.w / (1.0f + __expf(-n.w))};
    }
    case W2S_PRO_AFFINE_BWD:      // generic path, backward of the same: z = y scale + shift, gy = (scale g) act'(z) + z c + d
    case W2S_PRO_AFFINE_BWD + 1:
    case W2S_PRO_AFFINE_BWD + 2:
    case W2S_PRO_AFFINE_BWD + 3:
    case W2S_PRO_AFFINE_BWD + 4: {
      const f32x4 n = fma4(v2, k.a, k.b);
      f32x4 d;
      switch (pro - W2S_PRO_AFFINE_BWD) {
        case 1: d = (f32x4){n.x > 0.f ? 1.f : 0.f, n.y > 0.f ? 1.f : 0.f, n.z > 0.f ? 1.f : 0.f, n.w > 0.f ? 1.f : 0.f}; break;
        case 2: d = (f32x4){n.x > 0.f ? 1.f : 0.01f, n.y > 0.f ? 1.f : 0.01f, n.z > 0.f ? 1.f : 0.01f, n.w > 0.f ? 1.f : 0.01f}; break;
        case 3: d = gelu_grad4(n); break;
        case 4: {
          const f32x4 sg = {1.0f / (1.0f + __expf(-n.x)), 1.0f / (1.0f + __expf(-n.y)), 1.0f / (1.0f + __expf(-n.z)), 1.0f / (1.0f + __expf(-n.w))};
          d = sg * (n * (splat4(1.f) - sg) + 1.0f);
          break;
        }
        default: d = splat4(1.f);
      }
      return fma4(v * k.a, d, fma4(n, k.c, k.d));
    }
    default:
      return v;
  }
}

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

// counter-based RNG for dropout: splitmix64 of (seed, element index) -> uniform [0,1).  The same (seed, index) regenerates the same mask
// in the backward pass; nothing is stored.  Round 6 measured a cheaper generator -- three 64 x 64-bit multiplies per element are ~30 of
// gfx950's quarter-rate 32-bit integer multiplies, so: the element index times the golden ratio plus the seed's low word through murmur3's
// 32-bit finaliser, the seed's high word added between its two multiplies (3 multiplies; keep rate, cross-site / cross-step / lagged
// correlations and a 256-bin chi-square checked on 4 M draws) -- and found the step UNCHANGED (28.36 / 28.44 against 28.33 / 28.48 ms: the
// epilogues that draw masks are bound by their stores, docs/lab_notes_r6.md): splitmix64 stays, the other is -DW2S_DROPOUT_SPLITMIX=0.
#ifndef W2S_DROPOUT_SPLITMIX
#define W2S_DROPOUT_SPLITMIX 1
#endif
__device__ __forceinline__ float w2s_uniform(uint64_t seed, uint64_t idx) {
#if W2S_DROPOUT_SPLITMIX
  uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
#else
  uint32_t h = (uint32_t)idx * 0x9E3779B1u + (uint32_t)seed;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h += (uint32_t)(seed >> 32) * 0x27D4EB2Fu;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return (float)(h >> 8) * (1.0f / 16777216.0f);
#endif
}
__device__ __forceinline__ float w2s_dropscale(uint64_t seed, uint64_t idx, float p) {
  return (w2s_uniform(seed, idx) >= p) ? 1.0f / (1.0f - p) : 0.0f;
}

// ------------------------------------------------------------------------------------------------------------------
// Blocked tile assignment of the persistent producers: workgroup w owns a contiguous run of the flattened (sample, tile) list, so it meets
// one or two samples and carries (sample, tile) incrementally -- no integer division per tile.  Statistics leave the producers as per-tile
// fp32 partial sums [B][ntiles][2][C] that w2s_stats_finalize combines in fp64 in a fixed order (exact enough that the finalised values do
// not depend on how batch size and grid cut a sample).  Rounds 1 and 4 built three forms of finalising INSIDE the producer (ticket per
// tile; running double-double sums + one ticket per sample after the tile loop; extra finaliser workgroups polling a counter): correct,
// 168 launches fewer per step, and 0.3-0.9 ms per step SLOWER at four streams every time they were measured (docs/lab_notes_r4.md section 3,
// docs/lab_notes_r5.md) -- removed in round 5.
// ------------------------------------------------------------------------------------------------------------------
// balanced blocked partition of `total` items over `grid` workgroups: the first total % grid runs are one longer (surplus workgroups: empty)
struct W2SRun { int first, count; };
__host__ __device__ __forceinline__ W2SRun w2s_block_part(int total, int grid, int w) {
  if (grid > total) grid = total;
  if (w >= grid) return W2SRun{total, 0};
  const int base = total / grid, rem = total % grid;
  return W2SRun{w * base + (w < rem ? w : rem), base + (w < rem ? 1 : 0)};
}
// (sample, tile) of the i-th item of a run that starts at tile t0 of sample b0 -- without an integer division (a run is short against a
// sample except in the chunk-causal configuration, where the loop below takes a few more turns)
__device__ __forceinline__ void w2s_run_pos(int b0, int t0, int ntiles, int i, int& b, int& tile) {
  tile = t0 + i;
  b = b0;
  while (tile >= ntiles) { tile -= ntiles; ++b; }
}
// per-tile partial sums: plain stores, read by a later launch (w2s_stats_finalize)
__device__ __forceinline__ void w2s_part_store(float* p, float v) { *p = v; }

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
// row16_sum of EIGHT values at once (the statistics epilogue of every conv kernel: sums and sums of squares of four channels): four DPP
// steps, ONE fused v_add_f32_dpp each (the compiler's form of row16_sum is v_mov 0 + v_mov_dpp + add: 85 instructions for the eight
// values, this is 33).  The eight chains are interleaved, so a register written by one instruction is read as a DPP source eight
// instructions later: no VALU -> DPP wait states inside; the s_nop 4 (5 wait states, once per tile) covers an operand written right before the block AND a VALU write of EXEC (v_cmpx) the
// compiler might place there -- its hazard recogniser does not look into inline asm.  Same additions in
// the same order as row16_sum (a + dpp(a) is commutative bit for bit).  All 64 lanes must be active: call it from wave-uniform control flow only.
__device__ __forceinline__ void row16_sum8(f32x4& a, f32x4& b) {
#ifdef W2S_NO_DPP8   // fallback: the compiler's form of the same additions (bitwise the same results; tools/altlib.sh nodpp8 "-DW2S_NO_DPP8=1" ...)
  a = (f32x4){row16_sum(a.x), row16_sum(a.y), row16_sum(a.z), row16_sum(a.w)};
  b = (f32x4){row16_sum(b.x), row16_sum(b.y), row16_sum(b.z), row16_sum(b.w)};
  return;
#endif
  float a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
#define W2S_DPP_STEP(ctl) \
  "v_add_f32_dpp %0, %0, %0 " ctl " row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %1, %1, %1 " ctl " row_mask:0xf bank_mask:0xf\n\t" \
  "v_add_f32_dpp %2, %2, %2 " ctl " row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %3, %3, %3 " ctl " row_mask:0xf bank_mask:0xf\n\t" \
  "v_add_f32_dpp %4, %4, %4 " ctl " row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %5, %5, %5 " ctl " row_mask:0xf bank_mask:0xf\n\t" \
  "v_add_f32_dpp %6, %6, %6 " ctl " row_mask:0xf bank_mask:0xf\n\t" "v_add_f32_dpp %7, %7, %7 " ctl " row_mask:0xf bank_mask:0xf\n\t"
  asm volatile("s_nop 4\n\t" W2S_DPP_STEP("quad_perm:[1,0,3,2]") W2S_DPP_STEP("quad_perm:[2,3,0,1]") W2S_DPP_STEP("row_half_mirror") W2S_DPP_STEP("row_mirror")
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
#undef W2S_DPP_STEP
  a = (f32x4){a0, a1, a2, a3};
  b = (f32x4){b0, b1, b2, b3};
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
