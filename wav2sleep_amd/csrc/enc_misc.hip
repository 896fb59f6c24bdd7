// Encoder helper kernels that are not GEMM-shaped: the Cin = 1 first layer (HBM-bound, VALU), the block-0
// residual join, instance-norm statistics finalisation and the conv3 backward pre-pass.
// All reductions are two-level and fixed-order (no float atomics) so results are run-to-run identical.
#include "w2s_common.h"

// ---------------------------------------------------------------------------------------------------
// y1[b,t,o] = sum_j w[o][j] * san(x[b,t+j-1]),  o < 16.  One thread = one position x 4 channels
// (16-B store; 4 lanes cover a 64-B row; a wave writes 1 KiB contiguous).  blocks.py:46 with Cin = 1.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void enc_first_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                            float* __restrict__ part, int L, int tile, int ntiles, int shift) {
  __shared__ float red[4][4][8];
  const int b = blockIdx.y, tl = blockIdx.x, tid = threadIdx.x;
  const int og = tid & 3, lane = tid & 63, wave = tid >> 6;
  float wr[4][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) wr[i][j] = w[(og * 4 + i) * 3 + j];
  const float* xb = x + (size_t)b * L;
  float* yb = y + (size_t)b * L * 16;
  f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  const int t0 = tl * tile;
  for (int p = tid >> 2; p < tile; p += 64) {
    const int t = t0 + p;
    if (t >= L) break;
    // shift = 1: causal padding (blocks.py:150-152,178-182) -- the taps read x[t-2], x[t-1], x[t]
    const int tc = t - shift;
    const float xm = (tc > 0) ? sanitize_f(xb[tc - 1]) : 0.f;
    const float xc = (tc >= 0) ? sanitize_f(xb[tc]) : 0.f;
    const float xp = (tc + 1 < L) ? sanitize_f(xb[tc + 1]) : 0.f;
    f32x4 v;
    v.x = wr[0][0] * xm + wr[0][1] * xc + wr[0][2] * xp;
    v.y = wr[1][0] * xm + wr[1][1] * xc + wr[1][2] * xp;
    v.z = wr[2][0] * xm + wr[2][1] * xc + wr[2][2] * xp;
    v.w = wr[3][0] * xm + wr[3][1] * xc + wr[3][2] * xp;
    if (y) st4(yb + (size_t)t * 16 + og * 4, v);
    s1 += v;
    s2 += v * v;
  }
  // reduce over lanes with equal og (lane bits 2..5), then over the 4 waves
  float a[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float v = a[k];
    v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    a[k] = v;
  }
  if (lane < 4) {
#pragma unroll
    for (int k = 0; k < 8; ++k) red[wave][lane][k] = a[k];
  }
  __syncthreads();
  if (tid < 32) {
    const int k = tid >> 4, c = tid & 15;  // k: 0 sum, 1 sumsq
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) s += red[wv][c >> 2][k * 4 + (c & 3)];
    w2s_part_store(&part[(((size_t)b * ntiles + tl) * 2 + k) * 16 + c], s);
  }
}

int w2s_enc_first_stats_launch(const float* x, const float* w, float* part, int B, int L, int tile, int ntiles, int shift, hipStream_t s,
                               float* xmom = nullptr);   // enc_first_stats.hip

extern "C" int w2s_enc_first_fwd(const float* x, const float* w, float* y, float* part, int B, int L, int cout, int tile, int causal,
                                 void* stream) {
  if (!x || !w || !part || cout != 16 || tile < 64 || (tile & 63)) return W2S_EINVAL;  // y == NULL: statistics only
  const int ntiles = (L + tile - 1) / tile;
  if (!y) {
    return w2s_enc_first_stats_launch(x, w, part, B, L, tile, ntiles, causal ? 1 : 0, reinterpret_cast<hipStream_t>(stream));
  }
  hipLaunchKernelGGL(enc_first_fwd_kernel, dim3(ntiles, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, w, y, part, L, tile,
                     ntiles, causal ? 1 : 0);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// statistics-only form that also keeps the nine raw moments of every tile of the signal: xmom [B][ceil(L/tile)][9] (S0,S1,S2, A00,A11,A22,
// A01,A12,A02 of enc_first_stats.hip) -- w2s_enc_first_wgrad needs their sums and this kernel has them anyway
extern "C" int w2s_enc_first_stats(const float* x, const float* w, float* part, float* xmom, int B, int L, int tile, int causal,
                                   void* stream) {
  if (!x || !w || !part || tile < 64 || (tile & 63)) return W2S_EINVAL;
  return w2s_enc_first_stats_launch(x, w, part, B, L, tile, (L + tile - 1) / tile, causal ? 1 : 0, reinterpret_cast<hipStream_t>(stream), xmom);
}

// pre[b,u,o] = GELU(IN(y3[b,u,o])) + wd[o] * san(x[b,2u])     (blocks.py:67-69, stored pre-activation)
__global__ __launch_bounds__(256) void enc_first_join_kernel(const float* __restrict__ x, const float* __restrict__ wd,
                                                             const float* __restrict__ y3, const float* __restrict__ stats3,
                                                             float* __restrict__ pre, int L, int Lh) {
  const int b = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // float4 index within the sample
  if (i >= (size_t)Lh * 4) return;
  const int u = (int)(i >> 2), og = (int)(i & 3);
  const float* st = stats3 + ((size_t)b * 16 + og * 4) * 2;
  f32x4 s01 = ld4(st), s23 = ld4(st + 4);
  f32x4 mean = {s01.x, s01.z, s23.x, s23.z}, rstd = {s01.y, s01.w, s23.y, s23.w};
  f32x4 v = ld4(y3 + ((size_t)b * Lh + u) * 16 + og * 4);
  const float xv = sanitize_f(x[(size_t)b * L + 2 * u]);
  f32x4 o = gelu4((v - mean) * rstd) + ld4(wd + og * 4) * xv;
  st4(pre + ((size_t)b * Lh + u) * 16 + og * 4, o);
}

extern "C" int w2s_enc_first_join(const float* x, const float* wd, const float* y3, const float* stats3, float* pre, int B, int L, int cout,
                                  void* stream) {
  if (!x || !wd || !y3 || !stats3 || !pre || cout != 16 || (L & 1)) return W2S_EINVAL;
  const int Lh = L / 2;
  hipLaunchKernelGGL(enc_first_join_kernel, dim3((Lh * 4 + 255) / 256, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, wd, y3,
                     stats3, pre, L, Lh);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// Weight gradients of block 0's conv1 (16x1x3) and downsample (16x1x1):
//   dW1[o][j] = sum_{b,t} gy1[b,t,o] * san(x[b,t+j-1]),  gy1 = IN-backward(gn1; y1)
//   dWd[o]    = sum_{b,u} gpre[b,u,o] * san(x[b,2u])
// slab[wg][64]: [0..47] = dW1[o][j], [48..63] = dWd[o].  Sum the slabs with w2s_colsum.
// GH: gn1 and gpre are stored as fp16 (the fp16 gradient chain, w2s_common.h): scale_n / scale_p = their headers' scales.
template <int GH>
__global__ __launch_bounds__(256) void enc_first_bwd_kernel(const float* __restrict__ x, const void* __restrict__ gn1v,
                                                            const float* __restrict__ y1, const float* __restrict__ stats1,
                                                            const float* __restrict__ bstats1, const void* __restrict__ gprev,
                                                            float* __restrict__ slab, int B, int L, const float* __restrict__ w1, int shift,
                                                            const float* __restrict__ hdr_n, const float* __restrict__ hdr_p) {
  const float* gn1 = static_cast<const float*>(gn1v);
  const float* gpre = static_cast<const float*>(gprev);
  const float inv_n = GH ? 1.f / hdr_n[0] : 1.f, inv_p = GH ? 1.f / hdr_p[0] : 1.f;
  __shared__ float red[4][4][16];
  __shared__ float xs[1026];
  const int tid = threadIdx.x, og = tid & 3, lane = tid & 63, wave = tid >> 6;
  float wr[4][3];  // y1 == NULL: the conv1 output is recomputed from x (never stored)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) wr[i][j] = w1 ? w1[(og * 4 + i) * 3 + j] : 0.f;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const int chunks = (L + 1023) / 1024;  // 1024 positions per work item
  for (int item = blockIdx.x; item < B * chunks; item += gridDim.x) {
    const int b = item / chunks, t0 = (item % chunks) * 1024;
    const float* st = stats1 + ((size_t)b * 16 + og * 4) * 2;
    f32x4 s01 = ld4(st), s23 = ld4(st + 4);
    f32x4 mean = {s01.x, s01.z, s23.x, s23.z}, rstd = {s01.y, s01.w, s23.y, s23.w};
    const float* bs = bstats1 + ((size_t)b * 16 + og * 4) * 2;
    f32x4 b01 = ld4(bs), b23 = ld4(bs + 4);
    f32x4 q1 = {b01.x, b01.z, b23.x, b23.z}, q2 = {b01.y, b01.w, b23.y, b23.w};
    const float* xb = x + (size_t)b * L;
    // the item's 1026 sanitised signal samples t0-1 .. t0+1024 (causal padding, shift = 1: t0-2 .. t0+1023) through LDS (one
    // coalesced pass instead of three 4-B loads per position)
    __syncthreads();
    for (int i = tid; i < 1026; i += 256) {
      const int t = t0 - 1 - shift + i;
      const float xv = xb[min(max(t, 0), L - 1)];
      xs[i] = (t >= 0 && t < L && !isinf(xv)) ? xv : 0.f;
    }
    __syncthreads();
    const char* gb = reinterpret_cast<const char*>(gn1) + (size_t)b * L * 16 * (GH ? 2 : 4);
    const float* yb = y1 ? y1 + (size_t)b * L * 16 : nullptr;
    const char* pb = reinterpret_cast<const char*>(gpre) + (size_t)b * (L >> 1) * 16 * (GH ? 2 : 4);
    // straight-line body: no break / divergent branch inside (positions past the end and odd positions contribute through a 0/1
    // factor on clamped addresses).  The branchy form of this loop was NOT bit-reproducible when other kernels shared the CU
    // (tools/determinism_probe*.py: identical inputs, different dW1 sums in situ, correct in isolation).
    for (int p = tid >> 2; p < 1024; p += 64) {
      const int t = t0 + p;
      const float live = (t < L) ? 1.f : 0.f;
      const int tc = min(t, L - 1);
      const unsigned off = (unsigned)tc * 16 + og * 4;
      const float xm = xs[p], xc = xs[p + 1], xp = xs[p + 2];
      f32x4 yv;
      if (yb) yv = ld4o(yb, off);
      else {
        yv.x = wr[0][0] * xm + wr[0][1] * xc + wr[0][2] * xp;
        yv.y = wr[1][0] * xm + wr[1][1] * xc + wr[1][2] * xp;
        yv.z = wr[2][0] * xm + wr[2][1] * xc + wr[2][2] * xp;
        yv.w = wr[3][0] * xm + wr[3][1] * xc + wr[3][2] * xp;
      }
#ifdef W2S_FIRST_BWD_FLOAT4
      // The round-1 form of this block, kept for tools/first_bwd_race.py / tools/pk_fma_opsel_repro.hip only (build with tools/altlib.sh):
      // float4 expressions, which hipcc lowers to v_pk_mul_f32 / v_pk_fma_f32 -- among them two `v_pk_fma_f32 ... op_sel:[0,1,0]`
      // (both lanes multiply by the HIGH half of src1: acc4/acc5 += (xc, xp) * gy.y and acc10/acc11 += (xc, xp) * gy.w) whose LOW-lane
      // results (the odd channels' middle tap) were the sums that differed from launch to launch.
      const f32x4 n4 = (yv - mean) * rstd;
      f32x4 gy4 = rstd * (ld4o(reinterpret_cast<const float*>(gb), off) - q1 - n4 * q2);
#if W2S_FIRST_BWD_FLOAT4 == 2   // same arithmetic, the accumulate block fenced off from the producer of gy (scheduling / forwarding hazard?)
      __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 7"); __builtin_amdgcn_sched_barrier(0);
#endif
      gy4 = gy4 * live;
      acc[0] += gy4.x * xm; acc[1] += gy4.x * xc; acc[2] += gy4.x * xp;
      acc[3] += gy4.y * xm; acc[4] += gy4.y * xc; acc[5] += gy4.y * xp;
      acc[6] += gy4.z * xm; acc[7] += gy4.z * xc; acc[8] += gy4.z * xp;
      acc[9] += gy4.w * xm; acc[10] += gy4.w * xc; acc[11] += gy4.w * xp;
#else
      // scalar arithmetic on purpose: with float4 expressions hipcc built this block from v_pk_* instructions with op_sel operand
      // swizzles, and the sums of the odd channels' middle tap (gy.y * xc, gy.w * xc) came out different from launch to launch when
      // other kernels shared the CU -- a hazard of that instruction mix, not of the data (tools/determinism_probe3.py)
      f32x4 gv;
      if constexpr (GH) gv = h2f4(ld4h(gb, off)) * inv_n; else gv = ld4o(reinterpret_cast<const float*>(gb), off);
      const float yy[4] = {yv.x, yv.y, yv.z, yv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
      const float mm[4] = {mean.x, mean.y, mean.z, mean.w}, rr[4] = {rstd.x, rstd.y, rstd.z, rstd.w};
      const float a1[4] = {q1.x, q1.y, q1.z, q1.w}, a2[4] = {q2.x, q2.y, q2.z, q2.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float nn = (yy[i] - mm[i]) * rr[i];
        const float gy = rr[i] * (gg[i] - a1[i] - nn * a2[i]) * live;
        acc[3 * i] = fmaf(gy, xm, acc[3 * i]);
        acc[3 * i + 1] = fmaf(gy, xc, acc[3 * i + 1]);
        acc[3 * i + 2] = fmaf(gy, xp, acc[3 * i + 2]);
      }
#endif
      const float even = (t & 1) ? 0.f : live;
      f32x4 gp;
      if constexpr (GH) gp = h2f4(ld4h(pb, (unsigned)min(tc >> 1, (L >> 1) - 1) * 16 + og * 4)) * inv_p;
      else gp = ld4o(reinterpret_cast<const float*>(pb), (unsigned)min(tc >> 1, (L >> 1) - 1) * 16 + og * 4);
      const float x0 = xs[p + 1 + shift] * even;  // x[t]: the 1x1/stride-2 residual conv has no padding in either mode
      acc[12] += gp.x * x0; acc[13] += gp.y * x0; acc[14] += gp.z * x0; acc[15] += gp.w * x0;
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    float v = acc[k];
    v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    acc[k] = v;
  }
  if (lane < 4) {
#pragma unroll
    for (int k = 0; k < 16; ++k) red[wave][lane][k] = acc[k];
  }
  __syncthreads();
  if (tid < 64) {
    // output index -> (og, k): dW1[o][j]: o = tid/3, j = tid%3 for tid < 48; dWd[o] = tid-48
    int g4, k;
    if (tid < 48) { const int o = tid / 3, j = tid % 3; g4 = o >> 2; k = (o & 3) * 3 + j; }
    else { const int o = tid - 48; g4 = o >> 2; k = 12 + (o & 3); }
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) s += red[wv][g4][k];
    slab[(size_t)blockIdx.x * 64 + tid] = s;
  }
}

extern "C" int w2s_enc_first_bwd(const float* x, const float* gn1, const float* y1, const float* stats1, const float* bstats1,
                                 const float* gpre, float* slab, int nslab, int B, int L, int cout, const float* w1, int causal, void* stream) {
  if (!x || !gn1 || (!y1 && !w1) || !stats1 || !bstats1 || !gpre || !slab || cout != 16 || nslab <= 0) return W2S_EINVAL;
  hipLaunchKernelGGL(enc_first_bwd_kernel<0>, dim3(nslab), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, gn1, y1, stats1, bstats1,
                     gpre, slab, B, L, y1 ? nullptr : w1, causal ? 1 : 0, nullptr, nullptr);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
// gn1 / gpre stored as fp16 with the scales in their headers (the fp16 gradient chain)
extern "C" int w2s_enc_first_bwd_h(const float* x, const void* gn1, const float* hdr_n, const float* y1, const float* stats1, const float* bstats1,
                                   const void* gpre, const float* hdr_p, float* slab, int nslab, int B, int L, int cout, const float* w1, int causal,
                                   void* stream) {
  if (!x || !gn1 || !hdr_n || !hdr_p || (!y1 && !w1) || !stats1 || !bstats1 || !gpre || !slab || cout != 16 || nslab <= 0) return W2S_EINVAL;
  hipLaunchKernelGGL(enc_first_bwd_kernel<1>, dim3(nslab), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, gn1, y1, stats1, bstats1,
                     gpre, slab, B, L, y1 ? nullptr : w1, causal ? 1 : 0, hdr_n, hdr_p);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// ---------------------------------------------------------------------------------------------------
// First-layer weight gradient WITHOUT the gradient tensor gn1 (w2s_bwd_fused_w1 leaves per-tile sums A[o][j] = sum_t gn1[t][o] xs_j[t],
// xs_j[t] = sanitised zero-padded signal at conv1's tap j of position t).  gy1 = rstd (gn1 - q1 - n1 q2) is linear in gn1, so
//   dW1[o][j] = sum_b rstd_o ( A[o][j] - q1_o X[j] - q2_o N[o][j] ),   X[j] = sum_t xs_j[t],
//   N[o][j] = sum_t n1[t][o] xs_j[t] = rstd_o ( sum_k w1[o][k] XX[k][j] - mean_o X[j] ),   XX[k][j] = sum_t xs_k[t] xs_j[t]
// (n1 = IN(conv1(x)) is itself linear in the signal: the nine moments X, XX of each sample replace the [L][16] tensor; the forward's
// statistics kernel leaves them per tile, w2s_enc_first_stats).  One workgroup per sample, everything summed in a fixed order in fp64;
// out[b][48] = that sample's term (summed over b by w2s_colsum_batch like every other slab).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void enc_first_wgrad_kernel(const float* __restrict__ xmom, int ntx, const float* __restrict__ w1,
                                                               const float* __restrict__ part_w1, const float* __restrict__ stats1,
                                                               const float* __restrict__ bstats1, float* __restrict__ out, int ntiles) {
  __shared__ double mred[64][9];
  __shared__ double mom[9];
  __shared__ double ared[21][48];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < 64 * 9) {   // signal moments: thread (ln, k) = (tid / 9, tid % 9) walks the signal tiles ln, ln + 64, ...
    const int ln = tid / 9, k = tid % 9;
    double s = 0.0;
    for (int t = ln; t < ntx; t += 64) s += (double)xmom[((size_t)b * ntx + t) * 9 + k];
    mred[ln][k] = s;
  }
  if (tid < 21 * 48) {  // tile partials: thread (ln, k) = (tid / 48, tid % 48) walks tiles ln, ln + 21, ...
    const int ln = tid / 48, k = tid % 48;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const float* p = part_w1 + (size_t)b * ntiles * 48 + k;
    int t = ln;
    for (; t + 63 < ntiles; t += 84) {   // four loads in flight
      s0 += (double)p[(size_t)t * 48]; s1 += (double)p[(size_t)(t + 21) * 48];
      s2 += (double)p[(size_t)(t + 42) * 48]; s3 += (double)p[(size_t)(t + 63) * 48];
    }
    for (; t < ntiles; t += 21) s0 += (double)p[(size_t)t * 48];
    ared[ln][k] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (tid < 9) {
    double v = 0.0;
    for (int w = 0; w < 64; ++w) v += mred[w][tid];
    mom[tid] = v;
  }
  __syncthreads();
  if (tid < 48) {
    const int o = tid / 3, j = tid % 3;
    double A = 0.0;
    for (int ln = 0; ln < 21; ++ln) A += ared[ln][tid];
    const double X[3] = {mom[0], mom[1], mom[2]};
    const double XX[3][3] = {{mom[3], mom[6], mom[8]}, {mom[6], mom[4], mom[7]}, {mom[8], mom[7], mom[5]}};   // XX[k][j]
    const double mean = stats1[((size_t)b * 16 + o) * 2], rstd = stats1[((size_t)b * 16 + o) * 2 + 1];
    const double q1 = bstats1[((size_t)b * 16 + o) * 2], q2 = bstats1[((size_t)b * 16 + o) * 2 + 1];
    double wx = 0.0;
    for (int k = 0; k < 3; ++k) wx += (double)w1[o * 3 + k] * XX[k][j];
    const double N = rstd * (wx - mean * X[j]);
    out[(size_t)b * 48 + tid] = (float)(rstd * (A - q1 * X[j] - q2 * N));
  }
}
extern "C" int w2s_enc_first_wgrad(const float* xmom, int ntx, const float* w1, const float* part_w1, const float* stats1, const float* bstats1,
                                   float* out, int B, int ntiles, void* stream) {
  if (!xmom || !w1 || !part_w1 || !stats1 || !bstats1 || !out || B <= 0 || ntx <= 0 || ntiles <= 0) return W2S_EINVAL;
  hipLaunchKernelGGL(enc_first_wgrad_kernel, dim3(B), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), xmom, ntx, w1, part_w1, stats1, bstats1,
                     out, ntiles);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// Weight gradient of block 0's 1x1 / stride-2 residual conv alone: dWd[o] = sum_{b,u} gpre[b][u][o] * san(x[b][2u]); slab[wg][16].
__global__ __launch_bounds__(256) void enc_first_dwd_kernel(const float* __restrict__ x, const float* __restrict__ gpre, float* __restrict__ slab,
                                                            int B, int L) {
  __shared__ float red[4][4][4];
  const int tid = threadIdx.x, og = tid & 3, lane = tid & 63, wave = tid >> 6;
  const int Lh = L >> 1;
  const size_t total = (size_t)B * Lh;
  f32x4 acc = {0, 0, 0, 0};
  // thread = (position lane tid >> 2, channel group og); four positions in flight
  for (size_t u0 = (size_t)blockIdx.x * 256 + (tid >> 2); u0 < total; u0 += (size_t)gridDim.x * 256) {   // (a wave reads 16 consecutive rows)
    f32x4 gv[4];
    float xv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool ok = u0 + 64 * k < total;
      const size_t u = ok ? u0 + 64 * k : total - 1;
      gv[k] = ld4(gpre + u * 16 + og * 4);
      const size_t b = u / Lh, uu = u % Lh;
      const float v = x[b * L + 2 * uu];
      xv[k] = (ok && !isinf(v)) ? v : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc.x = fmaf(gv[k].x, xv[k], acc.x); acc.y = fmaf(gv[k].y, xv[k], acc.y);
      acc.z = fmaf(gv[k].z, xv[k], acc.z); acc.w = fmaf(gv[k].w, xv[k], acc.w);
    }
  }
  float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float s = v[e];
    s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
    if (lane < 4) red[wave][lane][e] = s;
  }
  __syncthreads();
  if (tid < 16) slab[(size_t)blockIdx.x * 16 + tid] = (red[0][tid >> 2][tid & 3] + red[1][tid >> 2][tid & 3]) + (red[2][tid >> 2][tid & 3] + red[3][tid >> 2][tid & 3]);
}
extern "C" int w2s_enc_first_dwd(const float* x, const float* gpre, float* slab, int nslab, int B, int L, void* stream) {
  if (!x || !gpre || !slab || nslab <= 0 || B <= 0 || L <= 0 || (L & 1)) return W2S_EINVAL;
  hipLaunchKernelGGL(enc_first_dwd_kernel, dim3(nslab), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, gpre, slab, B, L);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// ---------------------------------------------------------------------------------------------------
// partial sums -> per-(b,c) statistics.  kind 0: (mean, rstd) with biased variance + eps
// (nn.InstanceNorm1d, models/utils.py:89-92);  kind 1: (sum1/count, sum2/count).  fp64 accumulation.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void stats_finalize_kernel(const float* __restrict__ part, int ntiles, int C, double inv_count, float eps,
                                                              int kind, float* __restrict__ out) {
  // grid (B, 4): block y owns channels [y*C/4, (y+1)*C/4); 1024 threads = CQ channels x (1024/CQ) tile lanes
  __shared__ double red[2][1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int CQ = C >> 2, c0 = blockIdx.y * CQ;
  const int c = tid % CQ, rl = tid / CQ, nrl = 1024 / CQ;
  double s1 = 0.0, s2 = 0.0;
  int t = rl;
  for (; t + 3 * nrl < ntiles; t += 4 * nrl) {   // four rows in flight, added in the same order as the plain loop
    float v1[4], v2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* p = part + (((size_t)b * ntiles + t + u * nrl) * 2) * C + c0;
      v1[u] = p[c];
      v2[u] = p[C + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { s1 += (double)v1[u]; s2 += (double)v2[u]; }
  }
  for (; t < ntiles; t += nrl) {
    const float* p = part + (((size_t)b * ntiles + t) * 2) * C + c0;
    s1 += (double)p[c];
    s2 += (double)p[C + c];
  }
  red[0][tid] = s1;
  red[1][tid] = s2;
  __syncthreads();
  for (int half = nrl >> 1; half > 0; half >>= 1) {  // fixed-order tree over the tile lanes
    if (rl < half) {
      red[0][tid] += red[0][tid + half * CQ];
      red[1][tid] += red[1][tid + half * CQ];
    }
    __syncthreads();
  }
  if (tid < CQ) {
    const double a1 = red[0][tid], a2 = red[1][tid];
    float o0, o1;
    if (kind == 0) {
      const double mean = a1 * inv_count;
      double var = a2 * inv_count - mean * mean;
      if (var < 0.0) var = 0.0;
      o0 = (float)mean;
      o1 = (float)(1.0 / sqrt(var + (double)eps));
    } else {
      o0 = (float)(a1 * inv_count);
      o1 = (float)(a2 * inv_count);
    }
    out[((size_t)b * C + c0 + tid) * 2] = o0;
    out[((size_t)b * C + c0 + tid) * 2 + 1] = o1;
  }
}

extern "C" int w2s_stats_finalize(const float* part, int B, int ntiles, int C, long count, float eps, int kind, float* out, void* stream) {
  if (!part || !out || C < 16 || C > 256 || (C & (C - 1)) || count <= 0) return W2S_EINVAL;
  hipLaunchKernelGGL(stats_finalize_kernel, dim3(B, 4), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), part, ntiles, C,
                     1.0 / (double)count, eps, kind, out);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// ---------------------------------------------------------------------------------------------------
// conv3 backward pre-pass: gn = g * GELU'(n), n = IN(y);  partial sums of gn and gn*n per (b, tile, c).
// ---------------------------------------------------------------------------------------------------
// GH: g is stored as fp16 (header hdr_g = {scale, max}); hdr_amax != NULL (fp32 g only): also publish max |g| and scale 1 -- the
// entry of the fp16 gradient chain (w2s_common.h) derives its first output scale from it.
template <int GH>
__global__ __launch_bounds__(256) void gp_stats_kernel(const void* __restrict__ gv, const float* __restrict__ y,
                                                       const float* __restrict__ stats, float* __restrict__ part, int L, int C, int tile,
                                                       int ntiles, const float* __restrict__ hdr_g, float* __restrict__ hdr_amax) {
  const float* g = static_cast<const float*>(gv);
  const float inv_g = GH ? 1.f / hdr_g[0] : 1.f;
  float amax = 0.f;
  extern __shared__ float sm[];  // [256][8]
  const int b = blockIdx.y, tl = blockIdx.x, tid = threadIdx.x;
  const int c4n = C >> 2, myc4 = tid % c4n, row0 = tid / c4n, rstep = 256 / c4n;
  const float* st = stats + ((size_t)b * C + myc4 * 4) * 2;
  f32x4 s01 = ld4(st), s23 = ld4(st + 4);
  f32x4 mean = {s01.x, s01.z, s23.x, s23.z}, rstd = {s01.y, s01.w, s23.y, s23.w};
  f32x4 a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
  // four rows in flight per thread (the deep layers run this kernel with a few hundred workgroups: one dependent load pair per
  // iteration left it at 45 us for 60 MB); rows beyond the recording load row L-1 again and are masked out
  for (int rr = row0; rr < tile; rr += 4 * rstep) {
    f32x4 yv[4], gr[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = tl * tile + rr + u * rstep;
      ok[u] = (rr + u * rstep < tile) && (t < L);
      const size_t off = ((size_t)b * L + (ok[u] ? t : L - 1)) * C + myc4 * 4;
      yv[u] = ld4(y + off);
      if constexpr (GH) gr[u] = h2f4(*reinterpret_cast<const h16x4*>(static_cast<const _Float16*>(gv) + off)) * inv_g;
      else gr[u] = ld4(g + off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      const f32x4 n = (yv[u] - mean) * rstd;
      amax = amax4(amax, gr[u]);
      const f32x4 gn = gr[u] * gelu_grad4(n);
      a1 += gn;
      a2 += gn * n;
    }
  }
  st4(sm + tid * 8, a1);
  st4(sm + tid * 8 + 4, a2);
  __syncthreads();
  if (tid < 2 * C) {
    const int k = tid / C, c = tid % C;
    float s = 0.f;
    for (int rl = 0; rl < rstep; ++rl) s += sm[(rl * c4n + (c >> 2)) * 8 + k * 4 + (c & 3)];
    w2s_part_store(&part[(((size_t)b * ntiles + tl) * 2 + k) * C + c], s);
  }
  if (!GH && hdr_amax) w2s_amax_commit(hdr_amax, amax, 1.f);   // uniform
}

static int gp_stats_impl(const void* g, int g_half, const float* hdr_g, float* hdr_amax, const float* y, const float* stats, float* part, int B, int L,
                         int C, int tile, void* stream) {
  if (!g || !y || !stats || !part || C < 16 || C > 128 || (C & (C - 1)) || tile <= 0) return W2S_EINVAL;
  if (g_half ? (!hdr_g || hdr_amax) : (hdr_g != nullptr)) return W2S_EINVAL;
  const int ntiles = (L + tile - 1) / tile;
  if (g_half)
    hipLaunchKernelGGL(gp_stats_kernel<1>, dim3(ntiles, B), dim3(256), 256 * 8 * sizeof(float), reinterpret_cast<hipStream_t>(stream), g, y,
                       stats, part, L, C, tile, ntiles, hdr_g, hdr_amax);
  else
    hipLaunchKernelGGL(gp_stats_kernel<0>, dim3(ntiles, B), dim3(256), 256 * 8 * sizeof(float), reinterpret_cast<hipStream_t>(stream), g, y,
                       stats, part, L, C, tile, ntiles, hdr_g, hdr_amax);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
extern "C" int w2s_gp_stats(const float* g, const float* y, const float* stats, float* part, int B, int L, int C, int tile, void* stream) {
  return gp_stats_impl(g, 0, nullptr, nullptr, y, stats, part, B, L, C, tile, stream);
}
extern "C" int w2s_gp_stats_h(const void* g, int g_half, const float* hdr_g, float* hdr_amax, const float* y, const float* stats, float* part, int B,
                              int L, int C, int tile, void* stream) {
  return gp_stats_impl(g, g_half, hdr_g, hdr_amax, y, stats, part, B, L, C, tile, stream);
}
