// Statistics of block 0's conv1 output WITHOUT computing it (the first-layer recompute flow) -- in a translation unit of its own, compiled
// with -fno-slp-vectorize (Makefile): hipcc's SLP vectoriser turned the scalar products of the 32-thread expansion below into
// `v_pk_mul_f32 ... op_sel:[0,1]`, a packed form that returns wrong low-lane results on gfx950 while bf16 MFMA waves share the CU
// (wav2sleep_amd/isa_audit.py, tools/pk_fma_opsel_repro.hip).  The rest of enc_misc.hip keeps the vectoriser (its first-layer weight
// gradient kernel is 30 % slower without it).
#include "w2s_common.h"

// Statistics-only form (the W2S_PRO_FIRST flow never materialises y1).  y1 = conv(x) with ONE input channel, so every per-channel
// sum is a fixed combination of five per-tile scalars of the signal: with xs = sanitised, zero-padded x and t over the tile
//   sum_t y1[t][o]   = sum_j w[o][j] * S_j,            S_j  = sum_t xs[t+j-1]
//   sum_t y1[t][o]^2 = sum_{j,k} w[o][j] w[o][k] A_jk,  A_jk = sum_t xs[t+j-1] xs[t+k-1]
// => 6 multiply-adds per sample instead of 16 channels x 5, then a 32-thread expansion per tile.  fp32 partials per tile, fp64 in
// w2s_stats_finalize as for every other layer.
__global__ __launch_bounds__(256) void enc_first_stats_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ part,
                                                             int L, int tile, int ntiles, int shift, float* __restrict__ xmom) {
  __shared__ float red[4][9];
  __shared__ float tot[9];
  const int b = blockIdx.y, tl = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xb = x + (size_t)b * L;
  const int t0 = tl * tile, t1 = min(L, t0 + tile);
  auto xs = [&](int t) { const float v = (t >= 0 && t < L) ? xb[t] : 0.f; return isinf(v) ? 0.f : v; };
  // S0,S1,S2, A00,A11,A22, A01,A12,A02
  float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int t = t0 + tid; t < t1; t += 256) {
    const float m = xs(t - 1 - shift), c = xs(t - shift), p = xs(t + 1 - shift);   // shift = 1: causal padding (taps at t-2, t-1, t)
    a[0] += m; a[1] += c; a[2] += p;
    a[3] += m * m; a[4] += c * c; a[5] += p * p;
    a[6] += m * c; a[7] += c * p; a[8] += m * p;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) a[k] = wave_sum(a[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) red[wave][k] = a[k];
  }
  __syncthreads();
  if (tid < 9) tot[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  __syncthreads();
  if (xmom && tid >= 32 && tid < 41) xmom[((size_t)b * ntiles + tl) * 9 + (tid - 32)] = tot[tid - 32];   // the tile's nine raw moments (w2s_enc_first_wgrad)
  if (tid < 32) {
    const int k = tid >> 4, o = tid & 15;
    const float w0 = w[o * 3], w1 = w[o * 3 + 1], w2 = w[o * 3 + 2];
    float s;
    if (k == 0) s = w0 * tot[0] + w1 * tot[1] + w2 * tot[2];
    else s = w0 * w0 * tot[3] + w1 * w1 * tot[4] + w2 * w2 * tot[5] + 2.f * (w0 * w1 * tot[6] + w1 * w2 * tot[7] + w0 * w2 * tot[8]);
    w2s_part_store(&part[(((size_t)b * ntiles + tl) * 2 + k) * 16 + o], s);
  }
}

// launcher for w2s_enc_first_fwd (enc_misc.hip)
int w2s_enc_first_stats_launch(const float* x, const float* w, float* part, int B, int L, int tile, int ntiles, int shift, hipStream_t s,
                               float* xmom) {
  hipLaunchKernelGGL(enc_first_stats_kernel, dim3(ntiles, B), dim3(256), 0, s, x, w, part, L, tile, ntiles, shift, xmom);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
