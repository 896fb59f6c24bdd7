// Kernels of the GENERIC (untuned, inference) path: the module variants the reference can be configured into besides its shipped
// production model -- BatchNorm / LayerNorm / RMS / GroupNorm / no norm, ReLU / LeakyReLU / SiLU / linear activations, any feature_dim
// and head count, SleepPPGNet (models/utils.py:26-96, models/blocks.py:129-186, models/ppgnet.py).  The production configuration never
// comes here: its forward / backward is the fused path (conv_cl / conv_wide / fwd_fused / bwd_fused).  Convolutions and GEMMs of the
// generic path are w2s_conv_forward with no prologue; what is left is the three small kernels below (HBM-bound elementwise / row ops).
#include "w2s_common.h"

// act: 0 linear, 1 ReLU, 2 LeakyReLU(slope), 3 GELU (erf), 4 SiLU     -- models/utils.py:61-74 get_activation
__device__ __forceinline__ float act_f(float v, int act, float slope) {
  switch (act) {
    case 1: return v > 0.f ? v : 0.f;
    case 2: return v > 0.f ? v : v * slope;
    case 3: return gelu_f(v);
    case 4: return v / (1.0f + __expf(-v));
    default: return v;
  }
}

// y[row][c] = act(x[row][c] * scale[s][c] + shift[s][c]),  s = (row / rows_per_sample) * sample_stride   (sample_stride 0: one vector
// for all samples).  Instance norm (per-sample mean / rstd folded into scale / shift), eval-mode BatchNorm (running statistics and
// affine folded), GroupNorm (per-sample, per-group statistics x per-channel affine) and the bare activation (scale == NULL) all have
// this shape.  In place allowed (y == x).  One thread = 4 consecutive channels.
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int sample_stride, float* __restrict__ y, int ldy,
                                                         int rows_per_sample, long rows, int C, int act, float slope) {
  const int c4n = C >> 2;
  const long total = rows * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / c4n;
    const int c = (int)(i % c4n) * 4;
    f32x4 v = ld4(x + row * ldx + c);
    if (scale) {
      const long s = (row / rows_per_sample) * sample_stride + c;
      v = v * ld4(scale + s) + ld4(shift + s);
    }
    v.x = act_f(v.x, act, slope); v.y = act_f(v.y, act, slope); v.z = act_f(v.z, act, slope); v.w = act_f(v.w, act, slope);
    st4(y + row * ldy + c, v);
  }
}

extern "C" int w2s_affine_act(const float* x, int ldx, const float* scale, const float* shift, int sample_stride, float* y, int ldy,
                              int rows_per_sample, long rows, int C, int act, float slope, void* stream) {
  if (!x || !y || rows <= 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldy & 3) || rows_per_sample <= 0 || act < 0 || act > 4) return W2S_EINVAL;
  if ((scale == nullptr) != (shift == nullptr) || (sample_stride & 3)) return W2S_EINVAL;
  const long total = rows * (C >> 2);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(affine_act_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx, scale, shift,
                     sample_stride, y, ldy, rows_per_sample, rows, C, act, slope);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// Normalisation over the channel dimension of one position (channels-last: a row): ConvLayerNorm (models/utils.py:9-23), ConvRMSNorm
// (:26-38: rms != 0, no mean, no beta) and nn.LayerNorm, for ANY channel count, followed by the activation.  One wave per row.
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y, int ldy, long rows, int C,
                                                          float eps, int rms, int act, float slope) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float* xr = x + row * ldx;
    float s = 0.f;
    if (!rms) { for (int c = lane; c < C; c += 64) s += xr[c]; }
    const float mean = rms ? 0.f : wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    for (int c = lane; c < C; c += 64) {
      float o = (xr[c] - mean) * rstd * gamma[c];
      if (beta) o += beta[c];
      y[row * ldy + c] = act_f(o, act, slope);
    }
  }
}

extern "C" int w2s_rownorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, long rows, int C, float eps,
                               int rms, int act, float slope, void* stream) {
  if (!x || !gamma || !y || rows <= 0 || C <= 0 || act < 0 || act > 4) return W2S_EINVAL;
  long blocks = (rows + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(rownorm_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx, gamma, beta, y,
                     ldy, rows, C, eps, rms, act, slope);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// Scaled-dot-product attention core of nn.MultiheadAttention for ANY head size and up to 16 tokens per sentence (the set-fusion
// transformer attends over 1 + C modality tokens): qkv [N][D][3 H hd] (q | k | v), keypad [N][D] (1 = padded key), out [N][D][H hd].
// One thread per (sentence, head, query token); inference (no dropout).  Production (hd = 16, D <= 7) uses w2s_attn_fwd instead.
__global__ __launch_bounds__(256) void attn_generic_fwd_kernel(const float* __restrict__ qkv, const unsigned char* __restrict__ keypad,
                                                               float* __restrict__ out, long N, int D, int H, int hd, float scale, float p,
                                                               uint64_t seed) {
  const long total = N * H * D;
  const int F = H * hd;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int qi = (int)(i % D), h = (int)((i / D) % H);
    const long n = i / ((long)D * H);
    const float* base = qkv + n * D * 3 * F;
    const float* q = base + (long)qi * 3 * F + h * hd;
    float sc[16];
    float mx = -INFINITY;
    for (int j = 0; j < D; ++j) {
      const float* k = base + (long)j * 3 * F + F + h * hd;
      float s = 0.f;
      for (int e = 0; e < hd; ++e) s += q[e] * k[e];
      s = keypad[n * D + j] ? -INFINITY : s * scale;
      sc[j] = s;
      mx = fmaxf(mx, s);
    }
    float den = 0.f;
    for (int j = 0; j < D; ++j) { sc[j] = __expf(sc[j] - mx); den += sc[j]; }
    const float inv = 1.0f / den;
    if (p > 0.f)   // dropout on the attention weights (F.multi_head_attention_forward, training): mask index ((n H + h) D + query) D + key
      for (int j = 0; j < D; ++j) sc[j] *= w2s_dropscale(seed, (uint64_t)(i * D + j), p);
    float* o = out + (n * D + qi) * F + h * hd;
    for (int e = 0; e < hd; ++e) {
      float a = 0.f;
      for (int j = 0; j < D; ++j) a += sc[j] * base[(long)j * 3 * F + 2 * F + h * hd + e];
      o[e] = a * inv;
    }
  }
}

extern "C" int w2s_attn_generic_fwd(const float* qkv, const unsigned char* keypad, float* out, long N, int D, int H, int hd, float p_drop,
                                    uint64_t seed, void* stream) {
  if (!qkv || !keypad || !out || N <= 0 || D <= 0 || D > 16 || H <= 0 || hd <= 0 || p_drop < 0.f || p_drop >= 1.f) return W2S_EINVAL;
  const long total = N * H * D;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(attn_generic_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), qkv, keypad, out, N,
                     D, H, hd, 1.0f / sqrtf((float)hd), p_drop, seed);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// ===================================================================================================================
// Backward kernels of the generic path (round 6): training of the module variants above -- SleepPPGNet (BatchNorm / LeakyReLU, which the
// reference trains: models/ppgnet.py), feature sizes other than 128, the other norms and activations.  Data and weight gradients of the
// convolutions / GEMMs are w2s_conv_forward (flip / UP2 forms) and w2s_wgrad; what is left is the backward of norm -> activation
// (ConvLayer1D.forward, blocks.py:183-185), of the row norms and of the attention core.  HBM-bound elementwise / row kernels like the
// forward's three; per-(sample, channel) sums leave as per-tile fp32 partials for w2s_stats_finalize (fp64, fixed order).
// ===================================================================================================================
__device__ __forceinline__ float act_grad_f(float z, int act, float slope) {
  switch (act) {
    case 1: return z > 0.f ? 1.f : 0.f;        // threshold_backward
    case 2: return z > 0.f ? 1.f : slope;      // leaky_relu_backward
    case 3: return gelu_grad_f(z);
    case 4: { const float sg = 1.0f / (1.0f + __expf(-z)); return sg * (1.0f + z * (1.0f - sg)); }
    default: return 1.f;
  }
}

// the normalised value xh and the activation's input z of four channels:  xh = stats ? (y - mean) * rstd : y,   z = gamma ? xh * gamma + beta : xh
struct NormCh { f32x4 mean, rstd, gamma, beta; int has_stats, has_affine; };
__device__ __forceinline__ NormCh norm_ch(const float* stats, long soff, const float* gamma, const float* beta, int c) {
  NormCh k;
  k.has_stats = stats != nullptr;
  k.has_affine = gamma != nullptr;
  k.mean = splat4(0.f); k.rstd = splat4(1.f); k.gamma = splat4(1.f); k.beta = splat4(0.f);
  if (stats) {
    const f32x4 a = ld4(stats + soff + 2 * c), b = ld4(stats + soff + 2 * c + 4);   // (mean, rstd) interleaved
    k.mean = (f32x4){a.x, a.z, b.x, b.z};
    k.rstd = (f32x4){a.y, a.w, b.y, b.w};
  }
  if (gamma) { k.gamma = ld4(gamma + c); if (beta) k.beta = ld4(beta + c); }
  return k;
}
__device__ __forceinline__ void norm_act_grad4(const NormCh& k, f32x4 g, f32x4 y, int act, float slope, f32x4& ga, f32x4& xh) {
  xh = (y - k.mean) * k.rstd;
  const f32x4 z = xh * k.gamma + k.beta;
  ga = (f32x4){g.x * act_grad_f(z.x, act, slope), g.y * act_grad_f(z.y, act, slope), g.z * act_grad_f(z.z, act, slope),
               g.w * act_grad_f(z.w, act, slope)};
}

// pass 1: part[s][tile][0][c] = sum_rows ga, part[s][tile][1][c] = sum_rows ga * xh,   ga = g * act'(z)   (rows of sample s in the tile)
__global__ __launch_bounds__(256) void norm_act_bwd_part_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ y, int ldy,
                                                                const float* __restrict__ stats, int stats_stride, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int rows_per_sample, int C, int act, float slope,
                                                                int tile, float* __restrict__ part) {
  __shared__ float sm[2048];   // [rows in flight][2][C]
  const int s = blockIdx.y, tl = blockIdx.x, nt = gridDim.x, tid = threadIdx.x;
  const int c4n = C >> 2, rpb = 256 / c4n, q = tid % c4n, rr = tid / c4n, c = q * 4;
  f32x4 a1 = splat4(0.f), a2 = splat4(0.f);
  if (rr < rpb) {
    const NormCh k = norm_ch(stats, (long)s * stats_stride, gamma, beta, c);
    const long row0 = (long)s * rows_per_sample;
    const int t1 = min(rows_per_sample, (tl + 1) * tile);
    int t = tl * tile + rr;
    for (; t + 3 * rpb < t1; t += 4 * rpb) {   // four rows in flight per thread (one dependent load pair per iteration ran at 2 TB/s)
      f32x4 gv[4], yv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { gv[u] = ld4(g + (row0 + t + u * rpb) * ldg + c); yv[u] = ld4(y + (row0 + t + u * rpb) * ldy + c); }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4 ga, xh;
        norm_act_grad4(k, gv[u], yv[u], act, slope, ga, xh);
        a1 += ga;
        a2 += ga * xh;
      }
    }
    for (; t < t1; t += rpb) {
      f32x4 ga, xh;
      norm_act_grad4(k, ld4(g + (row0 + t) * ldg + c), ld4(y + (row0 + t) * ldy + c), act, slope, ga, xh);
      a1 += ga;
      a2 += ga * xh;
    }
    st4(sm + (rr * 2 + 0) * C + c, a1);
    st4(sm + (rr * 2 + 1) * C + c, a2);
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += 256) {
    float v = 0.f;
    for (int r = 0; r < rpb; ++r) v += sm[r * 2 * C + i];
    part[((size_t)s * nt + tl) * 2 * C + i] = v;
  }
}

extern "C" int w2s_norm_act_bwd_part(const float* g, int ldg, const float* y, int ldy, const float* stats, int stats_stride, const float* gamma,
                                     const float* beta, int rows_per_sample, int nsamples, int C, int act, float slope, int tile, float* part,
                                     void* stream) {
  if (!g || !y || !part || rows_per_sample <= 0 || nsamples <= 0 || C < 4 || C > 1024 || (C & 3) || (ldg & 3) || (ldy & 3) || (stats_stride & 3) ||
      act < 0 || act > 4 || tile <= 0 || (beta && !gamma))
    return W2S_EINVAL;
  const int nt = (rows_per_sample + tile - 1) / tile;
  hipLaunchKernelGGL(norm_act_bwd_part_kernel, dim3(nt, nsamples), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, ldg, y, ldy, stats,
                     stats_stride, gamma, beta, rows_per_sample, C, act, slope, tile, part);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// pass 2: gy = coef ? A * ga + B + Cx * xh : ga     coef [.][3][C] = (A, B, Cx) per (sample, channel) (coef_stride 0: one set for all samples):
// every statistics-based norm's backward has this shape -- instance (A = rstd, B = -rstd mean(ga), Cx = -rstd mean(ga xh)), BatchNorm in
// training (the same with the means over the batch and gamma folded), GroupNorm (means over the group), eval-mode BatchNorm (A only).
__global__ __launch_bounds__(256) void norm_act_bwd_apply_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ y, int ldy,
                                                                 const float* __restrict__ stats, int stats_stride, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ coef, int coef_stride,
                                                                 float* __restrict__ gy, int ldgy, int rows_per_sample, long rows, int C, int act,
                                                                 float slope) {
  const int c4n = C >> 2;
  const long total = rows * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / c4n;
    const int c = (int)(i % c4n) * 4;
    const long s = row / rows_per_sample;
    const NormCh k = norm_ch(stats, s * stats_stride, gamma, beta, c);
    f32x4 ga, xh;
    norm_act_grad4(k, ld4(g + row * ldg + c), ld4(y + row * ldy + c), act, slope, ga, xh);
    if (coef) {
      const float* cf = coef + s * coef_stride + c;
      ga = ld4(cf) * ga + ld4(cf + C) + ld4(cf + 2 * C) * xh;
    }
    st4(gy + row * ldgy + c, ga);
  }
}

extern "C" int w2s_norm_act_bwd_apply(const float* g, int ldg, const float* y, int ldy, const float* stats, int stats_stride, const float* gamma,
                                      const float* beta, const float* coef, int coef_stride, float* gy, int ldgy, int rows_per_sample, long rows,
                                      int C, int act, float slope, void* stream) {
  if (!g || !y || !gy || rows_per_sample <= 0 || rows <= 0 || C < 4 || (C & 3) || (ldg & 3) || (ldy & 3) || (ldgy & 3) || (stats_stride & 3) ||
      (coef_stride & 3) || act < 0 || act > 4 || (beta && !gamma))
    return W2S_EINVAL;
  const long total = rows * (C >> 2);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(norm_act_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, ldg, y, ldy, stats,
                     stats_stride, gamma, beta, coef, coef_stride, gy, ldgy, rows_per_sample, rows, C, act, slope);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// Backward of w2s_rownorm_fwd (norm over the channels of a row, affine, activation): x = the norm's INPUT, g = gradient of the activation's
// output.  gx = rstd (gh - mean_c(gh) - xh mean_c(gh xh)), gh = g act'(z) gamma  (RMS: no mean term);  part[block][0][c] = sum_rows
// g act'(z) xh (gamma's gradient), part[block][1][c] = sum_rows g act'(z) (beta's).  One wave per row, any C <= 1024; gx may alias g.
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ gx,
                                                          int ldgx, float* __restrict__ part, long rows, int C, float eps, int rms, int act,
                                                          float slope) {
  extern __shared__ float sm[];   // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* sg = sm + wave * 2 * C;
  float* sb = sg + C;
  for (int c = lane; c < C; c += 64) { sg[c] = 0.f; sb[c] = 0.f; }
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float* xr = x + row * ldx;
    const float* gr = g + row * ldg;
    float s = 0.f;
    if (!rms) { for (int c = lane; c < C; c += 64) s += xr[c]; }
    const float mean = rms ? 0.f : wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float xh = (xr[c] - mean) * rstd;
      const float z = xh * gamma[c] + (beta ? beta[c] : 0.f);
      const float ga = gr[c] * act_grad_f(z, act, slope);
      const float gh = ga * gamma[c];
      s1 += gh;
      s2 += gh * xh;
      sg[c] += ga * xh;
      sb[c] += ga;
    }
    const float m1 = rms ? 0.f : wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
    for (int c = lane; c < C; c += 64) {
      const float xh = (xr[c] - mean) * rstd;
      const float z = xh * gamma[c] + (beta ? beta[c] : 0.f);
      const float gh = gr[c] * act_grad_f(z, act, slope) * gamma[c];
      gx[row * ldgx + c] = rstd * (gh - m1 - xh * m2);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) part[(size_t)blockIdx.x * 2 * C + i] = sm[i] + sm[2 * C + i] + sm[4 * C + i] + sm[6 * C + i];
}

extern "C" int w2s_rownorm_bwd_blocks(long rows) {
  long blocks = (rows + 3) / 4;
  return (int)(blocks > 2048 ? 2048 : (blocks < 1 ? 1 : blocks));
}

extern "C" int w2s_rownorm_bwd(const float* g, int ldg, const float* x, int ldx, const float* gamma, const float* beta, float* gx, int ldgx,
                               float* part, long rows, int C, float eps, int rms, int act, float slope, void* stream) {
  if (!g || !x || !gamma || !gx || !part || rows <= 0 || C <= 0 || C > 1024 || act < 0 || act > 4) return W2S_EINVAL;
  const int blocks = w2s_rownorm_bwd_blocks(rows);
  hipLaunchKernelGGL(rownorm_bwd_kernel, dim3(blocks), dim3(256), (size_t)8 * C * sizeof(float), reinterpret_cast<hipStream_t>(stream), g, ldg, x,
                     ldx, gamma, beta, gx, ldgx, part, rows, C, eps, rms, act, slope);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// Backward of w2s_attn_generic_fwd: gqkv [N][D][3 H hd] from gout [N][D][H hd]; one thread per (sentence, head) walks the queries, so the
// key / value gradients it accumulates (in gqkv itself) are its own -- no atomics, fixed order.  The softmax (and the dropout mask of the
// forward, same seed) is recomputed per query.
__global__ __launch_bounds__(256) void attn_generic_bwd_kernel(const float* __restrict__ qkv, const unsigned char* __restrict__ keypad,
                                                               const float* __restrict__ gout, float* __restrict__ gqkv, long N, int D, int H,
                                                               int hd, float scale, float p, uint64_t seed) {
  const long total = N * H;
  const int F = H * hd;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int h = (int)(i % H);
    const long n = i / H;
    const float* base = qkv + n * D * 3 * F + h * hd;
    float* gbase = gqkv + n * D * 3 * F + h * hd;
    for (int j = 0; j < D; ++j)
      for (int e = 0; e < hd; ++e) { gbase[(long)j * 3 * F + F + e] = 0.f; gbase[(long)j * 3 * F + 2 * F + e] = 0.f; }
    for (int qi = 0; qi < D; ++qi) {
      const float* q = base + (long)qi * 3 * F;
      const float* go = gout + (n * D + qi) * F + h * hd;
      float pr[16], dp[16];
      float mx = -INFINITY;
      for (int j = 0; j < D; ++j) {
        const float* k = base + (long)j * 3 * F + F;
        float s = 0.f;
        for (int e = 0; e < hd; ++e) s += q[e] * k[e];
        s = keypad[n * D + j] ? -INFINITY : s * scale;
        pr[j] = s;
        mx = fmaxf(mx, s);
      }
      float den = 0.f;
      for (int j = 0; j < D; ++j) { pr[j] = __expf(pr[j] - mx); den += pr[j]; }
      const float inv = 1.0f / den;
      float delta = 0.f;
      for (int j = 0; j < D; ++j) {
        const float m = p > 0.f ? w2s_dropscale(seed, (uint64_t)((i * D + qi) * D + j), p) : 1.f;
        const float pj = pr[j] * inv;
        const float* v = base + (long)j * 3 * F + 2 * F;
        float* gv = gbase + (long)j * 3 * F + 2 * F;
        float d = 0.f;
        for (int e = 0; e < hd; ++e) { d += go[e] * v[e]; gv[e] += pj * m * go[e]; }
        d *= m;           // gradient of the pre-dropout weight
        pr[j] = pj;
        dp[j] = d;
        delta += pj * d;
      }
      float* gq = gbase + (long)qi * 3 * F;
      for (int e = 0; e < hd; ++e) gq[e] = 0.f;
      for (int j = 0; j < D; ++j) {
        const float ds = pr[j] * (dp[j] - delta) * scale;
        const float* k = base + (long)j * 3 * F + F;
        float* gk = gbase + (long)j * 3 * F + F;
        for (int e = 0; e < hd; ++e) { gq[e] += ds * k[e]; gk[e] += ds * q[e]; }
      }
    }
  }
}

extern "C" int w2s_attn_generic_bwd(const float* qkv, const unsigned char* keypad, const float* gout, float* gqkv, long N, int D, int H, int hd,
                                    float p_drop, uint64_t seed, void* stream) {
  if (!qkv || !keypad || !gout || !gqkv || N <= 0 || D <= 0 || D > 16 || H <= 0 || hd <= 0 || p_drop < 0.f || p_drop >= 1.f) return W2S_EINVAL;
  const long total = N * H;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(attn_generic_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), qkv, keypad, gout, gqkv,
                     N, D, H, hd, 1.0f / sqrtf((float)hd), p_drop, seed);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The per-(sample, channel) arithmetic around the statistics-based norms, one launch each instead of ~30 torch launches on [B, C] tensors
// per layer (a SleepPPGNet step spent 6 of its 32 ms of kernel time in 1300 such launches): fp64 inside, one thread per channel.
// kind: 0 InstanceNorm1d (stats = (mean, rstd) per (b, c), no affine), 1 BatchNorm1d in training (stats = (E[y], E[y^2]) per (b, c); the
// running statistics are updated with the unbiased variance as nn.BatchNorm1d does), 2 BatchNorm1d in eval mode (running statistics),
// 3 GroupNorm (stats = (E[y], E[y^2]); G groups of C / G consecutive channels).
// out: scale / shift [nset][C] for w2s_affine_act and mr [nset][C][2] = (mean, rstd) for the backward; nset = B (kinds 0, 3) or 1 (1, 2);
// ss (optional) [B][C][2] = (scale, shift) for every sample: the W2S_PRO_AFFINE operand of the next layer's conv / weight gradient.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void norm_fold_kernel(int kind, const float* __restrict__ stats, int B, int C, int G, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float* __restrict__ run_mean, float* __restrict__ run_var, float eps, float momentum,
                                 double count, float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mr,
                                 float* __restrict__ ss) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double gm = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
  if (kind == 1 || kind == 2) {
    double m, var;
    if (kind == 1) {
      double e1 = 0.0, e2 = 0.0;
      for (int b = 0; b < B; ++b) { e1 += (double)stats[((size_t)b * C + c) * 2]; e2 += (double)stats[((size_t)b * C + c) * 2 + 1]; }
      m = e1 / B;
      var = e2 / B - m * m;
      if (var < 0.0) var = 0.0;
      if (run_mean) {
        run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * m);
        run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
      }
    } else {
      m = run_mean[c];
      var = run_var[c];
    }
    const double rstd = 1.0 / sqrt(var + (double)eps);
    scale[c] = (float)(gm * rstd);
    shift[c] = (float)(bt - m * gm * rstd);
    mr[2 * c] = (float)m;
    mr[2 * c + 1] = (float)rstd;
    if (ss)
      for (int b = 0; b < B; ++b) { ss[((size_t)b * C + c) * 2] = scale[c]; ss[((size_t)b * C + c) * 2 + 1] = shift[c]; }
    return;
  }
  const int cg = C / G, c0 = (c / cg) * cg;
  for (int b = 0; b < B; ++b) {
    double m, rstd;
    if (kind == 0) {
      m = stats[((size_t)b * C + c) * 2];
      rstd = stats[((size_t)b * C + c) * 2 + 1];
    } else {
      double e1 = 0.0, e2 = 0.0;
      for (int k = 0; k < cg; ++k) { e1 += (double)stats[((size_t)b * C + c0 + k) * 2]; e2 += (double)stats[((size_t)b * C + c0 + k) * 2 + 1]; }
      e1 /= cg; e2 /= cg;
      double var = e2 - e1 * e1;
      if (var < 0.0) var = 0.0;
      m = e1;
      rstd = 1.0 / sqrt(var + (double)eps);
    }
    scale[(size_t)b * C + c] = (float)(gm * rstd);
    shift[(size_t)b * C + c] = (float)(bt - m * gm * rstd);
    mr[((size_t)b * C + c) * 2] = (float)m;
    mr[((size_t)b * C + c) * 2 + 1] = (float)rstd;
    if (ss) { ss[((size_t)b * C + c) * 2] = scale[(size_t)b * C + c]; ss[((size_t)b * C + c) * 2 + 1] = shift[(size_t)b * C + c]; }
  }
}

extern "C" int w2s_norm_fold(int kind, const float* stats, int B, int C, int G, const float* gamma, const float* beta, float* run_mean, float* run_var,
                             float eps, float momentum, double count, float* scale, float* shift, float* mr, float* ss, void* stream) {
  if (kind < 0 || kind > 3 || B <= 0 || C <= 0 || !scale || !shift || !mr || (kind != 2 && !stats) || (kind == 2 && (!run_mean || !run_var)) ||
      (kind == 3 && (G <= 0 || C % G)) || (beta && !gamma))
    return W2S_EINVAL;
  hipLaunchKernelGGL(norm_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), kind, stats, B, C, kind == 3 ? G : C,
                     gamma, beta, run_mean, run_var, eps, momentum, count, scale, shift, mr, ss);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// backward counterpart: means [B][C][2] = per-(sample, channel) means over L of ga and ga * xh (w2s_norm_act_bwd_part + w2s_stats_finalize
// kind 1), mr as written by w2s_norm_fold -> coef [nset][3][C] = (A, B, Cx) of w2s_norm_act_bwd_apply and the affine parameters' gradients
// dgamma[c] = L sum_b means[b][c][1], dbeta[c] = L sum_b means[b][c][0] (NULL for kind 0).
__global__ void norm_bwd_coef_kernel(int kind, const float* __restrict__ means, const float* __restrict__ mr, int B, int C, int G,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, double L, float* __restrict__ coef,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ cd, int y_sums) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double gm = gamma ? (double)gamma[c] : 1.0;
  const int per_sample = (kind == 0 || kind == 3);
  // y_sums: the second mean is of ga * y (the conv epilogue's W2S_EPI_AFFINE_PART sums) instead of ga * xh: xh = (y - mean) rstd
  auto M2 = [&](int b) -> double {
    const double e1 = means[((size_t)b * C + c) * 2], e2 = means[((size_t)b * C + c) * 2 + 1];
    if (!y_sums) return e2;
    const size_t q = per_sample ? (size_t)b * C + c : (size_t)c;
    return (double)mr[2 * q + 1] * (e2 - (double)mr[2 * q] * e1);
  };
  double s1 = 0.0, s2 = 0.0;
  for (int b = 0; b < B; ++b) { s1 += (double)means[((size_t)b * C + c) * 2]; s2 += M2(b); }
  if (dgamma) dgamma[c] = (float)(s2 * L);
  if (dbeta) dbeta[c] = (float)(s1 * L);
  if (kind == 1 || kind == 2) {
    const double rstd = mr[2 * c + 1];
    const double cb = kind == 1 ? -rstd * gm * s1 / B : 0.0, cx = kind == 1 ? -rstd * gm * s2 / B : 0.0;
    coef[c] = (float)(rstd * gm);
    coef[C + c] = (float)cb;
    coef[2 * C + c] = (float)cx;
    if (cd) {   // W2S_PRO_AFFINE_BWD operand: gy = (scale g) act'(z) + z c + d,  c = Cx / gamma, d = B - c beta  (gamma = 0: Cx = 0 as well)
      const double cc = gm != 0.0 ? cx / gm : 0.0, dd = cb - cc * (beta ? (double)beta[c] : 0.0);
      for (int b = 0; b < B; ++b) { cd[((size_t)b * C + c) * 2] = (float)cc; cd[((size_t)b * C + c) * 2 + 1] = (float)dd; }
    }
    return;
  }
  const int cg = C / G, c0 = (c / cg) * cg;
  for (int b = 0; b < B; ++b) {
    const double rstd = mr[((size_t)b * C + c) * 2 + 1];
    double m1, m2;
    if (kind == 0) {
      m1 = means[((size_t)b * C + c) * 2];
      m2 = M2(b);
    } else {
      m1 = 0.0; m2 = 0.0;
      for (int k = 0; k < cg; ++k) {
        const double gk = gamma ? (double)gamma[c0 + k] : 1.0;
        const size_t q = (size_t)b * C + c0 + k;
        const double e1 = means[q * 2], e2 = means[q * 2 + 1];
        m1 += gk * e1;
        m2 += gk * (y_sums ? (double)mr[2 * q + 1] * (e2 - (double)mr[2 * q] * e1) : e2);
      }
      m1 /= cg; m2 /= cg;
    }
    float* cf = coef + (size_t)b * 3 * C;
    cf[c] = (float)(rstd * gm);
    cf[C + c] = (float)(-rstd * m1);
    cf[2 * C + c] = (float)(-rstd * m2);
    if (cd) {
      const double cc = gm != 0.0 ? -rstd * m2 / gm : 0.0, dd = -rstd * m1 - cc * (beta ? (double)beta[c] : 0.0);
      cd[((size_t)b * C + c) * 2] = (float)cc;
      cd[((size_t)b * C + c) * 2 + 1] = (float)dd;
    }
  }
}

extern "C" int w2s_norm_bwd_coef(int kind, const float* means, const float* mr, int B, int C, int G, const float* gamma, const float* beta, double L,
                                 float* coef, float* dgamma, float* dbeta, float* cd, int y_sums, void* stream) {
  if (kind < 0 || kind > 3 || !means || !mr || !coef || B <= 0 || C <= 0 || (kind == 3 && (G <= 0 || C % G))) return W2S_EINVAL;
  hipLaunchKernelGGL(norm_bwd_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), kind, means, mr, B, C,
                     kind == 3 ? G : C, gamma, beta, L, coef, dgamma, dbeta, cd, y_sums);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The residual join of a ConvBlock1D in one pass: out = act2(act(x * scale + shift) + add)  (blocks.py:68-70: conv3's norm + activation,
// `+ downsample(x)`, the block's activation) and its backward gs = g * act2'(act(x * scale + shift) + add) -- the gradient of BOTH addends.
// x = conv3's raw output; its activated form and the sum are never written (three passes forward instead of seven).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int BWD>
__global__ __launch_bounds__(256) void affine_act_join_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ x, int ldx,
                                                              const float* __restrict__ scale, const float* __restrict__ shift, int sample_stride,
                                                              const float* __restrict__ add, int ld_add, float* __restrict__ y, int ldy,
                                                              int rows_per_sample, long rows, int C, int act, int act2, float slope) {
  const int c4n = C >> 2;
  const long total = rows * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / c4n;
    const int c = (int)(i % c4n) * 4;
    f32x4 v = ld4(x + row * ldx + c);
    if (scale) {
      const long s = (row / rows_per_sample) * sample_stride + c;
      v = v * ld4(scale + s) + ld4(shift + s);
    }
    const f32x4 a = ld4(add + row * ld_add + c);
    f32x4 s = {act_f(v.x, act, slope) + a.x, act_f(v.y, act, slope) + a.y, act_f(v.z, act, slope) + a.z, act_f(v.w, act, slope) + a.w};
    if (BWD) {
      const f32x4 gv = ld4(g + row * ldg + c);
      s = (f32x4){gv.x * act_grad_f(s.x, act2, slope), gv.y * act_grad_f(s.y, act2, slope), gv.z * act_grad_f(s.z, act2, slope),
                  gv.w * act_grad_f(s.w, act2, slope)};
    } else {
      s = (f32x4){act_f(s.x, act2, slope), act_f(s.y, act2, slope), act_f(s.z, act2, slope), act_f(s.w, act2, slope)};
    }
    st4(y + row * ldy + c, s);
  }
}

static int launch_join(int bwd, const float* g, int ldg, const float* x, int ldx, const float* scale, const float* shift, int sample_stride,
                       const float* add, int ld_add, float* y, int ldy, int rows_per_sample, long rows, int C, int act, int act2, float slope,
                       void* stream) {
  if (!x || !add || !y || (bwd && !g) || rows <= 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldy & 3) || (ld_add & 3) || (ldg & 3) || rows_per_sample <= 0 ||
      act < 0 || act > 4 || act2 < 0 || act2 > 4 || (scale == nullptr) != (shift == nullptr) || (sample_stride & 3))
    return W2S_EINVAL;
  const long total = rows * (C >> 2);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (bwd)
    hipLaunchKernelGGL(affine_act_join_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, ldg, x, ldx, scale, shift,
                       sample_stride, add, ld_add, y, ldy, rows_per_sample, rows, C, act, act2, slope);
  else
    hipLaunchKernelGGL(affine_act_join_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, ldg, x, ldx, scale, shift,
                       sample_stride, add, ld_add, y, ldy, rows_per_sample, rows, C, act, act2, slope);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

extern "C" int w2s_affine_act_join(const float* x, int ldx, const float* scale, const float* shift, int sample_stride, const float* add, int ld_add,
                                   float* y, int ldy, int rows_per_sample, long rows, int C, int act, int act2, float slope, void* stream) {
  return launch_join(0, nullptr, 0, x, ldx, scale, shift, sample_stride, add, ld_add, y, ldy, rows_per_sample, rows, C, act, act2, slope, stream);
}

extern "C" int w2s_affine_act_join_bwd(const float* g, int ldg, const float* x, int ldx, const float* scale, const float* shift, int sample_stride,
                                       const float* add, int ld_add, float* gs, int ldgs, int rows_per_sample, long rows, int C, int act, int act2,
                                       float slope, void* stream) {
  return launch_join(1, g, ldg, x, ldx, scale, shift, sample_stride, add, ld_add, gs, ldgs, rows_per_sample, rows, C, act, act2, slope, stream);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Convolutions of the ONE-channel input (block 0's conv1 and its 1x1 / stride-2 residual conv; blocks.py:44-55 with input_dim = 1) on the
// vector ALU: 3 (or 1) multiply-adds per output element, the signal read once from L1 / L2 instead of as a 16-channel zero-padded copy
// (1.26 GB per pass at batch 16 x 10 h).  Forward with the per-(sample, tile) statistics partials of the fused conv epilogue (tile = 1024
// positions); weight gradient with the norm + activation backward formed on the fly (the W2S_PRO_AFFINE_BWD formula) when y is given.
// ---------------------------------------------------------------------------------------------------------------------------------
#define W2S_C1_TILE 1024
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ y, float* __restrict__ part, int L_in, int L_out, int C, int K, int stride,
                                                        int pad) {
  __shared__ float sm[2048];
  const int b = blockIdx.y, tl = blockIdx.x, nt = gridDim.x, tid = threadIdx.x;
  const int c4n = C >> 2, rpb = 256 / c4n, q = tid % c4n, rr = tid / c4n, c = q * 4;
  f32x4 a1 = splat4(0.f), a2 = splat4(0.f);
  if (rr < rpb) {
    f32x4 wk[3];
    for (int j = 0; j < 3; ++j)
      wk[j] = j < K ? (f32x4){w[(c + 0) * K + j], w[(c + 1) * K + j], w[(c + 2) * K + j], w[(c + 3) * K + j]} : splat4(0.f);
    const f32x4 bv = bias ? ld4(bias + c) : splat4(0.f);
    const float* xb = x + (size_t)b * L_in;
    const int t1 = min(L_out, (tl + 1) * W2S_C1_TILE);
    for (int t = tl * W2S_C1_TILE + rr; t < t1; t += rpb) {
      f32x4 v = bv;
      for (int j = 0; j < K; ++j) {
        const int gi = t * stride + j - pad;
        const float xv = (gi >= 0 && gi < L_in) ? xb[gi] : 0.f;
        v += wk[j] * xv;
      }
      st4(y + ((size_t)b * L_out + t) * C + c, v);
      a1 += v;
      a2 += v * v;
    }
    st4(sm + (rr * 2 + 0) * C + c, a1);
    st4(sm + (rr * 2 + 1) * C + c, a2);
  }
  if (!part) return;
  __syncthreads();
  for (int i = tid; i < 2 * C; i += 256) {
    float v = 0.f;
    for (int r = 0; r < rpb; ++r) v += sm[r * 2 * C + i];
    part[((size_t)b * nt + tl) * 2 * C + i] = v;
  }
}

extern "C" int w2s_conv1_fwd(const float* x, const float* w, const float* bias, float* y, float* part, int B, int L_in, int L_out, int C, int K,
                             int stride, int pad, void* stream) {
  if (!x || !w || !y || B <= 0 || L_in <= 0 || L_out <= 0 || C < 4 || C > 256 || (C & 3) || K < 1 || K > 3 || stride < 1 || pad < 0) return W2S_EINVAL;
  const int nt = (L_out + W2S_C1_TILE - 1) / W2S_C1_TILE;
  hipLaunchKernelGGL(conv1_fwd_kernel, dim3(nt, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, w, bias, y, part, L_in, L_out, C, K,
                     stride, pad);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// dW[o][j] = sum_{b,t} gy[b,t,o] x[b, t stride + j - pad]: part [B * ntiles][C][K] per-(sample, tile) sums (w2s_colsum finishes them).
// y2 != NULL: gy = (scale g) act'(z) + z c + d, z = y2 scale + shift, with ss / cd [B][C][2] = (scale, shift) / (c, d).
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ y2, const float* __restrict__ ss,
                                                          const float* __restrict__ cd, const float* __restrict__ x, float* __restrict__ part, int L_in,
                                                          int L_out, int C, int K, int stride, int pad, int act, float slope) {
  __shared__ float sm[3072];   // [rows in flight][K][C]
  const int b = blockIdx.y, tl = blockIdx.x, nt = gridDim.x, tid = threadIdx.x;
  const int c4n = C >> 2, rpb = 256 / c4n, q = tid % c4n, rr = tid / c4n, c = q * 4;
  f32x4 acc[3] = {splat4(0.f), splat4(0.f), splat4(0.f)};
  if (rr < rpb) {
    f32x4 sc = splat4(1.f), sh = splat4(0.f), cc = splat4(0.f), dd = splat4(0.f);
    if (y2) {
      const f32x4 s01 = ld4(ss + ((size_t)b * C + c) * 2), s23 = ld4(ss + ((size_t)b * C + c) * 2 + 4);
      const f32x4 c01 = ld4(cd + ((size_t)b * C + c) * 2), c23 = ld4(cd + ((size_t)b * C + c) * 2 + 4);
      sc = (f32x4){s01.x, s01.z, s23.x, s23.z}; sh = (f32x4){s01.y, s01.w, s23.y, s23.w};
      cc = (f32x4){c01.x, c01.z, c23.x, c23.z}; dd = (f32x4){c01.y, c01.w, c23.y, c23.w};
    }
    const float* xb = x + (size_t)b * L_in;
    const int t1 = min(L_out, (tl + 1) * W2S_C1_TILE);
    for (int t = tl * W2S_C1_TILE + rr; t < t1; t += rpb) {
      const size_t o = ((size_t)b * L_out + t) * C + c;
      f32x4 gv = ld4(g + o);
      if (y2) {
        const f32x4 z = ld4(y2 + o) * sc + sh;
        gv = (f32x4){sc.x * gv.x * act_grad_f(z.x, act, slope), sc.y * gv.y * act_grad_f(z.y, act, slope), sc.z * gv.z * act_grad_f(z.z, act, slope),
                     sc.w * gv.w * act_grad_f(z.w, act, slope)} + z * cc + dd;
      }
      for (int j = 0; j < K; ++j) {
        const int gi = t * stride + j - pad;
        const float xv = (gi >= 0 && gi < L_in) ? xb[gi] : 0.f;
        acc[j] += gv * xv;
      }
    }
    for (int j = 0; j < K; ++j) st4(sm + (rr * K + j) * C + c, acc[j]);
  }
  __syncthreads();
  for (int i = tid; i < K * C; i += 256) {   // i = j * C + channel
    float v = 0.f;
    for (int r = 0; r < rpb; ++r) v += sm[r * K * C + i];
    const int j = i / C, ch = i % C;
    part[((size_t)b * nt + tl) * C * K + ch * K + j] = v;
  }
}

extern "C" int w2s_conv1_wgrad_parts(int B, int L_out) { return B * ((L_out + W2S_C1_TILE - 1) / W2S_C1_TILE); }

extern "C" int w2s_conv1_wgrad(const float* g, const float* y2, const float* ss, const float* cd, const float* x, float* part, int B, int L_in, int L_out,
                               int C, int K, int stride, int pad, int act, void* stream) {
  if (!g || !x || !part || B <= 0 || L_in <= 0 || L_out <= 0 || C < 4 || C > 256 || (C & 3) || K < 1 || K > 3 || stride < 1 || pad < 0 || act < 0 ||
      act > 4 || (y2 && (!ss || !cd)))
    return W2S_EINVAL;
  const int nt = (L_out + W2S_C1_TILE - 1) / W2S_C1_TILE;
  hipLaunchKernelGGL(conv1_wgrad_kernel, dim3(nt, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, y2, ss, cd, x, part, L_in, L_out, C, K,
                     stride, pad, act, 0.01f);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
