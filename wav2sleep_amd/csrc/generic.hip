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
                                                               float* __restrict__ out, long N, int D, int H, int hd, float scale) {
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
    float* o = out + (n * D + qi) * F + h * hd;
    for (int e = 0; e < hd; ++e) {
      float a = 0.f;
      for (int j = 0; j < D; ++j) a += sc[j] * base[(long)j * 3 * F + 2 * F + h * hd + e];
      o[e] = a * inv;
    }
  }
}

extern "C" int w2s_attn_generic_fwd(const float* qkv, const unsigned char* keypad, float* out, long N, int D, int H, int hd, void* stream) {
  if (!qkv || !keypad || !out || N <= 0 || D <= 0 || D > 16 || H <= 0 || hd <= 0) return W2S_EINVAL;
  const long total = N * H * D;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(attn_generic_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), qkv, keypad, out, N,
                     D, H, hd, 1.0f / sqrtf((float)hd));
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
