// Classifier head, masked cross-entropy (+ confusion matrix), and the fused clip + AdamW step.
#include "w2s_common.h"

#define W2S_MAXC 8  // num_classes <= 8 (reference uses 4 or 5)

// logits[row][c] = sum_f W[c][f] * (gelu_in ? GELU(pre[row][f]) : pre[row][f]) + b[c]   (wav2sleep.py:41,66)
// 16 lanes per row (each 8 features for F = 128), shuffle reduction.
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ pre, int ld, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ logits, int rows, int F, int nc,
                                                       int gelu_in) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t row = gid >> 4;
  const int e = threadIdx.x & 15;
  if (row >= (size_t)rows) return;
  float acc[W2S_MAXC];
#pragma unroll
  for (int c = 0; c < W2S_MAXC; ++c) acc[c] = 0.f;
  for (int f = e * 4; f < F; f += 64) {
    f32x4 v = ld4(pre + row * ld + f);
    if (gelu_in) v = gelu4(v);
#pragma unroll
    for (int c = 0; c < W2S_MAXC; ++c)
      if (c < nc) {
        f32x4 wv = ld4(w + (size_t)c * F + f);
        acc[c] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
      }
  }
#pragma unroll
  for (int c = 0; c < W2S_MAXC; ++c)
    if (c < nc) {
      const float s = row16_sum(acc[c]);
      if (e == 0) logits[row * nc + c] = s + bias[c];
    }
}

extern "C" int w2s_head_fwd(const float* pre, int ld, const float* w, const float* bias, float* logits, int rows, int F, int nc, int gelu_in,
                            void* stream) {
  if (!pre || !w || !bias || !logits || nc > W2S_MAXC || nc <= 0 || (F & 3) || (ld & 3)) return W2S_EINVAL;
  const unsigned blocks = (unsigned)(((size_t)rows * 16 + 255) / 256);
  hipLaunchKernelGGL(head_fwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pre, ld, w, bias, logits, rows, F,
                     nc, gelu_in);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// CrossEntropyLoss(reduction=mean, ignore_index=-1) (training/main.yaml:41-46, trainer/main.py:162-163):
// pass 1: per-block partial (sum nll, valid count) + confusion counts (rows = true, cols = argmax; integer atomics
// are order-independent);  pass 2 (1 block): loss = sum/count, written to out[0], count to out[1];
// pass 3: glogits = (softmax - onehot) / count for valid rows, 0 otherwise.
// A label outside {-1, 0 .. nc-1} (torch raises "Target out of bounds" there) touches no memory it must not: it poisons the loss
// (NaN in out[0], NaN gradient row), which is how a launch that cannot return an error code fails loudly.
__global__ __launch_bounds__(256) void ce_partial_kernel(const float* __restrict__ logits, const float* __restrict__ labels, int rows, int nc,
                                                         float* __restrict__ part, long long* __restrict__ cmat) {
  __shared__ float red[2][256];
  __shared__ unsigned int hist[W2S_MAXC * W2S_MAXC];   // confusion counts of this block: one global atomic per bin and block instead of
  if (threadIdx.x < W2S_MAXC * W2S_MAXC) hist[threadIdx.x] = 0u;   // one per row (15 360 atomics on 16 addresses took 59 us)
  __syncthreads();
  const int row = blockIdx.x * 256 + threadIdx.x;
  float nll = 0.f, cnt = 0.f;
  if (row < rows) {
    const float yf = labels[row];
    const int y = (yf >= -1.f && yf < (float)nc) ? (int)yf : nc;   // nc = out of range (NaN included)
    float mx = -INFINITY;
    int am = 0;
    for (int c = 0; c < nc; ++c) {
      const float v = logits[(size_t)row * nc + c];
      if (v > mx) { mx = v; am = c; }
    }
    if (y == nc) {
      nll = NAN;
      cnt = 1.f;
    } else if (y >= 0) {
      float den = 0.f;
      for (int c = 0; c < nc; ++c) den += expf(logits[(size_t)row * nc + c] - mx);
      nll = logf(den) + mx - logits[(size_t)row * nc + y];
      cnt = 1.f;
      if (cmat) atomicAdd(&hist[y * nc + am], 1u);
    }
  }
  red[0][threadIdx.x] = nll;
  red[1][threadIdx.x] = cnt;
  __syncthreads();
  if (cmat && threadIdx.x < nc * nc && hist[threadIdx.x])
    atomicAdd(reinterpret_cast<unsigned long long*>(cmat + threadIdx.x), (unsigned long long)hist[threadIdx.x]);
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[2 * blockIdx.x] = red[0][0]; part[2 * blockIdx.x + 1] = red[1][0]; }
}
__global__ void ce_final_kernel(const float* __restrict__ part, int nblocks, float* __restrict__ out) {
  double s = 0.0, c = 0.0;
  for (int i = 0; i < nblocks; ++i) { s += (double)part[2 * i]; c += (double)part[2 * i + 1]; }
  out[0] = (float)(s / c);  // all labels ignored => 0/0 = NaN, as torch
  out[1] = (float)c;
}
__global__ __launch_bounds__(256) void ce_grad_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                      const float* __restrict__ count, float* __restrict__ glogits, int rows, int nc,
                                                      float gscale) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const float yf = labels[row];
  const int y = (yf >= -1.f && yf < (float)nc) ? (int)yf : nc;
  const float inv = (y == nc) ? NAN : gscale / count[0];
  float mx = -INFINITY;
  for (int c = 0; c < nc; ++c) mx = fmaxf(mx, logits[(size_t)row * nc + c]);
  float den = 0.f;
  for (int c = 0; c < nc; ++c) den += expf(logits[(size_t)row * nc + c] - mx);
  for (int c = 0; c < nc; ++c) {
    float gv = 0.f;
    if (y >= 0) gv = (expf(logits[(size_t)row * nc + c] - mx) / den - (c == y ? 1.f : 0.f)) * inv;
    glogits[(size_t)row * nc + c] = gv;
  }
}

extern "C" int w2s_ce_fwd_bwd(const float* logits, const float* labels, int rows, int nc, float* part, float* loss_out, float* glogits,
                              long long* cmat, float gscale, void* stream) {
  if (!logits || !labels || !part || !loss_out || nc <= 0 || nc > W2S_MAXC) return W2S_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nb = (rows + 255) / 256;
  hipLaunchKernelGGL(ce_partial_kernel, dim3(nb), dim3(256), 0, s, logits, labels, rows, nc, part, cmat);
  hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(1), 0, s, part, nb, loss_out);
  if (glogits) hipLaunchKernelGGL(ce_grad_kernel, dim3(nb), dim3(256), 0, s, logits, labels, loss_out + 1, glogits, rows, nc, gscale);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// The same loss over a batch that arrives in sample waves (trainer.FusedTrainStep, W2S_WAVES): the number of counted labels is a function
// of the labels alone, so it is known before the first wave's logits exist (w2s_ce_count); every wave then runs passes 1 and 3 on its rows
// (w2s_ce_wave: partials into its slice of `part`, gradient scaled by the WHOLE batch's count), and pass 2 runs once over all partials.
__global__ __launch_bounds__(256) void ce_count_kernel(const float* __restrict__ labels, int rows, int nc, float* __restrict__ count) {
  __shared__ int red[256];
  int n = 0;
  for (int row = threadIdx.x; row < rows; row += 256) {
    const float yf = labels[row];
    const int y = (yf >= -1.f && yf < (float)nc) ? (int)yf : nc;
    n += (y != -1);
  }
  red[threadIdx.x] = n;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) count[0] = (float)red[0];
}
extern "C" int w2s_ce_count(const float* labels, int rows, int nc, float* count, void* stream) {
  if (!labels || !count || rows < 0 || nc <= 0 || nc > W2S_MAXC) return W2S_EINVAL;
  hipLaunchKernelGGL(ce_count_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), labels, rows, nc, count);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
extern "C" int w2s_ce_wave(const float* logits, const float* labels, int rows, int nc, float* part, const float* count, float* glogits,
                           long long* cmat, float gscale, void* stream) {
  if (!logits || !labels || !part || !count || !glogits || nc <= 0 || nc > W2S_MAXC) return W2S_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nb = (rows + 255) / 256;
  hipLaunchKernelGGL(ce_partial_kernel, dim3(nb), dim3(256), 0, s, logits, labels, rows, nc, part, cmat);
  hipLaunchKernelGGL(ce_grad_kernel, dim3(nb), dim3(256), 0, s, logits, labels, count, glogits, rows, nc, gscale);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
extern "C" int w2s_ce_final(const float* part, int nblocks, float* loss_out, void* stream) {
  if (!part || !loss_out || nblocks <= 0) return W2S_EINVAL;
  hipLaunchKernelGGL(ce_final_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), part, nblocks, loss_out);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// Head backward: gfeat[row][f] = sum_c glogits[row][c] * W[c][f];  gpre = gelu_in ? gfeat * GELU'(pre) : gfeat.
// Partials of dW[c][f] = sum_rows glogits[row][c]*feat[row][f] and db[c] per block of `rpb` rows:
// part[blk][nc*F + nc].  One thread per feature f.
__global__ void head_bwd_kernel(const float* __restrict__ pre, int ld, const float* __restrict__ w, const float* __restrict__ glogits,
                                float* __restrict__ gpre, int ldg, float* __restrict__ part, int rows, int F, int nc, int gelu_in, int rpb) {
  const int f = threadIdx.x;  // blockDim.x == F
  const int r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
  float wf[W2S_MAXC], dw[W2S_MAXC], db[W2S_MAXC];
#pragma unroll
  for (int c = 0; c < W2S_MAXC; ++c) { wf[c] = (c < nc) ? w[(size_t)c * F + f] : 0.f; dw[c] = 0.f; db[c] = 0.f; }
  // eight rows in flight (one dependent load pair per iteration left this kernel at 41 us for 8 MB: 64 L2 round trips in a row);
  // the sums run over the rows in the same order as the plain loop
  constexpr int U = 8;
  for (int rb = r0; rb < r1; rb += U) {
    float pv[U], gl[U][W2S_MAXC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = min(rb + u, r1 - 1);
      pv[u] = pre[(size_t)r * ld + f];
#pragma unroll
      for (int c = 0; c < W2S_MAXC; ++c) gl[u][c] = (c < nc) ? glogits[(size_t)r * nc + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = rb + u;
      if (r >= r1) break;
      const float p = pv[u];
      const float feat = gelu_in ? gelu_f(p) : p;
      float gf = 0.f;
#pragma unroll
      for (int c = 0; c < W2S_MAXC; ++c)
        if (c < nc) {
          gf += gl[u][c] * wf[c];
          dw[c] += gl[u][c] * feat;
          db[c] += gl[u][c];
        }
      gpre[(size_t)r * ldg + f] = gelu_in ? gf * gelu_grad_f(p) : gf;
    }
  }
  float* out = part + (size_t)blockIdx.x * (nc * F + nc);
#pragma unroll
  for (int c = 0; c < W2S_MAXC; ++c)
    if (c < nc) {
      out[c * F + f] = dw[c];
      if (f == 0) out[nc * F + c] = db[c];
    }
}

extern "C" int w2s_head_bwd(const float* pre, int ld, const float* w, const float* glogits, float* gpre, int ldg, float* part, int nparts,
                            int rows, int F, int nc, int gelu_in, void* stream) {
  if (!pre || !w || !glogits || !gpre || !part || nparts <= 0 || nc > W2S_MAXC || F > 1024) return W2S_EINVAL;
  const int rpb = (rows + nparts - 1) / nparts;
  hipLaunchKernelGGL(head_bwd_kernel, dim3(nparts), dim3(F), 0, reinterpret_cast<hipStream_t>(stream), pre, ld, w, glogits, gpre, ldg, part,
                     rows, F, nc, gelu_in, rpb);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// ------------------------------------------------------------------------------------------------
// Optimiser on the FLAT parameter / gradient buffers (one launch for all 183 tensors):
//   1. sumsq partials (fixed order)  2. total norm + clip coefficient  3. AdamW with the coefficient folded in.
// torch.nn.utils.clip_grad_norm_(max_norm) + torch.optim.AdamW (trainer/main.py:273-275, training/main.yaml:21-22).
// hyper (device): [lr, weight_decay, beta1, beta2, eps, 1-beta1^t, 1-beta2^t, max_norm]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ part) {
  __shared__ double red[256];
  double s = 0.0;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 v = ld4(g + 4 * i);
    s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += (double)v * v; }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = (float)red[0];
}
// out[0] = total L2 norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)); world>1: part holds all ranks' sums already
__global__ void clip_coef_kernel(const float* __restrict__ part, int nparts, const float* __restrict__ hyper, float* __restrict__ out) {
  double s = 0.0;
  for (int i = 0; i < nparts; ++i) s += (double)part[i];
  const float norm = (float)sqrt(s);
  float coef = hyper[7] / (norm + 1e-6f);
  if (coef > 1.0f) coef = 1.0f;
  if (hyper[7] <= 0.f) coef = 1.0f;
  out[0] = norm;
  out[1] = coef;
}
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, const float* __restrict__ hyper,
                                                    const float* __restrict__ normcoef) {
  const float lr = hyper[0], wd = hyper[1], b1 = hyper[2], b2 = hyper[3], eps = hyper[4], bc1 = hyper[5], bc2 = hyper[6];
  const float coef = normcoef[1];
  const float step_size = lr / bc1, rs_bc2 = 1.0f / sqrtf(bc2);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    const float denom = sqrtf(vi) * rs_bc2 + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

extern "C" int w2s_sumsq_partial(const float* g, long n, float* part, int nparts, void* stream) {
  if (!g || !part || n <= 0 || nparts <= 0) return W2S_EINVAL;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nparts), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, (size_t)n, part);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
extern "C" int w2s_clip_coef(const float* part, int nparts, const float* hyper, float* normcoef, void* stream) {
  if (!part || !hyper || !normcoef) return W2S_EINVAL;
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), part, nparts, hyper, normcoef);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
extern "C" int w2s_adamw(float* p, const float* g, float* m, float* v, long n, const float* hyper, const float* normcoef, void* stream) {
  if (!p || !g || !m || !v || !hyper || !normcoef || n <= 0) return W2S_EINVAL;
  size_t blocks = ((size_t)n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p, g, m, v, (size_t)n, hyper,
                     normcoef);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// ---- exponential moving average of the weights on the flat buffer (trainer/callbacks.py:61-64) and the in-place exchange
//      used to evaluate with the averaged weights (callbacks.py:84-98): one launch each for all 183 tensors
__global__ __launch_bounds__(256) void ema_update_kernel(float* __restrict__ ema, const float* __restrict__ p, size_t n, float decay) {
  const float a = 1.0f - decay;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float e = ema[i] * decay;   // ema.mul_(decay).add_(p, alpha=1-decay): two rounded steps, like the reference
    e = e + a * p[i];
    ema[i] = e;
  }
}
__global__ __launch_bounds__(256) void swap_kernel(float* __restrict__ a, float* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float t = a[i];
    a[i] = b[i];
    b[i] = t;
  }
}
extern "C" int w2s_ema_update(float* ema, const float* p, long n, float decay, void* stream) {
  if (!ema || !p || n <= 0 || !(decay >= 0.0f && decay <= 1.0f)) return W2S_EINVAL;
  size_t blocks = ((size_t)n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ema, p, (size_t)n, decay);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
extern "C" int w2s_swap(float* a, float* b, long n, void* stream) {
  if (!a || !b || n <= 0) return W2S_EINVAL;
  size_t blocks = ((size_t)n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(swap_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, b, (size_t)n);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

extern "C" const char* w2s_version(void) { return "w2s-hip 0.5 (gfx950; bf16x3 split-precision and fp32 MFMA)"; }
extern "C" int w2s_abi_version(void) { return W2S_ABI_VERSION; }
