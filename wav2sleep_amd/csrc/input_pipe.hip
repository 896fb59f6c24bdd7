// Device-side input pipeline ("next" row f-1): what the reference does on the CPU in
// ParquetDataset.__getitem__ (data/dataset.py:76-87,132-183) and in on_after_batch_transfer (trainer/main.py:131-138,
// 342-353; trainer/masker.py:49-50), as three bandwidth-bound kernels over the raw [rows][T] signals:
//   z-score per recording (mean, unbiased std, eps 1e-6, skipped when the row holds a non-finite value),
//   polarity flip + modality masking (-inf rows) in one pass, and the 5 -> 4/5 class label map.
#include "w2s_common.h"

// part[row][blk] = {sum, sumsq, nonfinite count} in fp64 over the block's chunk
__global__ __launch_bounds__(256) void rowstats_kernel(const float* __restrict__ x, long T, int nblk, double* __restrict__ part) {
  __shared__ double red[3][256];
  const int row = blockIdx.y, blk = blockIdx.x;
  const long per = (T + nblk - 1) / nblk;
  const long t0 = (long)blk * per, t1 = (t0 + per < T) ? t0 + per : T;
  const float* xr = x + (size_t)row * T;
  double s = 0.0, q = 0.0, bad = 0.0;
  for (long t = t0 + threadIdx.x; t < t1; t += 256) {
    const float v = xr[t];
    if (isfinite(v)) { s += v; q += (double)v * v; } else bad += 1.0;
  }
  red[0][threadIdx.x] = s; red[1][threadIdx.x] = q; red[2][threadIdx.x] = bad;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) {
      red[0][threadIdx.x] += red[0][threadIdx.x + k];
      red[1][threadIdx.x] += red[1][threadIdx.x + k];
      red[2][threadIdx.x] += red[2][threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* p = part + ((size_t)row * nblk + blk) * 3;
    p[0] = red[0][0]; p[1] = red[1][0]; p[2] = red[2][0];
  }
}

// y = (x - mu) / max(std, eps) per row, std unbiased (torch.std); rows with any non-finite value are copied unchanged
__global__ __launch_bounds__(256) void zscore_apply_kernel(const float* __restrict__ x, float* __restrict__ y, long T, int nblk,
                                                           const double* __restrict__ part, float eps, float* __restrict__ stats_out) {
  __shared__ float sh[2];
  const int row = blockIdx.y;
  if (threadIdx.x == 0) {
    double s = 0.0, q = 0.0, bad = 0.0;
    for (int b = 0; b < nblk; ++b) {
      const double* p = part + ((size_t)row * nblk + b) * 3;
      s += p[0]; q += p[1]; bad += p[2];
    }
    float mu = 0.f, inv = 1.f;
    if (bad == 0.0 && T > 0) {
      const double mean = s / (double)T;
      double var = (T > 1) ? (q - (double)T * mean * mean) / (double)(T - 1) : 0.0;
      if (var < 0.0) var = 0.0;
      float sd = (float)sqrt(var);
      if (!(sd > eps)) sd = eps;
      mu = (float)mean;
      inv = sd;
    }
    sh[0] = mu; sh[1] = inv;
    if (stats_out && blockIdx.x == 0) { stats_out[2 * row] = mu; stats_out[2 * row + 1] = inv; }
  }
  __syncthreads();
  const float mu = sh[0], sd = sh[1];
  const float* xr = x + (size_t)row * T;
  float* yr = y + (size_t)row * T;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < T; t += (long)gridDim.x * 256) yr[t] = (xr[t] - mu) / sd;
}

extern "C" int w2s_zscore(const float* x, float* y, int rows, long T, double* part, int nblk, float eps, float* stats_out, void* stream) {
  if (!x || !y || !part || rows <= 0 || T <= 0 || nblk <= 0) return W2S_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(rowstats_kernel, dim3(nblk, rows), dim3(256), 0, s, x, T, nblk, part);
  hipLaunchKernelGGL(zscore_apply_kernel, dim3(nblk, rows), dim3(256), 0, s, x, y, T, nblk, part, eps, stats_out);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// in place: x[b, :] = keep[b] ? x[b, :] * sign[b] : -inf      (invert_signals + SignalMasker's write)
__global__ __launch_bounds__(256) void augment_kernel(float* __restrict__ x, long T, const float* __restrict__ sign,
                                                      const uint8_t* __restrict__ keep) {
  const int b = blockIdx.y;
  const float sg = sign ? sign[b] : 1.0f;
  const bool kp = keep ? keep[b] != 0 : true;
  if (kp && sg == 1.0f) return;
  float* xr = x + (size_t)b * T;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < T; t += (long)gridDim.x * 256) xr[t] = kp ? xr[t] * sg : -INFINITY;
}
extern "C" int w2s_augment(float* x, int B, long T, const float* sign, const uint8_t* keep, void* stream) {
  if (!x || B <= 0 || T <= 0) return W2S_EINVAL;
  long blocks = (T + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(augment_kernel, dim3((unsigned)blocks, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, T, sign, keep);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// dst[i] = map[(int)src[i]] for src in 0..4, else -1 (NaN / unscored)   (settings.py:52-56, dataset.py:174-182)
__global__ void map_labels_kernel(const float* __restrict__ src, float* __restrict__ dst, long n, int m0, int m1, int m2, int m3, int m4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = src[i];
  const int map[5] = {m0, m1, m2, m3, m4};
  float o = -1.0f;
  if (v >= 0.0f && v <= 4.0f && v == floorf(v)) o = (float)map[(int)v];
  dst[i] = o;
}
extern "C" int w2s_map_labels(const float* src, float* dst, long n, int num_classes, void* stream) {
  if (!src || !dst || n <= 0 || (num_classes != 4 && num_classes != 5)) return W2S_EINVAL;
  const int m[5] = {0, 1, num_classes == 4 ? 1 : 2, num_classes == 4 ? 2 : 3, num_classes == 4 ? 3 : 4};
  hipLaunchKernelGGL(map_labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, n,
                     m[0], m[1], m[2], m[3], m[4]);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
