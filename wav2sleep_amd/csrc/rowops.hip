// Row-wise and elementwise kernels of the set-fusion transformer and the SequenceCNN (tensors here are
// ~100x smaller than the encoder's, so these stay simple and materialised): LayerNorm fwd/bwd (also the
// channel-LayerNorm of models/utils.py:17-23, which in channels-last layout IS a row LayerNorm), column sums
// (bias / gamma / beta / CLS gradients), elementwise GELU / residual / dropout, the D<=8-token attention core.
#include "w2s_common.h"

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim (C <= 512, multiple of 64); one wave per row; rstat[row] = (mean, rstd)
// ------------------------------------------------------------------------------------------------
template <int NPL>  // elements per lane = C / 64
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y, int ldy,
                                                            float* __restrict__ rstat, int rows, float eps, int gelu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int C = NPL * 64;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    float v[NPL];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) { v[k] = x[(size_t)row * ldx + lane + 64 * k]; s += v[k]; }
    const float mean = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) { const float d = v[k] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int c = lane + 64 * k;
      float o = gamma[c] * ((v[k] - mean) * rstd) + beta[c];
      if (gelu) o = gelu_f(o);
      y[(size_t)row * ldy + c] = o;
    }
    if (lane == 0 && rstat) { rstat[2 * (size_t)row] = mean; rstat[2 * (size_t)row + 1] = rstd; }
  }
}

// C = 128: 32 lanes x float4 per row, two rows per wave and step, U = 4 steps (8 rows per wave, 32 per block) of loads in flight -- the
// forward counterpart of layernorm_bwd128_kernel below.  The one-row-per-wave form above walks 19 rows per wave one round trip after the
// other on the 76 800-token tensors (28 us for 79 MB) and 4 on the 15 360-row SequenceCNN tensors (7 us for 16 MB).
__global__ __launch_bounds__(256) void layernorm_fwd128_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ y, int ldy,
                                                               float* __restrict__ rstat, int rows, float eps, int gelu) {
  constexpr int C = 128, U = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 5, c0 = (lane & 31) * 4;
  const f32x4 gm = ld4(gamma + c0), bt = ld4(beta + c0);
  for (int base = blockIdx.x * (8 * U); base < rows; base += gridDim.x * (8 * U)) {
    f32x4 xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = base + (u * 4 + wave) * 2 + sub;
      xv[u] = (row < rows) ? ld4(x + (size_t)row * ldx + c0) : (f32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = base + (u * 4 + wave) * 2 + sub;
      float sm = (xv[u].x + xv[u].y) + (xv[u].z + xv[u].w);
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) sm += __shfl_xor(sm, m);
      const float mean = sm * (1.0f / C);
      const f32x4 d = xv[u] - mean;
      float q = (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) q += __shfl_xor(q, m);
      const float rstd = 1.0f / sqrtf(q * (1.0f / C) + eps);
      f32x4 o = gm * (d * rstd) + bt;
      if (gelu) o = gelu4(o);
      if (row < rows) {
        st4(y + (size_t)row * ldy + c0, o);
        if ((lane & 31) == 0 && rstat) { rstat[2 * (size_t)row] = mean; rstat[2 * (size_t)row + 1] = rstd; }
      }
    }
  }
}

extern "C" int w2s_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* rstat, int rows,
                                 int C, float eps, int gelu, void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0) return W2S_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int grid = rows < 4096 ? (rows + 3) / 4 : 1024;
  const bool vec = !((ldx | ldy) & 3) && !(((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15);
  if (C == 128 && vec) {
    const int g128 = (rows + 31) / 32 < 1024 ? (rows + 31) / 32 : 1024;
    hipLaunchKernelGGL(layernorm_fwd128_kernel, dim3(g128), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rstat, rows, eps, gelu);
  } else if (C == 128) hipLaunchKernelGGL(layernorm_fwd_kernel<2>, dim3(grid), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rstat, rows, eps, gelu);
  else if (C == 64) hipLaunchKernelGGL(layernorm_fwd_kernel<1>, dim3(grid), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rstat, rows, eps, gelu);
  else if (C == 256) hipLaunchKernelGGL(layernorm_fwd_kernel<4>, dim3(grid), dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rstat, rows, eps, gelu);
  else return W2S_EINVAL;
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// gx = [gadd +] rstd*(gamma*gn - mean_c(gamma*gn) - xhat*mean_c(gamma*gn*xhat)),  gn = gelu ? g*GELU'(n) : g
// per-block partial sums: part_gamma[blk][c] = sum_rows gn*xhat, part_beta[blk][c] = sum_rows gn
template <int NPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ x, int ldx,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ rstat, const float* __restrict__ gadd,
                                                            float* __restrict__ gx, int ldgx, float* __restrict__ part_gamma,
                                                            float* __restrict__ part_beta, int rows, int gelu) {
  constexpr int C = NPL * 64;
  __shared__ float red[2][4][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dg[NPL], db[NPL], gm[NPL], bt[NPL];
#pragma unroll
  for (int k = 0; k < NPL; ++k) { dg[k] = 0.f; db[k] = 0.f; gm[k] = gamma[lane + 64 * k]; bt[k] = beta[lane + 64 * k]; }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const float mean = rstat[2 * (size_t)row], rstd = rstat[2 * (size_t)row + 1];
    float xh[NPL], gn[NPL];
    float a = 0.f, bsum = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int c = lane + 64 * k;
      xh[k] = (x[(size_t)row * ldx + c] - mean) * rstd;
      float gv = g[(size_t)row * ldg + c];
      if (gelu) gv *= gelu_grad_f(gm[k] * xh[k] + bt[k]);
      gn[k] = gv;
      dg[k] += gv * xh[k];
      db[k] += gv;
      a += gm[k] * gv;
      bsum += gm[k] * gv * xh[k];
    }
    a = wave_sum(a) * (1.0f / C);
    bsum = wave_sum(bsum) * (1.0f / C);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const int c = lane + 64 * k;
      float o = rstd * (gm[k] * gn[k] - a - xh[k] * bsum);
      if (gadd) o += gadd[(size_t)row * ldgx + c];
      gx[(size_t)row * ldgx + c] = o;
    }
  }
#pragma unroll
  for (int k = 0; k < NPL; ++k) { red[0][wave][lane + 64 * k] = dg[k]; red[1][wave][lane + 64 * k] = db[k]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    part_gamma[(size_t)blockIdx.x * C + c] = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    part_beta[(size_t)blockIdx.x * C + c] = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
  }
}

// C = 128: 32 lanes x float4 per row; a wave covers 2 rows per step and keeps U = 4 steps (8 rows) of loads in flight
// (the one-row-per-wave form above is latency-bound: 75 dependent row round trips per wave on the 76 800-token tensors).
__global__ __launch_bounds__(256) void layernorm_bwd128_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ x, int ldx,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ rstat, const float* __restrict__ gadd,
                                                               float* __restrict__ gx, int ldgx, float* __restrict__ part_gamma,
                                                               float* __restrict__ part_beta, int rows, int gelu) {
  constexpr int C = 128, U = 4;
  __shared__ float red[2][8][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 5, c0 = (lane & 31) * 4;
  const f32x4 gm = ld4(gamma + c0), bt = ld4(beta + c0);
  f32x4 dg = {0, 0, 0, 0}, db = {0, 0, 0, 0};
  for (int base = blockIdx.x * (8 * U); base < rows; base += gridDim.x * (8 * U)) {
    f32x4 xv[U], gv[U], ga[U];
    float mean[U], rstd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = base + (u * 4 + wave) * 2 + sub;
      const bool ok = row < rows;
      xv[u] = ok ? ld4(x + (size_t)row * ldx + c0) : (f32x4){0, 0, 0, 0};
      gv[u] = ok ? ld4(g + (size_t)row * ldg + c0) : (f32x4){0, 0, 0, 0};
      ga[u] = (ok && gadd) ? ld4(gadd + (size_t)row * ldgx + c0) : (f32x4){0, 0, 0, 0};
      mean[u] = ok ? rstat[2 * (size_t)row] : 0.f;
      rstd[u] = ok ? rstat[2 * (size_t)row + 1] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = base + (u * 4 + wave) * 2 + sub;
      const f32x4 xh = (xv[u] - mean[u]) * rstd[u];
      f32x4 gn = gv[u];
      if (gelu) gn = gn * gelu_grad4(gm * xh + bt);
      dg += gn * xh;
      db += gn;
      const f32x4 t = gm * gn, tx = t * xh;
      float a = (t.x + t.y) + (t.z + t.w), bs = (tx.x + tx.y) + (tx.z + tx.w);
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) { a += __shfl_xor(a, m); bs += __shfl_xor(bs, m); }
      a *= (1.0f / C); bs *= (1.0f / C);
      const f32x4 o = rstd[u] * (t - a - xh * bs) + ga[u];
      if (row < rows) st4(gx + (size_t)row * ldgx + c0, o);
    }
  }
  st4(&red[0][wave * 2 + sub][c0], dg);
  st4(&red[1][wave * 2 + sub][c0], db);
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    float s0 = red[0][0][c], s1 = red[1][0][c];
#pragma unroll
    for (int k = 1; k < 8; ++k) { s0 += red[0][k][c]; s1 += red[1][k][c]; }
    part_gamma[(size_t)blockIdx.x * C + c] = s0;
    part_beta[(size_t)blockIdx.x * C + c] = s1;
  }
}

extern "C" int w2s_layernorm_bwd(const float* g, int ldg, const float* x, int ldx, const float* gamma, const float* beta, const float* rstat,
                                 const float* gadd, float* gx, int ldgx, float* part_gamma, float* part_beta, int rows, int C, int gelu,
                                 int nparts, void* stream) {
  if (!g || !x || !gamma || !beta || !rstat || !gx || !part_gamma || !part_beta || nparts <= 0) return W2S_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool vec = !((ldg | ldx | ldgx) & 3) && !(((uintptr_t)g | (uintptr_t)x | (uintptr_t)gx | (uintptr_t)gadd | (uintptr_t)gamma | (uintptr_t)beta) & 15);
  if (C == 128 && vec) hipLaunchKernelGGL(layernorm_bwd128_kernel, dim3(nparts), dim3(256), 0, s, g, ldg, x, ldx, gamma, beta, rstat, gadd, gx, ldgx, part_gamma, part_beta, rows, gelu);
  else if (C == 128) hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(nparts), dim3(256), 0, s, g, ldg, x, ldx, gamma, beta, rstat, gadd, gx, ldgx, part_gamma, part_beta, rows, gelu);
  else if (C == 64) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, dim3(nparts), dim3(256), 0, s, g, ldg, x, ldx, gamma, beta, rstat, gadd, gx, ldgx, part_gamma, part_beta, rows, gelu);
  else if (C == 256) hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(nparts), dim3(256), 0, s, g, ldg, x, ldx, gamma, beta, rstat, gadd, gx, ldgx, part_gamma, part_beta, rows, gelu);
  else return W2S_EINVAL;
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// ------------------------------------------------------------------------------------------------
// column sums: part[blk][c] = sum over the block's rows of g[row][c]; then out[c] (+)= sum_p part[p][c]
// ------------------------------------------------------------------------------------------------
// 256 threads = 32 column lanes x 8 row lanes; fixed-order LDS tree
__global__ __launch_bounds__(256) void colsum_rows_kernel(const float* __restrict__ g, int rows, int C, int ldg, float* __restrict__ part) {
  __shared__ float red[8][33];
  const int per = (rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  for (int c0 = 0; c0 < C; c0 += 32) {
    const int c = c0 + cl;
    float s = 0.f;
    if (c < C)
      for (int r = r0 + rl; r < r1; r += 8) s += g[(size_t)r * ldg + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
      for (int k = 1; k < 8; ++k) s += red[k][cl];
      part[(size_t)blockIdx.x * C + c] = s;
    }
    __syncthreads();
  }
}
// float4 form: 16 column lanes (64 columns) x 16 row lanes per pass, 4 rows of loads in flight per thread
__global__ __launch_bounds__(256) void colsum_rows4_kernel(const float* __restrict__ g, int rows, int C, int ldg, float* __restrict__ part) {
  __shared__ float red[16][68];
  const int per = (rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + cl * 4;
    f32x4 s = {0, 0, 0, 0};
    if (c < C) {
      int r = r0 + rl;
      for (; r + 48 < r1; r += 64) {
        const f32x4 a0 = ld4(g + (size_t)r * ldg + c), a1 = ld4(g + (size_t)(r + 16) * ldg + c);
        const f32x4 a2 = ld4(g + (size_t)(r + 32) * ldg + c), a3 = ld4(g + (size_t)(r + 48) * ldg + c);
        s += (a0 + a1) + (a2 + a3);
      }
      for (; r < r1; r += 16) s += ld4(g + (size_t)r * ldg + c);
    }
    st4(&red[rl][cl * 4], s);
    __syncthreads();
    if (threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
      float t = red[0][threadIdx.x];
#pragma unroll
      for (int k = 1; k < 16; ++k) t += red[k][threadIdx.x];
      part[(size_t)blockIdx.x * C + c0 + threadIdx.x] = t;
    }
    __syncthreads();
  }
}
extern "C" int w2s_bias_grad(const float* g, int rows, int C, int ldg, float* part, int nparts, void* stream) {
  if (!g || !part || nparts <= 0) return W2S_EINVAL;
  if (!(C & 3) && !(ldg & 3) && !((uintptr_t)g & 15)) {
    hipLaunchKernelGGL(colsum_rows4_kernel, dim3(nparts), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, rows, C, ldg, part);
    W2S_CHECK_LAUNCH();
    return W2S_OK;
  }
  hipLaunchKernelGGL(colsum_rows_kernel, dim3(nparts), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, rows, C, ldg, part);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
// out[c] (+)= sum_p part[p*ld + c]: 16 column lanes x 16 part lanes, fp64, fixed order
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, int nparts, int C, int ld, float* __restrict__ out, int accumulate) {
  __shared__ double red[16][17];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s = 0.0;
  if (c < C)
    for (int p = pl; p < nparts; p += 16) s += (double)part[(size_t)p * ld + c];
  red[pl][cl] = s;
  __syncthreads();
  if (pl != 0 || c >= C) return;
#pragma unroll
  for (int k = 1; k < 16; ++k) s += red[k][cl];
  out[c] = accumulate ? out[c] + (float)s : (float)s;
}
// many column sums in one launch (a backward pass has ~55 of them, each a few microseconds): job table by value
#define W2S_COLSUM_BATCH 64
struct ColsumJobD { const float* part; float* out; int nparts, C, ld, accumulate, blk0; };
struct ColsumBatch { ColsumJobD j[W2S_COLSUM_BATCH]; int njobs; };
__global__ __launch_bounds__(256) void colsum_batch_kernel(ColsumBatch T) {
  __shared__ double red[16][17];
  int k = 0;
  while (k + 1 < T.njobs && (int)blockIdx.x >= T.j[k + 1].blk0) ++k;
  const ColsumJobD& J = T.j[k];
  const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c = ((int)blockIdx.x - J.blk0) * 16 + cl;
  double s = 0.0;
  if (c < J.C)
    for (int p = pl; p < J.nparts; p += 16) s += (double)J.part[(size_t)p * J.ld + c];
  red[pl][cl] = s;
  __syncthreads();
  if (pl != 0 || c >= J.C) return;
#pragma unroll
  for (int q = 1; q < 16; ++q) s += red[q][cl];
  J.out[c] = J.accumulate ? J.out[c] + (float)s : (float)s;
}
extern "C" int w2s_colsum_batch(const w2s_colsum_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs <= 0) return W2S_EINVAL;
  for (int base = 0; base < njobs; base += W2S_COLSUM_BATCH) {
    ColsumBatch T;
    T.njobs = (njobs - base < W2S_COLSUM_BATCH) ? njobs - base : W2S_COLSUM_BATCH;
    int blocks = 0;
    for (int i = 0; i < T.njobs; ++i) {
      const w2s_colsum_job& q = jobs[base + i];
      if (!q.part || !q.out || q.nparts <= 0 || q.C <= 0 || q.ld < q.C) return W2S_EINVAL;
      T.j[i] = ColsumJobD{q.part, q.out, q.nparts, q.C, q.ld, q.accumulate, blocks};
      blocks += (q.C + 15) / 16;
    }
    hipLaunchKernelGGL(colsum_batch_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), T);
    W2S_CHECK_LAUNCH();
  }
  return W2S_OK;
}

extern "C" int w2s_colsum(const float* part, int nparts, int C, int ld, float* out, int accumulate, void* stream) {
  if (!part || !out || nparts <= 0 || C <= 0 || ld < C) return W2S_EINVAL;
  hipLaunchKernelGGL(colsum_kernel, dim3((C + 15) / 16), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), part, nparts, C, ld, out, accumulate);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// dst[row][0..C) = src[0..C)  (CLS token rows of the set-fusion input, wav2sleep.py:330)
__global__ void fill_rows_kernel(float* __restrict__ dst, int ld, const float* __restrict__ src, int rows, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * C) return;
  const int c = (int)(i % C);
  dst[(i / C) * ld + c] = src[c];
}
extern "C" int w2s_fill_rows(float* dst, int ld, const float* src, int rows, int C, void* stream) {
  if (!dst || !src) return W2S_EINVAL;
  const size_t n = (size_t)rows * C;
  hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dst, ld, src, rows, C);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// dst[row][c] (+)= keep[row / rows_per_sample] * src[c * src_stride]: register tokens beyond the CLS column of the [1,1,F,R+1]
// parameter (wav2sleep.py:299,330; accumulate = 0, src_stride = R+1) and the signal-source embedding added to an encoder's
// output rows (wav2sleep.py:155-159; accumulate = 1; missing samples keep their zero rows)
__global__ void add_rows_kernel(float* __restrict__ dst, int ld, const float* __restrict__ src, int src_stride, const float* __restrict__ keep,
                                int rows_per_sample, int rows, int C, int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * C) return;
  const int c = (int)(i % C);
  const size_t row = i / C;
  const float k = keep ? keep[row / rows_per_sample] : 1.f;
  const float v = k * src[(size_t)c * src_stride];
  float* d = dst + row * ld + c;
  *d = accumulate ? *d + v : v;
}
extern "C" int w2s_add_rows(float* dst, int ld, const float* src, int src_stride, const float* keep, int rows_per_sample, int rows, int C,
                            int accumulate, void* stream) {
  if (!dst || !src || rows <= 0 || C <= 0 || src_stride <= 0 || (keep && rows_per_sample <= 0)) return W2S_EINVAL;
  const size_t n = (size_t)rows * C;
  hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dst, ld, src, src_stride,
                     keep, rows_per_sample > 0 ? rows_per_sample : 1, rows, C, accumulate);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// out[row][c] = g[row*ldg + c] * GELU'(pre[row][c]) * keep[row / rows_per_sample]   (encoder output GELU backward,
// reading the modality's column block of the token-gradient tensor; wav2sleep.py:154,265)
__global__ __launch_bounds__(256) void gelu_bwd_rows_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ pre,
                                                            const float* __restrict__ keep, int rps, float* __restrict__ out, int rows, int C) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // float4 index
  const int c4n = C >> 2;
  if (i >= (size_t)rows * c4n) return;
  const size_t row = i / c4n;
  const int c = (int)(i % c4n) * 4;
  const float k = keep ? keep[row / rps] : 1.0f;
  f32x4 o = ld4(g + row * ldg + c) * gelu_grad4(ld4(pre + row * C + c)) * k;
  st4(out + row * C + c, o);
}
extern "C" int w2s_gelu_bwd_rows(const float* g, int ldg, const float* pre, const float* keep, int rows_per_sample, float* out, int rows,
                                 int C, void* stream) {
  if (!g || !pre || !out || (C & 3) || rows_per_sample <= 0) return W2S_EINVAL;
  const size_t n4 = (size_t)rows * (C >> 2);
  hipLaunchKernelGGL(gelu_bwd_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, ldg,
                     pre, keep, rows_per_sample, out, rows, C);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// ------------------------------------------------------------------------------------------------
// elementwise (float4)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eltwise_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                                      size_t n4, float p, uint64_t seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 va = a ? ld4(a + 4 * i) : (f32x4){0, 0, 0, 0};
    f32x4 vb = b ? ld4(b + 4 * i) : (f32x4){0, 0, 0, 0};
    f32x4 m = {1, 1, 1, 1};
    if (op >= W2S_ELT_ADD_DROP && p > 0.f) {
      m.x = w2s_dropscale(seed, 4 * i + 0, p); m.y = w2s_dropscale(seed, 4 * i + 1, p);
      m.z = w2s_dropscale(seed, 4 * i + 2, p); m.w = w2s_dropscale(seed, 4 * i + 3, p);
    }
    f32x4 o;
    switch (op) {
      case W2S_ELT_GELU: o = gelu4(va); break;
      case W2S_ELT_GELU_BWD: o = vb * gelu_grad4(va); break;
      case W2S_ELT_ADD: o = va + vb; break;
      case W2S_ELT_ADD_DROP: o = va + vb * m; break;
      case W2S_ELT_DROP: o = va * m; break;
      case W2S_ELT_GELU_DROP: o = gelu4(va) * m; break;
      case W2S_ELT_GELU_DROP_BWD: o = vb * m * gelu_grad4(va); break;
      default: o = va; break;
    }
    st4(y + 4 * i, o);
  }
}
extern "C" int w2s_eltwise(int op, const float* a, const float* b, float* y, long n, float p_drop, uint64_t seed, void* stream) {
  if (!y || n <= 0 || (n & 3)) return W2S_EINVAL;
  const size_t n4 = (size_t)n / 4;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(eltwise_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), op, a, b, y, n4, p_drop, seed);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// ------------------------------------------------------------------------------------------------
// Set-fusion attention core.  qkv [N*D][3F] (q | k | v, F = H*16), keypad [N][D] (1 = missing modality).
// 16 lanes per (sentence n, head h): lane e holds feature e of every token's q/k/v, scores are 16-lane
// shuffle reductions, everything else is lane-local.  softmax(q k^T / 4 + (-inf on padded keys)),
// attention-probability dropout (p) regenerated from (seed, index) in the backward.
// Replaces F.scaled_dot_product_attention inside nn.MultiheadAttention (wav2sleep.py:286-296).
// ------------------------------------------------------------------------------------------------
// ---- D <= 6 tokens (every shipped map: 4 signals + CLS = 5, EOG pair = 3, single signal = 2): FOUR lanes per (n, head), four head
// dimensions each -- 16-byte loads / stores, a dot product is 4 FMAs + two quad DPP steps.  The 16-lane form below (one dimension per lane:
// dword accesses, four DPP steps per dot product) issued 4-5 x the wave instructions for the same bytes and ran at 2.2 TB/s; it stays
// for 7 .. 12 tokens, where six [D] float4 arrays no longer fit the register file.  Same dropout indices, same masks.
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_f(v, 0);
  v += dpp_f(v, 1);
  return v;
}
__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }

template <int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ keypad,
                                                       float* __restrict__ out, int N, int H, int nq, float p, uint64_t seed) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t pair = gid >> 2;  // (n, h); the four lanes of a quad share it, so a quad leaves together
  const int e = (threadIdx.x & 3) * 4;
  if (pair >= (size_t)N * H) return;
  const int n = (int)(pair / H), h = (int)(pair % H);
  const int F = H * 16, ld = 3 * F;
  f32x4 q[D], k[D], v[D];
  bool pad[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float* row = qkv + ((size_t)n * D + d) * ld + h * 16 + e;
    q[d] = (d < nq) ? ld4(row) : (f32x4){0, 0, 0, 0};
    k[d] = ld4(row + F); v[d] = ld4(row + 2 * F);
    pad[d] = keypad[(size_t)n * D + d] != 0;
  }
#pragma unroll
  for (int dq = 0; dq < D; ++dq) {
    if (dq >= nq) break;   // (uniform) nq = 1: only token 0's output is read downstream (the last layer: wav2sleep.py:345 returns the CLS token)
    float s[D];
    float mx = -INFINITY;
#pragma unroll
    for (int dk = 0; dk < D; ++dk) {
      s[dk] = pad[dk] ? -INFINITY : quad_sum(dot4(q[dq], k[dk])) * 0.25f;
      mx = fmaxf(mx, s[dk]);
    }
    float den = 0.f;
#pragma unroll
    for (int dk = 0; dk < D; ++dk) { s[dk] = __expf(s[dk] - mx); den += s[dk]; }
    const float inv = 1.0f / den;
    f32x4 o = {0, 0, 0, 0};
#pragma unroll
    for (int dk = 0; dk < D; ++dk) {
      float pr = s[dk] * inv;
      if (p > 0.f) pr *= w2s_dropscale(seed, ((pair * D + dq) * D + dk), p);
      o += pr * v[dk];
    }
    st4(out + ((size_t)n * D + dq) * F + h * 16 + e, o);
  }
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ keypad,
                                                       const float* __restrict__ gout, float* __restrict__ gqkv, int N, int H, int nq,
                                                       float p, uint64_t seed) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t pair = gid >> 2;
  const int e = (threadIdx.x & 3) * 4;
  if (pair >= (size_t)N * H) return;
  const int n = (int)(pair / H), h = (int)(pair % H);
  const int F = H * 16, ld = 3 * F;
  f32x4 q[D], k[D], v[D], go[D], dk_[D], dv_[D];
  bool pad[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float* row = qkv + ((size_t)n * D + d) * ld + h * 16 + e;
    const bool live = d < nq;   // (uniform) rows nobody read have no query gradient and their gout is never looked at
    q[d] = live ? ld4(row) : (f32x4){0, 0, 0, 0};
    k[d] = ld4(row + F); v[d] = ld4(row + 2 * F);
    go[d] = live ? ld4(gout + ((size_t)n * D + d) * F + h * 16 + e) : (f32x4){0, 0, 0, 0};
    pad[d] = keypad[(size_t)n * D + d] != 0;
    dk_[d] = (f32x4){0, 0, 0, 0}; dv_[d] = (f32x4){0, 0, 0, 0};
  }
#pragma unroll
  for (int dq = 0; dq < D; ++dq) {
    if (dq >= nq) {   // (uniform) a token whose output nobody read: its query gets no gradient and it adds nothing to the keys / values
      st4(gqkv + ((size_t)n * D + dq) * ld + h * 16 + e, (f32x4){0, 0, 0, 0});
      continue;
    }
    float pr[D], dp[D];
    float mx = -INFINITY;
#pragma unroll
    for (int d2 = 0; d2 < D; ++d2) {
      pr[d2] = pad[d2] ? -INFINITY : quad_sum(dot4(q[dq], k[d2])) * 0.25f;
      mx = fmaxf(mx, pr[d2]);
    }
    float den = 0.f;
#pragma unroll
    for (int d2 = 0; d2 < D; ++d2) { pr[d2] = __expf(pr[d2] - mx); den += pr[d2]; }
    const float inv = 1.0f / den;
    float dot = 0.f;
#pragma unroll
    for (int d2 = 0; d2 < D; ++d2) {
      pr[d2] *= inv;
      const float m = (p > 0.f) ? w2s_dropscale(seed, ((pair * D + dq) * D + d2), p) : 1.0f;
      // O = sum (P*m) V  =>  dV += (P*m) gO ; dP = m * (gO . V)
      dv_[d2] += (pr[d2] * m) * go[dq];
      dp[d2] = m * quad_sum(dot4(go[dq], v[d2]));
      dot += pr[d2] * dp[d2];
    }
    f32x4 dqv = {0, 0, 0, 0};
#pragma unroll
    for (int d2 = 0; d2 < D; ++d2) {
      const float ds = pr[d2] * (dp[d2] - dot) * 0.25f;  // dS (scaled)
      dqv += ds * k[d2];
      dk_[d2] += ds * q[dq];
    }
    st4(gqkv + ((size_t)n * D + dq) * ld + h * 16 + e, dqv);
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float* row = gqkv + ((size_t)n * D + d) * ld + h * 16 + e;
    st4(row + F, dk_[d]);
    st4(row + 2 * F, dv_[d]);
  }
}

// ---- 7 .. 12 tokens: sixteen lanes per (n, head), one head dimension each
template <int D>
__global__ __launch_bounds__(256) void attn_fwd16_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ keypad,
                                                       float* __restrict__ out, int N, int H, int nq, float p, uint64_t seed) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t pair = gid >> 4;  // (n, h)
  const int e = threadIdx.x & 15;
  if (pair >= (size_t)N * H) return;
  const int n = (int)(pair / H), h = (int)(pair % H);
  const int F = H * 16, ld = 3 * F;
  float q[D], k[D], v[D];
  bool pad[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float* row = qkv + ((size_t)n * D + d) * ld + h * 16 + e;
    q[d] = row[0]; k[d] = row[F]; v[d] = row[2 * F];
    pad[d] = keypad[(size_t)n * D + d] != 0;
  }
#pragma unroll
  for (int dq = 0; dq < D; ++dq) {
    if (dq >= nq) break;   // (uniform) nq = 1: only token 0's output is read downstream (the last layer: wav2sleep.py:345 returns the CLS token)
    float s[D];
    float mx = -INFINITY;
#pragma unroll
    for (int dk = 0; dk < D; ++dk) {
      s[dk] = pad[dk] ? -INFINITY : row16_sum(q[dq] * k[dk]) * 0.25f;
      mx = fmaxf(mx, s[dk]);
    }
    float den = 0.f;
#pragma unroll
    for (int dk = 0; dk < D; ++dk) { s[dk] = __expf(s[dk] - mx); den += s[dk]; }
    const float inv = 1.0f / den;
    float o = 0.f;
#pragma unroll
    for (int dk = 0; dk < D; ++dk) {
      float pr = s[dk] * inv;
      if (p > 0.f) pr *= w2s_dropscale(seed, ((pair * D + dq) * D + dk), p);
      o += pr * v[dk];
    }
    out[((size_t)n * D + dq) * F + h * 16 + e] = o;
  }
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd16_kernel(const float* __restrict__ qkv, const uint8_t* __restrict__ keypad,
                                                       const float* __restrict__ gout, float* __restrict__ gqkv, int N, int H, int nq,
                                                       float p, uint64_t seed) {
  const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t pair = gid >> 4;
  const int e = threadIdx.x & 15;
  if (pair >= (size_t)N * H) return;
  const int n = (int)(pair / H), h = (int)(pair % H);
  const int F = H * 16, ld = 3 * F;
  float q[D], k[D], v[D], go[D], dk_[D], dv_[D];
  bool pad[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float* row = qkv + ((size_t)n * D + d) * ld + h * 16 + e;
    q[d] = row[0]; k[d] = row[F]; v[d] = row[2 * F];
    go[d] = gout[((size_t)n * D + d) * F + h * 16 + e];
    pad[d] = keypad[(size_t)n * D + d] != 0;
    dk_[d] = 0.f; dv_[d] = 0.f;
  }
#pragma unroll
  for (int dq = 0; dq < D; ++dq) {
    if (dq >= nq) {   // (uniform) a token whose output nobody read: its query gets no gradient and it adds nothing to the keys / values
      gqkv[((size_t)n * D + dq) * ld + h * 16 + e] = 0.f;
      continue;
    }
    float pr[D], dp[D];
    float mx = -INFINITY;
#pragma unroll
    for (int d2 = 0; d2 < D; ++d2) {
      pr[d2] = pad[d2] ? -INFINITY : row16_sum(q[dq] * k[d2]) * 0.25f;
      mx = fmaxf(mx, pr[d2]);
    }
    float den = 0.f;
#pragma unroll
    for (int d2 = 0; d2 < D; ++d2) { pr[d2] = __expf(pr[d2] - mx); den += pr[d2]; }
    const float inv = 1.0f / den;
    float dot = 0.f;
#pragma unroll
    for (int d2 = 0; d2 < D; ++d2) {
      pr[d2] *= inv;
      const float m = (p > 0.f) ? w2s_dropscale(seed, ((pair * D + dq) * D + d2), p) : 1.0f;
      // O = sum (P*m) V  =>  dV += (P*m) gO ; dP = m * (gO . V)
      dv_[d2] += pr[d2] * m * go[dq];
      dp[d2] = m * row16_sum(go[dq] * v[d2]);
      dot += pr[d2] * dp[d2];
    }
    float dqv = 0.f;
#pragma unroll
    for (int d2 = 0; d2 < D; ++d2) {
      const float ds = pr[d2] * (dp[d2] - dot) * 0.25f;  // dS (scaled)
      dqv += ds * k[d2];
      dk_[d2] += ds * q[dq];
    }
    gqkv[((size_t)n * D + dq) * ld + h * 16 + e] = dqv;
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    float* row = gqkv + ((size_t)n * D + d) * ld + h * 16 + e;
    row[F] = dk_[d];
    row[2 * F] = dv_[d];
  }
}

#define W2S_ATTN_DISPATCH(KERN, KERN16, ...)                                                         \
  switch (D) {                                                                                       \
    case 2: hipLaunchKernelGGL(KERN<2>, dim3(blocks4), dim3(256), 0, s, __VA_ARGS__); break;         \
    case 3: hipLaunchKernelGGL(KERN<3>, dim3(blocks4), dim3(256), 0, s, __VA_ARGS__); break;         \
    case 4: hipLaunchKernelGGL(KERN<4>, dim3(blocks4), dim3(256), 0, s, __VA_ARGS__); break;         \
    case 5: hipLaunchKernelGGL(KERN<5>, dim3(blocks4), dim3(256), 0, s, __VA_ARGS__); break;         \
    case 6: hipLaunchKernelGGL(KERN<6>, dim3(blocks4), dim3(256), 0, s, __VA_ARGS__); break;         \
    case 7: hipLaunchKernelGGL(KERN16<7>, dim3(blocks), dim3(256), 0, s, __VA_ARGS__); break;        \
    case 8: hipLaunchKernelGGL(KERN16<8>, dim3(blocks), dim3(256), 0, s, __VA_ARGS__); break;        \
    case 9: hipLaunchKernelGGL(KERN16<9>, dim3(blocks), dim3(256), 0, s, __VA_ARGS__); break;        \
    case 10: hipLaunchKernelGGL(KERN16<10>, dim3(blocks), dim3(256), 0, s, __VA_ARGS__); break;      \
    case 11: hipLaunchKernelGGL(KERN16<11>, dim3(blocks), dim3(256), 0, s, __VA_ARGS__); break;      \
    case 12: hipLaunchKernelGGL(KERN16<12>, dim3(blocks), dim3(256), 0, s, __VA_ARGS__); break;      \
    default: return W2S_EINVAL;   /* 2 .. 12 tokens: six signals + CLS + five register tokens is all the reference's maps can ask for */ \
  }

extern "C" int w2s_attn_fwd(const float* qkv, const uint8_t* keypad, float* out, int N, int D, int H, int nq, float p_drop, uint64_t seed,
                            void* stream) {
  if (!qkv || !keypad || !out || N <= 0 || H <= 0 || (nq != D && nq != 1)) return W2S_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned blocks = (unsigned)(((size_t)N * H * 16 + 255) / 256), blocks4 = (unsigned)(((size_t)N * H * 4 + 255) / 256);
  W2S_ATTN_DISPATCH(attn_fwd_kernel, attn_fwd16_kernel, qkv, keypad, out, N, H, nq, p_drop, seed)
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
extern "C" int w2s_attn_bwd(const float* qkv, const uint8_t* keypad, const float* gout, float* gqkv, int N, int D, int H, int nq, float p_drop,
                            uint64_t seed, void* stream) {
  if (!qkv || !keypad || !gout || !gqkv || N <= 0 || H <= 0 || (nq != D && nq != 1)) return W2S_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned blocks = (unsigned)(((size_t)N * H * 16 + 255) / 256), blocks4 = (unsigned)(((size_t)N * H * 4 + 255) / 256);
  W2S_ATTN_DISPATCH(attn_bwd_kernel, attn_bwd16_kernel, qkv, keypad, gout, gqkv, N, H, nq, p_drop, seed)
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
