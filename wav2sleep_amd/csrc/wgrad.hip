// Weight gradient of the channels-last convolution on the fp32 matrix cores, + slab reduction, + weight repack.
//
//   dW[o][j][c] = sum_{b,t} GY[b,t,o] * H[b, t*stride + j*dil - pad, c]
// is a GEMM with M = cout, N = taps*cin and K = B*L_out (millions).  A workgroup walks position tiles
// (grid-stride over (b, tile)), stages GY (its 16*NTO output channels) and the H window into LDS with the same
// on-load transforms as conv_cl (GY = instance-norm backward of the incoming gradient, H = norm+GELU of the
// stored pre-norm tensor), and each of its 4 waves accumulates the FULL (16*NTO) x (TAPS_T*16*NTC) block over
// its own quarter of the positions.  Every wave then writes its accumulators as one raw-fragment slab;
// w2s_wgrad_reduce sums the slabs in a fixed order (deterministic) and scatters to torch layout [o][c][j].
// blockIdx.y enumerates (output-channel tile, tap group) so accumulators stay <= 32 tiles (128 VGPRs).
#include <cstdlib>
#include "w2s_common.h"

struct WgradP {
  w2s_wgrad_args a;
  int TM;      // positions per staged tile
  int ntiles;  // tiles per sample
  int ntg;     // tap groups = taps / TAPS_T
};

__device__ __forceinline__ f32x4 pro4(int pro, f32x4 v, f32x4 v2, f32x4 mean, f32x4 rstd, f32x4 s1, f32x4 s2) {
  switch (pro) {
    case W2S_PRO_SANITIZE:
      v.x = sanitize_f(v.x); v.y = sanitize_f(v.y); v.z = sanitize_f(v.z); v.w = sanitize_f(v.w);
      return v;
    case W2S_PRO_GELU: return gelu4(v);
    case W2S_PRO_IN_GELU: return gelu4((v - mean) * rstd);
    case W2S_PRO_INBWD: { f32x4 n = (v2 - mean) * rstd; return rstd * (v - s1 - n * s2); }
    case W2S_PRO_INBWD_GP: { f32x4 n = (v2 - mean) * rstd; f32x4 gn = v * gelu_grad4(n); return rstd * (gn - s1 - n * s2); }
    default: return v;
  }
}

__device__ __forceinline__ void load_chan_params(const float* stats, int b, int C, int ch, f32x4& p0, f32x4& p1) {
  const float* st = stats + ((size_t)b * C + ch) * 2;
  f32x4 s01 = ld4(st), s23 = ld4(st + 4);
  p0 = (f32x4){s01.x, s01.z, s23.x, s23.z};
  p1 = (f32x4){s01.y, s01.w, s23.y, s23.w};
}

template <int NTO, int NTC, int TAPS_T, int STRIDE>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradP P) {
  extern __shared__ f32x4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);
  const w2s_wgrad_args& a = P.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int TM = P.TM;
  const int cin = a.cin;             // == 16*NTC
  constexpr int WG = NTO * 16;       // gradient-side width handled here
  const int oy = blockIdx.y / P.ntg, tg = blockIdx.y % P.ntg;
  const int o0 = oy * WG, j0 = tg * TAPS_T;
  const int RSg = WG + 4, RSh = cin + 4;
  const int NRh = (TAPS_T == 1) ? TM : (TM - 1) * STRIDE + TAPS_T;  // TAPS_T>1 only with dil == 1
  float* gyL = smem;
  float* hL = smem + TM * RSg;

  f32x4 acc[NTO][TAPS_T][NTC];
#pragma unroll
  for (int i = 0; i < NTO; ++i)
#pragma unroll
    for (int j = 0; j < TAPS_T; ++j)
#pragma unroll
      for (int c = 0; c < NTC; ++c) acc[i][j][c] = (f32x4){0, 0, 0, 0};

  const int total = a.B * P.ntiles;
  for (int tl = blockIdx.x; tl < total; tl += gridDim.x) {
    const int b = tl / P.ntiles, tile = tl % P.ntiles;
    const int t0 = tile * TM;
    __syncthreads();
    {  // ---- stage GY tile: rows t0..t0+TM-1, channels o0..o0+WG-1
      constexpr int c4n = WG / 4, rstep = 256 / c4n;
      const int myc4 = tid % c4n, row0 = tid / c4n, ch = o0 + myc4 * 4;
      f32x4 pm = {0, 0, 0, 0}, pr = {1, 1, 1, 1}, ps1 = {0, 0, 0, 0}, ps2 = {0, 0, 0, 0};
      if (a.pro_g >= W2S_PRO_IN_GELU) {
        load_chan_params(a.g_stats, b, a.cout, ch, pm, pr);
        if (a.pro_g >= W2S_PRO_INBWD) load_chan_params(a.g_bstats, b, a.cout, ch, ps1, ps2);
      }
      const ProCoef kg = pro_coef(a.pro_g, pm, pr, ps1, ps2);
      const float* gb = a.g + (size_t)b * a.L_out * a.ldg;   // uniform base, 32-bit lane offsets (scalar-base addressing)
      const float* g2b = (a.pro_g >= W2S_PRO_INBWD) ? a.g2 + (size_t)b * a.L_out * a.ldg : nullptr;
      for (int row = row0; row < TM; row += rstep) {
        const int t = t0 + row;
        f32x4 v = {0, 0, 0, 0};
        if (t < a.L_out) {
          const unsigned go = (unsigned)t * (unsigned)a.ldg + ch;
          f32x4 x = ld4o(gb, go);
          f32x4 x2 = g2b ? ld4o(g2b, go) : (f32x4){0, 0, 0, 0};
          v = pro_apply_k(a.pro_g, x, x2, kg);
        }
        st4(gyL + row * RSg + myc4 * 4, v);
      }
    }
    {  // ---- stage H window
      const int c4n = cin >> 2, rstep = 256 / c4n;
      const int myc4 = tid % c4n, row0 = tid / c4n, ch = myc4 * 4;
      f32x4 pm = {0, 0, 0, 0}, pr = {1, 1, 1, 1}, z = {0, 0, 0, 0};
      if (a.pro_h >= W2S_PRO_IN_GELU) load_chan_params(a.x_stats, b, cin, ch, pm, pr);
      const ProCoef kh = pro_coef(a.pro_h, pm, pr, z, z);
      const float* xb = a.x + (size_t)b * a.L_in * a.ldx;
      const int rb = t0 * STRIDE - a.pad + j0 * a.dil;
      const int rowmul = (TAPS_T == 1) ? STRIDE : 1;
      for (int row = row0; row < NRh; row += rstep) {
        const int gr = rb + row * rowmul;
        f32x4 v = {0, 0, 0, 0};
        if (gr >= 0 && gr < a.L_in) v = pro_apply_k(a.pro_h, ld4o(xb, (unsigned)gr * (unsigned)a.ldx + ch), z, kh);
        st4(hL + row * RSh + ch, v);
      }
    }
    __syncthreads();
    // ---- MFMA: this wave's quarter of the positions; k-step = 4 positions (lane group g picks one)
    const int pw = wave * (TM >> 2);
    for (int s = 0; s < (TM >> 4); ++s) {
      const int p = pw + 4 * s + g;
      float ga[NTO], hb[TAPS_T][NTC];
#pragma unroll
      for (int i = 0; i < NTO; ++i) ga[i] = gyL[p * RSg + i * 16 + r];
#pragma unroll
      for (int j = 0; j < TAPS_T; ++j) {
        const int hr = (TAPS_T == 1) ? p : p * STRIDE + j;
#pragma unroll
        for (int c = 0; c < NTC; ++c) hb[j][c] = hL[hr * RSh + c * 16 + r];
      }
#pragma unroll
      for (int i = 0; i < NTO; ++i)
#pragma unroll
        for (int j = 0; j < TAPS_T; ++j)
#pragma unroll
          for (int c = 0; c < NTC; ++c) acc[i][j][c] = mfma16(ga[i], hb[j][c], acc[i][j][c]);
    }
  }
  // ---- raw-fragment slab: slab[(blockIdx.x*4+wave)][blockIdx.y][tile(i,j,c)][lane][4]
  constexpr int TILES = NTO * TAPS_T * NTC;
  float* out = a.slab + (((size_t)(blockIdx.x * 4 + wave) * gridDim.y + blockIdx.y) * TILES) * 256 + lane * 4;
#pragma unroll
  for (int i = 0; i < NTO; ++i)
#pragma unroll
    for (int j = 0; j < TAPS_T; ++j)
#pragma unroll
      for (int c = 0; c < NTC; ++c) st4(out + ((i * TAPS_T + j) * NTC + c) * 256, acc[i][j][c]);
}

// ------------------------------------------------------------------------------------------------------------------
// Tile-split variant for the matrix-core-bound layers (cin >= 64): ONE workgroup of NW waves covers a whole block of
// NW*OT*16 output channels x JT taps x all cin channels, wave w owning output-channel tiles [w*OT, (w+1)*OT).  Every wave
// walks all positions of the staged tile, so GY and H are staged (and GELU'd) once instead of once per output-channel
// slice (the position-split kernel above re-staged H up to 6x for 128 channels).  One slab per workgroup.
// ------------------------------------------------------------------------------------------------------------------
// PG / PH >= 0: gradient-side / input-side transform fixed at compile time (encoder k=3 convs; fewer live registers so two
// 8-wave workgroups fit a CU), -1: runtime switch.
template <int OT, int JT, int CT, int NW, int STRIDE, int PG, int PH>
__global__ __launch_bounds__(NW * 64) void wgrad_ts_kernel(WgradP P) {
  extern __shared__ f32x4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);
  const w2s_wgrad_args& a = P.a;
  const int pro_g = (PG >= 0) ? PG : a.pro_g, pro_h = (PH >= 0) ? PH : a.pro_h;
  constexpr int NT = NW * 64, WG = NW * OT * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int TM = P.TM;
  const int cin = a.cin;  // == 16*CT
  const int oy = blockIdx.y / P.ntg, tg = blockIdx.y % P.ntg;
  const int o0 = oy * WG, j0 = tg * JT;
  const int RSg = WG + 4, RSh = cin + 4;
  const int NRh = (JT == 1) ? TM : (TM - 1) * STRIDE + JT;
  float* gyL = smem;
  float* hL = smem + TM * RSg;

  f32x4 acc[OT][JT][CT];
#pragma unroll
  for (int i = 0; i < OT; ++i)
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[i][j][c] = (f32x4){0, 0, 0, 0};

  const int total = a.B * P.ntiles;
  for (int tl = blockIdx.x; tl < total; tl += gridDim.x) {
    const int b = tl / P.ntiles, tile = tl % P.ntiles;
    const int t0 = tile * TM;
    __syncthreads();
    {  // ---- stage GY tile (channels o0 .. o0+WG-1)
      constexpr int c4n = WG / 4;
      const float* gb = a.g + (size_t)b * a.L_out * a.ldg + o0;
      const float* g2b = (pro_g >= W2S_PRO_INBWD) ? a.g2 + (size_t)b * a.L_out * a.ldg + o0 : nullptr;
      if constexpr (NT % c4n == 0) {
        constexpr int rstep = NT / c4n;
        const int myc4 = tid % c4n, row0 = tid / c4n, ch = myc4 * 4;
        f32x4 pm = {0, 0, 0, 0}, pr = {1, 1, 1, 1}, ps1 = {0, 0, 0, 0}, ps2 = {0, 0, 0, 0};
        if (pro_g >= W2S_PRO_IN_GELU) {
          load_chan_params(a.g_stats, b, a.cout, o0 + ch, pm, pr);
          if (pro_g >= W2S_PRO_INBWD) load_chan_params(a.g_bstats, b, a.cout, o0 + ch, ps1, ps2);
        }
        const ProCoef kg = pro_coef(pro_g, pm, pr, ps1, ps2);
        for (int row = row0; row < TM; row += rstep) {
          const int t = t0 + row;
          f32x4 v = {0, 0, 0, 0};
          if (t < a.L_out) {
            const unsigned go = (unsigned)t * (unsigned)a.ldg + ch;
            f32x4 x = ld4o(gb, go);
            f32x4 x2 = g2b ? ld4o(g2b, go) : (f32x4){0, 0, 0, 0};
            v = pro_apply_k(pro_g, x, x2, kg);
          }
          st4(gyL + row * RSg + ch, v);
        }
      } else {  // wide untransformed gradients (transformer linears): launcher guarantees pro_g == NONE
        for (int f = tid; f < TM * c4n; f += NT) {
          const int row = f / c4n, ch = (f % c4n) * 4, t = t0 + row;
          st4(gyL + row * RSg + ch, t < a.L_out ? ld4o(gb, (unsigned)t * (unsigned)a.ldg + ch) : (f32x4){0, 0, 0, 0});
        }
      }
    }
    {  // ---- stage H window
      const int c4n = cin >> 2, rstep = NT / c4n;
      const int myc4 = tid % c4n, row0 = tid / c4n, ch = myc4 * 4;
      f32x4 pm = {0, 0, 0, 0}, pr = {1, 1, 1, 1}, z = {0, 0, 0, 0};
      if (pro_h >= W2S_PRO_IN_GELU) load_chan_params(a.x_stats, b, cin, ch, pm, pr);
      const ProCoef kh = pro_coef(pro_h, pm, pr, z, z);
      const float* xb = a.x + (size_t)b * a.L_in * a.ldx;
      const int rb = t0 * STRIDE - a.pad + j0 * a.dil;
      const int rowmul = (JT == 1) ? STRIDE : 1;
      for (int row = row0; row < NRh; row += rstep) {
        const int gr = rb + row * rowmul;
        f32x4 v = {0, 0, 0, 0};
        if (gr >= 0 && gr < a.L_in) v = pro_apply_k(pro_h, ld4o(xb, (unsigned)gr * (unsigned)a.ldx + ch), z, kh);
        st4(hL + row * RSh + ch, v);
      }
    }
    __syncthreads();
    for (int s = 0; s < (TM >> 2); ++s) {
      const int p = 4 * s + g;
      float ga[OT], hb[JT][CT];
#pragma unroll
      for (int i = 0; i < OT; ++i) ga[i] = gyL[p * RSg + (wave * OT + i) * 16 + r];
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int hr = (JT == 1) ? p : p * STRIDE + j;
#pragma unroll
        for (int c = 0; c < CT; ++c) hb[j][c] = hL[hr * RSh + c * 16 + r];
      }
#pragma unroll
      for (int i = 0; i < OT; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
          for (int c = 0; c < CT; ++c) acc[i][j][c] = mfma16(ga[i], hb[j][c], acc[i][j][c]);
    }
  }
  constexpr int TILES = NW * OT * JT * CT;
  float* out = a.slab + (((size_t)blockIdx.x * gridDim.y + blockIdx.y) * TILES) * 256 + lane * 4;
#pragma unroll
  for (int i = 0; i < OT; ++i)
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int c = 0; c < CT; ++c) st4(out + (((wave * OT + i) * JT + j) * CT + c) * 256, acc[i][j][c]);
}

// ------------------------------------------------------------------------------------------------------------------
// Split-precision ("bf16x3") tile-split weight gradient.  GY and H are staged as bf16 hi/lo planes in plain row-major
// [position][channel] LDS tiles; both MFMA operands need 8 CONSECUTIVE POSITIONS per lane (the contraction index), which
// is a column of those tiles, so they are fetched with the CDNA4 transposing LDS read ds_read_b64_tr_b16 (4 positions x
// 16 channels per 16-lane group, delivered position-major) -- two per operand, any tap shift is just a row offset.
// Wave w owns input-channel tile (w % CT) and output-channel tiles [(w / CT) * OW, +OW): the H fragments (3 taps) are
// read once per wave and the GY fragments are shared through LDS broadcast-free reads; 3 bf16 MFMAs (16 cycles each,
// K = 32 positions) replace 8 fp32 MFMAs (32 cycles each, K = 4).  Slab layout identical to wgrad_ts_kernel.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x4v __attribute__((__vector_size__(4 * sizeof(__bf16))));
__device__ __forceinline__ bf16x8 tr_read8(const __bf16* p0, const __bf16* p1) {
  typedef __attribute__((address_space(3))) bf16x4v* lds_p;
  bf16x4v a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p0));
  bf16x4v b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p1));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int OW, int JT, int CT, int NW, int STRIDE, int PG, int PH>
__global__ __launch_bounds__(NW * 64) void wgrad_bf_kernel(WgradP P) {
  extern __shared__ f32x4 smem4[];
  const w2s_wgrad_args& a = P.a;
  const int pro_g = (PG >= 0) ? PG : a.pro_g, pro_h = (PH >= 0) ? PH : a.pro_h;
  constexpr int NT = NW * 64, OGR = NW / CT, WG = OGR * OW * 16;  // WG = output channels of this block
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
  const int TM = P.TM;
  const int cin = a.cin;  // == 16*CT
  const int oy = blockIdx.y / P.ntg, tg = blockIdx.y % P.ntg;
  const int o0 = oy * WG, j0 = tg * JT;
  const int RSg = WG + 8, RSh = cin + 8;  // bf16 elements per row
  const int NRh = (JT == 1) ? TM : (TM - 1) * STRIDE + JT;
  __bf16* gH = reinterpret_cast<__bf16*>(smem4);
  __bf16* gL = gH + TM * RSg;
  __bf16* hH = gL + TM * RSg;
  __bf16* hL = hH + NRh * RSh;
  const int ctile = wave % CT, ogrp = wave / CT;

  f32x4 acc[OW][JT];
#pragma unroll
  for (int i = 0; i < OW; ++i)
#pragma unroll
    for (int j = 0; j < JT; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};

  auto split_store = [&](__bf16* hi, __bf16* lo, int off, f32x4 t) { split_store4(hi, lo, off, t); };

  const int total = a.B * P.ntiles;
  for (int tl = blockIdx.x; tl < total; tl += gridDim.x) {
    const int b = tl / P.ntiles, tile = tl % P.ntiles;
    const int t0 = tile * TM;
    __syncthreads();
    {  // ---- stage GY tile (channels o0 .. o0+WG-1) as bf16 hi/lo planes
      constexpr int c4n = WG / 4;
      const float* gb = a.g + (size_t)b * a.L_out * a.ldg + o0;
      const float* g2b = (pro_g >= W2S_PRO_INBWD) ? a.g2 + (size_t)b * a.L_out * a.ldg + o0 : nullptr;
      if constexpr (NT % c4n == 0) {
        constexpr int rstep = NT / c4n;
        const int myc4 = tid % c4n, row0 = tid / c4n, ch = myc4 * 4;
        f32x4 pm = {0, 0, 0, 0}, pr = {1, 1, 1, 1}, ps1 = {0, 0, 0, 0}, ps2 = {0, 0, 0, 0};
        if (pro_g >= W2S_PRO_IN_GELU) {
          load_chan_params(a.g_stats, b, a.cout, o0 + ch, pm, pr);
          if (pro_g >= W2S_PRO_INBWD) load_chan_params(a.g_bstats, b, a.cout, o0 + ch, ps1, ps2);
        }
        const ProCoef kg = pro_coef(pro_g, pm, pr, ps1, ps2);
        for (int row = row0; row < TM; row += rstep) {
          const int t = t0 + row;
          f32x4 v = {0, 0, 0, 0};
          if (t < a.L_out) {
            const unsigned go = (unsigned)t * (unsigned)a.ldg + ch;
            f32x4 x = ld4o(gb, go);
            f32x4 x2 = g2b ? ld4o(g2b, go) : (f32x4){0, 0, 0, 0};
            v = pro_apply_k(pro_g, x, x2, kg);
          }
          split_store(gH, gL, row * RSg + ch, v);
        }
      } else {
        for (int f = tid; f < TM * c4n; f += NT) {
          const int row = f / c4n, ch = (f % c4n) * 4, t = t0 + row;
          split_store(gH, gL, row * RSg + ch, t < a.L_out ? ld4o(gb, (unsigned)t * (unsigned)a.ldg + ch) : (f32x4){0, 0, 0, 0});
        }
      }
    }
    {  // ---- stage H window
      const int c4n = cin >> 2, rstep = NT / c4n;
      const int myc4 = tid % c4n, row0 = tid / c4n, ch = myc4 * 4;
      f32x4 pm = {0, 0, 0, 0}, pr = {1, 1, 1, 1}, z = {0, 0, 0, 0};
      if (pro_h >= W2S_PRO_IN_GELU) load_chan_params(a.x_stats, b, cin, ch, pm, pr);
      const ProCoef kh = pro_coef(pro_h, pm, pr, z, z);
      const float* xb = a.x + (size_t)b * a.L_in * a.ldx;
      const int rb = t0 * STRIDE - a.pad + j0 * a.dil;
      const int rowmul = (JT == 1) ? STRIDE : 1;
      for (int row = row0; row < NRh; row += rstep) {
        const int gr = rb + row * rowmul;
        f32x4 v = {0, 0, 0, 0};
        if (gr >= 0 && gr < a.L_in) v = pro_apply_k(pro_h, ld4o(xb, (unsigned)gr * (unsigned)a.ldx + ch), z, kh);
        split_store(hH, hL, row * RSh + ch, v);
      }
    }
    __syncthreads();
    // ---- k-steps of 32 positions; lane group g covers positions 8g..8g+7 of the step (two 4-row transposed reads)
    for (int s = 0; s < (TM >> 5); ++s) {
      const int p0 = 32 * s + 8 * g + q4;  // this lane's address row for the first 4-position block
      bf16x8 bh[JT], bl[JT];
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int r0 = (JT == 1) ? p0 : p0 * STRIDE + j;
        const int r1 = (JT == 1) ? p0 + 4 : (p0 + 4) * STRIDE + j;
        const int col = ctile * 16 + 4 * p4;
        bh[j] = tr_read8(hH + r0 * RSh + col, hH + r1 * RSh + col);
        bl[j] = tr_read8(hL + r0 * RSh + col, hL + r1 * RSh + col);
      }
#pragma unroll
      for (int i = 0; i < OW; ++i) {
        const int col = (ogrp * OW + i) * 16 + 4 * p4;
        const bf16x8 ah = tr_read8(gH + p0 * RSg + col, gH + (p0 + 4) * RSg + col);
        const bf16x8 al = tr_read8(gL + p0 * RSg + col, gL + (p0 + 4) * RSg + col);
#pragma unroll
        for (int j = 0; j < JT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  }
  constexpr int TILES = OGR * OW * JT * CT;
  float* out = a.slab + (((size_t)blockIdx.x * gridDim.y + blockIdx.y) * TILES) * 256 + lane * 4;
#pragma unroll
  for (int i = 0; i < OW; ++i)
#pragma unroll
    for (int j = 0; j < JT; ++j) st4(out + (((ogrp * OW + i) * JT + j) * CT + ctile) * 256, acc[i][j]);
}

// ------------------------------------------------------------------------------------------------------------------
// Round 6: the same kernel SOFTWARE-PIPELINED for the k = 1 / tap-split weight gradients without a gradient-side transform (the set-fusion
// transformer's linears over 76 800 token rows, the SequenceCNN's dilated convs, the encoders' dense layer and 1x1 joins): wgrad_bf_kernel
// above runs  load -> wait -> split -> LDS -> barrier -> MFMA -> barrier  per tile, and with 384-512 gradient channels a tile is ONE 32-position
// K step -- every 1500-cycle matrix phase waited for its own HBM round trip first (in_proj's weight gradient: 110 us for 157 MB).  Here the
// raw rows of tile i + 1 are requested (unconditional loads from clamped rows: no branch around the vector-memory queue) right after tile
// i's windows have been committed to LDS, and land while tile i runs through the matrix cores.  TM is a compile-time constant (register
// arrays); same products in the same order, same slab layout: results are bit-identical to wgrad_bf_kernel's.
// ------------------------------------------------------------------------------------------------------------------
template <int OW, int CT, int NW, int STRIDE, int TM>
__global__ __launch_bounds__(NW * 64) void wgrad_bf_pf_kernel(WgradP P) {
  extern __shared__ f32x4 smem4[];
  const w2s_wgrad_args& a = P.a;
  constexpr int NT = NW * 64, OGR = NW / CT, WG = OGR * OW * 16, HC = CT * 16;
  constexpr int RSg = WG + 8, RSh = HC + 8;
  constexpr int c4g = WG / 4, c4h = HC / 4;
  constexpr int NG = (TM * c4g + NT - 1) / NT, NH = (TM * c4h + NT - 1) / NT;   // float4s per thread and tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
  const int oy = blockIdx.y / P.ntg, tg = blockIdx.y % P.ntg;
  const int o0 = oy * WG, j0 = tg;
  __bf16* gH = reinterpret_cast<__bf16*>(smem4);
  __bf16* gL = gH + TM * RSg;
  __bf16* hH = gL + TM * RSg;
  __bf16* hL = hH + TM * RSh;
  const int ctile = wave % CT, ogrp = wave / CT;
  const int pro_h = a.pro_h;   // NONE or GELU (no statistics): uniform

  f32x4 acc[OW];
#pragma unroll
  for (int i = 0; i < OW; ++i) acc[i] = (f32x4){0, 0, 0, 0};

  f32x4 rg[NG], rh[NH];
  const int total = a.B * P.ntiles;
  auto prefetch = [&](int tl) {
    const int b = tl / P.ntiles, t0 = (tl - b * P.ntiles) * TM;
    const float* gb = a.g + (size_t)b * a.L_out * a.ldg + o0;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int f = min(tid + k * NT, TM * c4g - 1), row = f / c4g, ch = (f - row * c4g) * 4;
      rg[k] = ld4o(gb, (unsigned)min(t0 + row, a.L_out - 1) * (unsigned)a.ldg + ch);
    }
    const float* xb = a.x + (size_t)b * a.L_in * a.ldx;
    const int rb = t0 * STRIDE - a.pad + j0 * a.dil;
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const int f = min(tid + k * NT, TM * c4h - 1), row = f / c4h, ch = (f - row * c4h) * 4;
      rh[k] = ld4o(xb, (unsigned)min(max(rb + row * STRIDE, 0), a.L_in - 1) * (unsigned)a.ldx + ch);
    }
  };
  auto commit = [&](int tl) {
    const int b = tl / P.ntiles, t0 = (tl - b * P.ntiles) * TM;
    const f32x4 z = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int f = tid + k * NT, row = f / c4g, ch = (f - row * c4g) * 4;
      if (f < TM * c4g) split_store4(gH, gL, row * RSg + ch, (t0 + row < a.L_out) ? rg[k] : z);
    }
    const int rb = t0 * STRIDE - a.pad + j0 * a.dil;
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const int f = tid + k * NT, row = f / c4h, ch = (f - row * c4h) * 4, gr = rb + row * STRIDE;
      if (f < TM * c4h) {
        const f32x4 v = (pro_h == W2S_PRO_GELU) ? gelu4(rh[k]) : rh[k];
        split_store4(hH, hL, row * RSh + ch, (gr >= 0 && gr < a.L_in) ? v : z);
      }
    }
  };

  int tl = blockIdx.x;
  if (tl < total) prefetch(tl);
  for (; tl < total; tl += gridDim.x) {
    __syncthreads();   // the previous tile's fragment reads are done
    commit(tl);
    if (tl + (int)gridDim.x < total) prefetch(tl + gridDim.x);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < TM / 32; ++s) {
      const int p0 = 32 * s + 8 * g + q4;
      const int hcol = ctile * 16 + 4 * p4;
      const bf16x8 bh = tr_read8(hH + p0 * RSh + hcol, hH + (p0 + 4) * RSh + hcol);
      const bf16x8 bl = tr_read8(hL + p0 * RSh + hcol, hL + (p0 + 4) * RSh + hcol);
#pragma unroll
      for (int i = 0; i < OW; ++i) {
        const int col = (ogrp * OW + i) * 16 + 4 * p4;
        const bf16x8 ah = tr_read8(gH + p0 * RSg + col, gH + (p0 + 4) * RSg + col);
        const bf16x8 al = tr_read8(gL + p0 * RSg + col, gL + (p0 + 4) * RSg + col);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[i], 0, 0, 0);
      }
    }
  }
  constexpr int TILES = OGR * OW * CT;
  float* out = a.slab + (((size_t)blockIdx.x * gridDim.y + blockIdx.y) * TILES) * 256 + lane * 4;
#pragma unroll
  for (int i = 0; i < OW; ++i) st4(out + ((ogrp * OW + i) * CT + ctile) * 256, acc[i]);
}

// sum slabs in a fixed order (16 slab-lanes x sequential chunks, then a fixed LDS tree); decode the fragment
// index to (o, j, c); write torch layout grad[o][c][j] (layout 0) or [o][j][c] (layout 1).  Deterministic.
// One block = 256 consecutive slab elements (64 lanes x float4: a wave reads 1 KB contiguous runs of ONE slab; the earlier
// 16-element x 64-slab-lane form fetched 64-B segments of four different slabs per instruction and ran at 1.3 TB/s).
// 16 waves walk the slabs k = wave, wave + 16, ... with four loads in flight, then a fixed-order LDS tree.
__device__ __forceinline__ void wgrad_reduce_block(float (*red)[260], int blk, const float* __restrict__ slab, int nslab, float* __restrict__ grad,
                                                   int cout, int cin, int taps, int NTO, int NTC, int TAPS_T, int accumulate, int layout) {
  const size_t per = (size_t)cout * cin * taps;  // floats per slab (a multiple of 256: whole raw fragments)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t e0 = (size_t)blk * 256 + lane * 4;
  f32x4 s = {0, 0, 0, 0};
  const float* src = slab + e0;
  int k = wave;
  for (; k + 48 < nslab; k += 64) {
    const f32x4 a0 = ld4(src + (size_t)k * per), a1 = ld4(src + (size_t)(k + 16) * per);
    const f32x4 a2 = ld4(src + (size_t)(k + 32) * per), a3 = ld4(src + (size_t)(k + 48) * per);
    s += (a0 + a1) + (a2 + a3);
  }
  for (; k < nslab; k += 16) s += ld4(src + (size_t)k * per);
  st4(&red[wave][lane * 4], s);
  __syncthreads();
  if (threadIdx.x >= 256) return;
  const int el = threadIdx.x;
  float v = red[0][el];
#pragma unroll
  for (int w = 1; w < 16; ++w) v += red[w][el];
  const size_t idx = (size_t)blk * 256 + el;
  // idx = ((y * TILES + tile) * 64 + lane) * 4 + reg
  const int reg = idx & 3, fl = (idx >> 2) & 63;
  const int TILES = NTO * TAPS_T * NTC;
  const int tile = (idx >> 8) % TILES, y = (idx >> 8) / TILES;
  const int ntg = taps / TAPS_T;
  const int oy = y / ntg, tg = y % ntg;
  const int c_t = tile % NTC, j_t = (tile / NTC) % TAPS_T, i_t = tile / (NTC * TAPS_T);
  const int o = oy * NTO * 16 + i_t * 16 + 4 * (fl >> 4) + reg;
  const int j = tg * TAPS_T + j_t;
  const int c = c_t * 16 + (fl & 15);
  float* d = layout ? grad + ((size_t)o * taps + j) * cin + c : grad + ((size_t)o * cin + c) * taps + j;
  *d = accumulate ? (*d + v) : v;
}
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ slab, int nslab, float* __restrict__ grad, int cout,
                                                           int cin, int taps, int NTO, int NTC, int TAPS_T, int accumulate, int layout) {
  __shared__ float red[16][260];
  wgrad_reduce_block(red, blockIdx.x, slab, nslab, grad, cout, cin, taps, NTO, NTC, TAPS_T, accumulate, layout);
}

// Several layers' slab reductions in ONE launch (a backward pass has ~130 of them, most a few microseconds long):
// the job table travels by value in the kernel arguments; block -> job through the running block counts.
#define W2S_REDUCE_BATCH 48
struct ReduceJobD { const float* slab; float* grad; int nslab, cout, cin, taps, nto, ntc, tapst, flags, blk0; };
struct ReduceBatch { ReduceJobD j[W2S_REDUCE_BATCH]; int njobs; };
__global__ __launch_bounds__(1024) void wgrad_reduce_batch_kernel(ReduceBatch T) {
  __shared__ float red[16][260];
  int k = 0;
  while (k + 1 < T.njobs && (int)blockIdx.x >= T.j[k + 1].blk0) ++k;
  const ReduceJobD& J = T.j[k];
  wgrad_reduce_block(red, blockIdx.x - J.blk0, J.slab, J.nslab, J.grad, J.cout, J.cin, J.taps, J.nto, J.ntc, J.tapst, J.flags & 1, (J.flags >> 1) & 1);
}

struct WgCfg { int nto, ntc, tapst, ts, nw, ot; };
static inline WgCfg wg_cfg(int cin, int cout, int taps, int dil) {
  WgCfg c;
  c.ntc = cin / 16;
  c.ts = 0; c.nw = 4; c.ot = 0;
  if (cin >= 64 && cout >= 64) {  // matrix-core-bound: tile-split kernel, one workgroup per output-channel block
    c.ts = 1;
    c.tapst = (taps == 3 && dil == 1) ? 3 : 1;
    c.nw = (cout >= 128) ? 8 : 4;
    const int budget = 32 / (c.ntc * c.tapst);  // accumulator tiles per wave <= 32
    int ot = (cout / 16) / c.nw;
    while (ot > 1 && (ot > budget || ((cout / 16) / c.nw) % ot)) --ot;
    c.ot = ot;
    c.nto = c.nw * ot;
    return c;
  }
  c.tapst = (taps == 3 && dil == 1 && c.ntc <= 4) ? 3 : 1;
  const int budget = 32 / (c.ntc * c.tapst);  // accumulator tiles per wave <= 32 (128 VGPRs)
  int nto = 8;
  while (nto > 1 && (nto > budget || (cout / 16) % nto)) nto >>= 1;
  c.nto = nto;
  return c;
}

template <int OT, int JT, int CT, int NW, int STRIDE, int PG, int PH>
static int launch_wgrad_ts(const w2s_wgrad_args& a, hipStream_t s) {
  WgradP P;
  P.a = a;
  constexpr int WG = NW * OT * 16;
  if ((NW * 64) % (WG / 4) != 0 && a.pro_g != W2S_PRO_NONE) return W2S_EINVAL;
  int TM = 128;
  auto lds_of = [&](int tm) {
    const int nrh = (JT == 1) ? tm : (tm - 1) * STRIDE + JT;
    return (size_t)(tm * (WG + 4) + nrh * (a.cin + 4)) * sizeof(float);
  };
  while (TM > 16 && lds_of(TM) > 76 * 1024) TM >>= 1;
  while (TM > 16 && TM >= 2 * a.L_out) TM >>= 1;
  P.TM = TM;
  P.ntiles = (a.L_out + TM - 1) / TM;
  P.ntg = a.taps / JT;
  dim3 grid(a.nslab, (a.cout / WG) * P.ntg);
  size_t lds = lds_of(TM);
  auto kern = wgrad_ts_kernel<OT, JT, CT, NW, STRIDE, PG, PH>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return W2S_ELAUNCH;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// positions per tile of wgrad_bf_kernel with one tap per block: the largest of 128 / 64 / 32 whose two windows (bf16 hi + lo) fit 76 KB
__host__ __device__ constexpr int wgbf_tm(int wg, int cin) {
  return (128 * (wg + 8 + cin + 8) * 4 <= 76 * 1024) ? 128 : (64 * (wg + 8 + cin + 8) * 4 <= 76 * 1024) ? 64 : 32;
}
#ifndef W2S_WGRAD_PF
#define W2S_WGRAD_PF 1   // tuning: 0 = every launch on the unpipelined kernel
#endif
template <int OW, int JT, int CT, int NW, int STRIDE, int PG, int PH>
static int launch_wgrad_bf(const w2s_wgrad_args& a, hipStream_t s) {
  WgradP P;
  P.a = a;
  constexpr int WG = (NW / CT) * OW * 16;
  if ((NW * 64) % (WG / 4) != 0 && a.pro_g != W2S_PRO_NONE) return W2S_EINVAL;
  int TM = 128;
  auto lds_of = [&](int tm) {
    const int nrh = (JT == 1) ? tm : (tm - 1) * STRIDE + JT;
    return (size_t)(tm * (WG + 8) + nrh * (a.cin + 8)) * 4;  // two bf16 planes each
  };
  while (TM > 32 && lds_of(TM) > 76 * 1024) TM >>= 1;
  while (TM > 32 && TM >= 2 * a.L_out) TM >>= 1;
  P.TM = TM;
  P.ntiles = (a.L_out + TM - 1) / TM;
  P.ntg = a.taps / JT;
  dim3 grid(a.nslab, (a.cout / WG) * P.ntg);
  size_t lds = lds_of(TM);
  if constexpr (JT == 1 && PG < 0 && W2S_WGRAD_PF) {
    // no gradient-side transform, input side plain or GELU (no statistics): the software-pipelined form (same results bit for bit)
    constexpr int TMC = wgbf_tm(WG, CT * 16);
    const char* off = getenv("W2S_NO_WGRAD_PF");   // tuning / the bit-equality check of tests/gpu_check.py `wgrad` (read per launch on purpose)
    if (!off && a.pro_g == W2S_PRO_NONE && (a.pro_h == W2S_PRO_NONE || a.pro_h == W2S_PRO_GELU) && TM == TMC) {
      auto kpf = wgrad_bf_pf_kernel<OW, CT, NW, STRIDE, TMC>;
      if (lds > 64 * 1024 &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(kpf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return W2S_ELAUNCH;
      hipLaunchKernelGGL(kpf, grid, dim3(NW * 64), lds, s, P);
      W2S_CHECK_LAUNCH();
      return W2S_OK;
    }
  }
  auto kern = wgrad_bf_kernel<OW, JT, CT, NW, STRIDE, PG, PH>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return W2S_ELAUNCH;
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

template <int NTO, int NTC, int TAPS_T, int STRIDE>
static int launch_wgrad(const w2s_wgrad_args& a, hipStream_t s) {
  WgradP P;
  P.a = a;
  int TM = 256;
  auto lds_of = [&](int tm) {
    const int nrh = (TAPS_T == 1) ? tm : (tm - 1) * STRIDE + TAPS_T;
    return (size_t)(tm * (NTO * 16 + 4) + nrh * (a.cin + 4)) * sizeof(float);
  };
  while (TM > 32 && lds_of(TM) > 76 * 1024) TM >>= 1;
  while (TM > 32 && TM >= 2 * a.L_out) TM >>= 1;
  P.TM = TM;
  P.ntiles = (a.L_out + TM - 1) / TM;
  P.ntg = a.taps / TAPS_T;
  const int gy = (a.cout / (NTO * 16)) * P.ntg;
  int gx = a.nslab / 4;
  if (gx < 1) return W2S_EINVAL;
  dim3 grid(gx, gy);
  size_t lds = lds_of(TM);
  auto kern = wgrad_kernel<NTO, NTC, TAPS_T, STRIDE>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return W2S_ELAUNCH;
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

template <int STRIDE>
static int dispatch_wgrad(const w2s_wgrad_args& a, hipStream_t s) {
  const WgCfg c = wg_cfg(a.cin, a.cout, a.taps, a.dil);
  if (c.ts && a.cout % (c.nto * 16)) return W2S_EINVAL;   // the tile-split grid covers whole blocks of nw * ot o-tiles only (callers split cout)
  if (c.ts && a.split_precision && c.nw >= c.ntc && c.nw % c.ntc == 0) {
    // (OW = o-tiles per wave) = nto / (nw / ntc);  same slab layout / reduce as the fp32 tile-split kernel
    const int ow = c.nto / (c.nw / c.ntc);
#define W2S_BF(OW_, JT_, CT_, NW_, PG_, PH_) \
  if (ow == OW_ && c.tapst == JT_ && c.ntc == CT_ && c.nw == NW_ && (PG_ < 0 || (a.pro_g == PG_ && a.pro_h == PH_))) \
    return launch_wgrad_bf<OW_, JT_, CT_, NW_, STRIDE, PG_, PH_>(a, s);
    if constexpr (STRIDE == 1) {
      W2S_BF(8, 3, 8, 8, W2S_PRO_INBWD, W2S_PRO_IN_GELU) W2S_BF(4, 3, 4, 4, W2S_PRO_INBWD, W2S_PRO_IN_GELU) W2S_BF(4, 3, 4, 8, W2S_PRO_INBWD, W2S_PRO_IN_GELU)
      W2S_BF(8, 3, 8, 8, W2S_PRO_INBWD, W2S_PRO_GELU) W2S_BF(4, 3, 4, 4, W2S_PRO_INBWD, W2S_PRO_GELU) W2S_BF(4, 3, 4, 8, W2S_PRO_INBWD, W2S_PRO_GELU)
    }
    if constexpr (STRIDE == 2) {
      W2S_BF(8, 3, 8, 8, W2S_PRO_INBWD_GP, W2S_PRO_IN_GELU) W2S_BF(4, 3, 4, 4, W2S_PRO_INBWD_GP, W2S_PRO_IN_GELU)
    }
    if constexpr (STRIDE <= 2) { W2S_BF(8, 3, 8, 8, -1, -1) W2S_BF(4, 3, 4, 4, -1, -1) W2S_BF(4, 3, 4, 8, -1, -1) }
    W2S_BF(8, 1, 8, 8, -1, -1) W2S_BF(4, 1, 4, 8, -1, -1) W2S_BF(4, 1, 4, 4, -1, -1) W2S_BF(24, 1, 8, 8, -1, -1) W2S_BF(32, 1, 8, 8, -1, -1)
    W2S_BF(16, 1, 8, 8, -1, -1)
#undef W2S_BF
  }
  if (c.ts) {
#define W2S_TS3(CT_, NW_, PG_, PH_) \
  if (c.ot == 1 && c.tapst == 3 && c.ntc == CT_ && c.nw == NW_ && a.pro_g == PG_ && a.pro_h == PH_) \
    return launch_wgrad_ts<1, 3, CT_, NW_, STRIDE, PG_, PH_>(a, s);
#define W2S_TS(OT_, JT_, CT_, NW_) \
  if (c.ot == OT_ && c.tapst == JT_ && c.ntc == CT_ && c.nw == NW_) return launch_wgrad_ts<OT_, JT_, CT_, NW_, STRIDE, -1, -1>(a, s);
    if constexpr (STRIDE == 1) {
      W2S_TS3(8, 8, W2S_PRO_INBWD, W2S_PRO_IN_GELU) W2S_TS3(4, 4, W2S_PRO_INBWD, W2S_PRO_IN_GELU) W2S_TS3(4, 8, W2S_PRO_INBWD, W2S_PRO_IN_GELU)
      W2S_TS3(8, 8, W2S_PRO_INBWD, W2S_PRO_GELU) W2S_TS3(4, 4, W2S_PRO_INBWD, W2S_PRO_GELU) W2S_TS3(4, 8, W2S_PRO_INBWD, W2S_PRO_GELU)
    }
    if constexpr (STRIDE == 2) {
      W2S_TS3(8, 8, W2S_PRO_INBWD_GP, W2S_PRO_IN_GELU) W2S_TS3(4, 4, W2S_PRO_INBWD_GP, W2S_PRO_IN_GELU)
    }
    if constexpr (STRIDE <= 2) { W2S_TS(1, 3, 8, 8) W2S_TS(1, 3, 4, 4) W2S_TS(1, 3, 4, 8) W2S_TS(1, 3, 8, 4) }
    W2S_TS(1, 1, 8, 8) W2S_TS(1, 1, 4, 8) W2S_TS(1, 1, 4, 4) W2S_TS(3, 1, 8, 8) W2S_TS(4, 1, 8, 8) W2S_TS(2, 1, 8, 8) W2S_TS(1, 1, 8, 4)
#undef W2S_TS
#undef W2S_TS3
    return W2S_EINVAL;
  }
#define W2S_WG(NTO_, NTC_, TT_) \
  if (c.nto == NTO_ && c.ntc == NTC_ && c.tapst == TT_) return launch_wgrad<NTO_, NTC_, TT_, STRIDE>(a, s);
  if constexpr (STRIDE <= 2) {
    W2S_WG(1, 1, 3) W2S_WG(2, 1, 3) W2S_WG(2, 2, 3) W2S_WG(4, 2, 3) W2S_WG(2, 4, 3)
    // shapes only the generic training path reaches (generic.py: feature sizes 16 / 32 / 64, SleepPPGNet's pieces): the rest of wg_cfg's range
    W2S_WG(4, 1, 3) W2S_WG(8, 1, 3) W2S_WG(1, 2, 3) W2S_WG(1, 4, 3)
    W2S_WG(4, 1, 1) W2S_WG(8, 1, 1) W2S_WG(1, 2, 1) W2S_WG(8, 2, 1) W2S_WG(1, 4, 1) W2S_WG(2, 4, 1) W2S_WG(1, 8, 1) W2S_WG(2, 8, 1)
  }
  W2S_WG(1, 1, 1) W2S_WG(2, 1, 1) W2S_WG(2, 2, 1) W2S_WG(4, 2, 1) W2S_WG(4, 4, 1) W2S_WG(8, 4, 1) W2S_WG(4, 8, 1)
#undef W2S_WG
  return W2S_EINVAL;
}

int w2s_wgrad_wide_try(const w2s_wgrad_args& a, hipStream_t s, int dry);   // wgrad_wide.hip

// grid.x the caller should not exceed: the role-split kernel of the >= 64-channel k=3 layers runs one workgroup per CU
extern "C" int w2s_wgrad_max_blocks(const w2s_wgrad_args* ap) {
  if (!ap) return W2S_EINVAL;
  return w2s_wgrad_wide_try(*ap, nullptr, 1) == 0 ? 256 : 512;
}

// slabs one grid.x block of THIS launch writes (the role-split kernel: 1; otherwise w2s_wgrad_slabs_per_block)
extern "C" int w2s_wgrad_slabs_per_block_of(const w2s_wgrad_args* ap) {
  if (!ap) return W2S_EINVAL;
  return w2s_wgrad_wide_try(*ap, nullptr, 1) == 0 ? 1 : (wg_cfg(ap->cin, ap->cout, ap->taps, ap->dil).ts ? 1 : 4);
}

extern "C" int w2s_wgrad(const w2s_wgrad_args* ap, void* stream) {
  if (!ap) return W2S_EINVAL;
  const w2s_wgrad_args& a = *ap;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a.cin < 16 || a.cin > 128 || (a.cin & (a.cin - 1)) || (a.cout & 15) || !a.g || !a.x || !a.slab) return W2S_EINVAL;
  if ((size_t)a.L_out * (size_t)a.ldg * 4 >= ((size_t)1 << 32) || (size_t)a.L_in * (size_t)a.ldx * 4 >= ((size_t)1 << 32)) return W2S_EINVAL;  // 32-bit lane offsets
  if (a.pro_g >= W2S_PRO_IN_GELU && !a.g_stats) return W2S_EINVAL;
  if (a.pro_g >= W2S_PRO_INBWD && (!a.g_bstats || !a.g2)) return W2S_EINVAL;
  if (a.pro_h >= W2S_PRO_IN_GELU && !a.x_stats) return W2S_EINVAL;
  if (a.pro_h < 0 || a.pro_h > W2S_PRO_AFFINE + 4 || a.pro_g < 0 || a.pro_g > W2S_PRO_AFFINE_BWD + 4 ||
      (a.pro_g > W2S_PRO_INBWD_GP && a.pro_g < W2S_PRO_AFFINE_BWD))
    return W2S_EINVAL;   // input side: the forward transforms; gradient side: the backward ones
  if (const int rc = w2s_wgrad_wide_try(a, s, 0); rc != 1) return rc;
  if (a.stride == 1) return dispatch_wgrad<1>(a, s);
  if (a.stride == 2) return dispatch_wgrad<2>(a, s);
  if (a.stride == 4) return dispatch_wgrad<4>(a, s);
  return W2S_EINVAL;
}

extern "C" int w2s_wgrad_slabs_per_block(int cin, int cout, int taps, int dil) { return wg_cfg(cin, cout, taps, dil).ts ? 1 : 4; }

extern "C" int w2s_wgrad_grid_y(int cin, int cout, int taps, int dil) {
  const WgCfg c = wg_cfg(cin, cout, taps, dil);
  return (cout / (c.nto * 16)) * (taps / c.tapst);
}

extern "C" int w2s_wgrad_reduce(const float* slab, int nslab, float* grad, int cout, int cin, int taps, int dil, int accumulate,
                                int layout, void* stream) {
  if (!slab || !grad || nslab <= 0) return W2S_EINVAL;
  const WgCfg c = wg_cfg(cin, cout, taps, dil);
  const size_t per = (size_t)cout * cin * taps;
  if (per % 256) return W2S_EINVAL;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(per / 256)), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), slab,
                     nslab, grad, cout, cin, taps, c.nto, c.ntc, c.tapst, accumulate, layout);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

extern "C" int w2s_wgrad_reduce_batch(const w2s_reduce_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs <= 0) return W2S_EINVAL;
  for (int base = 0; base < njobs; base += W2S_REDUCE_BATCH) {
    ReduceBatch T;
    T.njobs = (njobs - base < W2S_REDUCE_BATCH) ? njobs - base : W2S_REDUCE_BATCH;
    int blocks = 0;
    for (int i = 0; i < T.njobs; ++i) {
      const w2s_reduce_job& q = jobs[base + i];
      if (!q.slab || !q.grad || q.nslab <= 0) return W2S_EINVAL;
      const WgCfg c = wg_cfg(q.cin, q.cout, q.taps, q.dil);
      const size_t per = (size_t)q.cout * q.cin * q.taps;
      T.j[i] = ReduceJobD{q.slab, q.grad, q.nslab, q.cout, q.cin, q.taps, c.nto, c.ntc, c.tapst, (q.accumulate ? 1 : 0) | (q.layout ? 2 : 0), blocks};
      if (per % 256) return W2S_EINVAL;
      blocks += (int)(per / 256);
    }
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), T);
    W2S_CHECK_LAUNCH();
  }
  return W2S_OK;
}

// all layers' repacks (fp32 kernel layouts + bf16 hi/lo planes) of one optimiser step in a few launches
#define W2S_REPACK_BATCH 48
struct RepackJobD { const float* w; float* fwd; float* bwd; __bf16* fh; __bf16* fl; __bf16* bh; __bf16* bl; int cout, cin, taps, blk0; };
struct RepackBatch { RepackJobD j[W2S_REPACK_BATCH]; int njobs; };
__global__ __launch_bounds__(256) void repack_batch_kernel(RepackBatch T) {
  int k = 0;
  while (k + 1 < T.njobs && (int)blockIdx.x >= T.j[k + 1].blk0) ++k;
  const RepackJobD& J = T.j[k];
  const size_t n = (size_t)J.cout * J.cin * J.taps;
  const size_t idx = (size_t)(blockIdx.x - J.blk0) * 256 + threadIdx.x;
  if (idx >= n) return;
  const int j = idx % J.taps;
  const int c = (idx / J.taps) % J.cin;
  const int o = idx / ((size_t)J.taps * J.cin);
  const float v = J.w[idx];
  const size_t df = ((size_t)o * J.taps + j) * J.cin + c, db = ((size_t)c * J.taps + j) * J.cout + o;
  if (J.fwd) J.fwd[df] = v;
  if (J.bwd) J.bwd[db] = v;
  if (J.fh || J.bh) {
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    if (J.fh) { const size_t d = w2s_frag_index(o, j * J.cin + c, J.cin == 16 ? ((J.taps + 1) / 2) * 32 : J.taps * J.cin); J.fh[d] = h; J.fl[d] = l; }
    if (J.bh) { const size_t d = w2s_frag_index(c, j * J.cout + o, J.taps * J.cout); J.bh[d] = h; J.bl[d] = l; }
  }
}
extern "C" int w2s_repack_batch(const w2s_repack_job* jobs, int njobs, void* stream) {
  if (!jobs || njobs <= 0) return W2S_EINVAL;
  for (int base = 0; base < njobs; base += W2S_REPACK_BATCH) {
    RepackBatch T;
    T.njobs = (njobs - base < W2S_REPACK_BATCH) ? njobs - base : W2S_REPACK_BATCH;
    int blocks = 0;
    for (int i = 0; i < T.njobs; ++i) {
      const w2s_repack_job& q = jobs[base + i];
      if (!q.w || (q.fwd_hi && !q.fwd_lo) || (q.bwd_hi && !q.bwd_lo)) return W2S_EINVAL;
      T.j[i] = RepackJobD{q.w, q.fwd, q.bwd, static_cast<__bf16*>(q.fwd_hi), static_cast<__bf16*>(q.fwd_lo), static_cast<__bf16*>(q.bwd_hi),
                          static_cast<__bf16*>(q.bwd_lo), q.cout, q.cin, q.taps, blocks};
      blocks += (int)(((size_t)q.cout * q.cin * q.taps + 255) / 256);
    }
    hipLaunchKernelGGL(repack_batch_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), T);
    W2S_CHECK_LAUNCH();
  }
  return W2S_OK;
}

// torch [cout][cin][taps] -> fwd [cout][taps][cin] and bwd [cin][taps][cout]
__global__ void repack_kernel(const float* __restrict__ w, float* __restrict__ fwd, float* __restrict__ bwd, int cout, int cin, int taps) {
  const size_t n = (size_t)cout * cin * taps;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int j = idx % taps;
  const int c = (idx / taps) % cin;
  const int o = idx / ((size_t)taps * cin);
  const float v = w[idx];
  if (fwd) fwd[((size_t)o * taps + j) * cin + c] = v;
  if (bwd) bwd[((size_t)c * taps + j) * cout + o] = v;
}

// torch [cout][cin][taps] fp32 -> bf16 hi/lo planes (w = hi + lo up to 2^-17 relative) of the forward GEMM operand
// A_f[o][j*cin + c] and the data-gradient operand A_b[c][j*cout + o], both in the fragment-major order of w2s_frag_index
// (needs cin, cout multiples of 32 / 16 as the split-precision path does), for conv_cl.
__global__ void repack_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ fh, __bf16* __restrict__ fl, __bf16* __restrict__ bh,
                                   __bf16* __restrict__ bl, int cout, int cin, int taps) {
  const size_t n = (size_t)cout * cin * taps;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int j = idx % taps;
  const int c = (idx / taps) % cin;
  const int o = idx / ((size_t)taps * cin);
  const float v = w[idx];
  const __bf16 h = (__bf16)v;
  const __bf16 l = (__bf16)(v - (float)h);
  if (fh) { const size_t d = w2s_frag_index(o, j * cin + c, cin == 16 ? ((taps + 1) / 2) * 32 : taps * cin); fh[d] = h; fl[d] = l; }
  if (bh) { const size_t d = w2s_frag_index(c, j * cout + o, taps * cout); bh[d] = h; bl[d] = l; }
}

extern "C" int w2s_repack_bf16(const float* w, void* fwd_hi, void* fwd_lo, void* bwd_hi, void* bwd_lo, int cout, int cin, int taps,
                               void* stream) {
  if (!w || (fwd_hi && !fwd_lo) || (bwd_hi && !bwd_lo)) return W2S_EINVAL;
  const size_t n = (size_t)cout * cin * taps;
  hipLaunchKernelGGL(repack_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                     static_cast<__bf16*>(fwd_hi), static_cast<__bf16*>(fwd_lo), static_cast<__bf16*>(bwd_hi), static_cast<__bf16*>(bwd_lo),
                     cout, cin, taps);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

extern "C" int w2s_repack(const float* w, float* fwd, float* bwd, int cout, int cin, int taps, void* stream) {
  if (!w) return W2S_EINVAL;
  const size_t n = (size_t)cout * cin * taps;
  hipLaunchKernelGGL(repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, fwd, bwd,
                     cout, cin, taps);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
