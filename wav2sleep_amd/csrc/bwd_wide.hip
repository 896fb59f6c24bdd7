// Backward of one k=3 encoder convolution (stride 1, or the block's stride-2 conv3) with 64 gradient-side channels in ONE pass: data gradient + weight gradient +
// GELU' + backward statistics from the same staged tiles -- the >= 64-channel counterpart of bwd_fused_bf_kernel, in the persistent
// role-split form of conv_wide_kernel / wgrad_wide_kernel (round 3).
//
//   gy     = rstd_k * (g - s1 - n_k s2),  n_k = (y_k - mean_k) rstd_k                      (W2S_PRO_INBWD of the incoming gradient g)
//   n_in   = (xin - mean_in) rstd_in  (st_in given: conv2, xin = y_{k-1})   or   xin   (conv1, xin = the previous block's stored pre-activation)
//   h      = GELU(n_in),  gp = GELU'(n_in)                                                 (ONE erf evaluation for both)
//   dgrad :  gout[t][c] = (sum_{j,o} W[o][c][j] gy[t+1-j][o]  [+ add_even[t/2][c] at even t]) * gp[t][c];  partial sums of gout, gout*n_in
//   wgrad :  dW[o][j][c] = sum_t gy[t][o] h[t+j-1][c]
// UP2 (the stride-2 conv3, g = dL/d(block pre-activation) at half the length): gy = instance-norm backward of g * GELU'(n_k) (W2S_PRO_INBWD_GP);
//   dgrad: gout[2u] = W_1^T gy[u], gout[2u+1] = W_2^T gy[u] + W_0^T gy[u+1];  wgrad: dW[o][j][c] = sum_u gy[u][o] h[2u+j-1][c]
// (written for symmetric padding, pad = 1.  Causal padding, pad = 2 -- the forward taps at t+j-2 / 2u+j-2, blocks.py:150-152,178-182 --
//  shifts the windows: gy[t+2-j] and h[t+j-2] in the stride-1 form; stride 2 (template CZ): gout[2u] = W_2^T gy[u] + W_0^T gy[u+1],
//  gout[2u+1] = W_1^T gy[u+1], dW from h[2u+j-2])
//
// Before: conv_wide (dgrad) and wgrad_wide each streamed g, y_k and xin, each ran the instance-norm backward on g, one evaluated GELU'
// and the other GELU of the same n_in (two erf's) -- 7 tensor passes and twice the producer arithmetic for 4 passes' worth of work
// (profiles/r02: traffic 1.10-1.20 x algorithmic, and the producers' VALU, not HBM, set the pace of both).  Here:
//   * 4 PRODUCER waves stream the raw rows of g, y_k and xin of the next PD tiles into registers (unconditional clamped loads), apply the
//     on-load transforms with the per-sample statistics from an LDS table, and write tile i + 1 into the other LDS buffer: gy as bf16
//     (hi, lo) planes [TM + 2 rows][OC], h as bf16 (hi, lo) planes [TM + 2 rows][HC], gp as fp32 [TM][HC];
//   * NWC CONSUMER waves: wave w owns input-side channels [16 (w % CI), +16) of a position group for the data gradient -- its 16 x K
//     weight slice (K = 3 OC) lives in registers for the whole launch, B operands are ds_read_b128 of the row-major gy planes -- and
//     IB x CB (cout tile, cin tile) pairs x 3 taps of weight-gradient accumulators, fed by transposing LDS reads of the SAME planes
//     (ds_read_b64_tr_b16: the contraction index is the position); epilogue: * gp, statistics partials, store;
//   * one barrier per tile; one slab per workgroup (raw-fragment layout of wgrad_wide_kernel), summed by w2s_wgrad_reduce.
#include <type_traits>
#include "conv_cl.inl"

struct BwdWideP {
  const float* g; const float* y; const float* st_k; const float* bst_k;
  const float* xin; const float* st_in; const float* add_even;
  const __bf16* w_hi; const __bf16* w_lo;   // data-gradient operand: [cin][taps][cout] as fragment-major planes (w2s_repack_batch bwd_hi / bwd_lo)
  float* gout; float* part; float* slab;
  // conv1 of a block (stride 1): fold the PREVIOUS block's conv3-backward pre-pass (w2s_gp_stats) in, as w2s_bwd_fused does: y3p = that block's
  // pre-norm conv3 output [B][L][HC] (same positions as gout), st3p = its (mean, rstd); `part` then holds the sums of gout*GELU'(n3) and *n3
  const float* y3p; const float* st3p;
  // residual fold (RD; conv1 of a block, stride 1, no add_even): the block's 1x1/stride-2 residual branch in the same pass, as w2s_bwd_fused
  // does for <= 32 channels: gout additionally receives Wd^T gpre[t/2] at even t before the GELU' factor, and slab_d the weight gradient
  // sum_u gpre[u] x h[2u].  gpre = dL/d(block pre-activation) [B][L/2][OC]; wd_hi / wd_lo = data-gradient planes of the downsample weight
  const float* gpre; const __bf16* wd_hi; const __bf16* wd_lo; float* slab_d;
  int B, L, Lg, ntiles;   // L: input-side length; Lg: gradient-side length (L, or L / 2 for the stride-2 form)
  int pad;                // the forward conv's left padding: 1 = symmetric, 2 = causal (blocks.py:150-152,178-182)
};

typedef __bf16 wbbf16x4v __attribute__((__vector_size__(4 * sizeof(__bf16))));
#define wb_split_store4 split_store4   /* w2s_common.h: the explicit bit form (10 instead of 14 instructions per four elements) */
__device__ __forceinline__ bf16x8 wb_tr8(const __bf16* p0, const __bf16* p1) {
  typedef __attribute__((address_space(3))) wbbf16x4v* lds_p;
  wbbf16x4v a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p0));
  wbbf16x4v b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p1));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// CO / CI: 16-channel tiles on the gradient / input side; HST: xin carries statistics (conv2); MT: 16-position m-tiles per tile;
// NWC consumer waves = (CO / IB) x (CI / CB) weight-gradient owners = CI x (NWC / CI) data-gradient owners; PD: producer prefetch depth
template <int CO, int CI, int HST, int MT, int NWC, int IB, int CB, int PD, int UP2, int RD, int CZ>
__device__ __forceinline__ void bwd_wide_body(const BwdWideP& P) {
  static_assert(!RD || (!UP2 && !HST && MT % 2 == 0), "residual fold: the stride-1 conv1 (its input is a stored pre-activation)");
  extern __shared__ f32x4 smem4[];
  static_assert((CO / IB) * (CI / CB) == NWC && NWC % CI == 0 && MT % (NWC / CI) == 0, "consumer wave grid");
  static_assert(!UP2 || (NWC == CI && MT % 2 == 0 && (8 * MT) % 32 == 0), "stride-2 form: one position group, even / odd m-tiles");
  static_assert(!CZ || UP2, "CZ: the stride-2 form with causal padding (stride 1 takes the padding at run time)");
  constexpr int OC = CO * 16, HC = CI * 16, TM = 16 * MT, NPT = 256, NR = TM + 2;   // h window rows: positions t0 - pad .. t0 - pad + TM + 1
  constexpr int NRG = UP2 ? TM / 2 + 1 : TM + 2;                                     // gy window rows: t0 - 1 .. t0 + TM (UP2: t0/2 .. t0/2 + TM/2)
  constexpr int TG = UP2 ? TM / 2 : TM;                                              // gradient-side positions per tile
  constexpr int RSg = OC + 16, RSh = HC + 8;                                         // bf16 elements per LDS row
  constexpr int RSp = HC + 4;                                                        // floats per row of the gp plane
  constexpr int NRP = RD ? TM / 2 + 1 : 0;                                           // gpre rows of the tile + one all-zero row (odd output positions)
  constexpr int BUFH = 2 * NRG * RSg + 2 * NR * RSh + 2 * NRP * RSg;                 // bf16 elements: gy hi, gy lo, h hi, h lo[, gpre hi, gpre lo]
  constexpr int BUFB = BUFH * 2 + TM * RSp * 4;                                      // bytes of one buffer (gp plane behind the bf16 planes)
  static_assert(BUFB % 16 == 0, "buffer alignment");
  constexpr int PG = NWC / CI, MTW = MT / PG;                                        // position groups / m-tiles per wave of the data gradient
  constexpr int QN = OC / 32, KS = 3 * QN;                                           // data-gradient K steps: ks = tap * QN + q
  char* lds = reinterpret_cast<char*>(smem4);
  float* stL = reinterpret_cast<float*>(lds + 2 * BUFB);   // [B][OC][2] (mean, rstd), [B][OC][2] backward sums, then [B][HC][2] (HST)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total = P.B * P.ntiles;
  const int G = (int)gridDim.x;
  // tiles of this workgroup: a contiguous run of the (sample, tile) list (blocked; w2s_common.h "Statistics finalisation")
  const W2SRun wrun = w2s_block_part(total, G, blockIdx.x);
  const int first = wrun.first;
  const int run_b0 = first / P.ntiles, run_t0 = first - run_b0 * P.ntiles;   // the run's first (sample, tile): the one division of the launch
  for (int i = tid; i < P.B * OC * 2; i += 64 * (NWC + 4)) {
    stL[i] = P.st_k[i];
    stL[P.B * OC * 2 + i] = P.bst_k[i];
  }
  if (HST)
    for (int i = tid; i < P.B * HC * 2; i += 64 * (NWC + 4)) stL[P.B * OC * 4 + i] = P.st_in[i];
  if (RD) {   // the zero row of the gpre planes, both buffers (the producers only ever write rows 0 .. TM/2 - 1)
    for (int i = tid; i < 2 * 2 * RSg; i += 64 * (NWC + 4)) {
      const int buf = i / (2 * RSg), k = i % (2 * RSg);
      __bf16* pz = reinterpret_cast<__bf16*>(lds + buf * BUFB) + 2 * NRG * RSg + 2 * NR * RSh + (k / RSg) * NRP * RSg + (TM / 2) * RSg + (k % RSg);
      *pz = (__bf16)0.f;
    }
  }
  __syncthreads();

  const int nt_wg = wrun.count;                                   // >= 1 (the grid never exceeds the tile count)
  const int NI = ((nt_wg + 1 + PD - 1) / PD) * PD;                // barrier rounds, padded to whole prefetch cycles
  const int L = P.L, Lg = P.Lg;
  const int PL = P.pad;   // window row 0: h position t0 - pad, gy position t0 + pad - 2 (stride 1; stride 2: t0 / 2)

  if (wave >= NWC) {
    // ================================================= producer waves =================================================
#ifndef W2S_BWW_PPRIO
#define W2S_BWW_PPRIO 1   // issue priority of the producer waves (the kernel is producer-bound: in-kernel stamps, docs/lab_notes_r4.md section 11)
#endif
    if (W2S_BWW_PPRIO) __builtin_amdgcn_s_setprio(W2S_BWW_PPRIO);
    const int pt = tid - 64 * NWC;
    constexpr int c4g = OC / 4, rsg = NPT / c4g, NG = (NRG + rsg - 1) / rsg;
    constexpr int c4h = HC / 4, rsh = NPT / c4h, NH = (NR + rsh - 1) / rsh;
    const int gch = (pt % c4g) * 4, grow0 = pt / c4g;
    const int hch = (pt % c4h) * 4, hrow0 = pt / c4h;
    constexpr int NP = RD ? (TM / 2 + rsg - 1) / rsg : 1;
    f32x4 rg[PD][NG] = {}, ry[PD][NG] = {}, rh[PD][NH] = {}, rp[RD ? PD : 1][NP] = {};
    // (the run position is looked up per row: hoisting it to once per tile was measured 5 % SLOWER here -- the scalar work spaces the loads)
    auto load_p = [&](auto SET, int i, int k) {
      constexpr int S = decltype(SET)::value;
      int b, tile_;
      w2s_run_pos(run_b0, run_t0, P.ntiles, min(i, nt_wg - 1), b, tile_);
      const int t0 = tile_ * TM;
      const int row = min(grow0 + k * rsg, TM / 2 - 1), gr = min(t0 / 2 + row, (L >> 1) - 1);
      rp[S][k] = ld4o(P.gpre + (size_t)b * (L >> 1) * OC, (unsigned)gr * OC + gch);
    };
    auto load_g = [&](auto SET, int i, int k) {
      constexpr int S = decltype(SET)::value;
      int b, tile_;
      w2s_run_pos(run_b0, run_t0, P.ntiles, min(i, nt_wg - 1), b, tile_);
      const int t0 = tile_ * TM;
      const int row = min(grow0 + k * rsg, NRG - 1), gr = min(max((UP2 ? t0 / 2 : t0 + PL - 2) + row, 0), Lg - 1);
      const unsigned off = (unsigned)gr * OC + gch;
      rg[S][k] = ld4o(P.g + (size_t)b * Lg * OC, off);
      ry[S][k] = ld4o(P.y + (size_t)b * Lg * OC, off);
    };
    auto load_h = [&](auto SET, int i, int k) {
      constexpr int S = decltype(SET)::value;
      int b, tile_;
      w2s_run_pos(run_b0, run_t0, P.ntiles, min(i, nt_wg - 1), b, tile_);
      const int t0 = tile_ * TM;
      const int row = min(hrow0 + k * rsh, NR - 1), gr = min(max(t0 - PL + row, 0), L - 1);
      rh[S][k] = ld4o(P.xin + (size_t)b * L * HC, (unsigned)gr * HC + hch);
    };
    auto stage = [&](auto SET, int i) {
      constexpr int S = decltype(SET)::value;
      const bool live = i < nt_wg;   // uniform; padding rounds only keep the load queue regular
      int b, tile_;
      w2s_run_pos(run_b0, run_t0, P.ntiles, min(i, nt_wg - 1), b, tile_);
      const int t0 = tile_ * TM;
      __bf16* gH = reinterpret_cast<__bf16*>(lds + (i & 1) * BUFB);
      __bf16* gL = gH + NRG * RSg;
      __bf16* hH = gL + NRG * RSg;
      __bf16* hL = hH + NR * RSh;
      __bf16* pH = hL + NR * RSh;
      __bf16* pL = pH + NRP * RSg;
      float* gpL = reinterpret_cast<float*>(pL + NRP * RSg);
      // instance-norm backward as two fused multiply-adds per element (coefficients per tile; see bwd_fused.hip `commit`):
      //   stride 1: gy = r g + (-r^2 s2) y + r (r s2 m - s1);   stride 2: n = r y + (-m r), gy = (r g) GELU'(n) + n (-r s2) + (-r s1)
      f32x4 cA, cB, cC, cD;
      {
        const float* st = stL + (b * OC + gch) * 2;
        const f32x4 s01 = ld4(st), s23 = ld4(st + 4);
        const f32x4 pm = {s01.x, s01.z, s23.x, s23.z}, pr = {s01.y, s01.w, s23.y, s23.w};
        const float* bs = stL + ((P.B + b) * OC + gch) * 2;
        const f32x4 b01 = ld4(bs), b23 = ld4(bs + 4);
        const f32x4 ps1 = {b01.x, b01.z, b23.x, b23.z}, ps2 = {b01.y, b01.w, b23.y, b23.w};
        cA = pr;
        if (UP2) { cB = -(pm * pr); cC = -(pr * ps2); cD = -(pr * ps1); }
        else { cB = -(pr * pr * ps2); cC = pr * (pr * ps2 * pm - ps1); cD = cC; }
      }
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        const int row = grow0 + k * rsg;
        const f32x4 v1 = rg[S][k], v2 = ry[S][k];
        load_g(SET, i + PD, k);
        if (live && row < NRG) {
          f32x4 tv;
#ifdef W2S_BWW_NOARITH   // diagnostic builds (numerics wrong on purpose): the producers without their transform arithmetic
          tv = v1 + v2 * cA;
#else
          if (UP2) {
            const f32x4 n = fma4(v2, cA, cB);
            tv = fma4(v1 * cA, gelu_grad4(n), fma4(n, cC, cD));
          } else {
            tv = fma4(cA, v1, fma4(cB, v2, cC));
          }
#endif
#ifndef W2S_BWW_NOLDSW
          wb_split_store4(gH, gL, row * RSg + gch, tv);
#else
          if (tv.x == 123.f) wb_split_store4(gH, gL, row * RSg + gch, tv);
#endif
        }
      }
      f32x4 hb = {0, 0, 0, 0}, hr = {1, 1, 1, 1};   // n = x rstd + (-mean rstd)
      if (HST) {
        const float* st = stL + P.B * OC * 4 + (b * HC + hch) * 2;
        const f32x4 s01 = ld4(st), s23 = ld4(st + 4);
        hr = (f32x4){s01.y, s01.w, s23.y, s23.w};
        hb = -((f32x4){s01.x, s01.z, s23.x, s23.z} * hr);
      }
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        const int row = hrow0 + k * rsh;
        const f32x4 v = rh[S][k];
        load_h(SET, i + PD, k);
        if (live && row < NR) {
          f32x4 hv, gpv;
#ifdef W2S_BWW_NOARITH
          hv = fma4(v, hr, hb); gpv = hv;
#else
          gelu_both4(fma4(v, hr, hb), hv, gpv);
#endif
#ifndef W2S_BWW_NOLDSW
          wb_split_store4(hH, hL, row * RSh + hch, hv);
          if (row >= PL && row < TM + PL) st4(gpL + (row - PL) * RSp + hch, gpv);
#else
          if (hv.x == 123.f) { wb_split_store4(hH, hL, row * RSh + hch, hv); st4(gpL + (row - PL) * RSp + hch, gpv); }
#endif
        }
      }
      if constexpr (RD) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const int row = grow0 + k * rsg;
          const f32x4 v = rp[S][k];
          load_p(SET, i + PD, k);
          if (live && row < TM / 2) wb_split_store4(pH, pL, row * RSg + gch, v);
        }
      }
      // rows outside the sample (the conv's zero padding; a sample's first / last tile only -- uniform branch): loaded from clamped
      // addresses and transformed like any row above, now overwritten with zeros by the lanes that stored them (bwd_fused.hip `commit`)
      const int rbg = UP2 ? t0 / 2 : t0 + PL - 2;
      if (live && (rbg < 0 || rbg + NRG > Lg || t0 - PL < 0 || t0 - PL + NR > L)) {
#pragma unroll
        for (int k = 0; k < NG; ++k) {
          const int row = grow0 + k * rsg, gr = rbg + row;
          if (row < NRG && (gr < 0 || gr >= Lg)) { zero_store4(gH, row * RSg + gch); zero_store4(gL, row * RSg + gch); }
        }
#pragma unroll
        for (int k = 0; k < NH; ++k) {
          const int row = hrow0 + k * rsh, gr = t0 - PL + row;
          if (row < NR && (gr < 0 || gr >= L)) { zero_store4(hH, row * RSh + hch); zero_store4(hL, row * RSh + hch); }
        }
        if constexpr (RD) {
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            const int row = grow0 + k * rsg;
            if (row < TM / 2 && t0 / 2 + row >= (L >> 1)) { zero_store4(pH, row * RSg + gch); zero_store4(pL, row * RSg + gch); }
          }
        }
      }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      load_g(I0{}, 0, k);
      if constexpr (PD > 1) load_g(I1{}, 1, k);
      if constexpr (PD > 2) load_g(I2{}, 2, k);
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      load_h(I0{}, 0, k);
      if constexpr (PD > 1) load_h(I1{}, 1, k);
      if constexpr (PD > 2) load_h(I2{}, 2, k);
    }
    if constexpr (RD) {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        load_p(I0{}, 0, k);
        if constexpr (PD > 1) load_p(I1{}, 1, k);
        if constexpr (PD > 2) load_p(I2{}, 2, k);
      }
    }
#ifdef W2S_WIDE_STAMP   // diagnostic build only (tools/altlib.sh): cycles of block 0's first producer wave in stage / at the barrier -> part[4..7]
    {
      unsigned long long ts = 0, tb = 0, c0, c1, c2;
      for (int it = 0; it < NI; it += PD) {
        c0 = __builtin_amdgcn_s_memtime(); stage(I0{}, it); c1 = __builtin_amdgcn_s_memtime(); __syncthreads(); c2 = __builtin_amdgcn_s_memtime(); ts += c1 - c0; tb += c2 - c1;
        if constexpr (PD > 1) { c0 = __builtin_amdgcn_s_memtime(); stage(I1{}, it + 1); c1 = __builtin_amdgcn_s_memtime(); __syncthreads(); c2 = __builtin_amdgcn_s_memtime(); ts += c1 - c0; tb += c2 - c1; }
        if constexpr (PD > 2) { c0 = __builtin_amdgcn_s_memtime(); stage(I2{}, it + 2); c1 = __builtin_amdgcn_s_memtime(); __syncthreads(); c2 = __builtin_amdgcn_s_memtime(); ts += c1 - c0; tb += c2 - c1; }
      }
      if (blockIdx.x == 0 && tid == 64 * NWC && P.part) { P.part[4] = (float)ts; P.part[5] = (float)tb; P.part[6] = (float)NI; P.part[7] = (float)nt_wg; }
      return;
    }
#endif
    for (int it = 0; it < NI; it += PD) {
      stage(I0{}, it);
      __syncthreads();
      if constexpr (PD > 1) { stage(I1{}, it + 1); __syncthreads(); }
      if constexpr (PD > 2) { stage(I2{}, it + 2); __syncthreads(); }
    }
    return;
  }

  // =================================================== consumer waves ===================================================
  const int r = lane & 15, g = lane >> 4, q4 = r >> 2, p4 = r & 3;
  const int wi = wave / (CI / CB), wc = wave % (CI / CB);   // weight gradient: cout tiles [wi IB, +IB) x cin tiles [wc CB, +CB)
  const int dn = wave % CI, dg = wave / CI;                 // data gradient: input-side channel tile dn, position group dg
  f32x4 accw[IB][3][CB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int c = 0; c < CB; ++c) accw[i][j][c] = (f32x4){0, 0, 0, 0};
  bf16x8 wh[KS], wl[KS];   // this wave's 16 x K data-gradient weight slice, once per launch: fragment-major planes [HC/16][KS][64 lanes][8]
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const size_t wo = ((size_t)dn * KS + ks) * 512 + lane * 8;
    wh[ks] = *reinterpret_cast<const bf16x8*>(P.w_hi + wo);
    wl[ks] = *reinterpret_cast<const bf16x8*>(P.w_lo + wo);
  }
  bf16x8 dh[RD ? QN : 1], dl[RD ? QN : 1];   // RD: the 16 x OC slice of Wd^T (planes [HC/16][QN][64 lanes][8])
  f32x4 accd[RD ? IB : 1][RD ? CB : 1];
  if (RD) {
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const size_t wo = ((size_t)dn * QN + q) * 512 + lane * 8;
      dh[q] = *reinterpret_cast<const bf16x8*>(P.wd_hi + wo);
      dl[q] = *reinterpret_cast<const bf16x8*>(P.wd_lo + wo);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int c = 0; c < CB; ++c) accd[i][c] = (f32x4){0, 0, 0, 0};
  }
  const int ch0 = dn * 16 + 4 * g;   // this lane's 4 consecutive output channels (D fragment: position r, channels 4g .. 4g+3)
  __syncthreads();                   // round 0 of the producers: the first windows are in buffer 0
#ifdef W2S_WIDE_STAMP
  unsigned long long tk = 0, te = 0, tg = 0, tw = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
#endif
  for (int it = 0; it < NI - 1; ++it) {
    if (it >= nt_wg) { __syncthreads(); continue; }   // padding rounds of the producers' prefetch cycle
#ifdef W2S_WIDE_STAMP
    c0 = __builtin_amdgcn_s_memtime();
#endif
    int b, tile;
    w2s_run_pos(run_b0, run_t0, P.ntiles, it, b, tile);
    const int t0 = tile * TM;
    const __bf16* gH = reinterpret_cast<const __bf16*>(lds + (it & 1) * BUFB);
    const __bf16* gL = gH + NRG * RSg;
    const __bf16* hH = gL + NRG * RSg;
    const __bf16* hL = hH + NR * RSh;
    const __bf16* pH = hL + NR * RSh;
    const __bf16* pL = pH + NRP * RSg;
    const float* gpL = reinterpret_cast<const float*>(pL + NRP * RSg);
    // epilogue operands of THIS tile, issued now so that their latency hides behind the MFMA loops: raw xin (statistics) and add_even
    f32x4 ax[MTW], ae[MTW];
    {
      const float* xb = P.y3p ? P.y3p + (size_t)b * L * HC : P.xin + (size_t)b * L * HC;   // statistics side: n_in, or the folded n3
      const float* eb = P.add_even ? P.add_even + (size_t)b * (L >> 1) * HC : nullptr;
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        const int pos = UP2 ? t0 + 2 * ((mt >> 1) * 16 + r) + (mt & 1) : t0 + (dg * MTW + mt) * 16 + r;
        ax[mt] = (pos < L) ? ld4o(xb, (unsigned)pos * HC + ch0) : (f32x4){0, 0, 0, 0};
        ae[mt] = (eb && !(pos & 1) && (pos >> 1) < (L >> 1)) ? ld4o(eb, (unsigned)(pos >> 1) * HC + ch0) : (f32x4){0, 0, 0, 0};
      }
    }
    // ---- data gradient: K = (tap, gradient channel); window row of gy[t + 1 - j] is (t - t0) + 2 - j
    f32x4 acc[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) acc[mt] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int j = ks / QN, q = ks % QN;
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) {
        // stride 2, symmetric padding: even outputs tap 1 (row u), odd outputs taps 2 (row u) and 0 (row u + 1); causal padding (forward taps
        // at 2u + j - 2): even outputs taps 2 (row u) and 0 (row u + 1), odd outputs tap 1 (row u + 1)
        if (UP2 && (mt & 1) != ((j == 1) != (CZ != 0) ? 0 : 1)) continue;
        const int row = UP2 ? (mt >> 1) * 16 + r + ((j == 0 || (CZ && j == 1)) ? 1 : 0) : (dg * MTW + mt) * 16 + r + 2 - j;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(gH + row * RSg + q * 32 + 8 * g);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(gL + row * RSg + q * 32 + 8 * g);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], bh, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], bl, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks], bh, acc[mt], 0, 0, 0);
      }
    }
    if constexpr (RD) {   // + Wd^T gpre[t/2] at even t (odd positions read the zero row)
#pragma unroll
      for (int q = 0; q < QN; ++q)
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
          const int m = (dg * MTW + mt) * 16 + r;
          const int prow = (m & 1) ? TM / 2 : (m >> 1);
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(pH + prow * RSg + q * 32 + 8 * g);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(pL + prow * RSg + q * 32 + 8 * g);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dh[q], bh, acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dh[q], bl, acc[mt], 0, 0, 0);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dl[q], bh, acc[mt], 0, 0, 0);
        }
    }
    // ---- epilogue: * GELU'(n_in), statistics partials, store
#ifdef W2S_WIDE_STAMP
    asm volatile("" :: "v"(acc[0]), "v"(acc[MTW - 1]));
    c1 = __builtin_amdgcn_s_memtime();
#endif
    f32x4 sA = {0, 0, 0, 0}, sB = {0, 0, 0, 0};
    f32x4 am = {0, 0, 0, 0}, ar = {1, 1, 1, 1};
    if (P.y3p) {
      const float* st = P.st3p + ((size_t)b * HC + ch0) * 2;
      const f32x4 s01 = ld4(st), s23 = ld4(st + 4);
      am = (f32x4){s01.x, s01.z, s23.x, s23.z}; ar = (f32x4){s01.y, s01.w, s23.y, s23.w};
    } else if (HST) {
      const float* st = stL + P.B * OC * 4 + (b * HC + ch0) * 2;
      const f32x4 s01 = ld4(st), s23 = ld4(st + 4);
      am = (f32x4){s01.x, s01.z, s23.x, s23.z}; ar = (f32x4){s01.y, s01.w, s23.y, s23.w};
    }
    float* ob = P.gout + (size_t)b * L * HC;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const int m = UP2 ? 2 * ((mt >> 1) * 16 + r) + (mt & 1) : (dg * MTW + mt) * 16 + r, pos = t0 + m;
      if (pos >= L) continue;
      const f32x4 v = (acc[mt] + ae[mt]) * ld4(gpL + m * RSp + ch0);
      const f32x4 n = (ax[mt] - am) * ar;
      const f32x4 sv = P.y3p ? v * gelu_grad4(n) : v;   // folded: gn = gout * GELU'(n3), sums of gn and gn * n3
      sA += sv;
      sB += sv * n;
      st4o(ob, (unsigned)pos * HC + ch0, v);
    }
    if (P.part) {   // [B][ntiles][PG][2][HC]: one row per (tile, position group), written once
      f32x4 x1, x2;
      x1 = sA; x2 = sB;
      row16_sum8(x1, x2);
      if (r == 0) {
        float* d = P.part + ((((size_t)b * P.ntiles + tile) * PG + dg) * 2) * HC + ch0;
        st4(d, x1);
        st4(d + HC, x2);
      }
    }
    // ---- weight gradient: k-step = 32 positions; gradient-side position p <-> gy window row p + 1, h[t + j - 1] <-> window row p + j
#ifdef W2S_WIDE_STAMP
    c2 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int s = 0; s < TG / 32; ++s) {
      const int p0 = 32 * s + 8 * g + q4;   // this lane's address row (gradient-side position) of the first 4-position block
      const int gr0 = UP2 ? p0 : p0 + 2 - PL;   // its gy window row (row 0 = position t0 + pad - 2)
      bf16x8 ah[IB], al[IB];
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        const int col = (wi * IB + i) * 16 + 4 * p4;
        ah[i] = wb_tr8(gH + gr0 * RSg + col, gH + (gr0 + 4) * RSg + col);
        al[i] = wb_tr8(gL + gr0 * RSg + col, gL + (gr0 + 4) * RSg + col);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int c = 0; c < CB; ++c) {
          const int col = (wc * CB + c) * 16 + 4 * p4;
          const int h0 = UP2 ? 2 * p0 + j : p0 + j, h1 = UP2 ? 2 * (p0 + 4) + j : p0 + 4 + j;   // window row of h[(t or 2u) + j - 1]
          const bf16x8 bh = wb_tr8(hH + h0 * RSh + col, hH + h1 * RSh + col);
          const bf16x8 bl = wb_tr8(hL + h0 * RSh + col, hL + h1 * RSh + col);
#pragma unroll
          for (int i = 0; i < IB; ++i) {
            accw[i][j][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, accw[i][j][c], 0, 0, 0);
            accw[i][j][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, accw[i][j][c], 0, 0, 0);
            accw[i][j][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, accw[i][j][c], 0, 0, 0);
          }
        }
      }
    }
    if constexpr (RD) {   // dWd[o][c] += sum_u gpre[u][o] h[2u][c]: window row of position t0 + 2u is 2u + pad
#pragma unroll
      for (int s = 0; s < TM / 64; ++s) {
        const int p0 = 32 * s + 8 * g + q4;
        bf16x8 ah[IB], al[IB];
#pragma unroll
        for (int i = 0; i < IB; ++i) {
          const int col = (wi * IB + i) * 16 + 4 * p4;
          ah[i] = wb_tr8(pH + p0 * RSg + col, pH + (p0 + 4) * RSg + col);
          al[i] = wb_tr8(pL + p0 * RSg + col, pL + (p0 + 4) * RSg + col);
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
          const int col = (wc * CB + c) * 16 + 4 * p4;
          const bf16x8 bh = wb_tr8(hH + (2 * p0 + PL) * RSh + col, hH + (2 * (p0 + 4) + PL) * RSh + col);
          const bf16x8 bl = wb_tr8(hL + (2 * p0 + PL) * RSh + col, hL + (2 * (p0 + 4) + PL) * RSh + col);
#pragma unroll
          for (int i = 0; i < IB; ++i) {
            accd[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, accd[i][c], 0, 0, 0);
            accd[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, accd[i][c], 0, 0, 0);
            accd[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, accd[i][c], 0, 0, 0);
          }
        }
      }
    }
#ifdef W2S_WIDE_STAMP
    asm volatile("" :: "v"(accw[0][0][0]), "v"(accw[IB - 1][2][CB - 1]));
    c3 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();   // the producers have staged the next windows; these may be overwritten
#ifdef W2S_WIDE_STAMP
    c4 = __builtin_amdgcn_s_memtime();
    tk += c1 - c0; te += c2 - c1; tg += c3 - c2; tw += c4 - c3;
#endif
  }
#ifdef W2S_WIDE_STAMP
  if (blockIdx.x == 0 && tid == 0 && P.part) { P.part[0] = (float)tk; P.part[1] = (float)te; P.part[2] = (float)tw; P.part[3] = (float)tg; }
#endif
  if (RD) {
    float* outd = P.slab_d + (size_t)blockIdx.x * (CO * CI) * 256 + lane * 4;
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int c = 0; c < CB; ++c) st4(outd + (size_t)((wi * IB + i) * CI + wc * CB + c) * 256, accd[i][c]);
  }
  float* out = P.slab + (size_t)blockIdx.x * (CO * 3 * CI) * 256 + lane * 4;
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int c = 0; c < CB; ++c) st4(out + (size_t)(((wi * IB + i) * 3 + j) * CI + wc * CB + c) * 256, accw[i][j][c]);
}

// (two entry points over one body, as conv_wide.hip: the second without packed-fp32 instruction selection)
template <int CO, int CI, int HST, int MT, int NWC, int IB, int CB, int PD, int UP2, int RD, int CZ>
__global__ __launch_bounds__(64 * (NWC + 4)) void bwd_wide_kernel(BwdWideP P) { bwd_wide_body<CO, CI, HST, MT, NWC, IB, CB, PD, UP2, RD, CZ>(P); }
template <int CO, int CI, int HST, int MT, int NWC, int IB, int CB, int PD, int UP2, int RD, int CZ>
__global__ __launch_bounds__(64 * (NWC + 4)) __attribute__((target("no-packed-fp32-ops"))) void bwd_wide_np_kernel(BwdWideP P) {
  bwd_wide_body<CO, CI, HST, MT, NWC, IB, CB, PD, UP2, RD, CZ>(P);
}

template <int CO, int CI, int HST, int MT, int NWC, int IB, int CB, int PD, int UP2 = 0, int RD = 0, int CZ = 0>
static int launch_bww(const BwdWideP& P0, int nslab, hipStream_t s, int dry) {
  constexpr int OC = CO * 16, HC = CI * 16, TM = 16 * MT, NR = TM + 2, NRG = UP2 ? TM / 2 + 1 : TM + 2, NRP = RD ? TM / 2 + 1 : 0;
  BwdWideP P = P0;
  P.ntiles = (P.L + TM - 1) / TM;
  const size_t lds = (size_t)2 * ((2 * (NRG + NRP) * (OC + 16) + 2 * NR * (HC + 8)) * 2 + TM * (HC + 4) * 4) + (size_t)P.B * OC * 4 * 4 + (HST ? (size_t)P.B * HC * 2 * 4 : 0);
  if (lds > 160 * 1024) return 1;   // (batch too large for the LDS statistics tables: the caller runs the separate kernels)
  if (dry) return 0;
  if (nslab <= 0 || (long)nslab > (long)P.B * P.ntiles) return W2S_EINVAL;   // every workgroup writes a slab: it needs a tile
#ifndef W2S_BWW_NP
#define W2S_BWW_NP 0   // tuning: 1 = the entry point without packed-fp32 selection
#endif
  void (*kern)(BwdWideP);
  if constexpr (W2S_BWW_NP) kern = bwd_wide_np_kernel<CO, CI, HST, MT, NWC, IB, CB, PD, UP2, RD, CZ>;
  else kern = bwd_wide_kernel<CO, CI, HST, MT, NWC, IB, CB, PD, UP2, RD, CZ>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return W2S_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(nslab), dim3(64 * (NWC + 4)), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// positions (input side) per tile and statistics-partial rows per tile of the (cg, ch, stride) instance; 0: no instance
extern "C" int w2s_bwd_wide_tile(int cg, int ch, int stride) {
  return (cg == 64 && ((ch == 64 && (stride == 1 || stride == 2)) || (ch == 32 && stride == 1))) ? 64 : 0;
}
extern "C" int w2s_bwd_wide_groups(int cg, int ch, int stride) { return !w2s_bwd_wide_tile(cg, ch, stride) ? 0 : (ch == 32) ? 2 : 1; }

// One pass for the backward of a k=3 encoder conv (pad: 1 = symmetric, 2 = causal left padding) with cg = 64 gradient-side channels: stride 1 with ch = 64 or 32
// input-side channels, or stride 2 (the block's conv3: g = dL/d(block pre-activation), [B][L/2][cg]) with ch = 64.  L: input-side length.
// w_hi / w_lo: the data-gradient operand planes (w2s_repack_batch bwd_hi / bwd_lo of the conv's weight).  part: [B][ntiles][groups][2][ch]
// partial sums of gout and gout * n_in (rows = ntiles * groups for w2s_stats_finalize), or NULL.  slab: nslab slabs of cg * 3 * ch floats
// (nslab = grid size <= B * ntiles) for w2s_wgrad_reduce(..., cg, ch, 3, 1, ...).  dry != 0: only answer whether an instance takes this
// launch (0) or not (1) -- st_in then only says WHETHER the input side carries statistics (any non-NULL value).  y3p / st3p (stride 1, with
// part): fold the previous block's conv3-backward statistics pre-pass in (see BwdWideP).  gpre / wd_hi / wd_lo / slab_d (stride 1, st_in and
// add_even NULL, L even): the residual fold (see BwdWideP); slab_d: nslab slabs of cg * ch floats for w2s_wgrad_reduce(..., cg, ch, 1, 1, ...).
// In dry mode gpre only says WHETHER the fold is asked for.
extern "C" int w2s_bwd_wide(const float* g, const float* y, const float* st_k, const float* bst_k, const float* xin, const float* st_in,
                            const float* add_even, const void* w_hi, const void* w_lo, float* gout, float* part, float* slab, int nslab, int B,
                            int L, int cg, int ch, int stride, const float* y3p, const float* st3p, const float* gpre, const void* wd_hi,
                            const void* wd_lo, float* slab_d, int pad, int dry, void* stream) {
  if (!w2s_bwd_wide_tile(cg, ch, stride)) return 1;
  if (pad != 1 && pad != 2) return dry ? 1 : W2S_EINVAL;
  const bool rd = gpre != nullptr;
  if (rd && (stride != 1 || st_in || add_even || (L & 1) || (!dry && (!wd_hi || !wd_lo || !slab_d)))) return dry ? 1 : W2S_EINVAL;
  if (stride == 2 && (!st_in || add_even || (L & 1) || y3p)) return dry ? 1 : W2S_EINVAL;
  if (y3p && (!st3p || !part)) return W2S_EINVAL;
  if (!dry && (!g || !y || !st_k || !bst_k || !xin || !w_hi || !w_lo || !gout || !slab)) return W2S_EINVAL;
  if ((size_t)L * 64 * 4 >= ((size_t)1 << 32)) return W2S_EINVAL;   // 32-bit lane offsets inside one sample
  static const char* off = getenv("W2S_NO_BWD_WIDE");   // tuning only
  if (off) return 1;
  BwdWideP P{g, y, st_k, bst_k, xin, st_in, add_even, static_cast<const __bf16*>(w_hi), static_cast<const __bf16*>(w_lo), gout, part, slab, y3p, st3p, gpre,
             static_cast<const __bf16*>(wd_hi), static_cast<const __bf16*>(wd_lo), slab_d, B, L, L / stride, 0, pad};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (stride == 2) return pad == 2 ? launch_bww<4, 4, 1, 4, 4, 2, 2, 2, 1, 0, 1>(P, nslab, s, dry) : launch_bww<4, 4, 1, 4, 4, 2, 2, 2, 1>(P, nslab, s, dry);
  if (rd) return ch == 64 ? launch_bww<4, 4, 0, 4, 4, 2, 2, 2, 0, 1>(P, nslab, s, dry) : launch_bww<4, 2, 0, 4, 4, 2, 1, 2, 0, 1>(P, nslab, s, dry);
  if (ch == 64) return st_in ? launch_bww<4, 4, 1, 4, 4, 2, 2, 2>(P, nslab, s, dry) : launch_bww<4, 4, 0, 4, 4, 2, 2, 2>(P, nslab, s, dry);
  return st_in ? launch_bww<4, 2, 1, 4, 4, 2, 1, 2>(P, nslab, s, dry) : launch_bww<4, 2, 0, 4, 4, 2, 1, 2>(P, nslab, s, dry);
}
