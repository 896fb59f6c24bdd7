// Weight gradient of the k=3 encoder convolutions with >= 64 channels on a side, as a PERSISTENT, ROLE-SPLIT split-precision kernel.
//
//   dW[o][j][c] = sum_{b,t} GY[b,t,o] * H[b, t*STRIDE + j - pad, c]        (w2s_wgrad's contract, taps = 3, dil = 1, pad = 1 or 2 = causal)
//   GY = instance-norm backward of the incoming gradient (PG = W2S_PRO_INBWD; W2S_PRO_INBWD_GP for the stride-2 conv3, whose incoming
//        gradient is with respect to the block's pre-activation output), H = GELU(IN(x)) or GELU(x) (PH)
//
// Why not wgrad_bf_kernel here (profiles/r02_*: 2.1-3.1 TB/s, traffic 1.36-1.39 x algorithmic): its workgroups run load -> transform ->
// LDS -> barrier -> MFMA strictly in sequence (no prefetch: every tile pays an HBM round trip), and it was launched with 512
// workgroups = 512 slabs of cout*3*cin floats.  Here, as in conv_wide_kernel:
//   * 4 PRODUCER waves stream the raw rows of the next PD tiles into registers (unconditional clamped loads => counted vmcnt waits),
//     apply the on-load transforms with the per-sample statistics read from an LDS table, split into bf16 (hi, lo) and write the
//     two row-major [position][channel] windows of tile i + 1 into the other LDS buffer;
//   * NWC CONSUMER waves own IB x CB (output-channel tile, input-channel tile) pairs with all three taps: accumulators stay in
//     registers for the whole launch, operands come from LDS with the transposing read (the contraction index is the position =
//     a column of the row-major windows), 3 bf16 MFMAs per product;
//   * one barrier per tile; one workgroup per CU => 256 slabs (written once at the end, summed by w2s_wgrad_reduce in fixed order).
// Slab layout = wgrad_bf_kernel's with gridDim.y == 1: [workgroup][tile (i*3 + j)*CI + c][lane][4].
#include <type_traits>
#include "conv_cl.inl"

struct WgWideP {
  const float* g; const float* g2; const float* gst; const float* gbst;
  const float* x; const float* xst;
  float* slab;
  int B, Lg, Lh, ntiles, pad;   // pad 1: symmetric; 2: causal
  int dbg;   // tuning only (W2S_WGW_DBG): 1 = no on-load arithmetic, 2 = no MFMA loop, 4 = no LDS staging
};

typedef __bf16 gbf16x4v __attribute__((__vector_size__(4 * sizeof(__bf16))));
#define gsplit_store4 split_store4   /* w2s_common.h: the explicit bit form */
// 8 consecutive positions (rows) of 16 channels for the 16x16x32 operand: two transposing reads of 4 rows each
__device__ __forceinline__ bf16x8 gtr_read8(const __bf16* p0, const __bf16* p1) {
  typedef __attribute__((address_space(3))) gbf16x4v* lds_p;
  gbf16x4v a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p0));
  gbf16x4v b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p1));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// CO / CI: 16-channel tiles on the gradient / input side; MT: k-steps (32 positions) per tile; NWC consumer waves in a
// (CO/IB) x (CI/CB) grid; PD: producer prefetch depth (register sets)
template <int CO, int CI, int STRIDE, int PG, int PH, int MT, int NWC, int IB, int CB, int PD>
__global__ __launch_bounds__(64 * (NWC + 4)) void wgrad_wide_kernel(WgWideP P) {
  extern __shared__ f32x4 smem4[];
  static_assert((CO / IB) * (CI / CB) == NWC, "consumer wave grid");
  constexpr int OC = CO * 16, HC = CI * 16, TM = 32 * MT, NPT = 256;
  constexpr int NRh = (TM - 1) * STRIDE + 3;            // input-side window rows; row 0 = position t0*STRIDE - pad
  constexpr int RSg = OC + 8, RSh = HC + 8;              // bf16 elements per LDS row
  constexpr int BUF = 2 * TM * RSg + 2 * NRh * RSh;      // one buffer: gy hi, gy lo, h hi, h lo
  constexpr bool HST = (PH == W2S_PRO_IN_GELU);
  __bf16* lds = reinterpret_cast<__bf16*>(smem4);
  float* gstL = reinterpret_cast<float*>(lds + 2 * BUF);  // [B][OC][2] (mean, rstd), then [B][OC][2] backward sums, then [B][HC][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total = P.B * P.ntiles;
  const int first = blockIdx.x, step = gridDim.x;
  for (int i = tid; i < P.B * OC * 2; i += 64 * (NWC + 4)) {
    gstL[i] = P.gst[i];
    gstL[P.B * OC * 2 + i] = P.gbst[i];
  }
  if (HST)
    for (int i = tid; i < P.B * HC * 2; i += 64 * (NWC + 4)) gstL[P.B * OC * 4 + i] = P.xst[i];
  __syncthreads();

  const int nt_wg = (total - first + step - 1) / step;            // >= 1
  const int NI = ((nt_wg + 1 + PD - 1) / PD) * PD;                // barrier rounds, padded to whole prefetch cycles

  if (wave >= NWC) {
    // ================================================= producer waves =================================================
#ifndef W2S_WGW_PPRIO
#define W2S_WGW_PPRIO 1   // issue priority of the producer waves (128 -> 128: -4 %, the other shapes neutral)
#endif
    if (W2S_WGW_PPRIO) __builtin_amdgcn_s_setprio(W2S_WGW_PPRIO);
    const int pt = tid - 64 * NWC;
    constexpr int c4g = OC / 4, rsg = NPT / c4g, NG = (TM + rsg - 1) / rsg;
    constexpr int c4h = HC / 4, rsh = NPT / c4h, NH = (NRh + rsh - 1) / rsh;
    const int gch = (pt % c4g) * 4, grow0 = pt / c4g;
    const int hch = (pt % c4h) * 4, hrow0 = pt / c4h;
    f32x4 rg[PD][NG] = {}, ry[PD][NG] = {}, rh[PD][NH] = {};
    auto load_g = [&](auto SET, int i, int k) {
      constexpr int S = decltype(SET)::value;
      const int tl = first + min(i, nt_wg - 1) * step;
      const int b = tl / P.ntiles, t0 = (tl % P.ntiles) * TM;
      const int row = min(grow0 + k * rsg, TM - 1), gr = min(t0 + row, P.Lg - 1);
      const unsigned off = (unsigned)gr * OC + gch;
      rg[S][k] = ld4o(P.g + (size_t)b * P.Lg * OC, off);
      ry[S][k] = ld4o(P.g2 + (size_t)b * P.Lg * OC, off);
    };
    auto load_h = [&](auto SET, int i, int k) {
      constexpr int S = decltype(SET)::value;
      const int tl = first + min(i, nt_wg - 1) * step;
      const int b = tl / P.ntiles, t0 = (tl % P.ntiles) * TM;
      const int row = min(hrow0 + k * rsh, NRh - 1), gr = min(max(t0 * STRIDE - P.pad + row, 0), P.Lh - 1);
      rh[S][k] = ld4o(P.x + (size_t)b * P.Lh * HC, (unsigned)gr * HC + hch);
    };
    auto stage = [&](auto SET, int i) {
      constexpr int S = decltype(SET)::value;
      const bool live = i < nt_wg;   // uniform; padding rounds only keep the load queue regular
      const int tl = first + min(i, nt_wg - 1) * step;
      const int b = tl / P.ntiles, t0 = (tl % P.ntiles) * TM;
      __bf16* gH = lds + (i & 1) * BUF;
      __bf16* gL = gH + TM * RSg;
      __bf16* hH = gL + TM * RSg;
      __bf16* hL = hH + NRh * RSh;
      ProCoef kg;
      {
        const float* st = gstL + (b * OC + gch) * 2;
        const f32x4 s01 = ld4(st), s23 = ld4(st + 4);
        const f32x4 pm = {s01.x, s01.z, s23.x, s23.z}, pr = {s01.y, s01.w, s23.y, s23.w};
        const float* bs = gstL + ((P.B + b) * OC + gch) * 2;
        const f32x4 b01 = ld4(bs), b23 = ld4(bs + 4);
        kg = pro_coef(PG, pm, pr, (f32x4){b01.x, b01.z, b23.x, b23.z}, (f32x4){b01.y, b01.w, b23.y, b23.w});
      }
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        const int row = grow0 + k * rsg;
        const f32x4 v1 = rg[S][k], v2 = ry[S][k];
        load_g(SET, i + PD, k);
        if (live && row < TM && !(P.dbg & 4)) {
          const f32x4 tv = (P.dbg & 1) ? v1 + v2 : pro_apply_k(PG, v1, v2, kg);
          gsplit_store4(gH, gL, row * RSg + gch, (t0 + row < P.Lg) ? tv : (f32x4){0, 0, 0, 0});
        }
      }
      f32x4 hm = {0, 0, 0, 0}, hr = {1, 1, 1, 1};
      if (HST) {
        const float* st = gstL + P.B * OC * 4 + (b * HC + hch) * 2;
        const f32x4 s01 = ld4(st), s23 = ld4(st + 4);
        hm = (f32x4){s01.x, s01.z, s23.x, s23.z}; hr = (f32x4){s01.y, s01.w, s23.y, s23.w};
      }
      const ProCoef kh = pro_coef(PH, hm, hr, hm, hm);
      const int rb = t0 * STRIDE - P.pad;
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        const int row = hrow0 + k * rsh, gr = rb + row;
        const f32x4 v = rh[S][k];
        load_h(SET, i + PD, k);
        if (live && row < NRh && !(P.dbg & 4)) {
          const f32x4 z = {0, 0, 0, 0};
          const f32x4 tv = (P.dbg & 1) ? v + hm : pro_apply_k(PH, v, z, kh);
          gsplit_store4(hH, hL, row * RSh + hch, (gr >= 0 && gr < P.Lh) ? tv : z);
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
    for (int it = 0; it < NI; it += PD) {
      stage(I0{}, it);
      __syncthreads();
      if constexpr (PD > 1) { stage(I1{}, it + 1); __syncthreads(); }
      if constexpr (PD > 2) { stage(I2{}, it + 2); __syncthreads(); }
    }
    return;
  }

  // =================================================== consumer waves ===================================================
  const int g = lane >> 4, li = lane & 15, q4 = li >> 2, p4 = li & 3;
  const int wi = wave / (CI / CB), wc = wave % (CI / CB);
  f32x4 acc[IB][3][CB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int c = 0; c < CB; ++c) acc[i][j][c] = (f32x4){0, 0, 0, 0};
  __syncthreads();                     // round 0 of the producers: the first windows are in buffer 0
  for (int it = 0; it < NI - 1; ++it) {
    if (it >= nt_wg) { __syncthreads(); continue; }   // padding rounds of the producers' prefetch cycle
    const __bf16* gH = lds + (it & 1) * BUF;
    const __bf16* gL = gH + TM * RSg;
    const __bf16* hH = gL + TM * RSg;
    const __bf16* hL = hH + NRh * RSh;
    if (!(P.dbg & 2))
#pragma unroll
    for (int s = 0; s < MT; ++s) {
      const int p0 = 32 * s + 8 * g + q4;   // this lane's address row (gradient-side position) of the first 4-position block
      bf16x8 ah[IB], al[IB];
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        const int col = (wi * IB + i) * 16 + 4 * p4;
        ah[i] = gtr_read8(gH + p0 * RSg + col, gH + (p0 + 4) * RSg + col);
        al[i] = gtr_read8(gL + p0 * RSg + col, gL + (p0 + 4) * RSg + col);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int r0 = p0 * STRIDE + j, r1 = (p0 + 4) * STRIDE + j;
#pragma unroll
        for (int c = 0; c < CB; ++c) {
          const int col = (wc * CB + c) * 16 + 4 * p4;
          const bf16x8 bh = gtr_read8(hH + r0 * RSh + col, hH + r1 * RSh + col);
          const bf16x8 bl = gtr_read8(hL + r0 * RSh + col, hL + r1 * RSh + col);
#pragma unroll
          for (int i = 0; i < IB; ++i) {
            acc[i][j][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, acc[i][j][c], 0, 0, 0);
            acc[i][j][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, acc[i][j][c], 0, 0, 0);
            acc[i][j][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, acc[i][j][c], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();   // the producers have staged the next windows; these may be overwritten
  }
  float* out = P.slab + (size_t)blockIdx.x * (CO * 3 * CI) * 256 + lane * 4;
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int c = 0; c < CB; ++c) st4(out + (size_t)(((wi * IB + i) * 3 + j) * CI + wc * CB + c) * 256, acc[i][j][c]);
}

template <int CO, int CI, int STRIDE, int PG, int PH, int MT, int NWC, int IB, int CB, int PD>
static int launch_wgw(const w2s_wgrad_args& a, hipStream_t s, int dry) {
  constexpr int OC = CO * 16, HC = CI * 16, TM = 32 * MT, NRh = (TM - 1) * STRIDE + 3;
  WgWideP P{a.g, a.g2, a.g_stats, a.g_bstats, a.x, a.x_stats, a.slab, a.B, a.L_out, a.L_in, (a.L_out + TM - 1) / TM, a.pad, 0};
  { static const char* d = getenv("W2S_WGW_DBG"); if (d) P.dbg = atoi(d); }
  size_t lds = (size_t)2 * (2 * TM * (OC + 8) + 2 * NRh * (HC + 8)) * 2 + (size_t)a.B * OC * 4 * 4 + (PH == W2S_PRO_IN_GELU ? (size_t)a.B * HC * 2 * 4 : 0);
  if (lds > 160 * 1024) return 1;   // (batch too large for the LDS statistics tables)
  if (dry) return 0;
  // every workgroup needs a tile: its slab must be written.  The caller sized nslab by w2s_wgrad_max_blocks / _slabs_per_block_of,
  // i.e. for THIS kernel (one slab per workgroup) -- the generic kernels' slab count may differ, so there is no falling through here.
  if (a.nslab <= 0 || (long)a.nslab > (long)P.B * P.ntiles) return W2S_EINVAL;
  auto kern = wgrad_wide_kernel<CO, CI, STRIDE, PG, PH, MT, NWC, IB, CB, PD>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return W2S_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(a.nslab), dim3(64 * (NWC + 4)), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

static bool wgw_shape(const w2s_wgrad_args& a) {
  if (!a.split_precision || a.taps != 3 || a.dil != 1 || (a.pad != 1 && a.pad != 2) || (a.stride != 1 && a.stride != 2)) return false;
  if (a.ldg != a.cout || a.ldx != a.cin || !a.g2 || !a.g_stats || !a.g_bstats) return false;
  if (a.cin < 32 || a.cout < 64) return false;
  static const char* off = getenv("W2S_NO_WGRAD_WIDE");   // tuning only
  return !off;
}
// 1 = not one of this kernel's shapes (the caller falls through to the generic kernels); dry: only answer (0 = would take it)
int w2s_wgrad_wide_try(const w2s_wgrad_args& a, hipStream_t s, int dry) {
  if (!wgw_shape(a)) return 1;
#define W2S_WGW(CO_, CI_, ST_, PG_, PH_, MT_, NWC_, IB_, CB_, PD_) \
  if (a.cout == 16 * CO_ && a.cin == 16 * CI_ && a.stride == ST_ && a.pro_g == PG_ && a.pro_h == PH_) \
    return launch_wgw<CO_, CI_, ST_, PG_, PH_, MT_, NWC_, IB_, CB_, PD_>(a, s, dry);
  W2S_WGW(4, 4, 1, W2S_PRO_INBWD, W2S_PRO_IN_GELU, 2, 4, 2, 2, 3)
  W2S_WGW(4, 4, 1, W2S_PRO_INBWD, W2S_PRO_GELU, 2, 4, 2, 2, 3)
  W2S_WGW(4, 4, 2, W2S_PRO_INBWD_GP, W2S_PRO_IN_GELU, 2, 4, 2, 2, 2)
  W2S_WGW(8, 8, 1, W2S_PRO_INBWD, W2S_PRO_IN_GELU, 1, 8, 2, 4, 2)
  W2S_WGW(8, 8, 1, W2S_PRO_INBWD, W2S_PRO_GELU, 1, 8, 2, 4, 2)
  W2S_WGW(8, 8, 2, W2S_PRO_INBWD_GP, W2S_PRO_IN_GELU, 1, 8, 2, 4, 1)
  W2S_WGW(8, 4, 1, W2S_PRO_INBWD, W2S_PRO_GELU, 1, 8, 2, 2, 2)
  W2S_WGW(4, 2, 1, W2S_PRO_INBWD, W2S_PRO_GELU, 2, 4, 2, 1, 3)
#undef W2S_WGW
  return 1;
}
