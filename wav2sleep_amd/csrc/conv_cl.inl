// Channels-last 1-D convolution as an implicit GEMM on the gfx950 fp32 matrix cores.
//
// One workgroup (256 threads = 4 waves) produces a tile of TM = 64*MT consecutive output positions of ONE
// sample for NT*16 output channels.  The input window is staged ONCE into LDS through registers with the
// producer's normalisation + activation applied on the way (instance-norm statistics are a global
// reduction over ~1M positions, so conv -> norm -> GELU cannot be fused forwards; instead the conv writes
// the PRE-norm tensor + partial statistics and the consumer normalises on load).  Each wave then owns
// 16*MT positions x all NT*16 channels and walks K = taps*cin in chunks of 16:
//   B operand (activations): ONE ds_read_b128 per lane feeds 4 MFMAs (the 4 k-slots of an MFMA are the
//       lane groups l>>4, so K is permuted as k = 16q + 4*(l>>4) + e, e = MFMA index);
//   A operand (weights [cout][taps][cin]): one global_load_dwordx4 per lane per n-tile, L1/L2 resident;
//   D fragment: lane (r = l&15, g = l>>4) holds 4 consecutive channels 4g..4g+3 of position r
//       => one coalesced 16-B store per lane, a wave writes whole rows.
// Reference ops replaced: see include/w2s.h (w2s_conv_args).
#pragma once
#include <cstdlib>
#include <type_traits>
#include "w2s_common.h"

__host__ __device__ constexpr int conv_bf_pad(int cin, int stride, int mode) { return (cin == 16 || (stride == 2 && mode == W2S_MODE_CONTIG)) ? 8 : 16; }

struct ConvP {
  w2s_conv_args a;
  int ntiles;
  int nr_lds;  // rows of the staged window (BF: offset of the lo plane)
};

// ---- on-load transform of one float4 (4 consecutive channels) --------------------------------------
__device__ __forceinline__ f32x4 pro_apply(int pro, f32x4 v, f32x4 v2, f32x4 mean, f32x4 rstd, f32x4 s1, f32x4 s2) {
  switch (pro) {
    case W2S_PRO_SANITIZE:
      v.x = sanitize_f(v.x); v.y = sanitize_f(v.y); v.z = sanitize_f(v.z); v.w = sanitize_f(v.w);
      return v;
    case W2S_PRO_GELU:
      return gelu4(v);
    case W2S_PRO_IN_GELU:
    case W2S_PRO_FIRST:  // v = the recomputed first-layer output
      return gelu4((v - mean) * rstd);
    case W2S_PRO_INBWD: {
      f32x4 n = (v2 - mean) * rstd;
      return rstd * (v - s1 - n * s2);
    }
    case W2S_PRO_INBWD_GP: {
      f32x4 n = (v2 - mean) * rstd;
      f32x4 gn = v * gelu_grad4(n);
      return rstd * (gn - s1 - n * s2);
    }
    default:
      return v;
  }
}

// NT: n-tiles (16 channels) per workgroup; MT: m-tiles (16 positions) per WAVE; WN: waves along channels (the 4 waves
// form a (4/WN) x WN grid, so a wave owns 16*MT positions x 16*NT/WN channels and each weight fragment fetched from
// L2 feeds MT MFMAs: WN = 2 quarters the L2 weight traffic of the 128-channel layers, which was their limiter).
// PRO / EPI >= 0: the prologue / epilogue mode is a compile-time constant (hot encoder paths: dead variants vanish and
// the kernel needs ~58 instead of ~84 VGPRs => more resident workgroups => more bytes in flight); -1: runtime switch.
// BF = 1: split-precision matrix cores ("bf16x3") for the MFMA-bound layers.  Every fp32 operand is split on the way into
// LDS as x = hi + lo (two bf16 planes, together the size of the fp32 window), weights arrive pre-split, and each
// 16x16x32 product runs as three bf16 MFMAs  hi*hi + hi*lo + lo*hi  with fp32 accumulation: 3 x 16 cycles per 32 channels
// instead of 8 x 32 cycles of fp32 MFMA (5.3x), relative error per product <= 2^-16.  This is the arithmetic class the
// reference itself trains with (torch.set_float32_matmul_precision('high') = TF32 / bf16_3x, scripts/train.py:117).
// 128-wide tiles are LDS-limited to two workgroups per CU anyway: cap the allocator at two waves per SIMD's worth of registers
// (the generic split-precision instance landed on 257 registers = one wave per SIMD, -35 % on the transformer GEMMs)
template <int NT, int MT, int TAPS, int STRIDE, int MODE, int WN, int PRO, int EPI, int BF>
#ifndef W2S_CL_OCC8
#define W2S_CL_OCC8 2   // tuning: waves per SIMD asked for the 128-wide instances
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT >= 8 ? W2S_CL_OCC8 : 1))) void conv_cl_kernel(ConvP P) {
  extern __shared__ f32x4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);
  const w2s_conv_args& a = P.a;
  constexpr int WM = 4 / WN, NTW = NT / WN, TM = 16 * MT * WM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave / WN, wave_n = wave % WN;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, tile = blockIdx.x, n0 = blockIdx.y * (NT * 16);
  const int t0 = tile * TM;
  const int cin = a.cin, RS = cin + 4, c4n = cin >> 2, rstep = 256 / c4n;
  // BF: bf16 elements per LDS row.  +32 B makes the 16-lane groups of the stride-1 B-operand ds_read_b128 hit 16 distinct 16-B slots
  // (+16 B rows were 2-way conflicted: SQ_LDS_BANK_CONFLICT = 48 % of SQ_LDS_IDX_ACTIVE); rows read 2 apart (stride 2) want +16 B
  const int RSE = cin + conv_bf_pad(cin, STRIDE, MODE);
  __bf16* hiL = reinterpret_cast<__bf16*>(smem);
  __bf16* loL = hiL + P.nr_lds * RSE;
  const int K = TAPS * cin;
  const int L_in = a.L_in, L_out = a.L_out;
  const int pro = (PRO >= 0) ? PRO : a.pro;

  f32x4 acc[MT][NTW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-thread channel-group parameters for the on-load transform
  const int myc4 = tid % c4n, row0 = tid / c4n;
  f32x4 pm = {0, 0, 0, 0}, pr = {1, 1, 1, 1}, ps1 = {0, 0, 0, 0}, ps2 = {0, 0, 0, 0};
  if (pro >= W2S_PRO_IN_GELU) {
    const float* st = a.pro_stats + ((size_t)b * cin + myc4 * 4) * 2;
    f32x4 s01 = ld4(st), s23 = ld4(st + 4);
    pm = (f32x4){s01.x, s01.z, s23.x, s23.z};
    pr = (f32x4){s01.y, s01.w, s23.y, s23.w};
    if (pro == W2S_PRO_INBWD || pro == W2S_PRO_INBWD_GP || pro >= W2S_PRO_AFFINE_BWD) {
      const float* bs = a.pro_bstats + ((size_t)b * cin + myc4 * 4) * 2;
      f32x4 b01 = ld4(bs), b23 = ld4(bs + 4);
      ps1 = (f32x4){b01.x, b01.z, b23.x, b23.z};
      ps2 = (f32x4){b01.y, b01.w, b23.y, b23.w};
    }
  }
  const ProCoef kc = pro_coef(pro, pm, pr, ps1, ps2);
  // wave-uniform per-sample bases + 32-bit lane offsets: scalar-base addressing (a sample's tensor is < 4 GB)
  const float* xb = ((PRO == W2S_PRO_FIRST)) ? a.x + (size_t)b * L_in : a.x + (size_t)b * L_in * a.ldx;
  const float* x2b = (pro == W2S_PRO_INBWD || pro == W2S_PRO_INBWD_GP || pro >= W2S_PRO_AFFINE_BWD) ? a.x2 + (size_t)b * L_in * a.ldx : nullptr;
  float w1r[4][3];  // W2S_PRO_FIRST: this thread's 4 output channels of block 0's conv1 (a.x2 = its weight [16][3])
  if ((PRO == W2S_PRO_FIRST)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) w1r[i][j] = a.x2[(myc4 * 4 + i) * 3 + j];
  }

  float* xsL = BF ? reinterpret_cast<float*>(loL + P.nr_lds * RSE) : smem + P.nr_lds * RS;  // W2S_PRO_FIRST: sanitised signal samples rb-1 .. rb+NR (zero outside the recording)
  auto stage = [&](int rb, int NR, int rowmul) {
    constexpr int U = (NT >= 8) ? 8 : 4;  // loads in flight per thread per batch (bigger windows: fewer round trips)
    if ((PRO == W2S_PRO_FIRST)) {
      for (int i = tid; i < NR + 2; i += 256) {
        const int gr = rb - 1 + i;
        const float xv = xb[min(max(gr, 0), L_in - 1)];
        xsL[i] = (gr >= 0 && gr < L_in && !isinf(xv)) ? xv : 0.f;
      }
      __syncthreads();
    }
    for (int row = row0; row < NR; row += rstep * U) {
      f32x4 v[U], v2[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rr = row + u * rstep, gr = rb + rr * rowmul;
        const bool ok = (rr < NR) && (gr >= 0) && (gr < L_in);
        if ((PRO == W2S_PRO_FIRST)) {  // conv1 (Cin = 1, k = 3, zero padding) recomputed from the raw signal window in LDS
          const int xi = (rr < NR) ? rr : 0;
          const float xm = xsL[xi], xc = xsL[xi + 1], xp = xsL[xi + 2];
          v[u].x = w1r[0][0] * xm + w1r[0][1] * xc + w1r[0][2] * xp;
          v[u].y = w1r[1][0] * xm + w1r[1][1] * xc + w1r[1][2] * xp;
          v[u].z = w1r[2][0] * xm + w1r[2][1] * xc + w1r[2][2] * xp;
          v[u].w = w1r[3][0] * xm + w1r[3][1] * xc + w1r[3][2] * xp;
          v2[u] = (f32x4){0, 0, 0, 0};
          continue;
        }
        const unsigned xo = (unsigned)gr * (unsigned)a.ldx + myc4 * 4;
        v[u] = ok ? ld4o(xb, xo) : (f32x4){0, 0, 0, 0};
        v2[u] = (ok && x2b) ? ld4o(x2b, xo) : (f32x4){0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rr = row + u * rstep, gr = rb + rr * rowmul;
        if (rr < NR) {
          const bool ok = (gr >= 0) && (gr < L_in);
          f32x4 t = ok ? pro_apply_k(pro, v[u], v2[u], kc) : (f32x4){0, 0, 0, 0};
          if constexpr (BF) {
            split_store4(hiL, loL, rr * RSE + myc4 * 4, t);
          } else {
            st4(smem + rr * RS + myc4 * 4, t);
          }
        }
      }
    }
  };

  const int wm0 = wave_m * (16 * MT);  // first tile-local output position of this wave
  const int wn0 = wave_n * (NTW * 16);  // first tile-local output channel of this wave

  // Weight prefetch (trunk GEMMs only: transformer linears and SequenceCNN convs, PRO_NONE): those launches have 1-2 workgroups per
  // CU, so nothing hides the L2 round trip of a K step's weight fragments (28 of them in a row for a k=7 conv: measured 41 us per launch
  // for 5 us of MFMA work).  The fragments of the NEXT K step are requested before the current step's MFMAs; the first request goes
  // out before the window is staged.  Not for the encoder layers: their occupancy hides the latency and the extra registers cost more.
#ifdef W2S_NO_WPF
  constexpr bool WPF = false;
#else
  constexpr bool WPF = BF && (PRO == W2S_PRO_NONE) && MODE != W2S_MODE_UP2;
#endif
  // two prefetch slots: the fragments of K steps n + 1 and n + 2 are in flight while step n runs (the taps are visited in order, so
  // the K-step index is simply linear: n = tap * (cin/32) + q).  One step ahead left the k=7 SequenceCNN conv at 30 us = 28 steps x
  // one exposed L2 round trip; cin = 32 (one K step per tap) keeps the single slot.
  bf16x8 pah[2][NTW], pal[2][NTW];
  const int NSTEP = TAPS * (cin >> 5);
  // (round 6) wave-uniform 64-bit bases of this wave's first fragment row + one 32-bit BYTE offset per lane and K step: the fetch of a
  // step is `base + (nt * NSTEP + kidx) KB + 16 lane` -- the first cut recomputed a 64-bit element index per load (~10 vector instructions
  // each, 1100 of the 1500 per wave of a SequenceCNN conv: profiles/r06 counters)
  const char* wfh = reinterpret_cast<const char*>(static_cast<const __bf16*>(a.w_hi) + (size_t)((n0 + wn0) / 16) * (K >> 5) * 512);
  const char* wfl = reinterpret_cast<const char*>(static_cast<const __bf16*>(a.w_lo) + (size_t)((n0 + wn0) / 16) * (K >> 5) * 512);
  auto load_frag = [&](auto SLOT, int kidx_) {
    constexpr int SL = decltype(SLOT)::value;
    const int kidx = min(kidx_, NSTEP - 1);   // no branch around the loads (a conditional load makes hipcc wait vmcnt(0) everywhere)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const unsigned wo = ((unsigned)(nt * (K >> 5) + kidx) << 10) + ((unsigned)lane << 4);
      pah[SL][nt] = *reinterpret_cast<const bf16x8*>(wfh + wo);
      pal[SL][nt] = *reinterpret_cast<const bf16x8*>(wfl + wo);
    }
  };
  using WS0 = std::integral_constant<int, 0>; using WS1 = std::integral_constant<int, 1>;
  if constexpr (WPF) {
    if (cin >= 32) { load_frag(WS0{}, 0); load_frag(WS1{}, 1); }
  }

  auto mma_tap = [&](int jw, int rowoff, int mtmask, int jw_next = -1) {
    // jw: weight tap index; rowoff: LDS row offset added to the per-position row; mtmask: which m-tiles take part
    if constexpr (WPF) {
      const int QN = cin >> 5;
      auto step = [&](auto SLOT, int q, int ahead) {
        constexpr int SL = decltype(SLOT)::value;
        bf16x8 bh[MT], bl[MT], ah[NTW], al[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) { ah[nt] = pah[SL][nt]; al[nt] = pal[SL][nt]; }
        load_frag(SLOT, jw * QN + q + ahead);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int row = (MODE == W2S_MODE_DILATED) ? wm0 + mt * 16 + r : (wm0 + mt * 16 + r) * STRIDE + rowoff;
          bh[mt] = *reinterpret_cast<const bf16x8*>(hiL + row * RSE + q * 32 + 8 * g);
          bl[mt] = *reinterpret_cast<const bf16x8*>(loL + row * RSE + q * 32 + 8 * g);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          if (mtmask & (1 << mt))
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[nt], bh[mt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[nt], bl[mt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[nt], bh[mt], acc[mt][nt], 0, 0, 0);
            }
      };
      if (QN & 1) {   // one K step per tap (cin = 32): slot 0 only, one step ahead
        for (int q = 0; q < QN; ++q) step(WS0{}, q, 1);
      } else {
        for (int q = 0; q < QN; q += 2) { step(WS0{}, q, 2); step(WS1{}, q + 1, 2); }
      }
      return;
    }
    if constexpr (BF) {
      // chunks of 32 input channels; lane (r, g) holds 8 consecutive channels 8g..8g+7 of its row (A: weights of output
      // channel r, B: activations of position r) -- the 16x16x32 bf16 operand layout
      for (int q = 0; q < (cin >> 5); ++q) {
        bf16x8 bh[MT], bl[MT], ah[NTW], al[NTW];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          int row;
          if (MODE == W2S_MODE_UP2) row = wave_m * (8 * MT) + (mt >> 1) * 16 + r + rowoff;
          else if (MODE == W2S_MODE_DILATED) row = wm0 + mt * 16 + r;
          else row = (wm0 + mt * 16 + r) * STRIDE + rowoff;
          bh[mt] = *reinterpret_cast<const bf16x8*>(hiL + row * RSE + q * 32 + 8 * g);
          bl[mt] = *reinterpret_cast<const bf16x8*>(loL + row * RSE + q * 32 + 8 * g);
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          // fragment-major planes [cout/16][K/32][lane][8] (w2s_frag_index): the wave's 64 x 16 B are ONE contiguous 1 KB run
          // (8 full cache lines) -- row-major weights made every fetch 16 separate 64-B segments, and the texture-address
          // path (~40 cycles per such instruction), not the matrix pipe, set the pace of the >= 64-channel layers.
          const size_t wo = ((size_t)((n0 + wn0) / 16 + nt) * (K >> 5) + (size_t)(jw * cin + q * 32) / 32) * 512 + lane * 8;
          ah[nt] = *reinterpret_cast<const bf16x8*>(static_cast<const __bf16*>(a.w_hi) + wo);
          al[nt] = *reinterpret_cast<const bf16x8*>(static_cast<const __bf16*>(a.w_lo) + wo);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          if (mtmask & (1 << mt))
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[nt], bh[mt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[nt], bl[mt], acc[mt][nt], 0, 0, 0);
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[nt], bh[mt], acc[mt][nt], 0, 0, 0);
            }
      }
      return;
    }
    for (int q = 0; q < (cin >> 4); ++q) {
      f32x4 bf[MT], af[NTW];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        int row;
        if (MODE == W2S_MODE_UP2) row = wave_m * (8 * MT) + (mt >> 1) * 16 + r + rowoff;
        else if (MODE == W2S_MODE_DILATED) row = wm0 + mt * 16 + r;
        else row = (wm0 + mt * 16 + r) * STRIDE + rowoff;
        bf[mt] = *reinterpret_cast<const f32x4*>(smem + row * RS + q * 16 + 4 * g);
      }
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
        af[nt] = ld4(a.w + (size_t)(n0 + wn0 + nt * 16 + r) * K + jw * cin + q * 16 + 4 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          if (mtmask & (1 << mt))
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = mfma16(af[nt][e], bf[mt][e], acc[mt][nt]);
    }
  };

  if (MODE == W2S_MODE_CONTIG) {
    // one staged window serves every tap (dilated taps included: rows j*dil apart -- the SequenceCNN convs up to
    // dilation 32 fit the 160 KB LDS and run with ONE global->LDS round trip instead of one per tap)
    const int dil = (TAPS > 1 && a.dil > 1) ? a.dil : 1;
    stage(t0 * STRIDE - a.pad, (TM - 1) * STRIDE + (TAPS - 1) * dil + 1, 1);
    __syncthreads();
    if (BF && cin == 16) {
      // 16 input channels: two taps share one K = 32 step (lane groups 0,1 carry tap 2*ks, groups 2,3 tap 2*ks+1; the weight
      // planes are zero beyond the last tap).  The fp32-MFMA form of these layers spends 31 % of the issue time in the
      // matrix pipe on top of 43 % VALU (profiles/: both compete for issue); this one needs a quarter of the MFMA cycles.
      constexpr int KSP = (TAPS + 1) / 2;
      const int half = g >> 1, col = 8 * (g & 1);
#pragma unroll
      for (int ks = 0; ks < KSP; ++ks) {
        const int j = (2 * ks + half < TAPS) ? 2 * ks + half : TAPS - 1;
        const int rowoff = (a.flip ? (TAPS - 1 - j) : j) * dil;
        bf16x8 bh[MT], bl[MT], ah[NTW], al[NTW];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int row = (wm0 + mt * 16 + r) * STRIDE + rowoff;
          bh[mt] = *reinterpret_cast<const bf16x8*>(hiL + row * RSE + col);
          bl[mt] = *reinterpret_cast<const bf16x8*>(loL + row * RSE + col);
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const size_t wo = ((size_t)((n0 + wn0) / 16 + nt) * KSP + ks) * 512 + lane * 8;
          ah[nt] = *reinterpret_cast<const bf16x8*>(static_cast<const __bf16*>(a.w_hi) + wo);
          al[nt] = *reinterpret_cast<const bf16x8*>(static_cast<const __bf16*>(a.w_lo) + wo);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[nt], bh[mt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[nt], bl[mt], acc[mt][nt], 0, 0, 0);
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[nt], bh[mt], acc[mt][nt], 0, 0, 0);
          }
      }
    } else {
#pragma unroll
      for (int j = 0; j < TAPS; ++j) mma_tap(j, (a.flip ? (TAPS - 1 - j) : j) * dil, (1 << MT) - 1, j + 1 < TAPS ? j + 1 : -1);
    }
  } else if (MODE == W2S_MODE_DILATED) {
    for (int j = 0; j < TAPS; ++j) {
      const int off = a.flip ? (TAPS - 1 - j) : j;
      __syncthreads();
      stage(t0 * STRIDE - a.pad + off * a.dil, TM, STRIDE);
      __syncthreads();
      mma_tap(j, 0, (1 << MT) - 1, j + 1 < TAPS ? j + 1 : -1);
    }
  } else {  // UP2: output position t' = 2u + phase; phase = mt & 1
    constexpr int EVEN = 0x55 & ((1 << MT) - 1), ODD = 0xAA & ((1 << MT) - 1);
    stage(t0 / 2, TM / 2 + 1, 1);
    __syncthreads();
    if (a.pad == 2) {       // causal padding (forward taps 2u'-2+k): the parities swap roles
      mma_tap(2, 0, EVEN);  // t' = 2u   : W_2^T g[u]
      mma_tap(0, 1, EVEN);  //            + W_0^T g[u+1]
      mma_tap(1, 1, ODD);   // t' = 2u+1 : W_1^T g[u+1]
    } else {
      mma_tap(1, 0, EVEN);  // t' = 2u   : W_1^T g[u]
      mma_tap(2, 0, ODD);   // t' = 2u+1 : W_2^T g[u]
      mma_tap(0, 1, ODD);   //            + W_0^T g[u+1]
    }
  }

  // ------------------------------------ epilogue ------------------------------------
  const int epi = (EPI >= 0) ? EPI : a.epi;
  const int cout = a.cout;
  f32x4 sA[NTW], sB[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) { sA[nt] = (f32x4){0, 0, 0, 0}; sB[nt] = (f32x4){0, 0, 0, 0}; }
  const float keep = a.rowkeep ? a.rowkeep[b] : 1.0f;

#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int pos;
    if (MODE == W2S_MODE_UP2) pos = t0 + 2 * (wave_m * (8 * MT) + (mt >> 1) * 16 + r) + (mt & 1);
    else pos = t0 + wm0 + mt * 16 + r;
    const bool valid = pos < L_out;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int ch = n0 + wn0 + nt * 16 + 4 * g;
      f32x4 v = acc[mt][nt];
      if (!valid) continue;
      const size_t ob = (size_t)b * L_out;   // uniform row base; per-lane parts stay 32-bit
      if (epi == W2S_EPI_BIAS) {
        if (a.bias) v += ld4(a.bias + ch);
        if (a.reserved & (W2S_FUSE_ADD_DROP | W2S_FUSE_GELU_BWD_DROP)) {   // uniform: the transformer's residual / activation-backward fusions
          const size_t i0 = (ob + pos) * (size_t)a.ldy + ch;
          f32x4 m = {1, 1, 1, 1};
          if (a.drop_p > 0.f) {
            m.x = w2s_dropscale(a.drop_seed, i0, a.drop_p); m.y = w2s_dropscale(a.drop_seed, i0 + 1, a.drop_p);
            m.z = w2s_dropscale(a.drop_seed, i0 + 2, a.drop_p); m.w = w2s_dropscale(a.drop_seed, i0 + 3, a.drop_p);
          }
          const f32x4 ax = ld4o(a.aux + ob * a.ld_aux, (unsigned)pos * (unsigned)a.ld_aux + ch);
          v = (a.reserved & W2S_FUSE_ADD_DROP) ? ax + v * m : v * m * gelu_grad4(ax);
        }
      } else if (epi == W2S_EPI_AUX_INGELU_ADD) {
        f32x4 ax = ld4o(a.aux + ob * a.ld_aux, (unsigned)pos * (unsigned)a.ld_aux + ch);
        const float* st = a.aux_stats + ((size_t)b * cout + ch) * 2;
        f32x4 s01 = ld4(st), s23 = ld4(st + 4);
        f32x4 mean = {s01.x, s01.z, s23.x, s23.z}, rstd = {s01.y, s01.w, s23.y, s23.w};
        v += gelu4((ax - mean) * rstd);
      } else if (epi == W2S_EPI_GP) {
        f32x4 n = ld4o(a.aux + ob * a.ld_aux, (unsigned)pos * (unsigned)a.ld_aux + ch);
        if (a.aux_stats) {
          const float* st = a.aux_stats + ((size_t)b * cout + ch) * 2;
          f32x4 s01 = ld4(st), s23 = ld4(st + 4);
          f32x4 mean = {s01.x, s01.z, s23.x, s23.z}, rstd = {s01.y, s01.w, s23.y, s23.w};
          n = (n - mean) * rstd;
        }
        if (a.add_even && !(pos & 1)) v += ld4o(a.add_even + (size_t)b * (L_out >> 1) * cout, (unsigned)(pos >> 1) * (unsigned)cout + ch);
        v = v * gelu_grad4(n);
        sA[nt] += v;
        sB[nt] += v * n;
      } else if (epi == W2S_EPI_STATS) {
        sA[nt] += v;
        sB[nt] += v * v;
      } else if (epi >= W2S_EPI_AFFINE_PART) {   // generic path: sums for the norm backward of the layer below (v is stored as it is)
        const f32x4 ax = ld4o(a.aux + ob * a.ld_aux, (unsigned)pos * (unsigned)a.ld_aux + ch);
        const float* st = a.aux_stats + ((size_t)b * cout + ch) * 2;
        const f32x4 s01 = ld4(st), s23 = ld4(st + 4);
        const ProCoef kz{(f32x4){s01.x, s01.z, s23.x, s23.z}, (f32x4){s01.y, s01.w, s23.y, s23.w}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        // ga = v act'(z): the backward prologue's formula with scale 1 on the gradient and no (c, d) terms, then undo its `* k.a`
        const f32x4 z = fma4(ax, kz.a, kz.b);
        f32x4 d;
        switch (epi - W2S_EPI_AFFINE_PART) {
          case 1: d = (f32x4){z.x > 0.f ? 1.f : 0.f, z.y > 0.f ? 1.f : 0.f, z.z > 0.f ? 1.f : 0.f, z.w > 0.f ? 1.f : 0.f}; break;
          case 2: d = (f32x4){z.x > 0.f ? 1.f : 0.01f, z.y > 0.f ? 1.f : 0.01f, z.z > 0.f ? 1.f : 0.01f, z.w > 0.f ? 1.f : 0.01f}; break;
          case 3: d = gelu_grad4(z); break;
          case 4: {
            const f32x4 sg = {1.0f / (1.0f + __expf(-z.x)), 1.0f / (1.0f + __expf(-z.y)), 1.0f / (1.0f + __expf(-z.z)), 1.0f / (1.0f + __expf(-z.w))};
            d = sg * (z * (splat4(1.f) - sg) + 1.0f);
            break;
          }
          default: d = splat4(1.f);
        }
        const f32x4 ga = v * d;
        sA[nt] += ga;
        sB[nt] += ga * ax;
      }
      v = v * keep;
      if (a.reserved & 1) v += ld4o(a.y + ob * a.ldy, (unsigned)pos * (unsigned)a.ldy + ch);   // accumulate: K split over several launches (generic path)
      st4o(a.y + ob * a.ldy, (unsigned)pos * (unsigned)a.ldy + ch, v);
      if (a.y2) {
        f32x4 h = gelu4(v);
        if ((a.reserved & W2S_FUSE_Y2_GELU_DROP) && a.drop_p > 0.f) {
          const size_t i0 = (ob + pos) * (size_t)a.ldy2 + ch;
          h.x *= w2s_dropscale(a.drop_seed, i0, a.drop_p); h.y *= w2s_dropscale(a.drop_seed, i0 + 1, a.drop_p);
          h.z *= w2s_dropscale(a.drop_seed, i0 + 2, a.drop_p); h.w *= w2s_dropscale(a.drop_seed, i0 + 3, a.drop_p);
        }
        st4o(a.y2 + ob * a.ldy2, (unsigned)pos * (unsigned)a.ldy2 + ch, h);
      }
    }
  }

  if ((epi == W2S_EPI_STATS || epi == W2S_EPI_GP || epi >= W2S_EPI_AFFINE_PART) && a.part) {
    // deterministic two-level reduction: 16 positions (shuffle) -> 4 waves (LDS) -> one partial per tile
    __syncthreads();  // LDS window no longer needed
    float* red = smem;  // [wave][ntw][g][2][4]
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      f32x4 x1, x2;
      x1 = sA[nt]; x2 = sB[nt];
      row16_sum8(x1, x2);
      if (r == 0) {
        float* d = red + ((wave * NTW + nt) * 4 + g) * 8;
        st4(d, x1);
        st4(d + 4, x2);
      }
    }
    __syncthreads();
    // NT*16 channels x 2 sums, one thread each
    if (tid < NT * 32) {
      const int k = tid / (NT * 16), c = tid % (NT * 16);
      const int wn = c / (NTW * 16), nt = (c >> 4) % NTW, gg = (c >> 2) & 3, e = c & 3;
      float s = 0.f;
#pragma unroll
      for (int wm = 0; wm < WM; ++wm) s += red[(((wm * WN + wn) * NTW + nt) * 4 + gg) * 8 + k * 4 + e];
      w2s_part_store(&a.part[(((size_t)b * P.ntiles + tile) * 2 + k) * cout + n0 + c], s);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// tile configuration: (NT n-tiles per workgroup, MT m-tiles per wave, WN waves along channels); TM = 16*MT*(4/WN).
// accumulators MT*NT/WN <= 16 float4 (64 VGPRs); staged window <= 72 KB so two workgroups share a CU.
// widest channel tile that divides cout (production: 16 / 32 / 64 / 128 / 384 / 512 -> as before; generic path: 48, 80, 96 ...)
static inline int pick_nt(int cout) {
  for (int nt = 8; nt > 1; nt >>= 1)
    if (cout >= 16 * nt && cout % (16 * nt) == 0) return nt;
  return 1;
}
static inline int window_rows(int tm, int taps, int stride, int mode, int dil = 1) {
  if (mode == W2S_MODE_CONTIG) return (tm - 1) * stride + (taps - 1) * dil + 1;
  if (mode == W2S_MODE_DILATED) return tm;
  return tm / 2 + 1;
}
struct TileCfg { int nt, mt, wn; };
static inline TileCfg pick_cfg(int cin, int cout, int taps, int stride, int mode, int B, int L_out, int dil = 1) {
  TileCfg c;
  c.nt = pick_nt(cout);
  c.wn = (c.nt >= 4) ? 2 : 1;
  c.mt = 4;
  if ((size_t)window_rows(16 * c.mt * (4 / c.wn), taps, stride, mode, dil) * (cin + 4) * 4 > 72 * 1024) c.mt = 2;
  // 128-wide tiles: 64 positions (four workgroups per CU) beat 128 now that a weight fragment is one 1 KB fetch, as a 1 x 4 wave grid:
  // every wave owns 32 output channels for all 64 positions, so each weight fragment is fetched ONCE per workgroup (the 2 x 2 grid
  // fetched it from both position halves: the K loop of these layers waits on L2 weight fetches, 12 TB/s aggregate)  -9 ... -14 %
  if (c.nt >= 8) { c.mt = 4; c.wn = 4; }
  // short problems (SequenceCNN: 16 x 960 rows): shrink the tile until the grid covers the 256 CUs about twice
  auto wgs = [&](const TileCfg& t) {
    const long tm = 16 * t.mt * (4 / t.wn);
    return (long)B * ((L_out + tm - 1) / tm) * (cout / (t.nt * 16));
  };
  if (B > 0 && L_out > 0) {
    // ... unless the window is so large that one workgroup owns a CU's LDS (SequenceCNN at dilation 16 / 32: 92 / 147 KB) and the
    // 128-wide grid fits the chip in ONE round: halving the tile width would make it two rounds of the same length (30 vs 18 us)
    const size_t win = (size_t)window_rows(16 * c.mt * (4 / c.wn), taps, stride, mode, dil) * (cin + 16) * 4;
    if (win > 80 * 1024 && wgs(c) <= 256) return c;
    if (wgs(c) < 512 && c.mt == 4 && c.wn != 4) c.mt = 2;
    if (wgs(c) < 512 && c.nt == 8) { c.nt = 4; if (c.wn == 4) { c.wn = 2; c.mt = 2; } }
  }
  return c;
}
static inline int cfg_tm(const TileCfg& c) { return 16 * c.mt * (4 / c.wn); }

template <int NT, int MT, int TAPS, int STRIDE, int MODE, int WN, int PRO, int EPI, int BF>
static int launch_conv(const w2s_conv_args& a, hipStream_t s) {
  constexpr int TM = 16 * MT * (4 / WN);
  ConvP P;
  P.a = a;
  P.ntiles = (a.L_out + TM - 1) / TM;
  const int NR = window_rows(TM, TAPS, STRIDE, MODE, a.dil > 0 ? a.dil : 1);
  P.nr_lds = NR;
  size_t lds = BF ? (size_t)2 * NR * (a.cin + conv_bf_pad(a.cin, STRIDE, MODE)) * 2 : (size_t)NR * (a.cin + 4) * sizeof(float);
  if (a.pro == W2S_PRO_FIRST) lds += (size_t)(NR + 2) * sizeof(float);
  size_t red = (size_t)4 * (NT / WN) * 4 * 8 * sizeof(float);
  if (lds < red) lds = red;
  dim3 grid(P.ntiles, a.cout / (NT * 16), a.B);
  auto kern = conv_cl_kernel<NT, MT, TAPS, STRIDE, MODE, WN, PRO, EPI, BF>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return W2S_ELAUNCH;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}


template <int TAPS, int STRIDE, int MODE, int PRO, int EPI>
static int dispatch_cfg(const w2s_conv_args& a, hipStream_t s) {
  const TileCfg c = pick_cfg(a.cin, a.cout, TAPS, STRIDE, MODE, a.B, a.L_out, a.dil > 0 ? a.dil : 1);
  if (a.cout % (c.nt * 16)) return W2S_EINVAL;
  // split-precision path: caller supplied bf16 weight planes (>= 32 input and output channels)
  const bool bf = a.w_hi && a.w_lo && ((a.cin >= 32 && c.nt >= 2) || (a.cin == 16 && MODE == W2S_MODE_CONTIG && (TAPS == 1 || TAPS == 3) && a.dil <= 1));
#define W2S_CFG(NT_, MT_, WN_) if (c.nt == NT_ && c.mt == MT_ && c.wn == WN_ && !bf) return launch_conv<NT_, MT_, TAPS, STRIDE, MODE, WN_, PRO, EPI, 0>(a, s);
#define W2S_CFGB(NT_, MT_, WN_) if (c.nt == NT_ && c.mt == MT_ && c.wn == WN_ && bf) return launch_conv<NT_, MT_, TAPS, STRIDE, MODE, WN_, PRO, EPI, 1>(a, s);
  W2S_CFG(1, 4, 1) W2S_CFG(2, 4, 1) W2S_CFG(2, 2, 1) W2S_CFG(4, 4, 2) W2S_CFG(4, 2, 2) W2S_CFG(8, 4, 2) W2S_CFG(8, 2, 2) W2S_CFG(1, 2, 1) W2S_CFG(8, 4, 4) W2S_CFG(4, 4, 4)
  W2S_CFGB(1, 4, 1) W2S_CFGB(1, 2, 1) W2S_CFGB(2, 4, 1) W2S_CFGB(2, 2, 1) W2S_CFGB(4, 4, 2) W2S_CFGB(4, 2, 2) W2S_CFGB(8, 4, 2) W2S_CFGB(8, 2, 2) W2S_CFGB(8, 4, 4) W2S_CFGB(4, 4, 4)
#undef W2S_CFG
#undef W2S_CFGB
  return W2S_EINVAL;
}

// hot (prologue, epilogue) pairs of the encoder get compile-time specialisations; everything else the runtime-switch kernel
template <int TAPS, int STRIDE, int MODE>
static int dispatch_tile(const w2s_conv_args& a, hipStream_t s) {
  const bool plain_io = !a.y2 && !a.rowkeep && !a.reserved;
  if constexpr (MODE == W2S_MODE_CONTIG && TAPS == 3 && STRIDE == 1) if (plain_io) {
    if (a.pro == W2S_PRO_GELU && a.epi == W2S_EPI_STATS) return dispatch_cfg<TAPS, STRIDE, MODE, W2S_PRO_GELU, W2S_EPI_STATS>(a, s);
    if (a.pro == W2S_PRO_IN_GELU && a.epi == W2S_EPI_STATS) return dispatch_cfg<TAPS, STRIDE, MODE, W2S_PRO_IN_GELU, W2S_EPI_STATS>(a, s);
    if (a.pro == W2S_PRO_INBWD && a.epi == W2S_EPI_GP) return dispatch_cfg<TAPS, STRIDE, MODE, W2S_PRO_INBWD, W2S_EPI_GP>(a, s);
    if (a.pro == W2S_PRO_FIRST && a.epi == W2S_EPI_STATS && a.cin == 16)
      return dispatch_cfg<TAPS, STRIDE, MODE, W2S_PRO_FIRST, W2S_EPI_STATS>(a, s);
  }
  if constexpr (MODE == W2S_MODE_CONTIG && TAPS == 3 && STRIDE == 2)
    if (plain_io && a.pro == W2S_PRO_IN_GELU && a.epi == W2S_EPI_STATS) return dispatch_cfg<TAPS, STRIDE, MODE, W2S_PRO_IN_GELU, W2S_EPI_STATS>(a, s);
  if constexpr (TAPS == 1 && STRIDE == 2)   // (contiguous or per-tap window)
    if (plain_io && a.pro == W2S_PRO_GELU && a.epi == W2S_EPI_AUX_INGELU_ADD)
      return dispatch_cfg<TAPS, STRIDE, MODE, W2S_PRO_GELU, W2S_EPI_AUX_INGELU_ADD>(a, s);
  // transformer / SequenceCNN GEMMs (no on-load transform): without the prologue variants the 128-wide instance keeps its
  // accumulators in AGPRs and two waves per SIMD
  if constexpr ((MODE == W2S_MODE_CONTIG && TAPS == 1 && STRIDE == 1) || (MODE == W2S_MODE_DILATED && TAPS == STRIDE))
    if (!a.rowkeep && !(a.reserved & 1) && (!a.y2 || (a.reserved & W2S_FUSE_Y2_GELU_DROP)) && a.pro == W2S_PRO_NONE && a.epi == W2S_EPI_BIAS)
      return dispatch_cfg<TAPS, STRIDE, MODE, W2S_PRO_NONE, W2S_EPI_BIAS>(a, s);   // (the residual / dropout / GELU fusions of the transformer included)
  if constexpr (MODE == W2S_MODE_CONTIG && TAPS == 1 && STRIDE == 1)
    if (plain_io && a.pro == W2S_PRO_NONE && a.epi == W2S_EPI_PLAIN) return dispatch_cfg<TAPS, STRIDE, MODE, W2S_PRO_NONE, W2S_EPI_PLAIN>(a, s);
  if constexpr (TAPS == 7 && STRIDE == 1)
    if (plain_io && a.pro == W2S_PRO_NONE && a.epi == W2S_EPI_PLAIN) return dispatch_cfg<TAPS, STRIDE, MODE, W2S_PRO_NONE, W2S_EPI_PLAIN>(a, s);
  if constexpr (MODE == W2S_MODE_UP2)
    if (plain_io && a.pro == W2S_PRO_INBWD_GP && a.epi == W2S_EPI_GP) return dispatch_cfg<TAPS, STRIDE, MODE, W2S_PRO_INBWD_GP, W2S_EPI_GP>(a, s);
  if (a.pro == W2S_PRO_FIRST) return W2S_EINVAL;  // only the specialised (FIRST, STATS) k=3/stride-1 instance implements it
  return dispatch_cfg<TAPS, STRIDE, MODE, -1, -1>(a, s);
}
