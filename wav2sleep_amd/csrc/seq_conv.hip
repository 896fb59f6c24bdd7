// SequenceCNN's dilated convolutions (DilatedConvBlock.forward, models/blocks.py:115-126: Conv1d(128, 128, k = 7, dilation d, no bias) ->
// ConvLayerNorm (channel LayerNorm, models/utils.py:9-23) -> GELU) with the LayerNorm IN THE CONV'S EPILOGUE -- forward and backward (round 6).
//
//   MODE 0:  y[b,t,:] = sum_j W_j x[b, t + roff(j) d - pad, :]                                            (the plain conv / data gradient)
//   MODE 1:  ... and  hn = GELU(gamma * (y - mean_c y) * rstd_c + beta),  rs = (mean_c, rstd_c) per position  (forward: y, hn, rs stored)
//   MODE 2:  gh = that sum over the gradient gy (flipped taps, the [cin][taps][cout] packing), then the LayerNorm + GELU backward of the layer
//            BELOW in the same epilogue:  gn = gh GELU'(gamma xhat + beta),  out = rstd (gamma gn - mean_c(gamma gn) - xhat mean_c(gamma gn xhat)),
//            per-tile partial sums of gn xhat and gn (the LayerNorm's weight / bias gradients); gh itself is never stored
//
// Until round 5 each of the 12 + 12 convs of a step ran on the generic tile kernel (20-25 us for 3.5 GFLOP, 7.8 MB tensors: launch ramp, one
// exposed window load, a K loop of 28 steps waiting on L2 weight fetches) followed by a LayerNorm launch of its own (7-9 us for 16 MB): the
// serial chain between the encoders' forward and backward.  A tile here holds ALL 128 output channels of its 64 positions, so the row
// statistics are a reduction over the workgroup's eight waves (wave w: output channels 16 w .. 16 w + 15, all 64 positions): 16-lane groups
// by two cross-row shuffles, waves through 4 KB of LDS, two-pass (mean, then centred squares) as the LayerNorm kernel does.
// The weights (896 x 128: 459 KB as bf16 hi / lo) fit neither registers nor LDS: every wave streams ITS 16-column fragments from L2 four K
// steps ahead through a register ring -- the only vector-memory stream inside the K loop, fully unrolled, so every wait is a counted one.
#include <cstdlib>
#include <type_traits>
#include "w2s_common.h"

struct SeqP {
  const float* x; const __bf16* w_hi; const __bf16* w_lo;
  float* y;                 // MODE 0 / 1: the conv output (pre-norm) [B][S][128]
  float* out;               // MODE 1: hn [B][S][128];  MODE 2: the gradient w.r.t. the lower layer's conv output [B][S][128]
  float* rs;                // MODE 1: (mean, rstd) per position [B*S][2] (written);  MODE 2: the lower layer's (read)
  const float* gamma; const float* beta;
  const float* yl;          // MODE 2: the lower layer's pre-norm conv output [B][S][128]
  float* part;              // MODE 2: [B * ntiles][2][128] partial sums (gn xhat, gn)
  int B, S, ldx, dil, padl, flip, ntiles;
  float eps;
};

template <int MODE>
__global__ __launch_bounds__(512) void seq_conv_kernel(SeqP P) {
  extern __shared__ f32x4 smem4[];
  constexpr int TM = 64, C = 128, RSE = C + 16, KS = 28;   // positions per tile; bf16 elements per LDS row; K steps of 32 (7 taps x 4)
  const int NR = TM + 6 * P.dil;                           // window rows; row 0 = position t0 - padl
  float* red = reinterpret_cast<float*>(smem4);            // [2][8 waves][64 rows] partial row sums
  __bf16* hiL = reinterpret_cast<__bf16*>(red + 2 * 8 * TM);
  __bf16* loL = hiL + NR * RSE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / P.ntiles, tile = blockIdx.x - b * P.ntiles;
  const int t0 = tile * TM;

  // ---- this wave's weight fragments: K step ks = tap * 4 + q of output-channel tile `wave`, from L2, four steps ahead
  const char* wfh = reinterpret_cast<const char*>(P.w_hi) + ((size_t)wave * KS << 10) + (lane << 4);
  const char* wfl = reinterpret_cast<const char*>(P.w_lo) + ((size_t)wave * KS << 10) + (lane << 4);
#ifndef W2S_SEQ_RING
#define W2S_SEQ_RING 4   // tuning: K steps of weight fragments in flight (8 registers each); 8 measured 2-4 us SLOWER per launch (lab notes r6)
#endif
  constexpr int RING = W2S_SEQ_RING;
  bf16x8 fh[RING], fl[RING];
  auto load_frag = [&](auto SLOT, int ks) {
    constexpr int SL = decltype(SLOT)::value;
    fh[SL] = *reinterpret_cast<const bf16x8*>(wfh + ((unsigned)ks << 10));
    fl[SL] = *reinterpret_cast<const bf16x8*>(wfl + ((unsigned)ks << 10));
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>; using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>;
  // the first RING steps' fragments go out before anything else (L2 hits: they are back long before the window's HBM rows)
  load_frag(I0{}, 0); load_frag(I1{}, 1); load_frag(I2{}, 2); load_frag(I3{}, 3);
  if constexpr (RING == 8) { load_frag(I4{}, 4); load_frag(I5{}, 5); load_frag(I6{}, 6); load_frag(I7{}, 7); }

  // ---- stage the window: 32 float4 per row, 16 rows per pass, eight passes of loads in flight; rows outside the sample are zero padding
  {
    const int myc4 = tid & 31, row0 = tid >> 5;
    const float* xb = P.x + (size_t)b * P.S * P.ldx + myc4 * 4;
    const int rb = t0 - P.padl;
    for (int base = 0; base < NR; base += 16 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int gr = min(max(rb + base + row0 + 16 * k, 0), P.S - 1);   // unconditional loads from clamped rows
        v[k] = ld4o(xb, (unsigned)gr * (unsigned)P.ldx);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int row = base + row0 + 16 * k, gr = rb + row;
        if (row < NR) split_store4(hiL, loL, row * RSE + myc4 * 4, (gr >= 0 && gr < P.S) ? v[k] : (f32x4){0, 0, 0, 0});
      }
    }
  }
  // epilogue operands of MODE 2 (the lower layer's conv output in this lane's D-fragment layout, its row statistics): requested now, they
  // land during the K loop (older than every weight fetch the loop waits for)
  const int ch = wave * 16 + 4 * g;
  f32x4 yl[MODE == 2 ? 4 : 1];
  float rmean[MODE == 2 ? 4 : 1], rrstd[MODE == 2 ? 4 : 1];
  if constexpr (MODE == 2) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int pos = min(t0 + mt * 16 + r, P.S - 1);
      yl[mt] = ld4o(P.yl + (size_t)b * P.S * C, (unsigned)pos * C + ch);
      const float* rp = P.rs + ((size_t)b * P.S + pos) * 2;
      rmean[mt] = rp[0]; rrstd[mt] = rp[1];
    }
  }
  __syncthreads();

  // ---- K loop: 7 taps x 4 steps of 32 input channels, fully unrolled (the ring's slots are compile-time, the waits counted)
  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0, 0, 0, 0};
  const int dil = P.dil, flip = P.flip;
  auto step = [&](auto SLOT, int ks) {
    constexpr int SL = decltype(SLOT)::value;
    const int j = ks >> 2, q = ks & 3;
    const int rowoff = (flip ? 6 - j : j) * dil;
    const bf16x8 ah = fh[SL], al = fl[SL];
    load_frag(SLOT, ks + RING < KS ? ks + RING : KS - 1);   // (past the last step: a harmless re-fetch instead of a branch around the loads)
    bf16x8 bh[4], bl[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      bh[mt] = *reinterpret_cast<const bf16x8*>(hiL + (mt * 16 + r + rowoff) * RSE + q * 32 + 8 * g);
      bl[mt] = *reinterpret_cast<const bf16x8*>(loL + (mt * 16 + r + rowoff) * RSE + q * 32 + 8 * g);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[mt], acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[mt], acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[mt], acc[mt], 0, 0, 0);
    }
  };
  if constexpr (RING == 8) {
#pragma unroll
    for (int ks = 0; ks < 24; ks += 8) {
      step(I0{}, ks); step(I1{}, ks + 1); step(I2{}, ks + 2); step(I3{}, ks + 3); step(I4{}, ks + 4); step(I5{}, ks + 5); step(I6{}, ks + 6); step(I7{}, ks + 7);
    }
    step(I0{}, 24); step(I1{}, 25); step(I2{}, 26); step(I3{}, 27);
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ks += 4) { step(I0{}, ks); step(I1{}, ks + 1); step(I2{}, ks + 2); step(I3{}, ks + 3); }
  }

  // ---- epilogue.  Lane (r, g) of wave w holds, for mt = 0..3, position t0 + 16 mt + r, channels 16 w + 4 g .. + 3
  // sum over the 128 channels of a row of up to two values per lane at once: 4 channels in the lane -> the wave's 16 (two cross-row
  // shuffles over g) -> the 8 waves through LDS; every lane ends with the full sums of ITS four rows
  auto rowsum2 = [&](float (&a)[4], float (&c)[4]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      a[mt] += __shfl_xor(a[mt], 16); a[mt] += __shfl_xor(a[mt], 32);
      c[mt] += __shfl_xor(c[mt], 16); c[mt] += __shfl_xor(c[mt], 32);
    }
    __syncthreads();   // the previous use of `red` has been read by everyone
    if (g == 0) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) { red[wave * TM + mt * 16 + r] = a[mt]; red[(8 + wave) * TM + mt * 16 + r] = c[mt]; }
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float s = 0.f, u = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { s += red[w * TM + mt * 16 + r]; u += red[(8 + w) * TM + mt * 16 + r]; }   // fixed order: deterministic
      a[mt] = s; c[mt] = u;
    }
  };
  const size_t ob = (size_t)b * P.S * C;
  if constexpr (MODE == 0) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int pos = t0 + mt * 16 + r;
      if (pos < P.S) st4o(P.y + ob, (unsigned)pos * C + ch, acc[mt]);
    }
  } else if constexpr (MODE == 1) {
    const f32x4 gm = ld4(P.gamma + ch), bt = ld4(P.beta + ch);
    float s1[4], z0[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { s1[mt] = (acc[mt].x + acc[mt].y) + (acc[mt].z + acc[mt].w); z0[mt] = 0.f; }
    rowsum2(s1, z0);
    float q2[4];
    f32x4 dv[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      s1[mt] *= (1.0f / C);                       // mean
      dv[mt] = acc[mt] - s1[mt];
      q2[mt] = (dv[mt].x * dv[mt].x + dv[mt].y * dv[mt].y) + (dv[mt].z * dv[mt].z + dv[mt].w * dv[mt].w);
      z0[mt] = 0.f;
    }
    rowsum2(q2, z0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int pos = t0 + mt * 16 + r;
      if (pos >= P.S) continue;
      const float rstd = 1.0f / sqrtf(q2[mt] * (1.0f / C) + P.eps);
      st4o(P.y + ob, (unsigned)pos * C + ch, acc[mt]);
      st4o(P.out + ob, (unsigned)pos * C + ch, gelu4(gm * (dv[mt] * rstd) + bt));
      if (wave == 0 && g == 0) { float* rp = P.rs + ((size_t)b * P.S + pos) * 2; rp[0] = s1[mt]; rp[1] = rstd; }
    }
  } else {
    const f32x4 gm = ld4(P.gamma + ch), bt = ld4(P.beta + ch);
    f32x4 xh[4], gn[4];
    float sa[4], sb[4];
    f32x4 dg = {0, 0, 0, 0}, db = {0, 0, 0, 0};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const bool ok = t0 + mt * 16 + r < P.S;
      xh[mt] = (yl[mt] - rmean[mt]) * rrstd[mt];
      gn[mt] = ok ? acc[mt] * gelu_grad4(gm * xh[mt] + bt) : (f32x4){0, 0, 0, 0};   // (rows past the sample: no contribution to any sum)
      const f32x4 t = gm * gn[mt];
      sa[mt] = (t.x + t.y) + (t.z + t.w);
      const f32x4 u = t * xh[mt];
      sb[mt] = (u.x + u.y) + (u.z + u.w);
      dg += gn[mt] * xh[mt];
      db += gn[mt];
    }
    rowsum2(sa, sb);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int pos = t0 + mt * 16 + r;
      if (pos >= P.S) continue;
      const float a = sa[mt] * (1.0f / C), c = sb[mt] * (1.0f / C);
      st4o(P.out + ob, (unsigned)pos * C + ch, rrstd[mt] * (gm * gn[mt] - a - xh[mt] * c));
    }
    // LayerNorm weight / bias gradient partials of this tile: sum over its 64 positions = 4 rows in the lane, 16 lanes of the row group
    row16_sum8(dg, db);
    if (r == 0) {
      float* d = P.part + ((size_t)blockIdx.x * 2) * C + ch;
      st4(d, dg);
      st4(d + C, db);
    }
  }
}

// include/w2s.h: w2s_seq_conv
extern "C" int w2s_seq_conv(const w2s_seq_conv_args* ap, void* stream) {
  if (!ap) return W2S_EINVAL;
  const w2s_seq_conv_args& a = *ap;
  if (!a.x || !a.w_hi || !a.w_lo || a.B <= 0 || a.S <= 0 || a.dil < 1 || a.dil > 32 || a.pad < 0 || a.pad > 6 * a.dil || (a.ldx & 3) || a.ldx < 128) return W2S_EINVAL;
  if (a.mode < 0 || a.mode > 2) return W2S_EINVAL;
  if (a.mode != 2 && !a.y) return W2S_EINVAL;
  if (a.mode == 1 && (!a.out || !a.rs || !a.gamma || !a.beta)) return W2S_EINVAL;
  if (a.mode == 2 && (!a.out || !a.rs || !a.gamma || !a.beta || !a.yl || !a.part)) return W2S_EINVAL;
  if ((size_t)a.S * (size_t)a.ldx * 4 >= ((size_t)1 << 32)) return W2S_EINVAL;
  const int ntiles = (a.S + 63) / 64;
  if ((long)a.B * ntiles > 0x7fffffffL) return W2S_EINVAL;
  SeqP P{a.x, static_cast<const __bf16*>(a.w_hi), static_cast<const __bf16*>(a.w_lo), a.y, a.out, a.rs, a.gamma, a.beta, a.yl, a.part,
         a.B, a.S, a.ldx, a.dil, a.pad, a.flip, ntiles, a.eps};
  const size_t lds = (size_t)2 * 8 * 64 * 4 + (size_t)2 * (64 + 6 * a.dil) * (128 + 16) * 2;   // row-sum scratch + the window's hi / lo planes (dil 32: 151.5 KB)
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  void (*kern)(SeqP) = a.mode == 0 ? seq_conv_kernel<0> : a.mode == 1 ? seq_conv_kernel<1> : seq_conv_kernel<2>;
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return W2S_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(a.B * ntiles), dim3(512), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}
