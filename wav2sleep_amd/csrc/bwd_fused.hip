// Fused backward of one encoder ConvLayer1D (k=3, pad=1; stride 1, or stride 2 = "UP2") for the HBM-bound layers
// (<= 32 channels): data gradient AND weight gradient from ONE pass over the tensors.
//
// The separate dgrad (conv_cl flip/UP2) and wgrad kernels read exactly the same three tensors (incoming gradient g,
// the layer's pre-norm output y_k, the layer's pre-norm input y_{k-1}); at 16/32 channels both are bandwidth-bound and
// the matrix cores idle ~75 % of the time, so doing both products on the same staged LDS tiles removes a third of the
// backward traffic of those layers.  Persistent workgroups walk position tiles (grid-stride), keep the weight-gradient
// accumulators in registers across tiles, and write one slab per workgroup at the end (summed by w2s_wgrad_reduce in a
// fixed order => deterministic).
//
//   gy  = rstd_k * (gn - s1 - n_k*s2),  n_k = (y_k - mean_k)*rstd_k,  gn = g            (PRO_INBWD)
//                                                                      gn = g*GELU'(n_k) (PRO_INBWD_GP, conv3: g = dL/d(block pre-act))
//   h   = GELU(IN(xin))  (st_in given)   or   GELU(xin)  (xin = previous block's stored pre-activation)
//   dgrad:  d[t'][c] = sum_{j,o} W[o][c][j] * gy[(t'+1-j)/stride][o]            (wb = [c][j][o])
//   out  :  gout[t'][c] = (d [+ add_even[t'/2] if t' even]) * GELU'(n_in[t'][c]);  partial sums of gout, gout*n_in
//   wgrad:  dW[o][j][c] = sum_t gy[t][o] * h[t*stride + j - 1][c]
#include <type_traits>
#include "conv_cl.inl"

struct BwdP {
  const float* g; const float* y; const float* st_k; const float* bst_k;
  const float* xin; const float* st_in; const float* add_even; const float* wb;
  float* gout; float* part; float* slab;
  int B, Lg, Lh, ntiles, pro, pad;   // pad 1: symmetric; 2: causal left padding (split-precision kernels only)
  // residual fold (conv1 of a block, split-precision kernel only): gpre = dL/d(block pre-activation) [B][Lh/2][GC],
  // wd = 1x1/stride-2 downsample weight as [HC][GC], slab_d = its weight-gradient slabs (one per workgroup)
  const float* gpre; const float* wd; float* slab_d;
  // first-layer recompute (conv2 of block 0, split-precision kernel only): xin = the raw signal [B][Lh], w1 = conv1 weight [16][3]
  const float* w1;
  // previous block's conv3 backward statistics folded into this block's conv1 (RD) kernel: y3p = that block's pre-norm conv3
  // output [B][Lh][HC] (same positions as gout), st3p = its (mean, rstd); part then receives sums of gn = gout*GELU'(n3) and gn*n3
  const float* y3p; const float* st3p;
  // fp16 gradient chain (split-precision kernels; w2s_common.h): gmode 0 = g / gpre / gout are fp32; 1 = g fp32 (header hdr_g: scale 1,
  // max from w2s_gp_stats), gout fp16; 2 = g, gpre and gout fp16.  hdr_o[1] must be zero at launch.
  int gmode; const float* hdr_g; const float* hdr_p; float* hdr_o;
  // first-layer weight gradient folded in (first-layer recompute form only): part_w1[b][tile][16][3] = sum over the tile's positions of
  // gout[t][o] * xs[t + j - pad] (xs = the sanitised, zero-padded signal) -- what w2s_enc_first_wgrad turns into dW1 without gout ever
  // being stored (gout may then be NULL)
  float* part_w1;
  // block 0's downsample weight gradient folded into block 1's conv1 (residual-fold) kernel, whose gout IS block 0's gpre:
  // part_wd[workgroup][16] = sum over the workgroup's tiles and positions u of gout[u][o] * san(x0[b][2u]), x0 = the raw signal [B][2 Lh]
  const float* x0; float* part_wd;
};

// LDS row strides: 16-channel rows stay unpadded (64-B rows: the three windows + weights of the 16x16 kernel then fit
// three workgroups per CU = 50 % more loads in flight); wider rows get 4 floats of padding against bank conflicts.
#ifndef W2S_BF_OCC3
#define W2S_BF_OCC3 1
#endif
__host__ __device__ constexpr int bwd_rs(int c) { return (c > 16 || !W2S_BF_OCC3) ? c + 4 : c; }
__host__ __device__ constexpr int bwd_redn(int ch) { return 4 * ch * 4 * 8; }  // floats of the statistics scratch

template <int CG, int CH, int MT, int UP2, int PF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((CG == 1 && CH == 1 && W2S_BF_OCC3) ? 3 : 2)))
void bwd_fused_kernel(BwdP P) {
  extern __shared__ f32x4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);
  constexpr int TM = 64 * MT;                       // output (h-side) positions per tile
  constexpr int GC = CG * 16, HC = CH * 16;         // channels on the gradient / input side
  constexpr int RSg = bwd_rs(GC), RSh = bwd_rs(HC);
  constexpr int NRg = UP2 ? TM / 2 + 1 : TM + 2;    // gy window rows
  constexpr int NRh = UP2 ? TM + 1 : TM + 2;        // h window rows (row 0 = position t0-1)
  float* gyL = smem;
  float* hL = smem + NRg * RSg;
  float* nL = hL + NRh * RSh;                       // normalised input n_in of the TM centre rows (epilogue: GELU'(n), stats)
  float* red = nL + TM * RSh;                       // [4][CH][4][8] stats scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int Lg = P.Lg, Lh = P.Lh;
  const int G = (int)gridDim.x;

  f32x4 accw[CG][3][CH];
#pragma unroll
  for (int i = 0; i < CG; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int c = 0; c < CH; ++c) accw[i][j][c] = (f32x4){0, 0, 0, 0};

  // ---- dgrad weights [HC][3][GC] -> LDS once (persistent kernel): the MFMA loops then touch no global memory, so the
  //      next tile's prefetch (below) is not drained by an in-order vmcnt wait on a weight load.
  constexpr int WROW = 3 * GC + 4;
  float* wL = red + bwd_redn(CH);
  for (int i = tid; i < HC * (3 * GC / 4); i += 256) {
    const int row = i / (3 * GC / 4), c4 = i % (3 * GC / 4);
    st4(wL + row * WROW + c4 * 4, ld4(P.wb + (size_t)row * (3 * GC) + c4 * 4));
  }

  // ---- software pipeline: raw global data of tile i+1 is prefetched into registers while tile i computes
  constexpr int c4g = GC / 4, rstep_g = 256 / c4g, NG = (NRg + rstep_g - 1) / rstep_g;
  constexpr int c4h = HC / 4, rstep_h = 256 / c4h, NH = (NRh + rstep_h - 1) / rstep_h;
  const int gc4 = tid % c4g, grow0 = tid / c4g, gch = gc4 * 4;
  const int hc4 = tid % c4h, hrow0 = tid / c4h, hch = hc4 * 4;
  f32x4 rg[NG], ry[NG], rh[NH];
  auto prefetch = [&](int b, int tile) {   // (sample, tile): carried incrementally by the tile loop, no division per tile
    const int t0 = tile * TM;
    const float* gb = P.g + (size_t)b * Lg * GC + gch;
    const float* yb = P.y + (size_t)b * Lg * GC + gch;
    const int rb = UP2 ? t0 / 2 : t0 - 1;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int row = grow0 + k * rstep_g, gr = rb + row;
      const bool ok = row < NRg && gr >= 0 && gr < Lg;
      rg[k] = ok ? ld4(gb + (size_t)gr * GC) : (f32x4){0, 0, 0, 0};
      ry[k] = ok ? ld4(yb + (size_t)gr * GC) : (f32x4){0, 0, 0, 0};
    }
    const float* xb = P.xin + (size_t)b * Lh * HC + hch;
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const int row = hrow0 + k * rstep_h, gr = t0 - 1 + row;
      const bool ok = row < NRh && gr >= 0 && gr < Lh;
      rh[k] = ok ? ld4(xb + (size_t)gr * HC) : (f32x4){0, 0, 0, 0};
    }
  };
  auto commit = [&](int b, int tile) {  // transform the prefetched registers and write both windows to LDS
    const int t0 = tile * TM;
    f32x4 pm, pr, ps1, ps2;
    {
      const float* st = P.st_k + ((size_t)b * GC + gch) * 2;
      f32x4 s01 = ld4(st), s23 = ld4(st + 4);
      pm = (f32x4){s01.x, s01.z, s23.x, s23.z}; pr = (f32x4){s01.y, s01.w, s23.y, s23.w};
      const float* bs = P.bst_k + ((size_t)b * GC + gch) * 2;
      f32x4 b01 = ld4(bs), b23 = ld4(bs + 4);
      ps1 = (f32x4){b01.x, b01.z, b23.x, b23.z}; ps2 = (f32x4){b01.y, b01.w, b23.y, b23.w};
    }
    const int rb = UP2 ? t0 / 2 : t0 - 1;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int row = grow0 + k * rstep_g, gr = rb + row;
      if (row < NRg) {
        const bool ok = gr >= 0 && gr < Lg;
        st4(gyL + row * RSg + gch, ok ? pro_apply(UP2 ? W2S_PRO_INBWD_GP : W2S_PRO_INBWD, rg[k], ry[k], pm, pr, ps1, ps2) : (f32x4){0, 0, 0, 0});
      }
    }
    f32x4 hm = {0, 0, 0, 0}, hr = {1, 1, 1, 1};
    if (P.st_in) {
      const float* st = P.st_in + ((size_t)b * HC + hch) * 2;
      f32x4 s01 = ld4(st), s23 = ld4(st + 4);
      hm = (f32x4){s01.x, s01.z, s23.x, s23.z}; hr = (f32x4){s01.y, s01.w, s23.y, s23.w};
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const int row = hrow0 + k * rstep_h, gr = t0 - 1 + row;
      if (row < NRh) {
        const bool ok = gr >= 0 && gr < Lh;
        const f32x4 nv = (rh[k] - hm) * hr;
        st4(hL + row * RSh + hch, ok ? gelu4(nv) : (f32x4){0, 0, 0, 0});
        if (row >= 1 && row <= TM) st4(nL + (row - 1) * RSh + hch, nv);
      }
    }
  };

  // tiles of this workgroup: a contiguous run of the (sample, tile) list (blocked; see w2s_common.h "Statistics finalisation")
  const int total = P.B * P.ntiles;
  const W2SRun wrun = w2s_block_part(total, G, blockIdx.x);   // the producers never exceed the tile count
  const int wend = wrun.first + wrun.count;
  int b = wrun.first / P.ntiles, tile = wrun.first - b * P.ntiles;   // one division per launch; the run is contiguous
  if (PF && wrun.count > 0) prefetch(b, tile);
  for (int tl = wrun.first; tl < wend; ++tl) {
    const int t0 = tile * TM;
    const int tile_n = tile + 1 == P.ntiles ? 0 : tile + 1, b_n = tile + 1 == P.ntiles ? b + 1 : b;   // the next tile of the run
    __syncthreads();  // everyone is done reading the previous tile's windows (and wL is written)
    if (!PF) prefetch(b, tile);
    commit(b, tile);
    if (PF && tl + 1 < wend) prefetch(b_n, tile_n);
    __syncthreads();

    // ---- data gradient: this wave's 16*MT output positions x HC channels
    f32x4 acc[MT][CH];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < CH; ++nt) acc[mt][nt] = (f32x4){0, 0, 0, 0};
    auto mma_tap = [&](int jw, int rowoff, int mtmask) {
#pragma unroll
      for (int q = 0; q < CG; ++q) {
        f32x4 bf[MT], af[CH];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int row = UP2 ? wave * (8 * MT) + (mt >> 1) * 16 + r + rowoff : wave * (16 * MT) + mt * 16 + r + rowoff;
          bf[mt] = *reinterpret_cast<const f32x4*>(gyL + row * RSg + q * 16 + 4 * g);
        }
#pragma unroll
        for (int nt = 0; nt < CH; ++nt) af[nt] = *reinterpret_cast<const f32x4*>(wL + (nt * 16 + r) * WROW + jw * GC + q * 16 + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            if (mtmask & (1 << mt))
#pragma unroll
              for (int nt = 0; nt < CH; ++nt) acc[mt][nt] = mfma16(af[nt][e], bf[mt][e], acc[mt][nt]);
      }
    };
    if (UP2) {
      constexpr int EVEN = 0x55 & ((1 << MT) - 1), ODD = 0xAA & ((1 << MT) - 1);
      mma_tap(1, 0, EVEN);
      mma_tap(2, 0, ODD);
      mma_tap(0, 1, ODD);
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j) mma_tap(j, 2 - j, (1 << MT) - 1);  // window row 0 = t0-1: gy[t'+1-j] is row (t'-t0) + (2-j)
    }

    // ---- epilogue: * GELU'(n_in), statistics, store
    f32x4 sA[CH], sB[CH];
#pragma unroll
    for (int nt = 0; nt < CH; ++nt) { sA[nt] = (f32x4){0, 0, 0, 0}; sB[nt] = (f32x4){0, 0, 0, 0}; }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int pos = UP2 ? t0 + 2 * (wave * (8 * MT) + (mt >> 1) * 16 + r) + (mt & 1) : t0 + wave * (16 * MT) + mt * 16 + r;
      if (pos >= Lh) continue;
#pragma unroll
      for (int nt = 0; nt < CH; ++nt) {
        const int ch = nt * 16 + 4 * g;
        const size_t orow = (size_t)b * Lh + pos;
        const f32x4 n = *reinterpret_cast<const f32x4*>(nL + (pos - t0) * RSh + ch);
        f32x4 v = acc[mt][nt];
        if (P.add_even && !(pos & 1)) v += ld4(P.add_even + ((size_t)b * (Lh >> 1) + (pos >> 1)) * HC + ch);
        v = v * gelu_grad4(n);
        sA[nt] += v;
        sB[nt] += v * n;
        st4(P.gout + orow * HC + ch, v);
      }
    }
    if (P.part) {
#pragma unroll
      for (int nt = 0; nt < CH; ++nt) {
        f32x4 x1, x2;
        x1 = sA[nt]; x2 = sB[nt];
        row16_sum8(x1, x2);
        if (r == 0) {
          float* d = red + ((wave * CH + nt) * 4 + g) * 8;
          st4(d, x1);
          st4(d + 4, x2);
        }
      }
      __syncthreads();
      if (tid < CH * 32) {
        const int k = tid / HC, c = tid % HC;
        const int nt = c >> 4, gg = (c >> 2) & 3, e = c & 3;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[((w * CH + nt) * 4 + gg) * 8 + k * 4 + e];
        w2s_part_store(&P.part[(((size_t)b * P.ntiles + tile) * 2 + k) * HC + c], s);
      }
    }

    // ---- weight gradient: k-step = 4 positions of the gradient side
    constexpr int KPW = (UP2 ? TM / 2 : TM) / 4;  // gradient-side positions per wave
    for (int s = 0; s < KPW / 4; ++s) {
      const int p = wave * KPW + 4 * s + g;       // tile-local gradient-side position
      float ga[CG], hb[3][CH];
      const int grow = UP2 ? p : p + 1;
#pragma unroll
      for (int i = 0; i < CG; ++i) ga[i] = gyL[grow * RSg + i * 16 + r];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int hr = UP2 ? 2 * p + j : p + j;
#pragma unroll
        for (int c = 0; c < CH; ++c) hb[j][c] = hL[hr * RSh + c * 16 + r];
      }
#pragma unroll
      for (int i = 0; i < CG; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int c = 0; c < CH; ++c) accw[i][j][c] = mfma16(ga[i], hb[j][c], accw[i][j][c]);
    }
    b = b_n; tile = tile_n;
  }

  // ---- one slab per workgroup: sum the 4 waves tile by tile through LDS (fixed order), raw-fragment layout of
  //      wgrad_kernel<CG, CH, 3, *> with gridDim.y == 1: [tile(i,j,c)][lane][4]
  float* out = P.slab + (size_t)blockIdx.x * (CG * 3 * CH) * 256;
#pragma unroll
  for (int i = 0; i < CG; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        __syncthreads();  // (first pass: every wave is done with the windows; the scratch aliases them)
        st4(smem + (wave * 64 + lane) * 4, accw[i][j][c]);
        __syncthreads();
        if (wave == 0) {
          f32x4 v = ld4(smem + lane * 4) + ld4(smem + (64 + lane) * 4) + ld4(smem + (128 + lane) * 4) + ld4(smem + (192 + lane) * 4);
          st4(out + ((i * 3 + j) * CH + c) * 256 + lane * 4, v);
        }
      }
}

template <int CG, int CH, int MT, int UP2, int PF>
static int launch_bwd(const BwdP& P0, int nslab, hipStream_t s) {
  constexpr int TM = 64 * MT;
  BwdP P = P0;
  P.ntiles = (P.Lh + TM - 1) / TM;
  constexpr int NRg = UP2 ? TM / 2 + 1 : TM + 2, NRh = UP2 ? TM + 1 : TM + 2;
  size_t lds = (size_t)(NRg * bwd_rs(CG * 16) + NRh * bwd_rs(CH * 16) + TM * bwd_rs(CH * 16)) * 4;
  lds += (size_t)bwd_redn(CH) * 4 + (size_t)(CH * 16) * (3 * CG * 16 + 4) * 4;
  lds = (lds + 15) & ~(size_t)15;
  auto kern = bwd_fused_kernel<CG, CH, MT, UP2, PF>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return W2S_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(nslab), dim3(256), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Split-precision ("bf16x3") form for 32 gradient-side channels: the fp32 kernel above spends ~70 % of a tile's time
// budget in the matrix pipe at 32x32 channels (2.7 TB/s); with both windows staged as bf16 (hi, lo) planes every product
// is 3 x v_mfma_f32_16x16x32_bf16 (hh + hl + lh, fp32 accumulate; K = 32 = the gradient channels for the data gradient,
// = 32 positions for the weight gradient, whose operands are columns of the row-major tiles => ds_read_b64_tr_b16).
// Weight-gradient tiles are split over waves by (cout tile, cin tile) so each wave owns its 3 tap accumulators: no
// cross-wave reduction at the end.  Staging, prefetch, epilogue and slab layout are those of bwd_fused_kernel.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x4t __attribute__((__vector_size__(4 * sizeof(__bf16))));
__device__ __forceinline__ bf16x8 lds_tr8(const __bf16* p0, const __bf16* p1) {
  typedef __attribute__((address_space(3))) bf16x4t* lds_p;
  bf16x4t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p0));
  bf16x4t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(p1));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ f32x4 mfma_bf3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
  return c;
}

// RD = 1 (conv1 of a residual block, stride 1): the block's 1x1/stride-2 residual branch is folded in -- its data gradient
// Wd^T gpre[t/2] (even t) rides in the K axis of the conv's own data gradient (16 channels: the unused half of the second
// K step; 32 channels: one more K step), and its weight gradient sum_u gpre[u] x h[2u] comes from the h window that is
// already staged.  Replaces a 1x1 conv launch (+ its output tensor, written and re-read) and a weight-gradient launch that
// re-read both gpre and the block input.
__host__ __device__ constexpr int bf_rs(int c) { return c == 16 ? 16 : c + 8; }  // 32-B rows are conflict-free as they are
// row stride (bf16 elements) of the h planes: stride 1 as the gy planes; stride 2: a lane's sixteen-lane store group covers EVERY OTHER row
// (the even / odd outputs are separate M tiles), so 32-B rows would put eight lanes on one bank pair -- 40-B / 72-B rows leave 2-way
__host__ __device__ constexpr int bf_rsh(int c, int up2) { return up2 ? (c == 16 ? 20 : 36) : bf_rs(c); }   // (16 / 40 and 24 / 44 measured the same)
#ifndef W2S_BF_HLO
#define W2S_BF_HLO 1   // 1: h = GELU(n) staged as bf16 hi + lo planes (three MFMAs per weight-gradient product); 0: hi plane only (two)
#endif
// FIRST = 1 (conv2 of block 0): the input side is block 0's conv1 output, which is never stored -- it is recomputed from
// the raw 1-channel signal while the window is staged (3 FMAs per element instead of a 64-B row per position).
// Occupancy: the kernels take what their registers allow (2 waves per SIMD; measured: forcing 2 on the variants that land on 1 changes
// nothing).  The residual-fold kernel of the 16-channel blocks is the exception: at 256-position tiles it needs 244 registers (one
// workgroup per CU); with 128-position tiles it fits three per CU and runs 21 % faster (1.39 -> 1.10 ms per step).
#ifndef W2S_BF_OCC11
#define W2S_BF_OCC11 1   // tuning: waves per SIMD asked for the plain 16 -> 16 kernels (with W2S_BF_MT11 = 2: 128-position tiles)
#endif
#ifndef W2S_BF_OCC22
#define W2S_BF_OCC22 1   // tuning: the same for the 32-channel kernels ((32,32) both strides, (32,16) fold)
#endif
__host__ __device__ constexpr int bfk_occ(int cg, int ch, int rd) { return (cg == 1 && ch == 1) ? W2S_BF_OCC11 : W2S_BF_OCC22; }
template <int CG, int CH, int MT, int UP2, int RD, int FIRST, int GM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(bfk_occ(CG, CH, RD))))
void bwd_fused_bf_kernel(BwdP P) {
  static_assert(GM != 1 || !RD, "a residual-fold kernel reads the chain from both sides: all fp32 or all fp16");
  constexpr bool GH = (GM == 2), OH = (GM != 0);   // g (and gpre) / gout stored as fp16
  using GRaw = std::conditional_t<GH, h16x4, f32x4>;
  float inv_g = 1.f, inv_p = 1.f, s_out = 1.f, amax = 0.f;
  if (OH) {   // uniform: the headers are final (their producers have completed)
    float ref = P.hdr_g[1];
    if (GH) inv_g = 1.f / P.hdr_g[0];
    if (RD) { ref = fmaxf(ref, P.hdr_p[1]); inv_p = 1.f / P.hdr_p[0]; }
    s_out = w2s_gscale_for(ref);
  }
  extern __shared__ f32x4 smem4[];
  constexpr int TM = 64 * MT;
  constexpr int GC = CG * 16, HC = CH * 16;
  constexpr int RSg = bf_rs(GC), RSh = bf_rsh(HC, UP2);   // bf16 elements per row of the gy / h planes
  static_assert(!(RD && UP2), "the residual fold belongs to the stride-1 conv1");
  static_assert(!FIRST || (HC == 16 && !UP2 && !RD), "first-layer recompute: conv2 of block 0");
  // A tile is TS = TM - 2 positions: its two windows (the positions and one halo row on each side) are then exactly TM rows -- a whole
  // number of staging passes (64 or 32 rows each).  With TM-position tiles the TM + 2 rows cost every tile a fifth pass for two rows,
  // executed by wave 0 alone at the price of a full one while the other three waited at the barrier (in-kernel stamps, lab notes r4
  // section 13: wave 0's commit 11-16 % longer than its siblings').  The MFMA tiles still cover TM positions; the last two are computed
  // on rows past the window (finite LDS contents) and neither stored nor summed, and the weight gradient masks them out of its K axis.
  constexpr int TS = TM - 2;
  constexpr int NRg = UP2 ? TM / 2 : TM;
  constexpr int NRh = TM;                                   // h window rows; row 0 = position t0 - pad
  const int PL = P.pad;                                     // 1: symmetric padding; 2: causal (left pad 2)
  // data-gradient K axis of the LDS weight image.  32 gradient channels: k = tap*32 + o (one tap per MFMA).
  // 16 gradient channels: two taps share one K = 32 step: stride 1: [tap0 | tap1] [tap2 | 0];
  // stride 2 (even outputs use tap 1, odd outputs taps 2 and 0): [tap1 | 0] [tap2 | tap0].
  constexpr int KD = (GC == 32) ? (RD ? 128 : 96) : 64;   // RD: [tap2 | Wd] (16 ch) / a fourth K step = Wd (32 ch)
  constexpr int WROW = KD + 8;
  // Round 5: the input-side window is staged by the lane that OWNS the position in the data gradient's D fragment (position = lane & 15 of
  // the wave's 16-position M tile, channels 4 * (lane >> 4) ...: bwd_hrow below), so n = IN(x) and GELU'(n) of the tile's positions never
  // leave their registers between the staging pass (one erf for GELU and GELU') and the epilogue (gout = d * GELU'(n), sum gout * n).
  // Rounds 3-4 kept them in two fp32 LDS planes (32 KB of the 16-channel kernels' 70 KB: two workgroups per CU): a ds_write_b128 + ds_read_b128
  // pair per plane and 4 elements.  Without them the 16 -> 16 kernels hold 31 KB and the register file sets the occupancy.
  // HLO: the lo plane of h = GELU(n), which only the weight gradient reads (dW = gy * h: gy_hi h_hi + gy_lo h_hi [+ gy_hi h_lo]).
  constexpr bool HLO = W2S_BF_HLO != 0;
  float* red = reinterpret_cast<float*>(smem4);             // [4][CH][4][8] stats scratch
  __bf16* gyH = reinterpret_cast<__bf16*>(red + bwd_redn(CH));
  __bf16* gyLo = gyH + NRg * RSg;
  __bf16* hH = gyLo + NRg * RSg;
  __bf16* hLo = hH + NRh * RSh;                             // (HLO == 0: not there -- the weight planes follow hH)
  __bf16* wH = hH + (HLO ? 2 : 1) * NRh * RSh;              // [HC][WROW]
  __bf16* wLo = wH + HC * WROW;
  constexpr int NRp = RD ? TM / 2 + 1 : 0;                  // gpre rows of the tile + one all-zero row (odd output positions)
  __bf16* pH = wLo + HC * WROW;
  __bf16* pLo = pH + NRp * RSg;
  float* xsL = reinterpret_cast<float*>(pLo + NRp * RSg);   // FIRST: TM + 4 signal samples
  float* redA = xsL + TM + 4;                               // FIRST: [4 waves][4 lane groups][12] scratch of the folded first-layer weight gradient
  float* redD = xsL;                                        // RD + part_wd: [4 waves][4 lane groups][4] scratch of the folded downsample weight gradient
  float* accD = redD + 64;                                  //               [16] running sums of this workgroup (thread 112 + o owns entry o)
  constexpr bool WDFC = RD && CG == 1 && CH == 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4, q4 = r >> 2, p4 = r & 3;
  const int Lg = P.Lg, Lh = P.Lh;
  const int G = (int)gridDim.x;
  if (WDFC && tid >= 112 && tid < 128) accD[tid - 112] = 0.f;

  // weight-gradient ownership: the CG*CH (cout tile, cin tile) pairs are spread over the 4 waves; KW waves share a
  // pair and take every KW-th k-step (summed through LDS at the end, fixed order)
  constexpr int NWT = CG * CH, KW = 4 / NWT;
  const int wt = wave / KW, wk = wave % KW, wi = wt / CH, wc = wt % CH;
  f32x4 accw[3], accd = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 3; ++j) accw[j] = (f32x4){0, 0, 0, 0};
  if (RD) {  // the zero row of the gpre planes
    for (int i = tid; i < RSg / 4; i += 256) split_store4(pH, pLo, (TM / 2) * RSg + i * 4, (f32x4){0, 0, 0, 0});
  }

  for (int i = tid; i < HC * (KD / 4); i += 256) {
    const int row = i / (KD / 4), k = (i % (KD / 4)) * 4;
    const float* wr = P.wb + (size_t)row * (3 * GC);
    f32x4 v = {0, 0, 0, 0};
    if (GC == 32) v = (k < 96) ? ld4(wr + k) : ld4(P.wd + (size_t)row * GC + (k - 96));
    else if (!UP2) { if (k < 48) v = ld4(wr + k); else if (RD) v = ld4(P.wd + (size_t)row * GC + (k - 48)); }
    else {
      const int seg = k >> 4, o = k & 15;
      if (seg == 0) v = ld4(wr + 16 + o);
      else if (seg == 2) v = ld4(wr + 32 + o);
      else if (seg == 3) v = ld4(wr + o);
    }
    split_store4(wH, wLo, row * WROW + k, v);
  }

  constexpr int c4g = GC / 4, rstep_g = 256 / c4g, NG = (NRg + rstep_g - 1) / rstep_g;
  constexpr int NH = MT * CH;   // float4s of the h window per thread: one per (M tile, channel tile) of this lane's D fragments
  const int gc4 = tid % c4g, grow0 = tid / c4g, gch = gc4 * 4;
  // tile-local position index of this lane in M tile mt (what the data gradient's D fragment holds), and the window row it stages: the
  // window's row i is position t0 - pad + i, so index m lives in row m + pad; the pad rows in front of the tile are staged by the lanes
  // of the last `pad` indices (>= TS: they own no output), i.e. row = (m + pad) mod TM -- every lane stages exactly MT * CH float4s
  auto bwd_hm = [&](int mt) { return UP2 ? 2 * (wave * (8 * MT) + (mt >> 1) * 16 + r) + (mt & 1) : wave * (16 * MT) + mt * 16 + r; };
  constexpr int NP = RD ? (TM / 2 + rstep_g - 1) / rstep_g : 1;
  GRaw rg[NG], rp[NP];
  f32x4 ry[NG], rh[FIRST ? 1 : NH], rq[RD ? MT * CH : 1];  // rq: previous block's y3 in the D-fragment layout
  f32x4 nR[NH], gR[NH];   // n = IN(x) and GELU'(n) of this lane's positions, from the staging pass to the epilogue
  float rxs[2], w1r[4][3];
  if (FIRST) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) w1r[i][j] = P.w1[(4 * g + i) * 3 + j];
  }
  // PART 0: the gradient side (g, y_k; the residual fold's gpre; the first-layer form's signal samples) -- issued right after the commit,
  // in flight during the matrix phase.  PART 1: the input side (x_in, and the previous block's y3 fragments of the statistics fold): in the
  // stride-1 forms issued AFTER the epilogue (n / GELU' of the current tile are dead by then: 16 registers less at the peak, which is what
  // lets the 16 -> 16 kernels keep three waves per SIMD without spilling); those loads have the weight-gradient phase and the next
  // commit's gradient half to land
  auto prefetch = [&](int b, int tile, auto PART) {   // (sample, tile): carried incrementally by the tile loop, no division per tile
    constexpr int PT = decltype(PART)::value;
    const int t0 = tile * TS;
    // wave-uniform 64-bit base per sample + 32-bit per-lane offsets (a sample's tensor is < 4 GB): scalar-base addressing,
    // no 64-bit VALU address arithmetic per load
    if constexpr (PT == 0) {
      const char* gb = reinterpret_cast<const char*>(P.g) + (size_t)b * Lg * GC * (GH ? 2 : 4);
      const float* yb = P.y + (size_t)b * Lg * GC;
      const int rb = UP2 ? t0 / 2 : t0 - 2 + PL;   // gy window: the data gradient reads gy[t' + pad - j] at window row (t' - t0) + 2 - j
#pragma unroll
      for (int k = 0; k < NG; ++k) {   // unconditional loads from clamped rows (rows outside the sample: zeroed in LDS by `zero_oob`)
        const int row = grow0 + k * rstep_g, gr = min(max(rb + row, 0), Lg - 1);
        const unsigned off = (unsigned)gr * GC + gch;
        if constexpr (GH) rg[k] = ld4h(gb, off);
        else rg[k] = ld4o(reinterpret_cast<const float*>(gb), off);
        ry[k] = ld4o(yb, off);
      }
      if (FIRST) {  // TM + 4 signal samples t0-2pad .. : one per thread (+4), exchanged through LDS at commit time (conv1 pads like this conv)
        const float* xs = P.xin + (size_t)b * Lh;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = tid + 256 * k, gr = t0 - 2 * PL + i;
          const float xv = xs[min(max(gr, 0), Lh - 1)];
          rxs[k] = (i < TM + 4 && gr >= 0 && gr < Lh && !isinf(xv)) ? xv : 0.f;
        }
      }
      if (RD) {
        const char* pb = reinterpret_cast<const char*>(P.gpre) + (size_t)b * (Lh >> 1) * GC * (GH ? 2 : 4);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const int row = grow0 + k * rstep_g, gr = min(t0 / 2 + row, (Lh >> 1) - 1);
          if constexpr (GH) rp[k] = ld4h(pb, (unsigned)gr * GC + gch);
          else rp[k] = ld4o(reinterpret_cast<const float*>(pb), (unsigned)gr * GC + gch);
        }
      }
    } else {
      if (!FIRST) {
        const float* xb = P.xin + (size_t)b * Lh * HC;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int row = (bwd_hm(mt) + PL) & (TM - 1), gr = min(max(t0 - PL + row, 0), Lh - 1);
#pragma unroll
          for (int nt = 0; nt < CH; ++nt) rh[mt * CH + nt] = ld4o(xb, (unsigned)gr * HC + nt * 16 + 4 * g);
        }
      }
      if (RD && P.y3p) {
        const float* qb = P.y3p + (size_t)b * Lh * HC;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < CH; ++nt) {
            const int pos = min(t0 + wave * (16 * MT) + mt * 16 + r, Lh - 1);   // (positions past the sample are not summed)
            rq[mt * CH + nt] = ld4o(qb, (unsigned)pos * HC + nt * 16 + 4 * g);
          }
      }
    }
  };
  static_assert(NRg % rstep_g == 0 && (TM / 2) % rstep_g == 0, "whole staging passes");
  auto commit = [&](int b, int tile) {
    const int t0 = tile * TS;
    // instance-norm backward of the gradient side as TWO fused multiply-adds per element: with n = (y - m) r,
    //   gy = r (g - s1 - n s2) = r g + (-r^2 s2) y + r (r s2 m - s1)
    // (stride 2, where g passes through GELU'(n) first: gy = (r g) GELU'(n) + n (-r s2) + (-r s1), n = r y + (-m r)).  The per-channel
    // coefficients are formed once per tile (a dozen operations); rounds 1-4 evaluated the textbook form: 5 (6) operations per element.
    f32x4 cA, cB, cC, cD;
    {
      const float* st = P.st_k + ((size_t)b * GC + gch) * 2;
      f32x4 s01 = ld4(st), s23 = ld4(st + 4);
      const f32x4 pm = {s01.x, s01.z, s23.x, s23.z}, pr = {s01.y, s01.w, s23.y, s23.w};
      const float* bs = P.bst_k + ((size_t)b * GC + gch) * 2;
      f32x4 b01 = ld4(bs), b23 = ld4(bs + 4);
      const f32x4 ps1 = {b01.x, b01.z, b23.x, b23.z}, ps2 = {b01.y, b01.w, b23.y, b23.w};
      cA = pr;
      if (UP2) { cB = -(pm * pr); cC = -(pr * ps2); cD = -(pr * ps1); }
      else { cB = -(pr * pr * ps2); cC = pr * (pr * ps2 * pm - ps1); cD = cC; }
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
      const int row = grow0 + k * rstep_g;
      f32x4 gv;
      if constexpr (GH) gv = h2f4(rg[k]) * inv_g; else gv = rg[k];
      f32x4 gy;
      if (UP2) {
        const f32x4 n = fma4(ry[k], cA, cB);
        gy = fma4(gv * cA, gelu_grad4(n), fma4(n, cC, cD));
      } else {
        gy = fma4(cA, gv, fma4(cB, ry[k], cC));
      }
      split_store4(gyH, gyLo, row * RSg + gch, gy);
    }
    f32x4 hr[CH], hb[CH];   // n = x * rstd + (-mean * rstd)
#pragma unroll
    for (int nt = 0; nt < CH; ++nt) {
      hr[nt] = (f32x4){1, 1, 1, 1}; hb[nt] = (f32x4){0, 0, 0, 0};
      if (P.st_in) {
        const float* st = P.st_in + ((size_t)b * HC + nt * 16 + 4 * g) * 2;
        f32x4 s01 = ld4(st), s23 = ld4(st + 4);
        hr[nt] = (f32x4){s01.y, s01.w, s23.y, s23.w};
        hb[nt] = -((f32x4){s01.x, s01.z, s23.x, s23.z} * hr[nt]);
      }
    }
    if (FIRST) {
      if (tid < TM + 4) xsL[tid] = rxs[0];
      if (tid < TM + 4 - 256) xsL[256 + tid] = rxs[1];
      __syncthreads();
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = (bwd_hm(mt) + PL) & (TM - 1);
#pragma unroll
      for (int nt = 0; nt < CH; ++nt) {
        const int k = mt * CH + nt, ch = nt * 16 + 4 * g;
        f32x4 xv;
        if (FIRST) {  // window row <-> position t0-pad+row; xsL[i] <-> position t0-2pad+i: conv1 taps at p-1,p,p+1 or p-2,p-1,p
          const float xm = xsL[row], xc = xsL[row + 1], xp = xsL[row + 2];
          xv.x = w1r[0][0] * xm + w1r[0][1] * xc + w1r[0][2] * xp;
          xv.y = w1r[1][0] * xm + w1r[1][1] * xc + w1r[1][2] * xp;
          xv.z = w1r[2][0] * xm + w1r[2][1] * xc + w1r[2][2] * xp;
          xv.w = w1r[3][0] * xm + w1r[3][1] * xc + w1r[3][2] * xp;
        } else {
          xv = rh[k];
        }
        const f32x4 nv = fma4(xv, hr[nt], hb[nt]);
        f32x4 hv;
        gelu_both4(nv, hv, gR[k]);
        nR[k] = nv;
        if (HLO) split_store4(hH, hLo, row * RSh + ch, hv);
        else {
          u32x2 h2;
          h2.x = __builtin_bit_cast(unsigned, (bf16x2){(__bf16)hv.x, (__bf16)hv.y});
          h2.y = __builtin_bit_cast(unsigned, (bf16x2){(__bf16)hv.z, (__bf16)hv.w});
          *reinterpret_cast<u32x2*>(hH + row * RSh + ch) = h2;
        }
      }
    }
    if (RD) {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int row = grow0 + k * rstep_g;
        f32x4 pv;
        if constexpr (GH) pv = h2f4(rp[k]) * inv_p; else pv = rp[k];
        split_store4(pH, pLo, row * RSg + gch, pv);
      }
    }
    // rows outside the sample (the zero padding of the conv; only a sample's first and last tiles have any -- a uniform branch): the passes
    // above loaded them from clamped addresses and transformed them like any other row; the lanes that own them now overwrite them with
    // zeros (the same thread stored the row: LDS keeps a thread's stores in order).  Rounds 1-4 predicated every load and every row of
    // every tile instead: ~120 of a tile's ~1250 instructions per wave.
    const int rbg = UP2 ? t0 / 2 : t0 - 2 + PL;
    if (rbg < 0 || rbg + NRg > Lg || t0 - PL < 0 || t0 - PL + TM > Lh) {
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        const int row = grow0 + k * rstep_g, gr = rbg + row;
        if (gr < 0 || gr >= Lg) { zero_store4(gyH, row * RSg + gch); zero_store4(gyLo, row * RSg + gch); }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = (bwd_hm(mt) + PL) & (TM - 1), gr = t0 - PL + row;
        if (gr < 0 || gr >= Lh) {
#pragma unroll
          for (int nt = 0; nt < CH; ++nt) { zero_store4(hH, row * RSh + nt * 16 + 4 * g); if (HLO) zero_store4(hLo, row * RSh + nt * 16 + 4 * g); }
        }
      }
      if (RD) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const int row = grow0 + k * rstep_g;
          if (t0 / 2 + row >= (Lh >> 1)) { zero_store4(pH, row * RSg + gch); zero_store4(pLo, row * RSg + gch); }
        }
      }
    }
  };

  // tiles of this workgroup: a contiguous run of the (sample, tile) list (blocked; see w2s_common.h "Statistics finalisation")
  const int total = P.B * P.ntiles;
  const W2SRun wrun = w2s_block_part(total, G, blockIdx.x);   // the producers never exceed the tile count
  const int wend = wrun.first + wrun.count;
  int b = wrun.first / P.ntiles, tile = wrun.first - b * P.ntiles;   // one division per launch; the run is contiguous
  using PT0 = std::integral_constant<int, 0>; using PT1 = std::integral_constant<int, 1>;
  if (wrun.count > 0) { prefetch(b, tile, PT0{}); prefetch(b, tile, PT1{}); }
#ifdef W2S_WIDE_STAMP   // diagnostic build only (tools/altlib.sh; W2S_STAMP=1 tools/kbench.py): cycles of workgroup 0's first wave per phase -> part[0..7]
  unsigned long long sA_ = 0, sW_ = 0, sC_ = 0, sB_ = 0, sD_ = 0, sG_ = 0, k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0, k6 = 0;
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ct0 = __builtin_amdgcn_s_memtime();   // (100 MHz / shader clock)
#endif
  for (int tl = wrun.first; tl < wend; ++tl) {
    const int t0 = tile * TS;
    const int tile_n = tile + 1 == P.ntiles ? 0 : tile + 1, b_n = tile + 1 == P.ntiles ? b + 1 : b;   // the next tile of the run
#ifdef W2S_WIDE_STAMP
    k0 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
#ifdef W2S_WIDE_STAMP
    k1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    k2 = __builtin_amdgcn_s_memtime();
#endif
    commit(b, tile);
    constexpr bool LATEH = !UP2;   // stride 1: -3 %; stride 2 (half the gradient-side registers in flight): +2 % -- measured per form
    f32x4 q3[RD ? MT * CH : 1];  // this tile's y3 fragments (the prefetch below reloads rq for the next tile)
    if (RD) {
#pragma unroll
      for (int i = 0; i < MT * CH; ++i) q3[i] = rq[i];
    }
#ifdef W2S_WIDE_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    k3 = __builtin_amdgcn_s_memtime();
#endif
    if (tl + 1 < wend) { prefetch(b_n, tile_n, PT0{}); if (!LATEH) prefetch(b_n, tile_n, PT1{}); }
    __syncthreads();
#ifdef W2S_WIDE_STAMP
    k4 = __builtin_amdgcn_s_memtime();
#endif

    // ---- data gradient
    f32x4 acc[MT][CH];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < CH; ++nt) acc[mt][nt] = (f32x4){0, 0, 0, 0};
    if constexpr (GC == 32) {  // K = the 32 gradient channels of one tap
      auto mma_tap = [&](int jw, int rowoff, int mtmask) {
        bf16x8 ah[CH], al[CH];
#pragma unroll
        for (int nt = 0; nt < CH; ++nt) {
          ah[nt] = *reinterpret_cast<const bf16x8*>(wH + (nt * 16 + r) * WROW + jw * GC + 8 * g);
          al[nt] = *reinterpret_cast<const bf16x8*>(wLo + (nt * 16 + r) * WROW + jw * GC + 8 * g);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (!(mtmask & (1 << mt))) continue;
          const int row = UP2 ? wave * (8 * MT) + (mt >> 1) * 16 + r + rowoff : wave * (16 * MT) + mt * 16 + r + rowoff;
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(gyH + row * RSg + 8 * g);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(gyLo + row * RSg + 8 * g);
#pragma unroll
          for (int nt = 0; nt < CH; ++nt) acc[mt][nt] = mfma_bf3(ah[nt], al[nt], bh, bl, acc[mt][nt]);
        }
      };
      if (UP2) {
        constexpr int EVEN = 0x55 & ((1 << MT) - 1), ODD = 0xAA & ((1 << MT) - 1);
        if (PL == 2) {          // causal (forward taps at 2u + j - 2): the parities swap roles
          mma_tap(2, 0, EVEN);  // t' = 2u   : W_2^T g[u]
          mma_tap(0, 1, EVEN);  //            + W_0^T g[u+1]
          mma_tap(1, 1, ODD);   // t' = 2u+1 : W_1^T g[u+1]
        } else {
          mma_tap(1, 0, EVEN);
          mma_tap(2, 0, ODD);
          mma_tap(0, 1, ODD);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) mma_tap(j, 2 - j, (1 << MT) - 1);
        if (RD) {  // + Wd^T gpre[t'/2] at even t' (odd rows read the zero row)
          bf16x8 ah[CH], al[CH];
#pragma unroll
          for (int nt = 0; nt < CH; ++nt) {
            ah[nt] = *reinterpret_cast<const bf16x8*>(wH + (nt * 16 + r) * WROW + 96 + 8 * g);
            al[nt] = *reinterpret_cast<const bf16x8*>(wLo + (nt * 16 + r) * WROW + 96 + 8 * g);
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int m = wave * (16 * MT) + mt * 16 + r;
            const int prow = (m & 1) ? TM / 2 : (m >> 1);
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(pH + prow * RSg + 8 * g);
            const bf16x8 bl = *reinterpret_cast<const bf16x8*>(pLo + prow * RSg + 8 * g);
#pragma unroll
            for (int nt = 0; nt < CH; ++nt) acc[mt][nt] = mfma_bf3(ah[nt], al[nt], bh, bl, acc[mt][nt]);
          }
        }
      }
    } else {  // 16 gradient channels: lane groups g = 0,1 carry the first tap of a K step, g = 2,3 the second
      bf16x8 ah[2][CH], al[2][CH];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nt = 0; nt < CH; ++nt) {
          ah[ks][nt] = *reinterpret_cast<const bf16x8*>(wH + (nt * 16 + r) * WROW + ks * 32 + 8 * g);
          al[ks][nt] = *reinterpret_cast<const bf16x8*>(wLo + (nt * 16 + r) * WROW + ks * 32 + 8 * g);
        }
      const int col = 8 * (g & 1), second = g >> 1;
      const bool cz = UP2 && PL == 2;   // causal stride 2: even outputs take [tap2 (row m) | tap0 (row m+1)], odd outputs [tap1 (row m+1) | 0]
      if (cz) {
#pragma unroll
        for (int nt = 0; nt < CH; ++nt) {
          const bf16x8 th = ah[0][nt], tl2 = al[0][nt];
          ah[0][nt] = ah[1][nt]; al[0][nt] = al[1][nt];
          ah[1][nt] = th; al[1][nt] = tl2;
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (UP2) {
          const int m = wave * (8 * MT) + (mt >> 1) * 16 + r;
          const int ks = mt & 1;                         // even outputs: [tap1 | 0]; odd outputs: [tap2 (row m) | tap0 (row m+1)]  (causal: swapped above)
          const int row = ((ks != 0) != cz) ? m + second : m + (cz ? 1 : 0);
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(gyH + row * RSg + col);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(gyLo + row * RSg + col);
#pragma unroll
          for (int nt = 0; nt < CH; ++nt) acc[mt][nt] = mfma_bf3(ah[ks][nt], al[ks][nt], bh, bl, acc[mt][nt]);
        } else {
          const int m = wave * (16 * MT) + mt * 16 + r;  // window row of gy[t'+1-j] is m + (2 - j)
          const int row0 = m + 2 - second;               // K step 0: taps 0 | 1
          const int row1 = m;                            // K step 1: tap 2 | zero weights
          const bf16x8 b0h = *reinterpret_cast<const bf16x8*>(gyH + row0 * RSg + col);
          const bf16x8 b0l = *reinterpret_cast<const bf16x8*>(gyLo + row0 * RSg + col);
          const int prow = (m & 1) ? TM / 2 : (m >> 1);  // RD: lane groups 2,3 of K step 1 carry Wd x gpre[t'/2] (zero row at odd t')
          const __bf16* s1h = (RD && second) ? pH + prow * RSg : gyH + row1 * RSg;
          const __bf16* s1l = (RD && second) ? pLo + prow * RSg : gyLo + row1 * RSg;
          const bf16x8 b1h = *reinterpret_cast<const bf16x8*>(s1h + col);
          const bf16x8 b1l = *reinterpret_cast<const bf16x8*>(s1l + col);
#pragma unroll
          for (int nt = 0; nt < CH; ++nt) {
            acc[mt][nt] = mfma_bf3(ah[0][nt], al[0][nt], b0h, b0l, acc[mt][nt]);
            acc[mt][nt] = mfma_bf3(ah[1][nt], al[1][nt], b1h, b1l, acc[mt][nt]);
          }
        }
      }
    }

    // ---- epilogue: * GELU'(n_in), statistics, store
    const int elim = min(TS, Lh - t0);   // uniform: tile-local positions below it are this tile's outputs
    f32x4 sA[CH], sB[CH];
#pragma unroll
    for (int nt = 0; nt < CH; ++nt) { sA[nt] = (f32x4){0, 0, 0, 0}; sB[nt] = (f32x4){0, 0, 0, 0}; }
    float aw[FIRST ? 12 : 1];   // FIRST + part_w1: this lane's sums of gout[pos][4g + e] * xs[pos + j - pad]  (index 3e + j)
#pragma unroll
    for (int k = 0; k < (FIRST ? 12 : 1); ++k) aw[k] = 0.f;
    const bool wdf = WDFC && P.part_wd;   // uniform
    float ad[4] = {0.f, 0.f, 0.f, 0.f}, xd[MT];   // RD + part_wd: sums of gout[pos][4g + e] * san(x0[2 pos])
    if (wdf) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int pos = t0 + wave * (16 * MT) + mt * 16 + r;
        const float xr = P.x0[(size_t)b * 2 * Lh + 2 * (size_t)min(pos, Lh - 1)];
        xd[mt] = isinf(xr) ? 0.f : xr;
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int pos = UP2 ? t0 + 2 * (wave * (8 * MT) + (mt >> 1) * 16 + r) + (mt & 1) : t0 + wave * (16 * MT) + mt * 16 + r;
      if (pos - t0 >= elim) continue;   // (the last two positions of the MFMA tiles belong to the next tile; positions past the sample)
#pragma unroll
      for (int nt = 0; nt < CH; ++nt) {
        const int ch = nt * 16 + 4 * g;
        const f32x4 n = nR[mt * CH + nt];
        f32x4 v = acc[mt][nt];
        if (P.add_even && !(pos & 1)) v += ld4o(P.add_even + (size_t)b * (Lh >> 1) * HC, (unsigned)(pos >> 1) * HC + ch);
        v = v * gR[mt * CH + nt];
        if (RD && P.y3p) {  // statistics of the previous block's conv3 backward: gn = gout * GELU'(n3), n3 = IN(y3)
          const float* st = P.st3p + ((size_t)b * HC + ch) * 2;
          const f32x4 s01 = ld4(st), s23 = ld4(st + 4);
          const f32x4 m3 = {s01.x, s01.z, s23.x, s23.z}, r3 = {s01.y, s01.w, s23.y, s23.w};
          const f32x4 n3 = (q3[mt * CH + nt] - m3) * r3;
          const f32x4 gn = v * gelu_grad4(n3);
          sA[nt] += gn;
          sB[nt] += gn * n3;
        } else {
          sA[nt] += v;
          sB[nt] += v * n;
        }
        if constexpr (FIRST) {
          if (P.part_w1) {   // uniform.  xsL[i] <-> position t0 - 2 pad + i: conv1's tap j of position pos reads xsL[pos - t0 + pad + j]
            const int xi = pos - t0 + PL;
            const float x0 = xsL[xi], x1 = xsL[xi + 1], x2 = xsL[xi + 2];
            const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              aw[3 * e] = fmaf(ve[e], x0, aw[3 * e]);
              aw[3 * e + 1] = fmaf(ve[e], x1, aw[3 * e + 1]);
              aw[3 * e + 2] = fmaf(ve[e], x2, aw[3 * e + 2]);
            }
          }
        }
        if (wdf) {   // (element by element on purpose: see wav2sleep_amd/isa_audit.py)
          const float vd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) ad[e] = fmaf(vd[e], xd[mt], ad[e]);
        }
        if (FIRST && !P.gout) continue;   // uniform: the folded first-layer weight gradient was this tensor's only reader
        if constexpr (OH) {
          amax = amax4(amax, v);
          st4h(reinterpret_cast<char*>(P.gout) + (size_t)b * Lh * HC * 2, (unsigned)pos * HC + ch, f2h4(v * s_out));
        } else {
          st4o(P.gout + (size_t)b * Lh * HC, (unsigned)pos * HC + ch, v);
        }
      }
    }
    if constexpr (FIRST) {
      if (P.part_w1) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
          const float sk = row16_sum(aw[k]);
          if (r == 0) redA[(wave * 4 + g) * 12 + k] = sk;
        }
      }
    }
    if (wdf) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float se = row16_sum(ad[e]);
        if (r == 0) redD[(wave * 4 + g) * 4 + e] = se;
      }
      if (!P.part) __syncthreads();
    }
    if (wdf && tid >= 112 && tid < 128 && !P.part) {
      const int o = tid - 112;
      float sd = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) sd += redD[(w * 4 + (o >> 2)) * 4 + (o & 3)];
      accD[o] += sd;
    }
    if (P.part) {
#pragma unroll
      for (int nt = 0; nt < CH; ++nt) {
        f32x4 x1, x2;
        x1 = sA[nt]; x2 = sB[nt];
        row16_sum8(x1, x2);
        if (r == 0) {
          float* d = red + ((wave * CH + nt) * 4 + g) * 8;
          st4(d, x1);
          st4(d + 4, x2);
        }
      }
      __syncthreads();
      if (tid < CH * 32) {
        const int k = tid / HC, c = tid % HC;
        const int nt = c >> 4, gg = (c >> 2) & 3, e = c & 3;
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[((w * CH + nt) * 4 + gg) * 8 + k * 4 + e];
        w2s_part_store(&P.part[(((size_t)b * P.ntiles + tile) * 2 + k) * HC + c], s);
      }
      if (wdf && tid >= 112 && tid < 128) {
        const int o = tid - 112;
        float sd = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sd += redD[(w * 4 + (o >> 2)) * 4 + (o & 3)];
        accD[o] += sd;
      }
      if constexpr (FIRST) {
        if (P.part_w1 && tid >= 64 && tid < 112) {   // (o, j) = ((tid - 64) / 3, (tid - 64) % 3); channel o sits in lane group o >> 2, slot o & 3
          const int idx = tid - 64, o = idx / 3, j = idx % 3;
          float s = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) s += redA[(w * 4 + (o >> 2)) * 12 + (o & 3) * 3 + j];
          P.part_w1[((size_t)b * P.ntiles + tile) * 48 + idx] = s;
        }
      }
    }

    // ---- weight gradient: k-step = 32 gradient-side positions; lane group g covers positions 8g..8g+7 of the step
#ifdef W2S_WIDE_STAMP
    k5 = __builtin_amdgcn_s_memtime();
#endif
    if (LATEH && tl + 1 < wend) prefetch(b_n, tile_n, PT1{});
    constexpr int KS = (UP2 ? TM / 2 : TM) / 32;
#pragma unroll
    for (int s0 = 0; s0 < KS; s0 += KW) {
      const int s = s0 + wk;
      if (KS % KW != 0 && s >= KS) break;         // fewer k-steps than sharing waves (128-position stride-2 tiles: 2 steps, 4 waves)
      const int p0 = 32 * s + 8 * g + q4;         // this lane's address row (gradient-side position) of the first 4-block
      const int gr0 = UP2 ? p0 : p0 + 2 - PL;     // window row of gradient-side position p0
      const int gcol = wi * 16 + 4 * p4;
      bf16x8 ah = lds_tr8(gyH + gr0 * RSg + gcol, gyH + (gr0 + 4) * RSg + gcol);
      bf16x8 al = lds_tr8(gyLo + gr0 * RSg + gcol, gyLo + (gr0 + 4) * RSg + gcol);
      if (s == KS - 1 && g == 3) {   // element e of the fragment is gradient-side position 32 s + 8 g + e: those past the tile's TS (TS / 2) are the next tile's
        ah[7] = (__bf16)0.f; al[7] = (__bf16)0.f;
        if (!UP2) { ah[6] = (__bf16)0.f; al[6] = (__bf16)0.f; }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int h0 = UP2 ? 2 * p0 + j : p0 + j, h1 = UP2 ? 2 * (p0 + 4) + j : p0 + 4 + j;
        const int hcol = wc * 16 + 4 * p4;
        const bf16x8 bh = lds_tr8(hH + h0 * RSh + hcol, hH + h1 * RSh + hcol);
        if (HLO) {
          const bf16x8 bl = lds_tr8(hLo + h0 * RSh + hcol, hLo + h1 * RSh + hcol);
          accw[j] = mfma_bf3(ah, al, bh, bl, accw[j]);
        } else {
          accw[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, accw[j], 0, 0, 0);
          accw[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, accw[j], 0, 0, 0);
        }
      }
    }
    if (RD) {  // dWd[o][c] += sum_u gpre[u][o] * h[2u][c]   (window row of position t0 + 2u is 2u + pad)
      constexpr int KSD = (TM / 2) / 32;
#pragma unroll
      for (int s0 = 0; s0 < KSD; s0 += KW) {
        const int s = s0 + wk;
        if (KSD % KW != 0 && s >= KSD) break;
        const int p0 = 32 * s + 8 * g + q4;
        const int gcol = wi * 16 + 4 * p4, hcol = wc * 16 + 4 * p4;
        bf16x8 ah = lds_tr8(pH + p0 * RSg + gcol, pH + (p0 + 4) * RSg + gcol);
        bf16x8 al = lds_tr8(pLo + p0 * RSg + gcol, pLo + (p0 + 4) * RSg + gcol);
        if (s == KSD - 1 && g == 3) { ah[7] = (__bf16)0.f; al[7] = (__bf16)0.f; }   // gpre row TM/2 - 1 = position t0 + TS: the next tile's
        const bf16x8 bh = lds_tr8(hH + (2 * p0 + PL) * RSh + hcol, hH + (2 * (p0 + 4) + PL) * RSh + hcol);
        if (HLO) {
          const bf16x8 bl = lds_tr8(hLo + (2 * p0 + PL) * RSh + hcol, hLo + (2 * (p0 + 4) + PL) * RSh + hcol);
          accd = mfma_bf3(ah, al, bh, bl, accd);
        } else {
          accd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, accd, 0, 0, 0);
          accd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, accd, 0, 0, 0);
        }
      }
    }
#ifdef W2S_WIDE_STAMP
    asm volatile("" :: "v"(accw[0]), "v"(accw[2]));
    k6 = __builtin_amdgcn_s_memtime();
    sA_ += k1 - k0; sW_ += k2 - k1; sC_ += k3 - k2; sB_ += k4 - k3; sD_ += k5 - k4; sG_ += k6 - k5;
#endif
    b = b_n; tile = tile_n;
  }
#ifdef W2S_WIDE_STAMP
  if (blockIdx.x == 0 && (tid & 63) == 0 && P.part) {   // every wave: its commit and its wait at the second barrier
    P.part[8 + 2 * (tid >> 6)] = (float)sC_; P.part[9 + 2 * (tid >> 6)] = (float)sB_;
  }
  if (blockIdx.x == 0 && tid == 0 && P.part) {
    P.part[0] = (float)sA_; P.part[1] = (float)sW_; P.part[2] = (float)sC_; P.part[3] = (float)sB_; P.part[4] = (float)sD_; P.part[5] = (float)sG_;
    P.part[6] = (float)wrun.count;
    P.part[7] = (float)(__builtin_amdgcn_s_memtime() - ct0) / (float)(__builtin_amdgcn_s_memrealtime() - rt0) * 100.f;   // in-kernel clock, MHz
  }
#endif

  if (OH) w2s_amax_commit(P.hdr_o, amax, s_out);
  if (WDFC && P.part_wd && tid >= 112 && tid < 128) P.part_wd[(size_t)blockIdx.x * 16 + (tid - 112)] = accD[tid - 112];
  // ---- one slab per workgroup, raw-fragment layout [tile(i,j,c)][lane][4]; waves sharing a tile pair sum in wave order
  float* out = P.slab + (size_t)blockIdx.x * (CG * 3 * CH) * 256;
  float* outd = RD ? P.slab_d + (size_t)blockIdx.x * (CG * CH) * 256 : nullptr;
  if (KW == 1) {
#pragma unroll
    for (int j = 0; j < 3; ++j) st4(out + ((wi * 3 + j) * CH + wc) * 256 + lane * 4, accw[j]);
    if (RD) st4(outd + (wi * CH + wc) * 256 + lane * 4, accd);
  } else {
    __syncthreads();
    float* sc = reinterpret_cast<float*>(smem4);  // the windows are free now: [wave][4][64][4]
    if (wk != 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) st4(sc + ((wave * 4 + j) * 64 + lane) * 4, accw[j]);
      if (RD) st4(sc + ((wave * 4 + 3) * 64 + lane) * 4, accd);
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        f32x4 v = accw[j];
#pragma unroll
        for (int k = 1; k < KW; ++k) v += ld4(sc + (((wave + k) * 4 + j) * 64 + lane) * 4);
        st4(out + ((wi * 3 + j) * CH + wc) * 256 + lane * 4, v);
      }
      if (RD) {
        f32x4 v = accd;
#pragma unroll
        for (int k = 1; k < KW; ++k) v += ld4(sc + (((wave + k) * 4 + 3) * 64 + lane) * 4);
        st4(outd + (wi * CH + wc) * 256 + lane * 4, v);
      }
    }
  }
}

template <int CG, int CH, int MT, int UP2, int RD, int FIRST = 0, int GM = 0>
static int launch_bwd_bf(const BwdP& P0, int nslab, hipStream_t s) {
  constexpr int TM = 64 * MT, TS = TM - 2, GC = CG * 16, HC = CH * 16, KD = (GC == 32) ? (RD ? 128 : 96) : 64;
  BwdP P = P0;
  P.ntiles = (P.Lh + TS - 1) / TS;
  constexpr int NRg = UP2 ? TM / 2 : TM, NRh = TM, NRp = RD ? TM / 2 + 1 : 0;
  size_t lds = (size_t)bwd_redn(CH) * 4 + (size_t)2 * (2 * (NRg + NRp) * bf_rs(GC) + (W2S_BF_HLO ? 2 : 1) * NRh * bf_rsh(HC, UP2) + 2 * HC * (KD + 8));
  if (FIRST) lds += (size_t)(TM + 4) * 4 + 4 * 4 * 12 * 4;
  if (RD) lds += (4 * 4 * 4 + 16) * 4;
  if (lds < 4 * 4 * 64 * 4 * 4) lds = 4 * 4 * 64 * 4 * 4;  // end-of-kernel scratch [wave][4][64][4]
  lds = (lds + 15) & ~(size_t)15;
  auto kern = bwd_fused_bf_kernel<CG, CH, MT, UP2, RD, FIRST, GM>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return W2S_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(nslab), dim3(256), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

#ifndef W2S_BF_MT22
#define W2S_BF_MT22 2
#endif
#ifndef W2S_BF_MT11
#define W2S_BF_MT11 4
#endif
#ifndef W2S_BF_PF
#define W2S_BF_PF 1
#endif
// 64-position sub-tiles per workgroup tile.  16 -> 16: 4 (256 positions; the residual-fold form 2: register budget, see bfk_occ);
// 32 -> 32 stride 1: W2S_BF_MT22; everything else 2 (the stride-2 form needs an even count: even / odd outputs are separate M tiles).
constexpr int bf_mt(int cg, int ch, int up2, int rd) {
  return (cg == 16 && ch == 16) ? (rd ? 2 : W2S_BF_MT11) : (cg == 32 && ch == 32 && !up2) ? W2S_BF_MT22 : 2;
}
// (cg, ch) pairs whose conv1 kernel can fold the residual branch (LDS budget: two workgroups per CU)
extern "C" int w2s_bwd_fused_folds_residual(int cg, int ch) { return (cg == 16 && ch == 16) || (cg == 32 && ch == 16); }
// positions of the input side per tile (= rows of `part` per sample: ceil(Lh / tile)); rd: the residual-fold form (gpre != NULL)
// split_precision: the split-precision kernels' tiles are two positions short of the MFMA tile (whole staging passes, see the kernel)
extern "C" int w2s_bwd_fused_tile(int cg, int ch, int stride, int rd, int split_precision) {
  return 64 * bf_mt(cg, ch, stride == 2, rd) - (split_precision ? 2 : 0);
}

// cg = channels of the gradient side (the forward conv's cout), ch = channels of the input side (its cin).
static int bwd_fused_impl(const void* gv, const float* y, const float* st_k, const float* bst_k, int pro, const float* xin,
                          const float* st_in, const float* add_even, const float* wb, void* goutv, float* part, float* slab, int nslab,
                          int B, int Lg, int Lh, int cg, int ch, int stride, int pad, int split_precision, const void* gprev, const float* wd,
                          float* slab_d, const float* w1, const float* y3p, const float* st3p,
                          int gmode, const float* hdr_g, const float* hdr_p, float* hdr_o, void* stream, float* part_w1 = nullptr,
                          const float* x0 = nullptr, float* part_wd = nullptr) {
  const float* g = static_cast<const float*>(gv);
  const float* gpre = static_cast<const float*>(gprev);
  float* gout = static_cast<float*>(goutv);
  if (!g || !y || !st_k || !bst_k || !xin || !wb || (!gout && !part_w1) || !slab || nslab <= 0) return W2S_EINVAL;
  if (part_w1 && (!w1 || !part || gmode)) return W2S_EINVAL;   // the fold lives in the first-layer recompute form (fp32 chain)
  if ((size_t)Lh * 32 * 4 >= ((size_t)1 << 32)) return W2S_EINVAL;  // 32-bit lane offsets inside one sample
  if (pro != W2S_PRO_INBWD && pro != W2S_PRO_INBWD_GP) return W2S_EINVAL;
  if (!((stride == 1 && Lg == Lh) || (stride == 2 && 2 * Lg == Lh))) return W2S_EINVAL;
  if (pro != (stride == 2 ? W2S_PRO_INBWD_GP : W2S_PRO_INBWD)) return W2S_EINVAL;  // the kernels bake the mode in
  if (pad != 1 && !(pad == 2 && split_precision)) return W2S_EINVAL;                // causal padding: split-precision kernels only
  BwdP P{g, y, st_k, bst_k, xin, st_in, add_even, wb, gout, part, slab, B, Lg, Lh, 0, pro, pad, gpre, wd, slab_d, w1, y3p, st3p,
         gmode, hdr_g, hdr_p, hdr_o, part_w1, x0, part_wd};
  if ((x0 != nullptr) != (part_wd != nullptr) || (part_wd && (!gpre || cg != 16 || ch != 16 || gmode))) return W2S_EINVAL;
  if (y3p && (!gpre || !st3p || !part)) return W2S_EINVAL;
  const bool rd = gpre != nullptr;
  if (rd && (!wd || !slab_d || add_even || stride != 1 || !split_precision || !w2s_bwd_fused_folds_residual(cg, ch) || (Lh & 1))) return W2S_EINVAL;
  if (gmode < 0 || gmode > 2 || (gmode && (!split_precision || !hdr_g || !hdr_o || (rd && (gmode != 2 || !hdr_p))))) return W2S_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int up2 = stride == 2;
  if (w1) {  // xin is the raw signal: conv2 of block 0
    if (rd || !st_in || stride != 1 || !split_precision || cg != 16 || ch != 16 || gmode == 1) return W2S_EINVAL;
    return gmode ? launch_bwd_bf<1, 1, bf_mt(16, 16, 0, 0), 0, 0, 1, 2>(P, nslab, s) : launch_bwd_bf<1, 1, bf_mt(16, 16, 0, 0), 0, 0, 1>(P, nslab, s);
  }
  if (rd && cg == 16 && ch == 16)
    return gmode ? launch_bwd_bf<1, 1, bf_mt(16, 16, 0, 1), 0, 1, 0, 2>(P, nslab, s) : launch_bwd_bf<1, 1, bf_mt(16, 16, 0, 1), 0, 1>(P, nslab, s);
  if (rd && cg == 32 && ch == 16)
    return gmode ? launch_bwd_bf<2, 1, bf_mt(32, 16, 0, 1), 0, 1, 0, 2>(P, nslab, s) : launch_bwd_bf<2, 1, bf_mt(32, 16, 0, 1), 0, 1>(P, nslab, s);
  // fp16 chain: the shapes the engine's chain reaches -- (16,16) and (32,32), both strides, fp16 in; (32,32) / (16,16) stride 2 also with
  // an fp32 gradient in (the chain's entry: conv3 of the topmost <= 32-channel block)
#define W2S_BFH(CG_, CH_, UP_, GM_) \
  if (gmode == GM_ && cg == 16 * CG_ && ch == 16 * CH_ && up2 == UP_) \
    return launch_bwd_bf<CG_, CH_, bf_mt(16 * CG_, 16 * CH_, UP_, 0), UP_, 0, 0, GM_>(P, nslab, s);
  W2S_BFH(1, 1, 0, 2) W2S_BFH(1, 1, 1, 2) W2S_BFH(2, 2, 0, 2) W2S_BFH(2, 2, 1, 2) W2S_BFH(2, 2, 1, 1) W2S_BFH(1, 1, 1, 1)
#undef W2S_BFH
  if (gmode) return W2S_EINVAL;
#define W2S_BFS(CG_, CH_) \
  if (split_precision && cg == 16 * CG_ && ch == 16 * CH_) \
    return up2 ? launch_bwd_bf<CG_, CH_, bf_mt(16 * CG_, 16 * CH_, 1, 0), 1, 0>(P, nslab, s) : launch_bwd_bf<CG_, CH_, bf_mt(16 * CG_, 16 * CH_, 0, 0), 0, 0>(P, nslab, s);
  W2S_BFS(1, 1) W2S_BFS(2, 1) W2S_BFS(2, 2)
#undef W2S_BFS
#define W2S_BF(CG_, CH_) \
  if (cg == 16 * CG_ && ch == 16 * CH_) \
    return up2 ? launch_bwd<CG_, CH_, bf_mt(16 * CG_, 16 * CH_, 1, 0), 1, W2S_BF_PF>(P, nslab, s) : launch_bwd<CG_, CH_, bf_mt(16 * CG_, 16 * CH_, 0, 0), 0, W2S_BF_PF>(P, nslab, s);
  W2S_BF(1, 1) W2S_BF(2, 1) W2S_BF(2, 2)
#undef W2S_BF
  return W2S_EINVAL;
}

extern "C" int w2s_bwd_fused(const float* g, const float* y, const float* st_k, const float* bst_k, int pro, const float* xin,
                             const float* st_in, const float* add_even, const float* wb, float* gout, float* part, float* slab, int nslab,
                             int B, int Lg, int Lh, int cg, int ch, int stride, int pad, int split_precision, const float* gpre, const float* wd,
                             float* slab_d, const float* w1, const float* y3p, const float* st3p, void* stream) {
  return bwd_fused_impl(g, y, st_k, bst_k, pro, xin, st_in, add_even, wb, gout, part, slab, nslab, B, Lg, Lh, cg, ch, stride, pad, split_precision, gpre,
                        wd, slab_d, w1, y3p, st3p, 0, nullptr, nullptr, nullptr, stream);
}
// conv2 of block 0 (first-layer recompute form: w1 != NULL) with the first layer's weight gradient folded in: per-tile partial sums into
// part_w1 [B][ceil(L / tile)][16][3]; gout may be NULL (nothing else reads it).  w2s_enc_first_wgrad finishes the job.
extern "C" int w2s_bwd_fused_w1(const float* g, const float* y, const float* st_k, const float* bst_k, const float* x, const float* st_in,
                                const float* wb, float* gout, float* part, float* part_w1, float* slab, int nslab, int B, int L, int pad,
                                const float* w1, void* stream) {
  if (!part_w1) return W2S_EINVAL;
  return bwd_fused_impl(g, y, st_k, bst_k, W2S_PRO_INBWD, x, st_in, nullptr, wb, gout, part, slab, nslab, B, L, L, 16, 16, 1, pad, 1, nullptr, nullptr,
                        nullptr, w1, nullptr, nullptr, 0, nullptr, nullptr, nullptr, stream, part_w1);
}
// conv1 of block 1 (residual-fold form, 16 -> 16) with block 0's downsample weight gradient folded in: this kernel's gout is block 0's
// gpre, so part_wd [nslab][16] = per-workgroup sums of gout[u][o] * san(x0[2u]) (x0 = the raw signal [B][2 Lh]) replaces a pass over
// that tensor.  Sum the rows with w2s_colsum_batch.
extern "C" int w2s_bwd_fused_wd(const float* g, const float* y, const float* st_k, const float* bst_k, const float* xin, const float* wb, float* gout,
                                float* part, float* slab, int nslab, int B, int L, int pad, const float* gpre, const float* wd, float* slab_d,
                                const float* y3p, const float* st3p, const float* x0, float* part_wd, void* stream) {
  if (!x0 || !part_wd) return W2S_EINVAL;
  return bwd_fused_impl(g, y, st_k, bst_k, W2S_PRO_INBWD, xin, nullptr, nullptr, wb, gout, part, slab, nslab, B, L, L, 16, 16, 1, pad, 1, gpre, wd, slab_d,
                        nullptr, y3p, st3p, 0, nullptr, nullptr, nullptr, stream, nullptr, x0, part_wd);
}
// the same launch with the gradient chain stored as fp16 (include/w2s.h, "fp16 gradient chain")
extern "C" int w2s_bwd_fused_h(const void* g, const float* y, const float* st_k, const float* bst_k, int pro, const float* xin,
                               const float* st_in, const float* add_even, const float* wb, void* gout, float* part, float* slab, int nslab,
                               int B, int Lg, int Lh, int cg, int ch, int stride, int pad, const void* gpre, const float* wd, float* slab_d,
                               const float* w1, const float* y3p, const float* st3p, int gmode, const float* hdr_g, const float* hdr_p,
                               float* hdr_o, void* stream) {
  return bwd_fused_impl(g, y, st_k, bst_k, pro, xin, st_in, add_even, wb, gout, part, slab, nslab, B, Lg, Lh, cg, ch, stride, pad, 1, gpre, wd, slab_d, w1,
                        y3p, st3p, gmode, hdr_g, hdr_p, hdr_o, stream);
}
