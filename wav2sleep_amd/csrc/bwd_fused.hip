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
#include "conv_cl.inl"

struct BwdP {
  const float* g; const float* y; const float* st_k; const float* bst_k;
  const float* xin; const float* st_in; const float* add_even; const float* wb;
  float* gout; float* part; float* slab;
  int B, Lg, Lh, ntiles, pro;
};

template <int CG, int CH, int MT, int UP2, int PF>
__global__ __launch_bounds__(256) void bwd_fused_kernel(BwdP P) {
  extern __shared__ f32x4 smem4[];
  float* smem = reinterpret_cast<float*>(smem4);
  constexpr int TM = 64 * MT;                       // output (h-side) positions per tile
  constexpr int GC = CG * 16, HC = CH * 16;         // channels on the gradient / input side
  constexpr int RSg = GC + 4, RSh = HC + 4;
  constexpr int NRg = UP2 ? TM / 2 + 1 : TM + 2;    // gy window rows
  constexpr int NRh = UP2 ? TM + 1 : TM + 2;        // h window rows (row 0 = position t0-1)
  float* gyL = smem;
  float* hL = smem + NRg * RSg;
  float* nL = hL + NRh * RSh;                       // normalised input n_in of the TM centre rows (epilogue: GELU'(n), stats)
  float* red = nL + TM * RSh;                       // [4][CH][4][8] stats scratch, later [4][64][4] slab reduce
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int Lg = P.Lg, Lh = P.Lh;

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
  float* wL = red + 1024;
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
  auto prefetch = [&](int tl) {
    const int b = tl / P.ntiles, t0 = (tl % P.ntiles) * TM;
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
  auto commit = [&](int tl) {  // transform the prefetched registers and write both windows to LDS
    const int b = tl / P.ntiles, t0 = (tl % P.ntiles) * TM;
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

  const int total = P.B * P.ntiles;
  if (PF && (int)blockIdx.x < total) prefetch(blockIdx.x);
  for (int tl = blockIdx.x; tl < total; tl += gridDim.x) {
    const int b = tl / P.ntiles, tile = tl % P.ntiles;
    const int t0 = tile * TM;
    __syncthreads();  // everyone is done reading the previous tile's windows (and wL is written)
    if (!PF) prefetch(tl);
    commit(tl);
    if (PF && tl + (int)gridDim.x < total) prefetch(tl + gridDim.x);
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
        x1.x = row16_sum(sA[nt].x); x1.y = row16_sum(sA[nt].y); x1.z = row16_sum(sA[nt].z); x1.w = row16_sum(sA[nt].w);
        x2.x = row16_sum(sB[nt].x); x2.y = row16_sum(sB[nt].y); x2.z = row16_sum(sB[nt].z); x2.w = row16_sum(sB[nt].w);
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
        P.part[(((size_t)b * P.ntiles + tile) * 2 + k) * HC + c] = s;
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
        __syncthreads();
        st4(red + (wave * 64 + lane) * 4, accw[i][j][c]);
        __syncthreads();
        if (wave == 0) {
          f32x4 v = ld4(red + lane * 4) + ld4(red + (64 + lane) * 4) + ld4(red + (128 + lane) * 4) + ld4(red + (192 + lane) * 4);
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
  size_t lds = (size_t)(NRg * (CG * 16 + 4) + NRh * (CH * 16 + 4)) * 4;
  size_t redb = (size_t)((4 * CH * 4 * 8 > 1024) ? 4 * CH * 4 * 8 : 1024) * 4;
  lds += redb + (size_t)(CH * 16) * (3 * CG * 16 + 4) * 4 + (size_t)TM * (CH * 16 + 4) * 4;
  auto kern = bwd_fused_kernel<CG, CH, MT, UP2, PF>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return W2S_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(nslab), dim3(256), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

#ifndef W2S_BF_MT11
#define W2S_BF_MT11 4
#endif
#ifndef W2S_BF_MT2
#define W2S_BF_MT2 2
#endif
#ifndef W2S_BF_PF
#define W2S_BF_PF 1
#endif
extern "C" int w2s_bwd_fused_tile(int cg, int ch) { return 64 * ((cg == 16 && ch == 16) ? W2S_BF_MT11 : W2S_BF_MT2); }

// cg = channels of the gradient side (the forward conv's cout), ch = channels of the input side (its cin).
extern "C" int w2s_bwd_fused(const float* g, const float* y, const float* st_k, const float* bst_k, int pro, const float* xin,
                             const float* st_in, const float* add_even, const float* wb, float* gout, float* part, float* slab, int nslab,
                             int B, int Lg, int Lh, int cg, int ch, int stride, void* stream) {
  if (!g || !y || !st_k || !bst_k || !xin || !wb || !gout || !slab || nslab <= 0) return W2S_EINVAL;
  if (pro != W2S_PRO_INBWD && pro != W2S_PRO_INBWD_GP) return W2S_EINVAL;
  if (!((stride == 1 && Lg == Lh) || (stride == 2 && 2 * Lg == Lh))) return W2S_EINVAL;
  if (pro != (stride == 2 ? W2S_PRO_INBWD_GP : W2S_PRO_INBWD)) return W2S_EINVAL;  // the kernels bake the mode in
  BwdP P{g, y, st_k, bst_k, xin, st_in, add_even, wb, gout, part, slab, B, Lg, Lh, 0, pro};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int up2 = stride == 2;
#define W2S_BF(CG_, CH_, MT_) \
  if (cg == 16 * CG_ && ch == 16 * CH_) return up2 ? launch_bwd<CG_, CH_, MT_, 1, W2S_BF_PF>(P, nslab, s) : launch_bwd<CG_, CH_, MT_, 0, W2S_BF_PF>(P, nslab, s);
  W2S_BF(1, 1, W2S_BF_MT11) W2S_BF(2, 1, W2S_BF_MT2) W2S_BF(2, 2, W2S_BF_MT2)
#undef W2S_BF
  return W2S_EINVAL;
}
