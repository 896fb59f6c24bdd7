// Forward k=3 encoder conv for the HBM-heavy <= 32-channel layers as a PERSISTENT kernel (split precision):
//
//   y[b,t,o] = sum_{j,c} W[o][c][j] * h[b, t*stride + j - pad, c],    h = GELU(x) | GELU(IN(x)) | GELU(IN(conv1(signal)))
//   pad = 1: symmetric padding; pad = 2: causal padding (blocks.py:150-152,178-182: left pad k-1, the right trim is not computed)
//   part[b][tile][2][CO] = per-tile sums of y and y^2   (instance-norm statistics of the NEXT layer, blocks.py:173-186)
//
// Same contract as conv_cl_kernel with EPI_STATS, different execution shape: every workgroup owns a contiguous run of the (sample, tile)
// list (blocked: w2s_block_part -- a workgroup then meets one or two samples, which is what lets it finalise the statistics itself, see
// w2s_common.h), the NEXT tile's window is prefetched into registers while the current tile runs through the matrix cores
// (conv_cl's one-tile workgroups serialise load latency, prologue arithmetic, MFMA and store drain), the weights live in
// LDS as bf16 hi/lo planes for the whole launch, and every product is 3 x v_mfma_f32_16x16x32_bf16 (16 input channels: two
// taps share one K = 32 step).  Replaces aten::convolution + native_batch_norm(statistics) of ConvLayer1D.forward.
#include <type_traits>
#include "conv_cl.inl"

#ifndef W2S_FF_DBG
#define W2S_FF_DBG 0   // tuning builds only: 1 = no on-load arithmetic, 2 = no MFMA loop, 4 = no LDS staging, 8 = no y stores
#endif
struct FwdP {
  const float* x; const float* w; const float* st_in; const float* w1;
  float* y; float* part;
  int B, L_in, L_out, ntiles, pro, pad;
};

__host__ __device__ constexpr int ff_rs(int c) { return c == 16 ? 16 : c + 8; }  // 32-B rows are conflict-free as they are
// Output positions per tile.  The window of a TM-position tile has TM*stride + 2 (stride 1) / + 1 (stride 2) rows; 256 threads stage
// 64 rows of 16 channels (32 rows of 32) per pass, so the two halo rows cost wave 0 a whole extra pass of prologue arithmetic (erf-GELU,
// split) -- +25 % on the critical path of a VALU-bound kernel, and on the SIMD that hosts wave 0 of every workgroup.  Tiles of
// TM - 2 (TM - 1) positions make the window exactly TM*stride (- 1) rows: no ragged pass; the matrix cores compute the 2 (1) spare
// rows from whatever follows the window in LDS and the epilogue drops them.
__host__ __device__ constexpr int ff_ts(int tm, int stride) { return stride == 1 ? tm - 2 : tm - 1; }
__host__ __device__ constexpr bool ff_db(int hc, int stride) { return hc == 16 && stride == 1; }

// CI / CO: input / output channel tiles (16 each); MT: 16-position m-tiles per wave (TM = 64*MT outputs per tile);
// PRO: W2S_PRO_GELU, W2S_PRO_IN_GELU or W2S_PRO_FIRST (x = raw signal, w1 = block 0's conv1 weight)
// tuning: waves per SIMD asked of the variants that otherwise land on 2 (stride 2, or 32 channels on a side: 164-192 VGPRs)
#ifndef W2S_FF_OCC2
#define W2S_FF_OCC2 1
#endif
__host__ __device__ constexpr int ffk_occ(int ci, int co, int stride) { return (ci == 1 && co == 1 && stride == 1) ? 1 : W2S_FF_OCC2; }
template <int CI, int CO, int MT, int STRIDE, int PRO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ffk_occ(CI, CO, STRIDE)))) void conv_fwd_bf_kernel(FwdP P) {
  extern __shared__ f32x4 smem4[];
  constexpr int TM = 64 * MT;                            // output rows the matrix cores compute per tile ...
  constexpr int TS = ff_ts(TM, STRIDE);                  // ... of which the first TS are the tile (the rest is discarded): see ff_ts
  constexpr int HC = CI * 16, OC = CO * 16;
  constexpr int RSh = ff_rs(HC);
  constexpr int NRh = (TS - 1) * STRIDE + 3;             // window rows; row 0 = input position t0*STRIDE - pad
  constexpr int KSP = (HC == 16) ? 2 : 3;                // K = 32 steps: [tap0|tap1] [tap2|0]  or one tap each
  constexpr int KD = KSP * 32, WROW = KD + 8;
  float* red = reinterpret_cast<float*>(smem4);          // [4][CO][4][8] statistics scratch
  // 16 input channels, stride 1: TWO window buffers (the LDS budget keeps four workgroups per CU), so a wave that is done with a tile's
  // MFMAs / stores starts transforming the next window at once instead of waiting at a barrier for the slowest wave
  constexpr bool DB = ff_db(HC, STRIDE);
  constexpr int WIN = 2 * NRh * RSh;                     // one window: hi plane, lo plane
  __bf16* hbase = reinterpret_cast<__bf16*>(red + 4 * CO * 4 * 8);
  __bf16* wH = hbase + (DB ? 2 : 1) * WIN;               // [OC][WROW]
  __bf16* wLo = wH + OC * WROW;
  float* xsL = reinterpret_cast<float*>(wLo + OC * WROW);  // FIRST: NRh + 2 signal samples
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int L_in = P.L_in, L_out = P.L_out;
  const int G = (int)gridDim.x;

  // ---- weights [OC][3][HC] (forward packing) -> LDS planes once per launch; 16 channels: k = tap*16 + c, zero tail
  for (int i = tid; i < OC * (KD / 4); i += 256) {
    const int row = i / (KD / 4), k = (i % (KD / 4)) * 4;
    f32x4 v = {0, 0, 0, 0};
    if (k < 3 * HC) v = ld4(P.w + (size_t)row * (3 * HC) + k);
    split_store4(wH, wLo, row * WROW + k, v);
  }

  constexpr int c4h = HC / 4, rstep = 256 / c4h, NH = (NRh + rstep - 1) / rstep;
  const int hc4 = tid % c4h, hrow0 = tid / c4h, hch = hc4 * 4;
  constexpr bool FIRST = PRO == W2S_PRO_FIRST;
  constexpr int NXS = FIRST ? (NRh + 2 + 255) / 256 : 1;
  // two register sets: the raw windows of the next TWO tiles are in flight while one is transformed (bytes in flight per CU, not
  // arithmetic, is what bounds these kernels).  Loads are unconditional (clamped addresses; out-of-range rows are zeroed at commit
  // time): a conditional load makes hipcc drain the whole queue (vmcnt(0)) at every wait.
  f32x4 rh[2][FIRST ? 1 : NH];
  float rxs[2][NXS], w1r[4][3];
  if (FIRST) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) w1r[i][j] = P.w1[(hch + i) * 3 + j];
  }
  const int total = P.B * P.ntiles;
  const W2SRun wrun = w2s_block_part(total, G, blockIdx.x);   // this workgroup's tiles [wfirst, wend): the producers never exceed the tile count
  const int wfirst = wrun.first, wend = wrun.first + wrun.count;
  // (sample, tile) of the body's tile and of the prefetch stream (two tiles ahead), carried incrementally: the run is contiguous, so a
  // position only ever steps to the next tile -- no integer division per tile (each was ~40 scalar + 5 vector instructions, four per tile)
  int b_cur = wfirst / P.ntiles, tile_cur = wfirst - b_cur * P.ntiles;
  int b_pf = b_cur, tile_pf = tile_cur, tl_pf = wfirst;   // the prefetch position stops at the run's last tile (later rounds reload it)
  auto prefetch = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    const int b = b_pf, t0 = tile_pf * TS;
    if (tl_pf + 1 < wend) {
      ++tl_pf;
      if (++tile_pf == P.ntiles) { tile_pf = 0; ++b_pf; }
    }
    const int rb = t0 * STRIDE - P.pad;
    if (FIRST) {  // signal samples rb-pad .. rb-pad+NRh+1 (conv1 has the same padding mode as this conv)
      const float* xs = P.x + (size_t)b * L_in;
#pragma unroll
      for (int k = 0; k < NXS; ++k) rxs[S][k] = xs[min(max(rb - P.pad + tid + 256 * k, 0), L_in - 1)];
    } else {
      const float* xb = P.x + (size_t)b * L_in * HC;
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        const int row = min(hrow0 + k * rstep, NRh - 1), gr = min(max(rb + row, 0), L_in - 1);
        rh[S][k] = ld4o(xb, (unsigned)gr * HC + hch);
      }
    }
  };
  auto commit = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    __bf16* hH = hbase + (DB ? S : 0) * WIN;
    __bf16* hLo = hH + NRh * RSh;
    const int b = b_cur, t0 = tile_cur * TS;
    const int rb = t0 * STRIDE - P.pad;
    const bool inside = rb >= 0 && rb + NRh <= L_in;   // uniform: every window row is a real position (all but a sample's first / last tile)
    f32x4 hb = {0, 0, 0, 0}, hr = {1, 1, 1, 1};   // n = x * rstd + (-mean * rstd): one fused multiply-add per element
    if (PRO != W2S_PRO_GELU) {
      const float* st = P.st_in + ((size_t)b * HC + hch) * 2;
      f32x4 s01 = ld4(st), s23 = ld4(st + 4);
      hr = (f32x4){s01.y, s01.w, s23.y, s23.w};
      hb = -((f32x4){s01.x, s01.z, s23.x, s23.z} * hr);
    }
    if (FIRST) {
#pragma unroll
      for (int k = 0; k < NXS; ++k) {
        const int i = tid + 256 * k, gr = rb - P.pad + i;
        const float xv = rxs[S][k];
        if (i < NRh + 2) xsL[i] = (gr >= 0 && gr < L_in && !isinf(xv)) ? xv : 0.f;
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const int row = hrow0 + k * rstep;
      if (row < NRh) {
        f32x4 xv;
        if (FIRST) {  // window row <-> position rb+row; xsL[i] <-> position rb-pad+i: taps at p-1,p,p+1 (pad 1) or p-2,p-1,p (pad 2)
          const float xm = xsL[row], xc = xsL[row + 1], xp = xsL[row + 2];
          xv.x = w1r[0][0] * xm + w1r[0][1] * xc + w1r[0][2] * xp;
          xv.y = w1r[1][0] * xm + w1r[1][1] * xc + w1r[1][2] * xp;
          xv.z = w1r[2][0] * xm + w1r[2][1] * xc + w1r[2][2] * xp;
          xv.w = w1r[3][0] * xm + w1r[3][1] * xc + w1r[3][2] * xp;
        } else {
          xv = rh[S][k];
        }
        const f32x4 hv = (W2S_FF_DBG & 1) ? xv + hb : gelu4(fma4(xv, hr, hb));
        if (!(W2S_FF_DBG & 4)) split_store4(hH, hLo, row * RSh + hch, hv);
      }
    }
    if (!inside) {   // (uniform: the zero padding only exists at a sample's ends) rows outside the sample were loaded from clamped addresses
                     // and transformed like the others; the lanes that stored them now overwrite them with zeros
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        const int row = hrow0 + k * rstep, gr = rb + row;
        if (row < NRh && (gr < 0 || gr >= L_in)) { zero_store4(hH, row * RSh + hch); zero_store4(hLo, row * RSh + hch); }
      }
    }
  };

  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  auto body = [&](auto SET, int tl) {
    const bool live = tl < wend;   // workgroup-uniform; a dead round only keeps the load queue regular
    const int b = b_cur, tile = tile_cur;
    const int t0 = tile * TS;
    if (!DB) __syncthreads();  // single buffer: the previous tile's LDS reads are done (two buffers: the barriers of the round in between did that)
    const __bf16* hH = hbase + (DB ? decltype(SET)::value : 0) * WIN;
    const __bf16* hLo = hH + NRh * RSh;
    if (live) commit(SET);
    prefetch(SET);
    __syncthreads();
    if (!live) return;

    f32x4 acc[MT][CO];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < CO; ++nt) acc[mt][nt] = (f32x4){0, 0, 0, 0};
    if (!(W2S_FF_DBG & 2))
#pragma unroll
    for (int ks = 0; ks < KSP; ++ks) {
      bf16x8 ah[CO], al[CO];
#pragma unroll
      for (int nt = 0; nt < CO; ++nt) {
        ah[nt] = *reinterpret_cast<const bf16x8*>(wH + (nt * 16 + r) * WROW + ks * 32 + 8 * g);
        al[nt] = *reinterpret_cast<const bf16x8*>(wLo + (nt * 16 + r) * WROW + ks * 32 + 8 * g);
      }
      // tap and column of this lane's 8 K slots
      int j, col;
      if (HC == 16) { j = 2 * ks + (g >> 1); if (j > 2) j = 2; col = 8 * (g & 1); }   // (the 4th half-step meets zero weights)
      else { j = ks; col = 8 * g; }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = (wave * (16 * MT) + mt * 16 + r) * STRIDE + j;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(hH + row * RSh + col);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(hLo + row * RSh + col);
#pragma unroll
        for (int nt = 0; nt < CO; ++nt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[nt], bh, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[nt], bl, acc[mt][nt], 0, 0, 0);
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[nt], bh, acc[mt][nt], 0, 0, 0);
        }
      }
    }

    // ---- epilogue: store the pre-norm tensor + statistics partials
    f32x4 sA[CO], sB[CO];
#pragma unroll
    for (int nt = 0; nt < CO; ++nt) { sA[nt] = (f32x4){0, 0, 0, 0}; sB[nt] = (f32x4){0, 0, 0, 0}; }
    float* yb = P.y + (size_t)b * L_out * OC;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = wave * (16 * MT) + mt * 16 + r, pos = t0 + m;
      if (pos >= L_out || m >= TS) continue;
#pragma unroll
      for (int nt = 0; nt < CO; ++nt) {
        const f32x4 v = acc[mt][nt];
        sA[nt] += v;
        sB[nt] += v * v;
        if (!(W2S_FF_DBG & 8)) st4o(yb, (unsigned)pos * OC + nt * 16 + 4 * g, v);
      }
    }
#pragma unroll
    for (int nt = 0; nt < CO; ++nt) {
      f32x4 x1, x2;
      x1 = sA[nt]; x2 = sB[nt];
      row16_sum8(x1, x2);
      if (r == 0) {
        float* d = red + ((wave * CO + nt) * 4 + g) * 8;
        st4(d, x1);
        st4(d + 4, x2);
      }
    }
    __syncthreads();
    if (tid < CO * 32) {
      const int k = tid / OC, c = tid % OC;
      const int nt = c >> 4, gg = (c >> 2) & 3, e = c & 3;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[((w * CO + nt) * 4 + gg) * 8 + k * 4 + e];
      w2s_part_store(&P.part[(((size_t)b * P.ntiles + tile) * 2 + k) * OC + c], s);
    }
    if (++tile_cur == P.ntiles) { tile_cur = 0; ++b_cur; }
  };
  prefetch(I0{});
  prefetch(I1{});
  for (int tl = wfirst; tl < wend; tl += 2) {
    body(I0{}, tl);
    body(I1{}, tl + 1);
  }
}

template <int CI, int CO, int MT, int STRIDE, int PRO>
static int launch_fwd(const FwdP& P0, int nwg, hipStream_t s) {
  constexpr int TM = 64 * MT, TS = ff_ts(TM, STRIDE), HC = CI * 16, OC = CO * 16, NRh = (TS - 1) * STRIDE + 3, KD = (HC == 16 ? 2 : 3) * 32;
  FwdP P = P0;
  P.ntiles = (P.L_out + TS - 1) / TS;
  size_t lds = (size_t)4 * CO * 4 * 8 * 4 + (size_t)2 * 2 * ((ff_db(HC, STRIDE) ? 2 : 1) * NRh * ff_rs(HC) + OC * (KD + 8));
  if (PRO == W2S_PRO_FIRST) lds += (size_t)(NRh + 2) * 4;
  lds = (lds + 15) & ~(size_t)15;
  auto kern = conv_fwd_bf_kernel<CI, CO, MT, STRIDE, PRO>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return W2S_ELAUNCH;
  const int total = P.B * P.ntiles, grid = nwg < total ? nwg : total;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

static inline int fwd_mt(int cin, int cout, int stride) { return (cin == 32 && stride == 2) ? 2 : 4; }
// positions per tile (sizes the statistics partials: [B][ceil(L_out/tile)][2][cout]); 0 = combination not covered
extern "C" int w2s_conv_fwd_fused_tile(int cin, int cout, int stride) {
  const bool ok = (cin == 16 || cin == 32) && (cout == 16 || cout == 32) && cout >= cin && (stride == 1 || stride == 2);
  return ok ? ff_ts(64 * fwd_mt(cin, cout, stride), stride) : 0;
}

extern "C" int w2s_conv_fwd_fused(const float* x, const float* w, const float* st_in, const float* w1, float* y, float* part, int B,
                                  int L_in, int L_out, int cin, int cout, int stride, int pad, int pro, int nwg, void* stream) {
  if (!x || !w || !y || !part || B <= 0 || L_out <= 0 || nwg <= 0) return W2S_EINVAL;
  if (!w2s_conv_fwd_fused_tile(cin, cout, stride)) return W2S_EINVAL;
  if (pro != W2S_PRO_GELU && pro != W2S_PRO_IN_GELU && pro != W2S_PRO_FIRST) return W2S_EINVAL;
  if ((pro != W2S_PRO_GELU && !st_in) || (pro == W2S_PRO_FIRST && (!w1 || cin != 16 || stride != 1))) return W2S_EINVAL;
  if ((stride == 1 && L_out != L_in) || (stride == 2 && 2 * L_out != L_in) || (pad != 1 && pad != 2)) return W2S_EINVAL;
  if ((size_t)L_in * 32 * 4 >= ((size_t)1 << 32)) return W2S_EINVAL;
  FwdP P{x, w, st_in, w1, y, part, B, L_in, L_out, 0, pro, pad};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define W2S_FF(CI_, CO_, MT_, ST_) \
  if (cin == 16 * CI_ && cout == 16 * CO_ && stride == ST_) { \
    if (pro == W2S_PRO_GELU) return launch_fwd<CI_, CO_, MT_, ST_, W2S_PRO_GELU>(P, nwg, s); \
    if (pro == W2S_PRO_IN_GELU) return launch_fwd<CI_, CO_, MT_, ST_, W2S_PRO_IN_GELU>(P, nwg, s); \
  }
  if (pro == W2S_PRO_FIRST) return launch_fwd<1, 1, 4, 1, W2S_PRO_FIRST>(P, nwg, s);
  W2S_FF(1, 1, 4, 1) W2S_FF(1, 1, 4, 2) W2S_FF(1, 2, 4, 1) W2S_FF(2, 2, 4, 1) W2S_FF(2, 2, 2, 2)
#undef W2S_FF
  return W2S_EINVAL;
}
