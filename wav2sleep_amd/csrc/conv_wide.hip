// k=3 encoder convolutions of the >= 64-channel layers (forward, and the stride-1 data gradients) as a PERSISTENT split-precision
// kernel with the weights held in REGISTERS.
//
//   y[b,t,o] = EPI( sum_{j<3} sum_{c<HC} w[o][j][c] * PRO(x[b, t*STRIDE + roff(j) - 1, c]) )          (w2s_conv_forward's contract)
//
// Why not conv_cl_kernel here (profiles/r01_*): its one-tile workgroups serialise stage -> K loop -> epilogue with 2-3 workgroups per
// CU, and every wave re-fetches the layer's weight fragments from L2 on every K step (two dependent L2 round trips per 32 channels);
// the 64/128-channel layers ran at 0.24-0.38 of the HBM roof with the matrix pipe 10-28 % busy -- on neither roof.  Here:
//   * wave w of the workgroup owns output channels [16w, 16w+16) for ALL positions of a tile, so its A operand -- the 16 x K weight
//     slice, K = 3*HC -- is loaded ONCE per launch into registers as bf16 hi/lo fragments (HC = 128: 96 VGPRs) and the K loop
//     touches no global memory at all: 2 ds_read_b128 (activation hi/lo) feed 3 MFMAs;
//   * workgroups are persistent (grid-stride over (sample, tile)); the next tile's raw window is prefetched into registers while
//     the current one runs through the matrix cores, and is transformed (norm + GELU, or the instance-norm backward) on its way
//     into LDS once the MFMAs are done -- HBM latency, VALU prologue and MFMA overlap across the two waves of each SIMD;
//   * a wave's 16 channels x 64 positions need no cross-wave statistics reduction: the per-tile sums come out of a 16-lane DPP
//     row reduction and go straight to the partials.
// NW = waves per workgroup = cout / 16 (64 channels: 256 threads, 128 channels: 512 threads).
#include "conv_cl.inl"

struct WideP {
  const float* x; const float* x2; const float* st; const float* bst;
  const __bf16* w_hi; const __bf16* w_lo;
  const float* aux; const float* aux_st; const float* add_even;
  float* y; float* part;
  int B, L_in, L_out, ntiles, flip;
};

typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wsplit_store4(__bf16* hi, __bf16* lo, int off, f32x4 t) {
  wbf16x4 h = {(__bf16)t.x, (__bf16)t.y, (__bf16)t.z, (__bf16)t.w};
  wbf16x4 l = {(__bf16)(t.x - (float)h.x), (__bf16)(t.y - (float)h.y), (__bf16)(t.z - (float)h.z), (__bf16)(t.w - (float)h.w)};
  *reinterpret_cast<wbf16x4*>(hi + off) = h;
  *reinterpret_cast<wbf16x4*>(lo + off) = l;
}

template <int CI, int NW, int STRIDE, int PRO, int EPI>
__global__ __launch_bounds__(64 * NW) void conv_wide_kernel(WideP P) {
  extern __shared__ f32x4 smem4[];
  constexpr int MT = 4, TM = 16 * MT;                    // 64 output positions per tile
  constexpr int HC = CI * 16, OC = NW * 16, NTH = 64 * NW;
  constexpr int RSE = HC + 8;                            // bf16 elements per LDS row (16 B pad: conflict-free ds_read_b128)
  constexpr int NR = (TM - 1) * STRIDE + 3;              // window rows; row 0 = input position t0*STRIDE - 1
  constexpr int QN = HC / 32, KS = 3 * QN;               // K steps of 32: ks = tap * QN + q
  constexpr bool TWO = (PRO == W2S_PRO_INBWD);
  constexpr bool FLIP = (PRO == W2S_PRO_INBWD);        // the data gradient runs the taps backwards over the [cin][taps][cout] packing
  __bf16* hiL = reinterpret_cast<__bf16*>(smem4);
  __bf16* loL = hiL + NR * RSE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int L_in = P.L_in, L_out = P.L_out;

  // ---- this wave's weight slice, once per launch: fragment-major planes [OC/16][KS][64 lanes][8]
  bf16x8 ah[KS], al[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const size_t wo = ((size_t)wave * KS + ks) * 512 + lane * 8;
    ah[ks] = *reinterpret_cast<const bf16x8*>(P.w_hi + wo);
    al[ks] = *reinterpret_cast<const bf16x8*>(P.w_lo + wo);
  }

  constexpr int c4n = HC / 4, rstep = NTH / c4n, NH = (NR + rstep - 1) / rstep;
  const int myc4 = tid % c4n, row0 = tid / c4n, mych = myc4 * 4;
  f32x4 rx[NH], rx2[TWO ? NH : 1];
  auto prefetch = [&](int tl) {
    const int b = tl / P.ntiles, t0 = (tl % P.ntiles) * TM;
    const int rb = t0 * STRIDE - 1;
    const float* xb = P.x + (size_t)b * L_in * HC;
    const float* x2b = TWO ? P.x2 + (size_t)b * L_in * HC : nullptr;
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const int row = row0 + k * rstep, gr = rb + row;
      const bool ok = row < NR && gr >= 0 && gr < L_in;
      const unsigned off = (unsigned)gr * HC + mych;
      rx[k] = ok ? ld4o(xb, off) : (f32x4){0, 0, 0, 0};
      if constexpr (TWO) rx2[k] = ok ? ld4o(x2b, off) : (f32x4){0, 0, 0, 0};
    }
  };
  auto commit = [&](int tl) {
    const int b = tl / P.ntiles, t0 = (tl % P.ntiles) * TM;
    const int rb = t0 * STRIDE - 1;
    f32x4 pm = {0, 0, 0, 0}, pr = {1, 1, 1, 1}, ps1 = {0, 0, 0, 0}, ps2 = {0, 0, 0, 0};
    if (PRO != W2S_PRO_GELU) {
      const float* st = P.st + ((size_t)b * HC + mych) * 2;
      const f32x4 s01 = ld4(st), s23 = ld4(st + 4);
      pm = (f32x4){s01.x, s01.z, s23.x, s23.z}; pr = (f32x4){s01.y, s01.w, s23.y, s23.w};
      if (TWO) {
        const float* bs = P.bst + ((size_t)b * HC + mych) * 2;
        const f32x4 b01 = ld4(bs), b23 = ld4(bs + 4);
        ps1 = (f32x4){b01.x, b01.z, b23.x, b23.z}; ps2 = (f32x4){b01.y, b01.w, b23.y, b23.w};
      }
    }
#pragma unroll
    for (int k = 0; k < NH; ++k) {
      const int row = row0 + k * rstep, gr = rb + row;
      if (row < NR) {
        const bool ok = gr >= 0 && gr < L_in;
        f32x4 v2 = rx[k];
        if constexpr (TWO) v2 = rx2[k];
        wsplit_store4(hiL, loL, row * RSE + mych, ok ? pro_apply(PRO, rx[k], v2, pm, pr, ps1, ps2) : (f32x4){0, 0, 0, 0});
      }
    }
  };

  const int total = P.B * P.ntiles;
  const int ch0 = wave * 16 + 4 * g;   // this lane's 4 consecutive output channels (D fragment: position r, channels 4g..4g+3)
  if ((int)blockIdx.x < total) prefetch(blockIdx.x);
  for (int tl = blockIdx.x; tl < total; tl += gridDim.x) {
    const int b = tl / P.ntiles, tile = tl % P.ntiles;
    const int t0 = tile * TM;
    __syncthreads();   // the previous tile's LDS reads are done
    commit(tl);
    if (tl + (int)gridDim.x < total) prefetch(tl + gridDim.x);
    // epilogue operands of THIS tile, issued now so that their latency hides behind the K loop
    f32x4 ax[EPI == W2S_EPI_GP ? MT : 1], ae[EPI == W2S_EPI_GP ? MT : 1];
    if (EPI == W2S_EPI_GP) {
      const float* ab = P.aux + (size_t)b * L_out * OC;
      const float* eb = P.add_even ? P.add_even + (size_t)b * (L_out >> 1) * OC : nullptr;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int pos = t0 + mt * 16 + r;
        ax[mt] = (pos < L_out) ? ld4o(ab, (unsigned)pos * OC + ch0) : (f32x4){0, 0, 0, 0};
        ae[mt] = (eb && pos < L_out && !(pos & 1)) ? ld4o(eb, (unsigned)(pos >> 1) * OC + ch0) : (f32x4){0, 0, 0, 0};
      }
    }
    __syncthreads();

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int j = ks / QN, q = ks % QN;
      const int rowoff = FLIP ? 2 - j : j;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = (mt * 16 + r) * STRIDE + rowoff;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(hiL + row * RSE + q * 32 + 8 * g);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(loL + row * RSE + q * 32 + 8 * g);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], bh, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], bl, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ks], bh, acc[mt], 0, 0, 0);
      }
    }

    // ---- epilogue
    f32x4 sA = {0, 0, 0, 0}, sB = {0, 0, 0, 0};
    float* yb = P.y + (size_t)b * L_out * OC;
    f32x4 am = {0, 0, 0, 0}, ar = {1, 1, 1, 1};
    if (EPI == W2S_EPI_GP && P.aux_st) {
      const float* st = P.aux_st + ((size_t)b * OC + ch0) * 2;
      const f32x4 s01 = ld4(st), s23 = ld4(st + 4);
      am = (f32x4){s01.x, s01.z, s23.x, s23.z}; ar = (f32x4){s01.y, s01.w, s23.y, s23.w};
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int pos = t0 + mt * 16 + r;
      if (pos >= L_out) continue;
      f32x4 v = acc[mt];
      if (EPI == W2S_EPI_GP) {
        const f32x4 n = (ax[mt] - am) * ar;
        v = (v + ae[mt]) * gelu_grad4(n);
        sA += v;
        sB += v * n;
      } else {
        sA += v;
        sB += v * v;
      }
      st4o(yb, (unsigned)pos * OC + ch0, v);
    }
    if (P.part) {
      f32x4 x1, x2;
      x1.x = row16_sum(sA.x); x1.y = row16_sum(sA.y); x1.z = row16_sum(sA.z); x1.w = row16_sum(sA.w);
      x2.x = row16_sum(sB.x); x2.y = row16_sum(sB.y); x2.z = row16_sum(sB.z); x2.w = row16_sum(sB.w);
      if (r == 0) {
        float* d = P.part + (((size_t)b * P.ntiles + tile) * 2) * OC + ch0;
        st4(d, x1);
        st4(d + OC, x2);
      }
    }
  }
}

template <int CI, int NW, int STRIDE, int PRO, int EPI>
static int launch_wide(const w2s_conv_args& a, hipStream_t s) {
  constexpr int TM = 64, HC = CI * 16, NR = (TM - 1) * STRIDE + 3;
  WideP P{a.x, a.x2, a.pro_stats, a.pro_bstats, static_cast<const __bf16*>(a.w_hi), static_cast<const __bf16*>(a.w_lo), a.aux, a.aux_stats,
          a.add_even, a.y, a.part, a.B, a.L_in, a.L_out, (a.L_out + TM - 1) / TM, a.flip};
  const size_t lds = (size_t)2 * NR * (HC + 8) * 2;
  auto kern = conv_wide_kernel<CI, NW, STRIDE, PRO, EPI>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return W2S_ELAUNCH;
  const int total = P.B * P.ntiles;
  static const char* e = getenv("W2S_WIDE_WGS");   // tuning only: workgroups per CU
  const int per_cu = e ? atoi(e) : (NW >= 8 ? 1 : 2);
  const int nwg = 256 * (per_cu > 0 ? per_cu : 1);
  hipLaunchKernelGGL(kern, dim3(nwg < total ? nwg : total), dim3(64 * NW), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// 1 = this launch is not one of the wide kernel's shapes (the caller falls through to conv_cl_kernel)
static bool wide_shape(const w2s_conv_args& a) {
  if (!a.w_hi || !a.w_lo || a.mode != W2S_MODE_CONTIG || a.taps != 3 || a.dil != 1 || a.pad != 1) return false;
  if (a.y2 || a.rowkeep || a.bias || a.stat_out) return false;
  if (a.ldx != a.cin || a.ldy != a.cout || (a.aux && a.ld_aux != a.cout)) return false;
  if (a.cin < 32 || a.cout < 32 || (a.cin < 64 && a.cout < 64)) return false;
  const bool fwd = a.epi == W2S_EPI_STATS && !a.flip && (a.pro == W2S_PRO_GELU || a.pro == W2S_PRO_IN_GELU);
  const bool dgr = a.epi == W2S_EPI_GP && a.flip && a.pro == W2S_PRO_INBWD && a.stride == 1;
  if (!fwd && !dgr) return false;
  if (a.pro == W2S_PRO_GELU && a.stride != 1) return false;
  static const char* off = getenv("W2S_NO_WIDE");   // tuning only
  return !off;
}
// dry != 0: only answer whether an instance takes this launch (0) or not (1) -- w2s_conv_tile sizes the statistics partials with it
// (the wide kernel's tile is 64 positions whatever the channel count)
int w2s_conv_wide_try(const w2s_conv_args& a, hipStream_t s, int dry) {
  if (!wide_shape(a)) return 1;
#define W2S_WIDE(CI_, NW_, ST_, PRO_, EPI_) \
  if (a.cin == 16 * CI_ && a.cout == 16 * NW_ && a.stride == ST_ && a.pro == PRO_ && a.epi == EPI_) \
    return dry ? 0 : launch_wide<CI_, NW_, ST_, PRO_, EPI_>(a, s);
  W2S_WIDE(2, 4, 1, W2S_PRO_GELU, W2S_EPI_STATS) W2S_WIDE(4, 4, 1, W2S_PRO_GELU, W2S_EPI_STATS)
  W2S_WIDE(4, 8, 1, W2S_PRO_GELU, W2S_EPI_STATS) W2S_WIDE(8, 8, 1, W2S_PRO_GELU, W2S_EPI_STATS)
  W2S_WIDE(4, 4, 1, W2S_PRO_IN_GELU, W2S_EPI_STATS) W2S_WIDE(8, 8, 1, W2S_PRO_IN_GELU, W2S_EPI_STATS)
  W2S_WIDE(4, 4, 2, W2S_PRO_IN_GELU, W2S_EPI_STATS) W2S_WIDE(8, 8, 2, W2S_PRO_IN_GELU, W2S_EPI_STATS)
  W2S_WIDE(4, 4, 1, W2S_PRO_INBWD, W2S_EPI_GP) W2S_WIDE(8, 8, 1, W2S_PRO_INBWD, W2S_EPI_GP)
  W2S_WIDE(4, 2, 1, W2S_PRO_INBWD, W2S_EPI_GP) W2S_WIDE(8, 4, 1, W2S_PRO_INBWD, W2S_EPI_GP)
#undef W2S_WIDE
  return 1;
}
