// k=3 encoder convolutions of the >= 64-channel layers (forward, the stride-1 data gradients and the transposed stride-2 data gradient)
// as a PERSISTENT, ROLE-SPLIT split-precision kernel with the weights held in REGISTERS.
//
//   y[b,t,o] = EPI( sum_{j<3} sum_{c<HC} w[o][j][c] * PRO(x[b, t*STRIDE + roff(j) - pad, c]) )          (w2s_conv_forward's contract)
//
// Why not conv_cl_kernel here (profiles/r01_*): its one-tile workgroups serialise stage -> K loop -> epilogue with 2-3 workgroups per
// CU, and every wave re-fetches the layer's weight fragments from L2 on every K step (two dependent L2 round trips per 32 channels);
// the 64/128-channel layers ran at 0.24-0.38 of the HBM roof with the matrix pipe 10-28 % busy -- on neither roof.  Here:
//   * NW = cout/16 CONSUMER waves: wave w owns output channels [16w, 16w+16) for ALL positions of a tile, so its A operand -- the
//     16 x K weight slice, K = 3*HC -- is loaded ONCE per launch into registers as bf16 hi/lo fragments (HC = 128: 96 VGPRs) and the
//     K loop touches no global memory at all: 2 ds_read_b128 (activation hi/lo) feed 3 MFMAs; the epilogue (statistics partials by a
//     16-lane DPP row reduction, or the GELU' product) stores y directly;
//   * 4 PRODUCER waves stream the raw window of the next PD tiles into registers (unconditional, clamped loads), apply the on-load
//     transform (norm + GELU, or the instance-norm backward) with the statistics from an LDS table, split into bf16 (hi, lo) and
//     write the window of tile i + 1 into the other LDS buffer while the consumers run tile i through the matrix cores;
//   * workgroups are persistent (grid-stride over (sample, tile)), one barrier per tile.
#include <type_traits>
#include "conv_cl.inl"
#ifndef W2S_WIDE_AE_LATE
#define W2S_WIDE_AE_LATE 1
#endif

struct WideP {
  const float* x; const float* x2; const float* st; const float* bst;
  const __bf16* w_hi; const __bf16* w_lo;
  const float* aux; const float* aux_st; const float* add_even;
  float* y; float* part;
  int B, L_in, L_out, ntiles, flip, pad;   // window row 0 = input position t0*STRIDE - pad (1: symmetric; forward 2 / data gradient 0: causal)
  int dbg;   // tuning only (W2S_WIDE_DBG): 1 = no prologue arithmetic, 2 = no MFMA loop, 4 = no LDS staging, 8 = no stores, 16 = no loads
};

#define wsplit_store4 split_store4   /* w2s_common.h: the explicit bit form */

// NW consumer waves (wave w: output channels [16w, 16w+16), weights in registers, MFMA + epilogue) and NP producer waves (global
// loads of the raw window, on-load transform, bf16 hi/lo split, LDS writes of the NEXT tile into the other buffer).  A SIMD hosts
// consumer and producer waves side by side (a workgroup's waves are dealt to the SIMDs cyclically), so the producers' VALU work
// runs beside the consumers' MFMAs instead of before them: with one role per wave the phases of a tile add up (measured on the
// one-role form of this kernel: VALU 38 % + MFMA 19-30 % busy, waves waiting 50 % of their lifetime, compute alone 135 / 91 us
// for the 64 / 128-channel conv2 against 93 / 38 us for its memory traffic alone).
// UP2 = 1: the transposed stride-2 form (data gradient of the stride-2 conv3, w2s_conv_forward's W2S_MODE_UP2): output position
// t' = 2u + phase reads the gradient rows u, u+1; m-tiles alternate phase (mt & 1), their row block is mt >> 1.  Symmetric padding:
// even outputs W_1^T g[u], odd outputs W_2^T g[u] + W_0^T g[u+1]; causal (pad 2): even W_2^T g[u] + W_0^T g[u+1], odd W_1^T g[u+1].
template <int CI, int NW, int NP, int STRIDE, int PRO, int EPI, int MT, int PD, int UP2, int CZ>   // CZ: UP2 with causal padding
__device__ __forceinline__ void conv_wide_body(const WideP& P) {
  extern __shared__ f32x4 smem4[];
  constexpr int TM = 16 * MT;                            // output positions per tile (all consumer waves share them)
  constexpr int HC = CI * 16, OC = NW * 16, NPT = 64 * NP;
  // LDS row stride in bf16 elements: +32 B (stride 1) / +16 B (stride 2) makes the 16-lane groups of the B-operand ds_read_b128
  // (lanes {0-3,12-15,20-27}, ... MI355X_MICROARCH.md LDS) hit 16 distinct 16-B slots: the +16 B rows of the first cut were 2-way
  // conflicted on every stride-1 read (SQ_LDS_BANK_CONFLICT = 48 % of SQ_LDS_IDX_ACTIVE)
  constexpr int RSE = HC + (STRIDE == 1 ? 16 : 8);
  constexpr int NR = UP2 ? TM / 2 + 1 : (TM - 1) * STRIDE + 3;   // window rows; row 0 = input position t0*STRIDE - pad (UP2: t0/2)
  constexpr int QN = HC / 32, KS = 3 * QN;               // K steps of 32: ks = tap * QN + q
  constexpr bool TWO = (PRO == W2S_PRO_INBWD || PRO == W2S_PRO_INBWD_GP);
  constexpr bool FLIP = (PRO == W2S_PRO_INBWD);          // the data gradient runs the taps backwards over the [cin][taps][cout] packing
  constexpr int BUF = 2 * NR * RSE;                      // bf16 elements of one window buffer (hi plane, lo plane)
  __bf16* lds = reinterpret_cast<__bf16*>(smem4);
  float* stL = reinterpret_cast<float*>(lds + 2 * BUF);   // [B][HC][2] (mean, rstd) and, data gradient, [B][HC][2] backward sums behind it
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L_in = P.L_in, L_out = P.L_out;
  const int total = P.B * P.ntiles;
  const int G = (int)gridDim.x;
  // tiles of this workgroup: a contiguous run of the (sample, tile) list (blocked; w2s_common.h "Statistics finalisation"); the producers
  // never exceed the tile count
  const W2SRun wrun = w2s_block_part(total, G, blockIdx.x);
  const int first = wrun.first;
  const int run_b0 = first / P.ntiles, run_t0 = first - run_b0 * P.ntiles;   // the run's first (sample, tile): the one division of the launch
  if (PRO != W2S_PRO_GELU) {
    for (int i = tid; i < P.B * HC * 2; i += 64 * (NW + NP)) {
      stL[i] = P.st[i];
      if (TWO) stL[P.B * HC * 2 + i] = P.bst[i];
    }
    __syncthreads();   // the producers read the table before the first round's barrier
  }

  // tiles of this workgroup: first, first + 1, ...; iteration i of the producers stages tile i, iteration i + 1 of the consumers eats it
  const int nt_wg = wrun.count;                                   // >= 1 (the grid never exceeds the tile count)
  const int NI = ((nt_wg + 1 + PD - 1) / PD) * PD;                // barrier rounds, padded to whole prefetch cycles

  if (wave >= NW) {
    // ================================================= producer waves =================================================
    // PD register sets: the raw rows of the next PD tiles are in flight while one is transformed.  Every global load below is
    // UNCONDITIONAL (addresses clamped into the tensor, out-of-range rows zeroed after the fact) and the loop is unrolled by PD, so the
    // only control flow around vector-memory instructions is the loop itself: with the loads under `ok ? load : 0` branches hipcc
    // waited vmcnt(0) before every row -- it drained the whole prefetch queue each time, i.e. one load in flight per wave (measured:
    // producers alone 165 us for the 64-channel conv2, 3 us per tile round trip).
#ifndef W2S_WIDE_PPRIO
#define W2S_WIDE_PPRIO 1   // tuning: issue priority of the forward instances' producer waves
#endif
    if (EPI == W2S_EPI_STATS && W2S_WIDE_PPRIO) __builtin_amdgcn_s_setprio(W2S_WIDE_PPRIO);
    const int pt = tid - 64 * NW;
    constexpr int c4n = HC / 4, rstep = NPT / c4n, NH = (NR + rstep - 1) / rstep;
    const int myc4 = pt % c4n, row0 = pt / c4n, mych = myc4 * 4;
    f32x4 rx[PD][NH], rx2[TWO ? PD : 1][TWO ? NH : 1];
    // (sample base, first window row) of this workgroup's tile number min(i, nt_wg - 1): ONCE per tile -- looked up per row, the run
    // position (a scalar loop) and the 64-bit base were recomputed NH times per stage and not merged by the compiler
    struct TileAt { const float* x; const float* x2; int rb; };
    auto tile_at = [&](int i) {
      int b, tile_;
      w2s_run_pos(run_b0, run_t0, P.ntiles, min(i, nt_wg - 1), b, tile_);
      const int t0 = tile_ * TM;
      const size_t sb = (size_t)b * L_in * HC;
      return TileAt{P.x + sb, TWO ? P.x2 + sb : nullptr, UP2 ? t0 / 2 : t0 * STRIDE - P.pad};
    };
    auto load_row = [&](auto SET, const TileAt& ta, int k) {   // window row k of that tile
      constexpr int S = decltype(SET)::value;
      const int row = min(row0 + k * rstep, NR - 1), gr = min(max(ta.rb + row, 0), L_in - 1);
      const unsigned off = (unsigned)gr * HC + mych;
      rx[S][k] = ld4o(ta.x, off);
      if constexpr (TWO) rx2[S][k] = ld4o(ta.x2, off);
    };
    // stage tile i (its raw rows are in register set SET) into LDS buffer i & 1, refilling the set with tile i + PD
    auto stage = [&](auto SET, int i) {
      constexpr int S = decltype(SET)::value;
      int b, tile_;   // (padding rounds i >= nt_wg only keep the load queue regular)
      w2s_run_pos(run_b0, run_t0, P.ntiles, min(i, nt_wg - 1), b, tile_);
      const int t0 = tile_ * TM;
      const int rb = UP2 ? t0 / 2 : t0 * STRIDE - P.pad;
      __bf16* hiL = lds + (i & 1) * BUF;
      __bf16* loL = hiL + NR * RSE;
      // per-channel coefficients of the on-load transform, formed once per tile so that the per-element work is fused multiply-adds:
      //   IN + GELU: n = x r + (-m r);  IN backward: gy = r g + (-r^2 s2) y + r (r s2 m - s1)  (with GELU': n = r y + (-m r),
      //   gy = (r g) GELU'(n) + n (-r s2) + (-r s1)) -- bwd_fused.hip `commit`
      f32x4 cA = {1, 1, 1, 1}, cB = {0, 0, 0, 0}, cC = {0, 0, 0, 0}, cD = {0, 0, 0, 0};
      if (PRO != W2S_PRO_GELU) {   // statistics from the LDS copy made at kernel start (lgkmcnt: does not touch the vector-memory queue)
        const float* st = stL + (b * HC + mych) * 2;
        const f32x4 s01 = ld4(st), s23 = ld4(st + 4);
        const f32x4 pm = {s01.x, s01.z, s23.x, s23.z}, pr = {s01.y, s01.w, s23.y, s23.w};
        cA = pr; cB = -(pm * pr);
        if (TWO) {
          const float* bs = stL + ((P.B + b) * HC + mych) * 2;
          const f32x4 b01 = ld4(bs), b23 = ld4(bs + 4);
          const f32x4 ps1 = {b01.x, b01.z, b23.x, b23.z}, ps2 = {b01.y, b01.w, b23.y, b23.w};
          if (PRO == W2S_PRO_INBWD_GP) { cC = -(pr * ps2); cD = -(pr * ps1); }
          else { cB = -(pr * pr * ps2); cC = pr * (pr * ps2 * pm - ps1); }
        }
      }
      const TileAt nxt = tile_at(i + PD);
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        const int row = row0 + k * rstep;
        f32x4 v1 = rx[S][k], v2 = rx[S][k];
        if constexpr (TWO) v2 = rx2[S][k];
        load_row(SET, nxt, k);   // the register is free again: the load of the tile PD rounds ahead goes out at once
        if (row < NR && !(P.dbg & 4)) {   // (a padding round stages its clamped tile again, into the buffer nobody reads: no `live` branch)
          f32x4 tv;   // (no run-time branch in here: control flow around the vector-memory queue makes hipcc drain it -- see above)
          if constexpr (PRO == W2S_PRO_GELU) tv = gelu4(v1);
          else if constexpr (PRO == W2S_PRO_IN_GELU) tv = gelu4(fma4(v1, cA, cB));
          else if constexpr (PRO == W2S_PRO_INBWD) tv = fma4(cA, v1, fma4(cB, v2, cC));
          else if constexpr (PRO == W2S_PRO_INBWD_GP) { const f32x4 n = fma4(v2, cA, cB); tv = fma4(v1 * cA, gelu_grad4(n), fma4(n, cC, cD)); }
          else tv = pro_apply(PRO, v1, v2, cA, cA, cA, cA);   // (not instantiated)
          wsplit_store4(hiL, loL, row * RSE + mych, (P.dbg & 1) ? v1 + v2 : tv);
        }
      }
      // rows outside the sample (zero padding; a sample's first / last tile only -- uniform): zeroed after the fact by the lanes that stored them
      if (!(P.dbg & 4) && (rb < 0 || rb + NR > L_in)) {
#pragma unroll
        for (int k = 0; k < NH; ++k) {
          const int row = row0 + k * rstep, gr = rb + row;
          if (row < NR && (gr < 0 || gr >= L_in)) { zero_store4(hiL, row * RSE + mych); zero_store4(loL, row * RSE + mych); }
        }
      }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    {
      const TileAt a0 = tile_at(0), a1 = tile_at(1), a2 = tile_at(2);
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        load_row(I0{}, a0, k);
        if constexpr (PD > 1) load_row(I1{}, a1, k);
        if constexpr (PD > 2) load_row(I2{}, a2, k);
      }
    }
#ifdef W2S_WIDE_STAMP   // diagnostic build only (tools/altlib.sh): cycles of block 0's first producer wave in stage / at the barrier -> part[4..7]
    unsigned long long ts = 0, tb = 0;
    for (int it = 0; it < NI; it += PD) {
      unsigned long long c0 = __builtin_amdgcn_s_memtime();
      stage(I0{}, it);
      unsigned long long c1 = __builtin_amdgcn_s_memtime();
      __syncthreads();
      unsigned long long c2 = __builtin_amdgcn_s_memtime();
      ts += c1 - c0; tb += c2 - c1;
      if constexpr (PD > 1) { c0 = __builtin_amdgcn_s_memtime(); stage(I1{}, it + 1); c1 = __builtin_amdgcn_s_memtime(); __syncthreads(); c2 = __builtin_amdgcn_s_memtime(); ts += c1 - c0; tb += c2 - c1; }
      if constexpr (PD > 2) { c0 = __builtin_amdgcn_s_memtime(); stage(I2{}, it + 2); c1 = __builtin_amdgcn_s_memtime(); __syncthreads(); c2 = __builtin_amdgcn_s_memtime(); ts += c1 - c0; tb += c2 - c1; }
    }
    if (blockIdx.x == 0 && tid == 64 * NW) { P.part[4] = (float)ts; P.part[5] = (float)tb; P.part[6] = (float)NI; P.part[7] = (float)nt_wg; }
    return;
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
  const int r = lane & 15, g = lane >> 4;
  bf16x8 ah[KS], al[KS];   // this wave's 16 x K weight slice, once per launch: fragment-major planes [OC/16][KS][64 lanes][8]
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const size_t wo = ((size_t)wave * KS + ks) * 512 + lane * 8;
    ah[ks] = *reinterpret_cast<const bf16x8*>(P.w_hi + wo);
    al[ks] = *reinterpret_cast<const bf16x8*>(P.w_lo + wo);
  }
  const int ch0 = wave * 16 + 4 * g;   // this lane's 4 consecutive output channels (D fragment: position r, channels 4g..4g+3)
  __syncthreads();                     // round 0 of the producers: the first window is in buffer 0
#ifdef W2S_WIDE_STAMP
  unsigned long long tk = 0, te = 0, tw = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#endif
  for (int i = 0; i < NI - 1; ++i) {
    if (i >= nt_wg) { __syncthreads(); continue; }   // padding rounds of the producers' prefetch cycle
#ifdef W2S_WIDE_STAMP
    c0 = __builtin_amdgcn_s_memtime();
#endif
    int b, tile;
    w2s_run_pos(run_b0, run_t0, P.ntiles, i, b, tile);
    const int t0 = tile * TM;
    const __bf16* hiL = lds + (i & 1) * BUF;
    const __bf16* loL = hiL + NR * RSE;
    // epilogue operands of THIS tile, issued now so that their latency hides behind the K loop
    constexpr bool AE_LATE = W2S_WIDE_AE_LATE && NW >= 8 && UP2;
    f32x4 ax[EPI == W2S_EPI_GP ? MT : 1], ae[(EPI == W2S_EPI_GP && !UP2) ? MT : 1];
    if (EPI == W2S_EPI_GP && !AE_LATE) {
      const float* ab = P.aux + (size_t)b * L_out * OC;
      const float* eb = P.add_even ? P.add_even + (size_t)b * (L_out >> 1) * OC : nullptr;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int pos = UP2 ? t0 + 2 * ((mt >> 1) * 16 + r) + (mt & 1) : t0 + mt * 16 + r;
        ax[mt] = (pos < L_out) ? ld4o(ab, (unsigned)pos * OC + ch0) : (f32x4){0, 0, 0, 0};
        if constexpr (!UP2 && !AE_LATE) ae[mt] = (eb && !(pos & 1) && (pos >> 1) < (L_out >> 1)) ? ld4o(eb, (unsigned)(pos >> 1) * OC + ch0) : (f32x4){0, 0, 0, 0};
      }
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0, 0, 0, 0};
    if (!(P.dbg & 2))
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int j = ks / QN, q = ks % QN;
      const int rowoff = UP2 ? ((j == 0) ? 1 : (j == 1 && CZ) ? 1 : 0) : (FLIP ? 2 - j : j);
      const int phase = (j == 1) ? (CZ ? 1 : 0) : (CZ ? 0 : 1);   // UP2: the output parity tap j feeds (compile-time: the other m-tiles vanish)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (UP2 && (mt & 1) != phase) continue;
        const int row = UP2 ? (mt >> 1) * 16 + r + rowoff : (mt * 16 + r) * STRIDE + rowoff;
#ifdef W2S_WIDE_NOLDS   // diagnostic builds (numerics wrong on purpose): the K loop without its LDS reads / without its MFMAs
        const bf16x8 bh = ah[(ks + mt) % KS], bl = al[(ks + mt) % KS];
#else
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(hiL + row * RSE + q * 32 + 8 * g);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(loL + row * RSE + q * 32 + 8 * g);
#endif
#ifdef W2S_WIDE_NOMFMA
        { typedef float f4 __attribute__((ext_vector_type(4))); f4 t1, t2; __builtin_memcpy(&t1, &bh, 16); __builtin_memcpy(&t2, &bl, 16); acc[mt] += t1 * t2; }
#else
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], bh, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], bl, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ks], bh, acc[mt], 0, 0, 0);
#endif
      }
    }

    if constexpr (EPI == W2S_EPI_GP && AE_LATE) {
      // 128 channels: 96 weight registers + accumulators + activation fragments leave no room for 16-32 epilogue operands across the K
      // loop.  Prefetched before it they spilled weight fragments, and every reload in the loop carried `s_waitcnt vmcnt(0)` -- i.e. the K
      // loop of each tile began by waiting for these very HBM loads, and ran with `lgkmcnt(0)` after every LDS read (3.7 x the cycles
      // of the forward kernel's identical K loop, in-kernel stamps).  Issued here their latency is exposed once per tile instead.
      const float* ab = P.aux + (size_t)b * L_out * OC;
      const float* eb = P.add_even ? P.add_even + (size_t)b * (L_out >> 1) * OC : nullptr;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int pos = UP2 ? t0 + 2 * ((mt >> 1) * 16 + r) + (mt & 1) : t0 + mt * 16 + r;
        ax[mt] = (pos < L_out) ? ld4o(ab, (unsigned)pos * OC + ch0) : (f32x4){0, 0, 0, 0};
        if constexpr (!UP2) ae[mt] = (eb && !(pos & 1) && (pos >> 1) < (L_out >> 1)) ? ld4o(eb, (unsigned)(pos >> 1) * OC + ch0) : (f32x4){0, 0, 0, 0};
      }
    }
    // ---- epilogue
#ifdef W2S_WIDE_STAMP
    asm volatile("" :: "v"(acc[0]), "v"(acc[MT - 1]));
    c1 = __builtin_amdgcn_s_memtime();
#endif
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
      const int pos = UP2 ? t0 + 2 * ((mt >> 1) * 16 + r) + (mt & 1) : t0 + mt * 16 + r;
      if (pos >= L_out) continue;
      f32x4 v = acc[mt];
      if (EPI == W2S_EPI_GP) {
        const f32x4 n = (ax[mt] - am) * ar;
        if constexpr (!UP2) v += ae[mt];
        v = v * gelu_grad4(n);
        sA += v;
        sB += v * n;
      } else {
        sA += v;
        sB += v * v;
      }
      if (!(P.dbg & 8)) st4o(yb, (unsigned)pos * OC + ch0, v);
    }
    if (P.part) {
      f32x4 x1, x2;
      x1 = sA; x2 = sB;
      row16_sum8(x1, x2);
      if (r == 0) {
        float* d = P.part + (((size_t)b * P.ntiles + tile) * 2) * OC + ch0;
        st4(d, x1);
        st4(d + OC, x2);
      }
    }
#ifdef W2S_WIDE_STAMP
    c2 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();   // the producers have staged the next window; this one may be overwritten
#ifdef W2S_WIDE_STAMP
    c3 = __builtin_amdgcn_s_memtime();
    tk += c1 - c0; te += c2 - c1; tw += c3 - c2;
#endif
  }
#ifdef W2S_WIDE_STAMP
  if (blockIdx.x == 0 && tid == 0) { P.part[0] = (float)tk; P.part[1] = (float)te; P.part[2] = (float)tw; P.part[3] = (float)nt_wg; }
#endif
}

// Two entry points over one body.  The forward instances (statistics epilogue) are PRODUCER-bound (in-kernel stamps, docs/lab_notes_r4.md
// section 11): their four producer waves issue ~700 vector instructions per tile beside the consumers' MFMAs, and a packed-fp32
// instruction issued beside a busy matrix pipe costs ~20 cycles more than the two plain ones it replaces (MI355X_MICROARCH.md, "price of
// one filler beside MFMAs").  `conv_wide_np_kernel` is the same code compiled without packed-fp32 selection, with the producers at
// s_setprio 1 (they are the younger waves and lose every issue arbitration otherwise).  The data-gradient instances are consumer-bound
// (GELU' epilogue, no MFMA beside it) and keep the packed form.
template <int CI, int NW, int NP, int STRIDE, int PRO, int EPI, int MT, int PD, int UP2, int CZ>
__global__ __launch_bounds__(64 * (NW + NP)) void conv_wide_kernel(WideP P) { conv_wide_body<CI, NW, NP, STRIDE, PRO, EPI, MT, PD, UP2, CZ>(P); }
template <int CI, int NW, int NP, int STRIDE, int PRO, int EPI, int MT, int PD, int UP2, int CZ>
__global__ __launch_bounds__(64 * (NW + NP)) __attribute__((target("no-packed-fp32-ops"))) void conv_wide_np_kernel(WideP P) {
  conv_wide_body<CI, NW, NP, STRIDE, PRO, EPI, MT, PD, UP2, CZ>(P);
}

template <int CI, int NW, int STRIDE, int PRO, int EPI, int MT, int NP = 4, int UP2 = 0, int CZ = 0>
static int launch_wide(const w2s_conv_args& a, hipStream_t s) {
  constexpr int TM = 16 * MT, HC = CI * 16, NR = UP2 ? TM / 2 + 1 : (TM - 1) * STRIDE + 3, RSE = HC + (STRIDE == 1 ? 16 : 8);
  WideP P{a.x, a.x2, a.pro_stats, a.pro_bstats, static_cast<const __bf16*>(a.w_hi), static_cast<const __bf16*>(a.w_lo), a.aux, a.aux_stats,
          a.add_even, a.y, a.part, a.B, a.L_in, a.L_out, (a.L_out + TM - 1) / TM, a.flip, a.pad, 0};
  { static const char* d = getenv("W2S_WIDE_DBG"); if (d) P.dbg = atoi(d); }
  size_t lds = (size_t)2 * 2 * NR * RSE * 2;   // two window buffers x (hi, lo) planes, bf16
  constexpr bool TWO = (PRO == W2S_PRO_INBWD || PRO == W2S_PRO_INBWD_GP);
  if (PRO != W2S_PRO_GELU) lds += (size_t)a.B * HC * 2 * 4 * (TWO ? 2 : 1);   // the statistics tables
  // producer prefetch depth: three tiles in flight where the register budget allows (kernel-wide allocation: 64-channel workgroups
  // of 8 waves run two per CU = 128 VGPRs; 128-channel workgroups of 12 waves run one per CU = 168 VGPRs)
  constexpr int NHr = (NR + (64 * NP) / (HC / 4) - 1) / ((64 * NP) / (HC / 4));
  constexpr int SETV = NHr * 4 * (TWO ? 2 : 1);
  // bytes in flight per CU ~ PD x tile bytes x workgroups per CU >= ~64 KB; register sets beyond that only cost occupancy / spills
#ifndef W2S_WIDE_PD_A   // tuning: prefetch depths (A: 128-channel stride 1 forward, B: its stride-2 form, C / D: the 64-channel ones)
#define W2S_WIDE_PD_A 2
#define W2S_WIDE_PD_B 1
#define W2S_WIDE_PD_C 2
#define W2S_WIDE_PD_D 1
#endif
  constexpr int PD = TWO ? (NW >= 8 ? (SETV <= 36 ? 2 : 1) : (SETV <= 12 ? 3 : SETV <= 24 ? 2 : 1))
                         : NW >= 8 ? (SETV <= 36 ? W2S_WIDE_PD_A : SETV <= 68 ? W2S_WIDE_PD_B : 1) : (SETV <= 12 ? 3 : SETV <= 24 ? W2S_WIDE_PD_C : SETV <= 36 ? W2S_WIDE_PD_D : 1);
#ifndef W2S_WIDE_NP
#define W2S_WIDE_NP 1   // tuning: 0 = every instance in the packed form
#endif
  constexpr bool NPK = W2S_WIDE_NP && EPI == W2S_EPI_STATS;
  void (*kern)(WideP);
  if constexpr (NPK) kern = conv_wide_np_kernel<CI, NW, NP, STRIDE, PRO, EPI, MT, PD, UP2, CZ>;
  else kern = conv_wide_kernel<CI, NW, NP, STRIDE, PRO, EPI, MT, PD, UP2, CZ>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return W2S_ELAUNCH;
  const int total = P.B * P.ntiles;
  const int per_cu = (NW >= 8 || UP2) ? 1 : 2;   // (the transposed form needs 166 registers: one 8-wave workgroup per CU)
  const int nwg = 256 * (per_cu > 0 ? per_cu : 1), grid = nwg < total ? nwg : total;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (NW + NP)), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// 1 = this launch is not one of the wide kernel's shapes (the caller falls through to conv_cl_kernel)
static bool wide_shape(const w2s_conv_args& a) {
  if (!a.w_hi || !a.w_lo || (a.mode != W2S_MODE_CONTIG && a.mode != W2S_MODE_UP2) || a.taps != 3 || a.dil != 1 || a.pad < 0 || a.pad > 2) return false;
  if (a.y2 || a.rowkeep || a.bias || a.reserved) return false;
  if ((size_t)a.B * a.cin * 16 > 32 * 1024) return false;   // the per-sample statistics tables live in LDS
  if (a.ldx != a.cin || a.ldy != a.cout || (a.aux && a.ld_aux != a.cout)) return false;
  if (a.cin < 32 || a.cout < 32 || (a.cin < 64 && a.cout < 64)) return false;
  const bool fwd = a.epi == W2S_EPI_STATS && !a.flip && (a.pro == W2S_PRO_GELU || a.pro == W2S_PRO_IN_GELU);
  const bool dgr = a.epi == W2S_EPI_GP && a.flip && a.pro == W2S_PRO_INBWD && a.stride == 1;
  const bool up2 = a.mode == W2S_MODE_UP2 && a.epi == W2S_EPI_GP && a.pro == W2S_PRO_INBWD_GP && a.stride == 2 && (a.pad == 1 || a.pad == 2) &&
                   a.L_out == 2 * a.L_in && !a.add_even && a.cin == a.cout;
  if (a.mode == W2S_MODE_UP2 ? !up2 : (!fwd && !dgr)) return false;
  if (a.pro == W2S_PRO_GELU && a.stride != 1) return false;
  static const char* off = getenv("W2S_NO_WIDE");   // tuning only
  return !off;
}
// dry != 0: only answer which tile an instance would use for this launch (> 0) or that none takes it (1 -> the caller's generic kernel);
// w2s_conv_tile sizes the statistics partials with it.
static int wide_mt() { return 4; }   // 64-position tiles (128 were tried: no gain, more registers)
int w2s_conv_wide_try(const w2s_conv_args& a, hipStream_t s, int dry) {
  if (!wide_shape(a)) return 1;
  const int mt = wide_mt();
#define W2S_WIDE_M(CI_, NW_, ST_, PRO_, EPI_, MT_) \
  if (a.mode == W2S_MODE_CONTIG && a.cin == 16 * CI_ && a.cout == 16 * NW_ && a.stride == ST_ && a.pro == PRO_ && a.epi == EPI_) { \
    if (dry) return 16 * MT_; \
    return launch_wide<CI_, NW_, ST_, PRO_, EPI_, MT_>(a, s); \
  }
#define W2S_WIDE(CI_, NW_, ST_, PRO_, EPI_) W2S_WIDE_M(CI_, NW_, ST_, PRO_, EPI_, 4)
#ifndef W2S_WIDE_MT_D128
#define W2S_WIDE_MT_D128 2   // tuning: m-tiles of the 128 -> 128 stride-1 data gradient
#endif
#ifndef W2S_UP2_MT128
#define W2S_UP2_MT128 4   // tuning: m-tiles of the 128-channel transposed data gradient
#endif
  W2S_WIDE(2, 4, 1, W2S_PRO_GELU, W2S_EPI_STATS) W2S_WIDE(4, 4, 1, W2S_PRO_GELU, W2S_EPI_STATS)
  W2S_WIDE(4, 8, 1, W2S_PRO_GELU, W2S_EPI_STATS) W2S_WIDE(8, 8, 1, W2S_PRO_GELU, W2S_EPI_STATS)
  W2S_WIDE(4, 4, 1, W2S_PRO_IN_GELU, W2S_EPI_STATS) W2S_WIDE(8, 8, 1, W2S_PRO_IN_GELU, W2S_EPI_STATS)
  W2S_WIDE(4, 4, 2, W2S_PRO_IN_GELU, W2S_EPI_STATS) W2S_WIDE(8, 8, 2, W2S_PRO_IN_GELU, W2S_EPI_STATS)
  W2S_WIDE(4, 4, 1, W2S_PRO_INBWD, W2S_EPI_GP) W2S_WIDE_M(8, 8, 1, W2S_PRO_INBWD, W2S_EPI_GP, W2S_WIDE_MT_D128)
  W2S_WIDE(4, 2, 1, W2S_PRO_INBWD, W2S_EPI_GP) W2S_WIDE(8, 4, 1, W2S_PRO_INBWD, W2S_EPI_GP)
#undef W2S_WIDE
#undef W2S_WIDE_M
  if (a.mode == W2S_MODE_UP2) {   // 64-position tiles = 32 gradient rows + 1 (128-position tiles spill)
#ifndef W2S_UP2_MT
#define W2S_UP2_MT 8
#endif
    if (a.cin == 64 && a.pad == 1) { return dry ? 16 * W2S_UP2_MT : launch_wide<4, 4, 1, W2S_PRO_INBWD_GP, W2S_EPI_GP, W2S_UP2_MT, 4, 1, 0>(a, s); }
    if (a.cin == 64 && a.pad == 2) { return dry ? 16 * W2S_UP2_MT : launch_wide<4, 4, 1, W2S_PRO_INBWD_GP, W2S_EPI_GP, W2S_UP2_MT, 4, 1, 1>(a, s); }
    if (a.cin == 128 && a.pad == 1) { return dry ? 16 * W2S_UP2_MT128 : launch_wide<8, 8, 1, W2S_PRO_INBWD_GP, W2S_EPI_GP, W2S_UP2_MT128, 4, 1, 0>(a, s); }
    if (a.cin == 128 && a.pad == 2) { return dry ? 16 * W2S_UP2_MT128 : launch_wide<8, 8, 1, W2S_PRO_INBWD_GP, W2S_EPI_GP, W2S_UP2_MT128, 4, 1, 1>(a, s); }
  }
  return 1;
}
