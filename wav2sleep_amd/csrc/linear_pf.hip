// Row-wise linear layers of the set-fusion transformer as a PERSISTENT, SOFTWARE-PIPELINED split-precision GEMM (round 6):
//
//   y[r][o] = EPI( bias[o] + sum_{k < K} x[r][k] * w[o][k] ),   r < rows (76 800 token rows at batch 16), K = 128 * KC, N = cout = 128 * NB
//
// the transformer's in_proj / out_proj / linear1 / linear2 (nn.TransformerEncoderLayer, models/wav2sleep.py:286-296) and their data gradients,
// with the epilogue fusions of w2s_conv_forward's bias epilogue (residual add + dropout, GELU + dropout into a second output, GELU' x
// dropout of the backward).  Same contract, same products in the same order as conv_cl_kernel<8, 4, *, *, *, 4, NONE, BIAS, 1>: bit-identical
// results (tests/gpu_check.py `linpf`).
//
// Why: conv_cl's one-tile workgroups run load -> wait -> split -> barrier -> K loop -> store per 64 rows and rely on four workgroups per CU
// to overlap each other; these launches are the serial chain between the encoders' forward and backward (one kernel at a time, nothing beside
// it) and ran at 2.4-3.4 TB/s, their waves parked on s_waitcnt / barriers for half of their lifetime (SQ_WAIT_ANY 51 % of SQ_WAVE_CYCLES,
// matrix pipe 16 %: docs/lab_notes_r6.md).  K > 128 (linear2: four 128-wide chunks of a 512-wide row) paid that round trip once per chunk.
// Here a workgroup walks 64-row tiles; the raw fp32 rows of chunk c + 1 are requested (unconditional loads from clamped rows) right after
// chunk c has been split into the OTHER LDS window, so one barrier per chunk suffices and the HBM latency of a chunk hides behind the previous
// chunk's matrix phase; the weight fragments (L2-resident, fragment-major 1 KB runs) are fetched two K steps ahead in one continuous
// pipeline across chunks, n-blocks and tiles (the step sequence of a tile is periodic).  N > 128 (in_proj, linear1): the staged rows serve
// every 128-column block -- x is read once, not once per block.
#include <cstdlib>
#include <type_traits>
#include "w2s_common.h"

struct LinP {
  const float* x; const __bf16* w_hi; const __bf16* w_lo; const float* bias; const float* aux;
  float* y; float* y2;
  int rows, xrow, ldy, ldy2, ld_aux;   // xrow: floats between two rows of x (ldx, or taps * ldx for the K = taps * 128 layouts)
  int ntiles, reserved;
  float drop_p; uint64_t drop_seed;
};

// NB: 128-column blocks of the output (N = 128 NB), KC: 128-wide chunks of a row (K = 128 KC); NB == 1 or KC == 1.
// 8 waves: wave w owns output columns [128 nb + 16 w, + 16) of every block nb for all 64 rows of a tile, and holds ITS weight fragments
// (NB KC x 4 K steps x hi / lo = 32 NB KC registers) for the whole launch: the K loop touches no global memory, so the only vector-memory
// stream of a wave is the activation prefetch and its counted waits never meet an unrelated load (vmcnt is in order: in the first cut of
// this kernel the L2 weight fetches of a K step waited for the HBM prefetch issued before them).
template <int NB, int KC>
__global__ __launch_bounds__(512) void linear_pf_kernel(LinP P) {
  extern __shared__ f32x4 smem4[];
  constexpr int TM = 64, RSE = 128 + 16;                  // rows per tile; bf16 elements per LDS row (+32 B: conflict-free ds_read_b128, conv_cl.inl)
  constexpr int WIN = 2 * TM * RSE;                       // one window: hi plane, lo plane
  constexpr int K32 = KC * 4;                             // K steps of 32 per output row
  __bf16* lds = reinterpret_cast<__bf16*>(smem4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int myc4 = tid & 31, row0 = tid >> 5;             // staging: 32 float4 per 128-wide row, 16 rows per pass, 4 passes

  bf16x8 wh[NB * KC * 4], wl[NB * KC * 4];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int k = 0; k < K32; ++k) {
      const size_t wo = ((size_t)((nb + blockIdx.y) * 8 + wave) * K32 + k) * 512 + lane * 8;
      wh[nb * K32 + k] = *reinterpret_cast<const bf16x8*>(P.w_hi + wo);
      wl[nb * K32 + k] = *reinterpret_cast<const bf16x8*>(P.w_lo + wo);
    }

  // chunk (tile, kc): 64 rows x 128 floats = 4 float4 per thread; the NEXT chunk's raw rows are in flight while this one runs
  f32x4 rx[4];
  auto prefetch = [&](int tile, int kc) {
    const float* xb = P.x + kc * 128 + myc4 * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = min(tile * TM + row0 + 16 * k, P.rows - 1);   // unconditional loads from clamped rows (rows past the end are not stored)
      rx[k] = ld4o(xb, (unsigned)row * (unsigned)P.xrow);
    }
  };
  auto commit = [&](int buf) {
    __bf16* hi = lds + buf * WIN;
    __bf16* lo = hi + TM * RSE;
#pragma unroll
    for (int k = 0; k < 4; ++k) split_store4(hi, lo, (row0 + 16 * k) * RSE + myc4 * 4, rx[k]);
  };

  // accumulators of G column blocks are live at a time (N = 512: block after block over the same staged rows -- 128 weight registers leave
  // room for one block's 16 accumulators beside the rest; N = 384 fits whole)
  constexpr int G = (NB == 4) ? 1 : NB;
  f32x4 acc[G][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < G; ++i)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[i][mt] = (f32x4){0, 0, 0, 0};
  };
  auto kloop = [&](const __bf16* hiL, const __bf16* loL, auto KCI, auto G0) {   // chunk KCI of the row, column blocks G0 .. G0 + G - 1
    constexpr int kc = decltype(KCI)::value, g0 = decltype(G0)::value;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bf16x8 bh[4], bl[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        bh[mt] = *reinterpret_cast<const bf16x8*>(hiL + (mt * 16 + r) * RSE + q * 32 + 8 * g);
        bl[mt] = *reinterpret_cast<const bf16x8*>(loL + (mt * 16 + r) * RSE + q * 32 + 8 * g);
      }
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const bf16x8 ah = wh[(g0 + i) * K32 + kc * 4 + q], al = wl[(g0 + i) * K32 + kc * 4 + q];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[mt], acc[i][mt], 0, 0, 0);
          acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[mt], acc[i][mt], 0, 0, 0);
          acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[mt], acc[i][mt], 0, 0, 0);
        }
      }
    }
  };
  // w2s_conv_forward's bias epilogue with its fusions (conv_cl.inl), column blocks g0 .. g0 + G - 1 of `tile`
  auto epilogue = [&](int tile, int g0) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int ch = (g0 + i + blockIdx.y) * 128 + wave * 16 + 4 * g;
      f32x4 bv = {0, 0, 0, 0};
      if (P.bias) bv = ld4(P.bias + ch);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int pos = tile * TM + mt * 16 + r;
        if (pos >= P.rows) continue;
        f32x4 v = acc[i][mt];
        if (P.bias) v += bv;
        if (P.reserved & (W2S_FUSE_ADD_DROP | W2S_FUSE_GELU_BWD_DROP)) {
          const size_t i0 = (size_t)pos * (size_t)P.ldy + ch;
          f32x4 m = {1, 1, 1, 1};
          if (P.drop_p > 0.f) {
            m.x = w2s_dropscale(P.drop_seed, i0, P.drop_p); m.y = w2s_dropscale(P.drop_seed, i0 + 1, P.drop_p);
            m.z = w2s_dropscale(P.drop_seed, i0 + 2, P.drop_p); m.w = w2s_dropscale(P.drop_seed, i0 + 3, P.drop_p);
          }
          const f32x4 ax = ld4o(P.aux, (unsigned)pos * (unsigned)P.ld_aux + ch);
          v = (P.reserved & W2S_FUSE_ADD_DROP) ? ax + v * m : v * m * gelu_grad4(ax);
        }
        st4o(P.y, (unsigned)pos * (unsigned)P.ldy + ch, v);
        if (P.y2) {
          f32x4 h = gelu4(v);
          if ((P.reserved & W2S_FUSE_Y2_GELU_DROP) && P.drop_p > 0.f) {
            const size_t i0 = (size_t)pos * (size_t)P.ldy2 + ch;
            h.x *= w2s_dropscale(P.drop_seed, i0, P.drop_p); h.y *= w2s_dropscale(P.drop_seed, i0 + 1, P.drop_p);
            h.z *= w2s_dropscale(P.drop_seed, i0 + 2, P.drop_p); h.w *= w2s_dropscale(P.drop_seed, i0 + 3, P.drop_p);
          }
          st4o(P.y2, (unsigned)pos * (unsigned)P.ldy2 + ch, h);
        }
      }
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

  int buf = 0;
  prefetch(blockIdx.x, 0);
  for (int tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x) {
    const int tn = tile + (int)gridDim.x < P.ntiles ? tile + (int)gridDim.x : tile;   // (past the last tile: re-read this one -- cheaper than a branch around the loads)
    // one chunk: commit the prefetched rows to window `buf`, request the next chunk, ONE barrier (window `buf` is complete; every wave is
    // done with the other window: it passed this barrier after reading it)
    auto stage = [&](int next_tile, int next_kc) {
      commit(buf);
      prefetch(next_tile, next_kc);
      __syncthreads();
      const __bf16* hiL = lds + buf * WIN;
      buf ^= 1;
      return hiL;
    };
    if constexpr (KC == 1) {
      const __bf16* hiL = stage(tn, 0);
      const __bf16* loL = hiL + TM * RSE;
      zero_acc(); kloop(hiL, loL, I0{}, I0{}); epilogue(tile, 0);
      if constexpr (NB == 4) {
        zero_acc(); kloop(hiL, loL, I0{}, I1{}); epilogue(tile, 1);
        zero_acc(); kloop(hiL, loL, I0{}, I2{}); epilogue(tile, 2);
        zero_acc(); kloop(hiL, loL, I0{}, I3{}); epilogue(tile, 3);
      }
    } else {
      zero_acc();
      { const __bf16* hiL = stage(tile, 1); kloop(hiL, hiL + TM * RSE, I0{}, I0{}); }
      { const __bf16* hiL = stage(KC > 2 ? tile : tn, KC > 2 ? 2 : 0); kloop(hiL, hiL + TM * RSE, I1{}, I0{}); }
      if constexpr (KC > 2) { const __bf16* hiL = stage(KC > 3 ? tile : tn, KC > 3 ? 3 : 0); kloop(hiL, hiL + TM * RSE, I2{}, I0{}); }
      if constexpr (KC > 3) { const __bf16* hiL = stage(tn, 0); kloop(hiL, hiL + TM * RSE, I3{}, I0{}); }
      epilogue(tile, 0);
    }
  }
}

template <int NB, int KC>
static int launch_linear_pf(const LinP& P, int gy, hipStream_t s) {
  const size_t lds = (size_t)2 * 2 * 64 * (128 + 16) * 2;   // two windows x (hi, lo) x 64 rows x 144 bf16 = 73 728 B
  auto kern = linear_pf_kernel<NB, KC>;
  // (per launch, like every launcher of this library: the attribute belongs to the CURRENT device's copy of the function)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return W2S_ELAUNCH;
  // one 8-wave workgroup per CU (two where 32 NB KC weight registers leave room: the register file sets it); every workgroup the same
  // number of tiles where the count allows it
  const char* gs = getenv("W2S_LINEAR_PF_GRID");   // tuning only
  const int cap = (gs ? atoi(gs) : (NB * KC == 1 ? 512 : 256)) / gy;
  const int per = (P.ntiles + cap - 1) / cap;
  const int grid = (P.ntiles + per - 1) / per;
  hipLaunchKernelGGL(kern, dim3(grid, gy), dim3(512), lds, s, P);
  W2S_CHECK_LAUNCH();
  return W2S_OK;
}

// 1 = not one of this kernel's launches (the caller falls through to conv_cl_kernel); otherwise the launch's return code (dry: 0 = would take it)
int w2s_linear_pf_try(const w2s_conv_args& a, hipStream_t s, int dry) {
  const char* off = getenv("W2S_NO_LINEAR_PF");   // tuning / the bit-equality check of tests/gpu_check.py `linpf` (read per launch on purpose)
  if (off) return 1;
  if (!a.w_hi || !a.w_lo || a.pro != W2S_PRO_NONE || a.epi != W2S_EPI_BIAS || a.rowkeep || (a.reserved & 1) || a.B != 1) return 1;
  if (a.y2 && !(a.reserved & W2S_FUSE_Y2_GELU_DROP)) return 1;
  if (a.cin != 128 || a.cout < 128 || (a.cout & 127) || a.cout > 512 || a.dil > 1 || a.flip) return 1;
  const bool k1 = a.mode == W2S_MODE_CONTIG && a.taps == 1 && a.stride == 1 && a.pad == 0;
  const bool kn = a.mode == W2S_MODE_DILATED && a.taps == a.stride && (a.taps == 3 || a.taps == 4) && a.pad == 0 && a.ldx == 128;
  if (!k1 && !kn) return 1;
  const int KC = a.taps, NB = a.cout / 128;
  if (KC > 1 && NB > 1) return 1;
  if (a.L_out < 256) return 1;   // (tiny launches: nothing to pipeline)
  if ((size_t)a.L_out * (size_t)(KC * a.ldx) * 4 >= ((size_t)1 << 32)) return 1;
  if (dry) return 0;
  LinP P{a.x, static_cast<const __bf16*>(a.w_hi), static_cast<const __bf16*>(a.w_lo), a.bias, a.aux, a.y, a.y2,
         a.L_out, KC * a.ldx, a.ldy, a.ldy2 ? a.ldy2 : a.cout, a.ld_aux ? a.ld_aux : a.cout, (a.L_out + 63) / 64, a.reserved, a.drop_p, a.drop_seed};
  // N > 128: one grid row per 128-column block (32 weight registers per wave: two workgroups per CU; the rows are then staged once per block,
  // from the Infinity Cache after the first -- holding 96-128 weight registers to stage them once spilled)
  if (KC == 1) return launch_linear_pf<1, 1>(P, NB, s);
  if (NB == 1 && KC == 3) return launch_linear_pf<1, 3>(P, 1, s);
  if (NB == 1 && KC == 4) return launch_linear_pf<1, 4>(P, 1, s);
  return 1;
}

// (bookkeeping: does w2s_conv_forward(a) run on linear_pf_kernel?  wav2sleep_amd/lib.py names its timer keys after the kernel rocprofv3 reports)
extern "C" int w2s_linear_pf_takes(const w2s_conv_args* a) { return a ? (w2s_linear_pf_try(*a, nullptr, 1) == 0) : 0; }
