// C-ABI entry of the channels-last implicit-GEMM convolution; kernels live in conv_cl.inl and are
// instantiated per geometry in conv_inst_*.hip (split only to parallelise the build).
#include <cstdlib>
#include "w2s_common.h"

int w2s_conv_dispatch_31(const w2s_conv_args& a, hipStream_t s);
int w2s_conv_dispatch_32(const w2s_conv_args& a, hipStream_t s);
int w2s_conv_dispatch_12(const w2s_conv_args& a, hipStream_t s);
int w2s_conv_dispatch_12d(const w2s_conv_args& a, hipStream_t s);
int w2s_conv_dispatch_11(const w2s_conv_args& a, hipStream_t s);
int w2s_conv_dispatch_71d(const w2s_conv_args& a, hipStream_t s);
int w2s_conv_dispatch_71(const w2s_conv_args& a, hipStream_t s);
int w2s_conv_dispatch_44d(const w2s_conv_args& a, hipStream_t s);
int w2s_conv_dispatch_33d(const w2s_conv_args& a, hipStream_t s);
int w2s_conv_dispatch_up2(const w2s_conv_args& a, hipStream_t s);
int w2s_conv_tile_impl(int cin, int cout, int taps, int stride, int mode, int B, int L_out);
int w2s_conv_wide_try(const w2s_conv_args& a, hipStream_t s, int dry);   // conv_wide.hip: 1 = not a wide-kernel shape
int w2s_linear_pf_try(const w2s_conv_args& a, hipStream_t s, int dry);   // linear_pf.hip: 1 = not one of the transformer's row-wise linears
void w2s_conv_cfg_impl(int cin, int cout, int taps, int stride, int mode, int B, int L_out, int dil, int* out3);

// (NT, MT, WN) template arguments and the effective MODE of the conv_cl_kernel instance w2s_conv_forward(a) launches (profiling keys)
extern "C" int w2s_conv_cfg(const w2s_conv_args* a, int* out4) {
  if (!a || !out4) return W2S_EINVAL;
  int mode = a->mode, dil = a->dil > 0 ? a->dil : 1;
  if (mode == W2S_MODE_DILATED && a->taps == 7 && a->stride == 1 && (size_t)(64 + 6 * dil) * (a->cin + 16) * 4 <= 150 * 1024)
    mode = W2S_MODE_CONTIG;   // single staging of the whole dilated window (w2s_conv_forward)
  if (mode == W2S_MODE_CONTIG && a->taps == 1 && a->stride == 2 && a->cin >= 32) mode = W2S_MODE_DILATED;   // (w2s_conv_forward)
  w2s_conv_cfg_impl(a->cin, a->cout, a->taps, a->stride, mode, a->B, a->L_out, mode == W2S_MODE_CONTIG ? dil : 1, out4);
  out4[3] = mode;
  return W2S_OK;
}

// positions per workgroup tile of the kernel that w2s_conv_forward(a) will launch (`a` filled as for the launch; y / part may be NULL)
extern "C" int w2s_conv_tile(const w2s_conv_args* a) {
  { const int t = w2s_conv_wide_try(*a, nullptr, 1); if (t > 1) return t; }
  return w2s_conv_tile_impl(a->cin, a->cout, a->taps, a->stride, a->mode, a->B, a->L_out);
}

extern "C" int w2s_conv_forward(const w2s_conv_args* ap, void* stream) {
  if (!ap) return W2S_EINVAL;
  const w2s_conv_args& a = *ap;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a.cin < 16 || a.cin > 128 || (a.cin & (a.cin - 1)) || (a.cout & 15) || a.B <= 0 || a.L_out <= 0) return W2S_EINVAL;
  if (((a.ldx & 3) && a.pro != W2S_PRO_FIRST) || (a.ldy & 3) || !a.x || !a.w || !a.y) return W2S_EINVAL;
  // lane offsets inside one sample are 32-bit byte offsets (scalar-base addressing): refuse tensors that do not fit
  {
    const size_t lim = (size_t)1 << 32;
    const size_t ldx = a.pro == W2S_PRO_FIRST ? 1 : (size_t)a.ldx;
    if ((size_t)a.L_in * ldx * 4 >= lim || (size_t)a.L_out * (size_t)a.ldy * 4 >= lim) return W2S_EINVAL;
    if (a.y2 && (size_t)a.L_out * (size_t)(a.ldy2 ? a.ldy2 : a.cout) * 4 >= lim) return W2S_EINVAL;
    if (a.aux && (size_t)a.L_out * (size_t)(a.ld_aux ? a.ld_aux : a.cout) * 4 >= lim) return W2S_EINVAL;
  }
  if (a.pro >= W2S_PRO_IN_GELU && !a.pro_stats) return W2S_EINVAL;
  if (a.pro < 0 || a.pro > W2S_PRO_AFFINE_BWD + 4) return W2S_EINVAL;
  if ((a.pro == W2S_PRO_INBWD || a.pro == W2S_PRO_INBWD_GP || a.pro >= W2S_PRO_AFFINE_BWD) && (!a.pro_bstats || !a.x2)) return W2S_EINVAL;
  if (a.pro == W2S_PRO_FIRST && (!a.x2 || a.cin != 16 || a.taps != 3 || a.stride != 1 || a.pad != 1 || a.mode != W2S_MODE_CONTIG)) return W2S_EINVAL;
  if ((a.epi == W2S_EPI_AUX_INGELU_ADD && (!a.aux || !a.aux_stats)) || (a.epi == W2S_EPI_GP && !a.aux)) return W2S_EINVAL;
  if (a.epi < 0 || a.epi > W2S_EPI_AFFINE_PART + 4) return W2S_EINVAL;
  if (a.epi >= W2S_EPI_AFFINE_PART && (!a.aux || !a.aux_stats || !a.part || (a.reserved & 1))) return W2S_EINVAL;   // (sums of the COMPLETE result only)
  // epilogue fusions of the transformer layer (`reserved` bits W2S_FUSE_*): only the bias epilogue implements them, and each reads the
  // operand it names -- refuse instead of faulting on a NULL aux / y2 or silently ignoring a bit
  if (a.reserved & (W2S_FUSE_ADD_DROP | W2S_FUSE_Y2_GELU_DROP | W2S_FUSE_GELU_BWD_DROP)) {
    if (a.epi != W2S_EPI_BIAS) return W2S_EINVAL;
    if ((a.reserved & (W2S_FUSE_ADD_DROP | W2S_FUSE_GELU_BWD_DROP)) && !a.aux) return W2S_EINVAL;
    if ((a.reserved & W2S_FUSE_Y2_GELU_DROP) && !a.y2) return W2S_EINVAL;
  }
  { const int rc = w2s_linear_pf_try(a, s, 0); if (rc != 1) return rc; }   // the transformer's linears: persistent pipelined GEMM (linear_pf.hip)
  if (a.mode == W2S_MODE_UP2) {
    if (a.taps != 3 || a.stride != 2 || (a.pad != 1 && a.pad != 2)) return W2S_EINVAL;  // pad 2 = gradient of the causal-padded conv
    { const int rc = w2s_conv_wide_try(a, s, 0); if (rc != 1) return rc; }   // >= 64 channels: persistent role-split kernel
    return w2s_conv_dispatch_up2(a, s);
  }
  if (a.mode == W2S_MODE_DILATED) {
    if (a.taps == 7 && a.stride == 1) {
      // whole dilated window (64-position tile + 6*dil halo rows, hi/lo planes or fp32) in LDS: single staging
      if (a.dil >= 1 && (size_t)(64 + 6 * a.dil) * (a.cin + 16) * 4 <= 150 * 1024) {
        w2s_conv_args b = a;
        b.mode = W2S_MODE_CONTIG;
        return w2s_conv_dispatch_71(b, s);
      }
      return w2s_conv_dispatch_71d(a, s);
    }
    if (a.taps == 4 && a.stride == 4) return w2s_conv_dispatch_44d(a, s);
    if (a.taps == 3 && a.stride == 3) return w2s_conv_dispatch_33d(a, s);
    return W2S_EINVAL;
  }
  if (a.dil != 1) return W2S_EINVAL;
  if (a.taps == 3) {   // >= 64-channel encoder layers: persistent kernel with register-resident weights (conv_wide.hip)
    const int rc = w2s_conv_wide_try(a, s, 0);
    if (rc != 1) return rc;
  }
  if (a.taps == 3 && a.stride == 1) return w2s_conv_dispatch_31(a, s);
  if (a.taps == 3 && a.stride == 2) return w2s_conv_dispatch_32(a, s);
  if (a.taps == 1 && a.stride == 2) {
    // the 1x1 / stride-2 residual conv reads every other input row: stage exactly those (the per-tap window form: TM rows at row stride 2)
    // instead of the contiguous 2 TM - 1 row window -- half the loads and half the on-load GELU work, and for >= 32 channels (rows of whole
    // cache lines) half the HBM traffic of that operand
    if (a.cin >= 32) {   // (16 channels: 64-B rows share their cache line with the skipped row, and the two-taps-per-K-step
      w2s_conv_args b = a;          //  split-precision packing of that width exists for the contiguous window only)
      b.mode = W2S_MODE_DILATED;
      return w2s_conv_dispatch_12d(b, s);
    }
    return w2s_conv_dispatch_12(a, s);
  }
  if (a.taps == 1 && a.stride == 1) return w2s_conv_dispatch_11(a, s);
  return W2S_EINVAL;
}
