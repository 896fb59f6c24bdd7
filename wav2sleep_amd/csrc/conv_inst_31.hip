#include "conv_cl.inl"
int w2s_conv_dispatch_31(const w2s_conv_args& a, hipStream_t s) { return dispatch_tile<3, 1, W2S_MODE_CONTIG>(a, s); }

int w2s_conv_tile_impl(int cin, int cout, int taps, int stride, int mode, int B, int L_out) { return cfg_tm(pick_cfg(cin, cout, taps, stride, mode, B, L_out)); }
void w2s_conv_cfg_impl(int cin, int cout, int taps, int stride, int mode, int B, int L_out, int dil, int* out3) {
  const TileCfg c = pick_cfg(cin, cout, taps, stride, mode, B, L_out, dil);
  out3[0] = c.nt; out3[1] = c.mt; out3[2] = c.wn;
}
