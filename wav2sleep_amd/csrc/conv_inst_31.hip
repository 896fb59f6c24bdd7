#include "conv_cl.inl"
int w2s_conv_dispatch_31(const w2s_conv_args& a, hipStream_t s) { return dispatch_tile<3, 1, W2S_MODE_CONTIG>(a, s); }

int w2s_conv_tile_impl(int cin, int cout, int taps, int stride, int mode, int B, int L_out) { return cfg_tm(pick_cfg(cin, cout, taps, stride, mode, B, L_out)); }
