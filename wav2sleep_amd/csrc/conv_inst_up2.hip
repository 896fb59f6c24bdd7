#include "conv_cl.inl"
int w2s_conv_dispatch_up2(const w2s_conv_args& a, hipStream_t s) { return dispatch_tile<3, 2, W2S_MODE_UP2>(a, s); }
