#include "conv_cl.inl"
int w2s_conv_dispatch_12d(const w2s_conv_args& a, hipStream_t s) { return dispatch_tile<1, 2, W2S_MODE_DILATED>(a, s); }
