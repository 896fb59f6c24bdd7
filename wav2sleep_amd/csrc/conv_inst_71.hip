#include "conv_cl.inl"
int w2s_conv_dispatch_71(const w2s_conv_args& a, hipStream_t s) { return dispatch_tile<7, 1, W2S_MODE_CONTIG>(a, s); }
