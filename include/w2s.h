/*
 * w2s.h -- C ABI of libw2s_hip.so, the MI355X (gfx950) implementation of the wav2sleep hot path.
 *
 * The reference (joncarter1/wav2sleep) is 100 % Python on stock PyTorch ATen ops; it has no FFI.  The
 * boundary this library replaces is therefore "the ATen ops dispatched by Wav2Sleep.forward / the train
 * step" (SURVEY.md 2.1 + 8a).  Each entry point below names the reference call site(s) it replaces.
 * The Python host (wav2sleep_amd/lib.py) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch types; all tensors fp32 device memory owned by the CALLER,
 *     16-byte aligned, ACTIVATIONS ARE CHANNELS-LAST: [B][L][C] (row = one time position, C contiguous);
 *   - asynchronous on `stream` (a hipStream_t passed as void*), no internal sync, no allocation, no global
 *     state => re-entrant across streams/devices and capturable into a hipGraph;
 *   - return 0 on success, negative W2S_E* on error (never throws, never aborts);
 *   - ONE SAMPLE's tensor (L * ld * 4 bytes) must stay below 4 GiB: lanes address inside a sample with 32-bit byte offsets from a
 *     wave-uniform 64-bit base (W2S_EINVAL otherwise).  An 8-h recording at 128 Hz x 16 channels is 252 MB.
 */
#ifndef W2S_H
#define W2S_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define W2S_OK 0
#define W2S_EINVAL (-1)     /* bad argument / unsupported shape */
#define W2S_ELAUNCH (-2)    /* hipLaunch failure */

/* ---- prologue (transform applied to the input operand while it is staged into LDS) ---- */
#define W2S_PRO_NONE 0
#define W2S_PRO_SANITIZE 1  /* inf -> 0                 models/wav2sleep.py:151 */
#define W2S_PRO_GELU 2      /* GELU(x)                  blocks.py:70 (block output is stored pre-activation) */
#define W2S_PRO_IN_GELU 3   /* GELU((x-mean)*rstd)      blocks.py:183-184 InstanceNorm1d(eps=1e-2)+GELU */
#define W2S_PRO_INBWD 4     /* g_y = rstd*(g - s1 - n*s2), n=(x2-mean)*rstd   (instance-norm backward) */
#define W2S_PRO_INBWD_GP 5  /* as 4 with g := g*GELU'(n) first */
#define W2S_PRO_FIRST 6     /* x = the raw 1-channel signal [B][L_in], x2 = block 0's conv1 weight [16][3]: the 16-channel conv1 output is
                              * RE-COMPUTED on load (3 FMAs / element; inf -> 0 as W2S_PRO_SANITIZE) and then normalised + GELU'd as mode 3.
                              * The largest tensor of the model (16 x T per recording) is then never written or read.  cin must be 16. */
#define W2S_PRO_AFFINE 7    /* act(x * scale + shift), act = pro - 7 in {0 linear, 1 ReLU, 2 LeakyReLU(0.01), 3 GELU, 4 SiLU} (W2S_PRO_AFFINE + act);
                              * pro_stats / x_stats hold (scale, shift) per (b, c) instead of (mean, rstd): any per-channel norm + activation of
                              * the generic path (BatchNorm, GroupNorm, instance, none) applied on load, so that its output is never written
                              * (ConvLayer1D.forward, blocks.py:183-184, for the NEXT layer's conv and that conv's weight gradient; round 6) */
#define W2S_PRO_AFFINE_BWD 12 /* + act: the BACKWARD of that norm + activation on load, two operands as the INBWD modes (x = g, x2 = y):
                              * z = y * scale + shift, gy = (scale g) act'(z) + z c + d with (scale, shift) in pro_stats / g_stats and (c, d) in
                              * pro_bstats / g_bstats per (b, c) (w2s_norm_bwd_coef writes them): the gradient of the conv output y is never
                              * written; its data-gradient conv and weight gradient form it while staging */

/* ---- epilogue (applied to the accumulator tile before the store) ---- */
#define W2S_EPI_PLAIN 0
#define W2S_EPI_STATS 1     /* store + per-(b,c) partial sum / sum-of-squares (instance-norm statistics) */
#define W2S_EPI_AUX_INGELU_ADD 2 /* v += GELU(IN(aux))    blocks.py:68-69 residual join, stored pre-activation */
#define W2S_EPI_BIAS 3      /* v += bias[c]; optional y2 = GELU(v) */
#define W2S_EPI_GP 4        /* v = (v [+ add_even[t/2] if t even]) * GELU'(n(aux)); partial sums of v and v*n */
#define W2S_EPI_AFFINE_PART 5 /* + act (generic path): store v unchanged; partial sums of ga = v * act'(aux * scale + shift) and ga * aux per (b, tile, c),
                              * aux = the raw conv output of the layer whose activated output this launch's result is the gradient of, aux_stats =
                              * (scale, shift) per (b, c): the reduction pass of that layer's norm backward rides in this data-gradient launch */

/* ---- geometry modes ---- */
#define W2S_MODE_CONTIG 0   /* dilation 1: one staged window serves all taps */
#define W2S_MODE_DILATED 1  /* one staged window per tap: dilated convs (7,1) and the taps=4/stride=4 linears */
#define W2S_MODE_UP2 2      /* transposed stride-2, 3 taps, pad 1 (data gradient of conv3) */

/*
 * Generic channels-last 1-D convolution as an implicit GEMM on the fp32 matrix cores.
 *   y[b,t,o] = EPI( sum_{j<taps} sum_{c<cin} w[o][j][c] * PRO(x[b, t*stride + roff(j)*dil - pad, c]) )
 * roff(j) = flip ? taps-1-j : j.  Rows outside [0,L_in) contribute zero (zero padding of the TRANSFORMED
 * operand, as torch's Conv1d pads the activation).  Weight layout is [cout][taps][cin] (see w2s_repack).
 * Replaces: F.conv1d in ConvLayer1D.forward (blocks.py:174), ConvBlock1D.downsample (blocks.py:68),
 * nn.Linear in SignalEncoder.forward (wav2sleep.py:264, as taps=4/stride=4 over the [B,4S,C] map),
 * the Linear/in_proj/out_proj GEMMs of nn.TransformerEncoderLayer (wav2sleep.py:286-296), the dilated
 * convs of DilatedConvBlock (blocks.py:95-108) -- and their data gradients (flip / UP2 modes).
 */
typedef struct w2s_conv_args {
  const float* x;          /* [B][L_in][ldx]  staged operand (or gradient g for PRO_INBWD*) */
  const float* x2;         /* PRO_INBWD*: pre-norm tensor y, same geometry as x */
  const float* w;          /* [cout][taps][cin] */
  float* y;                /* [B][L_out][ldy] */
  float* y2;               /* optional second output (GELU(v)) for EPI_BIAS, [B][L_out][ldy2] */
  const float* pro_stats;  /* [B][cin][2] (mean, rstd) */
  const float* pro_bstats; /* [B][cin][2] (s1, s2) */
  const float* aux;        /* epilogue aux tensor [B][L_out][ld_aux] */
  const float* aux_stats;  /* [B][cout][2] (mean, rstd) or NULL (then n = aux) */
  const float* add_even;   /* EPI_GP: [B][L_out/2][cout] added at even t, or NULL */
  const float* bias;       /* [cout] */
  const float* rowkeep;    /* [B] multiplier on the stored value (0 for a missing modality) or NULL */
  float* part;             /* EPI_STATS / EPI_GP: [B][ntiles][2][cout] partial sums, or NULL */
  const void* w_hi;        /* optional bf16 planes of w (w = hi + lo, layout as w; see w2s_repack_bf16): enables the split-  */
  const void* w_lo;        /* precision matrix-core path for cin >= 32 and cout >= 64; NULL = exact fp32 MFMA               */
  int32_t B, L_in, L_out, cin, cout, taps, stride, dil, pad, flip, mode;
  int32_t ldx, ldy, ldy2, ld_aux;
  int32_t pro, epi;
  int32_t reserved;        /* bit 0: y += result instead of y = result (a contraction split over several launches: cin > 128) */
                           /* EPI_BIAS fusions of the set-fusion transformer (TransformerEncoderLayer, wav2sleep.py:286-296; W2S_FUSE_*):      */
                           /* bit 1: y = aux + drop(result)            (residual add + dropout: x + Dropout(sublayer(x)))                       */
                           /* bit 2: y2 = drop(GELU(y)) instead of GELU(y)   (linear1 -> activation -> dropout; y keeps the pre-activation)     */
                           /* bit 3: y = result * GELU'(aux) * dropmask      (backward of bit 2 in the data-gradient GEMM of linear2)           */
  float drop_p;            /* dropout probability of those fusions (0: no dropout) and the mask seed: the mask of element i of the [rows][ld]    */
  uint64_t drop_seed;      /* output is keep / (1 - p) with the counter-based generator of w2s_eltwise's dropout modes at (drop_seed, i)            */
} w2s_conv_args;
#define W2S_FUSE_ADD_DROP 2
#define W2S_FUSE_Y2_GELU_DROP 4
#define W2S_FUSE_GELU_BWD_DROP 8

/* positions per workgroup tile for (cin,cout); ntiles = ceil(L_out / tile) sizes `part`. */
int w2s_conv_tile(const w2s_conv_args* a);
int w2s_conv_forward(const w2s_conv_args* a, void* stream);
/* out4 = (NT, MT, WN, effective MODE) of the generic kernel instance that launch would use (profiling / bookkeeping only) */
int w2s_conv_cfg(const w2s_conv_args* a, int32_t* out4);
/* 1: w2s_conv_forward(a) runs on the persistent pipelined GEMM of the transformer's row-wise linears (csrc/linear_pf.hip: cin 128 per chunk,
 * cout 128..512, no on-load transform, bias epilogue with its W2S_FUSE_* fusions; bit-identical to the generic kernel) -- bookkeeping only */
int w2s_linear_pf_takes(const w2s_conv_args* a);

/*
 * SequenceCNN's dilated convolution with its channel LayerNorm fused into the epilogue (csrc/seq_conv.hip), 128 -> 128 channels, 7 taps:
 *   DilatedConvBlock.forward (models/blocks.py:115-126: Conv1d(k 7, dilation d, bias False) -> ConvLayerNorm (models/utils.py:9-23) -> GELU)
 *   and its autograd backward.  v[b,t,:] = sum_j W_j x[b, t + roff(j) dil - pad, :], roff(j) = flip ? 6 - j : j, zero padding per sample.
 *   mode 0: y = v.
 *   mode 1: y = v (pre-norm, kept for backward), out = GELU(gamma * LN_c(v) + beta), rs[b*S + t] = (mean_c, rstd_c)   [eps inside the sqrt]
 *   mode 2: v is the gradient w.r.t. the LOWER layer's GELU(LN(.)) output (x = the gradient w.r.t. this layer's conv output, flip = 1, the
 *           [cin][7][cout] packing); out = the gradient w.r.t. the lower layer's conv output yl (LayerNorm + GELU backward with that layer's
 *           gamma / beta / rs), part[b * ntiles + tile][2][128] = per-tile sums over positions of (gn * xhat, gn), gn = v * GELU'(gamma xhat + beta):
 *           the lower LayerNorm's weight / bias gradients (ntiles = ceil(S / 64)); v itself is not stored (y unused).
 * w_hi / w_lo: fragment-major bf16 planes of the packed weight (w2s_repack_bf16).  Split precision only (no exact-fp32 form: that mode runs the
 * generic conv + w2s_layernorm_*).  Returns EINVAL for dil > 32 (the window must fit LDS).
 */
typedef struct w2s_seq_conv_args {
  const float* x;
  const void* w_hi; const void* w_lo;
  float* y; float* out; float* rs;
  const float* gamma; const float* beta; const float* yl;
  float* part;
  int32_t B, S, ldx, dil, pad, flip, mode;
  float eps;
} w2s_seq_conv_args;
int w2s_seq_conv(const w2s_seq_conv_args* a, void* stream);

/*
 * Weight gradient: dW[o][j][c] = sum_{b,t} GY(g)[b,t,o] * H(x)[b, t*stride + j*dil - pad, c]
 * GY/H are the same on-load transforms as above (pro_g on the gradient side, pro_h on the input side).
 * Writes per-workgroup partial slabs; w2s_wgrad_reduce sums them deterministically into the torch-layout
 * gradient.  Replaces the weight half of aten::convolution_backward / addmm backward.
 */
typedef struct w2s_wgrad_args {
  const float* g;  const float* g2;        /* gradient side: g (and pre-norm y for INBWD*) [B][L_out][cout] */
  const float* g_stats; const float* g_bstats;
  const float* x;                          /* input side [B][L_in][cin] */
  const float* x_stats;
  float* slab;                             /* nslab raw-fragment slabs of cout*taps*cin floats */
  int32_t B, L_in, L_out, cin, cout, taps, stride, dil, pad;
  int32_t ldg, ldx;
  int32_t pro_g, pro_h;
  int32_t nslab;                           /* number of partial slabs = grid.x * w2s_wgrad_slabs_per_block() */
  int32_t split_precision;                 /* != 0: bf16x3 matrix-core path where available (cin, cout >= 64) */
} w2s_wgrad_args;
int w2s_wgrad(const w2s_wgrad_args* a, void* stream);
int w2s_wgrad_max_blocks(const w2s_wgrad_args* a);   /* grid.x (= nslab / slabs_per_block) the caller should not exceed for this launch (nslab ignored) */
int w2s_wgrad_slabs_per_block_of(const w2s_wgrad_args* a); /* slabs one grid.x block of THIS launch writes: nslab = grid.x * this */
int w2s_wgrad_slabs_per_block(int cin, int cout, int taps, int dil); /* slabs written per grid.x block: nslab = grid.x * this */
int w2s_wgrad_grid_y(int cin, int cout, int taps, int dil); /* grid.y of w2s_wgrad: slab floats = nslab*cout*cin*taps */
/* grad (+)= sum_s slab[s]; layout 0: grad[o][c][j] (torch Conv1d), 1: grad[o][j][c] (Linear over the flattened taps) */
int w2s_wgrad_reduce(const float* slab, int nslab, float* grad, int cout, int cin, int taps, int dil, int accumulate, int layout, void* stream);
/* the reductions of many layers in one launch per 48 jobs (jobs: HOST array); jobs of one call must target distinct grads */
typedef struct w2s_reduce_job {
  const float* slab; float* grad;
  int32_t nslab, cout, cin, taps, dil, accumulate, layout, reserved;
} w2s_reduce_job;
int w2s_wgrad_reduce_batch(const w2s_reduce_job* jobs, int njobs, void* stream);

/* [cout][cin][taps] (torch) -> fwd pack [cout][taps][cin] and/or bwd pack [cin][taps][cout]; either dst may be NULL */
int w2s_repack(const float* w, float* fwd, float* bwd, int cout, int cin, int taps, void* stream);
/* the same for many layers at once (jobs: HOST array; any of the output pointers may be NULL, bf16 planes in pairs): the
 * ~130 per-layer repacks after an optimiser step become 3 launches */
typedef struct w2s_repack_job {
  const float* w; float* fwd; float* bwd; void* fwd_hi; void* fwd_lo; void* bwd_hi; void* bwd_lo;
  int32_t cout, cin, taps, reserved;
} w2s_repack_job;
int w2s_repack_batch(const w2s_repack_job* jobs, int njobs, void* stream);
/* the same two GEMM operands (rows o x K = taps*cin, rows c x K = taps*cout) as bf16 (hi, lo) planes with w = hi + lo, stored
 * fragment-major: element (row, k) at (((row/16)*(K/32) + k/32)*64 + ((k%32)/8)*16 + row%16)*8 + k%8, so that one wave fetch of a
 * 16x32 MFMA operand is a contiguous 1 KB run (cin == 16, forward planes only: K is padded to 32*ceil(taps/2) and the caller
 * zero-fills the planes once -- two taps share one K step); operands of the split-precision ("bf16x3") path; planes may be NULL in pairs */
int w2s_repack_bf16(const float* w, void* fwd_hi, void* fwd_lo, void* bwd_hi, void* bwd_lo, int cout, int cin, int taps, void* stream);

/*
 * Forward k=3 (pad 1, stride 1 or 2) encoder conv + instance-norm statistics partials for the <= 32-channel layers as a persistent
 * split-precision kernel (next tile prefetched into registers, weights LDS-resident): same result contract as w2s_conv_forward with
 * pro in {W2S_PRO_GELU, W2S_PRO_IN_GELU, W2S_PRO_FIRST} and W2S_EPI_STATS.  x: [B][L_in][cin] (pro FIRST: the raw signal [B][L_in],
 * w1 = block 0's conv1 weight), w: [cout][3][cin] (w2s_repack forward layout), st_in: [B][cin][2] (mean, rstd) unless pro == GELU,
 * y: [B][L_out][cout], part: [B][ceil(L_out/tile)][2][cout] with tile = w2s_conv_fwd_fused_tile (0 = combination not covered),
 * nwg = workgroups to launch (each takes a contiguous run of the (sample, tile) list).  Replaces aten::convolution + the statistics half of native_batch_norm
 * (blocks.py:173-186).
 */
int w2s_conv_fwd_fused_tile(int cin, int cout, int stride);
int w2s_conv_fwd_fused(const float* x, const float* w, const float* st_in, const float* w1, float* y, float* part, int B, int L_in,
                       int L_out, int cin, int cout, int stride, int pad, int pro, int nwg, void* stream);   /* pad: 1 = symmetric, 2 = causal (left pad k-1, blocks.py:150-152) */

/*
 * Fused backward of one encoder ConvLayer1D (k=3, pad=1, stride 1 or 2) for the bandwidth-bound <=32-channel layers:
 * data gradient + weight gradient from one pass over (g, y_k, y_{k-1}).  cg = channels of the gradient side (the
 * forward conv's cout), ch = channels of the input side (its cin); supported (cg,ch): (16,16) (32,16) (32,32).
 * gout[t][c] = (W^T gy [+ add_even[t/2] at even t]) * GELU'(n_in);  part: [B][ceil(Lh/tile)][2][ch] sums of gout, gout*n_in;
 * slab: nslab (= grid size) raw-fragment slabs of cg*3*ch floats -> w2s_wgrad_reduce(slab, nslab, grad, cg, ch, 3, 1, ...).
 * split_precision != 0: both products run as bf16x3 (hi/lo planes in LDS, fp32 accumulate).
 * Replaces aten::convolution_backward + native_batch_norm_backward + gelu_backward of blocks.py:173-186.
 */
int w2s_bwd_fused_tile(int cg, int ch, int stride, int rd, int split_precision);   /* rd: the residual-fold form (gpre != NULL); split_precision: the
 * split-precision kernels' tiles are two positions short of 256 / 128 (254 / 126: their staged windows are whole passes) */
int w2s_bwd_fused(const float* g, const float* y, const float* st_k, const float* bst_k, int pro, const float* xin,
                  const float* st_in, const float* add_even, const float* wb, float* gout, float* part, float* slab, int nslab,
                  int B, int Lg, int Lh, int cg, int ch, int stride, int pad, int split_precision, const float* gpre, const float* wd,
                  float* slab_d, const float* w1, const float* y3p, const float* st3p, void* stream);
/* pad: the forward conv's left padding: 1 = symmetric, 2 = causal (blocks.py:150-152; split_precision only) */
/* gpre != NULL (conv1 of a residual block; stride 1, split_precision, add_even NULL, w2s_bwd_fused_folds_residual(cg, ch)): the
 * block's 1x1/stride-2 residual branch (blocks.py:44-47,68) is folded in -- gout additionally receives Wd^T gpre[t/2] at even t
 * before the GELU' factor (gpre: [B][Lh/2][cg] = dL/d(block pre-activation), wd: [ch][cg]) and slab_d receives nslab raw-fragment
 * slabs of the downsample weight gradient -> w2s_wgrad_reduce(slab_d, nslab, grad_wd, cg, ch, 1, 1, ...). */
int w2s_bwd_fused_folds_residual(int cg, int ch);
/* y3p != NULL (only with gpre): additionally fold the PREVIOUS block's conv3-backward pre-pass (w2s_gp_stats) in: y3p = that block's
 * pre-norm conv3 output [B][Lh][ch], st3p = its (mean, rstd) [B][ch][2]; `part` then holds the partial sums of gout*GELU'(n3) and
 * gout*GELU'(n3)*n3, n3 = IN(y3p) -- the tensor gout is read by w2s_gp_stats otherwise. */
/* w1 != NULL (conv2 of block 0; cg = ch = 16, stride 1, split_precision, st_in given): xin is the RAW 1-channel signal [B][Lh] and
 * the conv's input (block 0's conv1 output) is recomputed from it with w1 = conv1 weight [16][3] (the W2S_PRO_FIRST flow). */

/*
 * fp16 gradient chain (round 3).  Autograd's inter-layer gradient tensors of the encoder (the `grad_output`s that
 * aten::convolution_backward / native_batch_norm_backward / gelu_backward hand to one another, blocks.py:173-186) are fp32 in the
 * reference.  The *_h entry points store the ones that live between two fused-backward launches as fp16 with one power-of-two scale per
 * tensor (fp32 accumulation and arithmetic throughout).  Every such tensor has a header of two floats: hdr[0] = scale (stored = true *
 * scale), hdr[1] = max |true value| (float bits, accumulated by the producer with an integer atomic maximum; ZERO it before the producer
 * runs).  A producer chooses its output scale from the maxima of the tensors it reads.
 *   gmode 0: g / gpre / gout fp32 (= w2s_bwd_fused);  1: g fp32 with header hdr_g = {1, max} (from w2s_gp_stats_h), gout fp16;
 *   2: g, gpre (headers hdr_g, hdr_p) and gout fp16.  hdr_o: header of gout.  Split-precision kernels only; shapes (16,16), (32,32),
 *   the residual-fold forms (16,16), (32,16) and the first-layer form; gmode 1 only for the stride-2 conv3.
 */
int w2s_bwd_fused_h(const void* g, const float* y, const float* st_k, const float* bst_k, int pro, const float* xin, const float* st_in,
                    const float* add_even, const float* wb, void* gout, float* part, float* slab, int nslab, int B, int Lg, int Lh, int cg,
                    int ch, int stride, int pad, const void* gpre, const float* wd, float* slab_d, const float* w1, const float* y3p,
                    const float* st3p, int gmode, const float* hdr_g, const float* hdr_p, float* hdr_o, void* stream);
/* w2s_gp_stats on an fp16 gradient (g_half != 0, hdr_g its header), or on an fp32 one while publishing its header (hdr_amax != NULL:
 * {1, max |g|}) -- the entry of the chain */
int w2s_gp_stats_h(const void* g, int g_half, const float* hdr_g, float* hdr_amax, const float* y, const float* stats, float* part, int B, int L,
                   int C, int tile, void* stream);
/* w2s_enc_first_bwd with gn1 / gpre stored as fp16 (headers hdr_n / hdr_p) */
int w2s_enc_first_bwd_h(const float* x, const void* gn1, const float* hdr_n, const float* y1, const float* stats1, const float* bstats1,
                        const void* gpre, const float* hdr_p, float* slab, int nslab, int B, int L, int cout, const float* w1, int causal,
                        void* stream);
/* Block 0's conv1 weight gradient without its gradient tensor.  w2s_bwd_fused_w1 = w2s_bwd_fused for conv2 of block 0 in the first-layer
 * recompute form (x = raw signal [B][L], w1 = conv1 weight [16][3], 16 -> 16 channels, stride 1, split precision) that ALSO leaves
 * part_w1 [B][ceil(L / w2s_bwd_fused_tile(16,16,1,0,1))][16][3] = per-tile sums of gout[t][o] * xs[t + j - pad]; gout may be NULL (the folded
 * sums were its only reader: 1 GB per 1024-samples-per-epoch signal at batch 16 neither written nor read back).  w2s_enc_first_wgrad turns
 * the partials into out [B][48] = each sample's contribution to dW1[o][j] (instance-norm backward applied through the signal's nine
 * moments xmom [B][ntx][9] from w2s_enc_first_stats; stats1 / bstats1 [B][16][2] as for w2s_enc_first_bwd); sum over B with
 * w2s_colsum_batch.  w2s_enc_first_dwd = the downsample
 * weight gradient alone: slab [nslab][16].  Replaces trainer-side autograd of models/wav2sleep.py:96-110 (block 0). */
int w2s_bwd_fused_w1(const float* g, const float* y, const float* st_k, const float* bst_k, const float* x, const float* st_in,
                     const float* wb, float* gout, float* part, float* part_w1, float* slab, int nslab, int B, int L, int pad,
                     const float* w1, void* stream);
int w2s_enc_first_wgrad(const float* xmom, int ntx, const float* w1, const float* part_w1, const float* stats1, const float* bstats1,
                        float* out, int B, int ntiles, void* stream);
/* w2s_enc_first_fwd's statistics-only form (y == NULL) that also keeps the nine raw moments of every signal tile:
 * xmom [B][ceil(L / tile)][9] (may be NULL) -- the sums w2s_enc_first_wgrad needs */
int w2s_enc_first_stats(const float* x, const float* w, float* part, float* xmom, int B, int L, int tile, int causal, void* stream);
int w2s_enc_first_dwd(const float* x, const float* gpre, float* slab, int nslab, int B, int L, void* stream);
/* ... or folded as well: w2s_bwd_fused_wd = w2s_bwd_fused for conv1 of block 1 in its residual-fold form (16 -> 16; gpre / wd / slab_d /
 * y3p / st3p as there) whose gout IS block 0's gpre; part_wd [nslab][16] = per-workgroup sums of gout[u][o] * san(x0[2u]), x0 = the raw
 * signal [B][2 L].  Sum the rows with w2s_colsum_batch; w2s_enc_first_dwd is then not needed. */
int w2s_bwd_fused_wd(const float* g, const float* y, const float* st_k, const float* bst_k, const float* xin, const float* wb, float* gout,
                     float* part, float* slab, int nslab, int B, int L, int pad, const float* gpre, const float* wd, float* slab_d,
                     const float* y3p, const float* st3p, const float* x0, float* part_wd, void* stream);

/*
 * Fused backward of one k=3 encoder conv (symmetric or causal padding) with cg = 64 gradient-side channels (round 3, csrc/bwd_wide.hip): data
 * gradient + weight gradient + GELU' + backward statistics in one persistent role-split pass -- the >= 64-channel counterpart of
 * w2s_bwd_fused (same formulas).  stride 1: ch = 64 or 32 input-side channels, gy = instance-norm backward of g with (st_k, bst_k), h =
 * GELU(IN(xin)) with st_in or GELU(xin) when st_in == NULL.  stride 2 (the block's conv3; ch = 64, st_in required): g = dL/d(block
 * pre-activation) [B][L/2][cg], gy = instance-norm backward of g * GELU'(n_k).  L = input-side length.  w_hi / w_lo: the data-gradient
 * operand planes of the conv weight (w2s_repack_batch bwd_hi / bwd_lo).  part: [B][ceil(L/tile)][groups][2][ch] partial sums of gout and
 * gout*n_in (tile / groups from the two queries below), or NULL.  slab: nslab (= grid size, <= B*ceil(L/tile)) raw-fragment slabs of
 * cg*3*ch floats -> w2s_wgrad_reduce(slab, nslab, grad, cg, ch, 3, 1, ...).  Returns 1 when no instance takes the launch (dry != 0: only
 * that answer, nothing is launched; st_in then only says whether the input side carries statistics).
 * Replaces aten::convolution_backward + native_batch_norm_backward + gelu_backward of blocks.py:173-186 for those layers.
 */
int w2s_bwd_wide_tile(int cg, int ch, int stride);     /* input-side positions per tile (0: no instance) */
int w2s_bwd_wide_groups(int cg, int ch, int stride);   /* statistics-partial rows per tile */
int w2s_bwd_wide(const float* g, const float* y, const float* st_k, const float* bst_k, const float* xin, const float* st_in,
                 const float* add_even, const void* w_hi, const void* w_lo, float* gout, float* part, float* slab, int nslab, int B, int L,
                 int cg, int ch, int stride, const float* y3p, const float* st3p, const float* gpre, const void* wd_hi, const void* wd_lo,
                 float* slab_d, int pad, int dry, void* stream);   /* pad: the forward conv's left padding (1 symmetric, 2 causal) */
/* gpre != NULL (conv1 of a residual block: stride 1, st_in and add_even NULL, L even): the block's 1x1/stride-2 residual branch
 * (blocks.py:44-47,68) folded in as in w2s_bwd_fused -- gout additionally receives Wd^T gpre[t/2] at even t before the GELU' factor (gpre:
 * [B][L/2][cg], wd_hi / wd_lo: w2s_repack_batch bwd planes of the downsample weight) and slab_d receives nslab raw-fragment slabs of the
 * downsample weight gradient -> w2s_wgrad_reduce(slab_d, nslab, grad_wd, cg, ch, 1, 1, ...). */
/* y3p != NULL (stride 1, part given): additionally fold the PREVIOUS block's conv3-backward pre-pass (w2s_gp_stats) in, as w2s_bwd_fused
 * does: y3p = that block's pre-norm conv3 output [B][L][ch], st3p = its (mean, rstd) [B][ch][2]; `part` then holds the partial sums of
 * gout*GELU'(n3) and gout*GELU'(n3)*n3, n3 = IN(y3p). */

/* Statistics leave every producer above as per-tile fp32 partial sums `part` (rounds 1 and 4 also built finalisation INSIDE the producers:
 * correct, 168 launches fewer per step and slower every time it was measured -- removed in round 5, docs/lab_notes_r5.md).
 * w2s_stats_finalize: partial sums [B][ntiles][2][C] -> per-(b,c) pairs [B][C][2]: kind 0 = (mean, rstd) with biased variance + eps
 * (nn.InstanceNorm1d, models/utils.py:89-92), kind 1 = (sum1, sum2)/count.  fp64 accumulation, fixed order. */
int w2s_stats_finalize(const float* part, int B, int ntiles, int C, long count, float eps, int kind, float* out, void* stream);

/* First encoder layer, Cin = 1 (blocks.py:46, conv1 of block 0): y[b,t,o] = sum_j w[o][j]*san(x[b,t+j-1]);
 * part [B][ceil(L/tile)][2][16] = partial sum / sum-of-squares.  w is the torch tensor [16][1][3]. */
/* y may be NULL (statistics only: the W2S_PRO_FIRST consumers recompute the values).
 * causal != 0: the causal padding of ConvLayer1D (blocks.py:150-152,178-182), taps x[t-2], x[t-1], x[t]; needs y. */
int w2s_enc_first_fwd(const float* x, const float* w, float* y, float* part, int B, int L, int cout, int tile, int causal, void* stream);
/* Block-0 residual join (blocks.py:67-69): pre[b,u,o] = GELU(IN(y3[b,u,o])) + wd[o]*san(x[b,2u]) */
int w2s_enc_first_join(const float* x, const float* wd, const float* y3, const float* stats3, float* pre, int B, int L, int cout, void* stream);
/* weight grads of block-0 conv1 / downsample; slab[nslab][64] = {dW1[o][j] (48), dWd[o] (16)}; sum with w2s_colsum */
int w2s_enc_first_bwd(const float* x, const float* gn1, const float* y1, const float* stats1, const float* bstats1,
                      const float* gpre, float* slab, int nslab, int B, int L, int cout, const float* w1, int causal, void* stream);
/* y1 == NULL: the conv1 output is recomputed from x and w1 = conv1 weight [16][3] (the W2S_PRO_FIRST flow) */
/* pre-pass of conv3's backward: part [B][ceil(L/tile)][2][C] = partial sums of g*GELU'(n) and g*GELU'(n)*n, n = IN(y) */
int w2s_gp_stats(const float* g, const float* y, const float* stats, float* part, int B, int L, int C, int tile, void* stream);

/* ---- row-wise ops on [rows][C] (nn.LayerNorm of the transformer; ConvLayerNorm models/utils.py:17-23) ---- */
int w2s_layernorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, float* rstat, int rows, int C,
                      float eps, int gelu, void* stream);
int w2s_layernorm_bwd(const float* g, int ldg, const float* x, int ldx, const float* gamma, const float* beta, const float* rstat,
                      const float* gadd, float* gx, int ldgx, float* part_gamma, float* part_beta, int rows, int C, int gelu,
                      int nparts, void* stream);
int w2s_bias_grad(const float* g, int rows, int C, int ldg, float* part, int nparts, void* stream); /* part[p][c] = block column sums */
int w2s_colsum(const float* part, int nparts, int C, int ld, float* out, int accumulate, void* stream);
/* many of them in one launch (jobs: HOST array, distinct outputs): the ~55 bias / gamma / beta / first-layer reductions of a backward pass */
typedef struct w2s_colsum_job { const float* part; float* out; int32_t nparts, C, ld, accumulate; } w2s_colsum_job;
int w2s_colsum_batch(const w2s_colsum_job* jobs, int njobs, void* stream); /* out[c] (+)= sum_p part[p*ld+c] */
/* out[row][c] = g[row*ldg+c] * GELU'(pre[row][c]) * keep[row/rows_per_sample]  (encoder-output GELU backward) */
int w2s_gelu_bwd_rows(const float* g, int ldg, const float* pre, const float* keep, int rows_per_sample, float* out, int rows, int C, void* stream);
int w2s_fill_rows(float* dst, int ld, const float* src, int rows, int C, void* stream);             /* CLS rows, wav2sleep.py:330 */
/* dst[row][c] (+)= keep[row / rows_per_sample] * src[c * src_stride] (keep may be NULL): register tokens r > 0 of the [1,1,F,R+1]
 * parameter (wav2sleep.py:299,330) and the signal-source embedding nn.Embedding row added to an encoder's output (wav2sleep.py:155-159) */
int w2s_add_rows(float* dst, int ld, const float* src, int src_stride, const float* keep, int rows_per_sample, int rows, int C, int accumulate,
                 void* stream);

/* Modality availability from the inputs alone (a sample lacks a modality when its row is -inf: models/wav2sleep.py:150,
 * trainer/masker.py:49-50).  xs / lds: HOST arrays of nsig <= W2S_MAX_SIGNALS device pointers [B][lds[m]] and their row lengths, in token
 * order.  keep [nsig][B] = 1 / 0 (the encoder epilogues' rowkeep); keypad [B*S][R1 + nsig] uint8 = the set-fusion transformer's
 * key-padding mask (tokens d < R1 = CLS / register tokens: never padded; wav2sleep.py:319-343).  Replaces ~20 stock element-wise launches
 * (isinf / bitwise_not / stack / expand / contiguous) per step. */
#define W2S_MAX_SIGNALS 8
int w2s_token_masks(const float* const* xs, const long* lds, int nsig, int R1, int B, int S, float* keep, uint8_t* keypad, void* stream);
/* dst [N][D][F]: rows d >= 1 zeroed, row 0 = src[n][:] (src NULL: left alone) -- a gradient that lives in the CLS rows only (wav2sleep.py:345) */
int w2s_cls_scatter(float* dst, const float* src, long N, int D, int F, void* stream);
/* strided row copy dst[r*ld_dst + c] = src[r*ld_src + c] (C and both strides multiples of 4) */
int w2s_copy_rows(float* dst, long ld_dst, const float* src, long ld_src, long rows, int C, void* stream);
int w2s_zero(void* p, long nbytes, void* stream);   /* nbytes % 4 == 0 */

/* elementwise on n floats (n % 4 == 0).  Dropout masks are a pure function of (seed, element index). */
#define W2S_ELT_GELU 0          /* y = GELU(a) */
#define W2S_ELT_GELU_BWD 1      /* y = b * GELU'(a) */
#define W2S_ELT_ADD 2           /* y = a + b */
#define W2S_ELT_ADD_DROP 3      /* y = a + drop(b) */
#define W2S_ELT_DROP 4          /* y = drop(a) */
#define W2S_ELT_GELU_DROP 5     /* y = drop(GELU(a)) */
#define W2S_ELT_GELU_DROP_BWD 6 /* y = drop(b) * GELU'(a) */
int w2s_eltwise(int op, const float* a, const float* b, float* y, long n, float p_drop, uint64_t seed, void* stream);

/* ---- set-fusion attention core: SDPA inside nn.MultiheadAttention (wav2sleep.py:286-296), D = 1+C tokens (2..7),
 * head_dim 16, qkv [N*D][3*H*16], keypad [N][D] (1 = missing modality), out [N*D][H*16].  nq = D, or 1: only token 0 of every epoch is a
 * QUERY (the last layer of the stack: wav2sleep.py:345 returns the CLS token alone) -- the other out rows are not written, and the
 * backward reads gout row 0 only and writes zeros to the other rows' query gradients ---- */
int w2s_attn_fwd(const float* qkv, const uint8_t* keypad, float* out, int N, int D, int H, int nq, float p_drop, uint64_t seed, void* stream);
int w2s_attn_bwd(const float* qkv, const uint8_t* keypad, const float* gout, float* gqkv, int N, int D, int H, int nq, float p_drop, uint64_t seed,
                 void* stream);

/* ---- classifier (wav2sleep.py:41,66), masked cross-entropy + confusion matrix (trainer/main.py:162-172) ---- */
int w2s_head_fwd(const float* pre, int ld, const float* w, const float* bias, float* logits, int rows, int F, int nc, int gelu_in, void* stream);
/* part: [ceil(rows/256)][2] scratch; loss_out[0] = mean NLL over labels != -1, loss_out[1] = count;
 * glogits (optional) = gscale * dLoss/dlogits; cmat (optional) int64 [nc][nc] += counts (rows = true, cols = argmax) */
int w2s_ce_fwd_bwd(const float* logits, const float* labels, int rows, int nc, float* part, float* loss_out, float* glogits,
                   long long* cmat, float gscale, void* stream);
/* The same loss when the batch is processed as sample waves (FusedTrainStep's pipelined step): count[0] = number of labels != -1 of the
 * WHOLE batch (a function of the labels alone: known before any logits exist); w2s_ce_wave = passes 1 and 3 of w2s_ce_fwd_bwd on one
 * wave's rows (partials into part[ceil(rows/256)][2] -- the caller lays the waves' slices out one after the other -- and the gradient
 * scaled by gscale / count[0]); w2s_ce_final = pass 2 over all `nblocks` partials: loss_out as above. */
int w2s_ce_count(const float* labels, int rows, int nc, float* count, void* stream);
int w2s_ce_wave(const float* logits, const float* labels, int rows, int nc, float* part, const float* count, float* glogits,
                long long* cmat, float gscale, void* stream);
int w2s_ce_final(const float* part, int nblocks, float* loss_out, void* stream);
/* gpre = (glogits . W) * (gelu_in ? GELU'(pre) : 1); part[nparts][nc*F + nc] = block partials of dW, db (sum with w2s_colsum) */
int w2s_head_bwd(const float* pre, int ld, const float* w, const float* glogits, float* gpre, int ldg, float* part, int nparts,
                 int rows, int F, int nc, int gelu_in, void* stream);

/* ---- optimiser on the flat buffers: clip_grad_norm_(L2) + AdamW (training/main.yaml:21-22, optimizer/adamw.yaml) ----
 * hyper (device, 8 floats): lr, weight_decay, beta1, beta2, eps, 1-beta1^t, 1-beta2^t, max_norm (<= 0: no clipping)
 * normcoef (device, 2 floats): total norm, clip coefficient */
int w2s_sumsq_partial(const float* g, long n, float* part, int nparts, void* stream);
int w2s_clip_coef(const float* part, int nparts, const float* hyper, float* normcoef, void* stream);
int w2s_adamw(float* p, const float* g, float* m, float* v, long n, const float* hyper, const float* normcoef, void* stream);
/* EMACallback on the flat parameter buffer (trainer/callbacks.py:55-64): ema = decay*ema + (1-decay)*p, evaluated as the
 * reference's mul_ then add_(alpha) (two roundings); w2s_swap exchanges two buffers in place (callbacks.py:84-98). */
int w2s_ema_update(float* ema, const float* p, long n, float decay, void* stream);
int w2s_swap(float* a, float* b, long n, void* stream);

/* ---- device-side input pipeline (data/dataset.py:76-87,174-182, trainer/main.py:342-353, masker.py:49-50) ----
 * w2s_zscore: y[row] = (x[row]-mean)/max(std_unbiased, eps) per row of T samples (rows with a non-finite value are copied);
 *   part: rows*nblk*3 doubles of scratch; stats_out (optional): [rows][2] = mean, std used.
 * w2s_augment: in place x[b,:] = keep[b] ? x[b,:]*sign[b] : -inf.   w2s_map_labels: AASM stages 0..4 -> 4/5 classes, else -1. */
int w2s_zscore(const float* x, float* y, int rows, long T, double* part, int nblk, float eps, float* stats_out, void* stream);
int w2s_augment(float* x, int B, long T, const float* sign, const uint8_t* keep, void* stream);
int w2s_map_labels(const float* src, float* dst, long n, int num_classes, void* stream);
/* HOST function (no GPU, no stream): causal EMA normalisation of one recording, the native counterpart of the reference's numba loop
 * (data/normalization.py:18-80,106-230), fp64 in and out like that loop.  baseline_tau_seconds <= 0 means "same as tau_seconds"; outlier may be NULL. */
int w2s_causal_normalize_host(const double* x, long n, double sampling_freq, double tau_seconds, double eps, double outlier_sigma,
                              double baseline_tau_seconds, double min_sigma, double* out, uint8_t* outlier);

const char* w2s_version(void);
/* Integer ABI number of this header: bumped whenever a signature or the meaning of an argument changes (round 5: 6; round 6: 7).  The host refuses a
 * library whose number differs from the one it was written against (wav2sleep_amd/lib.py), so a stale build_alt/ or W2S_LIB library is
 * an error at load time instead of shifted arguments at call time. */
#define W2S_ABI_VERSION 8
int w2s_abi_version(void);

/* ---- generic (untuned, inference) path: module variants outside the shipped production model (models/utils.py:26-96, ppgnet.py) ---- */
/* y[row][c] = act(x[row][c]*scale[s][c] + shift[s][c]), s = (row / rows_per_sample) * sample_stride (0: one vector for every sample);
 * scale == shift == NULL: activation only.  act: 0 linear, 1 ReLU, 2 LeakyReLU(slope), 3 GELU (erf), 4 SiLU (get_activation, utils.py:61-74).
 * Instance norm / eval BatchNorm / GroupNorm with their statistics and affine folded into (scale, shift).  In place allowed. */
int w2s_affine_act(const float* x, int ldx, const float* scale, const float* shift, int sample_stride, float* y, int ldy, int rows_per_sample,
                   long rows, int C, int act, float slope, void* stream);
/* per-position normalisation over C channels + activation: ConvLayerNorm (utils.py:9-23), ConvRMSNorm (rms != 0, beta NULL; :26-38), nn.LayerNorm */
int w2s_rownorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, long rows, int C, float eps, int rms,
                    int act, float slope, void* stream);
/* nn.MultiheadAttention's attention core for any head size hd, D <= 16 tokens: qkv [N][D][3*H*hd], keypad [N][D] (1 = padded key), out [N][D][H*hd];
 * p_drop > 0: dropout on the attention weights (training; mask from (seed, ((n H + h) D + query) D + key), regenerated by the backward) */
int w2s_attn_generic_fwd(const float* qkv, const unsigned char* keypad, float* out, long N, int D, int H, int hd, float p_drop, uint64_t seed,
                         void* stream);

/* ---- generic path, backward (round 6): what autograd runs for ConvLayer1D's norm -> activation (blocks.py:183-185), the row norms and
 * nn.MultiheadAttention when the reference trains SleepPPGNet (models/ppgnet.py) or a non-production Wav2Sleep configuration.  Data / weight
 * gradients of the convolutions and GEMMs are w2s_conv_forward (flip / W2S_MODE_UP2) and w2s_wgrad. ---- */
/* ga = g * act'(z), xh = stats ? (y - mean) * rstd : y, z = gamma ? xh * gamma + beta : xh; stats [.][C][2] = (mean, rstd), stats_stride = floats
 * between samples (0: one set for the batch).  part [nsamples][ceil(rows_per_sample / tile)][2][C] = per-tile sums of ga and ga * xh
 * (w2s_stats_finalize kind 1 turns them into the per-(sample, channel) means). */
int w2s_norm_act_bwd_part(const float* g, int ldg, const float* y, int ldy, const float* stats, int stats_stride, const float* gamma,
                          const float* beta, int rows_per_sample, int nsamples, int C, int act, float slope, int tile, float* part, void* stream);
/* gy = coef ? A * ga + B + Cx * xh : ga; coef [.][3][C] = (A, B, Cx), coef_stride = floats between samples (0: one set).  gy may alias g. */
int w2s_norm_act_bwd_apply(const float* g, int ldg, const float* y, int ldy, const float* stats, int stats_stride, const float* gamma,
                           const float* beta, const float* coef, int coef_stride, float* gy, int ldgy, int rows_per_sample, long rows, int C,
                           int act, float slope, void* stream);
/* backward of w2s_rownorm_fwd: x = the norm's input, g = gradient of the activation's output, gx = gradient of x (may alias g);
 * part [w2s_rownorm_bwd_blocks(rows)][2][C] = per-block sums for gamma's and beta's gradients (w2s_colsum finishes them).  C <= 1024. */
int w2s_rownorm_bwd_blocks(long rows);
int w2s_rownorm_bwd(const float* g, int ldg, const float* x, int ldx, const float* gamma, const float* beta, float* gx, int ldgx, float* part,
                    long rows, int C, float eps, int rms, int act, float slope, void* stream);
/* the per-(sample, channel) arithmetic around the statistics-based norms in one launch each (fp64 inside).  kind: 0 InstanceNorm1d (stats =
 * (mean, rstd) per (b, c)), 1 BatchNorm1d training (stats = (E[y], E[y^2]) per (b, c); run_mean / run_var updated in place with `momentum`
 * and the unbiased variance over `count` = B L elements, or NULL), 2 BatchNorm1d eval (running statistics; stats unused), 3 GroupNorm (stats =
 * (E[y], E[y^2]); G groups of consecutive channels).  scale / shift [nset][C] feed w2s_affine_act, mr [nset][C][2] = (mean, rstd) feeds the
 * backward kernels as `stats`; nset = B for kinds 0 and 3, 1 for kinds 1 and 2; ss (or NULL) [B][C][2] = (scale, shift) per sample, the
 * (scale, shift) operand of W2S_PRO_AFFINE. */
int w2s_norm_fold(int kind, const float* stats, int B, int C, int G, const float* gamma, const float* beta, float* run_mean, float* run_var,
                  float eps, float momentum, double count, float* scale, float* shift, float* mr, float* ss, void* stream);
/* means [B][C][2] = per-(sample, channel) means of ga and ga * xh -> coef [nset][3][C] for w2s_norm_act_bwd_apply, dgamma / dbeta [C] (or NULL),
 * cd (or NULL) [B][C][2] = the (c, d) operand of W2S_PRO_AFFINE_BWD for every sample */
int w2s_norm_bwd_coef(int kind, const float* means, const float* mr, int B, int C, int G, const float* gamma, const float* beta, double L,
                      float* coef, float* dgamma, float* dbeta, float* cd, int y_sums, void* stream);   /* y_sums: means[..][1] is of ga * y (W2S_EPI_AFFINE_PART) */
/* the residual join of a ConvBlock1D (blocks.py:68-70) in one pass: y = act2(act(x * scale + shift) + add) -- x = conv3's raw output, (scale, shift)
 * as for w2s_affine_act, add = downsample(x_in), act2 = the block's activation -- and its backward gs = g * act2'(act(x * scale + shift) + add),
 * the gradient of both addends.  y / gs may alias add. */
int w2s_affine_act_join(const float* x, int ldx, const float* scale, const float* shift, int sample_stride, const float* add, int ld_add, float* y,
                        int ldy, int rows_per_sample, long rows, int C, int act, int act2, float slope, void* stream);
int w2s_affine_act_join_bwd(const float* g, int ldg, const float* x, int ldx, const float* scale, const float* shift, int sample_stride,
                            const float* add, int ld_add, float* gs, int ldgs, int rows_per_sample, long rows, int C, int act, int act2, float slope,
                            void* stream);
/* convolutions of the ONE-channel input (block 0's conv1 and 1x1 / stride-2 residual conv, blocks.py:44-55 with input_dim = 1) on the vector ALU:
 * y[b,t,o] = bias[o] + sum_j w[o][j] x[b, t stride + j - pad], x [B][L_in], w [C][K] (K <= 3), y [B][L_out][C]; part (or NULL)
 * [B][ceil(L_out / 1024)][2][C] = per-tile sums of y and y^2 (the statistics partials of the fused conv epilogue, for w2s_stats_finalize) */
int w2s_conv1_fwd(const float* x, const float* w, const float* bias, float* y, float* part, int B, int L_in, int L_out, int C, int K, int stride,
                  int pad, void* stream);
/* its weight gradient: part [w2s_conv1_wgrad_parts(B, L_out)][C][K] per-(sample, tile) sums of gy[b,t,o] x[b, t stride + j - pad] (w2s_colsum
 * finishes them); y2 != NULL: gy is formed on the fly as W2S_PRO_AFFINE_BWD + act does (g, y2, ss = (scale, shift), cd = (c, d) per (b, c)) */
int w2s_conv1_wgrad_parts(int B, int L_out);
int w2s_conv1_wgrad(const float* g, const float* y2, const float* ss, const float* cd, const float* x, float* part, int B, int L_in, int L_out, int C,
                    int K, int stride, int pad, int act, void* stream);
/* backward of w2s_attn_generic_fwd: gqkv [N][D][3*H*hd] (fully written) from gout [N][D][H*hd] */
int w2s_attn_generic_bwd(const float* qkv, const unsigned char* keypad, const float* gout, float* gqkv, long N, int D, int H, int hd, float p_drop,
                         uint64_t seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif
