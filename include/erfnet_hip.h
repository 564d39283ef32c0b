/*
 * erfnet_hip.h — C ABI of liberfnet_hip.so: the ERFNet block forward/backward path as
 * hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (lorenwel/erfnet_pytorch) has no FFI: its hot path is the Python module API of
 * train/erfnet_blocks.py, whose arithmetic runs inside torch.nn.  Each entry point below is the
 * native replacement for one of those modules' forward()/backward pair; the reference line a
 * function replaces is cited next to it.  The Python host side (erfnet_pytorch_amd/) binds these
 * with ctypes and hands in raw device pointers of torch tensors plus the current HIP stream.
 *
 * Conventions
 *   - All activation tensors are fp32, NHWC in memory ("channels_last" in torch terms: logical
 *     shape [N,C,H,W], strides [H*W*C, 1, W*C, C]).  Exceptions are stated per function.
 *   - Weights / biases / BatchNorm vectors are in torch's native layouts
 *     (Conv2d [Cout][Cin][KH][KW], ConvTranspose2d [Cin][Cout][KH][KW]), fp32, on the device.
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls are asynchronous; nothing
 *     allocates or synchronises.  Scratch memory comes from the caller (`ws`, `ws_bytes`);
 *     query the size with erf_workspace_bytes().
 *   - Return value 0 = launched; nonzero = rejected before any launch (bad shape, small
 *     workspace, unsupported size); erf_last_error() then describes it.  A CPU fallback does
 *     not exist.
 */
#ifndef ERFNET_HIP_H
#define ERFNET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* nn.BatchNorm2d(C, eps=1e-3) attached to a block (train/erfnet_blocks.py:23,44,50,82).
 * mode 0: absent (use_batchnorm=False)        — all pointers may be NULL
 * mode 1: eval, normalise with running stats
 * mode 2: train, batch statistics; running_mean/var are updated in place (momentum, unbiased
 *         variance); save_mean/save_invstd ([C] each) receive the batch statistics for backward.
 * In modes 1 and 2 save_mean/save_invstd must be valid when a backward call will follow.
 * num_batches_tracked: optional device int64 (nn.BatchNorm2d's counter); a mode-2 forward adds 1. */
typedef struct {
    int mode;
    const float *gamma, *beta;
    float *running_mean, *running_var;
    float *save_mean, *save_invstd;
    float momentum, eps;
    long long *num_batches_tracked;
} erf_bn_t;

/* Library identification / diagnostics.  erf_version() returns the ABI revision the library was
 * built with; a caller compiled against this header compares it with ERF_ABI_VERSION and refuses a
 * mismatch (struct layouts are part of the ABI: revision 101 added erf_bn_t::num_batches_tracked and
 * changed erf_nb1d_block_t's member types, 102 added ERF_OPT_X3_ROLL and the capture rule of
 * erf_cross_entropy_forward_static, 103 added erf_nb1d_block_t::sign_bits / sign_bits_written and
 * erf_nb1d_backward_bits, 104 the erf_batch_* calls, 105 ERF_OPT_POOL_FUSE). */
#define ERF_ABI_VERSION 105
int erf_version(void);
const char *erf_last_error(void);
/* Number of kernel launches issued through this library by the calling process (tests use it to
 * prove the HIP path, not a fallback, produced a result). */
unsigned long long erf_launch_count(void);

/* Backward of a run of non_bottleneck_1d blocks (`for layer in self.layers`,
 * train/net_classification.py:32-36,42-45, walked in reverse by autograd) with the small launches of
 * its blocks batched: one weight-packing launch for the run instead of one per block, and the slab
 * reductions of all its weight gradients together at the end.  Protocol, on one thread and one stream:
 *   erf_batch_begin();
 *   for every block, last to first:  erf_nb1d_backward*( ... )     plans only, launches nothing
 *   erf_batch_pack(stream);
 *   for every block, same order, SAME arguments:  erf_nb1d_backward*( ... )
 *   erf_batch_end(stream);                                         the weight gradients are complete here
 * Every block needs a workspace of its own (erf_workspace_bytes each, disjoint), unchanged between the
 * two passes: its packed weights and gradient slabs live there until erf_batch_end.  Results are bit-
 * identical to the unbatched calls.  erf_batch_abort() leaves the protocol after an error.  Only
 * erf_nb1d_backward, erf_nb1d_backward_bits and their _bf16 forms may be called between begin and end. */
int erf_batch_begin(void);
int erf_batch_pack(void *stream);
int erf_batch_end(void *stream);
void erf_batch_abort(void);

/* Tuning switches (no counterpart in the reference).  erf_set_option returns the previous value,
 * or -1 for an unknown option.  ERF_OPT_PAIR_FUSION (default 1; environment ERFNET_NO_PAIR=1 turns
 * it off): consecutive 1-D convolutions of a non_bottleneck_1d run as one launch where the image
 * geometry allows; results are bit-identical either way. */
#define ERF_OPT_PAIR_FUSION 1
/* ERF_OPT_WGRAD_LINE (default 1; ERFNET_NO_WGRAD_LINE=1): weight gradients of the 64/128-channel
 * 1-D convolutions with all three taps per block; off = one block per tap.  ERF_OPT_STREAM16
 * (default 1; ERFNET_NO_STREAM16=1): LDS-free kernels for the 16-channel layers; off = the general
 * implicit-GEMM kernels.  Either way the results agree to fp32 rounding (different summation order). */
#define ERF_OPT_WGRAD_LINE 2
#define ERF_OPT_STREAM16 3
/* ERF_OPT_PAIR16 (default 0; ERFNET_PAIR16=1): the conv pairs of the 16-channel blocks as one launch
 * (pair16_kernel) under ERF_OPT_PAIR_FUSION; same results, currently slower than two launches. */
#define ERF_OPT_PAIR16 4
/* ERF_OPT_LINE2 (default 1; ERFNET_NO_LINE2=1): bf16 conv pairs on the eight-wave kernel whose wave
 * groups work on consecutive tiles (line2_bf16_kernel); off = the four-wave kernel; same results. */
#define ERF_OPT_LINE2 5
/* ERF_OPT_SPLIT_BF16 (default 1; ERFNET_NO_X3=1 -> 0, ERFNET_X3_ALWAYS=1 -> 2): the fp32 64/128-channel
 * 1-D convolutions and their weight gradients on the bf16 matrix pipe, each fp32 operand split into three
 * bf16 terms (exactly) and six of the nine partial products accumulated in fp32 (line_x3_kernel,
 * wgrad_line_x3_kernel): fp32 storage and fp32-level accuracy (the dropped terms are below 2^-24 of a
 * product), 2.7x the fp32 matrix rate.  1 = where it pays: a convolution launch with at least two line
 * tiles per resident block (smaller launches keep the fused fp32 pair); 2 = every eligible convolution;
 * 0 = the fp32-MFMA kernels (pair_f32_kernel, wgrad_line_kernel).  The paths agree to fp32 rounding.
 * Non-finite inputs give NaN on the split path. */
#define ERF_OPT_SPLIT_BF16 6
/* ERF_OPT_CONV16_ROWS (default 1; ERFNET_NO_CONV16_ROWS=1): bf16 16-channel 1-D convolutions on images
 * whose width is a multiple of 16 take the kernel that keeps a tile's coordinates in scalar registers
 * (conv16_bf16_rows_kernel); off = the general kernel.  Same results bit for bit. */
#define ERF_OPT_CONV16_ROWS 7
/* ERF_OPT_X3_ROLL (default 1; ERFNET_NO_X3R=1): the split-bf16 convolutions of ERF_OPT_SPLIT_BF16 on the
 * rolling kernel (an input line is split into its bf16 terms once and serves three output lines)
 * instead of once per tap. */
#define ERF_OPT_X3_ROLL 8
/* ERF_OPT_POOL_FUSE (default 1; ERFNET_NO_POOL_FUSE=1): erf_downsampler_backward with a 16-channel NHWC
 * input forms the max-pool's gradient (train/erfnet_blocks.py:29-30: the pooled half of the concatenation)
 * in the epilogue of the convolution's data-gradient kernel; off = a kernel of its own writes it into dx
 * first.  Same results bit for bit. */
#define ERF_OPT_POOL_FUSE 9
int erf_set_option(int option, int value);

/* Flags of the backward entry points.  Every block ends in a ReLU, so the gradient a block
 * receives is first masked by (its output > 0).  A block whose input x is itself such a ReLU
 * output may apply the producer's mask to the dx it hands back (ERF_BWD_MASK_DX: dx *= (x > 0),
 * fused into the data-gradient epilogue); the producer, told that the gradient it receives comes
 * from exactly that consumer (ERF_BWD_GY_PREMASKED), skips its own masking pass.  Results for all
 * parameter gradients and for dx of the first block are identical with and without the flags. */
#define ERF_BWD_GY_PREMASKED 1
#define ERF_BWD_MASK_DX 2

/* Upper bound of the scratch bytes any entry point needs for an activation tensor of
 * n_pixels = N*H*W output pixels and at most max_c channels. */
size_t erf_workspace_bytes(long long n_pixels, int max_c);

/* ---- DownsamplerBlock (train/erfnet_blocks.py:15-29; eval/erfnet.py:12-23) -------------------
 * y = relu([bn](cat([conv3x3 s2 p1 (x) -> Cout-Cin ch, maxpool2x2 s2 (x) -> Cin ch], dim=1)))
 * x: [N,Cin,H,W] with element strides (sn,sc,sh,sw) — any layout when Cin == 3 (the network
 *    input is read as it is), NHWC otherwise.  H and W must be even (the reference fails in
 *    torch.cat otherwise).
 * y: NHWC [N,Cout,H/2,W/2].  pre: NHWC buffer of the same size receiving the pre-BN
 *    concatenation; required when bn->mode != 0 and a backward call will follow, else NULL. */
int erf_downsampler_forward(const float *x, long long sn, long long sc, long long sh, long long sw,
                            const float *w, const float *b, const erf_bn_t *bn, float *y,
                            float *pre, int N, int Cin, int H, int W, int Cout, void *ws,
                            size_t ws_bytes, void *stream);
/* gy, y, pre: NHWC [N,Cout,H/2,W/2].  dx (NHWC [N,Cin,H,W]) may be NULL (network input).
 * dw [Cout-Cin][Cin][3][3], db [Cout-Cin], dgamma/dbeta [Cout] (BN modes) are overwritten. */
int erf_downsampler_backward(const float *x, long long sn, long long sc, long long sh,
                             long long sw, const float *w, const erf_bn_t *bn, const float *y,
                             const float *pre, const float *gy, float *dx, float *dw, float *db,
                             float *dgamma, float *dbeta, int N, int Cin, int H, int W, int Cout,
                             int flags, void *ws, size_t ws_bytes, void *stream);

/* ---- non_bottleneck_1d (train/erfnet_blocks.py:33-73; eval/erfnet.py:26-61) -----------------
 * t1 = relu(conv3x1_1(x)); t2 = relu([bn1](conv1x3_1(t1))); t3 = relu(conv3x1_2(t2));
 * t4 = [bn2](conv1x3_2(t3)); t4 = [dropout2d](t4); y = relu(t4 + x)
 * "3x1" convolves along H (kernel (3,1), padding (d,0), dilation (d,1)), "1x3" along W.
 * w[4]: conv3x1_1, conv1x3_1, conv3x1_2, conv1x3_2 weights, each [C][C][3]; b[4]: biases [C].
 * drop_scale: NULL, or [N][C] multipliers (0 or 1/(1-p)) realising nn.Dropout2d.
 * save: NULL for inference (two scratch tensors from ws are used), else 5 NHWC buffers of
 *       N*C*H*W floats: t1, pre2 (pre-BN conv1x3_1 output; unused when bn1 is off), t2, t3,
 *       pre4 (pre-BN conv1x3_2 output; unused when bn2 is off). */
int erf_nb1d_forward(const float *x, const float *const w[4], const float *const b[4],
                     const erf_bn_t *bn1, const erf_bn_t *bn2, const float *drop_scale, int dil,
                     float *y, float *const save[5], int N, int C, int H, int W, void *ws,
                     size_t ws_bytes, void *stream);

/* A run of consecutive non_bottleneck_1d blocks of one channel count and resolution, as the
 * encoders and decoders apply them (train/net_classification.py:32-36, 42-45: "for layer in
 * self.layers: output = layer(output)"): y_0 = block_0(x), y_i = block_i(y_{i-1}).  The results
 * equal nblocks calls of erf_nb1d_forward bit for bit; handing the run over at once lets the
 * library schedule the last convolution of a block and the first of the next as one launch.
 * Members as the arguments of erf_nb1d_forward; every block's y is written. */
typedef struct {
    const float *w[4], *b[4];
    const erf_bn_t *bn1, *bn2;
    const float *drop_scale;
    int dil;
    void *y;                /* fp32 tensors for erf_nb1d_chain_forward, bf16 for the _bf16 variant */
    void *const *save;
    /* The ReLU masks of the block for the backward pass, one bit per element instead of the
     * activation itself (no counterpart in the reference, whose autograd keeps the activations):
     * NULL, or 4 device buffers (any of them NULL) of N*H*W*C/32 words for t1, t2, t3 and y; word
     * [pixel][c / 32] holds (value > 0) of channel c in bit c % 32.  The library fills the buffers
     * it can fill for free (from the epilogue of the kernel that writes the tensor) and reports
     * which in *sign_bits_written (host memory, bit k = buffer k; set when the call returns, not
     * when the stream gets there).  The tensors themselves are written either way. */
    unsigned *const *sign_bits;
    unsigned *sign_bits_written;
} erf_nb1d_block_t;
int erf_nb1d_chain_forward(const float *x, const erf_nb1d_block_t *blocks, int nblocks, int N, int C,
                           int H, int W, void *ws, size_t ws_bytes, void *stream);
/* dw[4]/db[4] (and dgamma[2]/dbeta[2] in BN modes) are overwritten; dx is required. */
int erf_nb1d_backward(const float *x, const float *const w[4], const erf_bn_t *bn1,
                      const erf_bn_t *bn2, const float *drop_scale, int dil, const float *y,
                      const float *const save[5], const float *gy, float *dx, float *const dw[4],
                      float *const db[4], float *const dgamma[2], float *const dbeta[2], int N,
                      int C, int H, int W, int flags, void *ws, size_t ws_bytes, void *stream);
/* The same with the masks a forward call left in erf_nb1d_block_t::sign_bits: sign_bits[4] =
 * the bit buffers of t1, t2, t3 and of x (= the y buffer of the block that produced x), each NULL
 * when it was not written.  A data-gradient launch reads 1/32 of the bytes for its mask where the
 * bits are given; save[] and x are still required (the weight gradients read them).  Results are
 * bit-identical to erf_nb1d_backward. */
int erf_nb1d_backward_bits(const float *x, const float *const w[4], const erf_bn_t *bn1,
                           const erf_bn_t *bn2, const float *drop_scale, int dil, const float *y,
                           const float *const save[5], const unsigned *const sign_bits[4],
                           const float *gy, float *dx, float *const dw[4], float *const db[4],
                           float *const dgamma[2], float *const dbeta[2], int N, int C, int H, int W,
                           int flags, void *ws, size_t ws_bytes, void *stream);

/* ---- Two chained 1-D convolutions (train/erfnet_blocks.py:57-62 and 64-66: conv -> [relu] ->
 *      conv of a non_bottleneck_1d), C -> C channels, 3 taps each, as ONE launch ------------------
 * mid = act(conv_first(x) + b_first);  y = act(conv_second(mid) + b_second [+ residual])
 * first_axis: 1 = the first convolution runs along W (kernel (1,3)) and the second along H
 * (kernel (3,1)); 0 = the other way round.  A block computes a whole image line of the second
 * convolution (a column for first_axis = 1), so that line must fit a tile: C = 128 with 17..128
 * pixels per line, C = 64 with 33..256; otherwise the call is rejected
 * and the caller uses erf_nb1d_forward, which picks the launch plan itself.
 * Weights are repacked once with erf_conv_pair_pack (torch layout [C][C][3] each, `packed` holds
 * 2*3*C*C floats) and may then serve any number of forward calls — at inference they never change.
 * mid may be NULL (the intermediate then never leaves the chip).  x, mid, y, residual: NHWC, 16-byte
 * aligned, y and mid distinct from x. */
int erf_conv_pair_pack(const float *w_first, const float *w_second, int C, float *packed, void *stream);
int erf_conv_pair_forward(const float *x, const float *packed, const float *b_first,
                          const float *b_second, int first_axis, int dil_first, int dil_second,
                          int relu_first, int relu_second, const float *residual, float *mid, float *y,
                          int N, int C, int H, int W, void *stream);

/* ---- UpsamplerBlock (train/erfnet_blocks.py:77-88; eval/erfnet.py:99-108) -------------------
 * y = relu([bn](ConvTranspose2d(Cin,Cout,3,stride 2,padding 1,output_padding 1)(x)))
 * x NHWC [N,Cin,H,W]; y, pre NHWC [N,Cout,2H,2W]; w [Cin][Cout][3][3]. */
int erf_upsampler_forward(const float *x, const float *w, const float *b, const erf_bn_t *bn,
                          float *y, float *pre, int N, int Cin, int H, int W, int Cout, void *ws,
                          size_t ws_bytes, void *stream);
int erf_upsampler_backward(const float *x, const float *w, const erf_bn_t *bn, const float *y,
                           const float *pre, const float *gy, float *dx, float *dw, float *db,
                           float *dgamma, float *dbeta, int N, int Cin, int H, int W, int Cout,
                           int flags, void *ws, size_t ws_bytes, void *stream);

/* ---- output ConvTranspose2d(Cin, ncls, 2, stride=2) (train/net_classification.py:56,64;
 *      eval/erfnet.py:124,132; train/erfnet_blocks.py:99) --------------------------------------
 * out[n,o,2i+a,2j+b] = bias[o] + sum_c x[n,c,i,j] * w[c,o,a,b].
 * x NHWC [N,Cin,H,W]; y is written NCHW-contiguous [N,ncls,2H,2W] (it feeds torch's loss /
 * argmax directly); softmax != 0 applies softmax over the class dimension
 * (SoftMaxConv, train/erfnet_blocks.py:104-110). */
int erf_outconv_forward(const float *x, const float *w, const float *b, float *y, int N, int Cin,
                        int H, int W, int ncls, int softmax, void *stream);
/* labels[n][0][oh][ow] = argmax over classes of the logits erf_outconv_forward would write: the
 * `outputs.max(1)[1].unsqueeze(1)` of the scripts (eval/eval_iou.py:92) in one pass, int64, lowest index on ties. */
int erf_outconv_argmax(const float *x, const float *w, const float *b, long long *labels, int N,
                       int Cin, int H, int W, int ncls, void *stream);
/* gy NCHW [N,ncls,2H,2W] = gradient w.r.t. the conv output (pre-softmax); dx NHWC. */
int erf_outconv_backward(const float *x, const float *w, const float *gy, float *dx, float *dw,
                         float *db, int N, int Cin, int H, int W, int ncls, int flags, void *ws,
                         size_t ws_bytes, void *stream);
/* Backward of erf_cross_entropy_forward THROUGH the output convolution, for the training loop's
 * `criterion(model(x), targets).backward()` (train/main.py:333,379): equals erf_cross_entropy_backward
 * followed by erf_outconv_backward, but the gradient of the logits is formed in registers (logits
 * recomputed from x) and never written.  target / lse / stats as erf_cross_entropy_forward received
 * and produced them for the logits of this x, w, b; gy: device scalar, gradient of the loss. */
int erf_outconv_ce_backward(const float *x, const float *w, const float *b, const long long *target,
                            const float *lse, const float *stats, const float *gy, float *dx, float *dw,
                            float *db, int N, int Cin, int H, int W, int ncls, long long ignore_index,
                            int flags, void *ws, size_t ws_bytes, void *stream);

/* Forward of the training loop's `criterion(model(x), targets)` THROUGH the output convolution
 * (train/main.py:327,333: outputs = model(inputs); loss = criterion(outputs, targets), with
 * train/net_classification.py:56,64 producing the logits): equals erf_outconv_forward followed by
 * erf_cross_entropy_forward[_counts] on its logits, but the logits are formed in registers and never
 * written — the loop reads them for nothing else, and erf_outconv_ce_backward recomputes them.
 * stats / lse / counts as erf_cross_entropy_forward_counts produces them (lse bit for bit; the loss
 * to the rounding of a different summation order); counts may be NULL.  1 <= ncls <= 32. */
int erf_outconv_ce_forward(const float *x, const float *w, const float *b, const long long *target,
                           float *stats, float *lse, long long *counts, int N, int Cin, int H, int W,
                           int ncls, long long ignore_index, void *ws, size_t ws_bytes, void *stream);

/* ---- plain nn.Conv2d on NHWC tensors (Encoder.output_conv 1x1, eval/erfnet.py:85,93-94;
 *      LadderBlock's 3x3 conv, train/erfnet_blocks.py:145,151-154) -----------------------------
 * Square kernel k in {1,3}, stride 1, padding (k-1)/2.  relu != 0 fuses F.relu (no BN). */
int erf_conv2d_forward(const float *x, const float *w, const float *b, float *y, int N, int Cin,
                       int H, int W, int Cout, int k, int relu, void *ws, size_t ws_bytes,
                       void *stream);
/* gy is the gradient w.r.t. the block output; when relu != 0, y (the saved output) masks it. */
int erf_conv2d_backward(const float *x, const float *w, const float *y, const float *gy,
                        float *dx, float *dw, float *db, int N, int Cin, int H, int W, int Cout,
                        int k, int relu, void *ws, size_t ws_bytes, void *stream);

/* ---- CrossEntropyLoss2d = nn.CrossEntropyLoss(ignore_index=-1)
 *      (train/evaluation_functions.py:6-14; used by train/main.py:186-188,333,373) -------------
 * logits NCHW-contiguous [N,K,H,W] fp32 (what erf_outconv_forward writes); target [N,H,W] int64 with
 * values in [0,K) or ignore_index.  stats[2] (device) receives {mean of -log softmax[target] over
 * the labelled pixels, number of labelled pixels}; lse [N,H,W] receives the per-pixel log-sum-exp
 * that backward reuses.  No pixel labelled -> stats[0] is NaN, as in torch.  Any other label outside
 * [0,K) is treated like ignore_index (torch aborts with a device-side assert on such a label). */
int erf_cross_entropy_forward(const float *logits, const long long *target, float *stats, float *lse,
                              int N, int K, int H, int W, long long ignore_index, void *ws,
                              size_t ws_bytes, void *stream);
/* The same, and in the same pass over the logits the counts behind the reference's accuracy metrics
 * (ClassificationAccuracy / MeanAccuracy, train/evaluation_functions.py:133-163, which train/main.py:343-347
 * evaluates on the tensors the loss has just read): counts int64 [2][K] is OVERWRITTEN with, per label
 * class, the pixels whose arg-max class (first largest) equals the label, and the pixels carrying
 * that label.  K <= 32. */
int erf_cross_entropy_forward_counts(const float *logits, const long long *target, float *stats, float *lse,
                                     long long *counts, int N, int K, int H, int W, long long ignore_index,
                                     void *ws, size_t ws_bytes, void *stream);
/* The same loss for callers that cannot allocate device memory or bring a workspace: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the
 * reference.  The block partial sums go to a 1 MiB array inside the library's code object (one
 * copy per GPU, held until the process ends), so the call allocates nothing and takes no `ws`;
 * results equal erf_cross_entropy_forward bit for bit.  At most 33,554,432 pixels (N*H*W) per
 * call; larger inputs are rejected.  The device whose stream is passed must be current.  The first
 * call on a device resolves the array's address and is rejected while that stream is being captured
 * (any earlier erf_cross_entropy_forward / _counts call on the device resolves it as well); later
 * calls may be captured.  All calls
 * on one GPU share that array: calls that overlap in time on the same GPU (two streams) race;
 * calls on one stream are ordered and need no clearing in between. */
#define ERF_CE_STATIC_MAX_PIXELS 33554432LL
int erf_cross_entropy_forward_static(const float *logits, const long long *target, float *stats,
                                     float *lse, int N, int K, int H, int W, long long ignore_index,
                                     void *stream);
/* dlogits = d(stats[0])/d(logits) * gy[0]; gy is a device scalar (the upstream gradient). */
int erf_cross_entropy_backward(const float *logits, const long long *target, const float *lse,
                               const float *stats, const float *gy, float *dlogits, int N, int K, int H,
                               int W, long long ignore_index, void *stream);

/* ---- masked regression losses (train/evaluation_functions.py:32-97; train/main.py:173-184) ------
 * The mean, over the pixels whose float target is not -1, of a per-pixel value:
 *   kind 0  L1LossMasked (:89-97)                      |a - t|                         K = 1
 *   kind 1  LogLikelihoodLossMasked (:71-87)           log(2 pi s) + (a - t)^2 / s,  s = b + eps, K = 1
 *   kind 2  L1LossClassProbMasked (:32-47)             sum_k a_k |b_k - t|
 *   kind 3  LogLikelihoodLossClassProbMasked (:49-69)  sum_k a_k (log(2 pi s_k) + (b_k - t)^2 / s_k),
 *                                                      s_k = c_k + eps (the reference's class adds no
 *                                                      eps of its own: pass its opt_eps)
 * a: [N][K][H][W] (network output, cost, or class probabilities); b: variance [N][H][W] (kind 1) or
 * the per-class cost [K] (kinds 2, 3); c: the per-class variance [K] (kind 3); target: [N][H][W].
 * stats: device float[2] = {mean, number of labelled pixels}; mean is NaN when nothing is labelled,
 * as the reference's mean of an empty selection.  Backward: da like a; db like b (kinds 1-3, may be
 * NULL for 2, 3); dc like c (kind 3, may be NULL); gy is the device scalar d(loss).
 * workspace: 8 bytes per 1024 pixels forward, 8 K bytes per 1024 pixels backward (kinds 2, 3). */
int erf_masked_loss_forward(int kind, const float *a, const float *b, const float *c, const float *target,
                            float eps, float *stats, int N, int K, int H, int W, void *ws, size_t ws_bytes,
                            void *stream);
int erf_masked_loss_backward(int kind, const float *a, const float *b, const float *c, const float *target,
                             float eps, const float *stats, const float *gy, float *da, float *db, float *dc,
                             int N, int K, int H, int W, void *ws, size_t ws_bytes, void *stream);

/* ---- MSELossWeighted (train/evaluation_functions.py:122-131; the mean-teacher consistency loss,
 * train/main.py:191,341,353) ------------------------------------------------------------------------
 * stats[0] = mean((a - b)^2) over n elements, stats[1] = n; the caller multiplies by
 * scale_factor * weight.  Backward: da = 2 (a - b) gy[0] / n. */
int erf_mse_mean_forward(const float *a, const float *b, long long n, float *stats, void *ws, size_t ws_bytes,
                         void *stream);
int erf_mse_mean_backward(const float *a, const float *b, const float *gy, float *da, long long n, void *stream);

/* ---- iouEval.addBatch for label maps (train/iouEval.py:21-70, eval/iouEval.py) ------------------
 * pred, target: int64 class indices, n_pixels each (the [N,1,H,W] tensors the scripts pass:
 * `outputs.max(1)[1].unsqueeze(1)` and the label batch).  Classes [0, counted) are tallied, where
 * counted = ignore_index, or n_classes when ignore_index = -1 (the reference slices the one-hot
 * tensors to [:ignoreIndex]); a false positive on a pixel whose target is ignore_index is not
 * counted.  counts: int64 [3][counted] = tp, fp, fn; the call ADDS this batch (zero it to reset).
 * Exact integer arithmetic; IoU = tp / (tp + fp + fn + 1e-15) is left to the caller. */
int erf_iou_add_batch(const long long *pred, const long long *target, long long n_pixels,
                      int n_classes, int ignore_index, long long *counts, void *ws, size_t ws_bytes,
                      void *stream);

/* ---- ClassificationAccuracy / MeanAccuracy (train/evaluation_functions.py:133-163) ---------------
 * scores: NCHW fp32 [N,K,H,W] (logits or probabilities); target: int64 [N,H,W].  counts: int64
 * [2][K]; the call ADDS, per label class c, counts[0][c] += pixels labelled c whose argmax over the
 * K scores is c, counts[1][c] += pixels labelled c.  Labels outside [0,K) (the ignore label -1) are
 * skipped.  Accuracy = sum(correct) / sum(total); class accuracy = correct[c] / total[c]. */
int erf_accuracy_counts(const float *scores, const long long *target, int N, int K, int H, int W,
                        long long *counts, void *ws, size_t ws_bytes, void *stream);

/* ---- Parameter updates (the optimiser calls and the mean-teacher update of the training loops) as
 *      multi-tensor launches: `count` tensors per call, the pointer tables and `numel` in HOST memory,
 *      the tensors themselves fp32 on the device; tensors with numel 0 are skipped; 32 tensors per
 *      kernel launch (one launch for a model whose parameters live in a few flat buffers).
 * erf_adam_step replaces torch.optim.Adam(params, lr, betas, eps=eps, weight_decay=wd).step()
 *   (train/main.py:213, stepped at :381): L2 weight decay added to the gradient, no amsgrad;
 *   `step` is the 1-based count of this update (bias corrections are formed from it in double on the
 *   host, as torch does); exp_avg / exp_avg_sq start as zeros and are updated in place.
 * erf_sgd_step replaces torch.optim.SGD(params, lr, momentum=m, weight_decay=wd).step()
 *   (imagenet/main.py:83-85): buf = m * buf + (grad + wd * p), p -= lr * buf; a zero buffer on the
 *   first step gives torch's "buf = grad" initialisation.
 * erf_ema_update replaces copyWeightsToModelWithDiscount (train/main.py:140-142, called at :390):
 *   target = target * (1 - discount) + discount * source. */
int erf_adam_step(int count, float *const *params, const float *const *grads, float *const *exp_avg,
                  float *const *exp_avg_sq, const long long *numel, double lr, double beta1, double beta2,
                  double eps, double weight_decay, long long step, void *stream);
int erf_sgd_step(int count, float *const *params, const float *const *grads, float *const *momentum_buf,
                 const long long *numel, double lr, double momentum, double weight_decay, void *stream);
int erf_ema_update(int count, float *const *target, const float *const *source, const long long *numel,
                   double discount, void *stream);

/* ================================================================================================
 * bf16 storage.  Every block call above exists a second time, suffixed _bf16, for activation tensors
 * stored as bf16 (the reference's half-precision attempt is eval/eval_forwardTime.py:25,
 * `model.half()`; bf16 is the 16-bit type gfx950's matrix cores take).  Same arguments and meaning:
 *   - activation / gradient tensors (x, y, pre, save[], gy, dx, residual, mid, the block structs' y
 *     and save) are bf16, NHWC, 16-byte aligned;
 *   - parameters, biases, BatchNorm vectors, dropout factors and every parameter gradient stay fp32
 *     (the kernels round the weights to bf16 when they repack them; products are accumulated in
 *     fp32; bias, BatchNorm, residual and ReLU are applied in fp32 before the one rounding of the
 *     stored result);
 *   - the 64 / 128-channel 1-D convolutions run on v_mfma_f32_32x32x16_bf16 with the weights held
 *     in registers (csrc/line_bf16.h); other layers widen bf16 to fp32 on their way into the fp32
 *     kernels.
 * Expected agreement with the fp32 entry points: 1e-2 of the output's scale. */
int erf_downsampler_forward_bf16(const void *x, long long sn, long long sc, long long sh, long long sw,
                                 const float *w, const float *b, const erf_bn_t *bn, void *y,
                                 void *pre, int N, int Cin, int H, int W, int Cout, void *ws,
                                 size_t ws_bytes, void *stream);
int erf_downsampler_backward_bf16(const void *x, long long sn, long long sc, long long sh,
                                  long long sw, const float *w, const erf_bn_t *bn, const void *y,
                                  const void *pre, const void *gy, void *dx, float *dw, float *db,
                                  float *dgamma, float *dbeta, int N, int Cin, int H, int W,
                                  int Cout, int flags, void *ws, size_t ws_bytes, void *stream);
int erf_nb1d_forward_bf16(const void *x, const float *const w[4], const float *const b[4],
                          const erf_bn_t *bn1, const erf_bn_t *bn2, const float *drop_scale, int dil,
                          void *y, void *const save[5], int N, int C, int H, int W, void *ws,
                          size_t ws_bytes, void *stream);
int erf_nb1d_chain_forward_bf16(const void *x, const erf_nb1d_block_t *blocks, int nblocks, int N,
                                int C, int H, int W, void *ws, size_t ws_bytes, void *stream);
int erf_nb1d_backward_bf16(const void *x, const float *const w[4], const erf_bn_t *bn1,
                           const erf_bn_t *bn2, const float *drop_scale, int dil, const void *y,
                           const void *const save[5], const void *gy, void *dx, float *const dw[4],
                           float *const db[4], float *const dgamma[2], float *const dbeta[2], int N,
                           int C, int H, int W, int flags, void *ws, size_t ws_bytes, void *stream);
int erf_nb1d_backward_bits_bf16(const void *x, const float *const w[4], const erf_bn_t *bn1,
                                const erf_bn_t *bn2, const float *drop_scale, int dil, const void *y,
                                const void *const save[5], const unsigned *const sign_bits[4],
                                const void *gy, void *dx, float *const dw[4], float *const db[4],
                                float *const dgamma[2], float *const dbeta[2], int N, int C, int H,
                                int W, int flags, void *ws, size_t ws_bytes, void *stream);
/* packed: 2*3*C*C bf16 (MFMA weight fragments).  w_second may be NULL and dil_second 0: a single
 * convolution, y = act(conv_first(x) + b_first [+ residual]).  Lines longer than a tile (C = 128: 128
 * pixels, C = 64: 256) are accepted for a single convolution only. */
int erf_conv_pair_pack_bf16(const float *w_first, const float *w_second, int C, void *packed, void *stream);
int erf_conv_pair_forward_bf16(const void *x, const void *packed, const float *b_first,
                               const float *b_second, int first_axis, int dil_first, int dil_second,
                               int relu_first, int relu_second, const void *residual, void *mid,
                               void *y, int N, int C, int H, int W, void *stream);
/* ---- One 3-tap 1-D convolution C -> C (C = 64 / 128) of a non_bottleneck_1d (train/erfnet_blocks.py:
 *      57-66) on fp32 tensors, computed on the bf16 matrix pipe with split operands (see
 *      ERF_OPT_SPLIT_BF16) — the kernel the fp32 training step spends most of its time in.
 * y = [mask > 0 ?] act(conv(x) + bias [+ residual]) [: 0];  axis 0 = kernel (3,1) along H, 1 = kernel
 * (1,3) along W.  packed: 3 * 3*C*C bf16 from erf_conv_line_pack_split (w in torch's layout [C][C][3]).
 * x, y, residual, mask: NHWC, 16-byte aligned, y distinct from x. */
int erf_conv_line_pack_split(const float *w, int C, void *packed, void *stream);
int erf_conv_line_forward_split(const float *x, const void *packed, const float *bias, int axis, int dil,
                                int relu, const float *residual, const float *mask, float *y, int N,
                                int C, int H, int W, void *stream);
int erf_upsampler_forward_bf16(const void *x, const float *w, const float *b, const erf_bn_t *bn,
                               void *y, void *pre, int N, int Cin, int H, int W, int Cout, void *ws,
                               size_t ws_bytes, void *stream);
int erf_upsampler_backward_bf16(const void *x, const float *w, const erf_bn_t *bn, const void *y,
                                const void *pre, const void *gy, void *dx, float *dw, float *db,
                                float *dgamma, float *dbeta, int N, int Cin, int H, int W, int Cout,
                                int flags, void *ws, size_t ws_bytes, void *stream);
/* y_is_bf16 = 0: fp32 logits (they feed the loss in training); 1: bf16 logits (inference). */
int erf_outconv_forward_bf16(const void *x, const float *w, const float *b, void *y, int y_is_bf16,
                             int N, int Cin, int H, int W, int ncls, int softmax, void *stream);
int erf_outconv_argmax_bf16(const void *x, const float *w, const float *b, long long *labels, int N,
                            int Cin, int H, int W, int ncls, void *stream);
/* gy: fp32 gradient of the fp32 logits; dx bf16. */
int erf_outconv_backward_bf16(const void *x, const float *w, const float *gy, void *dx, float *dw,
                              float *db, int N, int Cin, int H, int W, int ncls, int flags, void *ws,
                              size_t ws_bytes, void *stream);
int erf_outconv_ce_forward_bf16(const void *x, const float *w, const float *b, const long long *target,
                                float *stats, float *lse, long long *counts, int N, int Cin, int H, int W,
                                int ncls, long long ignore_index, void *ws, size_t ws_bytes, void *stream);
int erf_outconv_ce_backward_bf16(const void *x, const float *w, const float *b, const long long *target,
                                 const float *lse, const float *stats, const float *gy, void *dx, float *dw,
                                 float *db, int N, int Cin, int H, int W, int ncls, long long ignore_index,
                                 int flags, void *ws, size_t ws_bytes, void *stream);
int erf_conv2d_forward_bf16(const void *x, const float *w, const float *b, void *y, int N, int Cin,
                            int H, int W, int Cout, int k, int relu, void *ws, size_t ws_bytes,
                            void *stream);
int erf_conv2d_backward_bf16(const void *x, const float *w, const void *y, const void *gy, void *dx,
                             float *dw, float *db, int N, int Cin, int H, int W, int Cout, int k,
                             int relu, void *ws, size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ERFNET_HIP_H */
