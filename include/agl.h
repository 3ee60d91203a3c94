/* agl.h — C ABI of libagl.so: the MI355X (gfx950) kernels behind the G+D train-step hot path of
 * ubc-vision/attribute-guided-image-generation-from-layout.
 *
 * The reference has no FFI of its own: its boundary is the Python nn.Module surface, and every number
 * is produced by a PyTorch op.  Each entry point below therefore names the torch
 * call site(s) in the reference it replaces (paths relative to the reference root).  The Python host
 * (attribute-guided-image-generation-from-layout_amd/agl/lib.py) binds these with ctypes; see
 * INTEGRATION.md for the binding a maintainer of the reference would add.
 *
 * Conventions: all tensors fp32, NCHW, contiguous, device memory borrowed for the duration of the
 * call; int64 index tensors (`long long`); `stream` is a hipStream_t (NULL = default stream); every
 * function only enqueues work (no allocation, no synchronisation) and returns 0 on success, non-zero
 * on a rejected call with a message in agl_last_error().  Workspaces are caller-provided.
 */
#ifndef AGL_H
#define AGL_H
#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped with every signature change; the Python binding refuses to bind a library of another version. */
#define AGL_ABI_VERSION 8
int agl_version(void);
const char* agl_last_error(void);

/* ---- convolutions (fp32 implicit GEMM on MFMA 32x32x2 f32) -------------------------------------
 * F.conv2d / nn.Conv2d call sites: models/generator_obj_att.py:374-386 (CropEncoder), :474-483
 * (LayoutEncoder), :93 (ConvLSTMCell), :53,56 (ResidualBlock), :432,435 (GlobalEncoder), :528,544
 * (Decoder); models/generator_obj_att128.py:549-557 (c5,c6,c7); models/spade/networks/normalization.py:87-91;
 * models/discriminator.py:37,39,44,72,74,79; nn.Linear (:218,252,253,162,122; generator :392,393,582-586)
 * is the ks=1, H=W=1 case.  ks in {1,3,4,5,7}; stride in {1,2}.
 *   up_log2  : x is nearest-upsampled by 2^up_log2 on the fly (F.interpolate(...,'nearest'),
 *              normalization.py:100 and generator_obj_att128.py:588) — H,W are the stored sizes.
 *   in_relu  : relu applied to x while gathering (discriminator.py:71 in-place ReLU).
 *   relu     : relu on the output;  accumulate: y += result (before relu).
 *   ws       : optional split-K scratch (agl_conv2d_splitk_ws_bytes); without it small grids run unsplit.      */
/* flags (per call; no process-wide switches):
 *   AGL_CONV_BF16        MFMA operands rounded to bf16 (RNE), fp32 accumulation (BASELINE configs 3/5); default is exact
 *                        fp32 MFMA (config 2).  Statistics, SN, losses and Adam are fp32 either way.
 *   AGL_CONV_NO_PATCH    never use the LDS-patch kernel (A/B tests);  AGL_CONV_NO_PATCH_S2: not its stride-2 form
 *   AGL_CONV_NO_POS      never use the position-major path on <= 8x8 maps (A/B tests)
 *   AGL_CONV_POS_ALL_KS  experiments: position-major path also for 3x3 / 4x4 kernels
 *   AGL_CONV_SPLIT3      fp32 tensors, fp32-accurate products on the 16-bit matrix cores (the name is historical): every
 *                        operand is carried as several 16-bit terms and the partial products that matter are accumulated
 *                        in fp32.  agl_conv2d_split_products() says which form the library was built with:
 *                          3  fp16 hi / lo terms (a = hi + lo' * 2^-11 to ~2^-23) under power-of-two block scales found
 *                             when the operand is staged / packed; hi*hi, hi*lo', lo'*hi (dropped: <= 2^-22 |ab|);
 *                          6  three bf16 terms (a = a1+a2+a3 to 2^-27), six of the nine partial products.
 *                        Used by the LDS-patch kernels of csrc/pconv.hip where they apply (1x1 / 3x3 / 5x5 stride 1,
 *                        4x4 / 3x3 stride 2 and their gradients on 4/8/16n-wide maps), exact fp32 MFMA elsewhere.  The
 *                        accuracy class is tested, not assumed: error against fp64 <= 2x that of the exact fp32 MFMA
 *                        kernel on the same problem (tests/test_ops_gpu.py)
 *   AGL_CONV_ANY_GRID    take the matrix-core kernels of csrc/pconv.hip also for grids below their occupancy threshold
 *                        (by default a launch of < 200 workgroups falls back to the split-K fp32 kernels, which are
 *                        faster there); lets unit tests exercise those kernels on small tensors                       */
#define AGL_CONV_BF16 1
#define AGL_CONV_NO_PATCH 2
#define AGL_CONV_NO_PATCH_S2 4
#define AGL_CONV_NO_POS 8
#define AGL_CONV_POS_ALL_KS 16
#define AGL_CONV_SPLIT3 32
#define AGL_CONV_ANY_GRID 64
#define AGL_CONV_PRIO 256 /* wave priority 1 for the conversion / LDS-store bursts of the patch kernels (A/B switch) */
#define AGL_CONV_W8 128   /* eight-wave (512-thread) workgroups in the split-mode stride-1 3x3 / 5x5 patch kernels: same tile and LDS
                           * footprint, twice the waves per SIMD (A/B switch; the host mirror sets it where it measured faster) */
#define AGL_CONV_X_BF16 (1 << 17) /* agl_conv2d_fwd / agl_conv2d_bwd_weight: x points to bf16 elements (same extents; 2 bytes each).  For a
                           * tensor that only bf16-mode convolutions read and whose producer wrote it in bf16 (agl_box2_fwd_bf16): they
                           * would round the fp32 tensor to these very values when staging it, so results are identical and the tensor
                           * costs half the HBM traffic.  Needs AGL_CONV_BF16 and a shape the matrix-core kernels take
                           * (agl_conv2d_fwd_packed_bytes != 0 / agl_conv2d_bwd_weight_takes_bf16_x); otherwise the call is rejected. */
#define AGL_CONV_Y_BF16 (1 << 18) /* agl_conv2d_fwd: y points to bf16 elements (round to nearest even; no accumulate) — only the few-input-
                           * channel stream kernel (Cin <= 4, 1x1 / 3x3, stride 1, W % 4 == 0) writes that form; otherwise rejected */
#define AGL_CONV_DY_BF16 (1 << 20) /* agl_conv2d_bwd_weight: dy points to bf16 elements; needs agl_conv2d_bwd_weight_takes_bf16_dy(...) == 1 */
#define AGL_CONV_MASK_BF16 (1 << 19) /* agl_conv2d_bwd_data: pos_mask points to bf16 elements (the bf16-stored output of the producer
                           * whose ReLU backward this call applies); needs agl_conv2d_bwd_data_takes_bf16_mask(...) == 1 */
long agl_conv2d_fwd_ws_bytes(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2);
long agl_conv2d_bwd_data_ws_bytes(int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int ks, int stride, int pad);
long agl_conv2d_splitk_ws_bytes(int M, long out_pixels, int phases, int K, long out_numel);
/* packed_w / packed_div (optional, both may be NULL): the weights already in the packed form of the matrix-core patch
 * kernel (agl_conv2d_pack_weights) — the call then launches no per-call weight re-pack.  The packed tensor w0 may differ from w
 * by a scalar divisor kept on the device, w = w0 / *packed_div: a spectrally normalised layer (discriminator.py:15-22) packs
 * weight_orig once per optimiser update and passes sigma, which changes with every forward call.  Used only when the call runs
 * on the patch kernel (agl_conv2d_*_packed_bytes != 0 for the same extents and flags); other launches read w, which may be NULL
 * only when packed_w is given (the call is rejected if it then cannot run on the patch kernel). */
int agl_conv2d_fwd(const float* x, const float* w, const void* packed_w, const float* packed_div, const float* bias, float* y, void* ws,
                   long ws_bytes, int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2, int in_relu, int relu,
                   int accumulate, int flags, void* stream);
long agl_conv2d_fwd_packed_bytes(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2, int flags);
long agl_conv2d_bwd_data_packed_bytes(int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int ks, int stride, int pad, int flags);
/* pass 0: packed form for agl_conv2d_fwd; 1: for agl_conv2d_bwd_data (stride 1 "same" form, or stride 2 with ks = 4).  It depends on
 * (Cin, Cout, ks, stride, arithmetic flags) only, so one packed buffer serves every call of that layer until the weights change. */
int agl_conv2d_pack_weights(const float* w, void* packed, long packed_bytes, int pass, int Cin, int Cout, int ks, int stride, int flags,
                            void* stream);
/* All packs of a parameter arena in ONE launch (ABI 6): a caller that re-packs every weight after each optimiser step (torch.optim.Adam.step
 * in train64.py:262 / :370) describes each pack once — agl_conv2d_pack_desc fills one host row of AGL_PACK_DESC_WORDS 64-bit words with the
 * same arguments as agl_conv2d_pack_weights; word 13 is the row's block count, word 12 its first block: the running sum, filled by the
 * caller — uploads the table, and calls agl_conv2d_pack_many(table, rows, total blocks) after every update. */
#define AGL_PACK_DESC_WORDS 14
int agl_conv2d_pack_desc(const float* w, void* packed, long packed_bytes, int pass, int Cin, int Cout, int ks, int stride, int flags,
                         long long* row);
int agl_conv2d_pack_many(const void* rows_dev, int n, long total_blocks, void* stream);
/* Arithmetic pipe of the main kernel the LAST agl_conv2d_fwd / _fwd_stats / _bwd_data / _bwd_weight call of the calling thread
 * launched: 0 = exact fp32 (fp32 MFMA or fp32 VALU), 1 = bf16 MFMA, one product per multiply-add (AGL_CONV_BF16), 3 = bf16 MFMA
 * with split operands (AGL_CONV_SPLIT3), agl_conv2d_split_products() products per multiply-add.  bench.py prices each launch
 * against that pipe's peak. */
int agl_conv2d_last_pipe(void);
/* Matrix-core products per fp32 multiply-add of the AGL_CONV_SPLIT3 arithmetic this library was built with: 3 (fp16 hi / lo
 * terms) or 6 (three bf16 terms).  The few-channel 7x7 vertical form always uses the six-product form. */
int agl_conv2d_split_products(void);
/* The same forward (no output ReLU, no accumulate) that may also leave the BatchNorm partial sums of its output in
 * `stats` — the statistics pass of the nn.BatchNorm2d that follows the convolution (generator_obj_att.py:54-57, 433, 583;
 * normalization.py:77-78 + 97) then needs no read of y.  stats: stats_floats floats (agl_conv2d_fwd_stats_floats());
 * *stat_rows = rows written, each row = [Cout][{count, mean, M2 = sum of squared deviations from that mean}] over a disjoint set
 * of output pixels (shift-invariant: a channel offset of 100 standard deviations costs no accuracy) — 0 when the
 * launch that ran does not produce them (the caller then uses agl_bn_stats).  Feed the rows to agl_bn_stats_from_partials. */
long agl_conv2d_fwd_stats_floats(int N, int Cout, int OH, int OW);
int agl_conv2d_fwd_stats(const float* x, const float* w, const void* packed_w, const float* packed_div, const float* bias, float* y,
                         void* ws, long ws_bytes, int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2,
                         int in_relu, int flags, float* stats, long stats_floats, int* stat_rows, void* stream);
/* Gradient wrt the input of the conv above; ALSO the forward of nn.ConvTranspose2d(k=4,s=2,p=1)
 * (generator_obj_att.py:532,536,540) with w stored [C_in_T][C_out_T][4][4].  pos_mask (optional, shaped
 * like dx): dx is zeroed where pos_mask <= 0 (backward of a fused input ReLU).                      */
int agl_conv2d_bwd_data(const float* dy, const float* w, const void* packed_w, const float* packed_div, const float* bias,
                        const float* pos_mask, float* dx, void* ws, long ws_bytes, int N, int Cin, int IH, int IW, int Cout, int OH, int OW,
                        int ks, int stride, int pad, int relu, int accumulate, int flags, void* stream);
long agl_conv2d_bwd_weight_ws_bytes(int N, int Cin, int Cout, int ks, int OH, int OW);
int agl_conv2d_bwd_weight_takes_bf16_x(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int flags);
int agl_conv2d_bwd_data_takes_bf16_mask(int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int ks, int stride, int pad, int flags);
/* bf16-stored operands of the other passes (bf16 arithmetic only; each producer asks the matching predicate before it writes bf16):
 * AGL_CONV_Y_BF16 on agl_conv2d_fwd also covers the matrix-core patch kernel (no reduction split): agl_conv2d_fwd_writes_bf16_y;
 * AGL_CONV_X_BF16 on agl_conv2d_bwd_data: dy holds bf16 (the bf16-stored input of a ConvTranspose2d, generator_obj_att.py:532-540,
 * whose forward IS this call): agl_conv2d_bwd_data_takes_bf16_dy;  AGL_CONV_DY_BF16 on agl_conv2d_bwd_weight: dy holds bf16 (the
 * same tensor in the weight gradient of that transposed convolution): agl_conv2d_bwd_weight_takes_bf16_dy. */
int agl_conv2d_fwd_takes_bf16_x(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int flags);
/* Channel-blocked bf16 activations (ABI 7): a tensor [N][C/8][H][W][8] of bf16 — the 8 channels of a pixel are one 16-byte piece, which is
 * what the matrix-core kernels stage (one aligned load per piece instead of eight element loads a channel stride apart, no conversion)
 * and what their epilogue stores whole.  Used inside the discriminators' block chain (agl/dtrunk.py: models/discriminator.py:29-99 with
 * h = relu(c1(.)) and the block outputs in this layout); reference counterpart of every call: the same nn.Conv2d / avg_pool2d on a
 * permuted tensor.  Per-operand flags (each implies the matching _BF16 flag, which must be set too; bf16 arithmetic only):
 *   AGL_CONV_X_BLOCKED     x of agl_conv2d_fwd / _addend / _shortcut and of agl_conv2d_bwd_weight (3x3 stride 1, 4x4 stride 2)
 *   AGL_CONV_Y_BLOCKED     y of agl_conv2d_fwd (3x3 stride 1 with a blocked x; the <= 4-input-channel 3x3 stream kernel), of
 *                          agl_conv2d_fwd_addend (1x1, fp32 NCHW x and addend) and of agl_conv2d_fwd_shortcut
 *   AGL_CONV_MASK_BLOCKED  pos_mask of agl_conv2d_bwd_data (with AGL_CONV_MASK_BF16; dy and dx stay fp32 NCHW)
 * agl_conv2d_fwd_takes_blocked says whether the forward forms exist for the extents; the weight gradient and the masked input
 * gradient take a blocked operand whenever they take the bf16 one (agl_conv2d_bwd_weight_takes_bf16_x / _bwd_data_takes_bf16_mask)
 * with ks 3 / stride 1 or ks 4 / stride 2. */
#define AGL_CONV_X_BLOCKED (1 << 21)
#define AGL_CONV_Y_BLOCKED (1 << 22)
#define AGL_CONV_MASK_BLOCKED (1 << 23)
#define AGL_CONV_BLOCKED (AGL_CONV_X_BLOCKED | AGL_CONV_Y_BLOCKED)
/* agl_conv2d_fwd / agl_conv2d_bwd_data with AGL_CONV_DEFER_SUM: when the kernel that runs cuts the reduction over workgroups (small
 * output grids: the recurrence steps of LayoutConvLSTM, generator_obj_att.py:99-104, :306-331) and the call has nothing but the sum left
 * to do (no bias, mask, accumulation, ReLU, divisor), the partial outputs are LEFT in ws and y / dx is NOT written: the caller's next
 * kernel on the same stream adds them itself (agl_lstm_gates_fwd_sum / _bwd_sum, in the epilogue's own order: identical numbers),
 * one launch less per step.  agl_conv2d_deferred reports, for the calling thread's last such call: splits >= 2, the first slab and the
 * distance between slabs in floats — or splits = 0 when the call wrote y / dx as usual.  The slabs live until ws is used again. */
#define AGL_CONV_DEFER_SUM (1 << 24)
int agl_conv2d_deferred(const float** slabs, int* splits, long long* stride);
int agl_conv2d_fwd_takes_blocked(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int flags);
/* NCHW fp32 (x_bf16 = 0) or bf16 (1) -> channel-blocked bf16; avg_pool2(relu?(x)) of a blocked x into an fp32 NCHW y (the shortcut's pool,
 * discriminator.py:58-60, :97-99); the backward of that pool with the ReLU mask read from the blocked x (dy, dx fp32 NCHW). */
int agl_to_blocked(const void* x, void* y_blk, int N, int C, int H, int W, int x_bf16, void* stream);
int agl_avgpool2_fwd_xblk(const void* x_blk, float* y, int N, int C, int H, int W, int in_relu, void* stream);
int agl_avgpool2_bwd_xblk(const float* dy, const void* x_blk, float* dx, int N, int C, int H, int W, int accumulate, void* stream);
int agl_conv2d_fwd_writes_bf16_y(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2, int relu, int accumulate,
                                 int flags);
int agl_conv2d_bwd_data_takes_bf16_dy(int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int ks, int stride, int pad, int flags);
int agl_conv2d_bwd_weight_takes_bf16_dy(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int flags);
/* ---- normalise-modulate folded into the consuming convolution (BASELINE north_star; SURVEY a2 "apply+ReLU into next conv prologue").
 * conv(relu(CondBN(x))) — generator_obj_att.py:395-416 (CropEncoder), :494-504 (LayoutEncoder), :437-446 (GlobalEncoder) — without
 * ever storing the normalised tensor: agl_norm_fold_table turns (mean, rstd, affine / class-table parameters) into per-(row, channel)
 * tables scale = rstd * gamma, shift = beta - mean * scale (row = object for ConditionalBatchNorm2d, one row otherwise); agl_conv2d_fwd_fold
 * applies v = fma(x, scale[r][c], shift[r][c]) (then the ReLU, then the zero padding) while it stages its input patch — forms of
 * the matrix-core patch kernel (agl_conv2d_fwd_fold_ok), optionally leaving the BatchNorm partial rows of its own output like
 * agl_conv2d_fwd_stats; agl_conv2d_bwd_weight_fold applies the same transform to the raw x in the weight gradient; agl_norm_bwd_fold
 * is agl_norm_bwd for a y that does not exist: the ReLU mask is recomputed from x with the staging pass's expression.
 * in_mean is NOT read by either convolution entry (NULL is accepted): the tables already carry the mean.  The fp32 shift re-admits an
 * error of about |mean|/std * 2^-24 per element that the two-pass (x - mean) * rstd form does not have — three orders below the bf16
 * operand rounding of the forms that accept a fold (tests/test_ops_gpu.py holds it to that bar at |mean|/std = 100). */
int agl_norm_fold_table(const float* mean, const float* rstd, int mode, const float* p0, const float* p1, const long long* labels, int N, int C,
                        float* scale, float* shift, void* stream);
int agl_conv2d_fwd_fold_ok(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int flags);
int agl_conv2d_fwd_fold(const float* x, const float* in_mean, const float* in_scale, const float* in_shift, int in_per_n, const float* w,
                        const void* packed_w, const float* packed_div, const float* bias, float* y, void* ws, long ws_bytes, int N, int Cin,
                        int H, int W, int Cout, int ks, int stride, int pad, int in_relu, int flags, float* stats, long stats_floats,
                        int* stat_rows, void* stream);
int agl_conv2d_bwd_weight_fold_ok(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int flags);
int agl_conv2d_bwd_weight_fold(const float* dy, const float* x, const float* in_mean, const float* in_scale, const float* in_shift,
                               int in_per_n, float* dw, float* dbias, int dbias_accumulate, int* dbias_done, void* ws, long ws_bytes, int N,
                               int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int in_relu, int accumulate,
                               int flags, void* stream);
/* ---- SPADE applied by the convolution that reads it (BASELINE north_star "SPADE normalization ... fused with the following conv").
 * relu(SPADE(x)) = relu((x - mean[c]) * rstd[c] * (1 + gamma[n][c][cell]) + beta[n][c][cell]) — models/spade/networks/normalization.py:97,106
 * with param_free_norm = BatchNorm2d — in front of the 128 px decoder's c6 (5x5) and c7 (7x7 to 3 channels),
 * models/generator_obj_att128.py:588-597, where gamma|beta live on a G x G class grid of ~10 % of the activation (pixel (iy, ix) reads
 * cell map[iy] * G + map[ix]): the modulated tensor is never stored.  agl_spade_cells lays gamma|beta (N, 2C, G, G) out for 16-byte
 * reads, cells[((n * C/8 + c/8) * G*G + cell) * 16 + j] = 1 + gamma of channel 8*(c/8) + j, [.. + 8 + j] = beta; agl_conv2d_fwd_spade /
 * agl_conv2d_bwd_weight_spade evaluate the stand-alone apply's own expression (csrc/spade.h) while they stage x — the same bits as
 * agl_norm_apply_fwd_y16 followed by the AGL_CONV_X_BF16 calls, bf16 arithmetic only (the *_ok predicates say for which extents);
 * agl_norm_bwd_spade is agl_norm_bwd for mode 3 with the ReLU mask recomputed from x (same expression again). */
int agl_spade_cells(const float* gb, int N, int C, int G, float* cells, void* stream);
int agl_conv2d_fwd_spade_ok(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int flags);
int agl_conv2d_fwd_spade(const float* x, const float* mean, const float* rstd, const float* cells, const int* map, int G, const float* w,
                         const void* packed_w, const float* packed_div, const float* bias, float* y, void* ws, long ws_bytes, int N, int Cin,
                         int H, int W, int Cout, int ks, int stride, int pad, int in_relu, int flags, float* stats, long stats_floats,
                         int* stat_rows, void* stream);
int agl_conv2d_bwd_weight_spade_ok(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int flags);
int agl_conv2d_bwd_weight_spade(const float* dy, const float* x, const float* mean, const float* rstd, const float* cells, const int* map, int G,
                                float* dw, float* dbias, int dbias_accumulate, int* dbias_done, void* ws, long ws_bytes, int N, int Cin, int H,
                                int W, int Cout, int OH, int OW, int ks, int stride, int pad, int in_relu, int accumulate, int flags,
                                void* stream);
int agl_norm_bwd_spade(const float* dy, const float* x, const float* mean, const float* rstd, const float* gb, int relu, int batch_stats,
                       float* dx, float* dgb, int N, int C, int HW, const int* gb_map, const int* gb_lo, int W, int src_w, void* ws,
                       long ws_bytes, void* stream);
/* y = conv(x) + addend (+ bias, output ReLU) written out of place; with AGL_CONV_Y_BF16 the fp32 sum is rounded once to bf16.
 * The shortcut sum of a discriminator block (discriminator.py:58-60, :97-99) in bf16 arithmetic.  Matrix-core patch kernel only
 * (agl_conv2d_fwd_writes_bf16_y / agl_conv2d_fwd_packed_bytes for the same extents). */
int agl_conv2d_fwd_addend(const float* x, const float* w, const void* packed_w, const float* packed_div, const float* bias, const float* addend,
                          float* y, void* ws, long ws_bytes, int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int in_relu,
                          int relu, int flags, void* stream);
/* y = conv3x3(x) + bias + [sc_bias[m] + sum_c sc_w[m][c] * sc_x[n][c][pixel]] (+ ReLU): the learnable 1x1 shortcut of the discriminators'
 * first block (discriminator.py:43-44, :58-60: out = resi(x) + sc(x), x has 3 channels) evaluated in the 3x3 convolution's epilogue —
 * one launch reads 3 input channels instead of a second launch adding a 64-channel tensor.  x may hold bf16 (AGL_CONV_X_BF16), y bf16
 * (AGL_CONV_Y_BF16).  bf16 arithmetic, Cout <= 64 (agl_conv2d_fwd_shortcut_ok). */
int agl_conv2d_fwd_shortcut_ok(int N, int Cin, int H, int W, int Cout, int ks, int pad, int flags);
int agl_conv2d_fwd_shortcut(const float* x, const float* w, const void* packed_w, const float* packed_div, const float* bias, const float* sc_x,
                            const float* sc_w, const float* sc_bias, int sc_cin, float* y, void* ws, long ws_bytes, int N, int Cin, int H, int W,
                            int Cout, int ks, int pad, int in_relu, int relu, int flags, void* stream);
/* dbias / dbias_done (optional, both NULL or both set): the bias gradient db[Cout] = sum over (n, oh, ow) of dy, added to dbias
 * when dbias_accumulate (its own flag: a spectrally normalised layer returns dw fresh but accumulates db in place).  The
 * matrix-core weight-gradient kernel forms it from the dy tiles it stages anyway; *dbias_done (host int) is 1 when the call did so
 * and 0 when the kernel that ran does not (the caller then uses agl_channel_sum). */
int agl_conv2d_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, int dbias_accumulate, int* dbias_done, void* ws,
                          long ws_bytes, int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int up_log2,
                          int in_relu, int accumulate, int flags, void* stream);
/* Executed FLOPs (2*MAC) of the launches one such call issues (dense count minus the padded taps the position-major
 * path skips) — what bench.py's roofline leg divides by the measured launch time. */
double agl_conv2d_fwd_flops(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2, int flags);
double agl_conv2d_bwd_data_flops(int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int ks, int stride, int pad,
                                 int flags);
double agl_conv2d_bwd_weight_flops(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad,
                                   int up_log2, int in_relu, int flags);

/* ---- batch-statistics normalisation --------------------------------------------------------------
 * nn.BatchNorm2d/1d in training mode (generator_obj_att.py:35,54,57,433,583,585; normalization.py:78):
 * per-channel mean / rstd over (N,HW), running stats with momentum and unbiased variance.            */
long agl_bn_stats_ws_bytes(int N, int C, int HW);
/* moments (optional, 2*C doubles): the batch mean and unbiased variance per channel exactly as the running update used them;
 * agl_bn_running_update re-applies that update from them (a second evaluation of the same layer on the same batch —
 * train64.py evaluates the generator twice per iteration with unchanged weights — without reading the activation again). */
int agl_bn_stats(const float* x, int N, int C, int HW, float eps, float momentum, float* mean, float* rstd,
                 float* running_mean, float* running_var, long long* num_batches_tracked, double* moments, void* ws, long ws_bytes,
                 void* stream);
int agl_bn_running_update(const double* moments, int C, float momentum, float* running_mean, float* running_var,
                          long long* num_batches_tracked, void* stream);
/* The same for n layers in ONE launch (n <= AGL_BN_UPDATE_MAX): the replay of a whole evaluation's tape.  Items are applied in
 * array order per channel, so a layer may appear more than once (its updates then chain as they would in separate calls). */
#define AGL_BN_UPDATE_MAX 24
typedef struct {
  const double* moments; float* running_mean; float* running_var; long long* num_batches_tracked; /* may be NULL */
  int C; float momentum;
} AglBnUpdate;
int agl_bn_running_update_multi(const AglBnUpdate* items, int n, void* stream);
int agl_bn_stats_eval(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* rstd,
                      void* stream);
/* agl_bn_stats from the partial rows of agl_conv2d_fwd_stats (count = N*HW elements per channel); rows are added in
 * double in a fixed order.  Same running-statistics update. */
int agl_bn_stats_from_partials(const float* partials, int rows, int C, long count, float eps, float momentum, float* running_mean,
                               float* running_var, long long* num_batches_tracked, float* mean, float* rstd, double* moments,
                               void* stream);
/* y = modulate(xhat) (+residual) (relu).  mode 0: none; 1: gamma[C],beta[C] (BatchNorm affine);
 * 2: table[V][2C] indexed by labels[N] (ConditionalBatchNorm2d, generator_obj_att.py:40-44);
 * 3: gb[N][2C][HW], y = xhat*(1+gamma)+beta (SPADE, normalization.py:106).                           */
/* gb_map (optional, mode 3 only; NULL = none): gamma|beta given on a coarser block-class grid — p0 is (N, 2C, src_w, src_w), the
 * map is square W x W (HW = W*W) and pixel (iy, ix) reads cell (gb_map[iy], gb_map[ix]) (W device ints): the nearest /
 * block-class expansion of normalization.py:100 (agl_grid_gather_fwd) folded into the reads instead of written out. */
int agl_norm_apply_fwd(const float* x, const float* mean, const float* rstd, int mode, const float* p0, const float* p1,
                       const long long* labels, const float* residual, int relu, float* y, int N, int C, int HW,
                       const int* gb_map, int W, int src_w, void* stream);
long agl_norm_bwd_ws_bytes(int N, int C);
/* dp0 / dp1: parameter gradients.  mode 1: dgamma[C], dbeta[C], overwritten — or added to when param_accumulate (gradient
 * accumulated in place, like autograd's AccumulateGrad); mode 2: dtable[V][2C], always added to; mode 3: dp0 = d(gb), overwritten —
 * at the FULL resolution (N, 2C, HW) also when gamma|beta are read through gb_map (reduce it with agl_grid_gather_bwd) — unless
 * gb_lo is given as well (the range starts of the map's inverse, src_w + 1 device ints; W = 64 or 128): then dp0 is
 * (N, 2C, src_w, src_w), reduced to the class grid inside the row pass. */
int agl_norm_bwd(const float* dy, const float* x, const float* y, const float* mean, const float* rstd, int mode,
                 const float* p0, const float* p1, const long long* labels, int relu, int batch_stats, float* dx,
                 float* dp0, float* dp1, int N, int C, int HW, int n_classes, int param_accumulate, const int* gb_map, const int* gb_lo,
                 int W, int src_w, void* ws, long ws_bytes, void* stream);
/* bf16-stored forward output (bf16 arithmetic): agl_norm_apply_fwd_y16 writes y as bf16 (round to nearest even) — for a tensor whose
 * only readers are bf16-mode convolutions, which would round the fp32 tensor to these very values when staging it (AGL_CONV_X_BF16 /
 * AGL_CONV_DY_BF16) — and agl_norm_bwd_y16 reads that tensor for the ReLU mask.  Same arguments as the fp32 forms otherwise. */
int agl_norm_apply_fwd_y16(const float* x, const float* mean, const float* rstd, int mode, const float* p0, const float* p1,
                           const long long* labels, const float* residual, int relu, void* y_bf16, int N, int C, int HW,
                           const int* gb_map, int W, int src_w, void* stream);
int agl_norm_bwd_y16(const float* dy, const float* x, const void* y_bf16, const float* mean, const float* rstd, int mode,
                     const float* p0, const float* p1, const long long* labels, int relu, int batch_stats, float* dx,
                     float* dp0, float* dp1, int N, int C, int HW, int n_classes, int param_accumulate, const int* gb_map, const int* gb_lo,
                     int W, int src_w, void* ws, long ws_bytes, void* stream);
int agl_norm_bwd_fold(const float* dy, const float* x, const float* mean, const float* rstd, const float* fold_scale, const float* fold_shift,
                      int fold_per_n, int mode, const float* p0, const float* p1, const long long* labels, int relu, int batch_stats,
                      float* dx, float* dp0, float* dp1, int N, int C, int HW, int n_classes, int param_accumulate, void* ws, long ws_bytes,
                      void* stream);

/* ---- per-object bilinear crop (models/bilinear.py:26 crop_bbox_batch -> :107 crop_bbox -> F.grid_sample :136)
 * out[b] = bilinear resample of feats[box_to_img[b]] over boxes[b]=[x0,y0,x1,y1] in [0,1]; zero padding;
 * align_corners as in torch (default 0).  Backward scatter-adds into dfeats (caller zero-fills).  A box whose
 * box_to_img is outside [0,N) never touches memory: its crop is filled with NaN / its gradient is dropped (the
 * reference asserts on the host, bilinear.py:122-123; the Python host mirror validates CPU-resident indices). */
int agl_crop_fwd(const float* feats, const float* boxes, const long long* box_to_img, float* out, int N, int B, int C,
                 int H, int W, int HH, int WW, int align_corners, void* stream);
int agl_crop_bwd(const float* dout, const float* boxes, const long long* box_to_img, float* dfeats, int N, int B, int C,
                 int H, int W, int HH, int WW, int align_corners, void* stream);
/* The same gradient for a NON-DECREASING box_to_img (the boxes of an image form one contiguous run — what bilinear.py:77-90 builds
 * and train64.py always passes): a gather in a fixed order instead of a scatter with atomics, bit-reproducible from run to run.
 * The caller vouches for the order (the Python host checks it where the map lives on the CPU, as in the reference's loop). */
int agl_crop_bwd_sorted(const float* dout, const float* boxes, const long long* box_to_img, float* dfeats, int N, int B, int C,
                        int H, int W, int HH, int WW, int align_corners, void* stream);

/* ---- ConvLSTM gate math (generator_obj_att.py:105-112), gates stored post-activation in i,f,o,g order */
int agl_lstm_gates_fwd(const float* ccx, const long long* rows, const float* cch, const float* c_prev, float* h, float* c,
                       float* gates, int B, int hid, int S, void* stream);
int agl_lstm_gates_bwd(const float* dh_a, const float* dh_b, int Bb, const float* dc_next, int Bc, const float* gates,
                       const float* c_prev, const float* c, float* dcc, float* dc_prev, int B, int hid, int S,
                       void* stream);
/* the same with cch / dh_b given as `splits` unreduced partial outputs `stride` floats apart (AGL_CONV_DEFER_SUM; splits <= 1: one tensor) */
int agl_lstm_gates_fwd_sum(const float* ccx, const long long* rows, const float* cch, int cch_splits, long long cch_stride,
                           const float* c_prev, float* h, float* c, float* gates, int B, int hid, int S, void* stream);
int agl_lstm_gates_bwd_sum(const float* dh_a, const float* dh_b, int dh_b_splits, long long dh_b_stride, int Bb, const float* dc_next, int Bc,
                           const float* gates, const float* c_prev, const float* c, float* dcc, float* dc_prev, int B, int hid, int S,
                           void* stream);

/* ---- small data movement / reductions ------------------------------------------------------------ */
int agl_relu_bwd(const float* dy, const float* y, float* dx, long n, void* stream);
int agl_axpby(const float* a, const float* b, float alpha, float beta, float* out, long n, void* stream);
int agl_gather_rows(const float* src, const long long* rows, float* out, long R, long len, int accumulate, void* stream);
int agl_scatter_rows(const float* src, const long long* rows, float* out, long R, long len, void* stream);
/* F.avg_pool2d(k=2) (discriminator.py:25-26) / nn.AdaptiveAvgPool2d(8) on 16x16 (generator_obj_att128.py:486) */
/* 2x2 stride-1 box filter over the zero-extended map ((H+1) x (W+1) outputs): avg_pool2(conv3x3(x, pad 1)) is the 3x3
 * stride-2 unpadded convolution of it (models/discriminator.py:25-26,90-97 down-sampling blocks) */
int agl_box2_fwd(const float* x, float* xb, long NC, int H, int W, void* stream);
/* the same, written as bf16 (round to nearest even; W % 4 == 0) for a consumer that reads it with AGL_CONV_X_BF16 */
int agl_box2_fwd_bf16(const float* x, void* xb, long NC, int H, int W, void* stream);
int agl_box2_bwd(const float* dxb, const float* mask, float* dx, long NC, int H, int W, void* stream);
/* y[nc][Y][X] = x[nc][map_y[Y]][map_x[X]] for monotone non-decreasing index maps (device int arrays of H / W entries);
 * backward sums dy over each source cell's preimage: lo_y / lo_x hold h+1 / w+1 range starts (lo[h] = H).  Nearest
 * up-sampling (models/spade/networks/normalization.py:100) and the block-class grids of the SPADE restructure. */
int agl_grid_gather_fwd(const float* x, const int* map_y, const int* map_x, float* y, long NC, int h, int w, int H, int W, void* stream);
int agl_grid_gather_bwd(const float* dy, const int* lo_y, const int* lo_x, float* dx, long NC, int h, int w, int H, int W, void* stream);
int agl_avgpool2_fwd(const float* x, float* y, long NC, int H, int W, int in_relu, void* stream);
int agl_avgpool2_bwd(const float* dy, const float* x, float* dx, long NC, int H, int W, int in_relu, int accumulate,
                     void* stream);
/* The same pair for an x stored as bf16 (a discriminator block output that only convolutions and this pool read, bf16 arithmetic):
 * the forward reads 8 elements per 16-byte load (W % 8 == 0), the backward reads x for the ReLU mask only (always in_relu). */
int agl_avgpool2_fwd_x16(const void* x_bf16, float* y, long NC, int H, int W, int in_relu, void* stream);
int agl_avgpool2_bwd_x16(const float* dy, const void* x_bf16, float* dx, long NC, int H, int W, int accumulate, void* stream);
int agl_upsample_nearest_fwd(const float* x, float* y, long NC, int H, int W, int log2_factor, void* stream);
int agl_upsample_nearest_bwd(const float* dy, float* dx, long NC, int H, int W, int log2_factor, int accumulate,
                             void* stream);
/* torch.sum(h, dim=(2,3)) (discriminator.py:226; generator_obj_att.py:444) and AdaptiveAvgPool2d(1) (:389) */
int agl_sum_hw_fwd(const float* x, float* y, long NC, int HW, int in_relu, float scale, void* stream);
int agl_sum_hw_bwd(const float* dy, const float* x, float* dx, long NC, int HW, int in_relu, float scale, void* stream);
long agl_channel_sum_ws_bytes(int C);
int agl_channel_sum(const float* x, float* out, int N, int C, int HW, int accumulate, void* ws, long ws_bytes, void* stream);
/* z = eps*exp(.5*logvar)+mu (generator_obj_att.py:418-420) */
int agl_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, long n, void* stream);
int agl_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dlogvar, long n, void* stream);

/* torch.cat along channels of two NCHW tensors (b optionally (N,Cb) broadcast over HW) and its adjoint;
 * nn.Embedding backward as a fixed-order row accumulation (generator_obj_att.py:489,549-552,589-590). */
int agl_concat2_fwd(const float* a, const float* b, float* out, long N, int Ca, int Cb, int HW, int bcast, void* stream);
int agl_concat2_bwd(const float* d, float* da, float* db, long N, int Ca, int Cb, int HW, int bcast, void* stream);
int agl_embedding_bwd(const float* dout, const long long* rows, float* dtable, int N, int D, int V, void* stream);

/* y[o,c] = u[o,c] (x) zero-padded mask[o]: the rank-1 layout tensor through LayoutEncoder.c0 (k1,p1)
 * (generator_obj_att.py:489-494) without materialising the (O,128,R,R) product. */
int agl_mask_outer_fwd(const float* u, const float* mask, float* y, int O, int C, int R, int pad, void* stream);
int agl_mask_outer_bwd(const float* dy, const float* mask, float* du, int O, int C, int R, int pad, void* stream);

/* conv3x3(p1) + avg_pool2d(2) == conv4x4(s2,p1): weight transform (n_filters = Cout*Cin) and its adjoint
 * (models/discriminator.py:46-51, :81-86 residual branches that end in _downsample). */
int agl_pool_fuse_weight_fwd(const float* w3, float* w4, long n_filters, void* stream);
int agl_pool_fuse_weight_bwd(const float* dw4, float* dw3, long n_filters, void* stream);

/* ---- layout-encoder first stage in closed form (generator_obj_att.py:489-497: rank-1 tensor -> c0 -> CondBN -> ReLU
 * -> c2): two-level images; see csrc/layout.hip.  OH = R/2 + 1. */
int agl_layout1_levels(const float* u, const float* mask, const long long* labels, const float* table, float* area, float* mean,
                       float* rstd, float* A, float* B, float* D, float* running_mean, float* running_var,
                       long long* num_batches_tracked, int O, int C, int R, float eps, float momentum, int training, void* stream);
int agl_layout1_permute(const float* src, float* dst, int Co, int C, int to_rows, void* stream);
int agl_layout1_pixels(const float* WB, const float* WD, const float* mask, float* y, int O, int Co, int R, void* stream);
int agl_layout1_tapsum(const float* dy, const float* mask, float* GB, float* GD, int O, int Co, int R, void* stream);
int agl_layout1_levels_bwd(const float* dA, const float* dB, const float* A, const float* B, const float* u, const float* area,
                           const float* mean, const float* rstd, const float* table, const long long* labels, float* du,
                           float* dtable, int O, int C, int R, int V, int training, void* ws, long ws_bytes, void* stream);

/* ---- spectral norm (torch.nn.utils.spectral_norm via add_sn, discriminator.py:15-22), batched per net */
struct AglSnLayer {
  const float* w; float* u; float* v; float* w_sn; float* sigma; float* tmp; float* u_used; float* v_used;
  const float* g; float* dw; int rows, cols;
};
long agl_sn_layer_desc_bytes(void);
long agl_sn_tmp_floats(int rows, int cols);
int agl_sn_forward(const void* host_layers, int n_layers, int power_iter, float eps, void* stream);
int agl_sn_backward(const void* host_layers, int n_layers, int accumulate, void* stream);

/* ---- losses of train64.py:195-245, 284-354 -----------------------------------------------------------
 * Single-workgroup, fixed-order reductions.  Each call writes the UNWEIGHTED loss to *loss_out and, when the
 * gradient pointer is not NULL, coef * d(loss)/d(input) (coef = lambda * mix weight of the training loop). */
/* mean(BCEWithLogits(x, target)) with a constant target (F.binary_cross_entropy_with_logits vs full_like) */
int agl_bce_logits_const(const float* x, long n, float target, float coef, float* loss_out, float* dx, void* stream);
/* mean over annotated rows (rows whose target row-sum != 0; train64.py:241,323) of BCEWithLogits with
 * pos_weight[A]; dx is written for all rows (zero for unselected ones). */
int agl_bce_logits_posw(const float* x, const float* targets, const float* pos_weight, long rows, int A, float coef,
                        float* loss_out, float* dx, void* stream);
/* F.cross_entropy(logits[R][V], labels[R]); a label outside [0,V) poisons the loss and its gradient row with NaN
 * (torch raises there; this call only enqueues). */
int agl_cross_entropy(const float* logits, const long long* labels, long R, int V, float coef, float* loss_out,
                      float* dlogits, void* stream);
/* The two row-wise losses over several workgroups (16 rows each) with agl_loss_rows_ws_bytes(rows) of scratch for the row terms, which ONE
 * workgroup then adds in a fixed order: the same gradients bit for bit, the cross-entropy value bit for bit, the attribute loss's value
 * to double rounding; ~10 us instead of ~75 on a 64-image batch.  The scratch must not be reused before the call's kernels have run. */
long agl_loss_rows_ws_bytes(long rows);
int agl_cross_entropy_ws(const float* logits, const long long* labels, long R, int V, float coef, float* loss_out, float* dlogits, void* ws,
                         long ws_bytes, void* stream);
int agl_bce_logits_posw_ws(const float* x, const float* targets, const float* pos_weight, long rows, int A, float coef, float* loss_out,
                           float* dx, void* ws, long ws_bytes, void* stream);
/* sum_n keep[n] * mean_len |a-b| / denom (train64.py:284-287); keep == NULL means all ones */
long agl_l1_rows_ws_bytes(void);
int agl_l1_rows(const float* a, const float* b, const float* keep, long N, long len, float coef, float denom,
                float* loss_out, float* da, void* ws, long ws_bytes, void* stream);
/* -0.5 * sum(1 + logvar - mu^2 - exp(logvar)) (train64.py:294-295) */
int agl_kl_sum(const float* mu, const float* logvar, long n, float coef, float* loss_out, float* dmu, float* dlogvar,
               void* stream);

/* Hinge GAN losses of the vendored SPADE GANLoss (models/spade/networks/loss.py:65-76; NOT on the reference's train path,
 * which uses BCE-with-logits).  mode 0: D on real, -mean(min(x-1,0)); 1: D on fake, -mean(min(-x-1,0)); 2: G, -mean(x). */
int agl_hinge_loss(const float* x, long n, int mode, float coef, float* loss_out, float* dx, void* stream);

/* ---- host logic of the loop moved on device: attribute estimate, train64.py:156-166 ------- */
/* N3: object masks from boxes on device (data/vg_custom_mask.py:136,158: python round(), slice semantics) */
int agl_rasterize_boxes(const float* boxes, float* masks, int O, int R, void* stream);
int agl_attr_estimate(const float* logits, const float* attribute, float* attribute_est, int O, int A, void* stream);
/* N3: everything the VG batch builder derives from the boxes (data/vg_custom_mask.py:136-158): masks, the shifted boxes
 * (0.8 x the larger horizontal border distance when the box is narrower than half the image) and their masks. */
int agl_layout_from_boxes(const float* boxes, float* boxes_shift, float* masks, float* masks_shift, int O, int R, void* stream);
/* N2: the attribute logic of the inference / attribute-editing loop (test64.py:143-184): clear a set of columns and set one
 * (cols_dev: device int array); membership of a column in a row's top-k logits; sigmoid(logit) > threshold. */
int agl_attr_edit(float* attribute, const int* cols_dev, int ncols, int tgt, int O, int A, void* stream);
int agl_topk_contains(const float* logits, unsigned char* out, int O, int A, int k, int tgt, void* stream);
int agl_sigmoid_threshold(const float* logits, unsigned char* pred, long n, float thr, void* stream);
/* N2: data/utils.py:47-66 imagenet_deprocess_batch — de-normalise, per-image min/max rescale, bytes. inv_std/mean are
 * HOST pointers to 3 floats (fp32(1/std_c), fp32(mean_c)). */
int agl_deprocess_u8(const float* x, unsigned char* out, int N, int C, int HW, int rescale, const float* inv_std, const float* mean,
                     void* stream);

/* ---- optimiser (torch.optim.Adam, train64.py:111-114) over a flat fp32 arena ---------------------- */
int agl_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2, double eps,
                  int step, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif
