// Internal interface of the bf16-MFMA LDS-patch convolution (pconv.hip), used by the dispatch in conv.hip.
#pragma once
#include "agl_internal.h"

// Input transform applied while an operand is staged: the normalise-modulate of the BatchNorm2d / ConditionalBatchNorm2d that reads the
// tensor (generator_obj_att.py:31-44, :433, :583) folded into the convolution that consumes its output —
//   v = fma(x, scale[r * C + c], shift[r * C + c]),  r = image index when per_n, else 0   (then the fused input ReLU)
// with scale = rstd * gamma, shift = beta - mean * rstd * gamma formed in double (agl_norm_fold_table).  Zero padding is applied AFTER
// the transform (padding stays 0).  (`mean` is kept for the ABI's symmetry with the statistics; the kernels read scale / shift only.)
//
// SPADE form (cells != NULL; normalization.py:97,106 in front of generator_obj_att128.py:588-597's c6 / c7): the modulation parameters
// live on a G x G class grid per image, cells[((n * C/8 + c/8) * G*G + cell) * 16 + {0..7: 1 + gamma of channels c..c+7, 8..15: beta}]
// (agl_spade_cells), pixel (iy, ix) reads cell map[iy] * G + map[ix]; mean / scale are then the per-channel batch mean and rstd and
//   v = spade_value(x, mean[c], scale[c], cells[.][j], cells[.][8 + j])          (csrc/spade.h — the stand-alone apply's own expression)
struct InFold { const float* mean; const float* scale; const float* shift; int per_n; const float* cells; const int* map; int G; };

struct PConvArgs {
  const float* x; const float* w; const float* bias; const float* pos_mask; float* y;
  int N, Cin, H, W, Cout, OH, OW;   // H, W: stored input map (logical size H<<up); OH, OW: output map
  int ks, stride, pad, up, in_relu, relu, accumulate;
  int w_sm, w_sc, flip;             // element strides of w for (output channel m, input channel c); flipped taps
  int nsplit;                       // 1: bf16 operands (AGL_CONV_BF16); 3: fp32 operands as three bf16 terms, six products
  int any_grid;                     // launch also below the occupancy threshold (AGL_CONV_ANY_GRID)
  int w8;                           // eight-wave workgroups where instantiated (AGL_CONV_W8)
  int prio;                         // priority 1 for the conversion segments (AGL_CONV_PRIO)
  int ablate;                       // diagnostic builds of pconv_k (flags bits 9..11): WRONG results, timing only
  float* stats; long stats_floats; int* stat_rows;   // optional BatchNorm partials of the output: buffer, its capacity, rows written
  const void* packed;               // optional: the weights already in packed form (pconv_pack / pconvT_pack with the same nsplit) — no
                                    // per-call pack_weights_k launch; the packed tensor w0 may differ from w by a scalar: w = w0 / *out_div
  const float* out_div;             // optional device scalar: the accumulated products are divided by it before bias / mask / ReLU
  int x_bf16;                       // x points to bf16 elements (AGL_CONV_X_BF16; nsplit 1 only)
  int mask_bf16;                    // pos_mask points to bf16 elements (AGL_CONV_MASK_BF16; launches without a reduction split only)
  InFold fold;                      // optional (scale != NULL): transform of x while it is staged (stride-1 / stride-2 forward forms)
  int y_bf16;                       // y points to bf16 elements (AGL_CONV_Y_BF16; launches without a reduction split only)
  const float* addend;              // optional fp32 tensor shaped like y, added before the output ReLU (no reduction split)
  int x_blk, y_blk;                 // x / y are CHANNEL-BLOCKED bf16 tensors [N][C/8][H][W][8] (AGL_CONV_X_BLOCKED / _Y_BLOCKED, with x_bf16 / y_bf16;
                                    // the forms pconv_takes_blocked lists)
  int mask_blk;                     // pos_mask is a channel-blocked bf16 tensor shaped like the output (AGL_CONV_MASK_BLOCKED, with mask_bf16)
  const float* sc_x; const float* sc_w; const float* sc_b; int sc_cin;   // optional few-channel 1x1 shortcut added in the epilogue:
                                    // y += sc_b[m] + sum_c sc_w[m*sc_cin + c] * sc_x[n][c][pixel] (bf16 3x3 stride-1 forms, <= 64-channel tiles)
};
// Packed form of a weight tensor for pconv_try (forward: flip 0, w_sm = Cin*ks*ks, w_sc = ks*ks; "same" input gradient: flip 1, roles
// swapped) and for pconvT_try (phase4): bytes = pconv_ws_bytes / pconvT_ws_bytes.  M = rows (output channels of the pass).
int pconv_pack(const float* w, void* packed, int M, int Cred, int ks, int w_sm, int w_sc, int flip, int nsplit, int phase4, hipStream_t st,
               const char* name);
// The same pack as one row of a descriptor table (14 64-bit words; row[13] = its 256-thread blocks, row[12] = its first block, to be
// filled by the caller with the running sum) and the launch that executes a whole table (pack_many_k)
void pconv_pack_desc(const float* w, void* packed, int M, int Cred, int ks, int w_sm, int w_sc, int flip, int nsplit, int phase4, long long* row);
int pconv_pack_many(const void* rows_dev, int n, long total_blocks, hipStream_t st, const char* name);
// Upper bound of the partial rows pconv_try writes for an output of N images of OH x OW pixels
long pconv_stat_rows_max(int N, int OH, int OW);
long pconv_stat_row_floats(int Cout);      // floats per partial row: [Cout][{count, mean, M2}]

// Matrix-core products per fp32 multiply-add of the split arithmetic (nsplit 3) of this build: 3 (fp16 hi / lo planes) or 6 (bf16 x3)
int pconv_split_products();
// Bytes of workspace pconv needs for these extents (packed weights), 0 when the shape is not eligible.
long pconv_ws_bytes(int Cin, int Cout, int ks, int nsplit);
// ... plus the slabs of a reduction split when the output (out_numel floats) is small enough for one
long pconv_ws_bytes_split(int Cin, int Cout, int ks, int nsplit, long out_numel);
// True when pconv_try would launch for these extents (given enough workspace).
bool pconv_eligible(const PConvArgs& a);
// True when pconv_try takes these extents with a.x_blk / a.y_blk (channel-blocked bf16 x and / or y)
bool pconv_takes_blocked(const PConvArgs& a);
// Reduction splits pconv_try would use for these extents (1: none), -1 when the shape is not eligible.
int pconv_plan_splits(const PConvArgs& a);
// Returns AGL_OK when launched, -1 when the shape is not eligible (caller falls back), or an error code.
int pconv_try(const PConvArgs& a, void* ws, long ws_bytes, hipStream_t st, const char* name);

// 4x4 / stride-2 / pad-1 input gradient on the same kernel (four 2x2-tap phases): a.x = dy (a.Cin reduction channels, a.H x a.W),
// a.y = dx (a.Cout channels, a.OH = 2H, a.OW = 2W), a.w_sm / a.w_sc = element strides of w for (dx channel, dy channel).
bool pconvT_eligible(const PConvArgs& a);
bool pconvT_takes_bf16_mask(const PConvArgs& a);
long pconvT_ws_bytes(int Cred, int Crow, int nsplit);
long pconvT_ws_bytes_split(int Cred, int Crow, int nsplit, long out_numel);
int pconvT_try(const PConvArgs& a, void* ws, long ws_bytes, hipStream_t st, const char* name);

// Few channels on one side (CO <= 4), ks x ks window, stride 1, "same" size: vertical ks x 1 convolution on the matrix cores + a
// diagonal sum (pconv.hip).  x: (N, Cred, H, W); y: (N, CO, H, W); w element (o, c, kh, kw) = w[o*w_so + c*w_sc + kh*ks + kw]
// (flip: taps reversed — the input-gradient form).
struct PVertArgs {
  const float* x; const float* w; const float* bias; const float* pos_mask; float* y;
  int N, Cred, H, W, CO, ks, pad, w_so, w_sc, flip, relu, accumulate, nsplit;
  int x_bf16;                       // x points to bf16 elements (nsplit 1 only)
  InFold fold; int in_relu;         // optional transform of x while it is staged (SPADE form, nsplit 1, fp32 x), then the fused input ReLU
};
long pconv_vert_ws_bytes(int N, int Cred, int H, int W, int CO, int ks, int nsplit);
int pconv_vert_try(const PVertArgs& a, void* ws, long ws_bytes, hipStream_t st, const char* name);

struct PBwwArgs {
  const float* dy; const float* x; float* dw;
  int N, Cin, H, W, Cout, OH, OW;   // H, W: stored input map (logical size H<<up)
  int ks, stride, pad, up, in_relu, accumulate, nsplit;
  int x_bf16;                       // x points to bf16 elements (AGL_CONV_X_BF16; nsplit 1 only)
  int dy_bf16;                      // dy points to bf16 elements (AGL_CONV_DY_BF16; nsplit 1 only)
  int x_blk;                        // x is a channel-blocked bf16 tensor [N][Cin/8][H][W][8] (AGL_CONV_X_BLOCKED, with x_bf16; 3x3 stride 1 and 4x4 stride 2)
  InFold fold;                      // optional transform of x while it is staged (as in PConvArgs)
  float* dbias; int dbias_accumulate; int* dbias_done;    // optional: also (+= when dbias_accumulate) the bias gradient sum_pixels dy into dbias[Cout]; *dbias_done = 1 when this path did it
};
long pbww_ws_bytes(const PBwwArgs& a);
bool pbww_takes_spade(const PBwwArgs& a);      // the weight-gradient kernel has the SPADE form of the input transform compiled in for these extents
int pbww_try(const PBwwArgs& a, void* ws, long ws_bytes, hipStream_t st, const char* name);
