// Internal interface of the few-channel convolution kernels (few.hip), used by the dispatch in conv.hip.
#pragma once
#include "agl_internal.h"

struct FewBwwShape { int N, Cin, H, W, Cout, OH, OW, ks, stride, pad, up, in_relu; };
// Bytes of slab workspace the few-input-channel weight gradient needs (0: shape not taken).
long few_bww_ws_bytes(const FewBwwShape& a);
// AGL_OK: slabs written ([*splits][Cout][Cin*ks*ks], to be added by slab_reduce); -1: shape not taken; else an error code.
// dy_fold (optional, SPADE form of pconv.h's InFold): dy is the raw input of a SPADE whose modulate + ReLU (dy_relu) this kernel applies while
// it stages the tensor, rounding the result to bf16 when dy_round_bf16 (the value a bf16-stored modulated tensor would hold)
struct InFold;
int few_bww_try(const FewBwwShape& a, const float* dy, const float* x, void* ws, long ws_bytes, int* splits, hipStream_t st,
                const char* name, int dy_bf16 = 0, const InFold* dy_fold = nullptr, int dy_relu = 0, int dy_round_bf16 = 0);      // dy_bf16: dy points to bf16 elements
// Forward convolution with <= 4 input channels, 1x1 / 3x3, stride 1, "same" size (exact fp32 on the vector units): AGL_OK, -1 when
// the shape is not taken, else an error code.
int few_cin_fwd_try(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W, int Cout, int ks, int in_relu,
                    int relu, int accumulate, int y_bf16, hipStream_t st, const char* name);      // y_bf16: 1 = y points to bf16 elements, 2 = channel-blocked bf16
// nn.Linear gradients (1x1 maps): dw[co][ci] (+)= sum_n dy[n][co] x[n][ci];  dx[n][ci] (+)= sum_co dy[n][co] w[co][ci] (masked by pos_mask > 0)
// round_bf16: bf16 arithmetic mode (both operands rounded to bf16 before the fp32 multiply-add)
int linear_bww_launch(const float* dy, const float* x, float* dw, int N, int Cin, int Cout, int in_relu, int accumulate, int round_bf16,
                      hipStream_t st, const char* name);
int linear_bwd_data_launch(const float* dy, const float* w, const float* pos_mask, float* dx, int N, int Cin, int Cout, int accumulate,
                           int round_bf16, hipStream_t st, const char* name);
