// Internal interface of the few-channel convolution kernels (few.hip), used by the dispatch in conv.hip.
#pragma once
#include "agl_internal.h"

struct FewBwwShape { int N, Cin, H, W, Cout, OH, OW, ks, stride, pad, up, in_relu; };
// Bytes of slab workspace the few-input-channel weight gradient needs (0: shape not taken).
long few_bww_ws_bytes(const FewBwwShape& a);
// AGL_OK: slabs written ([*splits][Cout][Cin*ks*ks], to be added by slab_reduce); -1: shape not taken; else an error code.
int few_bww_try(const FewBwwShape& a, const float* dy, const float* x, void* ws, long ws_bytes, int* splits, hipStream_t st,
                const char* name);
