// SPADE's normalise-modulate of one element (reference models/spade/networks/normalization.py:97,106 with param_free_norm = BatchNorm2d),
//   v = (x - mean[c]) * rstd[c] * (1 + gamma[n][c][cell]) + beta[n][c][cell],
// written ONCE: the stand-alone apply (norm.hip norm_apply_fwd*), the staging passes of the convolutions that apply it themselves
// (pconv.hip, few.hip: InFold::cells), and the backward's ReLU mask of a never-stored output (norm.hip) all call this function with
// g1 = 1.f + gamma rounded to fp32 — one subtraction, one multiplication, one fused multiply-add, in this order — so the folded and the
// two-pass forms produce the same bits and the mask is the forward's own decision.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ float spade_value(float x, float mu, float rs, float g1, float b) { return __builtin_fmaf((x - mu) * rs, g1, b); }
