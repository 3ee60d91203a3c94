// Batch-statistics normalisation family (HBM-bound): BatchNorm2d/1d statistics, the affine /
// class-conditional / SPADE modulation (+ReLU, +residual) and their backward passes.
// Tensors are fp32 NCHW viewed as rows (n,c) of HW contiguous floats.
//   mode 0: y = xhat                         (no affine)
//   mode 1: y = xhat*gamma[c] + beta[c]      (nn.BatchNorm affine)
//   mode 2: y = xhat*T[lab[n]][c] + T[lab[n]][C+c]   (ConditionalBatchNorm2d, table T[V][2C])
//   mode 3: y = xhat*(1+gb[n][c][hw]) + gb[n][C+c][hw]   (SPADE, gb = [gamma;beta] conv output)
#include <cstdint>
#include "agl_internal.h"
#include "spade.h"
#include <algorithm>

// (AglBnUpdate / AGL_BN_UPDATE_MAX: include/agl.h — the sources do not include the public header, tests/test_abi.py keeps them in step)
#define AGL_BN_UPDATE_MAX 24
struct AglBnUpdate { const double* moments; float* running_mean; float* running_var; long long* num_batches_tracked; int C; float momentum; };

namespace {

template <int LPR>
__device__ __forceinline__ float group_sum(float v, float* scratch4) {
  if constexpr (LPR == 256) {
    return block_sum_256(v, scratch4);
  } else {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
  }
}

// ---------------------------------------------------------------- statistics
// Partial sums over a contiguous range of the channel's elements.  With HW % 4 == 0 the range is walked in 16-byte pieces
// (a channel row is HW contiguous floats; taken only when the tensor starts on a 16-byte boundary); every element is widened to
// double before it is added, in both forms.
__device__ __forceinline__ void channel_moments(const float* __restrict__ x, int C, int HW, int c, long e0, long e1, double& a, double& b) {
  a = 0.0; b = 0.0;
  if ((HW & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {      // (a sub-view that starts off a 16-byte boundary: scalar walk)
    // (32-bit index arithmetic — the tensors hold < 2^30 elements — and two independent 16-byte loads in flight per thread: the
    //  64-bit division per load of the first form held the kernel at 2.6 TB/s)
    const unsigned Q = HW >> 2, g1 = (unsigned)(e1 >> 2);
    auto piece = [&](unsigned g) {
      const unsigned n = g / Q, q = g - n * Q;
      return *reinterpret_cast<const float4*>(x + ((long)n * C + c) * HW + 4 * q);
    };
    auto add = [&](const float4& v) {
      a += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
      b += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    };
    unsigned g = (unsigned)(e0 >> 2) + threadIdx.x;
    for (; g + 768 < g1; g += 1024) {      // four loads in flight; added in the order of the two-load loop below (same sums bit for bit)
      const float4 v0 = piece(g), v1 = piece(g + 256), v2 = piece(g + 512), v3 = piece(g + 768);
      add(v0);
      add(v1);
      add(v2);
      add(v3);
    }
    for (; g + 256 < g1; g += 512) {
      const float4 v0 = piece(g), v1 = piece(g + 256);
      add(v0);
      add(v1);
    }
    if (g < g1) add(piece(g));
  } else {
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
      const int n = (int)(e / HW), hw = (int)(e - (long)n * HW);
      const float v = x[((long)n * C + c) * HW + hw];
      a += v;
      b += (double)v * v;
    }
  }
}

__global__ __launch_bounds__(256) void bn_stats_partial(const float* __restrict__ x, int N, int C, int HW, int S,
                                                        double* __restrict__ part) {
  const int c = blockIdx.x, s = blockIdx.y;
  const long total = (long)N * HW;
  long chunk = (total + S - 1) / S;
  chunk = (chunk + 3) & ~3L;                               // whole 16-byte pieces per split
  const long e0 = min(total, s * chunk), e1 = min(total, e0 + chunk);
  double a, b;
  channel_moments(x, C, HW, c, e0, e1, a, b);
  __shared__ double sc[4];
  a = block_sum_256(a, sc);
  b = block_sum_256(b, sc);
  if (threadIdx.x == 0) {
    part[((long)c * S + s) * 2 + 0] = a;
    part[((long)c * S + s) * 2 + 1] = b;
  }
}

// Small tensors (one workgroup per channel covers them): statistics and the running update in ONE launch
__global__ __launch_bounds__(256) void bn_stats_single(const float* __restrict__ x, int N, int C, int HW, float eps, float momentum,
                                                       float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ rmean,
                                                       float* __restrict__ rvar, long long* __restrict__ nbt, double* __restrict__ moments) {
  const int c = blockIdx.x;
  const long M = (long)N * HW;
  double a, b;
  channel_moments(x, C, HW, c, 0, M, a, b);
  __shared__ double sc[4];
  a = block_sum_256(a, sc);
  b = block_sum_256(b, sc);
  if (threadIdx.x != 0) return;
  if (c == 0 && nbt) *nbt += 1;
  const double mu = a / (double)M;
  double var = b / (double)M - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
  if (moments) { moments[2 * c] = mu; moments[2 * c + 1] = unb; }      // for agl_bn_running_update (bit-identical replays)
  if (rmean) {
    rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mu);
    rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
  }
}

__global__ void bn_stats_final(const double* __restrict__ part, int C, int S, long M, float eps, float momentum,
                               float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ rmean,
                               float* __restrict__ rvar, long long* __restrict__ nbt, double* __restrict__ moments) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < S; ++s) {
    a += part[((long)c * S + s) * 2];
    b += part[((long)c * S + s) * 2 + 1];
  }
  double mu = a / (double)M;
  double var = b / (double)M - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  const double unb = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
  if (moments) { moments[2 * c] = mu; moments[2 * c + 1] = unb; }      // for agl_bn_running_update (bit-identical replays)
  if (rmean) {
    rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mu);
    rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
  }
}

// Statistics from the partial rows a convolution wrote beside its output (agl_conv2d_fwd_stats): part[row][c][{count, mean, M2}] over
// disjoint pixel sets.  One wave per channel; the rows are merged in double with Chan's update (mean and M2 of a union from the
// means and M2s of its parts: no difference of large sums anywhere), lanes in a fixed order, then across the wave.
__device__ __forceinline__ void chan_merge(double& n, double& mu, double& m2, double n2, double mu2, double m22) {
  const double nt = n + n2;
  if (nt > 0.0) {
    const double dl = mu2 - mu;
    mu += dl * (n2 / nt);
    m2 += m22 + dl * dl * (n * n2 / nt);
  }
  n = nt;
}
__global__ __launch_bounds__(64) void bn_stats_from_rows(const float* __restrict__ part, int rows, int C, long M, float eps, float momentum,
                                                         float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ rmean,
                                                         float* __restrict__ rvar, long long* __restrict__ nbt, double* __restrict__ moments) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (c == 0 && lane == 0 && nbt) *nbt += 1;
  double n = 0.0, mu = 0.0, m2 = 0.0;
  for (int r = lane; r < rows; r += 64) {
    const float* v = part + ((long)r * C + c) * 3;
    chan_merge(n, mu, m2, (double)v[0], (double)v[1], (double)v[2]);
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {      // (both partners compute the merged triple; lane 0's order is fixed)
    const double n2 = __shfl_xor(n, o), mu2 = __shfl_xor(mu, o), m22 = __shfl_xor(m2, o);
    if ((lane & o) == 0) chan_merge(n, mu, m2, n2, mu2, m22);
    else { double a = n2, b = mu2, d = m22; chan_merge(a, b, d, n, mu, m2); n = a; mu = b; m2 = d; }
  }
  if (lane != 0) return;
  (void)M;      // (= n: the rows cover every output element exactly once)
  double var = n > 0.0 ? m2 / n : 0.0;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  const double unb = n > 1.0 ? var * (n / (n - 1.0)) : var;
  if (moments) { moments[2 * c] = mu; moments[2 * c + 1] = unb; }      // for agl_bn_running_update (bit-identical replays)
  if (rmean) {
    rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mu);
    rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
  }
}

// The running-statistics update of a statistics call, replayed from the moments it left (same double arithmetic: bit-identical)
__global__ void bn_running_update_k(const double* __restrict__ moments, int C, float momentum, float* __restrict__ rmean,
                                    float* __restrict__ rvar, long long* __restrict__ nbt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  const double mu = moments[2 * c], unb = moments[2 * c + 1];
  rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mu);
  rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
}

// ... for up to AGL_BN_UPDATE_MAX layers in one launch: thread c applies the items in array order to channel c of each
struct BnUpdateBatch { AglBnUpdate it[AGL_BN_UPDATE_MAX]; int n; };
__global__ void bn_running_update_multi_k(BnUpdateBatch b) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  for (int e = 0; e < b.n; ++e) {
    const AglBnUpdate& u = b.it[e];
    if (c == 0 && u.num_batches_tracked) *u.num_batches_tracked += 1;
    if (c >= u.C) continue;
    const double mu = u.moments[2 * c], unb = u.moments[2 * c + 1];
    u.running_mean[c] = (float)((1.0 - u.momentum) * u.running_mean[c] + u.momentum * mu);
    u.running_var[c] = (float)((1.0 - u.momentum) * u.running_var[c] + u.momentum * unb);
  }
}

// Eval-mode statistics: mean = running_mean, rstd = 1/sqrt(running_var + eps)
__global__ void bn_stats_eval(const float* rmean, const float* rvar, int C, float eps, float* mean, float* rstd) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = rmean[c];
  rstd[c] = 1.0f / sqrtf(rvar[c] + eps);
}

// ---------------------------------------------------------------- forward apply
struct NormArgs {
  const float* x; const float* mean; const float* rstd;
  const float* p0; const float* p1; const long long* labels;
  int mode, relu, N, C, HW;
  // mode 3 with a gathered gamma|beta: p0 is (N, 2C, src_w, src_w) on a coarser (block-class) grid and pixel (iy, ix) of the
  // W-wide map reads cell (map[iy], map[ix]) — the expansion agl_grid_gather_fwd would write out is folded into the reads
  const int* map; int W, src_w;
  // backward of a FOLDED forward (agl_norm_fold_table + agl_conv2d_fwd_fold: the normalised tensor y was never stored): the ReLU mask
  // is recomputed from x with the very expression the consumer's staging pass evaluated, fmaf(x, fscale[r], fshift[r]) > 0
  const float* fscale; const float* fshift; int f_per_n;
  int y_bf16;      // the forward output y holds bf16 elements (written by agl_norm_apply_fwd_y16 for consumers that are bf16-mode convolutions)
};

// ReLU mask of element `idx` of row (n, c): from the stored output y, or recomputed from x (folded forward; fs / fh = the row's table entries)
__device__ __forceinline__ float bf16_at(const float* y, long idx) {      // element idx of a bf16 tensor, widened (a shift)
  return __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(y)[idx] << 16);
}
// mode 3 without a stored output (SPADE applied by its consumer's staging pass, agl_conv2d_fwd_spade): the value that pass computed
__device__ __forceinline__ long gb_plane(const NormArgs& a);
__device__ __forceinline__ float spade_at(const NormArgs& a, const float* __restrict__ gam, float x, float mu, float rs, int gi) {
  return spade_value(x, mu, rs, 1.f + gam[gi], gam[(long)a.C * gb_plane(a) + gi]);
}
__device__ __forceinline__ bool relu_dead(const NormArgs& a, const float* __restrict__ y, long idx, float x, float mu, float fs, float fh,
                                          const float* __restrict__ gam = nullptr, float rs = 0.f, int gi = 0) {
  if (!y) return a.mode == 3 ? !(spade_at(a, gam, x, mu, rs, gi) > 0.f) : !(fmaf(x, fs, fh) > 0.f);
  return a.y_bf16 ? !(bf16_at(y, idx) > 0.f) : !(y[idx] > 0.f);
}
// four consecutive elements of y starting at element 4*i of the row at `base` (fp32 or bf16 storage)
__device__ __forceinline__ float4 y_quad(const NormArgs& a, const float* y, long base, int i) {
  if (!a.y_bf16) return reinterpret_cast<const float4*>(y + base)[i];
  const uint2 b = reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(y) + base)[i];
  return float4{__builtin_bit_cast(float, b.x << 16), __builtin_bit_cast(float, b.x & 0xffff0000u),
                __builtin_bit_cast(float, b.y << 16), __builtin_bit_cast(float, b.y & 0xffff0000u)};
}
__device__ __forceinline__ void fold_row(const NormArgs& a, int n, int c, float& fs, float& fh) {
  fs = 0.f; fh = 0.f;
  if (a.fscale) { const long r = (a.f_per_n ? (long)n * a.C : 0) + c; fs = a.fscale[r]; fh = a.fshift[r]; }
}

// offset of pixel i's gamma (beta: + C planes) inside its (n, c) plane of p0, and the plane size
__device__ __forceinline__ int gb_index(const NormArgs& a, int i) {
  if (!a.map) return i;
  const int iy = i / a.W, ix = i - iy * a.W;
  return a.map[iy] * a.src_w + a.map[ix];
}
__device__ __forceinline__ long gb_plane(const NormArgs& a) { return a.map ? (long)a.src_w * a.src_w : (long)a.HW; }
// the same for pixels 4i .. 4i+3 of a row (W % 4 == 0 when gathered: they share an image row)
__device__ __forceinline__ float4 spade_quad(const NormArgs& a, const float* __restrict__ gam, float4 xv, float mu, float rs, int i) {
  int gi[4] = {4 * i, 4 * i + 1, 4 * i + 2, 4 * i + 3};
  if (a.map) {
    const int iy = (4 * i) / a.W, ix0 = 4 * i - iy * a.W, ro = a.map[iy] * a.src_w;
#pragma unroll
    for (int k = 0; k < 4; ++k) gi[k] = ro + a.map[ix0 + k];
  }
  return float4{spade_at(a, gam, xv.x, mu, rs, gi[0]), spade_at(a, gam, xv.y, mu, rs, gi[1]), spade_at(a, gam, xv.z, mu, rs, gi[2]),
                spade_at(a, gam, xv.w, mu, rs, gi[3])};
}

__device__ __forceinline__ void row_affine(const NormArgs& a, int n, int c, float& g, float& b) {
  g = 1.f; b = 0.f;
  if (a.mode == 1) { g = a.p0[c]; b = a.p1[c]; }
  else if (a.mode == 2) { const float* t = a.p0 + (long)a.labels[n] * 2 * a.C; g = t[c]; b = t[a.C + c]; }
}

template <int LPR>
__global__ __launch_bounds__(256) void norm_apply_fwd(NormArgs a, const float* __restrict__ residual, float* __restrict__ y) {
  const int row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  if (row >= a.N * a.C) return;
  const int n = row / a.C, c = row - n * a.C;
  const float mu = a.mean[c], rs = a.rstd[c];
  float g, b;
  row_affine(a, n, c, g, b);
  const long base = (long)row * a.HW;
  const long gp = gb_plane(a);
  const float* gam = a.mode == 3 ? a.p0 + ((long)n * 2 * a.C + c) * gp : nullptr;
  const float* bet = a.mode == 3 ? a.p0 + ((long)n * 2 * a.C + a.C + c) * gp : nullptr;
  for (int i = threadIdx.x % LPR; i < a.HW; i += LPR) {
    float xh = (a.x[base + i] - mu) * rs;
    const int gi = a.mode == 3 ? gb_index(a, i) : 0;
    float v = a.mode == 3 ? spade_value(a.x[base + i], mu, rs, 1.f + gam[gi], bet[gi]) : xh * g + b;
    if (residual) v += residual[base + i];
    if (a.relu) v = fmaxf(v, 0.f);
    if (a.y_bf16) reinterpret_cast<__bf16*>(y)[base + i] = (__bf16)v;
    else y[base + i] = v;
  }
}

// The same with 16-byte accesses: four consecutive pixels per thread and step (HW % 4 == 0, 16-byte aligned tensors, and W % 4 == 0
// for a gathered gamma|beta so that the four pixels share an image row).  Used on rows of >= 64 pixels (the 4-byte form ran at
// 3.3 TB/s on the large maps).
template <int LPR>
__global__ __launch_bounds__(256) void norm_apply_fwd4(NormArgs a, const float* __restrict__ residual, float* __restrict__ y) {
  const int row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  if (row >= a.N * a.C) return;
  const int n = row / a.C, c = row - n * a.C;
  const float mu = a.mean[c], rs = a.rstd[c];
  float g, b;
  row_affine(a, n, c, g, b);
  const long base = (long)row * a.HW;
  const long gp = gb_plane(a);
  const float* gam = a.mode == 3 ? a.p0 + ((long)n * 2 * a.C + c) * gp : nullptr;
  const float* bet = a.mode == 3 ? a.p0 + ((long)n * 2 * a.C + a.C + c) * gp : nullptr;
  const float4* x4 = reinterpret_cast<const float4*>(a.x + base);
  const float4* r4 = residual ? reinterpret_cast<const float4*>(residual + base) : nullptr;
  float4* y4 = reinterpret_cast<float4*>(y + base);
  uint2* y2 = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(y) + base);      // (bf16 storage: four elements per 8-byte store)
  for (int i = threadIdx.x % LPR; i < a.HW / 4; i += LPR) {
    const float4 xv = x4[i];
    float v[4] = {(xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs};
    if (a.mode == 3) {
      float gg[4], bb[4];
      if (a.map) {
        const int iy = (4 * i) / a.W, ix0 = 4 * i - iy * a.W, ro = a.map[iy] * a.src_w;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int gi = ro + a.map[ix0 + k]; gg[k] = gam[gi]; bb[k] = bet[gi]; }
      } else {
        const float4 g4 = reinterpret_cast<const float4*>(gam)[i], b4 = reinterpret_cast<const float4*>(bet)[i];
        gg[0] = g4.x; gg[1] = g4.y; gg[2] = g4.z; gg[3] = g4.w; bb[0] = b4.x; bb[1] = b4.y; bb[2] = b4.z; bb[3] = b4.w;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(v[k], 1.f + gg[k], bb[k]);      // (= spade_value of csrc/spade.h: v holds (x - mean) * rstd)
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] * g + b;
    }
    if (r4) { const float4 rv = r4[i]; v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w; }
    if (a.relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    if (a.y_bf16) {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      const bf16x4 ob = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      y2[i] = __builtin_bit_cast(uint2, ob);
    } else {
      y4[i] = float4{v[0], v[1], v[2], v[3]};
    }
  }
}

// ---------------------------------------------------------------- backward
// K1: per row (n,c): a1 = sum g*ge, a2 = sum g*ge*xhat  (ge = 1, or (1+gamma) in mode 3); mode 3 also writes dgb.
template <int LPR>
__global__ __launch_bounds__(256) void norm_bwd_rows(NormArgs a, const float* __restrict__ dy, const float* __restrict__ y,
                                                     float* __restrict__ rowsum, float* __restrict__ dgb) {
  __shared__ float sc[4];
  const int row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  const bool live = row < a.N * a.C;
  const int rr = live ? row : 0;
  const int n = rr / a.C, c = rr - n * a.C;
  const float mu = a.mean[c], rs = a.rstd[c];
  const long base = (long)rr * a.HW;
  const float* gam = a.mode == 3 ? a.p0 + ((long)n * 2 * a.C + c) * gb_plane(a) : nullptr;
  float* dgam = a.mode == 3 ? dgb + ((long)n * 2 * a.C + c) * a.HW : nullptr;
  float* dbet = a.mode == 3 ? dgb + ((long)n * 2 * a.C + a.C + c) * a.HW : nullptr;
  float s1 = 0.f, s2 = 0.f;
  float fs, fh;
  fold_row(a, n, c, fs, fh);
  if (live) {
    for (int i = threadIdx.x % LPR; i < a.HW; i += LPR) {
      float g = dy[base + i];
      const float xv = a.x[base + i];
      if (a.relu && relu_dead(a, y, base + i, xv, mu, fs, fh, gam, rs, a.mode == 3 ? gb_index(a, i) : 0)) g = 0.f;
      float xh = (xv - mu) * rs;
      if (a.mode == 3) {
        dgam[i] = g * xh;         // (full resolution either way: a gathered gamma|beta is reduced to its grid by agl_grid_gather_bwd)
        dbet[i] = g;
        g *= 1.f + gam[gb_index(a, i)];
      }
      s1 += g;
      s2 += g * xh;
    }
  }
  s1 = group_sum<LPR>(s1, sc);
  s2 = group_sum<LPR>(s2, sc);
  if (live && threadIdx.x % LPR == 0) {
    rowsum[2 * (long)row] = s1;
    rowsum[2 * (long)row + 1] = s2;
  }
}

// K1 with 16-byte accesses: one row per workgroup, four consecutive pixels per thread and step (HW % 4 == 0, aligned tensors,
// W % 4 == 0 for a gathered gamma|beta).
__global__ __launch_bounds__(256) void norm_bwd_rows4(NormArgs a, const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ rowsum, float* __restrict__ dgb) {
  __shared__ float sc[4];
  const int row = blockIdx.x;
  const int n = row / a.C, c = row - n * a.C;
  const float mu = a.mean[c], rs = a.rstd[c];
  const long base = (long)row * a.HW;
  const float* gam = a.mode == 3 ? a.p0 + ((long)n * 2 * a.C + c) * gb_plane(a) : nullptr;
  float4* dgam = a.mode == 3 ? reinterpret_cast<float4*>(dgb + ((long)n * 2 * a.C + c) * a.HW) : nullptr;
  float4* dbet = a.mode == 3 ? reinterpret_cast<float4*>(dgb + ((long)n * 2 * a.C + a.C + c) * a.HW) : nullptr;
  const float4* dy4 = reinterpret_cast<const float4*>(dy + base);
  const float4* x4 = reinterpret_cast<const float4*>(a.x + base);
  const bool have_y = a.relu && y != nullptr;
  float s1 = 0.f, s2 = 0.f;
  float fs, fh;
  fold_row(a, n, c, fs, fh);
  for (int i = threadIdx.x; i < a.HW / 4; i += 256) {
    const float4 gv = dy4[i], xv = x4[i];
    float g[4] = {gv.x, gv.y, gv.z, gv.w};
    const float xh[4] = {(xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs};
    if (a.relu) {
      const float4 yv = have_y ? y_quad(a, y, base, i) : (a.mode == 3 ? spade_quad(a, gam, xv, mu, rs, i) : float4{fmaf(xv.x, fs, fh), fmaf(xv.y, fs, fh), fmaf(xv.z, fs, fh), fmaf(xv.w, fs, fh)});
      if (!(yv.x > 0.f)) g[0] = 0.f;
      if (!(yv.y > 0.f)) g[1] = 0.f;
      if (!(yv.z > 0.f)) g[2] = 0.f;
      if (!(yv.w > 0.f)) g[3] = 0.f;
    }
    if (a.mode == 3) {
      dgam[i] = float4{g[0] * xh[0], g[1] * xh[1], g[2] * xh[2], g[3] * xh[3]};
      dbet[i] = float4{g[0], g[1], g[2], g[3]};
      if (a.map) {
        const int iy = (4 * i) / a.W, ix0 = 4 * i - iy * a.W, ro = a.map[iy] * a.src_w;
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] *= 1.f + gam[ro + a.map[ix0 + k]];
      } else {
        const float4 g4 = reinterpret_cast<const float4*>(gam)[i];
        g[0] *= 1.f + g4.x; g[1] *= 1.f + g4.y; g[2] *= 1.f + g4.z; g[3] *= 1.f + g4.w;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s1 += g[k]; s2 += g[k] * xh[k]; }
  }
  s1 = group_sum<256>(s1, sc);
  s2 = group_sum<256>(s2, sc);
  if (threadIdx.x == 0) {
    rowsum[2 * (long)row] = s1;
    rowsum[2 * (long)row + 1] = s2;
  }
}

// K1g (mode 3 with a gathered gamma|beta): the same row sums, and d(gamma|beta) reduced to the class grid in the same pass — a
// (n, c) plane per workgroup of W threads (one per pixel column).  Image row by image row: every thread leaves its pixel's
// (g*xhat, g) in LDS; the threads that own a class column add up their column range and keep running sums for the current class
// row, flushed to the cell when the class row ends.  Every cell has one owner and a fixed order (deterministic), the activation
// reads are coalesced, and neither the full-resolution d(gamma|beta) nor agl_grid_gather_bwd's pass over it exists.
// lo[]: range starts of the map's inverse (src_w + 1 entries, lo[src_w] = W).
__global__ __launch_bounds__(128) void norm_bwd_rows_gathered(NormArgs a, const float* __restrict__ dy, const float* __restrict__ y,
                                                              const int* __restrict__ lo, float* __restrict__ rowsum,
                                                              float* __restrict__ dgb) {
  __shared__ float ra[128], rb[128], red[4];
  const int row = blockIdx.x, n = row / a.C, c = row - n * a.C, W = a.W, sw = a.src_w, tid = threadIdx.x;
  const float mu = a.mean[c], rs = a.rstd[c];
  const long base = (long)row * a.HW, gp = (long)sw * sw;
  const float* gam = a.p0 + ((long)n * 2 * a.C + c) * gp;
  float* dgam = dgb + ((long)n * 2 * a.C + c) * gp;
  float* dbet = dgb + ((long)n * 2 * a.C + a.C + c) * gp;
  const bool owner = tid < sw;
  const int x0 = owner ? lo[tid] : 0, x1 = owner ? lo[tid + 1] : 0;
  float ca = 0.f, cb = 0.f, s1 = 0.f, s2 = 0.f;
  for (int iy = 0; iy < W; ++iy) {
    const long i = base + (long)iy * W + tid;
    float g = dy[i];
    const float xv = a.x[i];
    if (a.relu && !((y ? (a.y_bf16 ? bf16_at(y, i) : y[i]) : spade_at(a, gam, xv, mu, rs, a.map[iy] * sw + a.map[tid])) > 0.f)) g = 0.f;
    const float xh = (xv - mu) * rs;
    ra[tid] = g * xh; rb[tid] = g;
    __syncthreads();
    const int cy = a.map[iy];
    const bool last = iy == W - 1 || a.map[iy + 1] != cy;      // (uniform)
    if (owner) {
      for (int xx = x0; xx < x1; ++xx) { ca += ra[xx]; cb += rb[xx]; }
      if (last) {
        const long cell = (long)cy * sw + tid;
        dgam[cell] = ca; dbet[cell] = cb;
        const float ge = 1.f + gam[cell];
        s1 += ge * cb; s2 += ge * ca;
        ca = 0.f; cb = 0.f;
      }
    }
    __syncthreads();
  }
  // block sums of s1, s2 (W = 64 or 128 threads)
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s1; red[(tid >> 6) * 2 + 1] = s2; }
  __syncthreads();
  if (tid == 0) {
    rowsum[2 * (long)row] = W > 64 ? red[0] + red[2] : red[0];
    rowsum[2 * (long)row + 1] = W > 64 ? red[1] + red[3] : red[1];
  }
}

// K2b (mode 2): gradient of the class table: thread (v, c) scans the objects in increasing n and adds the rows whose
// label is v — every cell has one owner and a fixed order (deterministic, no atomics).
__global__ __launch_bounds__(256) void norm_bwd_table(NormArgs a, const float* __restrict__ rowsum, float* __restrict__ dtable) {
  __shared__ __attribute__((aligned(16))) int lab[1024];
  const int v = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  float g = 0.f, b = 0.f;
  for (int n0 = 0; n0 < a.N; n0 += 1024) {
    const int cnt = min(1024, a.N - n0);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += 256) lab[i] = (int)a.labels[n0 + i];
    __syncthreads();
    if (c < a.C) {
      // (four labels per LDS read — the loop is a chain of broadcast reads otherwise: 23 us per launch at 393 objects x 179 classes;
      //  matches are rare, ~2 objects per class, and are added in increasing n as before)
      int i = 0;
      for (; i + 4 <= cnt; i += 4) {
        const int4 l4 = *reinterpret_cast<const int4*>(lab + i);
        if (l4.x == v | l4.y == v | l4.z == v | l4.w == v) {
          const int ls[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (ls[k] == v) {
              g += rowsum[2 * ((long)(n0 + i + k) * a.C + c) + 1];
              b += rowsum[2 * ((long)(n0 + i + k) * a.C + c)];
            }
        }
      }
      for (; i < cnt; ++i)
        if (lab[i] == v) {
          g += rowsum[2 * ((long)(n0 + i) * a.C + c) + 1];
          b += rowsum[2 * ((long)(n0 + i) * a.C + c)];
        }
    }
  }
  if (c < a.C) {
    dtable[(long)v * 2 * a.C + c] += g;
    dtable[(long)v * 2 * a.C + a.C + c] += b;
  }
}

// K2: per channel: S1 = sum_n ge(n,c)*a1, S2 = sum_n ge(n,c)*a2 ; affine parameter grads (mode 1).  One wave per channel.
__global__ __launch_bounds__(256) void norm_bwd_channels(NormArgs a, const float* __restrict__ rowsum, float* __restrict__ chansum,
                                                         float* __restrict__ dp0, float* __restrict__ dp1, int param_accumulate) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= a.C) return;
  double S1 = 0.0, S2 = 0.0, A1 = 0.0, A2 = 0.0;
  for (int n = lane; n < a.N; n += 64) {
    const float a1 = rowsum[2 * ((long)n * a.C + c)], a2 = rowsum[2 * ((long)n * a.C + c) + 1];
    float g = 1.f;
    if (a.mode == 2) g = a.p0[(long)a.labels[n] * 2 * a.C + c];
    S1 += (double)g * a1; S2 += (double)g * a2; A1 += a1; A2 += a2;
  }
  S1 = wave_sum(S1); S2 = wave_sum(S2); A1 = wave_sum(A1); A2 = wave_sum(A2);
  if (lane == 0) {
    if (a.mode == 1) {
      if (dp0) {
        if (param_accumulate) { dp0[c] += (float)A2; dp1[c] += (float)A1; }
        else { dp0[c] = (float)A2; dp1[c] = (float)A1; }
      }
      S1 *= a.p0[c]; S2 *= a.p0[c];
    }
    chansum[2 * c] = (float)S1;
    chansum[2 * c + 1] = (float)S2;
  }
}

// K3: dx = rstd * (ge*g - S1/M - xhat*S2/M).  chansum == nullptr (batches of <= 64 images: every layer of the decoders and the
// residual blocks): K2 is folded in — every group of lanes that owns a row (n, c) adds the N row sums of its channel itself (one
// load per lane, the same order in every group of that channel: identical totals), the groups of image 0 also leave the affine
// parameter gradients, and the per-channel launch between the row pass and this one is gone.  (With hundreds of rows per channel —
// the crop encoder's per-object batches — that prologue costs more than the rows' own work: 2.2 -> 9.9 ms per iteration; those keep K2.)
template <int LPR>
__global__ __launch_bounds__(256) void norm_bwd_apply(NormArgs a, const float* __restrict__ dy, const float* __restrict__ y,
                                                      const float* __restrict__ rowsum, const float* __restrict__ chansum, float inv_m,
                                                      int batch_stats, float* __restrict__ dx, float* __restrict__ dp0,
                                                      float* __restrict__ dp1, int param_accumulate) {
  constexpr int G = LPR >= 64 ? 64 : LPR;          // lanes that add the channel's row sums together (a wave, or the row's lane group)
  const int row = blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
  const bool live = row < a.N * a.C;
  const int rr = live ? row : 0;
  const int n = rr / a.C, c = rr - n * a.C;
  const bool want_p = a.mode == 1 && dp0 != nullptr;
  float m1 = 0.f, m2 = 0.f;
  if (chansum) {
    if (batch_stats) { m1 = chansum[2 * c] * inv_m; m2 = chansum[2 * c + 1] * inv_m; }
  } else if (batch_stats || want_p) {               // (wave-uniform)
    double S1 = 0.0, S2 = 0.0, A1 = 0.0, A2 = 0.0;
    for (int k = threadIdx.x % G; k < a.N; k += G) {
      const float a1 = rowsum[2 * ((long)k * a.C + c)], a2 = rowsum[2 * ((long)k * a.C + c) + 1];
      float g = 1.f;
      if (a.mode == 2) g = a.p0[(long)a.labels[k] * 2 * a.C + c];
      S1 += (double)g * a1; S2 += (double)g * a2; A1 += a1; A2 += a2;
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
      S1 += __shfl_xor(S1, o); S2 += __shfl_xor(S2, o); A1 += __shfl_xor(A1, o); A2 += __shfl_xor(A2, o);
    }
    if (want_p && live && n == 0 && threadIdx.x % LPR == 0) {
      if (param_accumulate) { dp0[c] += (float)A2; dp1[c] += (float)A1; }
      else { dp0[c] = (float)A2; dp1[c] = (float)A1; }
    }
    if (a.mode == 1) { S1 *= a.p0[c]; S2 *= a.p0[c]; }
    if (batch_stats) { m1 = (float)S1 * inv_m; m2 = (float)S2 * inv_m; }
  }
  if (!live) return;
  const float mu = a.mean[c], rs = a.rstd[c];
  float ge, b;
  row_affine(a, n, c, ge, b);
  const long base = (long)row * a.HW;
  const float* gam = a.mode == 3 ? a.p0 + ((long)n * 2 * a.C + c) * gb_plane(a) : nullptr;
  float fs, fh;
  fold_row(a, n, c, fs, fh);
  for (int i = threadIdx.x % LPR; i < a.HW; i += LPR) {
    float g = dy[base + i];
    const float xv = a.x[base + i];
    if (a.relu && relu_dead(a, y, base + i, xv, mu, fs, fh, gam, rs, a.mode == 3 ? gb_index(a, i) : 0)) g = 0.f;
    float xh = (xv - mu) * rs;
    float gg = a.mode == 3 ? g * (1.f + gam[gb_index(a, i)]) : g * ge;
    dx[base + i] = rs * (gg - m1 - xh * m2);
  }
}

// K3 with 16-byte accesses (one row per workgroup; per-channel sums from K2 or folded in, as in norm_bwd_apply<256>)
__global__ __launch_bounds__(256) void norm_bwd_apply4(NormArgs a, const float* __restrict__ dy, const float* __restrict__ y,
                                                       const float* __restrict__ rowsum, const float* __restrict__ chansum, float inv_m,
                                                       int batch_stats, float* __restrict__ dx, float* __restrict__ dp0,
                                                       float* __restrict__ dp1, int param_accumulate) {
  const int row = blockIdx.x;
  const int n = row / a.C, c = row - n * a.C;
  const bool want_p = a.mode == 1 && dp0 != nullptr;
  float m1 = 0.f, m2 = 0.f;
  if (chansum) {
    if (batch_stats) { m1 = chansum[2 * c] * inv_m; m2 = chansum[2 * c + 1] * inv_m; }
  } else if (batch_stats || want_p) {
    double S1 = 0.0, S2 = 0.0, A1 = 0.0, A2 = 0.0;
    for (int k = threadIdx.x % 64; k < a.N; k += 64) {
      const float a1 = rowsum[2 * ((long)k * a.C + c)], a2 = rowsum[2 * ((long)k * a.C + c) + 1];
      float g = 1.f;
      if (a.mode == 2) g = a.p0[(long)a.labels[k] * 2 * a.C + c];
      S1 += (double)g * a1; S2 += (double)g * a2; A1 += a1; A2 += a2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      S1 += __shfl_xor(S1, o); S2 += __shfl_xor(S2, o); A1 += __shfl_xor(A1, o); A2 += __shfl_xor(A2, o);
    }
    if (want_p && n == 0 && threadIdx.x == 0) {
      if (param_accumulate) { dp0[c] += (float)A2; dp1[c] += (float)A1; }
      else { dp0[c] = (float)A2; dp1[c] = (float)A1; }
    }
    if (a.mode == 1) { S1 *= a.p0[c]; S2 *= a.p0[c]; }
    if (batch_stats) { m1 = (float)S1 * inv_m; m2 = (float)S2 * inv_m; }
  }
  const float mu = a.mean[c], rs = a.rstd[c];
  float ge, b;
  row_affine(a, n, c, ge, b);
  const long base = (long)row * a.HW;
  const float* gam = a.mode == 3 ? a.p0 + ((long)n * 2 * a.C + c) * gb_plane(a) : nullptr;
  const float4* dy4 = reinterpret_cast<const float4*>(dy + base);
  const float4* x4 = reinterpret_cast<const float4*>(a.x + base);
  const bool have_y = a.relu && y != nullptr;
  float4* dx4 = reinterpret_cast<float4*>(dx + base);
  float fs, fh;
  fold_row(a, n, c, fs, fh);
  for (int i = threadIdx.x; i < a.HW / 4; i += 256) {
    const float4 gv = dy4[i], xv = x4[i];
    float g[4] = {gv.x, gv.y, gv.z, gv.w};
    const float xh[4] = {(xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs};
    if (a.relu) {
      const float4 yv = have_y ? y_quad(a, y, base, i) : (a.mode == 3 ? spade_quad(a, gam, xv, mu, rs, i) : float4{fmaf(xv.x, fs, fh), fmaf(xv.y, fs, fh), fmaf(xv.z, fs, fh), fmaf(xv.w, fs, fh)});
      if (!(yv.x > 0.f)) g[0] = 0.f;
      if (!(yv.y > 0.f)) g[1] = 0.f;
      if (!(yv.z > 0.f)) g[2] = 0.f;
      if (!(yv.w > 0.f)) g[3] = 0.f;
    }
    float ge4[4] = {ge, ge, ge, ge};
    if (a.mode == 3) {
      if (a.map) {
        const int iy = (4 * i) / a.W, ix0 = 4 * i - iy * a.W, ro = a.map[iy] * a.src_w;
#pragma unroll
        for (int k = 0; k < 4; ++k) ge4[k] = 1.f + gam[ro + a.map[ix0 + k]];
      } else {
        const float4 g4 = reinterpret_cast<const float4*>(gam)[i];
        ge4[0] = 1.f + g4.x; ge4[1] = 1.f + g4.y; ge4[2] = 1.f + g4.z; ge4[3] = 1.f + g4.w;
      }
    }
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = rs * (g[k] * ge4[k] - m1 - xh[k] * m2);
    dx4[i] = float4{o[0], o[1], o[2], o[3]};
  }
}

// (N, 2C, G2) gamma|beta -> the blocked cell table of agl_spade_cells.  One thread per (image, channel octet, cell): sixteen reads, each
// coalesced along the cells of a channel plane, and one 64-byte line written (consecutive threads, consecutive lines)
__global__ void spade_cells_k(const float* __restrict__ gb, float* __restrict__ cells, int N, int C, int G2, long total) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (n * C/8 + c8) * G2 + cell
  if (q >= total) return;
  const int cell = (int)(q % G2);
  const long nc8 = q / G2;
  const int c8 = (int)(nc8 % (C >> 3)), n = (int)(nc8 / (C >> 3));
  const float* const ga = gb + ((long)n * 2 * C + 8 * c8) * G2 + cell;
  const float* const be = ga + (long)C * G2;
  float v[16];
#pragma unroll
  for (int j = 0; j < 8; ++j) { v[j] = 1.f + ga[(long)j * G2]; v[8 + j] = be[(long)j * G2]; }
  float4* const o = reinterpret_cast<float4*>(cells + q * 16);
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = float4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
}

int pick_lpr(int HW) { return HW <= 4 ? 4 : (HW <= 16 ? 16 : (HW <= 512 ? 64 : 256)); }

// Tables of the folded normalise-modulate (agl_norm_fold_table): scale[r][c] = rstd[c] * gamma(r, c), shift[r][c] = beta(r, c) - mean[c] * scale[r][c]
__global__ void norm_fold_table_k(NormArgs a, int rows, float* __restrict__ scale, float* __restrict__ shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * a.C) return;
  const int n = i / a.C, c = i - n * a.C;
  float g, b;
  row_affine(a, n, c, g, b);
  const double sc = (double)a.rstd[c] * (double)g;
  scale[i] = (float)sc;
  shift[i] = (float)((double)b - (double)a.mean[c] * sc);      // v = fma(x, scale, shift): ONE operation per staged element
}

}  // namespace

extern "C" {

long agl_bn_stats_ws_bytes(int N, int C, int HW) {
  (void)N; (void)HW;
  return (long)C * 64 * 2 * sizeof(double);
}

// Batch statistics of x[N,C,HW] (training mode): mean[C], rstd[C]; running stats updated like nn.BatchNorm
// (momentum, unbiased variance) when running_mean != NULL; *num_batches_tracked += 1 when not NULL.
int agl_bn_stats(const float* x, int N, int C, int HW, float eps, float momentum, float* mean, float* rstd,
                 float* running_mean, float* running_var, long long* num_batches_tracked, double* moments, void* ws, long ws_bytes,
                 void* stream) {
  AGL_REQUIRE(x && mean && rstd && N > 0 && C > 0 && HW > 0, "agl_bn_stats: bad argument");
  const long total = (long)N * HW;
  hipStream_t st = (hipStream_t)stream;
  // splits: enough workgroups to fill the chip (C x S >= ~1024), at least 2048 elements each
  int S = (int)std::min<long>(std::min<long>(64, (1024 + C - 1) / C), std::max<long>(1, total / 2048));
  if (S <= 1) {
    hipLaunchKernelGGL(bn_stats_single, dim3(C), dim3(256), 0, st, x, N, C, HW, eps, momentum, mean, rstd, running_mean, running_var,
                       num_batches_tracked, moments);
    AGL_CHECK_LAUNCH("agl_bn_stats(single)");
    return AGL_OK;
  }
  if (!ws || ws_bytes < (long)C * S * 2 * (long)sizeof(double)) {
    agl_set_error("agl_bn_stats: workspace too small");
    return AGL_ERR_WORKSPACE;
  }
  hipLaunchKernelGGL(bn_stats_partial, dim3(C, S), dim3(256), 0, st, x, N, C, HW, S, (double*)ws);
  AGL_CHECK_LAUNCH("agl_bn_stats(partial)");
  hipLaunchKernelGGL(bn_stats_final, dim3(agl_cdiv(C, 128)), dim3(128), 0, st, (const double*)ws, C, S, total, eps,
                     momentum, mean, rstd, running_mean, running_var, num_batches_tracked, moments);
  AGL_CHECK_LAUNCH("agl_bn_stats(final)");
  return AGL_OK;
}

int agl_bn_running_update(const double* moments, int C, float momentum, float* running_mean, float* running_var,
                          long long* num_batches_tracked, void* stream) {
  AGL_REQUIRE(moments && running_mean && running_var && C > 0, "agl_bn_running_update: bad argument");
  hipLaunchKernelGGL(bn_running_update_k, dim3(agl_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, moments, C, momentum, running_mean,
                     running_var, num_batches_tracked);
  AGL_CHECK_LAUNCH("agl_bn_running_update");
  return AGL_OK;
}

int agl_bn_running_update_multi(const AglBnUpdate* items, int n, void* stream) {
  AGL_REQUIRE(items && n > 0 && n <= AGL_BN_UPDATE_MAX, "agl_bn_running_update_multi: 1..%d items", AGL_BN_UPDATE_MAX);
  BnUpdateBatch b;
  b.n = n;
  int cmax = 0;
  for (int e = 0; e < n; ++e) {
    AGL_REQUIRE(items[e].moments && items[e].running_mean && items[e].running_var && items[e].C > 0, "agl_bn_running_update_multi: bad item %d", e);
    b.it[e] = items[e];
    cmax = std::max(cmax, items[e].C);
  }
  hipLaunchKernelGGL(bn_running_update_multi_k, dim3(agl_cdiv(cmax, 128)), dim3(128), 0, (hipStream_t)stream, b);
  AGL_CHECK_LAUNCH("agl_bn_running_update_multi");
  return AGL_OK;
}

int agl_bn_stats_from_partials(const float* partials, int rows, int C, long count, float eps, float momentum, float* running_mean,
                               float* running_var, long long* num_batches_tracked, float* mean, float* rstd, double* moments,
                               void* stream) {
  AGL_REQUIRE(partials && mean && rstd && rows > 0 && C > 0 && count > 0, "agl_bn_stats_from_partials: bad argument");
  AGL_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "agl_bn_stats_from_partials: running_mean / running_var go together");
  hipLaunchKernelGGL(bn_stats_from_rows, dim3(C), dim3(64), 0, (hipStream_t)stream, partials, rows, C, count, eps, momentum, mean, rstd,
                     running_mean, running_var, num_batches_tracked, moments);
  AGL_CHECK_LAUNCH("agl_bn_stats_from_partials");
  return AGL_OK;
}

int agl_bn_stats_eval(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* rstd,
                      void* stream) {
  AGL_REQUIRE(running_mean && running_var && mean && rstd && C > 0, "agl_bn_stats_eval: bad argument");
  hipLaunchKernelGGL(bn_stats_eval, dim3(agl_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, running_mean, running_var,
                     C, eps, mean, rstd);
  AGL_CHECK_LAUNCH("agl_bn_stats_eval");
  return AGL_OK;
}

#define AGL_LPR_DISPATCH(KERNEL, ...)                                                                              \
  do {                                                                                                              \
    const int lpr_ = pick_lpr(HW);                                                                                  \
    const int rows_ = N * C;                                                                                        \
    if (lpr_ == 4) hipLaunchKernelGGL((KERNEL<4>), dim3(agl_cdiv(rows_, 64)), dim3(256), 0, st, __VA_ARGS__);        \
    else if (lpr_ == 16) hipLaunchKernelGGL((KERNEL<16>), dim3(agl_cdiv(rows_, 16)), dim3(256), 0, st, __VA_ARGS__); \
    else if (lpr_ == 64) hipLaunchKernelGGL((KERNEL<64>), dim3(agl_cdiv(rows_, 4)), dim3(256), 0, st, __VA_ARGS__);  \
    else hipLaunchKernelGGL((KERNEL<256>), dim3(rows_), dim3(256), 0, st, __VA_ARGS__);                              \
  } while (0)

static int fill_args(NormArgs& a, const float* x, const float* mean, const float* rstd, int mode, const float* p0,
                     const float* p1, const long long* labels, int relu, int N, int C, int HW, const char* who) {
  AGL_REQUIRE(x && mean && rstd && N > 0 && C > 0 && HW > 0, "%s: bad argument", who);
  AGL_REQUIRE(mode >= 0 && mode <= 3, "%s: bad mode %d", who, mode);
  AGL_REQUIRE(mode == 0 || p0, "%s: mode %d needs p0", who, mode);
  AGL_REQUIRE(mode != 1 || p1, "%s: mode 1 needs p1", who);
  AGL_REQUIRE(mode != 2 || labels, "%s: mode 2 needs labels", who);
  AGL_REQUIRE((long)N * C * HW < (1L << 31), "%s: tensor too large", who);
  a.x = x; a.mean = mean; a.rstd = rstd; a.p0 = p0; a.p1 = p1; a.labels = labels;
  a.mode = mode; a.relu = relu; a.N = N; a.C = C; a.HW = HW;
  a.map = nullptr; a.W = 0; a.src_w = 0;
  a.fscale = nullptr; a.fshift = nullptr; a.f_per_n = 0; a.y_bf16 = 0;
  return AGL_OK;
}
// gathered gamma|beta of mode 3 (gb_map: W device ints, the row AND column map of a square W x W map onto a src_w x src_w grid)
static int fill_gather(NormArgs& a, const int* gb_map, int W, int src_w, const char* who) {
  if (!gb_map) return AGL_OK;
  AGL_REQUIRE(a.mode == 3 && W > 0 && src_w > 0 && (long)W * W == a.HW, "%s: gb_map needs mode 3 and a square W x W map (W=%d, HW=%d)", who, W, a.HW);
  a.map = gb_map; a.W = W; a.src_w = src_w;
  return AGL_OK;
}

static int norm_apply_impl(const float* x, const float* mean, const float* rstd, int mode, const float* p0, const float* p1,
                           const long long* labels, const float* residual, int relu, float* y, int N, int C, int HW,
                           const int* gb_map, int W, int src_w, void* stream, int y_bf16);
int agl_norm_apply_fwd(const float* x, const float* mean, const float* rstd, int mode, const float* p0, const float* p1,
                       const long long* labels, const float* residual, int relu, float* y, int N, int C, int HW,
                       const int* gb_map, int W, int src_w, void* stream) {
  return norm_apply_impl(x, mean, rstd, mode, p0, p1, labels, residual, relu, y, N, C, HW, gb_map, W, src_w, stream, 0);
}
// The same with y stored as bf16 (round to nearest even): for an output whose only readers are bf16-mode convolutions — they would
// round the fp32 tensor to these very values when staging it — and the ReLU mask of agl_norm_bwd_y16.
int agl_norm_apply_fwd_y16(const float* x, const float* mean, const float* rstd, int mode, const float* p0, const float* p1,
                           const long long* labels, const float* residual, int relu, void* y_bf16, int N, int C, int HW,
                           const int* gb_map, int W, int src_w, void* stream) {
  return norm_apply_impl(x, mean, rstd, mode, p0, p1, labels, residual, relu, (float*)y_bf16, N, C, HW, gb_map, W, src_w, stream, 1);
}
static int norm_apply_impl(const float* x, const float* mean, const float* rstd, int mode, const float* p0, const float* p1,
                           const long long* labels, const float* residual, int relu, float* y, int N, int C, int HW,
                           const int* gb_map, int W, int src_w, void* stream, int y_bf16) {
  NormArgs a;
  int rc = fill_args(a, x, mean, rstd, mode, p0, p1, labels, relu, N, C, HW, "agl_norm_apply_fwd");
  if (rc) return rc;
  rc = fill_gather(a, gb_map, W, src_w, "agl_norm_apply_fwd");
  if (rc) return rc;
  AGL_REQUIRE(y, "agl_norm_apply_fwd: null output");
  a.y_bf16 = y_bf16;
  hipStream_t st = (hipStream_t)stream;
  const bool al16 = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)(mode == 3 && !gb_map ? p0 : nullptr)) & 15) == 0;
  if (HW >= 64 && HW % 4 == 0 && al16 && (!gb_map || W % 4 == 0)) {
    const int q = HW / 4, rows = N * C;
    if (q >= 256) hipLaunchKernelGGL((norm_apply_fwd4<256>), dim3(rows), dim3(256), 0, st, a, residual, y);
    else if (q >= 64) hipLaunchKernelGGL((norm_apply_fwd4<64>), dim3(agl_cdiv(rows, 4)), dim3(256), 0, st, a, residual, y);
    else hipLaunchKernelGGL((norm_apply_fwd4<16>), dim3(agl_cdiv(rows, 16)), dim3(256), 0, st, a, residual, y);
  } else {
    AGL_LPR_DISPATCH(norm_apply_fwd, a, residual, y);
  }
  AGL_CHECK_LAUNCH("agl_norm_apply_fwd");
  return AGL_OK;
}

long agl_norm_bwd_ws_bytes(int N, int C) { return ((long)N * C * 2 + (long)C * 2) * 4; }

// Backward of (stats +) apply.  batch_stats=1: statistics were computed from x (training); 0: constants (eval).
// y is the forward output (ReLU mask) and may be NULL when relu==0.
// dp0/dp1: mode 1 -> dgamma[C], dbeta[C] (overwritten, or added to when param_accumulate); mode 2 -> dtable[V][2C] (ALWAYS
//          accumulated into; a caller without a gradient slot zeroes it first); mode 3 -> dp0 = dgb[N][2C][HW] (overwritten).
//          Either may be NULL to skip parameter gradients (not mode 3).
static int norm_bwd_impl(const float* dy, const float* x, const float* y, const float* mean, const float* rstd, int mode,
                         const float* p0, const float* p1, const long long* labels, int relu, int batch_stats, float* dx,
                         float* dp0, float* dp1, int N, int C, int HW, int n_classes, int param_accumulate, const int* gb_map, const int* gb_lo,
                         int W, int src_w, void* ws, long ws_bytes, void* stream, const float* fscale, const float* fshift, int f_per_n);

int agl_norm_bwd(const float* dy, const float* x, const float* y, const float* mean, const float* rstd, int mode,
                 const float* p0, const float* p1, const long long* labels, int relu, int batch_stats, float* dx,
                 float* dp0, float* dp1, int N, int C, int HW, int n_classes, int param_accumulate, const int* gb_map, const int* gb_lo,
                 int W, int src_w, void* ws, long ws_bytes, void* stream) {
  AGL_REQUIRE(!relu || y, "agl_norm_bwd: relu needs the forward output y (or agl_norm_bwd_fold)");
  return norm_bwd_impl(dy, x, y, mean, rstd, mode, p0, p1, labels, relu, batch_stats, dx, dp0, dp1, N, C, HW, n_classes, param_accumulate,
                       gb_map, gb_lo, W, src_w, ws, ws_bytes, stream, nullptr, nullptr, 0);
}

// agl_norm_bwd for a forward output stored as bf16 (agl_norm_apply_fwd_y16): y is read for the ReLU mask only
int agl_norm_bwd_y16(const float* dy, const float* x, const void* y_bf16, const float* mean, const float* rstd, int mode,
                     const float* p0, const float* p1, const long long* labels, int relu, int batch_stats, float* dx,
                     float* dp0, float* dp1, int N, int C, int HW, int n_classes, int param_accumulate, const int* gb_map, const int* gb_lo,
                     int W, int src_w, void* ws, long ws_bytes, void* stream) {
  AGL_REQUIRE(!relu || y_bf16, "agl_norm_bwd_y16: relu needs the forward output");
  return norm_bwd_impl(dy, x, (const float*)y_bf16, mean, rstd, mode, p0, p1, labels, relu, batch_stats, dx, dp0, dp1, N, C, HW, n_classes,
                       param_accumulate, gb_map, gb_lo, W, src_w, ws, ws_bytes, stream, nullptr, nullptr, -1);
}

// Backward of a FOLDED normalise-modulate(+ReLU) (agl_norm_fold_table + agl_conv2d_fwd_fold): dy is the gradient with respect to the
// never-stored activation y = relu?(fmaf(x, scale, shift)), shift = beta - mean * scale (include/agl.h); the ReLU mask is recomputed from x and the tables.  Modes 0-2.
int agl_norm_bwd_fold(const float* dy, const float* x, const float* mean, const float* rstd, const float* fold_scale, const float* fold_shift,
                      int fold_per_n, int mode, const float* p0, const float* p1, const long long* labels, int relu, int batch_stats,
                      float* dx, float* dp0, float* dp1, int N, int C, int HW, int n_classes, int param_accumulate, void* ws, long ws_bytes,
                      void* stream) {
  AGL_REQUIRE(mode >= 0 && mode <= 2 && fold_scale && fold_shift, "agl_norm_bwd_fold: modes 0-2 with the tables of agl_norm_fold_table");
  return norm_bwd_impl(dy, x, nullptr, mean, rstd, mode, p0, p1, labels, relu, batch_stats, dx, dp0, dp1, N, C, HW, n_classes, param_accumulate,
                       nullptr, nullptr, 0, 0, ws, ws_bytes, stream, fold_scale, fold_shift, fold_per_n);
}

// Backward of SPADE's modulate(+ReLU) when the consuming convolution applied it while staging (agl_conv2d_fwd_spade): dy is the gradient
// with respect to the never-stored activation; the ReLU mask is recomputed from x, mean, rstd and gb with csrc/spade.h's expression.
int agl_norm_bwd_spade(const float* dy, const float* x, const float* mean, const float* rstd, const float* gb, int relu, int batch_stats,
                       float* dx, float* dgb, int N, int C, int HW, const int* gb_map, const int* gb_lo, int W, int src_w, void* ws,
                       long ws_bytes, void* stream) {
  return norm_bwd_impl(dy, x, nullptr, mean, rstd, 3, gb, nullptr, nullptr, relu, batch_stats, dx, dgb, nullptr, N, C, HW, 0, 0, gb_map, gb_lo,
                       W, src_w, ws, ws_bytes, stream, nullptr, nullptr, 0);
}

// cells[((n * C/8 + c/8) * G*G + cell) * 16 + j] = 1 + gamma[n][c/8*8 + j][cell], [.. + 8 + j] = beta[..] from gb (N, 2C, G, G): the layout the
// staging passes of agl_conv2d_fwd_spade / agl_conv2d_bwd_weight_spade read with 16-byte loads (csrc/pconv.h InFold::cells)
int agl_spade_cells(const float* gb, int N, int C, int G, float* cells, void* stream) {
  AGL_REQUIRE(gb && cells && N > 0 && C > 0 && C % 8 == 0 && G > 0 && (long)N * C * G * G < (1L << 28), "agl_spade_cells: bad argument");
  const long total = (long)N * (C / 8) * G * G;
  hipLaunchKernelGGL(spade_cells_k, dim3(agl_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, gb, cells, N, C, G * G, total);
  AGL_CHECK_LAUNCH("agl_spade_cells");
  return AGL_OK;
}

// scale[rows][C] = rstd[c] * gamma(r, c), shift[rows][C] = beta(r, c) of the normalise-modulate in `mode` (0: gamma 1, beta 0;
// 1: affine; 2: class table rows picked by labels) — rows = N for mode 2 (per object), else 1.
int agl_norm_fold_table(const float* mean, const float* rstd, int mode, const float* p0, const float* p1, const long long* labels, int N, int C,
                        float* scale, float* shift, void* stream) {
  NormArgs a;
  AGL_REQUIRE(mode >= 0 && mode <= 2 && scale && shift, "agl_norm_fold_table: modes 0-2");
  int rc = fill_args(a, mean, mean, rstd, mode, p0, p1, labels, 0, N, C, 1, "agl_norm_fold_table");
  if (rc) return rc;
  const int rows = mode == 2 ? N : 1;
  hipLaunchKernelGGL(norm_fold_table_k, dim3(agl_cdiv((long)rows * C, 256)), dim3(256), 0, (hipStream_t)stream, a, rows, scale, shift);
  AGL_CHECK_LAUNCH("agl_norm_fold_table");
  return AGL_OK;
}

static int norm_bwd_impl(const float* dy, const float* x, const float* y, const float* mean, const float* rstd, int mode,
                         const float* p0, const float* p1, const long long* labels, int relu, int batch_stats, float* dx,
                         float* dp0, float* dp1, int N, int C, int HW, int n_classes, int param_accumulate, const int* gb_map, const int* gb_lo,
                         int W, int src_w, void* ws, long ws_bytes, void* stream, const float* fscale, const float* fshift, int f_per_n) {
  NormArgs a;
  int rc = fill_args(a, x, mean, rstd, mode, p0, p1, labels, relu, N, C, HW, "agl_norm_bwd");
  if (rc) return rc;
  rc = fill_gather(a, gb_map, W, src_w, "agl_norm_bwd");
  if (rc) return rc;
  a.fscale = fscale; a.fshift = fshift; a.f_per_n = f_per_n < 0 ? 0 : f_per_n;
  a.y_bf16 = f_per_n < 0;      // (f_per_n = -1: the y16 entry point)
  AGL_REQUIRE(dy && dx && (!relu || y || fscale || mode == 3), "agl_norm_bwd: null pointer");
  AGL_REQUIRE(mode != 3 || dp0, "agl_norm_bwd: SPADE mode needs dgb output");
  if (!ws || ws_bytes < agl_norm_bwd_ws_bytes(N, C)) {
    agl_set_error("agl_norm_bwd: workspace too small");
    return AGL_ERR_WORKSPACE;
  }
  float* rowsum = (float*)ws;
  hipStream_t st = (hipStream_t)stream;
  // rows of >= 1024 pixels: 16-byte accesses (a workgroup per row either way)
  const bool vec4 = HW >= 1024 && HW % 4 == 0 && (!gb_map || W % 4 == 0) &&
                    (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)(mode == 3 ? dp0 : nullptr) |
                      (uintptr_t)(mode == 3 && !gb_map ? p0 : nullptr)) & 15) == 0;
  if (gb_lo) {      // d(gamma|beta) reduced to the class grid in the row pass
    AGL_REQUIRE(gb_map && mode == 3 && (W == 64 || W == 128) && src_w <= W, "agl_norm_bwd: gb_lo needs mode 3, gb_map and a 64- or 128-wide map");
    hipLaunchKernelGGL(norm_bwd_rows_gathered, dim3(N * C), dim3(W), 0, st, a, dy, y, gb_lo, rowsum, dp0);
  } else if (vec4) {
    hipLaunchKernelGGL(norm_bwd_rows4, dim3(N * C), dim3(256), 0, st, a, dy, y, rowsum, dp0);
  } else {
    AGL_LPR_DISPATCH(norm_bwd_rows, a, dy, y, rowsum, dp0);
  }
  AGL_CHECK_LAUNCH("agl_norm_bwd(rows)");
  if (mode == 2 && dp0) {
    AGL_REQUIRE(n_classes > 0, "agl_norm_bwd: mode 2 needs the number of table rows");
    hipLaunchKernelGGL(norm_bwd_table, dim3(agl_cdiv(C, 256), n_classes), dim3(256), 0, st, a, (const float*)rowsum, dp0);
    AGL_CHECK_LAUNCH("agl_norm_bwd(table)");
  }
  const float inv_m = 1.0f / (float)((long)N * HW);
  const float* chansum = nullptr;
  if (N > 64) {      // many rows per channel: the per-channel sums by their own launch (see norm_bwd_apply)
    float* cs = rowsum + (long)N * C * 2;
    hipLaunchKernelGGL(norm_bwd_channels, dim3(agl_cdiv(C, 4)), dim3(256), 0, st, a, (const float*)rowsum, cs, mode == 3 ? nullptr : dp0, dp1,
                       param_accumulate);
    AGL_CHECK_LAUNCH("agl_norm_bwd(channels)");
    chansum = cs;
  }
  if (vec4)
    hipLaunchKernelGGL(norm_bwd_apply4, dim3(N * C), dim3(256), 0, st, a, dy, y, (const float*)rowsum, chansum, inv_m, batch_stats, dx,
                       mode == 3 ? nullptr : dp0, dp1, param_accumulate);
  else
    AGL_LPR_DISPATCH(norm_bwd_apply, a, dy, y, (const float*)rowsum, chansum, inv_m, batch_stats, dx, mode == 3 ? nullptr : dp0, dp1,
                     param_accumulate);
  AGL_CHECK_LAUNCH("agl_norm_bwd(apply)");
  return AGL_OK;
}

}  // extern "C"
