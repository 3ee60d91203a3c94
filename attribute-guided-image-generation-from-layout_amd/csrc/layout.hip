// First stage of the layout encoder in closed form (reference models/generator_obj_att.py:489-497:
//   h = ([obj_att; z] (x) mask) -> c0 (1x1, pad 1) -> ConditionalBatchNorm2d -> ReLU -> c2 (4x4, stride 2, pad 1)).
// The input of c0 is rank-1 per object, so after c0 the (O,64,R+2,R+2) tensor is u[o,c] inside the object's mask
// and 0 elsewhere (border included).  Consequences used here — all exact in real arithmetic:
//   * the BatchNorm statistics are sums over objects weighted by the mask area (no pass over a 438 MB tensor);
//   * after CondBN + ReLU every object/channel has only TWO values: A[o,c] inside the mask, B[o,c] outside;
//   * c2 of a two-level image is  out[o,co,p] = sum_t WB[o,co,t] I_t(p) + WD[o,co,t] m_t(p)  with
//     WB = W . B, WD = W . (A-B) (two small GEMMs), I_t = "tap t is inside the image", m_t = mask value under tap t:
//     32 FMAs per output instead of 1024 MACs, and the (O,64,R+2,R+2) activations are never materialised.
// The backward pass mirrors it: per-tap sums of the output gradient, small GEMMs, and the closed-form CondBN/BN
// backward over (object, channel) pairs.
#include "agl_internal.h"

namespace {

// area[o] = number of ones of mask o
__global__ __launch_bounds__(256) void l1_area_k(const float* __restrict__ mask, float* __restrict__ area, int O, int RR) {
  __shared__ float sc[4];
  const int o = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < RR; i += 256) s += mask[(long)o * RR + i];
  s = block_sum_256(s, sc);
  if (threadIdx.x == 0) area[o] = s;
}

// Per channel: batch statistics of the virtual tensor x[o,c,p] = u[o,c] * M[o,p] over O * P elements (P = (R+2)^2),
// running-stat update like nn.BatchNorm2d (momentum, unbiased variance), one wave per channel.
__global__ __launch_bounds__(256) void l1_stats_k(const float* __restrict__ u, const float* __restrict__ area, int O, int C, double P,
                                                  float eps, float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                  float* __restrict__ rmean, float* __restrict__ rvar, long long* __restrict__ nbt) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int o = lane; o < O; o += 64) {
    const double v = u[(long)o * C + c], a = area[o];
    s1 += a * v; s2 += a * v * v;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) {
    const double M = (double)O * P, mu = s1 / M;
    double var = s2 / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
      rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mu);
      rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * var * (M / (M - 1.0)));
    }
  }
}

// A = relu(gamma*(u-mean)*rstd + beta), B = relu(gamma*(0-mean)*rstd + beta), D = A - B
__global__ void l1_levels_k(const float* __restrict__ u, const float* __restrict__ mean, const float* __restrict__ rstd,
                            const float* __restrict__ table, const long long* __restrict__ labels, float* __restrict__ A,
                            float* __restrict__ B, float* __restrict__ D, int O, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= O * C) return;
  const int o = i / C, c = i - o * C;
  const float* t = table + (long)labels[o] * 2 * C;
  const float g = t[c], b = t[C + c], r = rstd[c], m = mean[c];
  const float a = fmaxf(g * ((u[i] - m) * r) + b, 0.f), bb = fmaxf(g * ((0.f - m) * r) + b, 0.f);
  A[i] = a; B[i] = bb; D[i] = a - bb;
}

// Wr[(co*16+t)][c] = W[co][c][t]  (and back)
__global__ void l1_perm_k(const float* __restrict__ src, float* __restrict__ dst, int Co, int C, int to_rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Co * C * 16) return;
  const int co = i / (C * 16), r = i - co * (C * 16);
  if (to_rows) { const int c = r / 16, t = r - c * 16; dst[((long)co * 16 + t) * C + c] = src[i]; }
  else { const int t = r / C, c = r - t * C; dst[((long)co * C + c) * 16 + t] = src[i]; }
}

// tap geometry of the 4x4 stride-2 pad-1 convolution over the (R+2)x(R+2) padded grid
__device__ __forceinline__ void tap_bits(const float* __restrict__ mask, int R, int oy, int ox, unsigned& in_img, unsigned& in_mask) {
  in_img = 0; in_mask = 0;
  const int RP = R + 2;
#pragma unroll
  for (int kh = 0; kh < 4; ++kh)
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) {
      const int iy = 2 * oy - 1 + kh, ix = 2 * ox - 1 + kw;
      const bool im = (unsigned)iy < (unsigned)RP && (unsigned)ix < (unsigned)RP;
      const int my = iy - 1, mx = ix - 1;
      const bool mk = im && (unsigned)my < (unsigned)R && (unsigned)mx < (unsigned)R && mask[(long)my * R + mx] != 0.f;
      in_img |= (unsigned)im << (kh * 4 + kw);
      in_mask |= (unsigned)mk << (kh * 4 + kw);
    }
}

// y[o,co,p] = sum_t WB[o,co,t] I_t(p) + WD[o,co,t] m_t(p).  Block: one object x 64 output pixels; wave g owns the output
// channels g, g+4, ...; the per-object filters are staged in LDS and read as wave-wide broadcasts.
__global__ __launch_bounds__(256) void l1_pixels_k(const float* __restrict__ WB, const float* __restrict__ WD,
                                                   const float* __restrict__ mask, float* __restrict__ y, int O, int Co, int R, int OH) {
  extern __shared__ __attribute__((aligned(16))) float lw[];   // [2][Co*16]
  const int o = blockIdx.y, p0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < Co * 16; i += 256) {
    lw[i] = WB[(long)o * Co * 16 + i];
    lw[Co * 16 + i] = WD[(long)o * Co * 16 + i];
  }
  __syncthreads();
  const int p = p0 + lane, OHW = OH * OH;
  if (p >= OHW) return;
  unsigned bi, bm;
  tap_bits(mask + (long)o * R * R, R, p / OH, p % OH, bi, bm);
  float fi[16], fm[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) { fi[t] = (bi >> t & 1) ? 1.f : 0.f; fm[t] = (bm >> t & 1) ? 1.f : 0.f; }
  for (int co = g; co < Co; co += 4) {
    const float4* wb = reinterpret_cast<const float4*>(lw + co * 16);
    const float4* wd = reinterpret_cast<const float4*>(lw + Co * 16 + co * 16);
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b4 = wb[q], d4 = wd[q];
      acc = fmaf(b4.x, fi[4 * q + 0], acc); acc = fmaf(d4.x, fm[4 * q + 0], acc);
      acc = fmaf(b4.y, fi[4 * q + 1], acc); acc = fmaf(d4.y, fm[4 * q + 1], acc);
      acc = fmaf(b4.z, fi[4 * q + 2], acc); acc = fmaf(d4.z, fm[4 * q + 2], acc);
      acc = fmaf(b4.w, fi[4 * q + 3], acc); acc = fmaf(d4.w, fm[4 * q + 3], acc);
    }
    y[((long)o * Co + co) * OHW + p] = acc;
  }
}

// GB[o,co,t] = sum_p dy[o,co,p] I_t(p), GD[o,co,t] = sum_p dy[o,co,p] m_t(p).  Block: one object x 8 output channels.
// The tap bits of every output pixel are computed once per block into LDS (the mask is read once per 8 channels, not
// once per channel); wave g then owns channels g, g+4, ... and reduces its 32 tap sums with shuffles (fixed order).
constexpr int TAPSUM_CH = 8;
// Cost per element (the kernel is bound by its vector ALU, not by the 4-byte reads): the sixteen in-image indicators differ from "all
// taps inside" only on the first / last output row and column (the window leaves the padded grid there by exactly one tap), so
// GB[kh][kw] = S - [kh=0] R0 - [kh=3] RL - [kw=0] C0 - [kw=3] CL + the four corner terms (inclusion - exclusion): one sum and eight
// border sums instead of sixteen conditional adds; the sixteen mask indicators are uniform over most 64-pixel groups (all outside
// the object's box: nothing to add; all inside: one add) and only the groups that touch the box's boundary take the sixteen
// conditional adds.  Exact for any mask; 32 -> ~13 vector operations per element (563 -> 280 us at 210 x 128 x 65 x 65).
__global__ __launch_bounds__(256) void l1_tapsum_k(const float* __restrict__ dy, const float* __restrict__ mask, float* __restrict__ GB,
                                                   float* __restrict__ GD, int O, int Co, int R, int OH) {
  extern __shared__ unsigned l1_bits[];       // [OH*OH]: in_img | in_mask << 16
  const int o = blockIdx.y, lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int OHW = OH * OH;
  for (int p = threadIdx.x; p < OHW; p += 256) {
    unsigned bi, bm;
    tap_bits(mask + (long)o * R * R, R, p / OH, p % OH, bi, bm);
    l1_bits[p] = bi | (bm << 16);
  }
  __syncthreads();
  const int c_end = min(Co, (int)(blockIdx.x + 1) * TAPSUM_CH);
  for (int co = blockIdx.x * TAPSUM_CH + g; co < c_end; co += 4) {
    float sd[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) sd[t] = 0.f;
    float s_all = 0.f, r0 = 0.f, rl = 0.f, c0 = 0.f, cl = 0.f, v00 = 0.f, v0l = 0.f, vl0 = 0.f, vll = 0.f, s_full = 0.f;
    const float* d = dy + ((long)o * Co + co) * OHW;
    for (int p0 = 0; p0 < OHW; p0 += 64) {
      const int p = p0 + lane;
      const bool live = p < OHW;
      const unsigned bits = live ? l1_bits[p] : 0u;
      const float v = live ? d[p] : 0.f;
      const int py = p / OH, px = p - py * OH;
      const bool top = py == 0, bot = py == OH - 1, lef = px == 0, rig = px == OH - 1;
      s_all += v;
      r0 += top ? v : 0.f; rl += bot ? v : 0.f; c0 += lef ? v : 0.f; cl += rig ? v : 0.f;
      v00 += (top && lef) ? v : 0.f; v0l += (top && rig) ? v : 0.f; vl0 += (bot && lef) ? v : 0.f; vll += (bot && rig) ? v : 0.f;
      const unsigned bm = bits >> 16;
      const bool any_in = __any(live && bm != 0u), all_in = __all(!live || bm == 0xffffu);      // (wave-uniform)
      if (!any_in) continue;
      if (all_in) { s_full += v; continue; }
#pragma unroll
      for (int t = 0; t < 16; ++t) sd[t] += (bm >> t & 1) ? v : 0.f;
    }
    s_all = wave_sum(s_all); r0 = wave_sum(r0); rl = wave_sum(rl); c0 = wave_sum(c0); cl = wave_sum(cl);
    v00 = wave_sum(v00); v0l = wave_sum(v0l); vl0 = wave_sum(vl0); vll = wave_sum(vll); s_full = wave_sum(s_full);
#pragma unroll
    for (int t = 0; t < 16; ++t) sd[t] = wave_sum(sd[t]) + s_full;
    if (lane == 0) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int kh = t >> 2, kw = t & 3;
        float b = s_all;
        if (OH > 1) {      // (a one-row grid would have both its borders on the same row: not a shape of the path — R >= 2 gives OH >= 2)
          if (kh == 0) b -= r0;
          if (kh == 3) b -= rl;
          if (kw == 0) b -= c0;
          if (kw == 3) b -= cl;
          if (kh == 0 && kw == 0) b += v00;
          if (kh == 0 && kw == 3) b += v0l;
          if (kh == 3 && kw == 0) b += vl0;
          if (kh == 3 && kw == 3) b += vll;
        }
        GB[((long)o * Co + co) * 16 + t] = b;
        GD[((long)o * Co + co) * 16 + t] = sd[t];
      }
    }
  }
}

// Closed-form backward of ReLU -> CondBN -> BN statistics for the two-level tensor.
// Pass 1 (per object/channel): pre-activation grads, gamma/beta row sums; pass 2 (per channel): S1, S2; pass 3: du.
__global__ void l1_bwd_rows_k(const float* __restrict__ dA, const float* __restrict__ dB, const float* __restrict__ A,
                              const float* __restrict__ B, const float* __restrict__ u, const float* __restrict__ mean,
                              const float* __restrict__ rstd, const float* __restrict__ table, const long long* __restrict__ labels,
                              float* __restrict__ rows /* [O][C][4]: dxi, dxo, dgamma, dbeta */, int O, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= O * C) return;
  const int o = i / C, c = i - o * C;
  const float g = table[(long)labels[o] * 2 * C + c], r = rstd[c], m = mean[c];
  const float xi = (u[i] - m) * r, xo = (0.f - m) * r;
  const float pa = A[i] > 0.f ? dA[i] : 0.f, pb = B[i] > 0.f ? dB[i] : 0.f;
  rows[4 * (long)i + 0] = pa * g;
  rows[4 * (long)i + 1] = pb * g;
  rows[4 * (long)i + 2] = pa * xi + pb * xo;
  rows[4 * (long)i + 3] = pa + pb;
}
__global__ __launch_bounds__(256) void l1_bwd_chan_k(const float* __restrict__ rows, const float* __restrict__ u, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ chan, int O, int C) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  const float r = rstd[c], m = mean[c];
  double s1 = 0.0, s2 = 0.0;
  for (int o = lane; o < O; o += 64) {
    const long i = (long)o * C + c;
    const float dxi = rows[4 * i], dxo = rows[4 * i + 1];
    s1 += (double)dxi + dxo;
    s2 += (double)dxi * ((u[i] - m) * r) + (double)dxo * ((0.f - m) * r);
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { chan[2 * c] = (float)s1; chan[2 * c + 1] = (float)s2; }
}
__global__ void l1_bwd_du_k(const float* __restrict__ rows, const float* __restrict__ chan, const float* __restrict__ u,
                            const float* __restrict__ area, const float* __restrict__ mean, const float* __restrict__ rstd,
                            float inv_m, float* __restrict__ du, int O, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= O * C) return;
  const int o = i / C, c = i - o * C;
  const float r = rstd[c], xi = (u[i] - mean[c]) * r, a = area[o];
  du[i] = r * (rows[4 * (long)i] - a * chan[2 * c] * inv_m - a * xi * chan[2 * c + 1] * inv_m);
}
// dtable[v][c] += sum_{o: label==v} dgamma ; dtable[v][C+c] += ... dbeta  (one owner per cell, fixed order)
__global__ __launch_bounds__(256) void l1_bwd_table_k(const float* __restrict__ rows, const long long* __restrict__ labels,
                                                      float* __restrict__ dtable, int O, int C) {
  __shared__ int lab[1024];
  const int v = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  float g = 0.f, b = 0.f;
  for (int n0 = 0; n0 < O; n0 += 1024) {
    const int cnt = min(1024, O - n0);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += 256) lab[i] = (int)labels[n0 + i];
    __syncthreads();
    if (c < C)
      for (int i = 0; i < cnt; ++i)
        if (lab[i] == v) {
          g += rows[4 * ((long)(n0 + i) * C + c) + 2];
          b += rows[4 * ((long)(n0 + i) * C + c) + 3];
        }
  }
  if (c < C) { dtable[(long)v * 2 * C + c] += g; dtable[(long)v * 2 * C + C + c] += b; }
}

}  // namespace

extern "C" {

// area[O]; mean/rstd[C] (+ running stats); A, B, D = A - B  [O][C]
int agl_layout1_levels(const float* u, const float* mask, const long long* labels, const float* table, float* area, float* mean,
                       float* rstd, float* A, float* B, float* D, float* running_mean, float* running_var,
                       long long* num_batches_tracked, int O, int C, int R, float eps, float momentum, int training, void* stream) {
  AGL_REQUIRE(u && mask && labels && table && area && mean && rstd && A && B && D && O > 0 && C > 0 && R > 0, "agl_layout1_levels: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(l1_area_k, dim3(O), dim3(256), 0, st, mask, area, O, R * R);
  AGL_CHECK_LAUNCH("agl_layout1_levels(area)");
  if (training) {
    hipLaunchKernelGGL(l1_stats_k, dim3(agl_cdiv(C, 4)), dim3(256), 0, st, u, (const float*)area, O, C, (double)(R + 2) * (R + 2), eps,
                       momentum, mean, rstd, running_mean, running_var, num_batches_tracked);
    AGL_CHECK_LAUNCH("agl_layout1_levels(stats)");
  }
  hipLaunchKernelGGL(l1_levels_k, dim3(agl_cdiv((long)O * C, 256)), dim3(256), 0, st, u, (const float*)mean, (const float*)rstd, table,
                     labels, A, B, D, O, C);
  AGL_CHECK_LAUNCH("agl_layout1_levels(levels)");
  return AGL_OK;
}

// to_rows=1: W[Co][C][16] -> Wr[Co*16][C];  0: the inverse
int agl_layout1_permute(const float* src, float* dst, int Co, int C, int to_rows, void* stream) {
  AGL_REQUIRE(src && dst && Co > 0 && C > 0, "agl_layout1_permute: bad argument");
  hipLaunchKernelGGL(l1_perm_k, dim3(agl_cdiv((long)Co * C * 16, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, Co, C, to_rows);
  AGL_CHECK_LAUNCH("agl_layout1_permute");
  return AGL_OK;
}

// y[O][Co][OH][OH], OH = R/2 + 1, from the per-object tap filters WB, WD [O][Co][16]
int agl_layout1_pixels(const float* WB, const float* WD, const float* mask, float* y, int O, int Co, int R, void* stream) {
  AGL_REQUIRE(WB && WD && mask && y && O > 0 && Co > 0 && R > 0 && R % 2 == 0 && Co * 16 * 2 * 4 <= 64 * 1024, "agl_layout1_pixels: bad argument");
  const int OH = R / 2 + 1;
  hipLaunchKernelGGL(l1_pixels_k, dim3(agl_cdiv(OH * OH, 64), O), dim3(256), (size_t)Co * 16 * 2 * sizeof(float), (hipStream_t)stream, WB, WD,
                     mask, y, O, Co, R, OH);
  AGL_CHECK_LAUNCH("agl_layout1_pixels");
  return AGL_OK;
}

int agl_layout1_tapsum(const float* dy, const float* mask, float* GB, float* GD, int O, int Co, int R, void* stream) {
  AGL_REQUIRE(dy && mask && GB && GD && O > 0 && Co > 0 && R > 0 && R % 2 == 0, "agl_layout1_tapsum: bad argument");
  const int OH = R / 2 + 1;
  AGL_REQUIRE((long)OH * OH * 4 <= 64 * 1024, "agl_layout1_tapsum: map too large for the LDS bit table");
  hipLaunchKernelGGL(l1_tapsum_k, dim3(agl_cdiv(Co, TAPSUM_CH), O), dim3(256), (size_t)OH * OH * sizeof(unsigned), (hipStream_t)stream, dy, mask,
                     GB, GD, O, Co, R, OH);
  AGL_CHECK_LAUNCH("agl_layout1_tapsum");
  return AGL_OK;
}

// du[O][C], dtable[V][2C] (+=) from dA, dB.  ws: (O*C*4 + 2*C) floats.
int agl_layout1_levels_bwd(const float* dA, const float* dB, const float* A, const float* B, const float* u, const float* area,
                           const float* mean, const float* rstd, const float* table, const long long* labels, float* du,
                           float* dtable, int O, int C, int R, int V, int training, void* ws, long ws_bytes, void* stream) {
  AGL_REQUIRE(dA && dB && A && B && u && area && mean && rstd && table && labels && du && O > 0 && C > 0 && V > 0, "agl_layout1_levels_bwd: bad argument");
  if (!ws || ws_bytes < ((long)O * C * 4 + 2 * C) * 4) { agl_set_error("agl_layout1_levels_bwd: workspace too small"); return AGL_ERR_WORKSPACE; }
  float* rows = (float*)ws;
  float* chan = rows + (long)O * C * 4;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(l1_bwd_rows_k, dim3(agl_cdiv((long)O * C, 256)), dim3(256), 0, st, dA, dB, A, B, u, mean, rstd, table, labels, rows, O, C);
  AGL_CHECK_LAUNCH("agl_layout1_levels_bwd(rows)");
  if (training) {
    hipLaunchKernelGGL(l1_bwd_chan_k, dim3(agl_cdiv(C, 4)), dim3(256), 0, st, (const float*)rows, u, mean, rstd, chan, O, C);
  } else {
    hipMemsetAsync(chan, 0, 2 * C * sizeof(float), st);
  }
  AGL_CHECK_LAUNCH("agl_layout1_levels_bwd(chan)");
  const float inv_m = 1.0f / ((float)O * (float)(R + 2) * (float)(R + 2));
  hipLaunchKernelGGL(l1_bwd_du_k, dim3(agl_cdiv((long)O * C, 256)), dim3(256), 0, st, (const float*)rows, (const float*)chan, u, area, mean,
                     rstd, inv_m, du, O, C);
  AGL_CHECK_LAUNCH("agl_layout1_levels_bwd(du)");
  if (dtable) {
    hipLaunchKernelGGL(l1_bwd_table_k, dim3(agl_cdiv(C, 256), V), dim3(256), 0, st, (const float*)rows, labels, dtable, O, C);
    AGL_CHECK_LAUNCH("agl_layout1_levels_bwd(table)");
  }
  return AGL_OK;
}

}  // extern "C"
