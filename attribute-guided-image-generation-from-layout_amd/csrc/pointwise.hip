// HBM-bound elementwise / gather / small-reduction kernels of the G+D path (fp32, NCHW):
// bilinear box crop (fwd gather, bwd scatter-add), ConvLSTM gate math, ReLU backward, 2x2 average
// pool, nearest up-sampling (and its sum-pool adjoint), spatial sums, row gathers, channel sums,
// the VAE re-parameterisation and fused Adam.
#include "agl_internal.h"

namespace {

constexpr int TPB = 256;
inline int nblocks(long n) { long b = (n + TPB - 1) / TPB; return (int)(b < 1 ? 1 : b); }

// ---------------------------------------------------------------- crop
// torch.linspace(0,1,steps)[j] and torch.linspace(1,0,steps)[j] as the CPU kernel evaluates them
// (two half ramps anchored at start / end); models/bilinear.py:272-275.
__device__ __forceinline__ void lin_weights(int j, int steps, float& w_start, float& w_end) {
  if (steps == 1) { w_start = 1.f; w_end = 0.f; return; }
  const float step_up = 1.0f / (float)(steps - 1);
  const float step_dn = -1.0f / (float)(steps - 1);
  if (j < steps / 2) {
    w_end = step_up * (float)j;
    w_start = 1.0f + step_dn * (float)j;
  } else {
    w_end = 1.0f - step_up * (float)(steps - 1 - j);
    w_start = 0.0f - step_dn * (float)(steps - 1 - j);
  }
}

__device__ __forceinline__ float unnormalize(float g, int size, int align) {
  return align ? (g + 1.f) * 0.5f * (float)(size - 1) : ((g + 1.f) * (float)size - 1.f) * 0.5f;
}

struct CropGeom { int x0, y0; float wx1, wy1; };  // north-west tap and the weight of the +1 taps

__device__ __forceinline__ CropGeom crop_geom(const float* box, int i, int j, int HH, int WW, int H, int W, int align) {
  const float bx0 = 2.f * box[0] - 1.f, by0 = 2.f * box[1] - 1.f, bx1 = 2.f * box[2] - 1.f, by1 = 2.f * box[3] - 1.f;
  float ws, we;
  lin_weights(j, WW, ws, we);
  const float gx = ws * bx0 + we * bx1;
  lin_weights(i, HH, ws, we);
  const float gy = ws * by0 + we * by1;
  const float ix = unnormalize(gx, W, align), iy = unnormalize(gy, H, align);
  const float fx = floorf(ix), fy = floorf(iy);
  CropGeom g;
  g.x0 = (int)fx; g.y0 = (int)fy; g.wx1 = ix - fx; g.wy1 = iy - fy;
  return g;
}

__global__ void crop_fwd(const float* __restrict__ feats, const float* __restrict__ boxes, const long long* __restrict__ o2i,
                         float* __restrict__ out, int N, int B, int C, int H, int W, int HH, int WW, int align) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  const long total = (long)B * C * HH * WW;
  if (idx >= total) return;
  const int j = (int)(idx % WW);
  long t = idx / WW;
  const int i = (int)(t % HH); t /= HH;
  const int c = (int)(t % C);
  const int b = (int)(t / C);
  const CropGeom g = crop_geom(boxes + 4 * b, i, j, HH, WW, H, W, align);
  const long long img = o2i[b];
  // a box that names no image of the batch (models/bilinear.py:122-123 asserts on the host) must not read foreign memory:
  // the crop is poisoned with NaN instead (this call only enqueues; the Python host validates CPU-resident indices first)
  if ((unsigned long long)img >= (unsigned long long)N) { out[idx] = nanf(""); return; }
  const float* src = feats + ((long)img * C + c) * H * W;
  const bool xa = (unsigned)g.x0 < (unsigned)W, xb = (unsigned)(g.x0 + 1) < (unsigned)W;
  const bool ya = (unsigned)g.y0 < (unsigned)H, yb = (unsigned)(g.y0 + 1) < (unsigned)H;
  const float wx0 = 1.f - g.wx1, wy0 = 1.f - g.wy1;
  float v = 0.f;
  if (ya && xa) v += src[g.y0 * W + g.x0] * (wx0 * wy0);
  if (ya && xb) v += src[g.y0 * W + g.x0 + 1] * (g.wx1 * wy0);
  if (yb && xa) v += src[(g.y0 + 1) * W + g.x0] * (wx0 * g.wy1);
  if (yb && xb) v += src[(g.y0 + 1) * W + g.x0 + 1] * (g.wx1 * g.wy1);
  out[idx] = v;
}

__global__ void crop_bwd(const float* __restrict__ dout, const float* __restrict__ boxes, const long long* __restrict__ o2i,
                         float* __restrict__ dfeats, int N, int B, int C, int H, int W, int HH, int WW, int align) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  const long total = (long)B * C * HH * WW;
  if (idx >= total) return;
  const int j = (int)(idx % WW);
  long t = idx / WW;
  const int i = (int)(t % HH); t /= HH;
  const int c = (int)(t % C);
  const int b = (int)(t / C);
  const CropGeom g = crop_geom(boxes + 4 * b, i, j, HH, WW, H, W, align);
  const long long img = o2i[b];
  if ((unsigned long long)img >= (unsigned long long)N) return;          // out-of-range box: nothing to scatter into
  float* dst = dfeats + ((long)img * C + c) * H * W;
  const bool xa = (unsigned)g.x0 < (unsigned)W, xb = (unsigned)(g.x0 + 1) < (unsigned)W;
  const bool ya = (unsigned)g.y0 < (unsigned)H, yb = (unsigned)(g.y0 + 1) < (unsigned)H;
  const float wx0 = 1.f - g.wx1, wy0 = 1.f - g.wy1;
  const float d = dout[idx];
  if (ya && xa) atomicAdd(dst + g.y0 * W + g.x0, d * (wx0 * wy0));
  if (ya && xb) atomicAdd(dst + g.y0 * W + g.x0 + 1, d * (g.wx1 * wy0));
  if (yb && xa) atomicAdd(dst + (g.y0 + 1) * W + g.x0, d * (wx0 * g.wy1));
  if (yb && xb) atomicAdd(dst + (g.y0 + 1) * W + g.x0 + 1, d * (g.wx1 * g.wy1));
}

// One axis of crop_geom: source coordinate of crop sample j (of `steps`) of a box side [b0, b1] (already mapped to [-1, 1]).
__device__ __forceinline__ float crop_axis(float b0, float b1, int j, int steps, int size, int align) {
  float ws, we;
  lin_weights(j, steps, ws, we);
  return unnormalize(ws * b0 + we * b1, size, align);
}
// Candidate samples of one axis whose two taps can touch source position p: the coordinate is monotone in j and linear up to
// rounding, so the range comes from the end points, widened by one sample on either side (every candidate is re-checked exactly).
__device__ __forceinline__ void crop_axis_range(float b0, float b1, int steps, int size, int align, int p, int& lo, int& hi) {
  lo = 0; hi = steps - 1;
  if (steps < 2) return;
  const float c0 = crop_axis(b0, b1, 0, steps, size, align), c1 = crop_axis(b0, b1, steps - 1, steps, size, align);
  const float slope = (c1 - c0) / (float)(steps - 1);
  if (!(fabsf(slope) > 1e-6f)) return;
  const float ta = ((float)(p - 1) - c0) / slope, tb = ((float)(p + 1) - c0) / slope;
  const float tlo = fminf(ta, tb), thi = fmaxf(ta, tb);
  if (!(thi >= -2.f && tlo <= (float)steps + 1.f)) { lo = 1; hi = 0; return; }      // (empty: the box never reaches p)
  lo = max(0, (int)floorf(fmaxf(tlo, -2.f)) - 1);
  hi = min(steps - 1, (int)ceilf(fminf(thi, (float)steps + 1.f)) + 1);
}

// Backward of crop_fwd as a GATHER, for a non-decreasing box -> image map (the boxes of an image are a contiguous run — what
// models/bilinear.py:77-90 builds and the training loop always passes): one thread per source element (n, c, y, x) walks the boxes
// of its image in order and, per box, the crop samples (i, j) whose bilinear footprint covers (y, x), adding d * (wx * wy) in a
// fixed order — no atomics, bit-reproducible.  Weights are those of crop_fwd (same functions).  dfeats is added to.
__global__ void crop_bwd_sorted(const float* __restrict__ dout, const float* __restrict__ boxes, const long long* __restrict__ o2i,
                                float* __restrict__ dfeats, int N, int B, int C, int H, int W, int HH, int WW, int align) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  const long total = (long)N * C * H * W;
  if (idx >= total) return;
  const int x = (int)(idx % W);
  long t = idx / W;
  const int y = (int)(t % H); t /= H;
  const int c = (int)(t % C);
  const int n = (int)(t / C);
  auto lower = [&](long long v) {          // first box whose image id is >= v
    int lo = 0, hi = B;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (o2i[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
  };
  const int b0 = lower(n), b1 = lower((long long)n + 1);
  float acc = 0.f;
  for (int b = b0; b < b1; ++b) {
    const float* box = boxes + 4 * b;
    const float bx0 = 2.f * box[0] - 1.f, by0 = 2.f * box[1] - 1.f, bx1 = 2.f * box[2] - 1.f, by1 = 2.f * box[3] - 1.f;
    int ilo, ihi, jlo, jhi;
    crop_axis_range(by0, by1, HH, H, align, y, ilo, ihi);
    crop_axis_range(bx0, bx1, WW, W, align, x, jlo, jhi);
    const float* d = dout + ((long)b * C + c) * HH * WW;
    for (int i = ilo; i <= ihi; ++i) {
      const float iy = crop_axis(by0, by1, i, HH, H, align), fy = floorf(iy);
      const int y0 = (int)fy;
      const float wy1 = iy - fy;
      float wy;
      if (y0 == y) wy = 1.f - wy1; else if (y0 + 1 == y) wy = wy1; else continue;
      for (int j = jlo; j <= jhi; ++j) {
        const float ix = crop_axis(bx0, bx1, j, WW, W, align), fx = floorf(ix);
        const int x0 = (int)fx;
        const float wx1 = ix - fx;
        float wx;
        if (x0 == x) wx = 1.f - wx1; else if (x0 + 1 == x) wx = wx1; else continue;
        acc += d[i * WW + j] * (wx * wy);
      }
    }
  }
  dfeats[idx] += acc;
}

// ---------------------------------------------------------------- ConvLSTM gates (i, f, o, g order)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// cch / dh_b may be `splits` partial outputs `stride` floats apart (the unreduced slabs of a convolution called with AGL_CONV_DEFER_SUM):
// they are added first, slab 0 upwards from 0.f — the order and association of the split-K epilogue, so the sums are the same numbers
__device__ __forceinline__ float slab_sum(const float* __restrict__ p, long o, int splits, long stride) {
  if (splits <= 1) return p[o];
  float v = 0.f;
  for (int z = 0; z < splits; ++z) v += p[(long)z * stride + o];
  return v;
}

__global__ void lstm_gates_fwd(const float* __restrict__ ccx, const long long* __restrict__ rows, const float* __restrict__ cch,
                               const float* __restrict__ c_prev, float* __restrict__ h, float* __restrict__ c,
                               float* __restrict__ gates, int B, int hid, int S, int splits, long stride) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  const long total = (long)B * hid * S;
  if (idx >= total) return;
  const int s = (int)(idx % S);
  long t = idx / S;
  const int ch = (int)(t % hid);
  const int b = (int)(t / hid);
  const long src = (rows ? (long)rows[b] : (long)b) * 4 * hid * S;
  const long loc = (long)b * 4 * hid * S;
  float pre[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const long o = (long)(q * hid + ch) * S + s;
    pre[q] = ccx[src + o] + (cch ? slab_sum(cch, loc + o, splits, stride) : 0.f);
  }
  const float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), go = sigmoidf_(pre[2]), gg = tanhf(pre[3]);
  const float cp = c_prev ? c_prev[idx] : 0.f;
  const float cn = gf * cp + gi * gg;
  c[idx] = cn;
  h[idx] = go * tanhf(cn);
  gates[loc + (long)(0 * hid + ch) * S + s] = gi;
  gates[loc + (long)(1 * hid + ch) * S + s] = gf;
  gates[loc + (long)(2 * hid + ch) * S + s] = go;
  gates[loc + (long)(3 * hid + ch) * S + s] = gg;
}

// dh_a (+ dh_b over the first Bb rows) and dc_next (first Bc rows) -> dcc (pre-activation grads), dc_prev
__global__ void lstm_gates_bwd(const float* __restrict__ dh_a, const float* __restrict__ dh_b, int Bb,
                               const float* __restrict__ dc_next, int Bc, const float* __restrict__ gates,
                               const float* __restrict__ c_prev, const float* __restrict__ c, float* __restrict__ dcc,
                               float* __restrict__ dc_prev, int B, int hid, int S, int splits, long stride) {
  const long idx = (long)blockIdx.x * TPB + threadIdx.x;
  const long total = (long)B * hid * S;
  if (idx >= total) return;
  const int s = (int)(idx % S);
  long t = idx / S;
  const int ch = (int)(t % hid);
  const int b = (int)(t / hid);
  const long loc = (long)b * 4 * hid * S;
  const float gi = gates[loc + (long)(0 * hid + ch) * S + s], gf = gates[loc + (long)(1 * hid + ch) * S + s];
  const float go = gates[loc + (long)(2 * hid + ch) * S + s], gg = gates[loc + (long)(3 * hid + ch) * S + s];
  float dh = dh_a ? dh_a[idx] : 0.f;
  if (dh_b && b < Bb) dh += slab_sum(dh_b, idx, splits, stride);
  const float tc = tanhf(c[idx]);
  float dc = dh * go * (1.f - tc * tc);
  if (dc_next && b < Bc) dc += dc_next[idx];
  const float cp = c_prev ? c_prev[idx] : 0.f;
  dcc[loc + (long)(0 * hid + ch) * S + s] = dc * gg * gi * (1.f - gi);
  dcc[loc + (long)(1 * hid + ch) * S + s] = dc * cp * gf * (1.f - gf);
  dcc[loc + (long)(2 * hid + ch) * S + s] = dh * tc * go * (1.f - go);
  dcc[loc + (long)(3 * hid + ch) * S + s] = dc * gi * (1.f - gg * gg);
  dc_prev[idx] = dc * gf;
}

// ---------------------------------------------------------------- small elementwise helpers
__global__ void relu_bwd_k(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long n) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

__global__ void axpby_k(const float* __restrict__ a, const float* __restrict__ b, float alpha, float beta, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) out[i] = alpha * a[i] + (b ? beta * b[i] : 0.f);
}

// out[r] = src[rows[r]] for rows of `len` floats (accumulate: out[r] += ...)
__global__ void gather_rows_k(const float* __restrict__ src, const long long* __restrict__ rows, float* __restrict__ out, long R, long len, int accumulate) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= R * len) return;
  const long r = i / len, o = i - r * len;
  const float v = src[(long)rows[r] * len + o];
  out[i] = accumulate ? out[i] + v : v;
}
// out[rows[r]] = src[r]  (rows must be unique)
__global__ void scatter_rows_k(const float* __restrict__ src, const long long* __restrict__ rows, float* __restrict__ out, long R, long len) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= R * len) return;
  const long r = i / len, o = i - r * len;
  out[(long)rows[r] * len + o] = src[i];
}

// 2x2 average pool, optional ReLU on the input (models/discriminator.py:25-26 and the in-place ReLU of :71)
// (index arithmetic in 32 bits: the entry points bound every tensor below 2^31 elements; 64-bit div/mod per element made
//  these kernels instruction-bound at half the HBM rate)
__global__ void avgpool2_fwd_k(const float* __restrict__ x, float* __restrict__ y, long NC, int H, int W, int in_relu) {
  const unsigned OH = H / 2, OW = W / 2;
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  if (i >= (unsigned)(NC * OH * OW)) return;
  const unsigned ow = i % OW, t = i / OW, oh = t % OH, nc = t / OH;
  const float* p = x + (long)nc * H * W + (long)(2 * oh) * W + 2 * ow;
  float a = p[0], b = p[1], c = p[W], d = p[W + 1];
  if (in_relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); c = fmaxf(c, 0.f); d = fmaxf(d, 0.f); }
  y[i] = (a + b + c + d) * 0.25f;
}
// dx = 0.25*dy broadcast over the 2x2 window (masked by x>0 when in_relu); accumulate: dx += ...
__global__ void avgpool2_bwd_k(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, long NC, int H, int W, int in_relu, int accumulate) {
  const unsigned OH = H / 2, OW = W / 2;
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  if (i >= (unsigned)(NC * H * W)) return;
  const unsigned w = i % W, t = i / W, h = t % H, nc = t / H;
  float v = 0.f;
  if (h < 2 * OH && w < 2 * OW) v = 0.25f * dy[(long)nc * OH * OW + (long)(h / 2) * OW + w / 2];
  if (in_relu && !(x[i] > 0.f)) v = 0.f;
  dx[i] = accumulate ? dx[i] + v : v;
}
// the same for W % 4 == 0: four consecutive columns per thread (16-byte accesses of x / dx, one 8-byte read of dy)
__global__ void avgpool2_bwd4_k(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, long NC, int H, int W, int in_relu, int accumulate) {
  const unsigned OW = W / 2, W4 = W / 4;
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  if (i >= (unsigned)(NC * H * W4)) return;
  const unsigned w4 = i % W4, t = i / W4, h = t % H, nc = t / H;
  const float2 d = *reinterpret_cast<const float2*>(dy + (long)nc * (H / 2) * OW + (long)(h / 2) * OW + 2 * w4);
  float4 v = {0.25f * d.x, 0.25f * d.x, 0.25f * d.y, 0.25f * d.y};
  const long o = 4L * i;
  if (in_relu) {
    const float4 xv = *reinterpret_cast<const float4*>(x + o);
    if (!(xv.x > 0.f)) v.x = 0.f;
    if (!(xv.y > 0.f)) v.y = 0.f;
    if (!(xv.z > 0.f)) v.z = 0.f;
    if (!(xv.w > 0.f)) v.w = 0.f;
  }
  if (accumulate) {
    const float4 ov = *reinterpret_cast<const float4*>(dx + o);
    v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w;
  }
  *reinterpret_cast<float4*>(dx + o) = v;
}

// Forward with wide accesses and an fp32 or bf16 input (IN16): a thread owns two output rows' worth of one 8-column strip — it reads
// 2 rows x 8 columns (fp32: two 16-byte loads per row; bf16: one) and writes 4 outputs (one 16-byte store).  W % 8 == 0, H even.
template <bool IN16>
__global__ void avgpool2_fwd_w_k(const void* __restrict__ xv, float* __restrict__ y, long NC, int H, int W, int in_relu) {
  const unsigned OH = H / 2, OW = W / 2, Q = W / 8;
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  if (i >= (unsigned)(NC * OH * Q)) return;
  const unsigned q = i % Q, t = i / Q, oh = t % OH, nc = t / OH;
  const long base = (long)nc * H * W + (long)(2 * oh) * W + 8 * q;
  float r[2][8];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if constexpr (IN16) {
      const uint4 b = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(xv) + base + (long)k * W);
      const unsigned u[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { r[k][2 * j] = __builtin_bit_cast(float, u[j] << 16); r[k][2 * j + 1] = __builtin_bit_cast(float, u[j] & 0xffff0000u); }
    } else {
      const float* p = reinterpret_cast<const float*>(xv) + base + (long)k * W;
      const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
      r[k][0] = a.x; r[k][1] = a.y; r[k][2] = a.z; r[k][3] = a.w; r[k][4] = b.x; r[k][5] = b.y; r[k][6] = b.z; r[k][7] = b.w;
    }
  }
  if (in_relu) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) r[k][j] = fmaxf(r[k][j], 0.f);
  }
  float4 o;      // (the same order of additions as avgpool2_fwd_k: a + b + c + d)
  o.x = (r[0][0] + r[0][1] + r[1][0] + r[1][1]) * 0.25f; o.y = (r[0][2] + r[0][3] + r[1][2] + r[1][3]) * 0.25f;
  o.z = (r[0][4] + r[0][5] + r[1][4] + r[1][5]) * 0.25f; o.w = (r[0][6] + r[0][7] + r[1][6] + r[1][7]) * 0.25f;
  *reinterpret_cast<float4*>(y + (long)nc * OH * OW + (long)oh * OW + 4 * q) = o;
}
// avgpool2_bwd4_k with the ReLU mask read from a bf16 x (four elements per 8-byte load)
__global__ void avgpool2_bwd4_m16_k(const float* __restrict__ dy, const unsigned short* __restrict__ x, float* __restrict__ dx, long NC, int H, int W,
                                    int accumulate) {
  const unsigned OW = W / 2, W4 = W / 4;
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  if (i >= (unsigned)(NC * H * W4)) return;
  const unsigned w4 = i % W4, t = i / W4, h = t % H, nc = t / H;
  const float2 d = *reinterpret_cast<const float2*>(dy + (long)nc * (H / 2) * OW + (long)(h / 2) * OW + 2 * w4);
  float4 v = {0.25f * d.x, 0.25f * d.x, 0.25f * d.y, 0.25f * d.y};
  const long o = 4L * i;
  const uint2 b = *reinterpret_cast<const uint2*>(x + o);
  if (!(__builtin_bit_cast(float, b.x << 16) > 0.f)) v.x = 0.f;
  if (!(__builtin_bit_cast(float, b.x & 0xffff0000u) > 0.f)) v.y = 0.f;
  if (!(__builtin_bit_cast(float, b.y << 16) > 0.f)) v.z = 0.f;
  if (!(__builtin_bit_cast(float, b.y & 0xffff0000u) > 0.f)) v.w = 0.f;
  if (accumulate) {
    const float4 ov = *reinterpret_cast<const float4*>(dx + o);
    v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w;
  }
  *reinterpret_cast<float4*>(dx + o) = v;
}

// ---- channel-blocked bf16 tensors [N][C/8][H][W][8] (the discriminators' block chain in bf16 arithmetic) -----------------------
// fp32 / bf16 NCHW -> blocked bf16: a thread owns one piece (8 channels of one pixel): 8 loads a channel stride apart (consecutive threads,
// consecutive pixels: coalesced per channel), one 16-byte store
template <bool IN16>
__global__ void to_blocked_k(const void* __restrict__ xv, uint4* __restrict__ y, long NG, int HW) {      // NG = N * C / 8
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= NG * HW) return;
  const long g = i / HW, pix = i - g * HW;
  unsigned short h[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const long o = (g * 8 + j) * HW + pix;
    if constexpr (IN16) h[j] = reinterpret_cast<const unsigned short*>(xv)[o];
    else h[j] = __builtin_bit_cast(unsigned short, (__bf16)reinterpret_cast<const float*>(xv)[o]);
  }
  y[i] = uint4{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16), (unsigned)h[4] | ((unsigned)h[5] << 16),
               (unsigned)h[6] | ((unsigned)h[7] << 16)};
}
// avg_pool2(relu?(x)) of a blocked bf16 x into an fp32 NCHW y: a thread owns one output pixel of 8 channels — four 16-byte pieces in
// (consecutive threads: consecutive 32-byte runs), eight 4-byte stores out, each coalesced along x across the threads (same order of
// additions as avgpool2_fwd_k)
__global__ void avgpool2_fwd_blk_k(const uint4* __restrict__ x, float* __restrict__ y, long NG, int H, int W, int in_relu) {
  const unsigned OH = H / 2, OW = W / 2;
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= NG * OH * OW) return;
  const unsigned ox = (unsigned)(i % OW);
  const long t = i / OW;
  const unsigned oy = (unsigned)(t % OH);
  const long g = t / OH;
  float v[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint4 b = x[(g * H + 2 * oy + (k >> 1)) * W + 2 * ox + (k & 1)];
    const unsigned u[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float lo = __builtin_bit_cast(float, u[j] << 16), hi = __builtin_bit_cast(float, u[j] & 0xffff0000u);
      if (in_relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
      v[k][2 * j] = lo; v[k][2 * j + 1] = hi;
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) y[((g * 8 + c) * OH + oy) * (long)OW + ox] = (v[0][c] + v[1][c] + v[2][c] + v[3][c]) * 0.25f;
}
// Backward of that pool with the ReLU mask read from the blocked x: a thread owns four consecutive input pixels of 8 channels — four mask
// pieces (64 contiguous bytes), one 8-byte dy load and one 16-byte store per channel.  W % 4 == 0.
__global__ void avgpool2_bwd_mblk_k(const float* __restrict__ dy, const uint4* __restrict__ x, float* __restrict__ dx, long NG, int H, int W,
                                    int accumulate) {
  const unsigned OH = H / 2, OW = W / 2, W4 = W / 4;
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= NG * H * W4) return;
  const unsigned x4 = (unsigned)(i % W4);
  const long t = i / W4;
  const unsigned Y = (unsigned)(t % H);
  const long g = t / H;
  uint4 mb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) mb[k] = x[(g * H + Y) * W + 4 * x4 + k];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float2 d = *reinterpret_cast<const float2*>(dy + ((g * 8 + c) * OH + (Y >> 1)) * (long)OW + 2 * x4);
    float4 v = {0.25f * d.x, 0.25f * d.x, 0.25f * d.y, 0.25f * d.y};
    float* vv = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned u = reinterpret_cast<const unsigned*>(&mb[k])[c >> 1];
      const float mval = __builtin_bit_cast(float, (c & 1) ? (u & 0xffff0000u) : (u << 16));
      if (!(mval > 0.f)) vv[k] = 0.f;
    }
    float4* o = reinterpret_cast<float4*>(dx + ((g * 8 + c) * H + Y) * (long)W + 4 * x4);
    if (accumulate) { const float4 ov = *o; v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w; }
    *o = v;
  }
}

// nearest up-sampling by 2^k (F.interpolate(mode='nearest') with integer factor) and its adjoint
__global__ void upsample_fwd_k(const float* __restrict__ x, float* __restrict__ y, long NC, int H, int W, int k) {
  const unsigned OH = H << k, OW = W << k;
  const unsigned i = blockIdx.x * TPB + threadIdx.x;        // (32-bit index arithmetic: tensors are bounded below 2^31 elements)
  if (i >= (unsigned)(NC * OH * OW)) return;
  const unsigned ow = i % OW, t = i / OW, oh = t % OH, nc = t / OH;
  y[i] = x[(long)nc * H * W + (long)(oh >> k) * W + (ow >> k)];
}
__global__ void upsample_bwd_k(const float* __restrict__ dy, float* __restrict__ dx, long NC, int H, int W, int k, int accumulate) {
  const int OH = H << k, OW = W << k, f = 1 << k;
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  if (i >= (unsigned)(NC * H * W)) return;
  const unsigned w = i % W, t = i / W, h = t % H, nc = t / H;
  const float* p = dy + (long)nc * OH * OW + (long)(h << k) * OW + (w << k);
  float s = 0.f;
  for (int a = 0; a < f; ++a)
    for (int b = 0; b < f; ++b) s += p[(long)a * OW + b];
  dx[i] = accumulate ? dx[i] + s : s;
}

// y[n,c] = scale * sum_hw (relu?)(x[n,c,hw]) ; one wave per row
__global__ __launch_bounds__(256) void sum_hw_fwd_k(const float* __restrict__ x, float* __restrict__ y, long NC, int HW, int in_relu, float scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= NC) return;
  float s = 0.f;
  for (int i = threadIdx.x & 63; i < HW; i += 64) {
    float v = x[row * HW + i];
    s += in_relu ? fmaxf(v, 0.f) : v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) y[row] = s * scale;
}
// The same for maps of 4 or 16 pixels (the discriminators' last maps: 2x2 / 4x4, up to 400 000 rows — a wave per row left 60 / 48 of
// its 64 lanes idle: 30 us per launch): one row per thread, read as 16-byte pieces, added in the ORDER of the wave reduction above
// (element i with element i + HW/2, then halving) — the same bits.
template <int HW>
__global__ __launch_bounds__(256) void sum_hw_small_k(const float* __restrict__ x, float* __restrict__ y, long NC, int in_relu, float scale) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= NC) return;
  float v[HW];
  const float4* const p = reinterpret_cast<const float4*>(x + row * HW);
#pragma unroll
  for (int q = 0; q < HW / 4; ++q) { const float4 t = p[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
  if (in_relu) {
#pragma unroll
    for (int i = 0; i < HW; ++i) v[i] = fmaxf(v[i], 0.f);
  }
#pragma unroll
  for (int o = HW / 2; o > 0; o >>= 1)
#pragma unroll
    for (int i = 0; i < o; ++i) v[i] += v[i + o];
  y[row] = v[0] * scale;
}
__global__ void sum_hw_bwd_k(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, long NC, int HW, int in_relu, float scale) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= NC * HW) return;
  float v = dy[i / HW] * scale;
  if (in_relu && !(x[i] > 0.f)) v = 0.f;
  dx[i] = v;
}

// out[c] = sum_{n,hw} x[n,c,hw]   (bias gradients): (C x S) partial blocks, then a fixed-order finish -> deterministic.
// HW % 4 == 0: 16-byte pieces of the channel rows; S == 1 (small tensors / many channels): one launch writes the result.
__device__ __forceinline__ double channel_range_sum(const float* __restrict__ x, int C, int HW, int c, long e0, long e1) {
  double s = 0.0;
  if ((HW & 3) == 0) {
    const int Q = HW >> 2;
    for (long g = (e0 >> 2) + threadIdx.x; g < (e1 >> 2); g += 256) {
      const int n = (int)(g / Q), q = (int)(g - (long)n * Q);
      const float4 v = *reinterpret_cast<const float4*>(x + ((long)n * C + c) * HW + 4 * q);
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    }
  } else {
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
      const long n = e / HW, hw = e - n * HW;
      s += x[(n * C + c) * HW + hw];
    }
  }
  return s;
}
__global__ __launch_bounds__(256) void channel_sum_partial(const float* __restrict__ x, double* __restrict__ part, float* __restrict__ out,
                                                           int N, int C, int HW, int S, int accumulate) {
  __shared__ double sc[4];
  const int c = blockIdx.x, sl = blockIdx.y;
  const long total = (long)N * HW;
  long chunk = (total + S - 1) / S;
  chunk = (chunk + 3) & ~3L;
  const long e0 = min(total, sl * chunk), e1 = min(total, e0 + chunk);
  double s = channel_range_sum(x, C, HW, c, e0, e1);
  s = block_sum_256(s, sc);
  if (threadIdx.x == 0) {
    if (S == 1) out[c] = accumulate ? out[c] + (float)s : (float)s;
    else part[(long)c * S + sl] = s;
  }
}
__global__ void channel_sum_final(const double* __restrict__ part, float* __restrict__ out, int C, int S, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int i = 0; i < S; ++i) s += part[(long)c * S + i];
  out[c] = accumulate ? out[c] + (float)s : (float)s;
}

// z = eps*exp(0.5*logvar) + mu  (models/generator_obj_att.py:418-420)
__global__ void reparam_fwd_k(const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ eps, float* __restrict__ z, long n) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) z[i] = eps[i] * expf(0.5f * logvar[i]) + mu[i];
}
__global__ void reparam_bwd_k(const float* __restrict__ dz, const float* __restrict__ logvar, const float* __restrict__ eps, float* __restrict__ dlogvar, long n) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i < n) dlogvar[i] = dz[i] * eps[i] * 0.5f * expf(0.5f * logvar[i]);
}

// torch.optim.Adam (no amsgrad / weight decay), single fused pass over a flat parameter arena
__global__ void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                       float omb1, float beta2, float omb2, float step_size, float bc2_sqrt, float eps, float grad_scale) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const float gr = g[i] * grad_scale;
  const float mi = m[i] + omb1 * (gr - m[i]);                  // exp_avg.lerp_(grad, 1-beta1)
  const float vi = v[i] * beta2 + omb2 * gr * gr;              // mul_(beta2).addcmul_(g, g, 1-beta2)
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;                 // (sqrt(v)/sqrt(bc2)).add_(eps)
  p[i] = p[i] - step_size * (mi / denom);                       // addcdiv_(m, denom, -step_size)
}

// Rank-1 layout tensor through the 1x1, pad-1 convolution c0 (generator_obj_att.py:489-494):
// y[o,c,i,j] = u[o,c] * mask[o,i-pad,j-pad] inside, 0 on the border; u = W_c0 . [obj_att ; z].
// The (O,128,R,R) tensor of the reference is never materialised.
__global__ void mask_outer_fwd_k(const float* __restrict__ u, const float* __restrict__ mask, float* __restrict__ y,
                                 long OC, int C, int R, int pad) {
  const int RP = R + 2 * pad;
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= OC * RP * RP) return;
  const int x = (int)(i % RP);
  long t = i / RP;
  const int yy = (int)(t % RP);
  const long oc = t / RP;
  const long o = oc / C;
  const int my = yy - pad, mx = x - pad;
  float v = 0.f;
  if ((unsigned)my < (unsigned)R && (unsigned)mx < (unsigned)R) v = u[oc] * mask[o * R * R + (long)my * R + mx];
  y[i] = v;
}
// du[o,c] = sum_{i,j} dy[o,c,i+pad,j+pad] * mask[o,i,j]   (one wave per (o,c))
__global__ __launch_bounds__(256) void mask_outer_bwd_k(const float* __restrict__ dy, const float* __restrict__ mask,
                                                        float* __restrict__ du, long OC, int C, int R, int pad) {
  const long oc = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (oc >= OC) return;
  const int RP = R + 2 * pad;
  const long o = oc / C;
  const float* d = dy + oc * RP * RP;
  const float* m = mask + o * R * R;
  float s = 0.f;
  for (int e = threadIdx.x & 63; e < R * R; e += 64) {
    const int i = e / R, j = e - i * R;
    s += d[(long)(i + pad) * RP + j + pad] * m[e];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) du[oc] = s;
}

// conv3x3(pad 1) followed by a 2x2 average pool == conv4x4(stride 2, pad 1) with
// w4[a][b] = 0.25 * sum_{dy,dx in {0,1}} w3[a-dy][b-dx]  (2.25x fewer MACs; same zero padding) — used for the second
// convolution of every down-sampling discriminator block (reference models/discriminator.py:46-51, 81-86).
__global__ void pool_fuse_w_fwd_k(const float* __restrict__ w3, float* __restrict__ w4, long n) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n * 16) return;
  const int b = (int)(i & 3), a = (int)((i >> 2) & 3);
  const long f = i >> 4;
  float s = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int kh = a - dy, kw = b - dx;
      if ((unsigned)kh < 3u && (unsigned)kw < 3u) s += w3[f * 9 + kh * 3 + kw];
    }
  w4[i] = 0.25f * s;
}
__global__ void pool_fuse_w_bwd_k(const float* __restrict__ dw4, float* __restrict__ dw3, long n) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n * 9) return;
  const long f = i / 9;
  const int r = (int)(i - f * 9), kh = r / 3, kw = r - kh * 3;
  float s = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) s += dw4[f * 16 + (kh + dy) * 4 + (kw + dx)];
  dw3[i] = 0.25f * s;
}

// out[n] = [a[n] ; b[n]] along the channel axis for rows of HW floats; b may be (N,Cb) broadcast over HW (bcast=1).
// Covers torch.cat((hidden, global.expand), 1) (generator_obj_att.py:549-552), cat((emb, attr), 1) (:589-590),
// cat((objs_att, z), 1) (:489) and contiguous weight concatenation.
__global__ void concat2_fwd_k(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long N, int Ca,
                              int Cb, int HW, int bcast) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  const long per = (long)(Ca + Cb) * HW;
  if (i >= N * per) return;
  const long n = i / per, r = i - n * per;
  const int c = (int)(r / HW), hw = (int)(r - (long)c * HW);
  out[i] = c < Ca ? a[(n * Ca + c) * HW + hw] : (bcast ? b[n * Cb + (c - Ca)] : b[(n * Cb + (c - Ca)) * HW + hw]);
}
// adjoint: da = d[:, :Ca], db = d[:, Ca:] (summed over HW when bcast)
__global__ void concat2_bwd_k(const float* __restrict__ d, float* __restrict__ da, float* __restrict__ db, long N, int Ca, int Cb,
                              int HW, int bcast) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  const long na = N * Ca * HW, nb = bcast ? N * Cb : N * Cb * HW;
  if (i < na) {
    if (!da) return;
    const long n = i / ((long)Ca * HW), r = i - n * (long)Ca * HW;
    da[i] = d[n * (long)(Ca + Cb) * HW + r];
  } else if (i < na + nb) {
    if (!db) return;
    const long j = i - na;
    if (bcast) {
      const long n = j / Cb; const int c = (int)(j - n * Cb);
      const float* p = d + (n * (Ca + Cb) + Ca + c) * HW;
      float s = 0.f;
      for (int k = 0; k < HW; ++k) s += p[k];
      db[j] = s;
    } else {
      const long n = j / ((long)Cb * HW), r = j - n * (long)Cb * HW;
      db[j] = d[(n * (Ca + Cb) + Ca) * HW + r];
    }
  }
}

// dtable[v][c] += sum_{n: rows[n]==v} dout[n][c]  — embedding backward, one owner per cell, fixed order
__global__ __launch_bounds__(256) void embedding_bwd_k(const float* __restrict__ dout, const long long* __restrict__ rows,
                                                       float* __restrict__ dtable, int N, int D) {
  __shared__ int lab[1024];
  const int v = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  float g = 0.f;
  for (int n0 = 0; n0 < N; n0 += 1024) {
    const int cnt = min(1024, N - n0);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += 256) lab[i] = (int)rows[n0 + i];
    __syncthreads();
    if (c < D)
      for (int i = 0; i < cnt; ++i)
        if (lab[i] == v) g += dout[(long)(n0 + i) * D + c];
  }
  if (c < D) dtable[(long)v * D + c] += g;
}

// Attribute estimate of the training loop (reference train64.py:156-166): rows without any annotated attribute get
// the arg-max of the attribute discriminator's logits as their single estimated attribute; annotated rows keep
// their annotation.  One wave per object; ties resolve to the lowest index like torch.argmax.
__global__ __launch_bounds__(256) void attr_estimate_k(const float* __restrict__ logits, const float* __restrict__ attr,
                                                       float* __restrict__ est, int O, int A) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= O) return;
  float best = -INFINITY, any = 0.f;
  int arg = 0x7fffffff;
  for (int a = lane; a < A; a += 64) {
    const float v = logits[(long)row * A + a];
    if (v > best) { best = v; arg = a; }          // strictly greater: keeps the lowest index within the lane
    any += attr[(long)row * A + a];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const int oa = __shfl_xor(arg, o);
    if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    any += __shfl_xor(any, o);
  }
  for (int a = lane; a < A; a += 64) {
    const float t = attr[(long)row * A + a];
    est[(long)row * A + a] = any != 0.f ? t : (a == arg ? 1.f : 0.f);
  }
}

// Box rasteriser of the dataset code (reference data/vg_custom_mask.py:136,158): mask[o,0,round(y0 R):round(y1 R),
// round(x0 R):round(x1 R)] = 1 with python round() (double arithmetic, half-to-even) — SURVEY.md §8f N3: only
// boxes need to cross PCIe, the (O,1,R,R) masks are built in HBM.
__global__ void rasterize_boxes_k(const float* __restrict__ boxes, float* __restrict__ masks, int O, int R) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= (long)O * R * R) return;
  const int x = (int)(i % R);
  long t = i / R;
  const int y = (int)(t % R);
  const int o = (int)(t / R);
  const float* b = boxes + 4 * o;
  const long x0 = (long)rint((double)b[0] * R), y0 = (long)rint((double)b[1] * R);
  const long x1 = (long)rint((double)b[2] * R), y1 = (long)rint((double)b[3] * R);
  // python slice semantics: negative bounds count from the end, everything is clipped to [0, R]
  auto clip = [R](long v) { if (v < 0) v += R; return v < 0 ? 0L : (v > R ? (long)R : v); };
  const long cx0 = clip(x0), cx1 = clip(x1), cy0 = clip(y0), cy1 = clip(y1);
  masks[i] = (x >= cx0 && x < cx1 && y >= cy0 && y < cy1) ? 1.f : 0.f;
}

// Per-object layout tensors of the VG batch builder from the boxes alone (data/vg_custom_mask.py:136-158), N3: the box mask,
// the shifted box (moved by 0.8 x the larger horizontal border distance when the box is narrower than half the image) and the
// shifted mask.  The reference computes with python floats (doubles) and python round (half to even): same here.  One workgroup
// column per object plane; thread -> pixel.
__global__ void layout_from_boxes_k(const float* __restrict__ boxes, float* __restrict__ boxes_shift, float* __restrict__ masks,
                                    float* __restrict__ masks_shift, int O, int R) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= (long)O * R * R) return;
  const int x = (int)(i % R);
  long t = i / R;
  const int y = (int)(t % R);
  const int o = (int)(t / R);
  const float* b = boxes + 4 * o;
  const double x0 = b[0], y0 = b[1], x1 = b[2], y1 = b[3];
  double sx0 = x0, sx1 = x1;
  if (x1 - x0 < 0.5) {                                   // :144
    const double left = x0, right = 1.0 - x1;
    if (left > right) { const double sh = left * 0.8; sx0 = x0 - sh; sx1 = x1 - sh; }          // :147-151
    else if (right > left) { const double sh = right * 0.8; sx0 = x0 + sh; sx1 = x1 + sh; }    // :152-156
  }
  auto clip = [R](long v) { if (v < 0) v += R; return v < 0 ? 0L : (v > R ? (long)R : v); };    // python slice semantics
  const long cy0 = clip((long)rint(y0 * R)), cy1 = clip((long)rint(y1 * R));
  const bool iny = y >= cy0 && y < cy1;
  masks[i] = (iny && x >= clip((long)rint(x0 * R)) && x < clip((long)rint(x1 * R))) ? 1.f : 0.f;                 // :136
  masks_shift[i] = (iny && x >= clip((long)rint(sx0 * R)) && x < clip((long)rint(sx1 * R))) ? 1.f : 0.f;       // :157
  if (x == 0 && y == 0) {                                                                                       // :158
    boxes_shift[4 * o + 0] = (float)sx0; boxes_shift[4 * o + 1] = (float)y0;
    boxes_shift[4 * o + 2] = (float)sx1; boxes_shift[4 * o + 3] = (float)y1;
  }
}

// ---- inference-side attribute logic (test64.py:114-198), N2
// rows of `attribute` (O, A): clear the listed columns and set column tgt (test64.py:160-167 attribute modification)
__global__ void attr_edit_k(float* __restrict__ attribute, const int* __restrict__ cols, int ncols, int tgt, int O, int A) {
  const int o = blockIdx.x * TPB + threadIdx.x;
  if (o >= O) return;
  for (int j = 0; j < ncols; ++j) attribute[(long)o * A + cols[j]] = 0.f;
  attribute[(long)o * A + tgt] = 1.f;
}
// out[o] = 1 if column tgt is among the k largest logits of row o (torch.topk(k) membership, test64.py:180-184): true iff fewer than
// k entries are strictly greater (ties resolved in favour of membership)
__global__ void topk_contains_k(const float* __restrict__ logits, unsigned char* __restrict__ out, int O, int A, int k, int tgt) {
  const int o = blockIdx.x * TPB + threadIdx.x;
  if (o >= O) return;
  const float* row = logits + (long)o * A;
  const float v = row[tgt];
  int greater = 0;
  for (int j = 0; j < A; ++j) greater += row[j] > v;
  out[o] = greater < k;
}
// pred[o][a] = sigmoid(logits[o][a]) > thr  (test64.py:143-150)
__global__ void sigmoid_threshold_k(const float* __restrict__ logits, unsigned char* __restrict__ pred, long n, float thr) {
  const long i = (long)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  pred[i] = 1.f / (1.f + expf(-logits[i])) > thr;
}

// ImageNet de-normalisation to bytes (data/utils.py:47-66 imagenet_deprocess_batch): per channel
// y = x / fp32(1/std_c) + fp32(mean_c) (two torchvision Normalize passes: sub 0, div 1/std; sub -mean, div 1),
// then per IMAGE (all channels) r = (y - lo) / (hi - lo) when rescaling, then byte(clamp(255 r, 0, 255)).
// One workgroup per image; the operation order is the reference's so the bytes are identical.
__global__ __launch_bounds__(256) void deprocess_u8_k(const float* __restrict__ x, unsigned char* __restrict__ out, int C, int HW,
                                                      int rescale, float is0, float is1, float is2, float m0, float m1, float m2) {
  __shared__ float s_lo[256], s_hi[256];
  const int n = blockIdx.x, tid = threadIdx.x;
  const long base = (long)n * C * HW;
  const int per = C * HW;
  auto denorm = [&](int e) {
    const int c = e / HW;
    const float is = c == 0 ? is0 : (c == 1 ? is1 : is2), mm = c == 0 ? m0 : (c == 1 ? m1 : m2);
    float y = (x[base + e] - 0.f) / is;
    return (y - (-mm)) / 1.f;
  };
  float lo = INFINITY, hi = -INFINITY;
  if (rescale) {
    for (int e = tid; e < per; e += 256) { const float y = denorm(e); lo = fminf(lo, y); hi = fmaxf(hi, y); }
    s_lo[tid] = lo; s_hi[tid] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) { s_lo[tid] = fminf(s_lo[tid], s_lo[tid + s]); s_hi[tid] = fmaxf(s_hi[tid], s_hi[tid + s]); }
      __syncthreads();
    }
    lo = s_lo[0]; hi = s_hi[0];
  }
  const float range = hi - lo;
  for (int e = tid; e < per; e += 256) {
    float y = denorm(e);
    if (rescale) y = (y - lo) / range;
    y = fminf(fmaxf(y * 255.f, 0.f), 255.f);
    out[base + e] = (unsigned char)(int)y;       // NaN (constant image, 0/0) -> 0 like the host cast
  }
}

// 2x2 box filter at stride 1 over the zero-extended map: xb[j][i] = (x[j-1][i-1] + x[j-1][i] + x[j][i-1] + x[j][i]) / 4 for
// j, i in [0, H] x [0, W] (out-of-range x = 0).  avg_pool2(conv3x3(x, pad 1)) equals a 3x3 STRIDE-2 convolution without
// padding of xb (agl.functional.conv3x3_avgpool2): 9 taps per output instead of the 16 of the fused 4x4 form.
__global__ void box2_fwd_any_k(const float* __restrict__ x, float* __restrict__ xb, long NC, int H, int W) {      // any width
  const unsigned HB = H + 1, WB = W + 1;
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  if (i >= (unsigned)(NC * HB * WB)) return;
  const unsigned ix = i % WB, t = i / WB, iy = t % HB, nc = t / HB;
  const float* p = x + (long)nc * H * W;
  float s = 0.f;
  if (iy > 0 && ix > 0) s += p[(iy - 1) * W + ix - 1];
  if (iy > 0 && ix < (unsigned)W) s += p[(iy - 1) * W + ix];
  if (iy < (unsigned)H && ix > 0) s += p[iy * W + ix - 1];
  if (iy < (unsigned)H && ix < (unsigned)W) s += p[iy * W + ix];
  xb[i] = 0.25f * s;
}
// One thread per four consecutive outputs of a row (W % 4 == 0): two 16-byte loads + two scalars in, four coalesced stores out
// (the one-output-per-thread form ran at 2.4 TB/s: 65-float rows, four scalar loads per store).  Same addition order as before:
// ((top-left + top-right) + bottom-left) + bottom-right.
template <typename OutT>       // float, or __bf16 (round to nearest even: the rounding the bf16-mode convolution applies when it stages a tensor)
__global__ void box2_fwd_k(const float* __restrict__ x, OutT* __restrict__ xb, long NC, int H, int W) {
  const unsigned HB = H + 1, WB = W + 1, G = W / 4 + 1;
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  if (i >= (unsigned)(NC * HB * G)) return;
  const unsigned g = i % G, t = i / G, iy = t % HB, nc = t / HB;
  const unsigned ix0 = 4 * g;
  const float* p = x + (long)nc * H * W;
  float top[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, bot[5] = {0.f, 0.f, 0.f, 0.f, 0.f};      // columns ix0 - 1 .. ix0 + 3
  if (iy > 0) {
    const float* r = p + (long)(iy - 1) * W + ix0;
    if (ix0 > 0) top[0] = r[-1];
    if (ix0 < (unsigned)W) { const float4 v = *reinterpret_cast<const float4*>(r); top[1] = v.x; top[2] = v.y; top[3] = v.z; top[4] = v.w; }
  }
  if (iy < (unsigned)H) {
    const float* r = p + (long)iy * W + ix0;
    if (ix0 > 0) bot[0] = r[-1];
    if (ix0 < (unsigned)W) { const float4 v = *reinterpret_cast<const float4*>(r); bot[1] = v.x; bot[2] = v.y; bot[3] = v.z; bot[4] = v.w; }
  }
  OutT* o = xb + ((long)nc * HB + iy) * WB + ix0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (ix0 + k < WB) o[k] = (OutT)(0.25f * (((top[k] + top[k + 1]) + bot[k]) + bot[k + 1]));
}
// dx[j][i] = (dxb[j][i] + dxb[j][i+1] + dxb[j+1][i] + dxb[j+1][i+1]) / 4, optionally masked by mask > 0 (ReLU backward)
__global__ void box2_bwd_k(const float* __restrict__ dxb, const float* __restrict__ mask, float* __restrict__ dx, long NC, int H, int W) {
  const unsigned WB = W + 1;
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  if (i >= (unsigned)(NC * H * W)) return;
  const unsigned ix = i % W, t = i / W, iy = t % H, nc = t / H;
  const float* p = dxb + (long)nc * (H + 1) * WB + (long)iy * WB + ix;
  float v = 0.25f * (p[0] + p[1] + p[WB] + p[WB + 1]);
  if (mask && !(mask[i] > 0.f)) v = 0.f;
  dx[i] = v;
}

// y[nc][Y][X] = x[nc][my[Y]][mx[X]] for monotone index maps (nearest up-sampling by any factor, and the block-class
// grids of agl.generator.SPADE).  Backward: dx[nc][i][j] = sum of dy over the preimage rectangle [ylo[i], ylo[i+1]) x
// [xlo[j], xlo[j+1]) — every cell has one owner and a fixed order (deterministic, no atomics).
__global__ void grid_gather_fwd_k(const float* __restrict__ x, const int* __restrict__ my, const int* __restrict__ mx, float* __restrict__ y,
                                  long NC, int h, int w, int H, int W) {
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  if (i >= (unsigned)(NC * H * W)) return;
  const unsigned X = i % W, t = i / W, Y = t % H, nc = t / H;
  y[i] = x[((long)nc * h + my[Y]) * w + mx[X]];
}
__global__ void grid_gather_bwd_k(const float* __restrict__ dy, const int* __restrict__ ylo, const int* __restrict__ xlo, float* __restrict__ dx,
                                  long NC, int h, int w, int H, int W) {
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  if (i >= (unsigned)(NC * h * w)) return;
  const unsigned xj = i % w, t = i / w, yi = t % h, nc = t / h;
  const float* p = dy + (long)nc * H * W;
  float s = 0.f;
  for (int Y = ylo[yi]; Y < ylo[yi + 1]; ++Y)
    for (int X = xlo[xj]; X < xlo[xj + 1]; ++X) s += p[(long)Y * W + X];
  dx[i] = s;
}

}  // namespace

#define LAUNCH1D(kernel, n, ...)                                                                   \
  hipLaunchKernelGGL(kernel, dim3(nblocks(n)), dim3(TPB), 0, (hipStream_t)stream, __VA_ARGS__)

extern "C" {

int agl_crop_fwd(const float* feats, const float* boxes, const long long* box_to_img, float* out, int N, int B, int C, int H,
                 int W, int HH, int WW, int align_corners, void* stream) {
  AGL_REQUIRE(feats && boxes && box_to_img && out, "agl_crop_fwd: null pointer");
  AGL_REQUIRE(N > 0 && B >= 0 && C > 0 && H > 0 && W > 0 && HH > 0 && WW > 0, "agl_crop_fwd: bad extent");
  if (B == 0) return AGL_OK;
  LAUNCH1D(crop_fwd, (long)B * C * HH * WW, feats, boxes, box_to_img, out, N, B, C, H, W, HH, WW, align_corners);
  AGL_CHECK_LAUNCH("agl_crop_fwd");
  return AGL_OK;
}

// dfeats must be zero-filled (or hold a gradient to accumulate into) by the caller.
int agl_crop_bwd(const float* dout, const float* boxes, const long long* box_to_img, float* dfeats, int N, int B, int C, int H,
                 int W, int HH, int WW, int align_corners, void* stream) {
  AGL_REQUIRE(dout && boxes && box_to_img && dfeats, "agl_crop_bwd: null pointer");
  AGL_REQUIRE(N > 0 && B >= 0 && C > 0 && H > 0 && W > 0 && HH > 0 && WW > 0, "agl_crop_bwd: bad extent");
  if (B == 0) return AGL_OK;
  LAUNCH1D(crop_bwd, (long)B * C * HH * WW, dout, boxes, box_to_img, dfeats, N, B, C, H, W, HH, WW, align_corners);
  AGL_CHECK_LAUNCH("agl_crop_bwd");
  return AGL_OK;
}

// The same gradient for a NON-DECREASING box_to_img (boxes of an image contiguous, as models/bilinear.py:77-90 and the training
// loop build them): a gather in fixed order, no atomics — bit-reproducible.  The caller guarantees the order (checked on the host
// where the map lives on the CPU); boxes whose image index is outside [0, N) contribute nothing.
int agl_crop_bwd_sorted(const float* dout, const float* boxes, const long long* box_to_img, float* dfeats, int N, int B, int C, int H,
                        int W, int HH, int WW, int align_corners, void* stream) {
  AGL_REQUIRE(dout && boxes && box_to_img && dfeats, "agl_crop_bwd_sorted: null pointer");
  AGL_REQUIRE(N > 0 && B >= 0 && C > 0 && H > 0 && W > 0 && HH > 0 && WW > 0, "agl_crop_bwd_sorted: bad extent");
  if (B == 0) return AGL_OK;
  LAUNCH1D(crop_bwd_sorted, (long)N * C * H * W, dout, boxes, box_to_img, dfeats, N, B, C, H, W, HH, WW, align_corners);
  AGL_CHECK_LAUNCH("agl_crop_bwd_sorted");
  return AGL_OK;
}

int agl_lstm_gates_fwd_sum(const float* ccx, const long long* rows, const float* cch, int cch_splits, long long cch_stride,
                           const float* c_prev, float* h, float* c, float* gates, int B, int hid, int S, void* stream) {
  AGL_REQUIRE(ccx && h && c && gates && B > 0 && hid > 0 && S > 0, "agl_lstm_gates_fwd: bad argument");
  AGL_REQUIRE(cch_splits <= 1 || (cch && cch_stride >= (long long)B * 4 * hid * S), "agl_lstm_gates_fwd: slabs shorter than the gate tensor");
  LAUNCH1D(lstm_gates_fwd, (long)B * hid * S, ccx, rows, cch, c_prev, h, c, gates, B, hid, S, cch_splits, (long)cch_stride);
  AGL_CHECK_LAUNCH("agl_lstm_gates_fwd");
  return AGL_OK;
}

int agl_lstm_gates_fwd(const float* ccx, const long long* rows, const float* cch, const float* c_prev, float* h, float* c,
                       float* gates, int B, int hid, int S, void* stream) {
  return agl_lstm_gates_fwd_sum(ccx, rows, cch, 1, 0, c_prev, h, c, gates, B, hid, S, stream);
}

int agl_lstm_gates_bwd_sum(const float* dh_a, const float* dh_b, int dh_b_splits, long long dh_b_stride, int Bb, const float* dc_next, int Bc,
                           const float* gates, const float* c_prev, const float* c, float* dcc, float* dc_prev, int B, int hid, int S,
                           void* stream) {
  AGL_REQUIRE(gates && c && dcc && dc_prev && B > 0 && hid > 0 && S > 0, "agl_lstm_gates_bwd: bad argument");
  AGL_REQUIRE(Bb <= B && Bc <= B, "agl_lstm_gates_bwd: prefix larger than batch");
  AGL_REQUIRE(dh_b_splits <= 1 || (dh_b && dh_b_stride >= (long long)Bb * hid * S), "agl_lstm_gates_bwd: slabs shorter than the prefix");
  LAUNCH1D(lstm_gates_bwd, (long)B * hid * S, dh_a, dh_b, Bb, dc_next, Bc, gates, c_prev, c, dcc, dc_prev, B, hid, S, dh_b_splits,
           (long)dh_b_stride);
  AGL_CHECK_LAUNCH("agl_lstm_gates_bwd");
  return AGL_OK;
}

int agl_lstm_gates_bwd(const float* dh_a, const float* dh_b, int Bb, const float* dc_next, int Bc, const float* gates,
                       const float* c_prev, const float* c, float* dcc, float* dc_prev, int B, int hid, int S, void* stream) {
  return agl_lstm_gates_bwd_sum(dh_a, dh_b, 1, 0, Bb, dc_next, Bc, gates, c_prev, c, dcc, dc_prev, B, hid, S, stream);
}

int agl_relu_bwd(const float* dy, const float* y, float* dx, long n, void* stream) {
  AGL_REQUIRE(dy && y && dx && n >= 0, "agl_relu_bwd: bad argument");
  if (n == 0) return AGL_OK;
  LAUNCH1D(relu_bwd_k, n, dy, y, dx, n);
  AGL_CHECK_LAUNCH("agl_relu_bwd");
  return AGL_OK;
}

// out = alpha*a + beta*b (b may be NULL)
int agl_axpby(const float* a, const float* b, float alpha, float beta, float* out, long n, void* stream) {
  AGL_REQUIRE(a && out && n >= 0, "agl_axpby: bad argument");
  if (n == 0) return AGL_OK;
  LAUNCH1D(axpby_k, n, a, b, alpha, beta, out, n);
  AGL_CHECK_LAUNCH("agl_axpby");
  return AGL_OK;
}

int agl_gather_rows(const float* src, const long long* rows, float* out, long R, long len, int accumulate, void* stream) {
  AGL_REQUIRE(src && rows && out && R >= 0 && len > 0, "agl_gather_rows: bad argument");
  if (R == 0) return AGL_OK;
  LAUNCH1D(gather_rows_k, R * len, src, rows, out, R, len, accumulate);
  AGL_CHECK_LAUNCH("agl_gather_rows");
  return AGL_OK;
}

int agl_scatter_rows(const float* src, const long long* rows, float* out, long R, long len, void* stream) {
  AGL_REQUIRE(src && rows && out && R >= 0 && len > 0, "agl_scatter_rows: bad argument");
  if (R == 0) return AGL_OK;
  LAUNCH1D(scatter_rows_k, R * len, src, rows, out, R, len);
  AGL_CHECK_LAUNCH("agl_scatter_rows");
  return AGL_OK;
}

int agl_avgpool2_fwd(const float* x, float* y, long NC, int H, int W, int in_relu, void* stream) {
  AGL_REQUIRE(x && y && NC > 0 && H >= 2 && W >= 2 && NC * H * W < (1L << 31), "agl_avgpool2_fwd: bad argument");
  if (W % 8 == 0 && H % 2 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0)
    LAUNCH1D(avgpool2_fwd_w_k<false>, NC * (H / 2) * (W / 8), (const void*)x, y, NC, H, W, in_relu);
  else
    LAUNCH1D(avgpool2_fwd_k, NC * (H / 2) * (W / 2), x, y, NC, H, W, in_relu);
  AGL_CHECK_LAUNCH("agl_avgpool2_fwd");
  return AGL_OK;
}

// The same for an input stored as bf16 (the output of a discriminator block that only convolutions and this pool read): W % 8 == 0
int agl_avgpool2_fwd_x16(const void* x_bf16, float* y, long NC, int H, int W, int in_relu, void* stream) {
  AGL_REQUIRE(x_bf16 && y && NC > 0 && H >= 2 && H % 2 == 0 && W >= 8 && W % 8 == 0 && NC * H * W < (1L << 31) &&
              (((uintptr_t)x_bf16 | (uintptr_t)y) & 15) == 0, "agl_avgpool2_fwd_x16: bad argument (W % 8 == 0, even H, 16-byte aligned)");
  LAUNCH1D(avgpool2_fwd_w_k<true>, NC * (H / 2) * (W / 8), x_bf16, y, NC, H, W, in_relu);
  AGL_CHECK_LAUNCH("agl_avgpool2_fwd_x16");
  return AGL_OK;
}
// Channel-blocked bf16 tensors (include/agl.h): NCHW fp32 (x_bf16 = 0) or bf16 (1) -> [N][C/8][H][W][8] bf16
int agl_to_blocked(const void* x, void* y_blk, int N, int C, int H, int W, int x_bf16, void* stream) {
  AGL_REQUIRE(x && y_blk && N > 0 && C > 0 && C % 8 == 0 && H > 0 && W > 0 && (long)N * C * H * W < (1L << 31) && ((uintptr_t)y_blk & 15) == 0,
              "agl_to_blocked: bad argument (C % 8 == 0, 16-byte aligned output)");
  const long NG = (long)N * (C / 8);
  if (x_bf16) LAUNCH1D(to_blocked_k<true>, NG * H * W, x, (uint4*)y_blk, NG, H * W);
  else LAUNCH1D(to_blocked_k<false>, NG * H * W, x, (uint4*)y_blk, NG, H * W);
  AGL_CHECK_LAUNCH("agl_to_blocked");
  return AGL_OK;
}
// y (fp32 NCHW, N x C x H/2 x W/2) = avg_pool2(relu?(x)) of a channel-blocked bf16 x.  W % 8 == 0, even H, C % 8 == 0.
int agl_avgpool2_fwd_xblk(const void* x_blk, float* y, int N, int C, int H, int W, int in_relu, void* stream) {
  AGL_REQUIRE(x_blk && y && N > 0 && C % 8 == 0 && H >= 2 && H % 2 == 0 && W >= 8 && W % 8 == 0 && (long)N * C * H * W < (1L << 31) &&
              (((uintptr_t)x_blk | (uintptr_t)y) & 15) == 0, "agl_avgpool2_fwd_xblk: bad argument (C % 8 == 0, W % 8 == 0, even H, 16-byte aligned)");
  const long NG = (long)N * (C / 8);
  LAUNCH1D(avgpool2_fwd_blk_k, NG * (H / 2) * (W / 2), (const uint4*)x_blk, y, NG, H, W, in_relu);
  AGL_CHECK_LAUNCH("agl_avgpool2_fwd_xblk");
  return AGL_OK;
}
// Backward of avg_pool2(relu(x)) with the mask read from a channel-blocked bf16 x: dx (fp32 NCHW) (+)= 0.25 * dy where x > 0
int agl_avgpool2_bwd_xblk(const float* dy, const void* x_blk, float* dx, int N, int C, int H, int W, int accumulate, void* stream) {
  AGL_REQUIRE(dy && x_blk && dx && N > 0 && C % 8 == 0 && H >= 2 && H % 2 == 0 && W % 4 == 0 && (long)N * C * H * W < (1L << 31),
              "agl_avgpool2_bwd_xblk: bad argument");
  const long NG = (long)N * (C / 8);
  LAUNCH1D(avgpool2_bwd_mblk_k, NG * H * (W / 4), dy, (const uint4*)x_blk, dx, NG, H, W, accumulate);
  AGL_CHECK_LAUNCH("agl_avgpool2_bwd_xblk");
  return AGL_OK;
}
// Backward of avg_pool2(relu(x)) with x stored as bf16 (read for the mask only): dx (+)= 0.25 * dy where x > 0.  W % 4 == 0, even H.
int agl_avgpool2_bwd_x16(const float* dy, const void* x_bf16, float* dx, long NC, int H, int W, int accumulate, void* stream) {
  AGL_REQUIRE(dy && x_bf16 && dx && NC > 0 && H >= 2 && H % 2 == 0 && W % 4 == 0 && NC * H * W < (1L << 31), "agl_avgpool2_bwd_x16: bad argument");
  LAUNCH1D(avgpool2_bwd4_m16_k, NC * H * (W / 4), dy, (const unsigned short*)x_bf16, dx, NC, H, W, accumulate);
  AGL_CHECK_LAUNCH("agl_avgpool2_bwd_x16");
  return AGL_OK;
}

int agl_avgpool2_bwd(const float* dy, const float* x, float* dx, long NC, int H, int W, int in_relu, int accumulate, void* stream) {
  AGL_REQUIRE(dy && dx && (!in_relu || x) && NC > 0 && H >= 2 && W >= 2 && NC * H * W < (1L << 31), "agl_avgpool2_bwd: bad argument");
  if (W % 4 == 0 && H % 2 == 0) LAUNCH1D(avgpool2_bwd4_k, NC * H * (W / 4), dy, x, dx, NC, H, W, in_relu, accumulate);
  else LAUNCH1D(avgpool2_bwd_k, NC * H * W, dy, x, dx, NC, H, W, in_relu, accumulate);
  AGL_CHECK_LAUNCH("agl_avgpool2_bwd");
  return AGL_OK;
}

int agl_upsample_nearest_fwd(const float* x, float* y, long NC, int H, int W, int log2_factor, void* stream) {
  AGL_REQUIRE(x && y && NC > 0 && H > 0 && W > 0 && log2_factor >= 0 && log2_factor <= 5 && NC * ((long)H << log2_factor) * ((long)W << log2_factor) < (1L << 31), "agl_upsample_nearest_fwd: bad argument");
  LAUNCH1D(upsample_fwd_k, NC * ((long)H << log2_factor) * ((long)W << log2_factor), x, y, NC, H, W, log2_factor);
  AGL_CHECK_LAUNCH("agl_upsample_nearest_fwd");
  return AGL_OK;
}

int agl_upsample_nearest_bwd(const float* dy, float* dx, long NC, int H, int W, int log2_factor, int accumulate, void* stream) {
  AGL_REQUIRE(dy && dx && NC > 0 && H > 0 && W > 0 && log2_factor >= 0 && log2_factor <= 5 && NC * ((long)H << log2_factor) * ((long)W << log2_factor) < (1L << 31), "agl_upsample_nearest_bwd: bad argument");
  LAUNCH1D(upsample_bwd_k, NC * H * W, dy, dx, NC, H, W, log2_factor, accumulate);
  AGL_CHECK_LAUNCH("agl_upsample_nearest_bwd");
  return AGL_OK;
}

int agl_sum_hw_fwd(const float* x, float* y, long NC, int HW, int in_relu, float scale, void* stream) {
  AGL_REQUIRE(x && y && NC > 0 && HW > 0, "agl_sum_hw_fwd: bad argument");
  if ((HW == 4 || HW == 16) && ((uintptr_t)x & 15) == 0) {
    if (HW == 4) hipLaunchKernelGGL(sum_hw_small_k<4>, dim3(agl_cdiv(NC, 256)), dim3(256), 0, (hipStream_t)stream, x, y, NC, in_relu, scale);
    else hipLaunchKernelGGL(sum_hw_small_k<16>, dim3(agl_cdiv(NC, 256)), dim3(256), 0, (hipStream_t)stream, x, y, NC, in_relu, scale);
  } else {
    hipLaunchKernelGGL(sum_hw_fwd_k, dim3(agl_cdiv(NC, 4)), dim3(256), 0, (hipStream_t)stream, x, y, NC, HW, in_relu, scale);
  }
  AGL_CHECK_LAUNCH("agl_sum_hw_fwd");
  return AGL_OK;
}

int agl_sum_hw_bwd(const float* dy, const float* x, float* dx, long NC, int HW, int in_relu, float scale, void* stream) {
  AGL_REQUIRE(dy && dx && (!in_relu || x) && NC > 0 && HW > 0, "agl_sum_hw_bwd: bad argument");
  LAUNCH1D(sum_hw_bwd_k, NC * HW, dy, x, dx, NC, HW, in_relu, scale);
  AGL_CHECK_LAUNCH("agl_sum_hw_bwd");
  return AGL_OK;
}

long agl_channel_sum_ws_bytes(int C) { return (long)C * 64 * sizeof(double); }

int agl_channel_sum(const float* x, float* out, int N, int C, int HW, int accumulate, void* ws, long ws_bytes, void* stream) {
  AGL_REQUIRE(x && out && N > 0 && C > 0 && HW > 0, "agl_channel_sum: bad argument");
  const long total = (long)N * HW;
  long S = (1024 + C - 1) / C;                              // C x S >= ~1024 workgroups, at least 2048 elements each
  if (S > total / 2048) S = total / 2048;
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  if (S > 1 && (!ws || ws_bytes < (long)C * S * (long)sizeof(double))) {
    agl_set_error("agl_channel_sum: workspace too small");
    return AGL_ERR_WORKSPACE;
  }
  hipLaunchKernelGGL(channel_sum_partial, dim3(C, (unsigned)S), dim3(256), 0, (hipStream_t)stream, x, (double*)ws, out, N, C, HW, (int)S, accumulate);
  AGL_CHECK_LAUNCH("agl_channel_sum(partial)");
  if (S > 1) {
    hipLaunchKernelGGL(channel_sum_final, dim3(agl_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, (const double*)ws, out, C, (int)S, accumulate);
    AGL_CHECK_LAUNCH("agl_channel_sum(final)");
  }
  return AGL_OK;
}

int agl_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, long n, void* stream) {
  AGL_REQUIRE(mu && logvar && eps && z && n > 0, "agl_reparam_fwd: bad argument");
  LAUNCH1D(reparam_fwd_k, n, mu, logvar, eps, z, n);
  AGL_CHECK_LAUNCH("agl_reparam_fwd");
  return AGL_OK;
}

int agl_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dlogvar, long n, void* stream) {
  AGL_REQUIRE(dz && logvar && eps && dlogvar && n > 0, "agl_reparam_bwd: bad argument");
  LAUNCH1D(reparam_bwd_k, n, dz, logvar, eps, dlogvar, n);
  AGL_CHECK_LAUNCH("agl_reparam_bwd");
  return AGL_OK;
}

// One Adam step over a flat arena of n floats; `step` is the 1-based step count.  grad_scale multiplies the
// gradient first (1/world_size after a sum all-reduce).
int agl_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2, double eps,
                  int step, float grad_scale, void* stream) {
  AGL_REQUIRE(p && g && m && v && n > 0 && step >= 1, "agl_adam_step: bad argument");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  // 1-beta is formed in double like torch's python scalars (1.f - 0.999f is off by 1.3e-5 relative)
  LAUNCH1D(adam_k, n, p, g, m, v, n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size, bc2_sqrt, (float)eps, grad_scale);
  AGL_CHECK_LAUNCH("agl_adam_step");
  return AGL_OK;
}

int agl_mask_outer_fwd(const float* u, const float* mask, float* y, int O, int C, int R, int pad, void* stream) {
  AGL_REQUIRE(u && mask && y && O > 0 && C > 0 && R > 0 && pad >= 0, "agl_mask_outer_fwd: bad argument");
  const long RP = R + 2 * pad;
  AGL_REQUIRE((long)O * C * RP * RP < (1L << 31), "agl_mask_outer_fwd: tensor too large");
  LAUNCH1D(mask_outer_fwd_k, (long)O * C * RP * RP, u, mask, y, (long)O * C, C, R, pad);
  AGL_CHECK_LAUNCH("agl_mask_outer_fwd");
  return AGL_OK;
}

int agl_mask_outer_bwd(const float* dy, const float* mask, float* du, int O, int C, int R, int pad, void* stream) {
  AGL_REQUIRE(dy && mask && du && O > 0 && C > 0 && R > 0 && pad >= 0, "agl_mask_outer_bwd: bad argument");
  hipLaunchKernelGGL(mask_outer_bwd_k, dim3(agl_cdiv((long)O * C, 4)), dim3(256), 0, (hipStream_t)stream, dy, mask, du,
                     (long)O * C, C, R, pad);
  AGL_CHECK_LAUNCH("agl_mask_outer_bwd");
  return AGL_OK;
}

int agl_pool_fuse_weight_fwd(const float* w3, float* w4, long n_filters, void* stream) {
  AGL_REQUIRE(w3 && w4 && n_filters > 0, "agl_pool_fuse_weight_fwd: bad argument");
  LAUNCH1D(pool_fuse_w_fwd_k, n_filters * 16, w3, w4, n_filters);
  AGL_CHECK_LAUNCH("agl_pool_fuse_weight_fwd");
  return AGL_OK;
}

int agl_pool_fuse_weight_bwd(const float* dw4, float* dw3, long n_filters, void* stream) {
  AGL_REQUIRE(dw4 && dw3 && n_filters > 0, "agl_pool_fuse_weight_bwd: bad argument");
  LAUNCH1D(pool_fuse_w_bwd_k, n_filters * 9, dw4, dw3, n_filters);
  AGL_CHECK_LAUNCH("agl_pool_fuse_weight_bwd");
  return AGL_OK;
}

int agl_concat2_fwd(const float* a, const float* b, float* out, long N, int Ca, int Cb, int HW, int bcast, void* stream) {
  AGL_REQUIRE(a && b && out && N > 0 && Ca > 0 && Cb > 0 && HW > 0, "agl_concat2_fwd: bad argument");
  LAUNCH1D(concat2_fwd_k, N * (long)(Ca + Cb) * HW, a, b, out, N, Ca, Cb, HW, bcast);
  AGL_CHECK_LAUNCH("agl_concat2_fwd");
  return AGL_OK;
}

int agl_concat2_bwd(const float* d, float* da, float* db, long N, int Ca, int Cb, int HW, int bcast, void* stream) {
  AGL_REQUIRE(d && (da || db) && N > 0 && Ca > 0 && Cb > 0 && HW > 0, "agl_concat2_bwd: bad argument");
  LAUNCH1D(concat2_bwd_k, N * (long)Ca * HW + (bcast ? N * (long)Cb : N * (long)Cb * HW), d, da, db, N, Ca, Cb, HW, bcast);
  AGL_CHECK_LAUNCH("agl_concat2_bwd");
  return AGL_OK;
}

int agl_embedding_bwd(const float* dout, const long long* rows, float* dtable, int N, int D, int V, void* stream) {
  AGL_REQUIRE(dout && rows && dtable && N > 0 && D > 0 && V > 0, "agl_embedding_bwd: bad argument");
  hipLaunchKernelGGL(embedding_bwd_k, dim3(agl_cdiv(D, 256), V), dim3(256), 0, (hipStream_t)stream, dout, rows, dtable, N, D);
  AGL_CHECK_LAUNCH("agl_embedding_bwd");
  return AGL_OK;
}

int agl_attr_estimate(const float* logits, const float* attribute, float* attribute_est, int O, int A, void* stream) {
  AGL_REQUIRE(logits && attribute && attribute_est && O > 0 && A > 0, "agl_attr_estimate: bad argument");
  hipLaunchKernelGGL(attr_estimate_k, dim3(agl_cdiv(O, 4)), dim3(256), 0, (hipStream_t)stream, logits, attribute, attribute_est, O, A);
  AGL_CHECK_LAUNCH("agl_attr_estimate");
  return AGL_OK;
}

int agl_rasterize_boxes(const float* boxes, float* masks, int O, int R, void* stream) {
  AGL_REQUIRE(boxes && masks && O > 0 && R > 0, "agl_rasterize_boxes: bad argument");
  LAUNCH1D(rasterize_boxes_k, (long)O * R * R, boxes, masks, O, R);
  AGL_CHECK_LAUNCH("agl_rasterize_boxes");
  return AGL_OK;
}

int agl_layout_from_boxes(const float* boxes, float* boxes_shift, float* masks, float* masks_shift, int O, int R, void* stream) {
  AGL_REQUIRE(boxes && boxes_shift && masks && masks_shift && O > 0 && R > 0, "agl_layout_from_boxes: bad argument");
  LAUNCH1D(layout_from_boxes_k, (long)O * R * R, boxes, boxes_shift, masks, masks_shift, O, R);
  AGL_CHECK_LAUNCH("agl_layout_from_boxes");
  return AGL_OK;
}

int agl_attr_edit(float* attribute, const int* cols_dev, int ncols, int tgt, int O, int A, void* stream) {
  AGL_REQUIRE(attribute && (cols_dev || ncols == 0) && O > 0 && A > 0 && tgt >= 0 && tgt < A && ncols >= 0, "agl_attr_edit: bad argument");
  LAUNCH1D(attr_edit_k, (long)O, attribute, cols_dev, ncols, tgt, O, A);
  AGL_CHECK_LAUNCH("agl_attr_edit");
  return AGL_OK;
}

int agl_topk_contains(const float* logits, unsigned char* out, int O, int A, int k, int tgt, void* stream) {
  AGL_REQUIRE(logits && out && O > 0 && A > 0 && k > 0 && tgt >= 0 && tgt < A, "agl_topk_contains: bad argument");
  LAUNCH1D(topk_contains_k, (long)O, logits, out, O, A, k, tgt);
  AGL_CHECK_LAUNCH("agl_topk_contains");
  return AGL_OK;
}

int agl_sigmoid_threshold(const float* logits, unsigned char* pred, long n, float thr, void* stream) {
  AGL_REQUIRE(logits && pred && n > 0, "agl_sigmoid_threshold: bad argument");
  LAUNCH1D(sigmoid_threshold_k, n, logits, pred, n, thr);
  AGL_CHECK_LAUNCH("agl_sigmoid_threshold");
  return AGL_OK;
}

int agl_deprocess_u8(const float* x, unsigned char* out, int N, int C, int HW, int rescale, const float* inv_std, const float* mean,
                     void* stream) {
  AGL_REQUIRE(x && out && inv_std && mean && N > 0 && C == 3 && HW > 0, "agl_deprocess_u8: bad argument (C must be 3)");
  hipLaunchKernelGGL(deprocess_u8_k, dim3(N), dim3(256), 0, (hipStream_t)stream, x, out, C, HW, rescale, inv_std[0], inv_std[1],
                     inv_std[2], mean[0], mean[1], mean[2]);
  AGL_CHECK_LAUNCH("agl_deprocess_u8");
  return AGL_OK;
}

int agl_box2_fwd(const float* x, float* xb, long NC, int H, int W, void* stream) {
  AGL_REQUIRE(x && xb && NC > 0 && H > 0 && W > 0 && NC * (H + 1) * (W + 1) < (1L << 31), "agl_box2_fwd: bad argument");
  if (W % 4 == 0) LAUNCH1D(box2_fwd_k<float>, NC * (H + 1) * (W / 4 + 1), x, xb, NC, H, W);
  else LAUNCH1D(box2_fwd_any_k, NC * (H + 1) * (W + 1), x, xb, NC, H, W);
  AGL_CHECK_LAUNCH("agl_box2_fwd");
  return AGL_OK;
}

// The same box filter written as bf16 (W % 4 == 0): for a consumer that is a bf16-mode convolution reading it with AGL_CONV_X_BF16
// — the values are those the convolution would have rounded the fp32 tensor to when staging it, so the results are identical.
int agl_box2_fwd_bf16(const float* x, void* xb, long NC, int H, int W, void* stream) {
  AGL_REQUIRE(x && xb && NC > 0 && H > 0 && W > 0 && W % 4 == 0 && NC * (H + 1) * (W + 1) < (1L << 31), "agl_box2_fwd_bf16: bad argument");
  LAUNCH1D(box2_fwd_k<__bf16>, NC * (H + 1) * (W / 4 + 1), x, (__bf16*)xb, NC, H, W);
  AGL_CHECK_LAUNCH("agl_box2_fwd_bf16");
  return AGL_OK;
}

int agl_box2_bwd(const float* dxb, const float* mask, float* dx, long NC, int H, int W, void* stream) {
  AGL_REQUIRE(dxb && dx && NC > 0 && H > 0 && W > 0 && NC * (H + 1) * (W + 1) < (1L << 31), "agl_box2_bwd: bad argument");
  LAUNCH1D(box2_bwd_k, NC * H * W, dxb, mask, dx, NC, H, W);
  AGL_CHECK_LAUNCH("agl_box2_bwd");
  return AGL_OK;
}

int agl_grid_gather_fwd(const float* x, const int* map_y, const int* map_x, float* y, long NC, int h, int w, int H, int W, void* stream) {
  AGL_REQUIRE(x && map_y && map_x && y && NC > 0 && h > 0 && w > 0 && H > 0 && W > 0 && NC * H * W < (1L << 31) && NC * h * w < (1L << 31), "agl_grid_gather_fwd: bad argument");
  LAUNCH1D(grid_gather_fwd_k, NC * H * W, x, map_y, map_x, y, NC, h, w, H, W);
  AGL_CHECK_LAUNCH("agl_grid_gather_fwd");
  return AGL_OK;
}

int agl_grid_gather_bwd(const float* dy, const int* lo_y, const int* lo_x, float* dx, long NC, int h, int w, int H, int W, void* stream) {
  AGL_REQUIRE(dy && lo_y && lo_x && dx && NC > 0 && h > 0 && w > 0 && H > 0 && W > 0 && NC * H * W < (1L << 31) && NC * h * w < (1L << 31), "agl_grid_gather_bwd: bad argument");
  LAUNCH1D(grid_gather_bwd_k, NC * h * w, dy, lo_y, lo_x, dx, NC, h, w, H, W);
  AGL_CHECK_LAUNCH("agl_grid_gather_bwd");
  return AGL_OK;
}

}  // extern "C"
