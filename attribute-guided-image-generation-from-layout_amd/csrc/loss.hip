// Loss terms of the reference training loop (train64.py:195-245, :284-354) as single-workgroup,
// fixed-order (deterministic) kernels: every call writes the UNWEIGHTED loss value to *loss_out and,
// when dx != NULL, coef * d(loss)/dx — the harness passes coef = lambda * mix weight so no autograd
// node or device-side scalar is needed for the loss combination.
#include "agl_internal.h"

namespace {

constexpr int LT = 1024;

__device__ __forceinline__ double block_sum_1024(double v, double* sc16) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sc16[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += sc16[i];
  return s;
}

// log(sigmoid(x)) = min(x,0) - log1p(exp(-|x|))
__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_neg(float x) { return 1.f / (1.f + expf(x)); }  // sigmoid(-x)

__global__ __launch_bounds__(LT) void bce_const_k(const float* __restrict__ x, long n, float target, float coef,
                                                  float* __restrict__ loss_out, float* __restrict__ dx) {
  __shared__ double sc[16];
  double s = 0.0;
  const float inv = 1.0f / (float)n;
  for (long i = threadIdx.x; i < n; i += LT) {
    const float xi = x[i];
    s += (double)((1.f - target) * xi - log_sigmoid(xi));
    if (dx) dx[i] = coef * inv * ((1.f - target) - sigmoid_neg(xi));
  }
  s = block_sum_1024(s, sc);
  if (threadIdx.x == 0) *loss_out = (float)(s / (double)n);
}

// rows with sum(targets[row]) != 0 are the annotated ones (train64.py:241,323).  Each wave walks rows r = wave, wave+16, ...
// with its lanes along the attributes; the row's annotated flag is recomputed in the second pass instead of being kept
// in a fixed-size table, so any number of rows is accepted (the reference has no limit).
__global__ __launch_bounds__(LT) void bce_posw_k(const float* __restrict__ x, const float* __restrict__ tg,
                                                 const float* __restrict__ pw, long rows, int A, float coef,
                                                 float* __restrict__ loss_out, float* __restrict__ dx) {
  __shared__ double sc[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double cnt = 0.0;
  for (long r = wave; r < rows; r += 16) {
    float t = 0.f;
    for (int a = lane; a < A; a += 64) t += tg[r * A + a];
    t = wave_sum(t);                       // targets are multi-hot (>= 0): the sum is zero iff every entry is
    if (lane == 0 && t != 0.f) cnt += 1.0;
  }
  cnt = block_sum_1024(cnt, sc);
  const double denom = cnt * (double)A;
  const float inv = denom > 0.0 ? (float)(1.0 / denom) : 0.f;
  double s = 0.0;
  for (long r = wave; r < rows; r += 16) {
    float t = 0.f;
    for (int a = lane; a < A; a += 64) t += tg[r * A + a];
    t = wave_sum(t);
    const bool sel = t != 0.f;
    for (int a = lane; a < A; a += 64) {
      const long i = r * A + a;
      float g = 0.f;
      if (sel) {
        const float xi = x[i], ti = tg[i];
        const float lw = (pw[a] - 1.f) * ti + 1.f;
        s += (double)((1.f - ti) * xi - lw * log_sigmoid(xi));
        g = coef * inv * ((1.f - ti) - lw * sigmoid_neg(xi));
      }
      if (dx) dx[i] = g;
    }
  }
  s = block_sum_1024(s, sc);
  if (threadIdx.x == 0) *loss_out = denom > 0.0 ? (float)(s / denom) : nanf("");
}

__global__ __launch_bounds__(LT) void ce_k(const float* __restrict__ lg, const long long* __restrict__ lab, long R, int V,
                                           float coef, float* __restrict__ loss_out, float* __restrict__ dl) {
  __shared__ double sc[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc = 0.0;
  const float inv = 1.0f / (float)R;
  for (long r = wave; r < R; r += 16) {
    const float* row = lg + r * V;
    float mx = -INFINITY;
    for (int j = lane; j < V; j += 64) mx = fmaxf(mx, row[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float se = 0.f;
    for (int j = lane; j < V; j += 64) se += expf(row[j] - mx);
    se = wave_sum(se);
    const float lse = mx + logf(se);
    const long long yl = lab[r];
    const bool ok = yl >= 0 && yl < V;            // torch raises on a bad label; here the result is poisoned instead
    const int y = ok ? (int)yl : 0;
    if (lane == 0) acc += ok ? (double)(lse - row[y]) : (double)nanf("");
    if (dl)
      for (int j = lane; j < V; j += 64)
        dl[r * V + j] = ok ? coef * inv * (expf(row[j] - lse) - (j == y ? 1.f : 0.f)) : nanf("");
  }
  acc = block_sum_1024(acc, sc);
  if (threadIdx.x == 0) *loss_out = (float)(acc / (double)R);
}

// The two row-wise losses over several workgroups (agl_cross_entropy_ws / agl_bce_logits_posw_ws): a wave per row as above, 16 rows per
// workgroup, the row's loss term left in scratch; ONE workgroup then adds the terms — wave w the rows w, w + 16, ... in increasing order,
// the sixteen sums in wave order: fixed order, deterministic — and writes the scalar.  (On one workgroup the 393 x 179 cross-entropy of
// a 64-image batch took 77 us, the 393 x 106 attribute loss 70 us, each between a discriminator's forward and its backward.)
__global__ __launch_bounds__(LT) void ce_rows_k(const float* __restrict__ lg, const long long* __restrict__ lab, long R, int V, float coef,
                                                double* __restrict__ rowloss, float* __restrict__ dl) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 16 + wave;
  if (r >= R) return;
  const float inv = 1.0f / (float)R;
  const float* row = lg + r * V;
  float mx = -INFINITY;
  for (int j = lane; j < V; j += 64) mx = fmaxf(mx, row[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float se = 0.f;
  for (int j = lane; j < V; j += 64) se += expf(row[j] - mx);
  se = wave_sum(se);
  const float lse = mx + logf(se);
  const long long yl = lab[r];
  const bool ok = yl >= 0 && yl < V;            // torch raises on a bad label; here the result is poisoned instead
  const int y = ok ? (int)yl : 0;
  if (lane == 0) rowloss[r] = ok ? (double)(lse - row[y]) : (double)nanf("");
  if (dl)
    for (int j = lane; j < V; j += 64)
      dl[r * V + j] = ok ? coef * inv * (expf(row[j] - lse) - (j == y ? 1.f : 0.f)) : nanf("");
}
// *loss_out = (sum of rowloss) / R  — the single-workgroup kernel's own order and arithmetic (bit-identical value)
__global__ __launch_bounds__(LT) void ce_final_k(const double* __restrict__ rowloss, long R, float* __restrict__ loss_out) {
  __shared__ double sc[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc = 0.0;
  if (lane == 0)
    for (long r = wave; r < R; r += 16) acc += rowloss[r];
  acc = block_sum_1024(acc, sc);
  if (threadIdx.x == 0) *loss_out = (float)(acc / (double)R);
}

// rowterm[r] = sum over the attributes of row r of its loss terms (0 for an un-annotated row), rowflag[r] = 1 for an annotated row
__global__ __launch_bounds__(LT) void bce_posw_rows_k(const float* __restrict__ x, const float* __restrict__ tg, const float* __restrict__ pw,
                                                      long rows, int A, double* __restrict__ rowterm, double* __restrict__ rowflag) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 16 + wave;
  if (r >= rows) return;
  float t = 0.f;
  for (int a = lane; a < A; a += 64) t += tg[r * A + a];
  t = wave_sum(t);                       // targets are multi-hot (>= 0): the sum is zero iff every entry is
  const bool sel = t != 0.f;
  double s = 0.0;
  if (sel)
    for (int a = lane; a < A; a += 64) {
      const long i = r * A + a;
      const float xi = x[i], ti = tg[i];
      const float lw = (pw[a] - 1.f) * ti + 1.f;
      s += (double)((1.f - ti) * xi - lw * log_sigmoid(xi));
    }
  s = wave_sum(s);
  if (lane == 0) { rowterm[r] = s; rowflag[r] = sel ? 1.0 : 0.0; }
}
// denominator and loss value (one workgroup, fixed order); leaves 1 / denominator for the gradient pass in *inv_out
__global__ __launch_bounds__(LT) void bce_posw_final_k(const double* __restrict__ rowterm, const double* __restrict__ rowflag, long rows, int A,
                                                       float* __restrict__ loss_out, float* __restrict__ inv_out) {
  __shared__ double sc[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double cnt = 0.0, s = 0.0;
  if (lane == 0)
    for (long r = wave; r < rows; r += 16) { cnt += rowflag[r]; s += rowterm[r]; }
  cnt = block_sum_1024(cnt, sc);
  s = block_sum_1024(s, sc);
  if (threadIdx.x == 0) {
    const double denom = cnt * (double)A;
    *loss_out = denom > 0.0 ? (float)(s / denom) : nanf("");
    *inv_out = denom > 0.0 ? (float)(1.0 / denom) : 0.f;
  }
}
__global__ __launch_bounds__(LT) void bce_posw_grad_k(const float* __restrict__ x, const float* __restrict__ tg, const float* __restrict__ pw,
                                                      const double* __restrict__ rowflag, const float* __restrict__ inv_in, long rows, int A,
                                                      float coef, float* __restrict__ dx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 16 + wave;
  if (r >= rows) return;
  const bool sel = rowflag[r] != 0.0;
  const float inv = *inv_in;
  for (int a = lane; a < A; a += 64) {
    const long i = r * A + a;
    float g = 0.f;
    if (sel) {
      const float xi = x[i], ti = tg[i];
      const float lw = (pw[a] - 1.f) * ti + 1.f;
      g = coef * inv * ((1.f - ti) - lw * sigmoid_neg(xi));
    }
    dx[i] = g;
  }
}

// pass 1: block b sums its contiguous chunk of the N*len elements (fixed order inside the block) into part[b];
// pass 2 (one wave) adds the partials in block order — deterministic for a given (N, len).
constexpr int L1_BLOCKS = 256;
__global__ __launch_bounds__(LT) void l1_rows_part_k(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ keep, long N, long len, float coef, float denom,
                                                     double* __restrict__ part, float* __restrict__ da) {
  __shared__ double sc[16];
  const long n = N * len, per = (n + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double s = 0.0;
  const float inv = 1.0f / ((float)len * denom);
  for (long i = lo + threadIdx.x; i < hi; i += LT) {
    const float k = keep ? keep[i / len] : 1.f;
    const float d = a[i] - b[i];
    s += (double)(k * fabsf(d));
    if (da) da[i] = coef * inv * k * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
  }
  s = block_sum_1024(s, sc);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void l1_rows_final_k(const double* __restrict__ part, int nb, double scale, float* __restrict__ loss_out) {
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nb; ++i) s += part[i];
    *loss_out = (float)(s * scale);
  }
}

// Hinge GAN losses of the vendored SPADE GANLoss (models/spade/networks/loss.py:65-76) — NOT on the reference's train
// path (train64.py uses BCE-with-logits); exported because the module surface names them.
//   mode 0: discriminator, real:  -mean(min(x - 1, 0));   1: discriminator, fake: -mean(min(-x - 1, 0));   2: generator: -mean(x)
__global__ __launch_bounds__(LT) void hinge_k(const float* __restrict__ x, long n, int mode, float coef, float* __restrict__ loss_out,
                                              float* __restrict__ dx) {
  __shared__ double sc[16];
  double s = 0.0;
  const float inv = 1.0f / (float)n;
  for (long i = threadIdx.x; i < n; i += LT) {
    const float xi = x[i];
    float v, g;
    if (mode == 0) { const float m = xi - 1.f; v = fminf(m, 0.f); g = m < 0.f ? -1.f : 0.f; }        // torch.min(a, 0): grad to a where a < 0
    else if (mode == 1) { const float m = -xi - 1.f; v = fminf(m, 0.f); g = m < 0.f ? 1.f : 0.f; }
    else { v = xi; g = -1.f; }
    s += (double)v;
    if (dx) dx[i] = coef * inv * g;
  }
  s = block_sum_1024(s, sc);
  if (threadIdx.x == 0) *loss_out = (float)(-s / (double)n);
}

__global__ __launch_bounds__(LT) void kl_k(const float* __restrict__ mu, const float* __restrict__ lv, long n, float coef,
                                           float* __restrict__ loss_out, float* __restrict__ dmu, float* __restrict__ dlv) {
  __shared__ double sc[16];
  double s = 0.0;
  for (long i = threadIdx.x; i < n; i += LT) {
    const float m = mu[i], l = lv[i], e = expf(l);
    s += (double)(1.f + l - m * m - e);
    if (dmu) dmu[i] = coef * m;
    if (dlv) dlv[i] = coef * (-0.5f) * (1.f - e);
  }
  s = block_sum_1024(s, sc);
  if (threadIdx.x == 0) *loss_out = (float)(-0.5 * s);
}

}  // namespace

extern "C" {

int agl_bce_logits_const(const float* x, long n, float target, float coef, float* loss_out, float* dx, void* stream) {
  AGL_REQUIRE(x && loss_out && n > 0, "agl_bce_logits_const: bad argument");
  hipLaunchKernelGGL(bce_const_k, dim3(1), dim3(LT), 0, (hipStream_t)stream, x, n, target, coef, loss_out, dx);
  AGL_CHECK_LAUNCH("agl_bce_logits_const");
  return AGL_OK;
}

int agl_bce_logits_posw(const float* x, const float* targets, const float* pos_weight, long rows, int A, float coef,
                        float* loss_out, float* dx, void* stream) {
  AGL_REQUIRE(x && targets && pos_weight && loss_out && rows > 0 && A > 0, "agl_bce_logits_posw: bad argument");
  hipLaunchKernelGGL(bce_posw_k, dim3(1), dim3(LT), 0, (hipStream_t)stream, x, targets, pos_weight, rows, A, coef, loss_out, dx);
  AGL_CHECK_LAUNCH("agl_bce_logits_posw");
  return AGL_OK;
}

int agl_cross_entropy(const float* logits, const long long* labels, long R, int V, float coef, float* loss_out,
                      float* dlogits, void* stream) {
  AGL_REQUIRE(logits && labels && loss_out && R > 0 && V > 0, "agl_cross_entropy: bad argument");
  hipLaunchKernelGGL(ce_k, dim3(1), dim3(LT), 0, (hipStream_t)stream, logits, labels, R, V, coef, loss_out, dlogits);
  AGL_CHECK_LAUNCH("agl_cross_entropy");
  return AGL_OK;
}

// The same two losses over several workgroups, with scratch for the row terms (include/agl.h)
long agl_loss_rows_ws_bytes(long rows) { return (2 * rows + 2) * (long)sizeof(double); }

int agl_cross_entropy_ws(const float* logits, const long long* labels, long R, int V, float coef, float* loss_out, float* dlogits, void* ws,
                         long ws_bytes, void* stream) {
  AGL_REQUIRE(logits && labels && loss_out && R > 0 && V > 0, "agl_cross_entropy_ws: bad argument");
  AGL_REQUIRE(ws && ws_bytes >= agl_loss_rows_ws_bytes(R), "agl_cross_entropy_ws: workspace too small");
  double* rowloss = (double*)ws;
  hipLaunchKernelGGL(ce_rows_k, dim3((unsigned)((R + 15) / 16)), dim3(LT), 0, (hipStream_t)stream, logits, labels, R, V, coef, rowloss, dlogits);
  AGL_CHECK_LAUNCH("agl_cross_entropy_ws(rows)");
  hipLaunchKernelGGL(ce_final_k, dim3(1), dim3(LT), 0, (hipStream_t)stream, (const double*)rowloss, R, loss_out);
  AGL_CHECK_LAUNCH("agl_cross_entropy_ws(final)");
  return AGL_OK;
}

int agl_bce_logits_posw_ws(const float* x, const float* targets, const float* pos_weight, long rows, int A, float coef, float* loss_out,
                           float* dx, void* ws, long ws_bytes, void* stream) {
  AGL_REQUIRE(x && targets && pos_weight && loss_out && rows > 0 && A > 0, "agl_bce_logits_posw_ws: bad argument");
  AGL_REQUIRE(ws && ws_bytes >= agl_loss_rows_ws_bytes(rows), "agl_bce_logits_posw_ws: workspace too small");
  double* rowterm = (double*)ws;
  double* rowflag = rowterm + rows;
  float* inv = (float*)(rowflag + rows);
  const dim3 g((unsigned)((rows + 15) / 16));
  hipLaunchKernelGGL(bce_posw_rows_k, g, dim3(LT), 0, (hipStream_t)stream, x, targets, pos_weight, rows, A, rowterm, rowflag);
  AGL_CHECK_LAUNCH("agl_bce_logits_posw_ws(rows)");
  hipLaunchKernelGGL(bce_posw_final_k, dim3(1), dim3(LT), 0, (hipStream_t)stream, (const double*)rowterm, (const double*)rowflag, rows, A, loss_out, inv);
  AGL_CHECK_LAUNCH("agl_bce_logits_posw_ws(final)");
  if (dx) {
    hipLaunchKernelGGL(bce_posw_grad_k, g, dim3(LT), 0, (hipStream_t)stream, x, targets, pos_weight, (const double*)rowflag, (const float*)inv, rows, A,
                       coef, dx);
    AGL_CHECK_LAUNCH("agl_bce_logits_posw_ws(gradient)");
  }
  return AGL_OK;
}

long agl_l1_rows_ws_bytes(void) { return (long)L1_BLOCKS * (long)sizeof(double); }

int agl_l1_rows(const float* a, const float* b, const float* keep, long N, long len, float coef, float denom,
                float* loss_out, float* da, void* ws, long ws_bytes, void* stream) {
  AGL_REQUIRE(a && b && loss_out && N > 0 && len > 0 && denom != 0.f, "agl_l1_rows: bad argument");
  AGL_REQUIRE(ws && ws_bytes >= agl_l1_rows_ws_bytes(), "agl_l1_rows: workspace too small (%ld < %ld)", ws_bytes, agl_l1_rows_ws_bytes());
  const long n = N * len;
  int nb = (int)((n + 4L * LT - 1) / (4L * LT));
  nb = nb < 1 ? 1 : (nb > L1_BLOCKS ? L1_BLOCKS : nb);
  hipLaunchKernelGGL(l1_rows_part_k, dim3(nb), dim3(LT), 0, (hipStream_t)stream, a, b, keep, N, len, coef, denom, (double*)ws, da);
  AGL_CHECK_LAUNCH("agl_l1_rows");
  hipLaunchKernelGGL(l1_rows_final_k, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)ws, nb,
                     1.0 / ((double)len * (double)denom), loss_out);
  AGL_CHECK_LAUNCH("agl_l1_rows(final)");
  return AGL_OK;
}

int agl_hinge_loss(const float* x, long n, int mode, float coef, float* loss_out, float* dx, void* stream) {
  AGL_REQUIRE(x && loss_out && n > 0 && mode >= 0 && mode <= 2, "agl_hinge_loss: bad argument");
  hipLaunchKernelGGL(hinge_k, dim3(1), dim3(LT), 0, (hipStream_t)stream, x, n, mode, coef, loss_out, dx);
  AGL_CHECK_LAUNCH("agl_hinge_loss");
  return AGL_OK;
}

int agl_kl_sum(const float* mu, const float* logvar, long n, float coef, float* loss_out, float* dmu, float* dlogvar,
               void* stream) {
  AGL_REQUIRE(mu && logvar && loss_out && n > 0, "agl_kl_sum: bad argument");
  hipLaunchKernelGGL(kl_k, dim3(1), dim3(LT), 0, (hipStream_t)stream, mu, logvar, n, coef, loss_out, dmu, dlogvar);
  AGL_CHECK_LAUNCH("agl_kl_sum");
  return AGL_OK;
}

}  // extern "C"
