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

// rows with sum(targets[row]) != 0 are the annotated ones (train64.py:241,323)
__global__ __launch_bounds__(LT) void bce_posw_k(const float* __restrict__ x, const float* __restrict__ tg,
                                                 const float* __restrict__ pw, long rows, int A, float coef,
                                                 float* __restrict__ loss_out, float* __restrict__ dx) {
  __shared__ double sc[16];
  __shared__ unsigned char sel[4096];

  double cnt = 0.0;
  for (long r = threadIdx.x; r < rows; r += LT) {
    float t = 0.f;
    for (int a = 0; a < A; ++a) t += tg[r * A + a];
    sel[r] = t != 0.f;
    cnt += sel[r] ? 1.0 : 0.0;
  }
  cnt = block_sum_1024(cnt, sc);
  const double denom = cnt * (double)A;
  const float inv = denom > 0.0 ? (float)(1.0 / denom) : 0.f;
  double s = 0.0;
  for (long i = threadIdx.x; i < rows * A; i += LT) {
    const long r = i / A;
    const int a = (int)(i - r * A);
    float g = 0.f;
    if (sel[r]) {
      const float xi = x[i], t = tg[i];
      const float lw = (pw[a] - 1.f) * t + 1.f;
      s += (double)((1.f - t) * xi - lw * log_sigmoid(xi));
      g = coef * inv * ((1.f - t) - lw * sigmoid_neg(xi));
    }
    if (dx) dx[i] = g;
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
    const int y = (int)lab[r];
    if (lane == 0) acc += (double)(lse - row[y]);
    if (dl)
      for (int j = lane; j < V; j += 64) dl[r * V + j] = coef * inv * (expf(row[j] - lse) - (j == y ? 1.f : 0.f));
  }
  acc = block_sum_1024(acc, sc);
  if (threadIdx.x == 0) *loss_out = (float)(acc / (double)R);
}

__global__ __launch_bounds__(LT) void l1_rows_k(const float* __restrict__ a, const float* __restrict__ b,
                                                const float* __restrict__ keep, long N, long len, float coef, float denom,
                                                float* __restrict__ loss_out, float* __restrict__ da) {
  __shared__ double sc[16];
  double s = 0.0;
  const float inv = 1.0f / ((float)len * denom);
  for (long i = threadIdx.x; i < N * len; i += LT) {
    const float k = keep ? keep[i / len] : 1.f;
    const float d = a[i] - b[i];
    s += (double)(k * fabsf(d));
    if (da) da[i] = coef * inv * k * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
  }
  s = block_sum_1024(s, sc);
  if (threadIdx.x == 0) *loss_out = (float)(s / ((double)len * (double)denom));
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
  AGL_REQUIRE(rows <= 4096, "agl_bce_logits_posw: at most 4096 rows per call (got %ld)", rows);
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

int agl_l1_rows(const float* a, const float* b, const float* keep, long N, long len, float coef, float denom,
                float* loss_out, float* da, void* stream) {
  AGL_REQUIRE(a && b && loss_out && N > 0 && len > 0 && denom != 0.f, "agl_l1_rows: bad argument");
  hipLaunchKernelGGL(l1_rows_k, dim3(1), dim3(LT), 0, (hipStream_t)stream, a, b, keep, N, len, coef, denom, loss_out, da);
  AGL_CHECK_LAUNCH("agl_l1_rows");
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
