// Spectral normalisation (torch.nn.utils.spectral_norm as applied by add_sn, reference
// models/discriminator.py:15-22), batched over all layers of one discriminator so a forward pass
// costs five launches instead of five per layer.  Per layer, W is [rows=Cout][cols=Cin*k*k]:
//   v <- normalize(W^T u) ; u <- normalize(W v) ; sigma = u . (W v) ; W_sn = W / sigma      (eps 1e-12)
// Backward of W_sn = W / sigma(W) with u, v constants:  dW = (G - <G, W_sn> u v^T) / sigma.
// All reductions use fixed orders (no atomics): u/v stay bit-identical across data-parallel replicas.
#include "agl_internal.h"

struct AglSnLayer {       // one entry per layer (host array, passed to the kernels by value)
  const float* w;         // weight_orig [rows][cols]
  float* u;               // [rows]  (power iteration updates it in place)
  float* v;               // [cols]  (power iteration updates it in place)
  float* w_sn;            // output [rows][cols]
  float* sigma;           // output scalar
  float* tmp;             // scratch: agl_sn_tmp_floats(rows, cols) floats
  float* u_used;          // output [rows]: the u that defines sigma (kept for backward)
  float* v_used;          // output [cols]
  const float* g;         // backward: grad wrt w_sn
  float* dw;              // backward: grad wrt weight_orig (overwritten, or += when accumulate is set)
  int rows, cols;
};
constexpr int AGL_SN_MAX_LAYERS = 24;
struct AglSnBatch { AglSnLayer l[AGL_SN_MAX_LAYERS]; };

namespace {

constexpr int RCH = 64;   // rows per partial chunk in the W^T u pass

// P1: part[rc][k] = sum_{r in chunk rc} W[r][k] * u[r]
__global__ __launch_bounds__(256) void sn_wtu_partial(const AglSnBatch L) {
  const AglSnLayer& l = L.l[blockIdx.z];
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * RCH;
  if (r0 >= l.rows || blockIdx.x * 256 >= l.cols) return;
  const int r1 = min(l.rows, r0 + RCH);
  if (k < l.cols) {
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += l.w[(long)r * l.cols + k] * l.u[r];
    l.tmp[max(l.rows, l.cols) + (long)blockIdx.y * l.cols + k] = s;
  }
}
// P2: t = sum of partials ; v = t / max(||t||, eps)
__global__ __launch_bounds__(256) void sn_finish_v(const AglSnBatch L, float eps) {
  __shared__ double sc[4];
  const AglSnLayer& l = L.l[blockIdx.x];
  const int nch = (l.rows + RCH - 1) / RCH;
  const float* part = l.tmp + max(l.rows, l.cols);
  double ss = 0.0;
  for (int k = threadIdx.x; k < l.cols; k += 256) {
    float t = 0.f;
#pragma unroll 8
    for (int c = 0; c < nch; ++c) t += part[(long)c * l.cols + k];      // (unrolled: the loads of a batch are in flight together)
    l.tmp[k] = t;
    ss += (double)t * t;
  }
  ss = block_sum_256(ss, sc);
  const float nrm = fmaxf((float)sqrt(ss), eps);
  for (int k = threadIdx.x; k < l.cols; k += 256) l.v[k] = l.tmp[k] / nrm;
}
// P3: s[r] = sum_k W[r][k] v[k]   (one wave per row)
__global__ __launch_bounds__(256) void sn_wv(const AglSnBatch L) {
  const AglSnLayer& l = L.l[blockIdx.y];
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= l.rows) return;
  float s = 0.f;
  for (int k = threadIdx.x & 63; k < l.cols; k += 64) s += l.w[(long)r * l.cols + k] * l.v[k];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) l.tmp[r] = s;
}
// P4: u = s / max(||s||, eps) ; sigma = u . s
__global__ __launch_bounds__(256) void sn_finish_u(const AglSnBatch L, float eps, int update_u) {
  __shared__ double sc[4];
  const AglSnLayer& l = L.l[blockIdx.x];
  double ss = 0.0;
  for (int r = threadIdx.x; r < l.rows; r += 256) ss += (double)l.tmp[r] * l.tmp[r];
  ss = block_sum_256(ss, sc);
  const float nrm = fmaxf((float)sqrt(ss), eps);
  double dot = 0.0;
  for (int r = threadIdx.x; r < l.rows; r += 256) {
    float un = update_u ? l.tmp[r] / nrm : l.u[r];
    if (update_u) l.u[r] = un;
    l.u_used[r] = un;
    dot += (double)un * l.tmp[r];
  }
  for (int k = threadIdx.x; k < l.cols; k += 256) l.v_used[k] = l.v[k];
  dot = block_sum_256(dot, sc);
  if (threadIdx.x == 0) *l.sigma = (float)dot;
}
// P5: W_sn = W / sigma
__global__ void sn_scale(const AglSnBatch L) {
  const AglSnLayer& l = L.l[blockIdx.y];
  const long n = (long)l.rows * l.cols;
  const float sg = *l.sigma;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) l.w_sn[i] = l.w[i] / sg;
}

// backward B1: per-row partial of <G, W_sn> into tmp[r]
__global__ __launch_bounds__(256) void sn_bwd_rowdot(const AglSnBatch L) {
  const AglSnLayer& l = L.l[blockIdx.y];
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= l.rows) return;
  float s = 0.f;
  for (int k = threadIdx.x & 63; k < l.cols; k += 64) s += l.g[(long)r * l.cols + k] * l.w_sn[(long)r * l.cols + k];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) l.tmp[r] = s;
}
// backward B2: dot = sum_r tmp[r] -> tmp[rows] (block 0 of each layer), then B3 elementwise
__global__ __launch_bounds__(256) void sn_bwd_dot(const AglSnBatch L) {
  __shared__ double sc[4];
  const AglSnLayer& l = L.l[blockIdx.x];
  double s = 0.0;
  for (int r = threadIdx.x; r < l.rows; r += 256) s += l.tmp[r];
  s = block_sum_256(s, sc);
  if (threadIdx.x == 0) l.tmp[max(l.rows, l.cols)] = (float)s;
}
__global__ void sn_bwd_apply(const AglSnBatch L, int accumulate) {
  const AglSnLayer& l = L.l[blockIdx.y];
  const long n = (long)l.rows * l.cols;
  const float sg = *l.sigma, dot = l.tmp[max(l.rows, l.cols)];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int r = (int)(i / l.cols), k = (int)(i - (long)r * l.cols);
    const float d = (l.g[i] - dot * l.u_used[r] * l.v_used[k]) / sg;
    l.dw[i] = accumulate ? l.dw[i] + d : d;
  }
}

}  // namespace

extern "C" {

long agl_sn_layer_desc_bytes(void) { return (long)sizeof(AglSnLayer); }
long agl_sn_tmp_floats(int rows, int cols) {
  return (long)(rows > cols ? rows : cols) + (long)((rows + RCH - 1) / RCH) * cols + 64;
}

static int sn_pack(const void* layers, int n_layers, AglSnBatch& L, int& max_rows, int& max_cols, const char* who) {
  AGL_REQUIRE(layers && n_layers > 0 && n_layers <= AGL_SN_MAX_LAYERS, "%s: need 1..%d layers", who, AGL_SN_MAX_LAYERS);
  const AglSnLayer* in = (const AglSnLayer*)layers;
  max_rows = max_cols = 0;
  for (int i = 0; i < n_layers; ++i) {
    L.l[i] = in[i];
    AGL_REQUIRE(in[i].rows > 0 && in[i].cols > 0 && in[i].w && in[i].u && in[i].v && in[i].w_sn && in[i].sigma && in[i].tmp &&
                    in[i].u_used && in[i].v_used, "%s: layer %d incomplete", who, i);
    max_rows = in[i].rows > max_rows ? in[i].rows : max_rows;
    max_cols = in[i].cols > max_cols ? in[i].cols : max_cols;
  }
  return AGL_OK;
}

// layers: HOST array of n_layers AglSnLayer (<= 24), passed to the kernels by value.
// power_iter=1: training forward (u, v advanced in place); 0: eval (sigma from the stored u, v).
int agl_sn_forward(const void* layers, int n_layers, int power_iter, float eps, void* stream) {
  AglSnBatch L; int max_rows, max_cols;
  int rc = sn_pack(layers, n_layers, L, max_rows, max_cols, "agl_sn_forward");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (power_iter) {
    hipLaunchKernelGGL(sn_wtu_partial, dim3(agl_cdiv(max_cols, 256), agl_cdiv(max_rows, RCH), n_layers), dim3(256), 0, st, L);
    AGL_CHECK_LAUNCH("agl_sn_forward(wtu)");
    hipLaunchKernelGGL(sn_finish_v, dim3(n_layers), dim3(256), 0, st, L, eps);
    AGL_CHECK_LAUNCH("agl_sn_forward(v)");
  }
  hipLaunchKernelGGL(sn_wv, dim3(agl_cdiv(max_rows, 4), n_layers), dim3(256), 0, st, L);
  AGL_CHECK_LAUNCH("agl_sn_forward(wv)");
  hipLaunchKernelGGL(sn_finish_u, dim3(n_layers), dim3(256), 0, st, L, eps, power_iter);
  AGL_CHECK_LAUNCH("agl_sn_forward(u)");
  long maxn = (long)max_rows * max_cols;
  int gx = agl_cdiv(maxn, 256 * 4);
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(sn_scale, dim3(gx, n_layers), dim3(256), 0, st, L);
  AGL_CHECK_LAUNCH("agl_sn_forward(scale)");
  return AGL_OK;
}

// dw = (g - <g, w_sn> u v^T) / sigma for every layer (u_used, v_used, sigma, w_sn as left by the matching forward).
int agl_sn_backward(const void* layers, int n_layers, int accumulate, void* stream) {
  AglSnBatch L; int max_rows, max_cols;
  int rc = sn_pack(layers, n_layers, L, max_rows, max_cols, "agl_sn_backward");
  if (rc) return rc;
  for (int i = 0; i < n_layers; ++i) AGL_REQUIRE(L.l[i].g && L.l[i].dw, "agl_sn_backward: layer %d has no g/dw", i);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sn_bwd_rowdot, dim3(agl_cdiv(max_rows, 4), n_layers), dim3(256), 0, st, L);
  AGL_CHECK_LAUNCH("agl_sn_backward(rowdot)");
  hipLaunchKernelGGL(sn_bwd_dot, dim3(n_layers), dim3(256), 0, st, L);
  AGL_CHECK_LAUNCH("agl_sn_backward(dot)");
  long maxn = (long)max_rows * max_cols;
  int gx = agl_cdiv(maxn, 256 * 4);
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(sn_bwd_apply, dim3(gx, n_layers), dim3(256), 0, st, L, accumulate);
  AGL_CHECK_LAUNCH("agl_sn_backward(apply)");
  return AGL_OK;
}

}  // extern "C"
