// Numerics of v_mfma_f32_32x32x16_f16 / _bf16 on gfx950: how the 16 products and the accumulator are summed (window, truncation, order).
// Every experiment is one dot product d = c + sum_k a[k] * b[k] replicated over the whole tile.  hipcc --offload-arch=gfx950 -o probe this.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Exp { float a[32], b[32], c; };
__global__ void run(const Exp* e, float* out, int n, int mode) {
  const int lane = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    if (mode < 2) {
      f32x16 acc;
      for (int r = 0; r < 16; ++r) acc[r] = e[i].c;
      if (mode == 0) {
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)e[i].a[8 * (lane >> 5) + j]; b[j] = (_Float16)e[i].b[8 * (lane >> 5) + j]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
      } else {
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)e[i].a[8 * (lane >> 5) + j]; b[j] = (__bf16)e[i].b[8 * (lane >> 5) + j]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
      if (lane == 0) out[i] = acc[0];
    } else {      // 16x16x32 f16: lane holds k = 8*(lane>>4)+j
      f32x4 acc = {e[i].c, e[i].c, e[i].c, e[i].c};
      f16x8 a, b;
      for (int j = 0; j < 8; ++j) { a[j] = (_Float16)e[i].a[8 * (lane >> 4) + j]; b[j] = (_Float16)e[i].b[8 * (lane >> 4) + j]; }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
      if (lane == 0) out[i] = acc[0];
    }
  }
}
int main() {
  std::vector<Exp> ex; std::vector<const char*> names;
  auto add = [&](const char* nm, std::vector<float> a, std::vector<float> b, float c) {
    Exp e{}; for (size_t k = 0; k < a.size(); ++k) { e.a[k] = a[k]; e.b[k] = b[k]; } e.c = c; ex.push_back(e); names.push_back(nm);
  };
  auto rep = [](int n, float v) { return std::vector<float>(n, v); };
  // guard bits below ulp(C): C = 1, 16 products of 2^-(23+g) each: exact = 1 + 2^-(19+g)
  for (int g = 1; g <= 6; ++g) {
    static char nm[8][64]; snprintf(nm[g], 64, "C=1 + 16 x 2^-%d", 23 + g);
    add(nm[g], rep(16, ldexpf(1.f, -12)), rep(16, ldexpf(1.f, -(11 + g))), 1.f);
  }
  // final rounding: C = 1 + one product 1.5 * 2^-24 (RNE -> 1 + 2^-23; truncation -> 1)
  add("C=1 + 1.5*2^-24", {1.5f}, {ldexpf(1.f, -24)}, 1.f);
  add("C=1 + 0.5*2^-23 (tie)", {1.f}, {ldexpf(1.f, -24)}, 1.f);
  add("C=1 - 16 x 2^-25", rep(16, -ldexpf(1.f, -12)), rep(16, ldexpf(1.f, -13)), 1.f);
  add("C=-1 + 16 x 2^-25", rep(16, ldexpf(1.f, -12)), rep(16, ldexpf(1.f, -13)), -1.f);
  add("C=-1 - 16 x 2^-25", rep(16, -ldexpf(1.f, -12)), rep(16, ldexpf(1.f, -13)), -1.f);
  // no accumulator: p0 = 1, 15 x 2^-24 (exact 1 + 15*2^-24: RNE 1 + 2^-20, trunc 1 + 7*2^-23)
  { std::vector<float> a = rep(16, 1.f), b = rep(16, ldexpf(1.f, -24)); b[0] = 1.f; add("p0=1, 15 x 2^-24, C=0", a, b, 0.f); }
  { std::vector<float> a = rep(16, 1.f), b = rep(16, ldexpf(1.f, -24)); b[15] = 1.f; add("p15=1, 15 x 2^-24, C=0", a, b, 0.f); }
  { std::vector<float> a = rep(16, 1.f), b = rep(16, ldexpf(1.f, -24)); b[7] = 1.f; add("p7=1, 15 x 2^-24, C=0", a, b, 0.f); }
  // 22-bit products against a larger one: p0 = 1, p1 = (1+2^-10)^2 * 2^-t = (1 + 2^-9 + 2^-20) 2^-t
  for (int t = 1; t <= 5; ++t) {
    static char nm[8][64]; snprintf(nm[t], 64, "1 + (1+2^-10)^2 * 2^-%d", t);
    add(nm[t], {1.f, 1.f + ldexpf(1.f, -10)}, {1.f, (1.f + ldexpf(1.f, -10)) * ldexpf(1.f, -t)}, 0.f);
  }
  // cancellation: big + small - big
  add("2^10 + 2^-14 - 2^10, C=0", {1024.f, ldexpf(1.f, -7), -1024.f}, {1.f, ldexpf(1.f, -7), 1.f}, 0.f);
  add("C=2^10; 2^-14 - 2^10", {ldexpf(1.f, -7), -1024.f}, {ldexpf(1.f, -7), 1.f}, 1024.f);
  add("k0: 2^10, k8: 2^-14, k9: -2^10", {1024.f, 0, 0, 0, 0, 0, 0, 0, ldexpf(1.f, -7), -1024.f}, {1.f, 0, 0, 0, 0, 0, 0, 0, ldexpf(1.f, -7), 1.f}, 0.f);
  // fp16 denormal inputs
  add("denormal a=2^-20 * b=2^10", {ldexpf(1.f, -20)}, {1024.f}, 0.f);
  Exp* d; float* o; const int n = (int)ex.size();
  hipMalloc(&d, n * sizeof(Exp)); hipMalloc(&o, n * sizeof(float));
  hipMemcpy(d, ex.data(), n * sizeof(Exp), hipMemcpyHostToDevice);
  const char* mn[3] = {"32x32x16 f16", "32x32x16 bf16", "16x16x32 f16"};
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(run, dim3(1), dim3(64), 0, 0, d, o, n, mode);
    std::vector<float> r(n); hipMemcpy(r.data(), o, n * sizeof(float), hipMemcpyDeviceToHost);
    printf("---- %s\n", mn[mode]);
    for (int i = 0; i < n; ++i) {
      double exact = ex[i].c;
      for (int k = 0; k < 32; ++k) {
        double a = mode == 1 ? (double)(float)(__bf16)ex[i].a[k] : (double)(float)(_Float16)ex[i].a[k];
        double b = mode == 1 ? (double)(float)(__bf16)ex[i].b[k] : (double)(float)(_Float16)ex[i].b[k];
        if (mode != 2 && k >= 16) break;
        exact += a * b;
      }
      printf("%-34s got %.10e  exact %.10e  (got - exact) / 2^-23 = %+.4f\n", names[i], r[i], exact, (r[i] - exact) / ldexp(1.0, -23) / fmax(1.0, pow(2.0, floor(log2(fabs(exact) > 0 ? fabs(exact) : 1.0)))));
    }
  }
  return 0;
}
