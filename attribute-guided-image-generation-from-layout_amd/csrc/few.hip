// Convolutions with a few (<= 4: RGB) channels on one side.  Their arithmetic is negligible — they are streams over the
// many-channel tensor — but as GEMMs they have a 3..147-wide dimension that leaves the generic 128-wide tiles almost empty
// (0.8..30 TFLOP/s, 0.5 TB/s).  The kernels here are shaped for them; all arithmetic is exact fp32.
//
//   few_bww_k   weight gradient with <= 4 INPUT channels (first layers: models/discriminator.py:29-60 OptimizedBlock 3->64,
//               generator_obj_att.py:367 CropEncoder c1 3->64 k7) — and, through the role swap of conv.hip, with <= 4 OUTPUT
//               channels (decoder c4 64->3 k7, generator_obj_att.py:516).  GEMM: rows = output channels (64 per workgroup),
//               columns = (input channel, tap) <= 160, reduction = pixels, on v_mfma_f32_32x32x2_f32.
//   few_cin_fwd_k   forward with <= 4 INPUT channels, 1x1 / 3x3, stride 1, "same" (OptimizedBlock c1 3->64 and its 1x1 shortcut
//               accumulated onto the residual branch, discriminator.py:36-60): a stream over the output — a thread owns four
//               consecutive pixels and walks all output channels with 16-byte stores (and 16-byte reads of the tensor it
//               accumulates onto); the 64 x 27 weights sit in LDS and are read as broadcasts.
#include <cstdint>
#include "agl_internal.h"
#include "few.h"
#include "pconv.h"
#include "spade.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT_ = 256;

struct FewBwwArgs {
  const float* dy; const float* x; float* slabs;
  int N, Cin, H, W, Cout, OH, OW, pad, in_relu;
  int tiles, tiles_per_split, ncols;
  int dy_bf16;      // dy holds bf16 elements (the bf16-stored input of a few-output-channel layer, in this role through the swap of conv.hip)
  InFold fold;      // optional (fold.cells): dy is a SPADE's raw input, modulated (+ ReLU: dy_relu) while it is staged (few.h)
  int dy_relu, dy_round_bf16;
};

// Tile = 128 output pixels = TH full rows of one image (OW in {32, 64, 128}, TH = 128 / OW).  The dy tile is staged as
// [64 channels][128 pixels] (pitch 132 floats: the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots), the input
// patch as [c][TH+KS-1][OW+KS-1] with zero padding.  Every wave takes 32 of the 128 pixels and accumulates the whole
// 64 x (32*NTL) product for them — an MFMA consumes 2 pixels: lanes 0-31 hold pixel 4g+j, lanes 32-63 pixel 4g+4+j of an
// 8-pixel group, so one 16-byte read of dy feeds 4 MFMAs.  The four partial products are added through LDS in a fixed
// order at the end and the workgroup writes one slab; slab_reduce (conv.hip) adds the slabs (deterministic).
template <int KS, int NTL, bool FOLD = false>      // FOLD: FewBwwArgs::fold compiled in (the 7x7 / 5-column-tile form: the 128 px decoder's c7)
__global__ __launch_bounds__(NT_, 2) void few_bww_k(FewBwwArgs p) {
  constexpr int KK = KS * KS, DP = 132, BMC = 64;
  constexpr int RED = BMC * NTL * 32;                  // floats of the cross-wave reduction buffer
  constexpr int DYF = BMC * DP;                        // floats of the dy tile
  // patch floats, worst case over OW in {32, 64, 128} (TH = 4, 2, 1) with 4 channels
  constexpr int P32 = 4 * (KS + 3) * (KS + 31), P64 = 4 * (KS + 1) * (KS + 63), P128 = 4 * KS * (KS + 127);
  constexpr int PATCHF = P32 > P64 ? (P32 > P128 ? P32 : P128) : (P64 > P128 ? P64 : P128);
  constexpr int STAGEF = DYF + PATCHF;
  __shared__ __attribute__((aligned(16))) float lds[STAGEF > RED ? STAGEF : RED];
  float* const Dl = lds;
  float* const Pl = lds + DYF;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int co0 = blockIdx.y * BMC;
  const int OW = p.OW, TH = 128 / OW, PH = TH + KS - 1, PW = OW + KS - 1;
  const int tpi = p.OH / TH;                           // tiles per image
  const int t_beg = blockIdx.x * p.tiles_per_split, t_end = min(p.tiles, t_beg + p.tiles_per_split);
  const long OHW = (long)p.OH * OW, HW = (long)p.H * p.W;

  // column of this lane in every column tile: (c, kh, kw) -> patch offset; columns >= ncols read offset 0 (discarded)
  int boff[NTL];
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt) {
    const int col = 32 * nt + l31;
    const int c = col / KK, t = col - c * KK, kh = t / KS, kw = t - kh * KS;
    boff[nt] = col < p.ncols ? (c * PH + kh) * PW + kw : 0;
  }
  // the wave's 32 pixels: pixel index q = 32*wave + 8*g + 4*lh + j  ->  (row, column) of the tile
  int pixoff[4];                                       // patch offset of pixel 32*wave + 8*g + 4*lh (j adds 1 per pixel: OW >= 32)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int q = 32 * wave + 8 * g + 4 * lh;
    pixoff[g] = (q / OW) * PW + (q % OW);
  }

  f32x16 acc[2][NTL];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][nt][r] = 0.f;

  // dy staging: item e = (channel, 16-byte piece of its 128 contiguous pixels); 8 items per thread
  float4 pd[8];
  auto gload_dy = [&](int tile) {
    const int img = tile / tpi, ty0 = (tile - img * tpi) * TH;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int e = tid + NT_ * r, co = e >> 5, pc = e & 31;
      const bool ok = co0 + co < p.Cout;
      const long idx = ok ? ((long)img * p.Cout + co0 + co) * OHW + (long)ty0 * OW + 4 * pc : 0;
      float4 v;
      if (p.dy_bf16) {      // four bf16 in one 8-byte load, widened (a shift)
        const uint2 b = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.dy) + idx);
        v = float4{__builtin_bit_cast(float, b.x << 16), __builtin_bit_cast(float, b.x & 0xffff0000u),
                   __builtin_bit_cast(float, b.y << 16), __builtin_bit_cast(float, b.y & 0xffff0000u)};
      } else {
        v = *reinterpret_cast<const float4*>(p.dy + idx);
      }
      if (FOLD && p.fold.cells && ok) {      // SPADE's modulate (+ ReLU) of this channel's four pixels (they share an image row: OW % 4 == 0)
        const int c = co0 + co, iy = ty0 + (4 * pc) / OW, ix0 = (4 * pc) % OW, G = p.fold.G;
        const float mu = p.fold.mean[c], rs = p.fold.scale[c];
        const float* const cb = p.fold.cells + ((size_t)(img * (p.Cout >> 3) + (c >> 3)) * (size_t)(G * G)) * 16 + (c & 7);
        const int ro = p.fold.map[iy] * G;
        float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float* const ce = cb + (size_t)(ro + p.fold.map[ix0 + k]) * 16;
          float t = spade_value(e4[k], mu, rs, ce[0], ce[8]);
          if (p.dy_relu) t = fmaxf(t, 0.f);
          if (p.dy_round_bf16) t = (float)(__bf16)t;
          e4[k] = t;
        }
        v = float4{e4[0], e4[1], e4[2], e4[3]};
      }
      pd[r] = ok ? v : float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto sstore_dy = [&]() {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int e = tid + NT_ * r, co = e >> 5, pc = e & 31;
      *reinterpret_cast<float4*>(Dl + co * DP + 4 * pc) = pd[r];
    }
  };
  auto stage_patch = [&](int tile) {
    const int img = tile / tpi, ty0 = (tile - img * tpi) * TH;
    const int np = p.Cin * PH * PW;
    for (int e = tid; e < np; e += NT_) {
      const int c = e / (PH * PW), r = e - c * (PH * PW), yy = r / PW, xx = r - yy * PW;
      const int iy = ty0 - p.pad + yy, ix = xx - p.pad;
      float v = 0.f;
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = p.x[((long)img * p.Cin + c) * HW + (long)iy * p.W + ix];
      if (p.in_relu) v = fmaxf(v, 0.f);
      Pl[e] = v;
    }
  };

  constexpr bool PREF = NTL < 5;                       // (160 accumulator registers leave no room for the prefetch: the 7x7 case is MFMA-bound)
  if (PREF && t_beg < t_end) gload_dy(t_beg);
  for (int tile = t_beg; tile < t_end; ++tile) {
    if (!PREF) gload_dy(tile);
    __syncthreads();                                   // everyone is done reading the previous tile
    sstore_dy();
    stage_patch(tile);
    __syncthreads();
    if (PREF && tile + 1 < t_end) gload_dy(tile + 1);  // in flight during this tile's MFMAs
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int q = 32 * wave + 8 * g + 4 * lh;
      float4 a[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const float4*>(Dl + (32 * m + l31) * DP + q);
      float b[NTL][4];
#pragma unroll
      for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) b[nt][j] = Pl[boff[nt] + pixoff[g] + j];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const float av = j == 0 ? a[m].x : (j == 1 ? a[m].y : (j == 2 ? a[m].z : a[m].w));
#pragma unroll
          for (int nt = 0; nt < NTL; ++nt) acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[nt][j], acc[m][nt], 0, 0, 0);
        }
      }
    }
  }

  // ---- cross-wave sum in wave order, then the slab.  C tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* const R = lds;
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float* dst = R + row * (NTL * 32) + 32 * nt + l31;
            *dst = w == 0 ? acc[m][nt][r] : *dst + acc[m][nt][r];
          }
    }
  }
  __syncthreads();
  float* out = p.slabs + (long)blockIdx.x * p.Cout * p.ncols;
  for (int e = tid; e < BMC * p.ncols; e += NT_) {
    const int co = e / p.ncols, col = e - co * p.ncols;
    if (co0 + co < p.Cout) out[(long)(co0 + co) * p.ncols + col] = R[co * (NTL * 32) + col];
  }
}


struct FewCinArgs {
  const float* x; const float* w; const float* bias; float* y;
  int N, Cin, H, W, Cout, in_relu, relu, accumulate;
};

// OutT = __bf16: the output is written as bf16 (round to nearest even) for a consumer that is a bf16-mode convolution reading it with
// AGL_CONV_X_BF16 / AGL_CONV_MASK_BF16 (no accumulation onto it in that form).
// YB (with OutT = __bf16): the output is channel-blocked bf16 [N][Cout/8][H][W][8] — a thread forms 8 channels x 4 pixels at a time and writes
// four consecutive 16-byte pieces (Cout % 8 == 0).
template <int KS, int CIN, typename OutT, bool YB = false>
__global__ __launch_bounds__(256) void few_cin_fwd_k(FewCinArgs p) {
  constexpr int KK = KS * KS, PADK = KS / 2, NW = CIN * KK;
  extern __shared__ __attribute__((aligned(16))) float lw[];       // [Cout][NW] (+ bias [Cout])
  for (int i = threadIdx.x; i < p.Cout * NW; i += 256) lw[i] = p.w[i];
  float* const lb = lw + p.Cout * NW;
  for (int i = threadIdx.x; i < p.Cout; i += 256) lb[i] = p.bias ? p.bias[i] : 0.f;
  __syncthreads();
  const int QW = p.W / 4;
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  if (q >= (long)p.N * p.H * QW) return;
  const int qx = (int)(q % QW), t = (int)(q / QW), iy = t % p.H, n = t / p.H, ix0 = 4 * qx;
  const long HW = (long)p.H * p.W;
  // the input window: CIN channels x KS rows x (4 + KS - 1) columns, zero outside the map
  float xin[CIN][KS][4 + KS - 1];
#pragma unroll
  for (int c = 0; c < CIN; ++c)
#pragma unroll
    for (int r = 0; r < KS; ++r) {
      const int yy = iy - PADK + r;
      const bool rok = (unsigned)yy < (unsigned)p.H;
      const float* row = p.x + ((long)n * CIN + c) * HW + (long)(rok ? yy : 0) * p.W + ix0;
      float4 mid = rok ? *reinterpret_cast<const float4*>(row) : float4{0.f, 0.f, 0.f, 0.f};
      if (p.in_relu) { mid.x = fmaxf(mid.x, 0.f); mid.y = fmaxf(mid.y, 0.f); mid.z = fmaxf(mid.z, 0.f); mid.w = fmaxf(mid.w, 0.f); }
      xin[c][r][PADK + 0] = mid.x; xin[c][r][PADK + 1] = mid.y; xin[c][r][PADK + 2] = mid.z; xin[c][r][PADK + 3] = mid.w;
      if constexpr (KS == 3) {
        float l = (rok && ix0 > 0) ? row[-1] : 0.f, rr = (rok && ix0 + 4 < p.W) ? row[4] : 0.f;
        if (p.in_relu) { l = fmaxf(l, 0.f); rr = fmaxf(rr, 0.f); }
        xin[c][r][0] = l; xin[c][r][5] = rr;
      }
    }
  if constexpr (YB) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    for (int cg = 0; cg < (p.Cout >> 3); ++cg) {
      float a[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* wc = lw + (8 * cg + u) * NW;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
          for (int r = 0; r < KS; ++r)
#pragma unroll
            for (int k = 0; k < KS; ++k) {
              const float wv = wc[(c * KS + r) * KS + k];
              a0 = fmaf(wv, xin[c][r][k + 0], a0); a1 = fmaf(wv, xin[c][r][k + 1], a1);
              a2 = fmaf(wv, xin[c][r][k + 2], a2); a3 = fmaf(wv, xin[c][r][k + 3], a3);
            }
        const float bb = lb[8 * cg + u];
        a[u][0] = a0 + bb; a[u][1] = a1 + bb; a[u][2] = a2 + bb; a[u][3] = a3 + bb;
      }
      bf16x8_t* dst = reinterpret_cast<bf16x8_t*>(p.y) + (((long)n * (p.Cout >> 3) + cg) * p.H + iy) * p.W + ix0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bf16x8_t o;
#pragma unroll
        for (int u = 0; u < 8; ++u) o[u] = (__bf16)(p.relu ? fmaxf(a[u][j], 0.f) : a[u][j]);
        dst[j] = o;
      }
    }
    return;
  }
  OutT* const yb = reinterpret_cast<OutT*>(p.y) + (long)n * p.Cout * HW + (long)iy * p.W + ix0;
  for (int co = 0; co < p.Cout; ++co) {
    const float* wc = lw + co * NW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int r = 0; r < KS; ++r)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const float wv = wc[(c * KS + r) * KS + k];
          a0 = fmaf(wv, xin[c][r][k + 0], a0); a1 = fmaf(wv, xin[c][r][k + 1], a1);
          a2 = fmaf(wv, xin[c][r][k + 2], a2); a3 = fmaf(wv, xin[c][r][k + 3], a3);
        }
    const float bb = lb[co];
    float4 v = {a0 + bb, a1 + bb, a2 + bb, a3 + bb};
    if constexpr (sizeof(OutT) == 4) {
      float4* dst = reinterpret_cast<float4*>(yb + (long)co * HW);
      if (p.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *dst = v;
    } else {
      if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      *reinterpret_cast<bf16x4_t*>(yb + (long)co * HW) = bf16x4_t{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    }
  }
}


// nn.Linear gradients (1x1 "convolutions" on 1x1 maps: the heads and embeddings of the discriminators, the crop encoder's fc
// layers).  As GEMMs they are a few hundred rows deep and the generic tiles leave them on 1..16 workgroups walking the reduction
// serially (75 us for the 393 x 64 -> 2048 weight gradient, 100 MFLOP).  Exact fp32, fixed order.
// dw[co][ci] (+)= sum_n dy[n][co] * x[n][ci]:  a workgroup owns 64 input features x 4 output features; its four waves take a quarter of
// the rows each (lane = input feature: x is read coalesced and once for the four outputs, dy as wave-uniform values), four rows in
// flight per wave, and the partial sums are added through LDS in wave order.  (One output per thread walking all N rows alone was a
// chain of N / 4 load round trips: 43 us per launch at N = 393, 29 launches per iteration.)
// (rb: bf16 arithmetic mode — both operands are rounded to bf16 first, as every kernel of that mode does)
__global__ __launch_bounds__(256) void linear_bww_k(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, int N,
                                                    int Cin, int Cout, int in_relu, int accumulate, int rb) {
  __shared__ float part[4][4][64];
  auto op = [&](float v) { return rb ? (float)(__bf16)v : v; };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ci = blockIdx.x * 64 + lane, co0 = blockIdx.y * 4;
  const bool ok = ci < Cin;
  const float* xp = x + (ok ? ci : 0);
  const int q = (N + 3) / 4, n_lo = wave * q, n_hi = min(N, n_lo + q);
  const int nco = min(4, Cout - co0);
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  int n = n_lo;
  for (; n + 4 <= n_hi; n += 4) {
    float xv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { xv[r] = xp[(long)(n + r) * Cin]; if (in_relu) xv[r] = fmaxf(xv[r], 0.f); xv[r] = op(xv[r]); }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nco) {      // (uniform)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[k] = fmaf(op(dy[(long)(n + r) * Cout + co0 + k]), xv[r], a[k]);
      }
  }
  for (; n < n_hi; ++n) {
    float x0 = xp[(long)n * Cin];
    if (in_relu) x0 = fmaxf(x0, 0.f);
    x0 = op(x0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < nco) a[k] = fmaf(op(dy[(long)n * Cout + co0 + k]), x0, a[k]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) part[wave][k][lane] = a[k];
  __syncthreads();
  const int k = wave;                            // wave k finishes output feature co0 + k
  if (k >= nco || !ok) return;
  const float v = ((part[0][k][lane] + part[1][k][lane]) + part[2][k][lane]) + part[3][k][lane];
  float* o = dw + (long)(co0 + k) * Cin + ci;
  *o = accumulate ? *o + v : v;
}

// dx[n][ci] (+)= sum_co dy[n][co] * w[co][ci], zeroed where pos_mask <= 0.  One workgroup per (row n, 64 input features): the
// output count is small (N x Cin) and the reduction long (Cout up to 2048), so the four waves take a quarter of the output features
// each and their partial sums are added through LDS in wave order.
__global__ __launch_bounds__(256) void linear_bwd_data_k(const float* __restrict__ dy, const float* __restrict__ w, const float* __restrict__ pos_mask,
                                                         float* __restrict__ dx, int N, int Cin, int Cout, int accumulate, int rb) {
  __shared__ float part[4][64];
  auto op = [&](float v) { return rb ? (float)(__bf16)v : v; };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ci = blockIdx.x * 64 + lane, n = blockIdx.y;
  const bool ok = ci < Cin;
  const float* wp = w + (ok ? ci : 0);
  const float* dp = dy + (long)n * Cout;
  const int q = (Cout + 3) / 4, c_lo = wave * q, c_hi = min(Cout, c_lo + q);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int co = c_lo;
  for (; co + 4 <= c_hi; co += 4) {
    a0 = fmaf(op(dp[co]), op(wp[(long)co * Cin]), a0); a1 = fmaf(op(dp[co + 1]), op(wp[(long)(co + 1) * Cin]), a1);
    a2 = fmaf(op(dp[co + 2]), op(wp[(long)(co + 2) * Cin]), a2); a3 = fmaf(op(dp[co + 3]), op(wp[(long)(co + 3) * Cin]), a3);
  }
  for (; co < c_hi; ++co) a0 = fmaf(op(dp[co]), op(wp[(long)co * Cin]), a0);
  part[wave][lane] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (wave != 0 || !ok) return;
  float v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  const long o = (long)n * Cin + ci;
  if (pos_mask && !(pos_mask[o] > 0.f)) v = 0.f;
  dx[o] = accumulate ? dx[o] + v : v;
}

}  // namespace

static int few_bww_plan(const FewBwwShape& a, int* ntl, int* splits, int* tps, int* tiles) {
  if (a.Cin > 4 || a.Cout < 16 || a.stride != 1 || a.up != 0) return -1;
  if (!(a.ks == 1 || a.ks == 3 || a.ks == 5 || a.ks == 7)) return -1;
  if (!(a.OW == 32 || a.OW == 64 || a.OW == 128) || a.OH % (128 / a.OW) != 0) return -1;
  if (a.H + 2 * a.pad - a.ks + 1 != a.OH || a.W + 2 * a.pad - a.ks + 1 != a.OW || a.pad >= a.ks) return -1;
  const int ncols = a.Cin * a.ks * a.ks;
  const int n = (ncols + 31) / 32;
  if (n > 5) return -1;                                // accumulators: 32 * n registers per lane
  if ((long)a.N * a.Cout * a.OH * a.OW >= (1L << 31) || (long)a.N * a.Cin * a.H * a.W >= (1L << 31)) return -1;
  const long t = (long)a.N * (a.OH / (128 / a.OW));
  const int blocks = agl_cdiv(a.Cout, 64);
  long z = (512 + blocks - 1) / blocks;                // two workgroups per CU
  if (z > t) z = t;
  const long per = (t + z - 1) / z;
  z = (t + per - 1) / per;
  *ntl = n; *splits = (int)z; *tps = (int)per; *tiles = (int)t;
  return 0;
}

long few_bww_ws_bytes(const FewBwwShape& a) {
  int ntl, splits, tps, tiles;
  if (few_bww_plan(a, &ntl, &splits, &tps, &tiles) != 0) return 0;
  return (long)splits * a.Cout * a.Cin * a.ks * a.ks * 4;
}

int few_bww_try(const FewBwwShape& a, const float* dy, const float* x, void* ws, long ws_bytes, int* splits_out, hipStream_t st,
                const char* name, int dy_bf16, const InFold* dy_fold, int dy_relu, int dy_round_bf16) {
  int ntl, splits, tps, tiles;
  if (few_bww_plan(a, &ntl, &splits, &tps, &tiles) != 0) return -1;
  const long need = (long)splits * a.Cout * a.Cin * a.ks * a.ks * 4;
  if (!ws || ws_bytes < need) return -1;
  FewBwwArgs p;
  p.dy = dy; p.x = x; p.slabs = (float*)ws; p.N = a.N; p.Cin = a.Cin; p.H = a.H; p.W = a.W; p.Cout = a.Cout; p.OH = a.OH; p.OW = a.OW;
  p.pad = a.pad; p.in_relu = a.in_relu; p.tiles = tiles; p.tiles_per_split = tps; p.ncols = a.Cin * a.ks * a.ks; p.dy_bf16 = dy_bf16;
  p.fold = dy_fold ? *dy_fold : InFold{}; p.dy_relu = dy_relu; p.dy_round_bf16 = dy_round_bf16;
  if (p.fold.cells && (dy_bf16 || !p.fold.map || !p.fold.mean || !p.fold.scale || a.Cout % 8 != 0 || a.OW % 4 != 0 || a.ks != 7)) return -1;
  dim3 g((unsigned)splits, agl_cdiv(a.Cout, 64));
#define FEW_LAUNCH(KS_, NT2_) hipLaunchKernelGGL((few_bww_k<KS_, NT2_>), g, dim3(NT_), 0, st, p)
  if (a.ks == 1) FEW_LAUNCH(1, 1);
  else if (a.ks == 3) { if (ntl == 1) FEW_LAUNCH(3, 1); else FEW_LAUNCH(3, 2); }
  else if (a.ks == 5) { if (ntl <= 3) FEW_LAUNCH(5, 3); else FEW_LAUNCH(5, 4); }
  else if (p.fold.cells) { if (ntl <= 4) return -1; hipLaunchKernelGGL((few_bww_k<7, 5, true>), g, dim3(NT_), 0, st, p); }
  else { if (ntl <= 4) FEW_LAUNCH(7, 4); else FEW_LAUNCH(7, 5); }
#undef FEW_LAUNCH
  AGL_CHECK_LAUNCH(name);
  *splits_out = splits;
  return AGL_OK;
}

// Forward with <= 4 input channels (see few_cin_fwd_k).  -1: shape not taken.
int few_cin_fwd_try(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W, int Cout, int ks, int in_relu,
                    int relu, int accumulate, int y_bf16, hipStream_t st, const char* name) {
  if (!(ks == 1 || ks == 3) || Cin < 1 || Cin > 4 || W % 4 != 0 || (relu && accumulate) || (y_bf16 && accumulate)) return -1;
  if (y_bf16 == 2 && (Cout % 8 != 0 || ks != 3 || Cin != 3)) return -1;      // (channel-blocked bf16 output: the 3 -> C first convolution of a discriminator)
  const long lds = ((long)Cout * Cin * ks * ks + Cout) * 4;
  if (lds > 48 * 1024) return -1;
  if ((((uintptr_t)x | (uintptr_t)y) & 15) != 0) return -1;
  FewCinArgs p{x, w, bias, y, N, Cin, H, W, Cout, in_relu, relu, accumulate};
  const dim3 g((unsigned)agl_cdiv((long)N * H * (W / 4), 256));
#define FC_LAUNCH(KS_, C_)                                                                                       \
  do {                                                                                                          \
    if (y_bf16 == 2) { if (KS_ == 3 && C_ == 3) hipLaunchKernelGGL((few_cin_fwd_k<3, 3, __bf16, true>), g, dim3(256), (size_t)lds, st, p); } \
    else if (y_bf16) hipLaunchKernelGGL((few_cin_fwd_k<KS_, C_, __bf16>), g, dim3(256), (size_t)lds, st, p);   \
    else hipLaunchKernelGGL((few_cin_fwd_k<KS_, C_, float>), g, dim3(256), (size_t)lds, st, p);               \
  } while (0)
  if (ks == 1) { if (Cin == 1) FC_LAUNCH(1, 1); else if (Cin == 2) FC_LAUNCH(1, 2); else if (Cin == 3) FC_LAUNCH(1, 3); else FC_LAUNCH(1, 4); }
  else { if (Cin == 1) FC_LAUNCH(3, 1); else if (Cin == 2) FC_LAUNCH(3, 2); else if (Cin == 3) FC_LAUNCH(3, 3); else FC_LAUNCH(3, 4); }
#undef FC_LAUNCH
  AGL_CHECK_LAUNCH(name);
  return AGL_OK;
}

int linear_bww_launch(const float* dy, const float* x, float* dw, int N, int Cin, int Cout, int in_relu, int accumulate, int round_bf16,
                      hipStream_t st, const char* name) {
  hipLaunchKernelGGL(linear_bww_k, dim3(agl_cdiv(Cin, 64), agl_cdiv(Cout, 4)), dim3(256), 0, st, dy, x, dw, N, Cin, Cout, in_relu, accumulate,
                     round_bf16);
  AGL_CHECK_LAUNCH(name);
  return AGL_OK;
}

int linear_bwd_data_launch(const float* dy, const float* w, const float* pos_mask, float* dx, int N, int Cin, int Cout, int accumulate,
                           int round_bf16, hipStream_t st, const char* name) {
  hipLaunchKernelGGL(linear_bwd_data_k, dim3(agl_cdiv(Cin, 64), N), dim3(256), 0, st, dy, w, pos_mask, dx, N, Cin, Cout, accumulate,
                     round_bf16);
  AGL_CHECK_LAUNCH(name);
  return AGL_OK;
}
