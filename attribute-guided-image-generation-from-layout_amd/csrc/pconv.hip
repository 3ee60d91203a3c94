// LDS-patch convolution on the bf16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16, fp32 accumulation).
//
// Replaces, for stride-1 3x3 / 5x5 convolutions on power-of-two maps, the im2col kernel of conv.hip in the two
// configurations where that kernel is bound by its 4-byte gathers rather than by the matrix pipe:
//   nsplit 1  (AGL_CONV_BF16, BASELINE configs 3/5): operands rounded to bf16 (RNE) once, when they are staged;
//   nsplit 3  (AGL_CONV_SPLIT3): fp32 operands carried as THREE bf16 terms each (a = a1 + a2 + a3 exactly, to 2^-27
//             relative) and multiplied as the six products a1b1, a1b2, a2b1, a2b2, a1b3, a3b1 (the dropped ones are
//             <= 2^-27 |ab|), accumulated in fp32 — fp32-accurate products at 6/16 of the fp32-MFMA cycle cost.
//
// Structure (one 256-thread workgroup = 2x2 waves, output tile BM channels x 128 pixels):
//   * the weights are re-packed once per call (pack_weights_k) into bf16 [plane][channel chunk][half][tap][m][8], so
//     a workgroup's slice for one chunk of 16 input channels is a set of contiguous 16-byte pieces that go to LDS
//     unchanged ([plane][half][tap][row]: the 32x32x16 A fragment of a lane is ONE conflict-free ds_read_b128);
//   * the input patch TI x (TH+ks-1) x (TW+ks-1) of the chunk is fetched ONCE from the fp32 NCHW tensor (each element
//     is used ks^2 times from LDS instead of being re-gathered ks^2 times), converted, and stored channel-fastest as
//     [plane][half][pixel] 16-byte pieces: the B fragment of a lane for tap (kh,kw) is ONE ds_read_b128 at
//     pixel(lane) + kh*pitch + kw.  Row pitches are padded so that every 16-lane read group hits 16 distinct
//     16-byte LDS slots (checked offline for each geometry);
//   * K order = (tap, channel): one MFMA K-step = 16 channels of one tap, matching the lane map
//     A[row][k = 8*(lane>>5) + j], B[k][col] of v_mfma_f32_32x32x16_bf16.
// The input-gradient of a "same" convolution runs on the same kernel through the packed weights (flipped taps,
// channel roles swapped).  Epilogues as in conv.hip (bias, ReLU-mask of the consumer, accumulate, ReLU).
#include "pconv.h"
#include "spade.h"
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Planes of the split (fp32-accurate) arithmetic, AGL_CONV_SPLIT3 at the ABI:
//   3  three bf16 terms per operand, six products per multiply-add (rounds 2-4);
//   2  fp16 hi / lo terms, THREE products (hi*hi into the main accumulator, hi*lo' + lo'*hi into a second one that is scaled back by
//      2^-11 once): a = hi + lo' * 2^-11 with hi = fp16(a * 2^-s), lo' = fp16((a * 2^-s - hi) * 2^11) — 22-23 significant bits; the
//      dropped lo*lo term is <= 2^-22 |ab| (rms 2^-24.6).  fp16 has 5 exponent bits, so every staged operand block carries a
//      power-of-two scale: weights per 16-channel chunk (found when they are packed, stored behind the packed planes), activations
//      per staged chunk / tile of a workgroup (absolute maximum of the registers it is about to convert, reduced over the workgroup);
//      the accumulators live at a running scale that only grows (a rescale multiplies them by a power of two: exact).
#ifndef AGL_SPLIT_PLANES
#define AGL_SPLIT_PLANES 2
#endif
constexpr int SPL = AGL_SPLIT_PLANES;
static_assert(SPL == 2 || SPL == 3, "split arithmetic: 2 (fp16 hi/lo, three products) or 3 (bf16 terms, six products) planes");
int pconv_split_products() { return SPL == 2 ? 3 : 6; }
// taps per weight stage of the 64-channel x 256-pixel 3x3 split tiles: one kernel row (3) keeps three planes within two workgroups per
// CU; two fp16 planes have room for the whole window (9: a third of the barriers)
#ifndef AGL_SPLIT_WIDE_TG
#define AGL_SPLIT_WIDE_TG 3
#endif
constexpr int STG = AGL_SPLIT_WIDE_TG;
#ifndef AGL_H16_PHASE_LONG
#define AGL_H16_PHASE_LONG 32      // (chunk, tap) steps above which the stride-2 input gradient takes its 64-pixel tiles (pconvT_plan)
#endif

namespace {

constexpr int NT = 256;
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int H16_TOP = 14;      // a scaled block's absolute maximum lies in [2^14, 2^15): below fp16's 65504 with a binade to spare
constexpr int H16_SMAX = 120;    // shifts are clamped to +-120 so that 2^-shift is a normal fp32

// 2^e as a float, e in [-126, 127]
__device__ __forceinline__ float exp2i(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }
// shift s of a block whose largest magnitude has the bits `mbits` (sign cleared): block * 2^-s has its maximum in [2^14, 2^15)
__device__ __forceinline__ int h16_shift(unsigned mbits) {
  const int s = (int)(mbits >> 23) - 127 - H16_TOP;
  return s < -H16_SMAX ? -H16_SMAX : (s > H16_SMAX ? H16_SMAX : s);
}
// v (already scaled) -> fp16 hi, fp16 lo' = (v - hi) * 2^11 (v - hi is exact in fp32)
__device__ __forceinline__ void split_h16(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)((v - (float)hi) * 2048.0f);
}
// ... with lo at its true scale (one accumulator for all three products).  An element more than 2^-18 below its block's maximum then has a
// subnormal lo — an absolute error <= 2^-40 of the block maximum, far below the accumulation's own rounding.
__device__ __forceinline__ void split_h16_unscaled(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}
// maximum over the wave of an unsigned value (DPP within rows of 16 lanes, then the four row values through the scalar unit)
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));    // row_half_mirror
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));    // row_mirror
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}

struct PArgs {
  const float* x; const u32x4* wp; const float* bias; const float* pos_mask; float* y;
  int N, Cin, H, W, Cout, OH, OW, pad, up, in_relu, relu, accumulate;
  int nch, mpad;
  unsigned x_bytes;
  int x_bf16;                // the input tensor is stored as bf16 (2-byte elements: a tensor only convolutions read, written by its producer in bf16)
  int mask_bf16;             // pos_mask points to bf16 elements (plain epilogue without a reduction split only)
  int xgy, xchunk, xitems;   // XCD-aware workgroup order (xchunk > 0): the grid is 1-D, see pconv_k
  float* stats;      // optional: per-channel (sum, sum of squares) of the stored outputs, one row per (pixel tile, wave column)
  float* slabs;      // optional reduction split: blockIdx.z takes `cps` channel chunks and writes its raw partial output to slab z
  int cps;           // (each slab is shaped like y; splitk_epilogue of conv.hip adds them and applies the epilogue)
  long out_numel;
  int prio;          // raise the wave priority for the conversion / LDS-store segment of every stage (A/B switch)
  int oh2, ow2;      // phase mode: extent of the output map (2*OH x 2*OW, or one more row and column for an odd-sized input)
  const float* odiv; // optional device scalar: the products are divided by it before bias / mask / ReLU (pre-packed weights of a
                     // spectrally normalised layer: packed W_orig, divisor sigma)
  InFold fold;       // optional input transform of the staged patch (pconv.h): the normalise-modulate of the BatchNorm that reads x
  int y_bf16;        // y points to bf16 elements (plain epilogue only: no phases, no reduction split)
  int blk;           // (host bookkeeping) FEAT 64 / 128: x / y are channel-blocked bf16 tensors [N][C/8][H][W][8]
  int mask_blk;      // pos_mask (bf16, mask_bf16 set) is channel-blocked: element (n, m, y, x) at (((n * Cout/8 + m/8) * H + y) * W + x) * 8 + m%8
  const float* addend;   // optional fp32 tensor shaped like y, added to the result before the output ReLU (out-of-place accumulate)
  // optional few-channel 1x1 shortcut evaluated in the epilogue (FEAT 16): y += sc_b[m] + sum_c sc_w[m * sc_cin + c] * sc_x[img][c][pixel],
  // sc_x an (N, sc_cin <= 4, OH, OW) fp32 tensor — the learnable shortcut of the discriminators' first block (discriminator.py:43-44, :58-60)
  const float* sc_x; const float* sc_w; const float* sc_b; int sc_cin;
  const int* wexp;       // NSPL 2 (fp16 hi / lo planes): shift of every 16-channel chunk of the packed weights (behind the planes)
};

// a = t0 + t1 + t2 with bf16 terms (each step's remainder is exact in fp32)
__device__ __forceinline__ void split3(float a, __bf16& t0, __bf16& t1, __bf16& t2) {
  t0 = (__bf16)a;
  const float r1 = a - (float)t0;
  t1 = (__bf16)r1;
  const float r2 = r1 - (float)t1;
  t2 = (__bf16)r2;
}

// Weight re-pack: wp[plane][cc][h][tap][m][j] = term_plane( w[m*w_sm + (16cc + 8h + j)*w_sc + tap'] ), zero for m >= M
__device__ __forceinline__ void pack_piece(const float* __restrict__ w, u32x4* __restrict__ wp, int M, int Cin, int KK, int w_sm, int w_sc,
                                           int flip, int mpad, int nsplit, int phase4, long per_plane, long i) {
  const int m = (int)(i % mpad);
  long r = i / mpad;
  const int tap = (int)(r % KK); r /= KK;
  const int h = (int)(r & 1);
  const int cc = (int)(r >> 1);
  int st = flip ? KK - 1 - tap : tap;
  if (phase4) {      // tap = phase*4 + th'*2 + tw' of the 4x4 / stride-2 / pad-1 input gradient: phase (ph,pw) of the output uses the
    const int phase = tap >> 2, thp = (tap >> 1) & 1, twp = tap & 1;   // window rows kh = kh0 + 2*th, kh0 = (ph+1)&1, th = 1 - th'
    const int kh = (((phase >> 1) + 1) & 1) + 2 * (1 - thp), kw = (((phase & 1) + 1) & 1) + 2 * (1 - twp);
    st = kh * 4 + kw;
  }
  bf16x8 t0, t1, t2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 16 * cc + 8 * h + j;
    const float v = (m < M && c < Cin) ? w[(long)m * w_sm + (long)c * w_sc + st] : 0.f;
    if (nsplit == 1) { t0[j] = (__bf16)v; }
    else { __bf16 a, b, d; split3(v, a, b, d); t0[j] = a; t1[j] = b; t2[j] = d; }
  }
  wp[i] = __builtin_bit_cast(u32x4, t0);
  if (nsplit == 3) {
    wp[per_plane + i] = __builtin_bit_cast(u32x4, t1);
    wp[2 * per_plane + i] = __builtin_bit_cast(u32x4, t2);
  }
}
// fp16 hi / lo form (SPL == 2).  A 16-channel chunk cc of the packed tensor shares one scale, so a pack is two launches: h16_scan_chunk
// (one 1024-thread workgroup per chunk reads the chunk's 16 * M * KK source floats in runs of consecutive addresses, eight loads in flight
// per thread, and leaves the shift in wexp[cc] behind the planes, where the consumer reads it as a scalar per stage) and then one thread
// per 16-byte piece as in pack_piece: wp[plane][i] = {hi, lo'} of w * 2^-shift.
constexpr int H16_SCAN_G = 8;      // workgroups that share the scan of one chunk (slices of its rows); partial maxima behind the shifts
__device__ __forceinline__ void h16_scan_chunk(const float* __restrict__ w, int* __restrict__ wexp, int M, int Cin, int KK, int w_sm, int w_sc,
                                               int cc, int g, unsigned* red) {
  const int cn = min(16, Cin - 16 * cc);                   // channels of this chunk that exist
  const int nch = Cin / 16, nchp = (nch + 3) & ~3;
  const int m_lo = (int)((long)M * g / H16_SCAN_G), m_hi = (int)((long)M * (g + 1) / H16_SCAN_G);      // this workgroup's rows
  w += (long)m_lo * w_sm;
  M = m_hi - m_lo;
  // (32-bit index arithmetic: a chunk slice has at most 16 * 2048 * 49 elements.  With 64-bit division in the address of every load the
  //  kernel spilled 228 registers under its 1024-thread bound and took 274 us for an arena)
  const int total = cn * M * KK;
  const float* const wc = w + (long)(16 * cc) * w_sc;
  const bool m_inner = w_sm < w_sc;                         // which of (row m, channel c) continues the run of taps
  const int inner = m_inner ? M : cn;
  auto at = [&](int idx) -> unsigned {
    const int r = idx / KK, st = idx - r * KK;
    const int hi = r / inner, lo = r - hi * inner;
    const int m = m_inner ? lo : hi, c = m_inner ? hi : lo;
    return __builtin_bit_cast(unsigned, wc[(long)m * w_sm + (long)c * w_sc + st]) & 0x7fffffffu;
  };
  unsigned tm = 0;
  const int step = blockDim.x;
  int idx = threadIdx.x;
  for (; idx + 3 * step < total; idx += 4 * step) {
    unsigned v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = at(idx + q * step);
#pragma unroll
    for (int q = 0; q < 4; ++q) tm = max(tm, v[q]);
  }
  for (; idx < total; idx += step) tm = max(tm, at(idx));
  tm = wave_umax(tm);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tm;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned mx = 0;
    for (int q = 0; q < (int)(blockDim.x >> 6); ++q) mx = max(mx, red[q]);
    reinterpret_cast<unsigned*>(wexp)[nchp + cc * H16_SCAN_G + g] = mx;      // partial maximum (h16_chunk_shift combines them)
  }
}
// shift of chunk cc from the partial maxima (every reader computes the same value)
__device__ __forceinline__ int h16_chunk_shift(const int* __restrict__ wexp, int nch, int cc) {
  const unsigned* pm = reinterpret_cast<const unsigned*>(wexp) + ((nch + 3) & ~3) + cc * H16_SCAN_G;
  unsigned mx = 0;
#pragma unroll
  for (int q = 0; q < H16_SCAN_G; ++q) mx = max(mx, pm[q]);
  return h16_shift(mx);
}
__device__ __forceinline__ void pack_piece_h16(const float* __restrict__ w, u32x4* __restrict__ wp, int M, int Cin, int KK, int w_sm, int w_sc,
                                               int flip, int mpad, int phase4, long per_plane, long i) {
  const int m = (int)(i % mpad);
  long r = i / mpad;
  const int tap = (int)(r % KK); r /= KK;
  const int cc = (int)(r >> 1), c0 = 16 * cc + 8 * (int)(r & 1);
  int st = flip ? KK - 1 - tap : tap;
  if (phase4) {
    const int phase = tap >> 2, thp = (tap >> 1) & 1, twp = tap & 1;
    st = ((((phase >> 1) + 1) & 1) + 2 * (1 - thp)) * 4 + (((phase & 1) + 1) & 1) + 2 * (1 - twp);
  }
  int* const wexp = reinterpret_cast<int*>(wp + 2 * per_plane);
  const int nch = (Cin + 15) / 16, sh = h16_chunk_shift(wexp, nch, cc);
  if (m == 0 && tap == 0 && (r & 1) == 0) {      // one piece per chunk publishes the shift the consumers read (wexp[cc]; padding defined)
    wexp[cc] = sh;
    if (cc == nch - 1)
      for (int q = nch; q < ((nch + 3) & ~3); ++q) wexp[q] = 0;
  }
  const float fac = exp2i(-sh);
  f16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = (m < M && c0 + j < Cin) ? w[(long)m * w_sm + (long)(c0 + j) * w_sc + st] * fac : 0.f;
    _Float16 a, b;
    split_h16(v, a, b);
    hi[j] = a; lo[j] = b;
  }
  wp[i] = __builtin_bit_cast(u32x4, hi);
  wp[per_plane + i] = __builtin_bit_cast(u32x4, lo);
}
__global__ __launch_bounds__(1024) void h16_scan_k(const float* __restrict__ w, u32x4* __restrict__ wp, int M, int Cin, int KK, int w_sm, int w_sc,
                                                   long per_plane) {
  __shared__ unsigned red[16];
  h16_scan_chunk(w, reinterpret_cast<int*>(wp + 2 * per_plane), M, Cin, KK, w_sm, w_sc, (int)blockIdx.x, (int)blockIdx.y, red);
}
// ... for a descriptor table (pack_many_k): blockIdx.y = row, blockIdx.x strides over the row's chunks
__global__ __launch_bounds__(1024) void h16_scan_many_k(const long long* __restrict__ d) {
  __shared__ unsigned red[16];
  const long long* r = d + (long)blockIdx.y * 14;
  if (r[9] != 2) return;
  const int nch = (int)r[3] / 16;
  for (int cc = blockIdx.x; cc < nch; cc += gridDim.x)
    h16_scan_chunk(reinterpret_cast<const float*>(r[0]), reinterpret_cast<int*>(reinterpret_cast<u32x4*>(r[1]) + 2 * r[11]), (int)r[2], (int)r[3], (int)r[4],
                   (int)r[5], (int)r[6], cc, (int)blockIdx.z, red);
}
__global__ void pack_weights_h16_k(const float* __restrict__ w, u32x4* __restrict__ wp, int M, int Cin, int KK, int w_sm, int w_sc, int flip,
                                   int mpad, int nch, int phase4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per_plane = (long)nch * 2 * KK * mpad;
  if (i >= per_plane) return;
  pack_piece_h16(w, wp, M, Cin, KK, w_sm, w_sc, flip, mpad, phase4, per_plane, i);
}
__global__ void pack_weights_k(const float* __restrict__ w, u32x4* __restrict__ wp, int M, int Cin, int KK, int w_sm, int w_sc,
                               int flip, int mpad, int nch, int nsplit, int phase4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per_plane = (long)nch * 2 * KK * mpad;
  if (i >= per_plane) return;
  pack_piece(w, wp, M, Cin, KK, w_sm, w_sc, flip, mpad, nsplit, phase4, per_plane, i);
}
// All packed forms of a parameter arena in ONE launch (after the optimiser step): d = n rows of AGL_PACK_DESC_WORDS 64-bit words
// {w, wp, M, Cin, KK, w_sm, w_sc, flip, mpad, nsplit, phase4, per_plane, first block, blocks} (pconv_pack_desc); a block finds its
// row by bisection over the first-block column.
__global__ void pack_many_k(const long long* __restrict__ d, int n) {
  constexpr int WD = 14;
  int lo = 0, hi = n - 1;
  const long b = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (d[(long)mid * WD + 12] <= b) lo = mid; else hi = mid - 1; }
  const long long* r = d + (long)lo * WD;
  const long i = (b - r[12]) * blockDim.x + threadIdx.x, per_plane = r[11];
  if (i >= per_plane) return;
  if (r[9] == 2) {      // fp16 hi / lo planes (the chunk shifts were left by h16_scan_many_k)
    pack_piece_h16(reinterpret_cast<const float*>(r[0]), reinterpret_cast<u32x4*>(r[1]), (int)r[2], (int)r[3], (int)r[4], (int)r[5], (int)r[6], (int)r[7],
                   (int)r[8], (int)r[10], per_plane, i);
    return;
  }
  pack_piece(reinterpret_cast<const float*>(r[0]), reinterpret_cast<u32x4*>(r[1]), (int)r[2], (int)r[3], (int)r[4], (int)r[5], (int)r[6], (int)r[7],
             (int)r[8], (int)r[9], (int)r[10], per_plane, i);
}

// LDS pixel pitches that make every ds_read_b128 lane group of the B fragments hit 16 distinct 16-byte slots
// (enumerated offline over the 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} of every 32-pixel sub-tile).
// The three-plane (split) mode trades the padding for LDS space — natural pitches, 2-way conflicts on the B reads of the
// 16- and 8-wide tiles — so that two workgroups fit a CU (<= 80 KB each) and their waves cover each other's stalls.
template <int TW, int KS, int NSPL> struct Pitch;
template <int KS, int NSPL> struct Pitch<32, KS, NSPL> { static constexpr int PWP = 32 + KS - 1; static constexpr int IMG_EXTRA = 0; };
template <int KS> struct Pitch<16, KS, 1> { static constexpr int PWP = 32; static constexpr int IMG_EXTRA = 0; };
template <int KS> struct Pitch<16, KS, 3> { static constexpr int PWP = 16 + KS - 1; static constexpr int IMG_EXTRA = 0; };
template <int KS> struct Pitch<8, KS, 1> { static constexpr int PWP = 24; static constexpr int IMG_EXTRA = 0; };
template <int KS> struct Pitch<8, KS, 3> { static constexpr int PWP = 12; static constexpr int IMG_EXTRA = 0; };
template <int NSPL> struct Pitch<4, 3, NSPL> { static constexpr int PWP = 8; static constexpr int IMG_EXTRA = 4; };
template <int NSPL> struct Pitch<4, 2, NSPL> { static constexpr int PWP = 8; static constexpr int IMG_EXTRA = 4; };
template <int NSPL> struct Pitch<4, 1, NSPL> { static constexpr int PWP = 4; static constexpr int IMG_EXTRA = 0; };   // 1x1: pixels are consecutive pieces
template <> struct Pitch<2, 2, 1> { static constexpr int PWP = 6; static constexpr int IMG_EXTRA = 2; };
template <> struct Pitch<2, 2, 3> { static constexpr int PWP = 3; static constexpr int IMG_EXTRA = 0; };   // 2-way, fits two workgroups
// (two fp16 planes leave room for the conflict-free pitches of the one-plane mode)
template <int KS> struct Pitch<16, KS, 2> : Pitch<16, KS, 1> {};
template <int KS> struct Pitch<8, KS, 2> : Pitch<8, KS, 1> {};
template <> struct Pitch<2, 2, 2> : Pitch<2, 2, 1> {};

// Stride 2: the patch is split by column parity ([half][parity][row][column/2]) so that a lane group again reads 16 distinct slots;
// PWH = pitch of a parity plane row.
template <int TW, int KS> struct Pitch2;
template <int KS> struct Pitch2<32, KS> { static constexpr int PWP = 33; static constexpr int IMG_EXTRA = 0; };
template <int KS> struct Pitch2<16, KS> { static constexpr int PWP = 24; static constexpr int IMG_EXTRA = 0; };
template <int KS> struct Pitch2<8, KS> { static constexpr int PWP = 12; static constexpr int IMG_EXTRA = 0; };
template <> struct Pitch2<4, 4> { static constexpr int PWP = 6; static constexpr int IMG_EXTRA = 4; };
template <> struct Pitch2<4, 3> { static constexpr int PWP = 6; static constexpr int IMG_EXTRA = 10; };
template <int KS> struct Pitch2<2, KS> { static constexpr int PWP = 3; static constexpr int IMG_EXTRA = 0; };   // 2x2 outputs (natural pitch)
template <int S, int TW, int KS, int NSPL> struct PitchSel { using type = Pitch<TW, KS, NSPL>; };
template <int TW, int KS, int NSPL> struct PitchSel<2, TW, KS, NSPL> { using type = Pitch2<TW, KS>; };

constexpr unsigned OOB31 = 0x80000000u;   // voffset of an out-of-window element: past every tensor (< 2 GB), and adding the
                                          // (< 2 GB) scalar chunk offset cannot wrap back into range

// TG: taps per weight stage (KS*KS: the whole window; KS: one kernel row at a time — keeps the 5x5 weight slice in LDS small)
// Output tile: BM channels x (TI*TH*TW = 128 or 256) pixels; 2x2 waves, each BM/2 x BN/2.
// PHS: the four stride phases of a 4x4 / stride-2 / pad-1 input gradient (= ConvTranspose2d(4,2,1) forward).  Output phase
// (ph,pw) = blockIdx.z is a 2x2-tap stride-1 correlation of dy with pads (1-ph, 1-pw) and its own quarter of the packed weights;
// its pixels (a,b) land at (2a+ph, 2b+pw) of the twice-as-large output map.
// NTH: threads per workgroup.  256 = 2x2 waves; 512 = 2x4 waves, each owning half as many pixels: twice the waves per SIMD for the
// same LDS footprint (two workgroups per CU either way), so that a wave waiting for its LDS fragments, a barrier or its staging
// loads leaves the matrix pipe to three others instead of one.
// ABL (diagnostic builds only, results are wrong), a bit mask: 1 no MFMAs, 2 no operand conversion (one rounding, the other planes
// are copies), 4 no LDS fragment reads inside the tap loop (tap 0's fragments for every tap), 8 no global loads after the first
// stage, 16 no LDS stores after the first stage, 32 no output stores (plain epilogue)
// VERT: a KS x 1 window (vertical taps only, no horizontal padding): the first half of the few-channel 7x7 layers (pconv_vert_try)
// FEAT (bit mask): features compiled into an instantiation — every one costs registers in ALL its launches (82 of 224 kernels lost a
// workgroup per CU when they were runtime switches), so only the shapes that use them are instantiated with them:
//   1  input transform of the staged patch (PArgs::fold): the 4x4 / stride-2 family (crop / layout / global encoder), and — for SPADE in
//      front of the 128 px decoder's c6 / c7 — the bf16 5x5 forms and the bf16 vertical 7 x 1 form
//   2  BatchNorm partial rows of the output (PArgs::stats): the same family, and the 5x5 bf16 forms (decoder c6)
//   4  bf16 output store / out-of-place addend (PArgs::y_bf16, addend)
//   8  bf16 pos_mask in the paired-phase (stride-2 input gradient) epilogue
//  16  few-channel 1x1 shortcut added in the plain epilogue (PArgs::sc_x; the discriminators' first block)
template <int KS, int S, int TW, int TH, int TI, int BM, int NSPL, int TG, bool PHS = false, bool DB = false, int NTH = 256, int ABL = 0,
          bool VERT = false, int FEAT = 0>
__global__ __launch_bounds__(NTH, NTH / 128) void pconv_k(PArgs p) {
  constexpr bool F_FOLD = (FEAT & 1) != 0, F_STATS = (FEAT & 2) != 0, F_YOUT = (FEAT & 4) != 0;
  constexpr bool F_SC = (FEAT & 16) != 0;          // 16: few-channel 1x1 shortcut added in the plain epilogue (PArgs::sc_x)
  constexpr bool F_PMASK16 = (FEAT & 8) != 0;      // 8: bf16 pos_mask in the paired-phase epilogue (the bf16-stored h of a discriminator block)
  // 64: channel-blocked bf16 tensors, x and y as [N][C/8][H][W][8] (PROTOTYPE, one family: DESIGN §3.3).  A staged piece — the 8 channels
  // of a pixel — is ONE aligned 16-byte load that goes to LDS as it is (no conversion, no per-channel element loads: a 16-channel chunk
  // of the 34 x 10 patch is 100 cache-line visits instead of 320), and the epilogue stores 8 channels of a pixel as one 16-byte piece.
  constexpr bool BLK = (FEAT & 64) != 0;       // x is channel-blocked bf16 (XBLK)
  // 128: y is written channel-blocked bf16 [N][Cout/8][OH][OW][8] straight from the accumulators (half a piece per lane, no LDS transpose),
  // with the optional fp32 NCHW addend / the few-channel shortcut (FEAT 16) / bias / ReLU applied before the one rounding
  constexpr bool YBLK = (FEAT & 128) != 0;
  // 256: channel-blocked bf16 pos_mask in the plain epilogue, applied to the accumulators before the transpose (the paired-phase kernels
  // have it with FEAT 8).  Its own instantiations: in the featureless 3x3 kernel the code cost registers and a resident workgroup.
  constexpr bool F_MBLK = (FEAT & 256) != 0;
  static_assert(!(BLK || YBLK) || (NSPL == 1 && !PHS && !VERT), "blocked layout: bf16 arithmetic, plain (non-phase) forms");
  constexpr int KSW = VERT ? 1 : KS;           // window columns
  static_assert(!VERT || (S == 1 && !PHS && TG == KS), "vertical window");
  constexpr int NT = NTH;                      // (shadows the file-level 256)
  constexpr int WNW = NTH / 128;               // wave columns (pixel direction); 2 wave rows (channel direction)
  static_assert(!PHS || (KS == 2 && S == 1 && TG == 4), "phase mode");
  // PAIR (phase mode, tiles >= 4 wide): one workgroup evaluates BOTH column phases pw = 0, 1 of its row phase — they share the
  // dy patch (one more column) and their outputs interleave to 16-byte stores of the twice-as-wide row instead of stride-2
  // 4-byte stores from two workgroups (which also made every output line a partial write: 1.85x the bytes at the HBM)
  constexpr bool PAIR = PHS && TW >= 4;
  constexpr int NPW = PAIR ? 2 : 1, TGA = PAIR ? 2 * TG : TG;      // column phases per workgroup; taps staged per stage
  constexpr int BN = TI * TH * TW, KK = KS * KSW, NTG = PHS ? 1 : KK / TG, KKW = PHS ? 16 : KK;
  static_assert((BN == 64 || BN == 128 || BN == 256) && (TG == KK || TG == KS) && (S == 1 || S == 2), "pconv geometry");
  constexpr int KSX = PAIR ? 3 : KSW;          // patch columns per output column
  constexpr int PH = S * (TH - 1) + KS, PW = S * (TW - 1) + KSX;
  using PitchT = typename PitchSel<S, TW, KSX, NSPL>::type;
  constexpr int PWP = PitchT::PWP;             // row pitch (of a parity plane when S == 2)
  static_assert(S * PWP >= PW, "pitch");
  constexpr int IMGP = PH * PWP + PitchT::IMG_EXTRA;
  constexpr int PAR = TI * IMGP;               // pieces per parity plane
  constexpr int NQ = S * PAR;
  constexpr int P_PLANE = 2 * NQ;              // 16-byte pieces per plane: [h][q]
  constexpr int A_PLANE = 2 * TGA * BM;        //                           [h][t][row]
  constexpr int NB = 2 * TI * PH * PW, BR = (NB + NT - 1) / NT;
  constexpr int NA = NSPL * A_PLANE, AR = (NA + NT - 1) / NT;
  constexpr int WTM = BM / 64, WTN = BN / (32 * WNW);   // 32x32 accumulator tiles per wave (a wave covers BM/2 channels x BN/WNW pixels)
  static_assert(WTN >= 1, "tile too small for this many waves");
  constexpr int NACC = NSPL >= 2 ? 2 : 1;      // split modes: the small products go to their own accumulator
  constexpr bool H16 = NSPL == 2;              // fp16 hi / lo planes with dynamic power-of-two scales (see SPL)
  constexpr int EP_PITCH = 36;                 // floats per row of the epilogue transpose tile (16-byte aligned rows)
  // DB: double buffering (where it does not cost a workgroup per CU).  With a second weight buffer the stores of stage s+1 need
  // no barrier of their own (one barrier per stage); a second patch buffer does the same for the once-per-chunk patch stores.
  // Measured per geometry (tools/conv_bench.py): -16..18 % on the 4x4-map 3x3 layers and -2..3 % on the 5x5 ConvLSTM layers in
  // split mode; +3..8 % (slower) on the larger 3x3 tiles, where it forces one kernel row per weight stage, and on the phase
  // kernels — so it is a per-instantiation choice of the dispatch, not a default.
  constexpr int PB = NSPL * P_PLANE, AB = NSPL * A_PLANE;          // 16-byte pieces per patch / weight buffer
  constexpr int LDS_CAP = (PB + AB <= 5120) ? 5120 : 10240;       // keep two workgroups per CU (80 KB each) when one buffer pair fits that
  constexpr bool DBA = DB && PB + 2 * AB <= LDS_CAP, DBB = DBA && 2 * PB + 2 * AB <= LDS_CAP;
  constexpr int STAGE_PIECES = (DBB ? 2 : 1) * PB + (DBA ? 2 : 1) * AB, EP_PIECES = (NTH / 64) * 32 * EP_PITCH / 4;   // 16-byte pieces
  __shared__ u32x4 lds[STAGE_PIECES > EP_PIECES ? STAGE_PIECES : EP_PIECES];
  __shared__ unsigned wmax_s[H16 ? NTH / 64 : 1];      // H16: the waves' maxima of the chunk about to be converted
  constexpr bool F_SPADE = F_FOLD && S == 1;      // the SPADE form of the transform (pconv.h): the stride-1 instantiations that carry FEAT 1
  __shared__ __attribute__((aligned(16))) float fmr[F_SPADE ? 1024 : 1];      // SPADE: per-channel mean [0, 512) and rstd [512, 1024) (Cin <= 512)
  u32x4* const Pl = lds;
  u32x4* const Al = lds + (DBB ? 2 : 1) * PB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm = wave / WNW, wn = wave % WNW;
  const int Hl = p.H << p.up, Wl = p.W << p.up;
  // Workgroup -> (pixel tile bx, channel tile by).  Workgroups are dealt round-robin over the 8 XCDs (linear ids L and L + 8 share an
  // XCD and its L2).  With the plain grid, vertically adjacent pixel tiles (which share two of their ten patch rows) and the channel
  // tiles of one pixel tile (which read the same patch) land on different XCDs, and every one of them fetches the shared rows from
  // memory again.  XCD-aware order (p.xchunk > 0, 1-D grid of 8 * xchunk): XCD k takes the contiguous run of items
  // [k * xchunk, (k + 1) * xchunk), item = pixel tile * gy + channel tile — neighbours in space are neighbours in time on one L2.
  // Phase kernels: the two / four phases of a pixel tile read the same dy patch and join the item index (fastest), the
  // reduction split stays in blockIdx.z.
  constexpr int NZP = PHS ? (PAIR ? 2 : 4) : 1;                  // phase values carried by blockIdx.z (the reduction split index is above)
  int bx = blockIdx.x, by = blockIdx.y;
  int zsplit = (int)blockIdx.z / NZP, zph = (int)blockIdx.z - zsplit * NZP;
  if (p.xchunk > 0) {
    int item = (int)(blockIdx.x & 7) * p.xchunk + (int)(blockIdx.x >> 3);
    if (item >= p.xitems) return;
    if constexpr (NZP > 1) { zsplit = blockIdx.z; zph = item % NZP; item /= NZP; }
    bx = item / p.xgy; by = item - bx * p.xgy;
  }
  int img0, ty0, tx0;
  if constexpr (TI == 1) {
    const int tpr = p.OW / TW, tpi = (p.OH / TH) * tpr;
    img0 = bx / tpi;
    const int t = bx - img0 * tpi;
    ty0 = (t / tpr) * TH; tx0 = (t % tpr) * TW;
  } else {
    img0 = bx * TI; ty0 = 0; tx0 = 0;
  }
  const int bm0 = by * BM;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const int phase = PHS ? (PAIR ? 2 * zph : zph) : 0, ph_y = phase >> 1, ph_x = phase & 1;
  const int pad_y = PHS ? 1 - ph_y : p.pad, pad_x = PHS ? (PAIR ? 1 : 1 - ph_x) : (VERT ? 0 : p.pad);

  // ---- per-thread constants of the two staging passes (everything that does not depend on the stage is computed once;
  // the stage term is wave-uniform and travels in the scalar offset of the loads: no per-load vector arithmetic)
  const unsigned esz = p.x_bf16 ? 2u : 4u;      // bytes per stored input element
  unsigned bsrc[BR];   // patch item e = (half h, image ti, row yy, column xx): byte offset of channel 8h, or OOB31
  int bdst[BR];        // LDS piece index h*NQ + q, or -1
  int faff[F_FOLD ? BR : 1];
  int fcell[F_FOLD ? BR : 1];      // SPADE form of the transform: float offset of the item's cell in channel group 0's plane of p.fold.cells
  const int fG2 = F_FOLD && p.fold.cells ? p.fold.G * p.fold.G : 0;
  if constexpr (F_SPADE) {
    if (p.fold.cells) {      // (uniform) the per-channel statistics, once per workgroup
      for (int i = tid; i < p.Cin; i += NTH) { fmr[i] = p.fold.mean[i]; fmr[512 + i] = p.fold.scale[i]; }
      __syncthreads();
    }
  }
#pragma unroll
  for (int r = 0; r < BR; ++r) {
    const int e = tid + NT * r;
    const int h = e / (TI * PH * PW), rem = e - h * (TI * PH * PW);
    const int ti = rem / (PH * PW), r2 = rem - ti * (PH * PW), yy = r2 / PW, xx = r2 - yy * PW;
    const int img = img0 + ti, ly = S * ty0 - pad_y + yy, lx = S * tx0 - pad_x + xx;
    const bool in = e < NB;
    const bool ok = in && img < p.N && (unsigned)ly < (unsigned)Hl && (unsigned)lx < (unsigned)Wl;
    if constexpr (BLK) bsrc[r] = ok ? (unsigned)(((img * (p.Cin >> 3) + h) * p.H + ly) * p.W + lx) * 16u : OOB31;      // the piece of channel group h
    else bsrc[r] = ok ? (unsigned)(((img * p.Cin + 8 * h) * p.H + (ly >> p.up)) * p.W + (lx >> p.up)) * esz : OOB31;
    bdst[r] = in ? h * NQ + (S == 2 ? (xx & 1) * PAR : 0) + ti * IMGP + yy * PWP + (S == 2 ? xx >> 1 : xx) : -1;
    if constexpr (F_FOLD) {
      faff[r] = 8 * h + (p.fold.per_n && ok ? img * p.Cin : 0);      // row offset of this item's 8 channels in the scale / shift tables
      fcell[r] = (p.fold.cells && ok) ? ((img * (p.Cin >> 3) + h) * fG2 + p.fold.map[ly] * p.fold.G + p.fold.map[lx]) * 16 : 0;
    }
  }
  const unsigned cstride = (unsigned)(p.H * p.W) * esz;
  unsigned asrc[AR];   // weight piece e = (plane, h, t, row) of stage (0, 0), in 16-byte units; clamped when e >= NA
#pragma unroll
  for (int r = 0; r < AR; ++r) {
    const int e = min(tid + NT * r, NA - 1);
    const int plane = e / A_PLANE, r1 = e - plane * A_PLANE;
    const int h = r1 / (TGA * BM), r2 = r1 - h * (TGA * BM), t = r2 / BM, row = r2 - t * BM;
    asrc[r] = (unsigned)((((plane * p.nch) * 2 + h) * KKW + t) * p.mpad + bm0 + row);
  }

  float pb[BLK ? 1 : BR][8];
  u32x4 pq[BLK ? BR : 1];      // blocked layout: the pieces as they are stored
  u32x4 pa[AR];
  int c_staged = 0;    // first channel of the chunk in pb
  auto gload_b = [&](int c0) {
    if ((ABL & 8) && c0 != 16 * (p.slabs ? zsplit * p.cps : 0)) return;
    if constexpr (F_FOLD) c_staged = c0;
    if constexpr (BLK) {      // one 16-byte load per piece, from a clamped address (an out-of-window piece is zeroed when it is stored)
      const char* const xb = reinterpret_cast<const char*>(p.x) + (size_t)(c0 >> 3) * (size_t)(p.H * p.W) * 16u;
#pragma unroll
      for (int r = 0; r < BR; ++r) pq[r] = *reinterpret_cast<const u32x4*>(xb + (bsrc[r] == OOB31 ? 0u : bsrc[r]));
      return;
    }
    if (p.x_bf16) {      // (a bf16 value is the upper half of its fp32 form: the conversion at the LDS store is then exact)
#pragma unroll
      for (int r = 0; r < BR; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          pb[r][j] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsX, bsrc[r], (unsigned)(c0 + j) * cstride, 0) << 16);
      return;
    }
#pragma unroll
    for (int r = 0; r < BR; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        pb[r][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, bsrc[r], (unsigned)(c0 + j) * cstride, 0));
  };
  auto gload_a = [&](int cc, int tg) {                     // cc is clamped by the caller (the last prefetch is never used)
    if ((ABL & 8) && (cc != 0 || tg != 0)) return;
    const u32x4* base = p.wp + (long)(cc * 2 * KKW + tg * TG) * p.mpad;
#pragma unroll
    for (int r = 0; r < AR; ++r) pa[r] = base[asrc[r]];
  };
  // the input transform of patch item r (PArgs::fold), applied to its eight prefetched channels; padding items are left alone (they stay 0)
  auto fold_item = [&](int r) {
    if constexpr (F_FOLD) {
      if (!p.fold.scale || bsrc[r] == OOB31) return;
      if constexpr (F_SPADE) {
        if (p.fold.cells) {      // SPADE: (1 + gamma | beta) of the item's class-grid cell (one 64-byte line), mean / rstd of its channels from LDS (pconv.h)
          const float4* const cb = reinterpret_cast<const float4*>(p.fold.cells + (size_t)(c_staged >> 3) * (size_t)(fG2 * 16) + fcell[r]);
          const float4* const mu4 = reinterpret_cast<const float4*>(fmr + c_staged + faff[r]);
          const float4* const rs4 = reinterpret_cast<const float4*>(fmr + 512 + c_staged + faff[r]);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float4 g1 = cb[q], be = cb[2 + q], mu = mu4[q], rs = rs4[q];
            pb[r][4 * q + 0] = spade_value(pb[r][4 * q + 0], mu.x, rs.x, g1.x, be.x); pb[r][4 * q + 1] = spade_value(pb[r][4 * q + 1], mu.y, rs.y, g1.y, be.y);
            pb[r][4 * q + 2] = spade_value(pb[r][4 * q + 2], mu.z, rs.z, g1.z, be.z); pb[r][4 * q + 3] = spade_value(pb[r][4 * q + 3], mu.w, rs.w, g1.w, be.w);
          }
          return;
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {              // BatchNorm: four channels at a time; one fused multiply-add per element
        const float4 sc = reinterpret_cast<const float4*>(p.fold.scale + c_staged + faff[r])[q];
        const float4 sh = reinterpret_cast<const float4*>(p.fold.shift + c_staged + faff[r])[q];
        pb[r][4 * q + 0] = fmaf(pb[r][4 * q + 0], sc.x, sh.x); pb[r][4 * q + 1] = fmaf(pb[r][4 * q + 1], sc.y, sh.y);
        pb[r][4 * q + 2] = fmaf(pb[r][4 * q + 2], sc.z, sh.z); pb[r][4 * q + 3] = fmaf(pb[r][4 * q + 3], sc.w, sh.w);
      }
    }
  };
  // H16: the input transform and the fused input ReLU are applied to the registers first, so that the chunk's maximum is that of the values
  // that get converted; every wave publishes its maximum (sign bit cleared: magnitudes order like unsigned integers)
  auto prep_b = [&]() {
    if constexpr (H16) {
      unsigned tm = 0;
#pragma unroll
      for (int r = 0; r < BR; ++r) {
        fold_item(r);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (p.in_relu) pb[r][j] = fmaxf(pb[r][j], 0.f);
          tm = max(tm, __builtin_bit_cast(unsigned, pb[r][j]) & 0x7fffffffu);
        }
      }
      tm = wave_umax(tm);
      if (lane == 0) wmax_s[wave] = tm;
    }
  };
  // H16: running shift of the accumulators (acc = sum x * w * 2^-S_run, workgroup-uniform, never decreases) and the factor 2^-sx the chunk
  // being staged is converted with, sx = S_run - wexp[chunk] >= its own shift.  Called by every thread after the barrier that follows prep_b.
  int S_run = -(1 << 20);
  float bfac = 1.f;
  auto sstore_b = [&](int buf) {
    u32x4* const Pl = lds + buf;
#pragma unroll
    for (int r = 0; r < BR; ++r) {
      if (bdst[r] < 0) continue;
      if constexpr (H16) {      // (transform and ReLU already applied by prep_b)
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) { _Float16 a, b; split_h16(pb[r][j] * bfac, a, b); hi[j] = a; lo[j] = b; }
        Pl[bdst[r]] = __builtin_bit_cast(u32x4, hi);
        Pl[P_PLANE + bdst[r]] = __builtin_bit_cast(u32x4, lo);
        continue;
      }
      if constexpr (BLK) {
        u32x4 v = pq[r];
        if (p.in_relu) {      // ReLU on the stored bf16 pairs: a set sign bit clears its 16-bit half
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] &= ~(((v[q] >> 15) & 0x00010001u) * 0xFFFFu);
        }
        Pl[bdst[r]] = bsrc[r] != OOB31 ? v : u32x4{0u, 0u, 0u, 0u};
        continue;
      }
      fold_item(r);      // the producing norm's normalise-modulate, applied on the way to LDS: padding stays 0
      bf16x8 t0, t1, t2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = p.in_relu ? fmaxf(pb[r][j], 0.f) : pb[r][j];
        if constexpr (NSPL == 1) { t0[j] = (__bf16)v; }
        else if constexpr ((ABL & 2) != 0) { t0[j] = (__bf16)v; t1[j] = t0[j]; t2[j] = t0[j]; }
        else { __bf16 a, b, d; split3(v, a, b, d); t0[j] = a; t1[j] = b; t2[j] = d; }
      }
      Pl[bdst[r]] = __builtin_bit_cast(u32x4, t0);
      if constexpr (NSPL == 3) {
        Pl[P_PLANE + bdst[r]] = __builtin_bit_cast(u32x4, t1);
        Pl[2 * P_PLANE + bdst[r]] = __builtin_bit_cast(u32x4, t2);
      }
    }
  };
  auto sstore_a = [&](int buf) {
#pragma unroll
    for (int r = 0; r < AR; ++r) {
      const int e = tid + NT * r;
      if (NA % NT == 0 || e < NA) Al[buf + e] = pa[r];
    }
  };

  // ---- fragment addressing
  int qlane[WTN];
#pragma unroll
  for (int jt = 0; jt < WTN; ++jt) {
    const int j = wn * (BN / WNW) + 32 * jt + l31;
    const int ti = j / (TH * TW), r = j - ti * (TH * TW), py = r / TW, px = r - py * TW;
    qlane[jt] = lh * NQ + ti * IMGP + S * py * PWP + px;
  }
  const int arow = lh * (TGA * BM) + wm * (BM / 2) + l31;

  f32x16 accs[NPW][NACC][WTM][WTN];
#pragma unroll
  for (int w = 0; w < NPW; ++w)
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[w][a][i][j][r] = 0.f;
  auto& acc = accs[0];
  // H16, tiles of <= 2 accumulator blocks per wave: the hi*hi accumulator is emptied into a total (plain fp32 additions, round to nearest)
  // after every channel chunk.  The matrix unit aligns the 16 products and the accumulator of an instruction to the largest of them and cuts
  // every addend 3 bits below that one's last place (tools/probe/mfma_f16_numerics.hip): a 22-bit fp16 product loses bits against a large
  // accumulator where a 16-bit bf16 product does not, and on long reductions (Cin * taps >= 2000) that error, not the operand split,
  // decided the distance to fp64 (2.2-3.3x the exact fp32 kernel's; 0.3-1.1x with short chains).  The cross-term accumulator may run
  // long: its errors are scaled by 2^-11.
  // (Only where the total fits the register budget: <= 2 blocks at 256 threads, 1 at 512 — the four-block tiles spill with it, and
  // pconv_plan keeps long reductions off them.)
  constexpr bool FLUSH = H16 && NPW * WTM * WTN * (NTH / 256) <= 2;
  constexpr int FLUSH_EVERY = (PHS ? 4 : KK) >= 32 ? 1 : 32 / (PHS ? 4 : KK);      // chunks per flush: chains of <= ~32-36 instructions
  f32x16 tot[FLUSH ? NPW : 1][FLUSH ? WTM : 1][FLUSH ? WTN : 1];
  if constexpr (FLUSH) {
#pragma unroll
    for (int w = 0; w < NPW; ++w)
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) tot[w][i][j][r] = 0.f;
  }
  auto scale_b = [&](int chunk) {
    if constexpr (H16) {
      unsigned m = wmax_s[0];
#pragma unroll
      for (int q = 1; q < NTH / 64; ++q) m = max(m, wmax_s[q]);
      m = (unsigned)__builtin_amdgcn_readfirstlane((int)m);
      const int sw = p.wexp[chunk];
      int Sn = max(S_run, h16_shift(m) + sw), sx = Sn - sw;
      if (sx < -H16_SMAX) { sx = -H16_SMAX; Sn = sx + sw; }      // (everything so far is tiny: a larger shift than needed, still consistent)
      bfac = sx > H16_SMAX ? 0.f : exp2i(-sx);                    // (a chunk 2^-100 below the running scale contributes nothing)
      if (Sn != S_run) {
        if (S_run > -(1 << 19)) {      // the accumulators hold products at the old scale: *= 2^(old - new), exact
          const int d = S_run - Sn;
          const float f = d < -126 ? 0.f : exp2i(d);
#pragma unroll
          for (int w = 0; w < NPW; ++w)
#pragma unroll
            for (int a = 0; a < NACC; ++a)
#pragma unroll
              for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j)
#pragma unroll
                  for (int r = 0; r < 16; ++r) accs[w][a][i][j][r] *= f;
          if constexpr (FLUSH) {
#pragma unroll
            for (int w = 0; w < NPW; ++w)
#pragma unroll
              for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j)
#pragma unroll
                  for (int r = 0; r < 16; ++r) tot[w][i][j][r] *= f;
          }
        }
        S_run = Sn;
      }
    }
  };

  // Stagger (speed only): the two workgroups that share a CU start together, run the same program for the same time and so
  // stay in lockstep — both in their matrix segment (sharing the pipe), then both in their staging segment (pipe idle).  The
  // workgroup in the odd slot of the first generation waits about half a stage; later generations inherit the offset because a
  // slot is refilled when its workgroup ends.  HW_REG_HW_ID[19:16] = workgroup slot on the CU.
  if (p.prio > 0) {
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned tg = __builtin_amdgcn_s_getreg((3 << 11) | (16 << 6) | 4);
    if (lin < 768 && (tg & 1)) {
      if (p.prio == 1) __builtin_amdgcn_s_sleep(16);
      else if (p.prio == 2) __builtin_amdgcn_s_sleep(32);
      else if (p.prio == 3) __builtin_amdgcn_s_sleep(64);
      else { __builtin_amdgcn_s_sleep(127); }
    }
  }
  const int c_lo = p.slabs ? zsplit * p.cps : 0;                  // reduction split: this workgroup's chunk range
  const int nchunk = p.slabs ? min(p.cps, p.nch - c_lo) : p.nch;
  const int nstage = nchunk * NTG;     // stage s = (channel chunk s / NTG, tap group s % NTG)
  gload_b(16 * c_lo);
  gload_a(c_lo, PHS ? phase : 0);
  if constexpr (H16) { prep_b(); __syncthreads(); scale_b(c_lo); }
  sstore_b(0);
  sstore_a(0);
  __syncthreads();
  for (int s = 0; s < nstage; ++s) {
    const int cc = s / NTG, tg = PHS ? phase : s - cc * NTG;
    const int s1 = min(s + 1, nstage - 1), cc1 = s1 / NTG, tg1 = PHS ? phase : s1 - cc1 * NTG;   // past the end: reload the last stage (unused)
    const bool first_tg = (NTG == 1) || s == cc * NTG, last_tg = (NTG == 1) || (s + 1) % NTG == 0;
    // the next chunk's patch: fetched at the chunk's first tap group when it has its own LDS buffer (the stores wait for the
    // loads only NTG stages later), else at the last one (the registers are live for one stage only)
    if (DBB ? first_tg : last_tg) gload_b(16 * (c_lo + min(cc + 1, nchunk - 1)));
    gload_a(c_lo + cc1, tg1);
    const u32x4* const Pc = Pl + (DBB ? (cc & 1) * PB : 0);
    const u32x4* const Ac = Al + (DBA ? (s & 1) * AB : 0);
#pragma unroll
    for (int t = 0; t < TGA; ++t) {
      int toff;                                  // LDS piece offset of tap (kh, kw)
      const int pw = PAIR ? t / TG : 0, tt = PAIR ? t % TG : t;      // (PAIR: taps 0..3 belong to column phase 0, 4..7 to phase 1)
      if constexpr (TG == KK) toff = (tt / KSW) * PWP + (S == 2 ? ((tt % KSW) & 1) * PAR + ((tt % KSW) >> 1) : (tt % KSW)) + pw;
      else toff = tg * PWP + (S == 2 ? (t & 1) * PAR + (t >> 1) : t);
      (void)tg;
      bf16x8 fa[NSPL][WTM], fb[NSPL][WTN];
#pragma unroll
      for (int pl = 0; pl < NSPL; ++pl) {
#pragma unroll
        for (int i = 0; i < WTM; ++i) fa[pl][i] = __builtin_bit_cast(bf16x8, Ac[pl * A_PLANE + arow + ((ABL & 4) ? 0 : t) * BM + 32 * i]);
#pragma unroll
        for (int jt = 0; jt < WTN; ++jt) fb[pl][jt] = __builtin_bit_cast(bf16x8, Pc[pl * P_PLANE + qlane[jt] + ((ABL & 4) ? 0 : toff)]);
      }
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int jt = 0; jt < WTN; ++jt) {
          if constexpr ((ABL & 1) != 0) {      // keep the fragments live without the matrix instructions
            asm volatile("" ::"v"(fa[0][i]), "v"(fb[0][jt]));
            if constexpr (NSPL == 3) asm volatile("" ::"v"(fa[1][i]), "v"(fb[1][jt]), "v"(fa[2][i]), "v"(fb[2][jt]));
            continue;
          }
          if constexpr (H16) {      // hi*hi; the two cross terms (2^11 too large: scaled back once at the end) accumulate apart
            const f16x8 ah = __builtin_bit_cast(f16x8, fa[0][i]), al = __builtin_bit_cast(f16x8, fa[1][i]);
            const f16x8 bh = __builtin_bit_cast(f16x8, fb[0][jt]), bl = __builtin_bit_cast(f16x8, fb[1][jt]);
            accs[pw][0][i][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accs[pw][0][i][jt], 0, 0, 0);
            f32x16& lo = accs[pw][1][i][jt];
            lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, lo, 0, 0, 0);
            continue;
          }
          accs[pw][0][i][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][jt], accs[pw][0][i][jt], 0, 0, 0);
          if constexpr (NSPL == 3) {      // the small terms (<= 2^-8 of the leading one) accumulate apart: their rounding errors
            f32x16& lo = accs[pw][NACC - 1][i][jt];   // are 2^-8 smaller and the leading chain sees one rounding per K-step
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][jt], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][jt], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][jt], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][jt], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][jt], lo, 0, 0, 0);
          }
        }
    }
    // a buffer that is about to be overwritten may still be read by a slower wave unless it is the other one of a pair
    const bool stb = H16 && last_tg && s + 1 < nstage;      // H16: the next chunk's patch is converted at the end of this stage
    if constexpr (FLUSH) {
      if (last_tg && (FLUSH_EVERY == 1 || (cc + 1) % FLUSH_EVERY == 0)) {
#pragma unroll
        for (int w = 0; w < NPW; ++w)
#pragma unroll
          for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) { tot[w][i][j][r] += accs[w][0][i][j][r]; accs[w][0][i][j][r] = 0.f; }
      }
    }
    if (stb) prep_b();
    if constexpr (!DBA) __syncthreads();
    else if constexpr (!DBB) { if (last_tg) __syncthreads(); }
    else { if (stb) __syncthreads(); }                       // (the waves' maxima cross the workgroup)
    if (stb) scale_b(c_lo + cc + 1);
    if (s + 1 < nstage && !(ABL & 16)) {      // (the last stage's prefetch is a dummy: nothing to convert or store)
      if (last_tg) sstore_b(DBB ? ((cc + 1) & 1) * PB : 0);
      sstore_a(DBA ? ((s + 1) & 1) * AB : 0);
    }
    __syncthreads();
  }
  if constexpr (NSPL == 3) {
#pragma unroll
    for (int w = 0; w < NPW; ++w)
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) accs[w][0][i][j][r] += accs[w][1][i][j][r];
  }
  if constexpr (H16) {
#pragma unroll
    for (int w = 0; w < NPW; ++w)
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if constexpr (FLUSH) accs[w][0][i][j][r] += tot[w][i][j][r];
            accs[w][0][i][j][r] = fmaf(accs[w][1][i][j][r], 1.0f / 2048.0f, accs[w][0][i][j][r]);
          }
  }

  // ---- epilogue.  D[row][col]: col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (output channel) — a lane holds 16
  // channels of ONE pixel, so storing straight from the accumulators costs 16 four-byte store instructions per 32x32 tile.
  // Each wave transposes its tiles through its own LDS scratch (the staging buffers are free after the loop's last barrier):
  // afterwards a lane holds 4 consecutive pixels of one channel and every tile takes 4 sixteen-byte stores (and 16-byte
  // loads for the optional mask / accumulate operands, issued together — no per-element load-then-wait chains).
  // Channel-blocked ReLU mask (p.mask_blk; input gradients of a discriminator block, no bias): applied to the ACCUMULATORS, whose layout is
  // the mask's — a lane holds channels 8g + 4*lh + q of one pixel, i.e. one 8-byte half of the mask's 16-byte piece per g: four coalesced
  // 8-byte loads per tile (a wave covers 32 consecutive pieces) instead of four 2-byte loads 16 bytes apart per channel row after the
  // transpose.  Paired phases: pixel (a, b) of column phase pw is output pixel (2a + ph, 2b + pw).
  bool mask_done = false;
  if constexpr ((!PHS && F_MBLK) || (PAIR && F_PMASK16)) {
    if (p.pos_mask && p.mask_bf16 && p.mask_blk) {
      mask_done = true;
      const uint2* const mk = reinterpret_cast<const uint2*>(p.pos_mask);
      const int mh = PHS ? p.oh2 : p.OH, mw = PHS ? p.ow2 : p.OW;
      const long gstride = (long)mh * mw * 2;      // uint2 units between consecutive 8-channel groups
#pragma unroll
      for (int jt = 0; jt < WTN; ++jt) {
        const int jq = wn * (BN / WNW) + 32 * jt + l31;
        const int tiq = jq / (TH * TW), rq = jq - tiq * (TH * TW), pyq = rq / TW, pxq = rq - pyq * TW;
        const int imq = img0 + tiq;
#pragma unroll
        for (int w = 0; w < NPW; ++w) {
          const int yy = PHS ? 2 * (ty0 + pyq) + ph_y : ty0 + pyq, xx = PHS ? 2 * (tx0 + pxq) + w : tx0 + pxq;
          // all loads of the tile first (clamped addresses, no branches around them), then the selects: one wait, not sixteen
          uint2 mb[WTM][4];
#pragma unroll
          for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int mch = bm0 + wm * (BM / 2) + 32 * i + 8 * g;
              const bool ok = mch < p.Cout && imq < p.N;
              const long idx = ok ? ((((long)imq * (p.Cout >> 3) + (mch >> 3)) * mh + yy) * mw + xx) * 2 + lh : 0;
              const uint2 v = mk[idx];
              mb[i][g] = ok ? v : uint2{0x3f803f80u, 0x3f803f80u};      // (outside the tensor: nothing to mask, nothing is stored)
            }
#pragma unroll
          for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const uint2 b2 = mb[i][g];
              if (!(__builtin_bit_cast(float, b2.x << 16) > 0.f)) accs[w][0][i][jt][4 * g + 0] = 0.f;
              if (!(__builtin_bit_cast(float, b2.x & 0xffff0000u) > 0.f)) accs[w][0][i][jt][4 * g + 1] = 0.f;
              if (!(__builtin_bit_cast(float, b2.y << 16) > 0.f)) accs[w][0][i][jt][4 * g + 2] = 0.f;
              if (!(__builtin_bit_cast(float, b2.y & 0xffff0000u) > 0.f)) accs[w][0][i][jt][4 * g + 3] = 0.f;
            }
        }
      }
      (void)gstride;
    }
  }
  float* const ep = reinterpret_cast<float*>(lds) + wave * (32 * EP_PITCH);
  // YBLK: a lane finishes 16 channels of one pixel, so the per-channel constants (bias; the shortcut's bias and <= 4 weights) would be
  // 16 x 6 broadcast loads from global memory per tile — they are staged once per workgroup in LDS instead ([BM][8] floats; the staging
  // buffers are free after the loop's last barrier)
  float* const ytab = reinterpret_cast<float*>(lds);
  if constexpr (YBLK) {
    for (int e = tid; e < BM; e += NT) {
      const int m = bm0 + e;
      const bool ok = m < p.Cout;
      ytab[8 * e + 0] = (ok && p.bias) ? p.bias[m] : 0.f;
      ytab[8 * e + 1] = (F_SC && ok && p.sc_x && p.sc_b) ? p.sc_b[m] : 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) ytab[8 * e + 2 + c] = (F_SC && ok && p.sc_x && c < p.sc_cin) ? p.sc_w[m * p.sc_cin + c] : 0.f;
    }
    __syncthreads();
  }
  const long OHW = (long)p.OH * p.OW;
  const float osc = (p.odiv && !p.slabs) ? 1.0f / *p.odiv : 1.0f;     // (slabs carry raw sums: the slab reduction divides)
  const int er = lane >> 3, ec = (lane & 7) * 4;            // read-back: row er + 8*pass, columns ec..ec+3
  // BatchNorm statistics of the output (p.stats): every lane accumulates the values it stores, per channel row, as deviations from
  // the FIRST value it sees for that row (sref): sum d, sum d^2 — a channel whose mean is large against its spread (mean / std = 100)
  // then loses nothing to cancellation, unlike raw sum / sum-of-squares partials in fp32
  constexpr int SW = F_STATS ? WTM : 1;
  float ssum[SW][4], ssq[SW][4], sref[SW][4];
  float scnt = 0.f;                 // values this lane has accumulated per channel row (the same for all its rows: validity depends on the image only)
#pragma unroll
  for (int i = 0; i < SW; ++i)
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) { ssum[i][ps] = 0.f; ssq[i][ps] = 0.f; sref[i][ps] = 0.f; }
#pragma unroll
  for (int jt = 0; jt < WTN; ++jt) {
    const int j = wn * (BN / WNW) + 32 * jt + ec;
    const int ti = j / (TH * TW), r = j - ti * (TH * TW), py = r / TW, px = r - py * TW;
    const int img = img0 + ti;
    const long pbase = (long)img * p.Cout * OHW + (long)(ty0 + py) * p.OW + tx0 + px;
    float4 scx[F_SC ? 4 : 1];      // the shortcut's input at this lane's four pixels, per input channel
    if constexpr (F_SC) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        scx[c] = (p.sc_x && c < p.sc_cin && img < p.N) ? *reinterpret_cast<const float4*>(p.sc_x + ((long)img * p.sc_cin + c) * OHW + (long)(ty0 + py) * p.OW + tx0 + px)
                                                       : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
      if constexpr (YBLK) {
        const int jq = wn * (BN / WNW) + 32 * jt + l31;
        const int tiq = jq / (TH * TW), rq = jq - tiq * (TH * TW), pyq = rq / TW, pxq = rq - pyq * TW;
        const int imq = img0 + tiq;
        const long pixq = (long)(ty0 + pyq) * p.OW + tx0 + pxq;
        float scq[F_SC ? 4 : 1];      // the shortcut's input at this lane's pixel, per input channel
        if constexpr (F_SC) {
#pragma unroll
          for (int c = 0; c < 4; ++c) scq[c] = (p.sc_x && c < p.sc_cin && imq < p.N) ? p.sc_x[((long)imq * p.sc_cin + c) * OHW + pixq] : 0.f;
        }
        float ad[4][4];      // the addend's 16 values of this lane, all loads issued before the first use (clamped addresses)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int m = bm0 + wm * (BM / 2) + 32 * i + 8 * g + 4 * lh + q;
            const bool ok = p.addend != nullptr && m < p.Cout && imq < p.N;
            const float v = (p.addend ? p.addend : reinterpret_cast<const float*>(p.wp))[ok ? ((long)imq * p.Cout + m) * OHW + pixq : 0];
            ad[g][q] = ok ? v : 0.f;
          }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int mch = bm0 + wm * (BM / 2) + 32 * i + 8 * g;      // first channel of the piece
          if (mch < p.Cout && imq < p.N) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 ob;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int m = mch + 4 * lh + q;
              const float* const tb = ytab + 8 * (m - bm0);
              float t = acc[0][i][jt][4 * g + q] * osc + tb[0];
              t += ad[g][q];      // (fp32 NCHW addend: consecutive lanes read consecutive pixels)
              if constexpr (F_SC) {
                float sv = tb[1];
#pragma unroll
                for (int c = 0; c < 4; ++c) sv = fmaf(tb[2 + c], scq[c], sv);      // (zero weights beyond sc_cin / without a shortcut)
                t += sv;
              }
              if (p.relu) t = fmaxf(t, 0.f);
              ob[q] = (__bf16)t;
            }
            const long piece = ((long)(imq * (p.Cout >> 3) + (mch >> 3)) * p.OH + ty0 + pyq) * p.OW + tx0 + pxq;
            reinterpret_cast<uint2*>(p.y)[2 * piece + lh] = __builtin_bit_cast(uint2, ob);
          }
        }
        continue;
      }
#pragma unroll
      for (int r2 = 0; r2 < 16; ++r2) ep[((r2 & 3) + 8 * (r2 >> 2) + 4 * lh) * EP_PITCH + l31] = acc[0][i][jt][r2];
      // (one wave writes and reads its own scratch: the LDS accesses of a wave are ordered, no barrier needed)
      float4 v[4];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        v[ps] = *reinterpret_cast<const float4*>(ep + (er + 8 * ps) * EP_PITCH + ec);
        v[ps].x *= osc; v[ps].y *= osc; v[ps].z *= osc; v[ps].w *= osc;
        if constexpr (H16) {      // back from the accumulators' running scale
          v[ps].x = __builtin_ldexpf(v[ps].x, S_run); v[ps].y = __builtin_ldexpf(v[ps].y, S_run);
          v[ps].z = __builtin_ldexpf(v[ps].z, S_run); v[ps].w = __builtin_ldexpf(v[ps].w, S_run);
        }
      }
      const int mb = bm0 + wm * (BM / 2) + 32 * i + er;
      if constexpr (PAIR) {       // both column phases: pixel (a, b..b+3) -> row 2a+ph, columns 2b .. 2b+7 — two 16-byte stores
        float4 v1[4];
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) ep[((r2 & 3) + 8 * (r2 >> 2) + 4 * lh) * EP_PITCH + l31] = accs[NPW - 1][0][i][jt][r2];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          v1[ps] = *reinterpret_cast<const float4*>(ep + (er + 8 * ps) * EP_PITCH + ec);
          v1[ps].x *= osc; v1[ps].y *= osc; v1[ps].z *= osc; v1[ps].w *= osc;
          if constexpr (H16) {
            v1[ps].x = __builtin_ldexpf(v1[ps].x, S_run); v1[ps].y = __builtin_ldexpf(v1[ps].y, S_run);
            v1[ps].z = __builtin_ldexpf(v1[ps].z, S_run); v1[ps].w = __builtin_ldexpf(v1[ps].w, S_run);
          }
        }
        // (odd-sized outputs — 33 x 33, 65 x 65 — have rows that are only 4-byte aligned: under-aligned vector type)
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        const long OHW2 = (long)p.oh2 * p.ow2;
        const bool iok = img < p.N;
        const long ob = (long)img * p.Cout * OHW2 + (long)(2 * (ty0 + py) + ph_y) * p.ow2 + 2 * (tx0 + px);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int m = mb + 8 * ps;
          if (m < p.Cout && iok) {
            const float bb = p.bias ? p.bias[m] : 0.f;
            float4 lo = {v[ps].x + bb, v1[ps].x + bb, v[ps].y + bb, v1[ps].y + bb};
            float4 hi = {v[ps].z + bb, v1[ps].z + bb, v[ps].w + bb, v1[ps].w + bb};
            float* dst = p.y + ob + (long)m * OHW2;
            if (p.slabs) {      // reduction split: raw partial sums (bias and the rest happen in the slab reduction)
              float* sd = p.slabs + (long)zsplit * p.out_numel + ob + (long)m * OHW2;
              *reinterpret_cast<f4u*>(sd) = f4u{v[ps].x, v1[ps].x, v[ps].y, v1[ps].y};
              *reinterpret_cast<f4u*>(sd + 4) = f4u{v[ps].z, v1[ps].z, v[ps].w, v1[ps].w};
              continue;
            }
            if (p.pos_mask && !mask_done) {
              f4u m0, m1;
              if (F_PMASK16 && p.mask_bf16) {      // eight bf16 in one 16-byte load (widened by a shift: sign and zero are the stored value's)
                const uint4 b = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.pos_mask) + ob + (long)m * OHW2);
                m0 = f4u{__builtin_bit_cast(float, b.x << 16), __builtin_bit_cast(float, b.x & 0xffff0000u),
                         __builtin_bit_cast(float, b.y << 16), __builtin_bit_cast(float, b.y & 0xffff0000u)};
                m1 = f4u{__builtin_bit_cast(float, b.z << 16), __builtin_bit_cast(float, b.z & 0xffff0000u),
                         __builtin_bit_cast(float, b.w << 16), __builtin_bit_cast(float, b.w & 0xffff0000u)};
              } else {
                m0 = *reinterpret_cast<const f4u*>(p.pos_mask + ob + (long)m * OHW2);
                m1 = *reinterpret_cast<const f4u*>(p.pos_mask + ob + (long)m * OHW2 + 4);
              }
              if (!(m0.x > 0.f)) lo.x = 0.f;
              if (!(m0.y > 0.f)) lo.y = 0.f;
              if (!(m0.z > 0.f)) lo.z = 0.f;
              if (!(m0.w > 0.f)) lo.w = 0.f;
              if (!(m1.x > 0.f)) hi.x = 0.f;
              if (!(m1.y > 0.f)) hi.y = 0.f;
              if (!(m1.z > 0.f)) hi.z = 0.f;
              if (!(m1.w > 0.f)) hi.w = 0.f;
            }
            if (p.accumulate) {
              const f4u o0 = *reinterpret_cast<const f4u*>(dst), o1 = *reinterpret_cast<const f4u*>(dst + 4);
              lo.x += o0.x; lo.y += o0.y; lo.z += o0.z; lo.w += o0.w;
              hi.x += o1.x; hi.y += o1.y; hi.z += o1.z; hi.w += o1.w;
            }
            if (p.relu) {
              lo.x = fmaxf(lo.x, 0.f); lo.y = fmaxf(lo.y, 0.f); lo.z = fmaxf(lo.z, 0.f); lo.w = fmaxf(lo.w, 0.f);
              hi.x = fmaxf(hi.x, 0.f); hi.y = fmaxf(hi.y, 0.f); hi.z = fmaxf(hi.z, 0.f); hi.w = fmaxf(hi.w, 0.f);
            }
            *reinterpret_cast<f4u*>(dst) = f4u{lo.x, lo.y, lo.z, lo.w};
            *reinterpret_cast<f4u*>(dst + 4) = f4u{hi.x, hi.y, hi.z, hi.w};
          }
        }
        continue;
      }
      if constexpr (PHS) {        // pixel (a, b) of this phase -> (2a+ph, 2b+pw) of the 2*OH x 2*OW map: scalar accesses
        const long OHW2 = (long)p.oh2 * p.ow2;
        long pb2[4];
        bool iok[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {      // (with 2-wide tiles the four pixels of a lane are not in one row)
          const int jq = j + q, tq = jq / (TH * TW), rq = jq - tq * (TH * TW), pyq = rq / TW, pxq = rq - pyq * TW;
          iok[q] = img0 + tq < p.N;
          pb2[q] = (long)(img0 + tq) * p.Cout * OHW2 + (long)(2 * (ty0 + pyq) + ph_y) * p.ow2 + 2 * (tx0 + pxq) + ph_x;
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int m = mb + 8 * ps;
          if (m < p.Cout) {
            const float vv[4] = {v[ps].x, v[ps].y, v[ps].z, v[ps].w};
            const float bb = p.bias ? p.bias[m] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (!iok[q]) continue;
              const long o = pb2[q] + (long)m * OHW2;
              float val = vv[q] + bb;
              if (p.pos_mask && !(p.pos_mask[o] > 0.f)) val = 0.f;
              if (p.accumulate) val += p.y[o];
              if (p.relu) val = fmaxf(val, 0.f);
              p.y[o] = val;
            }
          }
        }
        continue;
      }
      if (p.slabs) {      // reduction split: raw partial sums; bias / mask / accumulate / ReLU happen in the slab reduction
        float* const sl = p.slabs + (long)zsplit * p.out_numel;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int m = mb + 8 * ps;
          if (m < p.Cout && img < p.N) *reinterpret_cast<float4*>(sl + pbase + (long)m * OHW) = v[ps];
        }
        continue;
      }
      float4 old[4], msk[4];
      const bool add_old = p.accumulate || (F_YOUT && p.addend != nullptr);
      if (add_old) {
        const float* const src = (F_YOUT && p.addend) ? p.addend : p.y;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int m = mb + 8 * ps;
          old[ps] = (m < p.Cout && img < p.N) ? *reinterpret_cast<const float4*>(src + pbase + (long)m * OHW) : float4{0.f, 0.f, 0.f, 0.f};
        }
      }
      if (p.pos_mask && p.mask_bf16 && !mask_done) {      // (bf16 -> fp32 is a shift: sign and zero are those of the stored value)
        const unsigned short* const mk = reinterpret_cast<const unsigned short*>(p.pos_mask);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int m = mb + 8 * ps;
          uint2 b = {0u, 0u};
          if (m < p.Cout && img < p.N) b = *reinterpret_cast<const uint2*>(mk + pbase + (long)m * OHW);
          msk[ps] = float4{__builtin_bit_cast(float, b.x << 16), __builtin_bit_cast(float, b.x & 0xffff0000u),
                           __builtin_bit_cast(float, b.y << 16), __builtin_bit_cast(float, b.y & 0xffff0000u)};
        }
      } else if (p.pos_mask && !mask_done) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int m = mb + 8 * ps;
          msk[ps] = (m < p.Cout && img < p.N) ? *reinterpret_cast<const float4*>(p.pos_mask + pbase + (long)m * OHW) : float4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int m = mb + 8 * ps;
        if (m < p.Cout && img < p.N) {
          float4 o = v[ps];
          if (p.bias) { const float bb = p.bias[m]; o.x += bb; o.y += bb; o.z += bb; o.w += bb; }
          if (p.pos_mask && !mask_done) {
            if (!(msk[ps].x > 0.f)) o.x = 0.f;
            if (!(msk[ps].y > 0.f)) o.y = 0.f;
            if (!(msk[ps].z > 0.f)) o.z = 0.f;
            if (!(msk[ps].w > 0.f)) o.w = 0.f;
          }
          if (add_old) { o.x += old[ps].x; o.y += old[ps].y; o.z += old[ps].z; o.w += old[ps].w; }
          if constexpr (F_SC) {
            if (p.sc_x) {
              float4 sv = {0.f, 0.f, 0.f, 0.f};
              if (p.sc_b) { const float sb = p.sc_b[m]; sv = float4{sb, sb, sb, sb}; }
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                if (c < p.sc_cin) {
                  const float wv = p.sc_w[m * p.sc_cin + c];
                  sv.x = fmaf(wv, scx[c].x, sv.x); sv.y = fmaf(wv, scx[c].y, sv.y); sv.z = fmaf(wv, scx[c].z, sv.z); sv.w = fmaf(wv, scx[c].w, sv.w);
                }
              }
              o.x += sv.x; o.y += sv.y; o.z += sv.z; o.w += sv.w;
            }
          }
          if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
          if constexpr ((ABL & 32) != 0) { asm volatile("" ::"v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w)); }      // (diagnostic: no output store)
          else if (F_YOUT && p.y_bf16) {      // four bf16 (round to nearest even) in one 8-byte store
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            const bf16x4 ob = {(__bf16)o.x, (__bf16)o.y, (__bf16)o.z, (__bf16)o.w};
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.y) + pbase + (long)m * OHW) = __builtin_bit_cast(uint2, ob);
          } else {
            *reinterpret_cast<float4*>(p.y + pbase + (long)m * OHW) = o;
          }
          if constexpr (F_STATS) {
            if (p.stats) {
              if (scnt == 0.f) sref[i][ps] = o.x;
              const float rf = sref[i][ps];
              const float d0 = o.x - rf, d1 = o.y - rf, d2 = o.z - rf, d3 = o.w - rf;
              ssum[i][ps] += (d0 + d1) + (d2 + d3);
              ssq[i][ps] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
          }
        }
      }
    }
    if constexpr (F_STATS) {
      if (p.stats && img < p.N) scnt += 4.f;      // (after ALL channel rows of this pixel quad: the first-value test above reads it)
    }
  }
  if constexpr (!PHS && F_STATS) {
    if (p.stats) {      // the 8 lanes of a channel row hold its pixels: (count, mean, M2) per lane, merged pairwise within the octet
      float* const row = p.stats + ((long)bx * WNW + wn) * p.Cout * 3;      // (Chan's update: exact in real arithmetic, no cancellation)
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          float n = scnt;
          const float inv = n > 0.f ? 1.0f / n : 0.f;
          float mu = sref[i][ps] + ssum[i][ps] * inv;
          float m2 = fmaxf(ssq[i][ps] - ssum[i][ps] * ssum[i][ps] * inv, 0.f);
#pragma unroll
          for (int o = 1; o < 8; o <<= 1) {
            const float n2 = __shfl_xor(n, o), mu2 = __shfl_xor(mu, o), m22 = __shfl_xor(m2, o);
            const float nt = n + n2, dl = mu2 - mu, wgt = nt > 0.f ? n2 / nt : 0.f;
            // (symmetric in the two partners up to rounding; both lanes keep the merged triple)
            mu = n >= n2 ? mu + dl * wgt : mu2 - dl * (nt > 0.f ? n / nt : 0.f);
            m2 = m2 + m22 + dl * dl * (nt > 0.f ? n * n2 / nt : 0.f);
            n = nt;
          }
          const int m = bm0 + wm * (BM / 2) + 32 * i + er + 8 * ps;
          if ((lane & 7) == 0 && m < p.Cout) { row[3 * m] = n; row[3 * m + 1] = mu; row[3 * m + 2] = m2; }
        }
    }
  }
}

// Last row (Y = 2W) and last column (X = 2W) of the 4x4 / stride-2 / pad-1 input gradient of an odd-sized input (2W+1 x 2W+1, square
// maps): with oy = (Y + 1 - kh) / 2 only kh = 3 reaches a valid dy row there (oy = W-1), and likewise kw = 3 on the last column; along
// the edge the other index takes its usual taps.  blockIdx.x = 2*(image group) + {0: last row, 1: last column without the corner}; a thread
// owns one output channel m and all 2W+1 edge pixels of it (accumulators in registers); the dy line (last row or last column, all
// reduction channels) is staged in LDS 64 channels at a time and read as broadcasts; w[c][m][3][0..3] is one 16-byte load.
// fp32 FMAs (operands rounded to bf16 first in bf16 mode) — 1/(2W) of the layer's work.  w[c*w_sc + m*w_sm + kh*4 + kw].
// A workgroup takes EI consecutive images (phase_edge_images): the tap loads of a channel — 16 bytes out of every 64-byte tap block,
// the kernel's L2 traffic — are shared by them (one image per workgroup: 1.07 ms for 1 179 objects of the batched layout encoder).
// Few images (the un-batched calls: < 512 workgroups that way) keep one image per workgroup: EI = 1.
template <int W> constexpr int phase_edge_images() { return W <= 8 ? 8 : (W <= 16 ? 4 : 2); }
template <int W, int EI>
__global__ __launch_bounds__(128) void phase_edge_k(const float* __restrict__ dy, const float* __restrict__ w,
                                                    const float* __restrict__ pos_mask, float* __restrict__ dx,
                                                    const float* __restrict__ out_div, int N, int Cred, int M, int w_sm, int w_sc, int relu,
                                                    int accumulate, int round_bf16) {
  constexpr int CB = 64, OW = 2 * W + 1;
  __shared__ float line[EI][CB][W];
  const int n0 = (blockIdx.x >> 1) * EI, col = blockIdx.x & 1, m = blockIdx.y * 128 + threadIdx.x;
  auto op = [&](float v) { return round_bf16 ? (float)(__bf16)v : v; };
  float acc[EI][OW];
#pragma unroll
  for (int g = 0; g < EI; ++g)
#pragma unroll
    for (int i = 0; i < OW; ++i) acc[g][i] = 0.f;
  for (int c0 = 0; c0 < Cred; c0 += CB) {
    __syncthreads();
    for (int e = threadIdx.x; e < EI * CB * W; e += 128) {
      const int g = e / (CB * W), r = e - g * (CB * W), c = r / W, o = r - c * W, n = n0 + g;
      float v = 0.f;
      if (c0 + c < Cred && n < N) v = col ? dy[(((long)n * Cred + c0 + c) * W + o) * W + (W - 1)] : dy[(((long)n * Cred + c0 + c) * W + (W - 1)) * W + o];
      line[g][c][o] = op(v);
    }
    __syncthreads();
    if (m < M) {
      const int cn = min(CB, Cred - c0);
#pragma unroll 2
      for (int c = 0; c < cn; ++c) {      // (two channels in flight: the tap loads of one overlap the multiply-adds of the other)
        const float* wp = w + (long)(c0 + c) * w_sc + (long)m * w_sm;
        float wv[4];
        if (col) { wv[0] = wp[3]; wv[1] = wp[7]; wv[2] = wp[11]; wv[3] = wp[15]; }
        else { const float4 t = *reinterpret_cast<const float4*>(wp + 12); wv[0] = t.x; wv[1] = t.y; wv[2] = t.z; wv[3] = t.w; }
#pragma unroll
        for (int k = 0; k < 4; ++k) wv[k] = op(wv[k]);
#pragma unroll
        for (int g = 0; g < EI; ++g)
#pragma unroll
          for (int o = 0; o < W; ++o) {
            const float d = line[g][c][o];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int X = 2 * o - 1 + k;      // edge position this (pixel, tap) pair lands on
              if (X >= 0 && X < OW) acc[g][X] = fmaf(d, wv[k], acc[g][X]);
            }
          }
      }
    }
  }
  if (m >= M) return;
  const float sc = out_div ? 1.0f / *out_div : 1.0f;
  const int cnt = col ? OW - 1 : OW;            // (the corner belongs to the row pass)
#pragma unroll
  for (int g = 0; g < EI; ++g) {
    const int n = n0 + g;
#pragma unroll
    for (int i = 0; i < OW; ++i) {      // (no early exit: the loops must unroll completely, or acc[] is indexed at run time and lives in scratch)
      if (i < cnt && n < N) {
        const int Y = col ? i : OW - 1, X = col ? OW - 1 : i;
        const long o = (((long)n * M + m) * OW + Y) * OW + X;
        float v = acc[g][i] * sc;
        if (pos_mask && !(pos_mask[o] > 0.f)) v = 0.f;
        if (accumulate) v += dx[o];
        if (relu) v = fmaxf(v, 0.f);
        dx[o] = v;
      }
    }
  }
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }


// =====================================================================================================================
// Weight gradient on the bf16 matrix cores:  dw[co][c][kh][kw] = sum over (image, oy, ox) of dy[co][pix] * x[c][pix*S + tap].
// GEMM view: rows = output channels, columns = (input channel, tap), reduction = pixels.  v_mfma_f32_16x16x32_bf16 takes 32
// pixels per step; lane l holds A[row l&15][k = 8*(l>>4) + j] and B[k = 8*(l>>4) + j][col l&15].
//   * A (dy): the tile's rows are staged pixel-contiguous, [plane][co][pixel] bf16 with a 288-byte row pitch (conflict-free
//     ds_read_b128 of 8 consecutive pixels per lane);
//   * B (x):  the patch is staged ONCE per tile, pixel-major / channel-fastest [plane][q][BC channels].  A lane needs 8 pixels
//     of ONE channel — a column of that image — which ds_read_b64_tr_b16 delivers: per 16-lane group it reads a 4-pixel x
//     16-channel block (each lane supplies the address of one 8-byte row quarter, so the tap shift and the convolution
//     stride are just address arithmetic, with no alignment constraint) and hands lane i column i.
// A workgroup owns 64*RT output channels x 16*CT input channels x all taps (accumulators in registers), walks its share of
// the 128-pixel tiles, and writes one slab; the slabs are added in a fixed order (agl_launch_slab_reduce: deterministic).
struct WArgs {
  const float* dy; const float* x; float* slabs;
  float* direct;          // single split: the result goes (is added, when accumulate) straight to dw — no slab, no reduction launch
  int accumulate;
  float* bias_slabs;      // optional [split][Cout]: per-channel sums of dy (the bias gradient), by the workgroups of input-channel block 0
  int N, Cin, H, W, Cout, OH, OW, pad, up, in_relu;
  int tiles, tiles_per_split;
  unsigned x_bytes, dy_bytes;
  int x_bf16;             // x is stored as bf16 (see PArgs)
  int dy_bf16;            // dy is stored as bf16 (the bf16-stored input of a transposed convolution, whose weight gradient has it in this role)
  int x_blk;              // XB instantiations: x is channel-blocked bf16 [N][Cin/8][H][W][8]
  InFold fold;            // optional input transform of x (pconv.h): the normalise-modulate of the BatchNorm that reads x, as in pconv_k
  int xgyz, xchunk, xitems, xgy;      // XCD-aware workgroup order (xchunk > 0): 1-D grid, see pbww_k
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// TSUB: taps per pass.  When 4*RT*CT*KS^2 accumulator registers (x2 in split mode) do not fit, the taps are processed in
// ceil(KS^2/TSUB) passes by different workgroups (blockIdx.y = channel block * passes + pass), each staging the same tiles.
// AUX: the bf16 dy operand (WArgs::dy_bf16) and the input transform of x (WArgs::fold) compiled in — the 4x4 / stride-2 family only
// (weight gradients of ConvTranspose2d(4, 2, 1) and of the folded encoder convolutions); see pconv_k's FEAT for why not everywhere.
// XB: x is a channel-blocked bf16 tensor: a staged item — the 8 channels of a patch pixel — is ONE 16-byte load that goes to LDS as it is
template <int KS, int S, int TW, int TH, int TI, int RT, int CT, int NSPL, int TSUB = KS * KS, bool AUX = false, bool XB = false>
__global__ __launch_bounds__(NT, (NSPL == 3 && TSUB > 16) ? 1 : 2) void pbww_k(WArgs p) {
  static_assert(!XB || (NSPL == 1 && !AUX), "blocked x: bf16 arithmetic, no input transform");
  constexpr int NPX = TI * TH * TW, KK = KS * KS, BMCO = 64 * RT, BC = 16 * CT, KSTEPS = NPX / 32;
  constexpr int NPASS = (KK + TSUB - 1) / TSUB;
  static_assert((NPX == 128 || NPX == 64) && (TW >= 4 || (TW == 2 && TH == 2)), "pbww geometry");   // (TW == 4: whole 4x4 maps — the 8-pixel dy
  // pieces are two full rows; TW == TH == 2: whole 2x2 maps, an 8-pixel piece is the maps of TWO consecutive images — two 16-byte loads)
  static_assert(!(TW == 2 && XB), "2x2 maps: NCHW x");
  constexpr int PH = S * (TH - 1) + KS, PW = S * (TW - 1) + KS, IMGP = PH * PW, NQ = TI * IMGP;
  constexpr int DPITCH = NPX * 2 + 32;                  // bytes per dy row: 18 (10 for 64 pixels) sixteen-byte slots -> conflict-free b128 reads
  constexpr int XROW = 2 * BC;                          // bytes per patch pixel
  constexpr int D_PLANE = BMCO * DPITCH, X_PLANE = NQ * XROW;
  constexpr bool H16 = NSPL == 2;      // fp16 hi / lo planes, three products, power-of-two scales per staged tile (see SPL)
  // 5x5 (25 taps): two accumulator sets are 200 registers and the kernel spilled (188-212 bytes of scratch, one workgroup per CU).  There
  // the lo terms keep their true scale and all three products go to ONE set: 100 registers, two workgroups per CU.
  constexpr bool ONEACC = H16 && TSUB > 16;
  constexpr int NACC = (NSPL >= 2 && !ONEACC) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSPL * (D_PLANE + X_PLANE)];
  __shared__ unsigned wmax_s[H16 ? 2 * (NT / 64) : 1];      // H16: the waves' maxima of the dy tile and of the x patch about to be converted
  unsigned char* const Dl = lds;
  unsigned char* const Xl = lds + NSPL * D_PLANE;
  // bias gradient (sum of dy over the pixels, per output channel): the 16 consecutive lanes that stage the 8-pixel pieces of one
  // channel row add them up; one of them keeps the running sum in LDS (the same lane owns the same channel in every tile: no races)
  __shared__ float lbias[BMCO];
  __shared__ __attribute__((aligned(16))) float fmr[AUX ? 32 : 1];      // SPADE form of the transform: mean [0, 16) and rstd [16, 32) of the block's 16 input channels

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  // Workgroup -> (pixel split bx, input-channel block by, output-channel block bz).  The workgroups of one pixel split read the
  // same dy tiles (every by) and the same x patches (every bz), in the same order; dealt round-robin over the 8 XCDs by their linear
  // id they sit on different L2s and each fetches them from memory (measured: 2.8x the operand bytes).  XCD-aware order
  // (p.xchunk > 0, 1-D grid of 8 * xchunk): XCD k takes the items [k * xchunk, (k + 1) * xchunk), item = (bx * gz + bz) * gy + by —
  // the channel blocks of one split run side by side on one L2.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xchunk > 0) {
    const int item = (int)(blockIdx.x & 7) * p.xchunk + (int)(blockIdx.x >> 3);
    if (item >= p.xitems) return;
    bx = item / p.xgyz;
    const int r = item - bx * p.xgyz;
    bz = r / p.xgy; by = r - bz * p.xgy;
  }
  const int pass = NPASS == 1 ? 0 : by % NPASS, tap0 = pass * TSUB;
  const bool do_bias = p.bias_slabs != nullptr && by == 0;
  if (do_bias && tid < BMCO) lbias[tid] = 0.f;      // (ordered before the first update by the barrier at the top of the tile loop)
  const int c0 = (by / NPASS) * BC, co0 = bz * BMCO;
  if constexpr (AUX) {
    static_assert(BC == 16, "the input transform is compiled into one-column (16-channel) blocks");
    if (p.fold.cells) {      // (uniform) SPADE form: the block's per-channel statistics, once per workgroup
      if (tid < 32) fmr[tid] = tid < 16 ? p.fold.mean[c0 + tid] : p.fold.scale[c0 + tid - 16];
      __syncthreads();
    }
  }
  const int t_beg = bx * p.tiles_per_split, t_end = min(p.tiles, t_beg + p.tiles_per_split);
  const int Hl = p.H << p.up, Wl = p.W << p.up;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const unsigned esz = p.x_bf16 ? 2u : 4u;
  const unsigned cstride = (unsigned)(p.H * p.W) * esz;
  const int tpr = p.OW / TW, tpi = TI == 1 ? (p.OH / TH) * tpr : 1;

  f32x4 acc[NACC][RT][CT][TSUB];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int t = 0; t < TSUB; ++t) acc[a][i][j][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane patch positions of the tr reads: K-step ks, read rd (rows 4*rd .. 4*rd+3 of the group's 8 pixels)
  int xq[KSTEPS][2];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int j = 32 * ks + 8 * lg + 4 * rd + (l15 >> 2);
      const int ti = j / (TH * TW), r = j - ti * (TH * TW), py = r / TW, px = r - py * TW;
      xq[ks][rd] = (ti * IMGP + S * (py * PW + px)) * XROW + (l15 & 3) * 8;
    }
  const int arow = (wave * 16 * RT + l15) * DPITCH + lg * 16;

  // Register prefetch: the global loads of tile t+1 are issued before the MFMAs of tile t and converted / stored after them,
  // so that the (HBM-latency-long) loads are not exposed between the two barriers of the staging phase.
  constexpr int ND = BMCO * (NPX / 8), DR = (ND + NT - 1) / NT;          // dy items (output channel, 8 consecutive pixels)
  constexpr int NXI = NQ * (BC / 8), XR = (NXI + NT - 1) / NT;           // x items (channel octet, patch pixel)
  float4 pdy[DR][2];
  float px[XB ? 1 : XR][8];
  u32x4 pxq[XB ? XR : 1];      // blocked x: the pieces as they are stored
  int pxin[AUX ? XR : 1];            // (fold) < 0: the staged patch pixel lies outside the map (padding must stay zero); else its class-grid cell (SPADE form) or 0
  auto tile_origin = [&](int tile, int& img0, int& ty0, int& tx0) {
    if constexpr (TI == 1) {
      img0 = tile / tpi;
      const int t = tile - img0 * tpi;
      ty0 = (t / tpr) * TH; tx0 = (t % tpr) * TW;
    } else {
      img0 = tile * TI; ty0 = 0; tx0 = 0;
    }
  };
  auto gload_dy = [&](int img0, int ty0, int tx0, int r, int sl) {
    {
      const int e = min(tid + NT * r, ND - 1);
      const int co = e / (NPX / 8), oc = e - co * (NPX / 8);
      const int j = 8 * oc, ti = j / (TH * TW), rr = j - ti * (TH * TW), py = rr / TW, px_ = rr - py * TW;
      const int img = img0 + ti;
      const bool ok = co0 + co < p.Cout && img < p.N;
      // (16-byte buffer loads are not usable here: this ROCm build lowers __builtin_amdgcn_raw_buffer_load_b128 to ONE dword
      //  load; plain 16-byte global loads from a clamped, always-valid address + a select instead)
      const long idx = ok ? ((long)((img * p.Cout + co0 + co) * p.OH + ty0 + py) * p.OW + tx0 + px_) : 0;
      if constexpr (TW == 2) {      // whole 2x2 maps: pixels 0-3 of the piece are image img, 4-7 image img + 1 (same channel)
        const bool ok2 = ok && img + 1 < p.N;
        const float4 lo = *reinterpret_cast<const float4*>(p.dy + idx);
        const float4 hi = *reinterpret_cast<const float4*>(p.dy + (ok2 ? idx + (long)p.Cout * 4 : 0));
        pdy[sl][0] = ok ? lo : float4{0.f, 0.f, 0.f, 0.f};
        pdy[sl][1] = ok2 ? hi : float4{0.f, 0.f, 0.f, 0.f};
        return;
      }
      if (AUX && p.dy_bf16) {      // eight bf16 in one 16-byte load (bf16 -> fp32 is a shift)
        const uint4 b = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.dy) + idx);
        const float4 lo = {__builtin_bit_cast(float, b.x << 16), __builtin_bit_cast(float, b.x & 0xffff0000u),
                           __builtin_bit_cast(float, b.y << 16), __builtin_bit_cast(float, b.y & 0xffff0000u)};
        const float4 hi = {__builtin_bit_cast(float, b.z << 16), __builtin_bit_cast(float, b.z & 0xffff0000u),
                           __builtin_bit_cast(float, b.w << 16), __builtin_bit_cast(float, b.w & 0xffff0000u)};
        pdy[sl][0] = ok ? lo : float4{0.f, 0.f, 0.f, 0.f};
        pdy[sl][1] = ok ? hi : float4{0.f, 0.f, 0.f, 0.f};
        return;
      }
      const float4 lo = *reinterpret_cast<const float4*>(p.dy + idx);
      const float4 hi = *reinterpret_cast<const float4*>(p.dy + idx + 4);
      pdy[sl][0] = ok ? lo : float4{0.f, 0.f, 0.f, 0.f};
      pdy[sl][1] = ok ? hi : float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto gload_x = [&](int img0, int ty0, int tx0, int r, int sl) {
    {
      const int e = min(tid + NT * r, NXI - 1);
      const int oc = e / NQ, q = e - oc * NQ;
      const int ti = q / IMGP, r2 = q - ti * IMGP, yy = r2 / PW, xx = r2 - yy * PW;
      const int img = img0 + ti, ly = S * ty0 - p.pad + yy, lx = S * tx0 - p.pad + xx;
      const bool ok = img < p.N && (unsigned)ly < (unsigned)Hl && (unsigned)lx < (unsigned)Wl;
      if constexpr (XB) {
        const long idx = ok ? ((long)((img * (p.Cin >> 3) + (c0 >> 3) + oc) * p.H + ly) * p.W + lx) : 0;
        const u32x4 v = reinterpret_cast<const u32x4*>(p.x)[idx];
        pxq[sl] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        return;
      }
      const unsigned off = ok ? (unsigned)(((img * p.Cin + c0 + 8 * oc) * p.H + (ly >> p.up)) * p.W + (lx >> p.up)) * esz : OOB31;
      if constexpr (AUX) pxin[sl] = !ok ? -1 : (p.fold.cells ? p.fold.map[ly] * p.fold.G + p.fold.map[lx] : 0);
      if (p.x_bf16) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
          px[sl][jj] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsX, off, (unsigned)jj * cstride, 0) << 16);
      } else {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
          px[sl][jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, off, (unsigned)jj * cstride, 0));
      }
    }
  };
  // H16: conversion factors of the tile being staged and the running shift of the accumulators (acc = sum dy * x * 2^-S_run; workgroup-uniform)
  float dfac = 1.f, xfac = 1.f;
  int S_run = -(1 << 20);
  auto sstore_dy = [&](int r, int sl) {
    {
      const int e = tid + NT * r;
      if (ND % NT != 0 && e >= ND) return;
      const int co = e / (NPX / 8), oc = e - co * (NPX / 8);
      const float v[8] = {pdy[sl][0].x, pdy[sl][0].y, pdy[sl][0].z, pdy[sl][0].w, pdy[sl][1].x, pdy[sl][1].y, pdy[sl][1].z, pdy[sl][1].w};
      if (do_bias) {
        static_assert(NPX / 8 == 16 || NPX / 8 == 8, "lanes per dy row");
        float sm = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
#pragma unroll
        for (int o = 1; o < NPX / 8; o <<= 1) sm += __shfl_xor(sm, o);
        if (oc == 0) lbias[co] += sm;
      }
      unsigned char* dst = Dl + co * DPITCH + oc * 16;
      if constexpr (H16) {
        f16x8 hi, lo;
#pragma unroll
        for (int q = 0; q < 8; ++q) { _Float16 a, b; if constexpr (ONEACC) split_h16_unscaled(v[q] * dfac, a, b); else split_h16(v[q] * dfac, a, b); hi[q] = a; lo[q] = b; }
        *reinterpret_cast<u32x4*>(dst) = __builtin_bit_cast(u32x4, hi);
        *reinterpret_cast<u32x4*>(dst + D_PLANE) = __builtin_bit_cast(u32x4, lo);
        return;
      }
      bf16x8 t0, t1, t2;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if constexpr (NSPL == 1) { t0[q] = (__bf16)v[q]; }
        else { __bf16 a, b, d; split3(v[q], a, b, d); t0[q] = a; t1[q] = b; t2[q] = d; }
      }
      *reinterpret_cast<u32x4*>(dst) = __builtin_bit_cast(u32x4, t0);
      if constexpr (NSPL == 3) {
        *reinterpret_cast<u32x4*>(dst + D_PLANE) = __builtin_bit_cast(u32x4, t1);
        *reinterpret_cast<u32x4*>(dst + 2 * D_PLANE) = __builtin_bit_cast(u32x4, t2);
      }
    }
  };
  int img0_staged = 0;      // first image of the tile whose operands are in the prefetch registers
  // the input transform of x item (r: item index, sl: its prefetch slot), applied to its eight prefetched channels (WArgs::fold; see pconv_k)
  auto fold_x = [&](int r, int sl) {
    if constexpr (AUX) {
      if (!p.fold.scale || pxin[sl] < 0) return;
      const int e = min(tid + NT * r, NXI - 1), oc = e / NQ, q = e - oc * NQ;
      const int ch = c0 + 8 * oc, img = img0_staged + q / IMGP;
      if (p.fold.cells) {
        const float4* const cb = reinterpret_cast<const float4*>(p.fold.cells) +
                                 ((size_t)(img * (p.Cin >> 3) + (ch >> 3)) * (size_t)(p.fold.G * p.fold.G) + pxin[sl]) * 4;
        const float4* const mu4 = reinterpret_cast<const float4*>(fmr + 8 * oc);
        const float4* const rs4 = reinterpret_cast<const float4*>(fmr + 16 + 8 * oc);
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
          const float4 g1 = cb[hq], be = cb[2 + hq], mu = mu4[hq], rs = rs4[hq];
          px[sl][4 * hq + 0] = spade_value(px[sl][4 * hq + 0], mu.x, rs.x, g1.x, be.x); px[sl][4 * hq + 1] = spade_value(px[sl][4 * hq + 1], mu.y, rs.y, g1.y, be.y);
          px[sl][4 * hq + 2] = spade_value(px[sl][4 * hq + 2], mu.z, rs.z, g1.z, be.z); px[sl][4 * hq + 3] = spade_value(px[sl][4 * hq + 3], mu.w, rs.w, g1.w, be.w);
        }
        return;
      }
      const int rowo = p.fold.per_n ? img * p.Cin : 0;
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        const float4 sc = reinterpret_cast<const float4*>(p.fold.scale + rowo + ch)[hq];
        const float4 sh = reinterpret_cast<const float4*>(p.fold.shift + rowo + ch)[hq];
        px[sl][4 * hq + 0] = fmaf(px[sl][4 * hq + 0], sc.x, sh.x); px[sl][4 * hq + 1] = fmaf(px[sl][4 * hq + 1], sc.y, sh.y);
        px[sl][4 * hq + 2] = fmaf(px[sl][4 * hq + 2], sc.z, sh.z); px[sl][4 * hq + 3] = fmaf(px[sl][4 * hq + 3], sc.w, sh.w);
      }
    }
  };
  auto sstore_x = [&](int r, int sl) {
    {
      const int e = tid + NT * r;
      if (NXI % NT != 0 && e >= NXI) return;
      const int oc = e / NQ, q = e - oc * NQ;
      if constexpr (XB) {
        u32x4 v = pxq[sl];
        if (p.in_relu) {      // ReLU on the stored bf16 pairs: a set sign bit clears its 16-bit half
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) v[qq] &= ~(((v[qq] >> 15) & 0x00010001u) * 0xFFFFu);
        }
        *reinterpret_cast<u32x4*>(Xl + q * XROW + oc * 16) = v;
        return;
      }
      if constexpr (H16) {      // (input transform and ReLU already applied to the registers by prep_tile)
        f16x8 hi, lo;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) { _Float16 a, b; if constexpr (ONEACC) split_h16_unscaled(px[sl][jj] * xfac, a, b); else split_h16(px[sl][jj] * xfac, a, b); hi[jj] = a; lo[jj] = b; }
        unsigned char* dst = Xl + q * XROW + oc * 16;
        *reinterpret_cast<u32x4*>(dst) = __builtin_bit_cast(u32x4, hi);
        *reinterpret_cast<u32x4*>(dst + X_PLANE) = __builtin_bit_cast(u32x4, lo);
        return;
      }
      fold_x(r, sl);      // the producing norm's normalise-modulate, applied on the way to LDS (see pconv_k)
      bf16x8 t0, t1, t2;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        float v = px[sl][jj];
        if (p.in_relu) v = fmaxf(v, 0.f);
        if constexpr (NSPL == 1) { t0[jj] = (__bf16)v; }
        else { __bf16 a, b, d; split3(v, a, b, d); t0[jj] = a; t1[jj] = b; t2[jj] = d; }
      }
      unsigned char* dst = Xl + q * XROW + oc * 16;
      *reinterpret_cast<u32x4*>(dst) = __builtin_bit_cast(u32x4, t0);
      if constexpr (NSPL == 3) {
        *reinterpret_cast<u32x4*>(dst + X_PLANE) = __builtin_bit_cast(u32x4, t1);
        *reinterpret_cast<u32x4*>(dst + 2 * X_PLANE) = __builtin_bit_cast(u32x4, t2);
      }
    }
  };

  constexpr bool PREF = NSPL >= 2 || RT * CT == 1;
  static_assert(!H16 || PREF, "the fp16 split form converts from the prefetch registers");
  // H16: fold / ReLU on the prefetched x registers, then the maxima of the dy tile and the x patch (per wave; combined by scale_tile after
  // the barrier that follows)
  auto prep_tile = [&]() {
    if constexpr (H16) {
      unsigned md = 0, mx = 0;
#pragma unroll
      for (int r = 0; r < DR; ++r) {
        const float v[8] = {pdy[r][0].x, pdy[r][0].y, pdy[r][0].z, pdy[r][0].w, pdy[r][1].x, pdy[r][1].y, pdy[r][1].z, pdy[r][1].w};
#pragma unroll
        for (int q = 0; q < 8; ++q) md = max(md, __builtin_bit_cast(unsigned, v[q]) & 0x7fffffffu);
      }
#pragma unroll
      for (int r = 0; r < XR; ++r) {
        fold_x(r, r);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          if (p.in_relu) px[r][jj] = fmaxf(px[r][jj], 0.f);
          mx = max(mx, __builtin_bit_cast(unsigned, px[r][jj]) & 0x7fffffffu);
        }
      }
      md = wave_umax(md); mx = wave_umax(mx);
      if (lane == 0) { wmax_s[wave] = md; wmax_s[NT / 64 + wave] = mx; }
    }
  };
  auto scale_tile = [&]() {
    if constexpr (H16) {
      unsigned md = 0, mx = 0;
#pragma unroll
      for (int q = 0; q < NT / 64; ++q) { md = max(md, wmax_s[q]); mx = max(mx, wmax_s[NT / 64 + q]); }
      md = (unsigned)__builtin_amdgcn_readfirstlane((int)md); mx = (unsigned)__builtin_amdgcn_readfirstlane((int)mx);
      const int sd = h16_shift(md);
      const int Sn = max(S_run, sd + h16_shift(mx)), sx = Sn - sd;
      dfac = exp2i(-sd);
      xfac = sx > H16_SMAX ? 0.f : exp2i(-sx);      // (a tile 2^-100 below the running scale contributes nothing)
      if (Sn != S_run) {
        if (S_run > -(1 << 19)) {
          const int d = S_run - Sn;
          const float f = d < -126 ? 0.f : exp2i(d);
#pragma unroll
          for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
              for (int j = 0; j < CT; ++j)
#pragma unroll
                for (int t = 0; t < TSUB; ++t) acc[a][i][j][t] *= f;
        }
        S_run = Sn;
      }
    }
  };     // (the 2x2-fragment bf16 variants have no registers to spare)
  auto gload = [&](int tile) {
    int img0, ty0, tx0;
    tile_origin(tile, img0, ty0, tx0);
    if constexpr (AUX) img0_staged = img0;
#pragma unroll
    for (int r = 0; r < DR; ++r) gload_dy(img0, ty0, tx0, r, r);
#pragma unroll
    for (int r = 0; r < XR; ++r) gload_x(img0, ty0, tx0, r, r);
  };
  if (PREF && t_beg < t_end) gload(t_beg);
  for (int tile = t_beg; tile < t_end; ++tile) {
    prep_tile();
    __syncthreads();                                   // everyone is done reading the previous tile
    scale_tile();
    if constexpr (PREF) {
#pragma unroll
      for (int r = 0; r < DR; ++r) sstore_dy(r, r);
#pragma unroll
      for (int r = 0; r < XR; ++r) sstore_x(r, r);
    } else {                                           // item by item through one register set
      int img0, ty0, tx0;
      tile_origin(tile, img0, ty0, tx0);
      if constexpr (AUX) img0_staged = img0;
#pragma unroll 2
      for (int r = 0; r < DR; ++r) { gload_dy(img0, ty0, tx0, r, 0); sstore_dy(r, 0); }
#pragma unroll 2
      for (int r = 0; r < XR; ++r) { gload_x(img0, ty0, tx0, r, 0); sstore_x(r, 0); }
    }
    __syncthreads();
    if (PREF && tile + 1 < t_end) gload(tile + 1);
    // ---- 32 pixels per step
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      bf16x8 fa[NSPL][RT];
#pragma unroll
      for (int pl = 0; pl < NSPL; ++pl)
#pragma unroll
        for (int i = 0; i < RT; ++i)
          fa[pl][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Dl + pl * D_PLANE + arow + i * 16 * DPITCH + ks * 64));
#pragma unroll
      for (int t = 0; t < TSUB; ++t) {
        const int tap = tap0 + t;
        if (NPASS > 1 && tap >= KK) break;                      // (uniform) last pass of an uneven split
        const int toff = ((tap / KS) * PW + (tap % KS)) * XROW;
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          bf16x8 fb[NSPL];
#pragma unroll
          for (int pl = 0; pl < NSPL; ++pl) {
            typedef s16x4 __attribute__((address_space(3))) * lptr;
            const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(Xl + pl * X_PLANE + xq[ks][0] + toff + j * 32));
            const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(Xl + pl * X_PLANE + xq[ks][1] + toff + j * 32));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 both = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
            fb[pl] = __builtin_bit_cast(bf16x8, both);
          }
#pragma unroll
          for (int i = 0; i < RT; ++i) {
            if constexpr (H16) {
              const f16x8 ah = __builtin_bit_cast(f16x8, fa[0][i]), al = __builtin_bit_cast(f16x8, fa[1][i]);
              const f16x8 bh = __builtin_bit_cast(f16x8, fb[0]), bl = __builtin_bit_cast(f16x8, fb[1]);
              acc[0][i][j][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[0][i][j][t], 0, 0, 0);
              f32x4& lo = acc[NACC - 1][i][j][t];
              lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, lo, 0, 0, 0);
              lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, lo, 0, 0, 0);
              continue;
            }
            acc[0][i][j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[0], acc[0][i][j][t], 0, 0, 0);
            if constexpr (NSPL == 3) {
              f32x4& lo = acc[NACC - 1][i][j][t];
              lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2][i], fb[0], lo, 0, 0, 0);
              lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[2], lo, 0, 0, 0);
              lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[1], lo, 0, 0, 0);
              lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][i], fb[0], lo, 0, 0, 0);
              lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][i], fb[1], lo, 0, 0, 0);
            }
          }
        }
      }
    }
  }

  if (do_bias) {
    __syncthreads();
    if (tid < BMCO && co0 + tid < p.Cout) p.bias_slabs[(long)bx * p.Cout + co0 + tid] = lbias[tid];
  }
  // ---- slab: C tile col = lane&15 (input channel), row = 4*(lane>>4) + reg (output channel)
  float* out = p.direct ? p.direct : p.slabs + (long)bx * p.Cout * p.Cin * KK;
  const bool acc_out = p.direct != nullptr && p.accumulate;
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wave * 16 * RT + 16 * i + 4 * lg + r, c = c0 + 16 * j + l15;
        if (co < p.Cout) {
          float* o = out + ((long)co * p.Cin + c) * KK;
#pragma unroll
          for (int t = 0; t < TSUB; ++t) {
            if (tap0 + t >= KK) break;
            float v = acc[0][i][j][t][r];
            if constexpr (NSPL == 3) v += acc[1][i][j][t][r];
            if constexpr (H16 && !ONEACC) v = fmaf(acc[NACC - 1][i][j][t][r], 1.0f / 2048.0f, v);
            if constexpr (H16) v = __builtin_ldexpf(v, S_run);
            o[tap0 + t] = acc_out ? o[tap0 + t] + v : v;
          }
        }
      }
}

}  // namespace

// planes of a packed weight buffer for the ABI-level nsplit (1 or 3), and the bytes behind them (H16: one shift per 16-channel chunk)
static inline int pack_planes(int nsplit) { return nsplit == 3 ? SPL : 1; }
// (H16 tail: nch shifts rounded up to 16 bytes, then H16_SCAN_G partial maxima per chunk)
static inline long pack_tail_bytes(int nch, int nsplit) { return (nsplit == 3 && SPL == 2) ? (long)round_up(nch * 4, 16) + (long)nch * H16_SCAN_G * 4 : 0; }
long pconv_ws_bytes(int Cin, int Cout, int ks, int nsplit) {
  if (!(ks == 1 || ks == 3 || ks == 4 || ks == 5) || Cin % 16 != 0 || Cout < 48) return 0;
  return (long)pack_planes(nsplit) * (Cin / 16) * 2 * ks * ks * round_up(Cout, 128) * 16 + pack_tail_bytes(Cin / 16, nsplit);
}
constexpr int kPconvMaxSplits = 8;
// ... plus room for the slabs of a reduction split of a small output (out_numel floats per slab)
long pconv_ws_bytes_split(int Cin, int Cout, int ks, int nsplit, long out_numel) {
  const long packed = pconv_ws_bytes(Cin, Cout, ks, nsplit);
  if (!packed) return 0;
  const long slabs = out_numel * 4 * kPconvMaxSplits;
  return slabs <= (64L << 20) ? packed + slabs : packed;
}

struct PConvPlan { int geo, bm; bool s2, w32, wide, half; long ptiles; int oh, ow, splits; };
// Pure eligibility / tiling decision (no launches): 0 when pconv takes the call.
static int pconv_plan(const PConvArgs& a, PConvPlan& pl) {
  const bool s2 = a.stride == 2;
  if (s2 ? !((a.ks == 4 && a.pad == 1) || (a.ks == 3 && a.pad == 0)) || a.up != 0
         : !(a.stride == 1 && (a.ks == 3 || a.ks == 5 || (a.ks == 1 && a.pad == 0 && a.up == 0)))) return -1;
  if (a.Cin % 16 != 0 || a.Cout < 48) return -1;
  if (!(a.nsplit == 1 || a.nsplit == 3)) return -1;
  if (((a.H << a.up) + 2 * a.pad - a.ks) / a.stride + 1 != a.OH || ((a.W << a.up) + 2 * a.pad - a.ks) / a.stride + 1 != a.OW) return -1;
  // a 1x1 convolution has no spatial structure: its map is re-read as (HW/16) x 16 so that any HW % 128 == 0 tiles as geo 0
  int oh = a.OH, ow = a.OW;
  if (a.ks == 1) {
    const int hw = a.OH * a.OW;
    if (hw % 128 == 0) { ow = 16; oh = hw / 16; }
    else if (hw == 64) { ow = 8; oh = 8; }
    else if (hw == 16) { ow = 4; oh = 4; }
    else return -1;
  }
  int geo;
  if (ow % 16 == 0 && oh % 8 == 0) geo = 0;
  else if (ow == 8 && oh == 8) geo = 1;
  else if (ow == 4 && oh == 4 && (a.ks == 3 || a.ks == 4 || a.ks == 1)) geo = 2;
  else if (s2 && ow == 2 && oh == 2) geo = 5;               // stride 2 down to 2 x 2 maps: 16 (split mode) or 32 images per tile
  else if (!s2 && ow % 8 == 0 && oh % 8 == 0) geo = 3;      // 8 x 8 tiles of one image (64 pixels)
  else return -1;
  if ((long)a.N * a.Cin * a.H * a.W >= (1L << 29) || (long)a.N * a.Cout * a.OH * a.OW >= (1L << 30)) return -1;   // 32-bit offsets
  // nsplit 1: 128 channels (64 when Cout <= 64) x 256 pixels per workgroup, 128 pixels when the wider tile would leave CUs
  // without work; nsplit 3: 64 x 128 (three LDS planes).  Stride 2 (patch = 4x the tile): 128 pixels, 64 in split mode.
  const long px128 = geo == 0 ? (long)a.N * (oh / 8) * (ow / 16) : (geo == 1 ? agl_cdiv(a.N, 2) : (geo == 5 ? agl_cdiv(a.N, 32) : agl_cdiv(a.N, 8)));
  int bm = (a.nsplit == 3 || a.Cout <= 64) ? 64 : 128;
  // fp16 hi / lo split form: reductions of more than 64 (chunk, tap) steps stay on the two-block tiles, whose hi*hi accumulator is emptied
  // into an fp32 total every ~32 steps (pconv_k FLUSH) — on a long chain the matrix unit's addend truncation would decide the error
  const bool long_k = a.nsplit == 3 && SPL == 2 && (long)(a.Cin / 16) * a.ks * a.ks > 64;
  // 1x1: bandwidth-bound — read the input once per 128 output channels where that still leaves a workgroup per CU
  if (a.ks == 1 && a.Cout > 64 && px128 * agl_cdiv(a.Cout, 128) >= 256) bm = 128;
  // 3x3 split mode: 128 output channels per workgroup (one kernel row of weights per stage) halve the conversions and patch
  // reads per MFMA where the grid stays full: -3..-13 % per layer
  if (a.nsplit == 3 && !s2 && a.ks == 3 && geo != 2 && a.Cout >= 128 && px128 * agl_cdiv(a.Cout, 128) >= 512 && !long_k) bm = 128;
  const bool w32 = ow % 32 == 0 && a.ks != 1;                 // geo 0: 8 x 32 (wide) / 4 x 32 tiles; else 16 x 16 (wide) / 8 x 16
  // 256-pixel tiles: bf16 mode, and the 64-channel 3x3 layers of the split mode (one kernel row of weights per stage; -2..-12 %)
  bool wide = !s2 && geo != 3 && a.ks != 1 && (a.nsplit == 1 || (a.ks == 3 && bm == 64 && geo == 0)) && (px128 / 2) * agl_cdiv(a.Cout, bm) >= 512 && !long_k;
  if (geo == 0 && wide && !w32 && oh % 16 != 0) wide = false;
  const bool half = s2 && a.nsplit == 3;           // 64-pixel tiles
  long ptiles;
  if (geo == 0) ptiles = half ? px128 * 2 : px128 / (wide ? 2 : 1);
  else if (geo == 1) ptiles = half ? a.N : agl_cdiv(a.N, wide ? 4 : 2);
  else if (geo == 2) ptiles = agl_cdiv(a.N, half ? 4 : (wide ? 16 : 8));
  else if (geo == 5) ptiles = agl_cdiv(a.N, half ? 16 : 32);
  else ptiles = (long)a.N * (oh / 8) * (ow / 8);
  // no reduction split in this kernel: a grid that cannot occupy most CUs runs one long serial K loop per workgroup and is
  // slower than the split-K im2col / position-major kernels (ConvLSTM recurrence steps over the few images still active)
  // (measured on the ConvLSTM recurrence steps, 128 -> 512 5x5 on 8x8 maps: with a short reduction — <= 256 (chunk, tap) steps —
  //  the kernel still wins down to ~100 workgroups: 48 images 0.150 -> 0.099 ms, 32 images 0.110 -> 0.094 ms, 16 images slower)
  const long wgs = ptiles * agl_cdiv(a.Cout, bm);
  const bool short_k = (long)(a.Cin / 16) * a.ks * a.ks <= 256;
  // Below that, a long reduction is cut over blockIdx.z (raw partial outputs in slabs, summed by the split-K epilogue of
  // conv.hip): input gradients of the ConvLSTM recurrence steps — 512 -> 128 channels, 64 workgroups, 800 (chunk, tap) steps each
  int splits = 1;
  if (wgs < 200 && !(short_k && wgs >= 96) && !a.any_grid) {
    const int nch = a.Cin / 16;
    // (four chunks per split; 5x5 — the recurrence steps of the ConvLSTM and their input gradients over the few images still active —
    //  two chunks per split below 512 reduction channels, one below 128: 50 / 25 stages per workgroup still amortise its prologue)
    splits = std::min(kPconvMaxSplits, a.ks != 5 ? nch / 4 : (nch >= 32 ? nch / 4 : (nch >= 8 ? nch / 2 : nch)));
    while (splits > 2 && wgs * (splits / 2) >= 256) splits /= 2;
    if (splits < 2 || wgs * splits < 128 || (long)a.N * a.Cout * a.OH * a.OW * 4 * kPconvMaxSplits > (64L << 20)) return -1;
  }
  pl.splits = splits;
  pl.geo = geo; pl.bm = bm; pl.s2 = s2; pl.w32 = w32; pl.wide = wide; pl.half = half; pl.ptiles = ptiles; pl.oh = oh; pl.ow = ow;
  return 0;
}

bool pconv_eligible(const PConvArgs& a) { PConvPlan pl; return pconv_plan(a, pl) == 0; }
// Channel-blocked bf16 operands (a.x_blk / a.y_blk: [N][C/8][H][W][8]) — the forms compiled for them (pconv_k FEAT 64 / 128), bf16 arithmetic:
//   x and y blocked : 3x3 stride 1 (bias, input ReLU, output ReLU; with the few-channel shortcut of FEAT 16 on 64-channel tiles)
//   x blocked       : 4x4 stride 2 forward, fp32 NCHW output (the pooled second convolution of a discriminator block)
//   y blocked       : 1x1 with an fp32 NCHW x and an optional fp32 NCHW addend (the shortcut + sum of a discriminator block)
// No reduction split, no statistics rows, no input transform, no in-place accumulation, no mask.
bool pconv_takes_blocked(const PConvArgs& a) {
  PConvPlan pl;
  if (!(a.x_blk || a.y_blk) || pconv_plan(a, pl) != 0) return false;
  if (a.nsplit != 1 || a.up != 0 || a.accumulate || a.pos_mask || a.fold.scale || a.stats || pl.splits != 1) return false;
  if (a.x_blk && !(a.x_bf16 && a.Cin % 16 == 0)) return false;
  if (a.y_blk && !(a.y_bf16 && a.Cout % 8 == 0)) return false;
  if (a.sc_x && !(a.x_blk && a.y_blk)) return false;
  if (a.x_blk && a.y_blk) return a.ks == 3 && a.stride == 1 && pl.geo != 2;
  if (a.x_blk) return a.ks == 4 && a.stride == 2 && pl.geo != 5 && !a.addend && !a.sc_x && !a.y_bf16;
  return a.ks == 1 && !a.x_bf16 && !a.sc_x;      // y blocked only
}
int pconv_plan_splits(const PConvArgs& a) { PConvPlan pl; return pconv_plan(a, pl) == 0 ? pl.splits : -1; }
long pconv_stat_rows_max(int N, int OH, int OW) { return 2L * N * ((long)(OH * OW + 63) / 64) + 4; }   // (+4: the rounded-up last tile of the 4-column form)
long pconv_stat_row_floats(int Cout) { return 3L * Cout; }      // a partial row: [Cout][{count, mean, M2}]

#ifndef AGL_PCONV_XCD
#define AGL_PCONV_XCD 1
#endif
#ifndef AGL_PBWW_ROWS32
#define AGL_PBWW_ROWS32 1
#endif
// XCD-aware workgroup order of pconv_k (see there): (gx, gy, gz) -> (8 * ceil(gx * gy / 8), 1, gz) once the grid spans the XCDs
static void pconv_xcd_order(PArgs& p, dim3& g, int nzp = 1) {      // nzp: phases per pixel tile in g.z (phase kernels), else 1
  p.xgy = 1; p.xchunk = 0; p.xitems = 0;
  const long items = (long)g.x * g.y * nzp;
  if (!AGL_PCONV_XCD || items < 64 || items > (1L << 28)) return;
  p.xgy = (int)g.y; p.xitems = (int)items; p.xchunk = (int)((items + 7) / 8);
  g.x = 8u * (unsigned)p.xchunk; g.y = 1; g.z /= nzp;
}

int pconv_try(const PConvArgs& a, void* ws, long ws_bytes, hipStream_t st, const char* name) {
  PConvPlan pl;
  if (pconv_plan(a, pl) != 0) return -1;
  const long packed = a.packed ? 0 : pconv_ws_bytes(a.Cin, a.Cout, a.ks, a.nsplit);     // (pre-packed weights: ws holds the slabs only)
  const long out_numel = (long)a.N * a.Cout * a.OH * a.OW;
  const long need = packed + (pl.splits > 1 ? out_numel * 4 * pl.splits : 0);
  if (need > 0 && (!ws || ws_bytes < need)) return -1;
  const int geo = pl.geo, bm = pl.bm;
  const bool s2 = pl.s2, w32 = pl.w32, wide = pl.wide, half = pl.half;
  const long ptiles = pl.ptiles;
  (void)half;
  const int KK = a.ks * a.ks, nch = a.Cin / 16, mpad = round_up(a.Cout, 128);
  u32x4* wp = a.packed ? (u32x4*)a.packed : (u32x4*)ws;
  PArgs p;
  p.odiv = a.out_div; p.prio = a.prio;
  p.x = a.x; p.wp = wp; p.bias = a.bias; p.pos_mask = a.pos_mask; p.y = a.y;
  p.N = a.N; p.Cin = a.Cin; p.H = a.H; p.W = a.W; p.Cout = a.Cout; p.OH = a.OH; p.OW = a.OW; p.pad = a.pad; p.up = a.up;
  p.in_relu = a.in_relu; p.relu = a.relu; p.accumulate = a.accumulate; p.nch = nch; p.mpad = mpad;
  if (a.ks == 1) { p.H = p.OH = pl.oh; p.W = p.OW = pl.ow; }      // the re-read map of a 1x1 convolution
  p.stats = nullptr; p.slabs = nullptr; p.cps = nch; p.out_numel = out_numel; p.oh2 = 0; p.ow2 = 0;
  if (pl.splits > 1) { p.slabs = (float*)((char*)ws + packed); p.cps = agl_cdiv(nch, pl.splits); }
  // eight-wave workgroups (see pconv_k): the split-mode stride-1 3x3 / 5x5 kernels on tiles of >= 128 pixels
  // (measured, tools/conv_bench.py: no change on grids that fill the chip; -8..-14 % on the small ones — ConvLSTM recurrence steps,
  //  the decoder's 8x8 stem — where more waves per SIMD hide the staging latency nothing else covers: automatic below 512 workgroups)
  const bool small_grid = ptiles * agl_cdiv(a.Cout, bm) * (pl.splits > 1 ? pl.splits : 1) < 512;
  const bool w8 = (a.w8 || small_grid) && a.nsplit == 3 && !s2 && (a.ks == 3 || a.ks == 5) && geo != 2 && geo != 3;
  const int wcols = w8 ? 4 : 2;                 // wave columns = statistic rows per pixel tile
  // features compiled into some instantiations only (pconv_k FEAT): input transform + statistics rows in the 4x4 / stride-2 family,
  // statistics rows in the bf16 5x5 forms, bf16 output / addend in the bf16 3x3 stride-1 and 1x1 forms
  const bool fam42 = s2 && a.ks == 4, fam5 = !s2 && a.ks == 5 && a.nsplit == 1, famy = !s2 && a.nsplit == 1 && (a.ks == 3 || a.ks == 1);
  if (a.fold.scale && !(fam42 || fam5)) return -1;
  if (a.fold.cells && (a.up != 0 || a.x_bf16 || a.fold.per_n || !a.fold.map || a.fold.G <= 0 || a.Cin % 16 != 0 || a.Cin > 512 || s2)) return -1;      // (SPADE form: fp32 x, one map entry per pixel row / column)
  if ((a.y_bf16 || a.addend) && !famy) return -1;
  if (a.x_blk || a.y_blk) {      // (checked again below; here: keep the statistics rows off a blocked launch)
    if (a.stats) return -1;
  }
  if ((fam42 || fam5) && pl.splits == 1 && a.stats && !a.relu && !a.accumulate && !a.pos_mask && wcols * ptiles * a.Cout * 3 <= a.stats_floats) {
    p.stats = a.stats;
    *a.stat_rows = (int)(wcols * ptiles);
  }
  const int feat42 = (a.fold.scale || p.stats) ? 3 : 0, feat5 = (a.fold.scale || p.stats) ? 3 : 0, featy = (a.y_bf16 || a.addend) ? 4 : 0;
  const bool featsc = a.sc_x != nullptr;
  p.x_bf16 = a.x_bf16; p.mask_bf16 = a.mask_bf16;
  if (a.mask_bf16 && (pl.splits > 1 || !a.pos_mask)) return -1;      // (the slab reduction reads an fp32 mask)
  p.fold = a.fold; p.y_bf16 = a.y_bf16; p.addend = a.addend; p.blk = a.x_blk | (a.y_blk << 1); p.mask_blk = a.mask_blk;
  if ((a.x_blk || a.y_blk) && !pconv_takes_blocked(a)) return -1;
  if (a.mask_blk && (!a.mask_bf16 || a.bias)) return -1;      // (a blocked mask is applied to the accumulators, before any bias)
  if (a.mask_blk && !(a.ks == 3 && !s2 && a.nsplit == 1 && geo != 2 && !featy && !featsc && !a.x_blk && !a.y_blk)) return -1;      // (FEAT 256 instantiations)
  p.sc_x = a.sc_x; p.sc_w = a.sc_w; p.sc_b = a.sc_b; p.sc_cin = a.sc_cin;
  p.wexp = reinterpret_cast<const int*>(wp + (long)pack_planes(a.nsplit) * nch * 2 * KK * mpad);      // (read by the H16 instantiations only)
  if (a.sc_x && !(a.ks == 3 && !s2 && a.nsplit == 1 && bm == 64 && geo != 2 && pl.splits == 1 && a.sc_cin >= 1 && a.sc_cin <= 4 && a.OW % 4 == 0))
    return -1;                                                       // (compiled into the bf16 3x3 stride-1 instantiations with 64-channel tiles)
  if ((a.y_bf16 || a.addend) && pl.splits > 1) return -1;            // (... writes fp32 in place)
  if (a.y_bf16 && a.accumulate && !a.addend) return -1;              // (in-place accumulation reads y as fp32)
  p.x_bytes = (unsigned)((long)a.N * a.Cin * a.H * a.W * (a.x_bf16 ? 2 : 4));
  dim3 g((unsigned)ptiles, agl_cdiv(a.Cout, bm), pl.splits > 1 ? agl_cdiv(nch, p.cps) : 1);
  pconv_xcd_order(p, g);
  if (!a.packed) {
    const int prc = pconv_pack(a.w, wp, a.Cout, a.Cin, a.ks, a.w_sm, a.w_sc, a.flip, a.nsplit, 0, st, name);
    if (prc != AGL_OK) return prc;
  }
#define PC_LAUNCH(KS_, S_, TW_, TH_, TI_, BM_, NS_, TG_) \
  hipLaunchKernelGGL((pconv_k<KS_, S_, TW_, TH_, TI_, BM_, NS_, TG_>), g, dim3(NT), 0, st, p)
#define PC_LAUNCH_F(F_, KS_, S_, TW_, TH_, TI_, BM_, NS_, TG_) \
  hipLaunchKernelGGL((pconv_k<KS_, S_, TW_, TH_, TI_, BM_, NS_, TG_, false, false, 256, 0, false, F_>), g, dim3(NT), 0, st, p)
#define PC_LAUNCH_DB(KS_, S_, TW_, TH_, TI_, BM_, NS_, TG_) \
  hipLaunchKernelGGL((pconv_k<KS_, S_, TW_, TH_, TI_, BM_, NS_, TG_, false, true>), g, dim3(NT), 0, st, p)
#define PC_LAUNCH8(KS_, S_, TW_, TH_, TI_, BM_, NS_, TG_, DB_) \
  hipLaunchKernelGGL((pconv_k<KS_, S_, TW_, TH_, TI_, BM_, NS_, TG_, false, DB_, 512>), g, dim3(512), 0, st, p)
#define PC_SHAPES1(KS_, BM_, TG_) PC_SHAPES1F(0, KS_, BM_, TG_)
#define PC_SHAPES1F(F_, KS_, BM_, TG_)                                                                   \
  do {                                                                                                   \
    if (geo == 3) PC_LAUNCH_F(F_, KS_, 1, 8, 8, 1, BM_, 1, TG_);                                         \
    else if (geo == 0 && !wide) PC_LAUNCH_F(F_, KS_, 1, 16, 8, 1, BM_, 1, TG_);                          \
    else if (geo == 0 && w32) PC_LAUNCH_F(F_, KS_, 1, 32, 8, 1, BM_, 1, TG_);                            \
    else if (geo == 0) PC_LAUNCH_F(F_, KS_, 1, 16, 16, 1, BM_, 1, TG_);                                  \
    else if (geo == 1 && !wide) PC_LAUNCH_F(F_, KS_, 1, 8, 8, 2, BM_, 1, TG_);                           \
    else if (geo == 1) PC_LAUNCH_F(F_, KS_, 1, 8, 8, 4, BM_, 1, TG_);                                    \
  } while (0)
#define PC_SHAPES3(KS_, TG_)                                                                             \
  do {                                                                                                   \
    if (geo == 3) PC_LAUNCH(KS_, 1, 8, 8, 1, 64, SPL, TG_);                                                \
    else if (geo == 0 && w32) PC_LAUNCH(KS_, 1, 32, 4, 1, 64, SPL, TG_);                                   \
    else if (geo == 0) PC_LAUNCH(KS_, 1, 16, 8, 1, 64, SPL, TG_);                                          \
    else if (geo == 1) PC_LAUNCH(KS_, 1, 8, 8, 2, 64, SPL, TG_);                                           \
  } while (0)
#define PC_STRIDE2(F_, KS_, TG_)                                                                         \
  do {                                                                                                   \
    if (a.nsplit == 3) {                                                                                 \
      if (geo == 0) PC_LAUNCH_F(F_, KS_, 2, 16, 4, 1, 64, SPL, TG_);                                       \
      else if (geo == 1) PC_LAUNCH_F(F_, KS_, 2, 8, 8, 1, 64, SPL, TG_);                                   \
      else if (geo == 5) PC_LAUNCH_F(F_, KS_, 2, 2, 2, 16, 64, SPL, TG_);                                  \
      else PC_LAUNCH_F(F_, KS_, 2, 4, 4, 4, 64, SPL, TG_);                                                 \
    } else if (bm == 128) {                                                                              \
      if (geo == 0) PC_LAUNCH_F(F_, KS_, 2, 16, 8, 1, 128, 1, TG_);                                      \
      else if (geo == 1) PC_LAUNCH_F(F_, KS_, 2, 8, 8, 2, 128, 1, TG_);                                  \
      else if (geo == 5) PC_LAUNCH_F(F_, KS_, 2, 2, 2, 32, 128, 1, TG_);                                 \
      else PC_LAUNCH_F(F_, KS_, 2, 4, 4, 8, 128, 1, TG_);                                                \
    } else {                                                                                             \
      if (geo == 0) PC_LAUNCH_F(F_, KS_, 2, 16, 8, 1, 64, 1, TG_);                                       \
      else if (geo == 1) PC_LAUNCH_F(F_, KS_, 2, 8, 8, 2, 64, 1, TG_);                                   \
      else if (geo == 5) PC_LAUNCH_F(F_, KS_, 2, 2, 2, 32, 64, 1, TG_);                                  \
      else PC_LAUNCH_F(F_, KS_, 2, 4, 4, 8, 64, 1, TG_);                                                 \
    }                                                                                                    \
  } while (0)
#define PC_1X1(F_, BM_, NS_)                                                                             \
  do {                                                                                                   \
    if (geo == 0) PC_LAUNCH_F(F_, 1, 1, 16, 8, 1, BM_, NS_, 1);                                          \
    else if (geo == 1) PC_LAUNCH_F(F_, 1, 1, 8, 8, 2, BM_, NS_, 1);                                      \
    else PC_LAUNCH_F(F_, 1, 1, 4, 4, 8, BM_, NS_, 1);                                                    \
  } while (0)
  if (s2) {
    if (a.ks == 4 && a.x_blk) {      // blocked bf16 x, fp32 NCHW y
      if (bm == 128) { if (geo == 0) PC_LAUNCH_F(64, 4, 2, 16, 8, 1, 128, 1, 4); else if (geo == 1) PC_LAUNCH_F(64, 4, 2, 8, 8, 2, 128, 1, 4); else PC_LAUNCH_F(64, 4, 2, 4, 4, 8, 128, 1, 4); }
      else { if (geo == 0) PC_LAUNCH_F(64, 4, 2, 16, 8, 1, 64, 1, 4); else if (geo == 1) PC_LAUNCH_F(64, 4, 2, 8, 8, 2, 64, 1, 4); else PC_LAUNCH_F(64, 4, 2, 4, 4, 8, 64, 1, 4); }
    }
    else if (a.ks == 4) { if (feat42) PC_STRIDE2(3, 4, 4); else PC_STRIDE2(0, 4, 4); } else PC_STRIDE2(0, 3, 3);
  } else if (a.ks == 1) {
    if (a.nsplit == 3) { if (bm == 128) PC_1X1(0, 128, SPL); else PC_1X1(0, 64, SPL); }
    else if (a.y_blk) { if (bm == 128) PC_1X1(128, 128, 1); else PC_1X1(128, 64, 1); }
    else if (featy) { if (bm == 128) PC_1X1(4, 128, 1); else PC_1X1(4, 64, 1); }
    else { if (bm == 128) PC_1X1(0, 128, 1); else PC_1X1(0, 64, 1); }
  } else if (SPL == 3 && a.ablate > 0 && a.ks == 3 && a.nsplit == 3 && wide && w32) {      // diagnostic builds of ONE geometry (tools/ablate.sh)
#define PC_ABL(A_) hipLaunchKernelGGL((pconv_k<3, 1, 32, 8, 1, 64, 3, 3, false, false, 256, A_>), g, dim3(256), 0, st, p)
    switch (a.ablate) {
      case 1: PC_ABL(1); break; case 2: PC_ABL(2); break; case 4: PC_ABL(4); break; case 8: PC_ABL(8); break;
      case 14: PC_ABL(14); break; case 30: PC_ABL(30); break; case 31: PC_ABL(31); break; case 16: PC_ABL(16); break;
      default: PC_ABL(6); break;
    }
#undef PC_ABL
  } else if (a.ablate > 0 && a.ks == 3 && a.nsplit == 1 && geo == 0 && w32 && bm == 64 && !featy && !featsc) {      // the same for the bf16 form
#define PC_ABL(A_) hipLaunchKernelGGL((pconv_k<3, 1, 32, 8, 1, 64, 1, 9, false, false, 256, A_>), g, dim3(256), 0, st, p)
    switch (a.ablate) {      // (the flag field holds 5 bits: code 2 stands for bit 32, "no output store")
      case 1: PC_ABL(1); break; case 8: PC_ABL(8); break; case 24: PC_ABL(24); break; case 2: PC_ABL(32); break;
      case 3: PC_ABL(33); break;
      // occupancy probes: the unmodified kernel with idle dynamic LDS on top of its 29 KB, so that two / one workgroups fit a CU instead of three
      case 11: hipLaunchKernelGGL((pconv_k<3, 1, 32, 8, 1, 64, 1, 9>), g, dim3(256), 40 * 1024, st, p); break;
      case 12: hipLaunchKernelGGL((pconv_k<3, 1, 32, 8, 1, 64, 1, 9>), g, dim3(256), 90 * 1024, st, p); break;
      default: PC_ABL(57); break;
    }
#undef PC_ABL
  } else if (w8 && a.ks == 3) {
    if (wide) { if (w32) PC_LAUNCH8(3, 1, 32, 8, 1, 64, SPL, STG, false); else PC_LAUNCH8(3, 1, 16, 16, 1, 64, SPL, STG, false); }
    else if (bm == 128) {
      if (geo == 0 && w32) PC_LAUNCH8(3, 1, 32, 4, 1, 128, SPL, 3, false);
      else if (geo == 0) PC_LAUNCH8(3, 1, 16, 8, 1, 128, SPL, 3, false);
      else PC_LAUNCH8(3, 1, 8, 8, 2, 128, SPL, 3, false);
    } else {
      if (geo == 0 && w32) PC_LAUNCH8(3, 1, 32, 4, 1, 64, SPL, 9, false);
      else if (geo == 0) PC_LAUNCH8(3, 1, 16, 8, 1, 64, SPL, 9, false);
      else PC_LAUNCH8(3, 1, 8, 8, 2, 64, SPL, 9, false);
    }
  } else if (w8 && a.ks == 5) {
    if (geo == 0 && w32) PC_LAUNCH8(5, 1, 32, 4, 1, 64, SPL, 5, true);
    else if (geo == 0) PC_LAUNCH8(5, 1, 16, 8, 1, 64, SPL, 5, true);
    else PC_LAUNCH8(5, 1, 8, 8, 2, 64, SPL, 5, true);
  } else if (a.ks == 3) {
    if (geo == 2) {
      if (a.nsplit == 3) PC_LAUNCH_DB(3, 1, 4, 4, 8, 64, SPL, 3);    // one kernel row per stage, two weight + two patch buffers
      else if (featy) {
        if (wide) { if (bm == 128) PC_LAUNCH_F(4, 3, 1, 4, 4, 16, 128, 1, 9); else PC_LAUNCH_F(4, 3, 1, 4, 4, 16, 64, 1, 9); }
        else { if (bm == 128) PC_LAUNCH_F(4, 3, 1, 4, 4, 8, 128, 1, 9); else PC_LAUNCH_F(4, 3, 1, 4, 4, 8, 64, 1, 9); }
      }
      else if (wide) { if (bm == 128) PC_LAUNCH(3, 1, 4, 4, 16, 128, 1, 9); else PC_LAUNCH(3, 1, 4, 4, 16, 64, 1, 9); }
      else { if (bm == 128) PC_LAUNCH(3, 1, 4, 4, 8, 128, 1, 9); else PC_LAUNCH(3, 1, 4, 4, 8, 64, 1, 9); }
    } else if (a.nsplit == 1 && a.x_blk && a.y_blk && featsc) { PC_SHAPES1F(208, 3, 64, 9);      // (pconv_takes_blocked)
    } else if (a.nsplit == 1 && a.x_blk && a.y_blk) {
      if (bm == 128) PC_SHAPES1F(192, 3, 128, 9); else PC_SHAPES1F(192, 3, 64, 9);
    } else if (a.nsplit == 1 && a.mask_blk) { if (bm == 128) PC_SHAPES1F(256, 3, 128, 9); else PC_SHAPES1F(256, 3, 64, 9);
    } else if (a.nsplit == 1 && featsc) { PC_SHAPES1F(20, 3, 64, 9); }
    else if (a.nsplit == 1 && featy) { if (bm == 128) PC_SHAPES1F(4, 3, 128, 9); else PC_SHAPES1F(4, 3, 64, 9); }
    else if (a.nsplit == 1) { if (bm == 128) PC_SHAPES1(3, 128, 9); else PC_SHAPES1(3, 64, 9); }
    else if (wide) { if (w32) PC_LAUNCH(3, 1, 32, 8, 1, 64, SPL, STG); else PC_LAUNCH(3, 1, 16, 16, 1, 64, SPL, STG); }
    else if (bm == 128) {
      if (geo == 3) PC_LAUNCH(3, 1, 8, 8, 1, 128, SPL, 3);
      else if (geo == 0 && w32) PC_LAUNCH(3, 1, 32, 4, 1, 128, SPL, 3);
      else if (geo == 0) PC_LAUNCH(3, 1, 16, 8, 1, 128, SPL, 3);
      else PC_LAUNCH(3, 1, 8, 8, 2, 128, SPL, 3);
    }
    else PC_SHAPES3(3, 9);
  } else {
    if (a.nsplit == 1 && feat5) { if (bm == 128) PC_SHAPES1F(3, 5, 128, 5); else PC_SHAPES1F(3, 5, 64, 5); }
    else if (a.nsplit == 1) { if (bm == 128) PC_SHAPES1(5, 128, 5); else PC_SHAPES1(5, 64, 5); }
    else if (geo == 3) PC_LAUNCH_DB(5, 1, 8, 8, 1, 64, SPL, 5);
    else if (geo == 0 && w32) PC_LAUNCH_DB(5, 1, 32, 4, 1, 64, SPL, 5);
    else if (geo == 0) PC_LAUNCH_DB(5, 1, 16, 8, 1, 64, SPL, 5);
    else if (geo == 1) PC_LAUNCH_DB(5, 1, 8, 8, 2, 64, SPL, 5);
  }
#undef PC_STRIDE2
#undef PC_1X1
#undef PC_SHAPES1
#undef PC_SHAPES1F
#undef PC_LAUNCH_F
#undef PC_SHAPES3
#undef PC_LAUNCH
#undef PC_LAUNCH_DB
#undef PC_LAUNCH8
  AGL_CHECK_LAUNCH(name);
  if (pl.splits > 1)
    return agl_launch_splitk_epilogue(p.slabs, a.y, out_numel, (int)g.z, a.OH * a.OW, a.Cout, a.bias, a.pos_mask, a.accumulate, a.relu, st, name,
                                      a.out_div);
  return AGL_OK;
}

void pconv_pack_desc(const float* w, void* packed, int M, int Cred, int ks, int w_sm, int w_sc, int flip, int nsplit, int phase4, long long* row) {
  const int KK = phase4 ? 16 : ks * ks, nch = Cred / 16, mpad = round_up(M, 128);
  const long per_plane = (long)nch * 2 * KK * mpad;
  row[0] = (long long)(uintptr_t)w; row[1] = (long long)(uintptr_t)packed; row[2] = M; row[3] = Cred; row[4] = KK; row[5] = w_sm; row[6] = w_sc;
  row[7] = flip; row[8] = mpad; row[9] = pack_planes(nsplit); row[10] = phase4; row[11] = per_plane; row[12] = 0;
  row[13] = (per_plane + 255) / 256;
}
int pconv_pack_many(const void* rows_dev, int n, long total_blocks, hipStream_t st, const char* name) {
  if (SPL == 2) {      // the chunk scales of the fp16 hi / lo rows first (rows of the one-plane form return at once)
    hipLaunchKernelGGL(h16_scan_many_k, dim3(64, (unsigned)n, H16_SCAN_G), dim3(1024), 0, st, (const long long*)rows_dev);
    AGL_CHECK_LAUNCH(name);
  }
  hipLaunchKernelGGL(pack_many_k, dim3((unsigned)total_blocks), dim3(256), 0, st, (const long long*)rows_dev, n);
  AGL_CHECK_LAUNCH(name);
  return AGL_OK;
}
int pconv_pack(const float* w, void* packed, int M, int Cred, int ks, int w_sm, int w_sc, int flip, int nsplit, int phase4, hipStream_t st,
               const char* name) {
  const int KK = phase4 ? 16 : ks * ks, nch = Cred / 16, mpad = round_up(M, 128);
  const long per_plane = (long)nch * 2 * KK * mpad;
  if (pack_planes(nsplit) == 2) {
    hipLaunchKernelGGL(h16_scan_k, dim3((unsigned)nch, H16_SCAN_G), dim3(1024), 0, st, w, (u32x4*)packed, M, Cred, KK, w_sm, w_sc, per_plane);
    AGL_CHECK_LAUNCH(name);
    hipLaunchKernelGGL(pack_weights_h16_k, dim3((unsigned)((per_plane + 255) / 256)), dim3(256), 0, st, w, (u32x4*)packed, M, Cred, KK, w_sm, w_sc,
                       flip, mpad, nch, phase4);
  } else
    hipLaunchKernelGGL(pack_weights_k, dim3((unsigned)((per_plane + 255) / 256)), dim3(256), 0, st, w, (u32x4*)packed, M, Cred, KK, w_sm, w_sc,
                       flip, mpad, nch, nsplit, phase4);
  AGL_CHECK_LAUNCH(name);
  return AGL_OK;
}

// ---- few channels on one side, many on the other, ks x ks window (decoder c4 / c7: 64|128 -> 3, 7x7; the input gradients of the
// 3-channel first layers CropEncoder.c1 / Decoder.c5) -------------------------------------------------------------------------
// A matrix tile wants >= 32 rows; three output channels fill 3.  Decomposition: y[o][y][x] = sum_kw P[(kw,o)][y][x + kw - pad] with
// P[(kw,o)][y][x'] = sum_{c,kh} x[c][y + kh - pad][x'] * w[o][c][kh][kw] — a ks x 1 (vertical) convolution with ks*CO <= 28 output
// "channels" on the matrix cores (pconv_k<.., VERT>: reduction Cred*ks, one tile row of 64 channels), followed by a diagonal sum of
// ks shifted planes (vert_diag_sum_k: P is ks*CO/Cred of the input's size).  Replaces small_cout_conv (fp32 VALU, ~27 TFLOP/s) in
// the bf16 / split modes.
namespace {
// wp[plane][cc][h][kh][m][j] = term_plane(w[o*w_so + (16cc+8h+j)*w_sc + kh'*ks + kw']), m = kw*CO + o (zero rows for m >= ks*CO)
__global__ void pack_vert_k(const float* __restrict__ w, u32x4* __restrict__ wp, int CO, int Cred, int ks, int w_so, int w_sc, int flip,
                            int mpad, int nch, int nsplit) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per_plane = (long)nch * 2 * ks * mpad;
  if (i >= per_plane) return;
  const int m = (int)(i % mpad);
  long r = i / mpad;
  const int kh = (int)(r % ks); r /= ks;
  const int h = (int)(r & 1), cc = (int)(r >> 1);
  const int kw = m / CO, o = m - kw * CO;
  const int st = flip ? (ks - 1 - kh) * ks + (ks - 1 - kw) : kh * ks + kw;
  bf16x8 t0, t1, t2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 16 * cc + 8 * h + j;
    const float v = (m < ks * CO && c < Cred) ? w[(long)o * w_so + (long)c * w_sc + st] : 0.f;
    if (nsplit == 1) { t0[j] = (__bf16)v; }
    else { __bf16 a, b, d; split3(v, a, b, d); t0[j] = a; t1[j] = b; t2[j] = d; }
  }
  wp[i] = __builtin_bit_cast(u32x4, t0);
  if (nsplit == 3) {
    wp[per_plane + i] = __builtin_bit_cast(u32x4, t1);
    wp[2 * per_plane + i] = __builtin_bit_cast(u32x4, t2);
  }
}

// y[n][o][yy][x] = epilogue( sum_kw P[n][kw*CO + o][yy][x + kw - pad] + bias[o] ); columns outside the map contribute nothing
__global__ void vert_diag_sum_k(const float* __restrict__ P, const float* __restrict__ bias, const float* __restrict__ pos_mask,
                                float* __restrict__ y, long total, int CO, int HW, int W, int ks, int pad, int relu, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % W);
  const long plane = i / HW;                       // n*CO + o
  const int o = (int)(plane % CO);
  const long n = plane / CO;
  const long pix = i - plane * HW;
  const float* Pn = P + (n * ks * CO) * HW + pix;
  float v = 0.f;
  for (int kw = 0; kw < ks; ++kw) {
    const int xs = x + kw - pad;
    if (xs >= 0 && xs < W) v += Pn[(long)(kw * CO + o) * HW + (kw - pad)];
  }
  if (bias) v += bias[o];
  if (pos_mask && !(pos_mask[i] > 0.f)) v = 0.f;
  if (accumulate) v += y[i];
  if (relu) v = fmaxf(v, 0.f);
  y[i] = v;
}
}  // namespace

static bool pconv_vert_ok(int N, int Cred, int H, int W, int CO, int ks, int nsplit) {
  return ks == 7 && CO >= 1 && CO <= 4 && Cred % 16 == 0 && Cred >= 32 && W % 32 == 0 && H % 8 == 0 && (nsplit == 1 || nsplit == 3) &&
         (long)N * Cred * H * W < (1L << 29);
}
long pconv_vert_ws_bytes(int N, int Cred, int H, int W, int CO, int ks, int nsplit) {
  if (!pconv_vert_ok(N, Cred, H, W, CO, ks, nsplit)) return 0;
  const long packed = (long)nsplit * (Cred / 16) * 2 * ks * 128 * 16;
  return packed + (long)N * ks * CO * H * W * 4;
}
int pconv_vert_try(const PVertArgs& a, void* ws, long ws_bytes, hipStream_t st, const char* name) {
  if (!pconv_vert_ok(a.N, a.Cred, a.H, a.W, a.CO, a.ks, a.nsplit)) return -1;
  const long need = pconv_vert_ws_bytes(a.N, a.Cred, a.H, a.W, a.CO, a.ks, a.nsplit);
  if (!ws || ws_bytes < need) return -1;
  const int nch = a.Cred / 16, mpad = 128, M = a.ks * a.CO;
  const long packed = (long)a.nsplit * nch * 2 * a.ks * mpad * 16;
  u32x4* wp = (u32x4*)ws;
  float* P = (float*)((char*)ws + packed);
  const long per_plane = (long)nch * 2 * a.ks * mpad;
  hipLaunchKernelGGL(pack_vert_k, dim3((unsigned)((per_plane + 255) / 256)), dim3(256), 0, st, a.w, wp, a.CO, a.Cred, a.ks, a.w_so, a.w_sc,
                     a.flip, mpad, nch, a.nsplit);
  AGL_CHECK_LAUNCH(name);
  PArgs p;
  p.odiv = nullptr; p.prio = 0; p.oh2 = 0; p.ow2 = 0;
  p.x = a.x; p.wp = wp; p.bias = nullptr; p.pos_mask = nullptr; p.y = P;
  p.N = a.N; p.Cin = a.Cred; p.H = a.H; p.W = a.W; p.Cout = M; p.OH = a.H; p.OW = a.W; p.pad = a.pad; p.up = 0;
  p.in_relu = a.in_relu; p.relu = 0; p.accumulate = 0; p.nch = nch; p.mpad = mpad;
  p.stats = nullptr; p.slabs = nullptr; p.cps = nch; p.out_numel = (long)a.N * M * a.H * a.W;
  p.x_bf16 = a.x_bf16; p.mask_bf16 = 0; p.fold = a.fold; p.y_bf16 = 0; p.addend = nullptr; p.sc_x = nullptr;
  if (a.fold.scale && (a.nsplit != 1 || a.x_bf16 || !a.fold.cells || !a.fold.map || a.fold.per_n || a.Cred > 512)) return -1;      // (the transform: SPADE form, bf16 arithmetic, fp32 x)
  p.wexp = nullptr; p.blk = 0; p.mask_blk = 0;
  p.x_bytes = (unsigned)((long)a.N * a.Cred * a.H * a.W * (a.x_bf16 ? 2 : 4));
  if (a.x_bf16 && a.nsplit != 1) return -1;
  if (a.nsplit == 1) {
    dim3 g((unsigned)((long)a.N * (a.H / 8) * (a.W / 32)), 1, 1);
    pconv_xcd_order(p, g);
    hipLaunchKernelGGL((pconv_k<7, 1, 32, 8, 1, 64, 1, 7, false, false, 256, 0, true, 1>), g, dim3(256), 0, st, p);
  } else {
    dim3 g((unsigned)((long)a.N * (a.H / 4) * (a.W / 32)), 1, 1);
    pconv_xcd_order(p, g);
    hipLaunchKernelGGL((pconv_k<7, 1, 32, 4, 1, 64, 3, 7, false, false, 256, 0, true>), g, dim3(256), 0, st, p);
  }
  AGL_CHECK_LAUNCH(name);
  const long total = (long)a.N * a.CO * a.H * a.W;
  hipLaunchKernelGGL(vert_diag_sum_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)P, a.bias, a.pos_mask, a.y, total,
                     a.CO, a.H * a.W, a.W, a.ks, a.pad, a.relu, a.accumulate);
  AGL_CHECK_LAUNCH(name);
  return AGL_OK;
}

// ---- 4x4 / stride-2 / pad-1 input gradient (and ConvTranspose2d(4,2,1) forward): four 2x2-tap phases, blockIdx.z = phase ----
// a.x = dy (N, a.Cin = reduction channels, a.H x a.W), a.y = dx (N, a.Cout, 2H x 2W); w[co][ci][4][4] addressed through
// w_sm (stride of the output row m = ci: 16) and w_sc (stride of the reduction channel c = co: Cin*16).
static int pconvT_plan(const PConvArgs& a, int* geo, int* bm, long* ptiles, int* splits_out = nullptr) {
  if (a.ks != 4 || a.stride != 2 || a.pad != 1 || a.up != 0 || a.Cin % 16 != 0 || a.Cout < 48) return -1;
  if (!(a.nsplit == 1 || a.nsplit == 3)) return -1;
  // odd-sized inputs (33 x 33, 65 x 65: the layout encoder's c3): the phases cover rows / columns 0 .. 2H-1; the last row and column
  // (one tap each) come from phase_edge_k
  const bool odd = a.OH == 2 * a.H + 1 && a.OW == 2 * a.W + 1;
  if (!odd && (a.OH != 2 * a.H || a.OW != 2 * a.W)) return -1;
  if (odd && !(a.H == a.W && (a.W == 8 || a.W == 16 || a.W == 32))) return -1;      // paired-phase geometries, square maps (phase_edge_k)
  if (a.W % 16 == 0 && a.H % 8 == 0) *geo = 0;
  else if (a.W == 8 && a.H == 8) *geo = 1;
  else if (a.W == 4 && a.H == 4) *geo = 2;
  else if (a.W == 2 && a.H == 2) *geo = 3;
  else return -1;
  if ((long)a.N * a.Cin * a.H * a.W >= (1L << 29) || (long)a.N * a.Cout * a.OH * a.OW >= (1L << 30)) return -1;
  *bm = (a.nsplit == 3 || a.Cout <= 64) ? 64 : 128;
  *ptiles = *geo == 0 ? (long)a.N * (a.H / 8) * (a.W / 16) : (*geo == 1 ? agl_cdiv(a.N, 2) : (*geo == 2 ? agl_cdiv(a.N, 8) : agl_cdiv(a.N, 32)));
  // fp16 hi / lo split form, reductions of more than 32 (chunk, tap) steps: 64-pixel tiles (geo + 10) — two accumulator blocks per wave
  // for the two column phases, which leaves registers for the fp32 total the hi*hi accumulator is emptied into (pconv_k FLUSH); without
  // it a 128-step chain reached 2.7x the exact fp32 kernel's distance to fp64
  if (a.nsplit == 3 && SPL == 2 && (a.Cin / 16) * 4 > AGL_H16_PHASE_LONG && *geo != 3) { *geo += 10; *ptiles = *geo == 10 ? *ptiles * 2 : (*geo == 11 ? a.N : agl_cdiv(a.N, 4)); }
  int splits = 1;
  const long wgs = *ptiles * agl_cdiv(a.Cout, *bm) * 4;         // (in units of single-phase workgroups)
  if (wgs < 200 && !a.any_grid) {      // small grid: cut the reduction (paired-phase geometries only: they write whole slab rows)
    const int nch = a.Cin / 16;
    splits = std::min(kPconvMaxSplits, nch / 4);
    while (splits > 2 && wgs * (splits / 2) >= 512) splits /= 2;
    if (*geo % 10 == 3 || odd || splits < 2 || wgs * splits < 256 || (long)a.N * a.Cout * a.OH * a.OW * 4 * kPconvMaxSplits > (64L << 20)) return -1;
  }
  if (splits_out) *splits_out = splits;
  return 0;
}
bool pconvT_eligible(const PConvArgs& a) { int g, b; long t; return pconvT_plan(a, &g, &b, &t) == 0; }
// ... and in the form that reads a bf16 pos_mask (bf16 arithmetic, even size, paired phases, no reduction split)
bool pconvT_takes_bf16_mask(const PConvArgs& a) {
  int g, b, sp; long t;
  return pconvT_plan(a, &g, &b, &t, &sp) == 0 && sp == 1 && g % 10 != 3 && a.nsplit == 1 && a.OH == 2 * a.H && a.OW == 2 * a.W;
}
long pconvT_ws_bytes(int Cred, int Crow, int nsplit) {
  if (Cred % 16 != 0 || Crow < 48) return 0;
  return (long)pack_planes(nsplit) * (Cred / 16) * 2 * 16 * round_up(Crow, 128) * 16 + pack_tail_bytes(Cred / 16, nsplit);
}
long pconvT_ws_bytes_split(int Cred, int Crow, int nsplit, long out_numel) {
  const long packed = pconvT_ws_bytes(Cred, Crow, nsplit);
  if (!packed) return 0;
  const long slabs = out_numel * 4 * kPconvMaxSplits;
  return slabs <= (64L << 20) ? packed + slabs : packed;
}
int pconvT_try(const PConvArgs& a, void* ws, long ws_bytes, hipStream_t st, const char* name) {
  int geo, bm, splits; long ptiles;
  if (pconvT_plan(a, &geo, &bm, &ptiles, &splits) != 0) return -1;
  if (a.OH == 2 * a.H + 1 && (!a.w || a.x_bf16)) return -1;      // (the edge kernel of the odd-sized form reads the unpacked weights and an fp32 dy)
  const long packed = a.packed ? 0 : pconvT_ws_bytes(a.Cin, a.Cout, a.nsplit), out_numel = (long)a.N * a.Cout * a.OH * a.OW;
  const long need = packed + (splits > 1 ? out_numel * 4 * splits : 0);
  if (need > 0 && (!ws || ws_bytes < need)) return -1;
  const int nch = a.Cin / 16, mpad = round_up(a.Cout, 128);
  u32x4* wp = a.packed ? (u32x4*)a.packed : (u32x4*)ws;
  if (!a.packed) {
    const int prc = pconv_pack(a.w, wp, a.Cout, a.Cin, 4, a.w_sm, a.w_sc, 0, a.nsplit, 1, st, name);
    if (prc != AGL_OK) return prc;
  }
  PArgs p;
  p.odiv = a.out_div; p.prio = a.prio;
  p.x = a.x; p.wp = wp; p.bias = a.bias; p.pos_mask = a.pos_mask; p.y = a.y;
  p.N = a.N; p.Cin = a.Cin; p.H = a.H; p.W = a.W; p.Cout = a.Cout; p.OH = a.H; p.OW = a.W;      // tiles run over the dy map
  p.pad = 0; p.up = 0; p.in_relu = 0; p.relu = a.relu; p.accumulate = a.accumulate; p.nch = nch; p.mpad = mpad; p.stats = nullptr;
  p.slabs = nullptr; p.cps = nch; p.out_numel = out_numel; p.oh2 = a.OH; p.ow2 = a.OW;
  if (splits > 1) { p.slabs = (float*)((char*)ws + packed); p.cps = agl_cdiv(nch, splits); splits = agl_cdiv(nch, p.cps); }
  p.wexp = reinterpret_cast<const int*>(wp + (long)pack_planes(a.nsplit) * nch * 2 * 16 * mpad);
  p.x_bf16 = a.x_bf16; p.mask_bf16 = a.mask_bf16; p.fold = InFold{nullptr, nullptr, nullptr, 0}; p.y_bf16 = 0; p.addend = nullptr; p.sc_x = nullptr;
  p.blk = 0; p.mask_blk = a.mask_blk;
  if (a.mask_blk && (!a.mask_bf16 || a.bias)) return -1;      // (a blocked mask is applied to the accumulators, before any bias)
  if (a.x_bf16 && a.nsplit != 1) return -1;
  // bf16 ReLU mask: the paired-phase epilogue of the bf16 instantiations (16-byte pieces of 8 mask elements), no reduction split, even size
  if (a.mask_bf16 && (a.nsplit != 1 || !a.pos_mask || splits > 1 || geo == 3 || a.OH != 2 * a.H)) return -1;
  p.x_bytes = (unsigned)((long)a.N * a.Cin * a.H * a.W * (a.x_bf16 ? 2 : 4));
  dim3 g((unsigned)ptiles, agl_cdiv(a.Cout, bm), (geo == 3 ? 4 : 2) * splits);      // 2x2 maps: one workgroup per phase; else per row phase
  pconv_xcd_order(p, g, geo == 3 ? 4 : 2);
#define PT_LAUNCH(TW_, TH_, TI_, BM_, NS_) hipLaunchKernelGGL((pconv_k<2, 1, TW_, TH_, TI_, BM_, NS_, 4, true>), g, dim3(NT), 0, st, p)
#define PT_GEO(BM_, NS_)                                          \
  do {                                                            \
    if (geo == 0) PT_LAUNCH(16, 8, 1, BM_, NS_);                  \
    else if (geo == 1) PT_LAUNCH(8, 8, 2, BM_, NS_);              \
    else if (geo == 2) PT_LAUNCH(4, 4, 8, BM_, NS_);              \
    else PT_LAUNCH(2, 2, 32, BM_, NS_);                           \
  } while (0)
#define PT_LAUNCH_M(TW_, TH_, TI_, BM_) \
  hipLaunchKernelGGL((pconv_k<2, 1, TW_, TH_, TI_, BM_, 1, 4, true, false, 256, 0, false, 8>), g, dim3(NT), 0, st, p)
#define PT_GEO_M(BM_)                                             \
  do {                                                            \
    if (geo == 0) PT_LAUNCH_M(16, 8, 1, BM_);                     \
    else if (geo == 1) PT_LAUNCH_M(8, 8, 2, BM_);                 \
    else PT_LAUNCH_M(4, 4, 8, BM_);                               \
  } while (0)
  if (a.mask_bf16) { if (bm == 128) PT_GEO_M(128); else PT_GEO_M(64); }
  else if (geo >= 10) {      // (SPL == 2 only: 64-pixel tiles)
    if (geo == 10) PT_LAUNCH(16, 4, 1, 64, SPL); else if (geo == 11) PT_LAUNCH(8, 8, 1, 64, SPL); else PT_LAUNCH(4, 4, 4, 64, SPL);
  }
  else if (a.nsplit == 3) PT_GEO(64, SPL);
  else if (bm == 128) PT_GEO(128, 1);
  else PT_GEO(64, 1);
#undef PT_GEO_M
#undef PT_LAUNCH_M
#undef PT_GEO
#undef PT_LAUNCH
  AGL_CHECK_LAUNCH(name);
  if (a.OH == 2 * a.H + 1) {      // odd-sized input: its last row and column
    const int ei_max = a.W == 8 ? phase_edge_images<8>() : (a.W == 16 ? phase_edge_images<16>() : phase_edge_images<32>());
    const int ei = 2L * agl_cdiv(a.N, ei_max) * agl_cdiv(a.Cout, 128) >= 512 ? ei_max : 1;
    dim3 ge((unsigned)(2 * agl_cdiv(a.N, ei)), agl_cdiv(a.Cout, 128));
    // (the edge kernel reads the UNPACKED a.w, which by the ABI contract already is w0 / *out_div: no divisor here — the packed
    //  phases above read w0 and divide in their epilogue)
#define PE_LAUNCH(W_)                                                                                                                      \
  do {                                                                                                                                     \
    if (ei == 1) hipLaunchKernelGGL((phase_edge_k<W_, 1>), ge, dim3(128), 0, st, a.x, a.w, a.pos_mask, a.y, (const float*)nullptr, a.N,  \
                                    a.Cin, a.Cout, a.w_sm, a.w_sc, a.relu, a.accumulate, a.nsplit == 1);                                 \
    else hipLaunchKernelGGL((phase_edge_k<W_, phase_edge_images<W_>()>), ge, dim3(128), 0, st, a.x, a.w, a.pos_mask, a.y,                \
                            (const float*)nullptr, a.N, a.Cin, a.Cout, a.w_sm, a.w_sc, a.relu, a.accumulate, a.nsplit == 1);             \
  } while (0)
    if (a.W == 8) PE_LAUNCH(8); else if (a.W == 16) PE_LAUNCH(16); else PE_LAUNCH(32);
#undef PE_LAUNCH
    AGL_CHECK_LAUNCH(name);
  }
  if (splits > 1)
    return agl_launch_splitk_epilogue(p.slabs, a.y, out_numel, splits, a.OH * a.OW, a.Cout, a.bias, a.pos_mask, a.accumulate, a.relu, st, name,
                                      a.out_div);
  return AGL_OK;
}

// ---- weight gradient -------------------------------------------------------------------------------------------------
static int pbww_plan(const PBwwArgs& a0, int* splits, int* tps, long* tiles_out, int* rt, int* ct, int* half_out, int* oh_out = nullptr,
                     int* ow_out = nullptr) {
  PBwwArgs a = a0;
  const bool s2 = a.stride == 2;
  if (s2 ? !((a.ks == 4 && a.pad == 1) || (a.ks == 3 && a.pad == 0)) || a.up != 0
         : !(a.stride == 1 && (a.ks == 3 || a.ks == 5 || (a.ks == 1 && a.pad == 0 && a.up == 0)))) return -1;
  if (a.ks == 1) {      // no spatial structure: the map is re-read as (HW/16) x 16, or as it is when 8 x 8 (cf. pconv_plan)
    const int hw = a.OH * a.OW;
    if (a.H != a.OH || a.W != a.OW) return -1;
    if (hw % 128 == 0) { a.OW = a.W = 16; a.OH = a.H = hw / 16; }
    else if (hw == 64) { a.OW = a.W = 8; a.OH = a.H = 8; }
    else if (hw == 16) { a.OW = a.W = 4; a.OH = a.H = 4; }
    else return -1;
  }
  if (oh_out) { *oh_out = a.OH; *ow_out = a.OW; }
  if (a.Cin % 16 != 0 || a.Cout < 32) return -1;
  if (!(a.nsplit == 1 || a.nsplit == 3)) return -1;
  if (((a.H << a.up) + 2 * a.pad - a.ks) / a.stride + 1 != a.OH || ((a.W << a.up) + 2 * a.pad - a.ks) / a.stride + 1 != a.OW) return -1;
  const bool half = s2 && a.nsplit == 3;       // 64-pixel tiles: the stride-2 patch of a 128-pixel tile does not fit three planes
  long tiles;
  int shape;           // pixel tile: 0 = 8 x 16, 1 = two 8 x 8 images, 2 = 4 x 16 (64 pixels), 3 = 8 x 8 of one image (64 pixels),
                       // 4 = four whole 4 x 4 images (64 pixels), 5 = 4 x 32, 6 = sixteen whole 2 x 2 images (64 pixels; stride 2 only)
  // (5: rows of 32 pixels are whole 128-byte lines of dy and x — with 16-pixel rows two tiles share every line, and at three
  //  workgroups per CU the second one comes after the line has left the L2: measured 2.8x the operand bytes on 64 -> 64 at 64 x 64.
  //  Only where the layer is memory-bound — bf16 mode, <= 64 output channels: 245 -> 211 us there, 178 -> 155 us at 128 x 128; the
  //  wider patch (34 x 6 against 18 x 10 pixels to stage and convert) costs 6 % on 128-channel layers and 45 % in split mode.)
  if (AGL_PBWW_ROWS32 && !s2 && a.ks == 3 && a.nsplit == 1 && a.Cout <= 64 && a.OW % 32 == 0 && a.OH % 4 == 0) { tiles = (long)a.N * (a.OH / 4) * (a.OW / 32); shape = 5; }
  else if (a.OW % 16 == 0 && a.OH % 8 == 0) { tiles = (long)a.N * (a.OH / 8) * (a.OW / 16) * (half ? 2 : 1); shape = half ? 2 : 0; }
  else if (a.OW == 8 && a.OH == 8) { tiles = half ? a.N : agl_cdiv(a.N, 2); shape = half ? 3 : 1; }
  else if (a.OW % 8 == 0 && a.OH % 8 == 0) { tiles = (long)a.N * (a.OH / 8) * (a.OW / 8); shape = 3; }
  else if (a.OW == 4 && a.OH == 4 && a.ks != 5) { tiles = agl_cdiv(a.N, 4); shape = 4; }
  else if (a.OW == 2 && a.OH == 2 && s2 && !a.x_blk && !a.dy_bf16) { tiles = agl_cdiv(a.N, 16); shape = 6; }      // sixteen whole 2x2 maps (the crop encoder's last layers)
  else return -1;
  if ((long)a.N * a.Cin * a.H * a.W >= (1L << 29) || (long)a.N * a.Cout * a.OH * a.OW >= (1L << 29)) return -1;
  // accumulators per lane: 4 * RT * CT * ks^2 (x2 in split mode)
  // One 64 x 16 fragment block per workgroup (RT = CT = 1) also in bf16 mode: only that form has the registers to fetch the next tile
  // while the matrix cores work on this one (pbww_k PREF).  The 128 x 16 / 64 x 32 / 128 x 32 blocks stage fewer bytes per product
  // but load item by item, every load's latency exposed — measured (profiles/r04_pbww_blocks.txt, tools/one_conv.py, bf16): 3x3
  // 221 -> 166 us (210 x 64 x 64^2), 198 -> 160 (210 x 128 x 32^2), 149 -> 98 (32 x 512 x 16^2), never slower; 4x4 stride 2
  // 585 -> 532, 691 -> 659, 620 -> 470 us.
  *rt = 1;
  *ct = (a.ks == 1 && a.Cin % 32 == 0) ? 2 : 1;   // (1x1: staging-bound — dy is read once per 32 input channels)
  // (5x5 in split mode: 25 taps x 2 accumulator sets = 200 registers — one workgroup per CU with the accumulators in AGPRs, all
  //  taps in one pass: 2.43 -> 1.87 ms on the ConvLSTM layer against two passes of 13 + 12 taps that stage every tile twice)
  const int npass = 1;
  // Every split writes a slab of the whole weight tensor, and the reduction reads them all: on a small problem (SPADE's shared
  // convolution on a 24 x 24 class grid of 32 images: 14 MB of operands, 295 KB per slab) one split per tile meant 85 MB of slabs
  // — 0.13 ms for 3 GFLOP.  The slabs may not outweigh the operands; when that leaves too few workgroups for the chip, the
  // fragment block goes back to 64 x 16 channels (more, smaller workgroups instead of more slabs).
  const long slab_bytes = (long)a.Cout * a.Cin * a.ks * a.ks * 4;
  const long operand_bytes = ((long)a.N * a.Cout * a.OH * a.OW + (long)a.N * a.Cin * a.H * a.W) * 4;
#ifndef AGL_PBWW_ZCAP_DIV_BF16
#define AGL_PBWW_ZCAP_DIV_BF16 4
#endif
  // (bf16 mode — one product per multiply-add, the layer is memory-bound: the slabs may take a quarter of the operand bytes;
  //  256 -> 256 at 16 x 16 x 210: 236 -> 156 us, 128 px step 261 -> 266 images/s.  In split mode the same cap costs 3 %: the
  //  workgroups it removes were hiding the longer matrix segments.)
  const long zcap = std::max(4L, operand_bytes / (slab_bytes * (a.nsplit == 1 ? AGL_PBWW_ZCAP_DIV_BF16 : 1)));
  long blocks = (long)agl_cdiv(a.Cout, 64 * *rt) * (a.Cin / (16 * *ct)) * npass;
  // Two workgroups per CU — what the kernel's register budget keeps resident (__launch_bounds__(NT, 2)): one full round of
  // workgroups, no partial second round, and every further split would be slab traffic.  (Target 1024 / 768 / 640 / 512 / 384 / 256:
  // 128 px bf16 step 264-265 / 262-267 / - / 267-272 / 257-264 / 258-259 images/s; 64 px split-mode step 481-486 / 481-483 / 477 /
  // 485-492 / 474-478 / -.)
  const long wgs = 512;
  long z = std::min((wgs + blocks - 1) / blocks, zcap);
  if (blocks >= 384) z = 1;      // the channel blocks alone fill the chip: one split writes dw directly (no slab pass over a large tensor)
  if (blocks * z < 256 && (*rt > 1 || *ct > 1)) {
    *rt = 1; *ct = 1;
    blocks = (long)agl_cdiv(a.Cout, 64) * (a.Cin / 16) * npass;
    z = std::min((wgs + blocks - 1) / blocks, zcap);
  }
  if (z > tiles) z = tiles;
  if (z < 1) z = 1;
  const long per = (tiles + z - 1) / z;
  z = (tiles + per - 1) / per;
  *splits = (int)z; *tps = (int)per; *tiles_out = tiles; *half_out = shape;
  return 0;
}

long pbww_ws_bytes(const PBwwArgs& a) {
  int s, t, rt, ct, half; long tiles;
  if (pbww_plan(a, &s, &t, &tiles, &rt, &ct, &half) != 0) return 0;
  return (long)s * a.Cout * a.Cin * a.ks * a.ks * 4 + (long)s * a.Cout * 4;      // weight slabs + bias-gradient slabs
}

// shape 6 (sixteen whole 2x2 output maps per tile): stride-2 forms only — a template so that the stride-1 callers of the launch macros
// do not instantiate it
template <int KS, int S, int RT, int CT, int NS, bool AUX_>
static void pbww_launch_2x2(dim3 g, hipStream_t st, const WArgs& p) {
  if constexpr (S == 2) hipLaunchKernelGGL((pbww_k<KS, 2, 2, 2, 16, RT, CT, NS, KS * KS, AUX_>), g, dim3(NT), 0, st, p);
}

bool pbww_takes_spade(const PBwwArgs& a) {      // (pbww_try's aux5: the one bf16 5x5 tile shape the transform is compiled into)
  int splits, tps, rt, ct, half, oh, ow; long tiles;
  if (pbww_plan(a, &splits, &tps, &tiles, &rt, &ct, &half, &oh, &ow) != 0) return false;
  return a.ks == 5 && a.stride == 1 && a.nsplit == 1 && a.up == 0 && half == 0;
}

int pbww_try(const PBwwArgs& a, void* ws, long ws_bytes, hipStream_t st, const char* name) {
  int splits, tps, rt, ct, half, oh, ow; long tiles;
  if (pbww_plan(a, &splits, &tps, &tiles, &rt, &ct, &half, &oh, &ow) != 0) return -1;
  const long n = (long)a.Cout * a.Cin * a.ks * a.ks;
  const bool direct = splits == 1;
  if (!direct && (!ws || ws_bytes < (long)splits * n * 4)) return -1;
  WArgs p;
  p.direct = direct ? a.dw : nullptr; p.accumulate = a.accumulate;
  // bias gradient alongside (a.dbias): its slabs follow the weight slabs when the workspace has room for them
  const bool with_bias = a.dbias != nullptr && ws && ws_bytes >= (long)splits * n * 4 + (long)splits * a.Cout * 4;
  p.bias_slabs = with_bias ? (float*)ws + (long)splits * n : nullptr;
  p.dy = a.dy; p.x = a.x; p.slabs = (float*)ws; p.N = a.N; p.Cin = a.Cin; p.H = a.H; p.W = a.W; p.Cout = a.Cout; p.OH = a.OH; p.OW = a.OW;
  if (a.ks == 1) { p.H = p.OH = oh; p.W = p.OW = ow; }
  p.pad = a.pad; p.up = a.up; p.in_relu = a.in_relu; p.tiles = (int)tiles; p.tiles_per_split = tps;
  p.x_bf16 = a.x_bf16; p.dy_bf16 = a.dy_bf16; p.fold = a.fold; p.x_blk = a.x_blk;
  if (a.x_blk && !(a.x_bf16 && a.nsplit == 1 && a.up == 0 && !a.dy_bf16 && !a.fold.scale && a.Cin % 16 == 0 &&
                   ((a.ks == 3 && a.stride == 1) || (a.ks == 4 && a.stride == 2)))) return -1;
  if (a.dy_bf16 && (a.nsplit != 1 || (a.OW % 8 != 0 && !(a.OW == 4 && a.OH == 4)))) return -1;      // (16-byte pieces of 8 bf16)
  const bool aux = a.dy_bf16 || a.fold.scale != nullptr;
  const bool aux5 = aux && a.ks == 5 && a.stride == 1 && a.nsplit == 1 && !a.dy_bf16 && half == 0;      // (SPADE in front of the 128 px decoder's c6)
  if (aux && !(a.ks == 4 && a.stride == 2) && !aux5) return -1;      // (compiled into the 4x4 / stride-2 instantiations and one bf16 5x5 tile only)
  if (a.fold.cells && (a.up != 0 || a.x_bf16 || a.fold.per_n || !a.fold.map || a.fold.G <= 0)) return -1;
  p.x_bytes = (unsigned)((long)a.N * a.Cin * a.H * a.W * (a.x_bf16 ? 2 : 4)); p.dy_bytes = (unsigned)((long)a.N * a.Cout * a.OH * a.OW * 4);
  const int npass = 1;
  dim3 g((unsigned)splits, a.Cin / (16 * ct) * npass, agl_cdiv(a.Cout, 64 * rt));
  p.xgyz = 1; p.xchunk = 0; p.xitems = 0; p.xgy = 1;
  if (AGL_PCONV_XCD && (long)g.x * g.y * g.z >= 64 && g.y * g.z > 1) {
    const long items = (long)g.x * g.y * g.z;
    p.xgy = (int)g.y; p.xgyz = (int)(g.y * g.z); p.xitems = (int)items; p.xchunk = (int)((items + 7) / 8);
    g = dim3(8u * (unsigned)p.xchunk, 1, 1);
  }
#define PW_LAUNCH(KS_, S_, RT_, CT_, NS_)                                                                           \
  do {                                                                                                              \
    if (half == 6) pbww_launch_2x2<KS_, S_, RT_, CT_, NS_, false>(g, st, p);                                        \
    else if (half == 4) hipLaunchKernelGGL((pbww_k<KS_, S_, 4, 4, 4, RT_, CT_, NS_>), g, dim3(NT), 0, st, p);      \
    else if (half == 3) hipLaunchKernelGGL((pbww_k<KS_, S_, 8, 8, 1, RT_, CT_, NS_>), g, dim3(NT), 0, st, p);      \
    else if (half == 1) hipLaunchKernelGGL((pbww_k<KS_, S_, 8, 8, 2, RT_, CT_, NS_>), g, dim3(NT), 0, st, p);      \
    else if (half == 2) hipLaunchKernelGGL((pbww_k<KS_, S_, 16, 4, 1, RT_, CT_, NS_>), g, dim3(NT), 0, st, p);     \
    else hipLaunchKernelGGL((pbww_k<KS_, S_, 16, 8, 1, RT_, CT_, NS_>), g, dim3(NT), 0, st, p);                    \
  } while (0)
#define PW_LAUNCH_AUX(KS_, S_, RT_, CT_, NS_)                                                                       \
  do {                                                                                                              \
    if (half == 6) pbww_launch_2x2<KS_, S_, RT_, CT_, NS_, true>(g, st, p);                                                          \
    else if (half == 4) hipLaunchKernelGGL((pbww_k<KS_, S_, 4, 4, 4, RT_, CT_, NS_, KS_ * KS_, true>), g, dim3(NT), 0, st, p);      \
    else if (half == 3) hipLaunchKernelGGL((pbww_k<KS_, S_, 8, 8, 1, RT_, CT_, NS_, KS_ * KS_, true>), g, dim3(NT), 0, st, p);      \
    else if (half == 1) hipLaunchKernelGGL((pbww_k<KS_, S_, 8, 8, 2, RT_, CT_, NS_, KS_ * KS_, true>), g, dim3(NT), 0, st, p);      \
    else if (half == 2) hipLaunchKernelGGL((pbww_k<KS_, S_, 16, 4, 1, RT_, CT_, NS_, KS_ * KS_, true>), g, dim3(NT), 0, st, p);     \
    else hipLaunchKernelGGL((pbww_k<KS_, S_, 16, 8, 1, RT_, CT_, NS_, KS_ * KS_, true>), g, dim3(NT), 0, st, p);                    \
  } while (0)
#define PW_LAUNCH1(KS_, RT_, CT_, NS_)      /* 3x3, bf16 mode, <= 64 output channels: the 4 x 32 tile (shape 5) */              \
  do {                                                                                                              \
    if (half == 5) hipLaunchKernelGGL((pbww_k<KS_, 1, 32, 4, 1, RT_, CT_, NS_>), g, dim3(NT), 0, st, p);            \
    else PW_LAUNCH(KS_, 1, RT_, CT_, NS_);                                                                          \
  } while (0)
#define PW_LAUNCH_XB(KS_, S_)                                                                                       \
  do {                                                                                                              \
    if (half == 5) hipLaunchKernelGGL((pbww_k<KS_, S_, 32, 4, 1, 1, 1, 1, KS_ * KS_, false, true>), g, dim3(NT), 0, st, p);        \
    else if (half == 4) hipLaunchKernelGGL((pbww_k<KS_, S_, 4, 4, 4, 1, 1, 1, KS_ * KS_, false, true>), g, dim3(NT), 0, st, p);    \
    else if (half == 3) hipLaunchKernelGGL((pbww_k<KS_, S_, 8, 8, 1, 1, 1, 1, KS_ * KS_, false, true>), g, dim3(NT), 0, st, p);    \
    else if (half == 1) hipLaunchKernelGGL((pbww_k<KS_, S_, 8, 8, 2, 1, 1, 1, KS_ * KS_, false, true>), g, dim3(NT), 0, st, p);    \
    else if (half == 2) hipLaunchKernelGGL((pbww_k<KS_, S_, 16, 4, 1, 1, 1, 1, KS_ * KS_, false, true>), g, dim3(NT), 0, st, p);   \
    else hipLaunchKernelGGL((pbww_k<KS_, S_, 16, 8, 1, 1, 1, 1, KS_ * KS_, false, true>), g, dim3(NT), 0, st, p);                  \
  } while (0)
  // (bf16 mode launches the 64 x 16 block only — pbww_plan; 1x1 keeps its 64 x 32 block)
  if (a.x_blk) {      // channel-blocked bf16 x (checked above: 3x3 stride 1 or 4x4 stride 2, bf16 arithmetic)
    if (a.stride == 2) PW_LAUNCH_XB(4, 2); else PW_LAUNCH_XB(3, 1);
  } else if (a.stride == 2) {
    if (a.ks == 4 && aux) { if (a.nsplit == 3) PW_LAUNCH_AUX(4, 2, 1, 1, SPL); else PW_LAUNCH_AUX(4, 2, 1, 1, 1); }
    else if (a.ks == 4) { if (a.nsplit == 3) PW_LAUNCH(4, 2, 1, 1, SPL); else PW_LAUNCH(4, 2, 1, 1, 1); }
    else { if (a.nsplit == 3) PW_LAUNCH(3, 2, 1, 1, SPL); else PW_LAUNCH(3, 2, 1, 1, 1); }
  } else if (a.ks == 1) {
    if (a.nsplit == 3) { if (ct == 2) PW_LAUNCH(1, 1, 1, 2, SPL); else PW_LAUNCH(1, 1, 1, 1, SPL); }
    else if (ct == 2) PW_LAUNCH(1, 1, 1, 2, 1); else PW_LAUNCH(1, 1, 1, 1, 1);
  } else if (a.ks == 3 && a.nsplit == 1) {
    PW_LAUNCH1(3, 1, 1, 1);
  } else if (a.ks == 3) PW_LAUNCH(3, 1, 1, 1, SPL);
  else if (aux5) hipLaunchKernelGGL((pbww_k<5, 1, 16, 8, 1, 1, 1, 1, 25, true>), g, dim3(NT), 0, st, p);
  else if (a.nsplit == 1) PW_LAUNCH(5, 1, 1, 1, 1);
  else {
    if (half == 1) hipLaunchKernelGGL((pbww_k<5, 1, 8, 8, 2, 1, 1, SPL, 25>), g, dim3(NT), 0, st, p);
    else if (half == 3) hipLaunchKernelGGL((pbww_k<5, 1, 8, 8, 1, 1, 1, SPL, 25>), g, dim3(NT), 0, st, p);
    else hipLaunchKernelGGL((pbww_k<5, 1, 16, 8, 1, 1, 1, SPL, 25>), g, dim3(NT), 0, st, p);
  }
#undef PW_LAUNCH_XB
#undef PW_LAUNCH1
#undef PW_LAUNCH_AUX
#undef PW_LAUNCH
  AGL_CHECK_LAUNCH(name);
  if (with_bias && a.dbias_done) *a.dbias_done = 1;
  // the bias-gradient sums ride on the reduction launch of the weight slabs (one launch for both; when the single split wrote dw
  // directly there are no weight slabs and the bias sums are one slab: their own small launch).
  // (agl_launch_slab_reduce: 16 thread rows walk the splits of 16 consecutive elements — one thread per channel was a chain of up
  //  to 384 dependent loads, 27 us per launch)
  if (direct)
    return with_bias ? agl_launch_slab_reduce((const float*)p.bias_slabs, a.dbias, a.Cout, splits, a.dbias_accumulate, st, name) : AGL_OK;
  return agl_launch_slab_reduce((const float*)ws, a.dw, n, splits, a.accumulate, st, name, with_bias ? (const float*)p.bias_slabs : nullptr,
                                a.dbias, a.Cout, a.dbias_accumulate);
}
