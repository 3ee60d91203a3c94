// Convolution family for gfx950 (MI355X): fp32 implicit GEMM on v_mfma_f32_32x32x2_f32.
//
// One generic LDS-tiled kernel (igemm_f32) serves the three convolution passes; a small
// "problem" struct per pass tells it how to gather A and B elements and where an output
// element goes:
//   forward      C[co][pix]  = sum_k  W[co][k]            * im2col(x)[k][pix]   (k = ci,kh,kw)
//   bwd-data     C[ci][pix'] = sum_k  W[co][ci][kh][kw]   * dy-gather[k][pix']  (k = co,th,tw; one
//                launch z-slice per stride phase, so stride-2 k4 convs do 2x2 taps, not 4x4)
//   bwd-weight   C[co][kcol] = sum_r  dy[co][r]           * im2col(x)[r][kcol]  (r = n,oh,ow; split
//                over z with per-split slabs reduced deterministically afterwards)
// ConvTranspose2d(k4,s2,p1) forward is the bwd-data pass of the matching strided conv.
//
// Tile: BMxBN outputs per 256-thread workgroup (4 waves), BK=16 reduction slice, two LDS
// buffers with register prefetch of the next slice, 32x32x2 f32 MFMA accumulators in VGPRs.
// LDS images are [k][row] (row-fast gathers) or [row][k] padded to 17 (k-fast gathers) so both
// the staging writes and the 32-lane fragment reads stay (nearly) bank-conflict free.
#include "agl_internal.h"
#include "pconv.h"
#include "few.h"
#include <math.h>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

#ifndef AGL_SPLIT_FIXED
#define AGL_SPLIT_FIXED 4e-6   // fixed cost (s) the split cost models charge for the extra reduction launch
#endif
constexpr int BK = 16;
#ifndef AGL_GATHER_STEPS
#define AGL_GATHER_STEPS (BK / 2)   // MFMA k-steps of a slice over which the next slice's gathers are spread
#endif
constexpr int NT = 256;

template <int KS> struct KDiv {  // k -> (c, kh, kw) for a KSxKS window
  __device__ static __forceinline__ void split(int k, int& c, int& kh, int& kw) {
    c = k / (KS * KS);
    int r = k - c * (KS * KS);
    kh = r / KS;
    kw = r - kh * KS;
  }
};

constexpr unsigned OOB = 0xFFFFFFF0u;   // byte offset past every buffer: the hardware range check returns 0.0f

// Operand elements are fetched with raw buffer loads (SRD built from the tensor base and byte size): an
// out-of-window tap / out-of-range row simply gets the offset OOB and the hardware returns zero — no
// per-element select, no exec-mask branch, and nothing consumes the loaded value before it is staged to LDS.

// ------------------------------------------------------------------ forward
template <int KS>
struct FwdProb {
  static constexpr bool BIG_TILES = false;   // 256x128 / 128x256 tiles are instantiated for the weight gradient only
  static constexpr bool A_KFAST = true;   // weights: k contiguous
  static constexpr bool B_KFAST = false;  // im2col: pixels contiguous
  const float* x; const float* w; const float* bias; float* y;
  int N, Cin, H, W, Cout, OH, OW, stride, pad, up;  // up: log2 nearest-upsample of x folded into the gather
  int relu, accumulate, in_relu;
  int M, Nc, kbeg, kend;
  int HW, OHW, K;
  unsigned a_bytes, b_bytes;
  int splits, per_split; long slab; float* part;   // split-K: raw partial sums go to part + z*slab

  __device__ bool setup(int z) {
    if (splits > 1) { kbeg = z * per_split; kend = min(K, kbeg + per_split); part += (long)z * slab; }
    return kbeg < kend;
  }
  __device__ void tile(int, int) {}
  __device__ const float* a_ptr() const { return w; }
  __device__ const float* b_ptr() const { return x; }
  __device__ float fix_a(float v) const { return v; }
  __device__ float fix_b(float v) const { return in_relu ? fmaxf(v, 0.f) : v; }

  struct RowA { int base; bool ok; };
  struct KA { int k; bool ok; };
  __device__ RowA row_a(int m) const { return {m * K, m < M}; }
  __device__ KA k_a(int k) const { return {k, k < kend}; }
  __device__ unsigned off_a(const RowA& r, const KA& k) const { return (r.ok & k.ok) ? (unsigned)(r.base + k.k) * 4u : OOB; }

  struct RowB { int base, ih0, iw0; bool ok; };
  struct KB { int coff, kh, kw; bool ok; };
  __device__ RowB row_b(int n) const {
    RowB r; r.ok = n < Nc;
    int nn = r.ok ? n : 0;
    int img = nn / OHW, pix = nn - img * OHW;
    int oh = pix / OW, ow = pix - oh * OW;
    r.base = img * Cin * HW; r.ih0 = oh * stride - pad; r.iw0 = ow * stride - pad;
    return r;
  }
  __device__ KB k_b(int k) const {
    KB s; s.ok = k < kend; int c;
    KDiv<KS>::split(s.ok ? k : 0, c, s.kh, s.kw); s.coff = c * HW;
    return s;
  }
  __device__ unsigned off_b(const RowB& r, const KB& k) const {
    const int ih = r.ih0 + k.kh, iw = r.iw0 + k.kw;
    const bool ok = r.ok & k.ok & ((unsigned)ih < (unsigned)(H << up)) & ((unsigned)iw < (unsigned)(W << up));
    return ok ? (unsigned)(r.base + k.coff + (ih >> up) * W + (iw >> up)) * 4u : OOB;
  }
  struct Col { long off; bool ok; };
  __device__ Col col(int n) const {
    Col c; c.ok = n < Nc; int nn = c.ok ? n : 0;
    int img = nn / OHW, pix = nn - img * OHW;
    c.off = (long)img * Cout * OHW + pix; return c;
  }
  // epilogue API (see igemm_f32): the operands an epilogue needs from memory are fetched for a whole accumulator tile before any
  // of them is consumed — a per-element `if (accumulate) v += y[o]` makes the compiler wait for each load in turn
  __device__ bool out_ok(int m, const Col& c) const { return m < M && c.ok; }
  __device__ long out_off(int m, const Col& c) const { return c.off + (long)m * OHW; }
  __device__ bool reads_old() const { return splits <= 1 && accumulate; }
  __device__ bool reads_mask() const { return false; }
  __device__ float old_at(long o) const { return y[o]; }
  __device__ float mask_at(long) const { return 1.f; }
  __device__ void finish(int m, long o, float v, float old, float) const {
    if (splits > 1) { part[o] = v; return; }
    if (bias) v += bias[m];
    if (accumulate) v += old;
    if (relu) v = fmaxf(v, 0.f);
    y[o] = v;
  }
};

// ------------------------------------------------------------------ backward data (and ConvTranspose forward)
template <int KS, int S>
struct BwdDataProb {
  static constexpr bool BIG_TILES = false;
  static constexpr bool A_KFAST = false;
  static constexpr bool B_KFAST = false;
  static constexpr int TS = KS / S;  // taps per axis per phase
  const float* dy; const float* w; float* dx; const float* bias; const float* pos_mask;
  int N, Cin, IH, IW, Cout, OH, OW, pad;
  int accumulate, relu;
  int M, Nc, kbeg, kend;
  int IHW, OHW, ph, pw, kh0, kw0, ohb, owb, IHp, IWp;
  unsigned a_bytes, b_bytes;
  int splits, per_split; long slab; float* part;

  __device__ bool setup(int zz) {
    const int z = zz / splits, sp = zz - z * splits;
    ph = z / S; pw = z - ph * S;
    IHp = (IH - ph + S - 1) / S; IWp = (IW - pw + S - 1) / S;
    kh0 = (ph + pad) % S; kw0 = (pw + pad) % S;
    ohb = (ph + pad - kh0) / S; owb = (pw + pad - kw0) / S;
    Nc = N * IHp * IWp; M = Cin; kbeg = 0; kend = Cout * TS * TS;
    if (splits > 1) { kbeg = sp * per_split; kend = min(kend, kbeg + per_split); part += (long)sp * slab; }
    return Nc > 0 && kbeg < kend;
  }
  __device__ void tile(int, int) {}
  __device__ const float* a_ptr() const { return w; }
  __device__ const float* b_ptr() const { return dy; }
  __device__ float fix_a(float v) const { return v; }
  __device__ float fix_b(float v) const { return v; }

  struct RowA { int base; bool ok; };
  struct KA { int off; bool ok; };
  __device__ RowA row_a(int m) const { return {m * KS * KS, m < M}; }
  __device__ KA k_a(int k) const {
    KA s; s.ok = k < kend; int co, th, tw;
    KDiv<TS>::split(s.ok ? k : 0, co, th, tw);
    s.off = co * Cin * KS * KS + (kh0 + S * th) * KS + (kw0 + S * tw);
    return s;
  }
  __device__ unsigned off_a(const RowA& r, const KA& k) const { return (r.ok & k.ok) ? (unsigned)(r.base + k.off) * 4u : OOB; }

  struct RowB { int base, oh0, ow0; bool ok; };
  struct KB { int coff, th, tw; bool ok; };
  __device__ RowB row_b(int n) const {
    RowB r; r.ok = n < Nc; int nn = r.ok ? n : 0;
    int per = IHp * IWp;
    int img = nn / per, q = nn - img * per;
    int a = q / IWp, b = q - a * IWp;
    r.base = img * Cout * OHW; r.oh0 = a + ohb; r.ow0 = b + owb;
    return r;
  }
  __device__ KB k_b(int k) const {
    KB s; s.ok = k < kend; int co;
    KDiv<TS>::split(s.ok ? k : 0, co, s.th, s.tw); s.coff = co * OHW;
    return s;
  }
  __device__ unsigned off_b(const RowB& r, const KB& k) const {
    const int oh = r.oh0 - k.th, ow = r.ow0 - k.tw;
    const bool ok = r.ok & k.ok & ((unsigned)oh < (unsigned)OH) & ((unsigned)ow < (unsigned)OW);
    return ok ? (unsigned)(r.base + k.coff + oh * OW + ow) * 4u : OOB;
  }
  struct Col { long off; bool ok; };
  __device__ Col col(int n) const {
    Col c; c.ok = n < Nc; int nn = c.ok ? n : 0;
    int per = IHp * IWp;
    int img = nn / per, q = nn - img * per;
    int a = q / IWp, b = q - a * IWp;
    c.off = (long)img * Cin * IHW + (long)(a * S + ph) * IW + (b * S + pw); return c;
  }
  __device__ bool out_ok(int m, const Col& c) const { return m < M && c.ok; }
  __device__ long out_off(int m, const Col& c) const { return c.off + (long)m * IHW; }
  __device__ bool reads_old() const { return splits <= 1 && accumulate; }
  __device__ bool reads_mask() const { return splits <= 1 && pos_mask != nullptr; }
  __device__ float old_at(long o) const { return dx[o]; }
  __device__ float mask_at(long o) const { return pos_mask[o]; }
  __device__ void finish(int m, long o, float v, float old, float mk) const {
    if (splits > 1) { part[o] = v; return; }
    if (bias) v += bias[m];
    if (pos_mask && !(mk > 0.f)) v = 0.f;
    if (accumulate) v += old;
    if (relu) v = fmaxf(v, 0.f);
    dx[o] = v;
  }
};

// Input gradient of a strided convolution whose kernel size is not a multiple of the stride (3x3 stride 2: the
// down-sampling discriminator blocks in box-filter form): one launch z-slice per stride phase, the phase's taps are
// kh = kh0 + S*th with a phase-dependent count (2x2, 2x1, 1x2, 1x1 taps for 3x3/s2 — 9 in total, no zero taps).
template <int KS, int S>
struct BwdDataGenProb {
  static constexpr bool BIG_TILES = false;
  static constexpr bool A_KFAST = false;
  static constexpr bool B_KFAST = false;
  const float* dy; const float* w; float* dx; const float* bias; const float* pos_mask;
  int N, Cin, IH, IW, Cout, OH, OW, pad;
  int accumulate, relu;
  int M, Nc, kbeg, kend;
  int IHW, OHW, ph, pw, kh0, kw0, ohb, owb, IHp, IWp, tsw, tt;
  unsigned inv_tt, inv_tsw;
  unsigned a_bytes, b_bytes;
  int splits; long slab; float* part;
  int tap_split;     // 1: a phase with tt taps uses (splits / TTMAX) * tt of the `splits` slabs, so every workgroup reduces
                     // over the same length (the phases are 4:2:2:1 in size); its remaining slabs are stored as zeros
  static constexpr int TTMAX = ((KS + S - 1) / S) * ((KS + S - 1) / S);

  __device__ bool setup(int zz) {
    const int z = zz / splits, sp = zz - z * splits;
    ph = z / S; pw = z - ph * S;
    IHp = (IH - ph + S - 1) / S; IWp = (IW - pw + S - 1) / S;
    kh0 = (ph + pad) % S; kw0 = (pw + pad) % S;
    ohb = (ph + pad - kh0) / S; owb = (pw + pad - kw0) / S;
    const int tsh = kh0 < KS ? (KS - kh0 + S - 1) / S : 0;
    tsw = kw0 < KS ? (KS - kw0 + S - 1) / S : 0;
    tt = tsh * tsw;
    Nc = N * IHp * IWp; M = Cin;
    if (Nc <= 0) return false;
    if (tsw == 0) tsw = 1;
    inv_tt = tt > 1 ? (unsigned)((0x100000000ULL + tt - 1) / tt) : 0u;
    inv_tsw = 65536u / tsw + 1u;
    const int K = Cout * tt;
    const int s_p = tap_split ? (splits / TTMAX) * tt : splits;
    const int per = s_p > 0 ? ((K + s_p - 1) / s_p + BK - 1) / BK * BK : 0;
    kbeg = sp * per; kend = sp < s_p ? min(K, kbeg + per) : kbeg;
    if (kend < kbeg) kend = kbeg;                    // an empty split still stores its (zero) slab
    if (splits > 1) part += (long)sp * slab;
    return true;
  }
  __device__ void tile(int, int) {}
  __device__ const float* a_ptr() const { return w; }
  __device__ const float* b_ptr() const { return dy; }
  __device__ float fix_a(float v) const { return v; }
  __device__ float fix_b(float v) const { return v; }
  __device__ void decode(int k, int& co, int& th, int& tw) const {
    co = tt > 1 ? (int)__umulhi((unsigned)k, inv_tt) : k;
    const int r = k - co * tt;
    th = (int)(((unsigned)r * inv_tsw) >> 16);
    tw = r - th * tsw;
  }
  struct RowA { int base; bool ok; };
  struct KA { int off; bool ok; };
  __device__ RowA row_a(int m) const { return {m * KS * KS, m < M}; }
  __device__ KA k_a(int k) const {
    KA s; s.ok = k < kend; int co, th, tw;
    decode(s.ok ? k : 0, co, th, tw);
    s.off = co * Cin * KS * KS + (kh0 + S * th) * KS + (kw0 + S * tw);
    return s;
  }
  __device__ unsigned off_a(const RowA& r, const KA& k) const { return (r.ok & k.ok) ? (unsigned)(r.base + k.off) * 4u : OOB; }
  struct RowB { int base, oh0, ow0; bool ok; };
  struct KB { int coff, th, tw; bool ok; };
  __device__ RowB row_b(int n) const {
    RowB r; r.ok = n < Nc; int nn = r.ok ? n : 0;
    int per = IHp * IWp;
    int img = nn / per, q = nn - img * per;
    int a = q / IWp, b = q - a * IWp;
    r.base = img * Cout * OHW; r.oh0 = a + ohb; r.ow0 = b + owb;
    return r;
  }
  __device__ KB k_b(int k) const {
    KB s; s.ok = k < kend; int co;
    decode(s.ok ? k : 0, co, s.th, s.tw); s.coff = co * OHW;
    return s;
  }
  __device__ unsigned off_b(const RowB& r, const KB& k) const {
    const int oh = r.oh0 - k.th, ow = r.ow0 - k.tw;
    const bool ok = r.ok & k.ok & ((unsigned)oh < (unsigned)OH) & ((unsigned)ow < (unsigned)OW);
    return ok ? (unsigned)(r.base + k.coff + oh * OW + ow) * 4u : OOB;
  }
  struct Col { long off; bool ok; };
  __device__ Col col(int n) const {
    Col c; c.ok = n < Nc; int nn = c.ok ? n : 0;
    int per = IHp * IWp;
    int img = nn / per, q = nn - img * per;
    int a = q / IWp, b = q - a * IWp;
    c.off = (long)img * Cin * IHW + (long)(a * S + ph) * IW + (b * S + pw); return c;
  }
  __device__ bool out_ok(int m, const Col& c) const { return m < M && c.ok; }
  __device__ long out_off(int m, const Col& c) const { return c.off + (long)m * IHW; }
  __device__ bool reads_old() const { return splits <= 1 && accumulate; }
  __device__ bool reads_mask() const { return splits <= 1 && pos_mask != nullptr; }
  __device__ float old_at(long o) const { return dx[o]; }
  __device__ float mask_at(long o) const { return pos_mask[o]; }
  __device__ void finish(int m, long o, float v, float old, float mk) const {
    if (splits > 1) { part[o] = v; return; }
    if (bias) v += bias[m];
    if (pos_mask && !(mk > 0.f)) v = 0.f;
    if (accumulate) v += old;
    if (relu) v = fmaxf(v, 0.f);
    dx[o] = v;
  }
};

// ------------------------------------------------------------------ backward weight
template <int KS>
struct BwdWeightProb {
  static constexpr bool BIG_TILES = true;
  static constexpr bool A_KFAST = true;  // reduction index r = (n,oh,ow) is the contiguous one
  static constexpr bool B_KFAST = true;
  const float* dy; const float* x; float* out;  // out: dw (splits==1) or slab base
  int N, Cin, H, W, Cout, OH, OW, stride, pad, up, in_relu;
  int M, Nc, kbeg, kend;
  int HW, OHW, R, per_split;
  long slab;
  unsigned a_bytes, b_bytes;

  __device__ bool setup(int z) {
    kbeg = z * per_split; kend = min(R, kbeg + per_split);
    out += (long)z * slab;
    return kbeg < kend;
  }
  __device__ void tile(int, int) {}
  __device__ const float* a_ptr() const { return dy; }
  __device__ const float* b_ptr() const { return x; }
  __device__ float fix_a(float v) const { return v; }
  __device__ float fix_b(float v) const { return in_relu ? fmaxf(v, 0.f) : v; }

  struct RowA { int off; bool ok; };
  struct KA { int base; bool ok; };
  __device__ RowA row_a(int m) const { return {m * OHW, m < M}; }
  __device__ KA k_a(int r) const {
    KA s; s.ok = r < kend; int rr = s.ok ? r : 0;
    int img = rr / OHW, pix = rr - img * OHW;
    s.base = img * Cout * OHW + pix; return s;
  }
  __device__ unsigned off_a(const RowA& r, const KA& k) const { return (r.ok & k.ok) ? (unsigned)(k.base + r.off) * 4u : OOB; }

  struct RowB { int coff, kh, kw; bool ok; };
  struct KB { int base, ih0, iw0; bool ok; };
  __device__ RowB row_b(int n) const {
    RowB r; r.ok = n < Nc; int c;
    KDiv<KS>::split(r.ok ? n : 0, c, r.kh, r.kw); r.coff = c * HW; return r;
  }
  __device__ KB k_b(int r) const {
    KB s; s.ok = r < kend; int rr = s.ok ? r : 0;
    int img = rr / OHW, pix = rr - img * OHW;
    int oh = pix / OW, ow = pix - oh * OW;
    s.base = img * Cin * HW; s.ih0 = oh * stride - pad; s.iw0 = ow * stride - pad; return s;
  }
  __device__ unsigned off_b(const RowB& r, const KB& k) const {
    const int ih = k.ih0 + r.kh, iw = k.iw0 + r.kw;
    const bool ok = r.ok & k.ok & ((unsigned)ih < (unsigned)(H << up)) & ((unsigned)iw < (unsigned)(W << up));
    return ok ? (unsigned)(k.base + r.coff + (ih >> up) * W + (iw >> up)) * 4u : OOB;
  }
  struct Col { int n; bool ok; };
  __device__ Col col(int n) const { return {n, n < Nc}; }
  __device__ bool out_ok(int m, const Col& c) const { return m < M && c.ok; }
  __device__ long out_off(int m, const Col& c) const { return (long)m * Nc + c.n; }
  __device__ bool reads_old() const { return false; }
  __device__ bool reads_mask() const { return false; }
  __device__ float old_at(long) const { return 0.f; }
  __device__ float mask_at(long) const { return 1.f; }
  __device__ void finish(int, long o, float v, float, float) const { out[o] = v; }
};

// ------------------------------------------------------------------ position-major forward (small maps)
// On 4x4 / 8x8 maps a large share of the im2col taps multiply padding (5x5 pad 2 on 8x8: 28 %; 4x4 stride 2 on
// 4x4 -> 2x2: 44 %).  With the operands transposed to position-major form — Xt[pos][ci][img], Wt[tap][co][ci],
// Yt[q][co][img] — the convolution is, per output position q, a sum of PLAIN matrix products over the taps that fall
// inside the map:  Yt[q] = sum_{tap valid for q} Wt[tap] * Xt[pos(q,tap)],  so a GEMM column tile (128 images at one
// position) simply never visits the padded taps.  The valid taps of a position form a rectangle
// [kh_lo..kh_hi] x [kw_lo..kw_hi]; the reduction index is k = (tap_index, ci).  The transposes cost two passes over
// tensors that are tiny at these map sizes.
template <int KS>
struct PosFwdProb {
  static constexpr bool BIG_TILES = false;
  static constexpr bool A_KFAST = true;   // Wt: ci contiguous
  static constexpr bool B_KFAST = false;  // Xt: images contiguous
  const float* xt; const float* wt; const float* bias; float* yt;
  int N, Cin, cin_shift, H, W, Cout, OH, OW, stride, pad, relu;
  int M, Nc, kbeg, kend;            // Nc = Q * N: GEMM columns are (position, image), position-major
  int nfull, Q;                     // nfull = N rounded down to whole 128-column tiles (see col_of)
  int sp, kh_lo, kw_lo, nkw, inv_nkw;
  unsigned a_bytes, b_bytes;
  int splits; long slab; float* part;

  __device__ bool setup(int bz) {
    sp = bz;
    if (splits > 1) part += (long)sp * slab;
    return true;
  }
  // Column order: first, for every position, its images in whole 128-column tiles (nfull per position: tiles that
  // see ONE position and skip all of its padded taps); then the leftover N - nfull images of all positions packed
  // together.  A tile's taps are the bounding rectangle of the valid taps of the positions it covers; a lane whose own
  // position lacks a tap of the rectangle reads zero (out-of-range offset).
  __device__ void col_of(int n, int& q, int& img) const {
    const int pure = Q * nfull;
    if (n < pure) { q = n / nfull; img = n - q * nfull; }
    else { const int r = n - pure, nrem = N - nfull; q = r / nrem; img = nfull + r - q * nrem; }
  }
  __device__ void tile(int bn0, int bn) {
    int q_lo, q_hi, dummy;
    col_of(bn0, q_lo, dummy);
    col_of(min(bn0 + bn, Nc) - 1, q_hi, dummy);
    int kh_hi = -1, kw_hi = -1;
    kh_lo = KS; kw_lo = KS;
    for (int q = q_lo; q <= q_hi; ++q) {
      const int oh = q / OW, ow = q - oh * OW, ih0 = oh * stride - pad, iw0 = ow * stride - pad;
      kh_lo = min(kh_lo, max(0, -ih0)); kh_hi = max(kh_hi, min(KS - 1, H - 1 - ih0));
      kw_lo = min(kw_lo, max(0, -iw0)); kw_hi = max(kw_hi, min(KS - 1, W - 1 - iw0));
    }
    nkw = kw_hi - kw_lo + 1;
    inv_nkw = 65536 / nkw + 1;                       // j / nkw == (j * inv_nkw) >> 16 for j < 64
    const int Kq = (kh_hi - kh_lo + 1) * nkw * Cin;
    const int per = ((Kq + splits - 1) / splits + BK - 1) / BK * BK;
    kbeg = sp * per; kend = min(Kq, kbeg + per);
    if (kend < kbeg) kend = kbeg;                    // an empty split still stores its (zero) slab
  }
  __device__ const float* a_ptr() const { return wt; }
  __device__ const float* b_ptr() const { return xt; }
  __device__ float fix_a(float v) const { return v; }
  __device__ float fix_b(float v) const { return v; }
  __device__ void decode(int k, int& ci, int& kh, int& kw) const {
    const int j = k >> cin_shift;
    ci = k - (j << cin_shift);
    const int th = (j * inv_nkw) >> 16;
    kh = kh_lo + th; kw = kw_lo + j - th * nkw;
  }
  struct RowA { int base; bool ok; };
  struct KA { int off; bool ok; };
  __device__ RowA row_a(int m) const { return {m * Cin, m < M}; }
  __device__ KA k_a(int k) const {
    KA s; s.ok = k < kend; int ci, kh, kw;
    decode(s.ok ? k : 0, ci, kh, kw);
    s.off = (kh * KS + kw) * Cout * Cin + ci; return s;
  }
  __device__ unsigned off_a(const RowA& r, const KA& k) const { return (r.ok & k.ok) ? (unsigned)(r.base + k.off) * 4u : OOB; }
  struct RowB { int img, ih0, iw0; bool ok; };
  struct KB { int coff, kh, kw; bool ok; };
  __device__ RowB row_b(int n) const {
    RowB r; r.ok = n < Nc;
    int q;
    col_of(r.ok ? n : 0, q, r.img);
    const int oh = q / OW, ow = q - oh * OW;
    r.ih0 = oh * stride - pad; r.iw0 = ow * stride - pad;
    return r;
  }
  __device__ KB k_b(int k) const {
    KB s; s.ok = k < kend; int ci;
    decode(s.ok ? k : 0, ci, s.kh, s.kw);
    s.coff = ci * N; return s;
  }
  __device__ unsigned off_b(const RowB& r, const KB& k) const {
    const int ih = r.ih0 + k.kh, iw = r.iw0 + k.kw;
    const bool ok = r.ok & k.ok & ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
    return ok ? (unsigned)((ih * W + iw) * Cin * N + k.coff + r.img) * 4u : OOB;
  }
  struct Col { long off; bool ok; };
  __device__ Col col(int n) const {
    Col c; c.ok = n < Nc;
    int q, img;
    col_of(c.ok ? n : 0, q, img);
    c.off = (long)q * Cout * N + img; return c;
  }
  __device__ bool out_ok(int m, const Col& c) const { return m < M && c.ok; }
  __device__ long out_off(int m, const Col& c) const { return c.off + (long)m * N; }
  __device__ bool reads_old() const { return false; }
  __device__ bool reads_mask() const { return false; }
  __device__ float old_at(long) const { return 0.f; }
  __device__ float mask_at(long) const { return 1.f; }
  __device__ void finish(int m, long o, float v, float, float) const {
    if (splits > 1) { part[o] = v; return; }
    if (bias) v += bias[m];
    if (relu) v = fmaxf(v, 0.f);
    yt[o] = v;
  }
};

// Weight gradient in position-major form: per tap, dWt[tap] = sum over the output positions q whose tap falls inside
// the map of dYt[q] * Xt[pos(q,tap)]^T — reduction index k = (valid position, image); padded taps contribute nothing
// and are never visited.  grid z = tap * splits + split.
template <int KS>
struct PosBwwProb {
  static constexpr bool BIG_TILES = true;
  static constexpr bool A_KFAST = true;   // both operands: images (the reduction index) contiguous
  static constexpr bool B_KFAST = true;
  const float* dyt; const float* xt; float* dwt; float* part;
  int N, Cin, H, W, Cout, OH, OW, stride, pad;
  int M, Nc, kbeg, kend;
  int kh, kw, oh_lo, ow_lo, now;
  unsigned a_bytes, b_bytes;
  int splits; long slab;
  float* out;

  __device__ bool setup(int bz) {
    const int tap = bz / splits, sp = bz - tap * splits;
    kh = tap / KS; kw = tap - kh * KS;
    // ih = oh*stride - pad + kh in [0, H)
    oh_lo = max(0, (pad - kh + stride - 1) / stride); ow_lo = max(0, (pad - kw + stride - 1) / stride);
    const int oh_hi = min(OH - 1, (H - 1 + pad - kh) / stride), ow_hi = min(OW - 1, (W - 1 + pad - kw) / stride);
    const int noh = (H - 1 + pad - kh < 0) ? 0 : max(0, oh_hi - oh_lo + 1);
    now = (W - 1 + pad - kw < 0) ? 0 : max(0, ow_hi - ow_lo + 1);
    const int K = noh * now * N;
    const int per = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    kbeg = sp * per; kend = min(K, kbeg + per);
    if (kend < kbeg) kend = kbeg;
    if (now == 0) now = 1;
    out = (splits > 1 ? part + (long)sp * slab : dwt) + (long)tap * Cout * Cin;
    return true;
  }
  __device__ void tile(int, int) {}
  __device__ const float* a_ptr() const { return dyt; }
  __device__ const float* b_ptr() const { return xt; }
  __device__ float fix_a(float v) const { return v; }
  __device__ float fix_b(float v) const { return v; }
  __device__ void decode(int k, int& img, int& oh, int& ow) const {
    const int qi = k / N;
    img = k - qi * N;
    const int t = qi / now;
    oh = oh_lo + t; ow = ow_lo + qi - t * now;
  }
  struct RowA { int base; bool ok; };
  struct KA { int base; bool ok; };
  __device__ RowA row_a(int m) const { return {m * N, m < M}; }
  __device__ KA k_a(int k) const {
    KA s; s.ok = k < kend; int img, oh, ow;
    decode(s.ok ? k : 0, img, oh, ow);
    s.base = (oh * OW + ow) * Cout * N + img; return s;
  }
  __device__ unsigned off_a(const RowA& r, const KA& k) const { return (r.ok & k.ok) ? (unsigned)(k.base + r.base) * 4u : OOB; }
  struct RowB { int base; bool ok; };
  struct KB { int base; bool ok; };
  __device__ RowB row_b(int n) const { return {n * N, n < Nc}; }
  __device__ KB k_b(int k) const {
    KB s; s.ok = k < kend; int img, oh, ow;
    decode(s.ok ? k : 0, img, oh, ow);
    s.base = ((oh * stride - pad + kh) * W + ow * stride - pad + kw) * Cin * N + img; return s;
  }
  __device__ unsigned off_b(const RowB& r, const KB& k) const { return (r.ok & k.ok) ? (unsigned)(k.base + r.base) * 4u : OOB; }
  struct Col { int n; bool ok; };
  __device__ Col col(int n) const { return {n, n < Nc}; }
  __device__ bool out_ok(int m, const Col& c) const { return m < M && c.ok; }
  __device__ long out_off(int m, const Col& c) const { return (long)m * Nc + c.n; }
  __device__ bool reads_old() const { return false; }
  __device__ bool reads_mask() const { return false; }
  __device__ float old_at(long) const { return 0.f; }
  __device__ float mask_at(long) const { return 1.f; }
  __device__ void finish(int, long o, float v, float, float) const { out[o] = v; }
};

// dw[(a*B + b)*KK + t] (+)= dwt[(t*A + a)*B + b]
__global__ void tap_major_to_w_k(const float* __restrict__ dwt, float* __restrict__ dw, int A, int B, int KK, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)A * B * KK) return;
  const int t = (int)(i % KK), b = (int)(i / KK % B), a = (int)(i / ((long)KK * B));
  const float v = dwt[((long)t * A + a) * B + b];
  dw[i] = accumulate ? dw[i] + v : v;
}

// xt[(p*C + c)*N + n] = x[(n*C + c)*HW + p]  (optional ReLU); block = one channel x 64 images, HW <= 64
__global__ __launch_bounds__(256) void nchw_to_pcn_k(const float* __restrict__ x, float* __restrict__ xt, int N, int C, int HW, int in_relu) {
  __shared__ float t[64][65];
  const int c = blockIdx.x, n0 = blockIdx.y * 64;
  for (int e = threadIdx.x; e < 64 * HW; e += 256) {
    const int nl = e / HW, p = e - nl * HW, n = n0 + nl;
    float v = n < N ? x[((long)n * C + c) * HW + p] : 0.f;
    t[nl][p] = in_relu ? fmaxf(v, 0.f) : v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * HW; e += 256) {
    const int p = e >> 6, nl = e & 63, n = n0 + nl;
    if (n < N) xt[((long)p * C + c) * N + n] = t[nl][p];
  }
}
// y[(n*C + c)*HW + p] (+)= yt[(p*C + c)*N + n], optionally masked by pos_mask (same layout as y)
__global__ __launch_bounds__(256) void pcn_to_nchw_k(const float* __restrict__ yt, float* __restrict__ y, const float* __restrict__ pos_mask,
                                                     int N, int C, int HW, int accumulate) {
  __shared__ float t[64][65];
  const int c = blockIdx.x, n0 = blockIdx.y * 64;
  for (int e = threadIdx.x; e < 64 * HW; e += 256) {
    const int p = e >> 6, nl = e & 63, n = n0 + nl;
    t[nl][p] = n < N ? yt[((long)p * C + c) * N + n] : 0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * HW; e += 256) {
    const int nl = e / HW, p = e - nl * HW, n = n0 + nl;
    if (n < N) {
      const long o = ((long)n * C + c) * HW + p;
      float v = t[nl][p];
      if (pos_mask && !(pos_mask[o] > 0.f)) v = 0.f;
      y[o] = accumulate ? y[o] + v : v;
    }
  }
}
// wt[(t*A + a)*B + b] = w[...]: mode 0: w[a][b][t] (forward: a = co, b = ci);  mode 1: w[b][a][t];  mode 2: w[b][a][KK-1-t]
// (input gradient of a stride-1 convolution = forward convolution of dy with flipped taps and swapped channel roles)
__global__ void w_to_tap_major_k(const float* __restrict__ w, float* __restrict__ wt, int A, int B, int KK, int mode) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)A * B * KK) return;
  const int b = (int)(i % B), a = (int)(i / B % A), t = (int)(i / ((long)A * B));
  wt[i] = mode == 0 ? w[((long)a * B + b) * KK + t] : (mode == 1 ? w[((long)b * A + a) * KK + t] : w[((long)b * A + a) * KK + (KK - 1 - t)]);
}

// ------------------------------------------------------------------ workgroup -> tile, XCD-aware
// The grid is 1-D; workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an XCD and its L2).  Tiles
// that read the same operand slices are therefore placed 8 ids apart:
//   * z >= 8 slices (split-K parts / weight-gradient slabs: each slice reads its own K-range of both operands):
//     slice z lives on XCD z % 8, so one L2 fetches that K-range once instead of all eight;
//   * otherwise: the gy row tiles of a column tile (same im2col columns, different filters) sit on one XCD, adjacent
//     in dispatch order, so the gathered operand is fetched from HBM once per column tile, not once per row tile.
#ifndef AGL_XCD_REMAP
#define AGL_XCD_REMAP 1
#endif
__device__ __forceinline__ void tile_of_block(int gx, int gy, int gz, int& bx, int& by, int& bz) {
  const int L = blockIdx.x, T = gx * gy;
  if (!AGL_XCD_REMAP) { bz = L / T; const int l = L - bz * T; by = l / gx; bx = l - by * gx; return; }
  int t;
  if (gz >= 8) {
    const int z8 = gz & ~7;
    if (L < T * z8) { const int local = L >> 3; const int q = local / T; bz = q * 8 + (L & 7); t = local - q * T; }
    else { bz = L / T; t = L - bz * T; }
    by = t % gy; bx = t / gy;
    return;
  }
  bz = L / T;
  const int l = L - bz * T, gx8 = gx & ~7;
  if (l < gx8 * gy) { const int local = l >> 3; const int c = local / gy; by = local - c * gy; bx = c * 8 + (l & 7); }
  else { const int r = l - gx8 * gy; const int c = r / gy; by = r - c * gy; bx = gx8 + c; }
}

// ------------------------------------------------------------------ the kernel
// PREC 0: exact fp32 (v_mfma_f32_32x32x2_f32).  PREC 1: operands rounded to bf16 (RNE) when the fragments are read
// from LDS, v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the "bf16" configurations of BASELINE.json); the LDS
// images are then [row][k] with a 20-float row pitch so a lane's 8 consecutive k values are two 16-byte reads.
template <class P, int BM, int BN, int PREC>
__global__ __launch_bounds__(NT, (BM * BN > 128 * 128) ? 2 : 4) void igemm_f32(P p, int gx, int gy, int gz) {
  constexpr int WAVES_M = (BM >= 128) ? 2 : 1;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int WTM = BM / (32 * WAVES_M);
  constexpr int WTN = BN / (32 * WAVES_N);
  static_assert(WTM >= 1 && WTN >= 1, "tile too small for 4 waves");
  constexpr int A_PER = BM * BK / NT, B_PER = BN * BK / NT;
  constexpr int LDH = BK + 4;   // PREC 1 row pitch (floats): 80 B, 16-byte aligned, conflict-free for b128 reads
  constexpr int A_SZ = PREC ? BM * LDH : (P::A_KFAST ? BM * (BK + 1) : BK * BM);
  constexpr int B_SZ = PREC ? BN * LDH : (P::B_KFAST ? BN * (BK + 1) : BK * BN);
  __shared__ __attribute__((aligned(16))) float lds[2 * (A_SZ + B_SZ)];
  float* As = lds;
  float* Bs = lds + 2 * A_SZ;

  int bx, by, bz;
  tile_of_block(gx, gy, gz, bx, by, bz);
  if (!p.setup(bz)) return;
  const int bm0 = by * BM, bn0 = bx * BN;
  if (bm0 >= p.M || bn0 >= p.Nc) return;
  p.tile(bn0, BN);                                   // tile-dependent reduction range (position-major problems)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // hoisted row state (fixed per thread for the whole reduction)
  constexpr int A_ROWS = P::A_KFAST ? A_PER : 1;
  constexpr int B_ROWS = P::B_KFAST ? B_PER : 1;
  typename P::RowA ra_[A_ROWS];
  typename P::RowB rb_[B_ROWS];
  if constexpr (P::A_KFAST) {
#pragma unroll
    for (int j = 0; j < A_PER; ++j) ra_[j] = p.row_a(bm0 + tid / BK + (NT / BK) * j);
  } else {
    ra_[0] = p.row_a(bm0 + tid % BM);
  }
  if constexpr (P::B_KFAST) {
#pragma unroll
    for (int j = 0; j < B_PER; ++j) rb_[j] = p.row_b(bn0 + tid / BK + (NT / BK) * j);
  } else {
    rb_[0] = p.row_b(bn0 + tid % BN);
  }

  float va[A_PER], vb[B_PER];
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a_ptr(), 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.b_ptr(), 0, p.b_bytes, 0x00020000);
  // Element j of the next K-slice.  The loads are issued one or two at a time BETWEEN the MFMA groups of the
  // current slice (see the main loop) so their address arithmetic runs in the shadow of the matrix pipe.
  typename P::KA ka_tile;      // k state of this thread for k-fast operands (one per slice)
  typename P::KB kb_tile;
  const int kga = (!P::A_KFAST && BM >= 64) ? __builtin_amdgcn_readfirstlane(tid / BM) : tid / BM;   // wave-uniform
  const int kgb = (!P::B_KFAST && BN >= 64) ? __builtin_amdgcn_readfirstlane(tid / BN) : tid / BN;   //  -> scalar k decode
  auto gload_begin = [&](int k0) {
    if constexpr (P::A_KFAST) ka_tile = p.k_a(k0 + tid % BK);
    if constexpr (P::B_KFAST) kb_tile = p.k_b(k0 + tid % BK);
  };
  auto gload_a = [&](int j, int k0) {
    unsigned off;
    if constexpr (P::A_KFAST) off = p.off_a(ra_[j], ka_tile);
    else off = p.off_a(ra_[0], p.k_a(k0 + kga + (NT / BM) * j));
    va[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, off, 0, 0));
  };
  auto gload_b = [&](int j, int k0) {
    unsigned off;
    if constexpr (P::B_KFAST) off = p.off_b(rb_[j], kb_tile);
    else off = p.off_b(rb_[0], p.k_b(k0 + kgb + (NT / BN) * j));
    vb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, off, 0, 0));
  };
  auto gload = [&](int k0) {
    gload_begin(k0);
#pragma unroll
    for (int j = 0; j < A_PER; ++j) gload_a(j, k0);
#pragma unroll
    for (int j = 0; j < B_PER; ++j) gload_b(j, k0);
  };
  auto sstore = [&](int buf) {
    float* a = As + buf * A_SZ;
    float* b = Bs + buf * B_SZ;
    if constexpr (PREC) {
#pragma unroll
      for (int j = 0; j < A_PER; ++j) {
        if constexpr (P::A_KFAST) a[(tid / BK + (NT / BK) * j) * LDH + tid % BK] = p.fix_a(va[j]);
        else a[(tid % BM) * LDH + tid / BM + (NT / BM) * j] = p.fix_a(va[j]);
      }
#pragma unroll
      for (int j = 0; j < B_PER; ++j) {
        if constexpr (P::B_KFAST) b[(tid / BK + (NT / BK) * j) * LDH + tid % BK] = p.fix_b(vb[j]);
        else b[(tid % BN) * LDH + tid / BN + (NT / BN) * j] = p.fix_b(vb[j]);
      }
      return;
    }
    if constexpr (P::A_KFAST) {
#pragma unroll
      for (int j = 0; j < A_PER; ++j) a[(tid / BK + (NT / BK) * j) * (BK + 1) + tid % BK] = p.fix_a(va[j]);
    } else {
#pragma unroll
      for (int j = 0; j < A_PER; ++j) a[(tid / BM + (NT / BM) * j) * BM + tid % BM] = p.fix_a(va[j]);
    }
    if constexpr (P::B_KFAST) {
#pragma unroll
      for (int j = 0; j < B_PER; ++j) b[(tid / BK + (NT / BK) * j) * (BK + 1) + tid % BK] = p.fix_b(vb[j]);
    } else {
#pragma unroll
      for (int j = 0; j < B_PER; ++j) b[(tid / BN + (NT / BN) * j) * BN + tid % BN] = p.fix_b(vb[j]);
    }
  };

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  const int arow0 = wm * 32 * WTM + l31, brow0 = wn * 32 * WTN + l31;

  gload(p.kbeg);
  sstore(0);
  __syncthreads();
  int cur = 0;
  for (int k0 = p.kbeg; k0 < p.kend; k0 += BK) {
    // The next slice is fetched unconditionally: past kend every offset is OOB and the loads return zeros, which
    // keeps the loop body free of control flow (the scheduler can then spread the gathers between the MFMAs).
    gload_begin(k0 + BK);
    const float* a = As + cur * A_SZ;
    const float* b = Bs + cur * B_SZ;
    if constexpr (PREC == 0) {
      // Software pipeline, pinned with scheduling barriers (left alone, the compiler sinks every gather to the end of
      // the slice and waits on it at once, exposing the full memory latency each slice): K-step s issues its share of
      // the next slice's gathers and the LDS fragment reads of step s+1, then runs the 4 MFMAs of step s on fragments
      // read one step earlier.
      constexpr int STEPS = BK / 2;
      float fa[2][WTM], fb[2][WTN];
      auto read_frags = [&](int kk, float (&xa)[WTM], float (&xb)[WTN]) {
#pragma unroll
        for (int i = 0; i < WTM; ++i)
          xa[i] = P::A_KFAST ? a[(arow0 + 32 * i) * (BK + 1) + kk + lh] : a[(kk + lh) * BM + arow0 + 32 * i];
#pragma unroll
        for (int j = 0; j < WTN; ++j)
          xb[j] = P::B_KFAST ? b[(brow0 + 32 * j) * (BK + 1) + kk + lh] : b[(kk + lh) * BN + brow0 + 32 * j];
      };
      read_frags(0, fa[0], fb[0]);
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        if (st + 1 < STEPS) read_frags(2 * (st + 1), fa[(st + 1) & 1], fb[(st + 1) & 1]);
#pragma unroll
        for (int j = 0; j < A_PER; ++j)
          if (j * AGL_GATHER_STEPS / A_PER == st) gload_a(j, k0 + BK);
#pragma unroll
        for (int j = 0; j < B_PER; ++j)
          if (j * AGL_GATHER_STEPS / B_PER == st) gload_b(j, k0 + BK);
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
          for (int j = 0; j < WTN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st & 1][i], fb[st & 1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // one 32x32x16 bf16 MFMA per accumulator covers the whole BK=16 slice; lane l supplies k = 8*(l>>5) .. +7
      auto frag = [&](const float* base, int row) {
        const float4 lo = *reinterpret_cast<const float4*>(base + row * LDH + 8 * lh);
        const float4 hi = *reinterpret_cast<const float4*>(base + row * LDH + 8 * lh + 4);
        bf16x8 f;
        f[0] = (__bf16)lo.x; f[1] = (__bf16)lo.y; f[2] = (__bf16)lo.z; f[3] = (__bf16)lo.w;
        f[4] = (__bf16)hi.x; f[5] = (__bf16)hi.y; f[6] = (__bf16)hi.z; f[7] = (__bf16)hi.w;
        return f;
      };
#pragma unroll
      for (int j = 0; j < A_PER; ++j) gload_a(j, k0 + BK);
#pragma unroll
      for (int j = 0; j < B_PER; ++j) gload_b(j, k0 + BK);
      bf16x8 fa[WTM], fb[WTN];
#pragma unroll
      for (int i = 0; i < WTM; ++i) fa[i] = frag(a, arow0 + 32 * i);
#pragma unroll
      for (int j = 0; j < WTN; ++j) fb[j] = frag(b, brow0 + 32 * j);
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    sstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: D[row][col], col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < WTN; ++j) {
    auto c = p.col(bn0 + wn * 32 * WTN + 32 * j + l31);
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
      const int mb = bm0 + wm * 32 * WTM + 32 * i + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {      // (fetching the epilogue operands of a whole tile first was tried: it pushes these
        const int m = mb + (r & 3) + 8 * (r >> 2);   //  128-VGPR kernels into spills and measured 2 % slower at step level)
        if (p.out_ok(m, c)) {
          const long o = p.out_off(m, c);
          p.finish(m, o, acc[i][j][r], p.reads_old() ? p.old_at(o) : 0.f, p.reads_mask() ? p.mask_at(o) : 1.f);
        }
      }
    }
  }
}

// A launch may reduce a second, smaller set of slabs with the same slab count behind the first (the bias-gradient sums beside the
// weight-gradient slabs of csrc/pconv.hip: one launch instead of two): workgroups >= blocks_a work on (slabs_b, out_b, nb).
struct SlabPair { const float* slabs; float* out; long n; int accumulate; };

__global__ void slab_reduce(SlabPair a, SlabPair b, int blocks_a, int splits) {
  const bool second = (int)blockIdx.x >= blocks_a;
  const SlabPair q = second ? b : a;
  const long i = (long)(blockIdx.x - (second ? blocks_a : 0)) * blockDim.x + threadIdx.x;
  if (i >= q.n) return;
  float s = 0.f;
  for (int z = 0; z < splits; ++z) s += q.slabs[(long)z * q.n + i];
  q.out[i] = q.accumulate ? q.out[i] + s : s;
}

// Many slabs of few elements (few.hip: 512 slabs of 64 x 27): one thread per element would walk all slabs serially on a handful
// of workgroups.  Here a workgroup owns 16 consecutive elements and its 16 thread rows take every 16th slab each; the 16
// partial sums are added in row order (deterministic for a given slab count).
__global__ __launch_bounds__(256) void slab_reduce_wide(SlabPair a, SlabPair b, int blocks_a, int splits) {
  __shared__ float part[16][17];
  const bool second = (int)blockIdx.x >= blocks_a;
  const SlabPair q = second ? b : a;
  const int o = threadIdx.x & 15, zl = threadIdx.x >> 4;
  const long i = (long)(blockIdx.x - (second ? blocks_a : 0)) * 16 + o;
  float s = 0.f;
  if (i < q.n)
    for (int z = zl; z < splits; z += 16) s += q.slabs[(long)z * q.n + i];
  part[zl][o] = s;
  __syncthreads();
  if (zl == 0 && i < q.n) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += part[r][o];
    q.out[i] = q.accumulate ? q.out[i] + t : t;
  }
}

}  // namespace
// out[i] (+)= sum over slabs, in slab order (row-interleaved order in the wide form): shared by the weight-gradient paths.
// slabs_b (optional): a second set of `splits` slabs of nb elements reduced by the same launch.
int agl_launch_slab_reduce(const float* slabs, float* out, long n, int splits, int accumulate, hipStream_t st, const char* name,
                           const float* slabs_b, float* out_b, long nb, int accumulate_b) {
  const SlabPair a{slabs, out, n, accumulate}, b{slabs_b, out_b, slabs_b ? nb : 0, accumulate_b};
  if (splits >= 8 && n < (1L << 18)) {
    const int ba = agl_cdiv(n, 16);
    hipLaunchKernelGGL(slab_reduce_wide, dim3(ba + agl_cdiv(b.n, 16)), dim3(256), 0, st, a, b, ba, splits);
  } else {
    const int ba = agl_cdiv(n, 256);
    hipLaunchKernelGGL(slab_reduce, dim3(ba + agl_cdiv(b.n, 256)), dim3(256), 0, st, a, b, ba, splits);
  }
  AGL_CHECK_LAUNCH(name);
  return AGL_OK;
}
namespace {

// dw[co][ci][kh][kw] (+)= t[ci][co][ks-1-kh][ks-1-kw]
__global__ void flip_transpose_w(const float* __restrict__ t, float* __restrict__ dw, int Cout, int Cin, int ks, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int kk = ks * ks;
  if (i >= (long)Cout * Cin * kk) return;
  const int tap = (int)(i % kk), ci = (int)(i / kk % Cin), co = (int)(i / ((long)kk * Cin));
  const float v = t[((long)ci * Cout + co) * kk + (kk - 1 - tap)];
  dw[i] = accumulate ? dw[i] + v : v;
}

// AGL_CONV_DEFER_SUM (include/agl.h): agl_conv2d_fwd / agl_conv2d_bwd_data arm this for the duration of the call; a reduction split
// whose epilogue would only add the slabs (no bias, mask, accumulation, ReLU or divisor) then leaves them in the workspace for the
// caller's next kernel to add in the same fixed order (agl_conv2d_deferred reports where they are) — one launch less per call.
struct DeferredSum { bool armed; const float* slabs; int splits; long stride; };
thread_local DeferredSum g_defer = {false, nullptr, 0, 0};
struct DeferScope {
  explicit DeferScope(int flags) {
    g_defer.armed = (flags & (1 << 24)) != 0;      // AGL_CONV_DEFER_SUM
    if (g_defer.armed) { g_defer.slabs = nullptr; g_defer.splits = 0; g_defer.stride = 0; }
  }
  ~DeferScope() { g_defer.armed = false; }
};
bool defer_split_sum(const float* slabs, long n, int splits, const float* bias, const float* pos_mask, int accumulate, int relu,
                     const float* out_div) {
  if (!g_defer.armed || splits < 2 || bias || pos_mask || accumulate || relu || out_div) return false;
  g_defer.slabs = slabs; g_defer.splits = splits; g_defer.stride = n;
  return true;
}

// out[o] = epilogue(sum_z part[z][o]); channel of o = (o / HW) % C
__global__ void splitk_epilogue(const float* __restrict__ part, float* __restrict__ out, long n, int splits, int HW, int C,
                                const float* __restrict__ bias, const float* __restrict__ pos_mask, int accumulate, int relu,
                                const float* __restrict__ out_div = nullptr) {
  long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n) return;
  float v = 0.f;
  for (int z = 0; z < splits; ++z) v += part[(long)z * n + o];
  if (out_div) v *= 1.0f / *out_div;
  if (bias) v += bias[(o / HW) % C];
  if (pos_mask && !(pos_mask[o] > 0.f)) v = 0.f;
  if (accumulate) v += out[o];
  if (relu) v = fmaxf(v, 0.f);
  out[o] = v;
}

}  // namespace
int agl_launch_splitk_epilogue(const float* slabs, float* out, long n, int splits, int HW, int C, const float* bias, const float* pos_mask,
                               int accumulate, int relu, hipStream_t st, const char* name, const float* out_div) {
  if (defer_split_sum(slabs, n, splits, bias, pos_mask, accumulate, relu, out_div)) return AGL_OK;
  hipLaunchKernelGGL(splitk_epilogue, dim3(agl_cdiv(n, 256)), dim3(256), 0, st, slabs, out, n, splits, HW, C, bias, pos_mask, accumulate, relu,
                     out_div);
  AGL_CHECK_LAUNCH(name);
  return AGL_OK;
}
namespace {

// Reduction splits for the forward / input-gradient passes.  Two reasons to cut K: (1) the output grid alone cannot
// fill the chip (ConvLSTM recurrence steps, 8x8 decoder stem, 2x2 encoder tails); (2) wave quantisation — with 4
// resident workgroups per CU there are 1024 slots, and e.g. 1050 tiles cost two full rounds; s-way splitting turns that
// into ceil(1050 s / 1024) rounds of 1/s the length.  A small cost model picks s: rounds x per-round time + the slab
// traffic of the deterministic split (s writes + s reads of the output).
static int fwd_splits(int M, long Nc, int Z, int K, int* per_out) {
  const int bm = M <= 32 ? 32 : (M <= 64 ? 64 : 128);
  const int bn = M <= 32 ? 256 : 128;
  const long tiles = (long)agl_cdiv(M, bm) * agl_cdiv(Nc, bn) * Z;
  const double slots = 1024.0;                                 // 4 resident workgroups per CU (<=128 VGPRs, 35 KB LDS)
  const double t_k = (double)bm * bn * 2.0 / 115e9;            // seconds per unit of K per workgroup (4 per CU)
  const double out_bytes = (double)M * (double)Nc * Z * 4.0;    // upper bound of the output size
  int best = 1;
  double best_t = 1e30;
  for (int s = 1; s <= 16; ++s) {
    if (s > 1 && K / s < 128) break;
    const double rounds = ceil((double)tiles * s / slots);
    double t = rounds * ((double)K / s) * t_k + AGL_SPLIT_FIXED * (s > 1);
    if (s > 1) t += out_bytes * (2.0 * s + 1.0) / 4.0e12;
    if (t < best_t * 0.97) { best_t = t; best = s; }             // need a 3 % win to take a larger split
  }
  int s = best;
  int per = (K + s - 1) / s;
  per = (per + BK - 1) / BK * BK;
  s = (K + per - 1) / per;
  if (per_out) *per_out = per;
  return s;
}

// Per-call options (the `flags` argument of the agl_conv2d_* entry points; include/agl.h AGL_CONV_*).
struct ConvOpts {
  int prec;          // 0 = fp32 MFMA, 1 = bf16 MFMA operands with fp32 accumulation
  bool patch;        // LDS-patch kernel allowed
  bool patch_s2;     // ... also its stride-2 forward form
  bool pos;          // position-major path on small maps allowed
  bool pos_all_ks;   // experiments: position-major also for 3x3 / 4x4 kernels
  bool split3;       // fp32 operands as three bf16 terms, six bf16-MFMA products (pconv.hip) where that kernel applies
  bool any_grid;     // matrix-core kernels also below their occupancy threshold (unit tests)
  bool w8;           // eight-wave workgroups in the LDS-patch kernels that have that form
  int prio;          // start stagger of the co-resident workgroups of the LDS-patch kernels (0 off, 1..4: delay length)
  int ablate;        // diagnostic kernel builds (flags bits 9..11; results are wrong by construction)
  bool x_bf16;       // AGL_CONV_X_BF16 (1 << 17): the input tensor x holds bf16 elements (bf16 arithmetic, matrix-core kernels only)
  bool y_bf16;       // AGL_CONV_Y_BF16 (1 << 18): agl_conv2d_fwd writes y as bf16 (few-input-channel stream kernel only)
  bool mask_bf16;    // AGL_CONV_MASK_BF16 (1 << 19): agl_conv2d_bwd_data reads pos_mask as bf16 ("same" patch kernel without a reduction split)
  bool dy_bf16;      // AGL_CONV_DY_BF16 (1 << 20): agl_conv2d_bwd_weight reads dy as bf16 (matrix-core kernel, bf16 arithmetic)
  bool x_blk;        // AGL_CONV_X_BLOCKED (1 << 21): the bf16 x is channel-blocked [N][C/8][H][W][8] (forward, weight gradient)
  bool y_blk;        // AGL_CONV_Y_BLOCKED (1 << 22): agl_conv2d_fwd / _addend / _shortcut write y channel-blocked bf16
  bool mask_blk;     // AGL_CONV_MASK_BLOCKED (1 << 23): the bf16 pos_mask of agl_conv2d_bwd_data is channel-blocked
};
constexpr int kPosMinN = 96;    // smallest image count for the position-major path
static ConvOpts conv_opts(int flags) {
  ConvOpts o;
  o.prec = (flags & 1) ? 1 : 0;
  o.patch = !(flags & 2);
  o.patch_s2 = o.patch && !(flags & 4);
  o.pos = !(flags & 8);
  o.pos_all_ks = (flags & 16) != 0;
  o.split3 = (flags & 32) != 0 && o.prec == 0;
  o.any_grid = (flags & 64) != 0;
  o.w8 = (flags & 128) != 0;
  o.prio = (flags >> 14) & 7;
  o.ablate = (flags >> 9) & 31;
  o.x_bf16 = (flags & (1 << 17)) != 0;
  o.y_bf16 = (flags & (1 << 18)) != 0;
  o.mask_bf16 = (flags & (1 << 19)) != 0;
  o.dy_bf16 = (flags & (1 << 20)) != 0;
  o.x_blk = (flags & (1 << 21)) != 0;
  o.y_blk = (flags & (1 << 22)) != 0;
  o.mask_blk = (flags & (1 << 23)) != 0;
  return o;
}
constexpr ConvOpts kDefaultOpts = {0, true, true, true, false, false, false, false, 0, 0, false, false, false, false, false, false, false};

template <class P>
int launch_igemm(P& p, int M, long Nc, int Z, hipStream_t st, const char* name, int prec, int big_tile = 0) {
  AGL_REQUIRE(Nc > 0 && Nc < (1L << 31) && M > 0, "%s: bad GEMM extents M=%d Nc=%ld", name, M, Nc);
#define AGL_LAUNCH(BM_, BN_)                                                                      \
  do {                                                                                            \
    const long gx_ = agl_cdiv(Nc, BN_), gy_ = agl_cdiv(M, BM_);                                   \
    AGL_REQUIRE(gx_ * gy_ * Z < (1L << 31), "%s: grid too large", name);                          \
    dim3 g((unsigned)(gx_ * gy_ * Z));                                                            \
    if (prec) hipLaunchKernelGGL((igemm_f32<P, BM_, BN_, 1>), g, dim3(NT), 0, st, p, (int)gx_, (int)gy_, Z); \
    else hipLaunchKernelGGL((igemm_f32<P, BM_, BN_, 0>), g, dim3(NT), 0, st, p, (int)gx_, (int)gy_, Z);                  \
  } while (0)
  if (M <= 32) AGL_LAUNCH(32, 256);
  else if (M <= 64) AGL_LAUNCH(64, 128);
  else if (big_tile == 1) { if constexpr (P::BIG_TILES) AGL_LAUNCH(256, 128); }
  else if (big_tile == 2) { if constexpr (P::BIG_TILES) AGL_LAUNCH(128, 256); }
  else AGL_LAUNCH(128, 128);
#undef AGL_LAUNCH
  AGL_CHECK_LAUNCH(name);
  return AGL_OK;
}

// ------------------------------------------------------------------ few-output-channel direct convolution
// Convolutions with <= 4 output channels (decoder c4/c7: 64->3 k7; and the input gradients of the 3-channel first
// layers: CropEncoder.c1, OptimizedBlock.resi[0], decoder.c5) would waste 29 of 32 MFMA rows and re-gather every
// tap from HBM.  Here a workgroup owns a TY x (4*TXT) output tile of one image (16x64, or 32x32 on narrow maps), stages
// CB input channels' patches in LDS and accumulates on the VALU.  Each thread owns FOUR horizontally adjacent pixels:
// one row segment of 4+ks-1 patch values (16-byte LDS reads) feeds 4*ks*Cout FMAs, so the kernel is FMA-bound rather
// than LDS-bound (one value per 3 FMAs with a pixel per thread: 9.6 TFLOP/s).  Filter taps are wave-uniform scalar
// loads.  The weight tensor is addressed by strides so the same kernel evaluates the input-gradient form (flipped
// taps, channel roles swapped).  Stride 1 only.
template <int KS, int TXT, int NO>
__global__ __launch_bounds__(256) void small_cout_conv(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ pos_mask,
                                                       float* __restrict__ y, int Cin, int H, int W, int Cout, int OH, int OW,
                                                       int pad, int s_co, int s_ci, int flip, int relu, int accumulate) {
  // Two LDS buffers per operand and a register prefetch: the loads of chunk k+1 are in flight while chunk k is evaluated and are
  // stored into the other buffer before the single barrier of the iteration (the grids of these layers are one or two
  // workgroups per CU, so nothing else would cover the load latency).  NO = output channels actually evaluated (1..4).
  // CB = 4 keeps two buffer pairs under 40 KB: the 393-workgroup grids of the object crops run as one wave of co-resident
  // workgroups instead of two rounds (3x3: 101 -> 68 us)
  constexpr int PX = 4, TX = TXT * PX, TY = 256 / TXT, CB = KS == 1 ? 8 : 4;
  constexpr int PH = TY + KS - 1, PWV = TX + KS - 1, PITCH = (PWV + 3) / 4 * 4, PS = PH * PITCH;
  constexpr int SEG = PX + KS - 1, SEGV = (SEG + 3) / 4;           // row segment a thread reads, in float4s
  constexpr int NE = CB * PH * PWV, ER = (NE + 255) / 256;         // patch elements per chunk, per thread
  constexpr int NW = CB * KS * KS * 4, WR = (NW + 255) / 256;
  __shared__ __attribute__((aligned(16))) float patch[2][CB * PS + 4];      // (+4: dump slot of the table's padding entries)
  __shared__ __attribute__((aligned(16))) float wl[2][NW];          // [c][tap][o padded to 4]: one 16-byte broadcast read per tap
  const int tiles_x = (OW + TX - 1) / TX;
  const int n = blockIdx.y, ty0 = (blockIdx.x / tiles_x) * TY, tx0 = (blockIdx.x % tiles_x) * TX;
  const int tx = threadIdx.x % TXT, ty = threadIdx.x / TXT;
  float acc[NO][PX];
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int q = 0; q < PX; ++q) acc[o][q] = 0.f;
  const float* xn = x + (long)n * Cin * H * W;
  // per-thread element table of a chunk (the same for every chunk), one register per element: LDS slot in the upper 14 bits,
  // source offset within the chunk (< 2^18: launch_small_cout checks CB*H*W) or all ones (zero padding) in the lower 18
  unsigned etab[ER];
#pragma unroll
  for (int r = 0; r < ER; ++r) {
    const int e = threadIdx.x + 256 * r;
    const int c = e / (PH * PWV), rr = e - c * (PH * PWV), py = rr / PWV, pxx = rr - py * PWV;
    const int iy = ty0 - pad + py, ix = tx0 - pad + pxx;
    const bool in = e < NE;
    const unsigned src = (in && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? (unsigned)((c * H + iy) * W + ix) : 0x3FFFFu;
    etab[r] = ((unsigned)(in ? c * PS + py * PITCH + pxx : CB * PS) << 18) | src;     // slot CB*PS..: the 4 spare floats of the buffer
  }
  float pv[ER], pw[WR];
  auto gload = [&](int c0) {
#pragma unroll
    for (int r = 0; r < ER; ++r) {
      const int c = (threadIdx.x + 256 * r) / (PH * PWV);
      const unsigned src = etab[r] & 0x3FFFFu;
      pv[r] = (src != 0x3FFFFu && c0 + c < Cin) ? xn[(long)c0 * H * W + src] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < WR; ++r) {
      const int e = threadIdx.x + 256 * r;
      const int c = e / (KS * KS * 4), t = (e >> 2) % (KS * KS), o = e & 3;
      const int tap = flip ? KS * KS - 1 - t : t;
      pw[r] = (e < NW && c0 + c < Cin && o < Cout) ? w[(long)(c0 + c) * s_ci + (long)o * s_co + tap] : 0.f;
    }
  };
  auto sstore = [&](int b) {
#pragma unroll
    for (int r = 0; r < ER; ++r) patch[b][etab[r] >> 18] = pv[r];
#pragma unroll
    for (int r = 0; r < WR; ++r)
      if (threadIdx.x + 256 * r < NW) wl[b][threadIdx.x + 256 * r] = pw[r];
  };
  gload(0);
  sstore(0);
  __syncthreads();
  int b = 0;
  for (int c0 = 0; c0 < Cin; c0 += CB, b ^= 1) {
    const bool more = c0 + CB < Cin;
    if (more) gload(c0 + CB);
    const int cn = min(CB, Cin - c0);
    // one kernel row (segment + its KS filter taps) is fetched from LDS while the previous one is evaluated: with one or two
    // waves per SIMD nothing else hides the LDS latency
    float4 sg[2][SEGV], wq[2][KS];
    auto lload = [&](int c, int kh, int s) {
      const float4* row = reinterpret_cast<const float4*>(patch[b] + c * PS + (ty + kh) * PITCH + PX * tx);
#pragma unroll
      for (int v4 = 0; v4 < SEGV; ++v4) sg[s][v4] = row[v4];
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) wq[s][kw] = *reinterpret_cast<const float4*>(wl[b] + ((c * KS + kh) * KS + kw) * 4);
    };
    lload(0, 0, 0);
    for (int c = 0; c < cn; c += 2) {      // two channels per trip: the register set of every row is a compile-time choice
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        if (c + cc >= cn) break;
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
          const int cur = (cc * KS + kh) & 1;
          if (kh + 1 < KS) lload(c + cc, kh + 1, cur ^ 1);
          else if (c + cc + 1 < cn) lload(c + cc + 1, 0, cur ^ 1);
          float seg[SEGV * 4];
#pragma unroll
          for (int v4 = 0; v4 < SEGV; ++v4) {
            seg[4 * v4 + 0] = sg[cur][v4].x; seg[4 * v4 + 1] = sg[cur][v4].y; seg[4 * v4 + 2] = sg[cur][v4].z; seg[4 * v4 + 3] = sg[cur][v4].w;
          }
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            const float wv[4] = {wq[cur][kw].x, wq[cur][kw].y, wq[cur][kw].z, wq[cur][kw].w};
#pragma unroll
            for (int o = 0; o < NO; ++o)
#pragma unroll
              for (int q = 0; q < PX; ++q) acc[o][q] = fmaf(seg[q + kw], wv[o], acc[o][q]);
          }
        }
      }
    }
    if (more) sstore(b ^ 1);      // the other buffer: last read one iteration ago, before that iteration's barrier
    __syncthreads();
  }
  const int oy = ty0 + ty;
  if (oy < OH) {
#pragma unroll
    for (int o = 0; o < NO; ++o)
      if (o < Cout) {
#pragma unroll
        for (int q = 0; q < PX; ++q) {
          const int ox = tx0 + PX * tx + q;
          if (ox >= OW) continue;
          const long idx = ((long)n * Cout + o) * OH * OW + (long)oy * OW + ox;
          float v = acc[o][q];
          if (bias) v += bias[o];
          if (pos_mask && !(pos_mask[idx] > 0.f)) v = 0.f;
          if (accumulate) v += y[idx];
          if (relu) v = fmaxf(v, 0.f);
          y[idx] = v;
        }
      }
  }
}

int launch_small_cout(const float* x, const float* w, const float* bias, const float* pos_mask, float* y, int N, int Cin, int H,
                      int W, int Cout, int ks, int pad, int s_co, int s_ci, int flip, int relu, int accumulate, hipStream_t st,
                      const char* name) {
  const int OH = H + 2 * pad - ks + 1, OW = W + 2 * pad - ks + 1;
  const bool wide = OW > 32;                       // 16 x 64 tiles on wide maps, 32 x 32 otherwise
  dim3 g(wide ? agl_cdiv(OH, 16) * agl_cdiv(OW, 64) : agl_cdiv(OH, 32) * agl_cdiv(OW, 32), N);
#define AGL_SC2(KS_, TXT_, NO_)                                                                                                   \
  hipLaunchKernelGGL((small_cout_conv<KS_, TXT_, NO_>), g, dim3(256), 0, st, x, w, bias, pos_mask, y, Cin, H, W, Cout, OH, OW, pad, \
                     s_co, s_ci, flip, relu, accumulate)
#define AGL_SC(KS_)                                                                          \
  case KS_:                                                                                  \
    if (wide) { if (Cout == 3) AGL_SC2(KS_, 16, 3); else AGL_SC2(KS_, 16, 4); }              \
    else { if (Cout == 3) AGL_SC2(KS_, 8, 3); else AGL_SC2(KS_, 8, 4); }                     \
    break;
  switch (ks) { AGL_SC(1) AGL_SC(3) AGL_SC(4) AGL_SC(5) AGL_SC(7) default: return AGL_ERR_ARG; }
#undef AGL_SC
#undef AGL_SC2
  AGL_CHECK_LAUNCH(name);
  return AGL_OK;
}

// ------------------------------------------------------------------ LDS-patch ("direct") convolution, stride 1
// For 3x3 / 5x5 stride-1 convolutions on power-of-two maps the im2col gather fetches every input element ks^2
// times, each with its own address arithmetic.  Here a workgroup owns BM output channels x 128 output pixels laid
// out as TI images x TH x TW, and per chunk of CB input channels stages (a) the raw input patch
// TI x CB x (TH+ks-1) x (TW+ks-1) (zero padded, optional nearest up-sampling / input ReLU) and (b) the weight
// slice [CB*ks^2][BM] in LDS.  The MFMA B fragment for k = (c,kh,kw) is then read straight from the patch at
// pixel_offset(lane) + c*PS + kh*PW + kw — the tap offsets are compile-time constants.  The same kernel evaluates
// the stride-1 input gradient (flipped taps, channel roles swapped through the weight strides).
struct PatchArgs {
  const float* x; const float* w; const float* bias; const float* pos_mask; float* y; float* part;
  int N, Cin, H, W, Cout, OH, OW;   // H, W: stored input map (logical size H<<up); OH, OW: output map
  int pad, up, in_relu, relu, accumulate, stride;
  int w_sm, w_sc, flip;             // element strides of w for (output channel m, input channel c); flipped taps
  int splits, c_per_split;
  long slab;
  unsigned x_bytes, w_bytes;
};

// WVEC: the weight slice of a row is contiguous in memory (forward pass): stage it with coalesced 16-byte loads.
// S: convolution stride (forward form only): the patch covers S*(T-1)+KS input rows/columns and a lane's pixel offset
// is S*(py*PW + px) — the 4x4/stride-2 encoder layers and the 3x3/stride-2 box form of the discriminator blocks.
template <int KS, int TW, int TH, int TI, int CB, int BM, bool WVEC, int S = 1>
__global__ __launch_bounds__(NT, 3) void patch_conv(PatchArgs p) {
  constexpr int BN = 128, PH = S * (TH - 1) + KS, PW = S * (TW - 1) + KS, PS = PH * PW, KC = CB * KS * KS;
  constexpr int NP = TI * CB * PS, PL = (NP + NT - 1) / NT, WL = (KC * BM + NT - 1) / NT;
  constexpr int LDW = BM + 1;                                  // odd pitch: both staging patterns stay (nearly) conflict free
  constexpr int QR = KC / 4, WV = (QR * BM + NT - 1) / NT;     // WVEC: float4 per row, float4 loads per thread
  constexpr int WAVES_M = BM >= 128 ? 2 : 1, WAVES_N = 4 / WAVES_M, WTM = BM / (32 * WAVES_M), WTN = BN / (32 * WAVES_N);
  static_assert(KC % 2 == 0 && TI * TH * TW == BN, "patch_conv geometry");
  __shared__ float patch[NP];
  __shared__ float wt[KC * LDW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int Hl = p.H << p.up, Wl = p.W << p.up;
  int img0, ty0, tx0;
  if constexpr (TI == 1) {
    const int tpr = p.OW / TW, tpi = (p.OH / TH) * tpr;
    img0 = blockIdx.x / tpi;
    const int t = blockIdx.x - img0 * tpi;
    ty0 = (t / tpr) * TH; tx0 = (t % tpr) * TW;
  } else {
    img0 = blockIdx.x * TI; ty0 = 0; tx0 = 0;
  }
  const int bm0 = blockIdx.y * BM;
  int c_beg = 0, c_end = p.Cin;
  float* outp = p.y;
  if (p.splits > 1) {
    c_beg = blockIdx.z * p.c_per_split; c_end = min(p.Cin, c_beg + p.c_per_split);
    outp = p.part + (long)blockIdx.z * p.slab;
  }
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  // per-thread constants of the two staging passes
  unsigned poff_src[PL];   // byte offset of patch element e (without the channel-chunk term), or OOB
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const int e = tid + NT * i;
    const int ti = e / (CB * PS), r = e - ti * (CB * PS), c = r / PS, q = r - c * PS, yy = q / PW, xx = q - yy * PW;
    const int img = img0 + ti, ly = S * ty0 - p.pad + yy, lx = S * tx0 - p.pad + xx;
    const bool ok = e < NP && img < p.N && (unsigned)ly < (unsigned)Hl && (unsigned)lx < (unsigned)Wl;
    poff_src[i] = ok ? (unsigned)(((img * p.Cin + c) * p.H + (ly >> p.up)) * p.W + (lx >> p.up)) * 4u : OOB;
  }
  const int wm_row = tid % BM;                       // weight staging: lanes along m, k_local = tid/BM + (NT/BM)*j
  const int wkq = BM >= 64 ? __builtin_amdgcn_readfirstlane(tid / BM) : tid / BM;
  const unsigned w_row_off = (bm0 + wm_row < p.Cout) ? (unsigned)((bm0 + wm_row) * p.w_sm) * 4u : OOB;

  float pl[PL], wl[WVEC ? 4 * WV : WL];
  bool wok[WVEC ? WV : 1];
  auto gload = [&](int c0) {
    const unsigned cstep = (unsigned)(c0 * p.H * p.W) * 4u;
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int e = tid + NT * i;
      const int c = (e % (CB * PS)) / PS;
      const unsigned off = (poff_src[i] != OOB && c0 + c < c_end) ? poff_src[i] + cstep : OOB;
      pl[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, off, 0, 0));
    }
    if constexpr (WVEC) {
      static_assert(!WVEC || KC % 4 == 0, "vector weight staging needs KC % 4 == 0");
#pragma unroll
      for (int j = 0; j < WV; ++j) {
        const int f = tid + NT * j, m = f / QR, q = f - m * QR;          // lanes run along k inside a weight row
        const bool ok = f < QR * BM && bm0 + m < p.Cout && c0 < c_end;   // chunks are CB-aligned: whole row valid or not
        // (this ROCm build lowers __builtin_amdgcn_raw_buffer_load_b128 to a single dword load, so the 16-byte
        //  fetch is a plain global load from a clamped, always-valid address; invalid rows are zeroed when staged)
        const long idx = ok ? (long)(bm0 + m) * p.w_sm + (long)c0 * p.w_sc + 4 * q : 0;
        const float4 v = *reinterpret_cast<const float4*>(p.w + idx);
        wok[j] = ok;
        wl[4 * j + 0] = v.x; wl[4 * j + 1] = v.y; wl[4 * j + 2] = v.z; wl[4 * j + 3] = v.w;
      }
    } else {
#pragma unroll
    for (int j = 0; j < WL; ++j) {
      const int k = wkq + (NT / BM) * j;             // scalar
      const int c = k / (KS * KS), t = k - c * (KS * KS);
      const int tap = p.flip ? KS * KS - 1 - t : t;
      const unsigned off = (w_row_off != OOB && k < KC && c0 + c < c_end) ? w_row_off + (unsigned)((c0 + c) * p.w_sc + tap) * 4u : OOB;
      wl[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsW, off, 0, 0));
    }
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int e = tid + NT * i;
      if (e < NP) patch[e] = p.in_relu ? fmaxf(pl[i], 0.f) : pl[i];
    }
    if constexpr (WVEC) {
#pragma unroll
      for (int j = 0; j < WV; ++j) {
        const int f = tid + NT * j, m = f / QR, q = f - m * QR;
        if (f < QR * BM) {
#pragma unroll
          for (int e = 0; e < 4; ++e) wt[(4 * q + e) * LDW + m] = wok[j] ? wl[4 * j + e] : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < WL; ++j)
        if (tid / BM + (NT / BM) * j < KC) wt[(tid / BM + (NT / BM) * j) * LDW + wm_row] = wl[j];
    }
  };

  // per-lane pixel offsets inside the patch image for the WTN 32-pixel sub-tiles of this wave
  int ppix[WTN];
#pragma unroll
  for (int jt = 0; jt < WTN; ++jt) {
    const int j = wn * 32 * WTN + 32 * jt + l31;
    const int ti = j / (TH * TW), r = j - ti * (TH * TW), py = r / TW, px = r - py * TW;
    ppix[jt] = ti * (CB * PS) + S * (py * PW + px);
  }
  const int arow0 = wm * 32 * WTM + l31;

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gload(c_beg);
  sstore();
  __syncthreads();
  for (int c0 = c_beg; c0 < c_end; c0 += CB) {
    gload(c0 + CB);                                   // past c_end everything is OOB -> zeros
#pragma unroll
    for (int kk = 0; kk < KC; kk += 2) {
      // tap offsets of k = kk and kk+1 are compile-time constants
      constexpr auto PB = [](int k) { const int c = k / (KS * KS), t = k - c * (KS * KS); return c * PS + (t / KS) * PW + t % KS; };
      const int pb = lh ? PB(kk + 1) : PB(kk);
      float fa[WTM], fb[WTN];
#pragma unroll
      for (int i = 0; i < WTM; ++i) fa[i] = wt[(kk + lh) * LDW + arow0 + 32 * i];
#pragma unroll
      for (int jt = 0; jt < WTN; ++jt) fb[jt] = patch[ppix[jt] + pb];
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int jt = 0; jt < WTN; ++jt)
          acc[i][jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[jt], acc[i][jt], 0, 0, 0);
    }
    __syncthreads();
    sstore();
    __syncthreads();
  }

  const long OHW = (long)p.OH * p.OW;
#pragma unroll
  for (int jt = 0; jt < WTN; ++jt) {
    const int j = wn * 32 * WTN + 32 * jt + l31;
    const int ti = j / (TH * TW), r = j - ti * (TH * TW), py = r / TW, px = r - py * TW;
    const int img = img0 + ti;
    const long pbase = (long)img * p.Cout * OHW + (long)(ty0 + py) * p.OW + tx0 + px;
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
      const int mb = bm0 + wm * 32 * WTM + 32 * i + 4 * lh;
      float old[16], mk[16];
      const bool rd_old = p.splits <= 1 && p.accumulate, rd_mask = p.splits <= 1 && p.pos_mask != nullptr;
      if (rd_old) {          // operands of the epilogue are fetched for the whole tile first (no load-wait chain per element)
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) {
          const int m = mb + (r2 & 3) + 8 * (r2 >> 2);
          old[r2] = (m < p.Cout && img < p.N) ? outp[pbase + (long)m * OHW] : 0.f;
        }
      }
      if (rd_mask) {
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) {
          const int m = mb + (r2 & 3) + 8 * (r2 >> 2);
          mk[r2] = (m < p.Cout && img < p.N) ? p.pos_mask[pbase + (long)m * OHW] : 1.f;
        }
      }
#pragma unroll
      for (int r2 = 0; r2 < 16; ++r2) {
        const int m = mb + (r2 & 3) + 8 * (r2 >> 2);
        if (m < p.Cout && img < p.N) {
          const long o = pbase + (long)m * OHW;
          float v = acc[i][jt][r2];
          if (p.splits > 1) { outp[o] = v; continue; }
          if (p.bias) v += p.bias[m];
          if (rd_mask && !(mk[r2] > 0.f)) v = 0.f;
          if (rd_old) v += old[r2];
          if (p.relu) v = fmaxf(v, 0.f);
          outp[o] = v;
        }
      }
    }
  }
}

// returns AGL_OK when launched, -1 when the shape is not eligible (caller falls back to the im2col kernel)
int try_patch_conv(PatchArgs& a, int ks, void* ws, long ws_bytes, hipStream_t st, const char* name, const ConvOpts& co) {
  if (co.prec != 0) return -1;                       // fp32 path only (bf16 mode uses the im2col kernel)
  const bool s2 = a.stride == 2;
  if (s2 && (a.flip || a.up || (ks != 3 && ks != 4) || !co.patch_s2)) return -1;
  if (!s2 && ks != 3) return -1;                              // 5x5 (ConvLSTM at 8x8): measured no faster than im2col
  if (a.flip && a.OW < 32) return -1;                         // input-gradient form: wins on >= 32-wide maps only
  const int cb = s2 ? 4 : 8;
  if (a.Cin % cb != 0 || a.Cout < 48) return -1;
  int geo;
  if (a.OW >= 16 && a.OW % 16 == 0 && a.OH % 8 == 0) geo = 0;
  else if (a.OW == 8 && a.OH == 8) geo = 1;
  else if (a.OW == 4 && a.OH == 4) geo = 2;
  else return -1;
  const int bm = a.Cout <= 64 ? 64 : 128;
  const long ptiles = geo == 0 ? (long)a.N * (a.OH / 8) * (a.OW / 16) : (geo == 1 ? agl_cdiv(a.N, 2) : agl_cdiv(a.N, 8));
  const long tiles = ptiles * agl_cdiv(a.Cout, bm);
  // channel-chunk splits (same quantisation-aware idea as fwd_splits)
  const long out_numel = (long)a.N * a.Cout * a.OH * a.OW;
  int best = 1; double best_t = 1e30;
  const int chunks = a.Cin / cb;
  for (int s = 1; s <= 16 && s <= chunks; ++s) {
    if (s > 1 && (chunks / s) * cb * ks * ks < 128) break;
    const double rounds = ceil((double)tiles * s / 768.0);
    double t = rounds * ((double)a.Cin * ks * ks / s) * (bm * 128 * 2.0 / 150e9) + AGL_SPLIT_FIXED * (s > 1);
    if (s > 1) t += (double)out_numel * 4.0 * (2.0 * s + 1.0) / 4.0e12;
    if (t < best_t * 0.97) { best_t = t; best = s; }
  }
  int splits = best;
  int cps = agl_cdiv(agl_cdiv(chunks, splits), 1) * cb;
  splits = agl_cdiv(a.Cin, cps);
  if (splits > 1 && (!ws || ws_bytes < (long)splits * out_numel * 4)) { splits = 1; cps = a.Cin; }
  a.splits = splits; a.c_per_split = cps; a.slab = out_numel; a.part = (float*)ws;
  dim3 g((unsigned)ptiles, agl_cdiv(a.Cout, bm), splits);
#define AGL_PC(KS_, TW_, TH_, TI_, CB_, WV_)                                                                   \
  do {                                                                                                          \
    if (bm == 64) hipLaunchKernelGGL((patch_conv<KS_, TW_, TH_, TI_, CB_, 64, WV_>), g, dim3(NT), 0, st, a);     \
    else hipLaunchKernelGGL((patch_conv<KS_, TW_, TH_, TI_, CB_, 128, WV_>), g, dim3(NT), 0, st, a);            \
  } while (0)
#define AGL_PC2(KS_, TW_, TH_, TI_)                                                                             \
  do {                                                                                                          \
    if (bm == 64) hipLaunchKernelGGL((patch_conv<KS_, TW_, TH_, TI_, 4, 64, true, 2>), g, dim3(NT), 0, st, a);   \
    else hipLaunchKernelGGL((patch_conv<KS_, TW_, TH_, TI_, 4, 128, true, 2>), g, dim3(NT), 0, st, a);          \
  } while (0)
  if (s2 && ks == 3) {
    if (geo == 0) AGL_PC2(3, 16, 8, 1); else if (geo == 1) AGL_PC2(3, 8, 8, 2); else AGL_PC2(3, 4, 4, 8);
  } else if (s2) {
    if (geo == 0) AGL_PC2(4, 16, 8, 1); else if (geo == 1) AGL_PC2(4, 8, 8, 2); else AGL_PC2(4, 4, 4, 8);
  } else if (ks == 3 && !a.flip) {       // forward: weight rows contiguous -> vector staging
    if (geo == 0) AGL_PC(3, 16, 8, 1, 8, true); else if (geo == 1) AGL_PC(3, 8, 8, 2, 8, true); else AGL_PC(3, 4, 4, 8, 8, true);
  } else {
    AGL_PC(3, 16, 8, 1, 8, false);
  }
#undef AGL_PC2
#undef AGL_PC
  AGL_CHECK_LAUNCH(name);
  if (splits > 1 && !defer_split_sum((const float*)ws, out_numel, splits, a.bias, a.pos_mask, a.accumulate, a.relu, nullptr)) {
    hipLaunchKernelGGL(splitk_epilogue, dim3(agl_cdiv(out_numel, 256)), dim3(256), 0, st, (const float*)ws, a.y, out_numel, splits,
                       a.OH * a.OW, a.Cout, a.bias, a.pos_mask, a.accumulate, a.relu, (const float*)nullptr);
    AGL_CHECK_LAUNCH(name);
  }
  return AGL_OK;
}

// ---- position-major path (small maps): eligibility, scratch layout, forward launcher
// Eligibility.  The transposes are a fixed cost per activation byte, so the path pays only where the arithmetic intensity
// is high and the padded share large: measured +20 % on the 5x5 / 8x8 ConvLSTM input convolutions (512->512: 3.33 ->
// 2.67 ms), +9 % on 512->1024 4x4/s2 at 4x4, nothing on the 3x3 discriminator layers (their transposes cost what the
// skipped taps save) — so it is taken for 5x5 kernels only.
static bool pos_ok(const ConvOpts& co, int N, int Cred, int H, int W, int Crow, int ks, int up) {   // Cred: reduction channels, Crow: GEMM rows
  return co.pos && up == 0 && H <= 8 && W <= 8 && H * W >= 4 && (ks == 5 || (co.pos_all_ks && ks >= 3)) && N >= kPosMinN && Cred >= 64 && (Cred & (Cred - 1)) == 0 &&
         Crow >= 64;
}
static int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }
// valid taps summed over the output positions of a ks/stride/pad convolution on an H x W map
static long pos_valid_taps(int H, int W, int OH, int OW, int ks, int stride, int pad) {
  long rows = 0, cols = 0;
  for (int o = 0; o < OH; ++o) { const int i0 = o * stride - pad; rows += std::min(ks - 1, H - 1 - i0) - std::max(0, -i0) + 1; }
  for (int o = 0; o < OW; ++o) { const int i0 = o * stride - pad; cols += std::min(ks - 1, W - 1 - i0) - std::max(0, -i0) + 1; }
  return rows * cols;
}
struct PosPlan { long xt, wt, yt, slabs; int splits; long total() const { return (xt + wt + yt + slabs) * 4; } };
static PosPlan pos_fwd_plan(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad) {
  const int OH = (H + 2 * pad - ks) / stride + 1, OW = (W + 2 * pad - ks) / stride + 1, Q = OH * OW;
  PosPlan pl;
  pl.xt = (long)N * Cin * H * W; pl.wt = (long)Cout * Cin * ks * ks; pl.yt = (long)N * Cout * Q;
  const int Kavg = (int)(pos_valid_taps(H, W, OH, OW, ks, stride, pad) * Cin / Q);
  pl.splits = fwd_splits(Cout, (long)N * Q, 1, Kavg, nullptr);
  pl.slabs = pl.splits > 1 ? (long)pl.splits * pl.yt : 0;
  return pl;
}
static int launch_transpose_in(const float* x, float* xt, int N, int C, int HW, int in_relu, hipStream_t st) {
  hipLaunchKernelGGL(nchw_to_pcn_k, dim3(C, agl_cdiv(N, 64)), dim3(256), 0, st, x, xt, N, C, HW, in_relu);
  AGL_CHECK_LAUNCH("position-major transpose (in)");
  return AGL_OK;
}
static int launch_transpose_out(const float* yt, float* y, const float* pos_mask, int N, int C, int HW, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(pcn_to_nchw_k, dim3(C, agl_cdiv(N, 64)), dim3(256), 0, st, yt, y, pos_mask, N, C, HW, accumulate);
  AGL_CHECK_LAUNCH("position-major transpose (out)");
  return AGL_OK;
}
static int pos_conv_fwd(const float* x, const float* w, const float* bias, float* y, void* ws, int N, int Cin, int H, int W, int Cout,
                        int ks, int stride, int pad, int in_relu, int relu, int accumulate, hipStream_t st, int prec, int wmode = 0,
                        const float* pos_mask = nullptr) {
  const int OH = (H + 2 * pad - ks) / stride + 1, OW = (W + 2 * pad - ks) / stride + 1, Q = OH * OW, KK = ks * ks;
  const PosPlan pl = pos_fwd_plan(N, Cin, H, W, Cout, ks, stride, pad);
  float* xt = (float*)ws; float* wt = xt + pl.xt; float* yt = wt + pl.wt; float* slabs = yt + pl.yt;
  int rc = launch_transpose_in(x, xt, N, Cin, H * W, in_relu, st);
  if (rc) return rc;
  hipLaunchKernelGGL(w_to_tap_major_k, dim3(agl_cdiv(pl.wt, 256)), dim3(256), 0, st, w, wt, Cout, Cin, KK, wmode);
  AGL_CHECK_LAUNCH("agl_conv2d_fwd(position-major weights)");
  rc = AGL_ERR_ARG;
#define AGL_PF(KS_)                                                                                             \
  case KS_: {                                                                                                   \
    PosFwdProb<KS_> p;                                                                                          \
    p.xt = xt; p.wt = wt; p.bias = bias; p.yt = yt; p.N = N; p.Cin = Cin; p.cin_shift = ilog2(Cin); p.H = H; p.W = W;  \
    p.Cout = Cout; p.OH = OH; p.OW = OW; p.stride = stride; p.pad = pad; p.relu = relu; p.M = Cout; p.Nc = N * Q; \
    p.kbeg = 0; p.kend = 0; p.a_bytes = (unsigned)(pl.wt * 4); p.b_bytes = (unsigned)(pl.xt * 4);               \
    p.Q = Q; p.nfull = N / 128 * 128;                                                                            \
    p.splits = pl.splits; p.slab = pl.yt; p.part = slabs;                                                        \
    rc = launch_igemm(p, Cout, (long)N * Q, pl.splits, st, "agl_conv2d_fwd(position-major)", prec);                   \
  } break;
  switch (ks) { AGL_PF(3) AGL_PF(4) AGL_PF(5) AGL_PF(7) }
#undef AGL_PF
  if (rc != AGL_OK) return rc;
  if (pl.splits > 1) {
    hipLaunchKernelGGL(splitk_epilogue, dim3(agl_cdiv(pl.yt, 256)), dim3(256), 0, st, (const float*)slabs, yt, pl.yt, pl.splits, N, Cout,
                       bias, (const float*)nullptr, 0, relu, (const float*)nullptr);
    AGL_CHECK_LAUNCH("agl_conv2d_fwd(position-major split-K epilogue)");
  }
  return launch_transpose_out(yt, y, pos_mask, N, Cout, Q, accumulate, st);
}

bool ks_ok(int k) { return k == 1 || k == 3 || k == 4 || k == 5 || k == 7; }

// Arithmetic pipe of the main kernel the last agl_conv2d_* call of this thread launched (agl_conv2d_last_pipe):
// 0 exact fp32 (fp32 MFMA or VALU), 1 bf16 MFMA (one product per MAC), 3 bf16 MFMA with split operands (six products per MAC)
thread_local int g_last_pipe = 0;

}  // namespace

extern "C" {
long agl_conv2d_fwd_packed_bytes(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2, int flags);
long agl_conv2d_bwd_data_packed_bytes(int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int ks, int stride, int pad, int flags);

// Bytes of split-K scratch the forward / input-gradient pass wants for these extents (0 = none needed).
long agl_conv2d_splitk_ws_bytes(int M, long out_pixels, int phases, int K, long out_numel) {
  int per;
  const int s = fwd_splits(M, out_pixels, phases, K, &per);
  return s > 1 ? (long)s * out_numel * 4 : 0;
}

// Scratch the forward pass wants for these extents: the larger of the split-K slabs and, on small maps, the
// position-major operands (0 = none needed; with less, the pass falls back to paths that need less).
long agl_conv2d_fwd_ws_bytes(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2) {
  const int Hl = H << up_log2, Wl = W << up_log2;
  const int OH = (Hl + 2 * pad - ks) / stride + 1, OW = (Wl + 2 * pad - ks) / stride + 1;
  long need = agl_conv2d_splitk_ws_bytes(Cout, (long)N * OH * OW, 1, Cin * ks * ks, (long)N * Cout * OH * OW);
  if (pos_ok(kDefaultOpts, N, Cin, H, W, Cout, ks, up_log2)) {
    const long pn = pos_fwd_plan(N, Cin, H, W, Cout, ks, stride, pad).total();
    if (pn > need) need = pn;
  }
  need = std::max(need, pconv_ws_bytes_split(Cin, Cout, ks, 3, (long)N * Cout * OH * OW));
  if (Cout <= 4 && stride == 1 && up_log2 == 0) need = std::max(need, pconv_vert_ws_bytes(N, Cin, H, W, Cout, ks, 3));
  return need;
}

static int conv2d_fwd_impl(const float* x, const float* w, const void* packed_w, const float* packed_div, const float* bias, float* y,
                           void* ws, long ws_bytes, int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2,
                           int in_relu, int relu, int accumulate, int flags, void* stream, float* stats, long stats_floats,
                           int* stat_rows, const InFold* fold = nullptr, const float* addend = nullptr, const float* sc_x = nullptr,
                           const float* sc_w = nullptr, const float* sc_b = nullptr, int sc_cin = 0);

int agl_conv2d_fwd(const float* x, const float* w, const void* packed_w, const float* packed_div, const float* bias, float* y, void* ws,
                   long ws_bytes, int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2, int in_relu, int relu,
                   int accumulate, int flags, void* stream) {
  const DeferScope defer(flags);
  return conv2d_fwd_impl(x, w, packed_w, packed_div, bias, y, ws, ws_bytes, N, Cin, H, W, Cout, ks, stride, pad, up_log2, in_relu, relu,
                         accumulate, flags, stream, nullptr, 0, nullptr);
}

int agl_conv2d_deferred(const float** slabs, int* splits, long long* stride) {
  AGL_REQUIRE(slabs && splits && stride, "agl_conv2d_deferred: null pointer");
  *slabs = g_defer.slabs; *splits = g_defer.splits; *stride = g_defer.stride;
  return AGL_OK;
}

// Forward convolution that may also hand back the BatchNorm partial sums of its output (include/agl.h)
int agl_conv2d_fwd_stats(const float* x, const float* w, const void* packed_w, const float* packed_div, const float* bias, float* y,
                         void* ws, long ws_bytes, int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2,
                         int in_relu, int flags, float* stats, long stats_floats, int* stat_rows, void* stream) {
  AGL_REQUIRE(stats && stat_rows && stats_floats >= 0, "agl_conv2d_fwd_stats: null statistics buffer");
  *stat_rows = 0;
  return conv2d_fwd_impl(x, w, packed_w, packed_div, bias, y, ws, ws_bytes, N, Cin, H, W, Cout, ks, stride, pad, up_log2, in_relu, 0, 0,
                         flags, stream, stats, stats_floats, stat_rows);
}

long agl_conv2d_fwd_stats_floats(int N, int Cout, int OH, int OW) { return pconv_stat_rows_max(N, OH, OW) * Cout * 3; }

// Forward convolution with the normalise-modulate of the BatchNorm that reads x folded into the input staging (include/agl.h):
// y = conv(relu?(fma(x, scale, shift))) with the tables of agl_norm_fold_table (shift = beta - mean * scale), zero padding after the
// transform; optionally the BatchNorm partial rows of y.  in_mean is not read (it may be NULL): the tables carry the mean.
int agl_conv2d_fwd_fold(const float* x, const float* in_mean, const float* in_scale, const float* in_shift, int in_per_n, const float* w,
                        const void* packed_w, const float* packed_div, const float* bias, float* y, void* ws, long ws_bytes, int N, int Cin,
                        int H, int W, int Cout, int ks, int stride, int pad, int in_relu, int flags, float* stats, long stats_floats,
                        int* stat_rows, void* stream) {
  AGL_REQUIRE(in_scale && in_shift, "agl_conv2d_fwd_fold: null table");
  AGL_REQUIRE((stats == nullptr) == (stat_rows == nullptr), "agl_conv2d_fwd_fold: stats and stat_rows go together");
  if (stat_rows) *stat_rows = 0;
  const InFold f{in_mean, in_scale, in_shift, in_per_n};
  return conv2d_fwd_impl(x, w, packed_w, packed_div, bias, y, ws, ws_bytes, N, Cin, H, W, Cout, ks, stride, pad, 0, in_relu, 0, 0, flags,
                         stream, stats, stats_floats, stat_rows, &f, nullptr);
}
int agl_conv2d_fwd_fold_ok(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int flags) {
  if (!(ks == 4 && stride == 2)) return 0;      // (the transform is compiled into the 4x4 / stride-2 instantiations of the patch kernel)
  return agl_conv2d_fwd_packed_bytes(N, Cin, H, W, Cout, ks, stride, pad, 0, flags) > 0 ? 1 : 0;
}
int agl_conv2d_fwd_spade_ok(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int flags);
// y = conv(relu?(SPADE(x))) with SPADE's normalise-modulate (normalization.py:97,106) applied while the convolution stages x — the
// 128 px decoder's spade_4 -> c6 (5x5) and spade_5 -> c7 (7x7 to 3 channels: vertical + diagonal form), generator_obj_att128.py:588-597,
// in bf16 arithmetic.  cells / map / G: agl_spade_cells and the class-grid map of the modulation (include/agl.h).
int agl_conv2d_fwd_spade(const float* x, const float* mean, const float* rstd, const float* cells, const int* map, int G, const float* w,
                         const void* packed_w, const float* packed_div, const float* bias, float* y, void* ws, long ws_bytes, int N, int Cin,
                         int H, int W, int Cout, int ks, int stride, int pad, int in_relu, int flags, float* stats, long stats_floats,
                         int* stat_rows, void* stream) {
  AGL_REQUIRE(x && mean && rstd && cells && map && G > 0 && y && (w || packed_w), "agl_conv2d_fwd_spade: null pointer");
  AGL_REQUIRE(agl_conv2d_fwd_spade_ok(N, Cin, H, W, Cout, ks, stride, pad, flags), "agl_conv2d_fwd_spade: a shape / arithmetic the folded forms do not take (ask agl_conv2d_fwd_spade_ok)");
  if (stat_rows) *stat_rows = 0;
  const InFold f{mean, rstd, nullptr, 0, cells, map, G};
  if (Cout <= 4) {
    AGL_REQUIRE(w, "agl_conv2d_fwd_spade: the vertical form packs w itself");
    PVertArgs v{x, w, bias, nullptr, y, N, Cin, H, W, Cout, ks, pad, Cin * ks * ks, ks * ks, 0, 0, 0, 1, 0, f, in_relu};
    const int vrc = pconv_vert_try(v, ws, ws_bytes, (hipStream_t)stream, "agl_conv2d_fwd_spade(vertical + diagonal)");
    AGL_REQUIRE(vrc >= 0, "agl_conv2d_fwd_spade: the vertical form did not take the shape (workspace?)");
    g_last_pipe = 1;
    return vrc;
  }
  return conv2d_fwd_impl(x, w, packed_w, packed_div, bias, y, ws, ws_bytes, N, Cin, H, W, Cout, ks, stride, pad, 0, in_relu, 0, 0, flags, stream,
                         stats, stats_floats, stat_rows, &f);
}
int agl_conv2d_fwd_spade_ok(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int flags) {
  const ConvOpts co = conv_opts(flags);
  if (co.prec != 1 || !co.patch || stride != 1 || 2 * pad != ks - 1 || Cin % 16 != 0) return 0;
  if (Cout <= 4) return pconv_vert_ws_bytes(N, Cin, H, W, Cout, ks, 1) > 0 ? 1 : 0;
  if (ks != 5) return 0;      // (the transform is compiled into the bf16 5x5 instantiations of the patch kernel)
  return agl_conv2d_fwd_packed_bytes(N, Cin, H, W, Cout, ks, stride, pad, 0, flags) > 0 ? 1 : 0;
}
// y = conv(x) + addend (+ bias, output ReLU), written out of place — as bf16 with AGL_CONV_Y_BF16: the sum is formed in fp32 and rounded
// once.  The shortcut of a discriminator block (discriminator.py:58-60, :97-99) whose sum only convolutions read afterwards.
// y = conv(x) + bias + [1x1 shortcut of a few-channel tensor: sc_b[m] + sum_c sc_w[m][c] * sc_x[n][c][pixel]] (+ output ReLU), the shortcut
// evaluated in the convolution's epilogue — the first discriminator block, out = c2(h) + sc(x) (discriminator.py:36-44, :58-60), in ONE
// launch that reads 3 input channels instead of adding a second 64-channel tensor.  x may be bf16 (AGL_CONV_X_BF16), y bf16
// (AGL_CONV_Y_BF16).  bf16 3x3 stride-1 "same" forms of the patch kernel with <= 64 output channels (agl_conv2d_fwd_shortcut_ok).
int agl_conv2d_fwd_shortcut(const float* x, const float* w, const void* packed_w, const float* packed_div, const float* bias, const float* sc_x,
                            const float* sc_w, const float* sc_bias, int sc_cin, float* y, void* ws, long ws_bytes, int N, int Cin, int H, int W,
                            int Cout, int ks, int pad, int in_relu, int relu, int flags, void* stream) {
  AGL_REQUIRE(sc_x && sc_w && sc_cin >= 1 && sc_cin <= 4, "agl_conv2d_fwd_shortcut: bad shortcut operands");
  return conv2d_fwd_impl(x, w, packed_w, packed_div, bias, y, ws, ws_bytes, N, Cin, H, W, Cout, ks, 1, pad, 0, in_relu, relu, 0, flags, stream,
                         nullptr, 0, nullptr, nullptr, nullptr, sc_x, sc_w, sc_bias, sc_cin);
}
int agl_conv2d_fwd_shortcut_ok(int N, int Cin, int H, int W, int Cout, int ks, int pad, int flags) {
  const ConvOpts co = conv_opts(flags);
  if (co.prec != 1 || !co.patch || ks != 3 || pad != 1 || Cout > 64 || W % 4 != 0) return 0;
  PConvArgs a{};
  a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = H; a.OW = W; a.ks = 3; a.stride = 1; a.pad = 1; a.nsplit = 1;
  a.any_grid = co.any_grid;
  if (pconv_plan_splits(a) != 1) return 0;
  return (H == 4 && W == 4) ? 0 : 1;      // (not the 4 x 4-map tile form)
}
int agl_conv2d_fwd_addend(const float* x, const float* w, const void* packed_w, const float* packed_div, const float* bias, const float* addend,
                          float* y, void* ws, long ws_bytes, int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int in_relu,
                          int relu, int flags, void* stream) {
  AGL_REQUIRE(addend, "agl_conv2d_fwd_addend: null addend");
  return conv2d_fwd_impl(x, w, packed_w, packed_div, bias, y, ws, ws_bytes, N, Cin, H, W, Cout, ks, stride, pad, 0, in_relu, relu, 0, flags,
                         stream, nullptr, 0, nullptr, nullptr, addend);
}

static int conv2d_fwd_impl(const float* x, const float* w, const void* packed_w, const float* packed_div, const float* bias, float* y,
                           void* ws, long ws_bytes, int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2,
                           int in_relu, int relu, int accumulate, int flags, void* stream, float* stats, long stats_floats,
                           int* stat_rows, const InFold* fold, const float* addend, const float* sc_x, const float* sc_w, const float* sc_b,
                           int sc_cin) {
  AGL_REQUIRE(x && (w || packed_w) && y, "agl_conv2d_fwd: null pointer");
  g_last_pipe = 0;
  const ConvOpts co = conv_opts(flags);
  AGL_REQUIRE(ks_ok(ks) && (stride == 1 || stride == 2) && up_log2 >= 0 && up_log2 <= 4,
              "agl_conv2d_fwd: unsupported ks=%d stride=%d up=%d", ks, stride, up_log2);
  const int Hl = H << up_log2, Wl = W << up_log2;
  const int OH = (Hl + 2 * pad - ks) / stride + 1, OW = (Wl + 2 * pad - ks) / stride + 1;
  AGL_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && OH > 0 && OW > 0, "agl_conv2d_fwd: empty extent");
  AGL_REQUIRE((long)N * Cin * H * W < (1L << 30) && (long)N * Cout * OH * OW < (1L << 30) && (long)Cout * Cin * ks * ks < (1L << 30),
              "agl_conv2d_fwd: tensor too large (operands are addressed with 32-bit byte offsets: < 2^30 elements)");
  hipStream_t st = (hipStream_t)stream;
  if (fold || addend || sc_x) {      // folded input transform / out-of-place addend / epilogue shortcut: forms of the matrix-core patch kernel only
    AGL_REQUIRE((co.prec == 1 || co.split3) && co.patch && !accumulate && (!co.x_bf16 || !fold), "agl_conv2d_fwd_fold: needs the matrix-core patch kernel");
    AGL_REQUIRE(!co.y_bf16 || co.prec == 1, "agl_conv2d_fwd: AGL_CONV_Y_BF16 needs AGL_CONV_BF16");
    PConvArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.pos_mask = nullptr; a.y = y; a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout;
    a.OH = OH; a.OW = OW; a.ks = ks; a.stride = stride; a.pad = pad; a.up = up_log2; a.in_relu = in_relu; a.relu = relu;
    a.accumulate = 0; a.w_sm = Cin * ks * ks; a.w_sc = ks * ks; a.flip = 0; a.nsplit = co.prec == 1 ? 1 : 3; a.any_grid = co.any_grid;
    a.w8 = co.w8; a.prio = co.prio;
    a.stats = stats; a.stats_floats = stats_floats; a.stat_rows = stat_rows;
    a.packed = packed_w; a.out_div = packed_w ? packed_div : nullptr;
    if (fold) a.fold = *fold;
    a.addend = addend; a.y_bf16 = co.y_bf16; a.x_bf16 = co.x_bf16; a.x_blk = co.x_blk; a.y_blk = co.y_blk;
    AGL_REQUIRE((!co.x_blk || co.x_bf16) && (!co.y_blk || co.y_bf16), "agl_conv2d_fwd: a blocked operand is a bf16 operand (AGL_CONV_X_BF16 / _Y_BF16)");
    a.sc_x = sc_x; a.sc_w = sc_w; a.sc_b = sc_b; a.sc_cin = sc_cin;
    const int prc = pconv_try(a, ws, ws_bytes, st, "agl_conv2d_fwd(pconv, folded input transform / addend)");
    AGL_REQUIRE(prc >= 0, "agl_conv2d_fwd_fold: a shape the patch kernel does not take in this form (ask agl_conv2d_fwd_fold_ok)");
    g_last_pipe = a.nsplit;
    return prc;
  }
  if (co.x_bf16) {      // a bf16 input exists for the matrix-core patch kernel in bf16 arithmetic only (the caller asked agl_conv2d_fwd_packed_bytes)
    AGL_REQUIRE(co.prec == 1 && co.patch && !(relu && accumulate), "agl_conv2d_fwd: AGL_CONV_X_BF16 needs AGL_CONV_BF16 and the patch kernel");
    if (Cout <= 4 && stride == 1 && up_log2 == 0 && !in_relu && OH == H && OW == W && w && !co.y_bf16) {      // few output channels, 7x7
      PVertArgs v{x, w, bias, nullptr, y, N, Cin, H, W, Cout, ks, pad, Cin * ks * ks, ks * ks, 0, relu, accumulate, 1, 1};
      const int vrc = pconv_vert_try(v, ws, ws_bytes, st, "agl_conv2d_fwd(vertical + diagonal, bf16 input)");
      AGL_REQUIRE(vrc >= 0, "agl_conv2d_fwd: AGL_CONV_X_BF16 on a few-output-channel shape the vertical form does not take");
      g_last_pipe = 1;
      return vrc;
    }
    PConvArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.pos_mask = nullptr; a.y = y; a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout;
    a.OH = OH; a.OW = OW; a.ks = ks; a.stride = stride; a.pad = pad; a.up = up_log2; a.in_relu = in_relu; a.relu = relu;
    a.accumulate = accumulate; a.w_sm = Cin * ks * ks; a.w_sc = ks * ks; a.flip = 0; a.nsplit = 1; a.any_grid = co.any_grid;
    a.stats = stats; a.stats_floats = stats_floats; a.stat_rows = stat_rows;
    a.packed = packed_w; a.out_div = packed_w ? packed_div : nullptr; a.x_bf16 = 1; a.y_bf16 = co.y_bf16;
    a.x_blk = co.x_blk; a.y_blk = co.y_blk;
    AGL_REQUIRE(!co.y_blk || co.y_bf16, "agl_conv2d_fwd: AGL_CONV_Y_BLOCKED needs AGL_CONV_Y_BF16 (ask agl_conv2d_fwd_takes_blocked)");
    const int prc = pconv_try(a, ws, ws_bytes, st, "agl_conv2d_fwd(pconv, bf16 input)");
    AGL_REQUIRE(prc >= 0, "agl_conv2d_fwd: AGL_CONV_X_BF16 on a shape the patch kernel does not take");
    g_last_pipe = 1;
    return prc;
  }
  AGL_REQUIRE(!co.y_bf16 || Cout > 4, "agl_conv2d_fwd: AGL_CONV_Y_BF16 on a few-output-channel layer (ask agl_conv2d_fwd_writes_bf16_y)");
  if (Cout <= 4 && stride == 1 && up_log2 == 0 && !in_relu && OH == H && OW == W && w && co.patch && (co.prec == 1 || co.split3)) {
    // few output channels, 7x7: vertical convolution on the matrix cores + diagonal sum (pconv.hip)
    PVertArgs v{x, w, bias, nullptr, y, N, Cin, H, W, Cout, ks, pad, Cin * ks * ks, ks * ks, 0, relu, accumulate, co.prec == 1 ? 1 : 3};
    const int vrc = pconv_vert_try(v, ws, ws_bytes, st, "agl_conv2d_fwd(vertical + diagonal)");
    if (vrc >= 0) { g_last_pipe = v.nsplit; return vrc; }
  }
  if (Cout <= 4 && stride == 1 && up_log2 == 0 && !in_relu && OH * OW >= 64 && (long)H * W * 8 < (1L << 18) && w)   // (linear layers, HW = 1, stay on the GEMM; 18-bit patch table)
    return launch_small_cout(x, w, bias, nullptr, y, N, Cin, H, W, Cout, ks, pad, Cin * ks * ks, ks * ks, 0, relu, accumulate, st,
                             "agl_conv2d_fwd(small Cout)");
  if (Cin <= 4 && Cout >= 16 && stride == 1 && up_log2 == 0 && OH == H && OW == W && w && co.patch) {
    // few input channels, 1x1 / 3x3: a stream over the output (few.hip; exact fp32 on the vector units in every arithmetic mode)
    const int frc = few_cin_fwd_try(x, w, bias, y, N, Cin, H, W, Cout, ks, in_relu, relu, accumulate, co.y_blk ? 2 : (co.y_bf16 ? 1 : 0), st,
                                    "agl_conv2d_fwd(few input channels)");
    if (frc >= 0) return frc;
  }
  AGL_REQUIRE(!co.y_bf16 || co.prec == 1, "agl_conv2d_fwd: AGL_CONV_Y_BF16 needs AGL_CONV_BF16");
  if (co.patch && (co.prec == 1 || co.split3) && !(relu && accumulate)) {
    PConvArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.pos_mask = nullptr; a.y = y; a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout;
    a.OH = OH; a.OW = OW; a.ks = ks; a.stride = stride; a.pad = pad; a.up = up_log2; a.in_relu = in_relu; a.relu = relu;
    a.accumulate = accumulate; a.w_sm = Cin * ks * ks; a.w_sc = ks * ks; a.flip = 0; a.nsplit = co.prec == 1 ? 1 : 3; a.any_grid = co.any_grid; a.w8 = co.w8; a.prio = co.prio; a.ablate = co.ablate;
    a.stats = stats; a.stats_floats = stats_floats; a.stat_rows = stat_rows;
    a.packed = packed_w; a.out_div = packed_w ? packed_div : nullptr; a.y_bf16 = co.y_bf16; a.y_blk = co.y_blk;
    AGL_REQUIRE(!co.x_blk, "agl_conv2d_fwd: AGL_CONV_X_BLOCKED needs AGL_CONV_X_BF16");
    const int prc = pconv_try(a, ws, ws_bytes, st, "agl_conv2d_fwd(pconv)");
    if (prc >= 0) { g_last_pipe = a.nsplit; return prc; }
  }
  AGL_REQUIRE(!co.y_blk, "agl_conv2d_fwd: AGL_CONV_Y_BLOCKED on a call no blocked-writing kernel takes (ask agl_conv2d_fwd_takes_blocked)");
  AGL_REQUIRE(!co.y_bf16, "agl_conv2d_fwd: AGL_CONV_Y_BF16 on a call no bf16-writing kernel takes (ask agl_conv2d_fwd_writes_bf16_y)");
  AGL_REQUIRE(w, "agl_conv2d_fwd: packed weights were given for a call that does not run on the LDS-patch kernel, and w is NULL "
                 "(ask agl_conv2d_fwd_packed_bytes first)");
  g_last_pipe = co.prec;
  if (pos_ok(co, N, Cin, H, W, Cout, ks, up_log2) && !(relu && accumulate)) {
    const PosPlan pl = pos_fwd_plan(N, Cin, H, W, Cout, ks, stride, pad);
    if (ws && ws_bytes >= pl.total())
      return pos_conv_fwd(x, w, bias, y, ws, N, Cin, H, W, Cout, ks, stride, pad, in_relu, relu, accumulate, st, co.prec);
  }
  if (co.patch) {
    PatchArgs a;
    a.x = x; a.w = w; a.bias = bias; a.pos_mask = nullptr; a.y = y; a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout;
    a.OH = OH; a.OW = OW; a.pad = pad; a.up = up_log2; a.in_relu = in_relu; a.relu = relu; a.accumulate = accumulate;
    a.stride = stride;
    a.w_sm = Cin * ks * ks; a.w_sc = ks * ks; a.flip = 0;
    a.x_bytes = (unsigned)((long)N * Cin * H * W * 4); a.w_bytes = (unsigned)((long)Cout * Cin * ks * ks * 4);
    const int prc = try_patch_conv(a, ks, ws, ws_bytes, st, "agl_conv2d_fwd(patch)", co);
    if (prc >= 0) return prc;
  }
  const long out_numel = (long)N * Cout * OH * OW;
  int per = 0, rc = AGL_ERR_ARG;
  int splits = fwd_splits(Cout, (long)N * OH * OW, 1, Cin * ks * ks, &per);
  if (splits > 1 && (!ws || ws_bytes < (long)splits * out_numel * 4)) { splits = 1; per = Cin * ks * ks; }   // no scratch: unsplit
#define AGL_FWD(KS_)                                                                                                  \
  case KS_: {                                                                                                         \
    FwdProb<KS_> p;                                                                                                   \
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.OH = OH;      \
    p.OW = OW; p.stride = stride; p.pad = pad; p.up = up_log2; p.relu = relu; p.accumulate = accumulate; p.in_relu = in_relu; \
    p.M = Cout; p.Nc = N * OH * OW; p.K = Cin * KS_ * KS_; p.kbeg = 0; p.kend = p.K; p.HW = H * W; p.OHW = OH * OW;  \
    p.a_bytes = (unsigned)((long)Cout * Cin * KS_ * KS_ * 4); p.b_bytes = (unsigned)((long)N * Cin * H * W * 4);          \
    p.splits = splits; p.per_split = per; p.slab = out_numel; p.part = (float*)ws;                                    \
    rc = launch_igemm(p, p.M, p.Nc, splits, st, "agl_conv2d_fwd", co.prec);                                                    \
  } break;
  switch (ks) { AGL_FWD(1) AGL_FWD(3) AGL_FWD(4) AGL_FWD(5) AGL_FWD(7) }
#undef AGL_FWD
  if (rc != AGL_OK) return rc;
  if (splits > 1 && !defer_split_sum((const float*)ws, out_numel, splits, bias, nullptr, accumulate, relu, nullptr)) {
    hipLaunchKernelGGL(splitk_epilogue, dim3(agl_cdiv(out_numel, 256)), dim3(256), 0, st, (const float*)ws, y, out_numel, splits,
                       OH * OW, Cout, bias, (const float*)nullptr, accumulate, relu, (const float*)nullptr);
    AGL_CHECK_LAUNCH("agl_conv2d_fwd(split-K epilogue)");
  }
  return AGL_OK;
}

// 3x3 stride-2 input gradient: split plan.  Small maps: tap-proportional splits (balanced workgroups, 4*s slabs);
// larger maps: uniform splits.
static int bwd32_plan(int N, int Cin, int IH, int IW, int Cout, int* tap_split) {
  const long maxNc = (long)N * ((IH + 1) / 2) * ((IW + 1) / 2);
  if (IH <= 9 && IW <= 9) {
    *tap_split = 1;
    return 4 * fwd_splits(Cin, maxNc, 9, Cout, nullptr);        // nine (phase, tap) slices of K = Cout each
  }
  *tap_split = 0;
  return fwd_splits(Cin, maxNc, 4, Cout * 9 / 4, nullptr);
}

// Scratch the input-gradient pass wants (split-K slabs, position-major operands on small maps).
long agl_conv2d_bwd_data_ws_bytes(int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int ks, int stride, int pad) {
  const long out_numel = (long)N * Cin * IH * IW;
  long need;
  if (stride == 2 && ks == 3) {
    int ts;
    const int s = bwd32_plan(N, Cin, IH, IW, Cout, &ts);
    need = s > 1 ? (long)s * out_numel * 4 : 0;
  } else {
    const int tpa = (ks + stride - 1) / stride;
    need = agl_conv2d_splitk_ws_bytes(Cin, (long)N * ((IH + stride - 1) / stride) * ((IW + stride - 1) / stride), stride * stride,
                                      Cout * tpa * tpa, out_numel);
  }
  if (stride == 1 && IH == OH && IW == OW && pos_ok(kDefaultOpts, N, Cout, OH, OW, Cin, ks, 0)) {
    const long pn = pos_fwd_plan(N, Cout, OH, OW, Cin, ks, 1, ks - 1 - pad).total();
    if (pn > need) need = pn;
  }
  if (stride == 1) need = std::max(need, pconv_ws_bytes_split(Cout, Cin, ks, 3, (long)N * Cin * IH * IW));
  if (stride == 2 && ks == 4) need = std::max(need, pconvT_ws_bytes_split(Cout, Cin, 3, (long)N * Cin * IH * IW));
  if (Cin <= 4 && stride == 1) need = std::max(need, pconv_vert_ws_bytes(N, Cout, OH, OW, Cin, ks, 3));
  return need;
}

// dx[N,Cin,IH,IW] = conv2d_backward_input(dy[N,Cout,OH,OW], w[Cout,Cin,ks,ks]).  Also the forward of
// ConvTranspose2d (weight [C_in_T = Cout][C_out_T = Cin][ks][ks]).
int agl_conv2d_bwd_data(const float* dy, const float* w, const void* packed_w, const float* packed_div, const float* bias,
                        const float* pos_mask, float* dx, void* ws, long ws_bytes, int N, int Cin, int IH, int IW, int Cout, int OH, int OW,
                        int ks, int stride, int pad, int relu, int accumulate, int flags, void* stream) {
  AGL_REQUIRE(dy && (w || packed_w) && dx, "agl_conv2d_bwd_data: null pointer");
  const DeferScope defer(flags);
  g_last_pipe = 0;
  const ConvOpts co = conv_opts(flags);
  AGL_REQUIRE(ks_ok(ks) && ((stride == 1) || (stride == 2 && (ks == 4 || ks == 3))), "agl_conv2d_bwd_data: unsupported ks=%d stride=%d", ks, stride);
  AGL_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && OH > 0 && OW > 0 && IH > 0 && IW > 0, "agl_conv2d_bwd_data: empty extent");
  AGL_REQUIRE((IH + 2 * pad - ks) / stride + 1 == OH && (IW + 2 * pad - ks) / stride + 1 == OW,
              "agl_conv2d_bwd_data: inconsistent extents IH=%d IW=%d OH=%d OW=%d", IH, IW, OH, OW);
  AGL_REQUIRE((long)N * Cin * IH * IW < (1L << 30) && (long)N * Cout * OH * OW < (1L << 30) && (long)Cout * Cin * ks * ks < (1L << 30),
              "agl_conv2d_bwd_data: tensor too large (< 2^30 elements per operand)");
  hipStream_t st = (hipStream_t)stream;
  if (ks == 1 && IH == 1 && IW == 1 && OH == 1 && OW == 1 && pad == 0 && w && !bias && !relu && co.patch)      // nn.Linear (few.hip)
    return linear_bwd_data_launch(dy, w, pos_mask, dx, N, Cin, Cout, accumulate, co.prec == 1, st, "agl_conv2d_bwd_data(linear)");
  if (Cin <= 4 && stride == 1 && IH == OH && IW == OW && w && co.patch && (co.prec == 1 || co.split3)) {
    PVertArgs v{dy, w, bias, pos_mask, dx, N, Cout, OH, OW, Cin, ks, ks - 1 - pad, ks * ks, Cin * ks * ks, 1, relu, accumulate, co.prec == 1 ? 1 : 3};
    const int vrc = pconv_vert_try(v, ws, ws_bytes, st, "agl_conv2d_bwd_data(vertical + diagonal)");
    if (vrc >= 0) { g_last_pipe = v.nsplit; return vrc; }
  }
  if (Cin <= 4 && stride == 1 && IH * IW >= 64 && (long)OH * OW * 8 < (1L << 18) && w)   // dx = conv(dy, flipped taps, channel roles swapped), pad' = ks-1-pad
    return launch_small_cout(dy, w, bias, pos_mask, dx, N, Cout, OH, OW, Cin, ks, ks - 1 - pad, ks * ks, Cin * ks * ks, 1, relu,
                             accumulate, st, "agl_conv2d_bwd_data(small Cin)");
  if (co.x_bf16) {      // dy stored as bf16 (the bf16-stored input of a transposed convolution): matrix-core kernels in bf16 arithmetic only
    AGL_REQUIRE(co.prec == 1 && co.patch && Cin > 4 && !bias && !(relu && accumulate) && !co.mask_bf16,
                "agl_conv2d_bwd_data: AGL_CONV_X_BF16 needs AGL_CONV_BF16 and a matrix-core form (ask agl_conv2d_bwd_data_takes_bf16_dy)");
    PConvArgs a{};
    a.x = dy; a.w = w; a.bias = nullptr; a.pos_mask = pos_mask; a.y = dx; a.N = N; a.Cin = Cout; a.H = OH; a.W = OW; a.Cout = Cin;
    a.OH = IH; a.OW = IW; a.ks = ks; a.up = 0; a.in_relu = 0; a.relu = relu; a.accumulate = accumulate; a.nsplit = 1;
    a.any_grid = co.any_grid; a.packed = packed_w; a.out_div = packed_w ? packed_div : nullptr; a.x_bf16 = 1;
    int prc = -1;
    if (stride == 1 && IH == OH && IW == OW) {
      a.stride = 1; a.pad = ks - 1 - pad; a.w_sm = ks * ks; a.w_sc = Cin * ks * ks; a.flip = 1;
      prc = pconv_try(a, ws, ws_bytes, st, "agl_conv2d_bwd_data(pconv, bf16 dy)");
    } else if (stride == 2 && ks == 4) {
      a.stride = 2; a.pad = pad; a.w_sm = 16; a.w_sc = Cin * 16; a.flip = 0;
      prc = pconvT_try(a, ws, ws_bytes, st, "agl_conv2d_bwd_data(pconv phases, bf16 dy)");
    }
    AGL_REQUIRE(prc >= 0, "agl_conv2d_bwd_data: AGL_CONV_X_BF16 on a shape the matrix-core kernels do not take");
    g_last_pipe = 1;
    return prc;
  }
  if (co.mask_bf16 && stride == 2) {      // 4x4 / stride-2 / pad-1: the paired-phase kernel reads the bf16 mask
    AGL_REQUIRE(ks == 4 && co.patch && co.prec == 1 && pos_mask && Cin > 4 && !bias && !(relu && accumulate),
                "agl_conv2d_bwd_data: AGL_CONV_MASK_BF16 with stride 2 needs the 4x4 phase kernel in bf16 arithmetic");
    PConvArgs a{};
    a.x = dy; a.w = w; a.bias = nullptr; a.pos_mask = pos_mask; a.y = dx; a.N = N; a.Cin = Cout; a.H = OH; a.W = OW; a.Cout = Cin;
    a.OH = IH; a.OW = IW; a.ks = 4; a.stride = 2; a.pad = pad; a.up = 0; a.in_relu = 0; a.relu = relu; a.accumulate = accumulate;
    a.w_sm = 16; a.w_sc = Cin * 16; a.flip = 0; a.nsplit = 1; a.any_grid = co.any_grid;
    a.packed = packed_w; a.out_div = packed_w ? packed_div : nullptr; a.mask_bf16 = 1; a.mask_blk = co.mask_blk;
    const int prc = pconvT_try(a, ws, ws_bytes, st, "agl_conv2d_bwd_data(pconv phases, bf16 mask)");
    AGL_REQUIRE(prc >= 0, "agl_conv2d_bwd_data: AGL_CONV_MASK_BF16 on a shape the phase kernel does not take in that form (ask agl_conv2d_bwd_data_takes_bf16_mask)");
    g_last_pipe = 1;
    return prc;
  }
  if (co.mask_bf16)
    AGL_REQUIRE(stride == 1 && co.patch && IH == OH && IW == OW && co.prec == 1 && pos_mask && Cin > 4 && !(relu && accumulate),
                "agl_conv2d_bwd_data: AGL_CONV_MASK_BF16 needs the stride-1 patch kernel in bf16 arithmetic (ask agl_conv2d_bwd_data_takes_bf16_mask)");
  if (stride == 1 && co.patch && IH == OH && IW == OW && (co.prec == 1 || co.split3) && !(relu && accumulate)) {
    PConvArgs a{};    // "same" convolution: dx = conv(dy, flipped taps, channel roles swapped), pad' = ks-1-pad
    a.x = dy; a.w = w; a.bias = bias; a.pos_mask = pos_mask; a.y = dx; a.N = N; a.Cin = Cout; a.H = OH; a.W = OW; a.Cout = Cin;
    a.OH = IH; a.OW = IW; a.ks = ks; a.stride = 1; a.pad = ks - 1 - pad; a.up = 0; a.in_relu = 0; a.relu = relu;
    a.accumulate = accumulate; a.w_sm = ks * ks; a.w_sc = Cin * ks * ks; a.flip = 1; a.nsplit = co.prec == 1 ? 1 : 3; a.any_grid = co.any_grid; a.w8 = co.w8; a.prio = co.prio; a.ablate = co.ablate;
    a.packed = packed_w; a.out_div = packed_w ? packed_div : nullptr; a.mask_bf16 = co.mask_bf16; a.mask_blk = co.mask_blk;
    const int prc = pconv_try(a, ws, ws_bytes, st, "agl_conv2d_bwd_data(pconv)");
    if (prc >= 0) { g_last_pipe = a.nsplit; return prc; }
  }
  AGL_REQUIRE(!co.mask_bf16, "agl_conv2d_bwd_data: AGL_CONV_MASK_BF16 on a shape the patch kernel does not take without a reduction split");
  g_last_pipe = co.prec;
  if (stride == 1)
    AGL_REQUIRE(w, "agl_conv2d_bwd_data: packed weights were given for a call that does not run on the LDS-patch kernel, and w is NULL "
                   "(ask agl_conv2d_bwd_data_packed_bytes first)");
  if (stride == 1 && IH == OH && IW == OW && !bias && !relu && pos_ok(co, N, Cout, OH, OW, Cin, ks, 0)) {
    // "same" convolution: dx = forward convolution of dy with flipped taps, channel roles swapped, pad ks-1-pad
    const PosPlan pl = pos_fwd_plan(N, Cout, OH, OW, Cin, ks, 1, ks - 1 - pad);
    if (ws && ws_bytes >= pl.total())
      return pos_conv_fwd(dy, w, nullptr, dx, ws, N, Cout, OH, OW, Cin, ks, 1, ks - 1 - pad, 0, 0, accumulate, st, co.prec, 2, pos_mask);
  }
  if (stride == 1 && co.patch && IH == OH && IW == OW) {   // "same" convolution: dx = conv(dy, flipped taps, roles swapped)
    PatchArgs a;
    a.x = dy; a.w = w; a.bias = bias; a.pos_mask = pos_mask; a.y = dx; a.N = N; a.Cin = Cout; a.H = OH; a.W = OW; a.Cout = Cin;
    a.OH = IH; a.OW = IW; a.pad = ks - 1 - pad; a.up = 0; a.in_relu = 0; a.relu = relu; a.accumulate = accumulate;
    a.stride = 1;
    a.w_sm = ks * ks; a.w_sc = Cin * ks * ks; a.flip = 1;
    a.x_bytes = (unsigned)((long)N * Cout * OH * OW * 4); a.w_bytes = (unsigned)((long)Cout * Cin * ks * ks * 4);
    const int prc = try_patch_conv(a, ks, ws, ws_bytes, st, "agl_conv2d_bwd_data(patch)", co);
    if (prc >= 0) return prc;
  }
  if (stride == 2 && ks == 4 && co.patch && (co.prec == 1 || co.split3) && !bias) {
    PConvArgs a{};
    a.x = dy; a.w = w; a.bias = nullptr; a.pos_mask = pos_mask; a.y = dx; a.N = N; a.Cin = Cout; a.H = OH; a.W = OW; a.Cout = Cin;
    a.OH = IH; a.OW = IW; a.ks = 4; a.stride = 2; a.pad = pad; a.up = 0; a.in_relu = 0; a.relu = relu; a.accumulate = accumulate;
    a.w_sm = 16; a.w_sc = Cin * 16; a.flip = 0; a.nsplit = co.prec == 1 ? 1 : 3; a.any_grid = co.any_grid; a.w8 = co.w8; a.prio = co.prio; a.ablate = co.ablate;
    a.packed = packed_w; a.out_div = packed_w ? packed_div : nullptr;
    const int prc = pconvT_try(a, ws, ws_bytes, st, "agl_conv2d_bwd_data(pconv phases)");
    if (prc >= 0) { g_last_pipe = a.nsplit; return prc; }
  }
  AGL_REQUIRE(w, "agl_conv2d_bwd_data: packed weights were given for a call that does not run on the LDS-patch kernel, and w is NULL "
                 "(ask agl_conv2d_bwd_data_packed_bytes first)");
  g_last_pipe = co.prec;
  const long out_numel = (long)N * Cin * IH * IW;
  const int tpa = (ks + stride - 1) / stride;                    // taps per axis of the fullest stride phase
  const int phases = stride * stride, Kp = Cout * tpa * tpa;
  int per = 0, rc = AGL_ERR_ARG;
  int splits = fwd_splits(Cin, (long)N * ((IH + stride - 1) / stride) * ((IW + stride - 1) / stride), phases, Kp, &per);
  int tap_split = 0;
  if (stride == 2 && ks == 3) splits = bwd32_plan(N, Cin, IH, IW, Cout, &tap_split);
  if (splits > 1 && (!ws || ws_bytes < (long)splits * out_numel * 4)) { splits = 1; per = Kp; tap_split = 0; }
#define AGL_BWD(KS_, S_)                                                                                            \
  {                                                                                                                 \
    BwdDataProb<KS_, S_> p;                                                                                         \
    p.dy = dy; p.w = w; p.dx = dx; p.bias = bias; p.pos_mask = pos_mask; p.N = N; p.Cin = Cin; p.IH = IH; p.IW = IW; p.Cout = Cout;       \
    p.OH = OH; p.OW = OW; p.pad = pad; p.accumulate = accumulate; p.relu = relu; p.IHW = IH * IW; p.OHW = OH * OW; \
    p.a_bytes = (unsigned)((long)Cout * Cin * KS_ * KS_ * 4); p.b_bytes = (unsigned)((long)N * Cout * OH * OW * 4);      \
    long maxNc = (long)N * ((IH + S_ - 1) / S_) * ((IW + S_ - 1) / S_);                                             \
    p.splits = splits; p.per_split = per; p.slab = out_numel; p.part = (float*)ws;                                  \
    rc = launch_igemm(p, Cin, maxNc, S_ * S_ * splits, st, "agl_conv2d_bwd_data", co.prec);                                  \
  }
  if (stride == 2 && ks == 3) {
    BwdDataGenProb<3, 2> p;
    p.dy = dy; p.w = w; p.dx = dx; p.bias = bias; p.pos_mask = pos_mask; p.N = N; p.Cin = Cin; p.IH = IH; p.IW = IW; p.Cout = Cout;
    p.OH = OH; p.OW = OW; p.pad = pad; p.accumulate = accumulate; p.relu = relu; p.IHW = IH * IW; p.OHW = OH * OW;
    p.a_bytes = (unsigned)((long)Cout * Cin * 9 * 4); p.b_bytes = (unsigned)((long)N * Cout * OH * OW * 4);
    long maxNc = (long)N * ((IH + 1) / 2) * ((IW + 1) / 2);
    p.splits = splits; p.slab = out_numel; p.part = (float*)ws; p.kbeg = 0; p.kend = 0; p.M = Cin; p.Nc = 0;
    p.tap_split = tap_split;
    rc = launch_igemm(p, Cin, maxNc, 4 * splits, st, "agl_conv2d_bwd_data(3x3 stride 2)", co.prec);
  } else if (stride == 2) AGL_BWD(4, 2)
  else switch (ks) {
    case 1: AGL_BWD(1, 1) break;
    case 3: AGL_BWD(3, 1) break;
    case 4: AGL_BWD(4, 1) break;
    case 5: AGL_BWD(5, 1) break;
    case 7: AGL_BWD(7, 1) break;
  }
#undef AGL_BWD
  if (rc != AGL_OK) return rc;
  if (splits > 1 && !defer_split_sum((const float*)ws, out_numel, splits, bias, pos_mask, accumulate, relu, nullptr)) {
    hipLaunchKernelGGL(splitk_epilogue, dim3(agl_cdiv(out_numel, 256)), dim3(256), 0, st, (const float*)ws, dx, out_numel, splits,
                       IH * IW, Cin, bias, pos_mask, accumulate, relu, (const float*)nullptr);
    AGL_CHECK_LAUNCH("agl_conv2d_bwd_data(split-K epilogue)");
  }
  return AGL_OK;
}

// Number of reduction splits for bwd-weight and the (BK-aligned) reduction length of each; every split z < splits owns
// a non-empty range [z*per, min(R,(z+1)*per)).  Same quantisation-aware cost model as fwd_splits.
// Weight-gradient GEMMs use a 256x128 (Cout multiple of 256) or 128x256 (Cout <= 128) tile where it fits (2 workgroups
// per CU, 8 accumulators per wave): both operands are long streams there and the larger tile moves 25 % fewer bytes per
// FLOP through LDS (+8-16 % measured); forward and input-gradient passes measured no gain (or a loss) and stay on 128x128.
static int bww_big_tile(int Cout, long Nc) {
  if (Cout >= 256 && Cout % 256 == 0) return 1;
  if (Cout > 64 && Cout <= 128 && (Nc + 255) / 256 * 256 * 100 <= Nc * 112) return 2;   // 128x256 unless padding costs > 12 %
  return 0;
}

static int bww_splits(int Cout, long Nc, long R, long* per_out) {
  const int big = bww_big_tile(Cout, Nc);
  const int bm = big == 1 ? 256 : (Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128));
  const int bn = (Cout <= 32 || big == 2) ? 256 : 128;
  const long tiles = (long)agl_cdiv(Cout, bm) * agl_cdiv(Nc, bn);
  const double slots = big ? 512.0 : 1024.0;
  const double t_k = (double)bm * bn * 2.0 / (big ? 230e9 : 115e9);
  const double out_bytes = (double)Cout * (double)Nc * 4.0;
  long best = 1;
  double best_t = 1e30;
  const long smax = R / 256 > 0 ? (R / 256 > 256 ? 256 : R / 256) : 1;
  for (long s = 1; s <= smax; s = s < 16 ? s + 1 : s + s / 8) {
    const double rounds = ceil((double)tiles * s / slots);
    double t = rounds * ((double)R / s) * t_k + AGL_SPLIT_FIXED * (s > 1);
    if (s > 1) t += out_bytes * (2.0 * s + 1.0) / 4.0e12;
    if (t < best_t * 0.97) { best_t = t; best = s; }
  }
  long s = best;
  long per = (R + s - 1) / s;
  per = (per + BK - 1) / BK * BK;
  s = (R + per - 1) / per;
  if (per_out) *per_out = per;
  return (int)s;
}

struct PosBwwPlan { long dyt, xt, dwt, slabs; int splits; long total() const { return (dyt + xt + dwt + slabs) * 4; } };
static PosBwwPlan pos_bww_plan(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad) {
  PosBwwPlan pl;
  const int Q = OH * OW, KK = ks * ks;
  pl.dyt = (long)N * Cout * Q; pl.xt = (long)N * Cin * H * W; pl.dwt = (long)KK * Cout * Cin;
  const long Kavg = pos_valid_taps(H, W, OH, OW, ks, stride, pad) * N / KK;     // (position, image) pairs per tap
  pl.splits = bww_splits(Cout, (long)Cin * KK, Kavg, nullptr);
  if (pl.splits > 64) pl.splits = 64;
  pl.slabs = pl.splits > 1 ? (long)pl.splits * pl.dwt : 0;
  return pl;
}
static int pos_conv_bww(const float* dy, const float* x, float* dw, void* ws, int N, int Cin, int H, int W, int Cout, int OH, int OW,
                        int ks, int stride, int pad, int accumulate, hipStream_t st, int prec) {
  const int Q = OH * OW, KK = ks * ks;
  const PosBwwPlan pl = pos_bww_plan(N, Cin, H, W, Cout, OH, OW, ks, stride, pad);
  float* dyt = (float*)ws; float* xt = dyt + pl.dyt; float* dwt = xt + pl.xt; float* slabs = dwt + pl.dwt;
  int rc = launch_transpose_in(dy, dyt, N, Cout, Q, 0, st);
  if (rc) return rc;
  rc = launch_transpose_in(x, xt, N, Cin, H * W, 0, st);
  if (rc) return rc;
  rc = AGL_ERR_ARG;
#define AGL_PW(KS_)                                                                                             \
  case KS_: {                                                                                                   \
    PosBwwProb<KS_> p;                                                                                          \
    p.dyt = dyt; p.xt = xt; p.dwt = dwt; p.part = slabs; p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout;  \
    p.OH = OH; p.OW = OW; p.stride = stride; p.pad = pad; p.M = Cout; p.Nc = Cin; p.kbeg = 0; p.kend = 0;        \
    p.a_bytes = (unsigned)(pl.dyt * 4); p.b_bytes = (unsigned)(pl.xt * 4); p.splits = pl.splits; p.slab = pl.dwt; \
    rc = launch_igemm(p, Cout, Cin, KK * pl.splits, st, "agl_conv2d_bwd_weight(position-major)", prec, bww_big_tile(Cout, Cin)); \
  } break;
  switch (ks) { AGL_PW(3) AGL_PW(4) AGL_PW(5) AGL_PW(7) }
#undef AGL_PW
  if (rc != AGL_OK) return rc;
  if (pl.splits > 1) {
    rc = agl_launch_slab_reduce((const float*)slabs, dwt, pl.dwt, pl.splits, 0, st, "agl_conv2d_bwd_weight(position-major reduce)");
    if (rc != AGL_OK) return rc;
  }
  hipLaunchKernelGGL(tap_major_to_w_k, dim3(agl_cdiv(pl.dwt, 256)), dim3(256), 0, st, (const float*)dwt, dw, Cout, Cin, KK, accumulate);
  AGL_CHECK_LAUNCH("agl_conv2d_bwd_weight(position-major weights)");
  return AGL_OK;
}


// Weight gradients of convolutions with <= 4 output channels (decoder c4 / c7) would use 3 of 32 MFMA rows.  With the
// operand roles swapped — dw[co][ci][kh][kw] = dw'[ci][co][ks-1-kh][ks-1-kw], where dw' is the weight gradient of the
// convolution that maps dy (as input, pad ks-1-pad) to x (as output gradient) — the GEMM has M = Cin rows instead.
static bool bww_swapped(int Cin, int Cout, int stride, int up, int in_relu) {
  return Cout <= 4 && Cin >= 32 && stride == 1 && up == 0 && !in_relu;
}

static long bww_ws_core(int N, int Cin, int Cout, int ks, int OH, int OW);
// 1 when agl_conv2d_bwd_data with AGL_CONV_BF16 runs these extents on the stride-1 patch kernel without a reduction split — the
// launch that can read pos_mask as bf16 (AGL_CONV_MASK_BF16).
int agl_conv2d_bwd_data_takes_bf16_mask(int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int ks, int stride, int pad, int flags) {
  const ConvOpts co = conv_opts(flags);
  if (co.prec == 1 && co.patch && stride == 2 && ks == 4 && pad == 1 && Cin > 4 && IH == 2 * OH && IW == 2 * OW && OW >= 4) {
    PConvArgs t{};      // the paired-phase kernel without a reduction split (pconvT_try's conditions for mask_bf16)
    t.N = N; t.Cin = Cout; t.H = OH; t.W = OW; t.Cout = Cin; t.OH = IH; t.OW = IW; t.ks = 4; t.stride = 2; t.pad = 1; t.nsplit = 1;
    t.any_grid = co.any_grid;
    return pconvT_takes_bf16_mask(t) ? 1 : 0;
  }
  if (co.prec != 1 || !co.patch || stride != 1 || IH != OH || IW != OW || Cin <= 4) return 0;
  PConvArgs a{};
  a.N = N; a.Cin = Cout; a.H = OH; a.W = OW; a.Cout = Cin; a.OH = IH; a.OW = IW; a.ks = ks; a.stride = 1; a.pad = ks - 1 - pad;
  a.w_sm = ks * ks; a.w_sc = Cin * ks * ks; a.flip = 1; a.nsplit = 1; a.any_grid = co.any_grid;
  return pconv_plan_splits(a) == 1 ? 1 : 0;
}

// 1 when agl_conv2d_bwd_weight with AGL_CONV_BF16 runs these extents on the matrix-core kernel — the only one that reads a bf16 x
// (AGL_CONV_X_BF16); a producer asks before it writes the tensor in bf16.
int agl_conv2d_bwd_weight_takes_bf16_x(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int flags) {
  const ConvOpts co = conv_opts(flags);
  if (co.prec != 1 || !co.patch) return 0;
  if (bww_swapped(Cin, Cout, stride, 0, 0)) {      // few output channels: the few-channel kernel through the role swap (8-byte pieces of 4 bf16)
    const FewBwwShape f{N, Cout, OH, OW, Cin, H, W, ks, 1, ks - 1 - pad, 0, 0};
    return ((ks & 1) && few_bww_ws_bytes(f) > 0 && W % 4 == 0) ? 1 : 0;
  }
  PBwwArgs a{};
  a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = OH; a.OW = OW; a.ks = ks; a.stride = stride; a.pad = pad; a.nsplit = 1; a.x_bf16 = 1;
  return pbww_ws_bytes(a) > 0 ? 1 : 0;
}

long agl_conv2d_bwd_weight_ws_bytes(int N, int Cin, int Cout, int ks, int OH, int OW) {
  long need = bww_ws_core(N, Cin, Cout, ks, OH, OW);
  PBwwArgs a{};    // bf16 / split kernels: the input extent is not an argument here, so cover the convolution forms the path has
  a.dy = nullptr; a.x = nullptr; a.dw = nullptr; a.N = N; a.Cin = Cin; a.Cout = Cout; a.OH = OH; a.OW = OW; a.ks = ks;
  a.up = 0; a.in_relu = 0; a.accumulate = 0;
  for (int form = 0; form < 2; ++form) {       // stride 1 "same"; stride 2 (4x4 pad 1: H = 2*OH; 3x3 pad 0: H = 2*OH + 1)
    if (form == 0) { a.stride = 1; a.pad = ks / 2; a.H = OH; a.W = OW; }
    else if (ks == 4) { a.stride = 2; a.pad = 1; a.H = 2 * OH; a.W = 2 * OW; }
    else if (ks == 3) { a.stride = 2; a.pad = 0; a.H = 2 * OH + 1; a.W = 2 * OW + 1; }
    else continue;
    for (int ns = 1; ns <= 3; ns += 2) { a.nsplit = ns; need = std::max(need, pbww_ws_bytes(a)); }
  }
  return need;
}
static long bww_ws_core(int N, int Cin, int Cout, int ks, int OH, int OW) {
  if (Cin <= 4) {                                    // few-input-channel kernel (few.hip); the slab count does not depend on the padding
    const FewBwwShape f{N, Cin, OH, OW, Cout, OH, OW, ks, 1, ks / 2, 0, 0};
    const long few = (ks & 1) ? few_bww_ws_bytes(f) : 0;
    if (few) {
      long Nc = (long)Cin * ks * ks, R = (long)N * OH * OW;
      int s = bww_splits(Cout, Nc, R, nullptr);
      const long plain = s > 1 ? (long)s * Cout * Nc * 4 : 0;
      return few > plain ? few : plain;
    }
  }
  long pos_need = 0;
  if (OH <= 8 && OW <= 8 && ks == 5 && pos_ok(kDefaultOpts, N, 64, OH, OW, Cout, ks, 0) && Cin >= 64)   // stride-1 "same" 5x5 on a small map
    pos_need = pos_bww_plan(N, Cin, OH, OW, Cout, OH, OW, ks, 1, ks / 2).total();
  if (pos_need) {
    long Nc1 = (long)Cin * ks * ks, R1 = (long)N * OH * OW;
    int s1 = bww_splits(Cout, Nc1, R1, nullptr);
    const long plain = s1 > 1 ? (long)s1 * Cout * Nc1 * 4 : 0;
    return pos_need > plain ? pos_need : plain;
  }
  if (bww_swapped(Cin, Cout, 1, 0, 0)) {     // (upper bound: the entry point decides with the real stride / flags)
    long inner = 0;                                  // the input extent depends on the padding: cover every legal one
    for (int pad = 0; pad < ks; ++pad) {
      const int H = OH + ks - 1 - 2 * pad, W = OW + ks - 1 - 2 * pad;
      if (H <= 0 || W <= 0) break;
      const long need = bww_ws_core(N, Cout, Cin, ks, H, W);
      if (need > inner) inner = need;
    }
    long Nc0 = (long)Cin * ks * ks, R0 = (long)N * OH * OW;
    int s0 = bww_splits(Cout, Nc0, R0, nullptr);
    const long plain = s0 > 1 ? (long)s0 * Cout * Nc0 * 4 : 0;
    const long swapped = inner + (long)Cin * Cout * ks * ks * 4;
    return swapped > plain ? swapped : plain;
  }
  long Nc = (long)Cin * ks * ks, R = (long)N * OH * OW;
  int s = bww_splits(Cout, Nc, R, nullptr);
  return s > 1 ? (long)s * Cout * Nc * 4 : 0;
}

// dw[Cout,Cin,ks,ks] (+)= sum_{n,oh,ow} dy * im2col(x).  ws: agl_conv2d_bwd_weight_ws_bytes() bytes (may be null if 0).
static int conv2d_bwd_weight_impl(const float* dy, const float* x, float* dw, float* dbias, int dbias_accumulate, int* dbias_done, void* ws,
                                 long ws_bytes, int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int up_log2,
                                 int in_relu, int accumulate, int flags, void* stream, const InFold* fold);
int agl_conv2d_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, int dbias_accumulate, int* dbias_done, void* ws,
                          long ws_bytes, int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int up_log2,
                          int in_relu, int accumulate, int flags, void* stream) {
  return conv2d_bwd_weight_impl(dy, x, dw, dbias, dbias_accumulate, dbias_done, ws, ws_bytes, N, Cin, H, W, Cout, OH, OW, ks, stride, pad,
                                up_log2, in_relu, accumulate, flags, stream, nullptr);
}
// The weight gradient of a convolution whose input was normalised on the fly (agl_conv2d_fwd_fold): x is the RAW tensor, the same
// transform (+ in_relu) is applied while it is staged.  Matrix-core kernel only (agl_conv2d_bwd_weight_fold_ok).
int agl_conv2d_bwd_weight_fold(const float* dy, const float* x, const float* in_mean, const float* in_scale, const float* in_shift,
                               int in_per_n, float* dw, float* dbias, int dbias_accumulate, int* dbias_done, void* ws, long ws_bytes, int N,
                               int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int in_relu, int accumulate,
                               int flags, void* stream) {
  AGL_REQUIRE(in_scale && in_shift, "agl_conv2d_bwd_weight_fold: null table");
  const InFold f{in_mean, in_scale, in_shift, in_per_n};
  return conv2d_bwd_weight_impl(dy, x, dw, dbias, dbias_accumulate, dbias_done, ws, ws_bytes, N, Cin, H, W, Cout, OH, OW, ks, stride, pad, 0,
                                in_relu, accumulate, flags, stream, &f);
}
int agl_conv2d_bwd_weight_fold_ok(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int flags) {
  const ConvOpts co = conv_opts(flags);
  if (!co.patch || !(co.prec == 1 || co.split3) || Cin <= 4 || !(ks == 4 && stride == 2)) return 0;
  PBwwArgs a{};
  a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = OH; a.OW = OW; a.ks = ks; a.stride = stride; a.pad = pad;
  a.nsplit = co.prec == 1 ? 1 : 3;
  return pbww_ws_bytes(a) > 0 ? 1 : 0;
}
int agl_conv2d_bwd_weight_spade_ok(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int flags);
// Weight gradient of a convolution whose input is relu?(SPADE(x)) applied while x is staged (agl_conv2d_fwd_spade's partner).  Few output
// channels (c7): the role swap of agl_conv2d_bwd_weight, the modulated tensor in the few-channel kernel's dy role, rounded to bf16 as
// the stored tensor would be; the bias gradient is left to the caller (*dbias_done stays 0) as there.
int agl_conv2d_bwd_weight_spade(const float* dy, const float* x, const float* mean, const float* rstd, const float* cells, const int* map, int G,
                                float* dw, float* dbias, int dbias_accumulate, int* dbias_done, void* ws, long ws_bytes, int N, int Cin, int H,
                                int W, int Cout, int OH, int OW, int ks, int stride, int pad, int in_relu, int accumulate, int flags,
                                void* stream) {
  AGL_REQUIRE(dy && x && mean && rstd && cells && map && G > 0 && dw, "agl_conv2d_bwd_weight_spade: null pointer");
  AGL_REQUIRE(agl_conv2d_bwd_weight_spade_ok(N, Cin, H, W, Cout, OH, OW, ks, stride, pad, flags),
              "agl_conv2d_bwd_weight_spade: a shape / arithmetic the folded forms do not take (ask agl_conv2d_bwd_weight_spade_ok)");
  const InFold f{mean, rstd, nullptr, 0, cells, map, G};
  if (Cout <= 4) {
    if (dbias_done) *dbias_done = 0;
    const long inner = bww_ws_core(N, Cout, Cin, ks, H, W);
    const long n = (long)Cout * Cin * ks * ks;
    AGL_REQUIRE(ws && ws_bytes >= inner + n * 4, "agl_conv2d_bwd_weight_spade: workspace too small (%ld < %ld)", ws_bytes, inner + n * 4);
    float* tmp = (float*)((char*)ws + inner);
    const FewBwwShape fs{N, Cout, OH, OW, Cin, H, W, ks, 1, ks - 1 - pad, 0, 0};
    int fsplits = 0;
    const int frc = few_bww_try(fs, x, dy, ws, inner, &fsplits, (hipStream_t)stream, "agl_conv2d_bwd_weight_spade(few output channels)", 0, &f, in_relu, 1);
    AGL_REQUIRE(frc == AGL_OK, "agl_conv2d_bwd_weight_spade: the few-channel kernel did not take the shape");
    int rc = agl_launch_slab_reduce((const float*)ws, tmp, n, fsplits, 0, (hipStream_t)stream, "agl_conv2d_bwd_weight_spade(reduce)");
    if (rc != AGL_OK) return rc;
    hipLaunchKernelGGL(flip_transpose_w, dim3(agl_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)tmp, dw, Cout, Cin, ks, accumulate);
    AGL_CHECK_LAUNCH("agl_conv2d_bwd_weight_spade(flip)");
    g_last_pipe = 0;
    return AGL_OK;
  }
  return conv2d_bwd_weight_impl(dy, x, dw, dbias, dbias_accumulate, dbias_done, ws, ws_bytes, N, Cin, H, W, Cout, OH, OW, ks, stride, pad, 0,
                                in_relu, accumulate, flags, stream, &f);
}
int agl_conv2d_bwd_weight_spade_ok(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int flags) {
  const ConvOpts co = conv_opts(flags);
  if (co.prec != 1 || !co.patch || stride != 1 || 2 * pad != ks - 1 || Cin % 16 != 0 || OH != H || OW != W) return 0;
  if (Cout <= 4) {
    const FewBwwShape fs{N, Cout, OH, OW, Cin, H, W, ks, 1, ks - 1 - pad, 0, 0};
    return (W % 4 == 0 && few_bww_ws_bytes(fs) > 0) ? 1 : 0;
  }
  PBwwArgs a{};
  a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = OH; a.OW = OW; a.ks = ks; a.stride = stride; a.pad = pad; a.nsplit = 1;
  return pbww_takes_spade(a) ? 1 : 0;
}
// 1 when agl_conv2d_bwd_weight with AGL_CONV_BF16 | AGL_CONV_DY_BF16 runs these extents on the matrix-core kernel (8-pixel pieces of dy)
int agl_conv2d_bwd_weight_takes_bf16_dy(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int flags) {
  const ConvOpts co = conv_opts(flags);
  if (co.prec != 1 || !co.patch || Cin <= 4 || !(ks == 4 && stride == 2)) return 0;
  if (OW % 8 != 0 && !(OW == 4 && OH == 4)) return 0;
  PBwwArgs a{};
  a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = OH; a.OW = OW; a.ks = ks; a.stride = stride; a.pad = pad; a.nsplit = 1;
  a.dy_bf16 = 1;
  return pbww_ws_bytes(a) > 0 ? 1 : 0;
}
static int conv2d_bwd_weight_impl(const float* dy, const float* x, float* dw, float* dbias, int dbias_accumulate, int* dbias_done, void* ws,
                                 long ws_bytes, int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad, int up_log2,
                                 int in_relu, int accumulate, int flags, void* stream, const InFold* fold) {
  AGL_REQUIRE(dy && x && dw, "agl_conv2d_bwd_weight: null pointer");
  AGL_REQUIRE(!dbias || dbias_done, "agl_conv2d_bwd_weight: dbias needs dbias_done");
  if (dbias_done) *dbias_done = 0;
  g_last_pipe = 0;
  const ConvOpts co = conv_opts(flags);
  AGL_REQUIRE(ks_ok(ks) && (stride == 1 || stride == 2), "agl_conv2d_bwd_weight: unsupported ks=%d stride=%d", ks, stride);
  AGL_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && OH > 0 && OW > 0, "agl_conv2d_bwd_weight: empty extent");
  AGL_REQUIRE((long)N * Cin * H * W < (1L << 30) && (long)N * Cout * OH * OW < (1L << 30), "agl_conv2d_bwd_weight: tensor too large (< 2^30 elements per operand)");
  if (co.x_bf16 && !co.dy_bf16 && !fold && co.prec == 1 && co.patch && bww_swapped(Cin, Cout, stride, up_log2, in_relu)) {      // (a bias gradient is left to the caller: *dbias_done stays 0)
    // few OUTPUT channels (decoder c4 / c7) with a bf16-stored input: the role swap below hands the tensor to the few-channel
    // kernel in the dy role (AGL_CONV_DY_BF16 of the inner call)
    const long inner = bww_ws_core(N, Cout, Cin, ks, H, W);
    const long tmp_bytes = (long)Cin * Cout * ks * ks * 4;
    AGL_REQUIRE(ws && ws_bytes >= inner + tmp_bytes, "agl_conv2d_bwd_weight: workspace too small (%ld < %ld)", ws_bytes, inner + tmp_bytes);
    float* tmp = (float*)((char*)ws + inner);
    const int iflags = (flags & ~(1 << 17)) | (1 << 20);
    int rc = agl_conv2d_bwd_weight(x, dy, tmp, nullptr, 0, nullptr, ws, inner, N, Cout, OH, OW, Cin, H, W, ks, 1, ks - 1 - pad, 0, 0, 0, iflags, stream);
    if (rc != AGL_OK) return rc;
    const long n = (long)Cout * Cin * ks * ks;
    hipLaunchKernelGGL(flip_transpose_w, dim3(agl_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)tmp, dw, Cout, Cin, ks,
                       accumulate);
    AGL_CHECK_LAUNCH("agl_conv2d_bwd_weight(flip, bf16 input)");
    return AGL_OK;
  }
  if (co.dy_bf16 && !co.x_bf16 && !fold && Cin <= 4) {      // (the inner call of the swap above) few-input-channel kernel, dy as bf16
    const FewBwwShape f{N, Cin, H, W, Cout, OH, OW, ks, stride, pad, up_log2, in_relu};
    int fsplits = 0;
    const int frc = few_bww_try(f, dy, x, ws, ws_bytes, &fsplits, (hipStream_t)stream, "agl_conv2d_bwd_weight(few input channels, bf16 dy)", 1);
    AGL_REQUIRE(frc == AGL_OK, "agl_conv2d_bwd_weight: AGL_CONV_DY_BF16 with <= 4 input channels on a shape the few-channel kernel does not take");
    const long n = (long)Cout * Cin * ks * ks;
    return agl_launch_slab_reduce((const float*)ws, dw, n, fsplits, accumulate, (hipStream_t)stream, "agl_conv2d_bwd_weight(few input channels: reduce)");
  }
  if (co.x_bf16 || co.dy_bf16 || fold) {
    AGL_REQUIRE(co.patch && (co.prec == 1 || (co.split3 && !co.x_bf16 && !co.dy_bf16)),
                "agl_conv2d_bwd_weight: bf16-stored operands need AGL_CONV_BF16, a folded input transform the matrix-core kernel");
    PBwwArgs a{};
    a.dy = dy; a.x = x; a.dw = dw; a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = OH; a.OW = OW; a.ks = ks;
    a.stride = stride; a.pad = pad; a.up = up_log2; a.in_relu = in_relu; a.accumulate = accumulate; a.nsplit = co.prec == 1 ? 1 : 3;
    a.x_bf16 = co.x_bf16; a.dy_bf16 = co.dy_bf16; a.x_blk = co.x_blk;
    if (fold) a.fold = *fold;
    a.dbias = dbias; a.dbias_accumulate = dbias_accumulate; a.dbias_done = dbias_done;
    const int prc = pbww_try(a, ws, ws_bytes, (hipStream_t)stream, "agl_conv2d_bwd_weight(pbww, bf16 input)");
    AGL_REQUIRE(prc >= 0, "agl_conv2d_bwd_weight: bf16-stored operand / folded transform on a shape the matrix-core kernel does not take "
                          "(ask agl_conv2d_bwd_weight_takes_bf16_x / _takes_bf16_dy / _fold_ok)");
    g_last_pipe = a.nsplit;
    return prc;
  }
  if (ks == 1 && H == 1 && W == 1 && OH == 1 && OW == 1 && pad == 0 && up_log2 == 0 && co.patch)      // nn.Linear (few.hip)
    return linear_bww_launch(dy, x, dw, N, Cin, Cout, in_relu, accumulate, co.prec == 1, (hipStream_t)stream, "agl_conv2d_bwd_weight(linear)");
  if (Cin <= 4) {      // RGB-side layers: rows = output channels, columns = (input channel, tap), exact fp32 (few.hip)
    const FewBwwShape f{N, Cin, H, W, Cout, OH, OW, ks, stride, pad, up_log2, in_relu};
    int fsplits = 0;
    const int frc = few_bww_try(f, dy, x, ws, ws_bytes, &fsplits, (hipStream_t)stream, "agl_conv2d_bwd_weight(few input channels)");
    if (frc == AGL_OK) {
      const long n = (long)Cout * Cin * ks * ks;
      return agl_launch_slab_reduce((const float*)ws, dw, n, fsplits, accumulate, (hipStream_t)stream, "agl_conv2d_bwd_weight(few input channels: reduce)");
    }
    if (frc > 0) return frc;
  }
  if (co.patch && (co.prec == 1 || co.split3)) {
    PBwwArgs a{};
    a.dy = dy; a.x = x; a.dw = dw; a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = OH; a.OW = OW; a.ks = ks;
    a.stride = stride; a.pad = pad; a.up = up_log2; a.in_relu = in_relu; a.accumulate = accumulate; a.nsplit = co.prec == 1 ? 1 : 3;
    a.dbias = dbias; a.dbias_accumulate = dbias_accumulate; a.dbias_done = dbias_done;
    const int prc = pbww_try(a, ws, ws_bytes, (hipStream_t)stream, "agl_conv2d_bwd_weight(pbww)");
    if (prc >= 0) { g_last_pipe = a.nsplit; return prc; }
  }
  g_last_pipe = co.prec;
  if (stride == 1 && H == OH && W == OW && up_log2 == 0 && !in_relu && ks == 5 && pad == ks / 2 && Cin >= 64 &&
      pos_ok(co, N, 64, H, W, Cout, ks, 0)) {
    const PosBwwPlan pl = pos_bww_plan(N, Cin, H, W, Cout, OH, OW, ks, 1, pad);
    if (ws && ws_bytes >= pl.total())
      return pos_conv_bww(dy, x, dw, ws, N, Cin, H, W, Cout, OH, OW, ks, 1, pad, accumulate, (hipStream_t)stream, co.prec);
  }
  if (bww_swapped(Cin, Cout, stride, up_log2, in_relu)) {
    const long inner = bww_ws_core(N, Cout, Cin, ks, H, W);
    const long tmp_bytes = (long)Cin * Cout * ks * ks * 4;
    if (!ws || ws_bytes < inner + tmp_bytes) {
      agl_set_error("agl_conv2d_bwd_weight: workspace too small (%ld < %ld)", ws_bytes, inner + tmp_bytes);
      return AGL_ERR_WORKSPACE;
    }
    float* tmp = (float*)((char*)ws + inner);
    int rc = agl_conv2d_bwd_weight(x, dy, tmp, nullptr, 0, nullptr, ws, inner, N, Cout, OH, OW, Cin, H, W, ks, 1, ks - 1 - pad, 0, 0, 0, flags, stream);
    if (rc != AGL_OK) return rc;
    const long n = (long)Cout * Cin * ks * ks;
    hipLaunchKernelGGL(flip_transpose_w, dim3(agl_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)tmp, dw, Cout, Cin, ks,
                       accumulate);
    AGL_CHECK_LAUNCH("agl_conv2d_bwd_weight(flip)");
    return AGL_OK;
  }
  const long Nc = (long)Cin * ks * ks, R = (long)N * OH * OW;
  long per = 0;
  const int splits = bww_splits(Cout, Nc, R, &per);
  const long need = splits > 1 ? (long)splits * Cout * Nc * 4 : 0;
  if (need > ws_bytes || (need > 0 && !ws)) {
    agl_set_error("agl_conv2d_bwd_weight: workspace too small (%ld < %ld)", ws_bytes, need);
    return AGL_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const bool direct = splits == 1 && !accumulate;
  float* target = direct ? dw : (float*)ws;
  if (!direct && splits == 1) {  // accumulate with a single split: still go through a slab
    AGL_REQUIRE(ws && ws_bytes >= Cout * Nc * 4, "agl_conv2d_bwd_weight: accumulate needs a slab of Cout*Cin*ks*ks floats");
  }
  int rc = AGL_ERR_ARG;
#define AGL_BWW(KS_)                                                                                              \
  case KS_: {                                                                                                     \
    BwdWeightProb<KS_> p;                                                                                         \
    p.dy = dy; p.x = x; p.out = target; p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.OH = OH;        \
    p.OW = OW; p.stride = stride; p.pad = pad; p.up = up_log2; p.in_relu = in_relu; p.M = Cout; p.Nc = (int)Nc; p.HW = H * W;         \
    p.OHW = OH * OW; p.R = (int)R; p.per_split = (int)per; p.slab = (long)Cout * Nc; p.kbeg = 0; p.kend = 0;    \
    p.a_bytes = (unsigned)((long)N * Cout * OH * OW * 4); p.b_bytes = (unsigned)((long)N * Cin * H * W * 4);    \
    rc = launch_igemm(p, Cout, Nc, splits, st, "agl_conv2d_bwd_weight", co.prec, bww_big_tile(Cout, Nc));                \
  } break;
  switch (ks) { AGL_BWW(1) AGL_BWW(3) AGL_BWW(4) AGL_BWW(5) AGL_BWW(7) }
#undef AGL_BWW
  if (rc != AGL_OK) return rc;
  if (!direct) {
    long n = (long)Cout * Nc;
    rc = agl_launch_slab_reduce((const float*)ws, dw, n, splits, accumulate, st, "agl_conv2d_bwd_weight(reduce)");
    if (rc != AGL_OK) return rc;
  }
  return AGL_OK;
}

int agl_conv2d_last_pipe(void) { return g_last_pipe; }
int agl_conv2d_split_products(void) { return pconv_split_products(); }

// 1 when agl_conv2d_fwd with AGL_CONV_BF16 | AGL_CONV_Y_BF16 would write these extents as bf16: the few-input-channel stream kernel
// (Cin <= 4, 1x1 / 3x3 "same") or the matrix-core patch kernel without a reduction split (the two kernels that have that store).
int agl_conv2d_fwd_writes_bf16_y(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2, int relu, int accumulate,
                                 int flags) {
  const ConvOpts co = conv_opts(flags);
  if (co.prec != 1 || !co.patch || Cout <= 4 || accumulate) return 0;
  const int Hl = H << up_log2, Wl = W << up_log2;
  const int OH = (Hl + 2 * pad - ks) / stride + 1, OW = (Wl + 2 * pad - ks) / stride + 1;
  if (Cin <= 4) {      // the acceptance conditions of few_cin_fwd_try (few.hip), alignment aside (torch allocations are 16-byte aligned)
    return (stride == 1 && up_log2 == 0 && OH == H && OW == W && Cout >= 16 && (ks == 1 || ks == 3) && W % 4 == 0 &&
            ((long)Cout * Cin * ks * ks + Cout) * 4 <= 48 * 1024) ? 1 : 0;
  }
  (void)relu;
  if (stride != 1 || !(ks == 1 || ks == 3)) return 0;      // (the bf16 store is compiled into the 3x3 stride-1 and 1x1 instantiations)
  PConvArgs a{};
  a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = OH; a.OW = OW; a.ks = ks; a.stride = stride; a.pad = pad; a.up = up_log2;
  a.nsplit = 1; a.any_grid = co.any_grid;
  return pconv_plan_splits(a) == 1 && OW % 4 == 0 ? 1 : 0;
}
// 1 when agl_conv2d_fwd with AGL_CONV_BF16 | AGL_CONV_X_BF16 takes these extents: the patch kernel, or (<= 4 output channels, 7x7
// "same") the vertical + diagonal form
int agl_conv2d_fwd_takes_bf16_x(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int flags) {
  const ConvOpts co = conv_opts(flags);
  if (co.prec != 1 || !co.patch) return 0;
  if (Cout <= 4)
    return (stride == 1 && 2 * pad == ks - 1 && pconv_vert_ws_bytes(N, Cin, H, W, Cout, ks, 1) > 0) ? 1 : 0;
  return agl_conv2d_fwd_packed_bytes(N, Cin, H, W, Cout, ks, stride, pad, 0, flags) > 0 ? 1 : 0;
}
// 1 when agl_conv2d_fwd (/ _addend / _shortcut) takes the blocked operands the flags name (AGL_CONV_X_BLOCKED and / or AGL_CONV_Y_BLOCKED, with
// AGL_CONV_BF16 and the matching _BF16 operand flags) for these extents
int agl_conv2d_fwd_takes_blocked(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int flags) {
  const ConvOpts co = conv_opts(flags);
  if (co.prec != 1 || !co.patch || !(co.x_blk || co.y_blk)) return 0;
  PConvArgs a{};
  a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = (H + 2 * pad - ks) / stride + 1; a.OW = (W + 2 * pad - ks) / stride + 1;
  a.ks = ks; a.stride = stride; a.pad = pad; a.up = 0;
  a.nsplit = 1; a.any_grid = co.any_grid; a.x_bf16 = co.x_bf16; a.y_bf16 = co.y_bf16; a.x_blk = co.x_blk; a.y_blk = co.y_blk;
  return pconv_takes_blocked(a) ? 1 : 0;
}
// 1 when agl_conv2d_bwd_data with AGL_CONV_BF16 | AGL_CONV_X_BF16 (dy stored as bf16) runs these extents on a matrix-core kernel
int agl_conv2d_bwd_data_takes_bf16_dy(int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int ks, int stride, int pad, int flags) {
  const ConvOpts co = conv_opts(flags);
  if (co.prec != 1 || !co.patch || Cin <= 4) return 0;
  if (stride == 2 && (IH != 2 * OH || IW != 2 * OW)) return 0;      // (the odd-sized form's edge kernel reads an fp32 dy)
  return agl_conv2d_bwd_data_packed_bytes(N, Cin, IH, IW, Cout, OH, OW, ks, stride, pad, flags) > 0 ? 1 : 0;
}

// ---- pre-packed weights (include/agl.h): bytes of the packed form when the call would run on the LDS-patch kernel, else 0
long agl_conv2d_fwd_packed_bytes(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2, int flags) {
  const ConvOpts co = conv_opts(flags);
  if (!(co.patch && (co.prec == 1 || co.split3)) || !ks_ok(ks) || !(stride == 1 || stride == 2)) return 0;
  const int Hl = H << up_log2, Wl = W << up_log2;
  const int OH = (Hl + 2 * pad - ks) / stride + 1, OW = (Wl + 2 * pad - ks) / stride + 1;
  if (Cout <= 4 || OH <= 0 || OW <= 0) return 0;
  PConvArgs a{};
  a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = OH; a.OW = OW; a.ks = ks; a.stride = stride; a.pad = pad; a.up = up_log2;
  a.nsplit = co.prec == 1 ? 1 : 3; a.any_grid = co.any_grid; a.w8 = co.w8; a.prio = co.prio; a.ablate = co.ablate;
  return pconv_eligible(a) ? pconv_ws_bytes(Cin, Cout, ks, a.nsplit) : 0;
}
long agl_conv2d_bwd_data_packed_bytes(int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int ks, int stride, int pad, int flags) {
  const ConvOpts co = conv_opts(flags);
  if (!(co.patch && (co.prec == 1 || co.split3)) || !ks_ok(ks) || Cin <= 4) return 0;
  PConvArgs a{};
  a.N = N; a.Cin = Cout; a.H = OH; a.W = OW; a.Cout = Cin; a.OH = IH; a.OW = IW; a.ks = ks; a.up = 0;
  a.nsplit = co.prec == 1 ? 1 : 3; a.any_grid = co.any_grid; a.w8 = co.w8; a.prio = co.prio; a.ablate = co.ablate;
  if (stride == 1 && IH == OH && IW == OW) {
    a.stride = 1; a.pad = ks - 1 - pad;
    return pconv_eligible(a) ? pconv_ws_bytes(Cout, Cin, ks, a.nsplit) : 0;
  }
  if (stride == 2 && ks == 4) {
    a.stride = 2; a.pad = pad;
    return pconvT_eligible(a) ? pconvT_ws_bytes(Cout, Cin, a.nsplit) : 0;
  }
  return 0;
}
// pass 0: for agl_conv2d_fwd; 1: for agl_conv2d_bwd_data.  w is [Cout][Cin][ks][ks]; `packed` must hold the bytes the matching
// *_packed_bytes query returned for the calls it will be used with (the packed form depends on Cin, Cout, ks, stride and flags only).
int agl_conv2d_pack_weights(const float* w, void* packed, long packed_bytes, int pass, int Cin, int Cout, int ks, int stride, int flags,
                            void* stream) {
  AGL_REQUIRE(w && packed, "agl_conv2d_pack_weights: null pointer");
  const ConvOpts co = conv_opts(flags);
  AGL_REQUIRE(co.prec == 1 || co.split3, "agl_conv2d_pack_weights: flags select no matrix-core arithmetic (AGL_CONV_BF16 / AGL_CONV_SPLIT3)");
  const int ns = co.prec == 1 ? 1 : 3;
  hipStream_t st = (hipStream_t)stream;
  if (pass == 0) {
    const long need = pconv_ws_bytes(Cin, Cout, ks, ns);
    AGL_REQUIRE(need > 0 && packed_bytes >= need, "agl_conv2d_pack_weights: buffer too small or shape not packable (%ld < %ld)", packed_bytes, need);
    return pconv_pack(w, packed, Cout, Cin, ks, Cin * ks * ks, ks * ks, 0, ns, 0, st, "agl_conv2d_pack_weights(fwd)");
  }
  AGL_REQUIRE(pass == 1, "agl_conv2d_pack_weights: pass must be 0 (forward) or 1 (input gradient)");
  if (stride == 2) {
    AGL_REQUIRE(ks == 4, "agl_conv2d_pack_weights: the stride-2 input gradient is packed for 4x4 kernels only");
    const long need = pconvT_ws_bytes(Cout, Cin, ns);
    AGL_REQUIRE(need > 0 && packed_bytes >= need, "agl_conv2d_pack_weights: buffer too small or shape not packable (%ld < %ld)", packed_bytes, need);
    return pconv_pack(w, packed, Cin, Cout, 4, 16, Cin * 16, 0, ns, 1, st, "agl_conv2d_pack_weights(bwd_data, stride 2)");
  }
  const long need = pconv_ws_bytes(Cout, Cin, ks, ns);
  AGL_REQUIRE(need > 0 && packed_bytes >= need, "agl_conv2d_pack_weights: buffer too small or shape not packable (%ld < %ld)", packed_bytes, need);
  return pconv_pack(w, packed, Cin, Cout, ks, ks * ks, Cin * ks * ks, 1, ns, 0, st, "agl_conv2d_pack_weights(bwd_data)");
}

// The pack of agl_conv2d_pack_weights as one row of a descriptor table (host memory, AGL_PACK_DESC_WORDS 64-bit words) ...
int agl_conv2d_pack_desc(const float* w, void* packed, long packed_bytes, int pass, int Cin, int Cout, int ks, int stride, int flags,
                         long long* row) {
  AGL_REQUIRE(w && packed && row, "agl_conv2d_pack_desc: null pointer");
  const ConvOpts co = conv_opts(flags);
  AGL_REQUIRE(co.prec == 1 || co.split3, "agl_conv2d_pack_desc: flags select no matrix-core arithmetic (AGL_CONV_BF16 / AGL_CONV_SPLIT3)");
  const int ns = co.prec == 1 ? 1 : 3;
  if (pass == 0) {
    const long need = pconv_ws_bytes(Cin, Cout, ks, ns);
    AGL_REQUIRE(need > 0 && packed_bytes >= need, "agl_conv2d_pack_desc: buffer too small or shape not packable (%ld < %ld)", packed_bytes, need);
    pconv_pack_desc(w, packed, Cout, Cin, ks, Cin * ks * ks, ks * ks, 0, ns, 0, row);
    return AGL_OK;
  }
  AGL_REQUIRE(pass == 1, "agl_conv2d_pack_desc: pass must be 0 (forward) or 1 (input gradient)");
  if (stride == 2) {
    AGL_REQUIRE(ks == 4, "agl_conv2d_pack_desc: the stride-2 input gradient is packed for 4x4 kernels only");
    const long need = pconvT_ws_bytes(Cout, Cin, ns);
    AGL_REQUIRE(need > 0 && packed_bytes >= need, "agl_conv2d_pack_desc: buffer too small or shape not packable (%ld < %ld)", packed_bytes, need);
    pconv_pack_desc(w, packed, Cin, Cout, 4, 16, Cin * 16, 0, ns, 1, row);
    return AGL_OK;
  }
  const long need = pconv_ws_bytes(Cout, Cin, ks, ns);
  AGL_REQUIRE(need > 0 && packed_bytes >= need, "agl_conv2d_pack_desc: buffer too small or shape not packable (%ld < %ld)", packed_bytes, need);
  pconv_pack_desc(w, packed, Cin, Cout, ks, ks * ks, Cin * ks * ks, 1, ns, 0, row);
  return AGL_OK;
}
// ... and the launch that re-packs a whole table: rows_dev = n rows in device memory whose word 12 holds the row's first block
// (running sum of word 13), total_blocks = the sum.  One launch after the optimiser step instead of one per weight and form.
int agl_conv2d_pack_many(const void* rows_dev, int n, long total_blocks, void* stream) {
  AGL_REQUIRE(rows_dev && n > 0 && total_blocks > 0 && total_blocks < (1L << 31), "agl_conv2d_pack_many: bad argument");
  return pconv_pack_many(rows_dev, n, total_blocks, (hipStream_t)stream, "agl_conv2d_pack_many");
}

// Executed FLOPs (2*MAC) of the launches one call of the entry points above issues for these extents and flags, assuming
// the workspace the *_ws_bytes functions ask for is provided: dense 2*N*OH*OW*Cout*Cin*ks^2, minus the padded taps the
// position-major path never visits.  (bench.py's roofline leg divides the sum of these by the measured launch time.)
double agl_conv2d_fwd_flops(int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int up_log2, int flags) {
  const ConvOpts co = conv_opts(flags);
  const int Hl = H << up_log2, Wl = W << up_log2;
  const int OH = (Hl + 2 * pad - ks) / stride + 1, OW = (Wl + 2 * pad - ks) / stride + 1;
  const bool small = Cout <= 4 && stride == 1 && up_log2 == 0 && OH * OW >= 64 && (long)H * W * 8 < (1L << 18);
  bool mc = false;
  if (co.patch && (co.prec == 1 || co.split3)) {
    PConvArgs a{};
    a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.OH = OH; a.OW = OW; a.ks = ks; a.stride = stride; a.pad = pad; a.up = up_log2;
    a.nsplit = co.prec == 1 ? 1 : 3; a.any_grid = co.any_grid; a.w8 = co.w8; a.prio = co.prio; a.ablate = co.ablate;
    mc = pconv_eligible(a);
  }
  if (!small && !mc && pos_ok(co, N, Cin, H, W, Cout, ks, up_log2))
    return 2.0 * N * Cout * Cin * (double)pos_valid_taps(H, W, OH, OW, ks, stride, pad);
  return 2.0 * N * OH * OW * (double)Cout * Cin * ks * ks;
}
double agl_conv2d_bwd_data_flops(int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int ks, int stride, int pad, int flags) {
  const ConvOpts co = conv_opts(flags);
  const bool small = Cin <= 4 && stride == 1 && IH * IW >= 64 && (long)OH * OW * 8 < (1L << 18);
  bool mc = false;
  if (co.patch && (co.prec == 1 || co.split3) && stride == 1 && IH == OH && IW == OW) {
    PConvArgs a{};
    a.N = N; a.Cin = Cout; a.H = OH; a.W = OW; a.Cout = Cin; a.OH = IH; a.OW = IW; a.ks = ks; a.stride = 1; a.pad = ks - 1 - pad; a.up = 0;
    a.nsplit = co.prec == 1 ? 1 : 3; a.any_grid = co.any_grid; a.w8 = co.w8; a.prio = co.prio; a.ablate = co.ablate;
    mc = pconv_eligible(a);
  }
  if (!small && !mc && stride == 1 && IH == OH && IW == OW && pos_ok(co, N, Cout, OH, OW, Cin, ks, 0))
    return 2.0 * N * Cout * Cin * (double)pos_valid_taps(OH, OW, IH, IW, ks, 1, ks - 1 - pad);
  return 2.0 * N * OH * OW * (double)Cout * Cin * ks * ks;
}
double agl_conv2d_bwd_weight_flops(int N, int Cin, int H, int W, int Cout, int OH, int OW, int ks, int stride, int pad,
                                   int up_log2, int in_relu, int flags) {
  const ConvOpts co = conv_opts(flags);
  const bool mc = co.patch && (co.prec == 1 || co.split3) && Cin % 16 == 0 && Cout >= 32;     // pbww takes the call (dense taps)
  if (!mc && stride == 1 && H == OH && W == OW && up_log2 == 0 && !in_relu && ks == 5 && pad == ks / 2 && Cin >= 64 &&
      pos_ok(co, N, 64, H, W, Cout, ks, 0))
    return 2.0 * N * Cout * Cin * (double)pos_valid_taps(H, W, OH, OW, ks, 1, pad);
  return 2.0 * N * OH * OW * (double)Cout * Cin * ks * ks;
}

}  // extern "C"
