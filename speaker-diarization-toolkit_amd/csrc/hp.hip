// hp.hip - the PRECISE mode of the embedding forward ("precision" 1): every tensor that the default mode rounds to bf16 is carried as an
// fp16 hi + lo PAIR (22 significand bits), every MFMA product runs as three fp16 MFMAs
//        x . w  ~=  x_hi . w_hi  +  x_lo . w_hi  +  x_hi . w_lo            (the lo . lo term is 2^-22 of the product: dropped)
// with fp32 accumulation.  Why: north_star asks for cosine scores within 1e-5 of the fp32 model; the default mode's bf16 operands are
// 4e-3 away, and the committed error budget (profiles/r03_error_budget.md, tools/error_budget.py) shows that nothing short of ~18
// significand bits at EVERY rounding site gets under 1e-5 (bf16 hi+lo pairs = 16 bits: 2.3e-5; fp16 pairs = 22 bits: 1.5e-7).
// fp16 pairs cost 3 MFMAs per product at the bf16 rate; the exact-fp32 matrix pipe would cost 16.
//
// Storage ("planes"): a [rows, C] activation is [rows, ld] fp16 with the hi values in columns [0, C) and the lo values lo_off columns
// to the right (4 bytes per element, like fp32): x = float(hi) + float(lo) / 2^11 (hp.hpp).  Weights: a 256-byte header (float
// 1 / 2^s) followed by [2][N][K] fp16 planes (hi, lo unscaled) of 2^s * W, s chosen per layer so that max |2^s W| is in [2^12, 2^13]
// (weights_pack.hp_weight_planes): the lo plane of every weight that matters stays in the normal fp16 range; undone exactly in the epilogue.
// Range: |x| <= 65504 (fp16), saturating - activations behind BatchNorm / log-mel features are O(10).
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "ecapa_layout.h"
#include "hp.hpp"

namespace {

using namespace sdk_hp;

// ================================================================================================ GEMM
// conv_gemm in split-fp16: same operator as sdk_conv_gemm (dilated conv over frames, segment-local reflect, tap-major K), 128 x 128 x 32
// tile, 4 waves (2 x 2, 64 x 64 each as 4 x 4 v_mfma_f32_16x16x32_f16 x 3), register-staged global loads one K-step ahead.
// LDS: four operand planes [128 rows][32 k] fp16 = 64-byte rows; the 16-byte chunk c of row r sits at position c ^ g((r >> 2) & 3),
// g(q) = (-q) & 3: conflict-free for the ds_read_b128 fragment reads (lane groups of MI355X_MICROARCH 'LDS': every group holds four
// lanes per row residue mod 4, which the permutation sends to four different chunk slots) and for the ds_write_b128 fill.
// Epilogue: the fp32 tile goes through LDS once ([128][132] floats, overlaying the operand planes) and leaves as whole 16-byte pieces of
// each requested output (hi / lo planes, fp32, residual-sum planes).
constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int PLANE = BM * BK * 2;                // 8 KiB
constexpr int CT_LD = BN + 4;                     // floats per staged row
constexpr int LDS_BYTES = BM * CT_LD * 4;         // 67584 >= 4 planes
static_assert(LDS_BYTES >= 4 * PLANE, "epilogue tile overlays the operand planes");

struct HpParams {
  const uint16_t* A; int64_t lda, a_lo;
  const uint16_t* W;                 // [2][N][Ktot] planes (behind the slot's header)
  const float* winv;                 // the header: 1 / 2^s
  uint16_t* C; int64_t ldc, c_lo;
  float* C32; int64_t ldc32;
  const float* bias; const float* scale; const float* shift;
  const float* ubias; int64_t ldub;
  const uint16_t* X2; int64_t ldx2, x2_lo;
  uint16_t* S; int64_t lds, s_lo;
  int M, N, Cin, taps, dil, T;
  uint32_t flags;
  int half_tail;                     // 256^2 kernel: compute a mostly idle last round as half tiles (default 1; hp_gemm_variant 2 = off)
};

__device__ __forceinline__ int swz(int row) { return (-(row >> 2)) & 3; }

__global__ __launch_bounds__(NT, 2) void conv_gemm_hp_kernel(HpParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sAh = smem;
  char* sAl = smem + PLANE;
  char* sWh = smem + 2 * PLANE;
  char* sWl = smem + 3 * PLANE;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int nbn = p.N / BN, nbm = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, nbn * nbm);
  const int bn = tile % nbn, bm = tile / nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  // staging: thread -> rows (tid >> 2) and + 64, 16-byte chunk (tid & 3) of the 32-wide K-step
  const int ch = tid & 3, r0 = tid >> 2;
  int segbase[2], tloc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + r0 + 64 * i;
    m = m < p.M ? m : p.M - 1;
    if (p.taps > 1) {
      const int b = m / p.T;
      segbase[i] = b * p.T;
      tloc[i] = m - b * p.T;
    } else {
      segbase[i] = m;
      tloc[i] = 0;
    }
  }
  const int Ktot = p.taps * p.Cin;
  const int64_t wplane = (int64_t)p.N * Ktot;
  const uint16_t* wrow[2];
  uint32_t lds_w[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = r0 + 64 * i;
    wrow[i] = p.W + (int64_t)(n0 + row) * Ktot + ch * 8;
    lds_w[i] = row * 64 + ((ch ^ swz(row)) << 4);
  }
  const int ksteps_per_tap = p.Cin / BK;
  const int nk = p.taps * ksteps_per_tap;
  const int half = p.taps >> 1;

  u32x4 rah[2], ral[2], rwh[2], rwl[2];
  auto gload = [&](int s) {
    const int j = s / ksteps_per_tap;
    const int kc = (s - j * ksteps_per_tap) * BK;
    const int off = (j - half) * p.dil;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int src = p.taps > 1 ? segbase[i] + reflect_idx(tloc[i] + off, p.T) : segbase[i];
      const uint16_t* ap = p.A + (int64_t)src * p.lda + kc + ch * 8;
      rah[i] = *reinterpret_cast<const u32x4*>(ap);
      ral[i] = *reinterpret_cast<const u32x4*>(ap + p.a_lo);
      const uint16_t* wp = wrow[i] + j * p.Cin + kc;
      rwh[i] = *reinterpret_cast<const u32x4*>(wp);
      rwl[i] = *reinterpret_cast<const u32x4*>(wp + wplane);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t coff = (uint32_t)((fq ^ swz(fr)) << 4);       // (row >> 2) & 3 == (fr >> 2) & 3: tile / wave / sub-tile offsets are multiples of 16
  uint32_t a_off[4], b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_off[i] = (wm * 64 + i * 16 + fr) * 64 + coff;
    b_off[i] = (wn * 64 + i * 16 + fr) * 64 + coff;
  }

  gload(0);
  for (int s = 0; s < nk; ++s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(sAh + lds_w[i]) = rah[i];
      *reinterpret_cast<u32x4*>(sAl + lds_w[i]) = ral[i];
      *reinterpret_cast<u32x4*>(sWh + lds_w[i]) = rwh[i];
      *reinterpret_cast<u32x4*>(sWl + lds_w[i]) = rwl[i];
    }
    __syncthreads();
    if (s + 1 < nk) gload(s + 1);
    f16x8 ah[4], al[4], wh[4], wl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const f16x8*>(sAh + a_off[i]);
      al[i] = *reinterpret_cast<const f16x8*>(sAl + a_off[i]);
      wh[i] = *reinterpret_cast<const f16x8*>(sWh + b_off[i]);
      wl[i] = *reinterpret_cast<const f16x8*>(sWl + b_off[i]);
    }
    // the activation lo plane carries 2^11 * lo: it meets W_hi * 2^-11 (exact: a power of two; v_pk_mul_f16 on the fragment)
    f16x8 whs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) whs[i] = wh[i] * (_Float16)(1.0f / HP_LOSCALE);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mi], whs[ni], acc[mi][ni], 0, 0, 0);     // small terms first
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], wl[ni], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mi], wh[ni], acc[mi][ni], 0, 0, 0);
      }
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue (fp32 throughout)
  float* ct = reinterpret_cast<float*>(smem);
  const float winv = *p.winv;
  float cb[4], cs[4], csh[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int col = n0 + wn * 64 + ni * 16 + fr;
    cb[ni] = p.bias ? p.bias[col] : 0.f;
    cs[ni] = p.scale ? p.scale[col] : 1.f;
    csh[ni] = p.shift ? p.shift[col] : 0.f;
  }
  const bool relu = p.flags & SDK_GEMM_RELU, tnh = p.flags & SDK_GEMM_TANH;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wm * 64 + mi * 16 + fq * 4 + r;
      const int m = m0 + row;
      const float* ub = nullptr;
      if (p.ubias) {
        const int mm = m < p.M ? m : p.M - 1;
        ub = p.ubias + (int64_t)(mm / p.T) * p.ldub;
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int lc = wn * 64 + ni * 16 + fr;
        float v = acc[mi][ni][r] * winv + cb[ni];
        if (ub) v += ub[n0 + lc];
        if (relu) v = fmaxf(v, 0.f);
        v = v * cs[ni] + csh[ni];
        if (tnh) v = tanhf(v);
        ct[row * CT_LD + lc] = v;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int id = tid + NT * i;
    const int row = id >> 4, cc = id & 15;          // 16 chunks of 8 columns per row
    const int m = m0 + row;
    if (m < p.M) {
      float v[8];
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(ct + row * CT_LD + cc * 8);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(ct + row * CT_LD + cc * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = v0[e]; v[4 + e] = v1[e]; }
      const int col = n0 + cc * 8;
      if (p.C32) {
        *reinterpret_cast<f32x4*>(p.C32 + (int64_t)m * p.ldc32 + col) = v0;
        *reinterpret_cast<f32x4*>(p.C32 + (int64_t)m * p.ldc32 + col + 4) = v1;
      }
      if (p.C) store8(p.C + (int64_t)m * p.ldc + col, p.c_lo, v);
      if (p.S) {
        float x[8];
        load8(p.X2 + (int64_t)m * p.ldx2 + col, p.x2_lo, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += v[e];
        store8(p.S + (int64_t)m * p.lds + col, p.s_lo, x);
      }
    }
  }
}

// ================================================================================================ the same GEMM as a 256 x 256 LDS-DMA tile
// The big layers of the precise forward (blk0, the six 1024^2 TDNN layers, the 3072^2 MFA layer: 86 % of its flops) in the structure of
// conv_gemm256_kernel (conv_gemm.hip): 8 waves (2 x 4, 128 x 64 each), persistent workgroups in the unit tile order, operands by LDS-DMA
// into two 64-KiB stages, counted waits, one raw barrier per K-step placed before the last MFMA sub-phase, epilogue in registers.
// What differs: a K-step is 32 REAL k; an LDS row (128 B) holds [hi 32 k | lo 32 k] of one tile row, so the 16-byte chunks 0-3 / 4-7 of a
// row come from the hi / lo plane (per-lane DMA source address) and the fragment read at chunk offset c0 / c1 is the hi / lo fragment;
// six MFMA sub-phases of 32 per K-step:  Ah.Wh (both row halves), Al.(Wh 2^-11) (Wh is scaled IN PLACE once its last use has issued),
// Ah.Wl (the hi fragments are read from LDS a second time: cheaper than 32 more live registers);  the epilogue keeps the fp32 results in
// the accumulators and sends first the hi plane, then the lo plane through the bf16-sized tile image.  Plain planes output only
// (bias / ReLU / BN affine); fp32 / residual-sum / per-segment-bias / tanh outputs stay with the 128^2 kernel above.
constexpr int BM2 = 256, BN2 = 256, NT2 = 512;
constexpr int HSTAGE = (BM2 + BN2) * 128;            // 65536
constexpr int HLDS = 2 * HSTAGE;                     // 131072 = one fp16 plane image of a finished tile
constexpr int HLDS_TOTAL = HLDS + 3 * BN2 * 4;
static_assert(BM2 * BN2 * 2 <= HLDS, "a plane image of the tile must fit in the two pipeline stages");

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

template <bool TAPS>
__global__ __launch_bounds__(NT2, 2) void conv_gemm_hp256_kernel(HpParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int nbn = p.N / BN2, nbm = (p.M + BM2 - 1) / BM2, ntiles = nbn * nbm;
  const int Ktot = p.taps * p.Cin;
  const int ksteps_per_tap = p.Cin / BK;
  const int nk = p.taps * ksteps_per_tap;
  const int half = p.taps >> 1;

  // tile schedule: units of 8 m-tiles x 4 n-tiles = the 32 workgroups of one XCD per round (conv_gemm256_kernel: every A block is fetched
  // into one L2 once); other shapes: XCD-contiguous runs in groups of 8 m-tiles
  const int G = gridDim.x;
  const bool unit_order = G == 256 && (nbn & 3) == 0 && nbm >= 64;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nch = nbn >> 2, mg_full = nbm >> 3;
  const int full_units = mg_full * nch, R = full_units >> 3, rem_units = full_units - 8 * R;
  const int gm_tail = nbm & 7, tail_tiles = gm_tail * nbn;
  // HALF-TILE TAIL as in conv_gemm256_kernel (round 5): where the whole rounds leave a mostly idle last round (the K = 1024 layers: 12 x 256 + 72 tiles)
  // the left-over bottom rows are computed as 128 x 256 half tiles, at most one per workgroup - same K order per element, results bit-identical;
  // hp_gemm_variant 2 switches it off (A/B, tests)
  const int mt0 = unit_order && (8 * R) % nch == 0 ? ((8 * R) / nch) * 8 : 0;
  const int n_half_m = 2 * (nbm - mt0), half_cap = 32 / nbn;
  const bool half_tail = unit_order && p.half_tail && (8 * R) % nch == 0 && R > 0 && mt0 < nbm && n_half_m <= 8 * half_cap;
  const int nrounds = unit_order ? (half_tail ? R : R + (rem_units + (tail_tiles + 31) / 32 + 7) / 8) : (ntiles + G - 1) / G;
  auto tile_coords = [&](int i, int& tm0, int& tn0) -> bool {
    if (unit_order) {
      int u;
      if (i < R) {
        u = xcd * R + i;
      } else {
        const int q = (i - R) * 8 + xcd;
        if (q >= rem_units) {
          const int pj = (q - rem_units) * 32 + slot;
          if (pj >= tail_tiles) return false;
          tm0 = (mg_full * 8 + pj % gm_tail) * BM2;
          tn0 = (pj / gm_tail) * BN2;
          return true;
        }
        u = 8 * R + q;
      }
      const int mg = u / nch, ch = u - mg * nch;
      tm0 = (mg * 8 + (slot & 7)) * BM2;
      tn0 = (ch * 4 + (slot >> 3)) * BN2;
      return true;
    }
    const int vt = blockIdx.x + G * i;
    if (vt >= ntiles) return false;
    const int tile = xcd_remap(vt, ntiles);
    constexpr int GM = 8;
    const int per_group = GM * nbn;
    const int grp = tile / per_group, in_grp = tile - grp * per_group;
    const int gm = min(nbm - grp * GM, GM);
    tm0 = (grp * GM + in_grp % gm) * BM2;
    tn0 = (in_grp / gm) * BN2;
    return true;
  };

  // DMA assignment: wave w fills rows [32 w, 32 w + 32) of A and of W, 8 rows per wave-instruction.  Lane (rin = row in the piece, pos = chunk
  // position in the 128-byte LDS row) fetches source chunk pos ^ rin (the swizzle lives on the source side): chunks 0-3 = hi plane, 4-7 = lo plane
  const int rin = lane >> 3, pos = lane & 7;
  const int sch = pos ^ rin;
  const uint32_t kch = (uint32_t)(sch & 3) * 8u;                           // element offset of the chunk inside the 32-wide K-step
  const uint32_t a_pl = (sch & 4) ? (uint32_t)p.a_lo : 0u;                 // plane offsets (elements)
  const uint32_t w_pl = (sch & 4) ? (uint32_t)p.N * (uint32_t)Ktot : 0u;   // (the host checks 2 planes x N x Ktot x 2 B < 2^32)
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  // !TAPS: ONE per-lane byte offset (row of piece 0); pieces 1-3 are 8, 16, 24 rows further = a scalar stride on the uniform base.  Only a
  // tile that reaches past row M (the last m-tile) clamps per row, recomputed per K-step there (four 64-bit offset pairs live across the
  // K loop cost the registers the loop does not have: a scratch reload + vmcnt(0) in front of every DMA issue)
  uint32_t aoff0 = 0;
  int arow0 = 0;
  bool interior = true;
  int aseg0 = 0, atl0 = 0;
  uint32_t woff;
  const size_t astride = (size_t)8 * (size_t)p.lda * 2u;
  auto setup_dma = [&](int tm0, int tn0, auto ap_c) {
    constexpr int AP = decltype(ap_c)::value;                             // A pieces per wave: 4 (whole tile: rows [32 w, 32 w + 32)) or 2 (half tile: rows [16 w, 16 w + 16))
    const int row = 8 * AP * wu + rin;
    if constexpr (TAPS) {
      const int mm = min(tm0 + row, p.M - 1);
      aseg0 = (mm / p.T) * p.T;
      atl0 = mm - aseg0;
    } else {
      arow0 = tm0 + row;
      interior = tm0 + 32 * AP <= p.M;
      aoff0 = ((uint32_t)min(arow0, p.M - 1) * (uint32_t)p.lda + kch + a_pl) * 2u;
    }
    woff = ((uint32_t)(tn0 + 32 * wu + rin) * (uint32_t)Ktot + kch + w_pl) * 2u;
  };
  auto issue = [&](int t, int stage, auto ap_c) {
    constexpr int AP = decltype(ap_c)::value;
    const int j = t / ksteps_per_tap;
    const int kc = (t - j * ksteps_per_tap) * BK;
    char* sA = smem + stage * HSTAGE + (8 * AP * wu) * 128;
    char* sB = smem + stage * HSTAGE + BM2 * 128 + (32 * wu) * 128;
    const char* abase = reinterpret_cast<const char*>(p.A + kc);
    if constexpr (TAPS) {
      const int off = (j - half) * p.dil;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        int tl = atl0 + 8 * i, sb = aseg0;
        if (tl >= p.T) { tl -= p.T; sb += p.T; }
        const uint32_t src = (uint32_t)min(sb + reflect_idx(tl + off, p.T), p.M - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(abase + (size_t)((src * (uint32_t)p.lda + kch + a_pl) * 2u)), (lptr_t)(sA + i * 1024), 16, 0, 0);
      }
    } else if (interior) {
#pragma unroll
      for (int i = 0; i < AP; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(abase + i * astride + (size_t)aoff0), (lptr_t)(sA + i * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const uint32_t o = ((uint32_t)min(arow0 + 8 * i, p.M - 1) * (uint32_t)p.lda + kch + a_pl) * 2u;
        __builtin_amdgcn_global_load_lds((gptr_t)(abase + (size_t)o), (lptr_t)(sA + i * 1024), 16, 0, 0);
      }
    }
    const char* wbase = reinterpret_cast<const char*>(p.W + (j * p.Cin + kc));
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wbase + (size_t)i * 16 * Ktot + (size_t)woff), (lptr_t)(sB + i * 1024), 16, 0, 0);
  };

  const int fr = lane & 15, fq = lane >> 4, sw = lane & 7;
  const uint32_t a_base = (wm * 128 + fr) * 128;
  const uint32_t b_base = BM2 * 128 + (wn * 64 + fr) * 128;
  const uint32_t c0 = ((0 * 4 + fq) ^ sw) << 4, c1 = ((1 * 4 + fq) ^ sw) << 4;       // hi / lo fragment of the row
  const bool dma_early = wu < 4;
  const bool relu = p.flags & SDK_GEMM_RELU;
  const float winv = *p.winv;
  float* par = reinterpret_cast<float*>(smem + HLDS);
  constexpr std::integral_constant<int, 4> kWhole{};
  constexpr std::integral_constant<int, 2> kHalf{};
  auto ldB = [&](const char* st, f16x8* dst, uint32_t coff) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) dst[ni] = *reinterpret_cast<const f16x8*>(st + b_base + ni * 2048 + coff);
  };

  for (int rnd = 0; rnd < nrounds; ++rnd) {
    int m0, n0;
    if (!tile_coords(rnd, m0, n0)) continue;
    setup_dma(m0, n0, kWhole);
    float pb = 0.f, psc = 1.f, psh = 0.f;
    if (tid < BN2) {
      if (p.bias) pb = p.bias[n0 + tid];
      if (p.scale) { psc = p.scale[n0 + tid]; psh = p.shift[n0 + tid]; }
    }
    f32x4 acc[8][4];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    f16x8 b0[4], b1[4], a0[4], a1[4];
    auto ldA = [&](const char* st, f16x8* dst, int mh, uint32_t coff) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) dst[mi] = *reinterpret_cast<const f16x8*>(st + a_base + (mh * 4 + mi) * 2048 + coff);
    };
    // the WEIGHT fragment is the MFMA's row operand: a lane holds 4 CONSECUTIVE output columns of one output row
    auto mma_half = [&](const f16x8* af, const f16x8* bf, int mh, int part) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 2 * part; mi < 2 * part + 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mh * 4 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ni], af[mi], acc[mh * 4 + mi][ni], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };

    issue(0, 0, kWhole);
    if (nk > 1) {
      issue(1, 1, kWhole);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (tid < BN2) { par[tid] = pb; par[BN2 + tid] = psc; par[2 * BN2 + tid] = psh; }
    __builtin_amdgcn_s_barrier();
    ldB(smem, b0, c0);
    ldA(smem, a0, 0, c0);
    for (int t = 0; t < nk; ++t) {
      const char* st = smem + (t & 1) * HSTAGE;
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a0, b0, 0, 0);                           // P0: Ah[0] . Wh
      __builtin_amdgcn_sched_barrier(0);
      ldA(st, a1, 1, c0);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a0, b0, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a1, b0, 1, 0);                           // P1: Ah[1] . Wh
      __builtin_amdgcn_sched_barrier(0);
      ldB(st, b1, c1);
      ldA(st, a0, 0, c1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a1, b0, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) b0[ni] = b0[ni] * (_Float16)(1.0f / HP_LOSCALE);     // Wh -> Wh 2^-11 in place (exact)
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a0, b0, 0, 0);                           // P2: Al[0] . Wh 2^-11
      __builtin_amdgcn_sched_barrier(0);
      ldA(st, a1, 1, c1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a0, b0, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a1, b0, 1, 0);                           // P3: Al[1] . Wh 2^-11
      __builtin_amdgcn_sched_barrier(0);
      ldA(st, a0, 0, c0);                               // the hi fragments once more, for the W lo term
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a1, b0, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a0, b1, 0, 0);                           // P4: Ah[0] . Wl
      __builtin_amdgcn_sched_barrier(0);
      ldA(st, a1, 1, c0);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a0, b1, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // own reads of this stage done, own DMA of step t + 1 landed
        __builtin_amdgcn_s_barrier();
        if (dma_early && t + 2 < nk) issue(t + 2, t & 1, kWhole);
        const char* sn = smem + ((t + 1) & 1) * HSTAGE;
        ldB(sn, b0, c0);
        ldA(sn, a0, 0, c0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a1, b1, 1, 0);                           // P5: Ah[1] . Wl
      mma_half(a1, b1, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (!dma_early && t + 2 < nk) issue(t + 2, t & 1, kWhole);
    }

    // ------------------------------------------------------------------ epilogue: fp32 in the accumulators, hi plane then lo plane through the image
    f32x4 qb[4], qs[4], qt[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int c = wn * 64 + ni * 16 + fq * 4;
      qb[ni] = *reinterpret_cast<const f32x4*>(par + c);
      qs[ni] = *reinterpret_cast<const f32x4*>(par + BN2 + c);
      qt[ni] = *reinterpret_cast<const f32x4*>(par + 2 * BN2 + c);
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        f32x4 v = acc[mi][ni] * winv + qb[ni];
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        v = v * qs[ni] + qt[ni];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], -HP_MAX), HP_MAX);
        acc[mi][ni] = v;
      }
#pragma unroll
    for (int plane = 0; plane < 2; ++plane) {
      lds_barrier();                                    // every wave is done with the last stage / with the previous plane's image
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        const int row = wm * 128 + mi * 16 + fr;
        char* rowp = smem + row * (BN2 * 2);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const f32x4 v = acc[mi][ni];
          _Float16 h[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const _Float16 hi = (_Float16)v[e];
            h[e] = plane == 0 ? hi : (_Float16)((v[e] - (float)hi) * HP_LOSCALE);
          }
          uint2 pk;
          pk.x = (uint32_t)__builtin_bit_cast(uint16_t, h[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, h[1]) << 16);
          pk.y = (uint32_t)__builtin_bit_cast(uint16_t, h[2]) | ((uint32_t)__builtin_bit_cast(uint16_t, h[3]) << 16);
          const int u8 = (wn * 16 + ni * 4 + fq) ^ (fr << 1);
          *reinterpret_cast<uint2*>(rowp + u8 * 8) = pk;
        }
      }
      lds_barrier();
      {
        const int r0 = tid >> 5, cc = tid & 31;
        const char* src = smem + r0 * (BN2 * 2) + ((cc ^ (r0 & 15)) << 4);
        uint16_t* dst = p.C + (int64_t)(m0 + r0) * p.ldc + n0 + cc * 8 + (plane ? p.c_lo : 0);
        const int rows_left = p.M - m0 - r0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (16 * i < rows_left) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(src + i * 16 * (BN2 * 2));
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + (int64_t)(16 * i) * p.ldc));
          }
        }
      }
    }
    lds_barrier();                                      // the image is free: the next tile's DMA may overwrite it
  }

  // ------------------------------------------------------------------ half-tile tail: at most one 128 x 256 tile per workgroup (XCD x takes half row
  // blocks x, x + 8, ...: the nbn column tiles of a block share its A rows in one L2)
  if (half_tail && slot / nbn < half_cap && (slot / nbn) * 8 + xcd < n_half_m) {
    const int hq = slot / nbn;
    const int m0 = mt0 * BM2 + (hq * 8 + xcd) * 128, n0 = (slot - hq * nbn) * BN2;
    if (m0 < p.M) {
      setup_dma(m0, n0, kHalf);
      float pb = 0.f, psc = 1.f, psh = 0.f;
      if (tid < BN2) {
        if (p.bias) pb = p.bias[n0 + tid];
        if (p.scale) { psc = p.scale[n0 + tid]; psh = p.shift[n0 + tid]; }
      }
      f32x4 acc[4][4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      f16x8 b0[4], b1[4], a0[4], a1[4];
      const uint32_t a_base_h = (wm * 64 + fr) * 128;          // waves 2 (M) x 4 (N), 64 x 64 each
      auto ldAh = [&](const char* st, f16x8* dst, uint32_t coff) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) dst[mi] = *reinterpret_cast<const f16x8*>(st + a_base_h + mi * 2048 + coff);
      };
      auto mma_h = [&](const f16x8* af, const f16x8* bf, int part) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 2 * part; mi < 2 * part + 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      };
      issue(0, 0, kHalf);
      if (nk > 1) {
        issue(1, 1, kHalf);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (tid < BN2) { par[tid] = pb; par[BN2 + tid] = psc; par[2 * BN2 + tid] = psh; }
      __builtin_amdgcn_s_barrier();
      ldB(smem, b0, c0);
      ldAh(smem, a0, c0);
      for (int t = 0; t < nk; ++t) {                       // the whole tile's three product terms, one row half: Ah.Wh, Al.(Wh 2^-11), Ah.Wl
        const char* st = smem + (t & 1) * HSTAGE;
        __builtin_amdgcn_sched_barrier(0);
        mma_h(a0, b0, 0);                                 // Ah . Wh
        __builtin_amdgcn_sched_barrier(0);
        ldB(st, b1, c1);
        ldAh(st, a1, c1);
        __builtin_amdgcn_sched_barrier(0);
        mma_h(a0, b0, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) b0[ni] = b0[ni] * (_Float16)(1.0f / HP_LOSCALE);     // Wh -> Wh 2^-11 in place (exact)
        __builtin_amdgcn_sched_barrier(0);
        mma_h(a1, b0, 0);                                 // Al . Wh 2^-11
        mma_h(a1, b0, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_h(a0, b1, 0);                                 // Ah . Wl
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nk) {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (dma_early && t + 2 < nk) issue(t + 2, t & 1, kHalf);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_h(a0, b1, 1);                                 // (a0 = Ah of THIS step: still needed; the next step's fragments are read after it)
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nk) {
          const char* sn = smem + ((t + 1) & 1) * HSTAGE;
          ldB(sn, b0, c0);
          ldAh(sn, a0, c0);
        }
        if (!dma_early && t + 2 < nk) issue(t + 2, t & 1, kHalf);
      }
      f32x4 qb[4], qs[4], qt[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int c = wn * 64 + ni * 16 + fq * 4;
        qb[ni] = *reinterpret_cast<const f32x4*>(par + c);
        qs[ni] = *reinterpret_cast<const f32x4*>(par + BN2 + c);
        qt[ni] = *reinterpret_cast<const f32x4*>(par + 2 * BN2 + c);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          f32x4 v = acc[mi][ni] * winv + qb[ni];
          if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          v = v * qs[ni] + qt[ni];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e], -HP_MAX), HP_MAX);
          acc[mi][ni] = v;
        }
#pragma unroll
      for (int plane = 0; plane < 2; ++plane) {
        lds_barrier();
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int row = wm * 64 + mi * 16 + fr;
          char* rowp = smem + row * (BN2 * 2);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const f32x4 v = acc[mi][ni];
            _Float16 h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const _Float16 hi = (_Float16)v[e];
              h[e] = plane == 0 ? hi : (_Float16)((v[e] - (float)hi) * HP_LOSCALE);
            }
            uint2 pk;
            pk.x = (uint32_t)__builtin_bit_cast(uint16_t, h[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, h[1]) << 16);
            pk.y = (uint32_t)__builtin_bit_cast(uint16_t, h[2]) | ((uint32_t)__builtin_bit_cast(uint16_t, h[3]) << 16);
            const int u8 = (wn * 16 + ni * 4 + fq) ^ (fr << 1);
            *reinterpret_cast<uint2*>(rowp + u8 * 8) = pk;
          }
        }
        lds_barrier();
        {
          const int r0 = tid >> 5, cc = tid & 31;
          const char* src = smem + r0 * (BN2 * 2) + ((cc ^ (r0 & 15)) << 4);
          uint16_t* dst = p.C + (int64_t)(m0 + r0) * p.ldc + n0 + cc * 8 + (plane ? p.c_lo : 0);
          const int rows_left = p.M - m0 - r0;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (16 * i < rows_left) {
              const u32x4 v = *reinterpret_cast<const u32x4*>(src + i * 16 * (BN2 * 2));
              __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + (int64_t)(16 * i) * p.ldc));
            }
          }
        }
      }
    }
  }
}

// ================================================================================================ sweeps (HBM-bound, planes in / out)
// per-segment channel means of z (SE squeeze): grid (B, C / 512), 256 threads = 64 chunk-columns x 4 frame groups
__global__ __launch_bounds__(256) void seg_mean_hp_kernel(const uint16_t* __restrict__ z, int64_t ldz, int64_t z_lo, int T, int C,
                                                         float* __restrict__ out) {
  __shared__ float red[4][512];
  const int tid = threadIdx.x, c8 = tid & 63, grp = tid >> 6;
  const int cbase = blockIdx.y * 512 + c8 * 8;
  const int64_t base = (int64_t)blockIdx.x * T;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (cbase < C)
    for (int t = grp; t < T; t += 4) {
      float f[8];
      load8(z + (base + t) * ldz + cbase, z_lo, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += f[e];
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[grp][c8 * 8 + e] = s[e];
  __syncthreads();
  for (int c = tid; c < 512; c += 256)
    if (blockIdx.y * 512 + c < C)
      out[(int64_t)blockIdx.x * C + blockIdx.y * 512 + c] = (((red[0][c] + red[1][c]) + red[2][c]) + red[3][c]) * (1.0f / (float)T);
}

// out = g * z + x (SE gate + residual), planes in and out; grid (B, ceil(T / 8)), thread = (frame of 8, chunk column)
__global__ __launch_bounds__(256) void se_apply_hp_kernel(const uint16_t* __restrict__ z, int64_t ldz, int64_t z_lo,
                                                         const uint16_t* __restrict__ x, int64_t ldx, int64_t x_lo,
                                                         const float* __restrict__ gate, uint16_t* __restrict__ out, int64_t ldo,
                                                         int64_t o_lo, int T, int C) {
  const int nch8 = C >> 3;
  const int64_t base = (int64_t)blockIdx.x * T;
  const int t0 = blockIdx.y * 8;
  for (int i = threadIdx.x; i < 8 * nch8; i += 256) {
    const int t = t0 + i / nch8, c8 = i % nch8;
    if (t >= T) break;
    float fz[8], fx[8];
    load8(z + (base + t) * ldz + c8 * 8, z_lo, fz);
    load8(x + (base + t) * ldx + c8 * 8, x_lo, fx);
    const float* g = gate + (int64_t)blockIdx.x * C + c8 * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) fz[e] = g[e] * fz[e] + fx[e];
    store8(out + (base + t) * ldo + c8 * 8, o_lo, fz);
  }
}

// ASP global context: mean | std over frames (shifted single pass as the default-mode kernel: shift = frame 0)
__global__ __launch_bounds__(256) void asp_stats_hp_kernel(const uint16_t* __restrict__ h, int64_t ldh, int64_t h_lo, int T, int C,
                                                          float* __restrict__ out) {
  __shared__ float red[2][2][1024];
  const int tid = threadIdx.x, c8 = tid & 127, grp = tid >> 7;
  const int cbase = blockIdx.y * 1024 + c8 * 8;
  const int64_t base = (int64_t)blockIdx.x * T;
  const bool live = cbase < C;
  float K[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { K[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; }
  if (live) {
    load8(h + base * ldh + cbase, h_lo, K);
    for (int t = grp; t < T; t += 2) {
      float f[8];
      load8(h + (base + t) * ldh + cbase, h_lo, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = f[e] - K[e];
        s1[e] += d;
        s2[e] += d * d;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[grp][0][c8 * 8 + e] = s1[e]; red[grp][1][c8 * 8 + e] = s2[e]; }
  __syncthreads();
  if (grp == 0 && live) {
    const float invT = 1.0f / (float)T;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = (red[0][0][c8 * 8 + e] + red[1][0][c8 * 8 + e]) * invT;
      const float q = (red[0][1][c8 * 8 + e] + red[1][1][c8 * 8 + e]) * invT;
      out[(int64_t)blockIdx.x * 2 * C + cbase + e] = K[e] + a;
      out[(int64_t)blockIdx.x * 2 * C + C + cbase + e] = sqrtf(fmaxf(q - a * a, 1e-12f));
    }
  }
}

// attentive statistics pooling from fp32 logits: softmax over frames per channel, weighted mean / std of h.  One thread per channel
// (grid (B, C / 256)), ONE pass over the segment's column: online softmax (running maximum, the three running sums rescaled when it grows),
// moments around the first frame's value, all fp32.
__global__ __launch_bounds__(256) void asp_pool_hp_kernel(const float* __restrict__ logits, int64_t ldl, const uint16_t* __restrict__ h,
                                                         int64_t ldh, int64_t h_lo, int T, int C, float* __restrict__ pooled) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= C) return;
  const int64_t base = (int64_t)blockIdx.x * T;
  const float K = load1(h + base * ldh + c, h_lo);
  float mx = -INFINITY, se = 0.f, s1 = 0.f, s2 = 0.f;
  for (int t = 0; t < T; ++t) {
    const float l = logits[(base + t) * ldl + c];
    const float d = load1(h + (base + t) * ldh + c, h_lo) - K;
    if (l > mx) {                                     // rescale what has been summed so far to the new maximum
      const float r = expf(mx - l);                   // exp(-inf) = 0 on the first frame
      se *= r; s1 *= r; s2 *= r;
      mx = l;
    }
    const float w = expf(l - mx);
    se += w;
    s1 = fmaf(w, d, s1);
    s2 = fmaf(w * d, d, s2);
  }
  const float a = s1 / se, q = s2 / se;
  pooled[(int64_t)blockIdx.x * 2 * C + c] = K + a;
  pooled[(int64_t)blockIdx.x * 2 * C + C + c] = sqrtf(fmaxf(q - a * a, 1e-12f));
}

}  // namespace

// ================================================================================================ host side
extern "C" int sdk_conv_gemm_hp(sdk_ctx* ctx, const sdk_conv_gemm_hp_args* a, void* stream) {
  SDK_REQUIRE(ctx && a && a->A && a->W, "sdk_conv_gemm_hp: null argument");
  SDK_REQUIRE(a->M > 0 && a->N > 0 && a->N % BN == 0, "sdk_conv_gemm_hp: N=%d must be a positive multiple of %d", a->N, BN);
  SDK_REQUIRE(a->Cin > 0 && a->Cin % BK == 0, "sdk_conv_gemm_hp: Cin=%d must be a multiple of %d", a->Cin, BK);
  SDK_REQUIRE(a->taps >= 1 && (a->taps & 1), "sdk_conv_gemm_hp: taps=%d must be odd", a->taps);
  SDK_REQUIRE(a->T > 0 && a->M % a->T == 0, "sdk_conv_gemm_hp: M=%d must be a multiple of T=%d", a->M, a->T);
  SDK_REQUIRE((int64_t)a->N * a->taps * a->Cin < (1ll << 30), "sdk_conv_gemm_hp: weight matrix too large");
  SDK_REQUIRE(a->taps == 1 || (a->taps / 2) * a->dil < a->T, "sdk_conv_gemm_hp: segment of T=%d frames shorter than the conv halo", a->T);
  SDK_REQUIRE(a->lda % 8 == 0 && a->a_lo % 8 == 0 && a->a_lo >= a->Cin && a->lda >= a->a_lo + a->Cin, "sdk_conv_gemm_hp: bad lda / a_lo");
  SDK_REQUIRE(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->W % 16) == 0, "sdk_conv_gemm_hp: A/W must be 16-byte aligned");
  SDK_REQUIRE(((int64_t)a->N * a->taps * a->Cin) % 8 == 0, "sdk_conv_gemm_hp: weight planes must be 16-byte multiples");
  SDK_REQUIRE(!(a->flags & (SDK_GEMM_A_KBLOCKED | SDK_GEMM_C_KBLOCKED)), "sdk_conv_gemm_hp: the K-blocked layouts belong to the bf16 kernels (sdk_conv_gemm)");
  SDK_REQUIRE(a->C || a->C32 || a->S, "sdk_conv_gemm_hp: no output requested");
  if (a->C) SDK_REQUIRE(a->ldc % 8 == 0 && a->c_lo % 8 == 0 && a->c_lo >= a->N && ((uintptr_t)a->C % 16) == 0, "sdk_conv_gemm_hp: bad C planes");
  if (a->S) SDK_REQUIRE(a->X2 && a->lds % 8 == 0 && a->s_lo % 8 == 0 && a->ldx2 % 8 == 0 && a->x2_lo % 8 == 0 && ((uintptr_t)a->S % 16) == 0 && ((uintptr_t)a->X2 % 16) == 0, "sdk_conv_gemm_hp: S needs X2 planes, 16-byte aligned");
  if (a->C32) SDK_REQUIRE(a->ldc32 >= a->N && a->ldc32 % 4 == 0 && ((uintptr_t)a->C32 % 16) == 0, "sdk_conv_gemm_hp: bad C32");
  if (a->ubias) SDK_REQUIRE(a->ldub >= a->N, "sdk_conv_gemm_hp: bad ldub");
  if (sdk_lds_optin(ctx, (const void*)conv_gemm_hp_kernel, LDS_BYTES)) return 1;
  if (sdk_lds_optin(ctx, (const void*)conv_gemm_hp256_kernel<false>, HLDS_TOTAL)) return 1;
  if (sdk_lds_optin(ctx, (const void*)conv_gemm_hp256_kernel<true>, HLDS_TOTAL)) return 1;
  HpParams p;
  p.A = a->A; p.lda = a->lda; p.a_lo = a->a_lo; p.W = a->W + HP_WHDR; p.winv = reinterpret_cast<const float*>(a->W);
  p.C = a->C; p.ldc = a->ldc; p.c_lo = a->c_lo; p.C32 = a->C32; p.ldc32 = a->ldc32;
  p.bias = a->bias; p.scale = a->scale; p.shift = a->shift; p.ubias = a->ubias; p.ldub = a->ldub;
  p.X2 = a->X2; p.ldx2 = a->ldx2; p.x2_lo = a->x2_lo; p.S = a->S; p.lds = a->lds; p.s_lo = a->s_lo;
  p.M = a->M; p.N = a->N; p.Cin = a->Cin; p.taps = a->taps; p.dil = a->dil; p.T = a->T; p.flags = a->flags;
  p.half_tail = ctx->hp_gemm_variant != 2;
  const double kk = (double)a->taps * a->Cin;
  ProfScope ps(ctx, stream, SDK_K_CONV_GEMM_HP, 3 * 2.0 * a->M * a->N * kk,
               4.0 * a->M * a->Cin + 4.0 * a->N * kk + (a->C ? 4.0 : 0.0) * a->M * a->N + (a->C32 ? 4.0 : 0.0) * a->M * a->N + (a->S ? 8.0 : 0.0) * a->M * a->N);
  // the 256^2 LDS-DMA kernel takes the plain layer shape (planes out, bias / ReLU / BN affine); it addresses A by 32-bit byte offsets
  const bool use256 = ctx->hp_gemm_variant != 1 && a->N % BN2 == 0 && a->M >= BM2 && a->C && !a->C32 && !a->S && !a->ubias && !(a->flags & SDK_GEMM_TANH) &&
                      (a->taps == 1 || a->T >= 64) && (uint64_t)a->M * (uint64_t)a->lda * 2u < (1ull << 32) &&
                      (!a->bias || ((uintptr_t)a->bias % 16) == 0) && (!a->scale || (a->shift && (((uintptr_t)a->scale | (uintptr_t)a->shift) % 16) == 0));
  if (use256) {
    const int ntiles = (a->N / BN2) * ceil_div(a->M, BM2);
    const int cus = ctx->num_cu > 0 ? (ctx->num_cu / 8) * 8 : 256;
    const int grid = ntiles < cus ? ntiles : cus;
    hipLaunchKernelGGL(a->taps > 1 ? conv_gemm_hp256_kernel<true> : conv_gemm_hp256_kernel<false>, dim3(grid), dim3(NT2), HLDS_TOTAL, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(conv_gemm_hp_kernel, dim3((a->N / BN) * ceil_div(a->M, BM)), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
  }
  SDK_LAUNCH_CHECK();
  return 0;
}

int hp_seg_mean(sdk_ctx* ctx, const uint16_t* z, int64_t ldz, int64_t z_lo, int B, int T, int C, float* out, void* stream) {
  ProfScope ps(ctx, stream, SDK_K_SE_GATE, 1.0 * B * T * C, 4.0 * B * T * C);
  hipLaunchKernelGGL(seg_mean_hp_kernel, dim3(B, ceil_div(C, 512)), dim3(256), 0, (hipStream_t)stream, z, ldz, z_lo, T, C, out);
  SDK_LAUNCH_CHECK();
  return 0;
}

int hp_se_apply(sdk_ctx* ctx, const uint16_t* z, int64_t ldz, int64_t z_lo, const uint16_t* x, int64_t ldx, int64_t x_lo, const float* gate,
                uint16_t* out, int64_t ldo, int64_t o_lo, int B, int T, int C, void* stream) {
  ProfScope ps(ctx, stream, SDK_K_SE_GATE, 2.0 * B * T * C, 12.0 * B * T * C);
  hipLaunchKernelGGL(se_apply_hp_kernel, dim3(B, ceil_div(T, 8)), dim3(256), 0, (hipStream_t)stream, z, ldz, z_lo, x, ldx, x_lo, gate, out, ldo, o_lo, T, C);
  SDK_LAUNCH_CHECK();
  return 0;
}

int hp_asp_stats(sdk_ctx* ctx, const uint16_t* h, int64_t ldh, int64_t h_lo, int B, int T, int C, float* out, void* stream) {
  ProfScope ps(ctx, stream, SDK_K_ASP_STATS, 3.0 * B * T * C, 4.0 * B * T * C);
  hipLaunchKernelGGL(asp_stats_hp_kernel, dim3(B, ceil_div(C, 1024)), dim3(256), 0, (hipStream_t)stream, h, ldh, h_lo, T, C, out);
  SDK_LAUNCH_CHECK();
  return 0;
}

int hp_asp_pool(sdk_ctx* ctx, const float* logits, int64_t ldl, const uint16_t* h, int64_t ldh, int64_t h_lo, int B, int T, int C, float* pooled,
                void* stream) {
  ProfScope ps(ctx, stream, SDK_K_ASP_POOL, 8.0 * B * T * C, 12.0 * B * T * C);
  hipLaunchKernelGGL(asp_pool_hp_kernel, dim3(B, ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, logits, ldl, h, ldh, h_lo, T, C, pooled);
  SDK_LAUNCH_CHECK();
  return 0;
}
