// conv_gemm: dilated 1-D convolution over frames lowered to one bf16 MFMA GEMM (gfx950).
//
//   rows  = frames (M = B*T, channel-last activations [M, lda] bf16)
//   K     = taps * Cin, the A operand of tap j is the SAME activation tensor with its rows
//           re-indexed (segment-local reflect of t + (j - taps/2)*dil): no im2col buffer exists,
//           the shifted frame tile is gathered straight into LDS
//   cols  = output channels, W stored [N][taps*Cin] so both MFMA operands are K-contiguous
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 as 4x4 v_mfma_f32_16x16x32_bf16.
// LDS image: [row][64 k] bf16, 128-B rows, 16-B chunk index XOR (row & 7): conflict-free for the
// ds_read_b128 fragment reads (lane groups of MI355X_MICROARCH §LDS) and for the ds_write_b128 fill.
// Software pipeline: global loads of K-step s+1 are in flight (registers) while step s is computed.
// Epilogue (fp32): +bias +per-segment bias, ReLU, BN affine, tanh; result staged through LDS and
// written as whole 16-B chunks; optional second output S = bf16(C + X2) (Res2Net chain).
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#ifndef SDK_GEMM_CHAIN2
#define SDK_GEMM_CHAIN2 2      // 256 x 256 K loop: each accumulator's two products of a K-step back to back; 2 (default): the pair's second MFMA at a raised wave
#endif                         // priority; 1: plain priority; 0: the round-2 ... round-5 order (A/B builds: profiles/r05_gemm_chained_order.txt)

namespace {

// tanh for a value that is rounded to bf16 right away: 1 - 2 / (e^2x + 1) on the fast exp / rcp units (absolute error
// ~1e-7, far inside half a bf16 ulp wherever |tanh| > 1e-4; exact limits +-1).  libm's tanhf costs ~4x the instructions
// and was 13 % of the attention-hidden GEMM.
__device__ __forceinline__ float tanh_bf16(float x) { return 1.0f - __fdividef(2.0f, __expf(2.0f * x) + 1.0f); }

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;
constexpr int LDS_AB = (BM + BN) * BK * 2;        // 32 KiB
constexpr int CT_STRIDE = (BN + 8) * 2;           // bytes per staged C row (272: breaks the 256-B period)
constexpr int LDS_CT = BM * CT_STRIDE;            // 34816
constexpr int LDS_BYTES = LDS_CT > LDS_AB ? LDS_CT : LDS_AB;

struct Params {
  const bf16_t* A; int64_t lda;
  const bf16_t* W;
  bf16_t* C; int64_t ldc;
  float* C32; int64_t ldc32;
  const float* bias; const float* scale; const float* shift;
  const float* ubias; int64_t ldub;
  const bf16_t* X2; int64_t ldx2;
  bf16_t* S; int64_t lds;
  int M, N, Cin, taps, dil, T;
  uint32_t flags;
  int tune;
  float* stats_part; int stats_mode;   // fused per-segment column statistics (256^2 kernel only)
  const bf16_t* A2; int64_t lda2;      // optional addend of the A operand (128^2 kernel only): A := bf16(A + A2)
  int64_t ablk, cblk;                  // K-BLOCKED operands (elements between 64-column blocks, = M * 64; 0 = row-major): A read (128^2 kernel, taps == 1) /
                                       // C written (256^2 kernel) as [cols / 64][M][64]: every 128-byte row piece of a 64-column block is contiguous with its neighbours
  int tap_pack;                        // > 0: the taps are packed along K (W is [N][round64(taps * tap_pack)], a 16-byte chunk of 8 channels
                                       // belongs to tap chunk / (tap_pack / 8)): blk0's 5 x 80 mel channels in 7 K-steps instead of 5 x 128 in 10
  unsigned long long* clk;             // diagnostics (256^2 kernel): per workgroup {shader cycles, 100 MHz ticks} of its lifetime, or null
  unsigned long long* stamps;          // diagnostics (256^2 kernel): [4096] wall-clock stamps of workgroup 0's phases (own buffer: "gemm_stamps"), or null
};

template <bool F16>   // F16: fp16 operands / 2-byte outputs instead of bf16 (SDK_GEMM_F16)
__global__ __launch_bounds__(NT, 2) void conv_gemm_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + BM * BK * 2;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  const int nbn = p.N / BN;
  const int nbm = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, nbn * nbm);
  const int bn = tile % nbn, bm = tile / nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  // ---- staging assignment: thread -> 4 rows (tid>>3)+32i, one 16-B chunk column (tid&7)
  const int ch = tid & 7;
  const int r0 = tid >> 3;
  int segbase[4], tloc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i;
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
  const int cpt = p.tap_pack >> 3;                                       // packed taps: 16-byte chunks per tap
  const int cin_w = p.tap_pack ? BK : p.Cin;                             // (packed: one "tap" of the weight walk per K-step)
  const int Ktot = p.tap_pack ? ((p.taps * p.tap_pack + BK - 1) / BK) * BK : p.taps * p.Cin;
  const bf16_t* wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wrow[i] = p.W + (int64_t)(n0 + r0 + 32 * i) * Ktot + ch * 8;

  uint32_t lds_w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + 32 * i;
    lds_w[i] = row * 128 + ((ch ^ (row & 7)) << 4);
  }

  const int ksteps_per_tap = cin_w / BK;
  const int nk = p.tap_pack ? Ktot / BK : p.taps * ksteps_per_tap;
  const int half = p.taps >> 1;

  u32x4 ra[4], rb[4], ra2[4];
  auto gload = [&](int s) {
    const int j = s / ksteps_per_tap;
    const int kc = (s - j * ksteps_per_tap) * BK;
    int off = (j - half) * p.dil, acol = kc + ch * 8;
    if (p.tap_pack) {                                                    // this thread's chunk of the packed K: its own tap
      const int g = s * 8 + ch;
      int tap = g / cpt;
      acol = (g - tap * cpt) * 8;
      if (tap >= p.taps) { tap = half; acol = 0; }                       // K padding: any valid (finite) activation, its weights are zero
      off = (tap - half) * p.dil;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int src = p.taps > 1 ? segbase[i] + reflect_idx(tloc[i] + off, p.T) : segbase[i];
      ra[i] = *reinterpret_cast<const u32x4*>(p.ablk ? p.A + (int64_t)(acol >> 6) * p.ablk + (int64_t)src * 64 + (acol & 63) : p.A + (int64_t)src * p.lda + acol);
      rb[i] = *reinterpret_cast<const u32x4*>(wrow[i] + j * cin_w + kc);
      if (p.A2) ra2[i] = *reinterpret_cast<const u32x4*>(p.A2 + (int64_t)src * p.lda2 + acol);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read addresses (row & 7 == lane & 7 because tile/wave/sub-tile offsets are multiples of 8)
  const int fr = lane & 15, fq = lane >> 4;
  uint32_t a_off[4], b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_off[i] = (wm * 64 + i * 16 + fr) * 128;
    b_off[i] = (wn * 64 + i * 16 + fr) * 128;
  }
  const int sw = lane & 7;

  gload(0);
  for (int s = 0; s < nk; ++s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (p.A2) {                                            // Res2Net running sum formed on the way into LDS
        float fa[8], fb[8];
        unpack8t<F16>(ra[i], fa);
        unpack8t<F16>(ra2[i], fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[e] += fb[e];
        ra[i] = pack8t<F16>(fa);
      }
      *reinterpret_cast<u32x4*>(sA + lds_w[i]) = ra[i];
      *reinterpret_cast<u32x4*>(sB + lds_w[i]) = rb[i];
    }
    __syncthreads();
    if (s + 1 < nk) gload(s + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint32_t coff = ((ks * 4 + fq) ^ sw) << 4;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(sA + a_off[i] + coff);
        bfr[i] = *reinterpret_cast<const bf16x8*>(sB + b_off[i] + coff);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = mfma_16x16x32<F16>(af[mi], bfr[ni], acc[mi][ni]);
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  float cb[4], cs[4], ct[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int col = n0 + wn * 64 + ni * 16 + fr;
    cb[ni] = p.bias ? p.bias[col] : 0.f;
    cs[ni] = p.scale ? p.scale[col] : 1.f;
    ct[ni] = p.shift ? p.shift[col] : 0.f;
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
        float v = acc[mi][ni][r] + cb[ni];
        if (ub) v += ub[n0 + lc];
        if (relu) v = fmaxf(v, 0.f);
        v = v * cs[ni] + ct[ni];
        if (tnh) v = tanh_bf16(v);
        if (p.C32 && m < p.M) p.C32[(int64_t)m * p.ldc32 + n0 + lc] = v;
        *reinterpret_cast<uint16_t*>(smem + row * CT_STRIDE + lc * 2) = (uint16_t)pack2t<F16>(v, 0.f);
      }
    }
  }
  __syncthreads();
  if (p.C || p.S) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + NT * i;
      const int row = id >> 4, cc = id & 15;
      const int m = m0 + row;
      if (m < p.M) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * CT_STRIDE + cc * 16);
        if (p.C) *reinterpret_cast<u32x4*>(p.C + (int64_t)m * p.ldc + n0 + cc * 8) = v;
        if (p.S) {
          const u32x4 x = *reinterpret_cast<const u32x4*>(p.X2 + (int64_t)m * p.ldx2 + n0 + cc * 8);
          float fv[8], fx[8];
          unpack8t<F16>(v, fv);
          unpack8t<F16>(x, fx);
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] += fx[e];
          *reinterpret_cast<u32x4*>(p.S + (int64_t)m * p.lds + n0 + cc * 8) = pack8t<F16>(fv);
        }
      }
    }
  }
}

}  // namespace

// ================================================================================================
// v2: 256x256x64 tile, 8 waves (2 M x 4 N, 128x64 per wave), operands staged by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip), two 64-KiB LDS stages, counted vmcnt so the next
// K-tile stays in flight across the barriers, raw s_barrier (a __syncthreads() would drain the DMA).
//
//   LDS image per stage: A [256 rows][64 k] then B [256 rows][64 k], 128-B rows; the DMA writes each
//   wave-instruction's 1 KiB linearly (8 rows x 128 B), so the XOR swizzle lives on the SOURCE side:
//   the lane that fills chunk position p of row r fetches global chunk p ^ (r & 7); fragment reads
//   use position c ^ (r & 7) (same involution both sides).
//   The conv row gather (segment-local reflect of the tap offset) is just the per-lane global
//   address of the DMA - the shifted frame tile lands in LDS without ever existing in HBM.
namespace {

constexpr int BM2 = 256, BN2 = 256, NT2 = 512;
constexpr int STAGE2 = (BM2 + BN2) * BK * 2;         // 65536
constexpr int LDS2 = 2 * STAGE2;                     // 131072
constexpr int LDS2_PAR = LDS2 + 3 * BN2 * 4;         // + the tile's bias / scale / shift columns
constexpr int LDS2_TOTAL = LDS2_PAR + 8 * 2 * BN2 * 4;   // + the per-wave partials of the fused column statistics: [8 waves][2 parts][256 columns] fp32 (147 KiB of 160)
static_assert(BM2 * BN2 * 2 <= LDS2, "the bf16 image of a finished tile must fit in the two pipeline stages");

typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

// CKB: the output is written K-blocked (Params::cblk) - its own instantiation, so the default kernel's code is untouched.
// F16: operands and output are fp16 instead of bf16 (SDK_GEMM_F16: the single-plane fp16 contract) - the same kernel with mfma_f32_16x16x32_f16,
// the output clamped to fp16's finite range and rounded to nearest even; everything else (tile order, K order, epilogue, statistics) is shared.
template <bool TAPS, bool CKB = false, bool F16 = false>
__global__ __launch_bounds__(NT2, 2) void conv_gemm256_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // in-kernel clock probe (bench.py prints it next to the peaks): two scalar reads per workgroup, only when asked for
  const unsigned long long clk_c0 = p.clk ? __builtin_amdgcn_s_memtime() : 0, clk_r0 = p.clk ? __builtin_amdgcn_s_memrealtime() : 0;
  const int wm = wid >> 2, wn = wid & 3;

  const int nbn = p.N / BN2;
  const int nbm = (p.M + BM2 - 1) / BM2;
  const int ntiles = nbn * nbm;
  const int cpt = p.tap_pack >> 3;                                       // packed taps (blk0): 16-byte chunks per tap
  const int cin_w = p.tap_pack ? BK : p.Cin;
  const int Ktot = p.tap_pack ? ((p.taps * p.tap_pack + BK - 1) / BK) * BK : p.taps * p.Cin;
  const int ksteps_per_tap = cin_w / BK;
  const int nk = p.tap_pack ? Ktot / BK : p.taps * ksteps_per_tap;
  const int half = p.taps >> 1;

  // Tile schedule.  A workgroup's XCD is blockIdx & 7 (the grid is a multiple of 8), its slot on the XCD l = blockIdx >> 3.
  // UNIT ORDER (default when the grid is 256 workgroups and N is a multiple of 1024): the output is cut into UNITS of 8 m-tiles x 4
  // n-tiles = exactly the 32 workgroups of one XCD, slot l -> (m = l & 7, n = l >> 3), so in every round an XCD's workgroups
  // read 8 A row blocks (each shared by 4 of them) and 4 weight blocks (each shared by 8) - and nothing else: every A block is
  // fetched into exactly ONE L2, once, provided its four readers stay within the few K-steps that L2 holds.  Units are dealt
  // to the XCDs in contiguous runs (n-chunk fastest inside an m-group); whatever does not fill 8 XCDs evenly - the last
  // (full_units mod 8) units and the partial m-group - is dealt out unit by unit in the final round(s), which therefore
  // number exactly what the plain order needs.  (Round 1's order took an XCD's tile run in steps of 32 from a base that
  // is not a multiple of 32, so the four readers of an A block were split over two rounds on 7 of the 8 XCDs.)
  // LEGACY ORDER (tune bit 6 = `gemm_variant` 1026, kept for A/B): xcd_remap runs, groups of 8 m-tiles.
  // Measured (tools/pmc_gemm.sh, bytes leaving L2 per launch): K = 1024 layers 1209 -> 1031 MB (reads 1.93x -> 1.50x of A + W; what
  // is left is W re-fetched by every XCD in every round: 13 x 8 x 2 MB), the 3072^2 layer 9223 -> 6807 MB (A once per unit = 3x, W once per
  // unit: (8 + 4) x 1.57 MB x 296 units - the floor for 32 tiles of 256^2 per 4-MiB L2).  A per-XCD round barrier on top (bounded
  // spin on an XCD-local counter) changed neither the bytes nor the time: inside a unit the readers already stay together.
  // (Round 2's other order knobs - plain n-fastest, one workgroup walking all n-tiles of an m-tile - were measured behind both and are gone.)
  //
  // HALF-TILE TAIL (round 5).  What the whole rounds leave over is normally a last round with most CUs idle: the six K = 1024 layers of the
  // forward have 3144 tiles = 12 rounds of 256 + 72 tiles, a 13th round 28 % full that costs as much as a full one.  When the left-over region
  // - always the bottom rows of the matrix, all columns - fits the grid as 128 x 256 HALF tiles (here 35 half row blocks x 4 = 140 of them),
  // the rounds stop at R and every workgroup takes at most one half tile: same LDS image and fragment order, 4 instead of 8 accumulator
  // row blocks per wave (2 x 4 waves of 64 x 64), so every output element sees the same K order and the result - and the per-half-tile
  // partials of the fused column statistics, which were per 128 rows already - are bit-identical to the whole-tile schedule
  // (tests/test_gpu_kernels.py::test_conv_gemm_half_tile_tail_is_bit_identical).  tune bit 9 (`gemm_variant` 8194) switches it off: A/B.
  const int G = gridDim.x;
  const bool unit_order = !(p.tune & 64) && G == 256 && (nbn & 3) == 0 && nbm >= 64;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nch = nbn >> 2, mg_full = nbm >> 3;
  const int full_units = mg_full * nch, R = full_units >> 3, rem_units = full_units - 8 * R;
  const int gm_tail = nbm & 7, tail_tiles = gm_tail * nbn;
  // left-over region of the unit order: m-tiles [mt0, nbm) x all columns, provided the R whole rounds end on an m-group boundary
  const int mt0 = unit_order && (8 * R) % nch == 0 ? ((8 * R) / nch) * 8 : 0;
  const int n_half_m = 2 * (nbm - mt0);                                   // half row blocks of the region (the last one may lie wholly below the matrix)
  const int half_cap = 32 / nbn;                                          // half row blocks one XCD's 32 workgroups can take (all nbn columns each)
  const bool half_tail = unit_order && !(p.tune & 512) && (8 * R) % nch == 0 && R > 0 && mt0 < nbm && n_half_m <= 8 * half_cap;
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

  // ---- DMA assignment: wave wid fills rows [8 AP wid, 8 AP wid + 8 AP) of A (AP = 4 pieces per wave for a whole tile, 2 for a half tile) and rows
  // [32 wid, 32 wid + 32) of B, 8 rows per instruction.
  // Addresses are (uniform base pointer advanced per K-step on the scalar unit) + (32-bit per-lane byte offset fixed per
  // tile): a 1 x 1 layer's K loop then carries NO vector arithmetic for its 8 DMA pieces (it had ~40 VALU + 8 readfirstlane
  // per K-step; beside MFMAs that is clock, not cycles - MI355X_MICROARCH 'DVFS give-back' item 4).  The host routes operands
  // whose byte offsets do not fit 32 bits to the 128^2 kernel.
  const int rin = lane >> 3, pos = lane & 7;
  const int gch = (pos ^ rin) * 8;                   // source chunk (elements) for this lane's LDS position
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const bool a_nt = !TAPS && nbn <= 4;
  uint32_t aoff[4];                                  // !TAPS: byte offset of this lane's chunk in each of its A rows (rows step by 8)
  int aseg0 = 0, atl0 = 0;                           // TAPS: segment base row and in-segment frame of the first row
  uint32_t woff;
  auto setup_dma = [&](int tm0, int tn0, auto ap_c) {
    constexpr int AP = decltype(ap_c)::value;
    const int row = 8 * AP * wu + rin;
    if constexpr (TAPS) {
      const int mm = min(tm0 + row, p.M - 1);
      aseg0 = (mm / p.T) * p.T;
      atl0 = mm - aseg0;                             // rows past M fetch some valid row; their results are dropped
    } else {
#pragma unroll
      for (int i = 0; i < AP; ++i) aoff[i] = ((uint32_t)min(tm0 + row + 8 * i, p.M - 1) * (uint32_t)p.lda + (uint32_t)gch) * 2u;
    }
    woff = ((uint32_t)(tn0 + 32 * wu + rin) * (uint32_t)Ktot + (uint32_t)gch) * 2u;
  };
  auto issue = [&](int t, int stage, auto ap_c) {
    constexpr int AP = decltype(ap_c)::value;
    const int j = t / ksteps_per_tap;
    const int kc = (t - j * ksteps_per_tap) * BK;
    char* sA = smem + stage * STAGE2 + (8 * AP * wu) * 128;
    char* sB = smem + stage * STAGE2 + BM2 * BK * 2 + (32 * wu) * 128;
    const char* abase = reinterpret_cast<const char*>(p.A + kc);
    if constexpr (TAPS) {
      int off = (j - half) * p.dil;
      uint32_t acol = (uint32_t)gch;
      if (p.tap_pack) {
        // taps packed along K: this lane's 16-byte chunk of the K-step is chunk g of the packed row [tap 0 | tap 1 | ...], i.e. its OWN
        // tap's row shift - the gather is per lane anyway (abase carries no K offset here: kc = 0)
        const int g = t * 8 + (gch >> 3);
        int tap = g / cpt;
        acol = (uint32_t)(g - tap * cpt) * 8u;
        if (tap >= p.taps) { tap = half; acol = 0; }    // K padding: any valid (finite) activation, its weights are zero
        off = (tap - half) * p.dil;
      }
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        int tl = atl0 + 8 * i, sb = aseg0;
        if (tl >= p.T) { tl -= p.T; sb += p.T; }      // T >= 64 > 24: at most one segment boundary inside the rows of a wave
        const uint32_t src = (uint32_t)min(sb + reflect_idx(tl + off, p.T), p.M - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(abase + (size_t)((src * (uint32_t)p.lda + acol) * 2u)), (lptr_t)(sA + i * 1024), 16, 0, 0);
      }
    } else if (a_nt) {
      // A is read exactly ONCE per XCD when the layer has a single n-chunk (N = 1024): fetched non-temporal, its 4 MB per round no
      // longer push the 2 MB of weights out of the 4-MiB L2, which were re-fetched by every XCD in every round (PMC, tools/pmc_gemm.sh:
      // 619 -> 429 MB read per launch = 1.04 x A + W; the four readers of a block still hit each other's lines).  Not for wider layers:
      // on 3072^2 the m-group's A is re-read by three n-chunks from the Infinity Cache, and the hint costs 10 % there.
#pragma unroll
      for (int i = 0; i < AP; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(abase + (size_t)aoff[i]), (lptr_t)(sA + i * 1024), 16, 0, 2);
    } else {
#pragma unroll
      for (int i = 0; i < AP; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(abase + (size_t)aoff[i]), (lptr_t)(sA + i * 1024), 16, 0, 0);
    }
    const char* wbase = reinterpret_cast<const char*>(p.W + (j * cin_w + kc));
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wbase + (size_t)i * 16 * Ktot + (size_t)woff), (lptr_t)(sB + i * 1024), 16, 0, 0);
  };
  constexpr std::integral_constant<int, 4> kWhole{};
  constexpr std::integral_constant<int, 2> kHalf{};

  const int fr = lane & 15, fq = lane >> 4, sw = lane & 7;
  const uint32_t a_base = (wm * 128 + fr) * 128;
  const uint32_t b_base = BM2 * BK * 2 + (wn * 64 + fr) * 128;
  const uint32_t c0 = ((0 * 4 + fq) ^ sw) << 4, c1 = ((1 * 4 + fq) ^ sw) << 4;
  // the two waves that share a SIMD issue their DMA at different points of a K-step: waves 0-3 right after the barrier, 4-7 after the last MFMA
  // sub-phase (round 2 measured all-early, all-late and odd / even behind this; as a run-time policy it sat inside the K loop - a compile-time fact now)
  const bool dma_early = wu < 4;
  const bool relu = p.flags & SDK_GEMM_RELU;
  const bool stats = p.stats_part != nullptr;
  float* par = reinterpret_cast<float*>(smem + LDS2);            // [3][256]: bias, scale, shift of the tile's columns

  int nst = 0;
  auto stamp = [&]() {   // diagnostics (tools/gemm_timeline.py): per-tile phase stamps of workgroup 0, outside the K loop
    if (p.stamps && blockIdx.x == 0 && tid == 0 && nst < 4096) p.stamps[nst++] = __builtin_amdgcn_s_memrealtime();
  };
  auto ldB = [&](const char* st, bf16x8* dst, uint32_t coff) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) dst[ni] = *reinterpret_cast<const bf16x8*>(st + b_base + ni * 2048 + coff);
  };
  // Fused per-segment column statistics of the STORED (bf16-rounded) output, from the tile's LDS image (round 5 form; round 2's read the image
  // column by column, 128 two-byte LDS reads per thread, and cost 57 us of a 494-us K = 1024 launch - as much as a separate sweep of the output).
  //   phase 1: thread = (16-row slab g = tid >> 5, 8-column chunk cc = tid & 31): sixteen 16-byte reads, 8 running column sums (+ 8 sums of squares).
  //            A wave covers 32 consecutive rows and T >= 128, so at most ONE segment boundary falls inside it: rows before it go to part L, rows
  //            from it on to part U (no boundary: everything is L).  The two slabs of a wave are added across lane ^ 32; per wave
  //            wst[wave][L | U][256 columns].
  //   barrier (this is also the point after which the image may be overwritten by the next tile's DMA)
  //   phase 2: thread = (column, 128-row half): its half's four waves in order, L then U, each into the segment slot it belongs to.
  // Fixed order throughout (reproducible); a half tile runs the same code for its 128 rows, so its partials equal the whole tile's.
  // mp = origin of the 256-row tile the rows belong to (segment slots and stats_part rows are per such tile), lo = first image row's offset in
  // it (0, or 128 for the lower half tile), nrows = rows in the image (256 / 128).  Every thread of the workgroup calls it (barriers inside).
  // (Round 5, late: mode 2 with ONE barrier - sums of squares published in the same pass into a compact [8 + 2][256] buffer - measured the same tile
  // period, 87.9 against 88.1 us on the 3072^2 layer: the barriers are not what the statistics cost; the waves that hold a segment boundary or the
  // matrix edge run the masked path, ~2 x the vector work of the others, and everybody waits for them.  Not kept: 10 KB more LDS for nothing.)
  float* wst = reinterpret_cast<float*>(smem + LDS2_PAR);
  auto tile_stats = [&](int mp, int lo, int nrows, int n0) {
    const int sb1 = (mp / p.T + 1) * p.T - mp, sb2 = sb1 + p.T;          // tile-local rows where the tile's 2nd / 3rd segment start
    const int valid = min(nrows, p.M - mp - lo);                        // image rows [0, valid) are rows of the matrix
    const bool mode2 = p.stats_mode == 2;
    const bool active = 32 * wu < nrows;                                // (half tile: waves 0-3)
    const int g = tid >> 5, cc = tid & 31;
    float L[8], U[8], LQ[8], UQ[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { L[e] = 0.f; U[e] = 0.f; LQ[e] = 0.f; UQ[e] = 0.f; }
    if (active) {
      const int wlo = lo + 32 * wu;                                     // tile-local first row of this wave
      int bnd = wlo + 32;
      if (sb1 > wlo && sb1 < wlo + 32) bnd = sb1;
      else if (sb2 > wlo && sb2 < wlo + 32) bnd = sb2;
      const char* base = smem + (16 * g) * (BN2 * 2);
      if (bnd == wlo + 32 && 32 * wu + 32 <= valid) {                   // the usual wave: one segment, all rows inside the matrix (wave-uniform)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(base + i * (BN2 * 2) + ((cc ^ i) << 4));      // (16 g + i) & 15 == i
          float f[8];
          unpack8t<F16>(v, f);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            L[e] += f[e];
            if (mode2) LQ[e] = fmaf(f[e], f[e], LQ[e]);
          }
        }
      } else {
        const int ti = bnd - lo - 16 * g, ni = valid - 16 * g;          // this slab: rows i < ti are before the boundary, rows i < ni inside the matrix
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(base + i * (BN2 * 2) + ((cc ^ i) << 4));
          float f[8];
          unpack8t<F16>(v, f);
          // branches, not selects (round 5, late): a row goes to ONE part, so the other part's adds are not issued at all where no lane of the
          // wave needs them (execz skip), instead of three selects and both parts' arithmetic per element.  Adding nothing equals adding the
          // 0.f the select form added: same sums.  (Rows below the matrix hold whatever the padded MFMA rows produced: skipped.)
          if (i < ni) {
            if (i < ti) {
#pragma unroll
              for (int e = 0; e < 8; ++e) { L[e] += f[e]; if (mode2) LQ[e] = fmaf(f[e], f[e], LQ[e]); }
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) { U[e] += f[e]; if (mode2) UQ[e] = fmaf(f[e], f[e], UQ[e]); }
            }
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {                                     // the wave's two slabs (lanes l, l ^ 32): both lanes end up with the same totals
        L[e] += __shfl_xor(L[e], 32, 64); U[e] += __shfl_xor(U[e], 32, 64);
        if (mode2) { LQ[e] += __shfl_xor(LQ[e], 32, 64); UQ[e] += __shfl_xor(UQ[e], 32, 64); }
      }
    }
    float* wrow = wst + (size_t)((((lo >> 5) + wu) * 2 + (lane >> 5)) * BN2) + cc * 8;       // lanes 0-31 publish L, lanes 32-63 U
    const int c = tid & 255, hs = nrows == BM2 ? tid >> 8 : lo >> 7;
    const bool fin = nrows == BM2 || tid < BN2;
    auto publish = [&](const float* l, const float* u) {
      if (active) {
        const bool up = lane >= 32;
        *reinterpret_cast<f32x4*>(wrow) = f32x4{up ? u[0] : l[0], up ? u[1] : l[1], up ? u[2] : l[2], up ? u[3] : l[3]};
        *reinterpret_cast<f32x4*>(wrow + 4) = f32x4{up ? u[4] : l[4], up ? u[5] : l[5], up ? u[6] : l[6], up ? u[7] : l[7]};
      }
    };
    auto combine = [&](float* dst) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int w = 4 * hs + k, wl = 32 * w;                          // tile-local first row of wave w's 32 rows
        const int seg = (wl >= sb1) + (wl >= sb2);
        const float a = wst[(w * 2) * BN2 + c], b = wst[(w * 2 + 1) * BN2 + c];       // b = 0 where the wave holds no boundary
        s0 += seg == 0 ? a : 0.f; s1 += seg == 1 ? a : 0.f; s2 += seg == 2 ? a : 0.f;
        s1 += seg == 0 ? b : 0.f; s2 += seg == 1 ? b : 0.f;
      }
      dst[0] = s0; dst[p.N] = s1; dst[2 * (int64_t)p.N] = s2;
    };
    float* dst = p.stats_part + ((int64_t)((mp / BM2) * 2 + hs) * 3) * p.N + n0 + c;
    publish(L, U);
    lds_barrier();                                      // partials visible - and every read of the image is done: the next tile's DMA may overwrite it
    if (fin) combine(dst);
    if (mode2) {
      lds_barrier();                                    // the partial buffer is free again
      publish(LQ, UQ);
      lds_barrier();
      if (fin) combine(dst + (int64_t)nbm * 6 * p.N);
    }
  };
  // Persistent workgroups (one per CU; the grid is a multiple of 8 so a workgroup keeps its XCD class) walk
  // their tiles back to back.
  for (int rnd = 0; rnd < nrounds; ++rnd) {
    int m0, n0;
    if (!tile_coords(rnd, m0, n0)) continue;
    setup_dma(m0, n0, kWhole);
    // the tile's 256 columns of epilogue parameters travel through LDS: the loads ride under the K loop
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

    // The WEIGHT fragment is the MFMA's row operand: a lane then holds 4 CONSECUTIVE output columns (fq*4 + r) of one output row (fr), which the
    // epilogue packs into one 8-byte LDS write.
#if !SDK_GEMM_CHAIN2
    bf16x8 b0[4], b1[4], a0[4], a1[4];
    auto ldA = [&](const char* st, bf16x8* dst, int mh, uint32_t coff) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) dst[mi] = *reinterpret_cast<const bf16x8*>(st + a_base + (mh * 4 + mi) * 2048 + coff);
    };
    // 16 MFMAs of one sub-phase in two halves: the fragment reads of the NEXT sub-phase are issued between
    // the halves, so they complete under the second half instead of being waited for right after issue.
    // The WEIGHT fragment is the MFMA's row operand: a lane then holds 4 CONSECUTIVE output columns
    // (fq*4 + r) of one output row (fr), which the epilogue packs into one 8-byte LDS write.
    auto mma_half = [&](const bf16x8* af, const bf16x8* bf, int mh, int part) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 2 * part; mi < 2 * part + 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          if constexpr (F16)
            acc[mh * 4 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, bf[ni]), __builtin_bit_cast(f16x8_t, af[mi]), acc[mh * 4 + mi][ni], 0, 0, 0);
          else
            acc[mh * 4 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni], af[mi], acc[mh * 4 + mi][ni], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
#endif

    // Software pipeline over the K-tiles, one barrier per K-tile placed BEFORE the last MFMA sub-phase:
    //   P0 P1 P2 | own reads of tile t done, own DMA of tile t+1 landed, barrier |
    //   issue DMA of tile t+2 into the stage just freed, prefetch tile t+1's first fragments | P3
    // The two waves that share a SIMD issue their DMA at different points (before / after P3).
    issue(0, 0, kWhole);
    if (nk > 1) {
      issue(1, 1, kWhole);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // K-step 0 and the epilogue parameters have landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (tid < BN2) { par[tid] = pb; par[BN2 + tid] = psc; par[2 * BN2 + tid] = psh; }
    __builtin_amdgcn_s_barrier();
    stamp();
#if SDK_GEMM_CHAIN2
    // Round 5, late: the K-step's 64 MFMAs ordered so that every accumulator tile takes its TWO products (k 0-31, k 32-63) back to back.  At the
    // board's power cap the order is worth energy (tools/probe/mfma_energy.hip, bare loops on random operands, same 64 products per iteration):
    // the order below delivers 6.6-7.3 % more FLOP/s than the (ks, mh, mi, ni) order this kernel used - as much as removing ALL operand toggling -
    // because the second MFMA of a pair takes the accumulator from the matrix pipe instead of the register file (one other MFMA between the two and
    // the gain is gone: probe variants 7 / 8).  In THIS kernel the two waves of a SIMD feed one pipe, and a pair is split whenever the partner's MFMA
    // wins the slot in between: the plain chained order measured -0.6 % on the 3072^2 layer under sustained load and nothing on K = 1024; raising the
    // wave's priority for the pair's second MFMA (s_setprio 3 ... 1 around it) -1.8 % and -1.1 % at a 1-2 % higher clock
    // (profiles/r05_gemm_chained_order.txt).  Same sum order per element (k 0-31 before k 32-63): bit-identical output.  Four phases of two 16-row blocks each; the W fragments of both k-halves stay in registers for
    // the whole K-step and are replaced column block by column block in the last phase, right after their last use, so the peak stays at 16
    // fragments (8 W + 4 A in use + 4 A in flight), as before; 24 fragment reads per K-step, as before.
    bf16x8 wf[2][4], aA[2][2], aB[2][2];
    auto ldApair = [&](const char* st, bf16x8 (*dst)[2], int pr) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        dst[j][0] = *reinterpret_cast<const bf16x8*>(st + a_base + (pr * 2 + j) * 2048 + c0);
        dst[j][1] = *reinterpret_cast<const bf16x8*>(st + a_base + (pr * 2 + j) * 2048 + c1);
      }
    };
    auto ldWcol = [&](const char* st, int ni) {
      wf[0][ni] = *reinterpret_cast<const bf16x8*>(st + b_base + ni * 2048 + c0);
      wf[1][ni] = *reinterpret_cast<const bf16x8*>(st + b_base + ni * 2048 + c1);
    };
    auto mma_cols = [&](const bf16x8 (*af)[2], int pr, int n_lo, int n_hi) {   // column blocks [n_lo, n_hi) of row pair pr: per accumulator k 0-31 then k 32-63
#ifndef SDK_GEMM_PRIO_CLUSTER
#define SDK_GEMM_PRIO_CLUSTER 0        // the wave's priority inside an MFMA cluster beside the pair's 3 (outside a cluster: 0).  Rounds 2-5 ran clusters at 1; with the pairs
                                       // at 3 the cluster level only costs: 0 measured -1.1 % / -0.4 % (3072^2 / K = 1024, sustained) against 1, and 2 the same as 1
#endif
      if (SDK_GEMM_PRIO_CLUSTER) __builtin_amdgcn_s_setprio(SDK_GEMM_PRIO_CLUSTER);
#pragma unroll
      for (int ni = n_lo; ni < n_hi; ++ni)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
#if SDK_GEMM_CHAIN2 == 2     // the pair's second MFMA at a raised priority: the partner wave's MFMA does not go between the two (see above)
            if (ks == 1) __builtin_amdgcn_s_setprio(3);
#endif
            if constexpr (F16)
              acc[pr * 2 + j][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, wf[ks][ni]), __builtin_bit_cast(f16x8_t, af[j][ks]), acc[pr * 2 + j][ni], 0, 0, 0);
            else
              acc[pr * 2 + j][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][ni], af[j][ks], acc[pr * 2 + j][ni], 0, 0, 0);
#if SDK_GEMM_CHAIN2 == 2
            if (ks == 1) __builtin_amdgcn_s_setprio(SDK_GEMM_PRIO_CLUSTER);
#endif
          }
      if (SDK_GEMM_PRIO_CLUSTER) __builtin_amdgcn_s_setprio(0);
    };
    ldApair(smem, aA, 0);                                // (the order the last phase below issues them in: the counted wait at the head of the loop is the same on both edges)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) ldWcol(smem, ni);
    for (int t = 0; t < nk; ++t) {
      const char* st = smem + (t & 1) * STAGE2;
      __builtin_amdgcn_sched_barrier(0);
      mma_cols(aA, 0, 0, 2);                             // rows 0-31
      __builtin_amdgcn_sched_barrier(0);
      ldApair(st, aB, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_cols(aA, 0, 2, 4);
      __builtin_amdgcn_sched_barrier(0);
      mma_cols(aB, 1, 0, 2);                             // rows 32-63
      __builtin_amdgcn_sched_barrier(0);
      ldApair(st, aA, 2);
      __builtin_amdgcn_sched_barrier(0);
      mma_cols(aB, 1, 2, 4);
      __builtin_amdgcn_sched_barrier(0);
      mma_cols(aA, 2, 0, 2);                             // rows 64-95
      __builtin_amdgcn_sched_barrier(0);
      ldApair(st, aB, 3);
      __builtin_amdgcn_sched_barrier(0);
      mma_cols(aA, 2, 2, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // own reads of stage t done, own DMA of step t + 1 landed (see the note in the other branch)
        __builtin_amdgcn_s_barrier();
        if (dma_early && t + 2 < nk) issue(t + 2, t & 1, kWhole);
        ldApair(smem + ((t + 1) & 1) * STAGE2, aA, 0);
      }
      {
        const char* sn = smem + ((t + 1) & 1) * STAGE2;  // (after the last K-step the reads below fetch fragments nobody uses: in bounds, no branch in the phase)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {                 // rows 96-127, one column block at a time: its W fragments are free after it and take the next K-step's
          __builtin_amdgcn_sched_barrier(0);
          mma_cols(aB, 3, ni, ni + 1);
          __builtin_amdgcn_sched_barrier(0);
          ldWcol(sn, ni);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!dma_early && t + 2 < nk) issue(t + 2, t & 1, kWhole);
    }
#else
    ldB(smem, b0, c0);
    ldA(smem, a0, 0, c0);
    for (int t = 0; t < nk; ++t) {
      const char* st = smem + (t & 1) * STAGE2;
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a0, b0, 0, 0);                           // P0
      __builtin_amdgcn_sched_barrier(0);
      ldA(st, a1, 1, c0);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a0, b0, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a1, b0, 1, 0);                           // P1
      __builtin_amdgcn_sched_barrier(0);
      ldB(st, b1, c1);
      ldA(st, a0, 0, c1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a1, b0, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a0, b1, 0, 0);                           // P2
      __builtin_amdgcn_sched_barrier(0);
      ldA(st, a1, 1, c1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a0, b1, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) {
        // Where a K-step waits (round 3, tools/gemm_kstep.py on a diagnostic build with stamps around the wait and the barrier; the stamps are not
        // compiled in any more): MFMA time 1.02-1.14 us of a 1.56-1.64-us step; ~0.3 us waiting for the wave's own DMA of the next step and
        // ~0.1-0.2 us at the barrier.  Pulling the A lines into L2 two steps early (one 4-byte LDS-DMA "touch" per row and wave, younger than
        // the DMA pieces so the counted wait leaves it in flight) moved the wait from the vmcnt to the barrier - their sum stayed ~0.5 us - and
        // cost 2-4 % wall time: what the step waits for is the slowest wave's eight LDS-DMA issues (100-185 cycles each beside ds_reads), not HBM latency.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (dma_early && t + 2 < nk) issue(t + 2, t & 1, kWhole);
        const char* sn = smem + ((t + 1) & 1) * STAGE2;
        ldB(sn, b0, c0);
        ldA(sn, a0, 0, c0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_half(a1, b1, 1, 0);                           // P3
      mma_half(a1, b1, 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (!dma_early && t + 2 < nk) issue(t + 2, t & 1, kWhole);
    }
#endif

    stamp();
    // ------------------------------------------------------------------ epilogue
    // bias / ReLU / BN affine run on the accumulators in registers (packed fp32 ops: a lane's 4 values are 4
    // consecutive columns), the result is rounded to bf16 and written as 8-byte pieces into a [256][256] bf16
    // image of the tile that takes over both (now idle) pipeline stages; one barrier later all 512 threads copy
    // the image out, 16 bytes per lane and 512 contiguous bytes per row.  16-byte units of a row are XORed with
    // (row & 15): conflict-free for the 8-byte writes (16 rows x 2 columns groups per pass) and the 16-byte reads.
    f32x4 qb[4], qs[4], qt[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int c = wn * 64 + ni * 16 + fq * 4;
      qb[ni] = *reinterpret_cast<const f32x4*>(par + c);
      qs[ni] = *reinterpret_cast<const f32x4*>(par + BN2 + c);
      qt[ni] = *reinterpret_cast<const f32x4*>(par + 2 * BN2 + c);
    }
    lds_barrier();                                      // every wave is done reading the last stage
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const int row = wm * 128 + mi * 16 + fr;
      char* rowp = smem + row * (BN2 * 2);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        f32x4 v = acc[mi][ni] + qb[ni];
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        v = v * qs[ni] + qt[ni];
        uint2 pk;
        pk.x = pack2t<F16>(v[0], v[1]);
        pk.y = pack2t<F16>(v[2], v[3]);
        const int u8 = (wn * 16 + ni * 4 + fq) ^ (fr << 1);            // 8-byte unit inside the 512-byte row
        *reinterpret_cast<uint2*>(rowp + u8 * 8) = pk;
      }
    }
    lds_barrier();
    {
      // 256 rows x 32 chunks of 8 columns; thread -> (row r0 + 16 i, chunk cc): the swizzle term (row & 15) does
      // not depend on i, so every read is base + i * 8 KiB
      const int r0 = tid >> 5, cc = tid & 31;
      const char* src = smem + r0 * (BN2 * 2) + ((cc ^ (r0 & 15)) << 4);
      bf16_t* dst = CKB ? p.C + (int64_t)((n0 >> 6) + (cc >> 3)) * p.cblk + (int64_t)(m0 + r0) * 64 + (cc & 7) * 8
                        : p.C + (int64_t)(m0 + r0) * p.ldc + n0 + cc * 8;
      const int64_t ldc = CKB ? 64 : p.ldc;
      const int rows_left = p.M - m0 - r0;               // rows r0 + 16 i < rows_left are inside the matrix
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (16 * i < rows_left) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(src + i * 16 * (BN2 * 2));
          __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + (int64_t)(16 * i) * ldc));
        }
      }
    }
    if (stats) tile_stats(m0, 0, BM2, n0);
    else
    lds_barrier();                                      // the image is free: the next tile's DMA may overwrite it
    stamp();
  }   // persistent tile loop

  // ------------------------------------------------------------------ half-tile tail: at most one 128 x 256 tile per workgroup
  // XCD x takes half row blocks x, x + 8, ... (all nbn column tiles of a block on one XCD: its four readers share the A rows in one L2)
  if (half_tail && slot / nbn < half_cap && (slot / nbn) * 8 + xcd < n_half_m) {
    const int hq = slot / nbn;
    const int m0 = mt0 * BM2 + (hq * 8 + xcd) * 128, n0 = (slot - hq * nbn) * BN2;
    if (m0 >= p.M) {
      // the lower half of an edge tile, wholly below the matrix: nothing to compute, but its statistics slot is read by colstats_finish
      if (stats && tid < BN2) {
        float* dst = p.stats_part + ((int64_t)((m0 / BM2) * 2 + 1) * 3) * p.N + n0 + tid;
        dst[0] = 0.f; dst[p.N] = 0.f; dst[2 * (int64_t)p.N] = 0.f;
        if (p.stats_mode == 2) {
          float* dq = dst + (int64_t)nbm * 6 * p.N;
          dq[0] = 0.f; dq[p.N] = 0.f; dq[2 * (int64_t)p.N] = 0.f;
        }
      }
    } else {
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
    bf16x8 b0[4], b1[4], a0[4], a1[4];
    const uint32_t a_base_h = (wm * 64 + fr) * 128;      // waves 2 (M) x 4 (N), 64 x 64 each
    auto ldAh = [&](const char* st, bf16x8* dst, uint32_t coff) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) dst[mi] = *reinterpret_cast<const bf16x8*>(st + a_base_h + mi * 2048 + coff);
    };
    auto mma_h = [&](const bf16x8* af, const bf16x8* bf, int part) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 2 * part; mi < 2 * part + 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          if constexpr (F16)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, bf[ni]), __builtin_bit_cast(f16x8_t, af[mi]), acc[mi][ni], 0, 0, 0);
          else
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    // the same pipeline with two sub-phases per K-step (k 0-31, k 32-63) and 6 DMA pieces per wave (2 of A, 4 of B)
    issue(0, 0, kHalf);
    if (nk > 1) {
      issue(1, 1, kHalf);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (tid < BN2) { par[tid] = pb; par[BN2 + tid] = psc; par[2 * BN2 + tid] = psh; }
    __builtin_amdgcn_s_barrier();
    stamp();
    ldB(smem, b0, c0);
    ldAh(smem, a0, c0);
    for (int t = 0; t < nk; ++t) {
      const char* st = smem + (t & 1) * STAGE2;
      __builtin_amdgcn_sched_barrier(0);
      mma_h(a0, b0, 0);                                 // H0
      __builtin_amdgcn_sched_barrier(0);
      ldB(st, b1, c1);
      ldAh(st, a1, c1);
      __builtin_amdgcn_sched_barrier(0);
      mma_h(a0, b0, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_h(a1, b1, 0);                                 // H1
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (dma_early && t + 2 < nk) issue(t + 2, t & 1, kHalf);
        const char* sn = smem + ((t + 1) & 1) * STAGE2;
        ldB(sn, b0, c0);
        ldAh(sn, a0, c0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_h(a1, b1, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (!dma_early && t + 2 < nk) issue(t + 2, t & 1, kHalf);
    }
    stamp();
    // epilogue: the whole tile's, on a [128][256] image
    f32x4 qb[4], qs[4], qt[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int c = wn * 64 + ni * 16 + fq * 4;
      qb[ni] = *reinterpret_cast<const f32x4*>(par + c);
      qs[ni] = *reinterpret_cast<const f32x4*>(par + BN2 + c);
      qt[ni] = *reinterpret_cast<const f32x4*>(par + 2 * BN2 + c);
    }
    lds_barrier();
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int row = wm * 64 + mi * 16 + fr;
      char* rowp = smem + row * (BN2 * 2);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        f32x4 v = acc[mi][ni] + qb[ni];
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        v = v * qs[ni] + qt[ni];
        uint2 pk;
        pk.x = pack2t<F16>(v[0], v[1]);
        pk.y = pack2t<F16>(v[2], v[3]);
        const int u8 = (wn * 16 + ni * 4 + fq) ^ (fr << 1);
        *reinterpret_cast<uint2*>(rowp + u8 * 8) = pk;
      }
    }
    lds_barrier();
    {
      const int r0 = tid >> 5, cc = tid & 31;
      const char* src = smem + r0 * (BN2 * 2) + ((cc ^ (r0 & 15)) << 4);
      bf16_t* dst = CKB ? p.C + (int64_t)((n0 >> 6) + (cc >> 3)) * p.cblk + (int64_t)(m0 + r0) * 64 + (cc & 7) * 8
                        : p.C + (int64_t)(m0 + r0) * p.ldc + n0 + cc * 8;
      const int64_t ldc = CKB ? 64 : p.ldc;
      const int rows_left = p.M - m0 - r0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (16 * i < rows_left) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(src + i * 16 * (BN2 * 2));
          __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + (int64_t)(16 * i) * ldc));
        }
      }
    }
    // (the whole tile's statistics are per 128-row half: this half tile IS half (m0 >> 7) & 1 of its parent tile, same rows, same order, same slot)
    if (stats) tile_stats((m0 / BM2) * BM2, m0 & 128, 128, n0);
    stamp();
    }
  }
  if (p.clk && tid == 0 && blockIdx.x < 4096) {
    p.clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - clk_c0;
    p.clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  }
}

// Combine the per-tile partials of a segment in tile order: mean (mode 1) or mean | std (mode 2).
__global__ __launch_bounds__(256) void colstats_finish_kernel(const float* __restrict__ part, int M, int N, int T, int mode,
                                                             float* __restrict__ out) {
  const int b = blockIdx.x, n = blockIdx.y * 256 + threadIdx.x;
  if (n >= N) return;
  const int64_t nbm = (M + BM2 - 1) / BM2;
  const int r_lo = b * T, r_hi = (b + 1) * T - 1;
  float s = 0.f, q = 0.f;
  for (int t = r_lo / BM2; t <= r_hi / BM2; ++t) {
    const int slot = b - (t * BM2) / T;
    const float* ps = part + ((int64_t)(t * 2) * 3 + slot) * N + n;
    s += ps[0];
    s += ps[3 * (int64_t)N];
    if (mode == 2) {
      const float* pq = ps + nbm * 6 * N;
      q += pq[0];
      q += pq[3 * (int64_t)N];
    }
  }
  const float invT = 1.0f / (float)T;
  const float mean = s * invT;
  if (mode == 1) {
    out[(int64_t)b * N + n] = mean;
  } else {
    out[(int64_t)b * 2 * N + n] = mean;
    out[(int64_t)b * 2 * N + N + n] = sqrtf(fmaxf(q * invT - mean * mean, 1e-12f));
  }
}

int g_gemm_variant = -1;   // -1: read SDK_GEMM_VARIANT once; 1 = force the 128^2 kernel, 2 = prefer 256^2

}  // namespace

extern "C" size_t sdk_conv_gemm_stats_bytes(int M, int N, int mode) {
  if (M <= 0 || N <= 0 || mode < 1 || mode > 2) return 0;
  return (size_t)((M + BM2 - 1) / BM2) * 6 * (size_t)N * sizeof(float) * (size_t)mode;
}

extern "C" int sdk_conv_gemm_stats_fusable(int M, int N, int T) {
  return ((g_gemm_variant < 0 ? 2 : g_gemm_variant) & 15) != 1 && N % BN2 == 0 && M >= BM2 && T >= 128;
}

extern "C" int sdk_colstats_finish(sdk_ctx* ctx, const float* stats_part, int M, int N, int T, int mode, float* out, void* stream) {
  SDK_REQUIRE(ctx && stats_part && out, "sdk_colstats_finish: null argument");
  SDK_REQUIRE(M > 0 && T > 0 && M % T == 0 && (mode == 1 || mode == 2), "sdk_colstats_finish: bad shape");
  hipLaunchKernelGGL(colstats_finish_kernel, dim3(M / T, ceil_div(N, 256)), dim3(256), 0, (hipStream_t)stream, stats_part, M, N, T, mode, out);
  SDK_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdk_set_gemm_variant(int v) {   // tuning knob: 1 = 128^2 register-staged, 2 = 256^2 LDS-DMA (default)
  g_gemm_variant = v;
  return 0;
}

extern "C" int sdk_conv_gemm(sdk_ctx* ctx, const sdk_conv_gemm_args* a, void* stream) {
  SDK_REQUIRE(ctx && a, "sdk_conv_gemm: null ctx/args");
  SDK_REQUIRE(a->A && a->W, "sdk_conv_gemm: A and W are required");
  SDK_REQUIRE(a->M > 0 && a->N > 0 && a->N % BN == 0, "sdk_conv_gemm: N=%d must be a positive multiple of %d", a->N, BN);
  const int pack = a->tap_pack;
  if (pack) {
    SDK_REQUIRE(pack > 0 && pack % 8 == 0 && a->taps > 1 && !a->A2 && a->Cin == pack, "sdk_conv_gemm: tap_pack=%d needs taps > 1, a multiple of 8 channels per tap, Cin == tap_pack and no A2", pack);
  } else
  SDK_REQUIRE(a->Cin > 0 && a->Cin % BK == 0, "sdk_conv_gemm: Cin=%d must be a multiple of %d", a->Cin, BK);
  SDK_REQUIRE(a->taps >= 1 && (a->taps & 1), "sdk_conv_gemm: taps=%d must be odd", a->taps);
  SDK_REQUIRE(a->T > 0 && a->M % a->T == 0, "sdk_conv_gemm: M=%d must be a multiple of T=%d", a->M, a->T);
  SDK_REQUIRE((int64_t)a->N * a->taps * a->Cin < (1ll << 31), "sdk_conv_gemm: weight matrix of %d x %d elements exceeds 2^31", a->N, a->taps * a->Cin);
  SDK_REQUIRE(a->taps == 1 || (a->taps / 2) * a->dil < a->T, "sdk_conv_gemm: segment of T=%d frames shorter than the conv halo %d", a->T, (a->taps / 2) * a->dil);
  const bool a_kb = (a->flags & SDK_GEMM_A_KBLOCKED) != 0, c_kb = (a->flags & SDK_GEMM_C_KBLOCKED) != 0, f16 = (a->flags & SDK_GEMM_F16) != 0;
  if (a_kb) SDK_REQUIRE(a->taps == 1 && !pack && !a->A2, "sdk_conv_gemm: a K-blocked A operand needs taps == 1, no tap packing and no A2");
  else
  SDK_REQUIRE(a->lda % 8 == 0 && a->lda >= a->Cin, "sdk_conv_gemm: lda=%lld must be >= Cin and a multiple of 8", (long long)a->lda);
  SDK_REQUIRE(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->W % 16) == 0, "sdk_conv_gemm: A/W must be 16-byte aligned");
  SDK_REQUIRE(a->C || a->C32 || a->S, "sdk_conv_gemm: no output requested");
  if (a->C && !c_kb) SDK_REQUIRE(a->ldc % 8 == 0 && a->ldc >= a->N && ((uintptr_t)a->C % 16) == 0, "sdk_conv_gemm: bad C/ldc");
  if (c_kb) SDK_REQUIRE(a->C && ((uintptr_t)a->C % 16) == 0, "sdk_conv_gemm: a K-blocked output needs C (16-byte aligned)");
  if (a->S) SDK_REQUIRE(a->X2 && a->lds % 8 == 0 && a->ldx2 % 8 == 0 && ((uintptr_t)a->S % 16) == 0 && ((uintptr_t)a->X2 % 16) == 0, "sdk_conv_gemm: S needs X2 and 16-byte aligned rows");
  if (a->C32) SDK_REQUIRE(a->ldc32 >= a->N, "sdk_conv_gemm: bad ldc32");
  if (a->ubias) SDK_REQUIRE(a->ldub >= a->N, "sdk_conv_gemm: bad ldub");

  if (g_gemm_variant < 0) {
    const char* e = getenv("SDK_GEMM_VARIANT");
    g_gemm_variant = e ? atoi(e) : 2;
  }
  if (sdk_lds_optin(ctx, (const void*)conv_gemm256_kernel<false>, LDS2_TOTAL)) return 1;
  if (sdk_lds_optin(ctx, (const void*)conv_gemm256_kernel<true>, LDS2_TOTAL)) return 1;
  if (sdk_lds_optin(ctx, (const void*)conv_gemm256_kernel<false, true>, LDS2_TOTAL)) return 1;
  if (sdk_lds_optin(ctx, (const void*)conv_gemm256_kernel<false, false, true>, LDS2_TOTAL)) return 1;
  if (sdk_lds_optin(ctx, (const void*)conv_gemm256_kernel<true, false, true>, LDS2_TOTAL)) return 1;
  if (sdk_lds_optin(ctx, (const void*)conv_gemm256_kernel<false, true, true>, LDS2_TOTAL)) return 1;
  Params p;
  p.A = (const bf16_t*)a->A; p.lda = a->lda; p.W = (const bf16_t*)a->W;
  p.C = (bf16_t*)a->C; p.ldc = a->ldc; p.C32 = a->C32; p.ldc32 = a->ldc32;
  p.bias = a->bias; p.scale = a->scale; p.shift = a->shift; p.ubias = a->ubias; p.ldub = a->ldub;
  p.X2 = (const bf16_t*)a->X2; p.ldx2 = a->ldx2; p.S = (bf16_t*)a->S; p.lds = a->lds;
  p.M = a->M; p.N = a->N; p.Cin = a->Cin; p.taps = a->taps; p.dil = a->dil; p.T = a->T; p.flags = a->flags;
  p.tune = g_gemm_variant / 16;
  p.stats_part = nullptr; p.stats_mode = 0;
  p.A2 = (const bf16_t*)a->A2; p.lda2 = a->lda2;
  p.tap_pack = pack;
  p.ablk = a_kb ? (int64_t)a->M * 64 : 0;
  p.cblk = c_kb ? (int64_t)a->M * 64 : 0;
  p.clk = (unsigned long long*)ctx->gemm_clk_ptr;
  p.stamps = (unsigned long long*)ctx->gemm_stamps_ptr;
  if (a->A2) SDK_REQUIRE(a->lda2 % 8 == 0 && a->lda2 >= a->Cin && ((uintptr_t)a->A2 % 16) == 0, "sdk_conv_gemm: bad A2/lda2");

  const double kk = pack ? (double)(((a->taps * pack + BK - 1) / BK) * BK) : (double)a->taps * a->Cin;      // K as launched
  // the 256^2 kernel covers the plain layer shape (bias / ReLU / BN affine -> bf16, optional column statistics);
  // fp32 output, residual sum, per-segment bias, tanh and the A2 addend stay with the 128^2 kernel
  const bool use256 = !a_kb && ((g_gemm_variant & 15) != 1 || c_kb) && a->N % BN2 == 0 && a->M >= BM2 && !a->A2 && a->C && !a->C32 && !a->S &&
                      !a->ubias && !(a->flags & SDK_GEMM_TANH) && (a->taps == 1 || a->T >= 64) &&
                      (uint64_t)a->M * (uint64_t)a->lda * 2u < (1ull << 32);   // the 256^2 kernel addresses A by 32-bit byte offsets
  ProfScope ps(ctx, stream, use256 ? SDK_K_CONV_GEMM256 : SDK_K_CONV_GEMM, 2.0 * a->M * a->N * kk,
               2.0 * a->M * a->Cin + 2.0 * a->N * kk + (a->C ? 2.0 : 0.0) * a->M * a->N + (a->C32 ? 4.0 : 0.0) * a->M * a->N +
                   (a->S ? 4.0 : 0.0) * a->M * a->N);
  // K-blocked forms ([cols / 64][M][64], SDK_GEMM_*_KBLOCKED): the A read is the 128^2 kernel's (register-staged: any address form), the C write the
  // 256^2 kernel's copy-out sweep of its LDS image
  if (c_kb) SDK_REQUIRE(use256 && a->taps == 1, "sdk_conv_gemm: a K-blocked output needs the plain-layer shape of the 256^2 kernel (taps == 1, N %% 256 == 0, M >= 256, bf16 output only)");
  if (a->stats_mode) {
    SDK_REQUIRE(a->stats_mode == 1 || a->stats_mode == 2, "sdk_conv_gemm: stats_mode=%d", a->stats_mode);
    SDK_REQUIRE(use256 && a->T >= 128 && a->stats_part && a->C, "sdk_conv_gemm: fused column statistics need the 256^2 kernel (N %% 256 == 0, M >= 256), T >= 128 and a bf16 output");
    p.stats_part = a->stats_part; p.stats_mode = a->stats_mode;
  }
  if (use256) {
    const int ntiles = (a->N / BN2) * ceil_div(a->M, BM2);
    const int cus = ctx->num_cu > 0 ? (ctx->num_cu / 8) * 8 : 256;
    const int grid = (p.tune & 8) ? ntiles : (ntiles < cus ? ntiles : cus);       // tune bit 3: one workgroup per tile (A/B)
    // (tune bit 4 selected round 3's v3 kernel, the overlapped tile boundary: bit-identical, 6-7 % fewer cycles per K = 1024 tile, the same
    // wall time in interleaved A/B (0.97-1.01x) - the saved cycles came back as a lower clock; removed in round 5, see DESIGN.md and git history)
    if (f16) {
      void (*kern)(Params) = conv_gemm256_kernel<false, false, true>;
      if (c_kb) kern = conv_gemm256_kernel<false, true, true>;
      else if (p.taps > 1) kern = conv_gemm256_kernel<true, false, true>;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(NT2), LDS2_TOTAL, (hipStream_t)stream, p);
    } else if (c_kb)
      hipLaunchKernelGGL((conv_gemm256_kernel<false, true>), dim3(grid), dim3(NT2), LDS2_TOTAL, (hipStream_t)stream, p);
    else
      hipLaunchKernelGGL(p.taps > 1 ? conv_gemm256_kernel<true> : conv_gemm256_kernel<false>, dim3(grid), dim3(NT2), LDS2_TOTAL, (hipStream_t)stream, p);
  } else {
    hipLaunchKernelGGL(f16 ? conv_gemm_kernel<true> : conv_gemm_kernel<false>, dim3((a->N / BN) * ceil_div(a->M, BM)), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
  }
  SDK_LAUNCH_CHECK();
  return 0;
}
