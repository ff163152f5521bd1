// k4, k = 1 (argmax) fast path: segments x profiles cosine affinity without per-score bookkeeping.
//
// Rows consumed at speaker_detection:1085-1127 are "best enrolled profile + score" per segment.  The general kernel
// (scoring.hip) keeps a sorted candidate list per lane and pays 5 vector ops per score, which - not the matrix
// pipe - bounds it.  Here a lane (one segment, one half of a 32-profile tile = 16 accumulator registers) keeps
//     ROW maxima     T_t  = max over the 16 registers of tile t      (v_max3 tree, 8 ops per 16 scores)
//     COLUMN maxima  C_r  = max over all tiles of register r         (16 v_max per 16 scores)
// and a sorted top-4 of the tagged T_t (tile index in the low 10 mantissa bits; 5 ops per tile): 29 vector ops per
// 16 scores instead of 80+.  Every score x[t][r] sits in exactly one row and one column, hence
//     x[t][r] <= min(T_t, C_r),   and any score outside (top-3 rows) x (top-3 columns) is <= max(T_(4), C_(4)) =: u.
// The exact pass therefore re-scores, in fp32 with the one shared dot routine (scoring_exact.hpp), the intersections
// (t_i, r_j), i, j < 3, whose bound min(T_i, C_j) can still reach the best coarse score minus the rounding margin, and
// certifies the winner against max(u, cut) + eps exactly as the general path does; rows it cannot certify (~1 %) take the
// exact rescan, batched four rows per profile sweep.  Results equal an fp32 full scan: ties -> lowest profile index.
//
// Decomposition: ONE workgroup per CU, a contiguous range of (segment group, profile stage) units, so every CU gets
// the same amount of matrix work whatever N and P are (a group's sweep may be split over up to 3 workgroups = "parts";
// each part writes its own row/column record).  Profile stages of TPS tiles stream through a 3-deep LDS ring filled by
// LDS-DMA; the ring runs ahead across group boundaries.
#include <stdlib.h>

#include "common.hpp"
#include "scoring_exact.hpp"

using namespace sdk_exact;

namespace {

constexpr int KS = D / 16;             // 12 MFMA k-steps (32x32x16)
constexpr int PT = 32;                 // profiles per MFMA tile
constexpr int PROWB = 384;             // LDS bytes per profile row (24 chunks of 16 B, XOR-swizzled)
constexpr int TILE_BYTES = PT * PROWB; // 12 KiB
constexpr int MAXP = 3;                // record slots per segment = parts of a group's sweep (the geometry gives <= 3: tests/test_cabi_cpu.py)
constexpr uint32_t TMASK = 0x3ffu;     // tile tag: 10 bits (<= 1024 tiles = 32768 profiles per part)
constexpr uint32_t CMASK = 0xfu;       // column tag: the accumulator register
constexpr float MASKED = -4.0f;        // stands for "no such profile" (cosines are >= -1 - eps); finite so tags never make a NaN
constexpr float EMPTY = -8.0f;
constexpr int AFF_DEFAULT_PEN = 0;       // boundary penalty in stages (sdk_set_option "affinity_boundary_penalty").  Measured (tools/aff_pen_ab.py, interleaved): config #3
                                         // 49.6 us at 0, 47.8-48.7 at 1-4 (-3 %); 125k x 10k unchanged; 20k x 2k 41.9 -> 43.6-45.6 (+4-9 %): not a robust win, off

struct Geom {            // host-computed, identical on both sides
  int ngroups, nst, G, segs;   // segment groups, stages per group, workgroups, segments per group
  long long U;                 // units = ngroups * nst
  int pen;                     // cost of starting a portion in a NEW group, in stages (fragment reload + a first stage that waits for it)
};

// Work ranges balanced by COST, not by units (round 3; timeline profiles/r03_aff_timeline_config3.txt: with equal unit counts a workgroup whose
// range crossed a group boundary ended at 46-50 us where the others ended at 36-38).  Every group is given nst + pen VIRTUAL units, the first
// pen of which are the switch cost and carry no work; the virtual axis is cut evenly and mapped back: a workgroup that starts a second group gets
// fewer stages.  pen = 0 is the plain [i U / G, (i + 1) U / G) split.  Host and device share these two functions.
__host__ __device__ inline long long geom_real(const Geom& g, long long v) {
  const long long VU = g.nst + g.pen;
  const long long b = v / VU, o = v - b * VU;
  return b * g.nst + (o > g.pen ? o - g.pen : 0);
}
__host__ __device__ inline void geom_range(const Geom& g, long long i, long long& u0, long long& u1) {
  const long long V = (long long)g.ngroups * (g.nst + g.pen);
  u0 = geom_real(g, (i * V) / g.G);
  u1 = geom_real(g, ((i + 1) * V) / g.G);
}
// first workgroup whose range reaches into group b: smallest j with real(floor((j + 1) V / G)) > b nst  <=>  floor((j + 1) V / G) > b VU + pen
__host__ __device__ inline long long geom_first_wg(const Geom& g, long long b) {
  const long long VU = g.nst + g.pen, V = (long long)g.ngroups * VU, X = b * VU + g.pen;
  return ((X + 1) * g.G + V - 1) / V - 1;
}


// ---- hand-pipelined fragment reads (round 4) ---------------------------------------------------------------------------------------------------
// Left to hipcc, a tile's twelve profile-fragment reads are issued one or two at a time directly in front of the MFMAs that need them, each pair
// behind an `s_waitcnt lgkmcnt(0)` (the .s of both coarse kernels showed 7-8 such waits per tile): the wave exposes an LDS round trip every one or
// two MFMAs, and the stage loop ran at 58-65 % of the matrix pipe whatever was done to the vector work beside it (pipelined reduction, static
// priority, staggered waves: all measured, all null).  Here the reads are inline asm in a ring of RD registers, RD steps ahead of their MFMAs, with
// counted waits; the last steps of a tile already fetch the next tile's first fragments (PREF), which then arrive under the tile's reduction.
// Every wait is followed by sched_barrier(0): hipcc hoists register-only MFMAs across an inline-asm s_waitcnt (cdna_hip_programming.md rule 18).
// The kernels have no static LDS, so the dynamic array starts at LDS address 0 and `sqa` are plain byte offsets.
constexpr int RD = 4;
template <int OFF>
__device__ __forceinline__ void lds_rd128(bf16x8& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lgkm_wait_pinned() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// fragments 0..RD-1 of tile TT (byte offset TT * TILE_BYTES inside the stage): the stage's prologue, right behind its barrier
template <int TT>
__device__ __forceinline__ void tile_prologue(const uint32_t (&sqa)[4], bf16x8 (&ring)[RD]) {
  lds_rd128<TT * TILE_BYTES>(ring[0], sqa[0]);
  lds_rd128<TT * TILE_BYTES>(ring[1], sqa[1]);
  lds_rd128<TT * TILE_BYTES>(ring[2], sqa[2]);
  lds_rd128<TT * TILE_BYTES>(ring[3], sqa[3]);
}
// step KSI of tile TT for NB segment blocks: wait for fragment KSI, its NB MFMAs, then the read RD steps ahead into the slot just consumed
template <int TT, int NB, bool PREF, int KSI>
struct TileStep {
  static __device__ __forceinline__ void run(const uint32_t (&sqa)[4], const bf16x8 (&bfrag)[2][KS], f32x16 (&acc)[2], bf16x8 (&ring)[RD]) {
    constexpr int LAST = PREF ? KS + RD - 1 : KS - 1;                         // index of the last read of this tile's sequence (PREF: + the next tile's first RD)
    constexpr int YOUNGER = (LAST - KSI) < (RD - 1) ? (LAST - KSI) : (RD - 1); // reads issued after fragment KSI that may stay in flight
    lgkm_wait_pinned<YOUNGER>();
#pragma unroll
    for (int sb = 0; sb < NB; ++sb) acc[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[KSI % RD], bfrag[sb][KSI], acc[sb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NX = KSI + RD;
    if constexpr (NX < KS) lds_rd128<TT * TILE_BYTES + (NX >> 2) * 128>(ring[NX % RD], sqa[NX & 3]);
    else if constexpr (PREF) lds_rd128<(TT + 1) * TILE_BYTES + ((NX - KS) >> 2) * 128>(ring[NX % RD], sqa[(NX - KS) & 3]);
    if constexpr (KSI + 1 < KS) TileStep<TT, NB, PREF, KSI + 1>::run(sqa, bfrag, acc, ring);
  }
};
static_assert(KS % RD == 0, "the ring slot of fragment ks (ks % RD) and its address register (ks & 3) line up across tiles");

template <int DEPTH>
__device__ __forceinline__ void insert_sorted(float x, float* m) {
  float n[DEPTH];
#pragma unroll
  for (int q = DEPTH - 1; q >= 1; --q) n[q] = __builtin_amdgcn_fmed3f(x, m[q - 1], m[q]);
  n[0] = fmaxf(x, m[0]);
#pragma unroll
  for (int q = 0; q < DEPTH; ++q) m[q] = n[q];
}

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// WAVES waves x SEGB blocks of 32 segments; TPS tiles per LDS stage; NSTAGE stages.  (Rounds 3-4 also built, measured level or behind and removed: the
// reduction of tile t-1 dealt out by hand under tile t's MFMAs - 4-6 % slower, the two waves of a SIMD overlap those phases already -, static wave
// priorities, hand-pipelined fragment reads in this kernel, 4 tiles per stage, a late ring refill; docs/rounds/r01-r04_design_history.md 5.11.)
template <int WAVES, int SEGB, int TPS, int NSTAGE>
__global__ __launch_bounds__(WAVES * 64, 2) void aff_rowcol_kernel(const bf16_t* __restrict__ Eb, const bf16_t* __restrict__ Pb,
                                                               int N, int P, Geom gm, float* __restrict__ stats,
                                                               int32_t* __restrict__ part_base, int32_t* __restrict__ part_cnt,
                                                               int32_t* __restrict__ flag_count, unsigned long long* __restrict__ dbg) {
  // diagnostic time stamps (100 MHz wall clock) of wave 0: [0] start, [1] count, then one per event; never set in production
  int nstamp = 2;
  auto stamp = [&]() {
    if (dbg && threadIdx.x == 0 && nstamp < 62) dbg[blockIdx.x * 64 + nstamp++] = __builtin_amdgcn_s_memrealtime();
  };
  if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 64] = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) *flag_count = 0;          // the exact pass (next launch) counts its uncertain rows here
  const unsigned long long clk0 = dbg ? __builtin_amdgcn_s_memtime() : 0;
  constexpr int STAGE_BYTES = TPS * TILE_BYTES;
  constexpr int DMA_PER_STAGE = STAGE_BYTES / 1024;
  static_assert(DMA_PER_STAGE % WAVES == 0, "stage must split evenly over the waves");
  constexpr int DPW = DMA_PER_STAGE / WAVES;
  constexpr int SEGS = WAVES * SEGB * 32;
  extern __shared__ __attribute__((aligned(16))) char sP[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  long long u0, u1;
  geom_range(gm, blockIdx.x, u0, u1);
  const int nst = gm.nst;
  const int ntiles = (P + PT - 1) / PT;

  // DMA assignment: a stage is DMA_PER_STAGE wave-instructions of 1 KiB; wave w issues instructions w*DPW .. w*DPW+DPW-1
  int drow[DPW], dsrc[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int id = (wid * DPW + i) * 64 + lane;
    drow[i] = id / 24;
    const int pos = id - drow[i] * 24;
    dsrc[i] = ((pos & ~7) | ((pos & 7) ^ ((drow[i] >> 1) & 7))) * 8;    // source chunk (elements)
  }
  int s_issue = (int)(u0 % nst), k_issue = 0;
  auto issue = [&]() {                                               // next stage of the running sequence
    char* st = sP + (k_issue % NSTAGE) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      const int src = dsrc[i];
      int pr = s_issue * (TPS * PT) + drow[i];
      pr = pr < P ? pr : P - 1;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(Pb + (int64_t)pr * D + src),
                                       (void __attribute__((address_space(3)))*)(st + (wid * DPW + i) * 1024), 16, 0, 0);
    }
    ++k_issue;
    if (++s_issue == nst) s_issue = 0;
  };

  int b = (int)(u0 / nst), s = (int)(u0 % nst);
  // slot of the first portion = number of earlier workgroups that share its group
  int slot;
  {
    slot = (int)(blockIdx.x - geom_first_wg(gm, b));
  }

  bf16x8 bfrag[SEGB][KS];
  float C[SEGB][16], tl[SEGB][4];
  int ltile = 0;
  auto reduce = [&](const f32x16* ac, int tag) {
#pragma unroll
    for (int sb = 0; sb < SEGB; ++sb) {
      const f32x16& a = ac[sb];
      const float m0 = max3(a[0], a[1], a[2]), m1 = max3(a[3], a[4], a[5]), m2 = max3(a[6], a[7], a[8]);
      const float m3 = max3(a[9], a[10], a[11]), m4 = max3(a[12], a[13], a[14]);
      const float T = fmaxf(max3(m0, m1, m2), max3(m3, m4, a[15]));
#pragma unroll
      for (int r = 0; r < 16; ++r) C[sb][r] = fmaxf(C[sb][r], a[r]);
      insert_sorted<4>(__uint_as_float((__float_as_uint(T) & ~TMASK) | (uint32_t)tag), tl[sb]);
    }
  };
  auto begin_portion = [&]() {
#pragma unroll
    for (int sb = 0; sb < SEGB; ++sb) {
      const int seg = b * SEGS + (wid * SEGB + sb) * 32 + col;
      const int seg_c = seg < N ? seg : N - 1;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        bfrag[sb][ks] = *reinterpret_cast<const bf16x8*>(Eb + (int64_t)seg_c * D + ks * 16 + h * 8);
#pragma unroll
      for (int r = 0; r < 16; ++r) C[sb][r] = EMPTY;
#pragma unroll
      for (int q = 0; q < 4; ++q) tl[sb][q] = EMPTY;
    }
    ltile = 0;
    if (tid == 0) {
      if (slot < MAXP) part_base[b * MAXP + slot] = s * TPS;    // (a 4th part cannot happen - plan_geometry, tests/test_cabi_cpu.py - and would
                                                                //  not go unnoticed: part_cnt > MAXP sends the group's rows to the exact rescan)
    }
  };
  auto end_portion = [&](bool group_done) {
#pragma unroll
    for (int sb = 0; sb < SEGB; ++sb) {
      float cl[4] = {EMPTY, EMPTY, EMPTY, EMPTY};
#pragma unroll
      for (int r = 0; r < 16; ++r)
        insert_sorted<4>(__uint_as_float((__float_as_uint(C[sb][r]) & ~CMASK) | (uint32_t)r), cl);
      const int seg = b * SEGS + (wid * SEGB + sb) * 32 + col;
      if (seg < N && slot < MAXP) {
        float* dst = stats + (((int64_t)seg * MAXP + slot) * 2 + h) * 8;
        f32x4 tv, cv;
#pragma unroll
        for (int q = 0; q < 4; ++q) { tv[q] = tl[sb][q]; cv[q] = cl[q]; }
        *reinterpret_cast<f32x4*>(dst) = tv;
        *reinterpret_cast<f32x4*>(dst + 4) = cv;
      }
    }
    if (group_done && tid == 0) part_cnt[b] = slot + 1;
  };

  if (u0 >= u1) return;
  constexpr int AHEAD = NSTAGE - 1;                // stages in flight
#pragma unroll
  for (int a = 0; a < AHEAD; ++a)
    if (u0 + a < u1) issue();
  const int rsw = (col >> 1) & 7;
  int aoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) aoff[q] = col * PROWB + (((2 * q + h) ^ rsw) << 4);
  auto mask_partial = [&](int tile, f32x16* acc) {
    if ((tile + 1) * PT > P) {                     // last, partial tile: rows past the last profile never count
#pragma unroll
      for (int sb = 0; sb < SEGB; ++sb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (tile * PT + (r & 3) + 8 * (r >> 2) + 4 * h >= P) acc[sb][r] = MASKED;
    }
  };
  const char* sq[4];
  int k = 0;
  long long u = u0;
  while (u < u1) {                                 // one portion = this workgroup's share of one group's sweep
    const long long gend = (long long)(b + 1) * nst;
    const long long uend = gend < u1 ? gend : u1;
    // The segment fragments are ordinary loads issued OUTSIDE the stage loop, so the compiler drains them once here and
    // the stage loop keeps its counted waits (a load inside the loop makes it wait vmcnt(0) before every tile).
    begin_portion();
    stamp();
    for (; u < uend; ++u, ++k) {
      // stage u must have landed; the stages issued after it (at most AHEAD - 1) may stay in flight
      const long long after = u1 - 1 - u < AHEAD - 1 ? u1 - 1 - u : AHEAD - 1;
      if (after >= 2) {
        if constexpr (DPW == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (DPW == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else if (after == 1) {
        if constexpr (DPW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if constexpr (DPW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                 // stage k landed for everyone; the buffer of stage k-1 is free
      stamp();
      if (u + AHEAD < u1) issue();
      // Fragment addresses: chunk c = 2 ks + h of this lane's row sits at 16 * ((c & ~7) | ((c & 7) ^ rsw)); its low three
      // bits depend only on ks & 3, so four per-lane offsets + compile-time immediates (tile, ks >> 2) cover all 12 reads of
      // every tile.  (Left to the compiler, the XOR was re-derived per read: ~50 of the ~118 vector instructions per tile.)
#pragma unroll
      for (int q = 0; q < 4; ++q) sq[q] = sP + (k % NSTAGE) * STAGE_BYTES + aoff[q];
      {
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
          const int tile = s * TPS + tt;
          if (tile < ntiles) {                         // wave-uniform
            f32x16 acc[SEGB];
#pragma unroll
            for (int sb = 0; sb < SEGB; ++sb)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[sb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              const bf16x8 a = *reinterpret_cast<const bf16x8*>(sq[ks & 3] + tt * TILE_BYTES + (ks >> 2) * 128);
#pragma unroll
              for (int sb = 0; sb < SEGB; ++sb) acc[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[sb][ks], acc[sb], 0, 0, 0);
            }
            mask_partial(tile, acc);
            reduce(acc, ltile);
            ++ltile;
          }
        }
      }
      ++s;
    }
    stamp();
    end_portion(s == nst);
    if (s == nst) { ++b; s = 0; slot = 0; }         // the next portion starts a new group
  }
  stamp();
  if (dbg && threadIdx.x == 0) {
    dbg[blockIdx.x * 64 + 1] = nstamp;
    dbg[blockIdx.x * 64 + 63] = __builtin_amdgcn_s_memtime() - clk0;     // shader cycles between the first and this stamp
  }
}


// ---- round 4: the BLOCK plan (short sweeps, e.g. config #3: 100 000 x 1 000) ------------------------------------------------------------------
// What the range plan above loses at config #3 (profiles/r03_aff_timeline_config3.txt): a portion's 192 KB of segment fragments
// arrive at ~35-40 GB/s per CU (the Infinity-Cache rate) = 4.8 us at kernel start and ~6 us at every group boundary, three workgroups in four
// cross a boundary (16 stages per group, 12.25 per workgroup), and the fragments of a split group are fetched by every workgroup that takes part
// in its sweep (105 MB through the fabric against 45 MB).  Here NO workgroup changes its segments: the unit of work is a BLOCK of 32 segments
// with its whole sweep, blocks are dealt to (workgroup, wave, slot) - every workgroup sweeps all stages exactly once, in step with all others -
// and the blocks that do not divide evenly are swept in PARTS (thirds / halves of the stage range) by waves with a free second slot:
//   main    workgroup g owns blocks [g q, (g + 1) q), q a multiple of 4 (8 or 12): slot j -> wave j % 8, sub-slot j / 8: every SIMD carries q / 4
//           blocks (at q = 12: waves 0-3 two blocks, waves 4-7 one), so a stage costs q / 16 of a full workgroup's
//   extra   leftover block i, part p = item t = i PARTS + p -> workgroup t % G, its (t / G)-th free slot, active for stages [e0, e1) only
// Records are per BLOCK (segs = 32 for the exact pass): a main block has one part (slot 0, base tile 0), a leftover block PARTS parts (<= MAXP).
// The host picks this plan when its estimated cost (block-stages on the busiest SIMD) is under the range plan's (plan_blocks).
struct BlockPlan {
  int q, G, nst, parts, items;      // main blocks per workgroup, workgroups, stages per sweep, parts per leftover block, leftover items
  int main_parts;                   // records a WHOLE sweep writes (round 5): 1 = one record at the end; 2 / 3 = the row / column maxima are flushed and
                                    // reset at the stage boundaries p nst / main_parts, so the certificate is that of sweeps a half / a third as long
};

// the blocks of wave `wid` of workgroup `g` (host and device share this; tests/test_cabi_cpu.py replays it through sdk_affinity_block_plan_wave)
__host__ __device__ inline void block_slots(const BlockPlan& pl, int g, int wid, int& blk0, int& blk1, int& e0, int& e1, int& slot1, int& cnt1) {
  blk0 = g * pl.q + wid;
  blk1 = -1; e0 = 0; e1 = 0; slot1 = 0; cnt1 = 1;
  const int j1 = 8 + wid;
  if (j1 < pl.q) {
    blk1 = g * pl.q + j1; e0 = 0; e1 = pl.nst;
  } else {
    const long long t = (long long)(j1 - pl.q) * pl.G + g;
    if (t < pl.items) {
      const int i = (int)(t / pl.parts), pp = (int)(t - (long long)i * pl.parts);
      blk1 = pl.q * pl.G + i;
      e0 = (int)((long long)pp * pl.nst / pl.parts);
      e1 = (int)((long long)(pp + 1) * pl.nst / pl.parts);
      slot1 = pp; cnt1 = pl.parts;
    }
  }
}

template <int WAVES, int TPS, int NSTAGE>
__global__ __launch_bounds__(WAVES * 64, 2) void aff_rowcol_blocks_kernel(const bf16_t* __restrict__ Eb, const bf16_t* __restrict__ Pb,
                                                                      int N, int P, BlockPlan pl, float* __restrict__ stats,
                                                                      int32_t* __restrict__ part_base, int32_t* __restrict__ part_cnt,
                                                                      int32_t* __restrict__ flag_count, unsigned long long* __restrict__ dbg) {
  int nstamp = 2;
  auto stamp = [&]() {
    if (dbg && threadIdx.x == 0 && nstamp < 62) dbg[blockIdx.x * 64 + nstamp++] = __builtin_amdgcn_s_memrealtime();
  };
  if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 64] = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) *flag_count = 0;
  const unsigned long long clk0 = dbg ? __builtin_amdgcn_s_memtime() : 0;
  constexpr int STAGE_BYTES = TPS * TILE_BYTES;
  constexpr int DMA_PER_STAGE = STAGE_BYTES / 1024;
  static_assert(DMA_PER_STAGE % WAVES == 0 && WAVES == 8, "stage must split evenly over the 8 waves");
  constexpr int DPW = DMA_PER_STAGE / WAVES;
  extern __shared__ __attribute__((aligned(16))) char sP[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, h = lane >> 5;
  const int nst = pl.nst;
  const int ntiles = (P + PT - 1) / PT;
  const int g = blockIdx.x;

  // this wave's blocks: slot 0 is always a main block (q >= 8); slot 1 a main block, a leftover item, or nothing
  int blk0, blk1, e0, e1, slot1, cnt1;
  block_slots(pl, g, wid, blk0, blk1, e0, e1, slot1, cnt1);
  const bool has1 = blk1 >= 0;

  int drow[DPW], dsrc[DPW];
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int id = (wid * DPW + i) * 64 + lane;
    drow[i] = id / 24;
    const int pos = id - drow[i] * 24;
    dsrc[i] = ((pos & ~7) | ((pos & 7) ^ ((drow[i] >> 1) & 7))) * 8;
  }
  int s_issue = 0, k_issue = 0;
  auto issue = [&]() {
    char* st = sP + (k_issue % NSTAGE) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      int pr = s_issue * (TPS * PT) + drow[i];
      pr = pr < P ? pr : P - 1;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(Pb + (int64_t)pr * D + dsrc[i]),
                                       (void __attribute__((address_space(3)))*)(st + (wid * DPW + i) * 1024), 16, 0, 0);
    }
    ++k_issue;
    ++s_issue;
  };

  bf16x8 bfrag[2][KS];
  float C[2][16], tl[2][4];
#pragma unroll
  for (int sb = 0; sb < 2; ++sb) {
    const int blk = sb == 0 ? blk0 : (has1 ? blk1 : blk0);        // (no second block: a valid address, never used)
    const int seg = blk * 32 + col;
    const int seg_c = seg < N ? seg : N - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bfrag[sb][ks] = *reinterpret_cast<const bf16x8*>(Eb + (int64_t)seg_c * D + ks * 16 + h * 8);
#pragma unroll
    for (int r = 0; r < 16; ++r) C[sb][r] = EMPTY;
#pragma unroll
    for (int q = 0; q < 4; ++q) tl[sb][q] = EMPTY;
  }
  // A block whose whole sweep this wave makes alone (every main block; a leftover block when parts == 1) writes pl.main_parts records: a whole sweep's
  // fourth-largest row maximum is that of 32 tiles at config #3 - a weaker certificate than the range plan's parts of ~12 tiles (201 instead of
  // 88 rows to rescan, round 4) - so the lists are flushed and reset at the stage boundaries p nst / main_parts
  const int mp = pl.main_parts < 1 ? 1 : (pl.main_parts > MAXP ? MAXP : pl.main_parts);
  const bool whole1 = has1 && cnt1 == 1;
  if (lane == 0) {
    for (int pp = 0; pp < mp; ++pp) {
      part_base[blk0 * MAXP + pp] = (pp * nst / mp) * TPS;
      if (whole1) part_base[blk1 * MAXP + pp] = (pp * nst / mp) * TPS;
    }
    part_cnt[blk0] = mp;
    if (whole1) part_cnt[blk1] = mp;
    else if (has1) {
      part_base[blk1 * MAXP + slot1] = e0 * TPS;
      part_cnt[blk1] = cnt1;                                       // (every part of a leftover block writes the same count)
    }
  }
  constexpr int AHEAD = NSTAGE - 1;
#pragma unroll
  for (int a = 0; a < AHEAD; ++a)
    if (a < nst) issue();
  const int rsw = (col >> 1) & 7;
  int aoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) aoff[q] = col * PROWB + (((2 * q + h) ^ rsw) << 4);
  auto reduce1 = [&](const f32x16& a, int sb, int tag) {
    const float m0 = max3(a[0], a[1], a[2]), m1 = max3(a[3], a[4], a[5]), m2 = max3(a[6], a[7], a[8]);
    const float m3 = max3(a[9], a[10], a[11]), m4 = max3(a[12], a[13], a[14]);
    const float T = fmaxf(max3(m0, m1, m2), max3(m3, m4, a[15]));
#pragma unroll
    for (int r = 0; r < 16; ++r) C[sb][r] = fmaxf(C[sb][r], a[r]);
    insert_sorted<4>(__uint_as_float((__float_as_uint(T) & ~TMASK) | (uint32_t)tag), tl[sb]);
  };
  auto mask1 = [&](int tile, f32x16& acc) {
    if ((tile + 1) * PT > P) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (tile * PT + (r & 3) + 8 * (r >> 2) + 4 * h >= P) acc[r] = MASKED;
    }
  };
  // one record: the sorted top-4 row maxima (tile tags relative to the part) + the top-4 of the 16 column maxima; then the lists start afresh
  auto flush = [&](int sb, int slot) {
    float cl[4] = {EMPTY, EMPTY, EMPTY, EMPTY};
#pragma unroll
    for (int r = 0; r < 16; ++r) insert_sorted<4>(__uint_as_float((__float_as_uint(C[sb][r]) & ~CMASK) | (uint32_t)r), cl);
    const int seg = (sb == 0 ? blk0 : blk1) * 32 + col;
    if (seg < N) {
      float* dst = stats + (((int64_t)seg * MAXP + slot) * 2 + h) * 8;
      f32x4 tv, cv;
#pragma unroll
      for (int q = 0; q < 4; ++q) { tv[q] = tl[sb][q]; cv[q] = cl[q]; }
      *reinterpret_cast<f32x4*>(dst) = tv;
      *reinterpret_cast<f32x4*>(dst + 4) = cv;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) C[sb][r] = EMPTY;
#pragma unroll
    for (int q = 0; q < 4; ++q) tl[sb][q] = EMPTY;
  };
  int pcur = 0, tbase = 0, next_b = mp > 1 ? nst / mp : nst;        // current part of the whole sweeps, its first tile, the stage where the next one starts
  stamp();
  for (int s = 0; s < nst; ++s) {
    if (s == next_b) {                                             // (uniform over the workgroup)
      flush(0, pcur);
      if (whole1) flush(1, pcur);
      ++pcur;
      tbase = s * TPS;
      next_b = (pcur + 1) * nst / mp;
    }
    const int after = nst - 1 - s < AHEAD - 1 ? nst - 1 - s : AHEAD - 1;
    if (after >= 2) {
      if constexpr (DPW == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (after == 1) {
      if constexpr (DPW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    stamp();
    if (s + AHEAD < nst) issue();
    const bool two = has1 && s >= e0 && s < e1;                   // wave-uniform
    const int t0 = s * TPS, t1 = t0 + 1;
    static_assert(TPS == 2, "TileStep<0 / 1>: two tiles per stage");
    const bool has_t1 = t1 < ntiles;                              // wave-uniform
    const uint32_t stage_off = (uint32_t)((s % NSTAGE) * STAGE_BYTES);
    const uint32_t sqa[4] = {stage_off + (uint32_t)aoff[0], stage_off + (uint32_t)aoff[1], stage_off + (uint32_t)aoff[2], stage_off + (uint32_t)aoff[3]};
    bf16x8 ring[RD];
    tile_prologue<0>(sqa, ring);
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    if (two) {
      if (has_t1) TileStep<0, 2, true, 0>::run(sqa, bfrag, acc, ring);
      else TileStep<0, 2, false, 0>::run(sqa, bfrag, acc, ring);
    } else {
      if (has_t1) TileStep<0, 1, true, 0>::run(sqa, bfrag, acc, ring);
      else TileStep<0, 1, false, 0>::run(sqa, bfrag, acc, ring);
    }
    const int tb1 = whole1 ? tbase : e0 * TPS;                      // tag base of slot 1: its part's first tile
    mask1(t0, acc[0]);
    reduce1(acc[0], 0, t0 - tbase);
    if (two) { mask1(t0, acc[1]); reduce1(acc[1], 1, t0 - tb1); }
    if (has_t1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
      if (two) TileStep<1, 2, false, 0>::run(sqa, bfrag, acc, ring);
      else TileStep<1, 1, false, 0>::run(sqa, bfrag, acc, ring);
      mask1(t1, acc[0]);
      reduce1(acc[0], 0, t1 - tbase);
      if (two) { mask1(t1, acc[1]); reduce1(acc[1], 1, t1 - tb1); }
    }
  }
  stamp();
  flush(0, pcur);
  if (has1) flush(1, whole1 ? pcur : slot1);
  stamp();
  if (dbg && threadIdx.x == 0) {
    dbg[blockIdx.x * 64 + 1] = nstamp;
    dbg[blockIdx.x * 64 + 63] = __builtin_amdgcn_s_memtime() - clk0;
  }
}

// Host side of the block plan.  Returns false when the shape does not fit it (too few / too many blocks per workgroup, no free slots for the
// leftover items) or when the range plan is estimated cheaper (long sweeps: config #4's shape - the range plan balances to the stage, the block
// plan only to the block).  Costs in block-stages on the busiest SIMD; the range plan pays ~3.5 stages of fragment reload / refill per workgroup.
bool plan_blocks(int N, int P, int tps, int num_cu, BlockPlan* out, bool force) {
  const int NB = ceil_div(N, 32), G = num_cu;
  const int nst = ceil_div(ceil_div(P, PT), tps);
  if (G <= 0 || NB < 8 * G || nst < 3) return false;
  int q = NB / G;
  q = q >= 12 ? 12 : 8;
  const long long r = (long long)NB - (long long)q * G, free_slots = (long long)(16 - q) * G;
  int parts = 0;
  for (int pp = 3; pp >= 1; --pp)
    if (pp <= MAXP && r * pp <= free_slots) { parts = pp; break; }
  if (r > 0 && parts == 0) return false;
  if (r == 0) parts = 1;
  const long long items = r * parts;
  out->q = q; out->G = G; out->nst = nst; out->parts = parts; out->items = (int)items; out->main_parts = 1;
  const long long per_wg = (items + G - 1) / G;                                   // leftover items in the fullest workgroup
  const double cost_blocks = (double)nst * (q / 4) + (double)((per_wg + 3) / 4) * ceil_div(nst, parts);
  const int ngroups = ceil_div(N, 512);
  const double cost_ranges = ((double)ngroups * nst / (double)(G < 2 * ngroups ? G : 2 * ngroups) + 3.5) * 4.0;
  return force || cost_blocks < cost_ranges;
}

// ---- exact pass ------------------------------------------------------------------------------------------------------
// 8 lanes per segment row.  Lane j looks after half (j >> 2) of every part and, of that half's 3 x 3 intersections
// (i, j) = (c / 3, c % 3), the combinations c = (j & 3), (j & 3) + 4, (j & 3) + 8.  The live ones are gathered into one
// list per row (identical on the 8 lanes) and scored two at a time with the shared dot routine.
constexpr int MAXC = 8;                // candidates re-scored per row; more than that (rare) -> the row takes the rescan
// (Scanning the uncertain rows inside this kernel instead of queueing them for the rescan was measured in round 3: the few workgroups that own one become a
// 15-us tail - 25.7 -> 41-52 us against the 12 us of the two launches it saves - and was removed.)
constexpr int RS_SLICE = 64, RS_ROWS = 4;   // the exact rescan's work item: RS_ROWS flagged rows x a slice of profiles (below)

__global__ __launch_bounds__(256) void aff_rowcol_rescore_kernel(const float* __restrict__ E, const float* __restrict__ Pm,
                                                                const float* __restrict__ resid_e,
                                                                const float* __restrict__ resid_p, int N, int P, int segs,
                                                                const float* __restrict__ stats,
                                                                const int32_t* __restrict__ part_base,
                                                                const int32_t* __restrict__ part_cnt,
                                                                int32_t* __restrict__ idx, float* __restrict__ score,
                                                                int32_t* __restrict__ flag_count, int32_t* __restrict__ flag_rows,
                                                                int32_t* __restrict__ quad_done, unsigned long long* __restrict__ row_best) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = tid & 7, hh = j >> 2, sl = j & 3;
  const int row = blockIdx.x * 32 + (tid >> 3);
  const bool live = row < N;
  const int rc = live ? row : N - 1;
  const int grp = rc / segs;
  // everything that does not depend on other loads is requested up front
  f32x4 T4[MAXP], C4[MAXP];
  int pb[MAXP];
  const int np_raw = part_cnt[grp];
  {                                      // slot 0 always exists: requested together with the count
    const float* src = stats + (((int64_t)rc * MAXP) * 2 + hh) * 8;
    T4[0] = *reinterpret_cast<const f32x4*>(src);
    C4[0] = *reinterpret_cast<const f32x4*>(src + 4);
    pb[0] = part_base[grp * MAXP];
  }
  const int np = np_raw < MAXP ? np_raw : MAXP;
#pragma unroll
  for (int p = 1; p < MAXP; ++p) {       // further parts only where a sweep was split (block plan: 2 % of the blocks; 128 B per row and part saved)
    T4[p] = f32x4{EMPTY, EMPTY, EMPTY, EMPTY};
    C4[p] = T4[p];
    pb[p] = 0;
    if (p < np) {
      const float* src = stats + (((int64_t)rc * MAXP + p) * 2 + hh) * 8;
      T4[p] = *reinterpret_cast<const f32x4*>(src);
      C4[p] = *reinterpret_cast<const f32x4*>(src + 4);
      pb[p] = part_base[grp * MAXP + p];
    }
  }
  const float re = resid_e[rc];
  float e24[24];
  load_row24(E + (int64_t)rc * D, j, e24);     // plain loads: a non-temporal hint here cost 19 us (25.7 -> 44.6: E then comes from HBM, not the Infinity Cache)
  // rigorous |exact - coarse| bound for this row (resid_p = max profile residual); the slack covers the tag bits
  // (2^-13 relative), the MFMA's fp32 accumulation and this formula's own rounding
  const float eps = (re + (1.0f + re) * resid_p[0]) * 1.0001f + 1.5e-4f;
  float best = -INFINITY;
#pragma unroll
  for (int p = 0; p < MAXP; ++p)
    if (p < np) best = fmaxf(best, __uint_as_float(__float_as_uint(T4[p][0]) & ~TMASK));
  best = fmaxf(best, __shfl_xor(best, 4, 64));                       // the other half
  const int gsh = lane & ~7;
  // Phase 1: the exact score s1 of the best coarse entry (intersection (0, 0) of the half that holds it).  Any exact
  // score is a lower bound on the winner's, so everything whose bound is below s1 - eps cannot win: a far tighter cut
  // than best - 3 eps (which assumes the worst about the best entry's own rounding).
  int first = -1;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    if (p < np && sl == 0 && __uint_as_float(__float_as_uint(T4[p][0]) & ~TMASK) == best) {
      const uint32_t tb = __float_as_uint(T4[p][0]) & TMASK, r = __float_as_uint(C4[p][0]) & CMASK;
      const int pidx = ((int)tb + pb[p]) * PT + (int)((r & 3) + 8 * (r >> 2)) + 4 * hh;
      if (pidx < P) first = pidx;
    }
  }
  float bs;
  int bi;
  {
    const uint32_t bits = (uint32_t)(__ballot(first >= 0) >> gsh) & 0xffu;
    const int o = bits ? __ffs(bits) - 1 : 0;
    bi = __shfl(first, gsh + o, 64);
    bi = bi >= 0 ? bi : 0;                                           // cannot happen for P >= 1; stay in bounds regardless
    bs = dot192_group8(e24, Pm + (int64_t)bi * D, j);
  }
  const int c_first = bi;
  const float cut = fmaxf(best - 3.0f * eps, bs - eps);
  float u = -INFINITY;
  int mine[MAXP][3];
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
#pragma unroll
    for (int q = 0; q < 3; ++q) mine[p][q] = -1;
    if (p < np) {
      float Tv[4], Cv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Tv[q] = __uint_as_float(__float_as_uint(T4[p][q]) & ~TMASK);
        Cv[q] = __uint_as_float(__float_as_uint(C4[p][q]) & ~CMASK);
      }
      u = fmaxf(u, fmaxf(Tv[3], Cv[3]));
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int c = sl + 4 * q;                       // combination owned by this lane
        if (c < 9) {
          const int ii = c / 3, jj = c - 3 * ii;
          const float tv = ii == 0 ? Tv[0] : ii == 1 ? Tv[1] : Tv[2];
          const float cv = jj == 0 ? Cv[0] : jj == 1 ? Cv[1] : Cv[2];
          if (tv >= cut && cv >= cut) {                 // x[t][r] <= min(T_t, C_r): may still matter
            const uint32_t tb = __float_as_uint(ii == 0 ? T4[p][0] : ii == 1 ? T4[p][1] : T4[p][2]) & TMASK;
            const uint32_t r = __float_as_uint(jj == 0 ? C4[p][0] : jj == 1 ? C4[p][1] : C4[p][2]) & CMASK;
            const int pidx = ((int)tb + pb[p]) * PT + (int)((r & 3) + 8 * (r >> 2)) + 4 * hh;
            if (pidx < P && pidx != c_first) mine[p][q] = pidx;
          }
        }
      }
    }
  }
  // Phase 2: the row's remaining candidates, gathered into one list (identical on its 8 lanes), scored two at a time
  int list[MAXC];
#pragma unroll
  for (int m = 0; m < MAXC; ++m) list[m] = 0;
  int M = 0;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const unsigned long long bal = __ballot(mine[p][q] >= 0);
      uint32_t bits = (uint32_t)(bal >> gsh) & 0xffu;
      while (bits) {                                               // uniform inside a lane group
        const int o = __ffs(bits) - 1;
        bits &= bits - 1;
        const int ci = __shfl(mine[p][q], gsh + o, 64);
#pragma unroll
        for (int m = 0; m < MAXC; ++m)
          if (m == M) list[m] = ci;
        ++M;
      }
    }
  }
  for (int m = 0; m < MAXC; ++m) {                                 // one at a time: the list is empty for ~85 % of the rows, and 24 fewer
    if (m < M) {                                                   // registers buy a wave more per SIMD for the two dependent round trips
      int c0 = list[0];
#pragma unroll
      for (int q = 1; q < MAXC; ++q)
        if (q == m) c0 = list[q];
      const float s0 = dot192_group8(e24, Pm + (int64_t)c0 * D, j);
      if (better(s0, c0, bs, bi)) { bs = s0; bi = c0; }
    }
  }
  u = fmaxf(u, __shfl_xor(u, 4, 64));
  if (live && j == 0) {
    // What was not re-scored: (a) entries outside the top-3 rows x top-3 columns: coarse <= u, exact <= u + eps;
    // (b) intersections pruned by best - 3 eps: exact < best - 2 eps; (c) intersections pruned by s1 - eps: exact < s1
    // (the slack inside eps exceeds the tag truncation), and s1 <= the winner's score by construction.
    const float outside = fmaxf(u + eps, best - 2.0f * eps);
    const bool uncertain = M > MAXC || !(bs > outside) || np_raw > MAXP;   // (np_raw > MAXP: a part's record was dropped - cannot happen)
    row_best[row] = 0ull;                                            // the rescan's per-flagged-row keys (index < count <= N) ...
    if ((row & (RS_ROWS - 1)) == 0) quad_done[row / RS_ROWS] = 0;   // arrival counters of the rescan's row quads (index < count / RS_ROWS <= N / RS_ROWS)
    if (uncertain) flag_rows[atomicAdd(flag_count, 1)] = row;
    idx[row] = bi;
    score[row] = bs;
  }
}

// ---- exact rescan of the rows the certificate could not settle ----------------------------------------------------------
// Work item = (four flagged rows, slice of 64 profiles): a profile row fetched from L2 is scored against four segment rows
// held in registers (the old form streamed all of P once PER flagged row: 768 MB of L2 traffic at config #3), and an item
// is ONE round of loads (2 profile rows + 4 segment rows per lane group, all requested before the first use): with ~100
// flagged rows the kernel is a chain of memory round trips, so the chain is kept short and the items many.

constexpr int RS_GRID = 2048;
// slices per row quad: enough items to fill the grid once, never shorter than RS_SLICE profiles (same split in both kernels)
__device__ __forceinline__ void rescan_split(int P, int nq, int& nsl, int& slen) {
  const int nmax = (P + RS_SLICE - 1) / RS_SLICE;
  int want = nq > 0 ? RS_GRID / nq : 1;
  want = want < 1 ? 1 : (want > nmax ? nmax : want);
  slen = ((nmax + want - 1) / want) * RS_SLICE;
  nsl = (P + slen - 1) / slen;
}

__device__ __forceinline__ void group_best32(float& s, int& i) {     // best over the 32 lanes 0..31 of a wave
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    const float ts = __shfl_xor(s, o, 64);
    const int ti = __shfl_xor(i, o, 64);
    if (better(ts, ti, s, i)) { s = ts; i = ti; }
  }
}

// (score, index) as one 64-bit key whose unsigned order is better(): higher score first, then the LOWER index
__device__ __forceinline__ unsigned long long best_key(float s, int i) {
  const uint32_t b = __float_as_uint(s);
  const uint32_t o = (b & 0x80000000u) ? ~b : b | 0x80000000u;       // monotone map of the float order onto unsigned order
  return ((unsigned long long)o << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i);
}

__global__ __launch_bounds__(256) void aff_rescan4_kernel(const float* __restrict__ E, const float* __restrict__ Pm, int P,
                                                         const int32_t* __restrict__ flag_count,
                                                         const int32_t* __restrict__ flag_rows, unsigned long long* __restrict__ row_best,
                                                         int32_t* __restrict__ quad_done,
                                                         int32_t* __restrict__ idx, float* __restrict__ score) {
  __shared__ float ls[RS_ROWS][32];
  __shared__ int li[RS_ROWS][32];
  __shared__ int last_s;
  const int tid = threadIdx.x, j = tid & 7, g = tid >> 3;
  const int count = *flag_count;
  const int nq = (count + RS_ROWS - 1) / RS_ROWS;
  int nsl, slen;
  rescan_split(P, nq, nsl, slen);
  for (int item = blockIdx.x; item < nq * nsl; item += gridDim.x) {
    const int q = item / nsl, sl = item - q * nsl;
    const int s0 = sl * slen, s1 = min(P, s0 + slen);
    float e24[RS_ROWS][24];
#pragma unroll
    for (int x = 0; x < RS_ROWS; ++x) {
      const int f = min(q * RS_ROWS + x, count - 1);
      const int row = flag_rows[f];
      load_row24(E + (int64_t)row * D, j, e24[x]);
    }
    float bs[RS_ROWS];
    int bi[RS_ROWS];
#pragma unroll
    for (int x = 0; x < RS_ROWS; ++x) { bs[x] = -INFINITY; bi[x] = 0x7fffffff; }
    for (int p0 = s0; p0 < s1; p0 += RS_SLICE) {                  // ascending profile index per lane group
      const int pa = min(p0 + g, s1 - 1), pb = min(p0 + g + 32, s1 - 1);
      f32x4 pv[2][6];
      load_prow(Pm + (int64_t)pa * D, j, pv[0]);
      load_prow(Pm + (int64_t)pb * D, j, pv[1]);
#pragma unroll
      for (int x = 0; x < RS_ROWS; ++x) {
        const float sa = dot192_regs(e24[x], pv[0]);              // every lane of the group runs the shuffles
        const float sb = dot192_regs(e24[x], pv[1]);
        if (p0 + g < s1 && better(sa, pa, bs[x], bi[x])) { bs[x] = sa; bi[x] = pa; }
        if (p0 + g + 32 < s1 && better(sb, pb, bs[x], bi[x])) { bs[x] = sb; bi[x] = pb; }
      }
    }
    if (j == 0) {
#pragma unroll
      for (int x = 0; x < RS_ROWS; ++x) { ls[x][g] = bs[x]; li[x][g] = bi[x]; }
    }
    __syncthreads();
    {
      const int x = tid >> 6, l = tid & 63;                        // wave x settles row x
      float s = l < 32 ? ls[x][l] : -INFINITY;
      int i = l < 32 ? li[x][l] : 0x7fffffff;
      group_best32(s, i);
      const int f = q * RS_ROWS + x;
      if (l == 0 && f < count) {
        if (nsl == 1) { const int row = flag_rows[f]; idx[row] = i; score[row] = s; }
        else __hip_atomic_fetch_max(row_best + f, best_key(s, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (nsl > 1) {
      // The quad's slices meet here instead of in a second launch, through agent-scope ATOMICS only (they are performed at the
      // device's coherence point, so no cache write-back or invalidate is involved: a release/acquire fence pair per item made
      // this kernel 2-3x slower): every slice folds its four winners into the rows' 64-bit (score, index) keys with
      // atomic max, waits for those to be performed (s_waitcnt vmcnt(0)), then counts itself in; whoever
      // counts in last reads the four keys back with atomic loads and writes the answers.
      // ASSUMPTION (gfx950 only; sdk_init refuses every other device, so this path cannot run elsewhere): the ordering between the key
      // atomics and the counter add is NOT expressed in the HIP/LLVM memory model (all four operations are relaxed).  It rests on two
      // hardware facts measured on MI355X / ROCm 7.2 (MI355X_MICROARCH.md, 'Valid forms': "8-B agent atomics both sides"): a no-return
      // agent-scope atomic is counted in vmcnt until it has been performed at the device coherence point (memory side, past the 8 XCD
      // L2s), and inline asm is opaque to the compiler, so neither it nor the hardware can move the counter add ahead of the wait.
      // The answers (idx / score, plain stores by the last arriver) are read by no one inside this launch.  A port to another
      // architecture must replace this by the two-launch merge or by an acq_rel counter RMW; the determinism test
      // (tests/test_gpu_kernels.py::test_affinity_rescan_is_deterministic_and_sliced) covers this architecture and build only.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // (a workgroup-scope fence compiles to no vmcnt wait at all)
      __syncthreads();
      if (tid == 0) last_s = __hip_atomic_fetch_add(quad_done + q, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsl - 1;
      __syncthreads();
      if (last_s && tid < RS_ROWS && q * RS_ROWS + tid < count) {
        const int f = q * RS_ROWS + tid;
        const unsigned long long key = __hip_atomic_load(row_best + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int row = flag_rows[f];
        idx[row] = (int)(0xffffffffu - (uint32_t)key);
        const uint32_t o = (uint32_t)(key >> 32);
        score[row] = __uint_as_float((o & 0x80000000u) ? o & 0x7fffffffu : ~o);
      }
    }
    __syncthreads();                                                // ls / li / last_s are rewritten by the next item
  }
}

__global__ void copy_count_kernel(const int32_t* src, int32_t* dst) { *dst = *src; }

struct Ws {
  float* stats; int32_t* part_base; int32_t* part_cnt; int32_t* flag_count; int32_t* flag_rows; unsigned long long* row_best; int32_t* quad_done;
};
inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
size_t ws_layout(int N, int P, char* base, Ws* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al256(bytes); return p; };
  const size_t ngroups_max = (size_t)(N + 31) / 32;            // the smallest group any variant uses (the block plan: 32 segments)
  char* a = take((size_t)N * MAXP * 2 * 8 * 4);
  char* b = take(ngroups_max * MAXP * 4);
  char* c = take(ngroups_max * 4);
  char* d = take(256);                                          // flag_count
  char* e = take((size_t)N * 4);
  char* f1 = take((size_t)N * 8);                                // per flagged row: 64-bit (score, index) key the rescan's slices fold into
  char* g = take(((size_t)N / RS_ROWS + 1) * 4);
  if (w) { w->stats = (float*)a; w->part_base = (int32_t*)b; w->part_cnt = (int32_t*)c; w->flag_count = (int32_t*)d;
           w->flag_rows = (int32_t*)e; w->row_best = (unsigned long long*)f1; w->quad_done = (int32_t*)g; }
  return off;
}

// Work decomposition of the coarse kernel (host side; the kernel derives each workgroup's range from it).
Geom plan_geometry(int N, int P, int segs, int tps, long long max_wg, int pen = 0) {
  Geom gm;
  gm.segs = segs;
  gm.ngroups = ceil_div(N, segs);
  gm.nst = ceil_div(ceil_div(P, PT), tps);
  gm.U = (long long)gm.ngroups * gm.nst;
  long long G = max_wg;
  if (G > 2LL * gm.ngroups) G = 2LL * gm.ngroups;     // every range >= half a sweep: a group's sweep meets <= 3 workgroups (MAXP slots)
  if (G > gm.U) G = gm.U;
  gm.G = (int)G;
  // the boundary penalty needs every virtual range to hold at least one real unit (no empty workgroup inside a group's slot sequence):
  // floor(V / G) >= pen + 1; with G <= 2 groups that is nst >= pen + 2 - otherwise (very short sweeps) the plain split
  gm.pen = 0;
  if (pen > 0 && gm.nst >= pen + 2 && ((long long)gm.ngroups * (gm.nst + pen)) / G >= pen + 1) gm.pen = pen;
  return gm;
}

template <int WAVES, int SEGB, int TPS, int NSTAGE>
int launch_coarse(sdk_ctx* ctx, const bf16_t* Eb, const bf16_t* Pb, int N, int P, const Ws& w, hipStream_t s, int wg_per_cu,
                  int* segs) {
  constexpr int SEGS = WAVES * SEGB * 32;
  constexpr int LDS = NSTAGE * TPS * TILE_BYTES;
  *segs = SEGS;
  Geom gm = plan_geometry(N, P, SEGS, TPS, (long long)ctx->num_cu * wg_per_cu, ctx->aff_boundary_pen);
  auto kern = aff_rowcol_kernel<WAVES, SEGB, TPS, NSTAGE>;
  if (sdk_lds_optin(ctx, (const void*)kern, LDS)) return 1;
  hipLaunchKernelGGL(kern, dim3(gm.G), dim3(WAVES * 64), LDS, s, Eb, Pb, N, P, gm, w.stats, w.part_base, w.part_cnt, w.flag_count,
                     (unsigned long long*)ctx->dbg_ptr);
  return 0;
}

}  // namespace

size_t aff_rowcol_workspace_bytes(int N, int P) { return ws_layout(N, P, nullptr, nullptr); }

// Host-only view of the decomposition for tests (no device needed): out = {groups, stages per group, workgroups, segments per
// group, record slots per segment}, units = groups * stages.  Workgroup i sweeps units [i*units/wg, (i+1)*units/wg).
extern "C" int sdk_affinity_plan(int N, int P, int num_cu, int32_t* out5, int64_t* units) {
  SDK_REQUIRE(N > 0 && P > 0 && num_cu > 0 && out5 && units, "sdk_affinity_plan: bad argument");
  const Geom gm = plan_geometry(N, P, 8 * 2 * 32, 2, num_cu, AFF_DEFAULT_PEN);       // the default variant: 8 waves x 2 blocks, 2 tiles per stage
  out5[0] = gm.ngroups; out5[1] = gm.nst; out5[2] = gm.G; out5[3] = gm.segs; out5[4] = MAXP;
  *units = gm.U;
  return 0;
}
// Host-only: workgroup i's unit range [u0, u1) and the record slot of its first portion, under the default plan (tests replay the kernel's walk)
extern "C" int sdk_affinity_plan_range(int N, int P, int num_cu, int wg, int64_t* u0, int64_t* u1, int32_t* first_slot) {
  SDK_REQUIRE(N > 0 && P > 0 && num_cu > 0 && u0 && u1 && first_slot, "sdk_affinity_plan_range: bad argument");
  const Geom gm = plan_geometry(N, P, 8 * 2 * 32, 2, num_cu, AFF_DEFAULT_PEN);
  SDK_REQUIRE(wg >= 0 && wg < gm.G, "sdk_affinity_plan_range: workgroup %d of %d", wg, gm.G);
  long long a, b;
  geom_range(gm, wg, a, b);
  *u0 = a; *u1 = b;
  *first_slot = a < b ? (int32_t)(wg - geom_first_wg(gm, a / gm.nst)) : -1;
  return 0;
}
// Host-only: the block plan of a shape (tests).  out6 = {1 if the plan is taken / 0 if the range plan stays, q, workgroups, stages, parts per leftover block, items}
extern "C" int sdk_affinity_block_plan(int N, int P, int num_cu, int force, int32_t* out6) {
  SDK_REQUIRE(N > 0 && P > 0 && num_cu > 0 && out6, "sdk_affinity_block_plan: bad argument");
  BlockPlan bp = {0, 0, 0, 0, 0, 1};
  const bool ok = plan_blocks(N, P, 2, num_cu, &bp, force != 0);
  out6[0] = ok ? 1 : 0; out6[1] = bp.q; out6[2] = bp.G; out6[3] = bp.nst; out6[4] = bp.parts; out6[5] = bp.items;
  return 0;
}
// Host-only: the blocks of wave `wave` (0..7) of workgroup `wg` under that plan: out7 = {block of slot 0, block of slot 1 or -1, first stage, end stage,
// record slot, parts of that block}
extern "C" int sdk_affinity_block_plan_wave(int N, int P, int num_cu, int wg, int wave, int32_t* out6) {
  SDK_REQUIRE(N > 0 && P > 0 && num_cu > 0 && out6 && wave >= 0 && wave < 8, "sdk_affinity_block_plan_wave: bad argument");
  BlockPlan bp = {0, 0, 0, 0, 0, 1};
  SDK_REQUIRE(plan_blocks(N, P, 2, num_cu, &bp, true) && wg >= 0 && wg < bp.G, "sdk_affinity_block_plan_wave: the shape has no block plan, or workgroup %d out of range", wg);
  int b0, b1, e0, e1, s1, c1;
  block_slots(bp, wg, wave, b0, b1, e0, e1, s1, c1);
  out6[0] = b0; out6[1] = b1; out6[2] = e0; out6[3] = e1; out6[4] = s1; out6[5] = c1;
  return 0;
}
bool aff_rowcol_supported(int P) { return P <= 32768; }

// k = 1.  Same contract as sdk_affinity_topk (which dispatches here).
int aff_rowcol_top1(sdk_ctx* ctx, const float* E, const uint16_t* Eb, const float* resid_e, const float* P, const uint16_t* Pb,
                    const float* resid_p, int N, int Pn, int32_t* idx, float* score, int32_t* n_rescanned, void* ws,
                    void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Ws w;
  ws_layout(N, Pn, (char*)ws, &w);
  // no memset: the coarse kernel zeroes the uncertain-row counter, the exact pass zeroes the rescan's arrival counters
  int segs;
  {
    ProfScope ps(ctx, stream, SDK_K_AFF_COARSE, 2.0 * N * (double)Pn * D, 2.0 * ((double)N + Pn) * D + 64.0 * N);
    int rc;
    BlockPlan bp;
    // Coarse-pass plan.  Default (round 5): the BLOCK plan with two records per whole sweep wherever its cost model takes it (short sweeps: config #3),
    // else the range plan (long sweeps: config #4's shape, where it balances to the stage).  `affinity_variant`: 7 = always the range plan, 8 / 12 / 13
    // = always the block plan (where the shape fits it) with 1 / 2 / 3 records per sweep - the A/B pair tests/test_gpu_kernels.py keeps bit-identical.
    // Measured (profiles/r05_aff_block_plan_records.txt, interleaved): config #3 range 82.1 us (88 rows rescanned), blocks with 1 / 2 / 3 records
    // 83.1 (201) / 78.9 (73) / 79.4 (40); 125k x 10k: 463.8 / 482.0 / 459.3 / 466.0.
    const int av = ctx->aff_variant;
    const bool want_blocks = av == 8 || av == 12 || av == 13 || av == 0;
    if (want_blocks && plan_blocks(N, Pn, 2, ctx->num_cu, &bp, av != 0)) {
      bp.main_parts = av == 8 ? 1 : av == 13 ? 3 : 2;
      if (bp.nst < 2 * bp.main_parts) bp.main_parts = 1;        // (a part is at least two stages)
      auto kern = aff_rowcol_blocks_kernel<8, 2, 4>;
      constexpr int LDSB = 4 * 2 * TILE_BYTES;
      if (sdk_lds_optin(ctx, (const void*)kern, LDSB)) return 1;
      hipLaunchKernelGGL(kern, dim3(bp.G), dim3(512), LDSB, s, (const bf16_t*)Eb, (const bf16_t*)Pb, N, Pn, bp, w.stats, w.part_base, w.part_cnt, w.flag_count,
                         (unsigned long long*)ctx->dbg_ptr);
      segs = 32;
      rc = 0;
    } else {
      rc = launch_coarse<8, 2, 2, 4>(ctx, (const bf16_t*)Eb, (const bf16_t*)Pb, N, Pn, w, s, 1, &segs);
    }
    if (rc) return rc;
  }
  SDK_LAUNCH_CHECK();
  {
    ProfScope ps(ctx, stream, SDK_K_AFF_RESCORE, 0.0, 4.0 * N * D + 64.0 * N + 8.0 * N);
    hipLaunchKernelGGL(aff_rowcol_rescore_kernel, dim3(ceil_div(N, 32)), dim3(256), 0, s, E, P, resid_e, resid_p, N, Pn, segs,
                       w.stats, w.part_base, w.part_cnt, idx, score, w.flag_count, w.flag_rows, w.quad_done, w.row_best);
  }
  SDK_LAUNCH_CHECK();
  {
    ProfScope ps(ctx, stream, SDK_K_AFF_RESCAN, 0.0, 0.0);
    hipLaunchKernelGGL(aff_rescan4_kernel, dim3(RS_GRID), dim3(256), 0, s, E, P, Pn, w.flag_count, w.flag_rows, w.row_best,
                       w.quad_done, idx, score);
  }
  SDK_LAUNCH_CHECK();
  if (n_rescanned) {
    hipLaunchKernelGGL(copy_count_kernel, dim3(1), dim3(1), 0, s, w.flag_count, n_rescanned);
    SDK_LAUNCH_CHECK();
  }
  return 0;
}
