// k4: segments x profiles cosine affinity with fused top-k on gfx950.
//
// What a local backend's identify_speaker (speaker_detection_backends/base.py:130-151) needs per
// segment is "the best enrolled profiles and their scores", never the N x P score matrix.  So:
//
//  coarse pass (affinity_coarse_kernel)   bf16 MFMA, S^T tile = P_tile[32 x 192] . E_tile[32 x 192]^T
//      with the PROFILE index on the accumulator registers and the SEGMENT on the lane: every lane
//      owns one segment and keeps a sorted top-3 of the profiles it has seen in registers
//      (insert = 1 v_max + 2 v_med3 on values whose low 10 mantissa bits carry the profile index;
//      5 VALU ops per score in all - the kernel is VALU-issue-bound, so every op counts).
//      Profile tiles arrive by LDS-DMA into a 3-stage ring.  Only 6 candidates per segment reach HBM.
//  exact pass (affinity_rescore_kernel)   fp32 re-score of the candidates that can still win, fixed
//      summation order; certifies the result against the rigorous rounding bound
//          |exact - coarse| <= r_e + (1 + r_e) r_p + 2^-13 + K 2^-23 =: eps
//      (r_e, r_p = measured bf16 rounding residual norms from sdk_l2norm): any profile that is not a
//      candidate scores at most u + eps, u = the larger of the two lane-halves' 3rd-best coarse
//      score.  Rows with x_k <= u + eps (~1 % for k = 1) are queued for
//  exact rescan (affinity_rescan_kernel)  fp32 scan over all P with the same dot-product routine,
//      one work item per (row, 1024-profile slice), merged by affinity_rescan_merge_kernel.
// The reported (idx, score) therefore equal an fp32 full scan: ties -> lowest profile index.
#include <stdlib.h>

#include "common.hpp"
#include "scoring_exact.hpp"

using namespace sdk_exact;

namespace {

constexpr int KS = D / 16;
constexpr int SEG_PER_WAVE = 32;
constexpr int WAVES = 4;
constexpr int SEG_PER_WG = SEG_PER_WAVE * WAVES;   // 128
constexpr int PT = 32;                 // profiles per tile
constexpr int PROWB = 384;             // LDS bytes per profile row (24 chunks of 16 B, XOR-swizzled)
constexpr int NSTAGE = 3;              // LDS ring depth (2 tiles in flight); 36 KiB -> 4 workgroups per CU
constexpr int CHUNK_TILES = 32;        // 1024 profiles share one 10-bit index space
constexpr uint32_t IDX_MASK = 0x3ffu;
constexpr int MAXDEPTH = 4;            // per-lane sorted candidate list (one lane = one half of the profiles): 3 or 4 deep
constexpr int NCAND = 2 * MAXDEPTH;    // candidate slots per segment in the workspace (unused slots hold index -1)

struct Workspace {        // layout inside the caller's scratch buffer
  float* cand_val;        // [N][8] coarse value (index bits stripped)
  int32_t* cand_idx;      // [N][8] profile index, -1 = empty
  float* ubound;          // [N]    u (see header comment)
  int32_t* flag_count;    // [1]
  int32_t* flag_rows;     // [N]
  float* part_s;          // [N * nslices * 4] partial exact lists of the rescan (only when P > 1024)
  int32_t* part_i;
};

__host__ __device__ inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

inline size_t ws_layout(int N, int P, char* base, Workspace* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
  char* a = take((size_t)N * NCAND * 4);
  char* b = take((size_t)N * NCAND * 4);
  char* c = take((size_t)N * 4);
  char* d = take(256);
  char* e = take((size_t)N * 4);
  const size_t nsl = (size_t)((P + 1023) / 1024);
  char* f1 = take(nsl > 1 ? (size_t)N * nsl * 16 : 16);
  char* f2 = take(nsl > 1 ? (size_t)N * nsl * 16 : 16);
  if (w) { w->part_s = (float*)f1; w->part_i = (int32_t*)f2; }
  if (w) { w->cand_val = (float*)a; w->cand_idx = (int32_t*)b; w->ubound = (float*)c; w->flag_count = (int32_t*)d; w->flag_rows = (int32_t*)e; }
  return off;
}

// sorted insert of x into m[0] >= m[1] >= ... : 1 v_max + (DEPTH-1) v_med3; values are packed floats
template <int DEPTH>
__device__ __forceinline__ void insert_sorted(float x, float* m) {
  float n[DEPTH];
#pragma unroll
  for (int q = DEPTH - 1; q >= 1; --q) n[q] = __builtin_amdgcn_fmed3f(x, m[q - 1], m[q]);
  n[0] = fmaxf(x, m[0]);
#pragma unroll
  for (int q = 0; q < DEPTH; ++q) m[q] = n[q];
}

template <int DEPTH>
__global__ __launch_bounds__(WAVES * 64, 2) void affinity_coarse_kernel(const bf16_t* __restrict__ Eb,
                                                                       const bf16_t* __restrict__ Pb, int N, int P,
                                                                       float* __restrict__ cand_val,
                                                                       int32_t* __restrict__ cand_idx,
                                                                       float* __restrict__ ubound) {
  // Profile tiles stream through a 3-stage LDS ring filled by LDS-DMA (global_load_lds_dwordx4), two
  // tiles in flight behind a counted vmcnt and ONE raw barrier per tile.  Rows are 384 B (24 chunks of
  // 16 B), unpadded because the DMA writes 1 KiB linearly; bank conflicts are removed by XORing the low
  // 3 bits of the chunk index with (row >> 1) & 7 on the SOURCE address and again on the fragment read.
  __shared__ __attribute__((aligned(16))) char sP[NSTAGE][PT * PROWB];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int seg = blockIdx.x * SEG_PER_WG + wid * SEG_PER_WAVE + col;
  const int seg_c = seg < N ? seg : N - 1;

  // B operand (segments): B[k = 16 ks + 8h + j][col] = Eb[seg][16 ks + 8h + j]; resident for the whole sweep
  bf16x8 bfrag[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    bfrag[ks] = *reinterpret_cast<const bf16x8*>(Eb + (int64_t)seg_c * D + ks * 16 + h * 8);

  const int ntiles = (P + PT - 1) / PT;
  // DMA assignment: a tile is 768 chunks = 12 wave-instructions; wave w issues instructions 3w .. 3w+2
  int drow[3], dsrc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int id = (wid * 3 + i) * 64 + lane;
    const int row = id / 24, pos = id - row * 24;
    drow[i] = row;
    dsrc[i] = ((pos & ~7) | ((pos & 7) ^ ((row >> 1) & 7))) * 8;    // source chunk (elements)
  }
  auto issue = [&](int tile) {
    char* st = sP[tile % NSTAGE];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      int pr = tile * PT + drow[i];
      pr = pr < P ? pr : P - 1;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(Pb + (int64_t)pr * D + dsrc[i]),
                                       (void __attribute__((address_space(3)))*)(st + (wid * 3 + i) * 1024), 16, 0, 0);
    }
  };

  // global candidate list of this lane (its half of the profiles): value + full index
  float gv[DEPTH];
  int gi[DEPTH];
  float mv[DEPTH];                                             // chunk-local packed list
#pragma unroll
  for (int q = 0; q < DEPTH; ++q) { gv[q] = -INFINITY; gi[q] = -1; mv[q] = -INFINITY; }

  auto merge_chunk = [&](int chunk) {
#pragma unroll
    for (int e = 0; e < DEPTH; ++e) {
      const uint32_t bits = __float_as_uint(mv[e]);
      const bool live = mv[e] != -INFINITY;
      const float v = live ? __uint_as_float(bits & ~IDX_MASK) : -INFINITY;
      const int idx = chunk * (CHUNK_TILES * PT) + (int)(bits & IDX_MASK);
      // insert (v, idx) into the sorted global list (a dead entry compares below everything: no-op)
      int pos = DEPTH;
#pragma unroll
      for (int q = DEPTH - 1; q >= 0; --q)
        if (v > gv[q]) pos = q;
#pragma unroll
      for (int q = DEPTH - 1; q >= 1; --q)
        if (q > pos) { gv[q] = gv[q - 1]; gi[q] = gi[q - 1]; }
#pragma unroll
      for (int q = 0; q < DEPTH; ++q)
        if (q == pos) { gv[q] = v; gi[q] = idx; }
    }
#pragma unroll
    for (int q = 0; q < DEPTH; ++q) mv[q] = -INFINITY;
  };

  issue(0);
  if (ntiles > 1) issue(1);
  const int rsw = (col >> 1) & 7;
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // tile t+1 (3 DMA instructions) may stay in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // tile t landed for everyone; the stage of tile t-1 is free
    if (t + 2 < ntiles) issue(t + 2);
    const char* st = sP[t % NSTAGE] + col * PROWB;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks * 2 + h;
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(st + (((c & ~7) | ((c & 7) ^ rsw)) << 4));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[ks], acc, 0, 0, 0);
    }
    const int tl = t & (CHUNK_TILES - 1);
    const uint32_t tb = (uint32_t)(tl * PT + 4 * h);
    if ((t + 1) * PT <= P) {                                     // full tile: 5 VALU ops per score
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t rc = (uint32_t)((r & 3) + 8 * (r >> 2));
        insert_sorted<DEPTH>(__uint_as_float(((__float_as_uint(acc[r]) & ~IDX_MASK) | tb) | rc), mv);
      }
    } else {                                                     // last, partial tile: rows past the last profile never enter
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t rc = (uint32_t)((r & 3) + 8 * (r >> 2));
        float x = __uint_as_float(((__float_as_uint(acc[r]) & ~IDX_MASK) | tb) | rc);
        if (t * PT + (int)rc + 4 * h >= P) x = -INFINITY;
        insert_sorted<DEPTH>(x, mv);
      }
    }
    if (tl == CHUNK_TILES - 1 || t + 1 == ntiles) merge_chunk(t / CHUNK_TILES);
  }
  if (seg < N) {
#pragma unroll
    for (int e = 0; e < MAXDEPTH; ++e) {
      cand_val[(int64_t)seg * NCAND + h * MAXDEPTH + e] = e < DEPTH ? gv[e < DEPTH ? e : 0] : -INFINITY;
      cand_idx[(int64_t)seg * NCAND + h * MAXDEPTH + e] = e < DEPTH ? gi[e < DEPTH ? e : 0] : -1;
    }
    // u = max over the two halves of their last list entry (a bound on every profile that is not a candidate)
    const float other = __shfl_xor(gv[DEPTH - 1], 32, 64);
    if (h == 0) ubound[seg] = fmaxf(gv[DEPTH - 1], other);
  }
}

__global__ __launch_bounds__(256) void affinity_rescore_kernel(const float* __restrict__ E, const float* __restrict__ Pm,
                                                              const float* __restrict__ resid_e,
                                                              const float* __restrict__ resid_p, int N, int P, int k,
                                                              const float* __restrict__ cand_val,
                                                              const int32_t* __restrict__ cand_idx,
                                                              const float* __restrict__ ubound, int32_t* __restrict__ idx,
                                                              float* __restrict__ score, int32_t* __restrict__ flag_count,
                                                              int32_t* __restrict__ flag_rows) {
  const int tid = threadIdx.x;
  const int j = tid & 7;
  const int row = blockIdx.x * 32 + (tid >> 3);
  const bool live = row < N;
  const int rc = live ? row : N - 1;
  float e24[24];
  load_row24(E + (int64_t)rc * D, j, e24);
  const float re = resid_e[rc];
  // rigorous |exact - coarse| bound for this row (resid_p = max profile residual), small slack for fp32
  const float eps = (re + (1.0f + re) * resid_p[0]) * 1.0001f + 1.5e-4f;
  float cv[NCAND];
  int ci[NCAND];
  float best_coarse = -INFINITY;
#pragma unroll
  for (int c = 0; c < NCAND; ++c) {
    cv[c] = cand_val[(int64_t)rc * NCAND + c];
    ci[c] = cand_idx[(int64_t)rc * NCAND + c];
    if (ci[c] >= P) ci[c] = -1;   // never trust an index produced from non-finite scores
    if (ci[c] >= 0) best_coarse = fmaxf(best_coarse, cv[c]);
  }
  // the k best coarse values: anything below (k-th best coarse - 2 eps) cannot be in the exact top-k
  float kth = best_coarse;
  if (k > 1) {
    float tmp[NCAND];
#pragma unroll
    for (int c = 0; c < NCAND; ++c) tmp[c] = ci[c] >= 0 ? cv[c] : -INFINITY;
    for (int it = 1; it < k; ++it) {   // strip the current maximum k-1 times
      float mx = -INFINITY; int arg = -1;
#pragma unroll
      for (int c = 0; c < NCAND; ++c)
        if (tmp[c] > mx) { mx = tmp[c]; arg = c; }
#pragma unroll
      for (int c = 0; c < NCAND; ++c)
        if (c == arg) tmp[c] = -INFINITY;
    }
    kth = -INFINITY;
#pragma unroll
    for (int c = 0; c < NCAND; ++c) kth = fmaxf(kth, tmp[c]);
  }
  // 3 eps: the k best coarse candidates score >= kth - eps exactly, strictly above the pruning bound
  // (cut + eps = kth - 2 eps), so pruning alone can never make a row uncertain.
  const float cut = kth - 3.0f * eps;
  float bs[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
#pragma unroll
  for (int c = 0; c < NCAND; ++c) {
    // the 8 lanes of a group take the same branch (same row): no divergence inside the shuffles
    if (ci[c] >= 0 && ci[c] < P && cv[c] >= cut) {
      const float s = dot192_group8(e24, Pm + (int64_t)ci[c] * D, j);
      insert_exact<4>(s, ci[c], bs, bi);
    }
  }
  if (live && j == 0) {
    const float u = ubound[row];
    // every profile that was not re-scored has coarse <= max(u, cut), hence exact <= max(u, cut) + eps
    const float outside = fmaxf(u, cut) + eps;
    const bool uncertain = !(bs[k - 1] > outside);
    if (uncertain) {
      const int slot = atomicAdd(flag_count, 1);
      flag_rows[slot] = row;
    }
    for (int q = 0; q < k; ++q) {
      idx[(int64_t)row * k + q] = bi[q];
      score[(int64_t)row * k + q] = bs[q];
    }
  }
}

// Exact rescan of the uncertain rows.  Work item = (flagged row, slice of 1024 profiles): 32 lane groups
// score 32 profiles each (four rows in flight per group), the 128 partial entries meet in LDS and one
// wave selects the slice's best four by shuffles.  A second tiny kernel merges a row's slices.
constexpr int SLICE = 1024;

__global__ __launch_bounds__(256) void affinity_rescan_kernel(const float* __restrict__ E, const float* __restrict__ Pm,
                                                             int P, int k, const int32_t* __restrict__ flag_count,
                                                             const int32_t* __restrict__ flag_rows,
                                                             float* __restrict__ part_s, int32_t* __restrict__ part_i,
                                                             int32_t* __restrict__ idx, float* __restrict__ score) {
  __shared__ float ls[128];
  __shared__ int li[128];
  const int tid = threadIdx.x, j = tid & 7, g = tid >> 3;
  const int count = *flag_count;
  const int nsl = (P + SLICE - 1) / SLICE;
  for (int item = blockIdx.x; item < count * nsl; item += gridDim.x) {
    const int f = item / nsl, sl = item - f * nsl;
    const int row = flag_rows[f];
    const int p0 = sl * SLICE, p1 = min(P, p0 + SLICE);
    float e24[24];
    load_row24(E + (int64_t)row * D, j, e24);
    float bs[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    for (int p = p0 + g; p < p1; p += 4 * 32) {             // four profile rows in flight per lane group
      f32x4 pv[4][6];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = p + 32 * u;
        load_prow(Pm + (int64_t)(pp < p1 ? pp : p1 - 1) * D, j, pv[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = p + 32 * u;
        const float sc = dot192_regs(e24, pv[u]);         // every lane of the group runs the shuffles
        if (pp < p1) insert_exact<4>(sc, pp, bs, bi);
      }
    }
    if (j == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { ls[g * 4 + q] = bs[q]; li[g * 4 + q] = bi[q]; }
    }
    __syncthreads();
    if (tid < 64) {
      float os[4]; int oi[4];
      wave_select4(ls[tid], li[tid], ls[tid + 64], li[tid + 64], os, oi);
      if (tid == 0) {
        if (nsl == 1) {
          for (int q = 0; q < k; ++q) { idx[(int64_t)row * k + q] = oi[q]; score[(int64_t)row * k + q] = os[q]; }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) { part_s[(int64_t)item * 4 + q] = os[q]; part_i[(int64_t)item * 4 + q] = oi[q]; }
        }
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void affinity_rescan_merge_kernel(int P, int k, const int32_t* __restrict__ flag_count,
                                                                  const int32_t* __restrict__ flag_rows,
                                                                  const float* __restrict__ part_s, const int32_t* __restrict__ part_i,
                                                                  int32_t* __restrict__ idx, float* __restrict__ score) {
  const int nsl = (P + SLICE - 1) / SLICE;
  if (nsl == 1) return;
  const int count = *flag_count, lane = threadIdx.x;
  for (int f = blockIdx.x; f < count; f += gridDim.x) {
    float fs[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int fi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    for (int e0 = 0; e0 < nsl * 4; e0 += 128) {            // 128 partial entries per pass, folded into the running best
      const int ea = e0 + lane, eb = e0 + lane + 64;
      float s0 = ea < nsl * 4 ? part_s[(int64_t)f * nsl * 4 + ea] : -INFINITY;
      int i0 = ea < nsl * 4 ? part_i[(int64_t)f * nsl * 4 + ea] : 0x7fffffff;
      float s1 = eb < nsl * 4 ? part_s[(int64_t)f * nsl * 4 + eb] : -INFINITY;
      int i1 = eb < nsl * 4 ? part_i[(int64_t)f * nsl * 4 + eb] : 0x7fffffff;
      float os[4]; int oi[4];
      wave_select4(s0, i0, s1, i1, os, oi);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (oi[q] != 0x7fffffff) insert_exact<4>(os[q], oi[q], fs, fi);
    }
    if (lane == 0) {
      const int row = flag_rows[f];
      for (int q = 0; q < k; ++q) { idx[(int64_t)row * k + q] = fi[q]; score[(int64_t)row * k + q] = fs[q]; }
    }
  }
}

__global__ void copy_count_kernel(const int32_t* src, int32_t* dst) { *dst = *src; }

}  // namespace

// k = 1 fast path (affinity_rowcol.hip)
size_t aff_rowcol_workspace_bytes(int N, int P);
bool aff_rowcol_supported(int P);
int aff_rowcol_top1(sdk_ctx* ctx, const float* E, const uint16_t* Eb, const float* resid_e, const float* P, const uint16_t* Pb,
                    const float* resid_p, int N, int Pn, int32_t* idx, float* score, int32_t* n_rescanned, void* ws, void* stream);

extern "C" size_t sdk_affinity_workspace_bytes(int N, int P) {
  if (N <= 0 || P <= 0) return 0;
  const size_t a = ws_layout(N, P, nullptr, nullptr), b = aff_rowcol_workspace_bytes(N, P);
  return a > b ? a : b;
}

extern "C" int sdk_affinity_topk(sdk_ctx* ctx, const float* E, const uint16_t* Eb, const float* resid_e,
                                 const float* P, const uint16_t* Pb, const float* resid_p, int N, int Pn, int d, int k,
                                 int32_t* idx, float* score, int32_t* n_rescanned, void* ws, size_t ws_bytes,
                                 void* stream) {
  SDK_REQUIRE(ctx && E && Eb && resid_e && P && Pb && resid_p && idx && score && ws, "sdk_affinity_topk: null argument");
  SDK_REQUIRE(d == D, "sdk_affinity_topk: d=%d, this build is specialised for d=%d", d, D);
  SDK_REQUIRE(N > 0 && Pn > 0, "sdk_affinity_topk: empty problem (N=%d P=%d)", N, Pn);
  SDK_REQUIRE(k >= 1 && k <= 4 && k <= Pn, "sdk_affinity_topk: k=%d must be in [1, min(4, P)]", k);
  SDK_REQUIRE(ws_bytes >= sdk_affinity_workspace_bytes(N, Pn), "sdk_affinity_topk: workspace too small");
  SDK_REQUIRE(((uintptr_t)E % 16) == 0 && ((uintptr_t)Eb % 16) == 0 && ((uintptr_t)P % 16) == 0 && ((uintptr_t)Pb % 16) == 0,
              "sdk_affinity_topk: matrices must be 16-byte aligned");
  if (k == 1 && ctx->aff_fast && aff_rowcol_supported(Pn))
    return aff_rowcol_top1(ctx, E, Eb, resid_e, P, Pb, resid_p, N, Pn, idx, score, n_rescanned, ws, stream);
  hipStream_t s = (hipStream_t)stream;
  Workspace w;
  ws_layout(N, Pn, (char*)ws, &w);
  SDK_HIP_OK(hipMemsetAsync(w.flag_count, 0, sizeof(int32_t), s));
  {
  ProfScope ps(ctx, stream, SDK_K_AFF_COARSE, 2.0 * N * (double)Pn * D, 2.0 * ((double)N + Pn) * D + 68.0 * N);
  static const int aff_depth3_max_p = getenv("SDK_AFF_DEPTH3_MAXP") ? atoi(getenv("SDK_AFF_DEPTH3_MAXP")) : 2048;
  // list depth: 3 (5 VALU ops per score, ~1 % of rows re-scanned at P ~ 1k) up to 2048 profiles; 4 beyond, where
  // the order statistics crowd together and every re-scanned row costs a full pass over P
  if (Pn <= aff_depth3_max_p && k <= 2)
    hipLaunchKernelGGL(affinity_coarse_kernel<3>, dim3(ceil_div(N, SEG_PER_WG)), dim3(WAVES * 64), 0, s, (const bf16_t*)Eb,
                       (const bf16_t*)Pb, N, Pn, w.cand_val, w.cand_idx, w.ubound);
  else
    hipLaunchKernelGGL(affinity_coarse_kernel<4>, dim3(ceil_div(N, SEG_PER_WG)), dim3(WAVES * 64), 0, s, (const bf16_t*)Eb,
                       (const bf16_t*)Pb, N, Pn, w.cand_val, w.cand_idx, w.ubound);
  }
  SDK_LAUNCH_CHECK();
  {
  ProfScope ps(ctx, stream, SDK_K_AFF_RESCORE, 0.0, 4.0 * N * D + 68.0 * N + 8.0 * N * k);
  hipLaunchKernelGGL(affinity_rescore_kernel, dim3(ceil_div(N, 32)), dim3(256), 0, s, E, P, resid_e, resid_p, N, Pn, k,
                     w.cand_val, w.cand_idx, w.ubound, idx, score, w.flag_count, w.flag_rows);
  }
  SDK_LAUNCH_CHECK();
  {
  ProfScope ps(ctx, stream, SDK_K_AFF_RESCAN, 0.0, 0.0);
  hipLaunchKernelGGL(affinity_rescan_kernel, dim3(2048), dim3(256), 0, s, E, P, Pn, k, w.flag_count, w.flag_rows, w.part_s,
                     w.part_i, idx, score);
  if (Pn > SLICE)
    hipLaunchKernelGGL(affinity_rescan_merge_kernel, dim3(512), dim3(64), 0, s, Pn, k, w.flag_count, w.flag_rows, w.part_s, w.part_i,
                       idx, score);
  }
  SDK_LAUNCH_CHECK();
  if (n_rescanned) {
    hipLaunchKernelGGL(copy_count_kernel, dim3(1), dim3(1), 0, s, w.flag_count, n_rescanned);
    SDK_LAUNCH_CHECK();
  }
  return 0;
}
