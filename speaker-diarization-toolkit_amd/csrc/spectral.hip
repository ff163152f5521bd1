// k6: pieces of spectral clustering on the rectified cosine affinity A = max(E E^T, 0) (gfx950).
//
// A is never stored (100k x 100k fp32 = 40 GB): every application Y = A X recomputes its tiles on the
// matrix cores, flash-attention style without the softmax:
//     S^T tile [32 j x 32 i] = E_j [32 x 192] . E_i^T          12 x v_mfma_f32_32x32x16_bf16
//     relu, round to bf16 IN REGISTERS - the accumulator layout (column i on the lane, rows j on the
//     registers) is already the B-operand layout of the next MFMA, whose k index is j:
//     Y^T tile [32 c x 32 i] += X^T [32 c x 32 j] . S^T          2 k-steps x (hi + lo) = 4 MFMAs
// X (fp32 [N, kv], kv <= 32) is pre-split into bf16 hi + lo and pre-permuted into that k order by
// pack_x_kernel, so each lane fetches its A fragment with one 16-byte load.  A rank owns the row block
// [row0, row0 + rows) of A; j sweeps all N (the all-gathered embeddings).
//
// The thin [n, k] linear algebra of the subspace iteration and k-means (Gram matrices, X R, scaling,
// assignment) are small HBM-streaming kernels with two-stage, order-fixed reductions (reproducible).
#include "common.hpp"

namespace {

constexpr int D = 192, KS = D / 16, PT = 32, PROW = 400;
constexpr int KV = 32;                      // padded width of X / Y tiles

// Xp[tile][s][part][lane][8]: bf16 fragments of X^T in the k order of an accumulator-as-operand MFMA
__global__ __launch_bounds__(256) void pack_x_kernel(const float* __restrict__ X, int N, int kv, bf16_t* __restrict__ Xp,
                                                    const float* __restrict__ rowscale) {
  const int gid = blockIdx.x * 256 + threadIdx.x;          // one thread per (tile, s, lane)
  const int ntiles = (N + PT - 1) / PT;
  if (gid >= ntiles * 2 * 64) return;
  const int lane = gid & 63, s = (gid >> 6) & 1, tile = gid >> 7;
  const int c = lane & 31, h = lane >> 5;
  float hi[8], lo[8];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const int j = tile * PT + 16 * s + 8 * (jj >> 2) + 4 * h + (jj & 3);
    float v = (j < N && c < kv) ? X[(int64_t)j * kv + c] : 0.f;
    if (rowscale && j < N) v *= rowscale[j];
    const bf16_t b = f32_to_bf16(v);
    hi[jj] = bf16_to_f32(b);
    lo[jj] = v - hi[jj];
  }
  bf16_t* dst = Xp + (((int64_t)tile * 2 + s) * 2) * 64 * 8 + lane * 8;
  *reinterpret_cast<u32x4*>(dst) = pack8(hi);
  *reinterpret_cast<u32x4*>(dst + 64 * 8) = pack8(lo);
}

__global__ __launch_bounds__(256, 2) void affinity_matvec_kernel(const bf16_t* __restrict__ Eb, int N, int row0, int rows,
                                                                const bf16_t* __restrict__ Xp, int kv,
                                                                float* __restrict__ Y) {
  __shared__ __attribute__((aligned(16))) char sE[2][PT * PROW];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int iloc = blockIdx.x * 128 + wid * 32 + col;       // row inside this rank's block
  const int irow = row0 + (iloc < rows ? iloc : rows - 1);

  bf16x8 bfrag[KS];                                           // E_i fragments, resident
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) bfrag[ks] = *reinterpret_cast<const bf16x8*>(Eb + (int64_t)irow * D + ks * 16 + h * 8);

  const int ntiles = (N + PT - 1) / PT;
  u32x4 st[3];
  auto gload = [&](int tile) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int id = tid + 256 * i;
      const int row = id / 24, ch = id - row * 24;
      int pr = tile * PT + row;
      pr = pr < N ? pr : N - 1;
      st[i] = *reinterpret_cast<const u32x4*>(Eb + (int64_t)pr * D + ch * 8);
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int id = tid + 256 * i;
      const int row = id / 24, ch = id - row * 24;
      *reinterpret_cast<u32x4*>(&sE[buf][row * PROW + ch * 16]) = st[i];
    }
  };
  auto xload = [&](int tile, bf16x8* xf) {                    // [s][part]
    const bf16_t* p = Xp + (int64_t)tile * 4 * 64 * 8 + lane * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) xf[q] = *reinterpret_cast<const bf16x8*>(p + q * 64 * 8);
  };

  f32x16 yacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) yacc[r] = 0.f;

  bf16x8 xcur[4], xnext[4];
  gload(0);
  swrite(0);
  xload(0, xcur);
  __syncthreads();
  const int arow = col * PROW + h * 16;
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) {
      gload(t + 1);
      xload(t + 1, xnext);
    }
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(&sE[buf][arow + ks * 32]);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[ks], sacc, 0, 0, 0);
    }
    // rows j past the end of E were clamped to row N-1 when staged: their X rows are zero in Xp, so they add nothing
    bf16x8 sf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) sf[s][jj] = f32_to_bf16(fmaxf(sacc[8 * s + jj], 0.f));
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      yacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xcur[2 * s], sf[s], yacc, 0, 0, 0);
      yacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xcur[2 * s + 1], sf[s], yacc, 0, 0, 0);
    }
    if (t + 1 < ntiles) {
      swrite(buf ^ 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) xcur[q] = xnext[q];
    }
    __syncthreads();
  }
  if (iloc < rows) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (c < kv) Y[(int64_t)(row0 + iloc) * kv + c] = yacc[r];
    }
  }
}

// ---- round 3: the same contraction in the structure of aff_rowcol_kernel (affinity_rowcol.hip) ---------------------------------------
// PMC of the kernel above at 100k x 100k (profiles/r03_pmc_matvec.json): matrix pipe busy 22 % of a wave's lifetime, 39 % of wave cycles
// parked on s_waitcnt / the per-tile __syncthreads, 37 % issue-stalled behind a dependent MFMA: a wave owns ONE block of 32 rows of A, so
// every LDS fragment read feeds one MFMA, a workgroup barrier closes every 16 MFMAs, and 196 row groups of 512 leave 60 of 256 CUs idle
// in the second round.  Here:
//   * a wave owns 64 rows i (two blocks): every E_j fragment read from LDS feeds two MFMAs, two independent accumulation chains;
//   * ONE persistent workgroup per CU (8 waves x 64 = 512 rows i per group) walks a contiguous range of (row group, j stage) units, so
//     every CU gets the same matrix work whatever N is; a group's sweep over j may be split over <= 3 workgroups, each writing its partial
//     Y tile to the workspace, summed in slot order by matvec_reduce_kernel (reproducible: no atomics);
//   * j stages of 2 tiles (64 rows of E as swizzled 384-byte rows + their packed X fragments, 32 KiB) stream through a 4-deep LDS-DMA
//     ring with counted vmcnt and one raw barrier per stage (32 + 32 MFMAs per wave per barrier); nothing is loaded from global memory
//     inside the stage loop.
constexpr int MV_WAVES = 8, MV_TPS = 2, MV_NSTAGE = 4, MV_SEGS = MV_WAVES * 64, MV_MAX_WG = 1024;
constexpr int MV_PROWB = 384, MV_ETILE = PT * MV_PROWB, MV_XTILE = 4 * 64 * 16;          // 12 KiB + 4 KiB per tile
constexpr int MV_STAGE = MV_TPS * (MV_ETILE + MV_XTILE);                                  // 32 KiB
constexpr int MV_LDS = MV_NSTAGE * MV_STAGE;                                              // 128 KiB
constexpr int MV_DPW = MV_STAGE / 1024 / MV_WAVES;                                        // 4 DMA pieces per wave and stage
static_assert(MV_DPW * MV_WAVES * 1024 == MV_STAGE && MV_TPS * MV_ETILE / 1024 == 6 * MV_DPW, "waves 0-5 stage E, waves 6-7 stage X");

struct MvGeom { int ngroups, nst, G, maxp; long long U; };

// One workgroup per CU whatever the shape: with few row groups (a rank's 12 500-row block of config #5 at 8 GPUs = 25 groups) a group's
// sweep is cut into as many parts as it takes - maxp = the most workgroups any one group's sweep can meet.
MvGeom mv_geometry(int rows, int N, int num_cu) {
  MvGeom g;
  g.ngroups = ceil_div(rows, MV_SEGS);
  g.nst = ceil_div(ceil_div(N, PT), MV_TPS);
  g.U = (long long)g.ngroups * g.nst;
  long long G = num_cu > 0 ? num_cu : 256;
  if (G > MV_MAX_WG) G = MV_MAX_WG;
  if (G > g.U) G = g.U;
  g.G = (int)G;
  // range i starts at floor(i U / G): at most nst G / U + 1 = G / ngroups + 1 ranges start inside a group's nst units, plus the one reaching in
  g.maxp = (int)(G / g.ngroups) + 2;
  return g;
}

__global__ __launch_bounds__(MV_WAVES * 64, 2) void affinity_matvec2_kernel(const bf16_t* __restrict__ Eb, int N, int row0, int rows,
                                                                           const bf16_t* __restrict__ Xp, MvGeom gm, float* __restrict__ ypart,
                                                                           int32_t* __restrict__ part_cnt) {
  extern __shared__ __attribute__((aligned(16))) char sM[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const long long u0 = ((long long)blockIdx.x * gm.U) / gm.G, u1 = ((long long)(blockIdx.x + 1) * gm.U) / gm.G;
  if (u0 >= u1) return;
  const int nst = gm.nst;
  const int ntiles = (N + PT - 1) / PT;
  const int wu = __builtin_amdgcn_readfirstlane(wid);

  // DMA assignment: a stage is 32 wave-instructions of 1 KiB; wave w issues pieces 4 w .. 4 w + 3: pieces 0..23 = the stage's 64 E rows
  // (24 sixteen-byte chunks per row, source-side XOR swizzle as in aff_rowcol_kernel), pieces 24..31 = the two tiles' packed X fragments
  int drow[MV_DPW], dsrc[MV_DPW];
#pragma unroll
  for (int i = 0; i < MV_DPW; ++i) {
    const int id = (wu * MV_DPW + i) * 64 + lane;
    const int row = id / 24, pos = id - row * 24;
    drow[i] = row;
    dsrc[i] = ((pos & ~7) | ((pos & 7) ^ ((row >> 1) & 7))) * 8;
  }
  int s_issue = (int)(u0 % nst), k_issue = 0;
  auto issue = [&]() {
    char* st = sM + (k_issue % MV_NSTAGE) * MV_STAGE;
    if (wu < 6) {
#pragma unroll
      for (int i = 0; i < MV_DPW; ++i) {
        int pr = s_issue * (MV_TPS * PT) + drow[i];
        pr = pr < N ? pr : N - 1;                    // rows past the end: some valid row, their X rows are zero
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(Eb + (int64_t)pr * D + dsrc[i]),
                                         (void __attribute__((address_space(3)))*)(st + (wu * MV_DPW + i) * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < MV_DPW; ++i) {
        const int q = (wu - 6) * MV_DPW + i;         // 0..7: tile q >> 2 of the stage, fragment q & 3
        int tile = s_issue * MV_TPS + (q >> 2);
        tile = tile < ntiles ? tile : ntiles - 1;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(Xp + ((int64_t)tile * 4 + (q & 3)) * 512 + lane * 8),
                                         (void __attribute__((address_space(3)))*)(st + MV_TPS * MV_ETILE + q * 1024), 16, 0, 0);
      }
    }
    ++k_issue;
    if (++s_issue == nst) s_issue = 0;
  };

  int b = (int)(u0 / nst), s = (int)(u0 % nst);
  int slot;
  {
    const long long x = (long long)b * nst;
    const long long ifirst = ((x + 1) * gm.G + gm.U - 1) / gm.U - 1;      // first workgroup whose range reaches into group b
    slot = (int)(blockIdx.x - ifirst);
  }
  bf16x8 bfrag[2][KS];
  f32x16 yacc[2];
  auto begin_portion = [&]() {
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int iloc = b * MV_SEGS + (wid * 2 + sb) * 32 + col;
      const int irow = row0 + (iloc < rows ? iloc : rows - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bfrag[sb][ks] = *reinterpret_cast<const bf16x8*>(Eb + (int64_t)irow * D + ks * 16 + h * 8);
#pragma unroll
      for (int r = 0; r < 16; ++r) yacc[sb][r] = 0.f;
    }
  };
  auto end_portion = [&](bool group_done) {
    if (slot < gm.maxp) {                            // (always: mv_geometry sizes maxp for the shortest range)
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        float* dst = ypart + (((int64_t)b * gm.maxp + slot) * MV_SEGS + (wid * 2 + sb) * 32 + col) * KV + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {               // accumulator register r = 4 q + e holds column c = 8 q + 4 h + e of Y
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = yacc[sb][4 * q + e];
          *reinterpret_cast<f32x4*>(dst + 8 * q) = v;
        }
      }
    }
    if (group_done && tid == 0) part_cnt[b] = slot + 1 < gm.maxp ? slot + 1 : gm.maxp;
  };

  constexpr int AHEAD = MV_NSTAGE - 1;
#pragma unroll
  for (int a = 0; a < AHEAD; ++a)
    if (u0 + a < u1) issue();
  const int rsw = (col >> 1) & 7;
  int aoff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) aoff[q] = col * MV_PROWB + (((2 * q + h) ^ rsw) << 4);
  int k = 0;
  long long u = u0;
  while (u < u1) {
    const long long gend = (long long)(b + 1) * nst;
    const long long uend = gend < u1 ? gend : u1;
    begin_portion();                                 // plain loads OUTSIDE the stage loop: drained once here, the loop keeps its counted waits
    for (; u < uend; ++u, ++k) {
      const long long after = u1 - 1 - u < AHEAD - 1 ? u1 - 1 - u : AHEAD - 1;
      if (after >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (after == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                  // stage k landed for everyone; the buffer of stage k - 1 is free
      if (u + AHEAD < u1) issue();
      const char* stg = sM + (k % MV_NSTAGE) * MV_STAGE;
#pragma unroll
      for (int tt = 0; tt < MV_TPS; ++tt) {
        if (s * MV_TPS + tt < ntiles) {              // wave-uniform
          f32x16 sacc[2];
#pragma unroll
          for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[sb][r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(stg + tt * MV_ETILE + aoff[ks & 3] + (ks >> 2) * 128);
            sacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[0][ks], sacc[0], 0, 0, 0);
            sacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[1][ks], sacc[1], 0, 0, 0);
          }
          bf16x8 xf[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) xf[q] = *reinterpret_cast<const bf16x8*>(stg + MV_TPS * MV_ETILE + (tt * 4 + q) * 1024 + lane * 16);
#pragma unroll
          for (int sb = 0; sb < 2; ++sb) {
            bf16x8 sf[2];
#pragma unroll
            for (int hs = 0; hs < 2; ++hs)
#pragma unroll
              for (int jj = 0; jj < 8; ++jj) sf[hs][jj] = f32_to_bf16(fmaxf(sacc[sb][8 * hs + jj], 0.f));
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
              yacc[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[2 * hs], sf[hs], yacc[sb], 0, 0, 0);
              yacc[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[2 * hs + 1], sf[hs], yacc[sb], 0, 0, 0);
            }
          }
        }
      }
      ++s;
    }
    end_portion(s == nst);
    if (s == nst) { ++b; s = 0; slot = 0; }
  }
}

// Y[row0 + i, c] = sum over the parts of row group i / 512, in slot order
__global__ __launch_bounds__(256) void matvec_reduce_kernel(const float* __restrict__ ypart, const int32_t* __restrict__ part_cnt, int maxp, int row0,
                                                           int rows, int kv, float* __restrict__ Y) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;       // one thread per (row, 4 columns)
  const int i = (int)(gid >> 3), c4 = (int)(gid & 7) * 4;
  if (i >= rows || c4 >= kv) return;
  const int g = i / MV_SEGS, il = i - g * MV_SEGS;
  const int np = part_cnt[g];
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < np; ++p) a += *reinterpret_cast<const f32x4*>(ypart + (((int64_t)g * maxp + p) * MV_SEGS + il) * KV + c4);
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (c4 + e < kv) Y[(int64_t)(row0 + i) * kv + c4 + e] = a[e];
}

// ---- thin [n, k] helpers -------------------------------------------------------------------------
// G_part[block] = X_blk^T Y_blk (k x k), fixed-order in-block reduction; gram_reduce sums the blocks in order.
constexpr int GR_ROWS = 256;
__global__ __launch_bounds__(256) void gram_partial_kernel(const float* __restrict__ X, const float* __restrict__ Yv, int n,
                                                          int k, float* __restrict__ part) {
  __shared__ float xs[GR_ROWS][KV + 1];
  __shared__ float ys[GR_ROWS][KV + 1];
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * GR_ROWS;
  for (int i = tid; i < GR_ROWS * k; i += 256) {
    const int r = i / k, c = i - r * k;
    const bool ok = r0 + r < n;
    xs[r][c] = ok ? X[(int64_t)(r0 + r) * k + c] : 0.f;
    ys[r][c] = ok ? Yv[(int64_t)(r0 + r) * k + c] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < k * k; e += 256) {
    const int a = e / k, b = e - a * k;
    float s = 0.f;
    for (int r = 0; r < GR_ROWS; ++r) s = fmaf(xs[r][a], ys[r][b], s);
    part[(int64_t)blockIdx.x * k * k + e] = s;
  }
}
__global__ void gram_reduce_kernel(const float* __restrict__ part, int nblocks, int kk, float* __restrict__ G) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= kk) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += (double)part[(int64_t)b * kk + e];
  G[e] = (float)s;
}

// Y[i, :] = scale[i] * (X[i, :] @ R)   (R [k, k] row-major; scale may be null)
__global__ __launch_bounds__(256) void rows_apply_kernel(const float* __restrict__ X, const float* __restrict__ R,
                                                        const float* __restrict__ scale, int n, int k, float* __restrict__ Y) {
  __shared__ float rs[KV * KV];
  for (int i = threadIdx.x; i < k * k; i += 256) rs[i] = R[i];
  __syncthreads();
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)n * k) return;
  const int i = (int)(idx / k), c = (int)(idx - (int64_t)i * k);
  float s = 0.f;
  for (int a = 0; a < k; ++a) s = fmaf(X[(int64_t)i * k + a], rs[a * k + c], s);
  Y[idx] = scale ? s * scale[i] : s;
}

// rows scaled to unit length (k-means input)
__global__ __launch_bounds__(256) void rows_unit_kernel(const float* __restrict__ X, int n, int k, float* __restrict__ Y) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float ss = 0.f;
  for (int c = 0; c < k; ++c) ss = fmaf(X[(int64_t)i * k + c], X[(int64_t)i * k + c], ss);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  for (int c = 0; c < k; ++c) Y[(int64_t)i * k + c] = X[(int64_t)i * k + c] * inv;
}

// k-means step: label = nearest centre (ties -> lowest index), dist2 to it; per-block partial sums/counts
// accumulated in row order by a fixed (cluster, column) -> thread map (reproducible, no atomics).
__global__ __launch_bounds__(256) void kmeans_assign_kernel(const float* __restrict__ R, int n, int k, const float* __restrict__ Cn,
                                                           int kc, int32_t* __restrict__ label, float* __restrict__ dist2,
                                                           float* __restrict__ part_sum, int32_t* __restrict__ part_cnt) {
  __shared__ float cs[KV * KV];
  __shared__ float xs[256][KV + 1];
  __shared__ int lab[256];
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * 256;
  for (int i = tid; i < kc * k; i += 256) cs[i] = Cn[i];
  for (int i = tid; i < 256 * k; i += 256) {
    const int r = i / k, c = i - r * k;
    xs[r][c] = r0 + r < n ? R[(int64_t)(r0 + r) * k + c] : 0.f;
  }
  __syncthreads();
  const int i = r0 + tid;
  int best = -1;
  if (i < n) {
    float bd = INFINITY;
    for (int q = 0; q < kc; ++q) {
      float d = 0.f;
      for (int c = 0; c < k; ++c) {
        const float t = xs[tid][c] - cs[q * k + c];
        d = fmaf(t, t, d);
      }
      if (d < bd) { bd = d; best = q; }
    }
    label[i] = best;
    dist2[i] = bd;
  }
  lab[tid] = best;
  __syncthreads();
  if (part_sum) {
    for (int e = tid; e < kc * k; e += 256) {
      const int q = e / k, c = e - q * k;
      float sacc = 0.f;
      for (int r = 0; r < 256; ++r) sacc += lab[r] == q ? xs[r][c] : 0.f;
      part_sum[(int64_t)blockIdx.x * kc * k + e] = sacc;
    }
    if (tid < kc) {
      int cnt = 0;
      for (int r = 0; r < 256; ++r) cnt += lab[r] == tid;
      part_cnt[(int64_t)blockIdx.x * kc + tid] = cnt;
    }
  }
}

// min over chosen centres of squared distance (maximin initialisation): d2[i] = min(d2[i], |R[i] - c|^2)
__global__ __launch_bounds__(256) void kmeans_mindist_kernel(const float* __restrict__ R, int n, int k, const float* __restrict__ centre,
                                                            float* __restrict__ d2, int first) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float d = 0.f;
  for (int c = 0; c < k; ++c) {
    const float t = R[(int64_t)i * k + c] - centre[c];
    d = fmaf(t, t, d);
  }
  d2[i] = first ? d : fminf(d2[i], d);
}


// k x k (k <= 32) Cholesky-QR step on the device: G = sum_ranks Y^T Y  ->  Rinv with Y Rinv orthonormal, i.e.
// G_s = (G + G^T) / 2, G_s = L L^T, Rinv = (L^T)^-1, all in float64 by ONE lane (5 k flop at k = 16: the point is not speed but
// that the subspace iteration never leaves the stream - the host-side form synchronised twice per CholeskyQR pass).
__global__ __launch_bounds__(64) void chol_inverse_kernel(const float* __restrict__ G, int k, float* __restrict__ Rinv, int* __restrict__ bad,
                                                          double pivot_rtol, double shift_rel) {
  __shared__ double L[KV][KV + 1];
  __shared__ double X[KV][KV + 1];
  if (threadIdx.x != 0) return;
  int fail = 0;
  // shifted CholeskyQR ("chol_shift_ppb"): G + s I with s = shift_rel x the mean diagonal entry keeps a nearly rank-deficient block factorisable;
  // the factor is then only approximately orthonormalising, which the following unshifted passes repair (cluster._orth runs three in that mode)
  double shift = 0.0;
  if (shift_rel > 0.0) {
    for (int i = 0; i < k; ++i) shift += (double)G[i * k + i];
    shift *= shift_rel / (double)k;
  }
  for (int i = 0; i < k; ++i)
    for (int j = 0; j <= i; ++j) {
      double a = 0.5 * ((double)G[i * k + j] + (double)G[j * k + i]);
      if (i == j) a += 1e-30 + shift;
      for (int q = 0; q < j; ++q) a -= L[i][q] * L[j][q];
      if (i == j) {
        // a pivot at or below 1e-6 of its diagonal entry is rounding noise of the fp32 Gram matrix (cond(Y) > 1e3: the subspace
        // has lost rank) - flagged like a non-positive or NaN one; only an unusable pivot is replaced
        const double dii = (double)G[i * k + i] + shift;
        if (!(a > pivot_rtol * dii)) fail = 1;
        if (!(a > 0.0)) a = 1.0;
        L[i][i] = sqrt(a);
      } else {
        L[i][j] = a / L[j][j];
      }
    }
  // X = (L^T)^-1 : upper triangular, X[i][j] for i <= j;  sum_{q=i..j} U[i][q] X[q][j] = delta_ij with U = L^T (U[i][q] = L[q][i])
  for (int j = 0; j < k; ++j)
    for (int i = k - 1; i >= 0; --i) {
      if (i > j) { X[i][j] = 0.0; continue; }
      double a = i == j ? 1.0 : 0.0;
      for (int q = i + 1; q <= j; ++q) a -= L[q][i] * X[q][j];
      X[i][j] = a / L[i][i];
    }
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) Rinv[i * k + j] = (float)X[i][j];
  if (bad && fail) *bad = 1;   // sticky: the caller zeroes it once and reads it once, after any number of passes (cluster._orth)
}

}  // namespace

static size_t mv_xp_bytes(int N) { return (((size_t)((N + PT - 1) / PT) * 4 * 64 * 8 * sizeof(uint16_t)) + 255) & ~(size_t)255; }
// partial Y tiles: groups x maxp tiles of 64 KiB; maxp <= G / ngroups + 3, so groups x maxp <= G + 3 groups for any row block of <= N rows
static size_t mv_ypart_bytes(int N) { return ((size_t)MV_MAX_WG + 3 * (size_t)ceil_div(N, MV_SEGS)) * MV_SEGS * KV * sizeof(float); }

extern "C" size_t sdk_affinity_matvec_workspace_bytes(int N) {
  if (N <= 0) return 0;
  // packed X fragments | partial Y tiles of the parts of every row group | part counts
  return mv_xp_bytes(N) + mv_ypart_bytes(N) + (size_t)ceil_div(N, MV_SEGS) * sizeof(int32_t) + 256;
}

extern "C" int sdk_affinity_matvec_plan(int rows, int N, int num_cu, int32_t* out4, int64_t* units) {
  SDK_REQUIRE(out4 && units && rows > 0 && N >= rows, "sdk_affinity_matvec_plan: bad arguments");
  const MvGeom g = mv_geometry(rows, N, num_cu);
  out4[0] = g.ngroups; out4[1] = g.nst; out4[2] = g.G; out4[3] = g.maxp;
  *units = g.U;
  return 0;
}

extern "C" int sdk_affinity_matvec(sdk_ctx* ctx, const uint16_t* Eb, int N, int d, int row0, int rows, const float* X,
                                   const float* xscale, int kv, float* Y, void* ws, size_t ws_bytes, void* stream) {
  SDK_REQUIRE(ctx && Eb && X && Y && ws, "sdk_affinity_matvec: null argument");
  SDK_REQUIRE(d == D, "sdk_affinity_matvec: d=%d, this build is specialised for d=%d", d, D);
  SDK_REQUIRE(N > 0 && rows > 0 && row0 >= 0 && row0 + rows <= N, "sdk_affinity_matvec: bad row block [%d, %d) of %d", row0, row0 + rows, N);
  SDK_REQUIRE(kv >= 1 && kv <= KV, "sdk_affinity_matvec: kv=%d must be in [1, %d]", kv, KV);
  SDK_REQUIRE(ws_bytes >= sdk_affinity_matvec_workspace_bytes(N), "sdk_affinity_matvec: workspace too small");
  SDK_REQUIRE(((uintptr_t)Eb % 16) == 0 && ((uintptr_t)ws % 16) == 0, "sdk_affinity_matvec: Eb/ws must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int ntiles = ceil_div(N, PT);
  hipLaunchKernelGGL(pack_x_kernel, dim3(ceil_div(ntiles * 2 * 64, 256)), dim3(256), 0, s, X, N, kv, (bf16_t*)ws, xscale);
  SDK_LAUNCH_CHECK();
  if (ctx->matvec_variant == 1) {                    // A/B + test knob ("matvec_variant" 1): round 1's kernel, one 32-row block per wave
    ProfScope ps(ctx, stream, SDK_K_AFF_MATVEC, 2.0 * rows * (double)N * (D + KV), 2.0 * (double)N * D + 4.0 * N * kv);
    hipLaunchKernelGGL(affinity_matvec_kernel, dim3(ceil_div(rows, 128)), dim3(256), 0, s, (const bf16_t*)Eb, N, row0, rows,
                       (const bf16_t*)ws, kv, Y);
    SDK_LAUNCH_CHECK();
    return 0;
  }
  float* ypart = reinterpret_cast<float*>((char*)ws + mv_xp_bytes(N));
  int32_t* pcnt = reinterpret_cast<int32_t*>((char*)ws + mv_xp_bytes(N) + mv_ypart_bytes(N));
  const MvGeom gm = mv_geometry(rows, N, ctx->num_cu);
  SDK_REQUIRE((size_t)gm.ngroups * gm.maxp * MV_SEGS * KV * sizeof(float) <= mv_ypart_bytes(N), "sdk_affinity_matvec: internal: %d groups x %d parts exceed the workspace", gm.ngroups, gm.maxp);
  if (sdk_lds_optin(ctx, (const void*)affinity_matvec2_kernel, MV_LDS)) return 1;
  {
    ProfScope ps(ctx, stream, SDK_K_AFF_MATVEC, 2.0 * rows * (double)N * (D + KV), 2.0 * (double)N * D + 4.0 * N * kv);
    hipLaunchKernelGGL(affinity_matvec2_kernel, dim3(gm.G), dim3(MV_WAVES * 64), MV_LDS, s, (const bf16_t*)Eb, N, row0, rows, (const bf16_t*)ws, gm,
                       ypart, pcnt);
  }
  SDK_LAUNCH_CHECK();
  hipLaunchKernelGGL(matvec_reduce_kernel, dim3((unsigned)(((int64_t)rows * 8 + 255) / 256)), dim3(256), 0, s, (const float*)ypart, (const int32_t*)pcnt,
                     gm.maxp, row0, rows, kv, Y);
  SDK_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t sdk_rows_gram_workspace_bytes(int n, int k) { return n > 0 ? (size_t)ceil_div(n, GR_ROWS) * k * k * sizeof(float) : 0; }

extern "C" int sdk_rows_gram(sdk_ctx* ctx, const float* X, const float* Yv, int n, int k, float* G, void* ws, size_t ws_bytes,
                             void* stream) {
  SDK_REQUIRE(ctx && X && Yv && G && ws, "sdk_rows_gram: null argument");
  SDK_REQUIRE(n > 0 && k >= 1 && k <= KV, "sdk_rows_gram: bad shape n=%d k=%d", n, k);
  SDK_REQUIRE(ws_bytes >= sdk_rows_gram_workspace_bytes(n, k), "sdk_rows_gram: workspace too small");
  const int nb = ceil_div(n, GR_ROWS);
  hipLaunchKernelGGL(gram_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, X, Yv, n, k, (float*)ws);
  SDK_LAUNCH_CHECK();
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(ceil_div(k * k, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nb, k * k, G);
  SDK_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdk_rows_apply(sdk_ctx* ctx, const float* X, const float* R, const float* scale, int n, int k, float* Y, void* stream) {
  SDK_REQUIRE(ctx && X && R && Y, "sdk_rows_apply: null argument");
  SDK_REQUIRE(n > 0 && k >= 1 && k <= KV && X != Y, "sdk_rows_apply: bad arguments (n=%d k=%d, in-place not allowed)", n, k);
  hipLaunchKernelGGL(rows_apply_kernel, dim3((unsigned)(((int64_t)n * k + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, R, scale, n, k, Y);
  SDK_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdk_rows_unit(sdk_ctx* ctx, const float* X, int n, int k, float* Y, void* stream) {
  SDK_REQUIRE(ctx && X && Y && n > 0 && k >= 1 && k <= KV, "sdk_rows_unit: bad arguments");
  hipLaunchKernelGGL(rows_unit_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, X, n, k, Y);
  SDK_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdk_kmeans_assign(sdk_ctx* ctx, const float* R, int n, int k, const float* centres, int kc, int32_t* label,
                                 float* dist2, float* part_sum, int32_t* part_cnt, void* stream) {
  SDK_REQUIRE(ctx && R && centres && label && dist2, "sdk_kmeans_assign: null argument");
  SDK_REQUIRE(n > 0 && k >= 1 && k <= KV && kc >= 1 && kc <= KV, "sdk_kmeans_assign: bad shape n=%d k=%d kc=%d", n, k, kc);
  SDK_REQUIRE((part_sum == nullptr) == (part_cnt == nullptr), "sdk_kmeans_assign: part_sum and part_cnt go together");
  hipLaunchKernelGGL(kmeans_assign_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, R, n, k, centres, kc, label, dist2,
                     part_sum, part_cnt);
  SDK_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdk_kmeans_mindist(sdk_ctx* ctx, const float* R, int n, int k, const float* centre, float* d2, int first, void* stream) {
  SDK_REQUIRE(ctx && R && centre && d2 && n > 0 && k >= 1 && k <= KV, "sdk_kmeans_mindist: bad arguments");
  hipLaunchKernelGGL(kmeans_mindist_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, R, n, k, centre, d2, first);
  SDK_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdk_chol_inverse(sdk_ctx* ctx, const float* G, int k, float* Rinv, int32_t* not_spd, void* stream) {
  SDK_REQUIRE(ctx && G && Rinv && k >= 1 && k <= KV && G != Rinv, "sdk_chol_inverse: bad arguments (k=%d, in-place not allowed)", k);
  hipLaunchKernelGGL(chol_inverse_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, G, k, Rinv, not_spd, (double)ctx->chol_pivot_rtol_ppb * 1e-9,
                     (double)ctx->chol_shift_ppb * 1e-9);
  SDK_LAUNCH_CHECK();
  return 0;
}
