// k4: segments x profiles cosine affinity with fused top-k on gfx950.
//
// What a local backend's identify_speaker (speaker_detection_backends/base.py:130-151) needs per
// segment is "the best enrolled profiles and their scores", never the N x P score matrix.  So:
//
//  coarse pass (affinity_coarse_kernel)   bf16 MFMA, S^T tile = P_tile[32 x 192] . E_tile[32 x 192]^T
//      with the PROFILE index on the accumulator registers and the SEGMENT on the lane: every lane
//      owns one segment and keeps a sorted top-4 of the profiles it has seen in registers
//      (insert = 1 v_max + 3 v_med3 on values whose low 10 mantissa bits carry the profile index).
//      Nothing but 8 candidates per segment is written to HBM.
//  exact pass (affinity_rescore_kernel)   fp32 re-score of the candidates that can still win, fixed
//      summation order; certifies the result against the rigorous rounding bound
//          |exact - coarse| <= r_e + (1 + r_e) r_p + 2^-13 + K 2^-23 =: eps
//      (r_e, r_p = measured bf16 rounding residual norms from sdk_l2norm): any profile that is not a
//      candidate scores at most u + eps, u = the larger of the two lane-halves' 4th-best coarse
//      score.  Rows with x_k <= u + eps are queued for
//  exact rescan (affinity_rescan_kernel)  fp32 scan over all P with the same dot-product routine.
// The reported (idx, score) therefore equal an fp32 full scan: ties -> lowest profile index.
#include "common.hpp"

namespace {

constexpr int D = 192;                 // embedding width (12 MFMA k-steps of 16)
constexpr int KS = D / 16;
constexpr int SEG_PER_WAVE = 32;
constexpr int WAVES = 4;
constexpr int SEG_PER_WG = SEG_PER_WAVE * WAVES;   // 128
constexpr int PT = 32;                 // profiles per tile
constexpr int PROW = 400;              // LDS bytes per profile row (384 + 16 pad: 25 slots, odd -> conflict-free)
constexpr int CHUNK_TILES = 32;        // 1024 profiles share one 10-bit index space
constexpr uint32_t IDX_MASK = 0x3ffu;
constexpr int NCAND = 8;

struct Workspace {        // layout inside the caller's scratch buffer
  float* cand_val;        // [N][8] coarse value (index bits stripped)
  int32_t* cand_idx;      // [N][8] profile index, -1 = empty
  float* ubound;          // [N]    u (see header comment)
  int32_t* flag_count;    // [1]
  int32_t* flag_rows;     // [N]
};

__host__ __device__ inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

inline size_t ws_layout(int N, char* base, Workspace* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
  char* a = take((size_t)N * NCAND * 4);
  char* b = take((size_t)N * NCAND * 4);
  char* c = take((size_t)N * 4);
  char* d = take(256);
  char* e = take((size_t)N * 4);
  if (w) { w->cand_val = (float*)a; w->cand_idx = (int32_t*)b; w->ubound = (float*)c; w->flag_count = (int32_t*)d; w->flag_rows = (int32_t*)e; }
  return off;
}

// sorted insert of x into (m0 >= m1 >= m2 >= m3); values are packed floats
__device__ __forceinline__ void insert4(float x, float& m0, float& m1, float& m2, float& m3) {
  const float n3 = __builtin_amdgcn_fmed3f(x, m2, m3);
  const float n2 = __builtin_amdgcn_fmed3f(x, m1, m2);
  const float n1 = __builtin_amdgcn_fmed3f(x, m0, m1);
  m0 = fmaxf(x, m0);
  m1 = n1; m2 = n2; m3 = n3;
}

__global__ __launch_bounds__(WAVES * 64, 2) void affinity_coarse_kernel(const bf16_t* __restrict__ Eb,
                                                                       const bf16_t* __restrict__ Pb, int N, int P,
                                                                       float* __restrict__ cand_val,
                                                                       int32_t* __restrict__ cand_idx,
                                                                       float* __restrict__ ubound) {
  __shared__ __attribute__((aligned(16))) char sP[2][PT * PROW];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int seg = blockIdx.x * SEG_PER_WG + wid * SEG_PER_WAVE + col;
  const int seg_c = seg < N ? seg : N - 1;

  // B operand (segments): B[k = 16 ks + 8h + j][col] = Eb[seg][16 ks + 8h + j]; resident for the whole sweep
  bf16x8 bfrag[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    bfrag[ks] = *reinterpret_cast<const bf16x8*>(Eb + (int64_t)seg_c * D + ks * 16 + h * 8);

  // staging: 32 rows x 24 chunks of 16 B = 768 chunks, 3 per thread
  const int ntiles = (P + PT - 1) / PT;
  u32x4 st[3];
  auto gload = [&](int tile) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int id = tid + 256 * i;
      const int row = id / 24, ch = id - row * 24;
      int pr = tile * PT + row;
      pr = pr < P ? pr : P - 1;
      st[i] = *reinterpret_cast<const u32x4*>(Pb + (int64_t)pr * D + ch * 8);
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int id = tid + 256 * i;
      const int row = id / 24, ch = id - row * 24;
      *reinterpret_cast<u32x4*>(&sP[buf][row * PROW + ch * 16]) = st[i];
    }
  };

  // global candidate list of this lane (its half of the profiles): value + full index
  float gv[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int gi[4] = {-1, -1, -1, -1};
  // chunk-local packed list
  float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;

  auto merge_chunk = [&](int chunk) {
    float mv[4] = {m0, m1, m2, m3};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t bits = __float_as_uint(mv[e]);
      if (mv[e] == -INFINITY) continue;
      const float v = __uint_as_float(bits & ~IDX_MASK);
      const int idx = chunk * (CHUNK_TILES * PT) + (int)(bits & IDX_MASK);
      // insert (v, idx) into the sorted global list
      int pos = 4;
#pragma unroll
      for (int q = 3; q >= 0; --q)
        if (v > gv[q]) pos = q;
#pragma unroll
      for (int q = 3; q >= 1; --q)
        if (q > pos) { gv[q] = gv[q - 1]; gi[q] = gi[q - 1]; }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q == pos) { gv[q] = v; gi[q] = idx; }
    }
    m0 = m1 = m2 = m3 = -INFINITY;
  };

  gload(0);
  swrite(0);
  __syncthreads();
  const int arow = col * PROW + h * 16;
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) gload(t + 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(&sP[buf][arow + ks * 32]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfrag[ks], acc, 0, 0, 0);
    }
    const int tl = t & (CHUNK_TILES - 1);
    const uint32_t tb = (uint32_t)(tl * PT + 4 * h);
    const bool partial = (t + 1) * PT > P;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t rc = (uint32_t)((r & 3) + 8 * (r >> 2));
      float x = __uint_as_float(((__float_as_uint(acc[r]) & ~IDX_MASK) | tb) | rc);
      if (partial && (t * PT + (int)rc + 4 * h) >= P) x = -INFINITY;   // rows past the last profile
      insert4(x, m0, m1, m2, m3);
    }
    if (tl == CHUNK_TILES - 1 || t + 1 == ntiles) merge_chunk(t / CHUNK_TILES);
    if (t + 1 < ntiles) swrite(buf ^ 1);
    __syncthreads();
  }
  if (seg < N) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cand_val[(int64_t)seg * NCAND + h * 4 + e] = gv[e];
      cand_idx[(int64_t)seg * NCAND + h * 4 + e] = gi[e];
    }
    // u = max over the two halves of their 4th best (a bound on every profile that is not a candidate)
    const float other = __shfl_xor(gv[3], 32, 64);
    if (h == 0) ubound[seg] = fmaxf(gv[3], other);
  }
}

// ---- exact fp32 dot product of two 192-vectors by a group of 8 consecutive lanes ---------------
// lane j of the group owns elements [24 j, 24 j + 24); fixed order: sequential fma inside the lane,
// then the xor-butterfly 1,2,4 (fp add is commutative, so all 8 lanes hold the same bits).
__device__ __forceinline__ float dot192_group8(const float* __restrict__ e24, const float* __restrict__ prow, int j) {
  const f32x4* p = reinterpret_cast<const f32x4*>(prow + 24 * j);
  float a = 0.f;
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const f32x4 v = p[q];
    a = fmaf(e24[4 * q + 0], v[0], a);
    a = fmaf(e24[4 * q + 1], v[1], a);
    a = fmaf(e24[4 * q + 2], v[2], a);
    a = fmaf(e24[4 * q + 3], v[3], a);
  }
  a += __shfl_xor(a, 1, 64);
  a += __shfl_xor(a, 2, 64);
  a += __shfl_xor(a, 4, 64);
  return a;
}

__device__ __forceinline__ bool better(float s, int i, float s2, int i2) { return s > s2 || (s == s2 && i < i2); }

// insert (s, i) into a best-first list of length K kept in registers
template <int K>
__device__ __forceinline__ void insert_exact(float s, int i, float* bs, int* bi) {
  int pos = K;
#pragma unroll
  for (int q = K - 1; q >= 0; --q)
    if (better(s, i, bs[q], bi[q])) pos = q;
#pragma unroll
  for (int q = K - 1; q >= 1; --q)
    if (q > pos) { bs[q] = bs[q - 1]; bi[q] = bi[q - 1]; }
#pragma unroll
  for (int q = 0; q < K; ++q)
    if (q == pos) { bs[q] = s; bi[q] = i; }
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
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(E + (int64_t)rc * D + 24 * j + 4 * q);
    e24[4 * q] = v[0]; e24[4 * q + 1] = v[1]; e24[4 * q + 2] = v[2]; e24[4 * q + 3] = v[3];
  }
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

// One workgroup per uncertain row: 32 lane-groups scan P/32 profiles each, then merge.
__global__ __launch_bounds__(256) void affinity_rescan_kernel(const float* __restrict__ E, const float* __restrict__ Pm,
                                                             int P, int k, const int32_t* __restrict__ flag_count,
                                                             const int32_t* __restrict__ flag_rows,
                                                             int32_t* __restrict__ idx, float* __restrict__ score) {
  __shared__ float ls[32][4];
  __shared__ int li[32][4];
  const int tid = threadIdx.x, j = tid & 7, g = tid >> 3;
  const int count = *flag_count;
  for (int f = blockIdx.x; f < count; f += gridDim.x) {
    const int row = flag_rows[f];
    float e24[24];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(E + (int64_t)row * D + 24 * j + 4 * q);
      e24[4 * q] = v[0]; e24[4 * q + 1] = v[1]; e24[4 * q + 2] = v[2]; e24[4 * q + 3] = v[3];
    }
    float bs[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    for (int p = g; p < P; p += 32) {
      const float s = dot192_group8(e24, Pm + (int64_t)p * D, j);
      insert_exact<4>(s, p, bs, bi);
    }
    if (j == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { ls[g][q] = bs[q]; li[g][q] = bi[q]; }
    }
    __syncthreads();
    if (tid == 0) {
      float fs[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int fi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
      for (int gg = 0; gg < 32; ++gg)
        for (int q = 0; q < 4; ++q)
          if (li[gg][q] != 0x7fffffff) insert_exact<4>(ls[gg][q], li[gg][q], fs, fi);
      for (int q = 0; q < k; ++q) {
        idx[(int64_t)row * k + q] = fi[q];
        score[(int64_t)row * k + q] = fs[q];
      }
    }
    __syncthreads();
  }
}

__global__ void copy_count_kernel(const int32_t* src, int32_t* dst) { *dst = *src; }

}  // namespace

extern "C" size_t sdk_affinity_workspace_bytes(int N) { return N > 0 ? ws_layout(N, nullptr, nullptr) : 0; }

extern "C" int sdk_affinity_topk(sdk_ctx* ctx, const float* E, const uint16_t* Eb, const float* resid_e,
                                 const float* P, const uint16_t* Pb, const float* resid_p, int N, int Pn, int d, int k,
                                 int32_t* idx, float* score, int32_t* n_rescanned, void* ws, size_t ws_bytes,
                                 void* stream) {
  SDK_REQUIRE(ctx && E && Eb && resid_e && P && Pb && resid_p && idx && score && ws, "sdk_affinity_topk: null argument");
  SDK_REQUIRE(d == D, "sdk_affinity_topk: d=%d, this build is specialised for d=%d", d, D);
  SDK_REQUIRE(N > 0 && Pn > 0, "sdk_affinity_topk: empty problem (N=%d P=%d)", N, Pn);
  SDK_REQUIRE(k >= 1 && k <= 4 && k <= Pn, "sdk_affinity_topk: k=%d must be in [1, min(4, P)]", k);
  SDK_REQUIRE(ws_bytes >= sdk_affinity_workspace_bytes(N), "sdk_affinity_topk: workspace too small");
  SDK_REQUIRE(((uintptr_t)E % 16) == 0 && ((uintptr_t)Eb % 16) == 0 && ((uintptr_t)P % 16) == 0 && ((uintptr_t)Pb % 16) == 0,
              "sdk_affinity_topk: matrices must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  Workspace w;
  ws_layout(N, (char*)ws, &w);
  SDK_HIP_OK(hipMemsetAsync(w.flag_count, 0, sizeof(int32_t), s));
  {
  ProfScope ps(ctx, stream, SDK_K_AFF_COARSE, 2.0 * N * (double)Pn * D, 2.0 * ((double)N + Pn) * D + 68.0 * N);
  hipLaunchKernelGGL(affinity_coarse_kernel, dim3(ceil_div(N, SEG_PER_WG)), dim3(WAVES * 64), 0, s, (const bf16_t*)Eb,
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
  hipLaunchKernelGGL(affinity_rescan_kernel, dim3(1024), dim3(256), 0, s, E, P, Pn, k, w.flag_count, w.flag_rows, idx,
                     score);
  }
  SDK_LAUNCH_CHECK();
  if (n_rescanned) {
    hipLaunchKernelGGL(copy_count_kernel, dim3(1), dim3(1), 0, s, w.flag_count, n_rescanned);
    SDK_LAUNCH_CHECK();
  }
  return 0;
}
