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
__global__ __launch_bounds__(64) void chol_inverse_kernel(const float* __restrict__ G, int k, float* __restrict__ Rinv, int* __restrict__ bad) {
  __shared__ double L[KV][KV + 1];
  __shared__ double X[KV][KV + 1];
  if (threadIdx.x != 0) return;
  int fail = 0;
  for (int i = 0; i < k; ++i)
    for (int j = 0; j <= i; ++j) {
      double a = 0.5 * ((double)G[i * k + j] + (double)G[j * k + i]);
      if (i == j) a += 1e-30;
      for (int q = 0; q < j; ++q) a -= L[i][q] * L[j][q];
      if (i == j) {
        // a pivot at or below 1e-6 of its diagonal entry is rounding noise of the fp32 Gram matrix (cond(Y) > 1e3: the subspace
        // has lost rank) - flagged like a non-positive or NaN one; only an unusable pivot is replaced
        const double dii = (double)G[i * k + i];
        if (!(a > 1e-6 * dii)) fail = 1;
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

extern "C" size_t sdk_affinity_matvec_workspace_bytes(int N) {
  if (N <= 0) return 0;
  return (size_t)((N + PT - 1) / PT) * 4 * 64 * 8 * sizeof(uint16_t);
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
  {
    ProfScope ps(ctx, stream, SDK_K_AFF_MATVEC, 2.0 * rows * (double)N * (D + KV), 2.0 * (double)N * D + 4.0 * N * kv);
    hipLaunchKernelGGL(affinity_matvec_kernel, dim3(ceil_div(rows, 128)), dim3(256), 0, s, (const bf16_t*)Eb, N, row0, rows,
                       (const bf16_t*)ws, kv, Y);
  }
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
  hipLaunchKernelGGL(chol_inverse_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, G, k, Rinv, not_spd);
  SDK_LAUNCH_CHECK();
  return 0;
}
