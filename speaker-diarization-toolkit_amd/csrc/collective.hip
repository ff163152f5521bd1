// k5 / k6 drivers for a host that is NOT Python (SURVEY.md section 8b lists sdk_allgather and sdk_laplacian_topk among the exports):
//   sdk_allgather       one RCCL all-gather of equal shards on the caller's communicator and stream (the embedding exchange over xGMI);
//   sdk_laplacian_topk  the top-k eigenpairs of S = D^-1/2 A D^-1/2, A = max(E E^T, 0), by row-sharded subspace iteration: exactly the
//                       loop of cluster.spectral_cluster (degrees, CholeskyQR2, n_iter x [V all-gather, recomputed-affinity mat-vec,
//                       scaling, CholeskyQR2], Ritz) with every step on the stream - the k x k Ritz problem is solved on the device
//                       too (cyclic Jacobi in float64 by one lane), so the call never synchronises with the host.
// The library does not link RCCL: the symbols are resolved at first use from the copy the process already has (torch's) or from
// librccl.so.1, so a single-GPU host never needs it (comm == NULL skips every collective).  The Python host keeps using
// torch.distributed (dist.py): same collectives, same order.
#include <dlfcn.h>
#include <string.h>

#include "common.hpp"

namespace {

typedef int (*nccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*nccl_allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_sendrecv_t)(const void*, size_t, int, int, void*, hipStream_t);      // ncclSend (const buffer) / ncclRecv (same shape, buffer written)
typedef int (*nccl_group_t)(void);
typedef int (*nccl_count_t)(void*, int*);
typedef const char* (*nccl_errstr_t)(int);
constexpr int NCCL_CHAR = 0, NCCL_FLOAT32 = 7, NCCL_SUM = 0;

struct Rccl {
  nccl_allgather_t all_gather = nullptr;
  nccl_allreduce_t all_reduce = nullptr;
  nccl_sendrecv_t send = nullptr, recv = nullptr;
  nccl_group_t group_start = nullptr, group_end = nullptr;
  nccl_count_t comm_count = nullptr, comm_rank = nullptr;
  nccl_errstr_t errstr = nullptr;
  bool tried = false;
};
Rccl g_rccl;

int rccl_resolve() {
  if (g_rccl.all_gather) return 0;
  if (!g_rccl.tried) {
    g_rccl.tried = true;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);            // a copy already in the process (same soname) is returned, not loaded twice
      if (h) break;
    }
    if (h) {
      g_rccl.all_gather = (nccl_allgather_t)dlsym(h, "ncclAllGather");
      g_rccl.all_reduce = (nccl_allreduce_t)dlsym(h, "ncclAllReduce");
      g_rccl.send = (nccl_sendrecv_t)dlsym(h, "ncclSend");
      g_rccl.recv = (nccl_sendrecv_t)dlsym(h, "ncclRecv");
      g_rccl.group_start = (nccl_group_t)dlsym(h, "ncclGroupStart");
      g_rccl.group_end = (nccl_group_t)dlsym(h, "ncclGroupEnd");
      g_rccl.comm_count = (nccl_count_t)dlsym(h, "ncclCommCount");
      g_rccl.comm_rank = (nccl_count_t)dlsym(h, "ncclCommUserRank");
      g_rccl.errstr = (nccl_errstr_t)dlsym(h, "ncclGetErrorString");
    }
  }
  if (!g_rccl.all_gather || !g_rccl.all_reduce) {
    const char* why = dlerror();          // (one call: dlerror() clears the message it returns)
    sdk_set_error("RCCL not available: librccl.so.1 could not be loaded (%s); multi-GPU entry points need it", why ? why : "symbols missing");
    return 1;
  }
  return 0;
}

#define SDK_NCCL_OK(expr)                                                                                  \
  do {                                                                                                      \
    int _r = (expr);                                                                                        \
    if (_r != 0) {                                                                                          \
      sdk_set_error("%s failed: %s", #expr, g_rccl.errstr ? g_rccl.errstr(_r) : "RCCL error");             \
      return 1;                                                                                             \
    }                                                                                                       \
  } while (0)

constexpr int KV = 32;

__global__ void fill_kernel(float* p, int64_t n, float v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void rsqrt_rows_kernel(const float* __restrict__ Y, int64_t ld, int row0, int rows, float* __restrict__ out) {   // out[i] = 1 / sqrt(Y[row0 + i, 0])
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < rows) out[i] = rsqrtf(Y[(int64_t)(row0 + i) * ld]);
}
__global__ void copy_rows_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int64_t ldd, int rows, int k) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (int64_t)rows * k) dst[(i / k) * ldd + i % k] = src[(i / k) * lds + i % k];
}
__global__ void eye_kernel(float* R, int k) {
  const int i = threadIdx.x;
  if (i < k * k) R[i] = (i / k == i % k) ? 1.f : 0.f;
}

// Ritz step on the device: H = (G + G^T) / 2 (k x k, k <= 32), cyclic Jacobi in float64 by ONE lane (k = 16: ~10 sweeps x 120 rotations),
// eigenvalues sorted descending, Q [k, k] row-major with the eigenvectors in its columns, each normalised so that its largest-magnitude
// component is positive (a deterministic sign: the caller's k-means is invariant to it anyway).
__global__ __launch_bounds__(64) void jacobi_eigh_kernel(const float* __restrict__ G, int k, float* __restrict__ Q, float* __restrict__ lam) {
  __shared__ double A[KV][KV + 1];
  __shared__ double V[KV][KV + 1];
  if (threadIdx.x != 0) return;
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) {
      A[i][j] = 0.5 * ((double)G[i * k + j] + (double)G[j * k + i]);
      V[i][j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < k; ++i) {
      diag += A[i][i] * A[i][i];
      for (int j = i + 1; j < k; ++j) off += A[i][j] * A[i][j];
    }
    if (off <= 1e-30 * (diag + 1e-300)) break;
    for (int p = 0; p < k - 1; ++p)
      for (int q = p + 1; q < k; ++q) {
        const double apq = A[p][q];
        if (apq == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int r = 0; r < k; ++r) {
          const double arp = A[r][p], arq = A[r][q];
          A[r][p] = c * arp - s * arq;
          A[r][q] = s * arp + c * arq;
        }
        for (int r = 0; r < k; ++r) {
          const double apr = A[p][r], aqr = A[q][r];
          A[p][r] = c * apr - s * aqr;
          A[q][r] = s * apr + c * aqr;
        }
        for (int r = 0; r < k; ++r) {
          const double vrp = V[r][p], vrq = V[r][q];
          V[r][p] = c * vrp - s * vrq;
          V[r][q] = s * vrp + c * vrq;
        }
      }
  }
  int order[KV];
  for (int i = 0; i < k; ++i) order[i] = i;
  for (int i = 0; i < k; ++i)                                  // selection sort, descending, ties -> lower index first
    for (int j = i + 1; j < k; ++j)
      if (A[order[j]][order[j]] > A[order[i]][order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
  for (int c = 0; c < k; ++c) {
    const int src = order[c];
    lam[c] = (float)A[src][src];
    int big = 0;
    for (int r = 1; r < k; ++r)
      if (fabs(V[r][src]) > fabs(V[big][src])) big = r;
    const double sgn = V[big][src] < 0 ? -1.0 : 1.0;
    for (int r = 0; r < k; ++r) Q[r * k + c] = (float)(sgn * V[r][src]);
  }
}

inline size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }

struct LtWs {
  float *Xall, *Y, *dinv_all, *dinv_loc, *V2, *SV, *G, *Rinv, *eye, *Q;
  int32_t* flag;
  char *mv, *gram;
  size_t mv_bytes, gram_bytes;
};
size_t lt_layout(int N, int k, char* base, LtWs* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += a256(bytes); return p; };
  char* xall = take((size_t)N * k * 4);
  char* y = take((size_t)N * k * 4);
  char* da = take((size_t)N * 4);
  char* dl = take((size_t)N * 4);
  char* v2 = take((size_t)N * k * 4);
  char* sv = take((size_t)N * k * 4);
  char* g = take(KV * KV * 4);
  char* ri = take(KV * KV * 4);
  char* ey = take(KV * KV * 4);
  char* q = take(KV * KV * 4);
  char* fl = take(256);
  const size_t mvb = sdk_affinity_matvec_workspace_bytes(N), grb = sdk_rows_gram_workspace_bytes(N, k);
  char* mv = take(mvb);
  char* gr = take(grb);
  if (w) {
    w->Xall = (float*)xall; w->Y = (float*)y; w->dinv_all = (float*)da; w->dinv_loc = (float*)dl; w->V2 = (float*)v2; w->SV = (float*)sv;
    w->G = (float*)g; w->Rinv = (float*)ri; w->eye = (float*)ey; w->Q = (float*)q; w->flag = (int32_t*)fl; w->mv = mv; w->gram = gr;
    w->mv_bytes = mvb; w->gram_bytes = grb;
  }
  return off;
}

}  // namespace

extern "C" int sdk_allgather(sdk_ctx* ctx, const void* shard, void* out, size_t bytes_per_rank, void* comm, void* stream) {
  SDK_REQUIRE(ctx && shard && out && comm, "sdk_allgather: null argument (comm is the caller's ncclComm_t)");
  SDK_REQUIRE(bytes_per_rank > 0, "sdk_allgather: empty shard");
  if (rccl_resolve()) return 1;
  SDK_NCCL_OK(g_rccl.all_gather(shard, out, bytes_per_rank, NCCL_CHAR, comm, (hipStream_t)stream));
  return 0;
}

// The same exchange as world - 1 PAIRWISE transfers in one RCCL group (every rank sends its shard to every peer and receives every peer's):
// on a node whose GPUs are fully connected by point-to-point xGMI links (MI355X: 7 links x ~153 GB/s per GPU) each transfer takes the direct link
// of its pair and all seven run at once - the form SURVEY.md 5 / 8(e) asks for (floor 0.63 ms for 8 x 96-MB shards), whatever algorithm the
// library's own ncclAllGather would pick for the message size (a ring over the same links is 7 hops of one link each: 4.4 ms).  Same result as
// sdk_allgather, byte for byte.  Unmeasured on a multi-GPU node (none has been available to this build): bench.py --gpus N times both forms.
extern "C" int sdk_allgather_direct(sdk_ctx* ctx, const void* shard, void* out, size_t bytes_per_rank, void* comm, void* stream) {
  SDK_REQUIRE(ctx && shard && out && comm, "sdk_allgather_direct: null argument (comm is the caller's ncclComm_t)");
  SDK_REQUIRE(bytes_per_rank > 0, "sdk_allgather_direct: empty shard");
  if (rccl_resolve()) return 1;
  SDK_REQUIRE(g_rccl.send && g_rccl.recv && g_rccl.group_start && g_rccl.group_end && g_rccl.comm_count && g_rccl.comm_rank,
              "sdk_allgather_direct: this RCCL has no ncclSend / ncclRecv / ncclGroup* / ncclCommCount / ncclCommUserRank");
  int world = 0, rank = -1;
  SDK_NCCL_OK(g_rccl.comm_count(comm, &world));
  SDK_NCCL_OK(g_rccl.comm_rank(comm, &rank));
  SDK_REQUIRE(world >= 1 && rank >= 0 && rank < world, "sdk_allgather_direct: communicator reports rank %d of %d", rank, world);
  hipStream_t s = (hipStream_t)stream;
  char* o = (char*)out;
  if ((const char*)shard != o + (size_t)rank * bytes_per_rank)
    SDK_HIP_OK(hipMemcpyAsync(o + (size_t)rank * bytes_per_rank, shard, bytes_per_rank, hipMemcpyDeviceToDevice, s));
  if (world == 1) return 0;
  SDK_NCCL_OK(g_rccl.group_start());
  int rc = 0;
  for (int d = 1; d < world && !rc; ++d) {                   // peer at distance d: every pair (i, i + d) is scheduled by both ends in the same group
    const int to = (rank + d) % world, from = (rank - d + world) % world;
    rc = g_rccl.send(shard, bytes_per_rank, NCCL_CHAR, to, comm, s);
    if (!rc) rc = g_rccl.recv(o + (size_t)from * bytes_per_rank, bytes_per_rank, NCCL_CHAR, from, comm, s);
  }
  const int rc_end = g_rccl.group_end();
  if (rc || rc_end) {
    sdk_set_error("sdk_allgather_direct: RCCL error %d (%s)", rc ? rc : rc_end, g_rccl.errstr ? g_rccl.errstr(rc ? rc : rc_end) : "?");
    return 1;
  }
  return 0;
}

extern "C" size_t sdk_laplacian_topk_workspace_bytes(int N, int k) {
  if (N <= 0 || k < 1 || k > KV) return 0;
  return lt_layout(N, k, nullptr, nullptr);
}

extern "C" int sdk_laplacian_topk(sdk_ctx* ctx, const uint16_t* Eb_all, int N, int row0, int rows, int k, int n_iter, float* V,
                                  float* eigvals, int32_t* not_spd, void* ws, size_t ws_bytes, void* comm, int world, void* stream) {
  SDK_REQUIRE(ctx && Eb_all && V && eigvals && ws, "sdk_laplacian_topk: null argument");
  SDK_REQUIRE(N > 0 && rows > 0 && row0 >= 0 && row0 + rows <= N && k >= 1 && k <= KV && n_iter >= 0, "sdk_laplacian_topk: bad shape (N=%d rows=%d k=%d)", N, rows, k);
  SDK_REQUIRE(ws_bytes >= sdk_laplacian_topk_workspace_bytes(N, k) && ((uintptr_t)ws % 256) == 0, "sdk_laplacian_topk: workspace too small or misaligned");
  if (comm) {
    SDK_REQUIRE(world >= 1 && (int64_t)rows * world == N && row0 % rows == 0, "sdk_laplacian_topk: with a communicator every rank owns N / world = %d rows (got rows=%d row0=%d)", N / (world > 0 ? world : 1), rows, row0);
    if (rccl_resolve()) return 1;
  } else {
    SDK_REQUIRE(rows == N && row0 == 0, "sdk_laplacian_topk: without a communicator the call owns all N rows");
  }
  hipStream_t s = (hipStream_t)stream;
  LtWs w;
  lt_layout(N, k, (char*)ws, &w);
  auto grid = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  auto gather = [&](const float* loc, float* all, int width) -> int {     // rows x width floats per rank -> N x width
    if (!comm) {
      if (loc != all) SDK_HIP_OK(hipMemcpyAsync(all, loc, (size_t)rows * width * 4, hipMemcpyDeviceToDevice, s));
      return 0;
    }
    SDK_NCCL_OK(g_rccl.all_gather(loc, all, (size_t)rows * width, NCCL_FLOAT32, comm, s));
    return 0;
  };
  auto allreduce = [&](float* x, int n) -> int {
    if (!comm) return 0;
    SDK_NCCL_OK(g_rccl.all_reduce(x, x, (size_t)n, NCCL_FLOAT32, NCCL_SUM, comm, s));
    return 0;
  };
  auto orth = [&](float* Yin, float* tmp) -> int {                        // CholeskyQR2: Yin -> tmp -> Yin
    float *a = Yin, *b = tmp;
    for (int pass = 0; pass < 2; ++pass) {
      if (int rc = sdk_rows_gram(ctx, a, a, rows, k, w.G, w.gram, w.gram_bytes, stream)) return rc;
      if (int rc = allreduce(w.G, k * k)) return rc;
      if (int rc = sdk_chol_inverse(ctx, w.G, k, w.Rinv, not_spd, stream)) return rc;
      if (int rc = sdk_rows_apply(ctx, a, w.Rinv, nullptr, rows, k, b, stream)) return rc;
      float* t = a; a = b; b = t;
    }
    return 0;                                                             // two passes: the result is back in Yin
  };
  // degrees: A 1 over the owned rows -> D^-1/2, gathered
  hipLaunchKernelGGL(fill_kernel, grid(N), dim3(256), 0, s, w.Xall, (int64_t)N, 1.0f);
  SDK_LAUNCH_CHECK();
  if (int rc = sdk_affinity_matvec(ctx, Eb_all, N, 192, row0, rows, w.Xall, nullptr, 1, w.Y, w.mv, w.mv_bytes, stream)) return rc;
  hipLaunchKernelGGL(rsqrt_rows_kernel, grid(rows), dim3(256), 0, s, (const float*)w.Y, (int64_t)1, row0, rows, w.dinv_loc);
  SDK_LAUNCH_CHECK();
  if (int rc = gather(w.dinv_loc, w.dinv_all, 1)) return rc;
  hipLaunchKernelGGL(eye_kernel, dim3(1), dim3(1024), 0, s, w.eye, k);
  SDK_LAUNCH_CHECK();
  if (int rc = orth(V, w.V2)) return rc;
  auto apply_S = [&](const float* Vloc, float* out) -> int {              // out = D^-1/2 A D^-1/2 V on the owned rows
    if (int rc = gather(Vloc, w.Xall, k)) return rc;
    if (int rc = sdk_affinity_matvec(ctx, Eb_all, N, 192, row0, rows, w.Xall, w.dinv_all, k, w.Y, w.mv, w.mv_bytes, stream)) return rc;
    return sdk_rows_apply(ctx, w.Y + (int64_t)row0 * k, w.eye, w.dinv_loc, rows, k, out, stream);
  };
  for (int it = 0; it < n_iter; ++it) {
    if (int rc = apply_S(V, w.SV)) return rc;
    SDK_HIP_OK(hipMemcpyAsync(V, w.SV, (size_t)rows * k * 4, hipMemcpyDeviceToDevice, s));
    if (int rc = orth(V, w.V2)) return rc;
  }
  // Ritz: H = V^T S V, eigh on the device, U = V Q (eigenvalues descending)
  if (int rc = apply_S(V, w.SV)) return rc;
  if (int rc = sdk_rows_gram(ctx, V, w.SV, rows, k, w.G, w.gram, w.gram_bytes, stream)) return rc;
  if (int rc = allreduce(w.G, k * k)) return rc;
  hipLaunchKernelGGL(jacobi_eigh_kernel, dim3(1), dim3(64), 0, s, (const float*)w.G, k, w.Q, eigvals);
  SDK_LAUNCH_CHECK();
  if (int rc = sdk_rows_apply(ctx, V, w.Q, nullptr, rows, k, w.V2, stream)) return rc;
  SDK_HIP_OK(hipMemcpyAsync(V, w.V2, (size_t)rows * k * 4, hipMemcpyDeviceToDevice, s));
  return 0;
}
