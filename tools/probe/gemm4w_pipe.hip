// Round 5 probe: the 4-wave / 512-register 256x256x64 bf16 GEMM of round 4 (gemm4w_bench.hip: accumulators in AGPRs, one wave per SIMD, LDS-DMA
// staging) with a SOFTWARE-PIPELINED K loop: round 4's probe issued a K-step's 16 fragment reads and 16 DMA pieces in bursts with nothing to cover
// them (no partner wave on the SIMD) and reached 689 / 986 TF.  Here every phase of 64 MFMAs carries the next phase's 16 ds_read_b128 and, in the
// second phase, the 16 DMA pieces of the K-step after next, dealt out one per 4 MFMAs by sched_group_barrier.  Bare (one workgroup per tile, plain
// order, 8-byte epilogue stores).  Build: hipcc --offload-arch=gfx950 -O3 -o gemm4w_pipe gemm4w_pipe.hip ; run: ./gemm4w_pipe [launches per block]
#include <string.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#ifndef MI
#define MI 8
#endif
constexpr int BM = 32 * MI, BN = 256, BK = 64, NT = 256;      // MI = 16-row blocks per wave (8: 256 x 256 tile, 6: 192 x 256)
constexpr int STAGE = (BM + BN) * BK * 2;
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

template <int VARIANT>
__global__ __launch_bounds__(NT, 1) void gemm4w(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, uint16_t* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fq = lane >> 4, sw = lane & 7;
  const int nbn = N / BN;
  const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int rin = lane >> 3, pos = lane & 7, gch = (pos ^ rin) * 8;
  uint32_t aoff[MI], woff[8];
#pragma unroll
  for (int i = 0; i < MI; ++i) aoff[i] = ((uint32_t)min(m0 + 8 * MI * wid + rin + 8 * i, M - 1) * (uint32_t)K + gch) * 2u;
#pragma unroll
  for (int i = 0; i < 8; ++i) woff[i] = ((uint32_t)(n0 + 64 * wid + rin + 8 * i) * (uint32_t)K + gch) * 2u;
  f32x4 acc[MI][8];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = K / BK;
  const uint32_t a_base = (wm * (16 * MI) + fr) * 128, b_base = BM * BK * 2 + (wn * 128 + fr) * 128;
  const uint32_t c0 = ((0 * 4 + fq) ^ sw) << 4, c1 = ((1 * 4 + fq) ^ sw) << 4;
  auto dma_a = [&](int t, int st, int i) {
    __builtin_amdgcn_global_load_lds((gptr_t)((const char*)(A + t * BK) + aoff[i]), (lptr_t)(smem + st * STAGE + (8 * MI * wid) * 128 + i * 1024), 16, 0, 0);
  };
  auto dma_w = [&](int t, int st, int i) {
    __builtin_amdgcn_global_load_lds((gptr_t)((const char*)(W + t * BK) + woff[i]), (lptr_t)(smem + st * STAGE + BM * BK * 2 + (64 * wid) * 128 + i * 1024), 16, 0, 0);
  };
  auto issue = [&](int t, int st) {
#pragma unroll
    for (int i = 0; i < MI; ++i) dma_a(t, st, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_w(t, st, i);
  };
  bf16x8 a0[MI], b0[8], a1[MI], b1[8];
  auto ld = [&](const char* st, bf16x8* a, bf16x8* b, uint32_t coff) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { if (i < MI) a[i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 2048 + coff); b[i] = *reinterpret_cast<const bf16x8*>(st + b_base + i * 2048 + coff); }
  };
  auto mma = [&](const bf16x8* a, const bf16x8* b) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
  };
  if constexpr (VARIANT == 2) {
    // explicit program order, fenced per group of 8 MFMAs (one A fragment x 8 B fragments): group mi of a phase carries 1-2 fragment reads and,
    // in phase 1, two DMA pieces.  Phase 0 reads its A fragments just in time (ring of 3) and fills a1 / b1 (k 32-63) for phase 1; phase 1 (after
    // the barrier: nobody reads stage t any more) fills b0 and the first two A fragments of the next stage and issues the DMA of K-step t + 2.
    auto rdA = [&](const char* st, int mi, uint32_t coff) { return *reinterpret_cast<const bf16x8*>(st + a_base + mi * 2048 + coff); };
    auto rdB = [&](const char* st, int ni, uint32_t coff) { return *reinterpret_cast<const bf16x8*>(st + b_base + ni * 2048 + coff); };
    auto mma8 = [&](int mi, const bf16x8& a, const bf16x8* b) {
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a, acc[mi][ni], 0, 0, 0);
    };
    bf16x8 ar[3];
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if constexpr (MI == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) b0[i] = rdB(smem, i, c0);
    ar[0] = rdA(smem, 0, c0);
    ar[1] = rdA(smem, 1, c0);
    auto phase0 = [&](const char* st) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        __builtin_amdgcn_sched_barrier(0);
        if (mi + 2 < MI) ar[(mi + 2) % 3] = rdA(st, mi + 2, c0);
        a1[mi] = rdA(st, mi, c1);
        b1[mi] = rdB(st, mi, c1);
        if (mi + 1 == MI) { for (int j = MI; j < 8; ++j) b1[j] = rdB(st, j, c1); }
        __builtin_amdgcn_sched_barrier(0);
        mma8(mi, ar[mi % 3], b0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = 0; t < nk - 1; ++t) {                     // every K-step but the last: no branch inside (a branch splits the scheduling region)
      const char* st = smem + (t & 1) * STAGE;
      const char* sn = smem + ((t + 1) & 1) * STAGE;
      const int t2 = t + 2 < nk ? t + 2 : nk - 1;          // past the end: the last K-step once more, into the stage nobody reads any more
      phase0(st);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        __builtin_amdgcn_sched_barrier(0);
        b0[mi] = rdB(sn, mi, c0);
        if (mi + 1 == MI) { for (int j = MI; j < 8; ++j) { b0[j] = rdB(sn, j, c0); dma_w(t2, t & 1, j); } }
        if (mi < 2) ar[mi] = rdA(sn, mi, c0);
        dma_a(t2, t & 1, mi);
        dma_w(t2, t & 1, mi);
        __builtin_amdgcn_sched_barrier(0);
        mma8(mi, a1[mi], b1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    phase0(smem + ((nk - 1) & 1) * STAGE);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) mma8(mi, a1[mi], b1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  if constexpr (MI == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  ld(smem, a0, b0, c0);
  auto phase0 = [&](const char* st) {
    // 64 MFMAs on the k 0-31 fragments, the k 32-63 fragments of this stage read underneath (1 read per 4 MFMAs)
    __builtin_amdgcn_sched_barrier(0);
    ld(st, a1, b1, c1);
    mma(a0, b0);
    if constexpr (VARIANT >= 1) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);     // 4 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int t = 0; t < nk - 1; ++t) {                       // every K-step but the last: straight-line code (a branch would split the scheduling region)
    phase0(smem + (t & 1) * STAGE);
    // every wave has read all of stage t (reads issued in phase 0, waited for here) and its own DMA of step t + 1 has landed
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // phase 1: 64 MFMAs on the k 32-63 fragments; underneath: the next stage's k 0-31 fragments (16 reads) and the DMA of K-step t + 2 (16 pieces;
    // past the end the last K-step is fetched once more into the stage nobody reads any more: no branch)
    const int t2 = t + 2 < nk ? t + 2 : nk - 1;
    if constexpr (VARIANT == 0) { if (t + 2 < nk) issue(t2, t & 1); } else issue(t2, t & 1);
    ld(smem + ((t + 1) & 1) * STAGE, a0, b0, c0);
    mma(a1, b1);
    if constexpr (VARIANT >= 1) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);     // 4 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // 1 DS read
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // 1 VMEM read (LDS-DMA piece)
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  phase0(smem + ((nk - 1) & 1) * STAGE);                    // the last K-step: nothing left to fetch
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  mma(a1, b1);
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + wm * (16 * MI) + mi * 16 + fr;
    if (m < M) {
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        const int n = n0 + wn * 128 + ni * 16 + fq * 4;
        uint32_t lo = (__float_as_uint(acc[mi][ni][0]) >> 16) | (__float_as_uint(acc[mi][ni][1]) & 0xffff0000u);
        uint32_t hi = (__float_as_uint(acc[mi][ni][2]) >> 16) | (__float_as_uint(acc[mi][ni][3]) & 0xffff0000u);
        *reinterpret_cast<uint2*>(C + (int64_t)m * N + n) = make_uint2(lo, hi);
      }
    }
  }
}

#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
template <int V>
static void run(const char* name, uint16_t* dA, uint16_t* dW, uint16_t* dC, int M, int N, int K, int per_block, const std::vector<uint16_t>& hA, const std::vector<uint16_t>& hW) {
  hipFuncSetAttribute((const void*)gemm4w<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  const int grid = (N / BN) * ((M + BM - 1) / BM);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm4w<V>, dim3(grid), dim3(NT), 2 * STAGE, 0, dA, dW, dC, M, N, K);
  hipDeviceSynchronize();
  float iso = 0.f;
  for (int i = 0; i < 5; ++i) {
    hipEventRecord(a); hipLaunchKernelGGL(gemm4w<V>, dim3(grid), dim3(NT), 2 * STAGE, 0, dA, dW, dC, M, N, K); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); iso += ms / 5;
  }
  float sus = 1e9f;
  for (int r = 0; r < 4; ++r) {                      // sustained: blocks of back-to-back launches
    hipEventRecord(a);
    for (int i = 0; i < per_block; ++i) hipLaunchKernelGGL(gemm4w<V>, dim3(grid), dim3(NT), 2 * STAGE, 0, dA, dW, dC, M, N, K);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (r) sus = fminf(sus, ms / per_block);
  }
  std::vector<uint16_t> hC(256 * (size_t)N);
  hipMemcpy(hC.data(), dC + (size_t)100000 * N, hC.size() * 2, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int r = 0; r < 256; r += 37) for (int c = 0; c < N; c += 101) {
    double ref = 0; for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)(100000 + r) * K + k]) * bf2f(hW[(size_t)c * K + k]);
    maxerr = fmax(maxerr, fabs(ref - bf2f(hC[(size_t)r * N + c])) / (fabs(ref) + 1.0));
  }
  const double fl = 2.0 * M * (double)N * K;
  printf("%s %dx%dx%d: isolated %.1f us (%.0f TF), sustained %.1f us (%.0f TF), spot max rel err %.2e, hip: %s\n", name, M, N, K, iso * 1e3, fl / (iso * 1e-3) / 1e12,
         sus * 1e3, fl / (sus * 1e-3) / 1e12, maxerr, hipGetErrorString(hipGetLastError()));
}
int main(int argc, char** argv) {
  const int M = 201000;
  int shapes[2][2] = {{1024, 1024}, {3072, 3072}};
  for (int sh = 0; sh < 2; ++sh) {
    const int N = shapes[sh][0], K = shapes[sh][1];
    std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
    for (auto& v : hA) v = f2bf(rnd() * 0.8f);
    for (auto& v : hW) v = f2bf(rnd() * 0.05f);
    uint16_t *dA, *dW, *dC;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    const int pb = sh == 0 ? 60 : 10;
    run<0>("gemm4w burst (round 4)   ", dA, dW, dC, M, N, K, pb, hA, hW);
    run<1>("gemm4w software-pipelined", dA, dW, dC, M, N, K, pb, hA, hW);
    run<2>("gemm4w explicit order     ", dA, dW, dC, M, N, K, pb, hA, hW);
    hipFree(dA); hipFree(dW); hipFree(dC);
  }
  return 0;
}
