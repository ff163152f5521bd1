// Round 4 probe: a 4-wave / 512-register 256x256x64 bf16 GEMM with LDS-DMA staging (accumulators in AGPRs, one wave per SIMD), bare (one
// workgroup per tile, no XCD-aware order, 8-byte epilogue stores): is the structure worth a full kernel?  hipcc --offload-arch=gfx950 -O3 -o gemm4w_bench gemm4w_bench.hip
#include <string.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int BM = 256, BN = 256, BK = 64, NT = 256;
constexpr int STAGE = (BM + BN) * BK * 2;
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;
__global__ __launch_bounds__(NT, 1) void gemm4w(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, uint16_t* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fq = lane >> 4, sw = lane & 7;
  const int nbn = N / BN;
  const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int rin = lane >> 3, pos = lane & 7, gch = (pos ^ rin) * 8;
  uint32_t aoff[8], woff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    aoff[i] = ((uint32_t)min(m0 + 64 * wid + rin + 8 * i, M - 1) * (uint32_t)K + gch) * 2u;
    woff[i] = ((uint32_t)(n0 + 64 * wid + rin + 8 * i) * (uint32_t)K + gch) * 2u;
  }
  auto issue = [&](int t, int st) {
    char* sA = smem + st * STAGE + (64 * wid) * 128;
    char* sB = sA + BM * BK * 2;
    const char* ab = (const char*)(A + t * BK);
    const char* wb = (const char*)(W + t * BK);
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(ab + aoff[i]), (lptr_t)(sA + i * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(wb + woff[i]), (lptr_t)(sB + i * 1024), 16, 0, 0);
  };
  f32x4 acc[8][8];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = K / BK;
  const uint32_t a_base = (wm * 128 + fr) * 128, b_base = BM * BK * 2 + (wn * 128 + fr) * 128;
  const uint32_t c0 = ((0 * 4 + fq) ^ sw) << 4, c1 = ((1 * 4 + fq) ^ sw) << 4;
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8 a0[8], b0[8], a1[8], b1[8];
  auto ld = [&](const char* st, bf16x8* a, bf16x8* b, uint32_t coff) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 2048 + coff); b[i] = *reinterpret_cast<const bf16x8*>(st + b_base + i * 2048 + coff); }
  };
  ld(smem, a0, b0, c0);
  for (int t = 0; t < nk; ++t) {
    const char* st = smem + (t & 1) * STAGE;
    __builtin_amdgcn_sched_barrier(0);
    ld(st, a1, b1, c1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[ni], a0[mi], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (t + 2 < nk) issue(t + 2, t & 1);
      ld(smem + ((t + 1) & 1) * STAGE, a0, b0, c0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
  }
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int m = m0 + wm * 128 + mi * 16 + fr;
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
    hipFuncSetAttribute((const void*)gemm4w, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    const int grid = (N / BN) * ((M + BM - 1) / BM);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm4w, dim3(grid), dim3(NT), 2 * STAGE, 0, dA, dW, dC, M, N, K);
    hipDeviceSynchronize();
    float best = 1e9f, tot = 0.f;
    for (int i = 0; i < 10; ++i) {
      hipEventRecord(a); hipLaunchKernelGGL(gemm4w, dim3(grid), dim3(NT), 2 * STAGE, 0, dA, dW, dC, M, N, K); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best; tot += ms;
    }
    std::vector<uint16_t> hC(256 * (size_t)N);
    hipMemcpy(hC.data(), dC + (size_t)100000 * N, hC.size() * 2, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int r = 0; r < 256; r += 37) for (int c = 0; c < N; c += 101) {
      double ref = 0; for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)(100000 + r) * K + k]) * bf2f(hW[(size_t)c * K + k]);
      maxerr = fmax(maxerr, fabs(ref - bf2f(hC[(size_t)r * N + c])) / (fabs(ref) + 1.0));
    }
    const double fl = 2.0 * M * (double)N * K;
    printf("gemm4w %dx%dx%d: mean %.1f us (%.0f TF), best %.1f us (%.0f TF), spot max rel err %.2e, hip error: %s\n", M, N, K, tot / 10 * 1e3, fl / (tot / 10 * 1e-3) / 1e12,
           best * 1e3, fl / (best * 1e-3) / 1e12, maxerr, hipGetErrorString(hipGetLastError()));
    hipFree(dA); hipFree(dW); hipFree(dC);
  }
  return 0;
}
