// Feasibility probe (round 3): a 4-wave / 512-register 256x256x64 bf16 GEMM K loop with register-staged operands - does hipcc keep the 256
// accumulator registers in AGPRs without copies in the loop?   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int BM = 256, BN = 256, BK = 64, NT = 256;
constexpr int STAGE = (BM + BN) * BK * 2;   // 64 KiB

__global__ __launch_bounds__(NT, 1) void gemm4w(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, uint16_t* __restrict__ C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fq = lane >> 4;
  const int nbn = N / BN;
  const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  // staging: thread -> chunk (tid & 7) of rows (tid >> 3) + 32 i, i < 8
  const int ch = tid & 7, r0 = tid >> 3;
  const uint16_t* ap[8];
  const uint16_t* wp[8];
  uint32_t lw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = r0 + 32 * i;
    int m = m0 + row; m = m < M ? m : M - 1;
    ap[i] = A + (int64_t)m * K + ch * 8;
    wp[i] = W + (int64_t)(n0 + row) * K + ch * 8;
    lw[i] = row * 128 + ((ch ^ (row & 7)) << 4);
  }
  f32x4 acc[8][8];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 ra[8], rb[8];
  auto gload = [&](int s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ra[i] = *reinterpret_cast<const u32x4*>(ap[i] + s * BK);
      rb[i] = *reinterpret_cast<const u32x4*>(wp[i] + s * BK);
    }
  };
  auto lstore = [&](int st) {
    char* sA = smem + st * STAGE;
    char* sB = sA + BM * BK * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      *reinterpret_cast<u32x4*>(sA + lw[i]) = ra[i];
      *reinterpret_cast<u32x4*>(sB + lw[i]) = rb[i];
    }
  };
  const int nk = K / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int sw = lane & 7;
  for (int s = 0; s < nk; ++s) {
    if (s + 1 < nk) gload(s + 1);
    const char* sA = smem + (s & 1) * STAGE;
    const char* sB = sA + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint32_t coff = ((ks * 4 + fq) ^ sw) << 4;
      bf16x8 af[8], bfr[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(sA + (wm * 128 + i * 16 + fr) * 128 + coff);
        bfr[i] = *reinterpret_cast<const bf16x8*>(sB + (wn * 128 + i * 16 + fr) * 128 + coff);
      }
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mi][ni]) : "v"(bfr[ni]), "v"(af[mi]));   // accumulators pinned to AGPRs
    }
    if (s + 1 < nk) lstore((s + 1) & 1);
    __syncthreads();
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // MFMA results -> v_accvgpr_read: the compiler does not know the asm is an MFMA
  // plain epilogue: lane holds 4 consecutive columns of one row (weights are the MFMA row operand)
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
