// Round 5 probe: does the ORDER in which a K-step's 64 MFMAs are issued change what the chip delivers at its power cap?  Bare v_mfma_f32_16x16x32_bf16
// loops on random operands held in registers (no LDS, no memory), 2 waves per SIMD on every CU, the register picture of conv_gemm256's K-step: A
// fragments [mh 2][ks 2][mi 4], W fragments [ks 2][ni 4], 32 accumulator tiles [mh*4+mi][ni].  Every variant issues the SAME 64 products per iteration
// (bit-identical sums); only the order differs, fixed by asm volatile.  At the cap (profiles/r05_power_probe.json) FLOP/s IS energy per MFMA.
//   0  as the kernel: for ks, mh, mi, ni      (A operand held for 4 MFMAs, W changes every MFMA, an accumulator returns after 32 MFMAs)
//   1  ks innermost:  for mh, mi, ni, ks      (each accumulator takes its two products back to back: chain of 2)
//   2  as 0 with ni snaking (3,2,1,0 on odd mi): one operand change fewer per row
//   3  W held:        for ks, ni, mh, mi      (W operand held for 8 MFMAs, A changes every MFMA)
//   4  W held + ks innermost: for ni, mh, mi, ks
//   5  floor: one operand pair for all 64 MFMAs (no operand toggling), accumulators as 0
//   6  one accumulator for all 64 MFMAs (chain of 64), operands as 0
//   7  pairs interleaved two by two: (X k0, Y k0, X k1, Y k1) - an accumulator's second product follows ONE other MFMA
//   8  pairs interleaved four by four: (X, Y, Z, U) k0 then (X, Y, Z, U) k1 (one row block, all column blocks)
//   9  as 4 with the k-halves snaking from pair to pair (X k0 k1, Y k1 k0): the W operand is held across every pair boundary
// second argument 1: ONE wave per SIMD (256 threads per workgroup) - what a wave delivers while its partner is busy elsewhere
// hipcc --offload-arch=gfx950 -O3 -o mfma_energy mfma_energy.hip ; ./mfma_energy [seconds per variant = 1.0]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA(ACC, WF, AF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(WF), "v"(AF))

template <int V>
__global__ __launch_bounds__(512, 2) void probe(const bf16x8* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* clk) {
  const int tid = threadIdx.x;
  bf16x8 A[2][2][4], W[2][4];
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) A[mh][ks][mi] = in[(tid * 16 + mh * 8 + ks * 4 + mi) & 8191];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) W[ks][ni] = in[(tid * 8 + 4096 + ks * 4 + ni + blockIdx.x) & 8191];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (V == 0 || V == 2 || V == 5 || V == 6) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int nj = 0; nj < 4; ++nj) {
              const int ni = (V == 2 && (mi & 1)) ? 3 - nj : nj;
              if constexpr (V == 5) MFMA(acc[mh * 4 + mi][ni], W[0][0], A[0][0][0]);
              else if constexpr (V == 6) MFMA(acc[0][0], W[ks][ni], A[mh][ks][mi]);
              else MFMA(acc[mh * 4 + mi][ni], W[ks][ni], A[mh][ks][mi]);
            }
    }
    if constexpr (V == 1) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) MFMA(acc[mh * 4 + mi][ni], W[ks][ni], A[mh][ks][mi]);
    }
    if constexpr (V == 3) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) MFMA(acc[mh * 4 + mi][ni], W[ks][ni], A[mh][ks][mi]);
    }
    if constexpr (V == 7) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int np = 0; np < 2; ++np)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
              for (int q = 0; q < 2; ++q) MFMA(acc[mh * 4 + mi][np * 2 + q], W[ks][np * 2 + q], A[mh][ks][mi]);
    }
    if constexpr (V == 8) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) MFMA(acc[mh * 4 + mi][ni], W[ks][ni], A[mh][ks][mi]);
    }
    if constexpr (V == 9) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
              const int ks = (mi & 1) ? 1 - kk : kk;
              MFMA(acc[mh * 4 + mi][ni], W[ks][ni], A[mh][ks][mi]);
            }
    }
    if constexpr (V == 4) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) MFMA(acc[mh * 4 + mi][ni], W[ks][ni], A[mh][ks][mi]);
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

static double checksum(const float* d_out) {
  std::vector<float> h(256 * 256);
  hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
  double s = 0;
  for (float v : h) s += (double)v;
  return s;
}

template <int V>
static void run(const char* name, const bf16x8* in, float* out, unsigned long long* clk, double seconds, int nt) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int cal = 20000;
  probe<V><<<256, nt>>>(in, out, cal, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0); probe<V><<<256, nt>>>(in, out, cal, clk); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int iters = (int)(cal * (seconds * 1e3 / ms));
  probe<V><<<256, nt>>>(in, out, iters, clk);                 // bring the chip to this variant's steady state
  hipEventRecord(e0); probe<V><<<256, nt>>>(in, out, iters, clk); hipEventRecord(e1); hipDeviceSynchronize();
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(512); hipMemcpy(h.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
  const double mhz = (double)h[0] / (double)h[1] * 100.0;
  const double mfmas = 64.0 * iters;                            // per wave
  const double tf = mfmas * 16384.0 * (nt / 64) * 256 / (ms * 1e-3) / 1e12;
  probe<V><<<256, nt>>>(in, out, 8, clk);                      // a short run for the checksum (same products in every variant but 5 / 6)
  hipDeviceSynchronize();
  printf("%-58s %8.1f ms  %6.0f TF  clock %5.0f MHz  %5.2f wave-0 cycles per SIMD-MFMA  checksum(8 iters) %.6e  %s\n", name, ms, tf, mhz, (double)h[0] / (mfmas * (nt / 256)), checksum(out),
         hipGetErrorString(hipGetLastError()));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 1.0;
  const int nt = (argc > 2 && atoi(argv[2]) == 1) ? 256 : 512;
  printf("%d waves per SIMD\n", nt / 256);
  std::vector<unsigned short> h(8192 * 8);
  srand(1);
  for (auto& v : h) { float f = ((rand() / (float)RAND_MAX) + (rand() / (float)RAND_MAX) + (rand() / (float)RAND_MAX) - 1.5f) * 0.8f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
  bf16x8* in; float* out; unsigned long long* clk;
  hipMalloc(&in, h.size() * 2); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 512 * 8);
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("0 as the kernel (ks, mh, mi, ni)", in, out, clk, seconds, nt);
    run<1>("1 ks innermost (accumulator chains of 2)", in, out, clk, seconds, nt);
    run<2>("2 as 0, ni snaking", in, out, clk, seconds, nt);
    run<3>("3 W held for 8 MFMAs (ks, ni, mh, mi)", in, out, clk, seconds, nt);
    run<4>("4 W held + ks innermost (ni, mh, mi, ks)", in, out, clk, seconds, nt);
    run<5>("5 floor: one operand pair for every MFMA", in, out, clk, seconds, nt);
    run<6>("6 one accumulator for every MFMA (chain of 64)", in, out, clk, seconds, nt);
    run<7>("7 pairs interleaved two by two (X0 Y0 X1 Y1)", in, out, clk, seconds, nt);
    run<8>("8 pairs interleaved four by four (mh, mi, ks, ni)", in, out, clk, seconds, nt);
    run<9>("9 as 4, k-halves snaking (W held across pair boundaries)", in, out, clk, seconds, nt);
  }
  return 0;
}
