// Bare MFMA issue-rate probe on random operands (diagnostic; not part of the product path).
// hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip ; ./mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: one dependent chain 32x32x16; 1: two independent chains; 2: 16x16x32 four independent chains;
// 3: one chain + 3 v_max per MFMA on the OTHER accumulator; 4: chain + one ds_read_b128 per MFMA
template <int MODE>
__global__ __launch_bounds__(512) void probe(const bf16x8* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* clk) {
  __shared__ bf16x8 lds[2048];
  const int tid = threadIdx.x;
  bf16x8 a[12], b[12];
  for (int i = 0; i < 12; ++i) { a[i] = in[(tid * 12 + i) & 4095]; b[i] = in[(tid * 7 + i * 5) & 4095]; }
  for (int i = tid; i < 2048; i += 512) lds[i] = in[i];
  __syncthreads();
  f32x16 c0 = {0}, c1 = {0};
  f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
  float m = -1e30f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      if constexpr (MODE == 0) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], c0, 0, 0, 0);
      if constexpr (MODE == 1) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[(k + 1) % 12], c1, 0, 0, 0);
      }
      if constexpr (MODE == 2) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k], b[k], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k], b[(k + 1) % 12], d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(k + 1) % 12], b[k], d2, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(k + 2) % 12], b[k], d3, 0, 0, 0);
      }
      if constexpr (MODE == 3) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], c0, 0, 0, 0);
        m = fmaxf(m, c1[k]); m = fmaxf(m, c1[(k + 1) & 15]); m = fmaxf(m, c1[(k + 2) & 15]);
      }
      if constexpr (MODE == 4) {
        const bf16x8 f = lds[(tid * 3 + k * 64 + it) & 2047];
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, b[k], c0, 0, 0, 0);
      }
    }
    if constexpr (MODE == 3) { f32x16 t = c0; c0 = c1; c1 = t; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = m;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  for (int i = 0; i < 4; ++i) s += d0[i] + d1[i] + d2[i] + d3[i];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, const bf16x8* in, float* out, unsigned long long* clk, int iters, double flop_per_iter_wave) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<256, 512>>>(in, out, iters / 10, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<256, 512>>>(in, out, iters, clk);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(512); hipMemcpy(h.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
  const double mhz = (double)h[0] / (double)h[1] * 100.0;
  const double tf = flop_per_iter_wave * iters * 8 * 256 / (ms * 1e-3) / 1e12;
  const double mfma_per_wave = flop_per_iter_wave / (MODE == 2 ? 16384.0 : 32768.0) * iters;
  printf("%-44s %.3f ms  %.0f TF  clock %.0f MHz  %.1f cycles per MFMA per SIMD (2 waves)\n", name, ms, tf, mhz, (double)h[0] / (mfma_per_wave * 2));
}

int main() {
  std::vector<unsigned short> h(4096 * 8);
  srand(1);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
  bf16x8* in; float* out; unsigned long long* clk;
  hipMalloc(&in, h.size() * 2); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 512 * 8);
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("32x32x16 one dependent chain", in, out, clk, iters, 12 * 32768.0);
    run<1>("32x32x16 two independent chains", in, out, clk, iters, 24 * 32768.0);
    run<2>("16x16x32 four independent chains", in, out, clk, iters, 48 * 16384.0);
    run<3>("32x32x16 chain + 3 v_max per MFMA", in, out, clk, iters, 12 * 32768.0);
    run<4>("32x32x16 chain + ds_read_b128 per MFMA", in, out, clk, iters, 12 * 32768.0);
  }
  return 0;
}
