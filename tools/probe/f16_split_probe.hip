// Does the fp16 MFMA keep SUBNORMAL fp16 inputs, and what does an fp16 hi+lo split dot product reach?  (diagnostic, not product)
// hipcc --offload-arch=gfx950 -O3 -o f16_split_probe f16_split_probe.hip ; ./f16_split_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave: D[16x16] = A[16x32] * B[32x16]; lane l holds A[row l&15][k 8*(l>>4) .. +7] and B[k 8*(l>>4)..][col l&15]
__global__ void probe(const float* a, const float* b, float* out_plain, float* out_split, float* out_scaled) {
  const int l = threadIdx.x, r = l & 15, kq = l >> 4;
  f16x8 ah, al, als, bh, bl, bls;
  for (int j = 0; j < 8; ++j) {
    const float x = a[r * 32 + kq * 8 + j], y = b[(kq * 8 + j) * 16 + r];
    ah[j] = (_Float16)x; al[j] = (_Float16)(x - (float)ah[j]); als[j] = (_Float16)((x - (float)ah[j]) * 2048.0f);
    bh[j] = (_Float16)y; bl[j] = (_Float16)(y - (float)bh[j]); bls[j] = (_Float16)((y - (float)bh[j]) * 2048.0f);
  }
  f32x4 c = {0, 0, 0, 0}, d = {0, 0, 0, 0}, e = {0, 0, 0, 0}, e2 = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, d, 0, 0, 0);
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, d, 0, 0, 0);
  e = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, e, 0, 0, 0);
  e2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(als, bh, e2, 0, 0, 0);
  e2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bls, e2, 0, 0, 0);
  for (int i = 0; i < 4; ++i) {     // D layout: lane l holds rows 4*(l>>4)+i, col l&15
    const int row = 4 * kq + i, col = r;
    out_plain[row * 16 + col] = c[i];
    out_split[row * 16 + col] = d[i];
    out_scaled[row * 16 + col] = e[i] + e2[i] * (1.0f / 2048.0f);
  }
}

int main() {
  std::vector<float> a(16 * 32), b(32 * 16);
  float *da, *db, *o1, *o2, *o3;
  hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&o1, 1024); hipMalloc(&o2, 1024); hipMalloc(&o3, 1024);
  for (int test = 0; test < 3; ++test) {
    srand(1 + test);
    // test 0: O(1) values; test 1: A small (|x| ~ 1e-2: lo is an fp16 subnormal); test 2: A in the fp16-subnormal range itself (1e-6)
    const float amag = test == 0 ? 1.0f : test == 1 ? 1e-2f : 1e-6f;
    for (auto& v : a) v = amag * ((rand() / (float)RAND_MAX) * 2 - 1);
    for (auto& v : b) v = 0.05f * ((rand() / (float)RAND_MAX) * 2 - 1);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, o1, o2, o3);
    std::vector<float> r1(256), r2(256), r3(256);
    hipMemcpy(r1.data(), o1, 1024, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), o2, 1024, hipMemcpyDeviceToHost); hipMemcpy(r3.data(), o3, 1024, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0, e3 = 0, ref_norm = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double s = 0, sa = 0; for (int k = 0; k < 32; ++k) { s += (double)a[i * 32 + k] * b[k * 16 + j]; sa += fabs((double)a[i * 32 + k] * b[k * 16 + j]); }
      e1 = fmax(e1, fabs(r1[i * 16 + j] - s) / sa); e2 = fmax(e2, fabs(r2[i * 16 + j] - s) / sa); e3 = fmax(e3, fabs(r3[i * 16 + j] - s) / sa); ref_norm = fmax(ref_norm, sa);
    }
    printf("|a| ~ %g: max error / sum|a b|:  fp16 alone %.3e   hi+lo (3 MFMA, unscaled lo) %.3e   hi+lo with lo*2048 in a second accumulator %.3e\n", amag, e1, e2, e3);
  }
  // direct subnormal check: A = 2^-20 (an fp16 subnormal) in one element, B = 1024
  return 0;
}
