// Exact fp32 scoring helpers shared by the general top-k path (scoring.hip) and the k = 1 row/column-maxima
// fast path (affinity_rowcol.hip).  ONE dot-product routine everywhere: a profile's reported score is
// bit-identical whichever kernel computed it (tests: k = 3 top-1 == k = 1 result).
#pragma once
#include "common.hpp"

namespace sdk_exact {

constexpr int D = 192;                 // embedding width

// ---- exact fp32 dot product of two 192-vectors by a group of 8 consecutive lanes ---------------
// lane j of the group owns the elements {32 q + 4 j + 0..3 : q = 0..5}: for every q the group's 8 lanes read ONE 128-byte line
// of the row (a wave-instruction then touches 8 lines; with "lane j owns [24 j, 24 j + 24)", as in round 1, it touched 48).
// Fixed order: sequential fma inside the lane (q ascending), then the xor-butterfly 1,2,4 (fp add is commutative, so all
// 8 lanes hold the same bits).  EVERY kernel loads its rows through load_row24 / load_prow, so the order is one.
__device__ __forceinline__ void load_prow(const float* __restrict__ prow, int j, f32x4* pv) {
#pragma unroll
  for (int q = 0; q < 6; ++q) pv[q] = *reinterpret_cast<const f32x4*>(prow + 32 * q + 4 * j);
}
__device__ __forceinline__ void load_row24(const float* __restrict__ row, int j, float* e24) {
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 32 * q + 4 * j);
    e24[4 * q] = v[0]; e24[4 * q + 1] = v[1]; e24[4 * q + 2] = v[2]; e24[4 * q + 3] = v[3];
  }
}
__device__ __forceinline__ float dot192_regs(const float* __restrict__ e24, const f32x4* __restrict__ pv) {
  float a = 0.f;
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const f32x4 v = pv[q];
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
__device__ __forceinline__ float dot192_group8(const float* __restrict__ e24, const float* __restrict__ prow, int j) {
  f32x4 pv[6];
  load_prow(prow, j, pv);
  return dot192_regs(e24, pv);
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

// wave-wide selection of the 4 best (score desc, index asc) among 2 entries per lane; result on every lane
__device__ __forceinline__ void wave_select4(float s0, int i0, float s1, int i1, float* os, int* oi) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float bs = s0; int bi = i0;
    if (better(s1, i1, bs, bi)) { bs = s1; bi = i1; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ts = __shfl_xor(bs, o, 64);
      const int ti = __shfl_xor(bi, o, 64);
      if (better(ts, ti, bs, bi)) { bs = ts; bi = ti; }
    }
    os[q] = bs; oi[q] = bi;
    if (s0 == bs && i0 == bi) { s0 = -INFINITY; i0 = 0x7fffffff; }      // the winner leaves the pool
    if (s1 == bs && i1 == bi) { s1 = -INFINITY; i1 = 0x7fffffff; }
  }
}

}  // namespace sdk_exact
