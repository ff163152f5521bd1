// Per-utterance (HBM-streaming) pieces of the ECAPA-TDNN forward: squeeze-excitation gate +
// residual, attentive-statistics pooling, small fp32 fully-connected layers, L2-normalise.
// All of them are bandwidth-bound row sweeps: 16-byte loads per lane, channels on lanes so
// that every wave-instruction touches whole 128-B lines, fp32 accumulation.
#include "common.hpp"

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------------------------------
// SE: one workgroup per segment.  Thread (grp, c8) owns 8 channels and every ngrp-th frame.
// (F16 on this and the other sweeps: the 2-byte activations are fp16 instead of bf16 - the single-plane fp16 contract; fp32 arithmetic either way)
template <bool F16>
__global__ __launch_bounds__(NT) void se_gate_residual_kernel(
    const bf16_t* __restrict__ z, int64_t ldz, const bf16_t* __restrict__ x, int64_t ldx,
    const float* __restrict__ w1t, const float* __restrict__ b1, const float* __restrict__ w2t,
    const float* __restrict__ b2, bf16_t* __restrict__ out, int64_t ldo, int T, int C, int Cse) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nch8 = C >> 3;
  const int ngrp = NT / nch8;
  float* part = reinterpret_cast<float*>(smem);   // [ngrp][C]  (reused as FC1 partials)
  float* mean = part + ngrp * C;                  // [C]         (reused as gate g[C])
  float* hbuf = mean + C;                         // [Cse]

  const int tid = threadIdx.x;
  const int c8 = tid % nch8, grp = tid / nch8;
  const int64_t base = (int64_t)blockIdx.x * T;

  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  {
    // 4 independent 16-byte loads in flight per lane (the sweep is latency-bound otherwise)
    const bf16_t* zp = z + base * ldz + c8 * 8;
    int t = grp;
    for (; t + 3 * ngrp < T; t += 4 * ngrp) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4*>(zp + (int64_t)(t + u * ngrp) * ldz);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[8];
        unpack8t<F16>(v[u], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += f[e];
      }
    }
    for (; t < T; t += ngrp) {
      float f[8];
      unpack8t<F16>(*reinterpret_cast<const u32x4*>(zp + (int64_t)t * ldz), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += f[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[grp * C + c8 * 8 + e] = s[e];
  __syncthreads();
  const float invT = 1.0f / (float)T;
  for (int c = tid; c < C; c += NT) {
    float m = 0.f;
    for (int g = 0; g < ngrp; ++g) m += part[g * C + c];
    mean[c] = m * invT;
  }
  __syncthreads();
#ifndef SE_ABLATE_FC
  // FC1 (C -> Cse) : thread = (output j, slice k of the input range)
  {
    const int nsl = NT / Cse;              // Cse divides NT
    const int j = tid % Cse, k = tid / Cse;
    const int c_lo = (int)((int64_t)C * k / nsl), c_hi = (int)((int64_t)C * (k + 1) / nsl);
    float a = 0.f;
    for (int c = c_lo; c < c_hi; ++c) a += mean[c] * w1t[(int64_t)c * Cse + j];
    part[k * Cse + j] = a;                 // part is free again (mean already reduced)
    __syncthreads();
    if (tid < Cse) {
      float v = b1[tid];
      for (int kk = 0; kk < nsl; ++kk) v += part[kk * Cse + tid];
      hbuf[tid] = fmaxf(v, 0.f);
    }
    __syncthreads();
  }
  // FC2 (Cse -> C) + sigmoid; gate stored over `mean`
  for (int c = tid; c < C; c += NT) {
    float a = b2[c];
    for (int j = 0; j < Cse; ++j) a += hbuf[j] * w2t[(int64_t)j * C + c];
    mean[c] = 1.0f / (1.0f + __expf(-a));
  }
  __syncthreads();
#endif
  float g[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) g[e] = mean[c8 * 8 + e];
  {
    const bf16_t* zp = z + base * ldz + c8 * 8;
    const bf16_t* xp = x + base * ldx + c8 * 8;
    bf16_t* op = out + base * ldo + c8 * 8;
    auto one = [&](const u32x4& zv, const u32x4& xv, int t) {
      float fz[8], fx[8];
      unpack8t<F16>(zv, fz);
      unpack8t<F16>(xv, fx);
#pragma unroll
      for (int e = 0; e < 8; ++e) fz[e] = g[e] * fz[e] + fx[e];
      *reinterpret_cast<u32x4*>(op + (int64_t)t * ldo) = pack8t<F16>(fz);
    };
    int t = grp;
    for (; t + 3 * ngrp < T; t += 4 * ngrp) {
      u32x4 zv[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        zv[u] = *reinterpret_cast<const u32x4*>(zp + (int64_t)(t + u * ngrp) * ldz);
        xv[u] = *reinterpret_cast<const u32x4*>(xp + (int64_t)(t + u * ngrp) * ldx);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) one(zv[u], xv[u], t + u * ngrp);
    }
    for (; t < T; t += ngrp)
      one(*reinterpret_cast<const u32x4*>(zp + (int64_t)t * ldz), *reinterpret_cast<const u32x4*>(xp + (int64_t)t * ldx), t);
  }
}

// ------------------------------------------------------------------------------------------
// ASP global context: mean / std over frames.  grid (B, C/1024) ; 128 chunk-columns x 2 frame groups.
template <bool F16>
__global__ __launch_bounds__(NT) void asp_stats_kernel(const bf16_t* __restrict__ h, int64_t ldh, int T, int C,
                                                      float* __restrict__ out) {
  __shared__ float red[2][2][1024];
  const int tid = threadIdx.x;
  const int c8 = tid & 127, grp = tid >> 7;
  const int cbase = blockIdx.y * 1024 + c8 * 8;
  const int64_t base = (int64_t)blockIdx.x * T;
  const bool live = cbase < C;
  float K[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { K[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; }
  if (live) {
    unpack8t<F16>(*reinterpret_cast<const u32x4*>(h + base * ldh + cbase), K);   // shift = frame 0
    for (int t = grp; t < T; t += 2) {
      float f[8];
      unpack8t<F16>(*reinterpret_cast<const u32x4*>(h + (base + t) * ldh + cbase), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = f[e] - K[e];
        s1[e] += d;
        s2[e] += d * d;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[grp][0][c8 * 8 + e] = s1[e];
    red[grp][1][c8 * 8 + e] = s2[e];
  }
  __syncthreads();
  if (grp == 0 && live) {
    const float invT = 1.0f / (float)T;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = (red[0][0][c8 * 8 + e] + red[1][0][c8 * 8 + e]) * invT;
      const float q = (red[0][1][c8 * 8 + e] + red[1][1][c8 * 8 + e]) * invT;
      const float var = fmaxf(q - a * a, 1e-12f);
      out[(int64_t)blockIdx.x * 2 * C + cbase + e] = K[e] + a;
      out[(int64_t)blockIdx.x * 2 * C + C + cbase + e] = sqrtf(var);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Small fp32 fully-connected layer over rows (per-utterance GEMV batch: SE-style heads, global-context
// bias, final 6144->192 projection).  Block = 4 rows x 32 outputs x 8 K-slices: lane j of a 32-lane
// half-wave owns output j (coalesced 128-B weight rows), the 8 half-waves split every 512-wide input
// chunk, partial sums meet in LDS and are added in slice order (bitwise reproducible, no atomics).
constexpr int FC_ROWS = 4, FC_CHUNK = 512, FC_OUT = 32, FC_SLICES = NT / FC_OUT;   // 8 slices of 64 inputs
__global__ __launch_bounds__(NT) void rows_fc_kernel(const float* __restrict__ in, int64_t ldin,
                                                    const float* __restrict__ isc, const float* __restrict__ ish,
                                                    const float* __restrict__ wt, const float* __restrict__ bias,
                                                    float* __restrict__ out, int64_t ldout, int B, int Cin, int Nout,
                                                    int act) {
  __shared__ float xs[FC_ROWS][FC_CHUNK];
  __shared__ float part[FC_SLICES][FC_ROWS][FC_OUT];
  const int tid = threadIdx.x;
  const int jl = tid & (FC_OUT - 1), ks = tid / FC_OUT;
  const int j = blockIdx.y * FC_OUT + jl;
  const int jc = j < Nout ? j : Nout - 1;
  const int b0 = blockIdx.x * FC_ROWS;
  constexpr int SL = FC_CHUNK / FC_SLICES;
  float acc[FC_ROWS];
#pragma unroll
  for (int r = 0; r < FC_ROWS; ++r) acc[r] = 0.f;
  for (int c0 = 0; c0 < Cin; c0 += FC_CHUNK) {
    const int n = min(FC_CHUNK, Cin - c0);
    for (int i = tid; i < FC_ROWS * FC_CHUNK; i += NT) {
      const int r = i / FC_CHUNK, c = i - r * FC_CHUNK;
      float v = 0.f;
      if (c < n && b0 + r < B) {
        v = in[(int64_t)(b0 + r) * ldin + c0 + c];
        if (isc) v = v * isc[c0 + c] + ish[c0 + c];
      }
      xs[r][c] = v;
    }
    __syncthreads();
    const int lo = ks * SL, hi = min(n, lo + SL);
    const float* wp = wt + (int64_t)(c0 + lo) * Nout + jc;
#pragma unroll 8
    for (int c = lo; c < hi; ++c) {
      const float w = *wp;
      wp += Nout;
#pragma unroll
      for (int r = 0; r < FC_ROWS; ++r) acc[r] = fmaf(xs[r][c], w, acc[r]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < FC_ROWS; ++r) part[ks][r][jl] = acc[r];
  __syncthreads();
  if (tid < FC_ROWS * FC_OUT) {
    const int r = tid / FC_OUT, jj = tid & (FC_OUT - 1);
    const int jo = blockIdx.y * FC_OUT + jj;
    if (b0 + r < B && jo < Nout) {
      float v = bias ? bias[jo] : 0.f;
#pragma unroll
      for (int s2 = 0; s2 < FC_SLICES; ++s2) v += part[s2][r][jj];
      if (act == 1) v = fmaxf(v, 0.f);
      else if (act == 2) v = 1.0f / (1.0f + __expf(-v));
      out[(int64_t)(b0 + r) * ldout + jo] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// The same layer on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32 = a k-ordered fmaf chain, so the
// result is still plain fp32 arithmetic in a fixed order).  Block = 32 rows x 32 outputs, the 4 waves
// split K; lane half h of a wave takes k = 8g + 4h + u in MFMA step u, so every lane fetches its four
// A values with ONE 16-byte load and the B values as four coalesced 128-byte weight-row segments.
template <int KW, bool WLDS = false>   // waves per block = K slices; WLDS: weight groups fetched as 16-byte rows through a wave-private LDS tile
__global__ __launch_bounds__(KW * 64) void rows_fc_mfma_kernel(const float* __restrict__ in, int64_t ldin,
                                                         const float* __restrict__ isc, const float* __restrict__ ish,
                                                         const float* __restrict__ wt, const float* __restrict__ bias,
                                                         float* __restrict__ out, int64_t ldout, int B, int Cin, int Nout,
                                                         int act) {
  __shared__ float red[KW][32 * 33];
  __shared__ __attribute__((aligned(16))) float wtile[WLDS ? KW : 1][8 * 32];   // one group of weights [8 k][32 outputs] per wave
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int i = lane & 31, hh = lane >> 5;
  const int b0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int row = min(b0 + i, B - 1), colw = min(n0 + i, Nout - 1);
  const int kw = Cin / KW;                                 // K range of this wave (multiple of 8)
  const float* ap = in + (int64_t)row * ldin + wid * kw + 4 * hh;
  const float* bp = wt + (int64_t)(wid * kw + 4 * hh) * Nout + colw;
  // the input affine (the final FC's folded ASP batch-norm), when given: staged once per workgroup in dynamic LDS [2][Cin] - it depends on k only,
  // and carrying it in the fetch ring would cost 8 registers per slot
  extern __shared__ __attribute__((aligned(16))) float aff[];
  if (isc) {
    for (int e = tid; e < Cin; e += KW * 64) { aff[e] = isc[e]; aff[Cin + e] = ish[e]; }
    __syncthreads();
  }
  const float* sp = isc ? aff + wid * kw + 4 * hh : nullptr;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // software pipeline: a ring of D register slots, the operands of groups g+1 .. g+D-1 (8 k each) in flight while group g feeds the matrix pipe
  // (rounds 1-3 ran one group ahead).  The input affine is applied when a slot is consumed (from LDS), so a fetch is loads only.  What bounds the
  // 6144-long FCs is the fp32 matrix pipe, not latency: 192 `32x32x2` MFMAs x 64 cycles x 4 waves per SIMD = 23 us on the 128-192 workgroups
  // (32 x 32 output tiles of a 1000-row problem) of a 45-55 us launch; the deeper ring buys 5-15 % (`tools/rows_fc_bench.py`: 55 -> 50, 14 -> 12 us).
  // A timing probe with perfectly coalesced (wrong) A addresses - each lane's 16-byte piece lies in a different row - reads 45 -> 39 / 50 -> 45 us:
  // staging A through LDS would buy at most that; not built.
  constexpr int D = WLDS ? 8 : 4;
  const int ng = kw >> 3;
  f32x4 ra[D], rw[D];
  const float* wq = WLDS ? wt + (int64_t)(wid * kw + (lane >> 3)) * Nout + n0 + (lane & 7) * 4 : nullptr;
  auto fetch = [&](int g, int d) {
    ra[d] = *reinterpret_cast<const f32x4*>(ap + 8 * g);
    if (WLDS) {
      // A group's weights [8 k][32 outputs] = 1 KiB = ONE 16-byte load per lane (lane -> k = lane >> 3, 4 outputs) instead of
      // four 4-byte column loads per lane: the long-K FCs are bound by L2 request rate.
      rw[d] = *reinterpret_cast<const f32x4*>(wq + (int64_t)(8 * g) * Nout);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) rw[d][u] = bp[(int64_t)(8 * g + u) * Nout];
    }
  };
  // Every fetch is UNCONDITIONAL (the group index is clamped, the tail re-fetches the last group into slots nobody reads): a fetch under a branch
  // makes hipcc's wait-count pass join two different load histories and fall back to vmcnt(0) after every load - the ring then pipelines nothing.
#pragma unroll
  for (int d = 0; d < D; ++d) fetch(min(d, ng - 1), d);
  // (WLDS) the weight tile is private to the wave (same-wave LDS operations execute in order), so no barrier is involved.
  float* tw = &wtile[WLDS ? wid : 0][(lane >> 3) * 32 + (lane & 7) * 4];
  const float* tr = &wtile[WLDS ? wid : 0][(4 * hh) * 32 + i];
  auto consume = [&](int g, int d) {
    f32x4 a = ra[d];
    const f32x4 w4 = rw[d];
    if (sp) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(sp + 8 * g), t4 = *reinterpret_cast<const f32x4*>(sp + Cin + 8 * g);
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = a[u] * s4[u] + t4[u];
    }
    if (WLDS) {
      *reinterpret_cast<f32x4*>(tw) = w4;
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], tr[u * 32], acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], w4[u], acc, 0, 0, 0);
    }
  };
  const int nfull = ng / D * D;
  for (int g0 = 0; g0 < nfull; g0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      consume(g0 + d, d);
      fetch(min(g0 + d + D, ng - 1), d);
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (nfull + d < ng) consume(nfull + d, d);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wid][((r & 3) + 8 * (r >> 2) + 4 * hh) * 33 + i] = acc[r];
  __syncthreads();
  for (int e = tid; e < 32 * 32; e += KW * 64) {
    const int r = e >> 5, c = e & 31;
    if (b0 + r < B && n0 + c < Nout) {
      float v = 0.f;
#pragma unroll
      for (int k2 = 0; k2 < KW; ++k2) v += red[k2][r * 33 + c];        // slice order: reproducible
      v += bias ? bias[n0 + c] : 0.f;
      if (act == 1) v = fmaxf(v, 0.f);
      else if (act == 2) v = 1.0f / (1.0f + __expf(-v));
      out[(int64_t)(b0 + r) * ldout + n0 + c] = v;
    }
  }
}

// SE building blocks used by the forward schedule: per-segment channel means, and the gate application.
template <bool F16>
__global__ __launch_bounds__(NT) void seg_mean_kernel(const bf16_t* __restrict__ z, int64_t ldz, int T, int C, float* __restrict__ out) {
  __shared__ float red[2][1024];
  const int tid = threadIdx.x, c8 = tid & 127, grp = tid >> 7;
  const int cbase = blockIdx.y * 1024 + c8 * 8;
  const int64_t base = (int64_t)blockIdx.x * T;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (cbase < C) {
    const bf16_t* zp = z + base * ldz + cbase;
    int t = grp;
    for (; t + 6 < T; t += 8) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4*>(zp + (int64_t)(t + 2 * u) * ldz);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[8];
        unpack8t<F16>(v[u], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += f[e];
      }
    }
    for (; t < T; t += 2) {
      float f[8];
      unpack8t<F16>(*reinterpret_cast<const u32x4*>(zp + (int64_t)t * ldz), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += f[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[grp][c8 * 8 + e] = s[e];
  __syncthreads();
  if (grp == 0 && cbase < C) {
    const float invT = 1.0f / (float)T;
#pragma unroll
    for (int e = 0; e < 8; ++e) out[(int64_t)blockIdx.x * C + cbase + e] = (red[0][c8 * 8 + e] + red[1][c8 * 8 + e]) * invT;
  }
}

// out[b,t,c] = bf16(gate[b,c] * z[b,t,c] + x[b,t,c]);  grid (B, ceil(T/TCH)), thread = 8 channels x frame stripe
constexpr int SE_TCH = 32;
template <bool F16>
__global__ __launch_bounds__(NT) void se_apply_kernel(const bf16_t* __restrict__ z, int64_t ldz, const bf16_t* __restrict__ x,
                                                     int64_t ldx, const float* __restrict__ gate, bf16_t* __restrict__ out,
                                                     int64_t ldo, int T, int C) {
  const int nch8 = C >> 3, ngrp = NT / nch8;
  const int tid = threadIdx.x, c8 = tid % nch8, grp = tid / nch8;
  const int64_t base = (int64_t)blockIdx.x * T;
  const int t0 = blockIdx.y * SE_TCH, t1 = min(T, t0 + SE_TCH);
  float g[8];
  {
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gate + (int64_t)blockIdx.x * C + c8 * 8);
    const f32x4 g1 = *reinterpret_cast<const f32x4*>(gate + (int64_t)blockIdx.x * C + c8 * 8 + 4);
    g[0] = g0[0]; g[1] = g0[1]; g[2] = g0[2]; g[3] = g0[3]; g[4] = g1[0]; g[5] = g1[1]; g[6] = g1[2]; g[7] = g1[3];
  }
  const bf16_t* zp = z + base * ldz + c8 * 8;
  const bf16_t* xp = x + base * ldx + c8 * 8;
  bf16_t* op = out + base * ldo + c8 * 8;
  for (int t = t0 + grp; t < t1; t += ngrp) {
    float fz[8], fx[8];
    // z is read exactly once, here: non-temporal (5.0 -> 5.3 TB/s; nt on x and on the store as well measured no better)
    unpack8t<F16>(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(zp + (int64_t)t * ldz)), fz);
    unpack8t<F16>(*reinterpret_cast<const u32x4*>(xp + (int64_t)t * ldx), fx);
#pragma unroll
    for (int e = 0; e < 8; ++e) fz[e] = g[e] * fz[e] + fx[e];
    *reinterpret_cast<u32x4*>(op + (int64_t)t * ldo) = pack8t<F16>(fz);
  }
}

// ------------------------------------------------------------------------------------------
// ASP pooling: softmax over frames per (segment, channel), attention-weighted mean and std.
// grid (B, C/64); thread = (channel cl = tid & 63, frame group g = tid >> 6); three sweeps
// (max; sum-exp + weighted sum; weighted variance) - the 77-KB working set stays in L2.
template <bool F16>
__global__ __launch_bounds__(NT) void asp_pool_kernel(const float* __restrict__ logits, int64_t ldl,
                                                     const bf16_t* __restrict__ h, int64_t ldh, int T, int C,
                                                     float* __restrict__ pooled) {
  __shared__ float red[4][64];
  __shared__ float red2[4][64];
  const int tid = threadIdx.x;
  const int cl = tid & 63, g = tid >> 6;
  const int c = blockIdx.y * 64 + cl;
  const int64_t base = (int64_t)blockIdx.x * T;
  const float* lp = logits + base * ldl + c;
  const bf16_t* hp = h + base * ldh + c;

  float mx = -INFINITY;
  for (int t = g; t < T; t += 4) mx = fmaxf(mx, lp[(int64_t)t * ldl]);
  red[g][cl] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0][cl], red[1][cl]), fmaxf(red[2][cl], red[3][cl]));
  __syncthreads();

  float l = 0.f, s1 = 0.f;
  for (int t = g; t < T; t += 4) {
    const float w = __expf(lp[(int64_t)t * ldl] - mx);
    l += w;
    s1 += w * load1t<F16>(hp + (int64_t)t * ldh);
  }
  red[g][cl] = l;
  red2[g][cl] = s1;
  __syncthreads();
  l = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
  s1 = (red2[0][cl] + red2[1][cl]) + (red2[2][cl] + red2[3][cl]);
  __syncthreads();
  const float mu = s1 / l;

  float s2 = 0.f;
  for (int t = g; t < T; t += 4) {
    const float w = __expf(lp[(int64_t)t * ldl] - mx);
    const float d = load1t<F16>(hp + (int64_t)t * ldh) - mu;
    s2 += w * d * d;
  }
  red[g][cl] = s2;
  __syncthreads();
  if (g == 0) {
    s2 = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    pooled[(int64_t)blockIdx.x * 2 * C + c] = mu;
    pooled[(int64_t)blockIdx.x * 2 * C + C + c] = sqrtf(fmaxf(s2 / l, 1e-12f));
  }
}

// ------------------------------------------------------------------------------------------
// Fused attentive-statistics pooling: attention logits (MFMA) + softmax over frames + weighted
// mean/std, one workgroup per (segment, 128-channel block).  The [T, 3072] fp32 logits never exist
// in HBM.  The GEMM is oriented with FRAMES on the accumulator registers and the CHANNEL on the lane (A = attention hidden [T x 128], B = W2 rows of
// the wave's 32 channels), so each lane owns one channel and the softmax over frames is a
// register-local reduction plus one exchange with lane^32.
//   LDS (57 KiB, one buffer used twice): the segment's attention-hidden tile, 256-B rows with the
//   16-B chunk index XORed by (row & 15) (conflict-free ds_read_b128 A fragments), then - after the
//   MFMA phase - the segment's [T x 128] slab of h, read back 2 bytes per lane (64 B per half-wave).
template <int NTILES, bool F16>
__global__ __launch_bounds__(NT, 2) void asp_fused_kernel(const bf16_t* __restrict__ ah, int64_t ldah,
                                                      const bf16_t* __restrict__ w2, const float* __restrict__ b2,
                                                      const bf16_t* __restrict__ h, int64_t ldh, int T, int C,
                                                      float* __restrict__ pooled) {
  constexpr int ROWS = 32 * NTILES;
  __shared__ __attribute__((aligned(16))) char lds[ROWS * 256];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int col = lane & 31, hh = lane >> 5;
  const int seg = blockIdx.y;                        // channel block is the fast grid index: the workgroups of one
  const int64_t base = (int64_t)seg * T;             // segment run back to back and share its hidden tile in L2
  const int cblk = blockIdx.x * 128;
  const int ch = cblk + wid * 32 + col;

  // ---- phase A: attention-hidden tile -> LDS (rows >= T are zero)
  for (int id = tid; id < ROWS * 16; id += NT) {
    const int r = id >> 4, c = id & 15;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r < T) v = *reinterpret_cast<const u32x4*>(ah + (base + r) * ldah + c * 8);
    *reinterpret_cast<u32x4*>(lds + r * 256 + ((c ^ (r & 15)) << 4)) = v;
  }
  // B operand: W2 rows of this lane's channel, resident for the whole tile
  bf16x8 bfrag[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) bfrag[ks] = *reinterpret_cast<const bf16x8*>(w2 + (int64_t)ch * 128 + ks * 16 + hh * 8);
  __syncthreads();

  // h slab [T x 128 channels]: its HBM loads are issued NOW and ride under the MFMA phase; they land in the
  // same LDS buffer once every wave has finished reading the hidden tile
  constexpr int HP = (ROWS * 16 + NT - 1) / NT;
  u32x4 hpre[HP];
#pragma unroll
  for (int i = 0; i < HP; ++i) {
    const int id = tid + NT * i, r = id >> 4, c = id & 15;
    hpre[i] = u32x4{0u, 0u, 0u, 0u};
    if (r < T) hpre[i] = *reinterpret_cast<const u32x4*>(h + (base + r) * ldh + cblk + c * 8);
  }

  f32x16 acc[NTILES];
#pragma unroll
  for (int rt = 0; rt < NTILES; ++rt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
    const int row = rt * 32 + col;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(lds + row * 256 + (((ks * 2 + hh) ^ (row & 15)) << 4));
      acc[rt] = mfma_32x32x16<F16>(a, bfrag[ks], acc[rt]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < HP; ++i) {
    const int id = tid + NT * i, r = id >> 4, c = id & 15;
    if (r < T) *reinterpret_cast<u32x4*>(lds + r * 256 + c * 16) = hpre[i];
  }
  // softmax max while the loads are in flight (registers only; branch-free: rows >= T count as -inf)
  float mx = -INFINITY;
#pragma unroll
  for (int rt = 0; rt < NTILES; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      mx = fmaxf(mx, t < T ? acc[rt][r] : -INFINITY);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));          // bias is constant over frames: max(v + bias) = max(v) + bias
  __syncthreads();
  // ---- phase C: weighted moments about K = h[t = 0] (shifted single pass, fp32).  Rows >= T hold
  // stale but finite bytes of the hidden tile and get weight 0, so every LDS read is unconditional.
  const bf16_t* hl = reinterpret_cast<const bf16_t*>(lds) + wid * 32 + col;
  const float K = load1t<F16>(hl);
  float l = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int rt = 0; rt < NTILES; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      const float e = t < T ? __expf(acc[rt][r] - mx) : 0.f;
      const float d = load1t<F16>(hl + t * 128) - K;
      l += e;
      s1 = fmaf(e, d, s1);
      s2 = fmaf(e * d, d, s2);
      if (r == 15) __builtin_amdgcn_sched_barrier(0);   // one row tile at a time: keeps 16, not 112, reads in flight
    }
  l += __shfl_xor(l, 32, 64);
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  if (hh == 0) {
    const float a = s1 / l;
    pooled[(int64_t)seg * 2 * C + ch] = K + a;
    pooled[(int64_t)seg * 2 * C + C + ch] = sqrtf(fmaxf(s2 / l - a * a, 1e-12f));
  }
}

// ------------------------------------------------------------------------------------------
// The same computation, one workgroup (8 waves) per SEGMENT.  The attention-hidden tile is loaded into LDS once
// per segment instead of once per 128 channels, and every wave walks its own share of the 32-channel blocks with
// no workgroup barrier: per 32-frame row tile  MFMA logits -> ONLINE softmax (running max, the three running sums
// are rescaled when it grows) -> weighted moments against the wave's PRIVATE [208 x 32] slab of h in LDS.
//   * logits are evaluated once and a tile's 16 accumulator registers die with the tile (the two-pass form
//     needs all 112, or the MFMAs twice);
//   * a wave issues in order, so the NEXT tile's chain of 8 dependent MFMAs is interleaved, one MFMA per ~13
//     vector instructions, with THIS tile's arithmetic (sched_group_barrier), both in one basic block;
//   * the slab is refilled by LDS-DMA a row tile at a time: as soon as a tile's rows have been read, the same
//     rows of the wave's next block are requested, so every piece has a whole block (~4 us) to arrive.  vmcnt
//     retires in order and every block issues the same 13 DMA + 8 weight loads, so the wait before tile rt is the
//     constant "all but the 19 (20 for the last tile) youngest"; W2 fragments are double-buffered in registers and
//     requested a block ahead so that they are OLDER than the pieces they would otherwise force to land;
//   * elementwise steps run on frame pairs with packed fp32 instructions; slab values are read with
//     ds_read_u16_d16_hi into registers whose low half stays zero (the register IS the fp32 value);
//   * all LDS reads of the hot loop are inline asm: the compiler otherwise hoists the block's 112 + 56 reads to the
//     top and spills them.
//   LDS: hidden tile 224 x 256 B (16-B chunk index XOR (row & 15)) + 8 x (208 x 64 B) slabs (chunk XOR
//        ((row >> 2) & 3): the two half-waves read rows 4 apart) = 163 840 B, all of a CU's LDS.
// Results differ from asp_fused_kernel by fp32 rounding only (rescaling of the running sums).
constexpr int SEG_ROWS = 208;                 // frames a private h slab holds (T <= 208)
constexpr int SEG_NT = 512;
constexpr int SEG_HID_ROWS = 224;               // 7 MFMA row tiles of 32; rows >= T are zero
constexpr int SEG_HID_BYTES = SEG_HID_ROWS * 256;
constexpr int SEG_SLAB_BYTES = SEG_ROWS * 64;
constexpr int SEG_LDS = SEG_HID_BYTES + 8 * SEG_SLAB_BYTES;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const void __attribute__((address_space(1)))* seg_gptr_t;
typedef void __attribute__((address_space(3)))* seg_lptr_t;

#define SEG_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <bool PACKED, bool F16>
__global__ __launch_bounds__(SEG_NT, 2) void asp_seg_kernel(const bf16_t* __restrict__ ah, int64_t ldah,
                                                           const bf16_t* __restrict__ w2, const bf16_t* __restrict__ h,
                                                           int64_t ldh, int T, int C, float* __restrict__ pooled, int64_t hblk) {
  // hblk != 0: h is K-BLOCKED, [C / 64][rows][64] with hblk = rows * 64 elements between 64-channel blocks and ldh = 64 (sdk_hip.h
  // SDK_GEMM_C_KBLOCKED): a slab piece's 64-byte rows are 128 bytes apart instead of 2 ldh (-12 % on this kernel: DESIGN 5.3)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NTILES = 7;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, hh = lane >> 5;
  const int seg = blockIdx.x;
  const int64_t base = (int64_t)seg * T;

  // ---- attention-hidden tile -> LDS, once per segment (rows >= T are zero)
  for (int id = tid; id < SEG_HID_ROWS * 16; id += SEG_NT) {
    const int r = id >> 4, c = id & 15;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r < T) v = *reinterpret_cast<const u32x4*>(ah + (base + r) * ldah + c * 8);
    *reinterpret_cast<u32x4*>(lds + r * 256 + ((c ^ (r & 15)) << 4)) = v;
  }
  char* slab = lds + SEG_HID_BYTES + wid * SEG_SLAB_BYTES;
  const int nblk = C / 32;                                   // 32-channel blocks; wave w takes blocks w, w + 8, ...
  const int lrow = lane >> 2, lch = lane & 3;

  // Slab pieces: DMA instruction i fills rows [16 i, 16 i + 16) x 4 chunks of 16 B; row tile rt = instructions 2 rt and
  // 2 rt + 1 (the last tile has one: rows 192..207).  LDS position (row, lch) receives the SOURCE chunk
  // lch ^ ((row >> 2) & 3) (== ((lrow >> 2) & 3): 16 i does not reach those bits).  Rows past the segment re-read its
  // last frame instead of being skipped, so EVERY block issues exactly 13 instructions (the vmcnt arithmetic below
  // depends on it) and the slab never holds non-finite stale bytes.
  const uint32_t chunk_off = (uint32_t)((lch ^ ((lrow >> 2) & 3)) << 4);
  uint32_t voff0 = (uint32_t)lrow * (uint32_t)ldh * 2u + chunk_off;       // the per-lane part of every FULL piece's address
  auto fetch_piece = [&](int blk, int rt) {
    const char* hb = reinterpret_cast<const char*>(h + base * ldh + (hblk ? (int64_t)(blk >> 1) * hblk + (blk & 1) * 32 : (int64_t)blk * 32));     // wave-uniform
#pragma unroll
    for (int i = 2 * rt; i < 2 * rt + 2 && i < SEG_ROWS / 16; ++i) {
      // address = scalar base + 32-bit lane offset; the row part of a full piece moves into the scalar base
      const bool full = (i + 1) * 16 <= T;                        // wave-uniform
      const char* sb = full ? hb + (int64_t)i * 32 * ldh : hb;
      const uint32_t vo = full ? voff0 : (uint32_t)min(i * 16 + lrow, T - 1) * (uint32_t)ldh * 2u + chunk_off;
      __builtin_amdgcn_global_load_lds((seg_gptr_t)(sb + vo), (seg_lptr_t)(slab + i * 1024), 16, 0, 0);
    }
  };
  // hidden-tile fragment addresses: chunk (2 ks + hh) ^ (row & 15) with row & 15 == col & 15 for every row tile
  uint32_t aoff[8];                                          // dynamic LDS starts at offset 0 (the kernel has no static LDS)
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) aoff[ks] = col * 256 + (((ks * 2 + hh) ^ (col & 15)) << 4);
  bf16x8 bA[8], bB[8];                                       // W2 rows of this lane's channel, this block / next block
  // PACKED: w2 is the host's fragment-ordered copy [block][ks 8][lane 64][8] (ecapa_layout.h EL_ASP_W2PACK): one 1-KiB contiguous
  // load per k-step instead of 32 row pieces of 16 B - the row-major form touches every 128-byte line of W2 four times
  // (3.1 GB of L2 line traffic per 1000 segments for 0.79 GB of weights)
  auto fetch_w2 = [&](int blk, bf16x8* dst) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if constexpr (PACKED) dst[ks] = *reinterpret_cast<const bf16x8*>(w2 + ((int64_t)(blk * 8 + ks) * 64 + lane) * 8);
      else dst[ks] = *reinterpret_cast<const bf16x8*>(w2 + (int64_t)(blk * 32 + col) * 128 + ks * 16 + hh * 8);
    }
  };
  const uint32_t slab_off = (uint32_t)(SEG_HID_BYTES + wid * SEG_SLAB_BYTES);
  // frame t = 32 rt + 8 (r >> 2) + (r & 3) + 4 hh sits at chunk (col >> 3) ^ ((t >> 2) & 3) = (col >> 3) ^ hh ^ (2 (r >> 2) & 3):
  // two per-lane bases (compile-time part 0 or 2) + an immediate
  const uint32_t hoff0 = slab_off + hh * 4 * 64 + ((((col >> 3) ^ hh) ^ 0) << 4) + (col & 7) * 2;
  const uint32_t hoff2 = slab_off + hh * 4 * 64 + ((((col >> 3) ^ hh) ^ 2) << 4) + (col & 7) * 2;
  const uint32_t koff = slab_off + ((col >> 3) << 4) + (col & 7) * 2;               // h[t = 0]
  if (wid < nblk) {
    fetch_w2(wid, bA);                                        // oldest: older than every piece
#pragma unroll
    for (int rt = 0; rt < NTILES; ++rt) fetch_piece(wid, rt);
  }
  __syncthreads();                                           // the only workgroup barrier: the hidden tile is complete

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float LOG2E = 1.44269502162933349609375f;
  uint32_t hv[16];                                           // low halves stay zero: ds_read_u16_d16_hi writes the high half only
#pragma unroll
  for (int r = 0; r < 16; ++r) hv[r] = 0u;
  uint32_t kv = 0u;
  int tlim = T - 4 * hh;                                       // frame (const + 4 hh) exists  <=>  const < tlim

  // a slab element as read by ds_read_u16_d16_hi (value in the register's HIGH half, low half zero): bf16 - the register is the fp32 value;
  // fp16 - one conversion
  auto slab_f32 = [](uint32_t v) -> float {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (uint16_t)(v >> 16));
    else return __uint_as_float(v);
  };
  auto process = [&](int blk, const bf16x8* bcur, bf16x8* bnext) {
    const int ch = blk * 32 + col;
    const bool more = blk + 8 < nblk;
    asm volatile("" : "+v"(tlim), "+v"(voff0));                // keeps frame predicates / piece addresses from being hoisted and spilled
    SEG_WAIT_VM(13);                                           // this block's W2 rows (requested a block ago) are older than its 13 pieces
    if (more) fetch_w2(blk + 8, bnext);
    auto read_frags = [&](int rt, bf16x8* a) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[ks]) : "v"(aoff[ks]), "n"(rt * 32 * 256));
    };
    auto chain = [&](const bf16x8* a) {
      f32x16 acc = mfma_32x32x16<F16>(a[0], bcur[0], zero16);
#pragma unroll
      for (int ks = 1; ks < 8; ++ks) acc = mfma_32x32x16<F16>(a[ks], bcur[ks], acc);
      return acc;
    };
    f32x16 cur;
    {
      bf16x8 a[8];
      read_frags(0, a);
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) :: "memory");
      cur = chain(a);
    }
    float m = -INFINITY, Kf = 0.f;
    f32x2 l2 = {0.f, 0.f}, s12 = {0.f, 0.f}, s22 = {0.f, 0.f};   // softmax denominator and the two shifted moments, (even, odd) frames
#pragma unroll
    for (int rt = 0; rt < NTILES; ++rt) {
      // piece rt of this block has landed once all but the ops issued after it are done (see the header comment)
      if (more) { if (rt < NTILES - 1) SEG_WAIT_VM(19); else SEG_WAIT_VM(20); }
      else if (rt == 0) SEG_WAIT_VM(11); else if (rt == 1) SEG_WAIT_VM(9); else if (rt == 2) SEG_WAIT_VM(7);
      else if (rt == 3) SEG_WAIT_VM(5); else if (rt == 4) SEG_WAIT_VM(3); else if (rt == 5) SEG_WAIT_VM(1); else SEG_WAIT_VM(0);
      if (rt == 0) asm volatile("ds_read_u16_d16_hi %0, %1" : "+v"(kv) : "v"(koff));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (rt * 32 + 8 * (r >> 2) + (r & 3) + 4 < SEG_ROWS)                         // compile-time: frames 208..223 do not exist (weight 0, stale h)
          asm volatile("ds_read_u16_d16_hi %0, %1 offset:%2" : "+v"(hv[r]) : "v"(((2 * (r >> 2)) & 3) ? hoff2 : hoff0),
                       "n"((rt * 32 + 8 * (r >> 2) + (r & 3)) * 64));
      }
      // the next tile's fragments are read now; ONE wait covers both kinds of reads, so the next tile's MFMAs and this
      // tile's arithmetic sit in the same scheduling region
      bf16x8 a[8];
      if (rt + 1 < NTILES) {
        read_frags(rt + 1, a);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]), "+v"(hv[4]), "+v"(hv[5]), "+v"(hv[6]), "+v"(hv[7]),
                       "+v"(hv[8]), "+v"(hv[9]), "+v"(hv[10]), "+v"(hv[11]), "+v"(hv[12]), "+v"(hv[13]), "+v"(hv[14]), "+v"(hv[15]),
                       "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(kv)
                     :: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]), "+v"(hv[4]), "+v"(hv[5]), "+v"(hv[6]), "+v"(hv[7]),
                       "+v"(hv[8]), "+v"(hv[9]), "+v"(hv[10]), "+v"(hv[11]), "+v"(hv[12]), "+v"(hv[13]), "+v"(hv[14]), "+v"(hv[15]), "+v"(kv)
                     :: "memory");
      }
      // this tile's slab rows have been read: request the same rows of the wave's next block
      if (more) fetch_piece(blk + 8, rt);
      if (rt == 0) Kf = slab_f32(kv);
      f32x16 nxt = cur;
      auto step = [&](const bool masked) {
        if (rt + 1 < NTILES) nxt = chain(a);
        // online softmax: the running maximum may grow with this tile; the sums so far are rescaled to it
        float tm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) tm = fmaxf(tm, (!masked || rt * 32 + (r & 3) + 8 * (r >> 2) < tlim) ? cur[r] : -INFINITY);
        const float mn = fmaxf(m, tm);
        const float sc = __builtin_amdgcn_exp2f((m - mn) * LOG2E);   // first tile: exp2(-inf) = 0 (tile 0 always has frames)
        const f32x2 sc2 = {sc, sc};
        l2 *= sc2; s12 *= sc2; s22 *= sc2;
        m = mn;
        const float mc = -m * LOG2E;
        const f32x2 mc2 = {mc, mc}, K2 = {Kf, Kf}, log2e2 = {LOG2E, LOG2E};
        // frame PAIRS on packed fp32 instructions: even and odd frames of the lane accumulate side by side (merged at the
        // end of the block); 5 vector-issue slots per frame instead of 9
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 x = {cur[r], cur[r + 1]};
          const f32x2 y = x * log2e2 + mc2;                        // exp(v - m) = exp2(v log2(e) - m log2(e)): one packed FMA
          f32x2 e = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
          if (masked) {
            const int t0 = rt * 32 + (r & 3) + 8 * (r >> 2);
            e[0] = t0 < tlim ? e[0] : 0.f;
            e[1] = t0 + 1 < tlim ? e[1] : 0.f;
          }
          f32x2 d = {slab_f32(hv[r]), slab_f32(hv[r + 1])};
          d = d - K2;
          const f32x2 ed = e * d;
          l2 += e;
          s12 = e * d + s12;
          s22 = ed * d + s22;
        }
        if (rt + 1 < NTILES) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA of the next tile's chain
            __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);    // a slice of this tile's vector work
          }
        }
      };
      if ((rt + 1) * 32 <= T) step(false);                     // wave-uniform: no frame of this tile needs masking
      else step(true);
      asm volatile("" : "+v"(l2), "+v"(s12), "+v"(s22), "+v"(m));   // pins this tile's arithmetic before the next tile's reads
      cur = nxt;
    }
    // the two lanes of a channel (frame halves hh = 0 / 1) merge their running states
    const float M = fmaxf(m, __shfl_xor(m, 32, 64));
    const float f = __builtin_amdgcn_exp2f((m - M) * LOG2E);
    float l = (l2[0] + l2[1]) * f, s1 = (s12[0] + s12[1]) * f, s2 = (s22[0] + s22[1]) * f;
    l += __shfl_xor(l, 32, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (hh == 0) {
      const float a = s1 / l;
      pooled[(int64_t)seg * 2 * C + ch] = Kf + a;
      pooled[(int64_t)seg * 2 * C + C + ch] = sqrtf(fmaxf(s2 / l - a * a, 1e-12f));
    }
  };
  for (int blk = wid; blk < nblk; blk += 16) {
    process(blk, bA, bB);
    if (blk + 8 < nblk) process(blk + 8, bB, bA);
  }
}
#undef SEG_WAIT_VM

// ------------------------------------------------------------------------------------------
// k3: L2-normalise rows; one wave per row.
__global__ __launch_bounds__(NT) void l2norm_kernel(const float* __restrict__ X, int N, int d, float* __restrict__ E,
                                                   bf16_t* __restrict__ Eb, float* __restrict__ resid) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* x = X + (int64_t)row * d;
  float ss = 0.f;
  for (int i = lane; i < d; i += 64) ss += x[i] * x[i];
  ss = wave_sum(ss);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  float rr = 0.f;
  for (int i = lane; i < d; i += 64) {
    const float e = x[i] * inv;
    const bf16_t eb = f32_to_bf16(e);
    if (E) E[(int64_t)row * d + i] = e;
    if (Eb) Eb[(int64_t)row * d + i] = eb;
    const float dd = e - bf16_to_f32(eb);
    rr += dd * dd;
  }
  rr = wave_sum(rr);
  if (resid && lane == 0) resid[row] = sqrtf(rr);
}

}  // namespace

extern "C" size_t sdk_se_workspace_bytes(int B, int C, int Cse) {
  return B > 0 ? ((size_t)B * C * 2 + (size_t)B * Cse) * sizeof(float) : 0;
}

// (sdk_ecapa_forward passes the blob's format; the exported wrapper below takes the context's "precision" option)
int se_gate_residual_impl(sdk_ctx* ctx, const uint16_t* z, int64_t ldz, const uint16_t* x, int64_t ldx,
                          const float* w1t, const float* b1, const float* w2t, const float* b2,
                          uint16_t* out, int64_t ldo, int B, int T, int C, int Cse, const float* mean_in,
                          void* ws, size_t ws_bytes, void* stream, bool f16) {
  SDK_REQUIRE(ctx && z && x && w1t && b1 && w2t && b2 && out, "sdk_se_gate_residual: null argument");
  SDK_REQUIRE(B > 0 && T > 0, "sdk_se_gate_residual: empty batch");
  SDK_REQUIRE(C % 8 == 0 && C / 8 <= NT && NT % (C / 8) == 0, "sdk_se_gate_residual: C=%d unsupported (need C/8 | 256)", C);
  SDK_REQUIRE(Cse > 0 && Cse <= NT && NT % Cse == 0 && Cse <= C, "sdk_se_gate_residual: Cse=%d unsupported (need Cse | 256)", Cse);
  SDK_REQUIRE(ldz % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, "sdk_se_gate_residual: row strides must be multiples of 8");
  if (ws && ws_bytes >= sdk_se_workspace_bytes(B, C, Cse)) {
    // split schedule: channel means (one sweep of z) -> the two gate FCs batched over all segments on the
    // matrix pipe (weights read once per 32 segments instead of once per segment) -> gate*z + x sweep
    float* mean = (float*)ws;
    float* hid = mean + (size_t)B * C;
    float* gate = hid + (size_t)B * Cse;
    if (mean_in) {
      mean = const_cast<float*>(mean_in);                   // squeeze already produced by the GEMM epilogue
    } else {
      ProfScope ps(ctx, stream, SDK_K_SE_GATE, 1.0 * B * T * C, 2.0 * B * T * C);
      hipLaunchKernelGGL(f16 ? seg_mean_kernel<true> : seg_mean_kernel<false>, dim3(B, ceil_div(C, 1024)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)z, ldz, T, C, mean);
    }
    SDK_LAUNCH_CHECK();
    if (int rc = sdk_rows_fc(ctx, mean, C, nullptr, nullptr, w1t, b1, hid, Cse, B, C, Cse, 1, stream)) return rc;
    if (int rc = sdk_rows_fc(ctx, hid, Cse, nullptr, nullptr, w2t, b2, gate, C, B, Cse, C, 2, stream)) return rc;
    {
      ProfScope ps(ctx, stream, SDK_K_SE_GATE, 2.0 * B * T * C, 6.0 * B * T * C);
      hipLaunchKernelGGL(f16 ? se_apply_kernel<true> : se_apply_kernel<false>, dim3(B, ceil_div(T, SE_TCH)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)z, ldz,
                         (const bf16_t*)x, ldx, gate, (bf16_t*)out, ldo, T, C);
    }
    SDK_LAUNCH_CHECK();
    return 0;
  }
  SDK_REQUIRE(!mean_in, "sdk_se_gate_residual: a precomputed mean needs the workspace form");
  const int ngrp = NT / (C / 8);
  const size_t lds = (size_t)(ngrp * C + C + Cse) * sizeof(float);
  SDK_REQUIRE((size_t)(NT / Cse) * Cse <= (size_t)ngrp * C, "sdk_se_gate_residual: scratch too small");
  ProfScope ps(ctx, stream, SDK_K_SE_GATE, 3.0 * B * T * C, 6.0 * B * T * C);   // z, x read + out written (z re-read from L2)
  hipLaunchKernelGGL(f16 ? se_gate_residual_kernel<true> : se_gate_residual_kernel<false>, dim3(B), dim3(NT), lds, (hipStream_t)stream, (const bf16_t*)z, ldz,
                     (const bf16_t*)x, ldx, w1t, b1, w2t, b2, (bf16_t*)out, ldo, T, C, Cse);
  SDK_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdk_se_gate_residual(sdk_ctx* ctx, const uint16_t* z, int64_t ldz, const uint16_t* x, int64_t ldx,
                                    const float* w1t, const float* b1, const float* w2t, const float* b2,
                                    uint16_t* out, int64_t ldo, int B, int T, int C, int Cse, const float* mean_in,
                                    void* ws, size_t ws_bytes, void* stream) {
  return se_gate_residual_impl(ctx, z, ldz, x, ldx, w1t, b1, w2t, b2, out, ldo, B, T, C, Cse, mean_in, ws, ws_bytes, stream, ctx && ctx->precision == 2);
}

int asp_stats_impl(sdk_ctx* ctx, const uint16_t* h, int64_t ldh, int B, int T, int C, float* out_ctx, void* stream, bool f16) {
  SDK_REQUIRE(ctx && h && out_ctx, "sdk_asp_stats: null argument");
  SDK_REQUIRE(B > 0 && T > 0 && C % 8 == 0 && ldh % 8 == 0, "sdk_asp_stats: bad shape (C=%d ldh=%lld)", C, (long long)ldh);
  ProfScope ps(ctx, stream, SDK_K_ASP_STATS, 3.0 * B * T * C, 2.0 * B * T * C);
  hipLaunchKernelGGL(f16 ? asp_stats_kernel<true> : asp_stats_kernel<false>, dim3(B, ceil_div(C, 1024)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)h,
                     ldh, T, C, out_ctx);
  SDK_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdk_asp_stats(sdk_ctx* ctx, const uint16_t* h, int64_t ldh, int B, int T, int C, float* out_ctx,
                             void* stream) {
  return asp_stats_impl(ctx, h, ldh, B, T, C, out_ctx, stream, ctx && ctx->precision == 2);
}

extern "C" int sdk_asp_stats_fmt(sdk_ctx* ctx, const uint16_t* h, int64_t ldh, int B, int T, int C, float* out_ctx, int precision, void* stream) {
  SDK_REQUIRE(precision == 0 || precision == 2, "sdk_asp_stats_fmt: precision=%d (0: bf16 elements, 2: fp16 elements)", precision);
  return asp_stats_impl(ctx, h, ldh, B, T, C, out_ctx, stream, precision == 2);
}

extern "C" int sdk_rows_fc(sdk_ctx* ctx, const float* in, int64_t ldin, const float* in_scale, const float* in_shift,
                           const float* wt, const float* bias, float* out, int64_t ldout, int B, int Cin, int Nout,
                           int act, void* stream) {
  SDK_REQUIRE(ctx && in && wt && out, "sdk_rows_fc: null argument");
  SDK_REQUIRE(B > 0 && Cin > 0 && Nout > 0, "sdk_rows_fc: empty problem");
  SDK_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "sdk_rows_fc: in_scale and in_shift go together");
  SDK_REQUIRE(act >= 0 && act <= 2, "sdk_rows_fc: act=%d", act);
  ProfScope ps(ctx, stream, SDK_K_ROWS_FC, 2.0 * B * Cin * Nout, 4.0 * ((double)B * Cin + (double)Cin * Nout + (double)B * Nout));
  constexpr int kAffMax = 64 * 1024;                         // dynamic LDS of the MFMA kernels (beside up to 84 KB static): an input affine of Cin <= 8192
  const bool mfma_ok = ldin % 4 == 0 && ((uintptr_t)in % 16) == 0 && (!in_scale || (size_t)Cin * 8 <= (size_t)kAffMax);
  if (in_scale && mfma_ok) {
    if (sdk_lds_optin(ctx, (const void*)rows_fc_mfma_kernel<16, true>, kAffMax)) return 1;
    if (sdk_lds_optin(ctx, (const void*)rows_fc_mfma_kernel<16, false>, kAffMax)) return 1;
    if (sdk_lds_optin(ctx, (const void*)rows_fc_mfma_kernel<4, false>, kAffMax)) return 1;
  }
  const bool wlds_ok = Nout % 32 == 0 && ((uintptr_t)wt % 16) == 0;
  const size_t aff_bytes = in_scale ? (size_t)Cin * 8 : 0;      // the MFMA kernels stage the input affine in dynamic LDS ([2][Cin] fp32)
  if (mfma_ok && Cin % 128 == 0 && Cin >= 2048 && wlds_ok)   // long K (context bias, final FC): 16 K-slices, weights as 16-byte rows via LDS
    hipLaunchKernelGGL((rows_fc_mfma_kernel<16, true>), dim3(ceil_div(B, 32), ceil_div(Nout, 32)), dim3(1024), aff_bytes, (hipStream_t)stream,
                       in, ldin, in_scale, in_shift, wt, bias, out, ldout, B, Cin, Nout, act);
  else if (mfma_ok && Cin % 128 == 0 && Cin >= 2048)
    hipLaunchKernelGGL(rows_fc_mfma_kernel<16>, dim3(ceil_div(B, 32), ceil_div(Nout, 32)), dim3(1024), aff_bytes, (hipStream_t)stream,
                       in, ldin, in_scale, in_shift, wt, bias, out, ldout, B, Cin, Nout, act);
  else if (mfma_ok && Cin % 32 == 0)
    hipLaunchKernelGGL(rows_fc_mfma_kernel<4>, dim3(ceil_div(B, 32), ceil_div(Nout, 32)), dim3(NT), aff_bytes, (hipStream_t)stream,
                       in, ldin, in_scale, in_shift, wt, bias, out, ldout, B, Cin, Nout, act);
  else
    hipLaunchKernelGGL(rows_fc_kernel, dim3(ceil_div(B, FC_ROWS), ceil_div(Nout, FC_OUT)), dim3(NT), 0, (hipStream_t)stream,
                       in, ldin, in_scale, in_shift, wt, bias, out, ldout, B, Cin, Nout, act);
  SDK_LAUNCH_CHECK();
  return 0;
}

int asp_pool_impl(sdk_ctx* ctx, const float* logits, int64_t ldl, const uint16_t* h, int64_t ldh, int B,
                  int T, int C, float* pooled, void* stream, bool f16) {
  SDK_REQUIRE(ctx && logits && h && pooled, "sdk_asp_pool: null argument");
  SDK_REQUIRE(B > 0 && T > 0 && C % 64 == 0, "sdk_asp_pool: C=%d must be a multiple of 64", C);
  ProfScope ps(ctx, stream, SDK_K_ASP_POOL, 8.0 * B * T * C, 6.0 * B * T * C);
  hipLaunchKernelGGL(f16 ? asp_pool_kernel<true> : asp_pool_kernel<false>, dim3(B, C / 64), dim3(NT), 0, (hipStream_t)stream, logits, ldl,
                     (const bf16_t*)h, ldh, T, C, pooled);
  SDK_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdk_asp_pool(sdk_ctx* ctx, const float* logits, int64_t ldl, const uint16_t* h, int64_t ldh, int B,
                            int T, int C, float* pooled, void* stream) {
  return asp_pool_impl(ctx, logits, ldl, h, ldh, B, T, C, pooled, stream, ctx && ctx->precision == 2);
}

extern "C" int sdk_asp_fused_max_frames(void) { return 224; }

// Internal entry (sdk_ecapa_forward): w2p may be null, or the fragment-ordered copy of w2 (ecapa_layout.h EL_ASP_W2PACK).
static bool asp_seg_ok(const sdk_ctx* ctx, int T, int C) { return T > 96 && T <= SEG_ROWS && C % 256 == 0 && !ctx->no_asp_seg; }

int asp_fused_launch(sdk_ctx* ctx, const uint16_t* ah, int64_t ldah, const uint16_t* w2, const uint16_t* w2p, const float* b2,
                     const uint16_t* h, int64_t ldh, int B, int T, int C, int A, float* pooled, void* stream, bool kblocked, bool f16) {
  SDK_REQUIRE(ctx && ah && w2 && b2 && h && pooled, "sdk_asp_fused: null argument");
  const int64_t hblk = kblocked ? (int64_t)B * T * 64 : 0;
  if (kblocked) {
    SDK_REQUIRE(asp_seg_ok(ctx, T, C), "sdk_asp_fused_kblocked: only the per-segment form reads the K-blocked layout (96 < T <= %d, C %% 256 == 0, "
                "option asp_per_segment on); keep h row-major for T=%d, C=%d", SEG_ROWS, T, C);
    ldh = 64;
  }
  SDK_REQUIRE(A == 128, "sdk_asp_fused: attention width %d, this build is specialised for 128", A);
  SDK_REQUIRE(B > 0 && T > 0 && T <= 224, "sdk_asp_fused: T=%d frames unsupported (1..224); use sdk_conv_gemm + sdk_asp_pool", T);
  SDK_REQUIRE(C % 128 == 0 && ldah % 8 == 0 && ldh % 8 == 0, "sdk_asp_fused: C=%d must be a multiple of 128", C);
  ProfScope ps(ctx, stream, SDK_K_ASP_FUSED, 2.0 * B * T * (double)A * C, 2.0 * B * T * ((double)C + A) + 8.0 * B * C);
  if (asp_seg_ok(ctx, T, C)) {     // one workgroup per segment (hidden tile read once)
    const bool pk = w2p && !ctx->no_asp_packed;
    void (*kern)(const bf16_t*, int64_t, const bf16_t*, const bf16_t*, int64_t, int, int, float*, int64_t) =
        f16 ? (pk ? asp_seg_kernel<true, true> : asp_seg_kernel<false, true>) : (pk ? asp_seg_kernel<true, false> : asp_seg_kernel<false, false>);
    if (sdk_lds_optin(ctx, (const void*)kern, SEG_LDS)) return 1;
    hipLaunchKernelGGL(kern, dim3(B), dim3(SEG_NT), SEG_LDS, (hipStream_t)stream, (const bf16_t*)ah, ldah, (const bf16_t*)(pk ? w2p : w2),
                       (const bf16_t*)h, ldh, T, C, pooled, hblk);
    SDK_LAUNCH_CHECK();
    return 0;
  }
  const dim3 grid(C / 128, B);
  void (*kf)(const bf16_t*, int64_t, const bf16_t*, const float*, const bf16_t*, int64_t, int, int, float*) =
      T <= 96 ? (f16 ? asp_fused_kernel<3, true> : asp_fused_kernel<3, false>) : (f16 ? asp_fused_kernel<7, true> : asp_fused_kernel<7, false>);
  hipLaunchKernelGGL(kf, grid, dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)ah, ldah, (const bf16_t*)w2, b2, (const bf16_t*)h, ldh, T, C, pooled);
  SDK_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdk_asp_fused(sdk_ctx* ctx, const uint16_t* ah, int64_t ldah, const uint16_t* w2, const float* b2,
                             const uint16_t* h, int64_t ldh, int B, int T, int C, int A, float* pooled, void* stream) {
  return asp_fused_launch(ctx, ah, ldah, w2, nullptr, b2, h, ldh, B, T, C, A, pooled, stream, false, ctx && ctx->precision == 2);
}

extern "C" int sdk_asp_kblocked_ok(sdk_ctx* ctx, int T, int C) { return ctx && asp_seg_ok(ctx, T, C) ? 1 : 0; }

extern "C" int sdk_asp_fused_kblocked(sdk_ctx* ctx, const uint16_t* ah, int64_t ldah, const uint16_t* w2, const float* b2, const uint16_t* h,
                                      int B, int T, int C, int A, float* pooled, void* stream) {
  return asp_fused_launch(ctx, ah, ldah, w2, nullptr, b2, h, 64, B, T, C, A, pooled, stream, true, ctx && ctx->precision == 2);
}

extern "C" int sdk_l2norm(sdk_ctx* ctx, const float* X, int N, int d, float* E, uint16_t* Eb, float* resid,
                          void* stream) {
  SDK_REQUIRE(ctx && X, "sdk_l2norm: null argument");
  SDK_REQUIRE(N > 0 && d > 0, "sdk_l2norm: empty problem");
  ProfScope ps(ctx, stream, SDK_K_L2NORM, 4.0 * N * d, 10.0 * N * d);
  hipLaunchKernelGGL(l2norm_kernel, dim3(ceil_div(N, NT / 64)), dim3(NT), 0, (hipStream_t)stream, X, N, d, E,
                     (bf16_t*)Eb, resid);
  SDK_LAUNCH_CHECK();
  return 0;
}
