// Precise mode ("precision" 1): fp16 hi + lo plane helpers shared by hp.hip, fbank.hip and the forward schedule.
//   x = float(hi) + float(lo) / HP_LOSCALE,   hi = fp16(sat(x)),   lo = fp16((x - float(hi)) * HP_LOSCALE)
// x - float(hi) is exact in fp32 (|x - hi| <= ulp(hi) / 2); the pair carries 22 significand bits (fp16: 11 each, the sign of lo is the
// 23rd), down to an absolute floor of 2^-25 where lo reaches the fp16 subnormal range.
#pragma once
#include "common.hpp"

// What tools/probe/f16_split_probe measured on MI355X: v_mfma_f32_16x16x32_f16 KEEPS subnormal fp16 inputs (no flush), but a subnormal lo
// has an absolute grid of 2^-24, so an UNSCALED lo plane loses relative precision for |x| < 0.12 (error / sum|ab|: 4e-7 at |x| ~ 1,
// 2e-6 at |x| ~ 0.01).  Hence the ACTIVATION lo plane is stored times 2^11 (always in the normal range of its hi), and the GEMM multiplies
// it with W_hi * 2^-11 (a packed-fp16 multiply of the fragment in registers): the three products still share one accumulator.
// The WEIGHT planes are scaled per layer by a power of two instead (header in front of the planes, hp.hip).
namespace sdk_hp {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr float HP_LOSCALE = 2048.0f;                            // activation planes: lo = fp16((x - hi) * 2^11)
constexpr int HP_WHDR = 128;                                     // fp16 elements (256 B) in front of a weight slot's planes: float[0] = 1 / (layer scale 2^s)
constexpr float HP_MAX = 65504.0f;

__device__ __forceinline__ void split1(float v, _Float16& hi, _Float16& lo) {
  v = fminf(fmaxf(v, -HP_MAX), HP_MAX);                         // saturate instead of making an infinity (and then a NaN in lo)
  hi = (_Float16)v;
  lo = (_Float16)((v - (float)hi) * HP_LOSCALE);
}
__device__ __forceinline__ float join1(_Float16 hi, _Float16 lo) { return (float)hi + (float)lo * (1.0f / HP_LOSCALE); }

// 8 consecutive elements of a plane pair: hi at p, lo at p + lo_off (elements); both 16-byte aligned
__device__ __forceinline__ void load8(const uint16_t* p, int64_t lo_off, float* v) {
  const f16x8 h = *reinterpret_cast<const f16x8*>(p);
  const f16x8 l = *reinterpret_cast<const f16x8*>(p + lo_off);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = join1(h[e], l[e]);
}
__device__ __forceinline__ void store8(uint16_t* p, int64_t lo_off, const float* v) {
  f16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    _Float16 a, b;
    split1(v[e], a, b);
    h[e] = a;
    l[e] = b;
  }
  *reinterpret_cast<f16x8*>(p) = h;
  *reinterpret_cast<f16x8*>(p + lo_off) = l;
}
__device__ __forceinline__ float load1(const uint16_t* p, int64_t lo_off) {
  return join1(*reinterpret_cast<const _Float16*>(p), *reinterpret_cast<const _Float16*>(p + lo_off));
}

}  // namespace sdk_hp

// hp.hip launch helpers used by the forward schedule (sdk_api.hip)
int hp_seg_mean(sdk_ctx* ctx, const uint16_t* z, int64_t ldz, int64_t z_lo, int B, int T, int C, float* out, void* stream);
int hp_se_apply(sdk_ctx* ctx, const uint16_t* z, int64_t ldz, int64_t z_lo, const uint16_t* x, int64_t ldx, int64_t x_lo, const float* gate,
                uint16_t* out, int64_t ldo, int64_t o_lo, int B, int T, int C, void* stream);
int hp_asp_stats(sdk_ctx* ctx, const uint16_t* h, int64_t ldh, int64_t h_lo, int B, int T, int C, float* out, void* stream);
int hp_asp_pool(sdk_ctx* ctx, const float* logits, int64_t ldl, const uint16_t* h, int64_t ldh, int64_t h_lo, int B, int T, int C, float* pooled,
                void* stream);
