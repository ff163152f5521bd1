// Shared device/host helpers for libsdk_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#include "../../include/sdk_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Per-launch HIP-event timing, recorded ON THE STREAM THE KERNEL IS LAUNCHED ON, grouped by kernel
// family.  Off by default (no events are created); bench.py switches it on for its roofline pass.
struct sdk_prof_rec {
  int family;
  hipEvent_t a, b;
  double flops, bytes;
};
struct sdk_ctx {
  int device;
  int num_cu;
  hipDeviceProp_t prop;
  bool prof_on = false;
  bool no_chain_fusion = false;   // A/B + test knob: run the Res2Net chain as separate conv_gemm launches
  bool no_chain_two_per_cu = false;   // A/B + test knob: the Res2Net chain as 8-wave workgroups, one segment per CU ("res2net_two_per_cu" 0)
  bool no_chain_packed = false;   // A/B + test knob: ignore the fragment-ordered copies of the Res2Net chain weights (EL_CHAINPACK)
  bool no_asp_packed = false;     // A/B + test knob: ignore the fragment-ordered copy of the ASP logit weights (EL_ASP_W2PACK)
  bool no_asp_seg = false;        // A/B + test knob: ASP by (segment, 128-channel) workgroups instead of one per segment
  bool no_h_kblocked = false;     // A/B + test knob: keep the MFA output h row-major (default: K-blocked [C / 64][M][64] where the per-segment ASP reads it)
  int precision = 0;              // 0: bf16 operands (default); 1: fp16 hi+lo planes, 3 MFMAs per product ("precision": hp.hip)
  int aff_fast = 1;               // k = 1 affinity: 1 = row/column-maxima kernel (affinity_rowcol.hip), 0 = general sorted-list kernel
  void* dbg_ptr = nullptr;         // diagnostics only: device buffer for the affinity kernel's time stamps (sdk_debug_set_ptr "stamps")
  void* gemm_clk_ptr = nullptr;    // diagnostics only: [4096][2] uint64 {shader cycles, 100 MHz ticks} of conv_gemm256_kernel ("gemm_clock")
  void* gemm_stamps_ptr = nullptr; // diagnostics only: [4096] uint64 phase stamps of conv_gemm256 workgroup 0 ("gemm_stamps"; tools/gemm_timeline.py)
  int hp_gemm_variant = 0;        // A/B + test knob: 1 = the precise mode's GEMM always as the 128^2 register-staged kernel
  int matvec_variant = 0;         // A/B + test knob: 1 = round 1's affinity_matvec_kernel (one 32-row block per wave, a barrier per tile)
  int aff_boundary_pen = 0;       // k = 1 coarse pass: cost of a group boundary inside a workgroup's range, in stages (affinity_rowcol.hip Geom.pen; 0 = equal unit counts)
  int chol_pivot_rtol_ppb = 1000; // sdk_chol_inverse: a pivot <= this fraction (in 1e-9) of its diagonal entry sets the sticky not_spd flag (default 1e-6: cond(Y) > ~1e3)
  int chol_shift_ppb = 0;         // sdk_chol_inverse: shifted CholeskyQR, G + s I with s = this fraction (in 1e-9) of the mean diagonal entry (0 = off)
  int aff_variant = 0;            // A/B knob: coarse-pass plan of the row/column kernel (0 = cost model, 7 = range plan, 8 / 12 / 13 = block plan; affinity_rowcol.hip)
  std::vector<const void*> lds_optin;   // kernels of THIS context's device already opted in to > 64 KiB dynamic LDS
  std::vector<sdk_prof_rec> prof;
};

// > 64 KiB of dynamic LDS needs a per-function, per-device opt-in: tracked per context (one context = one device),
// with the context's device made current first, so a second Engine on another GPU of the same process gets its own.
int sdk_lds_optin(sdk_ctx* ctx, const void* func, int bytes);

// pool_se.hip: sdk_asp_fused with the optional fragment-ordered copy of w2 (internal; sdk_ecapa_forward)
int asp_fused_launch(sdk_ctx* ctx, const uint16_t* ah, int64_t ldah, const uint16_t* w2, const uint16_t* w2p, const float* b2,
                     const uint16_t* h, int64_t ldh, int B, int T, int C, int A, float* pooled, void* stream, bool kblocked = false, bool f16 = false);

// pool_se.hip: the sweeps with the 2-byte element format as an argument (f16: fp16 instead of bf16), for sdk_ecapa_forward
int se_gate_residual_impl(sdk_ctx* ctx, const uint16_t* z, int64_t ldz, const uint16_t* x, int64_t ldx, const float* w1t, const float* b1, const float* w2t,
                          const float* b2, uint16_t* out, int64_t ldo, int B, int T, int C, int Cse, const float* mean_in, void* ws, size_t ws_bytes,
                          void* stream, bool f16);
int asp_stats_impl(sdk_ctx* ctx, const uint16_t* h, int64_t ldh, int B, int T, int C, float* out_ctx, void* stream, bool f16);
int asp_pool_impl(sdk_ctx* ctx, const float* logits, int64_t ldl, const uint16_t* h, int64_t ldh, int B, int T, int C, float* pooled, void* stream, bool f16);

// res2net.hip: sdk_res2net_chain with the optional fragment-ordered weight copies (internal; sdk_ecapa_forward)
int res2net_chain_launch(sdk_ctx* ctx, const uint16_t* U, int64_t ldu, uint16_t* R, int64_t ldr, const uint16_t* const* W,
                         const uint16_t* const* Wpk, const float* const* bias, const float* const* scale, const float* const* shift,
                         int nconv, int B, int T, int dil, void* stream, bool f16 = false);

struct ProfScope {   // brackets one kernel launch with two events when profiling is enabled
  sdk_ctx* c; hipStream_t s; size_t slot; bool on;
  ProfScope(sdk_ctx* ctx, void* stream, int family, double flops, double bytes) : c(ctx), s((hipStream_t)stream), slot(0), on(ctx && ctx->prof_on) {
    if (!on) return;
    sdk_prof_rec r{family, nullptr, nullptr, flops, bytes};
    (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, s);
    slot = c->prof.size();
    c->prof.push_back(r);
  }
  ~ProfScope() { if (on) (void)hipEventRecord(c->prof[slot].b, s); }
};

void sdk_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define SDK_HIP_OK(expr)                                                              \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      sdk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

#define SDK_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      sdk_set_error(__VA_ARGS__);         \
      return 2;                           \
    }                                     \
  } while (0)

#define SDK_LAUNCH_CHECK()                                                         \
  do {                                                                             \
    hipError_t _e = hipGetLastError();                                             \
    if (_e != hipSuccess) {                                                        \
      sdk_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f32_to_bf16(float v) { return (bf16_t)v; }  // RNE (v_cvt_pk_bf16_f32)

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t lo16) { return __uint_as_float(lo16 << 16); }

// 8 bf16 packed in a 16-byte vector -> 8 floats
__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2 t;
  t[0] = (bf16_t)a;
  t[1] = (bf16_t)b;
  return __builtin_bit_cast(uint32_t, t);
}
// the same for a 2-byte storage type chosen at compile time: bf16 (default contract) or fp16 (the single-plane fp16 contract, round 5); fp16 values
// are clamped to the format's finite range first (a saturated activation, never an inf that turns the next layer into NaN)
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <bool F16>
__device__ __forceinline__ uint32_t pack2t(float a, float b) {
  if constexpr (F16) {
    f16x2_t t;
    t[0] = (_Float16)fminf(fmaxf(a, -65504.f), 65504.f);
    t[1] = (_Float16)fminf(fmaxf(b, -65504.f), 65504.f);
    return __builtin_bit_cast(uint32_t, t);
  } else {
    return pack2(a, b);
  }
}
template <bool F16>
__device__ __forceinline__ void unpack8t(const u32x4& v, float* f) {
  if constexpr (F16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t w = v[i];                               // (a bit_cast applied to the vector ELEMENT expression reads element 0 every time)
      const f16x2_t t = __builtin_bit_cast(f16x2_t, w);
      f[2 * i] = (float)t[0];
      f[2 * i + 1] = (float)t[1];
    }
  } else {
    unpack8(v, f);
  }
}

__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2(f[2 * i], f[2 * i + 1]);
  return v;
}
template <bool F16>
__device__ __forceinline__ u32x4 pack8t(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2t<F16>(f[2 * i], f[2 * i + 1]);
  return v;
}
// one stored 2-byte element -> fp32 (the element type of the pointers stays bf16_t in every signature: 16 bits either way)
template <bool F16>
__device__ __forceinline__ float load1t(const bf16_t* q) {
  if constexpr (F16) return (float)*reinterpret_cast<const _Float16*>(q);
  else return bf16_to_f32(*q);
}
// the dense 16-bit MFMAs on fragments held as bf16x8 registers (bit patterns): bf16 or fp16 arithmetic, fp32 accumulate
template <bool F16>
__device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL
// store of the wave (vmcnt(0): CDNA4 counts stores), i.e. a full HBM round trip per call in a store-heavy
// epilogue; where no global data is handed between waves this is the barrier to use.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// reflect (no edge repeat) t into [0, T); valid for -T < t < 2T-1
__device__ __forceinline__ int reflect_idx(int t, int T) {
  t = t < 0 ? -t : t;
  return t >= T ? 2 * (T - 1) - t : t;
}

// XCD-aware bijective block remap (8 XCDs, blocks dealt round-robin): blocks that share an XCD
// get a contiguous run of logical tile ids, so neighbouring tiles hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}
