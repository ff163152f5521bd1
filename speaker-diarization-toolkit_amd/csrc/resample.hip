// Audio conversion to the backend's AudioProfile (gfx950): channel down-mix + rational polyphase FIR resampling,
// s16 in -> s16 out, integer arithmetic throughout (s16 x Q30 taps, int64 accumulation) so every device and the
// CPU oracle agree bit for bit.  Replaces the ffmpeg subprocess of the reference's backends
// (speaker_detection_backends/audio_profiles.py:70-100, speechmatics_backend.py:231-281).
//
//   y[n] = sat16((sum_k taps[(n M) mod L][k] * mono[floor(n M / L) + k - K/2 + 1] + 2^29) >> 30)
//
// One thread per output sample, 2048 outputs per workgroup.  The workgroup first down-mixes the stretch of input
// it needs into an int16 LDS window (so the channel loop runs once per input sample, not once per tap) next to the
// tap table (L x K int32, <= 64 KiB for every common rate pair); each lane then walks its own phase row against
// its own window offset.  Inputs whose table or window does not fit fall back to global-memory operands.
#include "common.hpp"

namespace {

constexpr int RS_NT = 256;
constexpr int RS_PER_WG = 2048;          // outputs per workgroup: amortises the table load, bounds the input window in LDS

__device__ __forceinline__ int mono_at(const int16_t* __restrict__ x, int64_t i, int C) {
  if (C == 1) return x[i];
  const int16_t* p = x + i * C;
  int s = C >> 1;
  for (int c = 0; c < C; ++c) s += p[c];
  return s >= 0 ? s / C : -((-s + C - 1) / C);             // floor division
}

// LDS layout: [tap table L*K int32 (when it fits)] [down-mixed input window of this workgroup, int16]
template <bool LDS_TAPS, bool LDS_WIN>
__global__ __launch_bounds__(RS_NT) void resample_kernel(const int16_t* __restrict__ x, int64_t n_in, int C,
                                                        const int32_t* __restrict__ taps, int L, int M, int K,
                                                        int16_t* __restrict__ y, int64_t n_out) {
  extern __shared__ int32_t stab[];
  const int32_t* tab = taps;
  if (LDS_TAPS) {
    for (int i = threadIdx.x; i < L * K; i += RS_NT) stab[i] = taps[i];
    tab = stab;
  }
  const int64_t first = (int64_t)blockIdx.x * RS_PER_WG;
  const int64_t last = min(first + RS_PER_WG, n_out);                 // exclusive
  const int64_t w_lo = (first * M) / L - (K / 2 - 1);                 // first input sample any output of this WG reads
  int16_t* win = reinterpret_cast<int16_t*>(stab + (LDS_TAPS ? L * K : 0));
  if (LDS_WIN) {
    // channel down-mix once per input sample instead of once per tap; zero outside the signal
    const int nwin = (int)(((last - 1) * M) / L - (K / 2 - 1) + K - w_lo);
    for (int j = threadIdx.x; j < nwin; j += RS_NT) {
      const int64_t idx = w_lo + j;
      win[j] = (idx >= 0 && idx < n_in) ? (int16_t)mono_at(x, idx, C) : (int16_t)0;
    }
  }
  if (LDS_TAPS || LDS_WIN) __syncthreads();
  for (int j = threadIdx.x; j < RS_PER_WG; j += RS_NT) {
    const int64_t n = first + j;
    if (n >= n_out) break;
    const int64_t pos = n * M;
    const int64_t i0 = pos / L;
    const int ph = (int)(pos - i0 * L);
    const int32_t* h = tab + ph * K;
    const int64_t ib = i0 - (K / 2 - 1);
    int64_t acc = 0;
    if (LDS_WIN) {
      const int16_t* w = win + (int)(ib - w_lo);
#pragma unroll 8
      for (int k = 0; k < K; ++k) acc += (int64_t)h[k] * w[k];
    } else {
      int k0 = 0, k1 = K;
      if (ib < 0) k0 = (int)(-ib);
      if (ib + K > n_in) k1 = (int)(n_in - ib);
      for (int k = k0; k < k1; ++k) acc += (int64_t)h[k] * mono_at(x, ib + k, C);
    }
    int64_t v = (acc + (1ll << 29)) >> 30;
    v = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
    y[n] = (int16_t)v;
  }
}

}  // namespace

extern "C" int64_t sdk_resample_out_len(int64_t n_in, int L, int M) {
  return (L > 0 && M > 0 && n_in >= 0) ? (n_in * L + M - 1) / M : -1;
}

extern "C" int sdk_resample_s16(sdk_ctx* ctx, const int16_t* x, int64_t n_in, int channels, const int32_t* taps, int L,
                                int M, int K, int16_t* y, int64_t n_out, void* stream) {
  SDK_REQUIRE(ctx && x && taps && y, "sdk_resample_s16: null argument");
  SDK_REQUIRE(n_in > 0 && channels >= 1 && channels <= 64, "sdk_resample_s16: n_in=%lld channels=%d", (long long)n_in, channels);
  SDK_REQUIRE(L >= 1 && M >= 1 && K >= 2 && (K & 1) == 0 && (int64_t)L * K <= (1 << 22), "sdk_resample_s16: bad filter shape L=%d M=%d K=%d", L, M, K);
  SDK_REQUIRE(n_out == sdk_resample_out_len(n_in, L, M), "sdk_resample_s16: n_out=%lld, expected ceil(n_in*L/M)=%lld", (long long)n_out,
              (long long)sdk_resample_out_len(n_in, L, M));
  SDK_REQUIRE(n_in < (1ll << 40), "sdk_resample_s16: input too long");
  ProfScope ps(ctx, stream, SDK_K_RESAMPLE, 2.0 * (double)n_out * K, 2.0 * (double)n_in * channels + 2.0 * (double)n_out);
  const size_t tab_bytes = (size_t)L * K * sizeof(int32_t);
  const int64_t nwg = (n_out + RS_PER_WG - 1) / RS_PER_WG;
  SDK_REQUIRE(nwg < (1ll << 31), "sdk_resample_s16: output too long");
  // input samples one workgroup touches: (RS_PER_WG - 1) * M / L + K, rounded up
  const size_t win_bytes = ((size_t)(((int64_t)(RS_PER_WG - 1) * M) / L + K + 2) * sizeof(int16_t) + 3) & ~(size_t)3;
  const bool lds_taps = tab_bytes <= 64 * 1024;
  const bool lds_win = win_bytes <= 64 * 1024;
  if (sdk_lds_optin(ctx, (const void*)resample_kernel<true, true>, 128 * 1024)) return 1;
  if (sdk_lds_optin(ctx, (const void*)resample_kernel<true, false>, 64 * 1024)) return 1;
  if (sdk_lds_optin(ctx, (const void*)resample_kernel<false, true>, 64 * 1024)) return 1;
  const dim3 grid((unsigned)nwg), block(RS_NT);
  hipStream_t st = (hipStream_t)stream;
  if (lds_taps && lds_win)
    hipLaunchKernelGGL((resample_kernel<true, true>), grid, block, tab_bytes + win_bytes, st, x, n_in, channels, taps, L, M, K, y, n_out);
  else if (lds_taps)
    hipLaunchKernelGGL((resample_kernel<true, false>), grid, block, tab_bytes, st, x, n_in, channels, taps, L, M, K, y, n_out);
  else if (lds_win)
    hipLaunchKernelGGL((resample_kernel<false, true>), grid, block, win_bytes, st, x, n_in, channels, taps, L, M, K, y, n_out);
  else
    hipLaunchKernelGGL((resample_kernel<false, false>), grid, block, 0, st, x, n_in, channels, taps, L, M, K, y, n_out);
  SDK_LAUNCH_CHECK();
  return 0;
}
