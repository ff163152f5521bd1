// k1: 80-bin log-mel filterbank features on gfx950.
//
// Input contract = the reference's AudioProfile (speaker_detection_backends/audio_profiles.py:25-29):
// 16 kHz mono s16le.  Algorithm (restated in oracle/fbank.py): centred zero-padded frames of 400
// samples, hop 160, Hamming window, 400-point DFT, power, 80 triangular mel filters, dB log with an
// 80 dB floor under the utterance peak, per-utterance mean normalisation, bf16 output.
//
// Kernel A (fbank_tile_kernel): one workgroup = 32 frames of one segment, 7 waves.  The windowed
// real DFT runs on the bf16 matrix pipe with hi+lo split operands (3 MFMAs per product, ~fp32
// accuracy), K folded 400 -> 201 by the real-input/symmetric-window identity:
// wave w owns frequency bins 32w..32w+31 and keeps the cos and sin
// accumulators of those bins in the SAME lane/register positions, so |X|^2 is formed in registers.
//   A operand: folded (x[n] +- x[400-n]) hi/lo bf16 sample images built once per tile in LDS, 432-byte
//              rows (27 sixteen-byte slots: the 32 frames of a ds_read_b128 never share a slot).
//   B operand: the DFT matrix with the window folded in, split hi/lo, pre-packed on the host in
//              fragment order so each lane fetches a fragment with one 16-byte L2-resident load,
//              prefetched one k-step ahead.
// Mel projection is a sparse (triangular) VALU dot product over the LDS power tile.
// Kernel B (fbank_norm_kernel): per segment floor / mean-normalise / bf16 store (HBM streaming).
#include <math.h>
#include <string.h>

#include <type_traits>

#include "common.hpp"
#include "hp.hpp"

namespace {

constexpr int NFFT = 400, HOP = 160, NMEL = 80, NBIN = 201;
constexpr int FT = 32;                  // frames per tile
constexpr int NW = 7;                   // waves per workgroup = bin blocks of 32 (224 >= 201)
constexpr int NSYM = 208;                // folded sample index n = 0..200 (x[n] +- x[400-n]), padded to 13 k-steps of 16
constexpr int KSTEPS = NSYM / 16;        // 13 MFMA k-steps (v_mfma_f32_32x32x16_bf16)
constexpr int TILE_SAMPLES = (FT - 1) * HOP + NFFT;   // 5360
constexpr int AROW = NSYM + 8;           // bf16 elements per frame row of the folded images (432 B = 27 slots: conflict-free)
constexpr int PW_STRIDE = NW * 32 + 1;  // 225
constexpr int MELW_MAX = 512;

struct FbankTables {
  // DFT matrix with the window folded in, split hi + lo bf16, in B-fragment order:
  // [bin block][cos|sin][k-step][hi|lo][lane][8]   (lane l: bin 32w + (l & 31), k = 16 ks + 8 (l >> 5) + j)
  uint16_t dft[NW][2][KSTEPS][2][64][8];
  int32_t mstart[NMEL];
  int32_t mlen[NMEL];
  int32_t moff[NMEL];
  float melw[MELW_MAX];
  // precise mode ("precision" 1): the same matrix split into fp16 hi + lo (22 significand bits; the folded int16 samples split exactly
  // there too), same fragment order - three fp16 MFMAs per product, ~2^-22 relative instead of ~2^-16
  uint16_t dft16[NW][2][KSTEPS][2][64][8];
};

// The windowed real DFT of 32 frames as a [32 x 208] x [208 x 2*224] product on the bf16 matrix pipe at
// ~fp32 accuracy: both operands are split x = hi + lo (bf16 each; int16 samples split EXACTLY) and
//     x . y  ~=  hi.hi + hi.lo + lo.hi          (relative error ~2^-16 per product, ~ -120 dB of the peak)
// which is 3 v_mfma_f32_32x32x16_bf16 per 16 k - 5x fewer matrix-pipe cycles than v_mfma_f32_32x32x2_f32.
// K is already folded 400 -> 201 by the real-input / symmetric-window identity:
//   Re X[f] =  sum_{n=0..200} c_n w[n] (x[n] + x[400-n]) cos(2 pi f n / 400),  c_0 = c_200 = 1/2
//   Im X[f] = -sum_{n=1..199}     w[n] (x[n] - x[400-n]) sin(2 pi f n / 400)         (400-n taken mod 400)
// The folded, split sample images are built once per tile in LDS (rows of 432 B), so an A fragment is one
// ds_read_b128; wave w owns bins 32w..32w+31 and keeps cos and sin accumulators in the same lane/register
// positions, so |X|^2 forms in registers.
template <bool F16>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void fbank_tile_kernel(const int16_t* __restrict__ pcm, int S, int T,
                                                            int tiles_per_seg, const FbankTables* __restrict__ tab,
                                                            float* __restrict__ L, unsigned long long* __restrict__ dbg,
                                                            const int32_t* __restrict__ starts, int64_t n_total) {
  __shared__ __attribute__((aligned(16))) char lds[4 * FT * AROW * 2 + TILE_SAMPLES * 4];
  // diagnostics only (tools/fbank_timeline.py): 100 MHz stamps of thread 0 of workgroups 0..255
  int nstamp = 0;
  auto stamp = [&]() { if (dbg && threadIdx.x == 0 && blockIdx.x < 256) dbg[blockIdx.x * 16 + nstamp++] = __builtin_amdgcn_s_memrealtime(); };
  stamp();
  typedef typename std::conditional<F16, _Float16, bf16_t>::type el_t;       // operand element: bf16 (default) or fp16 (precise mode)
  typedef el_t elx8 __attribute__((ext_vector_type(8)));
  typedef el_t elx2 __attribute__((ext_vector_type(2)));
  el_t* img = reinterpret_cast<el_t*>(lds);                           // [cos hi | cos lo | sin hi | sin lo][32][AROW]
  float* xs = reinterpret_cast<float*>(lds + 4 * FT * AROW * 2);      // raw samples of the tile
  float* pw = reinterpret_cast<float*>(lds);                          // power tile, overlays the images after the MFMAs
  static_assert(FT * PW_STRIDE * 4 <= 4 * FT * AROW * 2, "power tile must fit over the folded images");
  static_assert((MELW_MAX + 3 * NMEL) * 4 <= TILE_SAMPLES * 4, "mel tables must fit over the raw samples");
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.x / tiles_per_seg;
  const int t0 = (blockIdx.x - b * tiles_per_seg) * FT;
  const int64_t s0 = (int64_t)t0 * HOP - NFFT / 2;     // first sample of the tile (may be < 0)
  // window b of the batch: row b of a [B, S] array, or (sdk_fbank_windows) the S samples from starts[b] of ONE resident recording of n_total
  // samples - what lies past the recording's end reads as zero, exactly the zero padding a host-side cut would have written
  const int64_t w0 = starts ? (int64_t)starts[b] : (int64_t)b * S;
  const int16_t* seg = pcm + w0;
  const int Sv = starts ? (int)(n_total - w0 < S ? (n_total - w0 < 0 ? 0 : n_total - w0) : S) : S;     // samples of the window that exist

  {
    // every request of the thread first, then the conversions: rolled, this loop was load - wait - store twelve times over,
    // i.e. twelve HBM round trips in a row at the head of every workgroup
    constexpr int NLD = (TILE_SAMPLES + NW * 64 - 1) / (NW * 64);
    int16_t raw[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int64_t g = s0 + tid + i * (NW * 64);
      raw[i] = seg[g < 0 ? 0 : (g < Sv ? g : (Sv > 0 ? Sv - 1 : 0))];          // clamped address, value masked below
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int q = tid + i * (NW * 64);
      const int64_t g = s0 + q;
      if (q < TILE_SAMPLES) xs[q] = (g >= 0 && g < Sv) ? (float)raw[i] * (1.0f / 32768.0f) : 0.f;
    }
  }
  __syncthreads();
  stamp();
  // folded, split sample images: two consecutive n per thread-iteration -> 4-byte LDS writes
  for (int e = tid; e < FT * (NSYM / 2); e += NW * 64) {
    const int i = e / (NSYM / 2), n = (e - i * (NSYM / 2)) * 2;
    float ec[2] = {0.f, 0.f}, es[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int nn = n + q;
      if (nn <= NFFT / 2) {
        const float xa = xs[i * HOP + nn], xb = xs[i * HOP + (nn == 0 ? 0 : NFFT - nn)];
        ec[q] = xa + xb;
        es[q] = xa - xb;
      }
    }
    const el_t ch0 = (el_t)ec[0], ch1 = (el_t)ec[1], sh0 = (el_t)es[0], sh1 = (el_t)es[1];     // RNE both ways
    auto pk = [](float a, float b) { elx2 t; t[0] = (el_t)a; t[1] = (el_t)b; return __builtin_bit_cast(uint32_t, t); };
    uint32_t* dst = reinterpret_cast<uint32_t*>(img + i * AROW + n);
    dst[(0 * FT * AROW) / 2] = pk((float)ch0, (float)ch1);
    dst[(1 * FT * AROW) / 2] = pk(ec[0] - (float)ch0, ec[1] - (float)ch1);
    dst[(2 * FT * AROW) / 2] = pk((float)sh0, (float)sh1);
    dst[(3 * FT * AROW) / 2] = pk(es[0] - (float)sh0, es[1] - (float)sh1);
  }
  __syncthreads();
  stamp();

  f32x16 are, aim;
#pragma unroll
  for (int r = 0; r < 16; ++r) { are[r] = 0.f; aim[r] = 0.f; }
  const int fi = lane & 31, kk = lane >> 5;
  const el_t* arow = img + fi * AROW + kk * 8;
  // table fragments through buffer loads: ONE per-lane offset register (lane * 16), everything else - wave, part, k-step, hi/lo -
  // in the scalar offset (the 53-KiB span of a wave's slice is far beyond a global load's immediate range, and flat addressing
  // kept several 64-bit base pointers alive through the MFMA loop)
  const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(F16 ? &tab->dft16[0][0][0][0][0][0] : &tab->dft[0][0][0][0][0][0]), 0, (int)sizeof(tab->dft), 0x00020000);
  const int wu = __builtin_amdgcn_readfirstlane(w);
  // DFT-matrix fragments (cos hi, cos lo, sin hi, sin lo) travel TWO k-steps ahead of their MFMAs through a ring of three
  // register sets: one step (6 MFMAs = 192 cycles) is far less than the L2 round trip they come from - with a single step of
  // look-ahead every k-step ended in a wait for the table.  The sample fragments (LDS) stay one step ahead.  The loop is
  // unrolled so that the ring indices are static; the scheduling fences keep hipcc from hoisting all 52 table loads at once.
  elx8 bt[3][4];
  elx8 ac[2][4];
  auto load_b = [&](int ks, elx8* dst) {
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      dst[v] = __builtin_bit_cast(elx8, __builtin_amdgcn_raw_buffer_load_b128(trs, lane * 16, ((((wu * 2 + 0) * KSTEPS + ks) * 2 + v) * 64) * 16, 0));
      dst[2 + v] = __builtin_bit_cast(elx8, __builtin_amdgcn_raw_buffer_load_b128(trs, lane * 16, ((((wu * 2 + 1) * KSTEPS + ks) * 2 + v) * 64) * 16, 0));
    }
  };
  auto load_a = [&](int ks, elx8* dst) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const elx8*>(arow + (q * FT) * AROW + ks * 16);
  };
  auto mma = [](const elx8& a, const elx8& b, const f32x16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  };
  load_b(0, bt[0]);
  load_b(1, bt[1]);
  load_a(0, ac[0]);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 2 < KSTEPS) load_b(ks + 2, bt[(ks + 2) % 3]);
    if (ks + 1 < KSTEPS) load_a(ks + 1, ac[(ks + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    const elx8* bc = bt[ks % 3];
    const elx8* aa = ac[ks & 1];
    are = mma(aa[1], bc[0], are);     // small terms first
    aim = mma(aa[3], bc[2], aim);
    are = mma(aa[0], bc[1], are);
    aim = mma(aa[2], bc[3], aim);
    are = mma(aa[0], bc[0], are);
    aim = mma(aa[2], bc[2], aim);
  }
  __syncthreads();                                       // every wave is done with the images: pw may overlay them
  // power tile -> LDS [frame][bin]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int frame = (r & 3) + 8 * (r >> 2) + 4 * kk;
    pw[frame * PW_STRIDE + w * 32 + fi] = are[r] * are[r] + aim[r] * aim[r];
  }
  // the triangular mel filters (start / length / offset per filter + 512 weights) move into the LDS the raw samples
  // no longer need: the dot products below then read only LDS instead of walking global memory per term
  float* melw = xs;
  int* mtab = reinterpret_cast<int*>(xs + MELW_MAX);
  {
    constexpr int NLD = (MELW_MAX + NW * 64 - 1) / (NW * 64);  // all requests first (the rolled loops waited for each load in turn)
    float wv[NLD];
    int tv[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < NLD; ++i) wv[i] = tab->melw[min(tid + i * (NW * 64), MELW_MAX - 1)];
    if (tid < NMEL) { tv[0] = tab->mstart[tid]; tv[1] = tab->mlen[tid]; tv[2] = tab->moff[tid]; }
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      if (tid + i * (NW * 64) < MELW_MAX) melw[tid + i * (NW * 64)] = wv[i];
    if (tid < NMEL) { mtab[tid] = tv[0]; mtab[NMEL + tid] = tv[1]; mtab[2 * NMEL + tid] = tv[2]; }
  }
  __syncthreads();
  stamp();
  for (int o = tid; o < FT * NMEL; o += NW * 64) {
    const int frame = o / NMEL, m = o - frame * NMEL;
    const int t = t0 + frame;
    if (t < T) {
      const int st = mtab[m], ln = mtab[NMEL + m], of = mtab[2 * NMEL + m];
      float acc = 0.f;
      for (int i = 0; i < ln; ++i) acc += pw[frame * PW_STRIDE + st + i] * melw[of + i];
      // 10 log10(x) on the hardware log2 unit (v_log_f32, ~1 ulp of log2 - far below the bf16 rounding of the feature): libm's
      // log10f was ~half of this phase (2560 calls per tile)
      if constexpr (F16) L[((int64_t)b * T + t) * NMEL + m] = 10.0f * log10f(fmaxf(acc, 1e-10f));      // precise mode: libm (the hardware log2 is ~1 ulp of log2 = 6e-6 dB)
      else L[((int64_t)b * T + t) * NMEL + m] = 3.0102999566398120f * __log2f(fmaxf(acc, 1e-10f));
    }
  }
  stamp();
}

template <bool HP, bool F16 = false>   // HP: fp16 hi | lo planes (precision 1); F16: one fp16 plane (precision 2); neither: bf16
__global__ __launch_bounds__(256) void fbank_norm_kernel(const float* __restrict__ L, int T, bf16_t* __restrict__ feats,
                                                        int ldf) {
  __shared__ float red[256];
  __shared__ float mean[NMEL];
  const int tid = threadIdx.x;
  const float* Ls = L + (int64_t)blockIdx.x * T * NMEL;
  float mx = -INFINITY;
  for (int i = tid; i < T * NMEL; i += 256) mx = fmaxf(mx, Ls[i]);
  red[tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
    __syncthreads();
  }
  const float flo = red[0] - 80.0f;
  __syncthreads();
  // mean over frames per mel bin: thread = (mel m, frame group g of 3)
  const int m = tid % NMEL, g = tid / NMEL;
  float s = 0.f;
  if (g < 3)
    for (int t = g; t < T; t += 3) s += fmaxf(Ls[t * NMEL + m], flo);
  red[tid] = s;
  __syncthreads();
  if (tid < NMEL) mean[tid] = (red[tid] + red[tid + NMEL] + red[tid + 2 * NMEL]) / (float)T;
  __syncthreads();
  const int c8n = (HP ? ldf >> 1 : ldf) >> 3;
  bf16_t* out = feats + (int64_t)blockIdx.x * T * ldf;
  for (int i = tid; i < T * c8n; i += 256) {
    const int t = i / c8n, c8 = i - t * c8n;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c8 * 8 + e;
      f[e] = c < NMEL ? fmaxf(Ls[t * NMEL + c], flo) - mean[c] : 0.f;
    }
    if constexpr (HP) sdk_hp::store8(reinterpret_cast<uint16_t*>(out) + (int64_t)t * ldf + c8 * 8, ldf >> 1, f);   // planes: hi | lo, ldf / 2 columns each
    else *reinterpret_cast<u32x4*>(out + (int64_t)t * ldf + c8 * 8) = pack8t<F16>(f);
  }
}

// The same per-segment normalisation with the segment's [T x 80] log-mel tile held in LDS: ONE sweep over HBM (16-byte loads, eight in
// flight per thread), the maximum / the means / the output all come from LDS.  The streaming form above walks global memory three
// times with a load - wait - use loop (63 dependent round trips per sweep at T = 201); it stays for segments whose tile exceeds LDS.
// Arithmetic (and its order per mel bin: frames g, g + 3, ... per thread, three partial sums) is the streaming form's, so the
// features are bit-identical.
constexpr int NORM_LDS_MAX_T = 480;                      // 480 x 80 x 4 B = 150 KiB
template <bool HP, bool F16 = false>   // HP: fp16 hi | lo planes (precision 1); F16: one fp16 plane (precision 2); neither: bf16
__global__ __launch_bounds__(256) void fbank_norm_lds_kernel(const float* __restrict__ L, int T, bf16_t* __restrict__ feats,
                                                            int ldf) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ float red[256];
  __shared__ float mean[NMEL];
  const int tid = threadIdx.x;
  const int n4 = T * (NMEL / 4);
  const f32x4* Ls = reinterpret_cast<const f32x4*>(L + (int64_t)blockIdx.x * T * NMEL);
  float mx = -INFINITY;
  constexpr int UN = 8;
  for (int i0 = tid; i0 < n4; i0 += 256 * UN) {
    f32x4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = Ls[min(i0 + 256 * u, n4 - 1)];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = i0 + 256 * u;
      if (i < n4) {
        reinterpret_cast<f32x4*>(tile)[i] = v[u];
        mx = fmaxf(fmaxf(mx, fmaxf(v[u][0], v[u][1])), fmaxf(v[u][2], v[u][3]));
      }
    }
  }
  red[tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
    __syncthreads();
  }
  const float flo = red[0] - 80.0f;
  __syncthreads();
  const int m = tid % NMEL, g = tid / NMEL;
  float s = 0.f;
  if (g < 3)
    for (int t = g; t < T; t += 3) s += fmaxf(tile[t * NMEL + m], flo);
  red[tid] = s;
  __syncthreads();
  if (tid < NMEL) mean[tid] = (red[tid] + red[tid + NMEL] + red[tid + 2 * NMEL]) / (float)T;
  __syncthreads();
  const int c8n = (HP ? ldf >> 1 : ldf) >> 3;
  bf16_t* out = feats + (int64_t)blockIdx.x * T * ldf;
  for (int i = tid; i < T * c8n; i += 256) {
    const int t = i / c8n, c8 = i - t * c8n;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c8 * 8 + e;
      f[e] = c < NMEL ? fmaxf(tile[t * NMEL + c], flo) - mean[c] : 0.f;
    }
    if constexpr (HP) sdk_hp::store8(reinterpret_cast<uint16_t*>(out) + (int64_t)t * ldf + c8 * 8, ldf >> 1, f);   // planes: hi | lo, ldf / 2 columns each
    else *reinterpret_cast<u32x4*>(out + (int64_t)t * ldf + c8 * 8) = pack8t<F16>(f);
  }
}

}  // namespace

extern "C" size_t sdk_fbank_tables_bytes(void) { return sizeof(FbankTables); }

extern "C" int sdk_fbank_tables_fill(void* host_dst, size_t bytes) {
  SDK_REQUIRE(host_dst && bytes >= sizeof(FbankTables), "sdk_fbank_tables_fill: buffer too small (%zu < %zu)", bytes,
              sizeof(FbankTables));
  FbankTables* t = (FbankTables*)host_dst;
  memset(t, 0, sizeof(FbankTables));
  const double PI = 3.14159265358979323846;
  auto bf16_bits = [](float v) -> uint16_t {               // round-to-nearest-even, finite inputs
    uint32_t u;
    memcpy(&u, &v, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
  };
  auto f16_bits = [](double v) -> uint16_t {                 // fp16 bits of v, round-to-nearest-even, subnormals kept (|v| <= 2 here)
    const double a = fabs(v);
    uint16_t out;
    if (a == 0.0) out = 0;
    else {
      int e;
      (void)frexp(a, &e);                                    // a = m * 2^e, 0.5 <= m < 1
      int ex = e - 1;                                        // a = 1.xxx * 2^ex
      if (ex < -14) ex = -14;                                // subnormal range: fixed quantum 2^-24
      const double q = ldexp(1.0, ex - 10);                  // spacing of fp16 at this exponent
      double n = nearbyint(a / q);                           // default rounding mode: to nearest even
      if (n >= 2048.0) { n /= 2.0; ex += 1; }
      const int mant = (int)n;
      out = mant < 1024 ? (uint16_t)mant : (uint16_t)(((ex + 15) << 10) | (mant - 1024));
    }
    return (uint16_t)(out | (v < 0 ? 0x8000u : 0u));
  };
  auto f16_val = [](uint16_t h) -> double {
    const int e = (h >> 10) & 31, m = h & 1023;
    const double a = e ? ldexp(1.0 + m / 1024.0, e - 15) : ldexp((double)m, -24);
    return (h & 0x8000u) ? -a : a;
  };
  auto bf16_val = [](uint16_t b) -> float {
    const uint32_t u = (uint32_t)b << 16;
    float v;
    memcpy(&v, &u, 4);
    return v;
  };
  for (int w = 0; w < NW; ++w)
    for (int ks = 0; ks < KSTEPS; ++ks)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int n = ks * 16 + 8 * (l >> 5) + j;
          const int f = w * 32 + (l & 31);
          double vc = 0.0, vs = 0.0;
          if (f < NBIN && n <= NFFT / 2) {
            const double win = 0.54 - 0.46 * cos(2.0 * PI * n / NFFT);
            const double ang = 2.0 * PI * (double)((n * f) % NFFT) / NFFT;
            const double half = (n == 0 || n == NFFT / 2) ? 0.5 : 1.0;   // these samples are their own mirror
            vc = half * win * cos(ang);
            vs = -win * sin(ang);
          }
          const uint16_t ch = bf16_bits((float)vc), sh = bf16_bits((float)vs);
          t->dft[w][0][ks][0][l][j] = ch;
          t->dft[w][0][ks][1][l][j] = bf16_bits((float)(vc - (double)bf16_val(ch)));
          t->dft[w][1][ks][0][l][j] = sh;
          t->dft[w][1][ks][1][l][j] = bf16_bits((float)(vs - (double)bf16_val(sh)));
          const uint16_t ch16 = f16_bits(vc), sh16 = f16_bits(vs);
          t->dft16[w][0][ks][0][l][j] = ch16;
          t->dft16[w][0][ks][1][l][j] = f16_bits(vc - f16_val(ch16));
          t->dft16[w][1][ks][0][l][j] = sh16;
          t->dft16[w][1][ks][1][l][j] = f16_bits(vs - f16_val(sh16));
        }
  // HTK-mel triangular filters, 0..8000 Hz, unit peak (oracle/fbank.py: mel_matrix)
  auto hz2mel = [](double f) { return 2595.0 * log10(1.0 + f / 700.0); };
  auto mel2hz = [](double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); };
  double pts[NMEL + 2];
  const double mlo = hz2mel(0.0), mhi = hz2mel(8000.0);
  for (int i = 0; i < NMEL + 2; ++i) pts[i] = mel2hz(mlo + (mhi - mlo) * i / (NMEL + 1));
  int off = 0;
  for (int m = 0; m < NMEL; ++m) {
    const double lo = pts[m], ce = pts[m + 1], hi = pts[m + 2];
    int first = -1, last = -1;
    float wv[NBIN];
    for (int f = 0; f < NBIN; ++f) {
      const double fr = 8000.0 * f / (NBIN - 1);
      const double up = (fr - lo) / (ce - lo), dn = (hi - fr) / (hi - ce);
      const double v = fmax(0.0, fmin(up, dn));
      wv[f] = (float)v;
      if (v > 0.0) {
        if (first < 0) first = f;
        last = f;
      }
    }
    t->mstart[m] = first < 0 ? 0 : first;
    t->mlen[m] = first < 0 ? 0 : last - first + 1;
    t->moff[m] = off;
    SDK_REQUIRE(off + t->mlen[m] <= MELW_MAX, "sdk_fbank_tables_fill: mel table overflow");
    for (int i = 0; i < t->mlen[m]; ++i) t->melw[off + i] = wv[first + i];
    off += t->mlen[m];
  }
  return 0;
}

extern "C" size_t sdk_fbank_workspace_bytes(int B, int S) {
  if (B <= 0 || S <= 0) return 0;
  return (size_t)B * (size_t)(1 + S / HOP) * NMEL * sizeof(float);
}

static int fbank_launch(sdk_ctx* ctx, const int16_t* pcm, const int32_t* starts, int64_t n_total, int B, int S, const void* tabs, uint16_t* feats,
                        int ldf, void* ws, size_t ws_bytes, int precision, void* stream, const char* who) {
  SDK_REQUIRE(ctx && pcm && tabs && feats && ws, "%s: null argument", who);
  SDK_REQUIRE(B > 0 && S > 0, "%s: empty batch (B=%d S=%d)", who, B, S);
  SDK_REQUIRE(precision >= 0 && precision <= 2, "%s: precision=%d must be 0 (bf16), 1 (fp16 hi | lo planes) or 2 (one fp16 plane)", who, precision);
  const bool hp = precision == 1;
  if (hp) SDK_REQUIRE((ldf >> 1) >= NMEL && ldf % 16 == 0, "%s: precise mode writes planes: ldf=%d must be >= 160 and a multiple of 16", who, ldf);
  SDK_REQUIRE(ldf >= NMEL && ldf % 8 == 0, "%s: ldf=%d must be >= 80 and a multiple of 8", who, ldf);
  SDK_REQUIRE(ws_bytes >= sdk_fbank_workspace_bytes(B, S), "%s: workspace too small", who);
  SDK_REQUIRE(((uintptr_t)feats % 16) == 0 && ((uintptr_t)tabs % 16) == 0, "%s: feats/tabs must be 16-byte aligned", who);
  const int T = 1 + S / HOP;
  const int tps = ceil_div(T, FT);
  SDK_REQUIRE((int64_t)B * tps < (1ll << 31), "%s: batch too large for one launch", who);
  {
  ProfScope ps(ctx, stream, SDK_K_FBANK_TILE, 3 * 2.0 * B * T * (double)NSYM * 2 * (NW * 32) + 2.0 * B * T * NBIN * NMEL, 2.0 * B * S + 4.0 * B * T * NMEL);
  hipLaunchKernelGGL(hp ? fbank_tile_kernel<true> : fbank_tile_kernel<false>, dim3(B * tps), dim3(NW * 64), 0, (hipStream_t)stream, pcm, S, T, tps,
                     (const FbankTables*)tabs, (float*)ws, (unsigned long long*)ctx->dbg_ptr, starts, n_total);
  }
  SDK_LAUNCH_CHECK();
  ProfScope ps2(ctx, stream, SDK_K_FBANK_NORM, 3.0 * B * T * NMEL, 4.0 * B * T * NMEL + 2.0 * B * T * ldf);
  const bool f16 = precision == 2;                       // one fp16 plane, the default mode's [B*T, ldf] layout
  void (*norm_lds)(const float*, int, bf16_t*, int) = hp ? fbank_norm_lds_kernel<true> : f16 ? fbank_norm_lds_kernel<false, true> : fbank_norm_lds_kernel<false>;
  void (*norm)(const float*, int, bf16_t*, int) = hp ? fbank_norm_kernel<true> : f16 ? fbank_norm_kernel<false, true> : fbank_norm_kernel<false>;
  if (T <= NORM_LDS_MAX_T) {
    const int lds = T * NMEL * 4;
    if (sdk_lds_optin(ctx, (const void*)norm_lds, NORM_LDS_MAX_T * NMEL * 4)) return 1;   // (opt-in is per function: the maximum)
    hipLaunchKernelGGL(norm_lds, dim3(B), dim3(256), lds, (hipStream_t)stream, (const float*)ws, T, (bf16_t*)feats, ldf);
  } else {
    hipLaunchKernelGGL(norm, dim3(B), dim3(256), 0, (hipStream_t)stream, (const float*)ws, T, (bf16_t*)feats, ldf);
  }
  SDK_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdk_fbank(sdk_ctx* ctx, const int16_t* pcm, int B, int S, const void* tabs, uint16_t* feats, int ldf,
                         void* ws, size_t ws_bytes, void* stream) {
  SDK_REQUIRE(ctx, "sdk_fbank: null context");
  return fbank_launch(ctx, pcm, nullptr, 0, B, S, tabs, feats, ldf, ws, ws_bytes, ctx->precision, stream, "sdk_fbank");
}

// The same with the output format as an ARGUMENT instead of the context's "precision" option (round 5): hosts that run several numerical
// contracts on one device - two engines, two threads - share no mutable state through the library this way.
extern "C" int sdk_fbank_fmt(sdk_ctx* ctx, const int16_t* pcm, int B, int S, const void* tabs, uint16_t* feats, int ldf,
                             void* ws, size_t ws_bytes, int precision, void* stream) {
  return fbank_launch(ctx, pcm, nullptr, 0, B, S, tabs, feats, ldf, ws, ws_bytes, precision, stream, "sdk_fbank_fmt");
}

// Windows cut ON THE DEVICE: the recording is resident once ([n_samples] s16) and window b is the S samples from starts[b] (device int32 table).
// The host never materialises the overlapping [B, S] windows (2x the samples at hop 1 s / window 2 s) and they never cross PCIe.  The START of
// every window must lie inside the recording (checked by the caller: starts is device memory); samples past its end read as zero.
extern "C" int sdk_fbank_windows(sdk_ctx* ctx, const int16_t* samples, int64_t n_samples, const int32_t* starts, int B, int S, const void* tabs,
                                 uint16_t* feats, int ldf, void* ws, size_t ws_bytes, void* stream) {
  SDK_REQUIRE(starts, "sdk_fbank_windows: null start table");
  SDK_REQUIRE(n_samples > 0 && n_samples < (1ll << 31), "sdk_fbank_windows: n_samples=%lld must be in [1, 2^31)", (long long)n_samples);
  SDK_REQUIRE(ctx, "sdk_fbank_windows: null context");
  return fbank_launch(ctx, samples, starts, n_samples, B, S, tabs, feats, ldf, ws, ws_bytes, ctx->precision, stream, "sdk_fbank_windows");
}

extern "C" int sdk_fbank_windows_fmt(sdk_ctx* ctx, const int16_t* samples, int64_t n_samples, const int32_t* starts, int B, int S, const void* tabs,
                                     uint16_t* feats, int ldf, void* ws, size_t ws_bytes, int precision, void* stream) {
  SDK_REQUIRE(starts, "sdk_fbank_windows_fmt: null start table");
  SDK_REQUIRE(n_samples > 0 && n_samples < (1ll << 31), "sdk_fbank_windows_fmt: n_samples=%lld must be in [1, 2^31)", (long long)n_samples);
  return fbank_launch(ctx, samples, starts, n_samples, B, S, tabs, feats, ldf, ws, ws_bytes, precision, stream, "sdk_fbank_windows_fmt");
}
