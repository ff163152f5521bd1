// k1: 80-bin log-mel filterbank features on gfx950.
//
// Input contract = the reference's AudioProfile (speaker_detection_backends/audio_profiles.py:25-29):
// 16 kHz mono s16le.  Algorithm (restated in oracle/fbank.py): centred zero-padded frames of 400
// samples, hop 160, Hamming window, 400-point DFT, power, 80 triangular mel filters, dB log with an
// 80 dB floor under the utterance peak, per-utterance mean normalisation, bf16 output.
//
// Kernel A (fbank_tile_kernel): one workgroup = 32 frames of one segment, 7 waves.  The windowed
// real DFT is a [32 x 400] x [400 x 2*224] product on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32; K folded 400 -> 201 by the real-input/symmetric-window identity):
// wave w owns frequency bins 32w..32w+31 and keeps the cos and sin
// accumulators of those bins in the SAME lane/register positions, so |X|^2 is formed in registers.
//   A operand: raw samples from an LDS image skewed by one word per hop (row stride 161 words:
//              the 32 frames of a wave-instruction hit 32 different banks instead of one).
//   B operand: the DFT matrix with the window folded in, pre-packed on the host in fragment order
//              ([wave][cos|sin][k-group][lane][4]) so each lane fetches four k-steps with one
//              16-byte L2-resident load, prefetched one group ahead.
// Mel projection is a sparse (triangular) VALU dot product over the LDS power tile.
// Kernel B (fbank_norm_kernel): per segment floor / mean-normalise / bf16 store (HBM streaming).
#include <math.h>
#include <string.h>

#include "common.hpp"

namespace {

constexpr int NFFT = 400, HOP = 160, NMEL = 80, NBIN = 201;
constexpr int FT = 32;                  // frames per tile
constexpr int NW = 7;                   // waves per workgroup = bin blocks of 32 (224 >= 201)
constexpr int NSYM = 208;                // folded sample index n = 0..200 (x[n] +- x[400-n]), padded to a multiple of 8
constexpr int KG = NSYM / 2 / 4;        // 26 groups of 4 MFMA k-steps (2 folded samples each)
constexpr int TILE_SAMPLES = (FT - 1) * HOP + NFFT;   // 5360
constexpr int XS_WORDS = TILE_SAMPLES + TILE_SAMPLES / HOP + 1;
constexpr int PW_STRIDE = NW * 32 + 1;  // 225
constexpr int MELW_MAX = 512;

struct FbankTables {
  float dft[NW][2][KG][64][4];
  int32_t mstart[NMEL];
  int32_t mlen[NMEL];
  int32_t moff[NMEL];
  float melw[MELW_MAX];
};

__device__ __forceinline__ int skew(int q) { return q + q / HOP; }

__global__ __launch_bounds__(NW * 64) void fbank_tile_kernel(const int16_t* __restrict__ pcm, int S, int T,
                                                            int tiles_per_seg, const FbankTables* __restrict__ tab,
                                                            float* __restrict__ L) {
  __shared__ float xs[XS_WORDS];
  __shared__ float pw[FT * PW_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.x / tiles_per_seg;
  const int t0 = (blockIdx.x - b * tiles_per_seg) * FT;
  const int64_t s0 = (int64_t)t0 * HOP - NFFT / 2;     // first sample of the tile (may be < 0)
  const int16_t* seg = pcm + (int64_t)b * S;

  for (int q = tid; q < TILE_SAMPLES; q += NW * 64) {
    const int64_t g = s0 + q;
    const float v = (g >= 0 && g < S) ? (float)seg[g] * (1.0f / 32768.0f) : 0.f;
    xs[skew(q)] = v;
  }
  __syncthreads();

  f32x16 are, aim;
#pragma unroll
  for (int r = 0; r < 16; ++r) { are[r] = 0.f; aim[r] = 0.f; }

  // The frame is real and the periodic Hamming window is symmetric (w[n] = w[400-n]), so
  //   Re X[f] =  sum_{n=0..200} c_n w[n] (x[n] + x[400-n]) cos(2 pi f n / 400),  c_0 = c_200 = 1/2
  //   Im X[f] = -sum_{n=1..199}     w[n] (x[n] - x[400-n]) sin(2 pi f n / 400)
  // (index 400-n taken mod 400): K shrinks from 400 to 201 - half the MFMAs - for one extra LDS read and
  // one add/sub per A value.  The constants c_n and the window live in the packed tables.
  const int fi = lane & 31, kk = lane >> 5;
  const int arow = fi * (HOP + 1);
  const f32x4* bre = reinterpret_cast<const f32x4*>(&tab->dft[w][0][0][lane][0]);
  const f32x4* bim = reinterpret_cast<const f32x4*>(&tab->dft[w][1][0][lane][0]);
  f32x4 cre = bre[0], cim = bim[0];
  for (int g = 0; g < KG; ++g) {
    f32x4 nre = cre, nim = cim;
    if (g + 1 < KG) {
      nre = bre[(g + 1) * 64];
      nim = bim[(g + 1) * 64];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int n = (g * 4 + u) * 2 + kk;
      n = n <= NFFT / 2 ? n : NFFT / 2;                       // padded k (201..207): table rows are zero
      const int m = n == 0 ? 0 : NFFT - n;                    // mirror sample, (400 - n) mod 400
      const float xa = xs[arow + n + (n >= HOP)];             // n <= 200 < 2*HOP
      const float xb = xs[arow + m + (m >= HOP) + (m >= 2 * HOP)];
      are = __builtin_amdgcn_mfma_f32_32x32x2f32(xa + xb, cre[u], are, 0, 0, 0);
      aim = __builtin_amdgcn_mfma_f32_32x32x2f32(xa - xb, cim[u], aim, 0, 0, 0);
    }
    cre = nre;
    cim = nim;
  }
  // power tile -> LDS [frame][bin]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int frame = (r & 3) + 8 * (r >> 2) + 4 * kk;
    pw[frame * PW_STRIDE + w * 32 + fi] = are[r] * are[r] + aim[r] * aim[r];
  }
  __syncthreads();
  for (int o = tid; o < FT * NMEL; o += NW * 64) {
    const int frame = o / NMEL, m = o - frame * NMEL;
    const int t = t0 + frame;
    if (t < T) {
      const int st = tab->mstart[m], ln = tab->mlen[m], of = tab->moff[m];
      float acc = 0.f;
      for (int i = 0; i < ln; ++i) acc += pw[frame * PW_STRIDE + st + i] * tab->melw[of + i];
      L[((int64_t)b * T + t) * NMEL + m] = 10.0f * log10f(fmaxf(acc, 1e-10f));
    }
  }
}

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
  const int c8n = ldf >> 3;
  bf16_t* out = feats + (int64_t)blockIdx.x * T * ldf;
  for (int i = tid; i < T * c8n; i += 256) {
    const int t = i / c8n, c8 = i - t * c8n;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c8 * 8 + e;
      f[e] = c < NMEL ? fmaxf(Ls[t * NMEL + c], flo) - mean[c] : 0.f;
    }
    *reinterpret_cast<u32x4*>(out + (int64_t)t * ldf + c8 * 8) = pack8(f);
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
  for (int w = 0; w < NW; ++w)
    for (int g = 0; g < KG; ++g)
      for (int l = 0; l < 64; ++l)
        for (int u = 0; u < 4; ++u) {
          const int n = (g * 4 + u) * 2 + (l >> 5);
          const int f = w * 32 + (l & 31);
          if (f >= NBIN || n > NFFT / 2) continue;
          const double win = 0.54 - 0.46 * cos(2.0 * PI * n / NFFT);
          const double ang = 2.0 * PI * (double)((n * f) % NFFT) / NFFT;
          const double half = (n == 0 || n == NFFT / 2) ? 0.5 : 1.0;   // these samples are their own mirror
          t->dft[w][0][g][l][u] = (float)(half * win * cos(ang));
          t->dft[w][1][g][l][u] = (float)(-win * sin(ang));
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

extern "C" int sdk_fbank(sdk_ctx* ctx, const int16_t* pcm, int B, int S, const void* tabs, uint16_t* feats, int ldf,
                         void* ws, size_t ws_bytes, void* stream) {
  SDK_REQUIRE(ctx && pcm && tabs && feats && ws, "sdk_fbank: null argument");
  SDK_REQUIRE(B > 0 && S > 0, "sdk_fbank: empty batch (B=%d S=%d)", B, S);
  SDK_REQUIRE(ldf >= NMEL && ldf % 8 == 0, "sdk_fbank: ldf=%d must be >= 80 and a multiple of 8", ldf);
  SDK_REQUIRE(ws_bytes >= sdk_fbank_workspace_bytes(B, S), "sdk_fbank: workspace too small");
  SDK_REQUIRE(((uintptr_t)feats % 16) == 0 && ((uintptr_t)tabs % 16) == 0, "sdk_fbank: feats/tabs must be 16-byte aligned");
  const int T = 1 + S / HOP;
  const int tps = ceil_div(T, FT);
  SDK_REQUIRE((int64_t)B * tps < (1ll << 31), "sdk_fbank: batch too large for one launch");
  {
  ProfScope ps(ctx, stream, SDK_K_FBANK_TILE, 2.0 * B * T * (double)NSYM * 2 * (NW * 32) + 2.0 * B * T * NBIN * NMEL, 2.0 * B * S + 4.0 * B * T * NMEL);
  hipLaunchKernelGGL(fbank_tile_kernel, dim3(B * tps), dim3(NW * 64), 0, (hipStream_t)stream, pcm, S, T, tps,
                     (const FbankTables*)tabs, (float*)ws);
  }
  SDK_LAUNCH_CHECK();
  ProfScope ps2(ctx, stream, SDK_K_FBANK_NORM, 3.0 * B * T * NMEL, 4.0 * B * T * NMEL + 2.0 * B * T * ldf);
  hipLaunchKernelGGL(fbank_norm_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const float*)ws, T,
                     (bf16_t*)feats, ldf);
  SDK_LAUNCH_CHECK();
  return 0;
}
