// conv_gemm: dilated 1-D convolution over frames lowered to one bf16 MFMA GEMM (gfx950).
//
//   rows  = frames (M = B*T, channel-last activations [M, lda] bf16)
//   K     = taps * Cin, the A operand of tap j is the SAME activation tensor with its rows
//           re-indexed (segment-local reflect of t + (j - taps/2)*dil): no im2col buffer exists,
//           the shifted frame tile is gathered straight into LDS
//   cols  = output channels, W stored [N][taps*Cin] so both MFMA operands are K-contiguous
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 as 4x4 v_mfma_f32_16x16x32_bf16.
// LDS image: [row][64 k] bf16, 128-B rows, 16-B chunk index XOR (row & 7): conflict-free for the
// ds_read_b128 fragment reads (lane groups of MI355X_MICROARCH §LDS) and for the ds_write_b128 fill.
// Software pipeline: global loads of K-step s+1 are in flight (registers) while step s is computed.
// Epilogue (fp32): +bias +per-segment bias, ReLU, BN affine, tanh; result staged through LDS and
// written as whole 16-B chunks; optional second output S = bf16(C + X2) (Res2Net chain).
#include "common.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;
constexpr int LDS_AB = (BM + BN) * BK * 2;        // 32 KiB
constexpr int CT_STRIDE = (BN + 8) * 2;           // bytes per staged C row (272: breaks the 256-B period)
constexpr int LDS_CT = BM * CT_STRIDE;            // 34816
constexpr int LDS_BYTES = LDS_CT > LDS_AB ? LDS_CT : LDS_AB;

struct Params {
  const bf16_t* A; int64_t lda;
  const bf16_t* W;
  bf16_t* C; int64_t ldc;
  float* C32; int64_t ldc32;
  const float* bias; const float* scale; const float* shift;
  const float* ubias; int64_t ldub;
  const bf16_t* X2; int64_t ldx2;
  bf16_t* S; int64_t lds;
  int M, N, Cin, taps, dil, T;
  uint32_t flags;
};

__global__ __launch_bounds__(NT, 2) void conv_gemm_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + BM * BK * 2;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  const int nbn = p.N / BN;
  const int nbm = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, nbn * nbm);
  const int bn = tile % nbn, bm = tile / nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  // ---- staging assignment: thread -> 4 rows (tid>>3)+32i, one 16-B chunk column (tid&7)
  const int ch = tid & 7;
  const int r0 = tid >> 3;
  int segbase[4], tloc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i;
    m = m < p.M ? m : p.M - 1;
    if (p.taps > 1) {
      const int b = m / p.T;
      segbase[i] = b * p.T;
      tloc[i] = m - b * p.T;
    } else {
      segbase[i] = m;
      tloc[i] = 0;
    }
  }
  const int Ktot = p.taps * p.Cin;
  const bf16_t* wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wrow[i] = p.W + (int64_t)(n0 + r0 + 32 * i) * Ktot + ch * 8;

  uint32_t lds_w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + 32 * i;
    lds_w[i] = row * 128 + ((ch ^ (row & 7)) << 4);
  }

  const int ksteps_per_tap = p.Cin / BK;
  const int nk = p.taps * ksteps_per_tap;
  const int half = p.taps >> 1;

  u32x4 ra[4], rb[4];
  auto gload = [&](int s) {
    const int j = s / ksteps_per_tap;
    const int kc = (s - j * ksteps_per_tap) * BK;
    const int off = (j - half) * p.dil;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int src = p.taps > 1 ? segbase[i] + reflect_idx(tloc[i] + off, p.T) : segbase[i];
      ra[i] = *reinterpret_cast<const u32x4*>(p.A + (int64_t)src * p.lda + kc + ch * 8);
      rb[i] = *reinterpret_cast<const u32x4*>(wrow[i] + j * p.Cin + kc);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read addresses (row & 7 == lane & 7 because tile/wave/sub-tile offsets are multiples of 8)
  const int fr = lane & 15, fq = lane >> 4;
  uint32_t a_off[4], b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_off[i] = (wm * 64 + i * 16 + fr) * 128;
    b_off[i] = (wn * 64 + i * 16 + fr) * 128;
  }
  const int sw = lane & 7;

  gload(0);
  for (int s = 0; s < nk; ++s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<u32x4*>(sA + lds_w[i]) = ra[i];
      *reinterpret_cast<u32x4*>(sB + lds_w[i]) = rb[i];
    }
    __syncthreads();
    if (s + 1 < nk) gload(s + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint32_t coff = ((ks * 4 + fq) ^ sw) << 4;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(sA + a_off[i] + coff);
        bfr[i] = *reinterpret_cast<const bf16x8*>(sB + b_off[i] + coff);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }

  // ------------------------------------------------------------------ epilogue
  float cb[4], cs[4], ct[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int col = n0 + wn * 64 + ni * 16 + fr;
    cb[ni] = p.bias ? p.bias[col] : 0.f;
    cs[ni] = p.scale ? p.scale[col] : 1.f;
    ct[ni] = p.shift ? p.shift[col] : 0.f;
  }
  const bool relu = p.flags & SDK_GEMM_RELU, tnh = p.flags & SDK_GEMM_TANH;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wm * 64 + mi * 16 + fq * 4 + r;
      const int m = m0 + row;
      const float* ub = nullptr;
      if (p.ubias) {
        const int mm = m < p.M ? m : p.M - 1;
        ub = p.ubias + (int64_t)(mm / p.T) * p.ldub;
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int lc = wn * 64 + ni * 16 + fr;
        float v = acc[mi][ni][r] + cb[ni];
        if (ub) v += ub[n0 + lc];
        if (relu) v = fmaxf(v, 0.f);
        v = v * cs[ni] + ct[ni];
        if (tnh) v = tanhf(v);
        if (p.C32 && m < p.M) p.C32[(int64_t)m * p.ldc32 + n0 + lc] = v;
        *reinterpret_cast<bf16_t*>(smem + row * CT_STRIDE + lc * 2) = f32_to_bf16(v);
      }
    }
  }
  __syncthreads();
  if (p.C || p.S) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int id = tid + NT * i;
      const int row = id >> 4, cc = id & 15;
      const int m = m0 + row;
      if (m < p.M) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + row * CT_STRIDE + cc * 16);
        if (p.C) *reinterpret_cast<u32x4*>(p.C + (int64_t)m * p.ldc + n0 + cc * 8) = v;
        if (p.S) {
          const u32x4 x = *reinterpret_cast<const u32x4*>(p.X2 + (int64_t)m * p.ldx2 + n0 + cc * 8);
          float fv[8], fx[8];
          unpack8(v, fv);
          unpack8(x, fx);
#pragma unroll
          for (int e = 0; e < 8; ++e) fv[e] += fx[e];
          *reinterpret_cast<u32x4*>(p.S + (int64_t)m * p.lds + n0 + cc * 8) = pack8(fv);
        }
      }
    }
  }
}

}  // namespace

extern "C" int sdk_conv_gemm(sdk_ctx* ctx, const sdk_conv_gemm_args* a, void* stream) {
  SDK_REQUIRE(ctx && a, "sdk_conv_gemm: null ctx/args");
  SDK_REQUIRE(a->A && a->W, "sdk_conv_gemm: A and W are required");
  SDK_REQUIRE(a->M > 0 && a->N > 0 && a->N % BN == 0, "sdk_conv_gemm: N=%d must be a positive multiple of %d", a->N, BN);
  SDK_REQUIRE(a->Cin > 0 && a->Cin % BK == 0, "sdk_conv_gemm: Cin=%d must be a multiple of %d", a->Cin, BK);
  SDK_REQUIRE(a->taps >= 1 && (a->taps & 1), "sdk_conv_gemm: taps=%d must be odd", a->taps);
  SDK_REQUIRE(a->T > 0 && a->M % a->T == 0, "sdk_conv_gemm: M=%d must be a multiple of T=%d", a->M, a->T);
  SDK_REQUIRE(a->taps == 1 || (a->taps / 2) * a->dil < a->T, "sdk_conv_gemm: segment of T=%d frames shorter than the conv halo %d", a->T, (a->taps / 2) * a->dil);
  SDK_REQUIRE(a->lda % 8 == 0 && a->lda >= a->Cin, "sdk_conv_gemm: lda=%lld must be >= Cin and a multiple of 8", (long long)a->lda);
  SDK_REQUIRE(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->W % 16) == 0, "sdk_conv_gemm: A/W must be 16-byte aligned");
  SDK_REQUIRE(a->C || a->C32 || a->S, "sdk_conv_gemm: no output requested");
  if (a->C) SDK_REQUIRE(a->ldc % 8 == 0 && a->ldc >= a->N && ((uintptr_t)a->C % 16) == 0, "sdk_conv_gemm: bad C/ldc");
  if (a->S) SDK_REQUIRE(a->X2 && a->lds % 8 == 0 && a->ldx2 % 8 == 0 && ((uintptr_t)a->S % 16) == 0 && ((uintptr_t)a->X2 % 16) == 0, "sdk_conv_gemm: S needs X2 and 16-byte aligned rows");
  if (a->C32) SDK_REQUIRE(a->ldc32 >= a->N, "sdk_conv_gemm: bad ldc32");
  if (a->ubias) SDK_REQUIRE(a->ldub >= a->N, "sdk_conv_gemm: bad ldub");

  Params p;
  p.A = (const bf16_t*)a->A; p.lda = a->lda; p.W = (const bf16_t*)a->W;
  p.C = (bf16_t*)a->C; p.ldc = a->ldc; p.C32 = a->C32; p.ldc32 = a->ldc32;
  p.bias = a->bias; p.scale = a->scale; p.shift = a->shift; p.ubias = a->ubias; p.ldub = a->ldub;
  p.X2 = (const bf16_t*)a->X2; p.ldx2 = a->ldx2; p.S = (bf16_t*)a->S; p.lds = a->lds;
  p.M = a->M; p.N = a->N; p.Cin = a->Cin; p.taps = a->taps; p.dil = a->dil; p.T = a->T; p.flags = a->flags;

  const int nwg = (a->N / BN) * ceil_div(a->M, BM);
  const double kk = (double)a->taps * a->Cin;
  ProfScope ps(ctx, stream, SDK_K_CONV_GEMM, 2.0 * a->M * a->N * kk,
               2.0 * a->M * a->Cin + 2.0 * a->N * kk + (a->C ? 2.0 : 0.0) * a->M * a->N + (a->C32 ? 4.0 : 0.0) * a->M * a->N +
                   (a->S ? 4.0 : 0.0) * a->M * a->N);
  hipLaunchKernelGGL(conv_gemm_kernel, dim3(nwg), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
  SDK_LAUNCH_CHECK();
  return 0;
}
