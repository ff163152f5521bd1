// Res2Net chain of one SE-Res2Net block fused per segment (gfx950).
//
//   y_1 = TDNN(u_1),  y_c = TDNN(bf16(u_c + y_{c-1}))  for c = 2..7,   TDNN = dilated k3 conv 128->128, ReLU, BN
//
// One workgroup owns one segment and walks the seven dependent convolutions with the running tile kept
// in LDS: u is read from HBM once and y written once (824 MB per block at 1000 segments instead of
// 1.4 GB for seven separate launches), and the conv taps read the frame tile - halo included - straight
// from LDS ("LDS-staged frame tiles"): row t + (j-1)*dil with segment-local reflection is just the
// fragment's row address.
//   LDS: two [TP rows][128 ch] bf16 images (256-B rows, 16-B chunk index XOR (row & 15): the 16 rows of a
//        ds_read_b128 lane group land on 16 distinct slots whatever the tap shift).
//   MFMA: 2 x 4 waves: half of the MT row tiles x 32 output channels each, so a frame fragment read from LDS
//        feeds two MFMAs; weight fragments (L2-resident) are prefetched one whole tap (4 k-steps) ahead;
//        the weight fragment is the row operand, so the epilogue packs 4 channels into one 8-byte LDS write;
//        the next chunk of u is prefetched into registers across the conv.
//   Arithmetic order is identical to seven conv_gemm launches (tap-major, 32-wide k-steps, fp32 epilogue,
//   bf16 rounding points), so results are bit-identical to the unfused schedule.
#include "common.hpp"

namespace {

constexpr int RS = 128;                  // sub-band width (channels per Res2Net chunk)
constexpr int PAR_BYTES = 7 * 3 * RS * 4;   // LDS table of the convs' epilogue parameters
constexpr int RNT = 512;                 // 8 waves: two per SIMD, each owns one 16-column output tile

struct ChainParams {
  const bf16_t* U; int64_t ldu;          // [M, scale*128] tdnn1 output
  bf16_t* R; int64_t ldr;                // [M, scale*128] chain output (chunk 0 is copied by the caller)
  const bf16_t* W[7];                    // [128][3*128] bf16 each
  const float* bias[7]; const float* scale[7]; const float* shift[7];
  int T, dil, nconv;
};

template <int MT>
__global__ __launch_bounds__(RNT, 2) void res2net_chain_kernel(ChainParams p) {
  constexpr int TP = MT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int T = p.T;
  const int64_t base = (int64_t)blockIdx.x * T;
  constexpr int NPASS = (TP * 16 + RNT - 1) / RNT;   // 16-byte chunks per thread per tile pass

  auto lds_off = [](int row, int ch16) { return row * 256 + ((ch16 ^ (row & 15)) << 4); };

  // bias / scale / shift of the (up to) seven convs: [7][3][128] fp32 behind the two images
  float* par = reinterpret_cast<float*>(smem + 2 * TP * 256);
  for (int i = tid; i < p.nconv * 3 * RS; i += RNT) {
    const int cc = i / (3 * RS), w = (i / RS) % 3, ch = i % RS;
    par[i] = (w == 0 ? p.bias[cc] : w == 1 ? p.scale[cc] : p.shift[cc])[ch];
  }
  // ---- s_1 = u_1 -> buf 0
  {
    char* b0 = smem;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int id = tid + RNT * i, row = id >> 4, ch16 = id & 15;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < T) v = *reinterpret_cast<const u32x4*>(p.U + (base + row) * p.ldu + RS + ch16 * 8);
      if (row < TP) *reinterpret_cast<u32x4*>(b0 + lds_off(row, ch16)) = v;
    }
  }

  // 2 x 4 waves: wave (wm, wq) owns row tiles [wm*MH, wm*MH + MH) and the 32 output channels [32 wq, 32 wq + 32).
  // The WEIGHT fragment is the MFMA's row operand, so a lane ends up with 4 consecutive channels of one frame.
  const int wm = wn >> 2, wq = wn & 3;
  constexpr int MH = (MT + 1) / 2;
  const int mt0 = wm * MH;
  const int64_t wofs0 = (int64_t)(wq * 32 + fr) * (3 * RS) + fq * 8;
  const int64_t wofs1 = wofs0 + (int64_t)16 * (3 * RS);
  bf16x8 bcur[2][4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    bcur[0][ks] = *reinterpret_cast<const bf16x8*>(p.W[0] + wofs0 + ks * 32);
    bcur[1][ks] = *reinterpret_cast<const bf16x8*>(p.W[0] + wofs1 + ks * 32);
  }
  for (int c = 1; c <= p.nconv; ++c) {
    const char* cur = smem + ((c - 1) & 1) * (TP * 256);
    char* nxt = smem + (c & 1) * (TP * 256);
    __syncthreads();                                        // s_c complete; the other image is free
    // prefetch u_{c+1} (consumed after the conv)
    u32x4 upre[NPASS];
    if (c < p.nconv) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int id = tid + RNT * i, row = id >> 4, ch16 = id & 15;
        upre[i] = u32x4{0u, 0u, 0u, 0u};
        if (row < T) upre[i] = *reinterpret_cast<const u32x4*>(p.U + (base + row) * p.ldu + (int64_t)RS * (c + 1) + ch16 * 8);
      }
    }
    // ---- conv c: [TP x 384] x [384 x 128]
    f32x4 acc[MH][2];
#pragma unroll
    for (int mi = 0; mi < MH; ++mi) acc[mi][0] = acc[mi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* Wc = p.W[c - 1];
    bf16x8 bnext[2][4];                                      // one tap (4 k-steps of 32) of weights ahead
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
      const int off = (j - 1) * p.dil;
      // next tap of this conv, or tap 0 of the next conv (its latency hides under the epilogue and the y pass)
      const bf16_t* Wn = j < 2 ? Wc + (j + 1) * RS : (c < p.nconv ? p.W[c] : Wc);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bnext[0][ks] = *reinterpret_cast<const bf16x8*>(Wn + wofs0 + ks * 32);
        bnext[1][ks] = *reinterpret_cast<const bf16x8*>(Wn + wofs1 + ks * 32);
      }
      int rr[MH];
#pragma unroll
      for (int mi = 0; mi < MH; ++mi) rr[mi] = reflect_idx(min((mt0 + mi) * 16 + fr, TP - 1) + off, T);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int mi = 0; mi < MH; ++mi) {
          if (mt0 + mi < MT) {                               // wave-uniform: the second row half has MT - MH tiles
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(cur + rr[mi] * 256 + (((ks * 4 + fq) ^ (rr[mi] & 15)) << 4));
            acc[mi][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bcur[0][ks], a, acc[mi][0], 0, 0, 0);
            acc[mi][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bcur[1][ks], a, acc[mi][1], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { bcur[0][ks] = bnext[0][ks]; bcur[1][ks] = bnext[1][ks]; }
    }
    // ---- epilogue: y_c = bf16(relu(acc + bias) * scale + shift) -> the free image, 8 bytes (4 channels) per write
    f32x4 cb[2], cs[2], ct[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* q = par + (c - 1) * (3 * RS) + wq * 32 + h * 16 + fq * 4;
      cb[h] = *reinterpret_cast<const f32x4*>(q);
      cs[h] = *reinterpret_cast<const f32x4*>(q + RS);
      ct[h] = *reinterpret_cast<const f32x4*>(q + 2 * RS);
    }
#pragma unroll
    for (int mi = 0; mi < MH; ++mi) {
      const int row = (mt0 + mi) * 16 + fr;
      if (mt0 + mi < MT && row < T) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 v = acc[mi][h] + cb[h];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          v = v * cs[h] + ct[h];
          uint2 pk;
          pk.x = pack2(v[0], v[1]);
          pk.y = pack2(v[2], v[3]);
          const int ch16 = wq * 4 + h * 2 + (fq >> 1);
          *reinterpret_cast<uint2*>(nxt + lds_off(row, ch16) + (fq & 1) * 8) = pk;
        }
      }
    }
    __syncthreads();                                        // y_c complete in `nxt`; every read of `cur` is done
    // ---- y_c -> HBM; s_{c+1} = bf16(y_c + u_{c+1}) in place
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int id = tid + RNT * i, row = id >> 4, ch16 = id & 15;
      if (row < T) {
        char* q = nxt + lds_off(row, ch16);
        const u32x4 y = *reinterpret_cast<const u32x4*>(q);
        *reinterpret_cast<u32x4*>(p.R + (base + row) * p.ldr + (int64_t)RS * c + ch16 * 8) = y;
        if (c < p.nconv) {
          float fy[8], fu[8];
          unpack8(y, fy);
          unpack8(upre[i], fu);
#pragma unroll
          for (int e = 0; e < 8; ++e) fy[e] += fu[e];
          *reinterpret_cast<u32x4*>(q) = pack8(fy);
        }
      }
    }
  }
}

}  // namespace

extern "C" int sdk_res2net_chain_max_frames(void) { return 208; }

extern "C" int sdk_res2net_chain(sdk_ctx* ctx, const uint16_t* U, int64_t ldu, uint16_t* R, int64_t ldr, const uint16_t* const* W,
                                 const float* const* bias, const float* const* scale, const float* const* shift, int nconv,
                                 int B, int T, int dil, void* stream) {
  SDK_REQUIRE(ctx && U && R && W && bias && scale && shift, "sdk_res2net_chain: null argument");
  SDK_REQUIRE(nconv >= 1 && nconv <= 7, "sdk_res2net_chain: nconv=%d must be in [1, 7]", nconv);
  SDK_REQUIRE(B > 0 && T > dil && T <= 208 && dil >= 1, "sdk_res2net_chain: T=%d frames (dilation %d) unsupported (dil < T <= 208)", T, dil);
  SDK_REQUIRE(ldu % 8 == 0 && ldr % 8 == 0 && ldu >= (int64_t)RS * (nconv + 1) && ldr >= (int64_t)RS * (nconv + 1), "sdk_res2net_chain: bad row strides");
  ChainParams p;
  p.U = (const bf16_t*)U; p.ldu = ldu; p.R = (bf16_t*)R; p.ldr = ldr; p.T = T; p.dil = dil; p.nconv = nconv;
  for (int i = 0; i < 7; ++i) {
    const int k = i < nconv ? i : 0;
    SDK_REQUIRE(W[k] && bias[k] && scale[k] && shift[k], "sdk_res2net_chain: conv %d parameters missing", k);
    p.W[i] = (const bf16_t*)W[k]; p.bias[i] = bias[k]; p.scale[i] = scale[k]; p.shift[i] = shift[k];
  }
  ProfScope ps(ctx, stream, SDK_K_RES2NET, 2.0 * B * T * (double)RS * 3 * RS * nconv, 2.0 * 2.0 * B * T * RS * nconv);
  if (sdk_lds_optin(ctx, (const void*)res2net_chain_kernel<13>, 2 * 208 * 256 + PAR_BYTES)) return 1;
  if (sdk_lds_optin(ctx, (const void*)res2net_chain_kernel<7>, 2 * 112 * 256 + PAR_BYTES)) return 1;
  if (T <= 112) hipLaunchKernelGGL(res2net_chain_kernel<7>, dim3(B), dim3(RNT), 2 * 112 * 256 + PAR_BYTES, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(res2net_chain_kernel<13>, dim3(B), dim3(RNT), 2 * 208 * 256 + PAR_BYTES, (hipStream_t)stream, p);
  SDK_LAUNCH_CHECK();
  return 0;
}
