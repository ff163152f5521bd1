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
//        feeds two MFMAs; weight fragments (L2-resident; from the blob's fragment-ordered copy when present: one
//        contiguous KiB per load) are prefetched one whole tap (4 k-steps) ahead; the weight fragment is the row
//        operand, so the epilogue packs 4 channels into one 8-byte LDS write; the next chunk of u is requested
//        BEHIND the conv's last weight prefetch (vmcnt retires in order) and waited for after the epilogue.
//   u and y move through buffer instructions (uniform resource + one 32-bit lane offset + scalar pass offset).
//   Arithmetic order is identical to seven conv_gemm launches (tap-major, 32-wide k-steps, fp32 epilogue,
//   bf16 rounding points), so results are bit-identical to the unfused schedule.
#include <type_traits>

#include "common.hpp"

namespace {

constexpr int RS = 128;                  // sub-band width (channels per Res2Net chunk)
constexpr int PAR_BYTES = 7 * 3 * RS * 4;   // LDS table of the convs' epilogue parameters
constexpr int RNT = 512;                 // 8 waves: two per SIMD, each owns one 16-column output tile

struct ChainParams {
  const bf16_t* U; int64_t ldu;          // [M, scale*128] tdnn1 output
  bf16_t* R; int64_t ldr;                // [M, scale*128] chain output (chunk 0 is copied by the caller)
  const bf16_t* W[7];                    // [128][3*128] bf16 each
  const bf16_t* Wpk[7];                  // the same weights in fragment order (ecapa_layout.h EL_CHAINPACK), used when PACKED
  const float* bias[7]; const float* scale[7]; const float* shift[7];
  int T, dil, nconv;
  unsigned long long* dbg;               // diagnostics only: [workgroup][64] 100 MHz time stamps of wave 0 (tools/res2net_timeline.py), or null
};

template <int MT, bool PACKED, bool F16 = false>   // F16: fp16 storage and operands (the single-plane fp16 contract) instead of bf16
__global__ __launch_bounds__(RNT, 2) void res2net_chain_kernel(ChainParams p) {
  constexpr int TP = MT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int T = p.T;
  const int64_t base = (int64_t)blockIdx.x * T;
  constexpr int NPASS = (TP * 16 + RNT - 1) / RNT;   // 16-byte chunks per thread per tile pass

  int nstamp = 2;
  auto stamp = [&]() {
    if (p.dbg && tid == 0 && blockIdx.x < 256 && nstamp < 64) p.dbg[blockIdx.x * 64 + nstamp++] = __builtin_amdgcn_s_memrealtime();
  };
  if (p.dbg && tid == 0 && blockIdx.x < 256) p.dbg[blockIdx.x * 64] = __builtin_amdgcn_s_memrealtime();
  auto lds_off = [](int row, int ch16) { return row * 256 + ((ch16 ^ (row & 15)) << 4); };

  // bias / scale / shift of the (up to) seven convs: [7][3][128] fp32 behind the two images.  Wave w fetches conv w's three
  // arrays (uniform pointers, 2 floats per lane): three loads per lane in flight together - one round trip, where the
  // element-by-element form (pointer picked per element) took five dependent ones (the prologue was 8 us of a 70-us workgroup)
  float* par = reinterpret_cast<float*>(smem + 2 * TP * 256);
  float2 pv[3] = {};
  if (wn < p.nconv) {
    pv[0] = reinterpret_cast<const float2*>(p.bias[wn])[lane];
    pv[1] = reinterpret_cast<const float2*>(p.scale[wn])[lane];
    pv[2] = reinterpret_cast<const float2*>(p.shift[wn])[lane];
  }
  // Tile passes (u in, y out): thread -> (row r0 + 32 i, 16-byte chunk c16), i < NPASS.  32 i leaves row & 15 alone, so ONE LDS
  // offset, ONE u offset and ONE y offset per lane serve every pass (immediate / uniform strides), and `i < nrow` is "row < T".
  const int r0 = tid >> 4, c16 = tid & 15;
  const int nrow = (T - r0 + 31) >> 5;
  const int loff = lds_off(r0, c16);
  // u and y move through BUFFER instructions: (uniform resource = this segment's rows) + (ONE 32-bit per-lane offset) + (scalar
  // offset = pass stride x i + chunk) - flat addressing kept a zero-extended 64-bit register pair per pass alive across the conv
  // loop (14 + 14 registers, and a spill reload in front of the y stores).
  const uint32_t uoff = ((uint32_t)r0 * (uint32_t)p.ldu + (uint32_t)c16 * 8u) * 2u, ustride = 64u * (uint32_t)p.ldu;
  const uint32_t ulast = ((uint32_t)(T - 1) * (uint32_t)p.ldu + (uint32_t)c16 * 8u) * 2u;
  const uint32_t yoff = ((uint32_t)r0 * (uint32_t)p.ldr + (uint32_t)c16 * 8u) * 2u, ystride = 64u * (uint32_t)p.ldr;
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.U + base * p.ldu), 0, (int)((uint32_t)T * (uint32_t)p.ldu * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.R + base * p.ldr, 0, (int)((uint32_t)T * (uint32_t)p.ldr * 2u), 0x00020000);
  // pass i of chunk ch: rows past T re-read row T - 1 (the request count stays static; the value is never used)
  auto uload = [&](int i, int ch) -> u32x4 {
    return __builtin_amdgcn_raw_buffer_load_b128(urs, i < nrow ? uoff : ulast - (uint32_t)i * ustride, (uint32_t)i * ustride + (uint32_t)(ch * RS * 2), 0);
  };
  // ---- s_1 = u_1 -> buf 0
  {
    char* b0 = smem;
    u32x4 v[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) v[i] = uload(i, 1);      // all requests first (one HBM round trip, not NPASS)
    if (wn < p.nconv) {                                      // (the parameter loads are older than the u loads: a counted wait)
#pragma unroll
      for (int q = 0; q < 3; ++q) reinterpret_cast<float2*>(par + wn * (3 * RS) + q * RS)[lane] = pv[q];
    }
#pragma unroll
    for (int i = 0; i < NPASS; ++i)
      if (r0 + 32 * i < TP) *reinterpret_cast<u32x4*>(b0 + loff + i * (32 * 256)) = i < nrow ? v[i] : u32x4{0u, 0u, 0u, 0u};
  }

  // 2 x 4 waves: wave (wm, wq) owns row tiles [wm*MH, wm*MH + MH) and the 32 output channels [32 wq, 32 wq + 32).
  // The WEIGHT fragment is the MFMA's row operand, so a lane ends up with 4 consecutive channels of one frame.
  const int wm = wn >> 2, wq = wn & 3;
  constexpr int MH = (MT + 1) / 2;
  const int mt0 = wm * MH;
  const int64_t wofs0 = (int64_t)(wq * 32 + fr) * (3 * RS) + fq * 8;
  const int64_t wofs1 = wofs0 + (int64_t)16 * (3 * RS);
  int ebase[2];                                              // epilogue write offsets of row tile mt0 (8 bytes = 4 channels per lane)
#pragma unroll
  for (int h = 0; h < 2; ++h) ebase[h] = lds_off(mt0 * 16 + fr, wq * 4 + h * 2 + (fq >> 1)) + (fq & 1) * 8;
  // weight fragment (conv cc, tap, column tile h, k-step ks) of this lane: 16 bytes.  PACKED: the host's fragment-ordered copy -
  // a wave's load is 1 KiB contiguous (8 cache lines); otherwise 16 row pieces of 64 B of the [128][384] matrix (16 lines
  // half used: with every workgroup fetching 64 KiB per tap that is what the L2 was busy with - tap waits of ~1 us)
  auto wfrag = [&](int cc, int tap, int h, int ks) -> bf16x8 {
    if constexpr (PACKED)
      return *reinterpret_cast<const bf16x8*>(p.Wpk[cc] + ((((tap * 4 + wq) * 2 + h) * 4 + ks) * 64 + lane) * 8);
    else
      return *reinterpret_cast<const bf16x8*>(p.W[cc] + tap * RS + (h ? wofs1 : wofs0) + ks * 32);
  };
  bf16x8 bcur[2][4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    bcur[0][ks] = wfrag(0, 0, 0, ks);
    bcur[1][ks] = wfrag(0, 0, 1, ks);
  }
  // (same pin as at the end of tap 2: with every load complete on BOTH ways into the conv loop, its first wait does not have to
  // cover the previous conv's y stores)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    asm volatile("" : "+v"(bcur[0][ks]));
    asm volatile("" : "+v"(bcur[1][ks]));
  }
  for (int c = 1; c <= p.nconv; ++c) {
    const char* cur = smem + ((c - 1) & 1) * (TP * 256);
    char* nxt = smem + (c & 1) * (TP * 256);
    __syncthreads();                                        // s_c complete; the other image is free
    stamp();
    u32x4 upre[NPASS];                                       // u_{c+1}, consumed after the conv; requested inside tap 2 (see there)
    // ---- conv c: [TP x 384] x [384 x 128]
    f32x4 acc[MH][2];
#pragma unroll
    for (int mi = 0; mi < MH; ++mi) acc[mi][0] = acc[mi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 bnext[2][4];                                      // one tap (4 k-steps of 32) of weights ahead
    // vmcnt retires in order, so a wait for weights also waits for every OLDER request.  The HBM fetch of u_{c+1}
    // (microseconds) is therefore issued BEHIND the last weight prefetch of the conv (tap 2 fetches the next conv's tap 0):
    // the wait for those weights can leave the u loads in flight (counted vmcnt), and u has all of tap 2, the epilogue and a
    // barrier to arrive.  Issued at the top of the conv, as before, every conv stalled at the end of tap 0 until u had come
    // back from HBM.  Taps 0 and 1 stay a rolled loop (unrolled, hipcc hoists every fragment read and spills 89 registers);
    // tap 2 is its own copy so that the request order around the u prefetch is static.
    auto tap = [&](const int j, const bool fetch_u) {
      const int off = (j - 1) * p.dil;
      // next tap of this conv, or tap 0 of the next conv (its latency hides under the epilogue and the y pass; the last conv
      // re-reads its own tap 0)
      const int ncc = j < 2 ? c - 1 : min(c, p.nconv - 1), ntap = j < 2 ? j + 1 : 0;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bnext[0][ks] = wfrag(ncc, ntap, 0, ks);
        bnext[1][ks] = wfrag(ncc, ntap, 1, ks);
      }
      if (fetch_u) {
        __builtin_amdgcn_sched_barrier(0);                   // the u requests must stay BEHIND the weight requests
#pragma unroll
        for (int i = 0; i < NPASS; ++i) upre[i] = uload(i, min(c + 1, p.nconv));   // unconditional: the request count stays static
      }
      int rr[MH];
#pragma unroll
      for (int mi = 0; mi < MH; ++mi) rr[mi] = reflect_idx(min((mt0 + mi) * 16 + fr, TP - 1) + off, T);
      // frame fragments travel RING steps (one step = one fragment = two MFMAs) ahead of their use through a small register
      // ring; the scheduling fences keep hipcc from hoisting a whole tap's 28 reads to the top (which spills)
      constexpr int RING = 5, NSTEP = 4 * MH;
      bf16x8 af[RING];
      auto rd = [&](int s) {
        const int ks = s / MH, mi = s % MH;                  // (the second row half has MT - MH tiles: its last step recomputes the
        af[s % RING] = *reinterpret_cast<const bf16x8*>(cur + rr[mi] * 256 + (((ks * 4 + fq) ^ (rr[mi] & 15)) << 4));   // clamped last row
      };                                                     //  tile and drops it in the epilogue - the other half has MH real tiles anyway)
#pragma unroll
      for (int s = 0; s < RING - 1; ++s) rd(s);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        if (s + RING - 1 < NSTEP) rd(s + RING - 1);
        const int ks = s / MH, mi = s % MH;
        acc[mi][0] = mfma_16x16x32<F16>(bcur[0][ks], af[s % RING], acc[mi][0]);
        acc[mi][1] = mfma_16x16x32<F16>(bcur[1][ks], af[s % RING], acc[mi][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { bcur[0][ks] = bnext[0][ks]; bcur[1][ks] = bnext[1][ks]; }
    };
#pragma unroll 1
    for (int j = 0; j < 2; ++j) { tap(j, false); stamp(); }
    tap(2, true);                                           // (the last conv re-reads its own chunk: one code copy less)
    // Pin "the next conv's tap-0 weights have arrived" HERE (a counted wait that leaves the u loads in flight): otherwise the
    // first wait of the next conv sits behind this conv's y stores and waits for their acknowledgement from HBM.
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      asm volatile("" : "+v"(bcur[0][ks]));
      asm volatile("" : "+v"(bcur[1][ks]));
    }
    stamp();
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
          pk.x = pack2t<F16>(v[0], v[1]);
          pk.y = pack2t<F16>(v[2], v[3]);
          *reinterpret_cast<uint2*>(nxt + ebase[h] + mi * (16 * 256)) = pk;    // row & 15 == fr for every mi: one offset per h
        }
      }
    }
    stamp();
    __syncthreads();                                        // y_c complete in `nxt`; every read of `cur` is done
    stamp();
    // ---- y_c -> HBM; s_{c+1} = bf16(y_c + u_{c+1}) in place.  (Doing both straight from the accumulators in the epilogue - no
    // barrier, no second pass - was measured: the 8-byte-per-lane u loads and y stores it needs touch 16 cache lines per
    // instruction, and the chain went from 0.66 to 0.96 ms.)
    // every u request is waited for HERE, on every lane: a load left pending on some path would make the first wait of the next
    // conv cover it - and, vmcnt being in-order, the y stores issued behind it (an HBM round trip per conv)
#pragma unroll
    for (int i = 0; i < NPASS; ++i) asm volatile("" : "+v"(upre[i]));
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      if (i < nrow) {
        char* q = nxt + loff + i * (32 * 256);
        const u32x4 y = *reinterpret_cast<const u32x4*>(q);
        __builtin_amdgcn_raw_buffer_store_b128(y, yrs, yoff, (uint32_t)i * ystride + (uint32_t)(c * RS * 2), 0);
        if (c < p.nconv) {
          float fy[8], fu[8];
          unpack8t<F16>(y, fy);
          unpack8t<F16>(upre[i], fu);
#pragma unroll
          for (int e = 0; e < 8; ++e) fy[e] += fu[e];
          *reinterpret_cast<u32x4*>(q) = pack8t<F16>(fy);
        }
      }
    }
    stamp();
  }
  if (p.dbg && tid == 0 && blockIdx.x < 256) p.dbg[blockIdx.x * 64 + 1] = nstamp;
}


// ---- The same chain with TWO segments per CU --------------------------------------------------------------------------------
// The kernel above spends 43 % of a conv outside the taps (epilogue, a barrier, the y pass), with one workgroup per CU: the matrix
// pipe idles meanwhile.  Here a workgroup is 4 waves (one per SIMD, 256 registers each) and keeps ONE image, written in place, so
// two workgroups (two segments) fit a CU and one's epilogue / y pass / barriers run under the other's taps.
//   wave wq owns output channels [32 wq, 32 wq + 32) of ALL MT row tiles (a frame fragment still feeds two MFMAs);
//   in place: the taps read s_c (halo included) from the image, a barrier, the epilogue writes y_c over it, a barrier, the y pass
//   stores y_c and writes s_{c+1} = bf16(y_c + u_{c+1}) - three barriers per conv instead of two, hidden by the other workgroup;
//   u_{c+1} is requested at the top of the epilogue (behind the next conv's tap-0 weights, which are pinned first).
// Same arithmetic order: bit-identical to the 8-wave kernel and to seven conv_gemm launches.
constexpr int RNT4 = 256;
template <int MT, bool PACKED, bool F16 = false>
__global__ __launch_bounds__(RNT4, 2) void res2net_chain4_kernel(ChainParams p) {
  constexpr int TP = MT * 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int T = p.T;
  const int64_t base = (int64_t)blockIdx.x * T;
  constexpr int NPASS = (TP * 16 + RNT4 - 1) / RNT4;  // 16-byte chunks per thread per tile pass (13 at MT = 13)
  int nstamp = 2;
  auto stamp = [&]() {
    if (p.dbg && tid == 0 && blockIdx.x < 256 && nstamp < 64) p.dbg[blockIdx.x * 64 + nstamp++] = __builtin_amdgcn_s_memrealtime();
  };
  if (p.dbg && tid == 0 && blockIdx.x < 256) p.dbg[blockIdx.x * 64] = __builtin_amdgcn_s_memrealtime();
  auto lds_off = [](int row, int ch16) { return row * 256 + ((ch16 ^ (row & 15)) << 4); };
  char* img = smem;
  float* par = reinterpret_cast<float*>(smem + TP * 256);
  // epilogue parameters: wave w fetches convs w and w + 4 (uniform pointers, 2 floats per lane)
  float2 pv[2][3] = {};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int cc = wq + 4 * q;
    if (cc < p.nconv) {
      pv[q][0] = reinterpret_cast<const float2*>(p.bias[cc])[lane];
      pv[q][1] = reinterpret_cast<const float2*>(p.scale[cc])[lane];
      pv[q][2] = reinterpret_cast<const float2*>(p.shift[cc])[lane];
    }
  }
  // tile passes: thread -> (row r0 + 16 i, 16-byte chunk c16): 16 i leaves row & 15 alone -> one LDS / u / y offset per lane
  const int r0 = tid >> 4, c16 = tid & 15;
  const int nrow = (T - r0 + 15) >> 4;                      // passes with row < T
  const int loff = lds_off(r0, c16);
  const uint32_t uoff = ((uint32_t)r0 * (uint32_t)p.ldu + (uint32_t)c16 * 8u) * 2u, ustride = 32u * (uint32_t)p.ldu;
  const uint32_t ulast = ((uint32_t)(T - 1) * (uint32_t)p.ldu + (uint32_t)c16 * 8u) * 2u;
  const uint32_t yoff = ((uint32_t)r0 * (uint32_t)p.ldr + (uint32_t)c16 * 8u) * 2u, ystride = 32u * (uint32_t)p.ldr;
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.U + base * p.ldu), 0, (int)((uint32_t)T * (uint32_t)p.ldu * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.R + base * p.ldr, 0, (int)((uint32_t)T * (uint32_t)p.ldr * 2u), 0x00020000);
  auto uload = [&](int i, int ch) -> u32x4 {
    return __builtin_amdgcn_raw_buffer_load_b128(urs, i < nrow ? uoff : ulast - (uint32_t)i * ustride, (uint32_t)i * ustride + (uint32_t)(ch * RS * 2), 0);
  };
  {
    u32x4 v[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) v[i] = uload(i, 1);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int cc = wq + 4 * q;
      if (cc < p.nconv) {
#pragma unroll
        for (int w = 0; w < 3; ++w) reinterpret_cast<float2*>(par + cc * (3 * RS) + w * RS)[lane] = pv[q][w];
      }
    }
#pragma unroll
    for (int i = 0; i < NPASS; ++i)
      if (r0 + 16 * i < TP) *reinterpret_cast<u32x4*>(img + loff + i * (16 * 256)) = i < nrow ? v[i] : u32x4{0u, 0u, 0u, 0u};
  }
  const int64_t wofs0 = (int64_t)(wq * 32 + fr) * (3 * RS) + fq * 8;
  const int64_t wofs1 = wofs0 + (int64_t)16 * (3 * RS);
  int ebase[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) ebase[h] = lds_off(fr, wq * 4 + h * 2 + (fq >> 1)) + (fq & 1) * 8;
  auto wfrag = [&](int cc, int tap, int h, int ks) -> bf16x8 {
    if constexpr (PACKED)
      return *reinterpret_cast<const bf16x8*>(p.Wpk[cc] + ((((tap * 4 + wq) * 2 + h) * 4 + ks) * 64 + lane) * 8);
    else
      return *reinterpret_cast<const bf16x8*>(p.W[cc] + tap * RS + (h ? wofs1 : wofs0) + ks * 32);
  };
  bf16x8 bcur[2][4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    bcur[0][ks] = wfrag(0, 0, 0, ks);
    bcur[1][ks] = wfrag(0, 0, 1, ks);
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    asm volatile("" : "+v"(bcur[0][ks]));
    asm volatile("" : "+v"(bcur[1][ks]));
  }
  for (int c = 1; c <= p.nconv; ++c) {
    __syncthreads();                                        // s_c complete in the image
    stamp();
    f32x4 acc[MT][2];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) acc[mi][0] = acc[mi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 bnext[2][4];
    auto tap = [&](const int j) {
      const int off = (j - 1) * p.dil;
      const int ncc = j < 2 ? c - 1 : min(c, p.nconv - 1), ntap = j < 2 ? j + 1 : 0;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bnext[0][ks] = wfrag(ncc, ntap, 0, ks);
        bnext[1][ks] = wfrag(ncc, ntap, 1, ks);
      }
      int rr[MT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) rr[mi] = reflect_idx(min(mi * 16 + fr, TP - 1) + off, T);
      constexpr int RING = 5, NSTEP = 4 * MT;
      bf16x8 af[RING];
      auto rd = [&](int s) {
        const int ks = s / MT, mi = s % MT;
        af[s % RING] = *reinterpret_cast<const bf16x8*>(img + rr[mi] * 256 + (((ks * 4 + fq) ^ (rr[mi] & 15)) << 4));
      };
#pragma unroll
      for (int s = 0; s < RING - 1; ++s) rd(s);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        __builtin_amdgcn_sched_barrier(0);
        if (s + RING - 1 < NSTEP) rd(s + RING - 1);
        const int ks = s / MT, mi = s % MT;
        acc[mi][0] = mfma_16x16x32<F16>(bcur[0][ks], af[s % RING], acc[mi][0]);
        acc[mi][1] = mfma_16x16x32<F16>(bcur[1][ks], af[s % RING], acc[mi][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { bcur[0][ks] = bnext[0][ks]; bcur[1][ks] = bnext[1][ks]; }
    };
#pragma unroll 1
    for (int j = 0; j < 3; ++j) { tap(j); stamp(); }
    u32x4 upre[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) upre[i] = uload(i, min(c + 1, p.nconv));   // has the epilogue and two barriers to arrive
    stamp();
    __syncthreads();                                        // every wave has read s_c for good: the image may be overwritten
    stamp();
    f32x4 cb[2], cs[2], ct[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* q = par + (c - 1) * (3 * RS) + wq * 32 + h * 16 + fq * 4;
      cb[h] = *reinterpret_cast<const f32x4*>(q);
      cs[h] = *reinterpret_cast<const f32x4*>(q + RS);
      ct[h] = *reinterpret_cast<const f32x4*>(q + 2 * RS);
    }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      if (mi * 16 + fr < T) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x4 v = acc[mi][h] + cb[h];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          v = v * cs[h] + ct[h];
          uint2 pk;
          pk.x = pack2t<F16>(v[0], v[1]);
          pk.y = pack2t<F16>(v[2], v[3]);
          *reinterpret_cast<uint2*>(img + ebase[h] + mi * (16 * 256)) = pk;
        }
      }
    }
    // the next conv's tap-0 weights (requested at the top of tap 2, OLDER than the u requests) are pinned here, behind the epilogue
    // they had to arrive under: a counted wait that leaves the u loads in flight, and no later wait can sit behind the y stores
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      asm volatile("" : "+v"(bcur[0][ks]));
      asm volatile("" : "+v"(bcur[1][ks]));
    }
    stamp();
    __syncthreads();                                        // y_c complete in the image
    stamp();
#pragma unroll
    for (int i = 0; i < NPASS; ++i) asm volatile("" : "+v"(upre[i]));
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      if (i < nrow) {
        char* q = img + loff + i * (16 * 256);
        const u32x4 y = *reinterpret_cast<const u32x4*>(q);
        __builtin_amdgcn_raw_buffer_store_b128(y, yrs, yoff, (uint32_t)i * ystride + (uint32_t)(c * RS * 2), 0);
        if (c < p.nconv) {
          float fy[8], fu[8];
          unpack8t<F16>(y, fy);
          unpack8t<F16>(upre[i], fu);
#pragma unroll
          for (int e = 0; e < 8; ++e) fy[e] += fu[e];
          *reinterpret_cast<u32x4*>(q) = pack8t<F16>(fy);
        }
      }
    }
    stamp();
  }
  if (p.dbg && tid == 0 && blockIdx.x < 256) p.dbg[blockIdx.x * 64 + 1] = nstamp;
}

}  // namespace

extern "C" int sdk_res2net_chain_max_frames(void) { return 208; }

// Internal entry (sdk_ecapa_forward): Wpk may be null, or hold the fragment-ordered copies of W (ecapa_layout.h EL_CHAINPACK).
int res2net_chain_launch(sdk_ctx* ctx, const uint16_t* U, int64_t ldu, uint16_t* R, int64_t ldr, const uint16_t* const* W,
                         const uint16_t* const* Wpk, const float* const* bias, const float* const* scale, const float* const* shift,
                         int nconv, int B, int T, int dil, void* stream, bool f16) {
  SDK_REQUIRE(ctx && U && R && W && bias && scale && shift, "sdk_res2net_chain: null argument");
  SDK_REQUIRE(nconv >= 1 && nconv <= 7, "sdk_res2net_chain: nconv=%d must be in [1, 7]", nconv);
  SDK_REQUIRE(B > 0 && T > dil && T <= 208 && dil >= 1, "sdk_res2net_chain: T=%d frames (dilation %d) unsupported (dil < T <= 208)", T, dil);
  SDK_REQUIRE(ldu % 8 == 0 && ldr % 8 == 0 && ldu >= (int64_t)RS * (nconv + 1) && ldr >= (int64_t)RS * (nconv + 1), "sdk_res2net_chain: bad row strides");
  SDK_REQUIRE((uint64_t)T * (uint64_t)(ldu > ldr ? ldu : ldr) * 2u < (1ull << 32), "sdk_res2net_chain: row strides too large for 32-bit in-segment offsets");
  ChainParams p;
  p.U = (const bf16_t*)U; p.ldu = ldu; p.R = (bf16_t*)R; p.ldr = ldr; p.T = T; p.dil = dil; p.nconv = nconv;
  p.dbg = (unsigned long long*)ctx->dbg_ptr;
  bool packed = Wpk != nullptr && !ctx->no_chain_packed;
  for (int i = 0; i < 7; ++i) {
    const int k = i < nconv ? i : 0;
    SDK_REQUIRE(W[k] && bias[k] && scale[k] && shift[k], "sdk_res2net_chain: conv %d parameters missing", k);
    SDK_REQUIRE(((uintptr_t)bias[k] % 8) == 0 && ((uintptr_t)scale[k] % 8) == 0 && ((uintptr_t)shift[k] % 8) == 0, "sdk_res2net_chain: conv %d parameter arrays must be 8-byte aligned", k);
    p.W[i] = (const bf16_t*)W[k]; p.bias[i] = bias[k]; p.scale[i] = scale[k]; p.shift[i] = shift[k];
    if (packed && !Wpk[k]) packed = false;
    p.Wpk[i] = packed ? (const bf16_t*)Wpk[k] : nullptr;
  }
  ProfScope ps(ctx, stream, SDK_K_RES2NET, 2.0 * B * T * (double)RS * 3 * RS * nconv, 2.0 * 2.0 * B * T * RS * nconv);
  auto launch = [&](auto kern, int lds, int nt) -> int {
    if (sdk_lds_optin(ctx, (const void*)kern, lds)) return 1;
    hipLaunchKernelGGL(kern, dim3(B), dim3(nt), lds, (hipStream_t)stream, p);
    return 0;
  };
  // kernel choice: two segments per CU (4-wave workgroups, one image in place) for T > 112, else the 8-wave form; the weights' fragment-ordered
  // copy when present; fp16 or bf16 arithmetic
  auto pick = [&](auto f16c) -> int {
    constexpr bool F = decltype(f16c)::value;
    if (T > 112 && !ctx->no_chain_two_per_cu) {
      const int lds4 = 208 * 256 + PAR_BYTES;
      return packed ? launch(res2net_chain4_kernel<13, true, F>, lds4, RNT4) : launch(res2net_chain4_kernel<13, false, F>, lds4, RNT4);
    }
    if (T <= 112) return packed ? launch(res2net_chain_kernel<7, true, F>, 2 * 112 * 256 + PAR_BYTES, RNT) : launch(res2net_chain_kernel<7, false, F>, 2 * 112 * 256 + PAR_BYTES, RNT);
    return packed ? launch(res2net_chain_kernel<13, true, F>, 2 * 208 * 256 + PAR_BYTES, RNT) : launch(res2net_chain_kernel<13, false, F>, 2 * 208 * 256 + PAR_BYTES, RNT);
  };
  const int rc = f16 ? pick(std::true_type{}) : pick(std::false_type{});
  if (rc) return rc;
  SDK_LAUNCH_CHECK();
  return 0;
}

extern "C" int sdk_res2net_chain(sdk_ctx* ctx, const uint16_t* U, int64_t ldu, uint16_t* R, int64_t ldr, const uint16_t* const* W,
                                 const float* const* bias, const float* const* scale, const float* const* shift, int nconv,
                                 int B, int T, int dil, void* stream) {
  return res2net_chain_launch(ctx, U, ldu, R, ldr, W, nullptr, bias, scale, shift, nconv, B, T, dil, stream, ctx && ctx->precision == 2);
}
