// libsdk_hip.so: context / error plumbing and the ECAPA-TDNN forward schedule (one C call per
// batch of segments; every step is an asynchronous launch on the caller's stream, so the whole
// forward can be captured in a hipGraph by the host).
#include <stdarg.h>
#include <string.h>

#include "common.hpp"
#include "ecapa_layout.h"
#include "hp.hpp"

static thread_local char g_err[512] = "";

void sdk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sdk_last_error(void) { return g_err; }
extern "C" int sdk_abi_version(void) { return SDK_ABI_VERSION; }

extern "C" int sdk_init(int device, sdk_ctx** out) {
  SDK_REQUIRE(out, "sdk_init: out is null");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    sdk_set_error("sdk_init: no HIP device available (%s); this library has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    return 1;
  }
  SDK_REQUIRE(device >= 0 && device < n, "sdk_init: device %d out of range (0..%d)", device, n - 1);
  sdk_ctx* c = new sdk_ctx();
  c->device = device;
  if (hipGetDeviceProperties(&c->prop, device) != hipSuccess) {
    delete c;
    sdk_set_error("sdk_init: hipGetDeviceProperties failed");
    return 1;
  }
  if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
    sdk_set_error("sdk_init: device %d is %s; libsdk_hip.so carries gfx950 (MI355X) code objects only", device,
                  c->prop.gcnArchName);
    delete c;
    return 1;
  }
  c->num_cu = c->prop.multiProcessorCount;
  SDK_HIP_OK(hipSetDevice(device));
  *out = c;
  return 0;
}

int sdk_lds_optin(sdk_ctx* ctx, const void* func, int bytes) {
  for (const void* f : ctx->lds_optin)
    if (f == func) return 0;
  SDK_HIP_OK(hipSetDevice(ctx->device));
  SDK_HIP_OK(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  ctx->lds_optin.push_back(func);
  return 0;
}

// ---- device memory for hosts that bring no allocator of their own (a C / C++ host, or the torch-free Python path lite.py) -----------
extern "C" int sdk_device_malloc(sdk_ctx* ctx, size_t bytes, void** out) {
  SDK_REQUIRE(ctx && out, "sdk_device_malloc: null argument");
  *out = nullptr;
  SDK_HIP_OK(hipSetDevice(ctx->device));
  SDK_HIP_OK(hipMalloc(out, bytes ? bytes : 1));
  return 0;
}
extern "C" int sdk_device_free(sdk_ctx* ctx, void* p) {
  SDK_REQUIRE(ctx, "sdk_device_free: null context");
  if (p) SDK_HIP_OK(hipFree(p));
  return 0;
}
// kind 1 = host -> device, 2 = device -> host, 3 = device -> device; ordered on `stream`, and the call returns when the copy is complete
extern "C" int sdk_memcpy(sdk_ctx* ctx, void* dst, const void* src, size_t bytes, int kind, void* stream) {
  SDK_REQUIRE(ctx && (bytes == 0 || (dst && src)), "sdk_memcpy: null argument");
  SDK_REQUIRE(kind >= 1 && kind <= 3, "sdk_memcpy: kind=%d (1 = host to device, 2 = device to host, 3 = device to device)", kind);
  if (!bytes) return 0;
  const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  SDK_HIP_OK(hipMemcpyAsync(dst, src, bytes, k, (hipStream_t)stream));
  SDK_HIP_OK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
extern "C" int sdk_stream_synchronize(sdk_ctx* ctx, void* stream) {
  SDK_REQUIRE(ctx, "sdk_stream_synchronize: null context");
  SDK_HIP_OK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

extern "C" int sdk_shutdown(sdk_ctx* ctx) {
  delete ctx;
  return 0;
}

extern "C" int sdk_get_device_info(sdk_ctx* ctx, sdk_device_info* out) {
  SDK_REQUIRE(ctx && out, "sdk_get_device_info: null argument");
  memset(out, 0, sizeof(*out));
  out->device = ctx->device;
  out->compute_units = ctx->prop.multiProcessorCount;
  out->clock_khz = ctx->prop.clockRate;
  out->wavefront_size = ctx->prop.warpSize;
  out->hbm_bytes = ctx->prop.totalGlobalMem;
  strncpy(out->name, ctx->prop.name, sizeof(out->name) - 1);
  strncpy(out->arch, ctx->prop.gcnArchName, sizeof(out->arch) - 1);
  return 0;
}

extern "C" int sdk_set_option(sdk_ctx* ctx, const char* name, int value) {
  SDK_REQUIRE(ctx && name, "sdk_set_option: null argument");
  if (strcmp(name, "res2net_chain_fusion") == 0) { ctx->no_chain_fusion = value == 0; return 0; }
  if (strcmp(name, "res2net_packed_weights") == 0) { ctx->no_chain_packed = value == 0; return 0; }
  if (strcmp(name, "res2net_two_per_cu") == 0) { ctx->no_chain_two_per_cu = value == 0; return 0; }
  if (strcmp(name, "asp_packed_weights") == 0) { ctx->no_asp_packed = value == 0; return 0; }
  if (strcmp(name, "asp_per_segment") == 0) { ctx->no_asp_seg = value == 0; return 0; }
  if (strcmp(name, "h_kblocked") == 0) { ctx->no_h_kblocked = value == 0; return 0; }
  if (strcmp(name, "precision") == 0) {
    SDK_REQUIRE(value >= 0 && value <= 2, "sdk_set_option: precision must be 0 (bf16 operands), 1 (fp16 hi+lo planes) or 2 (one fp16 plane), got %d", value);
    ctx->precision = value;
    return 0;
  }
  if (strcmp(name, "gemm_variant") == 0) return sdk_set_gemm_variant(value);
  if (strcmp(name, "affinity_fast_path") == 0) { ctx->aff_fast = value; return 0; }
  if (strcmp(name, "affinity_variant") == 0) { ctx->aff_variant = value; return 0; }
  if (strcmp(name, "affinity_boundary_penalty") == 0) { ctx->aff_boundary_pen = value < 0 ? 0 : value; return 0; }
  if (strcmp(name, "matvec_variant") == 0) { ctx->matvec_variant = value; return 0; }
  if (strcmp(name, "hp_gemm_variant") == 0) { ctx->hp_gemm_variant = value; return 0; }
  if (strcmp(name, "affinity_whole_groups") == 0) return 0;      // (round-3 knob, measured behind and removed in round 5: accepted and ignored)
  if (strcmp(name, "chol_pivot_rtol_ppb") == 0) { ctx->chol_pivot_rtol_ppb = value < 0 ? 0 : value; return 0; }
  if (strcmp(name, "chol_shift_ppb") == 0) { ctx->chol_shift_ppb = value < 0 ? 0 : value; return 0; }
  sdk_set_error("sdk_set_option: unknown option '%s'", name);
  return 2;
}

extern "C" int sdk_debug_set_ptr(sdk_ctx* ctx, const char* name, void* p) {
  SDK_REQUIRE(ctx && name, "sdk_debug_set_ptr: null argument");
  if (strcmp(name, "stamps") == 0) { ctx->dbg_ptr = p; return 0; }
  if (strcmp(name, "gemm_clock") == 0) { ctx->gemm_clk_ptr = p; return 0; }
  if (strcmp(name, "gemm_stamps") == 0) { ctx->gemm_stamps_ptr = p; return 0; }
  sdk_set_error("sdk_debug_set_ptr: unknown name '%s'", name);
  return 2;
}

extern "C" int sdk_profile_begin(sdk_ctx* ctx) {
  SDK_REQUIRE(ctx, "sdk_profile_begin: null ctx");
  for (auto& r : ctx->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  ctx->prof.clear();
  ctx->prof_on = true;
  return 0;
}

extern "C" int sdk_profile_end(sdk_ctx* ctx, sdk_profile_report* out) {
  SDK_REQUIRE(ctx && out, "sdk_profile_end: null argument");
  ctx->prof_on = false;
  memset(out, 0, sizeof(*out));
  SDK_HIP_OK(hipDeviceSynchronize());
  for (auto& r : ctx->prof) {
    float ms = 0.f;
    SDK_HIP_OK(hipEventElapsedTime(&ms, r.a, r.b));
    if (r.family >= 0 && r.family < SDK_K_COUNT) {
      out->launches[r.family] += 1;
      out->ms[r.family] += ms;
      out->flops[r.family] += r.flops;
      out->bytes[r.family] += r.bytes;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  ctx->prof.clear();
  return 0;
}

// ------------------------------------------------------------------------------ ECAPA forward
namespace {

inline size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }

struct FwdWs {
  uint16_t *X0, *U, *R, *Z, *Sa, *Sb, *CAT, *H, *AH;
  float *ctx, *ubias, *logits, *pooled, *se, *stats, *mean;
};

size_t fwd_layout(const sdk_ecapa_desc* d, int B, int T, char* base, FwdWs* w) {
  const size_t M = (size_t)B * T, C = d->channels, Cm = d->mfa_channels, S = d->sub_channels, A = d->attn_channels;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += a256(bytes); return p; };
  char* x0 = take(M * C * 2);
  char* u = take(M * C * 2);
  char* r = take(M * C * 2);
  char* z = take(M * C * 2);
  char* sa = take(M * S * 2);
  char* sb = take(M * S * 2);
  char* cat = take(M * Cm * 2);
  char* h = take(M * Cm * 2);
  char* ah = take(M * A * 2);
  char* cx = take((size_t)B * 2 * Cm * 4);
  char* ub = take((size_t)B * A * 4);
  // the [M, Cm] fp32 attention logits exist in HBM only on the unfused pooling path (37 % of the workspace otherwise)
  const bool fused_asp = A == 128 && T <= sdk_asp_fused_max_frames();
  char* lg = take(fused_asp ? 256 : M * Cm * 4);
  char* po = take((size_t)B * 2 * Cm * 4);
  char* se = take(sdk_se_workspace_bytes(B, (int)C, d->se_channels));
  char* stp = take(sdk_conv_gemm_stats_bytes((int)M, (int)Cm, 2));
  char* mn = take((size_t)B * C * 4);
  if (w) { w->se = (float*)se; w->stats = (float*)stp; w->mean = (float*)mn; }
  if (w) {
    w->X0 = (uint16_t*)x0; w->U = (uint16_t*)u; w->R = (uint16_t*)r; w->Z = (uint16_t*)z;
    w->Sa = (uint16_t*)sa; w->Sb = (uint16_t*)sb; w->CAT = (uint16_t*)cat; w->H = (uint16_t*)h; w->AH = (uint16_t*)ah;
    w->ctx = (float*)cx; w->ubias = (float*)ub; w->logits = (float*)lg; w->pooled = (float*)po;
  }
  return off;
}

// ---- precise mode: every activation a pair of fp16 planes [M, 2 C] (hi | lo), logits fp32
struct FwdWsHp {
  uint16_t *X0, *U, *R, *Z, *Sa, *Sb, *CAT, *H, *AH;
  float *ctx, *ubias, *logits, *pooled, *mean, *hid, *gate;
};

size_t fwd_layout_hp(const sdk_ecapa_desc* d, int B, int T, char* base, FwdWsHp* w) {
  const size_t M = (size_t)B * T, C = d->channels, Cm = d->mfa_channels, S = d->sub_channels, A = d->attn_channels;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += a256(bytes); return p; };
  char* x0 = take(M * C * 4);
  char* u = take(M * C * 4);
  char* r = take(M * C * 4);
  char* z = take(M * C * 4);
  char* sa = take(M * S * 4);
  char* sb = take(M * S * 4);
  char* cat = take(M * Cm * 4);
  char* h = take(M * Cm * 4);
  char* ah = take(M * A * 4);
  char* cx = take((size_t)B * 2 * Cm * 4);
  char* ub = take((size_t)B * A * 4);
  char* lg = take(M * Cm * 4);
  char* po = take((size_t)B * 2 * Cm * 4);
  char* mn = take((size_t)B * C * 4);
  char* hd = take((size_t)B * d->se_channels * 4);
  char* gt = take((size_t)B * C * 4);
  if (w) {
    w->X0 = (uint16_t*)x0; w->U = (uint16_t*)u; w->R = (uint16_t*)r; w->Z = (uint16_t*)z; w->Sa = (uint16_t*)sa; w->Sb = (uint16_t*)sb;
    w->CAT = (uint16_t*)cat; w->H = (uint16_t*)h; w->AH = (uint16_t*)ah; w->ctx = (float*)cx; w->ubias = (float*)ub; w->logits = (float*)lg;
    w->pooled = (float*)po; w->mean = (float*)mn; w->hid = (float*)hd; w->gate = (float*)gt;
  }
  return off;
}

int check_desc(const sdk_ecapa_desc* d) {
  SDK_REQUIRE(d, "ecapa desc is null");
  SDK_REQUIRE(d->n_blocks >= 1 && d->n_blocks <= 4, "ecapa desc: n_blocks=%d", d->n_blocks);
  SDK_REQUIRE(d->channels % 128 == 0 && d->mfa_channels == d->n_blocks * d->channels, "ecapa desc: channels=%d mfa=%d", d->channels, d->mfa_channels);
  SDK_REQUIRE(d->scale >= 2 && d->sub_channels * d->scale == d->channels && d->sub_channels % 128 == 0, "ecapa desc: res2net scale=%d sub=%d", d->scale, d->sub_channels);
  SDK_REQUIRE(d->attn_channels % 128 == 0 && d->n_mels_padded % (d->precision == 1 ? 32 : 64) == 0, "ecapa desc: attn=%d mels=%d", d->attn_channels, d->n_mels_padded);
  SDK_REQUIRE((d->kernel0 & 1) == 1, "ecapa desc: kernel0=%d must be odd", d->kernel0);
  SDK_REQUIRE(d->precision >= 0 && d->precision <= 2, "ecapa desc: precision=%d", d->precision);
  return 0;
}

// The forward in precise mode (hp.hip): the same layer sequence on fp16 hi+lo planes, every GEMM as three fp16 MFMAs per product,
// no fusion (the Res2Net chain as seven launches with the running sum in the epilogue, SE as mean sweep + two FCs + apply sweep,
// attention logits in fp32 through HBM): this mode buys accuracy, the default mode is the fast one.
int ecapa_forward_hp(sdk_ctx* ctx, const void* wblob, const sdk_ecapa_desc* d, const uint16_t* feats, int ldf, int B, int T, void* ws,
                     float* emb, void* stream) {
  FwdWsHp w;
  fwd_layout_hp(d, B, T, (char*)ws, &w);
  const char* wb = (const char*)wblob;
  auto P16 = [&](int slot) -> const uint16_t* { return d->off[slot] < 0 ? nullptr : (const uint16_t*)(wb + d->off[slot]); };
  auto P32 = [&](int slot) -> const float* { return d->off[slot] < 0 ? nullptr : (const float*)(wb + d->off[slot]); };
  const int M = B * T, C = d->channels, Cm = d->mfa_channels, S = d->sub_channels, A = d->attn_channels;
  const int flo = ldf >> 1;
  SDK_REQUIRE(ldf % 16 == 0 && flo >= d->n_mels_padded, "sdk_ecapa_forward: precise mode takes feature planes, ldf=%d must be >= 2 x %d", ldf, d->n_mels_padded);

  auto tdnn = [&](const uint16_t* Ain, int64_t lda, int64_t a_lo, int Cin, int taps, int dil, int slot, int N, uint16_t* Cout, int64_t ldc, int64_t c_lo,
                  const uint16_t* X2, int64_t ldx2, int64_t x2_lo, uint16_t* Sout, int64_t lds, int64_t s_lo) -> int {
    sdk_conv_gemm_hp_args g;
    memset(&g, 0, sizeof(g));
    g.A = Ain; g.lda = lda; g.a_lo = a_lo; g.W = P16(slot + EL_W); g.C = Cout; g.ldc = ldc; g.c_lo = c_lo;
    g.bias = P32(slot + EL_B); g.scale = P32(slot + EL_SCALE); g.shift = P32(slot + EL_SHIFT);
    g.X2 = X2; g.ldx2 = ldx2; g.x2_lo = x2_lo; g.S = Sout; g.lds = lds; g.s_lo = s_lo;
    g.M = M; g.N = N; g.Cin = Cin; g.taps = taps; g.dil = dil; g.T = T; g.flags = SDK_GEMM_RELU;
    SDK_REQUIRE(g.W && g.bias && g.scale && g.shift, "sdk_ecapa_forward: weight slot %d missing", slot);
    return sdk_conv_gemm_hp(ctx, &g, stream);
  };
  hipStream_t st = (hipStream_t)stream;

  if (int rc = tdnn(feats, ldf, flo, d->n_mels_padded, d->kernel0, 1, EL_BLK0, C, w.X0, 2 * C, C, nullptr, 0, 0, nullptr, 0, 0)) return rc;
  const uint16_t* xin = w.X0;
  int64_t ldx = 2 * C, xlo = C;
  for (int i = 1; i <= d->n_blocks; ++i) {
    const int base = EL_BLOCK_BASE(i), dil = d->dilation[i - 1];
    if (int rc = tdnn(xin, ldx, xlo, C, 1, 1, base + EL_TDNN1, C, w.U, 2 * C, C, nullptr, 0, 0, nullptr, 0, 0)) return rc;
    {   // Res2Net chunk 0 passes through: both planes
      ProfScope ps(ctx, stream, SDK_K_COPY, 0.0, 2.0 * M * S * 4);
      SDK_HIP_OK(hipMemcpy2DAsync(w.R, (size_t)C * 4, w.U, (size_t)C * 4, (size_t)S * 2, (size_t)M, hipMemcpyDeviceToDevice, st));
      SDK_HIP_OK(hipMemcpy2DAsync(w.R + C, (size_t)C * 4, w.U + C, (size_t)C * 4, (size_t)S * 2, (size_t)M, hipMemcpyDeviceToDevice, st));
    }
    for (int j = 0; j < d->scale - 1; ++j) {
      const uint16_t* Ain = j == 0 ? w.U + S : ((j & 1) ? w.Sa : w.Sb);
      const int64_t lda = j == 0 ? 2 * C : 2 * S, alo = j == 0 ? C : S;
      const bool more = j + 1 < d->scale - 1;
      uint16_t* Sout = more ? ((j & 1) ? w.Sb : w.Sa) : nullptr;
      const uint16_t* X2 = more ? w.U + (int64_t)S * (j + 2) : nullptr;
      if (int rc = tdnn(Ain, lda, alo, S, 3, dil, base + EL_RES2NET(j), S, w.R + (int64_t)S * (j + 1), 2 * C, C, X2, 2 * C, C, Sout, 2 * S, S)) return rc;
    }
    if (int rc = tdnn(w.R, 2 * C, C, C, 1, 1, base + EL_TDNN2, C, w.Z, 2 * C, C, nullptr, 0, 0, nullptr, 0, 0)) return rc;
    // squeeze-excitation: mean sweep, two per-utterance FCs (fp32 matrix pipe), apply sweep
    if (int rc = hp_seg_mean(ctx, w.Z, 2 * C, C, B, T, C, w.mean, stream)) return rc;
    if (int rc = sdk_rows_fc(ctx, w.mean, C, nullptr, nullptr, P32(base + EL_SE_W1T), P32(base + EL_SE_B1), w.hid, d->se_channels, B, C, d->se_channels, 1, stream)) return rc;
    if (int rc = sdk_rows_fc(ctx, w.hid, d->se_channels, nullptr, nullptr, P32(base + EL_SE_W2T), P32(base + EL_SE_B2), w.gate, C, B, d->se_channels, C, 2, stream)) return rc;
    uint16_t* slab = w.CAT + (int64_t)C * (i - 1);
    if (int rc = hp_se_apply(ctx, w.Z, 2 * C, C, xin, ldx, xlo, w.gate, slab, 2 * Cm, Cm, B, T, C, stream)) return rc;
    xin = slab;
    ldx = 2 * Cm;
    xlo = Cm;
  }
  const int tb = EL_TAIL_BASE(d->n_blocks);
  if (int rc = tdnn(w.CAT, 2 * Cm, Cm, Cm, 1, 1, tb + EL_MFA, Cm, w.H, 2 * Cm, Cm, nullptr, 0, 0, nullptr, 0, 0)) return rc;
  if (int rc = hp_asp_stats(ctx, w.H, 2 * Cm, Cm, B, T, Cm, w.ctx, stream)) return rc;
  if (int rc = sdk_rows_fc(ctx, w.ctx, 2 * Cm, nullptr, nullptr, P32(tb + EL_ASP_WMS_T), P32(tb + EL_ASP_B), w.ubias, A, B, 2 * Cm, A, 0, stream)) return rc;
  {
    sdk_conv_gemm_hp_args g;
    memset(&g, 0, sizeof(g));
    g.A = w.H; g.lda = 2 * Cm; g.a_lo = Cm; g.W = P16(tb + EL_ASP_WH); g.C = w.AH; g.ldc = 2 * A; g.c_lo = A;
    g.ubias = w.ubias; g.ldub = A; g.scale = P32(tb + EL_ASP_SCALE); g.shift = P32(tb + EL_ASP_SHIFT);
    g.M = M; g.N = A; g.Cin = Cm; g.taps = 1; g.dil = 1; g.T = T; g.flags = SDK_GEMM_RELU | SDK_GEMM_TANH;
    if (int rc = sdk_conv_gemm_hp(ctx, &g, stream)) return rc;
    memset(&g, 0, sizeof(g));
    g.A = w.AH; g.lda = 2 * A; g.a_lo = A; g.W = P16(tb + EL_ASP_W2); g.C32 = w.logits; g.ldc32 = Cm; g.bias = P32(tb + EL_ASP_B2);
    g.M = M; g.N = Cm; g.Cin = A; g.taps = 1; g.dil = 1; g.T = T; g.flags = 0;
    if (int rc = sdk_conv_gemm_hp(ctx, &g, stream)) return rc;
  }
  if (int rc = hp_asp_pool(ctx, w.logits, Cm, w.H, 2 * Cm, Cm, B, T, Cm, w.pooled, stream)) return rc;
  return sdk_rows_fc(ctx, w.pooled, 2 * Cm, P32(tb + EL_ASPBN_SCALE), P32(tb + EL_ASPBN_SHIFT), P32(tb + EL_FC_WT), P32(tb + EL_FC_B), emb,
                     d->embed_dim, B, 2 * Cm, d->embed_dim, 0, stream);
}

}  // namespace

extern "C" size_t sdk_ecapa_workspace_bytes(const sdk_ecapa_desc* d, int B, int T) {
  if (!d || B <= 0 || T <= 0) return 0;
  if (d->precision == 1) return fwd_layout_hp(d, B, T, nullptr, nullptr);
  return fwd_layout(d, B, T, nullptr, nullptr);
}

// Calibration slots (sdk_ecapa_forward_calib): per-segment mean | std (sdk_asp_stats layout [B, 2 C_l]) of the INPUT of every bf16 GEMM layer whose
// weight rounding the host corrects in the bias (weights_pack.bias_corrections): per block tdnn1, the Res2Net convs, tdnn2; then MFA, ASP hidden.
static size_t calib_offset(const sdk_ecapa_desc* d, int B, int block, int layer) {   // block < n_blocks: layer 0 = tdnn1, 1..scale-1 = Res2Net conv
  const size_t C = d->channels, S = d->sub_channels, Cm = d->mfa_channels, nr = d->scale - 1;   // layer - 1, scale = tdnn2; block == n_blocks: 0 = MFA,
  const size_t per_block = 2 * C + nr * 2 * S + 2 * C;                                            // 1 = ASP hidden, 2 = end
  size_t off;
  if (block < d->n_blocks) {
    off = (size_t)block * per_block;
    if (layer >= 1) off += 2 * C + (size_t)(layer - 1) * 2 * S;        // (layer == scale: behind the nr Res2Net slots)
  } else {
    off = (size_t)d->n_blocks * per_block + (size_t)layer * 2 * Cm;
  }
  return off * (size_t)B;
}

extern "C" size_t sdk_ecapa_calib_floats(const sdk_ecapa_desc* d, int B) {
  if (!d || B <= 0) return 0;
  return calib_offset(d, B, d->n_blocks, 2);
}

static int ecapa_forward_impl(sdk_ctx* ctx, const void* wblob, const sdk_ecapa_desc* d, const uint16_t* feats, int ldf,
                              int B, int T, void* ws, size_t ws_bytes, float* emb, float* calib, void* stream);

extern "C" int sdk_ecapa_forward(sdk_ctx* ctx, const void* wblob, const sdk_ecapa_desc* d, const uint16_t* feats, int ldf,
                                 int B, int T, void* ws, size_t ws_bytes, float* emb, void* stream) {
  return ecapa_forward_impl(ctx, wblob, d, feats, ldf, B, T, ws, ws_bytes, emb, nullptr, stream);
}

extern "C" int sdk_ecapa_forward_calib(sdk_ctx* ctx, const void* wblob, const sdk_ecapa_desc* d, const uint16_t* feats, int ldf,
                                       int B, int T, void* ws, size_t ws_bytes, float* emb, float* calib, void* stream) {
  SDK_REQUIRE(calib && d && d->precision != 1, "sdk_ecapa_forward_calib: calib is null or the blob is a precise-mode blob (calibration serves the single-plane modes 0 and 2)");
  return ecapa_forward_impl(ctx, wblob, d, feats, ldf, B, T, ws, ws_bytes, emb, calib, stream);
}

static int ecapa_forward_impl(sdk_ctx* ctx, const void* wblob, const sdk_ecapa_desc* d, const uint16_t* feats, int ldf,
                              int B, int T, void* ws, size_t ws_bytes, float* emb, float* calib, void* stream) {
  SDK_REQUIRE(ctx && wblob && feats && ws && emb, "sdk_ecapa_forward: null argument");
  if (int rc = check_desc(d)) return rc;
  SDK_REQUIRE(B > 0 && T > 0, "sdk_ecapa_forward: empty batch (B=%d T=%d)", B, T);
  SDK_REQUIRE((int64_t)B * T < (1ll << 31), "sdk_ecapa_forward: B*T overflows int32; split the batch");
  SDK_REQUIRE(ldf >= d->n_mels_padded && ldf % 8 == 0, "sdk_ecapa_forward: ldf=%d < padded mel width %d", ldf, d->n_mels_padded);
  // The numerical contract is the DESCRIPTOR's (per call): features must be in that format (sdk_fbank_fmt with the same precision).  The context's
  // "precision" option plays no part here since round 5 - two engines with different contracts on one device share no mutable state.
  int maxhalo = d->kernel0 / 2;
  for (int i = 0; i < d->n_blocks; ++i) maxhalo = d->dilation[i] > maxhalo ? d->dilation[i] : maxhalo;
  SDK_REQUIRE(T > maxhalo, "sdk_ecapa_forward: segments of %d frames are shorter than the receptive halo %d", T, maxhalo);
  SDK_REQUIRE(ws_bytes >= sdk_ecapa_workspace_bytes(d, B, T), "sdk_ecapa_forward: workspace too small (%zu < %zu)", ws_bytes,
              sdk_ecapa_workspace_bytes(d, B, T));
  SDK_REQUIRE(((uintptr_t)ws % 256) == 0 && ((uintptr_t)wblob % 256) == 0, "sdk_ecapa_forward: ws/wblob must be 256-byte aligned");

  {  // every slot the schedule dereferences must be present in the blob
    const int tbs = EL_TAIL_BASE(d->n_blocks);
    for (int i = 1; i <= d->n_blocks; ++i)
      for (int s = EL_SE_W1T; s <= EL_SE_B2; ++s)
        SDK_REQUIRE(d->off[EL_BLOCK_BASE(i) + s] >= 0, "sdk_ecapa_forward: SE weight slot %d of block %d missing", s, i);
    for (int s = EL_ASP_WH; s <= EL_FC_B; ++s) SDK_REQUIRE(d->off[tbs + s] >= 0, "sdk_ecapa_forward: tail weight slot %d missing", s);
  }
  if (d->precision == 1) return ecapa_forward_hp(ctx, wblob, d, feats, ldf, B, T, ws, emb, stream);
  FwdWs w;
  fwd_layout(d, B, T, (char*)ws, &w);
  const char* wb = (const char*)wblob;
  auto P16 = [&](int slot) -> const uint16_t* { return d->off[slot] < 0 ? nullptr : (const uint16_t*)(wb + d->off[slot]); };
  auto P32 = [&](int slot) -> const float* { return d->off[slot] < 0 ? nullptr : (const float*)(wb + d->off[slot]); };
  const int M = B * T, C = d->channels, Cm = d->mfa_channels, S = d->sub_channels, A = d->attn_channels;
  hipStream_t st = (hipStream_t)stream;
  // precision 2 (round 5): the same schedule with every 2-byte tensor - features, weights, activations - in fp16 instead of bf16 (11 significand
  // bits: 7 x closer to the fp32 model at one MFMA per product, profiles/r05_fp16_decision.txt); every stage takes the
  // format per call: the GEMMs from their flag, the sweeps and the two fused launches from an argument of their internal entry
  const bool f16 = d->precision == 2;
  const uint32_t fmt = f16 ? SDK_GEMM_F16 : 0u;

  auto tdnn = [&](const uint16_t* Ain, int64_t lda, int Cin, int taps, int dil, int slot, int N, uint16_t* Cout, int64_t ldc,
                  const uint16_t* X2, int64_t ldx2, uint16_t* Sout, int64_t lds, int stats_mode = 0, uint32_t layout = 0) -> int {
    sdk_conv_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = Ain; g.lda = lda; g.W = P16(slot + EL_W); g.C = Cout; g.ldc = ldc;
    g.bias = P32(slot + EL_B); g.scale = P32(slot + EL_SCALE); g.shift = P32(slot + EL_SHIFT);
    g.X2 = X2; g.ldx2 = ldx2; g.S = Sout; g.lds = lds;
    g.M = M; g.N = N; g.Cin = Cin; g.taps = taps; g.dil = dil; g.T = T; g.flags = SDK_GEMM_RELU | layout | fmt;
    g.stats_mode = stats_mode; g.stats_part = stats_mode ? w.stats : nullptr;
    SDK_REQUIRE(g.W && g.bias && g.scale && g.shift, "sdk_ecapa_forward: weight slot %d missing", slot);
    return sdk_conv_gemm(ctx, &g, stream);
  };

  // blk0: k5 conv over the mel channels; with a packed weight slot the five taps share K (5 x 80 -> 448: 7 K-steps instead of 10)
  if (d->blk0_tap_pack > 0) {
    sdk_conv_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = feats; g.lda = ldf; g.W = P16(EL_BLK0 + EL_W); g.C = w.X0; g.ldc = C;
    g.bias = P32(EL_BLK0 + EL_B); g.scale = P32(EL_BLK0 + EL_SCALE); g.shift = P32(EL_BLK0 + EL_SHIFT);
    g.M = M; g.N = C; g.Cin = d->blk0_tap_pack; g.taps = d->kernel0; g.dil = 1; g.T = T; g.flags = SDK_GEMM_RELU | fmt; g.tap_pack = d->blk0_tap_pack;
    SDK_REQUIRE(g.W && g.bias && g.scale && g.shift && d->blk0_tap_pack <= ldf, "sdk_ecapa_forward: blk0 weight slots missing or tap_pack > ldf");
    if (int rc = sdk_conv_gemm(ctx, &g, stream)) return rc;
  } else if (int rc = tdnn(feats, ldf, d->n_mels_padded, d->kernel0, 1, EL_BLK0, C, w.X0, C, nullptr, 0, nullptr, 0)) return rc;

  const uint16_t* xin = w.X0;
  int64_t ldx = C;
  for (int i = 1; i <= d->n_blocks; ++i) {
    const int base = EL_BLOCK_BASE(i), dil = d->dilation[i - 1];
    if (calib)
      if (int rc = asp_stats_impl(ctx, xin, ldx, B, T, C, calib + calib_offset(d, B, i - 1, 0), stream, f16)) return rc;
    if (int rc = tdnn(xin, ldx, C, 1, 1, base + EL_TDNN1, C, w.U, C, nullptr, 0, nullptr, 0)) return rc;
    // Res2Net: chunk 0 passes through, chunk c>=1 = TDNN(chunk c + y_{c-1})
    const uint16_t* r2out = w.R;
    if (S == 128 && d->scale - 1 <= 7 && T <= sdk_res2net_chain_max_frames() && !ctx->no_chain_fusion && !calib) {
      // the seven dependent convolutions in ONE launch, the running tile resident in LDS per segment.  The chain runs
      // IN PLACE on the tdnn1 output: a workgroup has read u_c (into LDS / registers) before it writes y_c over it, and
      // segments do not overlap - so chunk 0 needs no copy
      const uint16_t* Wp[7]; const uint16_t* Wk[7]; const float* bp[7]; const float* sp[7]; const float* tp[7];
      for (int j = 0; j < d->scale - 1; ++j) {
        const int slot = base + EL_RES2NET(j);
        Wp[j] = P16(slot + EL_W); bp[j] = P32(slot + EL_B); sp[j] = P32(slot + EL_SCALE); tp[j] = P32(slot + EL_SHIFT);
        Wk[j] = (i <= 4 && d->off[EL_CHAINPACK(i, j)] >= 0) ? P16(EL_CHAINPACK(i, j)) : nullptr;   // optional fragment-ordered copy
      }
      if (int rc = res2net_chain_launch(ctx, w.U, C, w.U, C, Wp, Wk, bp, sp, tp, d->scale - 1, B, T, dil, stream, f16)) return rc;
      r2out = w.U;
    } else {
      // separate launches: the running sum is produced by the previous conv's epilogue (S output), ping-ponging
      // between two [M, S] buffers
      {
        ProfScope ps(ctx, stream, SDK_K_COPY, 0.0, 2.0 * M * S * 2);
        SDK_HIP_OK(hipMemcpy2DAsync(w.R, (size_t)C * 2, w.U, (size_t)C * 2, (size_t)S * 2, (size_t)M, hipMemcpyDeviceToDevice, st));
      }
      for (int j = 0; j < d->scale - 1; ++j) {
        const uint16_t* Ain = j == 0 ? w.U + S : ((j & 1) ? w.Sa : w.Sb);
        const int64_t lda = j == 0 ? C : S;
        const bool more = j + 1 < d->scale - 1;
        uint16_t* Sout = more ? ((j & 1) ? w.Sb : w.Sa) : nullptr;
        const uint16_t* X2 = more ? w.U + (int64_t)S * (j + 2) : nullptr;
        if (calib)
          if (int rc = asp_stats_impl(ctx, Ain, lda, B, T, S, calib + calib_offset(d, B, i - 1, 1 + j), stream, f16)) return rc;
        if (int rc = tdnn(Ain, lda, S, 3, dil, base + EL_RES2NET(j), S, w.R + (int64_t)S * (j + 1), C, X2, C, Sout, S)) return rc;
      }
    }
    if (calib)
      if (int rc = asp_stats_impl(ctx, r2out, C, B, T, C, calib + calib_offset(d, B, i - 1, d->scale), stream, f16)) return rc;
    // the SE squeeze (per-segment channel means of z) comes out of the tdnn2 epilogue where the shape allows it
    const bool fuse_se = sdk_conv_gemm_stats_fusable(M, C, T) != 0;
    if (int rc = tdnn(r2out, C, C, 1, 1, base + EL_TDNN2, C, w.Z, C, nullptr, 0, nullptr, 0, fuse_se ? 1 : 0)) return rc;
    if (fuse_se)
      if (int rc = sdk_colstats_finish(ctx, w.stats, M, C, T, 1, w.mean, stream)) return rc;
    uint16_t* slab = w.CAT + (int64_t)C * (i - 1);
    if (int rc = se_gate_residual_impl(ctx, w.Z, C, xin, ldx, P32(base + EL_SE_W1T), P32(base + EL_SE_B1), P32(base + EL_SE_W2T),
                                       P32(base + EL_SE_B2), slab, Cm, B, T, C, d->se_channels, fuse_se ? w.mean : nullptr, w.se,
                                       sdk_se_workspace_bytes(B, C, d->se_channels), stream, f16)) return rc;
    xin = slab;
    ldx = Cm;
  }

  const int tb = EL_TAIL_BASE(d->n_blocks);
  // attentive statistics pooling with global context; the context (mean | std of h over frames) comes out of
  // the MFA epilogue where the shape allows it, else from a separate sweep of h
  const bool fuse_ctx = sdk_conv_gemm_stats_fusable(M, Cm, T) != 0;
  if (calib)
    if (int rc = asp_stats_impl(ctx, w.CAT, Cm, B, T, Cm, calib + calib_offset(d, B, d->n_blocks, 0), stream, f16)) return rc;
  // h [M, Cm] is the widest activation and is read twice, both times in 64- / 32-channel pieces of its 6-KB rows (the skinny attention-hidden GEMM,
  // the per-segment ASP slabs): where those two are its only readers it is written K-BLOCKED, [Cm / 64][M][64] (sdk_hip.h SDK_GEMM_C_KBLOCKED)
  const bool h_kb = !ctx->no_h_kblocked && !calib && fuse_ctx && A == 128 && sdk_asp_kblocked_ok(ctx, T, Cm) != 0;
  if (int rc = tdnn(w.CAT, Cm, Cm, 1, 1, tb + EL_MFA, Cm, w.H, Cm, nullptr, 0, nullptr, 0, fuse_ctx ? 2 : 0, h_kb ? SDK_GEMM_C_KBLOCKED : 0)) return rc;
  if (calib)
    if (int rc = asp_stats_impl(ctx, w.H, Cm, B, T, Cm, calib + calib_offset(d, B, d->n_blocks, 1), stream, f16)) return rc;
  if (fuse_ctx) {
    if (int rc = sdk_colstats_finish(ctx, w.stats, M, Cm, T, 2, w.ctx, stream)) return rc;
  } else {
    if (int rc = asp_stats_impl(ctx, w.H, Cm, B, T, Cm, w.ctx, stream, f16)) return rc;
  }
  if (int rc = sdk_rows_fc(ctx, w.ctx, 2 * Cm, nullptr, nullptr, P32(tb + EL_ASP_WMS_T), P32(tb + EL_ASP_B), w.ubias, A, B, 2 * Cm, A, 0, stream)) return rc;
  {
    sdk_conv_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = w.H; g.lda = Cm; g.W = P16(tb + EL_ASP_WH); g.C = w.AH; g.ldc = A;
    g.ubias = w.ubias; g.ldub = A; g.scale = P32(tb + EL_ASP_SCALE); g.shift = P32(tb + EL_ASP_SHIFT);
    g.M = M; g.N = A; g.Cin = Cm; g.taps = 1; g.dil = 1; g.T = T; g.flags = SDK_GEMM_RELU | SDK_GEMM_TANH | (h_kb ? SDK_GEMM_A_KBLOCKED : 0) | fmt;
    if (int rc = sdk_conv_gemm(ctx, &g, stream)) return rc;
    if (A == 128 && T <= sdk_asp_fused_max_frames()) {
      // logits GEMM + softmax pooling fused: no [M, Cm] fp32 logits round trip through HBM
      if (int rc = asp_fused_launch(ctx, w.AH, A, P16(tb + EL_ASP_W2), P16(tb + EL_ASP_W2PACK), P32(tb + EL_ASP_B2), w.H, Cm, B, T, Cm, A, w.pooled, stream, h_kb, f16)) return rc;
    } else {
      memset(&g, 0, sizeof(g));
      g.A = w.AH; g.lda = A; g.W = P16(tb + EL_ASP_W2); g.C32 = w.logits; g.ldc32 = Cm; g.bias = P32(tb + EL_ASP_B2);
      g.M = M; g.N = Cm; g.Cin = A; g.taps = 1; g.dil = 1; g.T = T; g.flags = fmt;
      if (int rc = sdk_conv_gemm(ctx, &g, stream)) return rc;
      if (int rc = asp_pool_impl(ctx, w.logits, Cm, w.H, Cm, B, T, Cm, w.pooled, stream, f16)) return rc;
    }
  }
  return sdk_rows_fc(ctx, w.pooled, 2 * Cm, P32(tb + EL_ASPBN_SCALE), P32(tb + EL_ASPBN_SHIFT), P32(tb + EL_FC_WT),
                     P32(tb + EL_FC_B), emb, d->embed_dim, B, 2 * Cm, d->embed_dim, 0, stream);
}

// ------------------------------------------------------------------------------ x-vector forward
// The plain TDNN embedding extractor (Snyder et al. 2018) composed from the same pieces: every frame layer is one sdk_conv_gemm
// (conv -> ReLU -> folded BN -> bf16), the pooled statistics and the embedding layer are fp32.
namespace {
int check_xdesc(const sdk_xvector_desc* d) {
  SDK_REQUIRE(d, "xvector desc is null");
  SDK_REQUIRE(d->n_frame_layers >= 1 && d->n_frame_layers <= 8, "xvector desc: n_frame_layers=%d", d->n_frame_layers);
  SDK_REQUIRE(d->embed_dim > 0 && d->n_feats > 0, "xvector desc: embed_dim=%d n_feats=%d", d->embed_dim, d->n_feats);
  for (int l = 0; l < d->n_frame_layers; ++l) {
    SDK_REQUIRE((d->kernel[l] & 1) == 1 && d->dilation[l] >= 1 && d->cout[l] % 128 == 0 && d->cout[l] > 0, "xvector desc: layer %d kernel=%d dil=%d cout=%d", l,
                d->kernel[l], d->dilation[l], d->cout[l]);
    if (l > 0) SDK_REQUIRE(d->cin[l] == d->cout[l - 1] && d->cin[l] % 64 == 0, "xvector desc: layer %d cin=%d does not follow cout=%d", l, d->cin[l], d->cout[l - 1]);
    for (int q = 0; q < 4; ++q) SDK_REQUIRE(d->off[4 * l + q] >= 0, "xvector desc: slot %d of layer %d missing", q, l);
  }
  SDK_REQUIRE(d->off[62] >= -1 && d->off[62] <= 2, "xvector desc: off[62] (precision of the blob) must be -1 / 0 (bf16 operands), 1 (fp16 hi + lo planes) or 2 (one fp16 plane)");
  if (d->off[62] == 1)
    SDK_REQUIRE(d->cin[0] == d->n_feats && d->n_feats % 32 == 0 && d->first_tap_pack == 0, "xvector desc: a precise-mode blob reads %d feature channels (a multiple of 32, no tap packing)", d->n_feats);
  else
  SDK_REQUIRE(d->cin[0] == d->n_feats && (d->first_tap_pack == 0 ? d->n_feats % 64 == 0 : (d->first_tap_pack == d->n_feats && d->n_feats % 8 == 0 && d->kernel[0] > 1)),
              "xvector desc: first layer over %d features needs them to be a multiple of 64, or its taps packed (first_tap_pack = n_feats, a multiple of 8)", d->n_feats);
  SDK_REQUIRE(d->off[60] >= 0 && d->off[61] >= 0, "xvector desc: embedding layer slots missing");
  return 0;
}
size_t xv_layout(const sdk_xvector_desc* d, int B, int T, char* base, uint16_t** buf0, uint16_t** buf1, float** stats) {
  size_t cmax = 0;
  for (int l = 0; l < d->n_frame_layers; ++l) cmax = (size_t)d->cout[l] > cmax ? (size_t)d->cout[l] : cmax;
  const size_t M = (size_t)B * T;
  const size_t esz = d->off[62] == 1 ? 4 : 2;                  // precise blob: activations travel as fp16 hi + lo planes (4 bytes per element)
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += a256(bytes); return p; };
  char* a = take(M * cmax * esz);
  char* b = take(M * cmax * esz);
  char* c = take((size_t)B * 2 * d->cout[d->n_frame_layers - 1] * 4);
  if (buf0) { *buf0 = (uint16_t*)a; *buf1 = (uint16_t*)b; *stats = (float*)c; }
  return off;
}
}  // namespace

extern "C" size_t sdk_xvector_workspace_bytes(const sdk_xvector_desc* d, int B, int T) {
  if (!d || B <= 0 || T <= 0 || d->n_frame_layers < 1 || d->n_frame_layers > 8) return 0;
  return xv_layout(d, B, T, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int sdk_xvector_forward(sdk_ctx* ctx, const void* wblob, const sdk_xvector_desc* d, const uint16_t* feats, int ldf, int B, int T,
                                   void* ws, size_t ws_bytes, float* emb, void* stream) {
  SDK_REQUIRE(ctx && wblob && feats && ws && emb, "sdk_xvector_forward: null argument");
  if (int rc = check_xdesc(d)) return rc;
  const bool hp = d->off[62] == 1, f16 = d->off[62] == 2;
  // (the blob's precision decides, per call; the features must be in its format: bf16 for 0, fp16 hi | lo planes for 1, one fp16 plane for 2)
  SDK_REQUIRE(B > 0 && T > 0 && (int64_t)B * T < (1ll << 31), "sdk_xvector_forward: bad batch (B=%d T=%d)", B, T);
  if (hp) SDK_REQUIRE(ldf % 16 == 0 && (ldf >> 1) >= d->n_feats && d->first_tap_pack == 0 && d->n_feats % 32 == 0,
                      "sdk_xvector_forward: precise mode reads fp16 planes [B*T, ldf] with the lo plane ldf/2 columns to the right: ldf=%d, n_feats=%d (a multiple of 32, no tap packing)", ldf, d->n_feats);
  else SDK_REQUIRE(ldf >= d->n_feats && ldf % 8 == 0, "sdk_xvector_forward: ldf=%d < feature width %d", ldf, d->n_feats);
  for (int l = 0; l < d->n_frame_layers; ++l)
    SDK_REQUIRE(T > (d->kernel[l] / 2) * d->dilation[l], "sdk_xvector_forward: segments of %d frames are shorter than layer %d's halo", T, l);
  SDK_REQUIRE(ws_bytes >= sdk_xvector_workspace_bytes(d, B, T) && ((uintptr_t)ws % 256) == 0 && ((uintptr_t)wblob % 256) == 0, "sdk_xvector_forward: workspace too small or misaligned");
  uint16_t *b0, *b1;
  float* stats;
  xv_layout(d, B, T, (char*)ws, &b0, &b1, &stats);
  const char* wb = (const char*)wblob;
  const uint16_t* in = feats;
  int64_t ldin = ldf;
  const int M = B * T;
  const int Cl = d->cout[d->n_frame_layers - 1];
  if (hp) {
    // precise mode (north_star's 1e-5): every frame layer on the fp16 hi+lo plane GEMM (three MFMAs per product, csrc/hp.hip), planes between the
    // layers, statistics pooling on the planes; the embedding layer is fp32 either way
    int64_t in_lo = ldf >> 1;
    for (int l = 0; l < d->n_frame_layers; ++l) {
      uint16_t* out = (l & 1) ? b1 : b0;
      sdk_conv_gemm_hp_args g;
      memset(&g, 0, sizeof(g));
      g.A = in; g.lda = ldin; g.a_lo = in_lo; g.W = (const uint16_t*)(wb + d->off[4 * l]);
      g.C = out; g.ldc = 2 * (int64_t)d->cout[l]; g.c_lo = d->cout[l];
      g.bias = (const float*)(wb + d->off[4 * l + 1]); g.scale = (const float*)(wb + d->off[4 * l + 2]); g.shift = (const float*)(wb + d->off[4 * l + 3]);
      g.M = M; g.N = d->cout[l]; g.Cin = d->cin[l]; g.taps = d->kernel[l]; g.dil = d->dilation[l]; g.T = T; g.flags = SDK_GEMM_RELU;
      if (int rc = sdk_conv_gemm_hp(ctx, &g, stream)) return rc;
      in = out;
      ldin = 2 * (int64_t)d->cout[l];
      in_lo = d->cout[l];
    }
    if (int rc = hp_asp_stats(ctx, in, ldin, in_lo, B, T, Cl, stats, stream)) return rc;
  } else {
    for (int l = 0; l < d->n_frame_layers; ++l) {
      uint16_t* out = (l & 1) ? b1 : b0;
      sdk_conv_gemm_args g;
      memset(&g, 0, sizeof(g));
      g.A = in; g.lda = ldin; g.W = (const uint16_t*)(wb + d->off[4 * l]); g.C = out; g.ldc = d->cout[l];
      g.bias = (const float*)(wb + d->off[4 * l + 1]); g.scale = (const float*)(wb + d->off[4 * l + 2]); g.shift = (const float*)(wb + d->off[4 * l + 3]);
      g.M = M; g.N = d->cout[l]; g.Cin = d->cin[l]; g.taps = d->kernel[l]; g.dil = d->dilation[l]; g.T = T; g.flags = SDK_GEMM_RELU | (f16 ? SDK_GEMM_F16 : 0u);
      if (l == 0 && d->first_tap_pack) g.tap_pack = d->first_tap_pack;
      if (int rc = sdk_conv_gemm(ctx, &g, stream)) return rc;
      in = out;
      ldin = d->cout[l];
    }
    if (int rc = asp_stats_impl(ctx, in, ldin, B, T, Cl, stats, stream, f16)) return rc;
  }
  return sdk_rows_fc(ctx, stats, 2 * Cl, nullptr, nullptr, (const float*)(wb + d->off[60]), (const float*)(wb + d->off[61]), emb, d->embed_dim, B, 2 * Cl,
                     d->embed_dim, 0, stream);
}
