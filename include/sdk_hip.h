/*
 * sdk_hip.h - C ABI of libsdk_hip.so, the MI355X (gfx950) speaker-embedding + assignment path.
 *
 * The reference (CLIAI/speaker-diarization-toolkit) is pure Python and calls no native code;
 * its plug-in boundary is the Python class contract
 *     speaker_detection_backends/base.py:22-200   (EmbeddingBackend)
 *     speaker_detection_backends/base.py:272-293  (get_backend -> module.Backend())
 * The entry points below are what a local backend behind that contract binds (ctypes stub in
 * INTEGRATION.md).  Each one cites the reference interface whose work it performs.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 on success, non-zero on failure; sdk_last_error() then holds a
 *     thread-local message.  Nothing falls back to the CPU.
 *   - all data pointers are DEVICE pointers owned by the caller (e.g. torch tensors'
 *     data_ptr()), unless a parameter is documented as host memory.
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised.
 *   - bf16 tensors are passed as uint16_t* (raw bfloat16 bits).
 *   - "rows" of an activation tensor are frames: row m = segment (m / T), frame (m % T),
 *     channel-last, row stride given explicitly (ld*, in elements).
 *
 * STABLE SURFACE - what a backend binds (INTEGRATION.md shows the stub); SDK_ABI_VERSION changes when any of these does:
 *     sdk_abi_version  sdk_init  sdk_shutdown  sdk_last_error  sdk_get_device_info
 *     sdk_device_malloc  sdk_device_free  sdk_memcpy  sdk_stream_synchronize          device memory for hosts without an allocator of their own
 *     sdk_resample_out_len  sdk_resample_s16                                   audio -> AudioProfile format
 *     sdk_fbank_tables_bytes  sdk_fbank_tables_fill  sdk_fbank_workspace_bytes  sdk_fbank  sdk_fbank_windows          k1
 *     sdk_ingest_create / _destroy / _acquire / _commit / _submit / _release / _copy_ms          host audio -> HBM, pinned + double-buffered
 *     sdk_ecapa_workspace_bytes  sdk_ecapa_forward  sdk_ecapa_calib_floats  sdk_ecapa_forward_calib            k2
 *     sdk_xvector_workspace_bytes  sdk_xvector_forward                                                  k2 (second model family)
 *     sdk_l2norm                                                                                        k3
 *     sdk_affinity_workspace_bytes  sdk_affinity_topk                                                   k4
 *     sdk_affinity_matvec_workspace_bytes  sdk_affinity_matvec  sdk_rows_gram_workspace_bytes  sdk_rows_gram
 *     sdk_rows_apply  sdk_chol_inverse  sdk_rows_unit  sdk_kmeans_mindist  sdk_kmeans_assign            k6 (driven by cluster.py)
 * BUILDING BLOCKS AND KNOBS - exported for the parity tests and the A/B tools, free to change between rounds, not for binding:
 *     sdk_conv_gemm*  sdk_colstats_finish  sdk_res2net_chain*  sdk_se_*  sdk_asp_*  sdk_rows_fc  (pieces of sdk_ecapa_forward)
 *     sdk_set_option  sdk_set_gemm_variant  sdk_profile_begin / _end  sdk_debug_set_ptr  sdk_affinity_plan*  sdk_affinity_block_plan*  sdk_affinity_matvec_plan  sdk_conv_gemm_hp
 *     sdk_allgather  sdk_laplacian_topk_workspace_bytes  sdk_laplacian_topk        k5 / k6 drivers for a non-Python host (the library holds no
 *                                                                                     communicator: the caller passes its ncclComm_t; the Python
 *                                                                                     host layer uses torch.distributed, dist.py / cluster.py)
 */
#ifndef SDK_HIP_H
#define SDK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDK_ABI_VERSION 4   /* 2: sdk_ecapa_desc.precision (round 3); 3: sdk_fbank_windows + sdk_ingest_* (round 4); 4: precision 2, sdk_fbank_fmt, SDK_GEMM_F16, lazy ingest slots (round 5) */

typedef struct sdk_ctx sdk_ctx;

typedef struct sdk_device_info {
  int device;
  int compute_units;
  int clock_khz;
  int wavefront_size;
  uint64_t hbm_bytes;
  char name[128];
  char arch[64];
} sdk_device_info;

/* ---- lifecycle (replaces Backend.__init__ of a reference backend: base.py:291-293) ---- */
int sdk_abi_version(void);
int sdk_init(int device, sdk_ctx** out);
int sdk_shutdown(sdk_ctx* ctx);
const char* sdk_last_error(void);
int sdk_get_device_info(sdk_ctx* ctx, sdk_device_info* out);
/* Device memory for hosts that bring no allocator (a C / C++ binding; the torch-free Python path lite.py that serves a one-recording CLI
 * call without the 0.8-s `import torch` - the reference constructs its backend fresh in every process, base.py:291-293).  sdk_memcpy: kind 1 =
 * host -> device, 2 = device -> host, 3 = device -> device; ordered on `stream` and COMPLETE when the call returns (the one entry point that
 * synchronises, next to sdk_stream_synchronize).  Pointers from any other allocator of the same HIP runtime (torch tensors) work equally. */
int sdk_device_malloc(sdk_ctx* ctx, size_t bytes, void** out);
int sdk_device_free(sdk_ctx* ctx, void* p);
int sdk_memcpy(sdk_ctx* ctx, void* dst, const void* src, size_t bytes, int kind, void* stream);
int sdk_stream_synchronize(sdk_ctx* ctx, void* stream);
/* A/B and test knobs: "res2net_chain_fusion" (1 default / 0 = seven conv_gemm launches), "res2net_packed_weights" (1 default /
 * 0 = the chain ignores the blob's optional fragment-ordered weight copies, ecapa_layout.h EL_CHAINPACK), "res2net_two_per_cu" (1 default /
 * 0 = 8-wave workgroups, one segment per CU), "asp_packed_weights"
 * (likewise for the ASP logit weights, EL_ASP_W2PACK), "asp_per_segment"
 * (1 default / 0 = one workgroup per (segment, 128 channels)), "h_kblocked" (1 default / 0 = sdk_ecapa_forward keeps the MFA output row-major
 * instead of K-blocked, SDK_GEMM_C_KBLOCKED below), "affinity_fast_path" / "affinity_variant" /
 * "affinity_boundary_penalty" (k = 1 affinity kernel selection and work split), "chol_pivot_rtol_ppb" / "chol_shift_ppb"
 * (sdk_chol_inverse, below), "matvec_variant" (0 default: persistent row-group kernel / 1 = round 1's
 * kernel; sums differ in the last bits only), "gemm_variant" (see sdk_set_gemm_variant).  Results do not depend on them.
 * NOT a knob - a numerical contract: "precision" 0 (default: bf16 operands, bf16 layer-boundary storage; PCM -> score within ~4e-3 of
 * the fp32 model, ~9e-4 with the host's bias correction) / 1 (fp16 hi+lo planes, three MFMAs per product: within 1e-5, ~3x the GEMM time) /
 * 2 (round 5: ONE fp16 plane - the default schedule with fp16 instead of bf16 storage and MFMA operands: ~5e-4, ~1.7e-4 bias-corrected, ~0.96x the
 * default's throughput; ECAPA-TDNN forward only).  The OPTION is only the default of the entry points that have no format argument: the output
 * format of sdk_fbank / sdk_fbank_windows (bf16 / planes / fp16; sdk_fbank_fmt takes it per call) and the element format the stand-alone sweeps
 * (sdk_se_gate_residual, sdk_asp_stats, sdk_asp_pool, sdk_asp_fused*, sdk_res2net_chain) read and write.  The forwards take the contract from the
 * weight blob's descriptor (sdk_ecapa_desc.precision), sdk_conv_gemm from its flags (SDK_GEMM_F16): per call, no shared state. */
int sdk_set_option(sdk_ctx* ctx, const char* name, int value);
/* Diagnostics: "stamps" = device buffer [workgroups][64] of uint64 that the affinity kernel (tools/aff_timeline.py) and the
 * Res2Net chain (tools/res2net_timeline.py) fill with in-kernel wall-clock stamps; "gemm_clock" = EXACTLY [4096][2] uint64 {shader cycles, 100 MHz ticks} of each
 * conv_gemm256 workgroup's lifetime (bench.py: the clock the chip holds inside the dominant kernel; workgroups >= 4096 do not write);
 * "gemm_stamps" = EXACTLY [4096] uint64 wall-clock stamps of conv_gemm256 workgroup 0's tile phases (tools/gemm_timeline.py;
 * a buffer of its own - the clock probe never writes outside its [4096][2]).  NULL (default) = off. */
int sdk_debug_set_ptr(sdk_ctx* ctx, const char* name, void* device_ptr);

/* ---- measurement: per-kernel-family HIP-event timing on the launch stream (bench.py roofline) -- */
enum {
  SDK_K_CONV_GEMM = 0, SDK_K_SE_GATE, SDK_K_ASP_STATS, SDK_K_ROWS_FC, SDK_K_ASP_POOL, SDK_K_FBANK_TILE,
  SDK_K_FBANK_NORM, SDK_K_L2NORM, SDK_K_AFF_COARSE, SDK_K_AFF_RESCORE, SDK_K_AFF_RESCAN, SDK_K_COPY,
  SDK_K_AFF_MATVEC, SDK_K_CONV_GEMM256, SDK_K_ASP_FUSED, SDK_K_RES2NET, SDK_K_RESAMPLE, SDK_K_CONV_GEMM_HP, SDK_K_COUNT
};
typedef struct sdk_profile_report {
  int32_t launches[24];
  double ms[24];          /* summed device time of the family's launches */
  double flops[24];       /* executed flops as launched (2*M*N*K for GEMMs) */
  double bytes[24];       /* compulsory bytes as launched (inputs once + outputs once) */
} sdk_profile_report;
int sdk_profile_begin(sdk_ctx* ctx);
int sdk_profile_end(sdk_ctx* ctx, sdk_profile_report* out);   /* synchronises the device */

/* ---- k1: fbank.  Input contract = audio_profiles.py:25-29 (16 kHz mono s16le). --------
 * pcm   [B, S] int16 (device)          T = 1 + S/160 frames per segment
 * tabs  packed DFT/mel tables from sdk_fbank_tables_bytes()/sdk_fbank_tables_fill(), copied to
 *       the device by the caller (16-byte aligned)
 * ws    caller scratch of sdk_fbank_workspace_bytes(B, S)
 * feats [B*T, ldf] bf16, channels 0..79 = mean-normalised log-mel, 80..ldf-1 = 0 (ldf >= 80)
 *       precise mode ("precision" 1): fp16 planes, hi in columns [0, ldf/2), lo in [ldf/2, ldf), channels >= 80 of each zero
 *       (ldf/2 >= 80, ldf % 16 == 0); the DFT then runs on fp16 MFMAs with an fp16 hi+lo table (the folded int16 samples split exactly in
 *       fp16 too: three MFMAs per product)
 */
size_t sdk_fbank_tables_bytes(void);
int sdk_fbank_tables_fill(void* host_dst, size_t bytes);          /* HOST buffer */
size_t sdk_fbank_workspace_bytes(int B, int S);                    /* fp32 log-mel scratch */
int sdk_fbank(sdk_ctx* ctx, const int16_t* pcm, int B, int S, const void* tabs,
              uint16_t* feats, int ldf, void* ws, size_t ws_bytes, void* stream);
/* The same, with the windows cut ON THE DEVICE from ONE resident recording (the boundary hands a backend a path and segments,
 * base.py:130-151; the reference cuts with ffmpeg per segment list, speechmatics_backend.py:231-281): samples [n_samples] int16 (device),
 * starts [B] int32 (device) = first sample of every window, each inside the recording; window b = samples[starts[b] .. starts[b] + S), samples
 * past the recording's end read as zero.  Bit-identical to sdk_fbank on the materialised [B, S] windows; the overlapping windows (2x the
 * samples at hop 1 s / window 2 s) never exist on the host and never cross PCIe.  n_samples < 2^31. */
int sdk_fbank_windows(sdk_ctx* ctx, const int16_t* samples, int64_t n_samples, const int32_t* starts, int B, int S, const void* tabs,
                      uint16_t* feats, int ldf, void* ws, size_t ws_bytes, void* stream);
/* Both with the output format as an ARGUMENT (precision 0 / 1 / 2, see sdk_set_option) instead of the context's "precision" option (round 5): a
 * host that runs several numerical contracts on one device - two engines, two threads - then shares no mutable state through the library.
 * sdk_ecapa_forward / sdk_xvector_forward take the contract from their descriptor alone; the features must have been written in that format. */
int sdk_fbank_fmt(sdk_ctx* ctx, const int16_t* pcm, int B, int S, const void* tabs, uint16_t* feats, int ldf, void* ws, size_t ws_bytes, int precision,
                  void* stream);
int sdk_fbank_windows_fmt(sdk_ctx* ctx, const int16_t* samples, int64_t n_samples, const int32_t* starts, int B, int S, const void* tabs,
                          uint16_t* feats, int ldf, void* ws, size_t ws_bytes, int precision, void* stream);

/* ---- ingest: host audio -> HBM, staged through pinned memory on a copy stream of its own, `depth`-deep (2 = double-buffered) so that the
 *      upload of recording i + 1 runs under the forward pass of recording i.  A slot = pinned host buffers for int16 samples and
 *      int32 window starts + their device twins, allocated on the slot's first use at the size of that upload (max_samples / max_windows bound it).
 *   sdk_ingest_acquire : next slot (ring order); *pinned_samples / *pinned_starts are HOST pointers the caller fills (a file reader can
 *                        read straight into them: no second host copy).  Fails if that slot was committed and never released; a slot
 *                        acquired and never committed (its filler gave up) is handed out again when the ring comes round.
 *   sdk_ingest_commit  : validates the start table (every start inside [0, n_samples)), enqueues the uploads on the copy stream - behind the
 *                        slot's previous consumer, without blocking the host - and makes `compute_stream` wait for them;
 *                        *dev_samples / *dev_starts are DEVICE pointers for sdk_fbank_windows (any sub-range of the table may be launched)
 *   sdk_ingest_submit  : acquire + memcpy from pageable host memory + commit
 *   sdk_ingest_release : call after the LAST kernel reading the slot has been enqueued on compute_stream; the slot's next upload waits for it
 *   sdk_ingest_copy_ms : duration / bytes of the slot's last upload (HIP events on the copy stream; synchronises with that upload) */
typedef struct sdk_ingest sdk_ingest;
int sdk_ingest_create(sdk_ctx* ctx, int64_t max_samples, int max_windows, int depth, sdk_ingest** out);
int sdk_ingest_destroy(sdk_ingest* ing);
int sdk_ingest_acquire(sdk_ingest* ing, int* ticket, int16_t** pinned_samples, int32_t** pinned_starts);   /* = _sized(max_samples, max_windows) */
/* a slot's buffers are allocated on its first use and sized to what it is asked to take (grown when a larger upload arrives; where the host
 * refuses page-locked memory the slot stages through ordinary memory): max_samples / max_windows of sdk_ingest_create are upper bounds only */
int sdk_ingest_acquire_sized(sdk_ingest* ing, int64_t n_samples, int n_windows, int* ticket, int16_t** pinned_samples, int32_t** pinned_starts);
int sdk_ingest_slot_info(sdk_ingest* ing, int slot, int64_t* cap_samples, int* pinned);
int sdk_ingest_commit(sdk_ingest* ing, int ticket, int64_t n_samples, int n_windows, int window_len, void* compute_stream,
                      const int16_t** dev_samples, const int32_t** dev_starts);
int sdk_ingest_submit(sdk_ingest* ing, const int16_t* host_samples, int64_t n_samples, const int32_t* host_starts, int n_windows,
                      int window_len, void* compute_stream, int* ticket, const int16_t** dev_samples, const int32_t** dev_starts);
int sdk_ingest_release(sdk_ingest* ing, int ticket, void* compute_stream);
int sdk_ingest_copy_ms(sdk_ingest* ing, int ticket, float* ms, double* bytes);

/* ---- k2 building blocks (ECAPA-TDNN forward; behind EmbeddingBackend.enroll_speaker /
 *      identify_speaker, base.py:107-151) -------------------------------------------------- */

#define SDK_GEMM_RELU 1u
#define SDK_GEMM_TANH 2u
/* K-BLOCKED activation layout (round 4): a [M, cols] bf16 matrix stored as [cols / 64][M][64] - element (m, c) at (c / 64) * M * 64 + m * 64 + c % 64 -
 * so that the 128-byte row piece a GEMM K-step (or an ASP channel block) takes from a row lies next to its neighbours' instead of `ld` apart.  Made for
 * the widest activation of the forward, the MFA output h [B*T, 3072]: written once (C_KBLOCKED: plain-layer shape of the 256^2 kernel only, ldc ignored),
 * read by the skinny attention-hidden GEMM (A_KBLOCKED: taps == 1, lda ignored; runs on the 128^2 kernel) and by sdk_asp_fused_kblocked. */
#define SDK_GEMM_A_KBLOCKED 4u
#define SDK_GEMM_C_KBLOCKED 8u
#define SDK_GEMM_F16 16u          /* A, W and the 2-byte outputs are fp16 (IEEE binary16) instead of bf16: the single-plane fp16 contract (precision 2) */

/* Dilated 1-D convolution over frames as one MFMA GEMM:
 *   pre[m, n] = bias[n] + ubias[m / T, n] + sum_{j<taps} sum_{c<Cin}
 *                 A[seg(m)*T + reflect(t(m) + (j - taps/2)*dil), c] * W[n, j*Cin + c]
 *   v = RELU? max(pre,0) : pre;  v = v*scale[n] + shift[n];  v = TANH? tanh(v) : v
 *   C[m,n] = bf16(v);  C32[m,n] = v;  S[m,n] = bf16(float(bf16(v)) + float(X2[m,n]))
 * Requirements: Cin % 64 == 0, N % 128 == 0, T > (taps/2)*dil, M % T == 0.
 * Any of bias/scale/shift/ubias/C/C32/X2+S may be NULL. */
typedef struct sdk_conv_gemm_args {
  const uint16_t* A;  int64_t lda;
  const uint16_t* W;               /* [N, taps*Cin] bf16, K contiguous */
  uint16_t* C;        int64_t ldc;
  float* C32;         int64_t ldc32;
  const float* bias;  const float* scale;  const float* shift;
  const float* ubias; int64_t ldub;
  const uint16_t* X2; int64_t ldx2;
  uint16_t* S;        int64_t lds;
  int M, N, Cin, taps, dil, T;
  uint32_t flags;
  /* optional fused per-segment column statistics of the stored output (SE squeeze means, ASP global
   * context): stats_mode 1 = sum, 2 = sum and sum of squares; stats_part = scratch of
   * sdk_conv_gemm_stats_bytes(); finish with sdk_colstats_finish().  Only where
   * sdk_conv_gemm_stats_fusable(M, N, T) is true. */
  int32_t stats_mode;
  float* stats_part;
  /* optional addend of the A operand, same rows/row map/channels: the GEMM consumes bf16(A + A2)
   * (Res2Net: y_{c-1} + u_c formed on the way into LDS instead of round-tripping through HBM) */
  const uint16_t* A2; int64_t lda2;
  /* > 0: the taps are PACKED along K: W is [N, round_up(taps * tap_pack, 64)] (tap-major, tap_pack channels per tap, zero K padding),
   * Cin == tap_pack (a multiple of 8, not of 64): the first layer's 5 x 80 mel channels take 7 K-steps of 64 instead of 5 x 128 -> 10 */
  int32_t tap_pack;
  int32_t reserved;
} sdk_conv_gemm_args;
int sdk_conv_gemm(sdk_ctx* ctx, const sdk_conv_gemm_args* a, void* stream);
size_t sdk_conv_gemm_stats_bytes(int M, int N, int mode);
int sdk_conv_gemm_stats_fusable(int M, int N, int T);
/* out: mode 1 -> [B, N] per-segment column means; mode 2 -> [B, 2N] mean | sqrt(max(var, 1e-12)) */
int sdk_colstats_finish(sdk_ctx* ctx, const float* stats_part, int M, int N, int T, int mode, float* out, void* stream);
/* Tuning knob (A/B measurements): 1 = 128x128 register-staged tile, 2 = 256x256 LDS-DMA tile where
 * the shape allows it (default; also settable once via $SDK_GEMM_VARIANT). */
int sdk_set_gemm_variant(int variant);

/* Squeeze-excitation gate + residual, one workgroup per segment:
 *   mean[c] = (1/T) sum_t z[b,t,c];  h = relu(W1 mean + b1);  g = sigmoid(W2 h + b2)
 *   out[b,t,c] = bf16(g[c]*z[b,t,c] + x[b,t,c])
 * w1t [C, Cse] fp32 (transposed), w2t [Cse, C] fp32 (transposed).  C % 8 == 0 and (C/8) | 256, Cse | 256.
 * With a workspace the work is split into a mean sweep, two batched FCs on the fp32 matrix pipe and an
 * apply sweep (same arithmetic, weights read once per 32 segments). */
/* Res2Net chain of one block fused per segment (T <= sdk_res2net_chain_max_frames(), 128-channel sub-bands):
 *   R[:, 128c : 128(c+1)] = y_c,  y_1 = TDNN_0(U chunk 1),  y_c = TDNN_{c-1}(bf16(U chunk c + y_{c-1})),  c = 2..nconv
 * W/bias/scale/shift: HOST arrays of nconv device pointers (W[i]: bf16 [128][3*128]).  Bit-identical to the
 * same chain expressed as nconv sdk_conv_gemm launches.  R may be U itself (in place: a segment's chunk c has been
 * read before y_c is written over it), which also leaves chunk 0 where the next layer expects it. */
int sdk_res2net_chain_max_frames(void);
int sdk_res2net_chain(sdk_ctx* ctx, const uint16_t* U, int64_t ldu, uint16_t* R, int64_t ldr, const uint16_t* const* W,
                      const float* const* bias, const float* const* scale, const float* const* shift, int nconv,
                      int B, int T, int dil, void* stream);
size_t sdk_se_workspace_bytes(int B, int C, int Cse);   /* fp32 [B,C] means + [B,Cse] hidden + [B,C] gates */
int sdk_se_gate_residual(sdk_ctx* ctx, const uint16_t* z, int64_t ldz, const uint16_t* x, int64_t ldx,
                         const float* w1t, const float* b1, const float* w2t, const float* b2,
                         uint16_t* out, int64_t ldo, int B, int T, int C, int Cse,
                         const float* mean_in,      /* optional [B, C] squeeze means already computed (fused GEMM epilogue) */
                         void* ws, size_t ws_bytes, void* stream);   /* ws may be NULL: one-kernel-per-segment form */

/* Attentive statistics pooling pieces.
 *   sdk_asp_stats : ctx[b, 0:C] = mean_t h, ctx[b, C:2C] = sqrt(max(var_t h, 1e-12))   fp32
 *   sdk_rows_fc   : out[b, j] = act(bias[j] + sum_c (in[b,c]*in_scale[c]+in_shift[c]) * wt[c, j])
 *                   (wt [Cin, Nout] fp32 transposed; act 0 none, 1 relu, 2 sigmoid)
 *   sdk_asp_pool  : softmax over t of logits[b,t,c] -> weighted mean / std of h -> pooled[b, 0:C | C:2C]
 */
int sdk_asp_stats(sdk_ctx* ctx, const uint16_t* h, int64_t ldh, int B, int T, int C, float* out_ctx, void* stream);
/* the same with the element format of h as an argument (0: bf16, 2: fp16) instead of the context's default - per call, like sdk_fbank_fmt */
int sdk_asp_stats_fmt(sdk_ctx* ctx, const uint16_t* h, int64_t ldh, int B, int T, int C, float* out_ctx, int precision, void* stream);
int sdk_rows_fc(sdk_ctx* ctx, const float* in, int64_t ldin, const float* in_scale, const float* in_shift,
                const float* wt, const float* bias, float* out, int64_t ldout,
                int B, int Cin, int Nout, int act, void* stream);
int sdk_asp_pool(sdk_ctx* ctx, const float* logits, int64_t ldl, const uint16_t* h, int64_t ldh,
                 int B, int T, int C, float* pooled, void* stream);
/* Fused form of (attention-logit GEMM + sdk_asp_pool) for T <= sdk_asp_fused_max_frames(): the fp32
 * logits stay in accumulator registers.  ah [B*T, A=128] bf16 attention hidden, w2 [C, A] bf16, b2 [C]. */
int sdk_asp_fused_max_frames(void);
int sdk_asp_fused(sdk_ctx* ctx, const uint16_t* ah, int64_t ldah, const uint16_t* w2, const float* b2,
                  const uint16_t* h, int64_t ldh, int B, int T, int C, int A, float* pooled, void* stream);
/* The same with h in the K-blocked layout [C / 64][B*T][64] (SDK_GEMM_C_KBLOCKED above).  Only the per-segment form reads it: where
 * sdk_asp_kblocked_ok(ctx, T, C) is 0 (short or long windows, option asp_per_segment off) the call is an error and the caller keeps h row-major.
 * Bit-identical to sdk_asp_fused on the same values. */
int sdk_asp_kblocked_ok(sdk_ctx* ctx, int T, int C);
int sdk_asp_fused_kblocked(sdk_ctx* ctx, const uint16_t* ah, int64_t ldah, const uint16_t* w2, const float* b2,
                           const uint16_t* h, int B, int T, int C, int A, float* pooled, void* stream);

/* ---- PRECISE MODE (sdk_set_option "precision" 1; north_star: cosine scores within 1e-5 of the fp32 model, which bf16 operands miss
 *      by 4e-3 - profiles/r03_error_budget.md).  Tensors the default mode rounds to bf16 travel as fp16 hi + lo PLANES: a [rows, C]
 *      activation is [rows, ld] fp16, hi values in columns [0, C), lo values `lo` columns to the right, x = float(hi) + float(lo);
 *      the lo values are stored times 2^11 (x = float(hi) + float(lo) / 2048: always in the normal fp16 range of their hi);
 *      GEMM weights are a 256-byte header (float 1 / 2^s) + [2][N][K] fp16 planes (hi, lo) of 2^s * W, s per layer
 *      (weights_pack.hp_weight_planes); every product runs as three fp16 MFMAs (hi.hi + lo.hi + hi.lo), fp32 accumulate, fp32 epilogue
 *      with libm tanh.  Same operator as sdk_conv_gemm otherwise (Cin % 32 == 0, N % 128 == 0). */
typedef struct sdk_conv_gemm_hp_args {
  const uint16_t* A;  int64_t lda, a_lo;
  const uint16_t* W;                       /* the weight slot: header + [2][N][taps*Cin] fp16 planes */
  uint16_t* C;        int64_t ldc, c_lo;   /* planes out (may be NULL) */
  float* C32;         int64_t ldc32;       /* fp32 out (may be NULL) */
  const float* bias;  const float* scale;  const float* shift;
  const float* ubias; int64_t ldub;
  const uint16_t* X2; int64_t ldx2, x2_lo; /* S = planes(v + X2) (Res2Net running sum; may be NULL) */
  uint16_t* S;        int64_t lds, s_lo;
  int M, N, Cin, taps, dil, T;
  uint32_t flags;
} sdk_conv_gemm_hp_args;
int sdk_conv_gemm_hp(sdk_ctx* ctx, const sdk_conv_gemm_hp_args* a, void* stream);

/* Whole forward: feats [B*T, ldf] bf16 -> raw embeddings emb [B, 192] fp32.
 * `wblob` is the packed device weight blob and `wdesc` (HOST) its offset table, both produced by
 * the host packer (weights_pack.py); `ws` is caller-owned scratch of sdk_ecapa_workspace_bytes(). */
typedef struct sdk_ecapa_desc {
  int32_t n_mels_padded, channels, sub_channels, scale, se_channels, attn_channels, mfa_channels, embed_dim;
  int32_t n_blocks, kernel0;
  int32_t dilation[4];
  int32_t precision;       /* 0: bf16 operand blob (default mode); 1: fp16 hi+lo plane blob (precise mode: feats are planes [B*T, ldf] with
                              the lo plane ldf/2 columns to the right, n_mels_padded = 96); must equal the context's "precision" option */
  int32_t blk0_tap_pack;   /* default mode: > 0 = the first layer's weight slot is packed along K (sdk_conv_gemm_args.tap_pack), value = mel channels per
                              tap (80); 0 = [C][kernel0 * n_mels_padded] */
  /* byte offsets into wblob; -1 = absent.  Layout of the index space: see weights_pack.py */
  int64_t off[256];
} sdk_ecapa_desc;
size_t sdk_ecapa_workspace_bytes(const sdk_ecapa_desc* d, int B, int T);
int sdk_ecapa_forward(sdk_ctx* ctx, const void* wblob, const sdk_ecapa_desc* wdesc,
                      const uint16_t* feats, int ldf, int B, int T,
                      void* ws, size_t ws_bytes, float* emb, void* stream);

/* Calibration pass for the host's BIAS CORRECTION of the bf16 weight rounding (weights_pack.bias_corrections): the same
 * forward (Res2Net chain unfused), additionally writing the per-segment mean | std ([B, 2 C_l] fp32, sdk_asp_stats layout) of the INPUT of every
 * corrected GEMM layer into `calib`, slots in this order: per block {tdnn1 [C], Res2Net conv 0..scale-2 [sub_channels each], tdnn2 [C]}, then
 * MFA [mfa_channels], ASP hidden [mfa_channels]; sdk_ecapa_calib_floats() = total floats.  Default-mode blobs only. */
size_t sdk_ecapa_calib_floats(const sdk_ecapa_desc* d, int B);
int sdk_ecapa_forward_calib(sdk_ctx* ctx, const void* wblob, const sdk_ecapa_desc* wdesc, const uint16_t* feats, int ldf, int B, int T,
                            void* ws, size_t ws_bytes, float* emb, float* calib, void* stream);

/* ---- x-vector (plain TDNN) forward - north_star names "ECAPA-TDNN/x-vector".  Frame layers l = 0..n_frame_layers-1:
 *      dilated conv (kernel[l], dilation[l], "same" length by segment-local reflection) -> ReLU -> BatchNorm(eval), bf16 layer-boundary
 *      storage, all on sdk_conv_gemm (layer 0 with its taps packed along K when the feature width is not a multiple of 64); then statistics
 *      pooling (mean | std over frames, sdk_asp_stats) and the embedding layer (fp32, sdk_rows_fc).  feats as sdk_fbank writes them
 *      ([B*T, ldf] bf16); emb [B, embed_dim] fp32 (pre-activation of the first segment layer, the usual x-vector).
 *      off[4 l + {0,1,2,3}] = W (bf16 [cout][K]), bias, BN scale, BN shift of frame layer l; off[60] = FC weight (fp32 [2 cout_last, embed_dim],
 *      transposed), off[61] = FC bias.  cout[] are multiples of 128 (pad a 1500-wide layer to 1536 with zero weights).
 *      off[62] = numerical contract of the blob: -1 / 0 = bf16 operands (default mode); 1 = PRECISE mode (round 4): W slots are sdk_conv_gemm_hp
 *      weight slots (header + fp16 hi / lo planes), feats are fp16 planes [B*T, ldf] (lo plane ldf/2 columns to the right, n_feats = 96 padded
 *      mel channels, no tap packing), the layers run on sdk_conv_gemm_hp and the pooling on the planes; 2 = ONE fp16 plane
 *      (round 5): the default layout with fp16 bits in the W slots and in feats (sdk_fbank_fmt(..., 2, ...)), sdk_conv_gemm with SDK_GEMM_F16.
 *      The blob decides per call; the context's "precision" option is not consulted. */
typedef struct sdk_xvector_desc {
  int32_t n_frame_layers, n_feats, embed_dim, first_tap_pack;
  int32_t kernel[8], dilation[8], cin[8], cout[8];
  int64_t off[64];
} sdk_xvector_desc;
size_t sdk_xvector_workspace_bytes(const sdk_xvector_desc* d, int B, int T);
int sdk_xvector_forward(sdk_ctx* ctx, const void* wblob, const sdk_xvector_desc* d, const uint16_t* feats, int ldf, int B, int T,
                        void* ws, size_t ws_bytes, float* emb, void* stream);

/* ---- audio conversion to the AudioProfile: replaces the ffmpeg subprocess the reference's backends
 *      run before upload (audio_profiles.py:70-100 `format_ffmpeg_args`; speechmatics_backend.py:231-281).
 *      x [n_in, channels] s16 interleaved -> y [n_out] s16 mono at rate_in * L / M, n_out = ceil(n_in * L / M).
 *      Channel down-mix (rounded mean) + polyphase FIR, integer arithmetic: taps [L][K] int32 Q30 (device memory,
 *      designed by the host layer, every phase summing to 2^30), int64 accumulation, round-half-up, s16 saturation;
 *      samples outside the input are zero.  Bit-exact against oracle/resample.py. -- */
int64_t sdk_resample_out_len(int64_t n_in, int L, int M);
int sdk_resample_s16(sdk_ctx* ctx, const int16_t* x, int64_t n_in, int channels, const int32_t* taps, int L, int M, int K,
                     int16_t* y, int64_t n_out, void* stream);

/* ---- k3: L2-normalise rows.  X [N, d] fp32 -> E fp32 unit rows, Eb bf16 copy,
 *      resid[n] = || E[n] - float(Eb[n]) ||_2 (rigorous per-row bf16 rounding residual). -- */
int sdk_l2norm(sdk_ctx* ctx, const float* X, int N, int d, float* E, uint16_t* Eb, float* resid, void* stream);

/* ---- k4: segments x profiles cosine affinity with fused top-k (replaces the scoring a local
 *      identify_speaker performs per candidate: base.py:130-151; rows consumed by
 *      speaker_detection:1085-1127).
 *   coarse pass : bf16 MFMA  Eb [N,d] x Pb [P,d]^T, fused per-row candidate lists (no N x P matrix in HBM)
 *   exact pass  : fp32 re-score of the candidates, sorted, ties -> lowest profile index
 *   guarantee   : rows whose last coarse candidate is within the rounding margin of the k-th exact score
 *                 are re-scanned exactly in fp32 over all P, so idx/score equal an fp32 full scan
 *                 (k = 1 is the fast path: ~1 % of rows rescanned; larger k rescans more).
 * d must be 192 (= 12 MFMA k-steps), k <= 4.  idx [N,k] int32, score [N,k] fp32.
 * n_rescanned (device int32, may be NULL) receives the number of rows that took the exact path.
 * ws: sdk_affinity_workspace_bytes(N, P). */
size_t sdk_affinity_workspace_bytes(int N, int P);
/* Host-only (no device): the k = 1 path's work decomposition, for tests.  out5 = {segment groups, profile stages per group,
 * workgroups, segments per group, record slots per segment}; *units = groups * stages; workgroup i sweeps the units
 * [i * units / workgroups, (i + 1) * units / workgroups) in (group, stage) order. */
int sdk_affinity_plan(int N, int P, int num_cu, int32_t* out5, int64_t* units);
/* Host-only: the unit range [u0, u1) of workgroup `wg` under that plan and the record slot of its first portion.  The ranges can be balanced by
 * cost instead of unit count ("affinity_boundary_penalty" p: a group boundary inside a range counts as p stages; default 0 - measured, not a robust win). */
int sdk_affinity_plan_range(int N, int P, int num_cu, int wg, int64_t* u0, int64_t* u1, int32_t* first_slot);
/* Host-only: the BLOCK plan of the coarse pass for short sweeps (config #3), for tests.  Since round 5 it is what sdk_affinity_topk takes where its cost
 * model prefers it, with two records per whole sweep (`affinity_variant` 0 = that choice, 7 = always the range plan, 8 / 12 / 13 = the block plan
 * wherever the shape fits with 1 / 2 / 3 records: the A/B pair the tests keep bit-identical).  Unit of work = a block of 32 segments with its whole
 * sweep; workgroup g owns blocks [g q, (g + 1) q), the leftover blocks are swept in `parts` stage ranges by waves with a free second slot.
 * sdk_affinity_block_plan: out6 = {1 = plan taken / 0 = the range plan stays (force != 0: taken whenever the shape fits), q, workgroups, stages per
 * sweep, parts per leftover block, leftover items}.  sdk_affinity_block_plan_wave: wave `wave` (0..7) of workgroup `wg`: out6 = {block of slot 0, block of
 * slot 1 or -1, its first stage, its end stage, its record slot, parts of that block}. */
int sdk_affinity_block_plan(int N, int P, int num_cu, int force, int32_t* out6);
int sdk_affinity_block_plan_wave(int N, int P, int num_cu, int wg, int wave, int32_t* out6);
int sdk_affinity_topk(sdk_ctx* ctx, const float* E, const uint16_t* Eb, const float* resid_e,
                      const float* P, const uint16_t* Pb, const float* resid_p,
                      int N, int Pn, int d, int k, int32_t* idx, float* score,
                      int32_t* n_rescanned, void* ws, size_t ws_bytes, void* stream);

/* ---- k6: spectral clustering pieces on the rectified cosine affinity A = max(E E^T, 0) (BASELINE.json
 *      config #5).  A is never stored: its tiles are recomputed on the matrix cores per application.
 *   sdk_affinity_matvec : Y[row0+i, :] = sum_j max(<e_i, e_j>, 0) * xscale[j] * X[j, :]   for i < rows
 *                         Eb [N,192] bf16 (ALL rows, i.e. the all-gathered embeddings), X [N,kv] fp32,
 *                         kv <= 32, xscale [N] or NULL, Y [N,kv] fp32 (only the owned rows are written).
 *                         Degrees are the case X = ones.  ws: sdk_affinity_matvec_workspace_bytes(N).
 *   sdk_rows_gram       : G [k,k] = X^T Y over n rows (order-fixed two-stage reduction)
 *   sdk_rows_apply      : Y[i,:] = scale[i] * (X[i,:] @ R),  R [k,k] row-major, scale may be NULL
 *   sdk_chol_inverse    : Rinv [k,k] = (L^T)^-1 with (G + G^T)/2 = L L^T, float64 inside (CholeskyQR without leaving the stream);
 *                         *not_spd (device int32, may be NULL) is SET to 1 if a pivot is not a number, not positive, OR at most 1e-6 of its
 *                         diagonal entry (the relative-pivot rule: column i lies within 1e-3 of the span of the columns before it, i.e.
 *                         cond(Y) > ~1e3 - beyond that the fp32 Gram matrix is rounding noise; sdk_set_option "chol_pivot_rtol_ppb", in 1e-9,
 *                         default 1000).  Never cleared: zero it once, run any number of passes, read it at the next host synchronisation.
 *                         Only a non-positive / NaN pivot is replaced (by 1, so the stream keeps running; the result is then meaningless).
 *                         "chol_shift_ppb" > 0: shifted CholeskyQR (G + s I, s = that fraction, in 1e-9, of the mean diagonal entry) for
 *                         nearly rank-deficient blocks; follow it with unshifted passes (cluster.spectral_cluster retries once that way
 *                         when the flag was raised - over-clustered or near-duplicate inputs - before it reports a lost rank)
 *   sdk_rows_unit       : rows scaled to unit length
 *   sdk_kmeans_mindist  : d2[i] = (first ? : min(d2[i],)) |R[i] - centre|^2      (maximin initialisation)
 *   sdk_kmeans_assign   : label[i] = nearest of kc centres (ties -> lowest), dist2, optional per-256-row-block
 *                         partial sums [nblk, kc, k] and counts [nblk, kc]
 */
size_t sdk_affinity_matvec_workspace_bytes(int N);
/* Host-only (no device): the mat-vec kernel's work decomposition, for tests.  out4 = {row groups of 512, j stages per group, workgroups,
 * partial-tile slots per group}; *units = groups * stages; workgroup i sweeps the units [i * units / workgroups, (i + 1) * units / workgroups)
 * in (group, stage) order and writes one partial Y tile per group it touches; a group's tiles are summed in slot order. */
int sdk_affinity_matvec_plan(int rows, int N, int num_cu, int32_t* out4, int64_t* units);
int sdk_affinity_matvec(sdk_ctx* ctx, const uint16_t* Eb, int N, int d, int row0, int rows, const float* X,
                        const float* xscale, int kv, float* Y, void* ws, size_t ws_bytes, void* stream);
/* ---- k5 / k6 for a non-Python host.  The Python host layer uses torch.distributed for the same collectives (dist.py).
 *   sdk_allgather       : ONE RCCL all-gather of equal shards (bytes_per_rank each) on the caller's communicator (an ncclComm_t passed as
 *                         void*) and stream: the [N/G, 192] embedding exchange over xGMI.  RCCL is resolved at first use (the copy already in
 *                         the process, else librccl.so.1); the library does not link it.
 *   sdk_allgather_direct: the same exchange as world - 1 PAIRWISE ncclSend / ncclRecv transfers in one RCCL group: on fully connected point-to-point
 *                         xGMI (7 links per GPU) every transfer takes its pair's direct link and all run at once (floor 0.63 ms for 8 x 96 MB),
 *                         whatever algorithm ncclAllGather itself would choose for the size (a ring: 4.4 ms).  Same bytes in `out`.
 *   sdk_laplacian_topk  : top-k eigenpairs of S = D^-1/2 A D^-1/2, A = max(E E^T, 0), by row-sharded subspace iteration (the loop of
 *                         cluster.spectral_cluster: degrees, CholeskyQR2, n_iter x [V all-gather, recomputed-affinity mat-vec, scaling,
 *                         CholeskyQR2], Ritz with a device-side k x k Jacobi eigh): never synchronises with the host.
 *                         Eb_all [N,192] bf16 = ALL embeddings (already gathered); this call owns rows [row0, row0 + rows); V [rows, k] fp32
 *                         in: any full-rank start block (e.g. seeded gaussian), out: the Ritz vectors (columns = eigenvectors, eigenvalue
 *                         descending; signs fixed by a deterministic rule on the k x k Ritz eigenvectors); eigvals DEVICE [k]; not_spd as sdk_chol_inverse (sticky, may be NULL).
 *                         comm NULL: single GPU (rows == N).  comm != NULL: every rank owns N / world rows (equal shards) and calls with the
 *                         same arguments; collectives: all-gather of D^-1/2 and of V per iteration, all-reduce of the k x k Gram matrices. */
int sdk_allgather(sdk_ctx* ctx, const void* shard, void* out, size_t bytes_per_rank, void* comm, void* stream);
int sdk_allgather_direct(sdk_ctx* ctx, const void* shard, void* out, size_t bytes_per_rank, void* comm, void* stream);
size_t sdk_laplacian_topk_workspace_bytes(int N, int k);
int sdk_laplacian_topk(sdk_ctx* ctx, const uint16_t* Eb_all, int N, int row0, int rows, int k, int n_iter, float* V, float* eigvals,
                       int32_t* not_spd, void* ws, size_t ws_bytes, void* comm, int world, void* stream);
size_t sdk_rows_gram_workspace_bytes(int n, int k);
int sdk_rows_gram(sdk_ctx* ctx, const float* X, const float* Y, int n, int k, float* G, void* ws, size_t ws_bytes, void* stream);
int sdk_rows_apply(sdk_ctx* ctx, const float* X, const float* R, const float* scale, int n, int k, float* Y, void* stream);
int sdk_chol_inverse(sdk_ctx* ctx, const float* G, int k, float* Rinv, int32_t* not_spd, void* stream);
int sdk_rows_unit(sdk_ctx* ctx, const float* X, int n, int k, float* Y, void* stream);
int sdk_kmeans_mindist(sdk_ctx* ctx, const float* R, int n, int k, const float* centre, float* d2, int first, void* stream);
int sdk_kmeans_assign(sdk_ctx* ctx, const float* R, int n, int k, const float* centres, int kc, int32_t* label,
                      float* dist2, float* part_sum, int32_t* part_cnt, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDK_HIP_H */
