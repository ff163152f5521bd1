/* A host that is not Python: the whole identify path (k1 -> k4) through include/sdk_hip.h alone - C99, no HIP headers, no framework.
 * It is the binding a maintainer of the toolkit would write if the backend lived in a compiled plug-in instead of a Python class
 * (speaker_detection_backends/base.py:107-151: audio in the AudioProfile format -> "which enrolled profile, what score" per window).
 *
 *   cc -std=c99 -O2 -Iinclude examples/c_host_identify.c -o c_host_identify -L<pkg dir> -lsdk_hip -Wl,-rpath,<pkg dir>
 *   ./c_host_identify pcm.s16 B S  blob.bin desc.bin  profiles.f32 P  out.bin
 *
 * Inputs are raw little-endian files written by the caller (tests/test_c_host.py writes them from the same arrays it hands to the Python
 * engine): pcm [B][S] int16, the packed weight blob + its sdk_ecapa_desc (weights_pack.py), profiles [P][192] float32.
 * out.bin = int32 idx[B] then float score[B] then float E[B][192] (unit-norm embeddings).
 * Every device buffer comes from sdk_device_malloc; everything runs on the NULL stream; sdk_memcpy is complete when it returns. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdk_hip.h"

#define OK(call)                                                                    \
  do {                                                                              \
    if ((call) != 0) {                                                              \
      fprintf(stderr, "%s failed: %s\n", #call, sdk_last_error());                  \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

static void* slurp(const char* path, size_t* bytes) {
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc(n > 0 ? (size_t)n : 1);
  if (fread(p, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read on %s\n", path); exit(2); }
  fclose(f);
  *bytes = (size_t)n;
  return p;
}

int main(int argc, char** argv) {
  if (argc != 9) { fprintf(stderr, "usage: %s pcm.s16 B S blob.bin desc.bin profiles.f32 P out.bin\n", argv[0]); return 2; }
  const int B = atoi(argv[2]), S = atoi(argv[3]), P = atoi(argv[7]), D = 192, LDF = 128;
  const int T = 1 + S / 160;                       /* frames per window: 10-ms hop at 16 kHz */
  size_t n_pcm, n_blob, n_desc, n_prof;
  int16_t* pcm = (int16_t*)slurp(argv[1], &n_pcm);
  void* blob = slurp(argv[4], &n_blob);
  sdk_ecapa_desc* desc = (sdk_ecapa_desc*)slurp(argv[5], &n_desc);
  float* prof = (float*)slurp(argv[6], &n_prof);
  if (n_pcm != (size_t)B * S * 2 || n_desc != sizeof(sdk_ecapa_desc) || n_prof != (size_t)P * D * 4) {
    fprintf(stderr, "input sizes do not match B=%d S=%d P=%d (desc %zu vs %zu)\n", B, S, P, n_desc, sizeof(sdk_ecapa_desc));
    return 2;
  }
  if (sdk_abi_version() != SDK_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", sdk_abi_version(), SDK_ABI_VERSION); return 2; }

  sdk_ctx* ctx = NULL;
  OK(sdk_init(0, &ctx));                           /* what Backend.__init__ does (base.py:291-293) */

  /* resident state: fbank tables, weights */
  const size_t n_tabs = sdk_fbank_tables_bytes();
  void* tabs_h = malloc(n_tabs);
  OK(sdk_fbank_tables_fill(tabs_h, n_tabs));
  void *d_tabs, *d_blob, *d_pcm, *d_feats, *d_emb, *d_E, *d_Eb, *d_re, *d_P, *d_Pn, *d_Pb, *d_rp, *d_rpmax, *d_idx, *d_sc;
  OK(sdk_device_malloc(ctx, n_tabs, &d_tabs));
  OK(sdk_memcpy(ctx, d_tabs, tabs_h, n_tabs, 1, NULL));
  OK(sdk_device_malloc(ctx, n_blob, &d_blob));
  OK(sdk_memcpy(ctx, d_blob, blob, n_blob, 1, NULL));

  /* k1: PCM -> log-mel features [B*T][128] bf16 */
  OK(sdk_device_malloc(ctx, n_pcm, &d_pcm));
  OK(sdk_memcpy(ctx, d_pcm, pcm, n_pcm, 1, NULL));
  OK(sdk_device_malloc(ctx, (size_t)B * T * LDF * 2, &d_feats));
  size_t n_ws = sdk_fbank_workspace_bytes(B, S);
  void* d_ws;
  OK(sdk_device_malloc(ctx, n_ws, &d_ws));
  OK(sdk_fbank(ctx, (const int16_t*)d_pcm, B, S, d_tabs, (uint16_t*)d_feats, LDF, d_ws, n_ws, NULL));
  OK(sdk_device_free(ctx, d_ws));

  /* k2: ECAPA-TDNN forward -> [B][192] fp32 */
  n_ws = sdk_ecapa_workspace_bytes(desc, B, T);
  OK(sdk_device_malloc(ctx, n_ws, &d_ws));
  OK(sdk_device_malloc(ctx, (size_t)B * D * 4, &d_emb));
  OK(sdk_ecapa_forward(ctx, d_blob, desc, (const uint16_t*)d_feats, LDF, B, T, d_ws, n_ws, (float*)d_emb, NULL));
  OK(sdk_device_free(ctx, d_ws));

  /* k3: L2-normalise segments and profiles (fp32 rows, bf16 copy, residual norm) */
  OK(sdk_device_malloc(ctx, (size_t)B * D * 4, &d_E));
  OK(sdk_device_malloc(ctx, (size_t)B * D * 2, &d_Eb));
  OK(sdk_device_malloc(ctx, (size_t)B * 4, &d_re));
  OK(sdk_l2norm(ctx, (const float*)d_emb, B, D, (float*)d_E, (uint16_t*)d_Eb, (float*)d_re, NULL));
  OK(sdk_device_malloc(ctx, n_prof, &d_P));
  OK(sdk_memcpy(ctx, d_P, prof, n_prof, 1, NULL));
  OK(sdk_device_malloc(ctx, n_prof, &d_Pn));
  OK(sdk_device_malloc(ctx, n_prof / 2, &d_Pb));
  OK(sdk_device_malloc(ctx, (size_t)P * 4, &d_rp));
  OK(sdk_l2norm(ctx, (const float*)d_P, P, D, (float*)d_Pn, (uint16_t*)d_Pb, (float*)d_rp, NULL));
  float* rp = (float*)malloc((size_t)P * 4);
  OK(sdk_memcpy(ctx, rp, d_rp, (size_t)P * 4, 2, NULL));
  float rpmax = 0.f;
  for (int i = 0; i < P; ++i) rpmax = rp[i] > rpmax ? rp[i] : rpmax;
  OK(sdk_device_malloc(ctx, 4, &d_rpmax));
  OK(sdk_memcpy(ctx, d_rpmax, &rpmax, 4, 1, NULL));

  /* k4: best profile + exact fp32 cosine per window */
  OK(sdk_device_malloc(ctx, (size_t)B * 4, &d_idx));
  OK(sdk_device_malloc(ctx, (size_t)B * 4, &d_sc));
  n_ws = sdk_affinity_workspace_bytes(B, P);
  OK(sdk_device_malloc(ctx, n_ws, &d_ws));
  OK(sdk_affinity_topk(ctx, (const float*)d_E, (const uint16_t*)d_Eb, (const float*)d_re, (const float*)d_Pn, (const uint16_t*)d_Pb,
                       (const float*)d_rpmax, B, P, D, 1, (int32_t*)d_idx, (float*)d_sc, NULL, d_ws, n_ws, NULL));

  int32_t* idx = (int32_t*)malloc((size_t)B * 4);
  float* sc = (float*)malloc((size_t)B * 4);
  float* E = (float*)malloc((size_t)B * D * 4);
  OK(sdk_memcpy(ctx, idx, d_idx, (size_t)B * 4, 2, NULL));
  OK(sdk_memcpy(ctx, sc, d_sc, (size_t)B * 4, 2, NULL));
  OK(sdk_memcpy(ctx, E, d_E, (size_t)B * D * 4, 2, NULL));
  FILE* f = fopen(argv[8], "wb");
  if (!f) { fprintf(stderr, "cannot write %s\n", argv[8]); return 2; }
  fwrite(idx, 4, (size_t)B, f);
  fwrite(sc, 4, (size_t)B, f);
  fwrite(E, 4, (size_t)B * D, f);
  fclose(f);
  for (int i = 0; i < B && i < 4; ++i) printf("window %d -> profile %d, cosine %.6f\n", i, idx[i], sc[i]);

  void* all[] = {d_tabs, d_blob, d_pcm, d_feats, d_emb, d_E, d_Eb, d_re, d_P, d_Pn, d_Pb, d_rp, d_rpmax, d_idx, d_sc, d_ws};
  for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); ++i) OK(sdk_device_free(ctx, all[i]));
  OK(sdk_shutdown(ctx));
  return 0;
}
