// ingest: host audio -> HBM at batch scale.
//
// The plug-in boundary hands a backend a PATH and segments (speaker_detection_backends/base.py:130-151; the reference's cloud backend cuts
// the file with ffmpeg per segment list, speechmatics_backend.py:231-281).  Here a recording crosses PCIe ONCE, as it lies in the file, plus an
// int32 table of window starts; the windows are cut on the device by sdk_fbank_windows.  This file is the staging runtime around that:
//
//   sdk_ingest          `depth` slots, each = {pinned host buffers for samples and window starts, their device twins, two events};
//                       one copy stream of its own.  A slot's buffers are allocated on its FIRST use and sized to the upload it is asked to take
//                       (rounded up, grown when a larger one arrives): a single-recording CLI call (base.py:130-151, one identify per process)
//                       touches one slot of the recording's size, not `depth` slots of the largest size the ring may ever see.
//                       Where the host refuses page-locked memory the slot stages through ordinary memory (slower upload, same results)
//   acquire             next slot in ring order; waits (host side) only until the slot's PREVIOUS upload has left its pinned buffers
//   [the caller fills the pinned buffers - a WAVE reader can readinto() them: no second host copy - or lets submit() memcpy]
//   commit              H2D copies on the copy stream, issued behind the slot's previous consumer (stream wait on `consumed`, no host block);
//                       the caller's compute stream is made to wait for them (`copied`); returns the device pointers
//   release             marks the point on the compute stream after which the slot's device buffers may be overwritten
//
// With depth >= 2 the upload of recording i + 1 (host memcpy into pinned memory + DMA) runs under the forward pass of recording i.
// No kernel here: HIP runtime calls only; the library's other entry points take the returned device pointers as they take any others.
#include <stdlib.h>
#include <string.h>

#include <thread>

#include "common.hpp"

// Pageable -> pinned staging copy.  One core moves ~10 GB/s; a 64 MB batch (1000 two-second segments) would then take most of the 8.9 ms its
// forward pass needs, so large copies are cut over a few threads (the GPU box gives a process 16 cores per GPU).
static void staging_copy(void* dst, const void* src, size_t bytes) {
  constexpr size_t kMin = 8u << 20;
  const int nt = bytes < 2 * kMin ? 1 : (int)(bytes / kMin < 4 ? bytes / kMin : 4);
  if (nt <= 1) { memcpy(dst, src, bytes); return; }
  std::thread th[4];
  const size_t chunk = ((bytes / nt) + 4095) & ~(size_t)4095;
  for (int i = 1; i < nt; ++i) {
    const size_t o = (size_t)i * chunk, len = o >= bytes ? 0 : (bytes - o < chunk ? bytes - o : chunk);
    th[i] = std::thread([=] { if (len) memcpy((char*)dst + o, (const char*)src + o, len); });
  }
  memcpy(dst, src, chunk < bytes ? chunk : bytes);
  for (int i = 1; i < nt; ++i) th[i].join();
}

struct sdk_ingest {
  sdk_ctx* ctx = nullptr;
  int depth = 0;
  int64_t max_samples = 0;
  int max_windows = 0;
  hipStream_t copy = nullptr;
  struct Slot {
    int16_t* h_s = nullptr; int32_t* h_w = nullptr;
    int16_t* d_s = nullptr; int32_t* d_w = nullptr;
    int64_t cap_s = 0; int cap_w = 0;   // what the buffers hold now (0: not allocated yet)
    bool pinned = true;                 // false: hipHostMalloc was refused, h_s / h_w come from malloc
    hipEvent_t copy_begin = nullptr, copied = nullptr, consumed = nullptr;
    int state = 0;            // 0 free, 1 acquired (being filled), 2 committed (in use by the compute stream), 3 released (consumed event recorded)
    bool ever_copied = false, ever_consumed = false;
    int64_t n = 0; int B = 0;
  } slot[8];
  int next = 0;
  double bytes_total = 0.0;
};

static void slot_free(sdk_ingest::Slot& s) {
  if (s.pinned) {
    if (s.h_s) (void)hipHostFree(s.h_s);
    if (s.h_w) (void)hipHostFree(s.h_w);
  } else {
    free(s.h_s);
    free(s.h_w);
  }
  if (s.d_s) (void)hipFree(s.d_s);
  if (s.d_w) (void)hipFree(s.d_w);
  s.h_s = nullptr; s.h_w = nullptr; s.d_s = nullptr; s.d_w = nullptr;
  s.cap_s = 0; s.cap_w = 0; s.pinned = true;
}

// Make the slot hold n_samples / n_windows.  Called with the slot idle on the host side (its previous upload has left the staging buffers);
// its previous CONSUMER may still be running on the device, so growing waits for that one event - the other slots and both streams keep going.
static int slot_reserve(sdk_ingest* g, sdk_ingest::Slot& s, int64_t n_samples, int n_windows) {
  if (n_samples <= s.cap_s && n_windows <= s.cap_w) return 0;
  if (s.ever_consumed) SDK_HIP_OK(hipEventSynchronize(s.consumed));
  // sizes: at least what is asked, rounded up to 2^20 samples (2 MiB) / 4096 windows so that recordings of similar length reuse the buffers
  int64_t cs = n_samples > s.cap_s ? n_samples : s.cap_s;
  int cw = n_windows > s.cap_w ? n_windows : s.cap_w;
  cs = (cs + (1 << 20) - 1) & ~(int64_t)((1 << 20) - 1);
  cw = (cw + 4095) & ~4095;
  if (cs > g->max_samples) cs = g->max_samples;
  if (cw > g->max_windows) cw = g->max_windows;
  slot_free(s);
  const bool try_pinned = !getenv("SDK_INGEST_NO_PINNED");               // (tests: the pageable fallback without exhausting the host's lock limit)
  if (!try_pinned || hipHostMalloc((void**)&s.h_s, (size_t)cs * 2, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&s.h_w, (size_t)cw * 4, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    if (s.h_s) { (void)hipHostFree(s.h_s); s.h_s = nullptr; }
    s.pinned = false;
    s.h_s = (int16_t*)malloc((size_t)cs * 2);
    s.h_w = (int32_t*)malloc((size_t)cw * 4);
    if (!s.h_s || !s.h_w) { slot_free(s); SDK_REQUIRE(false, "sdk_ingest: no host memory for a staging slot of %lld samples", (long long)cs); }
  }
  hipError_t e = hipMalloc((void**)&s.d_s, (size_t)cs * 2);
  if (e == hipSuccess) e = hipMalloc((void**)&s.d_w, (size_t)cw * 4);
  if (e != hipSuccess) { slot_free(s); SDK_REQUIRE(false, "sdk_ingest: hipMalloc of a %lld-sample slot failed: %s", (long long)cs, hipGetErrorString(e)); }
  s.cap_s = cs; s.cap_w = cw;
  return 0;
}

extern "C" int sdk_ingest_destroy(sdk_ingest* g) {
  if (!g) return 0;
  if (g->copy) (void)hipStreamSynchronize(g->copy);
  for (int i = 0; i < g->depth; ++i) {
    auto& s = g->slot[i];
    if (s.consumed && s.ever_consumed) (void)hipEventSynchronize(s.consumed);
    slot_free(s);
    if (s.copy_begin) (void)hipEventDestroy(s.copy_begin);
    if (s.copied) (void)hipEventDestroy(s.copied);
    if (s.consumed) (void)hipEventDestroy(s.consumed);
  }
  if (g->copy) (void)hipStreamDestroy(g->copy);
  delete g;
  return 0;
}

extern "C" int sdk_ingest_create(sdk_ctx* ctx, int64_t max_samples, int max_windows, int depth, sdk_ingest** out) {
  SDK_REQUIRE(ctx && out, "sdk_ingest_create: null argument");
  *out = nullptr;
  SDK_REQUIRE(max_samples > 0 && max_samples < (1ll << 31) && max_windows > 0, "sdk_ingest_create: max_samples=%lld (1 .. 2^31-1), max_windows=%d",
              (long long)max_samples, max_windows);
  SDK_REQUIRE(depth >= 1 && depth <= 8, "sdk_ingest_create: depth=%d must be 1..8", depth);
  SDK_HIP_OK(hipSetDevice(ctx->device));
  sdk_ingest* g = new sdk_ingest();
  g->ctx = ctx; g->depth = depth; g->max_samples = max_samples; g->max_windows = max_windows;
#define ING_OK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { sdk_set_error("%s failed: %s", #expr, hipGetErrorString(_e)); sdk_ingest_destroy(g); return 1; } } while (0)
  ING_OK(hipStreamCreateWithFlags(&g->copy, hipStreamNonBlocking));
  for (int i = 0; i < depth; ++i) {
    auto& s = g->slot[i];
    ING_OK(hipEventCreate(&s.copy_begin));
    ING_OK(hipEventCreate(&s.copied));
    ING_OK(hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming));
  }
#undef ING_OK
  *out = g;
  return 0;
}

extern "C" int sdk_ingest_acquire(sdk_ingest* g, int* ticket, int16_t** pinned_samples, int32_t** pinned_starts) {
  return sdk_ingest_acquire_sized(g, g ? g->max_samples : 0, g ? g->max_windows : 0, ticket, pinned_samples, pinned_starts);
}

extern "C" int sdk_ingest_acquire_sized(sdk_ingest* g, int64_t n_samples, int n_windows, int* ticket, int16_t** pinned_samples, int32_t** pinned_starts) {
  SDK_REQUIRE(g && ticket && pinned_samples && pinned_starts, "sdk_ingest_acquire: null argument");
  SDK_REQUIRE(n_samples > 0 && n_samples <= g->max_samples && n_windows >= 0 && n_windows <= g->max_windows,
              "sdk_ingest_acquire: %lld samples / %d windows do not fit the slots (%lld / %d)", (long long)n_samples, n_windows,
              (long long)g->max_samples, g->max_windows);
  auto& s = g->slot[g->next];
  // (a slot that was acquired but never committed - its filler failed - has nothing enqueued and is simply handed out again)
  SDK_REQUIRE(s.state != 2, "sdk_ingest_acquire: all %d slots are in flight (slot %d was committed but never released: call sdk_ingest_release "
              "after the last kernel that reads it has been enqueued)", g->depth, g->next);
  if (s.ever_copied) SDK_HIP_OK(hipEventSynchronize(s.copied));     // the previous upload has left the pinned buffers (it finished long ago unless depth == 1)
  SDK_HIP_OK(hipSetDevice(g->ctx->device));
  if (int rc = slot_reserve(g, s, n_samples, n_windows > 0 ? n_windows : 1)) return rc;
  s.state = 1;
  *ticket = g->next;
  *pinned_samples = s.h_s;
  *pinned_starts = s.h_w;
  g->next = (g->next + 1) % g->depth;
  return 0;
}

extern "C" int sdk_ingest_commit(sdk_ingest* g, int ticket, int64_t n_samples, int n_windows, int window_len, void* compute_stream,
                                 const int16_t** dev_samples, const int32_t** dev_starts) {
  SDK_REQUIRE(g && dev_samples && dev_starts, "sdk_ingest_commit: null argument");
  SDK_REQUIRE(ticket >= 0 && ticket < g->depth && g->slot[ticket].state == 1, "sdk_ingest_commit: ticket %d is not an acquired slot", ticket);
  auto& s = g->slot[ticket];
  if (n_samples <= 0 || n_samples > s.cap_s || n_windows < 0 || n_windows > s.cap_w) {
    s.state = 0;
    SDK_REQUIRE(false, "sdk_ingest_commit: %lld samples / %d windows do not fit the slot as acquired (%lld / %d)", (long long)n_samples, n_windows,
                (long long)s.cap_s, s.cap_w);
  }
  // the start table is dereferenced on the device: every window must START inside the recording (window_len > 0: and end no further than one
  // window past it - what lies beyond the end reads as zero, sdk_fbank_windows)
  for (int b = 0; b < n_windows; ++b) {
    const int32_t w = s.h_w[b];
    if (w < 0 || (int64_t)w >= n_samples) {
      s.state = 0;
      SDK_REQUIRE(false, "sdk_ingest_commit: window %d starts at sample %d, outside the recording of %lld samples", b, (int)w, (long long)n_samples);
    }
  }
  (void)window_len;
  SDK_HIP_OK(hipSetDevice(g->ctx->device));
  if (s.ever_consumed) SDK_HIP_OK(hipStreamWaitEvent(g->copy, s.consumed, 0));     // the kernels that read the slot's previous contents
  SDK_HIP_OK(hipEventRecord(s.copy_begin, g->copy));
  SDK_HIP_OK(hipMemcpyAsync(s.d_s, s.h_s, (size_t)n_samples * 2, hipMemcpyHostToDevice, g->copy));
  if (n_windows) SDK_HIP_OK(hipMemcpyAsync(s.d_w, s.h_w, (size_t)n_windows * 4, hipMemcpyHostToDevice, g->copy));
  SDK_HIP_OK(hipEventRecord(s.copied, g->copy));
  SDK_HIP_OK(hipStreamWaitEvent((hipStream_t)compute_stream, s.copied, 0));
  s.ever_copied = true;
  s.state = 2;
  s.n = n_samples; s.B = n_windows;
  g->bytes_total += (double)n_samples * 2 + (double)n_windows * 4;
  *dev_samples = s.d_s;
  *dev_starts = s.d_w;
  return 0;
}

extern "C" int sdk_ingest_submit(sdk_ingest* g, const int16_t* host_samples, int64_t n_samples, const int32_t* host_starts, int n_windows,
                                 int window_len, void* compute_stream, int* ticket, const int16_t** dev_samples, const int32_t** dev_starts) {
  SDK_REQUIRE(g && host_samples && (host_starts || n_windows == 0) && ticket, "sdk_ingest_submit: null argument");
  SDK_REQUIRE(n_samples > 0 && n_samples <= g->max_samples && n_windows >= 0 && n_windows <= g->max_windows,
              "sdk_ingest_submit: %lld samples / %d windows do not fit the slots (%lld / %d)", (long long)n_samples, n_windows,
              (long long)g->max_samples, g->max_windows);
  int16_t* ps; int32_t* pw;
  if (int rc = sdk_ingest_acquire_sized(g, n_samples, n_windows, ticket, &ps, &pw)) return rc;
  staging_copy(ps, host_samples, (size_t)n_samples * 2);
  if (n_windows) memcpy(pw, host_starts, (size_t)n_windows * 4);
  return sdk_ingest_commit(g, *ticket, n_samples, n_windows, window_len, compute_stream, dev_samples, dev_starts);
}

extern "C" int sdk_ingest_release(sdk_ingest* g, int ticket, void* compute_stream) {
  SDK_REQUIRE(g && ticket >= 0 && ticket < g->depth && g->slot[ticket].state == 2, "sdk_ingest_release: ticket %d is not a committed slot", ticket);
  auto& s = g->slot[ticket];
  SDK_HIP_OK(hipEventRecord(s.consumed, (hipStream_t)compute_stream));
  s.ever_consumed = true;
  s.state = 3;
  return 0;
}

// Duration of the slot's last upload (copy stream, HIP events) - synchronises with that upload.  For bench.py's achieved PCIe rate.
extern "C" int sdk_ingest_copy_ms(sdk_ingest* g, int ticket, float* ms, double* bytes) {
  SDK_REQUIRE(g && ms && ticket >= 0 && ticket < g->depth && g->slot[ticket].ever_copied, "sdk_ingest_copy_ms: ticket %d has no upload", ticket);
  auto& s = g->slot[ticket];
  SDK_HIP_OK(hipEventSynchronize(s.copied));
  SDK_HIP_OK(hipEventElapsedTime(ms, s.copy_begin, s.copied));
  if (bytes) *bytes = (double)s.n * 2 + (double)s.B * 4;
  return 0;
}

// What a slot holds right now: its capacity in samples (0 = never used: nothing allocated) and whether its staging memory is page-locked.
extern "C" int sdk_ingest_slot_info(sdk_ingest* g, int slot, int64_t* cap_samples, int* pinned) {
  SDK_REQUIRE(g && slot >= 0 && slot < g->depth, "sdk_ingest_slot_info: slot %d of %d", slot, g ? g->depth : 0);
  if (cap_samples) *cap_samples = g->slot[slot].cap_s;
  if (pinned) *pinned = g->slot[slot].pinned ? 1 : 0;
  return 0;
}
