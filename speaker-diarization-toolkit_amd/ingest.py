"""Host side of the staging runtime in csrc/ingest.hip: one recording crosses PCIe ONCE, as it lies in the file, next to an int32 table of
window starts; the windows are cut on the device (sdk_fbank_windows).  The boundary being served hands a backend a path and segments
(speaker_detection_backends/base.py:130-151); the reference's cloud backend cuts with ffmpeg per segment list (speechmatics_backend.py:231-281).

ctypes + numpy only (shared by ops.Engine and the torch-free lite.LiteEngine).  Pinned staging, a copy stream of its own and `depth` slots:
with depth 2 the upload of recording i + 1 runs under the forward pass of recording i."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from ._lib import check


def chunk_samples() -> int:
    """Longest piece of a recording that is staged at once (samples; $SDK_INGEST_CHUNK, default 2^25 = 35 minutes at 16 kHz = 64 MiB per slot):
    a longer recording goes through the ring in pieces (plan_chunks), so the page-locked memory of a process is bounded by depth x this,
    whatever the length of the file (speaker-process:627-629 runs up to four such processes)."""
    import os
    return max(1 << 16, int(os.environ.get("SDK_INGEST_CHUNK", str(1 << 25))))


def plan_chunks(n_samples: int, tables, cap: int):
    """Cut one recording + its window-start tables {S: int32 [B_S]} into uploads of at most `cap` samples.
    -> [(lo, hi, {S: (rows int64 [b], local starts int32 [b])})]: piece [lo, hi) of the recording and, per window length, which rows of the
    table start inside it and where (relative to lo).  A window belongs to the piece that holds its START; pieces overlap by the longest window,
    so every window lies wholly inside its piece - or runs past the END of the recording, where it reads zeros exactly as in the one-piece form
    (sdk_fbank_windows).  One piece (the usual case) returns the tables unchanged."""
    order = sorted(tables)
    smax = max(order) if order else 0
    if n_samples <= cap or not order:
        return [(0, n_samples, {S: (None, np.ascontiguousarray(tables[S], dtype=np.int32)) for S in order})]
    if cap <= 2 * smax:
        raise ValueError(f"ingest: chunk of {cap} samples is too short for windows of {smax}")
    L = cap - smax                                   # starts per piece: [c L, (c + 1) L)
    out = []
    for c in range((n_samples + L - 1) // L):
        lo, hi = c * L, min(n_samples, (c + 1) * L + smax)
        sub = {}
        for S in order:
            st = np.asarray(tables[S], dtype=np.int64)
            rows = np.nonzero((st >= lo) & (st < lo + L))[0]
            if rows.size:
                sub[S] = (rows, (st[rows] - lo).astype(np.int32))
        if sub:
            out.append((lo, hi, sub))
    return out


class Ingest:
    def __init__(self, lib, ctx, max_samples: int = (1 << 31) - 1, max_windows: int = 1 << 24, depth: int = 2):
        """max_samples / max_windows are upper bounds only: a slot's buffers are allocated on its first use, at the size of that upload."""
        self.lib, self.ctx, self.depth = lib, ctx, int(depth)
        self._h = None
        self._open = 0
        h = C.c_void_p()
        check(self.lib.sdk_ingest_create(self.ctx, min(int(max_samples), (1 << 31) - 1), int(max_windows), self.depth, C.byref(h)), "sdk_ingest_create")
        self._h = h

    def close(self) -> None:
        if self._h is not None:
            self.lib.sdk_ingest_destroy(self._h)
            self._h = None

    def __del__(self):  # noqa: D105
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def slot_info(self, slot: int) -> Tuple[int, bool]:
        """(capacity in samples - 0: never used, nothing allocated -, staging memory is page-locked)"""
        cap, pinned = C.c_int64(), C.c_int()
        check(self.lib.sdk_ingest_slot_info(self._h, slot, C.byref(cap), C.byref(pinned)), "sdk_ingest_slot_info")
        return cap.value, bool(pinned.value)

    def pinned(self, n_samples: int, n_windows: int) -> Tuple[int, np.ndarray, np.ndarray]:
        """Next slot's pinned buffers as numpy views (fill them, then commit): a reader can `readinto` the sample view - no second host copy."""
        t, ps, pw = C.c_int(), C.c_void_p(), C.c_void_p()
        check(self.lib.sdk_ingest_acquire_sized(self._h, n_samples, n_windows, C.byref(t), C.byref(ps), C.byref(pw)), "sdk_ingest_acquire")
        s = np.ctypeslib.as_array(C.cast(ps, C.POINTER(C.c_int16)), shape=(n_samples,))
        w = np.ctypeslib.as_array(C.cast(pw, C.POINTER(C.c_int32)), shape=(max(n_windows, 1),))[:n_windows]
        return t.value, s, w

    def commit(self, ticket: int, n_samples: int, n_windows: int, window_len: int, stream: Optional[int]) -> Tuple[int, int]:
        ds, dw = C.c_void_p(), C.c_void_p()
        check(self.lib.sdk_ingest_commit(self._h, ticket, n_samples, n_windows, window_len, stream, C.byref(ds), C.byref(dw)), "sdk_ingest_commit")
        self._open += 1
        return ds.value, dw.value

    def submit(self, samples: np.ndarray, starts: np.ndarray, window_len: int, stream: Optional[int]) -> Tuple[int, int, int]:
        """samples int16 [n] and starts int32 [B] in pageable host memory -> (ticket, device samples pointer, device start-table pointer)."""
        samples = np.ascontiguousarray(samples, dtype=np.int16).reshape(-1)
        starts = np.ascontiguousarray(starts, dtype=np.int32).reshape(-1)
        t, ds, dw = C.c_int(), C.c_void_p(), C.c_void_p()
        check(self.lib.sdk_ingest_submit(self._h, samples.ctypes.data, samples.size, starts.ctypes.data if starts.size else None, starts.size, window_len,
                                         stream, C.byref(t), C.byref(ds), C.byref(dw)), "sdk_ingest_submit")
        self._open += 1
        return t.value, ds.value, dw.value

    def release(self, ticket: int, stream: Optional[int]) -> None:
        check(self.lib.sdk_ingest_release(self._h, ticket, stream), "sdk_ingest_release")
        self._open -= 1

    def copy_ms(self, ticket: int) -> Tuple[float, float]:
        ms, nb = C.c_float(), C.c_double()
        check(self.lib.sdk_ingest_copy_ms(self._h, ticket, C.byref(ms), C.byref(nb)), "sdk_ingest_copy_ms")
        return ms.value, nb.value
