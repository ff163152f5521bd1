"""Host side of the staging runtime in csrc/ingest.hip: one recording crosses PCIe ONCE, as it lies in the file, next to an int32 table of
window starts; the windows are cut on the device (sdk_fbank_windows).  The boundary being served hands a backend a path and segments
(speaker_detection_backends/base.py:130-151); the reference's cloud backend cuts with ffmpeg per segment list (speechmatics_backend.py:231-281).

ctypes + numpy only (shared by ops.Engine and the torch-free lite.LiteEngine).  Pinned staging, a copy stream of its own and `depth` slots:
with depth 2 the upload of recording i + 1 runs under the forward pass of recording i."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from ._lib import SdkError, check


class Ingest:
    def __init__(self, lib, ctx, max_samples: int = 1 << 22, max_windows: int = 1 << 12, depth: int = 2):
        self.lib, self.ctx, self.depth = lib, ctx, int(depth)
        self._h = None
        self._cap = (0, 0)
        self._open = 0
        self._grow(max_samples, max_windows)

    def _grow(self, n_samples: int, n_windows: int) -> None:
        cap_s = max(self._cap[0], 1 << max(16, int(n_samples - 1).bit_length()))
        cap_w = max(self._cap[1], 1 << max(8, int(max(n_windows, 1) - 1).bit_length()))
        if (cap_s, cap_w) == self._cap:
            return
        if self._open:
            raise SdkError("ingest: a larger recording arrived while slots are still committed (release them first)")
        self.close()
        h = C.c_void_p()
        check(self.lib.sdk_ingest_create(self.ctx, min(cap_s, (1 << 31) - 1), cap_w, self.depth, C.byref(h)), "sdk_ingest_create")
        self._h, self._cap = h, (cap_s, cap_w)

    def close(self) -> None:
        if self._h is not None:
            self.lib.sdk_ingest_destroy(self._h)
            self._h = None

    def __del__(self):  # noqa: D105
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def pinned(self, n_samples: int, n_windows: int) -> Tuple[int, np.ndarray, np.ndarray]:
        """Next slot's pinned buffers as numpy views (fill them, then commit): a reader can `readinto` the sample view - no second host copy."""
        self._grow(n_samples, n_windows)
        t, ps, pw = C.c_int(), C.c_void_p(), C.c_void_p()
        check(self.lib.sdk_ingest_acquire(self._h, C.byref(t), C.byref(ps), C.byref(pw)), "sdk_ingest_acquire")
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
        self._grow(samples.size, starts.size)
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
