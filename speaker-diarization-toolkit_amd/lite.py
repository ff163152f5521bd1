"""Torch-free host path for the toolkit's process model (SDK_NO_TORCH=1).

The reference constructs its backend fresh in every CLI process (speaker_detection_backends/base.py:291-293) and runs up to four of
them at once (speaker-process:627-629); of the 1.1 s a warm-cache process needs to its first identify row, 0.8 s is `import torch`.
This module drives the SAME library calls as ops.Engine (fbank -> ECAPA-TDNN / x-vector forward -> L2 -> cosine top-k, and the audio
resampler) with numpy on the host side and device memory from the library itself (sdk_device_malloc / sdk_memcpy): no torch, no second
allocator.  Everything runs on the null stream; sdk_memcpy completes before it returns, so results are ready when they reach numpy.

Scope: what Backend.enroll_speaker / identify_speaker / verify_speaker need.  It takes the packed (bias-corrected) weight blob from the
on-disk cache (weights_cache.py); when the entry is missing the caller (backend.py) builds it once in a child process through the torch
engine.  Clustering (k6), multi-GPU and the tuning tools stay on ops.Engine.  A process that uses this module must never import torch
(two HIP runtimes in one process).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import EcapaDesc, SdkError, check
from .weights import DEFAULT_CONFIG, EcapaConfig
from .weights_pack import N_MELS_PADDED

_lib.NO_TORCH = True        # if THIS module is what loads libsdk_hip.so, torch stays out (process-local: the environment is not touched; a library
                            # that ops.Engine already loaded - with torch - is simply reused)

HOP = 160
H2D, D2H = 1, 2


def num_frames(n_samples: int) -> int:
    return 1 + n_samples // HOP


class DevBuf:
    """A device allocation owned by this object (sdk_device_malloc / sdk_device_free)."""

    def __init__(self, eng: "LiteEngine", nbytes: int):
        self.eng, self.nbytes = eng, int(nbytes)
        p = C.c_void_p()
        check(eng.lib.sdk_device_malloc(eng.ctx, self.nbytes, C.byref(p)), "sdk_device_malloc")
        self.ptr = p.value

    def upload(self, host: np.ndarray) -> "DevBuf":
        host = np.ascontiguousarray(host)
        if host.nbytes > self.nbytes:
            raise SdkError(f"upload of {host.nbytes} bytes into a {self.nbytes}-byte device buffer")
        check(self.eng.lib.sdk_memcpy(self.eng.ctx, self.ptr, host.ctypes.data, host.nbytes, H2D, None), "sdk_memcpy")
        return self

    def download(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        if out.nbytes > self.nbytes:
            raise SdkError(f"download of {out.nbytes} bytes from a {self.nbytes}-byte device buffer")
        check(self.eng.lib.sdk_memcpy(self.eng.ctx, out.ctypes.data, self.ptr, out.nbytes, D2H, None), "sdk_memcpy")
        return out

    def free(self) -> None:
        if self.ptr:
            self.eng.lib.sdk_device_free(self.eng.ctx, self.ptr)
            self.ptr = None

    def __del__(self):  # noqa: D105
        try:
            self.free()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class LiteEngine:
    """Resident state of one process on one GPU: context, fbank tables, weight blob, growing scratch buffers."""

    def __init__(self, device: int = 0, cache_key: Optional[str] = None, cfg: EcapaConfig = DEFAULT_CONFIG,
                 bias_correction: Optional[bool] = None, precision: int = 0):
        if precision not in (0, 2):
            raise SdkError(f"the torch-free engine serves the single-plane contracts (precision 0: bf16, 2: fp16), got {precision}")
        self.precision = int(precision)
        self.lib = _lib.load_library()
        self.ctx = _lib.get_ctx(device)
        self.cfg = cfg
        self._cache_key = cache_key
        self.bias_correction = (os.environ.get("SDK_BIAS_CORRECTION", "1") != "0") if bias_correction is None else bool(bias_correction)
        self._scratch: Dict[str, DevBuf] = {}
        self._tabs: Optional[DevBuf] = None
        self._blob: Optional[DevBuf] = None
        self._desc: Optional[EcapaDesc] = None
        self._xv = None                     # (DevBuf blob, XVectorDesc) when the x-vector family is loaded
        self._taps: Dict[Tuple[int, int], DevBuf] = {}
        self._ingest = None
        self.cache_hit = False

    # ---------------------------------------------------------------- resident state
    def _buf(self, key: str, nbytes: int) -> DevBuf:
        b = self._scratch.get(key)
        if b is None or b.nbytes < nbytes:
            if b is not None:
                b.free()
            self._scratch[key] = b = DevBuf(self, max(int(nbytes), 256))
        return b

    def _tables(self) -> DevBuf:
        if self._tabs is None:
            n = self.lib.sdk_fbank_tables_bytes()
            host = np.zeros(n, dtype=np.uint8)
            check(self.lib.sdk_fbank_tables_fill(host.ctypes.data, n), "sdk_fbank_tables_fill")
            self._tabs = DevBuf(self, n).upload(host)
        return self._tabs

    def cache_entry_name(self):
        from .weights_pack import calibration_tag
        return (f"{self.precision}c" + calibration_tag()) if self.bias_correction else self.precision

    def has_cached_weights(self) -> bool:
        """True only if the entry LOADS: meta present, blob present, size / offset table / SHA-256 good (weights_cache.load_blob) - a meta file whose
        blob is missing, truncated or was rewritten for another precision is a miss, and the caller rebuilds the entry."""
        from . import weights_cache
        return bool(self._cache_key) and weights_cache.load_blob(self._cache_key, self.cache_entry_name()) is not None

    def load_weights(self) -> None:
        """ECAPA-TDNN blob from the cache entry ops.Engine wrote (memory-mapped, validated, one upload)."""
        if self._blob is not None:
            return
        from . import weights_cache
        hit = weights_cache.load_blob(self._cache_key, self.cache_entry_name()) if self._cache_key else None
        if hit is None:
            raise SdkError("lite path: no cached packed weights for this model (backend.py builds the entry in a child process first)")
        blob, f = hit
        d = EcapaDesc()
        for k, v in f.items():
            if k == "dilation":
                d.dilation = (C.c_int32 * 4)(*v)
            elif k == "off":
                d.off = (C.c_int64 * 256)(*v)
            else:
                setattr(d, k, v)
        self._blob = DevBuf(self, blob.size).upload(np.asarray(blob))
        self._desc = d
        self.cache_hit = True

    def load_xvector(self, weights: Dict[str, np.ndarray]) -> None:
        """Pack and upload the x-vector family; with the bias correction on (the shipped default) the same calibration pass ops-side XVector runs
        (xvector.calibration_means: library calls only), so both host paths hold bit-identical blobs."""
        from . import xvector
        from .weights_pack import calibration_pcm
        blob, desc = xvector.pack_weights(weights, precision=self.precision)
        dev = DevBuf(self, blob.size).upload(blob)
        if self.bias_correction:
            pcm = calibration_pcm()
            B, S = pcm.shape
            feats = self.fbank(self._buf("pcm", pcm.nbytes).upload(pcm), B, S)
            keep = []

            def alloc(nbytes):
                keep.append(DevBuf(self, nbytes))
                return keep[-1].ptr

            def download(ptr, n_floats):
                return next(k for k in keep if k.ptr == ptr).download(np.float32, (n_floats,))
            means = xvector.calibration_means(self.lib, self.ctx, desc, xvector.DEFAULT_XVECTOR, dev.ptr, feats.ptr, N_MELS_PADDED, B, num_frames(S),
                                              alloc, download, None)
            blob, desc = xvector.pack_weights(dict(weights, **xvector.bias_corrections(weights, means, precision=self.precision)), precision=self.precision)
            dev.free()
            dev = DevBuf(self, blob.size).upload(blob)
            for k in keep:
                k.free()
        self._xv = (dev, desc)

    # ---------------------------------------------------------------- the path (device buffers in, device buffers out)
    def fbank(self, pcm_dev: DevBuf, B: int, S: int) -> DevBuf:
        T = num_frames(S)
        feats = self._buf("feats", B * T * N_MELS_PADDED * 2)
        ws = self._buf("fbank_ws", self.lib.sdk_fbank_workspace_bytes(B, S))
        check(self.lib.sdk_fbank_fmt(self.ctx, pcm_dev.ptr, B, S, self._tables().ptr, feats.ptr, N_MELS_PADDED, ws.ptr, ws.nbytes, self.precision, None), "sdk_fbank")
        return feats

    def fbank_windows(self, samples_ptr: int, n_samples: int, starts_ptr: int, B: int, S: int) -> DevBuf:
        """fbank with the windows cut on the device from a resident recording (sdk_fbank_windows; pointers from the ingest slots)."""
        T = num_frames(S)
        feats = self._buf("feats", B * T * N_MELS_PADDED * 2)
        ws = self._buf("fbank_ws", self.lib.sdk_fbank_workspace_bytes(B, S))
        check(self.lib.sdk_fbank_windows_fmt(self.ctx, samples_ptr, n_samples, starts_ptr, B, S, self._tables().ptr, feats.ptr, N_MELS_PADDED, ws.ptr, ws.nbytes, self.precision, None),
              "sdk_fbank_windows")
        return feats

    def ingest(self):
        if self._ingest is None:
            from .ingest import Ingest
            self._ingest = Ingest(self.lib, self.ctx, depth=2)
        return self._ingest

    def embed_from_host(self, samples: np.ndarray, tables: Dict[int, np.ndarray], step: int = 2048, profiles=None, k: int = 1, score_kw=None):
        """One recording (int16 [n], host) + window-start tables {S: int32 [B_S]} -> {S: (E [B_S, d] fp32 host, idx or None, score or None)}.
        The recording is uploaded ONCE (pinned staging slot); the windows are cut on the device, batch by batch; with `profiles` every batch is
        scored while its embeddings are resident (score_last).  Same results as embed_pcm on host-cut windows, bit for bit."""
        from .ingest import chunk_samples, plan_chunks
        ing = self.ingest()
        samples = np.ascontiguousarray(samples, dtype=np.int16).reshape(-1)
        d = self.cfg.embed_dim
        acc: Dict[int, list] = {}
        for lo, hi, sub in plan_chunks(len(samples), tables, chunk_samples()):    # one piece unless the recording is longer than a staging slot may be
            order = sorted(sub)
            flat = np.concatenate([sub[S][1] for S in order]) if order else np.zeros((0,), np.int32)
            ticket, ds, dw = ing.submit(samples[lo:hi], flat, max(order) if order else 0, None)
            off = 0
            try:
                for S in order:
                    rows, local = sub[S]
                    Bs = len(local)
                    if Bs == 0:
                        continue
                    Es, idxs, scs = [], [], []
                    for a in range(0, Bs, step):
                        b = min(step, Bs - a)
                        emb = self.forward(self.fbank_windows(ds, hi - lo, dw + 4 * (off + a), b, S), b, num_frames(S))
                        self._last = (self.l2norm(emb, b, d, "seg"), b)
                        Es.append(self._last[0][0].download(np.float32, (b, d)))
                        if profiles is not None:
                            i, s_ = self.score_last(profiles, k, **(score_kw or {}))
                            idxs.append(i); scs.append(s_)
                    acc.setdefault(S, []).append((rows, np.concatenate(Es), np.concatenate(idxs) if idxs else None, np.concatenate(scs) if scs else None))
                    off += Bs
            finally:
                ing.release(ticket, None)
        out = {}
        for S in tables:
            got = acc.get(S, [])
            if len(got) == 1 and got[0][0] is None:
                out[S] = got[0][1:]
                continue
            n = len(tables[S])
            E = np.empty((n, d), np.float32)
            idx = np.empty((n, k), np.int32) if profiles is not None and n else None
            sc = np.empty((n, k), np.float32) if profiles is not None and n else None
            for rows, e, i, s_ in got:                                  # scatter every piece's windows to their rows of the table
                E[rows] = e
                if idx is not None:
                    idx[rows] = i; sc[rows] = s_
            out[S] = (E, idx, sc)
        return out

    def forward(self, feats: DevBuf, B: int, T: int) -> DevBuf:
        emb = self._buf("emb", B * self.cfg.embed_dim * 4)
        if self._xv is not None:
            blob, d = self._xv
            ws = self._buf("fwd_ws", self.lib.sdk_xvector_workspace_bytes(C.byref(d), B, T))
            check(self.lib.sdk_xvector_forward(self.ctx, blob.ptr, C.byref(d), feats.ptr, N_MELS_PADDED, B, T, ws.ptr, ws.nbytes, emb.ptr, None),
                  "sdk_xvector_forward")
        else:
            self.load_weights()
            ws = self._buf("fwd_ws", self.lib.sdk_ecapa_workspace_bytes(C.byref(self._desc), B, T))
            check(self.lib.sdk_ecapa_forward(self.ctx, self._blob.ptr, C.byref(self._desc), feats.ptr, N_MELS_PADDED, B, T, ws.ptr, ws.nbytes,
                                             emb.ptr, None), "sdk_ecapa_forward")
        return emb

    def l2norm(self, x: DevBuf, N: int, d: int, tag: str) -> Tuple[DevBuf, DevBuf, DevBuf]:
        E, Eb, r = self._buf(tag + "_E", N * d * 4), self._buf(tag + "_Eb", N * d * 2), self._buf(tag + "_r", N * 4)
        check(self.lib.sdk_l2norm(self.ctx, x.ptr, N, d, E.ptr, Eb.ptr, r.ptr, None), "sdk_l2norm")
        return E, Eb, r

    # ---------------------------------------------------------------- host-level entry points (numpy in, numpy out)
    def embed_pcm(self, pcm: np.ndarray) -> np.ndarray:
        """pcm [B, S] int16 -> unit-norm embeddings [B, d] fp32 (host); the device copies (E, Eb, resid) stay resident for score_last()."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        B, S = pcm.shape
        dev = self._buf("pcm", pcm.nbytes).upload(pcm)
        emb = self.forward(self.fbank(dev, B, S), B, num_frames(S))
        self._last = (self.l2norm(emb, B, self.cfg.embed_dim, "seg"), B)
        return self._last[0][0].download(np.float32, (B, self.cfg.embed_dim))

    def score_last(self, profiles: np.ndarray, k: int = 1, norm=None, on_norm=None, tag=None) -> Tuple[np.ndarray, np.ndarray]:
        """Cosine top-k of the segments of the last embed_pcm() call against `profiles` [P, d] fp32 -> (idx [B, k] int32, score [B, k] fp32).
        norm = (E fp32, Eb bf16 bits, resid) of a packed profile store (store.load_pack): uploaded as they are, no normalisation pass;
        on_norm(E, Eb bits, resid): called once with the device's normalisation of `profiles` (store.publish_pack); tag: identity of the
        profile set - the device copies of the previous call are reused when it repeats (per-bucket / per-label scoring of one batch)."""
        (E, Eb, re), B = self._last
        Pn, d = profiles.shape
        k = min(k, Pn)
        if tag is not None and getattr(self, "_prof_tag", None) == (tag, Pn):
            Pe, Pb, rmax = self._prof_dev
        else:
            if norm is not None:
                nE, nEb, nr = (np.ascontiguousarray(x) for x in norm)
                Pe = self._buf("prof_E", nE.nbytes).upload(nE)
                Pb = self._buf("prof_Eb", nEb.nbytes).upload(nEb)
                r_host = np.asarray(nr, dtype=np.float32)
            else:
                P = np.ascontiguousarray(profiles, dtype=np.float32)
                Pd = self._buf("prof", P.nbytes).upload(P)
                Pe, Pb, rp = self.l2norm(Pd, Pn, d, "prof")
                r_host = rp.download(np.float32, (Pn,))
                if on_norm is not None:
                    on_norm(Pe.download(np.float32, (Pn, d)), Pb.download(np.uint16, (Pn, d)), r_host)
            rmax = self._buf("rpmax", 4).upload(np.array([r_host.max()], dtype=np.float32))
            self._prof_tag, self._prof_dev = ((tag, Pn) if tag is not None else None), (Pe, Pb, rmax)
        idx, sc = self._buf("idx", B * k * 4), self._buf("sc", B * k * 4)
        ws = self._buf("aff_ws", self.lib.sdk_affinity_workspace_bytes(B, Pn))
        check(self.lib.sdk_affinity_topk(self.ctx, E.ptr, Eb.ptr, re.ptr, Pe.ptr, Pb.ptr, rmax.ptr, B, Pn, d, k, idx.ptr, sc.ptr, None,
                                         ws.ptr, ws.nbytes, None), "sdk_affinity_topk")
        return idx.download(np.int32, (B, k)), sc.download(np.float32, (B, k))

    def resample_s16_host(self, x: np.ndarray, rate_in: int, rate_out: int = 16000) -> np.ndarray:
        """x int16 [n] or [n, C] interleaved at rate_in -> mono int16 at rate_out (GPU integer polyphase FIR, bit-exact vs oracle/resample.py)."""
        from . import resample
        x = np.ascontiguousarray(x, dtype=np.int16)
        n_in = x.shape[0]
        ch = 1 if x.ndim == 1 else x.shape[1]
        taps, L, M, K = resample.design_taps(int(rate_in), int(rate_out))
        key = (int(rate_in), int(rate_out))
        if key not in self._taps:
            t = np.array(taps)
            self._taps[key] = DevBuf(self, t.nbytes).upload(t)
        n_out = resample.out_len(n_in, L, M)
        xd = self._buf("rs_in", x.nbytes).upload(x)
        yd = self._buf("rs_out", n_out * 2)
        check(self.lib.sdk_resample_s16(self.ctx, xd.ptr, n_in, ch, self._taps[key].ptr, L, M, K, yd.ptr, n_out, None), "sdk_resample_s16")
        return yd.download(np.int16, (n_out,))
