"""Host-side wrappers over the C ABI: torch tensors in, torch tensors out, every byte of compute
inside libsdk_hip.so.  torch supplies device memory and the current HIP stream only.

`Engine` is the long-lived per-process, per-GPU object (context + resident weights + fbank
tables + reusable scratch): the reference constructs its backend once per CLI process
(speaker_detection_backends/base.py:291-293), so everything expensive lives here and is built
lazily exactly once.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib, resample
from ._lib import ConvGemmArgs, EcapaDesc, SdkError, check
from .weights import DEFAULT_CONFIG, EcapaConfig, synthetic_weights
from .weights_pack import N_MELS_PADDED, N_MELS_PADDED_HP, pack_weights

HOP = 160
EMBED_DIM = 192


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise SdkError(f"{name} must live in device memory (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise SdkError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def num_frames(n_samples: int) -> int:
    return 1 + n_samples // HOP


class Engine:
    def __init__(self, device: int = 0, weights: Optional[Dict[str, np.ndarray]] = None,
                 cfg: EcapaConfig = DEFAULT_CONFIG, seed: int = 0, cache_key: Optional[str] = None, weights_fn=None, digest_fn=None,
                 bias_correction: Optional[bool] = None):
        """weights: host dict (weights.py naming) or None; weights_fn: called only when the dict is really needed (cache miss);
        cache_key / digest_fn: identity of the weights for the packed-blob cache (weights_cache.py) - None = no caching.
        bias_correction (default mode only): fold the constant part of the bf16 weight-rounding error of every GEMM layer into its bias,
        from a calibration pass on built-in synthetic audio (weights_pack.bias_corrections; PCM -> score deviation from the fp32 model
        4.3e-3 -> ~8e-4 at no run-time cost).  False = the plain bf16 layer-boundary model (kernel parity tests).  None = $SDK_BIAS_CORRECTION
        (default "1")."""
        self.lib = _lib.load_library()
        self.ctx = _lib.get_ctx(device)        # raises SdkError without a gfx950 device
        self.device = torch.device("cuda", device)
        self.cfg = cfg
        self._weights_host = weights
        self._weights_fn = weights_fn
        self._cache_key = cache_key
        self._digest_fn = digest_fn
        self.cache_hit = False
        import os
        self.bias_correction = (os.environ.get("SDK_BIAS_CORRECTION", "1") != "0") if bias_correction is None else bool(bias_correction)
        self._bias_host: Dict[str, np.ndarray] = {}        # layer name -> corrected bias of the CURRENT single-plane mode, for effective_weights()
        self._bias_by_precision: Dict[int, Dict[str, np.ndarray]] = {}
        self._seed = seed
        self._wblob = None
        self._desc = None
        self._packed = {}                 # precision -> (device blob, desc): each numerical contract has its own weight format
        self.precision = 0
        self._fbank_tabs = None
        self._scratch: Dict[str, torch.Tensor] = {}
        self._graphs: Dict[tuple, tuple] = {}
        self._streams = []
        self._taps: Dict[tuple, torch.Tensor] = {}
        self._ingest = None

    # ------------------------------------------------------------------ resident state
    def _scratch_bytes(self, key: str, nbytes: int) -> torch.Tensor:
        key = f"{key}@{torch.cuda.current_stream().cuda_stream}"       # scratch is per stream: sub-batches overlap
        buf = self._scratch.get(key)
        if buf is None or buf.numel() < nbytes:
            self._scratch[key] = buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        return buf

    def fbank_tables(self) -> torch.Tensor:
        if self._fbank_tabs is None:
            n = self.lib.sdk_fbank_tables_bytes()
            host = np.zeros(n, dtype=np.uint8)
            check(self.lib.sdk_fbank_tables_fill(host.ctypes.data, n), "sdk_fbank_tables_fill")
            self._fbank_tabs = torch.from_numpy(host).to(self.device)
        return self._fbank_tabs

    def load_weights(self, weights: Optional[Dict[str, np.ndarray]] = None) -> None:
        if weights is not None:
            self._weights_host = weights
            self._packed.clear()
            self._cache_key = None                      # explicit weights: their identity is unknown to the cache
        if self.precision not in self._packed:
            from . import weights_cache
            correct = self.bias_correction and self.precision in (0, 2)
            from .weights_pack import calibration_tag
            ckey = (f"{self.precision}c" + calibration_tag()) if correct else self.precision   # a bias-corrected blob is its own cache entry (per calibration recording)
            hit = weights_cache.load_blob(self._cache_key, ckey) if self._cache_key else None
            need_store = False
            if hit is not None:
                blob, f = hit                            # memory-mapped: the upload below is the only pass over the bytes
                self.cache_hit = True
            else:
                if self._weights_host is None:
                    self._weights_host = self._weights_fn() if self._weights_fn else synthetic_weights(self._seed, self.cfg)
                blob, f = pack_weights(self._weights_host, self.cfg, precision=self.precision)
                need_store = bool(self._cache_key and self._digest_fn)
            d = EcapaDesc()
            for k, v in f.items():
                if k == "dilation":
                    d.dilation = (C.c_int32 * 4)(*v)
                elif k == "off":
                    d.off = (C.c_int64 * 256)(*v)
                else:
                    setattr(d, k, v)
            import warnings
            with warnings.catch_warnings():          # a cache hit is a read-only memory map: it is only read (uploaded) here
                warnings.simplefilter("ignore", UserWarning)
                host = torch.from_numpy(np.asarray(blob))
            self._packed[self.precision] = (host.to(self.device), d)
            if correct and hit is None:
                blob = self._apply_bias_correction(np.array(blob, copy=True), f)
            if need_store:
                weights_cache.store(self._cache_key, self._digest_fn(), ckey, blob, f)
        self._wblob, self._desc = self._packed[self.precision]

    # ------------------------------------------------------------------ bias correction of the bf16 weight rounding (default mode)
    @staticmethod
    def calibration_pcm(n: int = 24, seed: int = 20240) -> np.ndarray:
        """Built-in calibration audio (weights_pack.calibration_pcm: torch-free, shared with the lite host path)."""
        from .weights_pack import calibration_pcm
        return calibration_pcm(n, seed)

    def _apply_bias_correction(self, blob: np.ndarray, fields: dict) -> np.ndarray:
        """One calibration forward with the plain blob (already uploaded), corrected biases computed on the host from the measured channel means of
        every GEMM layer's input, written into the device blob and into `blob` (returned, for the cache)."""
        from .weights_pack import bias_corrections, bias_slot, calib_layout
        wdev, d = self._packed[self.precision]
        pcm = torch.from_numpy(self.calibration_pcm()).to(self.device)
        B, S = pcm.shape
        T = num_frames(S)
        saved = (self._wblob, self._desc)
        self._wblob, self._desc = wdev, d
        try:
            feats = self.fbank(pcm)
            nfl = self.lib.sdk_ecapa_calib_floats(C.byref(d), B)
            calib = torch.zeros((nfl,), dtype=torch.float32, device=self.device)
            ws = self._scratch_bytes("ecapa", self.lib.sdk_ecapa_workspace_bytes(C.byref(d), B, T))
            emb = torch.empty((B, self.cfg.embed_dim), dtype=torch.float32, device=self.device)
            check(self.lib.sdk_ecapa_forward_calib(self.ctx, wdev.data_ptr(), C.byref(d), feats.data_ptr(), feats.stride(0), B, T, ws.data_ptr(),
                                                   ws.numel(), emb.data_ptr(), calib.data_ptr(), _stream()), "sdk_ecapa_forward_calib")
            cal = calib.cpu().numpy().astype(np.float64)
        finally:
            self._wblob, self._desc = saved
        layout, per_seg = calib_layout(self.cfg)
        assert per_seg * B == nfl, (per_seg, B, nfl)
        means = {name: cal[off * B:(off + 2 * ch) * B].reshape(B, 2 * ch)[:, :ch].mean(axis=0) for name, ch, off in layout}
        self._bias_by_precision[self.precision] = self._bias_host = bias_corrections(self._weights_host, means, self.cfg, precision=self.precision)
        for name, bias in self._bias_host.items():
            o = fields["off"][bias_slot(name, self.cfg)]
            raw = np.ascontiguousarray(bias, dtype=np.float32).view(np.uint8)
            blob[o:o + raw.size] = raw
            wdev[o:o + raw.size] = torch.from_numpy(raw.copy()).to(self.device)
        return blob

    def effective_weights(self) -> Dict[str, np.ndarray]:
        """The weight dictionary whose bf16 layer-boundary model the default mode computes: the loaded weights with the corrected biases
        (identical to the loaded weights when bias_correction is off).  For checkers (tests, smoke): the product never needs it."""
        self.desc
        if self._weights_host is None:
            self._weights_host = self._weights_fn() if self._weights_fn else synthetic_weights(self._seed, self.cfg)
        w = dict(self._weights_host)
        if self.bias_correction and self.precision in (0, 2):
            self._bias_host = self._bias_by_precision.setdefault(self.precision, {})
            if not self._bias_host:                   # cache hit: read the corrected biases back from the device blob
                from .weights_pack import bias_slot, calib_layout
                for name, ch, _ in calib_layout(self.cfg)[0]:
                    n_out = self._weights_host[f"{name}.conv.b"].shape[0]
                    o = int(self._desc.off[bias_slot(name, self.cfg)])
                    self._bias_host[name] = self._wblob[o:o + 4 * n_out].cpu().numpy().view(np.float32).copy()
            for name, b in self._bias_host.items():
                w[f"{name}.conv.b"] = b
        return w

    def set_precision(self, precision: int) -> None:
        """0 (default): bf16 GEMM operands, bf16 layer-boundary storage - PCM -> cosine score within ~4e-3 of the fp32 model.
        1: the precise mode (csrc/hp.hip): fp16 hi+lo planes and three MFMAs per product everywhere, within 1e-5 (north_star's
        tolerance), ~3x the GEMM time.  Selects the fbank output format and the weight blob together; embeddings of the two modes
        are comparable with each other at the 4e-3 level only.
        2 (round 5): one fp16 plane - the default mode's kernels and schedule with fp16 instead of bf16 storage and MFMA operands (11
        significand bits instead of 8): PCM -> score within ~1e-4 of the fp32 model with the bias correction, at the default mode's MFMA
        count (the chip holds a ~5 % lower clock on fp16 products: profiles/r05_gemm_f16_probe.txt)."""
        if precision not in (0, 1, 2):
            raise SdkError(f"precision must be 0, 1 or 2, got {precision}")
        self.precision = precision          # (every call names its format: the library context holds no per-engine state)
        self._wblob = self._desc = None
        self._graphs.clear()

    @property
    def desc(self) -> EcapaDesc:
        if self._desc is None:
            self.load_weights()
        return self._desc

    def debug_ptr(self, name: str, tensor: Optional[torch.Tensor]) -> None:
        """Diagnostics: hand a device buffer to (or, with None, take it away from) an in-kernel probe (sdk_debug_set_ptr)."""
        check(self.lib.sdk_debug_set_ptr(self.ctx, name.encode(), tensor.data_ptr() if tensor is not None else None), "sdk_debug_set_ptr")

    def set_option(self, name: str, value: int) -> None:
        check(self.lib.sdk_set_option(self.ctx, name.encode(), int(value)), "sdk_set_option")

    # ------------------------------------------------------------------ measurement
    def profile_begin(self) -> None:
        check(self.lib.sdk_profile_begin(self.ctx), "sdk_profile_begin")

    def profile_end(self) -> Dict[str, Dict[str, float]]:
        """Per kernel family: launches, summed device ms (HIP events on the launch stream), flops, bytes."""
        rep = _lib.ProfileReport()
        check(self.lib.sdk_profile_end(self.ctx, C.byref(rep)), "sdk_profile_end")
        return {name: {"launches": rep.launches[i], "ms": rep.ms[i], "flops": rep.flops[i], "bytes": rep.bytes[i]}
                for i, name in enumerate(_lib.KERNEL_FAMILIES) if rep.launches[i]}

    # ------------------------------------------------------------------ audio conversion
    def resample_s16(self, x: torch.Tensor, rate_in: int, rate_out: int = 16000) -> torch.Tensor:
        """x int16 [n] or [n, C] interleaved (device) at rate_in -> mono int16 [ceil(n*L/M)] at rate_out."""
        _need(x, torch.int16, "x")
        x = x.contiguous()
        n_in = x.shape[0]
        ch = 1 if x.dim() == 1 else x.shape[1]
        taps, L, M, K = resample.design_taps(int(rate_in), int(rate_out))
        key = (int(rate_in), int(rate_out))
        tdev = self._taps.get(key)
        if tdev is None:
            tdev = self._taps[key] = torch.from_numpy(np.array(taps)).to(self.device)
        n_out = resample.out_len(n_in, L, M)
        y = torch.empty((n_out,), dtype=torch.int16, device=self.device)
        check(self.lib.sdk_resample_s16(self.ctx, x.data_ptr(), n_in, ch, tdev.data_ptr(), L, M, K, y.data_ptr(), n_out, _stream()),
              "sdk_resample_s16")
        return y

    def resample_s16_host(self, x: np.ndarray, rate_in: int, rate_out: int = 16000) -> np.ndarray:
        """Host arrays in and out (wav.decode_to_profile; lite.LiteEngine has the same method)."""
        return self.resample_s16(torch.from_numpy(np.ascontiguousarray(x)).to(self.device), rate_in, rate_out).cpu().numpy()

    # ------------------------------------------------------------------ k1
    def fbank(self, pcm: torch.Tensor, ldf: Optional[int] = None) -> torch.Tensor:
        """pcm [B, S] int16 (device) -> feats [B*T, ldf] bf16 (channels >= 80 are zero); in precise mode fp16 planes
        [B*T, 2 x 96]: hi values in columns [0, 96), lo values (times 2^11) in [96, 192)."""
        _need(pcm, torch.int16, "pcm")
        pcm = pcm.contiguous()
        B, S = pcm.shape
        T = num_frames(S)
        if ldf is None:
            ldf = 2 * N_MELS_PADDED_HP if self.precision == 1 else N_MELS_PADDED
        feats = torch.empty((B * T, ldf), dtype=torch.bfloat16 if self.precision == 0 else torch.float16, device=self.device)
        wsb = self.lib.sdk_fbank_workspace_bytes(B, S)
        ws = self._scratch_bytes("fbank", wsb)
        check(self.lib.sdk_fbank_fmt(self.ctx, pcm.data_ptr(), B, S, self.fbank_tables().data_ptr(), feats.data_ptr(), ldf,
                                     ws.data_ptr(), ws.numel(), self.precision, _stream()), "sdk_fbank")
        return feats

    def fbank_windows(self, samples_ptr: int, n_samples: int, starts_ptr: int, B: int, S: int, ldf: Optional[int] = None) -> torch.Tensor:
        """fbank with the windows cut on the device (sdk_fbank_windows): samples_ptr -> int16 [n_samples] resident recording, starts_ptr -> int32 [B]
        first samples; same features as fbank() on the materialised [B, S] windows, bit for bit."""
        T = num_frames(S)
        if ldf is None:
            ldf = 2 * N_MELS_PADDED_HP if self.precision == 1 else N_MELS_PADDED
        feats = torch.empty((B * T, ldf), dtype=torch.bfloat16 if self.precision == 0 else torch.float16, device=self.device)
        ws = self._scratch_bytes("fbank", self.lib.sdk_fbank_workspace_bytes(B, S))
        check(self.lib.sdk_fbank_windows_fmt(self.ctx, samples_ptr, n_samples, starts_ptr, B, S, self.fbank_tables().data_ptr(), feats.data_ptr(), ldf,
                                             ws.data_ptr(), ws.numel(), self.precision, _stream()), "sdk_fbank_windows")
        return feats

    def ingest(self):
        """The staging runtime (ingest.py / csrc/ingest.hip): pinned, double-buffered, on a copy stream of its own."""
        if self._ingest is None:
            from .ingest import Ingest
            self._ingest = Ingest(self.lib, self.ctx, depth=2)
        return self._ingest

    def embed_from_host(self, samples: np.ndarray, tables: Dict[int, np.ndarray], step: int = 2048, forward=None):
        """One recording in HOST memory (int16 [n]) + window-start tables {window samples S: int32 [B_S]} -> {S: (E, Eb, resid)} device tensors.
        The recording is uploaded once through the pinned staging slots (the next call's upload overlaps this call's forward); the overlapping
        windows are never materialised.  Windows go through in batches of `step`; forward(feats, B, T) defaults to the ECAPA-TDNN forward
        (Backend passes the x-vector's).  Bit-identical to embed_pcm on the host-cut windows."""
        from .ingest import chunk_samples, plan_chunks
        forward = forward or self.ecapa_forward
        self.desc if forward == self.ecapa_forward else None          # lazy weight load (its uploads run on the current stream) before the fan-out
        st = _stream()
        ing = self.ingest()
        samples = np.ascontiguousarray(samples, dtype=np.int16).reshape(-1)
        pieces = plan_chunks(len(samples), tables, chunk_samples())   # one piece unless the recording is longer than a staging slot may be
        out: Dict[int, tuple] = {}
        for lo, hi, sub in pieces:
            order = sorted(sub)
            flat = np.concatenate([sub[S][1] for S in order]) if order else np.zeros((0,), np.int32)
            ticket, ds, dw = ing.submit(samples[lo:hi], flat, max(order) if order else 0, st)      # piece i + 1 uploads under piece i's forward
            off = 0
            try:
                for S in order:
                    rows, local = sub[S]
                    Bs = len(local)
                    if Bs == 0:                                        # an empty table: its (empty) result is made below
                        continue
                    parts = []
                    for a in range(0, Bs, step):
                        b = min(step, Bs - a)
                        feats = self.fbank_windows(ds, hi - lo, dw + 4 * (off + a), b, S)
                        parts.append(self.l2norm(forward(feats, b, num_frames(S))))
                    res = parts[0] if len(parts) == 1 else tuple(torch.cat([p[i] for p in parts], dim=0) for i in range(3))
                    if rows is None:
                        out[S] = res
                    else:                                              # scatter this piece's windows to their rows of the table
                        if S not in out:
                            out[S] = tuple(torch.empty((len(tables[S]),) + tuple(r.shape[1:]), dtype=r.dtype, device=r.device) for r in res)
                        ridx = torch.from_numpy(rows).to(self.device)
                        for dst, r in zip(out[S], res):
                            dst[ridx] = r
                    off += Bs
            finally:
                ing.release(ticket, st)
            self.last_ingest_ticket = ticket
        for S in tables:                                               # an empty table still has an (empty) result
            if S not in out:
                out[S] = (torch.empty((0, self.cfg.embed_dim), dtype=torch.float32, device=self.device),
                          torch.empty((0, self.cfg.embed_dim), dtype=torch.bfloat16, device=self.device), torch.empty((0,), dtype=torch.float32, device=self.device))
        return out

    # ------------------------------------------------------------------ k2
    def ecapa_forward(self, feats: torch.Tensor, B: int, T: int) -> torch.Tensor:
        """feats [B*T, ldf] bf16 (precise mode: fp16 planes) -> raw embeddings [B, 192] fp32."""
        _need(feats, torch.bfloat16 if self.precision == 0 else torch.float16, "feats")
        if feats.shape[0] != B * T or feats.stride(1) != 1:
            raise SdkError(f"feats must be [B*T={B * T}, ldf] row-major, got {tuple(feats.shape)}")
        d = self.desc
        wsb = self.lib.sdk_ecapa_workspace_bytes(C.byref(d), B, T)
        ws = self._scratch_bytes("ecapa", wsb)
        emb = torch.empty((B, self.cfg.embed_dim), dtype=torch.float32, device=self.device)
        check(self.lib.sdk_ecapa_forward(self.ctx, self._wblob.data_ptr(), C.byref(d), feats.data_ptr(), feats.stride(0),
                                         B, T, ws.data_ptr(), ws.numel(), emb.data_ptr(), _stream()), "sdk_ecapa_forward")
        return emb

    # ------------------------------------------------------------------ k3
    def l2norm(self, X: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """X [N, d] fp32 -> (E fp32 unit rows, Eb bf16 copy, resid [N] = ||E - Eb||)."""
        _need(X, torch.float32, "X")
        X = X.contiguous()
        N, d = X.shape
        E = torch.empty_like(X)
        Eb = torch.empty((N, d), dtype=torch.bfloat16, device=self.device)
        r = torch.empty((N,), dtype=torch.float32, device=self.device)
        check(self.lib.sdk_l2norm(self.ctx, X.data_ptr(), N, d, E.data_ptr(), Eb.data_ptr(), r.data_ptr(), _stream()), "sdk_l2norm")
        return E, Eb, r

    # ------------------------------------------------------------------ k4
    def affinity_topk(self, E, Eb, re, P, Pb, rp_max, k: int = 1, want_count: bool = False):
        """Cosine affinity of unit rows E [N,192] vs P [Pn,192] -> (idx [N,k] int32, score [N,k] fp32).
        rp_max: device tensor [1] = max profile residual (from l2norm)."""
        for t, dt, nm in ((E, torch.float32, "E"), (Eb, torch.bfloat16, "Eb"), (re, torch.float32, "resid_e"),
                          (P, torch.float32, "P"), (Pb, torch.bfloat16, "Pb"), (rp_max, torch.float32, "resid_p")):
            _need(t, dt, nm)
        N, d = E.shape
        Pn = P.shape[0]
        idx = torch.empty((N, k), dtype=torch.int32, device=self.device)
        sc = torch.empty((N, k), dtype=torch.float32, device=self.device)
        cnt = torch.zeros((1,), dtype=torch.int32, device=self.device) if want_count else None
        ws = self._scratch_bytes("affinity", self.lib.sdk_affinity_workspace_bytes(N, Pn))
        check(self.lib.sdk_affinity_topk(self.ctx, E.data_ptr(), Eb.data_ptr(), re.data_ptr(), P.data_ptr(), Pb.data_ptr(),
                                         rp_max.data_ptr(), N, Pn, d, k, idx.data_ptr(), sc.data_ptr(), _ptr(cnt),
                                         ws.data_ptr(), ws.numel(), _stream()), "sdk_affinity_topk")
        return (idx, sc, cnt) if want_count else (idx, sc)

    # ------------------------------------------------------------------ whole path
    def embed_pcm(self, pcm: torch.Tensor):
        """pcm [B, S] int16 on device -> (E, Eb, resid) L2-normalised embeddings."""
        B, S = pcm.shape
        feats = self.fbank(pcm)
        emb = self.ecapa_forward(feats, B, num_frames(S))
        return self.l2norm(emb)

    def embed_pcm_overlapped(self, pcm: torch.Tensor, nsplit: int = 2):
        """embed_pcm with the batch cut into `nsplit` sub-batches, each on its own HIP stream.  Segments are
        independent, so the sub-batches need no ordering between them; the dispatcher co-schedules the
        HBM-bound sweeps (SE, pooling, fbank) of one sub-batch into the wave slots, registers and LDS the
        MFMA-bound GEMMs of the other leave free on every CU, and one sub-batch's tail overlaps the other's head."""
        B = pcm.shape[0]
        if nsplit <= 1 or B < 2 * nsplit:
            return self.embed_pcm(pcm)
        # lazy state first, on THIS stream and complete: the weight upload, the calibration pass and the 29 bias patches of a first load must not
        # race with the side streams that read the same blob
        self.desc
        self.fbank_tables()
        torch.cuda.current_stream().synchronize()
        if len(self._streams) < nsplit:
            self._streams += [torch.cuda.Stream(device=self.device) for _ in range(nsplit - len(self._streams))]
        cur = torch.cuda.current_stream()
        parts = []
        for i, chunk in enumerate(pcm.chunk(nsplit)):
            st = self._streams[i]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                parts.append(self.embed_pcm(chunk))
        for i in range(len(parts)):
            cur.wait_stream(self._streams[i])
        E = torch.cat([p[0] for p in parts]); Eb = torch.cat([p[1] for p in parts]); r = torch.cat([p[2] for p in parts])
        for t in (E, Eb, r):
            t.record_stream(cur)
        return E, Eb, r

    def embed_pcm_graph(self, pcm: torch.Tensor):
        """embed_pcm replayed from a captured HIP graph (one graph per input shape): ~45 launches become one
        submission, which is what bounds the latency of a single-recording identify (B <= ~64 windows).
        The returned tensors are the graph's static outputs - consume them before the next call."""
        key = tuple(pcm.shape)
        entry = self._graphs.get(key)
        if entry is None:
            static_in = pcm.clone()
            self.embed_pcm(static_in)                    # warm-up: lazy state (weights, tables, scratch, function attributes)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.embed_pcm(static_in)
            entry = self._graphs[key] = (graph, static_in, out)
        graph, static_in, out = entry
        static_in.copy_(pcm)
        graph.replay()
        return out

    # ------------------------------------------------------------------ building blocks (tests / tuning)
    def conv_gemm(self, A, W, N, Cin, taps=1, dil=1, T=None, bias=None, scale=None, shift=None, ubias=None,
                  relu=False, tanh=False, out_bf16=True, out_f32=False, X2=None, stats_mode=0, A2=None, tap_pack=0,
                  a_kblocked=False, c_kblocked=False):
        """stats_mode 1/2 additionally returns the fused per-segment column statistics as a 4th value
        ([B, N] means, or [B, 2N] mean | std).
        a_kblocked: A is a K-blocked tensor [Cin / 64, M, 64] (to_kblocked); c_kblocked: the bf16 output comes back as [N / 64, M, 64]
        (include/sdk_hip.h SDK_GEMM_A_KBLOCKED / SDK_GEMM_C_KBLOCKED)."""
        _need(A, torch.bfloat16, "A"); _need(W, torch.bfloat16, "W")
        if a_kblocked and (A.dim() != 3 or A.shape[2] != 64 or A.shape[0] * 64 != Cin or not A.is_contiguous()):
            raise ValueError(f"conv_gemm: a K-blocked A must be a contiguous [Cin / 64, M, 64] tensor, got {tuple(A.shape)} for Cin={Cin}")
        M = A.shape[1] if a_kblocked else A.shape[0]
        T = T or M
        g = ConvGemmArgs()
        g.A, g.lda, g.W = A.data_ptr(), (64 if a_kblocked else A.stride(0)), W.data_ptr()
        if A2 is not None:
            g.A2, g.lda2 = A2.data_ptr(), A2.stride(0)
        Cout = torch.empty((N // 64, M, 64) if c_kblocked else (M, N), dtype=torch.bfloat16, device=self.device) if out_bf16 else None
        C32 = torch.empty((M, N), dtype=torch.float32, device=self.device) if out_f32 else None
        S = torch.empty((M, N), dtype=torch.bfloat16, device=self.device) if X2 is not None else None
        g.C, g.ldc, g.C32, g.ldc32 = _ptr(Cout), N, _ptr(C32), N
        g.bias, g.scale, g.shift = _ptr(bias), _ptr(scale), _ptr(shift)
        g.ubias, g.ldub = _ptr(ubias), (ubias.stride(0) if ubias is not None else 0)
        g.X2, g.ldx2 = _ptr(X2), (X2.stride(0) if X2 is not None else 0)
        g.S, g.lds = _ptr(S), N
        g.M, g.N, g.Cin, g.taps, g.dil, g.T = M, N, Cin, taps, dil, T
        g.flags = ((_lib.GEMM_RELU if relu else 0) | (_lib.GEMM_TANH if tanh else 0) | (_lib.GEMM_A_KBLOCKED if a_kblocked else 0)
                   | (_lib.GEMM_C_KBLOCKED if c_kblocked else 0))
        g.tap_pack = tap_pack
        part = None
        if stats_mode:
            part = torch.empty(self.lib.sdk_conv_gemm_stats_bytes(M, N, stats_mode), dtype=torch.uint8, device=self.device)
            g.stats_mode, g.stats_part = stats_mode, part.data_ptr()
        check(self.lib.sdk_conv_gemm(self.ctx, C.byref(g), _stream()), "sdk_conv_gemm")
        if stats_mode:
            st = torch.empty((M // T, N * stats_mode), dtype=torch.float32, device=self.device)
            check(self.lib.sdk_colstats_finish(self.ctx, part.data_ptr(), M, N, T, stats_mode, st.data_ptr(), _stream()), "sdk_colstats_finish")
            return Cout, C32, S, st
        return Cout, C32, S

    @staticmethod
    def to_kblocked(x: torch.Tensor) -> torch.Tensor:
        """[M, C] -> the K-blocked layout [C / 64, M, 64] (element (m, c) at (c // 64, m, c % 64)); for tests and tools."""
        M, Cc = x.shape
        return x.reshape(M, Cc // 64, 64).permute(1, 0, 2).contiguous()

    @staticmethod
    def from_kblocked(x: torch.Tensor) -> torch.Tensor:
        return x.permute(1, 0, 2).reshape(x.shape[1], x.shape[0] * 64).contiguous()

    # ---- precise mode building blocks (csrc/hp.hip) ------------------------------------------------
    @staticmethod
    def to_planes(x: torch.Tensor) -> torch.Tensor:
        """fp32 [M, C] -> fp16 planes [M, 2C]: hi | lo * 2^11 (the format csrc/hp.hpp stores; same arithmetic, for tests and tools)."""
        x = x.float().clamp(-65504.0, 65504.0)
        hi = x.to(torch.float16)
        lo = ((x - hi.float()) * 2048.0).to(torch.float16)
        return torch.cat([hi, lo], dim=1).contiguous()

    @staticmethod
    def from_planes(p: torch.Tensor) -> torch.Tensor:
        c = p.shape[1] // 2
        return p[:, :c].float() + p[:, c:].float() * (1.0 / 2048.0)

    def conv_gemm_hp(self, A, Wslot, N, Cin, taps=1, dil=1, T=None, bias=None, scale=None, shift=None, ubias=None,
                     relu=False, tanh=False, out_planes=True, out_f32=False, X2=None):
        """A: fp16 planes [M, 2*Cin']; Wslot: uint16/fp16 device tensor holding weights_pack.hp_weight_planes(W [N, taps*Cin]).
        Returns (C planes [M, 2N] or None, C32 fp32 or None, S planes or None)."""
        from ._lib import ConvGemmHpArgs
        _need(A, torch.float16, "A")
        M = A.shape[0]
        T = T or M
        g = ConvGemmHpArgs()
        g.A, g.lda, g.a_lo, g.W = A.data_ptr(), A.stride(0), A.shape[1] // 2, Wslot.data_ptr()
        Cout = torch.empty((M, 2 * N), dtype=torch.float16, device=self.device) if out_planes else None
        C32 = torch.empty((M, N), dtype=torch.float32, device=self.device) if out_f32 else None
        S = torch.empty((M, 2 * N), dtype=torch.float16, device=self.device) if X2 is not None else None
        g.C, g.ldc, g.c_lo, g.C32, g.ldc32 = _ptr(Cout), 2 * N, N, _ptr(C32), N
        g.bias, g.scale, g.shift = _ptr(bias), _ptr(scale), _ptr(shift)
        g.ubias, g.ldub = _ptr(ubias), (ubias.stride(0) if ubias is not None else 0)
        if X2 is not None:
            _need(X2, torch.float16, "X2")
            g.X2, g.ldx2, g.x2_lo = X2.data_ptr(), X2.stride(0), X2.shape[1] // 2
            g.S, g.lds, g.s_lo = S.data_ptr(), 2 * N, N
        g.M, g.N, g.Cin, g.taps, g.dil, g.T = M, N, Cin, taps, dil, T
        g.flags = (_lib.GEMM_RELU if relu else 0) | (_lib.GEMM_TANH if tanh else 0)
        check(self.lib.sdk_conv_gemm_hp(self.ctx, C.byref(g), _stream()), "sdk_conv_gemm_hp")
        return Cout, C32, S

    def se_gate_residual(self, z, x, w1t, b1, w2t, b2, B, T, split: bool = True):
        C_ = z.shape[1]
        out = torch.empty_like(z)
        ws = self._scratch_bytes("se", self.lib.sdk_se_workspace_bytes(B, C_, w1t.shape[1])) if split else None
        check(self.lib.sdk_se_gate_residual(self.ctx, z.data_ptr(), z.stride(0), x.data_ptr(), x.stride(0), w1t.data_ptr(),
                                            b1.data_ptr(), w2t.data_ptr(), b2.data_ptr(), out.data_ptr(), out.stride(0),
                                            B, T, C_, w1t.shape[1], None, _ptr(ws), ws.numel() if split else 0, _stream()), "sdk_se_gate_residual")
        return out

    def asp_stats(self, h, B, T):
        Cm = h.shape[1]
        out = torch.empty((B, 2 * Cm), dtype=torch.float32, device=self.device)
        check(self.lib.sdk_asp_stats(self.ctx, h.data_ptr(), h.stride(0), B, T, Cm, out.data_ptr(), _stream()), "sdk_asp_stats")
        return out

    def rows_fc(self, x, wt, bias=None, in_scale=None, in_shift=None, act=0):
        B, Cin = x.shape
        Nout = wt.shape[1]
        out = torch.empty((B, Nout), dtype=torch.float32, device=self.device)
        check(self.lib.sdk_rows_fc(self.ctx, x.data_ptr(), x.stride(0), _ptr(in_scale), _ptr(in_shift), wt.data_ptr(), _ptr(bias),
                                   out.data_ptr(), Nout, B, Cin, Nout, act, _stream()), "sdk_rows_fc")
        return out

    def asp_pool(self, logits, h, B, T):
        Cm = h.shape[1]
        out = torch.empty((B, 2 * Cm), dtype=torch.float32, device=self.device)
        check(self.lib.sdk_asp_pool(self.ctx, logits.data_ptr(), logits.stride(0), h.data_ptr(), h.stride(0), B, T, Cm,
                                    out.data_ptr(), _stream()), "sdk_asp_pool")
        return out


    # ------------------------------------------------------------------ k6 (spectral clustering pieces)
    def affinity_matvec(self, Eb, X, row0: int = 0, rows: Optional[int] = None, xscale=None, out=None):
        """Y[row0:row0+rows] = max(E E^T, 0)[row0:row0+rows, :] @ (xscale[:, None] * X);  X [N, kv] fp32."""
        _need(Eb, torch.bfloat16, "Eb"); _need(X, torch.float32, "X")
        N, d = Eb.shape
        kv = X.shape[1]
        rows = N - row0 if rows is None else rows
        Y = out if out is not None else torch.zeros((N, kv), dtype=torch.float32, device=self.device)
        ws = self._scratch_bytes("matvec", self.lib.sdk_affinity_matvec_workspace_bytes(N))
        check(self.lib.sdk_affinity_matvec(self.ctx, Eb.data_ptr(), N, d, row0, rows, X.contiguous().data_ptr(), _ptr(xscale), kv,
                                           Y.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "sdk_affinity_matvec")
        return Y

    def laplacian_topk(self, Eb_all, V0, n_iter: int, row0: int = 0, rows: Optional[int] = None, comm: Optional[int] = None, world: int = 1,
                       flag: Optional[torch.Tensor] = None):
        """The C-ABI spectral driver for non-Python hosts (sdk_laplacian_topk): top-k Ritz pairs of D^-1/2 A D^-1/2 by subspace iteration,
        never leaving the stream.  V0 [rows, k] fp32 start block (not modified) -> (U [rows, k], eigenvalues [k] device, descending).
        comm: an ncclComm_t as an integer (or None on one GPU)."""
        _need(Eb_all, torch.bfloat16, "Eb_all"); _need(V0, torch.float32, "V0")
        N = Eb_all.shape[0]
        rows = N - row0 if rows is None else rows
        k = V0.shape[1]
        V = V0.contiguous().clone()
        lam = torch.empty((k,), dtype=torch.float32, device=self.device)
        ws = self._scratch_bytes("laplacian", self.lib.sdk_laplacian_topk_workspace_bytes(N, k))
        check(self.lib.sdk_laplacian_topk(self.ctx, Eb_all.data_ptr(), N, row0, rows, k, n_iter, V.data_ptr(), lam.data_ptr(), _ptr(flag), ws.data_ptr(),
                                          ws.numel(), comm, world, _stream()), "sdk_laplacian_topk")
        return V, lam

    def rows_gram(self, X, Y):
        n, k = X.shape
        G = torch.empty((k, k), dtype=torch.float32, device=self.device)
        ws = self._scratch_bytes("gram", self.lib.sdk_rows_gram_workspace_bytes(n, k))
        check(self.lib.sdk_rows_gram(self.ctx, X.data_ptr(), Y.data_ptr(), n, k, G.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "sdk_rows_gram")
        return G

    def rows_apply(self, X, R, scale=None):
        n, k = X.shape
        Y = torch.empty_like(X)
        check(self.lib.sdk_rows_apply(self.ctx, X.data_ptr(), R.contiguous().data_ptr(), _ptr(scale), n, k, Y.data_ptr(), _stream()), "sdk_rows_apply")
        return Y

    def chol_inverse(self, G, flag: Optional[torch.Tensor] = None):
        """Rinv [k,k] with (G+G^T)/2 = L L^T, Rinv = (L^T)^-1 (float64 inside, stays on the stream).  `flag` (device int32 [1], zeroed
        by the caller) is set to 1 when G is not positive definite (rank-deficient or NaN input); it is sticky, so one flag can watch a
        whole iteration and be read at the caller's next host synchronisation."""
        k = G.shape[0]
        if flag is not None:
            _need(flag, torch.int32, "flag")
        Rinv = torch.empty((k, k), dtype=torch.float32, device=self.device)
        check(self.lib.sdk_chol_inverse(self.ctx, G.contiguous().data_ptr(), k, Rinv.data_ptr(), _ptr(flag), _stream()), "sdk_chol_inverse")
        return Rinv

    def rows_unit(self, X):
        n, k = X.shape
        Y = torch.empty_like(X)
        check(self.lib.sdk_rows_unit(self.ctx, X.data_ptr(), n, k, Y.data_ptr(), _stream()), "sdk_rows_unit")
        return Y

    def kmeans_mindist(self, R, centre, d2, first: bool):
        n, k = R.shape
        check(self.lib.sdk_kmeans_mindist(self.ctx, R.data_ptr(), n, k, centre.contiguous().data_ptr(), d2.data_ptr(), int(first), _stream()), "sdk_kmeans_mindist")
        return d2

    def kmeans_assign(self, R, centres, want_sums: bool = True):
        n, k = R.shape
        kc = centres.shape[0]
        lab = torch.empty((n,), dtype=torch.int32, device=self.device)
        d2 = torch.empty((n,), dtype=torch.float32, device=self.device)
        nb = (n + 255) // 256
        ps = torch.empty((nb, kc, k), dtype=torch.float32, device=self.device) if want_sums else None
        pc = torch.empty((nb, kc), dtype=torch.int32, device=self.device) if want_sums else None
        check(self.lib.sdk_kmeans_assign(self.ctx, R.data_ptr(), n, k, centres.contiguous().data_ptr(), kc, lab.data_ptr(), d2.data_ptr(),
                                         _ptr(ps), _ptr(pc), _stream()), "sdk_kmeans_assign")
        return lab, d2, ps, pc

    def asp_fused(self, ah, w2, b2, h, B, T, kblocked=False):
        """kblocked: h is [Cm / 64, B*T, 64] (to_kblocked) - the per-segment form only (sdk_asp_kblocked_ok)."""
        if kblocked:
            Cm = h.shape[0] * 64
            out = torch.empty((B, 2 * Cm), dtype=torch.float32, device=self.device)
            check(self.lib.sdk_asp_fused_kblocked(self.ctx, ah.data_ptr(), ah.stride(0), w2.data_ptr(), b2.data_ptr(), h.data_ptr(),
                                                  B, T, Cm, ah.shape[1], out.data_ptr(), _stream()), "sdk_asp_fused_kblocked")
            return out
        Cm = h.shape[1]
        out = torch.empty((B, 2 * Cm), dtype=torch.float32, device=self.device)
        check(self.lib.sdk_asp_fused(self.ctx, ah.data_ptr(), ah.stride(0), w2.data_ptr(), b2.data_ptr(), h.data_ptr(), h.stride(0),
                                     B, T, Cm, ah.shape[1], out.data_ptr(), _stream()), "sdk_asp_fused")
        return out


_engines: Dict[int, Engine] = {}


def get_engine(device: int = 0, **kw) -> Engine:
    if device not in _engines:
        _engines[device] = Engine(device, **kw)
    return _engines[device]
