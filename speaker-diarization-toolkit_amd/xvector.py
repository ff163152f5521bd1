"""x-vector (plain TDNN) embedding extractor - the second model family north_star names ("ECAPA-TDNN/x-vector forward pass").

The reference has no local embedding model at all (speaker_detection_backends/backends.yaml:22-31 lists them as "future"); like the
ECAPA-TDNN it is the PUBLIC architecture (Snyder et al. 2018, "X-vectors: robust DNN embeddings for speaker recognition"): five frame
layers (contexts k5 / k3 dil 2 / k3 dil 3 / k1 / k1, 512-512-512-512-1500 channels, each affine -> ReLU -> BatchNorm), statistics
pooling (mean | std over frames), and the first segment layer whose pre-activation is the embedding.  Differences stated: "same"-length
frame layers by segment-local reflection (the toolkit cuts fixed windows; Kaldi shrinks the context instead), the front-end is this
build's 80-bin log-mel fbank, and the default embedding width is 192 (not 512) so that k3 / k4 / the .npy store - specialised for 192-d -
serve both model families.  PARITY UNPINNED (no reference implementation, no checkpoint).

Everything runs in libsdk_hip.so through ONE C call per batch (sdk_xvector_forward: sdk_conv_gemm per frame layer, the first with its
taps packed along K; sdk_asp_stats; sdk_rows_fc).  Weights: fp32 host dict, keys  frame{l}.conv.{w,b}  frame{l}.bn.{gamma,beta,mean,var}
embed.{w,b}.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

from .weights import bn_affine
from .weights_pack import ALIGN, N_MELS_PADDED_HP, bf16_bits_to_f32, conv_weight_kmajor, conv_weight_kmajor_f32, f32_to_bf16_bits, hp_weight_planes, round16


@dataclass(frozen=True)
class XVectorConfig:
    n_feats: int = 80
    kernels: Tuple[int, ...] = (5, 3, 3, 1, 1)
    dilations: Tuple[int, ...] = (1, 2, 3, 1, 1)
    channels: Tuple[int, ...] = (512, 512, 512, 512, 1500)
    embed_dim: int = 192

    def padded_channels(self) -> Tuple[int, ...]:
        return tuple((c + 127) // 128 * 128 for c in self.channels)

    def macs_per_frame(self) -> int:
        cin = (self.n_feats,) + self.channels[:-1]
        return sum(k * ci * co for k, ci, co in zip(self.kernels, cin, self.channels))


DEFAULT_XVECTOR = XVectorConfig()


class XVectorDesc(C.Structure):
    _fields_ = [("n_frame_layers", C.c_int32), ("n_feats", C.c_int32), ("embed_dim", C.c_int32), ("first_tap_pack", C.c_int32),
                ("kernel", C.c_int32 * 8), ("dilation", C.c_int32 * 8), ("cin", C.c_int32 * 8), ("cout", C.c_int32 * 8), ("off", C.c_int64 * 64)]


def param_shapes(cfg: XVectorConfig = DEFAULT_XVECTOR) -> Dict[str, Tuple[int, ...]]:
    sh: Dict[str, Tuple[int, ...]] = {}
    cin = (cfg.n_feats,) + cfg.channels[:-1]
    for l, (k, ci, co) in enumerate(zip(cfg.kernels, cin, cfg.channels)):
        sh[f"frame{l}.conv.w"] = (co, ci, k)
        sh[f"frame{l}.conv.b"] = (co,)
        for f in ("gamma", "beta", "mean", "var"):
            sh[f"frame{l}.bn.{f}"] = (co,)
    sh["embed.w"] = (cfg.embed_dim, 2 * cfg.channels[-1])
    sh["embed.b"] = (cfg.embed_dim,)
    return sh


def synthetic_weights(seed: int = 0, cfg: XVectorConfig = DEFAULT_XVECTOR) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith(".w"):
            fan_in = int(np.prod(shape[1:]))
            gain = 1.0 if name.startswith("embed") else 2.0
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(gain / fan_in))
        elif name.endswith((".b", ".beta", ".mean")):
            a = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
        elif name.endswith(".gamma"):
            a = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        else:
            a = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        out[name] = np.ascontiguousarray(a, dtype=np.float32)
    return out


def pack_weights(weights: Dict[str, np.ndarray], cfg: XVectorConfig = DEFAULT_XVECTOR, precision: int = 0):
    """-> (blob uint8, XVectorDesc).  Channel counts are padded to multiples of 128 with zero weights / bias, BN scale 1, shift 0 (a padded
    channel is exactly 0 after ReLU; its pooled mean is 0, its std the 1e-6 floor, and the embedding layer's weights for it are zero).
    precision 1: the precise mode's blob (off[62] = 1): every frame layer's weights as a power-of-two-scaled fp16 hi + lo plane slot
    (weights_pack.hp_weight_planes), the first layer over the 96 padded mel channels of the plane-format features, no tap packing.
    precision 2: the default layout with fp16 bits in the weight slots (off[62] = 2; features from sdk_fbank_fmt(..., 2, ...))."""
    for k, s in param_shapes(cfg).items():
        if k not in weights or tuple(weights[k].shape) != s:
            raise ValueError(f"x-vector weight {k}: expected shape {s}, got {None if k not in weights else tuple(weights[k].shape)}")
    if precision not in (0, 1, 2):
        raise ValueError(f"precision must be 0, 1 or 2, got {precision}")
    pc = cfg.padded_channels()
    cin_real = (cfg.n_feats,) + cfg.channels[:-1]
    n_in = N_MELS_PADDED_HP if precision == 1 else cfg.n_feats
    if precision == 1 and cfg.n_feats > N_MELS_PADDED_HP:
        raise ValueError(f"precise mode reads {N_MELS_PADDED_HP} padded feature channels, the configuration has {cfg.n_feats}")
    cin_pad = (n_in,) + pc[:-1]
    d = XVectorDesc()
    d.n_frame_layers, d.n_feats, d.embed_dim = len(cfg.kernels), n_in, cfg.embed_dim
    d.first_tap_pack = 0 if precision == 1 else (cfg.n_feats if cfg.n_feats % 64 else 0)
    off = [-1] * 64
    off[62] = precision
    chunks, cur = [], 0

    def put(slot, arr):
        nonlocal cur
        a = np.ascontiguousarray(arr)
        assert off[slot] == -1 and a.dtype in (np.uint16, np.float32)
        off[slot] = cur
        chunks.append((cur, a.view(np.uint8).reshape(-1)))
        cur += (a.nbytes + ALIGN - 1) // ALIGN * ALIGN

    for l in range(len(cfg.kernels)):
        w = np.zeros((pc[l], cin_pad[l], cfg.kernels[l]), np.float32)
        w[:cfg.channels[l], :cin_real[l], :] = weights[f"frame{l}.conv.w"]
        if precision == 1:
            wk = hp_weight_planes(conv_weight_kmajor_f32(w))                    # 256-byte header + fp16 hi / lo planes of 2^s W, [cout, k * cin] each
        else:
            wk = conv_weight_kmajor(w, precision=precision)                     # bf16 (precision 2: fp16) bits [cout, k * cin]
            if l == 0 and d.first_tap_pack:
                kp = (wk.shape[1] + 63) // 64 * 64
                wp = np.zeros((wk.shape[0], kp), np.uint16)
                wp[:, :wk.shape[1]] = wk
                wk = wp
        put(4 * l, wk)
        b = np.zeros(pc[l], np.float32); b[:cfg.channels[l]] = weights[f"frame{l}.conv.b"]
        s, sh = bn_affine(weights, f"frame{l}.bn")
        sp = np.ones(pc[l], np.float32); sp[:cfg.channels[l]] = s
        shp = np.zeros(pc[l], np.float32); shp[:cfg.channels[l]] = sh
        put(4 * l + 1, b); put(4 * l + 2, sp); put(4 * l + 3, shp)
        d.kernel[l], d.dilation[l], d.cin[l], d.cout[l] = cfg.kernels[l], cfg.dilations[l], cin_pad[l], pc[l]
    cl, cr = pc[-1], cfg.channels[-1]
    wt = np.zeros((2 * cl, cfg.embed_dim), np.float32)                        # [mean | std] inputs, transposed for rows_fc
    wt[:cr] = weights["embed.w"][:, :cr].T
    wt[cl:cl + cr] = weights["embed.w"][:, cr:].T
    put(60, wt)
    put(61, weights["embed.b"].astype(np.float32))
    d.off = (C.c_int64 * 64)(*off)
    blob = np.zeros(cur, np.uint8)
    for o, a in chunks:
        blob[o:o + a.size] = a
    return blob, d


def bias_corrections(weights: Dict[str, np.ndarray], means: Dict[int, np.ndarray], cfg: XVectorConfig = DEFAULT_XVECTOR,
                     precision: int = 0) -> Dict[str, np.ndarray]:
    """frame layer l -> corrected fp32 bias  b + (W - round16(W)) . mu_l  (float64 inside; every tap sees the same channel means; round16 =
    bf16 for precision 0, fp16 for precision 2) - the same post-training bias correction of the weight rounding the ECAPA-TDNN family gets
    (weights_pack.bias_corrections)."""
    out = {}
    for l, mu in means.items():
        w = weights[f"frame{l}.conv.w"].astype(np.float64)
        dw = w - round16(w.astype(np.float32), precision).astype(np.float64)
        corr = np.tensordot(dw.sum(axis=2), np.asarray(mu, np.float64)[:dw.shape[1]], axes=([1], [0]))
        out[f"frame{l}.conv.b"] = (weights[f"frame{l}.conv.b"].astype(np.float64) + corr).astype(np.float32)
    return out


def calibration_means(lib, ctx, d: "XVectorDesc", cfg: XVectorConfig, blob_ptr: int, feats_ptr: int, ldf: int, B: int, T: int, alloc, download,
                      stream) -> Dict[int, np.ndarray]:
    """Per-channel means of the INPUT of frame layers 1..L-1 of a single-plane blob (off[62] = 0 or 2): the layers run one at a time through
    sdk_conv_gemm, the per-segment channel means come from sdk_asp_stats_fmt in the blob's element format - C-ABI calls only, so ops.Engine (torch tensors) and lite.LiteEngine (sdk_device_malloc)
    run the identical sequence and get identical means.  alloc(nbytes) -> device pointer (kept alive by the caller); download(ptr, n_floats)
    -> float32 host array (synchronising)."""
    from ._lib import ConvGemmArgs, GEMM_F16, GEMM_RELU, check
    fmt = 2 if int(d.off[62]) == 2 else 0
    means: Dict[int, np.ndarray] = {}
    x, ldx = feats_ptr, ldf
    M = B * T
    for l in range(d.n_frame_layers - 1):
        cout = d.cout[l]
        out = alloc(M * cout * 2)
        g = ConvGemmArgs()
        g.A, g.lda, g.W = x, ldx, blob_ptr + int(d.off[4 * l])
        g.C, g.ldc = out, cout
        g.bias, g.scale, g.shift = blob_ptr + int(d.off[4 * l + 1]), blob_ptr + int(d.off[4 * l + 2]), blob_ptr + int(d.off[4 * l + 3])
        g.M, g.N, g.Cin, g.taps, g.dil, g.T = M, cout, d.cin[l], d.kernel[l], d.dilation[l], T
        g.flags = GEMM_RELU | (GEMM_F16 if fmt == 2 else 0)
        g.tap_pack = d.first_tap_pack if l == 0 else 0
        check(lib.sdk_conv_gemm(ctx, C.byref(g), stream), "sdk_conv_gemm")
        st = alloc(B * 2 * cout * 4)
        check(lib.sdk_asp_stats_fmt(ctx, out, cout, B, T, cout, st, fmt, stream), "sdk_asp_stats_fmt")
        stats = download(st, B * 2 * cout).reshape(B, 2 * cout)                  # mean | std per segment
        means[l + 1] = stats[:, :cout].astype(np.float64).mean(axis=0)[:cfg.channels[l]]
        x, ldx = out, cout
    return means


class XVector:
    """Resident x-vector extractor on an ops.Engine (device blob + descriptor); embed_pcm mirrors Engine.embed_pcm.
    bias_correction (default: $SDK_BIAS_CORRECTION, on): single-plane modes (0: bf16, 2: fp16) - fold the constant part of the weight-rounding error of frame
    layers 1.. into their biases, from ONE calibration pass on the engine's built-in synthetic audio (the layer inputs' channel means are
    measured on the GPU with the library's own kernels; layer 0 needs none: its input is mean-normalised).  precision 1: the precise mode
    (fp16 hi+lo planes, three MFMAs per product); precision 2: one fp16 plane.  The engine's front end must write that format
    (Engine.set_precision), which XVector does itself."""

    def __init__(self, engine, weights: Dict[str, np.ndarray] = None, cfg: XVectorConfig = DEFAULT_XVECTOR, seed: int = 0,
                 bias_correction=None, precision: int = 0):
        import os
        import torch
        self.eng, self.cfg, self.precision = engine, cfg, int(precision)
        self.weights = dict(weights if weights is not None else synthetic_weights(seed, cfg))
        self.bias_correction = ((os.environ.get("SDK_BIAS_CORRECTION", "1") != "0") if bias_correction is None else bool(bias_correction)) and self.precision in (0, 2)
        blob, self.desc = pack_weights(self.weights, cfg, precision=self.precision)
        self.blob = torch.from_numpy(blob).to(engine.device)
        self._effective = self.weights
        if self.bias_correction:
            self._effective = dict(self.weights, **bias_corrections(self.weights, self._calibrate(), cfg, precision=self.precision))
            blob, self.desc = pack_weights(self._effective, cfg, precision=self.precision)
            self.blob = torch.from_numpy(blob).to(engine.device)

    def effective_weights(self) -> Dict[str, np.ndarray]:
        """The weights whose bf16 layer-boundary model the default mode computes (= the loaded ones with the corrected biases)."""
        return self._effective

    def _calibrate(self) -> Dict[int, np.ndarray]:
        """Channel means of the inputs of frame layers 1..L-1 on the built-in calibration audio (calibration_means: library calls only)."""
        import torch
        from .ops import _stream, num_frames
        from .weights_pack import calibration_pcm
        eng = self.eng
        if eng.precision != self.precision:
            eng.set_precision(self.precision)                   # the calibration features in the blob's element format
        pcm = torch.from_numpy(calibration_pcm()).to(eng.device)
        B, S = pcm.shape
        feats = eng.fbank(pcm)
        keep = []

        def alloc(nbytes):
            keep.append(torch.empty(nbytes, dtype=torch.uint8, device=eng.device))
            return keep[-1].data_ptr()

        def download(ptr, n_floats):
            t = next(k for k in keep if k.data_ptr() == ptr)
            return t[:4 * n_floats].view(torch.float32).cpu().numpy()
        return calibration_means(eng.lib, eng.ctx, self.desc, self.cfg, self.blob.data_ptr(), feats.data_ptr(), feats.stride(0), B, num_frames(S),
                                 alloc, download, _stream())

    def forward(self, feats, B: int, T: int):
        """feats [B*T, ldf] bf16 as Engine.fbank writes them (precise mode: fp16 planes) -> raw embeddings [B, embed_dim] fp32."""
        import torch
        from ._lib import check
        from .ops import _stream
        lib = self.eng.lib
        ws = self.eng._scratch_bytes("xvector", lib.sdk_xvector_workspace_bytes(C.byref(self.desc), B, T))
        emb = torch.empty((B, self.cfg.embed_dim), dtype=torch.float32, device=self.eng.device)
        check(lib.sdk_xvector_forward(self.eng.ctx, self.blob.data_ptr(), C.byref(self.desc), feats.data_ptr(), feats.stride(0), B, T,
                                      ws.data_ptr(), ws.numel(), emb.data_ptr(), _stream()), "sdk_xvector_forward")
        return emb

    def embed_pcm(self, pcm):
        from .ops import num_frames
        B, S = pcm.shape
        if self.eng.precision != self.precision:
            self.eng.set_precision(self.precision)              # the front end's output format follows the numerical contract
        return self.eng.l2norm(self.forward(self.eng.fbank(pcm), B, num_frames(S)))
