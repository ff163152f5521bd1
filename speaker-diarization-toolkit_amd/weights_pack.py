"""Pack an ECAPA-TDNN weight dict (weights.py naming) into the single device blob + offset table
that sdk_ecapa_forward consumes.  Slot numbering mirrors csrc/ecapa_layout.h.

Layout rules:
  * operands of MFMA GEMMs are bf16, stored [C_out][taps * C_in_padded] (tap-major, K contiguous);
    the mel input is zero-padded 80 -> 128 channels so every K-step is a full 64-channel tile;
  * eval-mode BatchNorm is folded to a per-channel fp32 (scale, shift) applied in the epilogue;
  * everything evaluated on the VALU (SE gates, global-context bias, final FC) stays fp32 and is
    stored transposed ([C_in][C_out]) so consecutive lanes read consecutive addresses.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .weights import DEFAULT_CONFIG, EcapaConfig, bn_affine, check_weights

N_MELS_PADDED = 128
ALIGN = 256

EL_W, EL_B, EL_SCALE, EL_SHIFT = 0, 1, 2, 3
EL_BLK0 = 0
EL_TDNN1, EL_TDNN2 = 0, 32
EL_SE_W1T, EL_SE_B1, EL_SE_W2T, EL_SE_B2 = 36, 37, 38, 39
EL_MFA, EL_ASP_WH, EL_ASP_WMS_T, EL_ASP_B, EL_ASP_SCALE, EL_ASP_SHIFT = 0, 4, 5, 6, 7, 8
EL_ASP_W2, EL_ASP_B2, EL_ASPBN_SCALE, EL_ASPBN_SHIFT, EL_FC_WT, EL_FC_B, EL_ASP_W2PACK = 9, 10, 11, 12, 13, 14, 15


def block_base(i: int) -> int:
    return 4 + (i - 1) * 40


def res2net_slot(j: int) -> int:
    return 4 + 4 * j


def tail_base(n_blocks: int) -> int:
    return 4 + n_blocks * 40


def chainpack_slot(i: int, j: int) -> int:
    """EL_CHAINPACK(i, j): optional fragment-ordered copy of block i's Res2Net conv j."""
    return 200 + (i - 1) * 8 + j


def chain_fragment_order(wk: np.ndarray) -> np.ndarray:
    """[128, 3*128] bf16 bits (tap-major K) -> [tap 3][wq 4][h 2][ks 4][lane 64][8] (csrc/ecapa_layout.h): the 16 bytes lane l of
    wave-column-block wq needs for column tile h, tap, k-step ks sit at a lane-contiguous address."""
    assert wk.shape == (128, 384) and wk.dtype == np.uint16
    lane = np.arange(64)
    fr, fq = lane & 15, lane >> 4
    out = np.empty((3, 4, 2, 4, 64, 8), dtype=np.uint16)
    for tap in range(3):
        for wq in range(4):
            for h in range(2):
                for ks in range(4):
                    rows = 32 * wq + 16 * h + fr                                   # [64]
                    cols = 128 * tap + 32 * ks + 8 * fq                             # [64]
                    out[tap, wq, h, ks] = wk[rows[:, None], cols[:, None] + np.arange(8)[None, :]]
    return out


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bit pattern, round-to-nearest-even (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_f16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> IEEE binary16 bit pattern, round-to-nearest-even, saturated at the format's finite range (precision 2's storage)."""
    return np.clip(np.ascontiguousarray(x, dtype=np.float32), -65504.0, 65504.0).astype(np.float16).view(np.uint16)


def to_bits16(x: np.ndarray, precision: int) -> np.ndarray:
    """the 2-byte GEMM operand format of a single-plane contract: bf16 (precision 0) or fp16 (precision 2)"""
    return f32_to_f16_bits(x) if precision == 2 else f32_to_bf16_bits(x)


def round16(x: np.ndarray, precision: int) -> np.ndarray:
    """x rounded to that format, as fp32 values"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    return f32_to_f16_bits(x).view(np.float16).astype(np.float32) if precision == 2 else bf16_bits_to_f32(f32_to_bf16_bits(x))


def asp_w2_fragment_order(w2: np.ndarray) -> np.ndarray:
    """[Cm, 128] bf16 bits -> [Cm/32][ks 8][lane 64][8] (csrc/ecapa_layout.h EL_ASP_W2PACK): lane l of a 32x32x16 fragment holds
    W2[32 blk + (l & 31)][16 ks + 8 (l >> 5) + 0..7]."""
    cm, a = w2.shape
    assert a == 128 and cm % 32 == 0 and w2.dtype == np.uint16
    lane = np.arange(64)
    col, hh = lane & 31, lane >> 5
    out = np.empty((cm // 32, 8, 64, 8), dtype=np.uint16)
    for ks in range(8):
        k0 = 16 * ks + 8 * hh                                                      # [64]
        for blk in range(cm // 32):
            out[blk, ks] = w2[(32 * blk + col)[:, None], k0[:, None] + np.arange(8)[None, :]]
    return out


def conv_weight_kmajor(w: np.ndarray, cin_pad: int | None = None, precision: int = 0) -> np.ndarray:
    """[C_out, C_in, k] -> bf16 (precision 2: fp16) bits [C_out, k * C_in_pad] (tap-major, zero-padded channels)."""
    co, ci, k = w.shape
    cp = cin_pad or ci
    out = np.zeros((co, k, cp), dtype=np.float32)
    out[:, :, :ci] = np.transpose(w, (0, 2, 1))
    return to_bits16(out.reshape(co, k * cp), precision)


N_MELS_PADDED_HP = 96          # precise mode: the mel channels padded to the next multiple of its 32-wide K-step
HP_WHDR = 128                  # fp16 elements (256 bytes) in front of a precise weight slot's planes (csrc/hp.hpp)


def hp_weight_planes(wk: np.ndarray) -> np.ndarray:
    """fp32 [N, K] (K contiguous) -> the precise mode's weight slot (csrc/hp.hip): a 256-byte header whose first float is 2^-s, then
    the fp16 planes hi = fp16(2^s W), lo = fp16(2^s W - hi), both [N, K].  s is the largest power of two that keeps max |2^s W| <= 2^13:
    the lo plane of every weight above 2^-13 of the largest stays in fp16's normal range (22 significand bits for the pair)."""
    wk = np.ascontiguousarray(wk, dtype=np.float32)
    amax = float(np.abs(wk).max())
    s = 0 if amax == 0.0 else int(np.clip(np.floor(np.log2(8192.0 / amax)), -14, 60))
    scaled = wk.astype(np.float64) * (2.0 ** s)
    hi = scaled.astype(np.float16)
    lo = (scaled - hi.astype(np.float64)).astype(np.float16)
    assert np.isfinite(hi.astype(np.float32)).all()
    out = np.zeros(HP_WHDR + 2 * wk.size, dtype=np.uint16)
    out[:2] = np.array([2.0 ** -s], dtype=np.float32).view(np.uint16)
    out[HP_WHDR:HP_WHDR + wk.size] = hi.view(np.uint16).reshape(-1)
    out[HP_WHDR + wk.size:] = lo.view(np.uint16).reshape(-1)
    return out


def hp_planes_to_f64(slot: np.ndarray, n: int, k: int) -> np.ndarray:
    """Inverse of hp_weight_planes (tests): the value the precise GEMM multiplies by."""
    inv = float(slot[:2].view(np.float32)[0])
    hi = slot[HP_WHDR:HP_WHDR + n * k].view(np.float16).astype(np.float64)
    lo = slot[HP_WHDR + n * k:HP_WHDR + 2 * n * k].view(np.float16).astype(np.float64)
    return ((hi + lo) * inv).reshape(n, k)


def conv_weight_kmajor_f32(w: np.ndarray, cin_pad: int | None = None) -> np.ndarray:
    """[C_out, C_in, k] -> fp32 [C_out, k * C_in_pad] (tap-major, zero-padded channels)."""
    co, ci, k = w.shape
    cp = cin_pad or ci
    out = np.zeros((co, k, cp), dtype=np.float32)
    out[:, :, :ci] = np.transpose(w, (0, 2, 1))
    return out.reshape(co, k * cp)


def pack_weights(weights: Dict[str, np.ndarray], cfg: EcapaConfig = DEFAULT_CONFIG, precision: int = 0):
    """Return (blob: np.uint8 [bytes], desc_fields: dict) - host side only, no device access.
    precision 0: bf16 GEMM operands (default mode).  precision 1 (precise mode, csrc/hp.hip): every GEMM weight slot holds fp16
    hi+lo planes behind a scale header (hp_weight_planes), the mel channels are padded to 96, no fragment-ordered copies;
    everything fp32 is identical in both blobs.  precision 2 (one fp16 plane, round 5): the default mode's blob layout - packed blk0 taps,
    fragment-ordered copies - with every 2-byte weight in fp16 instead of bf16."""
    check_weights(weights, cfg)
    assert precision in (0, 1, 2)
    hp = precision == 1
    mel_pad = N_MELS_PADDED_HP if hp else N_MELS_PADDED
    chunks: List[Tuple[int, np.ndarray]] = []
    off = [-1] * 256
    cur = 0

    def put(slot: int, arr: np.ndarray):
        nonlocal cur
        a = np.ascontiguousarray(arr)
        assert a.dtype in (np.uint16, np.float32), a.dtype
        assert 0 <= slot < 256 and off[slot] == -1, f"weight slot {slot} written twice (ecapa_layout.h index space)"
        off[slot] = cur
        chunks.append((cur, a.view(np.uint8).reshape(-1)))
        cur += (a.nbytes + ALIGN - 1) // ALIGN * ALIGN

    def gemm_w(w3: np.ndarray, cin_pad=None) -> np.ndarray:
        """[C_out, C_in, k] conv weight -> the slot content of the blob's precision"""
        return hp_weight_planes(conv_weight_kmajor_f32(w3, cin_pad)) if hp else conv_weight_kmajor(w3, cin_pad, precision)

    def tdnn(slot: int, name: str, cin_pad=None):
        put(slot + EL_W, gemm_w(weights[f"{name}.conv.w"], cin_pad))
        put(slot + EL_B, weights[f"{name}.conv.b"].astype(np.float32))
        s, sh = bn_affine(weights, f"{name}.bn")
        put(slot + EL_SCALE, s)
        put(slot + EL_SHIFT, sh)

    nb = len(cfg.dilations)
    blk0_pack = 0
    if not hp and cfg.n_mels % 8 == 0 and cfg.kernel0 > 1:
        # default mode: the first layer's taps packed along K - [C][round_up(kernel0 * n_mels, 64)] (5 x 80 = 400 -> 448: 7 K-steps of 64
        # where the per-tap padding 80 -> 128 took 10); the activations stay [M, 128] with zero columns 80.., read 80 wide per tap
        blk0_pack = cfg.n_mels
        wk = conv_weight_kmajor(weights["blk0.conv.w"], None, precision)      # [C, kernel0 * n_mels] bf16 / fp16 bits
        kp = (wk.shape[1] + 63) // 64 * 64
        wp = np.zeros((wk.shape[0], kp), dtype=np.uint16)
        wp[:, :wk.shape[1]] = wk
        put(EL_BLK0 + EL_W, wp)
        put(EL_BLK0 + EL_B, weights["blk0.conv.b"].astype(np.float32))
        s0, sh0 = bn_affine(weights, "blk0.bn")
        put(EL_BLK0 + EL_SCALE, s0)
        put(EL_BLK0 + EL_SHIFT, sh0)
    else:
        tdnn(EL_BLK0, "blk0", mel_pad)
    for i in range(1, nb + 1):
        b = block_base(i)
        tdnn(b + EL_TDNN1, f"blk{i}.tdnn1")
        for j in range(cfg.res2net_scale - 1):
            tdnn(b + res2net_slot(j), f"blk{i}.res2net.{j}")
            if cfg.sub_channels == 128 and i <= 4 and j < 7 and not hp:
                put(chainpack_slot(i, j), chain_fragment_order(conv_weight_kmajor(weights[f"blk{i}.res2net.{j}.conv.w"], None, precision)))
        tdnn(b + EL_TDNN2, f"blk{i}.tdnn2")
        put(b + EL_SE_W1T, weights[f"blk{i}.se.conv1.w"][:, :, 0].T.astype(np.float32))
        put(b + EL_SE_B1, weights[f"blk{i}.se.conv1.b"])
        put(b + EL_SE_W2T, weights[f"blk{i}.se.conv2.w"][:, :, 0].T.astype(np.float32))
        put(b + EL_SE_B2, weights[f"blk{i}.se.conv2.b"])
    t = tail_base(nb)
    tdnn(t + EL_MFA, "mfa")
    m = cfg.mfa_channels
    wt = weights["asp.tdnn.conv.w"][:, :, 0]
    put(t + EL_ASP_WH, hp_weight_planes(wt[:, :m]) if hp else to_bits16(wt[:, :m], precision))
    put(t + EL_ASP_WMS_T, wt[:, m:].T.astype(np.float32))
    put(t + EL_ASP_B, weights["asp.tdnn.conv.b"])
    s, sh = bn_affine(weights, "asp.tdnn.bn")
    put(t + EL_ASP_SCALE, s)
    put(t + EL_ASP_SHIFT, sh)
    put(t + EL_ASP_W2, hp_weight_planes(weights["asp.conv.w"][:, :, 0]) if hp else to_bits16(weights["asp.conv.w"][:, :, 0], precision))
    if cfg.attn_channels == 128 and m % 32 == 0 and not hp:
        put(t + EL_ASP_W2PACK, asp_w2_fragment_order(to_bits16(weights["asp.conv.w"][:, :, 0], precision)))
    put(t + EL_ASP_B2, weights["asp.conv.b"])
    s, sh = bn_affine(weights, "asp_bn")
    put(t + EL_ASPBN_SCALE, s)
    put(t + EL_ASPBN_SHIFT, sh)
    put(t + EL_FC_WT, weights["fc.w"][:, :, 0].T.astype(np.float32))
    put(t + EL_FC_B, weights["fc.b"])

    blob = np.zeros(cur, dtype=np.uint8)
    for o, a in chunks:
        blob[o:o + a.size] = a
    fields = dict(n_mels_padded=mel_pad, precision=precision, blk0_tap_pack=blk0_pack, channels=cfg.channels, sub_channels=cfg.sub_channels,
                  scale=cfg.res2net_scale, se_channels=cfg.se_channels, attn_channels=cfg.attn_channels,
                  mfa_channels=cfg.mfa_channels, embed_dim=cfg.embed_dim, n_blocks=nb, kernel0=cfg.kernel0,
                  dilation=list(cfg.dilations) + [0] * (4 - nb), off=off)
    return blob, fields


# ---------------------------------------------------------------------------------------------------------------------------------
# Bias correction of the bf16 WEIGHT rounding (round 3).  The error budget (profiles/r03_error_budget.md) says the default
# mode's 4.3e-3 score deviation is almost all weight rounding, and tools/ experiments say that error is almost all a per-output-channel
# CONSTANT: (W - bf16(W)) . mean(x) with a channel mean of the layer input that barely depends on the utterance (the input is a BatchNorm
# output).  Folding that constant into the layer's bias - standard post-training-quantisation bias correction - removes it at no run-time cost.
# The means come from a calibration pass on the GPU (sdk_ecapa_forward_calib on built-in synthetic audio; ops.Engine), never from the oracle.
def calib_layout(cfg: EcapaConfig = DEFAULT_CONFIG):
    """[(layer name, channels, float offset per segment)] in the slot order of sdk_ecapa_forward_calib (each slot = mean | std, 2 C floats)."""
    out, off = [], 0
    for i in range(1, len(cfg.dilations) + 1):
        out.append((f"blk{i}.tdnn1", cfg.channels, off)); off += 2 * cfg.channels
        for j in range(cfg.res2net_scale - 1):
            out.append((f"blk{i}.res2net.{j}", cfg.sub_channels, off)); off += 2 * cfg.sub_channels
        out.append((f"blk{i}.tdnn2", cfg.channels, off)); off += 2 * cfg.channels
    out.append(("mfa", cfg.mfa_channels, off)); off += 2 * cfg.mfa_channels
    out.append(("asp.tdnn", cfg.mfa_channels, off)); off += 2 * cfg.mfa_channels
    return out, off


def bias_corrections(weights: Dict[str, np.ndarray], means: Dict[str, np.ndarray], cfg: EcapaConfig = DEFAULT_CONFIG, precision: int = 0) -> Dict[str, np.ndarray]:
    """layer name -> corrected fp32 bias  b + (W - round16(W)) . mu  (float64 inside; every tap of a k3 conv sees the same channel means);
    round16 = the blob's 2-byte weight format: bf16 (precision 0) or fp16 (precision 2)."""
    out = {}
    for name, mu in means.items():
        w = weights[f"{name}.conv.w"].astype(np.float64)                     # [C_out, C_in(, k)]
        if name == "asp.tdnn":
            w = w[:, :cfg.mfa_channels]                                       # the per-frame part of the attention hidden layer (the context part is fp32)
        dw = w - round16(w.astype(np.float32), precision).astype(np.float64)
        corr = np.tensordot(dw.sum(axis=2) if dw.ndim == 3 else dw, np.asarray(mu, np.float64)[:dw.shape[1]], axes=([1], [0]))
        out[name] = (weights[f"{name}.conv.b"].astype(np.float64) + corr).astype(np.float32)
    return out


def bias_slot(name: str, cfg: EcapaConfig = DEFAULT_CONFIG) -> int:
    """descriptor slot that holds the fp32 bias of a corrected layer"""
    if name == "mfa":
        return tail_base(len(cfg.dilations)) + EL_MFA + EL_B
    if name == "asp.tdnn":
        return tail_base(len(cfg.dilations)) + EL_ASP_B
    blk, kind = name.split(".", 1)
    b = block_base(int(blk[3:]))
    if kind == "tdnn1":
        return b + EL_TDNN1 + EL_B
    if kind == "tdnn2":
        return b + EL_TDNN2 + EL_B
    return b + res2net_slot(int(kind.split(".")[1])) + EL_B


def calibration_tag() -> str:
    """'' for the built-in calibration audio, '-<8 hex>' naming the file in $SDK_CALIBRATION_WAV (path, size, mtime): part of the corrected blob's
    cache entry name, so another calibration recording is another entry."""
    import hashlib
    import os
    path = os.environ.get("SDK_CALIBRATION_WAV")
    if not path:
        return ""
    try:
        st = os.stat(path)
    except OSError as exc:
        raise ValueError(f"SDK_CALIBRATION_WAV={path}: cannot read the calibration recording ({exc.strerror or exc})") from exc
    return "-" + hashlib.sha256(f"{os.path.abspath(path)}|{st.st_size}|{st.st_mtime_ns}".encode()).hexdigest()[:8]


def calibration_pcm(n: int = 24, seed: int = 20240) -> np.ndarray:
    """Calibration audio of the bias correction: [segments, 32000] int16.  $SDK_CALIBRATION_WAV (a 16 kHz mono s16 WAVE file, the backend's audio
    contract): its two-second windows at a one-second hop, at most 64 - with a trained checkpoint, calibrate on SPEECH.
    Otherwise the built-in deterministic set, see _builtin_calibration_pcm."""
    import os
    path = os.environ.get("SDK_CALIBRATION_WAV")
    if path:
        from .wav import cut_windows, read_wav_s16
        pcm, _ = cut_windows(read_wav_s16(path), None)
        if len(pcm) == 0:
            raise ValueError(f"SDK_CALIBRATION_WAV={path}: shorter than 0.5 s")
        return np.ascontiguousarray(pcm[:64])
    return _builtin_calibration_pcm(n, seed)


def _builtin_calibration_pcm(n: int = 24, seed: int = 20240) -> np.ndarray:
    """Built-in calibration audio (deterministic): n two-second segments, half of them noise at several levels plus two tones, half of them
    harmonic stacks with a pitch, a spectral tilt and a slow amplitude modulation ("voices").  Only the per-channel MEANS of the layer inputs
    are taken from it; measured on noise, voices, quiet / clipped noise, 0.5-s and 5-s windows the correction always helps (2.3-5.7x,
    profiles/r03_bias_correction_generalisation*.json) - with a trained checkpoint, calibrate on speech."""
    rng = np.random.default_rng(seed)
    t = np.arange(32000, dtype=np.float32) / 16000.0
    h = n // 2
    level = rng.uniform(0.02, 0.2, (h, 1)).astype(np.float32)
    x = rng.standard_normal((h, 32000)).astype(np.float32) * level
    x += 0.2 * np.sin(2 * np.pi * rng.uniform(90, 400, (h, 1)).astype(np.float32) * t) + 0.1 * np.sin(2 * np.pi * rng.uniform(800, 3500, (h, 1)).astype(np.float32) * t)
    out = [np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)]
    v = np.zeros((n - h, 32000), np.float32)
    for i in range(n - h):
        f0 = rng.uniform(80, 260)
        tilt = rng.uniform(1.0, 1.6)
        for k in range(1, 12):
            v[i] += (0.5 / k ** tilt) * np.sin(2 * np.pi * f0 * k * t + rng.uniform(0, 6.28))
        v[i] = v[i] * (0.6 + 0.4 * np.sin(2 * np.pi * rng.uniform(2, 5) * t)) + rng.normal(0, 0.02, t.shape).astype(np.float32)
        v[i] *= 0.5 / np.abs(v[i]).max()
    out.append(np.clip(np.round(v * 32767.0), -32768, 32767).astype(np.int16))
    return np.concatenate(out)
