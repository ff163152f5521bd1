"""Bias correction of the bf16 weight rounding (round 3): the default mode's PCM -> score deviation from the fp32 model is
almost all WEIGHT rounding (error budget), and that error is almost all a per-channel constant (W - bf16(W)) . mean(layer input).  The product
folds it into the layer biases from a calibration pass on built-in synthetic audio - no run-time cost.  Here: the corrected engine computes
exactly the bf16 layer-boundary model of its effective weights (the blob was patched correctly), and its distance from the UN-ROUNDED model on
config-#2 segments drops from 4.3e-3 to under 1.6e-3 (measured ~8e-4), IDs unchanged."""
import importlib
import json
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, sub
from oracle import ecapa as oecapa
from oracle import fbank as ofbank

pytestmark = pytest.mark.gpu

W = sub("weights")
WP = sub("weights_pack")
OPS = sub("ops")
CORRECTED_BUDGET = 1.6e-3         # 2 x the ~8e-4 measured on MI355X (uncorrected: 4.3e-3)


@pytest.fixture(scope="module")
def corrected():
    return OPS.Engine(0, bias_correction=True)


def test_corrected_engine_is_the_bf16_model_of_its_effective_weights(engine, corrected):
    eff = corrected.effective_weights()
    plain = W.synthetic_weights(0)
    lay, _ = WP.calib_layout()
    changed = [n for n, _, _ in lay if not np.array_equal(eff[f"{n}.conv.b"], plain[f"{n}.conv.b"])]
    assert len(changed) == len(lay) == 29                                    # every corrected layer's bias moved, nothing else did
    assert all(np.array_equal(eff[k], plain[k]) for k in plain if not k.endswith(".conv.b"))
    assert np.array_equal(eff["blk0.conv.b"], plain["blk0.conv.b"])          # mean-normalised features: nothing to correct in blk0
    dmax = max(float(np.abs(eff[f"{n}.conv.b"] - plain[f"{n}.conv.b"]).max()) for n, _, _ in lay)
    assert 1e-4 < dmax < 10.0, dmax        # 2^-9 of |W| x channel means that grow with depth under these synthetic weights (0.02 in block 1, 0.8 in block 3)
    g = torch.Generator().manual_seed(31)
    feats = torch.randn(3, 201, 80, generator=g) * 3.0
    f = torch.zeros(3 * 201, 128, dtype=torch.bfloat16)
    f[:, :80] = feats.reshape(-1, 80).to(torch.bfloat16)
    emb = corrected.ecapa_forward(f.cuda(), 3, 201).cpu()
    want = oecapa.EcapaOracle(eff, "bf16", torch.float64).embed(feats)
    cos = (emb.double() * want.double()).sum(1) / (emb.double().norm(dim=1) * want.double().norm(dim=1))
    assert (cos > 1 - 2e-5).all(), cos
    # ... and it is NOT the model of the plain weights any more (the shared fixture engine is)
    emb0 = engine.ecapa_forward(f.cuda(), 3, 201).cpu()
    assert float((emb - emb0).abs().max()) > 1e-4 * float(emb0.abs().max())


def test_bias_correction_brings_the_default_mode_closer_to_the_fp32_model(engine, corrected):
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    n = 200
    pcm = bench.synth_pcm(n, seed=0)
    P = bench.unit_rows(100, 192, seed=1)
    model = oecapa.EcapaOracle(W.synthetic_weights(0), "fp32", torch.float32)
    Eo = oecapa.l2_normalise(np.concatenate([model.embed(torch.from_numpy(ofbank.fbank(pcm[a:a + 50]))).numpy() for a in range(0, n, 50)]))
    rep = {}
    for name, eng in (("plain", engine), ("corrected", corrected)):
        E, Eb, re = eng.embed_pcm(torch.from_numpy(pcm).cuda())
        Pn, Pb, rp = eng.l2norm(torch.from_numpy(P).cuda())
        gi, gs = eng.affinity_topk(E, Eb, re, Pn, Pb, rp.max().reshape(1), k=1)
        torch.cuda.synchronize()
        r = bench.parity_object(E.cpu().numpy(), gi.cpu().numpy()[:, 0], gs.cpu().numpy()[:, 0], Eo, P)
        rep[name] = {k: r[k] for k in ("max_abs_dscore_all_pairs", "max_abs_dscore_top1", "min_cos_embedding", "id_mismatches")}
    print("\nbias correction, 200 segments x 100 profiles vs the fp32 model:", json.dumps(rep))
    assert rep["corrected"]["id_mismatches"] == 0 and rep["plain"]["id_mismatches"] == 0
    assert rep["corrected"]["max_abs_dscore_all_pairs"] < CORRECTED_BUDGET
    assert rep["corrected"]["max_abs_dscore_all_pairs"] < 0.4 * rep["plain"]["max_abs_dscore_all_pairs"]
    assert rep["plain"]["max_abs_dscore_all_pairs"] > 2.5e-3                   # (the fixture engine really is the uncorrected one)


LITE_CHILD = r"""
import importlib, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
lite = importlib.import_module(sys.argv[2] + ".lite")
eng = lite.LiteEngine(0, cache_key=sys.argv[3], bias_correction=True)
assert eng.cache_entry_name() == "0c" and eng.has_cached_weights()
E = eng.embed_pcm(np.load(sys.argv[4]))
assert eng.cache_hit and "torch" not in sys.modules
np.save(sys.argv[5], E)
"""


def test_cache_entry_0c_is_bit_identical_across_fresh_hit_and_lite(tmp_path, monkeypatch):
    """ADVICE r3 (medium) / VERDICT r3 next #2a: the shipped default through the packed-blob cache.  The process that calibrates and stores entry
    "0c", a second engine that maps it, and the torch-free host path reading the same entry give bit-identical embeddings and identical
    effective biases."""
    import subprocess
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    WC = sub("weights_cache")
    monkeypatch.setenv("SDK_CACHE_DIR", str(tmp_path / "cache"))
    monkeypatch.delenv("SDK_WEIGHTS_CACHE", raising=False)
    w = W.synthetic_weights(0)
    key = WC.key_for_seed(0, W.DEFAULT_CONFIG)
    digest = W.weights_digest(w)
    mk = lambda: OPS.Engine(0, cache_key=key, weights_fn=lambda: w, digest_fn=lambda: digest, bias_correction=True)  # noqa: E731
    pcm = bench.synth_pcm(5, seed=9)
    cold = mk()
    E_cold = cold.embed_pcm(torch.from_numpy(pcm).cuda())[0].cpu()
    assert cold.cache_hit is False
    assert WC.load_blob(key, "0c") is not None and WC.load_blob(key, 0) is None          # the corrected blob is its own entry; the plain one was not stored
    assert WC.load_meta(key)["digest"] == digest
    warm = mk()
    E_warm = warm.embed_pcm(torch.from_numpy(pcm).cuda())[0].cpu()
    assert warm.cache_hit is True and torch.equal(E_cold, E_warm)
    eff_c, eff_w = cold.effective_weights(), warm.effective_weights()                    # warm: read back from the device blob
    assert set(eff_c) == set(eff_w) and all(np.array_equal(eff_c[k], eff_w[k]) for k in eff_c)
    assert any(not np.array_equal(eff_c[k], w[k]) for k in w)
    np.save(tmp_path / "pcm.npy", pcm)
    r = subprocess.run([sys.executable, "-c", LITE_CHILD, str(ROOT), OPS.__package__, key, str(tmp_path / "pcm.npy"), str(tmp_path / "E_lite.npy")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert np.array_equal(np.load(tmp_path / "E_lite.npy"), E_cold.numpy())


def _input_kinds(n):
    """The six input kinds of tools/bias_corr_generalise.py (same generators, n segments each)."""
    bench = importlib.import_module("bench")
    rng = np.random.default_rng(77)

    def voices(m, S):
        t = np.arange(S) / 16000.0
        out = []
        for _ in range(m):
            f0 = rng.uniform(80, 260)
            x = sum((0.5 / h ** rng.uniform(1.0, 1.6)) * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28)) for h in range(1, 12))
            x = x * (0.6 + 0.4 * np.sin(2 * np.pi * rng.uniform(2, 5) * t)) + rng.normal(0, 0.02, t.shape)
            out.append(np.clip(np.round(x / np.abs(x).max() * 0.5 * 32767), -32768, 32767).astype(np.int16))
        return np.stack(out)

    return {
        "config2_other_seed": bench.synth_pcm(n, seed=9),
        "harmonic_voices": voices(n, 32000),
        "quiet_white_noise": np.clip(np.round(rng.normal(0, 0.01, (n, 32000)) * 32768), -32768, 32767).astype(np.int16),
        "loud_clipped_noise": np.clip(np.round(rng.normal(0, 0.6, (n, 32000)) * 32768), -32768, 32767).astype(np.int16),
        "half_second_windows": bench.synth_pcm(n, seed=10)[:, :8000].copy(),
        "five_second_windows": np.concatenate([bench.synth_pcm(n, seed=11), bench.synth_pcm(n, seed=12), bench.synth_pcm(n, seed=13)[:, :16000]], axis=1),
    }


def test_corrected_budget_holds_on_other_input_kinds(corrected, engine):
    """the correction is calibrated once on 24 built-in synthetic segments; assert that it still helps on inputs unlike
    the calibration set - each kind with its own bound of 2 x the deviation measured on MI355X over 64 segments
    (profiles/r03_bias_correction_generalisation.json: 7.3e-4, 1.5e-3, 8.3e-4, 1.0e-3, 1.2e-3, 2.3e-3)."""
    sys.path.insert(0, str(ROOT))
    P = importlib.import_module("bench").unit_rows(100, 192, seed=1).astype(np.float64)
    model = oecapa.EcapaOracle(W.synthetic_weights(0), "fp32", torch.float32)
    bounds = {"config2_other_seed": 1.5e-3, "harmonic_voices": 3.0e-3, "quiet_white_noise": 1.7e-3, "loud_clipped_noise": 2.1e-3,
              "half_second_windows": 2.4e-3, "five_second_windows": 4.7e-3}
    rep = {}
    for kind, pcm in _input_kinds(16).items():
        Eo = oecapa.l2_normalise(model.embed(torch.from_numpy(ofbank.fbank(pcm))).numpy()).astype(np.float64)
        dev = {}
        for name, eng in (("plain", engine), ("corrected", corrected)):
            E = eng.embed_pcm(torch.from_numpy(pcm).cuda())[0].cpu().numpy().astype(np.float64)
            dev[name] = float(np.abs(E @ P.T - Eo @ P.T).max())
        rep[kind] = dev
        assert dev["corrected"] < bounds[kind], (kind, dev)
        assert dev["corrected"] < dev["plain"], (kind, dev)
    print("\nbias correction on other input kinds (max |dscore| vs the fp32 model):", json.dumps(rep))


def test_calibration_on_caller_supplied_audio(tmp_path, monkeypatch, engine):
    """SDK_CALIBRATION_WAV: the calibration pass runs on the caller's recording (with a trained checkpoint: speech) instead of
    the built-in synthetic set; the engine is then the bf16 model of THOSE effective weights, the correction still helps, and the cache entry is
    named after the calibration file."""
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    wav = sub("wav")
    WC = sub("weights_cache")
    rec = np.concatenate(list(_input_kinds(6)["harmonic_voices"]))                       # 12 s of "voices"
    wav.write_wav_s16(tmp_path / "cal.wav", rec)
    monkeypatch.setenv("SDK_CALIBRATION_WAV", str(tmp_path / "cal.wav"))
    monkeypatch.setenv("SDK_CACHE_DIR", str(tmp_path / "cache"))
    assert WP.calibration_pcm().shape == (11, 32000) and WP.calibration_tag().startswith("-")
    w = W.synthetic_weights(0)
    key = WC.key_for_seed(0, W.DEFAULT_CONFIG)
    eng = OPS.Engine(0, cache_key=key, weights_fn=lambda: w, digest_fn=lambda: "d", bias_correction=True)
    eff = eng.effective_weights()
    assert WC.load_blob(key, "0c" + WP.calibration_tag()) is not None and WC.load_blob(key, "0c") is None
    monkeypatch.delenv("SDK_CALIBRATION_WAV")
    builtin = OPS.Engine(0, bias_correction=True).effective_weights()
    assert any(not np.array_equal(eff[k], builtin[k]) for k in eff)                        # other audio, other means, other biases
    pcm = bench.synth_pcm(16, seed=3)
    P = bench.unit_rows(100, 192, seed=1).astype(np.float64)
    feats = torch.from_numpy(ofbank.fbank(pcm))
    E = eng.embed_pcm(torch.from_numpy(pcm).cuda())[0].cpu()
    want = torch.from_numpy(oecapa.l2_normalise(oecapa.EcapaOracle(eff, "bf16", torch.float64).embed(feats).numpy()))
    assert float(((E.double() * want.double()).sum(1)).min()) > 1 - 2e-5
    E32 = oecapa.l2_normalise(oecapa.EcapaOracle(w, "fp32", torch.float32).embed(feats).numpy()).astype(np.float64)
    Ep = engine.embed_pcm(torch.from_numpy(pcm).cuda())[0].cpu().numpy().astype(np.float64)
    d_corr = float(np.abs(E.numpy().astype(np.float64) @ P.T - E32 @ P.T).max())
    d_plain = float(np.abs(Ep @ P.T - E32 @ P.T).max())
    print("\ncalibrated on a caller-supplied recording: plain", d_plain, "corrected", d_corr)
    assert d_corr < 0.6 * d_plain
