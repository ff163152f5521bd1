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
