"""PCM -> score parity of the DEFAULT (bf16-operand) GPU path against the UN-ROUNDED fp32 model, not only against the bf16 layer-boundary
model the kernels are shaped like (VERDICT r1 weak #1, r2 next #1b): ALL 1000 config-#2 segments x 100 profiles, plus a NEAR-TIE set
that samples the ID-agreement rate against the fp32 decision margin.  Reports max |d score|, the rows whose argmax ID differs and their
fp32 margins; asserts what bf16 operands can promise: IDs identical wherever the fp32 margin exceeds twice the measured deviation, and
the deviation itself inside a budget of 2x what was measured when the budget was written.  north_star's 1e-5 / identical-ID criterion holds
for k4 GIVEN the embeddings (tests/test_gpu_kernels.py::test_affinity_*) and for the whole path in the PRECISE mode
(tests/test_gpu_precise.py); the error budget of this mode is profiles/r03_error_budget.md (the bf16 WEIGHTS make 4.15e-3 of it)."""
import importlib
import json
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, sub
from oracle import ecapa as oecapa
from oracle import fbank as ofbank
from oracle import scoring as oscoring

pytestmark = pytest.mark.gpu

W = sub("weights")
# measured on MI355X, round 3, 1000 segments x 100 profiles: max |d score| 4.31e-3 over all 100 000 pairs (4.0e-3 on the first 64 segments), 0 IDs differ;
# the budget is 2x that, so a regression by a factor of two fails (round 2's 2e-2 would have let a 4x regression through)
BF16_SCORE_BUDGET = 8.7e-3
N_SEG = 1000


@pytest.fixture(scope="module")
def both(engine):
    """GPU embeddings and the un-rounded oracle's embeddings of the same 1000 segments.  The oracle accumulates in float32 here (30 s
    instead of 4 min on 16 host threads); its distance from float64 accumulation is 3e-7 on a score (error-budget table), three
    orders of magnitude below what this test measures."""
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    pcm = bench.synth_pcm(N_SEG, seed=0)
    P = bench.unit_rows(100, 192, seed=1)
    E, Eb, re = engine.embed_pcm(torch.from_numpy(pcm).cuda())
    torch.cuda.synchronize()
    model = oecapa.EcapaOracle(W.synthetic_weights(0), "fp32", torch.float32)
    Eo = np.concatenate([model.embed(torch.from_numpy(ofbank.fbank(pcm[a:a + 50]))).numpy() for a in range(0, N_SEG, 50)])
    return bench, P, (E, Eb, re), oecapa.l2_normalise(Eo)


def _gpu_top1(engine, Et, P):
    Pn, Pb, rp = engine.l2norm(torch.from_numpy(np.ascontiguousarray(P, dtype=np.float32)).cuda())
    idx, sc = engine.affinity_topk(*Et, Pn, Pb, rp.max().reshape(1), k=1)
    torch.cuda.synchronize()
    return idx.cpu().numpy()[:, 0], sc.cpu().numpy()[:, 0]


def test_pcm_to_score_against_the_fp32_oracle(engine, both):
    bench, P, Et, Eo = both
    gi, gs = _gpu_top1(engine, Et, P)
    rep = bench.parity_object(Et[0].cpu().numpy(), gi, gs, Eo, P)
    print("\nparity vs fp32 oracle:", json.dumps(rep))
    assert rep["segments"] == N_SEG and rep["profiles"] == 100
    # every ID mismatch must sit inside the measured deviation (a decision the fp32 model itself holds by less than that)
    for m in rep["mismatches"]:
        assert m["fp32_margin"] <= 2.0 * rep["max_abs_dscore_all_pairs"], m
    assert rep["ids_identical_where_margin_exceeds_bound"] is True
    assert rep["max_abs_dscore_all_pairs"] < BF16_SCORE_BUDGET and rep["min_cos_embedding"] > 0.9997
    assert rep["max_abs_dscore_all_pairs"] > 1e-4, "suspiciously exact for bf16 operands: is this the default mode?"
    # the GPU's own reported top-1 score equals its embedding's exact cosine (k4's 1e-5 criterion, given the embeddings)
    assert rep["max_abs_top1_score_vs_own_embedding"] <= 1e-5


def near_tie_report(engine, Et, Eo, P, bound, eps_list=(1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 3e-2, 0.1, 0.4)):
    """Profiles built as p + eps * q around each segment's fp32 winner: the decision between a profile and its perturbed twin is held by
    a margin of order eps * 1, so sweeping eps samples how often the bf16 path keeps the fp32 model's ID as a function of that margin."""
    rng = np.random.default_rng(7)
    edges = [0.0, 1e-5, 1e-4, 1e-3, 2 * bound, np.inf]
    agree = np.zeros(len(edges) - 1, np.int64)
    total = np.zeros(len(edges) - 1, np.int64)
    worst_kept = 0.0
    for eps in eps_list:
        q = rng.standard_normal(P.shape)
        twin = P.astype(np.float64) + eps * q / np.linalg.norm(q, axis=1, keepdims=True)
        twin = (twin / np.linalg.norm(twin, axis=1, keepdims=True)).astype(np.float32)
        Pj = np.concatenate([P, twin]).astype(np.float32)                        # every profile next to its perturbed twin: the fp32 margin is ~0.07 eps
        So = Eo.astype(np.float64) @ Pj.astype(np.float64).T
        oi = So.argmax(1)
        srt = np.sort(So, axis=1)
        margin = srt[:, -1] - srt[:, -2]
        gi, _ = _gpu_top1(engine, Et, Pj)
        same = gi == oi
        # a different ID is acceptable only as an (almost-)tie of the fp32 model itself
        lost = ~same
        if lost.any():
            worst_kept = max(worst_kept, float(margin[lost].max()))
        b = np.digitize(margin, edges) - 1
        for k in range(len(edges) - 1):
            total[k] += int((b == k).sum())
            agree[k] += int((same & (b == k)).sum())
    return {"margin_bins": [f"[{edges[k]:g}, {edges[k + 1]:g})" for k in range(len(edges) - 1)], "rows": total.tolist(), "ids_agree": agree.tolist(),
            "agreement_rate": [round(float(a) / t, 4) if t else None for a, t in zip(agree, total)], "largest_fp32_margin_of_a_changed_id": worst_kept,
            "bound_used": bound, "eps": list(eps_list)}


def test_near_tie_id_agreement_vs_fp32_margin(engine, both):
    bench, P, Et, Eo = both
    S_gpu = Et[0].cpu().numpy().astype(np.float64) @ P.astype(np.float64).T
    bound = float(np.abs(S_gpu - Eo.astype(np.float64) @ P.astype(np.float64).T).max())
    rep = near_tie_report(engine, Et, Eo, P, bound)
    print("\nnear-tie ID agreement vs fp32 margin:", json.dumps(rep))
    assert sum(rep["rows"]) == 8 * N_SEG
    assert rep["rows"][1] + rep["rows"][2] + rep["rows"][3] > 1000, "the sweep must actually sample margins below the deviation"
    # the promise: every decision the fp32 model holds by more than twice the measured deviation is kept
    assert rep["largest_fp32_margin_of_a_changed_id"] <= 2.0 * bound
    assert rep["rows"][-1] > 100 and rep["agreement_rate"][-1] == 1.0
    # ... below that the rate is what it is (measured round 3, twins per segment: 99.0 % under 1e-5, 99.8 % in [1e-5, 1e-4), 100 % from 1e-4 up; twins
    # per profile: 100 % in every bin but one row at 7.7e-4: the
    # deviation of an embedding moves the scores of a profile and of its near twin almost equally, so near-tie decisions survive far
    # better than the worst-case bound says); only sanity is asserted
    rates = [r for r, n in zip(rep["agreement_rate"], rep["rows"]) if n >= 100]
    assert rates[-1] >= rates[0] and min(rates) > 0.9
