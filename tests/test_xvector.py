"""The x-vector (plain TDNN) family - north_star: "ECAPA-TDNN/x-vector forward pass".  CPU: the oracle against an independently composed
torch.nn model (Conv1d with reflect padding, BatchNorm1d.eval()), the packer's padding rules.  GPU: sdk_xvector_forward (one C call:
sdk_conv_gemm per frame layer - the first with its taps packed along K -, sdk_asp_stats, sdk_rows_fc) against the oracle's bf16 model,
at the 2-s window (T = 201) and at short / odd lengths, and through fbank -> L2 -> cosine k4 on the shared 192-d back end."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import sub
from oracle import xvector as oxv
from oracle import ecapa as oecapa
from oracle import scoring as oscoring

XV = sub("xvector")


def test_xvector_oracle_against_torch_nn():
    cfg = XV.XVectorConfig(channels=(128, 128, 128, 128, 384), embed_dim=64)
    w = XV.synthetic_weights(3, cfg)
    feats = torch.randn(2, 60, 80, generator=torch.Generator().manual_seed(1), dtype=torch.float64) * 3
    x = feats.transpose(1, 2)
    cin = (80,) + cfg.channels[:-1]
    with torch.no_grad():
        for l, (k, dil) in enumerate(zip(cfg.kernels, cfg.dilations)):
            conv = torch.nn.Conv1d(cin[l], cfg.channels[l], k, dilation=dil, padding=dil * (k - 1) // 2, padding_mode="reflect").double()
            conv.weight.data = torch.from_numpy(w[f"frame{l}.conv.w"]).double(); conv.bias.data = torch.from_numpy(w[f"frame{l}.conv.b"]).double()
            bn = torch.nn.BatchNorm1d(cfg.channels[l], eps=1e-5).double().eval()
            bn.weight.data = torch.from_numpy(w[f"frame{l}.bn.gamma"]).double(); bn.bias.data = torch.from_numpy(w[f"frame{l}.bn.beta"]).double()
            bn.running_mean.data = torch.from_numpy(w[f"frame{l}.bn.mean"]).double(); bn.running_var.data = torch.from_numpy(w[f"frame{l}.bn.var"]).double()
            x = bn(torch.relu(conv(x)))
        stats = torch.cat([x.mean(dim=2), x.std(dim=2, unbiased=False)], dim=1)
        ref = stats @ torch.from_numpy(w["embed.w"]).double().T + torch.from_numpy(w["embed.b"]).double()
    got = oxv.xvector_embed(w, feats.float(), cfg.kernels, cfg.dilations, mode="fp32")
    assert got.shape == (2, 64) and float((got.double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())


def test_xvector_packer_pads_channels_and_packs_the_first_layer():
    WP = sub("weights_pack")
    cfg = XV.DEFAULT_XVECTOR
    w = XV.synthetic_weights(1, cfg)
    blob, d = XV.pack_weights(w, cfg)
    assert list(d.cout)[:5] == [512, 512, 512, 512, 1536] and d.first_tap_pack == 80 and d.n_frame_layers == 5 and d.embed_dim == 192
    assert cfg.macs_per_frame() == 80 * 5 * 512 + 2 * 3 * 512 * 512 + 512 * 512 + 512 * 1500
    w0 = blob[d.off[0]:d.off[0] + 512 * 448 * 2].view(np.uint16).reshape(512, 448)
    assert not w0[:, 400:].any()
    assert np.array_equal(w0[:, :400].reshape(512, 5, 80), WP.f32_to_bf16_bits(np.transpose(w["frame0.conv.w"], (0, 2, 1))))
    w4 = blob[d.off[16]:d.off[16] + 1536 * 512 * 2].view(np.uint16).reshape(1536, 512)
    assert not w4[1500:].any()                                               # padded output channels: zero weights ...
    s4 = blob[d.off[18]:d.off[18] + 1536 * 4].view(np.float32)
    assert (s4[1500:] == 1).all() and not blob[d.off[19]:d.off[19] + 1536 * 4].view(np.float32)[1500:].any()   # ... BN scale 1, shift 0
    fc = blob[d.off[60]:d.off[60] + 3072 * 192 * 4].view(np.float32).reshape(3072, 192)
    assert not fc[1500:1536].any() and not fc[3036:].any()                   # the embedding layer ignores the padded statistics
    assert np.array_equal(fc[:1500], w["embed.w"][:, :1500].T) and np.array_equal(fc[1536:3036], w["embed.w"][:, 1500:].T)
    with pytest.raises(ValueError):
        XV.pack_weights({k: v for k, v in w.items() if k != "embed.b"}, cfg)


def _cos(a, b):
    a, b = a.double(), b.double()
    return (a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))


@pytest.mark.gpu
@pytest.mark.parametrize("B,T", [(4, 201), (3, 50), (2, 301), (1, 9)])
def test_xvector_forward_matches_oracle(engine, B, T):
    """Full-size x-vector (512-512-512-512-1500, 192-d embedding) vs the oracle's bf16 layer-boundary model: same tolerance as the
    ECAPA-TDNN forward (fp32-vs-float64 accumulation and rare last-bit bf16 flips only)."""
    WP = sub("weights_pack")
    w = XV.synthetic_weights(0)
    xv = XV.XVector(engine, w)
    feats = torch.randn(B, T, 80, generator=torch.Generator().manual_seed(T)) * 3.0
    f = torch.zeros(B * T, WP.N_MELS_PADDED, dtype=torch.bfloat16)
    f[:, :80] = feats.reshape(-1, 80).to(torch.bfloat16)
    f[:, 80:] = 5.0                                                          # must never be read as data (packed taps read 80 wide)
    emb = xv.forward(f.cuda(), B, T).cpu()
    torch.cuda.synchronize()
    want = oxv.xvector_embed(w, feats, mode="bf16")
    assert (_cos(emb, want) > 1 - 2e-5).all(), _cos(emb, want)
    assert torch.allclose(emb, want, rtol=0, atol=2e-3 * float(want.abs().max())), float((emb - want).abs().max())
    assert (_cos(emb, oxv.xvector_embed(w, feats, mode="fp32")) > 0.999).all()


@pytest.mark.gpu
def test_xvector_pcm_to_assignment(engine):
    """PCM -> fbank -> x-vector -> L2 -> cosine argmax vs profiles: the 192-d embeddings go through the SAME k3 / k4 as the ECAPA ones;
    IDs identical to and scores within 1e-5 of the exact scan of the GPU's own embeddings; refuses precise mode loudly."""
    import importlib, sys
    from conftest import ROOT
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    xv = XV.XVector(engine, seed=2)
    pcm = torch.from_numpy(bench.synth_pcm(40, seed=5)).cuda()
    E, Eb, re = xv.embed_pcm(pcm)
    assert E.shape == (40, 192) and float((E.double().norm(dim=1) - 1).abs().max()) < 1e-6
    P = bench.unit_rows(30, 192, seed=6)
    Pn, Pb, rp = engine.l2norm(torch.from_numpy(P).cuda())
    idx, sc = engine.affinity_topk(E, Eb, re, Pn, Pb, rp.max().reshape(1), k=1)
    torch.cuda.synchronize()
    oidx, osc = oscoring.affinity_topk(E.cpu().numpy(), Pn.cpu().numpy(), 1)
    assert np.array_equal(idx.cpu().numpy(), oidx) and np.abs(sc.cpu().numpy() - osc).max() <= 1e-5
    # the numerical contract is the BLOB's, per call (round 5): the context's "precision" option - mutable state shared by
    # every engine of a device - plays no part in the forward, so another engine (or thread) that set it cannot change this one's results
    L = sub("_lib")
    ws = torch.empty(engine.lib.sdk_xvector_workspace_bytes(C.byref(xv.desc), 2, 201), dtype=torch.uint8, device="cuda")
    feats = engine.fbank(pcm[:2].contiguous())
    outs = []
    for opt in (0, 1):
        engine.set_option("precision", opt)
        try:
            out = torch.empty(2, 192, device="cuda")
            L.check(engine.lib.sdk_xvector_forward(engine.ctx, xv.blob.data_ptr(), C.byref(xv.desc), feats.data_ptr(), feats.stride(0), 2, 201, ws.data_ptr(), ws.numel(),
                                                   out.data_ptr(), None), "sdk_xvector_forward")
            torch.cuda.synchronize()
            outs.append(out)
        finally:
            engine.set_option("precision", 0)
    assert torch.equal(outs[0], outs[1])


def test_backend_model_selection_metadata(monkeypatch):
    """SDK_MODEL selects the family behind the SAME plug-in class (base.py:291-293: zero-arg constructible); model_version names it, so the
    reference's prefix rule (base.py:92-93) accepts both and store.load_profile_batch's exact match keeps the two spaces apart."""
    B = sub("backend")
    monkeypatch.setenv("SDK_MODEL", "xvector")
    be = B.Backend()
    assert be.model == "xvector" and be.embedding_dim == 192 and be.name == "mi355x"
    mv = be.model_version
    assert mv.startswith("mi355x-xvector512-") and len(mv.split("-")[-1]) == 12 and mv == B.Backend().model_version
    assert be.check_embedding_compatibility({"model_version": mv})["compatible"] is True
    monkeypatch.setenv("SDK_MODEL", "ecapa")
    assert B.Backend().model == "ecapa"
    monkeypatch.setenv("SDK_MODEL", "resnet")
    with pytest.raises(ValueError, match="SDK_MODEL"):
        B.Backend()


def test_backend_xvector_weight_file(monkeypatch, tmp_path):
    B = sub("backend")
    cfg = XV.DEFAULT_XVECTOR
    w = XV.synthetic_weights(5, cfg)
    np.savez(tmp_path / "xv.npz", **w)
    monkeypatch.setenv("SDK_MODEL", "xvector")
    monkeypatch.setenv("SDK_XVECTOR_WEIGHTS", str(tmp_path / "xv.npz"))
    be = B.Backend()
    assert be.model_version == f"mi355x-xvector512-{sub('weights').weights_digest(w)}"
    bad = dict(w); bad["embed.w"] = bad["embed.w"][:, :-1]
    np.savez(tmp_path / "bad.npz", **bad)
    monkeypatch.setenv("SDK_XVECTOR_WEIGHTS", str(tmp_path / "bad.npz"))
    with pytest.raises(ValueError, match="embed.w"):
        B.Backend().model_version


@pytest.mark.gpu
def test_backend_xvector_enroll_identify_roundtrip(tmp_path, monkeypatch):
    """The x-vector family through the drop-in boundary (enroll_speaker / identify_speaker, base.py:107-151): stored vectors and window
    scores against the oracle's x-vector on the same windows; vectors enrolled with it are refused by the ECAPA-TDNN configuration."""
    from oracle import fbank as ofbank
    from test_gpu_backend_e2e import _voice
    wav = sub("wav")
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path / "store"))
    monkeypatch.setenv("SDK_MODEL", "xvector")
    be = sub("backend").Backend()
    w = XV.synthetic_weights(0)

    def oracle_embed(pcm):
        return oecapa.l2_normalise(oxv.xvector_embed(w, torch.from_numpy(ofbank.fbank(pcm)), mode="bf16").numpy())

    profiles = []
    for i, (sid, f0) in enumerate({"alice": 140.0, "bob": 95.0}.items()):
        path = tmp_path / f"enroll_{sid}.wav"
        wav.write_wav_s16(path, _voice(10 + i, 6.0, f0))
        rec = be.enroll_speaker(path, [(0.5, 5.5)])
        assert rec["model_version"] == be.model_version and rec["model_version"].startswith("mi355x-xvector512-") and rec["embedding_dim"] == 192
        profiles.append({"id": sid, "names": {"default": sid}, "embeddings": {"mi355x": [
            {"id": f"emb-{sid}", "external_id": rec["external_id"], "model_version": rec["model_version"], "trust_level": "high"}]}})
        pcm, _ = wav.cut_windows(wav.read_wav_s16(path), [(0.5, 5.5)])
        e = oracle_embed(pcm).astype(np.float64).mean(0)
        assert float(np.load(rec["file"]) @ (e / np.linalg.norm(e))) > 1 - 1e-4, "stored enrollment vector vs the x-vector oracle"
    tpath = tmp_path / "meeting.wav"
    wav.write_wav_s16(tpath, np.concatenate([_voice(40, 4.0, 95.0), _voice(41, 4.0, 140.0)]))
    rows = be.identify_speaker(tpath, profiles, threshold=-1.0)
    assert {r["speaker_id"] for r in rows} == {"alice", "bob"} and all(r["confidence"] == r["similarity"] for r in rows)
    pcm, _ = wav.cut_windows(wav.read_wav_s16(tpath), None)
    Eo = oracle_embed(pcm)
    E, Eb, re = be.embed_windows(pcm)
    assert ((E.cpu().numpy().astype(np.float64) * Eo).sum(1) > 1 - 1e-4).all()
    batch = sub("store").load_profile_batch(profiles, "mi355x", model_prefix="mi355x-", model_version=be.model_version)
    gidx, gsc = be.score_windows(E, Eb, re, batch)
    oidx, osc = oscoring.affinity_topk(Eo, oecapa.l2_normalise(batch.matrix), 1)
    full = np.sort(oscoring.affinity(Eo, oecapa.l2_normalise(batch.matrix)), axis=1)
    clear = (full[:, -1] - full[:, -2]) > 1e-3
    assert np.array_equal(gidx[clear, 0], oidx[clear, 0]) and np.abs(gsc[:, 0] - osc[:, 0])[clear].max() < 5e-4
    # the other family must not compare against these vectors: same backend name, other model_version
    monkeypatch.setenv("SDK_MODEL", "ecapa")
    with pytest.raises(ValueError, match="enrolled under other weights"):
        sub("backend").Backend().identify_speaker(tpath, profiles)


@pytest.mark.gpu
def test_xvector_bias_correction_is_the_bf16_model_of_its_effective_weights_and_closer_to_fp32(engine):
    """the second model family gets the same post-training bias correction of the bf16 weight rounding
    (b' = b + (W - bf16(W)) . mu on frame layers 1..4, means measured on the GPU by a calibration pass).  The corrected extractor computes exactly
    the bf16 layer-boundary model of its EFFECTIVE weights, and its PCM -> score deviation from the fp32 model shrinks."""
    import importlib, sys
    from conftest import ROOT
    from oracle import fbank as ofbank
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    w = XV.synthetic_weights(0)
    plain = XV.XVector(engine, w, bias_correction=False)
    corr = XV.XVector(engine, w, bias_correction=True)
    eff = corr.effective_weights()
    changed = [k for k in w if not np.array_equal(eff[k], w[k])]
    assert sorted(changed) == [f"frame{l}.conv.b" for l in (1, 2, 3, 4)]              # layer 0 reads mean-normalised features: nothing to correct
    assert plain.effective_weights() is plain.weights
    pcm = bench.synth_pcm(48, seed=2)
    feats = torch.from_numpy(ofbank.fbank(pcm))
    P = bench.unit_rows(100, 192, seed=1).astype(np.float64)
    Ec = corr.embed_pcm(torch.from_numpy(pcm).cuda())[0].cpu()
    Ep = plain.embed_pcm(torch.from_numpy(pcm).cuda())[0].cpu()
    want = torch.from_numpy(oecapa.l2_normalise(oxv.xvector_embed(eff, feats, mode="bf16").numpy()))
    assert (_cos(Ec, want) > 1 - 2e-5).all()
    E32 = oecapa.l2_normalise(oxv.xvector_embed(w, feats, mode="fp32").numpy()).astype(np.float64)
    dev = {n: float(np.abs(E.numpy().astype(np.float64) @ P.T - E32 @ P.T).max()) for n, E in (("plain", Ep), ("corrected", Ec))}
    print("\nx-vector PCM -> score deviation from the fp32 model (48 segments x 100 profiles):", dev)
    assert dev["corrected"] < 0.75 * dev["plain"], dev


@pytest.mark.gpu
def test_xvector_fp16_mode_is_the_11_bit_model_and_5x_closer_to_fp32(engine):
    """Precision 2 for the second family (round 5): the default layout and kernels with one fp16 plane (off[62] = 2, SDK_GEMM_F16 per layer,
    sdk_asp_stats_fmt, features from sdk_fbank_fmt(..., 2)).  Plain weights: the forward is the oracle's fp16 layer-boundary model (the bf16 test's
    tolerance scaled by 2^-3: 8 -> 11 significand bits).  Corrected: the fp16 model of its effective weights; PCM -> score deviation from the fp32
    model well below the corrected bf16 default's."""
    import importlib, sys
    from conftest import ROOT
    from oracle import fbank as ofbank
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    w = XV.synthetic_weights(0)
    eng = sub("ops").Engine(0, bias_correction=False)              # an engine of its own: the shared one stays in its format
    plain = XV.XVector(eng, w, bias_correction=False, precision=2)
    corr = XV.XVector(eng, w, bias_correction=True, precision=2)
    bf = XV.XVector(engine, w, bias_correction=True)
    assert int(plain.desc.off[62]) == 2 and plain.desc.first_tap_pack == 80
    eff = corr.effective_weights()
    assert sorted(k for k in w if not np.array_equal(eff[k], w[k])) == [f"frame{l}.conv.b" for l in (1, 2, 3, 4)]
    pcm = bench.synth_pcm(48, seed=2)
    feats = torch.from_numpy(ofbank.fbank(pcm))
    P = bench.unit_rows(100, 192, seed=1).astype(np.float64)
    dpcm = torch.from_numpy(pcm).cuda()
    Ep, Ec = plain.embed_pcm(dpcm)[0].cpu(), corr.embed_pcm(dpcm)[0].cpu()
    assert eng.precision == 2 and engine.precision == 0
    Eb = bf.embed_pcm(dpcm)[0].cpu()
    for E, ww in ((Ep, w), (Ec, eff)):
        want = torch.from_numpy(oecapa.l2_normalise(oxv.xvector_embed(ww, feats, mode="fp16").numpy()))
        assert (_cos(E, want) > 1 - 2e-5 / 8).all(), float((1 - _cos(E, want)).max())
    E32 = oecapa.l2_normalise(oxv.xvector_embed(w, feats, mode="fp32").numpy()).astype(np.float64)
    dev = {n: float(np.abs(E.numpy().astype(np.float64) @ P.T - E32 @ P.T).max()) for n, E in (("fp16_plain", Ep), ("fp16_corrected", Ec), ("bf16_corrected", Eb))}
    print("\nx-vector PCM -> score deviation from the fp32 model (48 segments x 100 profiles):", dev)
    assert dev["fp16_corrected"] < 0.25 * dev["bf16_corrected"] and dev["fp16_plain"] < 0.25 * dev["bf16_corrected"] * 4, dev
    # the format travels with the blob: the forward does not consult the context's default
    f2 = eng.fbank(dpcm[:2].contiguous())
    assert torch.equal(plain.forward(f2, 2, 201), plain.forward(f2, 2, 201))


@pytest.mark.gpu
def test_xvector_precise_mode_meets_1e5(engine):
    """the x-vector family in the precise mode (fp16 hi+lo planes, three MFMAs per product, sdk_conv_gemm_hp per frame layer,
    pooling on the planes): PCM -> cosine score within north_star's 1e-5 of the un-rounded model (float64 accumulation), IDs identical."""
    import importlib, sys
    from conftest import ROOT
    from oracle import fbank as ofbank
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    w = XV.synthetic_weights(0)
    xv = XV.XVector(engine, w, precision=1)
    assert xv.bias_correction is False and int(xv.desc.off[62]) == 1 and xv.desc.n_feats == 96 and xv.desc.first_tap_pack == 0
    pcm = bench.synth_pcm(32, seed=4)
    P = bench.unit_rows(100, 192, seed=1)
    try:
        E, Eb, re = xv.embed_pcm(torch.from_numpy(pcm).cuda())
        assert engine.precision == 1
        Pn, Pb, rp = engine.l2norm(torch.from_numpy(P).cuda())
        gi, gs = engine.affinity_topk(E, Eb, re, Pn, Pb, rp.max().reshape(1), k=1)
        torch.cuda.synchronize()
    finally:
        engine.set_precision(0)
    Eo = oecapa.l2_normalise(oxv.xvector_embed(w, torch.from_numpy(ofbank.fbank(pcm)), mode="fp32", acc=torch.float64).numpy())
    par = bench.parity_object(E.cpu().numpy(), gi.cpu().numpy()[:, 0], gs.cpu().numpy()[:, 0], Eo, P)
    print("\nx-vector precise mode vs the un-rounded oracle:", {k: par[k] for k in ("max_abs_dscore_all_pairs", "max_abs_dscore_top1", "min_cos_embedding", "id_mismatches")})
    assert par["max_abs_dscore_all_pairs"] <= 1e-5 and par["id_mismatches"] == 0
    # and the default-mode extractor still runs afterwards on the same engine (the context follows each extractor's contract)
    E0 = XV.XVector(engine, w, bias_correction=False).embed_pcm(torch.from_numpy(pcm).cuda())[0]
    assert float((E0.cpu().double() * torch.from_numpy(Eo).double()).sum(1).min()) > 0.999
