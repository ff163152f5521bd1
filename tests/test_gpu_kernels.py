"""GPU parity tests: every HIP kernel, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Integer / index results must be identical; floating point within the tolerance
written next to each assert (bf16 layer-boundary model: DESIGN.md §3)."""
import numpy as np
import pytest
import torch

from conftest import sub
from oracle import ecapa as oecapa
from oracle import fbank as ofbank
from oracle import scoring as oscoring

pytestmark = pytest.mark.gpu

W = sub("weights")
WP = sub("weights_pack")
OPS = sub("ops")
SdkError = sub("_lib").SdkError

BF16_ULP = 2.0 ** -8          # relative spacing of bf16 (8 significand bits)


def dev(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def assert_bf16_close(got: torch.Tensor, want_f32: torch.Tensor, what: str, max_ulps: float = 2.0, frac_exact: float = 0.98,
                      magnitude: torch.Tensor = None):
    """got: bf16 result of the GPU; want_f32: the oracle's pre-rounding fp32 value.
    Tolerance: every element within `max_ulps` bf16 ulps of the oracle value (fp32-vs-fp64
    accumulation can flip the last bf16 bit), and >= frac_exact of the elements bit-identical to
    the rounded oracle.  `magnitude` (optional) replaces |oracle| as the ulp scale where the value is a
    difference of larger terms (the fp32 rounding error follows the terms, not the cancelled result)."""
    g = got.float().cpu()
    w = bf16_round(want_f32.float().cpu())
    mag = w.abs() if magnitude is None else torch.maximum(w.abs(), magnitude.float().cpu())
    tol = max_ulps * BF16_ULP * 1.01 * mag.clamp_min(1e-30) + 1e-30      # 1.01: a last-bit flip across a binade edge
    bad = (g - w).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} / {bad.numel()} elements off by more than {max_ulps} bf16 ulps; worst {float((g - w).abs().max())}"
    same = float((g == w).float().mean())
    assert same >= frac_exact, f"{what}: only {same:.4f} of elements bit-identical to the rounded oracle"


# ------------------------------------------------------------------------------------------------
# conv_gemm
def _conv_ref(A, Wt, Cin, taps, dil, T, bias=None, scale=None, shift=None, ubias=None, relu=False, tanh=False):
    """float64 oracle of sdk_conv_gemm on bf16-valued inputs.  A [M, lda>=Cin], Wt [N, taps*Cin]."""
    A = A.double()[:, :Cin]
    M = A.shape[0]
    B = M // T
    t = torch.arange(T)
    out = torch.zeros(M, Wt.shape[0], dtype=torch.float64)
    Ab = A.reshape(B, T, Cin)
    for j in range(taps):
        src = oecapa.reflect_index(t + (j - taps // 2) * dil, T)
        out += (Ab[:, src, :].reshape(M, Cin)) @ Wt.double()[:, j * Cin:(j + 1) * Cin].T
    out = out.float()
    if bias is not None:
        out = out + bias
    if ubias is not None:
        out = out + ubias.repeat_interleave(T, dim=0)
    if relu:
        out = torch.relu(out)
    if scale is not None:
        out = out * scale + shift
    if tanh:
        out = torch.tanh(out)
    return out


def test_conv_gemm_identity_layout(engine):
    """A = I (exact), asymmetric integer W: catches any row/col swap or k-permutation exactly."""
    K = N = 128
    A = torch.eye(K, dtype=torch.float32)
    Wt = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251) - 125      # W[n][k], asymmetric
    C, _, _ = engine.conv_gemm(dev(A, torch.bfloat16), dev(Wt, torch.bfloat16), N, K)
    torch.cuda.synchronize()
    assert torch.equal(C.float().cpu(), Wt.T.contiguous()), "C[m][n] must equal W[n][m] for A = I"


@pytest.mark.parametrize("M,T,N,Cin,taps,dil", [
    (384, 384, 128, 64, 1, 1), (300, 100, 256, 128, 1, 1), (603, 201, 128, 128, 3, 2), (402, 201, 128, 128, 3, 4),
    (250, 50, 256, 128, 5, 1), (1005, 201, 384, 192, 3, 3), (130, 130, 128, 3072, 1, 1), (64, 8, 128, 64, 3, 3),
    (1005, 201, 512, 192, 3, 3), (520, 130, 256, 64, 5, 1), (256, 256, 256, 3072, 1, 1), (2010, 201, 1024, 128, 3, 4), (257, 257, 256, 64, 1, 1)])
def test_conv_gemm_integer_exact(engine, M, T, N, Cin, taps, dil):
    """Small-integer operands: every product and partial sum is exact in fp32, so the GPU result must
    equal the oracle bit for bit (tests the reflect row gather, tails and the K loop)."""
    g = torch.Generator().manual_seed(M * 7 + N)
    lda = Cin + 64                                         # strided input (slab of a wider tensor)
    A = torch.randint(-3, 4, (M, lda), generator=g).float()
    Wt = torch.randint(-2, 3, (N, taps * Cin), generator=g).float()
    _, C32, _ = engine.conv_gemm(dev(A, torch.bfloat16)[:, :], dev(Wt, torch.bfloat16), N, Cin, taps=taps, dil=dil, T=T,
                                 out_bf16=False, out_f32=True)
    want = _conv_ref(A, Wt, Cin, taps, dil, T)
    torch.cuda.synchronize()
    assert torch.equal(C32.cpu(), want), f"max diff {float((C32.cpu() - want).abs().max())}"


@pytest.mark.parametrize("M,T,N,C,taps,dil", [(1005, 201, 256, 80, 5, 1), (2010, 201, 1024, 80, 5, 1), (250, 50, 256, 80, 5, 1), (402, 201, 128, 24, 3, 2),
                                              (603, 201, 256, 40, 7, 1)])
def test_conv_gemm_packed_taps_integer_exact(engine, M, T, N, C, taps, dil):
    """tap_pack (round 3): the taps share K - W is [N, round_up(taps * C, 64)], a 16-byte chunk of 8 channels belongs to tap chunk / (C / 8),
    the K padding multiplies finite activations by zero weights.  Small integers: bit-exact against the oracle, in the 256^2 kernel
    (M >= 256, N % 256 == 0) and in the 128^2 kernel (the other shapes).  blk0 of the model is the first case's shape with N = 1024."""
    g = torch.Generator().manual_seed(M + N + C)
    lda = 128                                             # the activations stay 128 wide (feats), C channels are read per tap
    A = torch.randint(-1, 2, (M, lda), generator=g).float()
    Wt = torch.randint(-1, 2, (N, taps * C), generator=g).float()
    kp = (taps * C + 63) // 64 * 64
    Wp = torch.zeros(N, kp)
    Wp[:, :taps * C] = Wt
    A[:, C:] = 7.0                                        # columns beyond C must never be read as data (only as zero-weighted padding)
    want = _conv_ref(A, Wt, C, taps, dil, T)
    assert float(want.abs().max()) < 256                  # exactly representable in bf16: the 256^2 kernel (bf16 output only) is exact too
    Ad, Wd = dev(A, torch.bfloat16), dev(Wp, torch.bfloat16)
    Cb, _, _ = engine.conv_gemm(Ad, Wd, N, C, taps=taps, dil=dil, T=T, tap_pack=C)                                  # 256^2 kernel where the shape allows
    _, C32, _ = engine.conv_gemm(Ad, Wd, N, C, taps=taps, dil=dil, T=T, out_bf16=False, out_f32=True, tap_pack=C)   # always the 128^2 kernel
    torch.cuda.synchronize()
    assert torch.equal(Cb.float().cpu(), want), f"max diff {float((Cb.float().cpu() - want).abs().max())}"
    assert torch.equal(C32.cpu(), want), f"max diff {float((C32.cpu() - want).abs().max())}"


def test_conv_gemm_epilogue(engine):
    M, T, N, Cin = 402, 201, 256, 128
    g = torch.Generator().manual_seed(5)
    A = bf16_round(torch.randn(M, Cin, generator=g))
    Wt = bf16_round(torch.randn(N, 3 * Cin, generator=g) * 0.05)
    bias, shift = torch.randn(N, generator=g), torch.randn(N, generator=g) * 0.1
    scale = torch.rand(N, generator=g) + 0.5
    ubias = torch.randn(M // T, N, generator=g)
    X2 = bf16_round(torch.randn(M, N, generator=g))
    for relu, tanh in [(True, False), (True, True), (False, False)]:
        C, C32, S = engine.conv_gemm(dev(A, torch.bfloat16), dev(Wt, torch.bfloat16), N, Cin, taps=3, dil=2, T=T, bias=dev(bias),
                                     scale=dev(scale), shift=dev(shift), ubias=dev(ubias), relu=relu, tanh=tanh, out_f32=True,
                                     X2=dev(X2, torch.bfloat16))
        want = _conv_ref(A, Wt, Cin, 3, 2, T, bias, scale, shift, ubias, relu, tanh)
        torch.cuda.synchronize()
        # fp32 epilogue vs float64-accumulated oracle: |diff| <= 2e-5 * (1 + |want|)  (K = 384 fp32 accumulation)
        assert torch.allclose(C32.cpu(), want, rtol=2e-5, atol=2e-5), float((C32.cpu() - want).abs().max())
        assert_bf16_close(C, want, f"C relu={relu} tanh={tanh}", magnitude=torch.full_like(want, 2e-2))   # |pre| ~ 1: fp32 error 2e-5 abs
        # S is defined on the ROUNDED C: S = bf16(float(bf16(v)) + X2)
        assert torch.equal(S.float().cpu(), bf16_round(C.float().cpu() + X2)), "S = bf16(C + X2)"


# ------------------------------------------------------------------------------------------------
# per-utterance kernels
def test_se_gate_residual(engine):
    B, T, C, Cse = 5, 201, 1024, 128
    g = torch.Generator().manual_seed(1)
    z = bf16_round(torch.randn(B * T, C, generator=g))
    x = bf16_round(torch.randn(B * T, C, generator=g))
    w1 = torch.randn(Cse, C, generator=g) / 32
    w2 = torch.randn(C, Cse, generator=g) / 11
    b1, b2 = torch.randn(Cse, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    out = engine.se_gate_residual(dev(z, torch.bfloat16), dev(x, torch.bfloat16), dev(w1.T.contiguous()), dev(b1),
                                  dev(w2.T.contiguous()), dev(b2), B, T, split=True)
    mono = engine.se_gate_residual(dev(z, torch.bfloat16), dev(x, torch.bfloat16), dev(w1.T.contiguous()), dev(b1),
                                   dev(w2.T.contiguous()), dev(b2), B, T, split=False)
    mean = z.double().reshape(B, T, C).mean(1)
    h = torch.relu(mean @ w1.double().T + b1)
    gate = torch.sigmoid(h @ w2.double().T + b2).float()
    want = gate[:, None, :] * z.reshape(B, T, C) + x.reshape(B, T, C)
    torch.cuda.synchronize()
    mag = (gate[:, None, :] * z.reshape(B, T, C)).abs() + x.reshape(B, T, C).abs()
    assert_bf16_close(out, want.reshape(B * T, C), "se_gate_residual (split schedule)", frac_exact=0.995, magnitude=mag.reshape(B * T, C) * 2 ** -9)
    assert_bf16_close(mono, want.reshape(B * T, C), "se_gate_residual (one kernel per segment)", frac_exact=0.995, magnitude=mag.reshape(B * T, C) * 2 ** -9)


def test_asp_stats_and_pool(engine):
    B, T, C = 3, 201, 3072
    g = torch.Generator().manual_seed(2)
    h = bf16_round(torch.randn(B * T, C, generator=g) * 20 + 5)
    logits = torch.randn(B * T, C, generator=g) * 3
    ctx = engine.asp_stats(dev(h, torch.bfloat16), B, T)
    pooled = engine.asp_pool(dev(logits), dev(h, torch.bfloat16), B, T)
    hd = h.double().reshape(B, T, C)
    mu = hd.mean(1)
    sd = ((hd - mu[:, None]) ** 2).mean(1).clamp_min(1e-12).sqrt()
    w = torch.softmax(logits.double().reshape(B, T, C), dim=1)
    wmu = (w * hd).sum(1)
    wsd = (w * (hd - wmu[:, None]) ** 2).sum(1).clamp_min(1e-12).sqrt()
    torch.cuda.synchronize()
    # fp32 single-sweep statistics vs float64: rtol 2e-5 on values of magnitude ~20
    assert torch.allclose(ctx.cpu().double(), torch.cat([mu, sd], 1), rtol=2e-5, atol=2e-5)
    assert torch.allclose(pooled.cpu().double(), torch.cat([wmu, wsd], 1), rtol=5e-5, atol=5e-5)


@pytest.mark.parametrize("B,T", [(3, 201), (2, 224), (4, 50), (1, 9), (2, 97)])
def test_asp_fused_matches_unfused_oracle(engine, B, T):
    C, A = 3072, 128
    g = torch.Generator().manual_seed(T)
    h = bf16_round(torch.randn(B * T, C, generator=g) * 20 + 5)
    ah = bf16_round(torch.tanh(torch.randn(B * T, A, generator=g)))
    w2 = bf16_round(torch.randn(C, A, generator=g) * 0.3)
    b2 = torch.randn(C, generator=g)
    pooled = engine.asp_fused(dev(ah, torch.bfloat16), dev(w2, torch.bfloat16), dev(b2), dev(h, torch.bfloat16), B, T)
    logits = (ah.double() @ w2.double().T + b2).reshape(B, T, C)
    hd = h.double().reshape(B, T, C)
    w = torch.softmax(logits, dim=1)
    wmu = (w * hd).sum(1)
    wsd = (w * (hd - wmu[:, None]) ** 2).sum(1).clamp_min(1e-12).sqrt()
    torch.cuda.synchronize()
    # fp32 MFMA accumulation (K = 128) + __expf + shifted fp32 moments vs float64: rtol 1e-4 on values ~ 5..25
    assert torch.allclose(pooled.cpu().double(), torch.cat([wmu, wsd], 1), rtol=1e-4, atol=1e-4), \
        float((pooled.cpu().double() - torch.cat([wmu, wsd], 1)).abs().max())


def test_rows_fc(engine):
    g = torch.Generator().manual_seed(3)
    for B, Cin, Nout, act in [(7, 6144, 192, 0), (5, 6144, 128, 0), (9, 1000, 70, 1), (1, 33, 200, 2), (1000, 1024, 128, 1), (77, 128, 1024, 2), (33, 96, 40, 0),
                               (3, 8192, 64, 0), (3, 8320, 64, 0)]:      # input affine in LDS at its 64-KB limit / past it (plain kernel)
        x = torch.randn(B, Cin, generator=g)
        wt = torch.randn(Cin, Nout, generator=g) / Cin ** 0.5
        bias, isc, ish = torch.randn(Nout, generator=g), torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
        out = engine.rows_fc(dev(x), dev(wt), dev(bias), dev(isc), dev(ish), act)
        want = (x.double() * isc + ish) @ wt.double() + bias
        want = torch.relu(want) if act == 1 else torch.sigmoid(want) if act == 2 else want
        torch.cuda.synchronize()
        assert torch.allclose(out.cpu().double(), want, rtol=1e-5, atol=2e-5), (B, Cin, Nout, float((out.cpu().double() - want).abs().max()))


def test_l2norm(engine):
    g = torch.Generator().manual_seed(4)
    X = torch.randn(1001, 192, generator=g) * 7
    X[5] = 0                                                 # zero row: must not produce NaN
    E, Eb, r = engine.l2norm(dev(X))
    want = torch.from_numpy(oecapa.l2_normalise(X.numpy()))
    torch.cuda.synchronize()
    assert torch.allclose(E.cpu(), want, rtol=0, atol=2e-7)                       # fp32 rounding only
    assert torch.equal(Eb.float().cpu(), bf16_round(E.cpu()))                       # bf16 copy is RNE of E
    rr = (E.cpu().double() - Eb.double().cpu()).norm(dim=1)
    assert torch.allclose(r.cpu().double(), rr, rtol=1e-4, atol=1e-9)
    assert torch.isfinite(E).all()


# ------------------------------------------------------------------------------------------------
# k1 fbank
def _pcm(B, S, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(S) / 16000.0
    x = rng.normal(0, 0.1, (B, S))
    for b in range(B):                                        # two sinusoids per segment so the spectrum is not flat
        x[b] += 0.2 * np.sin(2 * np.pi * (200 + 37 * b) * t) + 0.1 * np.sin(2 * np.pi * (1800 + 91 * b) * t)
    return np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("B,S", [(3, 32000), (2, 16000), (1, 4805), (2, 800), (2, 80000)])   # 80000 samples: T = 501 > the LDS-tile normaliser's limit
def test_fbank(engine, B, S):
    pcm = _pcm(B, S, 11)
    feats = engine.fbank(torch.from_numpy(pcm).cuda())
    want = ofbank.fbank(pcm)                                   # [B, T, 80] float32 (float64 internally)
    T = want.shape[1]
    torch.cuda.synchronize()
    got = feats.float().cpu().reshape(B, T, -1)
    assert torch.count_nonzero(got[:, :, 80:]) == 0, "padding channels must be zero"
    # fp32 DFT + log vs float64, then bf16 storage: within 2 bf16 ulps (+1e-3 dB absolute for values near 0)
    w = torch.from_numpy(want)
    err = (got[:, :, :80] - bf16_round(w)).abs()
    tol = 2 * BF16_ULP * w.abs() + 2e-3
    assert (err <= tol).all(), f"fbank: worst {float(err.max())} dB at |x|={float(w.abs().flatten()[err.argmax()])}"
    assert float((got[:, :, :80] == bf16_round(w)).float().mean()) > 0.97


# ------------------------------------------------------------------------------------------------
# k2 ECAPA-TDNN forward
SMALL = W.EcapaConfig(channels=256, mfa_channels=768, res2net_scale=2, se_channels=64, attn_channels=128)


def _feats(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, T, 80, generator=g) * 3.0)


def _run_forward(cfg, weights, feats):
    eng = OPS.Engine(0, weights=weights, cfg=cfg)
    B, T, _ = feats.shape
    f = torch.zeros(B * T, WP.N_MELS_PADDED, dtype=torch.bfloat16)
    f[:, :80] = feats.reshape(B * T, 80).to(torch.bfloat16)
    emb = eng.ecapa_forward(f.cuda(), B, T)
    torch.cuda.synchronize()
    return emb.cpu()


def _cos(a, b):
    a, b = a.double(), b.double()
    return (a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))


@pytest.mark.parametrize("B,T", [(3, 50), (2, 9), (1, 201)])
def test_ecapa_forward_small_config(B, T):
    weights = W.synthetic_weights(3, SMALL)
    feats = _feats(B, T, 21)
    emb = _run_forward(SMALL, weights, feats)
    o = oecapa.EcapaOracle(weights, "bf16", torch.float64, n_dilations=SMALL.dilations, scale=SMALL.res2net_scale)
    want = o.embed(feats)
    # bf16 layer-boundary model on both sides; the only differences are fp32-vs-fp64 accumulation and the
    # resulting rare last-bit bf16 flips: cosine >= 1 - 2e-5 and elementwise 2e-3 of the embedding scale.
    c = _cos(emb, want)
    assert (c > 1 - 2e-5).all(), c
    assert torch.allclose(emb, want, rtol=0, atol=2e-3 * float(want.abs().max())), float((emb - want).abs().max())


def test_ecapa_forward_full_config(engine):
    """C = 1024 (the BASELINE.json architecture), 4 two-second segments (T = 201)."""
    weights = W.synthetic_weights(0)
    feats = _feats(4, 201, 22)
    f = torch.zeros(4 * 201, WP.N_MELS_PADDED, dtype=torch.bfloat16)
    f[:, :80] = feats.reshape(-1, 80).to(torch.bfloat16)
    emb = engine.ecapa_forward(f.cuda(), 4, 201).cpu()
    want = oecapa.EcapaOracle(weights, "bf16", torch.float64).embed(feats)
    c = _cos(emb, want)
    assert (c > 1 - 2e-5).all(), c
    assert torch.allclose(emb, want, rtol=0, atol=2e-3 * float(want.abs().max())), float((emb - want).abs().max())
    # and the bf16 model itself stays close to the unrounded fp32 model (reported, loose bound)
    ref32 = oecapa.EcapaOracle(weights, "fp32", torch.float64).embed(feats)
    assert (_cos(emb, ref32) > 0.999).all()


def test_ecapa_forward_kblocked_h_is_bit_identical(engine):
    """Round 4: the MFA output h is written K-blocked ([C / 64][M][64]) where the skinny attention-hidden GEMM and the per-segment ASP are its
    only readers (T = 201: yes; T = 64: no - asp_stats and the short-window ASP read it row-major).  A layout, not arithmetic: the embeddings
    equal the row-major schedule's bit for bit (option h_kblocked)."""
    for B, T in [(6, 201), (3, 130), (2, 64)]:
        feats = _feats(B, T, 40 + T)
        f = torch.zeros(B * T, WP.N_MELS_PADDED, dtype=torch.bfloat16)
        f[:, :80] = feats.reshape(-1, 80).to(torch.bfloat16)
        f = f.cuda()
        on = engine.ecapa_forward(f, B, T).clone()
        engine.set_option("h_kblocked", 0)
        try:
            off = engine.ecapa_forward(f, B, T).clone()
        finally:
            engine.set_option("h_kblocked", 1)
        assert torch.equal(on, off), (B, T, float((on - off).abs().max()))
        assert engine.lib.sdk_asp_kblocked_ok(engine.ctx, T, 3072) == (1 if T > 96 else 0)


@pytest.mark.parametrize("M,T,N,Cin", [(2010, 201, 3072, 128), (1000, 200, 256, 64), (2613, 201, 512, 192)])
def test_conv_gemm_kblocked_output_and_input(engine, M, T, N, Cin):
    """SDK_GEMM_C_KBLOCKED: the 256^2 kernel's copy-out writes [N / 64][M][64] - the same values (and the same fused column statistics) as the
    row-major output, edge tiles included.  SDK_GEMM_A_KBLOCKED: the 128^2 kernel reads that layout - the same product as from row-major A."""
    g = torch.Generator().manual_seed(M + N)
    A = dev(bf16_round(torch.randn(M, Cin, generator=g)), torch.bfloat16)
    Wt = dev(bf16_round(torch.randn(N, Cin, generator=g) * 0.2), torch.bfloat16)
    bias, sc, sh = dev(torch.randn(N, generator=g)), dev(torch.rand(N, generator=g) + 0.5), dev(torch.randn(N, generator=g))
    C0, _, _, st0 = engine.conv_gemm(A, Wt, N, Cin, T=T, bias=bias, scale=sc, shift=sh, relu=True, stats_mode=2)
    C1, _, _, st1 = engine.conv_gemm(A, Wt, N, Cin, T=T, bias=bias, scale=sc, shift=sh, relu=True, stats_mode=2, c_kblocked=True)
    assert torch.equal(st0, st1)
    assert C1.shape == (N // 64, M, 64)
    assert torch.equal(engine.from_kblocked(C1), C0)
    assert torch.equal(engine.to_kblocked(C0), C1)
    # the K-blocked tensor as the A operand of a skinny layer (N2 = 128: the attention-hidden shape)
    W2 = dev(bf16_round(torch.randn(128, N, generator=g) * 0.05), torch.bfloat16)
    ub = dev(torch.randn(M // T, 128, generator=g))
    D0, _, _ = engine.conv_gemm(C0, W2, 128, N, T=T, ubias=ub, relu=True, tanh=True)
    D1, _, _ = engine.conv_gemm(C1, W2, 128, N, T=T, ubias=ub, relu=True, tanh=True, a_kblocked=True)
    torch.cuda.synchronize()
    assert torch.equal(D0, D1)


def test_conv_gemm_kblocked_refusals(engine):
    A = torch.zeros(512, 64, dtype=torch.bfloat16, device="cuda")
    Wt = torch.zeros(128, 64, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(SdkError, match="K-blocked output needs"):
        engine.conv_gemm(A, Wt, 128, 64, c_kblocked=True)                       # N = 128: not the 256^2 kernel's shape
    W3 = torch.zeros(256, 192, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(SdkError, match="K-blocked A operand needs taps == 1"):
        engine.conv_gemm(engine.to_kblocked(A), W3, 256, 64, taps=3, T=128, a_kblocked=True)
    with pytest.raises(ValueError, match="K-blocked A must be"):
        engine.conv_gemm(A, Wt, 128, 64, a_kblocked=True)


@pytest.mark.parametrize("B,T", [(3, 201), (2, 97), (2, 208)])
def test_asp_fused_kblocked_is_bit_identical(engine, B, T):
    C, A = 3072, 128
    g = torch.Generator().manual_seed(T + 7)
    h = dev(bf16_round(torch.randn(B * T, C, generator=g) * 20 + 5), torch.bfloat16)
    ah = dev(bf16_round(torch.tanh(torch.randn(B * T, A, generator=g))), torch.bfloat16)
    w2 = dev(bf16_round(torch.randn(C, A, generator=g) * 0.3), torch.bfloat16)
    b2 = dev(torch.randn(C, generator=g))
    p0 = engine.asp_fused(ah, w2, b2, h, B, T)
    p1 = engine.asp_fused(ah, w2, b2, engine.to_kblocked(h), B, T, kblocked=True)
    torch.cuda.synchronize()
    assert torch.equal(p0, p1)
    with pytest.raises(SdkError, match="only the per-segment form"):
        engine.asp_fused(ah[:50], w2, b2, engine.to_kblocked(h[:50]), 1, 50, kblocked=True)


@pytest.mark.parametrize("B,T", [(2, 301), (2, 501)])
def test_ecapa_forward_full_config_long_windows(engine, B, T):
    """SDK_WINDOW_S lets a user pick any window.  T > 208 drops the Res2Net chain to seven conv_gemm launches
    (sdk_res2net_chain_max_frames) and T > 224 takes the unfused ASP path (fp32 logits in HBM + sdk_asp_pool): those pieces are
    tested alone above; here the whole C = 1024 forward at 3-s / 5-s windows against the bf16-model oracle, same tolerance as T = 201."""
    weights = W.synthetic_weights(0)
    feats = _feats(B, T, 23 + T)
    f = torch.zeros(B * T, WP.N_MELS_PADDED, dtype=torch.bfloat16)
    f[:, :80] = feats.reshape(-1, 80).to(torch.bfloat16)
    assert T > engine.lib.sdk_res2net_chain_max_frames() and T > engine.lib.sdk_asp_fused_max_frames()
    emb = engine.ecapa_forward(f.cuda(), B, T).cpu()
    want = oecapa.EcapaOracle(weights, "bf16", torch.float64).embed(feats)
    c = _cos(emb, want)
    assert (c > 1 - 2e-5).all(), c
    assert torch.allclose(emb, want, rtol=0, atol=2e-3 * float(want.abs().max())), float((emb - want).abs().max())


# ------------------------------------------------------------------------------------------------
# k4 affinity + top-k
def _unit(n, d, seed):
    x = np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)
    return oecapa.l2_normalise(x)


def _score_gpu(engine, E, P, k):
    En, Eb, re = engine.l2norm(dev(E))
    Pn, Pb, rp = engine.l2norm(dev(P))
    idx, sc, cnt = engine.affinity_topk(En, Eb, re, Pn, Pb, rp.max().reshape(1), k=k, want_count=True)
    torch.cuda.synchronize()
    return idx.cpu().numpy(), sc.cpu().numpy(), int(cnt.item()), En.cpu().numpy(), Pn.cpu().numpy()


@pytest.mark.parametrize("N,P,k", [(1000, 100, 1), (5000, 1000, 1), (777, 45, 3), (129, 3, 3), (300, 2500, 4), (1, 1, 1), (4096, 1024, 2), (2000, 10000, 1), (513, 1025, 2)])
def test_affinity_topk_matches_oracle(engine, N, P, k):
    E, Pm = _unit(N, 192, N + P), _unit(P, 192, 7 * P + 1)
    idx, sc, cnt, En, Pn = _score_gpu(engine, E, Pm, k)
    oidx, osc = oscoring.affinity_topk(En, Pn, k)          # oracle on the SAME normalised rows
    # scores: fp32 fma chain vs float64 -> 1e-5 absolute (north_star tolerance); typically ~1e-7
    assert np.abs(sc - osc).max() <= 1e-5, np.abs(sc - osc).max()
    # integer IDs identical, except where the oracle itself has a near-tie below fp32 resolution
    full = oscoring.affinity(En, Pn)
    mism = np.argwhere(idx != oidx)
    for n, j in mism:
        assert abs(full[n, idx[n, j]] - full[n, oidx[n, j]]) <= 2e-7, (n, j, idx[n], oidx[n])
    assert len(mism) <= max(1, N // 1000), f"{len(mism)} index mismatches"
    # k = 1: the 3-deep per-half candidate lists certify all but ~1 % of the rows; larger k sits next to
    # the list depth, so more rows are (correctly) certified by the exact rescan instead.
    if k == 1:
        assert cnt <= max(4, N // 20), f"{cnt} of {N} rows needed the exact rescan"


def test_affinity_exact_ties(engine):
    """Duplicate profiles give bitwise-equal fp32 scores: the lower profile index must rank first."""
    base = _unit(40, 192, 99)
    Pm = np.concatenate([base, base[:10]], 0)              # rows 40..49 duplicate rows 0..9
    E = _unit(500, 192, 101)
    E[:10] = base[:10] + 0.05 * _unit(10, 192, 102)        # segments sitting next to a duplicated profile
    idx, sc, cnt, En, Pn = _score_gpu(engine, E, Pm, 2)
    oidx, osc = oscoring.affinity_topk(En, Pn, 2)
    assert np.abs(sc - osc).max() <= 1e-5
    for n in range(10):
        assert idx[n, 0] == n and idx[n, 1] == n + 40 and sc[n, 0] == sc[n, 1], (n, idx[n], sc[n])
    assert np.array_equal(idx, oidx)


def test_affinity_near_duplicates_force_rescan(engine):
    """Profiles that differ by less than the bf16 rounding of the coarse pass: the certification must
    notice it cannot separate them and take the exact fp32 rescan - results still equal the oracle."""
    base = _unit(40, 192, 99)
    near = base + 1e-4 * _unit(40, 192, 100)               # indistinguishable in bf16
    Pm = np.concatenate([base, near, near + 1e-4 * _unit(40, 192, 103), base + 2e-4 * _unit(40, 192, 104),
                         base + 3e-4 * _unit(40, 192, 105)], 0)
    E = _unit(500, 192, 101)
    E[:40] = base + 0.05 * _unit(40, 192, 102)
    idx, sc, cnt, En, Pn = _score_gpu(engine, E, Pm, 1)
    full = (En.astype(np.float64) @ Pn.astype(np.float64).T)
    oidx, osc = oscoring.affinity_topk(En, Pn, 1)
    assert np.abs(sc - osc).max() <= 1e-5
    for n in np.argwhere(idx[:, 0] != oidx[:, 0]).flatten():
        assert abs(full[n, idx[n, 0]] - full[n, oidx[n, 0]]) <= 2e-7, n      # below fp32 resolution of the score itself
    assert cnt >= 30, f"only {cnt} rows took the exact rescan; the construction should force ~40"


def test_affinity_rescan_is_deterministic_and_sliced(engine):
    """The k = 1 exact rescan splits the profiles of a flagged row quad into slices that meet through agent-scope atomics (64-bit
    (score, index) keys + an arrival counter): whichever slice arrives last, the answer must be the oracle's - many flagged rows,
    several slices per quad, repeated runs."""
    P = 1024
    base = _unit(64, 192, 7)
    Pm = np.concatenate([base + (1e-4 * q) * _unit(64, 192, 200 + q) for q in range(P // 64)], 0)   # 16 near-copies of every direction
    E = _unit(3000, 192, 8)
    E[:1500] = base[np.arange(1500) % 64] + 0.05 * _unit(1500, 192, 9)
    ref = None
    for rep in range(6):
        idx, sc, cnt, En, Pn = _score_gpu(engine, E, Pm, 1)
        if ref is None:
            ref = (idx.copy(), sc.copy())
            oidx, osc = oscoring.affinity_topk(En, Pn, 1)
            full = (En.astype(np.float64) @ Pn.astype(np.float64).T)
            assert np.abs(sc - osc).max() <= 1e-5
            for n in np.argwhere(idx[:, 0] != oidx[:, 0]).flatten():
                assert abs(full[n, idx[n, 0]] - full[n, oidx[n, 0]]) <= 2e-7, n
            assert cnt >= 1000, f"only {cnt} rows took the exact rescan"
        else:
            assert np.array_equal(idx, ref[0]) and np.array_equal(sc, ref[1]), f"run {rep} differs"


@pytest.mark.parametrize("N,P", [(100_000, 1000), (70_001, 333), (109_215, 1000), (131_000, 197), (98_303, 2049)])
def test_affinity_block_plan_equals_the_range_plan(engine, N, P):
    """The coarse pass's two plans - the BLOCK plan (every workgroup sweeps all stages once; blocks of 32 segments dealt to waves, leftover blocks swept in
    parts; 1, 2 or 3 records per whole sweep; the default where its cost model takes it, round 5) and the RANGE plan (`affinity_variant` 7) - on the same inputs: the exact pass certifies or rescans every row, so indices and
    scores must agree bit for bit whatever the decomposition - near-duplicate profiles, exact ties and a partial last tile included - and both
    equal the fp64 scan of the GPU's own embeddings (IDs identical, scores within 1e-5)."""
    E, Pm = _unit(N, 192, N + P), _unit(P, 192, 7 * P + 1)
    Pm[1] = Pm[0]                                                   # an exact tie: the lower index must win under both plans
    Pm[5] = oecapa.l2_normalise((Pm[4] + 1e-4 * Pm[6])[None])[0]    # a near-duplicate pair (below fp32 resolution for some rows: compared between the plans only)
    E[:64] = oecapa.l2_normalise(Pm[np.arange(64) % 8] + 0.05 * _unit(64, 192, 3))
    out = {}
    for name, var in (("ranges", 7), ("blocks", 8), ("blocks2", 12), ("blocks3", 13)):      # 12 / 13 (round 5): 2 / 3 records per whole sweep
        engine.set_option("affinity_variant", var)
        try:
            out[name] = _score_gpu(engine, E, Pm, 1)
        finally:
            engine.set_option("affinity_variant", 0)
    a, b = out["ranges"], out["blocks"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for k in ("blocks2", "blocks3"):
        assert np.array_equal(a[0], out[k][0]) and np.array_equal(a[1], out[k][1]), k
    En, Pn = a[3], a[4]
    step = 20_000
    for lo in range(0, N, step):
        oidx, osc = oscoring.affinity_topk(En[lo:lo + step], Pn, 1)
        full = oscoring.affinity(En[lo:lo + step], Pn)
        srt = np.sort(full, axis=1)
        clear = (srt[:, -1] - srt[:, -2]) > 1e-6                   # rows the fp64 oracle itself decides by more than fp32 resolution
        assert np.array_equal(b[0][lo:lo + step][clear], oidx[clear]) and np.abs(b[1][lo:lo + step] - osc).max() <= 1e-5
        ties = ~clear
        assert (np.abs(np.take_along_axis(full, b[0][lo:lo + step].astype(np.int64), 1)[:, 0] - srt[:, -1])[ties] <= 1e-6).all()


def test_affinity_threshold_assignment(engine):
    """Config #2 shape: 1000 segments x 100 profiles, threshold 0.354 (the ABC default)."""
    E, Pm = _unit(1000, 192, 0), _unit(100, 192, 1)
    E[::3] = oecapa.l2_normalise(Pm[np.arange(0, 1000, 3) % 100] + 0.6 * _unit(334, 192, 2))
    idx, sc, _, En, Pn = _score_gpu(engine, E, Pm, 1)
    best, osc = oscoring.assign(En, Pn, 0.354)
    mine = np.where(sc[:, 0] >= np.float32(0.354), idx[:, 0], -1)
    edge = np.abs(osc - 0.354) < 1e-6                        # a score within fp32 rounding of the threshold may land either side
    assert np.array_equal(mine[~edge], best[~edge])
    assert (best >= 0).sum() > 300


# ------------------------------------------------------------------------------------------------
# k6 spectral clustering pieces
from oracle import spectral as ospec  # noqa: E402

CL = sub("cluster")


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("N,kv,row0,rows", [(1000, 16, 0, 1000), (777, 5, 100, 300), (4096, 32, 0, 4096), (130, 1, 0, 130), (5000, 16, 1200, 2600), (33, 3, 0, 33)])
def test_affinity_matvec(engine, N, kv, row0, rows, variant):
    """variant 0 = the persistent row-group kernel (round 3: 64 rows per wave, LDS-DMA ring, parts summed in slot order),
    variant 1 = round 1's kernel; both against the float64 product of the bf16 rows."""
    engine.set_option("matvec_variant", variant)
    try:
        _check_matvec(engine, N, kv, row0, rows)
    finally:
        engine.set_option("matvec_variant", 0)


def test_affinity_matvec_is_reproducible_and_splits_groups(engine):
    """A row group's sweep over j is split over up to three workgroups whose partial tiles are summed in slot order: two runs give the
    same bits, and a shape whose groups are all split (rows = 4 groups on 256 CUs -> 8 workgroups) still matches the reference."""
    N, kv = 6000, 16
    _, Eb, _ = engine.l2norm(dev(_unit(N, 192, 77)))
    X = dev(np.random.default_rng(3).standard_normal((N, kv)).astype(np.float32))
    Y1 = engine.affinity_matvec(Eb, X, 0, 2048).clone()
    Y2 = engine.affinity_matvec(Eb, X, 0, 2048).clone()
    torch.cuda.synchronize()
    assert torch.equal(Y1, Y2)
    _check_matvec(engine, N, kv, 0, 2048)


def _check_matvec(engine, N, kv, row0, rows):
    E = _unit(N, 192, N)
    _, Eb, _ = engine.l2norm(dev(E))
    Ef = Eb.float().cpu().numpy().astype(np.float64)            # the kernel sees the bf16 rows
    rng = np.random.default_rng(kv)
    X = rng.standard_normal((N, kv)).astype(np.float32)
    xs = rng.uniform(0.5, 1.5, N).astype(np.float32)
    Y = engine.affinity_matvec(Eb, dev(X), row0, rows, xscale=dev(xs))
    torch.cuda.synchronize()
    A = np.maximum(Ef[row0:row0 + rows] @ Ef.T, 0.0)
    want = A @ (X.astype(np.float64) * xs[:, None])
    got = Y.cpu().numpy()[row0:row0 + rows]
    # S is rounded to bf16 inside the kernel (2^-9 relative per term, random sign) and X is split hi+lo:
    # error ~ 2^-9 * |A| |X| / sqrt(N_eff); tolerance 1.5e-3 of the row's absolute sum
    scale = (A @ np.abs(X.astype(np.float64) * xs[:, None])) + 1e-6
    assert (np.abs(got - want) <= 1.5e-3 * scale).all(), float((np.abs(got - want) / scale).max())
    if rows < N:
        assert not Y.cpu().numpy()[:row0].any() and not Y.cpu().numpy()[row0 + rows:].any(), "rows outside the block must stay untouched"


def test_thin_helpers(engine):
    rng = np.random.default_rng(0)
    X = rng.standard_normal((1000, 16)).astype(np.float32)
    Yv = rng.standard_normal((1000, 16)).astype(np.float32)
    G = engine.rows_gram(dev(X), dev(Yv)).cpu().numpy()
    assert np.allclose(G, X.astype(np.float64).T @ Yv.astype(np.float64), rtol=1e-5, atol=1e-4)
    R = rng.standard_normal((16, 16)).astype(np.float32)
    sc = rng.uniform(0.5, 2, 1000).astype(np.float32)
    Z = engine.rows_apply(dev(X), dev(R), dev(sc)).cpu().numpy()
    assert np.allclose(Z, (X.astype(np.float64) @ R) * sc[:, None], rtol=1e-5, atol=1e-5)
    U = engine.rows_unit(dev(X)).cpu().numpy()
    assert np.allclose(np.linalg.norm(U, axis=1), 1, atol=1e-6)
    C = X[:5].copy()
    lab, d2, ps, pc = engine.kmeans_assign(dev(X), dev(C))
    dist = ((X[:, None, :].astype(np.float64) - C[None]) ** 2).sum(-1)
    assert np.array_equal(lab.cpu().numpy(), dist.argmin(1))
    assert np.allclose(d2.cpu().numpy(), dist.min(1), rtol=1e-5, atol=1e-6)
    sums = ps.cpu().numpy().astype(np.float64).sum(0)
    for q in range(5):
        assert np.allclose(sums[q], X[dist.argmin(1) == q].astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)
    assert np.array_equal(pc.cpu().numpy().sum(0), np.bincount(dist.argmin(1), minlength=5))


@pytest.mark.parametrize("k", [1, 5, 16, 32])
def test_chol_inverse_on_device(engine, k):
    """CholeskyQR's k x k step without leaving the stream: Rinv = (L^T)^-1, (G + G^T)/2 = L L^T, float64 inside."""
    g = torch.Generator().manual_seed(k)
    Y = torch.randn(500, k, generator=g, dtype=torch.float64)
    G = (Y.T @ Y).float()
    G[0, k - 1] += 1e-3                                       # a slightly asymmetric input is symmetrised, as the host form did
    Rinv = engine.chol_inverse(dev(G)).cpu().double()
    Gs = 0.5 * (G.double() + G.double().T)
    L = np.linalg.cholesky(Gs.numpy())
    want = np.linalg.inv(L.T)
    assert np.abs(Rinv.numpy() - want).max() <= 1e-6 * np.abs(want).max()
    assert np.abs(np.tril(Rinv.numpy(), -1)).max() == 0.0    # upper triangular
    Q = Y @ Rinv
    assert float((Q.T @ Q - torch.eye(k, dtype=torch.float64)).abs().max()) < 1e-3


def test_chol_inverse_flags_a_rank_deficient_gram_matrix(engine):
    """the device CholeskyQR must not swallow a non-SPD Gram matrix.  The flag is sticky (set, never cleared)."""
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    g = torch.Generator().manual_seed(1)
    Y = torch.randn(400, 6, generator=g, dtype=torch.float64)
    engine.chol_inverse(dev((Y.T @ Y).float()), flag)
    assert int(flag.item()) == 0
    Y[:, 5] = Y[:, 0] + Y[:, 1]                               # rank 5
    engine.chol_inverse(dev((Y.T @ Y).float()), flag)
    assert int(flag.item()) == 1
    engine.chol_inverse(dev(torch.eye(6)), flag)              # a good matrix afterwards does not clear it
    assert int(flag.item()) == 1
    flag.zero_()
    Gn = torch.eye(6); Gn[2, 2] = float("nan")
    engine.chol_inverse(dev(Gn), flag)
    assert int(flag.item()) == 1


def test_spectral_cluster_raises_when_k_exceeds_the_rank(engine):
    """Duplicate embeddings, k greater than the number of distinct rows: the host-side CholeskyQR of round 1 raised LinAlgError; the
    sync-free device form must too (checked once, at the Ritz step's host synchronisation), instead of returning plausible labels."""
    base = _unit(3, 192, 5)
    E = np.repeat(base, 200, axis=0)
    En, Eb, _ = engine.l2norm(dev(E))
    with pytest.raises(np.linalg.LinAlgError):
        CL.spectral_cluster(engine, En, Eb, 600, 6, n_iter=5, n_kmeans=5, seed=0)
    res = CL.spectral_cluster(engine, En, Eb, 600, 3, n_iter=8, n_kmeans=5, seed=0)     # k = rank works
    assert np.array_equal(res.labels, np.repeat(np.arange(3), 200))


def test_spectral_cluster_survives_over_clustering_on_tight_data(engine, monkeypatch):
    """k just above the true number of speakers on low-noise embeddings puts lambda_k / lambda_1 near the relative-pivot rule of
    sdk_chol_inverse (1e-6 of the diagonal: cond(Y) ~ 1e3), where plain CholeskyQR2 used to end the whole shard in LinAlgError after all
    embeddings had been computed.  spectral_cluster now retries once with shifted CholeskyQR; exact duplicates (a true rank loss: the test
    above) still raise.

    Compared with oracle/spectral.py on the SAME bf16 rows, k, iteration counts and seed (VERDICT r4 next #1a).  With k above the speaker
    count c the surplus eigenvalues sit at the noise floor (oracle: 1e-5 / 0, 0 / 4.9e-4, -1.9e-4 for the three cases), so the surplus Ritz
    DIRECTIONS are not determined by the data - float64 QR and fp32 CholeskyQR legitimately pick different ones - and a label-for-label match
    is not a property of the algorithm.  What is determined, and asserted:
      * the c leading eigenvalues equal the oracle's (they are separated from the floor by > 0.6) and the surplus ones are at the floor in both;
      * k-means itself: the oracle's k-means on the GPU's own rows gives the GPU's labels, row for row;
      * purity wherever the ORACLE is pure: cases (0.02, 4, 5) and (0.005, 3, 5) split speakers and mix none, in the oracle and here.  At
        (0.05, 6, 8) the oracle itself puts 63 rows of FOUR speakers into one of its two surplus clusters (two noise directions: k-means
        trades a split for a merge), so the two-speaker cluster the GPU path showed in round 4 is the algorithm's behaviour, not the retry's:
        there the assertion is that at least c clusters are pure and the mixed clusters are small (no worse than twice the oracle's mixed rows)."""
    shifted = []
    real = engine.set_option
    monkeypatch.setattr(engine, "set_option", lambda n, v: (shifted.append((n, v)), real(n, v))[1])

    def mixed_rows(lab, truth):
        return sum(int((lab == l).sum()) for l in np.unique(lab) if len(np.unique(truth[lab == l])) > 1)

    for noise, c, k in ((0.02, 4, 5), (0.005, 3, 5), (0.05, 6, 8)):
        E, truth = ospec.vmf_mixture(1500, 192, c, seed=7 + c, noise=noise)
        En, Eb, _ = engine.l2norm(dev(E))
        res = CL.spectral_cluster(engine, En, Eb, 1500, k, n_iter=12, n_kmeans=10, seed=0, keep_rows=True)
        olab, olam = ospec.spectral_cluster(Eb.float().cpu().numpy(), k, n_iter=12, n_kmeans=10, seed=0)
        assert np.isfinite(res.eigenvalues).all() and len(np.unique(res.labels)) <= k
        assert np.abs(res.eigenvalues[:c] - olam[:c]).max() < 2e-3, (noise, c, k, res.eigenvalues, olam)   # bf16 tile rounding inside A V, as in test_spectral_cluster_matches_oracle
        assert np.abs(res.eigenvalues[c:]).max() < 5e-3 and np.abs(olam[c:]).max() < 5e-3, (res.eigenvalues, olam)
        klab, _ = ospec.kmeans_maximin(res.rows.astype(np.float64), k, 10)
        assert np.array_equal(ospec.canonical_labels(klab), res.labels), (noise, c, k)
        o_mixed, g_mixed = mixed_rows(olab, truth), mixed_rows(res.labels, truth)
        print(f"\n(noise {noise}, c {c}, k {k}): retried {res.retried}; rows in mixed clusters: oracle {o_mixed}, gpu {g_mixed}; "
              f"cluster sizes oracle {np.bincount(olab).tolist()} gpu {np.bincount(res.labels).tolist()}")
        if o_mixed == 0:
            assert g_mixed == 0, (noise, c, k, g_mixed)
        else:
            pure = sum(1 for l in np.unique(res.labels) if len(np.unique(truth[res.labels == l])) == 1)
            assert pure >= c and g_mixed <= 2 * o_mixed, (noise, c, k, pure, g_mixed, o_mixed)
    print("shifted CholeskyQR retries:", sum(1 for n, v in shifted if n == "chol_shift_ppb" and v > 0))
    assert all(v in (0, 10_000) for n, v in shifted if n == "chol_shift_ppb")


@pytest.mark.parametrize("N,k", [(2000, 6), (5000, 16)])
def test_laplacian_topk_c_driver_matches_the_host_driver(engine, N, k):
    """sdk_laplacian_topk (the k6 driver a non-Python host binds) runs cluster.spectral_cluster's subspace iteration in ONE C call
    with a device-side Ritz step.  Same seeded start block -> the same eigenvalues (1e-5), the same invariant subspace (projector difference
    ~1e-4), and - fed to the exported k-means primitives exactly as cluster.py does - the same integer labels as the oracle."""
    E, truth = ospec.vmf_mixture(N, 192, k, seed=N + k, noise=0.6)
    En, Eb, _ = engine.l2norm(dev(E))
    res = CL.spectral_cluster(engine, En, Eb, N, k, n_iter=25, n_kmeans=20, seed=0)
    V0 = dev(np.random.default_rng(0).standard_normal((N, k)).astype(np.float32))
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    U, lam = engine.laplacian_topk(Eb, V0, 25, flag=flag)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    lam = lam.cpu().numpy()
    assert np.all(np.diff(lam) <= 1e-7) and np.abs(lam - res.eigenvalues).max() < 1e-5, (lam, res.eigenvalues)
    Ud = U.double().cpu()
    assert float((Ud.T @ Ud - torch.eye(k, dtype=torch.float64)).abs().max()) < 1e-4          # orthonormal Ritz vectors
    # k-means on the driver's rows, through cluster.py's own (exported-primitive) routine: labels of the oracle
    R = engine.rows_unit(U)
    lab = CL.canonical_labels(CL._kmeans(engine, CL._Comm(None), R, 0, N, k, 20).cpu().numpy())
    olab, _ = ospec.spectral_cluster(Eb.float().cpu().numpy(), k, n_iter=25, n_kmeans=20, seed=0)
    assert np.array_equal(lab, olab) and np.array_equal(lab, res.labels)
    # k greater than the rank: the sticky flag, no exception, no hang
    base = _unit(3, 192, 5)
    _, Eb3, _ = engine.l2norm(dev(np.repeat(base, 100, axis=0)))
    flag.zero_()
    engine.laplacian_topk(Eb3, dev(np.random.default_rng(1).standard_normal((300, 6)).astype(np.float32)), 4, flag=flag)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("N,k", [(2000, 6), (5000, 16)])
def test_spectral_cluster_matches_oracle(engine, N, k):
    """Config #5 scaled down: mixture-of-clusters embeddings, GPU pipeline vs the CPU oracle run on the
    same bf16-rounded rows.  Integer labels must be identical (canonical order of first appearance)."""
    E, truth = ospec.vmf_mixture(N, 192, k, seed=N + k, noise=0.6)
    En, Eb, _ = engine.l2norm(dev(E))
    res = CL.spectral_cluster(engine, En, Eb, N, k, n_iter=25, n_kmeans=20, seed=0)
    Ef = Eb.float().cpu().numpy()
    olab, olam = ospec.spectral_cluster(Ef, k, n_iter=25, n_kmeans=20, seed=0)
    assert np.array_equal(res.labels, olab), f"ARI vs oracle {ospec.adjusted_rand_index(res.labels, olab)}"
    assert ospec.adjusted_rand_index(res.labels, truth) == 1.0
    # eigenvalues of S: bf16 tile rounding inside A V -> 2e-3 absolute (values in [0, 1])
    assert np.abs(res.eigenvalues - olam).max() < 2e-3, (res.eigenvalues, olam)


@pytest.mark.parametrize("M,T,N,Cin,mode", [(2010, 201, 1024, 128, 1), (1608, 201, 512, 64, 2), (1280, 128, 256, 64, 2), (3000, 1500, 256, 64, 1)])
def test_conv_gemm_fused_column_stats(engine, M, T, N, Cin, mode):
    """Per-segment column statistics produced by the GEMM epilogue (SE squeeze / ASP context) equal the
    statistics of the STORED bf16 output."""
    g = torch.Generator().manual_seed(M + mode)
    A = bf16_round(torch.randn(M, Cin, generator=g))
    Wt = bf16_round(torch.randn(N, Cin, generator=g) * 0.2)
    bias = torch.randn(N, generator=g)
    C, _, _, st = engine.conv_gemm(dev(A, torch.bfloat16), dev(Wt, torch.bfloat16), N, Cin, T=T, bias=dev(bias), relu=True, stats_mode=mode)
    torch.cuda.synchronize()
    z = C.float().cpu().double().reshape(M // T, T, N)
    mean = z.mean(1)
    got = st.cpu().double()
    # fp32 partial sums in a fixed order vs float64: rtol 1e-5 (|values| ~ 1)
    assert torch.allclose(got[:, :N], mean, rtol=1e-5, atol=1e-5), float((got[:, :N] - mean).abs().max())
    if mode == 2:
        sd = ((z - mean[:, None]) ** 2).mean(1).clamp_min(1e-12).sqrt()
        assert torch.allclose(got[:, N:], sd, rtol=2e-4, atol=2e-5), float((got[:, N:] - sd).abs().max())


@pytest.mark.parametrize("M,T,N,Cin,taps,dil,mode,pack", [(40200, 201, 1024, 128, 1, 1, 1, 0), (20100, 201, 1024, 64, 1, 1, 2, 0), (66000, 200, 1024, 192, 3, 2, 2, 0),
                                                          (201000, 201, 1024, 256, 1, 1, 2, 0), (201000, 201, 1024, 80, 5, 1, 0, 80), (50250, 201, 1024, 64, 1, 1, 1, 0),
                                                          (40200, 201, 2048, 64, 1, 1, 0, 0), (33600, 200, 2048, 64, 1, 1, 1, 0), (99200, 200, 3072, 64, 1, 1, 2, 0)])
def test_conv_gemm_half_tile_tail_is_bit_identical(engine, M, T, N, Cin, taps, dil, mode, pack):
    """Round 5: where the whole rounds of the 256^2 kernel leave a last round that is mostly idle (the forward's K = 1024
    layers: 3144 tiles = 12 x 256 + 72), the left-over rows are computed as 128 x 256 half tiles, at most one per workgroup, instead of a 13th
    round of whole tiles.  Same K order per output element, same rows in the same order per column-statistics partial: output and statistics
    must equal the whole-tile schedule's (gemm_variant 8194 = tune bit 9: half tiles off) bit for bit - random operands, ReLU + BN epilogue,
    conv taps, blk0's packed taps, an edge half tile that is partly / wholly outside the matrix, a shape where the rule does not apply
    (N = 2048, 60 half row blocks left over: they do not fit), and wider layers where it does (N = 2048 with 8 half row blocks left: every XCD
    takes one block's eight column tiles; N = 3072 with 8: one block's twelve column tiles per XCD)."""
    g = torch.Generator().manual_seed(M + N + taps)
    lda = Cin
    A = dev(bf16_round(torch.randn(M, lda, generator=g)), torch.bfloat16)
    K = ((taps * pack + 63) // 64) * 64 if pack else taps * Cin
    Wt = bf16_round(torch.randn(N, K, generator=g) * 0.1)
    if pack:
        Wt[:, taps * pack:] = 0
    Wt = dev(Wt, torch.bfloat16)
    bias, sc, sh = dev(torch.randn(N, generator=g)), dev(torch.rand(N, generator=g) + 0.5), dev(torch.randn(N, generator=g))
    outs = {}
    try:
        for v in (2, 8194):
            engine.lib.sdk_set_gemm_variant(v)
            outs[v] = engine.conv_gemm(A, Wt, N, Cin, taps=taps, dil=dil, T=T, bias=bias, scale=sc, shift=sh, relu=True, stats_mode=mode, tap_pack=pack)
            torch.cuda.synchronize()
    finally:
        engine.lib.sdk_set_gemm_variant(2)
    a, b = outs[2], outs[8194]
    assert torch.equal(a[0].view(torch.int16), b[0].view(torch.int16))
    if mode:
        assert torch.equal(a[3], b[3])
    # and against an independent result on a sample of rows from the tail region (fp32 matmul of the same bf16 operands; taps == 1 only)
    if taps == 1:
        rows = torch.arange(M - 300, M, device="cuda")
        want = torch.relu(A[rows].float() @ Wt.float().T + bias) * sc + sh
        assert torch.allclose(a[0][rows].float(), want, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("M,T,N,Cin,mode", [(40200, 201, 1024, 128, 1), (40200, 201, 2048, 64, 0), (20100, 201, 1024, 64, 2), (66000, 200, 1024, 64, 0)])
def test_conv_gemm_unit_order_covers_every_tile(engine, M, T, N, Cin, mode):
    """The 256^2 kernel's tile schedule (default: 8 x 4-tile units per XCD round; 1026: round 1's order) must write every tile exactly
    once: integer operands, so the output is exact and can be compared with a float32 matmul as well as between schedules (full
    unit rounds, left-over units, a partial m-group, an edge tile)."""
    g = torch.Generator().manual_seed(M + N)
    A = dev(torch.randint(-2, 3, (M, Cin), generator=g).float(), torch.bfloat16)
    Wt = dev(torch.randint(-1, 2, (N, Cin), generator=g).float(), torch.bfloat16)
    want = (A.float() @ Wt.float().T).to(torch.bfloat16)
    outs = {}
    try:
        for v in (2, 1026):
            engine.lib.sdk_set_gemm_variant(v)
            o = engine.conv_gemm(A, Wt, N, Cin, T=T, stats_mode=mode)
            torch.cuda.synchronize()
            assert torch.equal(o[0], want), f"variant {v}"
            outs.setdefault(v, o)
    finally:
        engine.lib.sdk_set_gemm_variant(2)
    if mode:
        assert torch.equal(outs[2][3], outs[1026][3])


def test_conv_gemm_a2_addend(engine):
    """A2: the GEMM consumes bf16(A + A2) (Res2Net running sum formed on the way into LDS)."""
    M, T, N, Cin = 603, 201, 128, 128
    g = torch.Generator().manual_seed(9)
    A = torch.randint(-3, 4, (M, 256), generator=g).float()
    A2 = torch.randint(-2, 3, (M, 384), generator=g).float()
    Wt = torch.randint(-2, 3, (N, 3 * Cin), generator=g).float()
    _, C32, _ = engine.conv_gemm(dev(A, torch.bfloat16)[:, 64:192], dev(Wt, torch.bfloat16), N, Cin, taps=3, dil=3, T=T, out_bf16=False,
                                 out_f32=True, A2=dev(A2, torch.bfloat16)[:, 128:256])
    want = _conv_ref(A[:, 64:192] + A2[:, 128:256], Wt, Cin, 3, 3, T)
    torch.cuda.synchronize()
    assert torch.equal(C32.cpu(), want)
    # non-integer data: the sum is rounded to bf16 before the MFMA, exactly like a stored S = bf16(A + A2)
    Af, A2f = bf16_round(torch.randn(M, Cin, generator=g)), bf16_round(torch.randn(M, Cin, generator=g))
    Wf = bf16_round(torch.randn(N, 3 * Cin, generator=g) * 0.1)
    _, C32, _ = engine.conv_gemm(dev(Af, torch.bfloat16), dev(Wf, torch.bfloat16), N, Cin, taps=3, dil=2, T=T, out_bf16=False, out_f32=True,
                                 A2=dev(A2f, torch.bfloat16))
    want = _conv_ref(bf16_round(Af + A2f), Wf, Cin, 3, 2, T)
    torch.cuda.synchronize()
    assert torch.allclose(C32.cpu(), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("B,T", [(5, 201), (3, 100), (2, 208), (1, 9)])
def test_res2net_chain_fusion_is_bit_identical(engine, B, T):
    """The per-segment fused Res2Net chain (one launch, tile resident in LDS) must reproduce the seven
    separate conv_gemm launches bit for bit (same k order, same rounding points)."""
    feats = _feats(B, T, 31)
    f = torch.zeros(B * T, WP.N_MELS_PADDED, dtype=torch.bfloat16)
    f[:, :80] = feats.reshape(-1, 80).to(torch.bfloat16)
    f = f.cuda()
    engine.set_option("res2net_chain_fusion", 0)
    ref = engine.ecapa_forward(f, B, T).cpu()
    engine.set_option("res2net_chain_fusion", 1)
    fused = engine.ecapa_forward(f, B, T).cpu()
    assert torch.equal(ref, fused), float((ref - fused).abs().max())
    # ... and so must the chain when it reads the [128][384] weight matrices instead of their fragment-ordered copies (EL_CHAINPACK)
    try:
        engine.set_option("res2net_packed_weights", 0)
        plain = engine.ecapa_forward(f, B, T).cpu()
    finally:
        engine.set_option("res2net_packed_weights", 1)
    assert torch.equal(ref, plain), float((ref - plain).abs().max())
    # ... and the 8-wave one-segment-per-CU form of the chain (the default for T > 112 is two 4-wave workgroups per CU, one image in place)
    try:
        engine.set_option("res2net_two_per_cu", 0)
        plain = engine.ecapa_forward(f, B, T).cpu()
    finally:
        engine.set_option("res2net_two_per_cu", 1)
    assert torch.equal(ref, plain), float((ref - plain).abs().max())
    # ... and the per-segment ASP kernel with the row-major logit weights instead of their fragment-ordered copy (EL_ASP_W2PACK)
    try:
        engine.set_option("asp_packed_weights", 0)
        plain = engine.ecapa_forward(f, B, T).cpu()
    finally:
        engine.set_option("asp_packed_weights", 1)
    assert torch.equal(ref, plain), float((ref - plain).abs().max())
