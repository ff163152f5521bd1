"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the
header declares, the host binding matches it, and - without a GPU - the product path fails loudly
instead of falling back to anything."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT, sub

LIB = sub("_lib")
HEADER = ROOT / "include" / "sdk_hip.h"


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(sdk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = LIB.load_library()
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sdk_hip.h but not exported by libsdk_hip.so"
    assert sorted(LIB.SIGNATURES) == names, "host binding (_lib.SIGNATURES) out of sync with the header"
    assert lib.sdk_abi_version() == 4


def test_struct_layouts_match_header():
    assert C.sizeof(LIB.EcapaDesc) == 16 * 4 + 256 * 8            # ABI 2: + precision, reserved0
    assert LIB.EcapaDesc.off.offset == 64 and LIB.EcapaDesc.precision.offset == 56 and LIB.EcapaDesc.blk0_tap_pack.offset == 60
    assert C.sizeof(LIB.ConvGemmHpArgs) == 20 * 8 + 7 * 4 + 4     # 20 pointer/int64 slots, 6 ints + flags, tail padding
    assert C.sizeof(LIB.ProfileReport) == 24 * 4 + 3 * 24 * 8
    assert C.sizeof(LIB.ConvGemmArgs) == 16 * 8 + 8 * 4 + 8 + 16 + 8  # 16 pointer/int64 slots, 6 ints + flags + stats_mode, stats_part, A2 + lda2, tap_pack + reserved


def test_host_only_entry_points_work_without_gpu():
    lib = LIB.load_library()
    n = lib.sdk_fbank_tables_bytes()
    buf = np.zeros(n, dtype=np.uint8)
    assert lib.sdk_fbank_tables_fill(buf.ctypes.data, n) == 0
    WP = sub("weights_pack")
    raw = buf[:7 * 2 * 13 * 2 * 64 * 8 * 2].view(np.uint16).reshape(7, 2, 13, 2, 64, 8)
    tab = WP.bf16_bits_to_f32(raw).reshape(raw.shape)
    val = tab[:, :, :, 0] + tab[:, :, :, 1]                 # hi + lo
    # folded DFT table (n = 0..200), B-fragment order: [bin block][cos|sin][k-step][hi|lo][lane][8];
    # lane l, element j -> n = 16 ks + 8 (l >> 5) + j, bin 32 w + (l & 31)
    assert abs(val[0, 0, 0, 0, 0] - 0.04) < 1e-6 and val[0, 1, 0, 0, 0] == 0          # n = 0: 1/2 w[0] cos 0; sin row zero
    n, f = 16 * 3 + 8 + 5, 32 * 2 + 7
    want_c = (0.54 - 0.46 * np.cos(2 * np.pi * n / 400)) * np.cos(2 * np.pi * n * f / 400)
    want_s = -(0.54 - 0.46 * np.cos(2 * np.pi * n / 400)) * np.sin(2 * np.pi * n * f / 400)
    assert abs(val[2, 0, 3, 32 + 7, 5] - want_c) < 2e-5 and abs(val[2, 1, 3, 32 + 7, 5] - want_s) < 2e-5   # hi+lo: ~16 bits
    assert not val[:, :, 12, 32:, 1:].any()                  # n > 200 (padding) is zero
    tab = raw                                                # byte size used below
    # mel table must equal the oracle's filterbank
    from oracle import fbank as ofb
    ints = buf[tab.nbytes:tab.nbytes + 3 * 80 * 4].view(np.int32).reshape(3, 80)
    melw = buf[tab.nbytes + 3 * 80 * 4:].view(np.float32)
    Wm = ofb.mel_matrix().astype(np.float32)
    for m in range(80):
        st, ln, of = ints[0, m], ints[1, m], ints[2, m]
        assert np.array_equal(np.nonzero(Wm[:, m])[0], np.arange(st, st + ln))
        assert np.allclose(melw[of:of + ln], Wm[st:st + ln, m], rtol=0, atol=1e-7)
    # precise mode: the same matrix split into fp16 hi + lo (22 bits), same fragment order, behind the mel table
    t16 = buf[tab.nbytes + 3 * 80 * 4 + 512 * 4:][:tab.nbytes].view(np.float16).reshape(7, 2, 13, 2, 64, 8).astype(np.float64)
    val16 = t16[:, :, :, 0] + t16[:, :, :, 1]
    assert abs(val16[2, 0, 3, 32 + 7, 5] - want_c) < 3e-7 and abs(val16[2, 1, 3, 32 + 7, 5] - want_s) < 3e-7
    assert np.abs(val16 - val.astype(np.float64)).max() < 2e-5 and not val16[:, :, 12, 32:, 1:].any()
    assert lib.sdk_fbank_tables_fill(buf.ctypes.data, 10) != 0 and b"too small" in lib.sdk_last_error()
    assert lib.sdk_fbank_workspace_bytes(1000, 32000) == 1000 * 201 * 80 * 4
    assert lib.sdk_affinity_workspace_bytes(100000, 1000) > 100000 * 56


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(LIB.SdkError, match="no HIP device|no CPU fallback"):
        LIB.get_ctx(0)
    be = sub("backend").Backend()
    assert be.name == "mi355x" and be.requires_api_key is False and be.embedding_dim == 192
    with pytest.raises(LIB.SdkError):
        be.engine()


def test_missing_library_is_reported(monkeypatch, tmp_path):
    monkeypatch.setattr(LIB, "_lib", None)
    monkeypatch.setattr(LIB, "LIB_PATH", tmp_path / "libsdk_hip.so")
    with pytest.raises(LIB.SdkError, match="has not been built"):
        LIB.load_library()


def test_weight_packing_layout():
    W, WP = sub("weights"), sub("weights_pack")
    cfg = W.EcapaConfig(channels=256, mfa_channels=768, res2net_scale=2, se_channels=64, attn_channels=128)
    w = W.synthetic_weights(5, cfg)
    blob, f = WP.pack_weights(w, cfg)
    assert f["n_blocks"] == 3 and f["dilation"] == [2, 3, 4, 0] and f["n_mels_padded"] == 128
    off = f["off"]
    used = [o for o in off if o >= 0]
    assert all(o % 256 == 0 for o in used) and len(used) == len(set(used))
    # blk0 weight, taps packed along K (round 3): [256][448] = tap-major 5 x 80 channels + 48 zero columns, values = bf16(w)
    assert f["blk0_tap_pack"] == 80
    k0 = blob[off[0]:off[0] + 256 * 448 * 2].view(np.uint16).reshape(256, 448)
    assert not k0[:, 400:].any()
    want = WP.f32_to_bf16_bits(np.transpose(w["blk0.conv.w"], (0, 2, 1)))
    assert np.array_equal(k0[:, :400].reshape(256, 5, 80), want)
    # folded BN of blk0
    s, sh = W.bn_affine(w, "blk0.bn")
    assert np.array_equal(blob[off[2]:off[2] + 1024].view(np.float32), s)
    # SE weights are stored transposed
    b1 = WP.block_base(1)
    w1t = blob[off[b1 + WP.EL_SE_W1T]:off[b1 + WP.EL_SE_W1T] + 256 * 64 * 4].view(np.float32).reshape(256, 64)
    assert np.array_equal(w1t, w["blk1.se.conv1.w"][:, :, 0].T)
    assert W.DEFAULT_CONFIG.param_count() == 20_767_552 and W.DEFAULT_CONFIG.macs_per_frame() == 18_743_296


def test_fragment_ordered_weight_copies():
    """The optional blob slots that hold weights a second time in MFMA fragment order (ecapa_layout.h EL_CHAINPACK, EL_ASP_W2PACK):
    every 16-byte piece must be exactly the piece of the row-major matrix the kernels' strided loads would fetch."""
    W, WP = sub("weights"), sub("weights_pack")
    rng = np.random.default_rng(3)
    wk = rng.integers(0, 65536, (128, 384), dtype=np.uint16)
    o = WP.chain_fragment_order(wk)
    assert o.shape == (3, 4, 2, 4, 64, 8)
    for tap, wq, h, ks, lane in ((0, 0, 0, 0, 0), (2, 3, 1, 2, 37), (1, 2, 0, 3, 63), (2, 0, 1, 1, 16)):
        row, k0 = 32 * wq + 16 * h + (lane & 15), 128 * tap + 32 * ks + 8 * (lane >> 4)
        assert np.array_equal(o[tap, wq, h, ks, lane], wk[row, k0:k0 + 8])
    assert np.array_equal(np.sort(o.reshape(-1)), np.sort(wk.reshape(-1)))            # a permutation of the matrix
    w2 = rng.integers(0, 65536, (3072, 128), dtype=np.uint16)
    q = WP.asp_w2_fragment_order(w2)
    assert q.shape == (96, 8, 64, 8)
    for blk, ks, lane in ((0, 0, 0), (95, 7, 63), (17, 3, 40), (50, 5, 31)):
        assert np.array_equal(q[blk, ks, lane], w2[32 * blk + (lane & 31), 16 * ks + 8 * (lane >> 5):16 * ks + 8 * (lane >> 5) + 8])
    assert np.array_equal(np.sort(q.reshape(-1)), np.sort(w2.reshape(-1)))
    # ... and the packer fills the slots for the full-size configuration
    cfg = W.DEFAULT_CONFIG
    blob, f = WP.pack_weights(W.synthetic_weights(1, cfg), cfg)
    off = f["off"]
    for i in range(1, 4):
        for j in range(7):
            a = off[WP.chainpack_slot(i, j)]
            assert a >= 0
            plain = blob[off[WP.block_base(i) + WP.res2net_slot(j) + WP.EL_W]:][:128 * 384 * 2].view(np.uint16).reshape(128, 384)
            assert np.array_equal(blob[a:a + 98304].view(np.uint16).reshape(3, 4, 2, 4, 64, 8), WP.chain_fragment_order(plain))
    t = WP.tail_base(3)
    plain = blob[off[t + WP.EL_ASP_W2]:][:3072 * 128 * 2].view(np.uint16).reshape(3072, 128)
    assert np.array_equal(blob[off[t + WP.EL_ASP_W2PACK]:][:3072 * 128 * 2].view(np.uint16).reshape(96, 8, 64, 8), WP.asp_w2_fragment_order(plain))


def test_four_block_layout_has_no_slot_collisions():
    """with n_blocks == 4 the tail occupies slots 164..179; the optional fragment-ordered copies must not land inside it (they
    did at 160 + ...: the chain then read MFA / ASP weights as conv weights).  The packer refuses a slot written twice; here every slot of a
    4-dilation model (small channels: the slot arithmetic does not depend on them) and of the full-size 3-block model is unique and in range."""
    W, WP = sub("weights"), sub("weights_pack")
    cfg4 = W.EcapaConfig(channels=1024, mfa_channels=4096, dilations=(2, 3, 4, 5))
    # slot arithmetic only (packing 4 blocks of C = 1024 weights costs seconds, not needed for the index check)
    used = set(range(4))
    for i in range(1, 5):
        b = WP.block_base(i)
        used |= {b + k for k in range(40)}
    t = WP.tail_base(4)
    tail = {t + k for k in range(16)}
    assert not (used & tail) and max(tail) == 179
    chain = {WP.chainpack_slot(i, j) for i in range(1, 5) for j in range(7)}
    assert len(chain) == 28 and not (chain & (used | tail)) and max(chain) < 256 and min(chain) > max(tail)
    # the real packer on a small 4-block model: every offset distinct, no assertion from the double-write guard
    small = W.EcapaConfig(channels=1024, mfa_channels=4096, dilations=(2, 3, 4, 5))
    blob, f = WP.pack_weights(W.synthetic_weights(2, small), small)
    offs = [o for o in f["off"] if o >= 0]
    assert len(offs) == len(set(offs)) and f["n_blocks"] == 4
    for i in range(1, 5):
        for j in range(7):
            a = f["off"][WP.chainpack_slot(i, j)]
            plain = blob[f["off"][WP.block_base(i) + WP.res2net_slot(j) + WP.EL_W]:][:128 * 384 * 2].view(np.uint16).reshape(128, 384)
            assert np.array_equal(blob[a:a + 98304].view(np.uint16).reshape(3, 4, 2, 4, 64, 8), WP.chain_fragment_order(plain))
    # the header's macro agrees with the packer
    hdr = (ROOT / "speaker-diarization-toolkit_amd" / "csrc" / "ecapa_layout.h").read_text()
    assert "#define EL_CHAINPACK(i, j) (200 + ((i) - 1) * 8 + (j))" in hdr


def test_precise_mode_weight_planes():
    """weights_pack.hp_weight_planes (csrc/hp.hip's weight slot): 256-byte header with 2^-s, fp16 hi / lo planes of 2^s W; the pair
    reproduces W to 22 bits whatever the layer's scale, and the precise blob keeps every fp32 slot of the default blob bit for bit."""
    W, WP = sub("weights"), sub("weights_pack")
    rng = np.random.default_rng(4)
    for scale in (1e-5, 3e-2, 1.0, 40.0):
        w = (rng.standard_normal((64, 96)) * scale).astype(np.float32)
        slot = WP.hp_weight_planes(w)
        assert slot.dtype == np.uint16 and slot.size == WP.HP_WHDR + 2 * w.size
        inv = float(slot[:2].view(np.float32)[0])
        assert inv > 0 and np.log2(inv) == np.round(np.log2(inv))                      # an exact power of two
        hi = slot[WP.HP_WHDR:WP.HP_WHDR + w.size].view(np.float16).astype(np.float64)
        assert 4096 <= np.abs(hi).max() <= 8192 + 4                                    # max |2^s W| in [2^12, 2^13]
        back = WP.hp_planes_to_f64(slot, 64, 96)
        assert np.abs(back - w).max() <= 2.0 ** -21 * np.abs(w).max()
    cfg = W.EcapaConfig(channels=256, mfa_channels=768)
    wts = W.synthetic_weights(5, cfg)
    b0, f0 = WP.pack_weights(wts, cfg, precision=0)
    b1, f1 = WP.pack_weights(wts, cfg, precision=1)
    assert f0["precision"] == 0 and f1["precision"] == 1 and f1["n_mels_padded"] == 96 and f0["n_mels_padded"] == 128
    gemm_w = {0} | {WP.block_base(i) + s for i in (1, 2, 3) for s in [WP.EL_TDNN1, WP.EL_TDNN2] + [WP.res2net_slot(j) for j in range(7)]}
    t = WP.tail_base(3)
    gemm_w |= {t + WP.EL_MFA, t + WP.EL_ASP_WH, t + WP.EL_ASP_W2}
    for slot in range(256):
        o0, o1 = f0["off"][slot], f1["off"][slot]
        if slot in gemm_w:
            assert o0 >= 0 and o1 >= 0
        elif slot >= 200 or slot == t + WP.EL_ASP_W2PACK:
            assert o1 == -1                                                           # no fragment-ordered copies in the precise blob
        elif o0 >= 0:
            nxt0 = min([x for x in f0["off"] if x > o0] + [b0.size])
            nxt1 = min([x for x in f1["off"] if x > o1] + [b1.size])
            assert nxt0 - o0 == nxt1 - o1 and np.array_equal(b0[o0:nxt0], b1[o1:nxt1]), slot
    # blk0: taps x 96 padded mel channels, channels 80.. are zero in both planes
    w0 = WP.hp_planes_to_f64(b1[f1["off"][0]:].view(np.uint16)[:WP.HP_WHDR + 2 * 256 * 5 * 96], 256, 5 * 96).reshape(256, 5, 96)
    assert not w0[:, :, 80:].any()
    assert np.abs(w0[:, :, :80] - np.transpose(wts["blk0.conv.w"], (0, 2, 1))).max() < 2.0 ** -21 * np.abs(wts["blk0.conv.w"]).max()


def test_matvec_plan_every_group_fits_its_partial_tile_slots():
    """Host side of the config #5 tile kernel's decomposition (csrc/spectral.hip affinity_matvec2_kernel): workgroup i owns the units
    [i*U/G, (i+1)*U/G) of the (row group, j stage) grid and writes one partial Y tile per group it touches, into slot = number of
    earlier workgroups that touch the same group.  Replays that arithmetic: every group's sweep is tiled exactly once, by consecutive
    slots 0..n-1 with n <= the slot count the workspace is sized for - also for a rank's thin row block (12 500 rows of 100 000)."""
    lib = LIB.load_library()
    rng = np.random.default_rng(12)
    shapes = [(100_000, 100_000, 256), (12_500, 100_000, 256), (33, 33, 256), (5000, 5000, 304), (2600, 5000, 64), (1, 1, 256), (512, 100_000, 256)]
    shapes += [(int(r), int(r + rng.integers(0, 50_000)), int(rng.choice([256, 304, 8, 64]))) for r in rng.integers(1, 120_000, 40)]
    for rows, N, cus in shapes:
        out = (C.c_int32 * 4)()
        units = C.c_int64()
        assert lib.sdk_affinity_matvec_plan(rows, N, cus, out, C.byref(units)) == 0
        ng, nst, G, maxp = list(out)
        U = units.value
        assert ng == -(-rows // 512) and nst == -(-(-(-N // 32)) // 2) and U == ng * nst and 1 <= G <= min(cus, U)
        covered = np.zeros(U, np.int32)
        slots = [[] for _ in range(ng)]
        for i in range(G):
            u0, u1 = i * U // G, (i + 1) * U // G
            covered[u0:u1] += 1
            if u0 < u1:
                for b in range(u0 // nst, (u1 - 1) // nst + 1):
                    slots[b].append(i)
        assert (covered == 1).all()
        assert max(len(x) for x in slots) <= maxp, (rows, N, cus)
        assert ng * maxp * 512 * 32 * 4 + (-(-N // 32)) * 4096 <= lib.sdk_affinity_matvec_workspace_bytes(N)
        # the kernel's closed form for "first workgroup whose range reaches into group b"
        for b in range(0, ng, max(1, ng // 7)):
            x = b * nst
            ifirst = ((x + 1) * G + U - 1) // U - 1
            assert slots[b][0] == ifirst, (rows, N, cus, b)


def test_bf16_bits_roundtrip():
    WP = sub("weights_pack")
    import torch
    x = np.random.default_rng(0).standard_normal(10000).astype(np.float32) * 100
    ours = WP.bf16_bits_to_f32(WP.f32_to_bf16_bits(x))
    ref = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    assert np.array_equal(ours, ref)


def test_affinity_plan_every_group_sweep_is_covered_by_at_most_three_workgroups():
    """Host side of the k = 1 affinity decomposition (csrc/affinity_rowcol.hip): workgroup i owns a contiguous range of the (segment group,
    profile stage) units, ranges from sdk_affinity_plan_range - the same functions the kernel uses (they can weigh a group boundary inside a
    range as extra stages; default weight 0: equal unit counts).  Replays the kernel's walk and checks that the ranges tile [0, U) in order with no
    idle workgroup, that every group's sweep is covered exactly once by consecutive record slots 0..n-1 with n <= the slot count, that the slot
    the kernel derives for a workgroup's first portion is the number of earlier workgroups in that group, and that the split is balanced."""
    lib = LIB.load_library()
    rng = np.random.default_rng(11)
    worst = 0
    cases = [(1, 1), (2, 3), (31, 100), (512, 64), (513, 65), (1000, 100), (5000, 1000), (100_000, 1000), (125_000, 10_000),
             (2000, 10_000), (300_000, 32_768)]
    cases += [(int(rng.integers(1, 300_000)), int(rng.integers(1, 32_769))) for _ in range(120)]
    for N, P in cases:
        for cu in (256, 240, 304):
            out = (C.c_int32 * 5)()
            units = C.c_int64()
            assert lib.sdk_affinity_plan(N, P, cu, out, C.byref(units)) == 0
            ngroups, nst, G, segs, slots = list(out)
            U = units.value
            assert ngroups == -(-N // segs) and U == ngroups * nst and 1 <= G <= min(cu, U)
            parts, prev_end, costs = {}, 0, []
            for i in range(G):
                u0, u1, fs = C.c_int64(), C.c_int64(), C.c_int32()
                assert lib.sdk_affinity_plan_range(N, P, cu, i, C.byref(u0), C.byref(u1), C.byref(fs)) == 0
                u0, u1, fs = u0.value, u1.value, fs.value
                assert u0 == prev_end and u1 > u0, (N, P, cu, i, u0, u1)     # ranges tile [0, U) in order, no idle workgroup
                prev_end = u1
                u, first, started = u0, True, 0
                while u < u1:
                    b = u // nst
                    e = min(u1, (b + 1) * nst)
                    slot = fs if first else 0
                    assert slot == len(parts.get(b, [])), (N, P, cu, i, b)     # the kernel's slot = number of earlier workgroups in this group
                    parts.setdefault(b, []).append((slot, u - b * nst, e - b * nst))
                    started += 1
                    first, u = False, e
                costs.append((u1 - u0) + 2 * (started - 1))
            assert prev_end == U and len(parts) == ngroups
            for b, lst in parts.items():
                assert lst[0][1] == 0 and lst[-1][2] == nst and all(x[2] == y[1] for x, y in zip(lst, lst[1:]))
                worst = max(worst, len(lst))
            assert worst <= slots
            if nst >= 8 and G == cu:                                          # long enough sweeps: the cost spread is a stage or two, not a switch
                assert max(costs) - min(costs) <= 6, (N, P, cu, min(costs), max(costs))   # (a range that BEGINS at a group start is charged the penalty too)
    assert worst == 3
    assert lib.sdk_affinity_plan_range(1000, 100, 256, 999, C.byref(C.c_int64()), C.byref(C.c_int64()), C.byref(C.c_int32())) != 0


def test_bias_correction_host_side():
    """weights_pack.bias_corrections / calib_layout / bias_slot (round 3): the slot walk matches sdk_ecapa_forward_calib's, the correction is
    b + (W - bf16(W)) . mu in float64 with every tap of a k3 conv seeing the same means, and the bias slots it patches are the layers' own."""
    W, WP = sub("weights"), sub("weights_pack")
    cfg = W.DEFAULT_CONFIG
    lay, per = WP.calib_layout(cfg)
    assert len(lay) == 3 * 9 + 2 and per == 3 * (2 * 1024 + 7 * 2 * 128 + 2 * 1024) + 2 * 2 * 3072
    assert [n for n, _, _ in lay[:10]] == ["blk1.tdnn1"] + [f"blk1.res2net.{j}" for j in range(7)] + ["blk1.tdnn2", "blk2.tdnn1"]
    lib = LIB.load_library()
    d = LIB.EcapaDesc()
    d.channels, d.sub_channels, d.scale, d.mfa_channels, d.n_blocks = 1024, 128, 8, 3072, 3
    assert lib.sdk_ecapa_calib_floats(C.byref(d), 5) == per * 5
    small = W.EcapaConfig(channels=256, mfa_channels=768)
    w = W.synthetic_weights(4, small)
    rng = np.random.default_rng(0)
    lay_s, _ = WP.calib_layout(small)
    means = {n: rng.uniform(0, 1, c) for n, c, _ in lay_s}
    bc = WP.bias_corrections(w, means, small)
    name = "blk2.res2net.3"
    wk = w[f"{name}.conv.w"].astype(np.float64)
    dw = wk - WP.bf16_bits_to_f32(WP.f32_to_bf16_bits(w[f"{name}.conv.w"])).astype(np.float64)
    want = w[f"{name}.conv.b"].astype(np.float64) + np.einsum("nkj,k->n", dw, means[name])
    assert np.allclose(bc[name], want, rtol=0, atol=1e-7)
    wa = w["asp.tdnn.conv.w"][:, :768, 0].astype(np.float64)
    dwa = wa - WP.bf16_bits_to_f32(WP.f32_to_bf16_bits(wa.astype(np.float32))).astype(np.float64)
    assert np.allclose(bc["asp.tdnn"], w["asp.tdnn.conv.b"] + dwa @ means["asp.tdnn"], rtol=0, atol=1e-7)
    blob, f = WP.pack_weights(w, small)
    for n in ("blk1.tdnn1", "blk3.res2net.6", "blk2.tdnn2", "mfa", "asp.tdnn"):
        o = f["off"][WP.bias_slot(n, small)]
        nb = w[f"{n}.conv.b"].shape[0]
        assert np.array_equal(blob[o:o + 4 * nb].view(np.float32), w[f"{n}.conv.b"]), n


def test_affinity_block_plan_covers_every_block_and_stage_exactly_once():
    """Round 4, k4 at short sweeps: the block plan (csrc/affinity_rowcol.hip plan_blocks / block_slots, shared by host and kernel).  Replayed here for
    every wave of every workgroup: each block of 32 segments is swept over every profile stage exactly once - a main block by one wave for the
    whole sweep, a leftover block by `parts` waves whose stage ranges tile [0, stages) - record slots 0..parts-1 each written once, and no
    SIMD (waves w and w + 4) carries more than q / 4 + 1 blocks.  The plan's own cost estimate prefers it at config #3 and not at config #4's long
    sweeps; measured end to end it is level with the range plan (weaker certificates: more rescans), so it runs on request only
    (`affinity_variant` 8; tests/test_gpu_kernels.py::test_affinity_block_plan_equals_the_range_plan)."""
    lib = LIB.load_library()
    out = (C.c_int32 * 6)()
    assert lib.sdk_affinity_block_plan(100_000, 1000, 256, 0, out) == 0 and list(out) == [1, 12, 256, 16, 3, 159]
    assert lib.sdk_affinity_block_plan(125_000, 10_000, 256, 0, out) == 0 and out[0] == 0          # long sweeps: the range plan balances to the stage
    assert lib.sdk_affinity_block_plan(125_000, 10_000, 256, 1, out) == 0 and out[0] == 1 and out[1] == 12 and out[4] == 1
    assert lib.sdk_affinity_block_plan(50_000, 1000, 256, 1, out) == 0 and out[0] == 0            # fewer than 8 blocks per workgroup: no block plan
    assert lib.sdk_affinity_block_plan(140_000, 1000, 256, 1, out) == 0 and out[0] == 0           # more leftover blocks than free slots
    w = (C.c_int32 * 6)()
    for N, P, cu in ((100_000, 1000, 256), (70_001, 333, 256), (98_303, 4100, 256), (131_000, 197, 256), (109_215, 1000, 256), (114_000, 640, 256), (26_000, 500, 64)):
        assert lib.sdk_affinity_block_plan(N, P, cu, 1, out) == 0 and out[0] == 1, (N, P, cu)
        _, q, G, nst, parts, items = list(out)
        NB = (N + 31) // 32
        assert q in (8, 12) and G == cu and nst == -(-(-(-P // 32)) // 2) and items == (NB - q * G) * parts and 1 <= parts <= 3
        cover = np.zeros((NB, nst), np.int32)
        slots = [set() for _ in range(NB)]
        for g in range(G):
            per_simd = [0, 0, 0, 0]
            for wave in range(8):
                assert lib.sdk_affinity_block_plan_wave(N, P, cu, g, wave, w) == 0
                b0, b1, e0, e1, s1, c1 = list(w)
                assert b0 == g * q + wave
                cover[b0, :] += 1
                slots[b0].add(0)
                per_simd[wave % 4] += 1
                if b1 >= 0:
                    assert 0 <= e0 < e1 <= nst and 0 <= s1 < c1 <= 3 and b1 < NB
                    cover[b1, e0:e1] += 1
                    assert s1 not in slots[b1]
                    slots[b1].add(s1)
                    per_simd[wave % 4] += 1
            assert max(per_simd) <= q // 4 + 1
        assert (cover == 1).all(), (N, P, np.argwhere(cover != 1)[:5])
        assert all(sl == set(range(len(sl))) for sl in slots)
        assert all(len(slots[b]) == (1 if b < q * G else parts) for b in range(NB))
