"""Error behaviour at the boundary.  C-ABI: an entry point that is handed arguments it cannot serve returns non-zero,
`sdk_last_error()` names the function and the offending value, NOTHING is launched, and the context keeps working.
Plug-in: the errors the toolkit's CLIs turn into `Error during identification: ...` / rc 1
(speaker_detection:1064-1074) are ordinary Python exceptions with a usable message."""
import struct

import numpy as np
import pytest
import torch

from conftest import sub

pytestmark = pytest.mark.gpu

LIB = sub("_lib")
wav = sub("wav")


def test_cabi_rejects_bad_shapes_and_recovers(engine):
    A = torch.zeros((512, 128), dtype=torch.bfloat16, device="cuda")
    W = torch.zeros((128, 128), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(LIB.SdkError, match=r"sdk_conv_gemm: N=100 must be a positive multiple of 128"):
        engine.conv_gemm(A, W, 100, 128)
    with pytest.raises(LIB.SdkError, match=r"sdk_conv_gemm: Cin=96 must be a multiple of 64"):
        engine.conv_gemm(A, W, 128, 96)
    with pytest.raises(LIB.SdkError, match=r"taps=2 must be odd"):
        engine.conv_gemm(A, W, 128, 64, taps=2)
    with pytest.raises(LIB.SdkError, match=r"M=512 must be a multiple of T=200"):
        engine.conv_gemm(A, W, 128, 128, T=200)
    with pytest.raises(LIB.SdkError, match=r"shorter than the conv halo"):
        engine.conv_gemm(torch.zeros((512, 128), dtype=torch.bfloat16, device="cuda"), torch.zeros((128, 384), dtype=torch.bfloat16, device="cuda"),
                         128, 128, taps=3, dil=4, T=4)
    with pytest.raises(LIB.SdkError, match=r"sdk_affinity_topk"):
        E, Eb, re = engine.l2norm(torch.randn(10, 192, device="cuda"))
        engine.affinity_topk(E, Eb, re, E, Eb, re.max().reshape(1), k=11)          # k > number of profiles
    with pytest.raises(LIB.SdkError, match=r"sdk_resample_s16: n_out="):
        taps = torch.zeros((1, 2), dtype=torch.int32, device="cuda")
        x = torch.zeros((10,), dtype=torch.int16, device="cuda")
        y = torch.zeros((7,), dtype=torch.int16, device="cuda")
        LIB.check(engine.lib.sdk_resample_s16(engine.ctx, x.data_ptr(), 10, 1, taps.data_ptr(), 1, 1, 2, y.data_ptr(), 7, 0), "sdk_resample_s16")
    with pytest.raises(LIB.SdkError, match=r"null"):
        LIB.check(engine.lib.sdk_l2norm(engine.ctx, None, 4, 192, None, None, None, 0), "sdk_l2norm")
    # the context is intact: a valid call right after the failures gives the right answer
    A = torch.ones((256, 64), dtype=torch.bfloat16, device="cuda")
    W = torch.ones((128, 64), dtype=torch.bfloat16, device="cuda")
    C, _, _ = engine.conv_gemm(A, W, 128, 64)
    torch.cuda.synchronize()
    assert torch.equal(C.float(), torch.full((256, 128), 64.0, device="cuda"))


def test_wrong_dtype_or_host_tensor_is_refused_before_the_call(engine):
    with pytest.raises(LIB.SdkError, match="must be torch.float32"):
        engine.l2norm(torch.zeros((4, 192), dtype=torch.float64, device="cuda"))
    with pytest.raises(LIB.SdkError, match="there is no CPU path"):
        engine.l2norm(torch.zeros((4, 192), dtype=torch.float32))                   # host memory


def test_plugin_errors(tmp_path, monkeypatch):
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path / "store"))
    be = sub("backend").Backend()
    good = tmp_path / "a.wav"
    wav.write_wav_s16(good, (3000 * np.sin(np.arange(48000) / 20.0)).astype(np.int16))
    # no candidates with an embedding for this backend: nothing to do, no GPU work, empty list (the CLI prints [])
    assert be.identify_speaker(good, [], threshold=0.354) == []
    assert be.identify_speaker(good, [{"id": "x", "embeddings": {"speechmatics": [{"id": "e", "external_id": "spk_1"}]}}]) == []
    # missing / unreadable / non-PCM audio
    with pytest.raises(FileNotFoundError):
        be.enroll_speaker(tmp_path / "missing.wav")
    bad = tmp_path / "b.wav"
    bad.write_bytes(b"RIFF" + struct.pack("<I", 4) + b"WAVE")
    with pytest.raises(wav.AudioFormatError, match="missing fmt/data chunk"):
        be.enroll_speaker(bad)
    # every requested segment shorter than 0.5 s
    with pytest.raises(ValueError, match="no analysable audio"):
        be.enroll_speaker(good, [(0.0, 0.2), (1.0, 1.3)])
    # compatibility follows the toolkit's rule: model_version must start with "<backend name>-" (base.py:92-93)
    rec = be.enroll_speaker(good)
    same = be.check_embedding_compatibility({"model_version": "mi355x-other-model"})
    other = be.check_embedding_compatibility({"model_version": "speechmatics-v2"})
    assert same["compatible"] is True and same["current"] == rec["model_version"] and other["compatible"] is False
    # a profile pointing at a vanished .npy is reported and skipped; when that leaves NOTHING to compare against, the call is loud
    # (the CLI's "Error during identification: ..." + rc 1) instead of answering "no match"
    ghost = {"id": "g", "embeddings": {"mi355x": [{"id": "emb-g", "external_id": "npy:000000000000000000000000", "model_version": rec["model_version"]}]}}
    with pytest.raises(ValueError, match="1 of 1 enrolled embeddings are unusable"):
        be.identify_speaker(good, [ghost])
    ok = {"id": "a", "embeddings": {"mi355x": [{"id": "emb-a", "external_id": rec["external_id"], "model_version": rec["model_version"]}]}}
    rows = be.identify_speaker(good, [ghost, ok], threshold=0.354)
    assert [r["speaker_id"] for r in rows] == ["a"] and rows[0]["similarity"] > 0.99


def test_cabi_device_memory_entry_points(engine):
    """sdk_device_malloc / sdk_memcpy / sdk_device_free (the allocator of a host that brings none, lite.py): a round trip through library-owned
    memory, interoperability with torch's pointers (same HIP runtime in this process), and the argument checks."""
    import ctypes as C
    lib, ctx = engine.lib, engine.ctx
    host = np.arange(1000, dtype=np.float32)
    p = C.c_void_p()
    LIB.check(lib.sdk_device_malloc(ctx, host.nbytes, C.byref(p)), "sdk_device_malloc")
    assert p.value
    LIB.check(lib.sdk_memcpy(ctx, p, host.ctypes.data, host.nbytes, 1, None), "sdk_memcpy")
    t = torch.empty(1000, dtype=torch.float32, device="cuda")
    LIB.check(lib.sdk_memcpy(ctx, t.data_ptr(), p, host.nbytes, 3, None), "sdk_memcpy")        # device -> device into a torch tensor
    back = np.empty_like(host)
    LIB.check(lib.sdk_memcpy(ctx, back.ctypes.data, t.data_ptr(), host.nbytes, 2, None), "sdk_memcpy")
    assert np.array_equal(back, host) and torch.equal(t.cpu(), torch.from_numpy(host))
    with pytest.raises(LIB.SdkError, match=r"sdk_memcpy: kind=7"):
        LIB.check(lib.sdk_memcpy(ctx, p, host.ctypes.data, 4, 7, None), "sdk_memcpy")
    with pytest.raises(LIB.SdkError, match=r"sdk_memcpy: null argument"):
        LIB.check(lib.sdk_memcpy(ctx, None, host.ctypes.data, 4, 1, None), "sdk_memcpy")
    LIB.check(lib.sdk_memcpy(ctx, None, None, 0, 1, None), "sdk_memcpy")                          # empty copies are fine
    LIB.check(lib.sdk_stream_synchronize(ctx, None), "sdk_stream_synchronize")
    LIB.check(lib.sdk_device_free(ctx, p), "sdk_device_free")
    LIB.check(lib.sdk_device_free(ctx, None), "sdk_device_free")
    with pytest.raises(LIB.SdkError, match=r"sdk_device_malloc: null argument"):
        LIB.check(lib.sdk_device_malloc(ctx, 16, None), "sdk_device_malloc")
