"""examples/c_host_identify.c: the identify path (k1 -> k4) driven by a C99 program through include/sdk_hip.h alone - no Python, no torch, device
memory from sdk_device_malloc.  CPU: the header is valid C99 and the program links against libsdk_hip.so.  GPU: its answers (best profile, exact
cosine, unit-norm embeddings) equal the Python engine's bit for bit on the same PCM, weights and profiles."""
import subprocess

import numpy as np
import pytest

from conftest import ROOT, sub

PKG_DIR = ROOT / sub("backend").__package__


def _build(tmp_path):
    exe = tmp_path / "c_host_identify"
    r = subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / "examples" / "c_host_identify.c"), "-o", str(exe),
                        f"-L{PKG_DIR}", "-lsdk_hip", f"-Wl,-rpath,{PKG_DIR}"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def test_c_host_compiles_against_the_header(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)      # no arguments: usage, rc 2 (no device is touched)
    assert r.returncode == 2 and "usage:" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("bias_correction", [False, True])
def test_c_host_equals_the_python_engine(tmp_path, engine, bias_correction):
    import sys
    import torch
    if bias_correction:                      # the shipped default: the blob the C host maps is the bias-corrected one
        engine = sub("ops").Engine(0, bias_correction=True)
    sys.path.insert(0, str(ROOT))
    import bench
    B, S, P = 7, 32000, 33
    pcm = bench.synth_pcm(B, seed=3)
    prof = bench.unit_rows(P, 192, seed=4) * np.float32(1.7)                      # not unit length: the C host normalises them itself
    E, Eb, re = engine.embed_pcm(torch.from_numpy(pcm).cuda())
    Pn, Pb, rp = engine.l2norm(torch.from_numpy(prof).cuda())
    idx, sc = engine.affinity_topk(E, Eb, re, Pn, Pb, rp.max().reshape(1), k=1)
    torch.cuda.synchronize()
    blob, desc = engine._wblob, engine.desc
    (tmp_path / "pcm.s16").write_bytes(pcm.astype("<i2").tobytes())
    (tmp_path / "blob.bin").write_bytes(blob.cpu().numpy().tobytes())
    (tmp_path / "desc.bin").write_bytes(bytes(desc))
    (tmp_path / "prof.f32").write_bytes(prof.astype("<f4").tobytes())
    exe = _build(tmp_path)
    r = subprocess.run([str(exe), str(tmp_path / "pcm.s16"), str(B), str(S), str(tmp_path / "blob.bin"), str(tmp_path / "desc.bin"),
                        str(tmp_path / "prof.f32"), str(P), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-1500:])
    raw = (tmp_path / "out.bin").read_bytes()
    c_idx = np.frombuffer(raw[:4 * B], dtype="<i4")
    c_sc = np.frombuffer(raw[4 * B:8 * B], dtype="<f4")
    c_E = np.frombuffer(raw[8 * B:], dtype="<f4").reshape(B, 192)
    assert np.array_equal(c_idx, idx.cpu().numpy()[:, 0]) and np.array_equal(c_sc, sc.cpu().numpy()[:, 0])
    assert np.array_equal(c_E, E.cpu().numpy())
    assert "window 0 -> profile" in r.stdout
