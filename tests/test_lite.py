"""The torch-free host path (SDK_NO_TORCH=1, lite.py): the same library calls as ops.Engine with numpy on the host side and device memory from
sdk_device_malloc, so that a CLI process - the reference builds its backend fresh in every process (speaker_detection_backends/base.py:291-293) - reaches
its first row without `import torch`.  CPU: the import chain really is torch-free.  GPU (child processes: this test process has torch loaded, and one
process must hold one HIP runtime): enroll / identify / verify rows equal to the torch engine's, for both model families, from an EMPTY cache (the
entry is built once by a grandchild through the torch engine) and from the warm cache."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, sub

PKG = sub("backend").__package__

CHILD = r"""
import importlib, json, os, sys
sys.path.insert(0, sys.argv[1])
from pathlib import Path
B = importlib.import_module(sys.argv[2] + ".backend")
be = B.Backend()
tmp = Path(sys.argv[3])
profiles = []
for sid in ("alice", "bob"):
    rec = be.enroll_speaker(tmp / f"enroll_{sid}.wav", [(0.5, 5.5)] if sid == "alice" else None)
    profiles.append({"id": sid, "embeddings": {"mi355x": [{"id": "emb-" + sid, "external_id": rec["external_id"], "model_version": rec["model_version"]}]}})
rows = be.identify_speaker(tmp / "meeting.wav", profiles, threshold=-1.0)
ver = be.verify_speaker(tmp / "meeting48k.wav", profiles[0], threshold=-1.0)          # 48 kHz stereo: the GPU resampler on the way in
vecs = {p["id"]: __import__("numpy").load(str(B.vector_path(p["embeddings"]["mi355x"][0]["external_id"]))).tolist() for p in profiles}
# the speaker-assign compatible driver on top (per-label identify through Backend.score_ranges): speaker-assign:262-328, 418-492
db = Path(os.environ["SPEAKERS_EMBEDDINGS_DIR"]) / "db"
db.mkdir(parents=True, exist_ok=True)
for p in profiles:
    p["names"] = {"default": p["id"]}
    p["embeddings"]["mi355x"][0]["trust_level"] = "high"
    (db / (p["id"] + ".json")).write_text(json.dumps(p))
transcript = tmp / ("t_" + os.environ["TAG"] + ".json")
transcript.write_text(json.dumps({"results": [
    {"type": "word", "start_time": 0.2, "end_time": 3.8, "alternatives": [{"content": "hello", "speaker": "S1"}]},
    {"type": "word", "start_time": 4.2, "end_time": 7.8, "alternatives": [{"content": "there", "speaker": "S2"}]}]}))
asg, ident = importlib.import_module(sys.argv[2] + ".assign"), importlib.import_module(sys.argv[2] + ".identify")
out = asg.assign_recording(tmp / "meeting.wav", transcript, rows_fn=ident.make_rows_fn(tmp / "meeting.wav", per_label=True, backend=be), use_embeddings=True, threshold=0.1)
maps = {k: {f: v[f] for f in ("speaker_id", "confidence", "score", "signals")} for k, v in out["mappings"].items()}
print(json.dumps({"rows": rows, "verify": ver, "vecs": vecs, "mappings": maps, "torch": "torch" in sys.modules, "model_version": be.model_version, "cache_hit": bool(be._cache_hit)}))
"""


def _run_child(tmp_path, env_extra):
    env = dict(os.environ, SPEAKERS_EMBEDDINGS_DIR=str(tmp_path / ("store_" + env_extra["TAG"])), HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD, str(ROOT), PKG, str(tmp_path)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_lite_import_chain_is_torch_free():
    code = ("import importlib, sys, os; os.environ['SDK_NO_TORCH'] = '1'; sys.path.insert(0, sys.argv[1]);"
            "[importlib.import_module(sys.argv[2] + m) for m in ('.lite', '.backend', '.store', '.wav', '.plugin_api', '.xvector')];"
            "print('torch' in sys.modules)")
    r = subprocess.run([sys.executable, "-c", code, str(ROOT), PKG], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "False", (r.stdout, r.stderr[-500:])


def test_lite_serves_the_single_plane_contracts_only(monkeypatch):
    monkeypatch.setenv("SDK_NO_TORCH", "1")
    monkeypatch.setenv("SDK_PRECISION", "1")
    with pytest.raises(ValueError, match="SDK_PRECISION=1"):
        sub("backend").Backend()._lite_engine()


@pytest.mark.gpu
@pytest.mark.parametrize("model,bias_correction,precision", [("ecapa", "1", "0"), ("ecapa", "0", "0"), ("xvector", "1", "0"), ("ecapa", "1", "2"), ("xvector", "1", "2")])
def test_lite_rows_equal_the_torch_engine(tmp_path, model, bias_correction, precision):
    from test_gpu_backend_e2e import _voice
    wav = sub("wav")
    wav.write_wav_s16(tmp_path / "enroll_alice.wav", _voice(10, 6.0, 140.0))
    wav.write_wav_s16(tmp_path / "enroll_bob.wav", _voice(11, 6.0, 95.0))
    meeting = np.concatenate([_voice(40, 4.0, 95.0), _voice(41, 4.0, 140.0)])
    wav.write_wav_s16(tmp_path / "meeting.wav", meeting)
    up = np.repeat(meeting, 3)                                                   # crude 48 kHz stereo rendition: both paths decode it through the same GPU resampler
    import struct
    pcm = np.stack([up, up], axis=1).astype("<i2").tobytes()
    (tmp_path / "meeting48k.wav").write_bytes(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, 48000, 48000 * 4, 4, 16)
                                              + b"data" + struct.pack("<I", len(pcm)) + pcm)
    cache = tmp_path / "cache"
    base = {"SDK_MODEL": model, "SDK_CACHE_DIR": str(cache), "SDK_BIAS_CORRECTION": bias_correction, "SDK_PRECISION": precision}     # "1", "0" = the shipped default (cache entry "p0c"); precision 2 = one fp16 plane ("p2c")
    lite_cold = _run_child(tmp_path, dict(base, SDK_NO_TORCH="1", TAG="lite_cold"))       # empty cache: a grandchild builds the entry (ecapa)
    ref = _run_child(tmp_path, dict(base, TAG="torch"))
    lite_warm = _run_child(tmp_path, dict(base, SDK_NO_TORCH="1", TAG="lite_warm"))
    assert lite_cold["torch"] is False and lite_warm["torch"] is False and ref["torch"] is True
    assert lite_cold["model_version"] == ref["model_version"] == lite_warm["model_version"]
    if model == "ecapa":
        assert lite_warm["cache_hit"] is True and any(cache.glob(f"*.p{precision}c.npy" if bias_correction == "1" else f"*.p{precision}.npy"))
        assert not any(cache.glob(f"*.p{precision}.npy" if bias_correction == "1" else f"*.p{precision}c.npy"))
    for got in (lite_cold, lite_warm):
        # same kernels on the same inputs: stored vectors, scores and rows are the torch engine's, bit for bit
        assert got["vecs"] == ref["vecs"]
        assert got["rows"] == ref["rows"] and got["verify"] == ref["verify"] and got["mappings"] == ref["mappings"]
    if model == "ecapa" and bias_correction == "1" and precision == "0":
        # a recording longer than a staging slot may be goes through the ring in pieces (ingest.plan_chunks; here 4.1-s pieces of an 8-s file):
        # both host paths cut the same pieces, so their rows agree bit for bit again, and the decisions are those of the one-piece run
        ch = {"SDK_INGEST_CHUNK": "65536"}
        lite_ch = _run_child(tmp_path, dict(base, SDK_NO_TORCH="1", TAG="lite_chunk", **ch))
        ref_ch = _run_child(tmp_path, dict(base, TAG="torch_chunk", **ch))
        assert lite_ch["rows"] == ref_ch["rows"] and lite_ch["verify"] == ref_ch["verify"] and lite_ch["mappings"] == ref_ch["mappings"] and lite_ch["vecs"] == ref_ch["vecs"]
        assert [r["speaker_id"] for r in ref_ch["rows"]] == [r["speaker_id"] for r in ref["rows"]]
        assert all(abs(a["confidence"] - b["confidence"]) < 1e-3 for a, b in zip(ref_ch["rows"], ref["rows"]))
    assert {r["speaker_id"] for r in ref["rows"]} == {"alice", "bob"} and ref["verify"]["match"] is True
    assert ref["mappings"]["S1"]["speaker_id"] == "bob" and ref["mappings"]["S2"]["speaker_id"] == "alice"
