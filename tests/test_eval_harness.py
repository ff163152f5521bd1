"""evals/run_eval.py - the enrol/identify accuracy check modelled on the toolkit's
evals/speaker_detection/benchmark.py:72-185 (same case files, same pass rule).  CPU: case loading, dry run, return
codes, loud failure without a GPU.  GPU: the three cases pass on rendered stand-in voices."""
import importlib.util
import json
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("run_eval", ROOT / "evals" / "run_eval.py")
run_eval = importlib.util.module_from_spec(spec)
spec.loader.exec_module(run_eval)


def test_case_files_and_filter():
    cases = run_eval.load_cases(ROOT / "evals" / "samples")
    assert [c["id"] for c in cases] == ["001-two-speakers", "002-three-speakers", "003-absent-speaker"]
    assert cases[0]["expected_speakers"] == ["alice", "bob"] and set(cases[1]["speakers"]) == {"alice", "bob", "charlie"}
    assert [c["id"] for c in run_eval.load_cases(ROOT / "evals" / "samples", "002,003")] == ["002-three-speakers", "003-absent-speaker"]
    assert run_eval.load_cases(ROOT / "evals" / "samples", "9") == []


def test_dry_run_and_missing_audio_codes(tmp_path, capsys):
    assert run_eval.main(["--dry-run", "--audio-root", str(tmp_path)]) == 0
    out = capsys.readouterr().out
    assert "001-two-speakers: MISSING" in out and "Tests: 3" in out
    assert run_eval.main(["--audio-root", str(tmp_path), "-t", "001"]) == 2          # the toolkit's "run make first" exit
    assert run_eval.main(["-t", "nope"]) == 1


def test_rendered_voices_are_deterministic_and_distinct():
    a1, a2 = run_eval.render_voice("en-us", 1.0, 5), run_eval.render_voice("en-us", 1.0, 5)
    b = run_eval.render_voice("en-gb", 1.0, 5)
    assert a1.dtype == np.int16 and len(a1) == 16000 and np.array_equal(a1, a2) and not np.array_equal(a1, b)
    assert np.abs(a1).max() <= 16384 and run_eval._voice_params("en-us") != run_eval._voice_params("en-au")


def test_without_gpu_fails_loudly(tmp_path, capsys):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_cases_pass_on_gpu")
    rc = run_eval.main(["--synthesize", "--audio-root", str(tmp_path), "-t", "001"])
    out = capsys.readouterr().out
    assert rc == 1 and "FAIL: Enrollment failed for alice" in out and "no CPU fallback" in out


@pytest.mark.gpu
def test_cases_pass_on_gpu(tmp_path, capsys):
    rc = run_eval.main(["--synthesize", "--audio-root", str(tmp_path), "--json"])
    doc = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert rc == 0 and doc["passed"] == 3 and doc["failed"] == 0, doc
    by_id = {r["test_id"]: r for r in doc["results"]}
    assert by_id["003-absent-speaker"]["enrolled"] == ["alice", "bob", "charlie"]
    assert sorted(by_id["003-absent-speaker"]["identified"]) == ["alice", "bob"]
    assert all(0.354 <= s <= 1.0 for r in doc["results"] for s in r["scores"].values())
