import importlib
import json
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

PKG = "speaker-diarization-toolkit_amd"
# KERNEL-parity tests (the session `engine` fixture) compare the GPU with the PLAIN bf16 layer-boundary model of the given weights, so the default
# of the test processes and their spawned ranks is the uncorrected engine.  The SHIPPED default (bias correction ON: calibration pass -> 29 patched
# bias slots -> cache entry "0c" -> lite path maps that blob) is exercised explicitly: every Backend-level GPU test is parametrised over
# SDK_BIAS_CORRECTION in {0, 1} (test_gpu_backend_e2e, test_lite, test_gpu_sentences, test_c_host) against the oracle on the engine's EFFECTIVE
# weights, and tests/test_gpu_bias_correction.py covers the cache round trip (fresh / hit / lite bit-identical) and the correction's budget.
import os
os.environ.setdefault("SDK_BIAS_CORRECTION", "0")
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def sub(name: str):
    """Import a submodule of the (hyphenated) product package."""
    return importlib.import_module(f"{PKG}.{name}" if name else PKG)


@pytest.fixture(scope="session")
def golden():
    return json.loads((GOLDEN / "plumbing_golden.json").read_text())


@pytest.fixture(scope="session")
def fixture_transcript_path():
    return GOLDEN / "test_001-two-speakers.wav.speechmatics.json"


@pytest.fixture(scope="session")
def engine():
    """Session-wide GPU engine (C=1024 synthetic weights, seed 0).  gpu tests only."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test collected without a GPU: run with -m 'not gpu' on CPU boxes")
    return sub("ops").get_engine(0)
