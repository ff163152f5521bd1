"""the embedding-record writer and the record / profile checks, against verdicts captured from the
reference's own validators and its own `enroll` CLI (tests/golden/make_golden.py -> schema_golden.json)."""
import json
from pathlib import Path

import pytest

from conftest import sub

REC = sub("records")
G = json.loads((Path(__file__).parent / "golden" / "schema_golden.json").read_text())


@pytest.mark.parametrize("case", G["embedding"], ids=lambda c: c["name"])
def test_embedding_findings_match_the_reference_validator(case):
    assert REC.embedding_issues(case["record"]) == case["warnings"]
    if case["strict_error"] is None:
        REC.embedding_issues(case["record"], strict=True)
    else:
        with pytest.raises(REC.RecordError) as exc:
            REC.embedding_issues(case["record"], strict=True)
        assert str(exc.value) == case["strict_error"]


@pytest.mark.parametrize("case", G["profile"], ids=lambda c: c["name"])
def test_profile_findings_match_the_reference_validator(case):
    assert REC.profile_issues(case["profile"]) == case["warnings"]
    if case["strict_error"] is None:
        REC.profile_issues(case["profile"], strict=True)
    else:
        with pytest.raises(REC.RecordError) as exc:
            REC.profile_issues(case["profile"], strict=True)
        assert str(exc.value) == case["strict_error"]


@pytest.mark.parametrize("run", G["enroll_cli"]["runs"], ids=lambda r: r["name"])
def test_record_writer_reproduces_what_the_reference_cli_stored(run):
    """Same backend result, same audio, same segments -> the same record, key for key and in the same order
    (id and created_at are minted per call and passed in here; the stored `file` key is dropped, as the CLI drops it)."""
    assert run["rc"] == 0 and run["validate"] == [] and run["profile_validate"] == []
    want = run["record"]
    result = {"external_id": "npy:0123456789abcdef01234567", "file": "dropped.npy", "model_version": "mi355x-ecapa1024-2f6c1e0d9b7a",
              "source_audio": "x", "source_segments": None}
    segs = [(s["start"], s["end"]) for s in want["source_segments"]] if want["source_segments"] else None
    trust = "high" if run["name"] == "trust_override" else None
    got = REC.make_embedding_record(result, Path(G["enroll_cli"]["audio_path"]), G["enroll_cli"]["wav_sha256_32"], segs,
                                    trust_level=trust, emb_id=want["id"], created_at=want["created_at"])
    assert got == want and list(got) == list(want)
    assert REC.embedding_issues(got) == []


def test_minted_fields_and_attach():
    rec = REC.make_embedding_record({"external_id": "npy:" + "ab" * 12, "model_version": "mi355x-ecapa1024-x"}, Path("a.wav"), "00" * 16)
    assert rec["id"].startswith("emb-") and len(rec["id"]) == 12 and rec["trust_level"] == "low"
    assert REC.embedding_issues(rec) == []
    prof = {"id": "bob", "names": {"default": "Bob"}}
    REC.attach_embedding(prof, "mi355x", rec)
    assert prof["embeddings"]["mi355x"] == [rec] and REC.profile_issues(prof) == []
    with pytest.raises(REC.RecordError):
        REC.attach_embedding(prof, "mi355x", {"id": "emb-1"})                    # the toolkit would reject it too
    for samples, want in (({"reviewed": ["a"]}, "high"), ({"reviewed": ["a"], "unreviewed": ["b"]}, "medium"),
                          ({"unreviewed": ["b"]}, "low"), ({"reviewed": ["a"], "rejected": ["c"]}, "invalidated")):
        assert REC.trust_from_samples(samples) == want
