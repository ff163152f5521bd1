"""ingest.plan_chunks (host logic, no GPU): a recording longer than a staging slot is uploaded in overlapping pieces; every window goes to
exactly one piece, lies wholly inside it (or runs past the END of the recording, as in the one-piece form) and keeps its samples."""
import numpy as np
import pytest

from conftest import sub

ingest = sub("ingest")


def test_one_piece_returns_the_tables_unchanged():
    t = {32000: np.array([0, 16000, 5], np.int32), 8000: np.zeros((0,), np.int32)}
    (lo, hi, sub_), = ingest.plan_chunks(100000, t, 1 << 20)
    assert (lo, hi) == (0, 100000) and sub_[32000][0] is None and np.array_equal(sub_[32000][1], t[32000]) and len(sub_[8000][1]) == 0
    assert ingest.plan_chunks(5, {}, 1 << 20) == [(0, 5, {})]


@pytest.mark.parametrize("n, cap, seed", [(1_000_000, 1 << 17, 0), (700_001, 70_000, 1), (1 << 22, 1 << 20, 2)])
def test_every_window_lands_in_exactly_one_piece_with_its_samples(n, cap, seed):
    rng = np.random.default_rng(seed)
    rec = rng.integers(-30000, 30000, n).astype(np.int16)
    tables = {32000: rng.integers(0, n, 300).astype(np.int32), 8000: np.sort(rng.integers(0, n, 200)).astype(np.int32), 16000: np.array([n - 1, 0], np.int32)}
    pieces = ingest.plan_chunks(n, tables, cap)
    assert len(pieces) > 1 and all(0 <= lo < hi <= n and hi - lo <= cap for lo, hi, _ in pieces)
    for S, st in tables.items():
        seen = np.zeros(len(st), int)
        for lo, hi, sub_ in pieces:
            if S not in sub_:
                continue
            rows, local = sub_[S]
            seen[rows] += 1
            assert np.array_equal(st[rows], local.astype(np.int64) + lo) and (local >= 0).all() and (local < hi - lo).all()
            for r, l in zip(rows[:20], local[:20]):                     # the window as the device will read it: zeros past the piece's end
                w = np.zeros(S, np.int16)
                m = min(S, hi - lo - l)
                w[:m] = rec[lo + l:lo + l + m]
                ref = np.zeros(S, np.int16)
                k = min(S, n - st[r])
                ref[:k] = rec[st[r]:st[r] + k]
                assert np.array_equal(w, ref)
        assert (seen == 1).all()


def test_a_chunk_shorter_than_two_windows_is_refused():
    with pytest.raises(ValueError, match="too short"):
        ingest.plan_chunks(1_000_000, {32000: np.array([0], np.int32)}, 60000)
