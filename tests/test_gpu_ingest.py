"""Ingest from host audio: a recording crosses PCIe once, as it lies in the file, next to an int32 table of window starts;
the windows are cut on the device (sdk_fbank_windows) and the upload is staged through pinned slots on a copy stream (csrc/ingest.hip).
Every result must equal the host-windowed path bit for bit.  Boundary being served: speaker_detection_backends/base.py:130-151 (a path and
segments); the reference cuts with ffmpeg per segment list (speechmatics_backend.py:231-281)."""
import numpy as np
import pytest
import torch

from conftest import sub

pytestmark = pytest.mark.gpu

wav = sub("wav")
SdkError = sub("_lib").SdkError


def _recording(seconds, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(int(16000 * seconds)) / 16000.0
    x = 0.2 * np.sin(2 * np.pi * (110 + 7 * seed) * t) + 0.1 * np.sin(2 * np.pi * (900 + 31 * seed) * t) + rng.normal(0, 0.08, t.shape)
    return np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("precision", [0, 1])
def test_fbank_windows_equals_fbank_on_materialised_windows(engine, precision):
    rec = _recording(7.3, 1)
    n = len(rec)
    # overlapping windows, one flush with the end, two running PAST the end (zero padded), one of a single valid sample
    starts = np.array([0, 16000, 32000, 5, 12345, n - 32000, n - 20000, n - 3000, n - 1], dtype=np.int32)
    for S in (32000, 8000):
        pcm = wav.materialise_windows(rec, starts, S)
        assert pcm[-1, 1:].max() == 0 and pcm[-1, 0] == rec[-1]
        engine.set_precision(precision)
        try:
            want = engine.fbank(torch.from_numpy(pcm).cuda())
            ing = engine.ingest()
            t, ds, dw = ing.submit(rec, starts, S, torch.cuda.current_stream().cuda_stream)
            got = engine.fbank_windows(ds, n, dw, len(starts), S)
            sub_ = engine.fbank_windows(ds, n, dw + 4 * 3, 4, S)                  # any sub-range of the table
            ing.release(t, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
        finally:
            engine.set_precision(0)
        T = 1 + S // 160
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
        assert torch.equal(sub_.view(torch.int16), want.view(torch.int16)[3 * T:7 * T])


def test_embed_from_host_equals_the_host_windowed_path_and_pipelines(engine):
    """Several recordings back to back through the two staging slots (the upload of recording i + 1 is issued while recording i computes), two
    window lengths each, batches of 5 windows: embeddings bit-identical to embed_pcm on host-cut windows; slots reused three times over."""
    recs = [_recording(5.0 + 1.7 * i, 10 + i) for i in range(6)]
    outs, wants = [], []
    for rec in recs:                                       # enqueue everything first: no synchronisation between recordings
        s2, _, W = wav.window_starts(len(rec), None)
        s1 = np.arange(0, len(rec) - 16000, 9000, dtype=np.int32)
        outs.append((engine.embed_from_host(rec, {W: s2, 16000: s1}, step=5), s2, s1, W))
    for rec, (_, s2, s1, W) in zip(recs, outs):
        wants.append({W: engine.embed_pcm(torch.from_numpy(wav.materialise_windows(rec, s2, W)).cuda()),
                      16000: engine.embed_pcm(torch.from_numpy(wav.materialise_windows(rec, s1, 16000)).cuda())})
    torch.cuda.synchronize()
    for (got, s2, s1, W), want in zip(outs, wants):
        for S in (W, 16000):
            # batches of 5 start on other tile rows than one batch of all windows: per-segment fp32 statistics are summed in another grouping
            # (test_embed_windows_in_bounded_batches); the first batch is bit-identical, the rest within the batch-invariance tolerance
            assert torch.equal(got[S][0][:5], want[S][0][:5]) and torch.equal(got[S][2][:5], want[S][2][:5])
            assert float((got[S][0] - want[S][0]).abs().max()) < 1e-3
    # with one batch per bucket everything is bit-identical
    rec = recs[3]
    s2, _, W = wav.window_starts(len(rec), None)
    got = engine.embed_from_host(rec, {W: s2})[W]
    want = engine.embed_pcm(torch.from_numpy(wav.materialise_windows(rec, s2, W)).cuda())
    torch.cuda.synchronize()
    assert all(torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b)
               for a, b in zip(got, want))
    ms, nbytes = engine.ingest().copy_ms(engine.last_ingest_ticket)
    assert nbytes == len(rec) * 2 + len(s2) * 4 and ms > 0


def test_ingest_refuses_bad_tables_and_unreleased_slots(engine):
    ing = sub("ingest").Ingest(engine.lib, engine.ctx, max_samples=1 << 16, max_windows=64, depth=2)
    rec = _recording(2.0, 3)
    st = torch.cuda.current_stream().cuda_stream
    with pytest.raises(SdkError, match="outside the recording"):
        ing.submit(rec, np.array([0, len(rec)], np.int32), 8000, st)
    with pytest.raises(SdkError, match="outside the recording"):
        ing.submit(rec, np.array([-1], np.int32), 8000, st)
    t0, _, _ = ing.submit(rec, np.array([0], np.int32), 8000, st)      # a refused table leaves its slot free
    t1, _, _ = ing.submit(rec, np.array([5], np.int32), 8000, st)
    with pytest.raises(SdkError, match="never released"):
        ing.submit(rec, np.array([9], np.int32), 8000, st)
    ing.release(t0, st)
    with pytest.raises(SdkError, match="not a committed slot"):
        ing.release(t0, st)
    t2, _, _ = ing.submit(rec, np.array([9], np.int32), 8000, st)
    assert t2 == t0
    ing.release(t1, st); ing.release(t2, st)
    # the zero-copy form: fill the pinned views, then commit
    t, ps, pw = ing.pinned(len(rec), 2)
    ps[:] = rec; pw[:] = (0, 100)
    ds, dw = ing.commit(t, len(rec), 2, 8000, st)
    back = torch.empty(len(rec), dtype=torch.int16, device="cuda")
    engine.lib.sdk_memcpy(engine.ctx, back.data_ptr(), ds, len(rec) * 2, 3, st)
    ing.release(t, st)
    assert np.array_equal(back.cpu().numpy(), rec)
    # a filler that gives up between acquire and commit does not wedge the ring: the slot comes round again
    seen = [ing.pinned(len(rec), 1)[0] for _ in range(5)]
    assert seen == [seen[0], 1 - seen[0]] * 2 + [seen[0]]
    t3, _, _ = ing.submit(rec, np.array([0], np.int32), 8000, st)
    ing.release(t3, st)
    ing.close()


def test_backend_ingest_path_equals_host_cut_windows(tmp_path, monkeypatch):
    """Through the plug-in class: what identify / enroll now run (recording uploaded once, windows cut on the device) against the previous form
    (windows materialised on the host, uploaded as [B, 32000])."""
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path / "store"))
    be = sub("backend").Backend()
    rec = _recording(9.4, 21)
    samples, starts, W, spans = None, None, None, None
    path = tmp_path / "r.wav"
    wav.write_wav_s16(path, rec)
    samples, starts, W, spans = be._windows(path, None)
    pcm, spans2 = wav.cut_windows(rec, None)
    assert spans == spans2 and np.array_equal(samples, rec) and len(starts) == len(pcm) == 9
    a = be.embed_tables(samples, {W: starts})[W]
    b = be.embed_windows(pcm)
    torch.cuda.synchronize()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1].view(torch.int16), b[1].view(torch.int16)) and torch.equal(a[2], b[2])
    ra = be.embed_ranges(rec, [(0.2, 3.1), (3.3, 4.0), (4.2, 9.0)])
    pcm_by_len, wins, _ = wav.cut_ranges(rec, [(0.2, 3.1), (3.3, 4.0), (4.2, 9.0)])
    for w, (_, S, row, _, _) in enumerate(wins):
        want = be.embed_windows(pcm_by_len[S])[0][row]
        assert torch.equal(ra[0][w], want)


def test_ingest_edge_cases(engine, tmp_path, monkeypatch):
    """Empty tables, a recording shorter than a window (one zero-padded window), a recording too short to analyse (the error the CLI prints), and a
    window table far larger than the staging slots' first size (they grow)."""
    assert engine.embed_from_host(_recording(1.0, 2), {}) == {}
    empty = engine.embed_from_host(_recording(3.0, 2), {32000: np.array([0], np.int32), 8000: np.zeros((0,), np.int32)})      # one bucket without windows
    assert empty[8000][0].shape == (0, 192) and empty[32000][0].shape == (1, 192)
    with pytest.raises(SdkError, match="empty batch|null argument"):
        engine.fbank_windows(1, 100, 1, 0, 32000)
    short = _recording(1.2, 4)                                     # 19 200 samples < one 2-s window
    st, spans, W = wav.window_starts(len(short), None)
    assert st.tolist() == [0] and W == 32000
    got = engine.embed_from_host(short, {W: st})[W][0]
    want = engine.embed_pcm(torch.from_numpy(wav.materialise_windows(short, st, W)).cuda())[0]
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path / "store"))
    be = sub("backend").Backend()
    wav.write_wav_s16(tmp_path / "tiny.wav", _recording(0.3, 5))
    with pytest.raises(ValueError, match="no analysable audio"):
        be.enroll_speaker(tmp_path / "tiny.wav")
    long = _recording(40.0, 6)
    dense = np.arange(0, len(long) - 8000, 100, dtype=np.int32)     # 6320 windows of 0.5 s: more than the slots' initial 4096-entry table
    out = engine.embed_from_host(long, {8000: dense}, step=4096)[8000]
    assert out[0].shape == (len(dense), 192) and bool(torch.isfinite(out[0]).all())


def test_identify_many_equals_identify_speaker_per_recording(tmp_path, monkeypatch):
    """Backend.identify_many: several recordings against one candidate set in one pipelined pass (profiles uploaded once, recording i + 1 uploaded
    under recording i's forward, one host synchronisation): the row lists must equal identify_speaker's, recording by recording."""
    from test_gpu_backend_e2e import _voice
    monkeypatch.setenv("SPEAKERS_EMBEDDINGS_DIR", str(tmp_path / "store"))
    be = sub("backend").Backend()
    profiles = []
    for i, (sid, f0) in enumerate((("alice", 140.0), ("bob", 95.0), ("carol", 210.0))):
        p = tmp_path / f"enroll_{sid}.wav"
        wav.write_wav_s16(p, _voice(10 + i, 6.0, f0))
        rec = be.enroll_speaker(p)
        profiles.append({"id": sid, "embeddings": {"mi355x": [{"id": f"emb-{sid}", "external_id": rec["external_id"], "model_version": rec["model_version"]}]}})
    paths = []
    for i, parts in enumerate(([(95.0, 4.0), (140.0, 4.0)], [(210.0, 7.5)], [(140.0, 3.0), (210.0, 3.0), (95.0, 3.0)], [(95.0, 2.2)], [(140.0, 9.0), (95.0, 5.0)])):
        rec = np.concatenate([_voice(50 + 7 * i + j, sec, f0) for j, (f0, sec) in enumerate(parts)])
        paths.append(tmp_path / f"meeting{i}.wav")
        wav.write_wav_s16(paths[-1], rec)
    many = be.identify_many(paths, profiles, threshold=-1.0)
    one_by_one = [be.identify_speaker(p, profiles, threshold=-1.0) for p in paths]
    assert many == one_by_one and len(many) == 5 and all(len(r) >= 1 for r in many)
    assert many[1][0]["speaker_id"] == "carol" and many[3][0]["speaker_id"] == "bob"
    assert be.identify_many([], profiles) == []


def test_ingest_slots_are_lazy_sized_and_fall_back_to_pageable_staging(engine, monkeypatch):
    """a single-recording call (base.py:130-151: one identify per CLI process) touches ONE slot, sized to that recording -
    not `depth` slots of the largest size; a longer recording grows the slot it lands in and nothing else; where page-locked memory is refused
    the slot stages through ordinary memory with the same results."""
    Ingest = sub("ingest").Ingest
    st = torch.cuda.current_stream().cuda_stream
    ing = Ingest(engine.lib, engine.ctx, depth=2)
    assert ing.slot_info(0) == (0, True) and ing.slot_info(1) == (0, True)            # nothing allocated at creation
    rec = _recording(70.0, 8)                                                         # 1 120 000 samples -> one 2^21-sample slot
    t, _, _ = ing.submit(rec, np.array([0], np.int32), 32000, st)
    ing.release(t, st)
    assert ing.slot_info(t) == (1 << 21, True) and ing.slot_info(1 - t)[0] == 0
    t2, _, _ = ing.submit(rec[:40000], np.array([0], np.int32), 32000, st)            # the other slot: sized to ITS upload
    ing.release(t2, st)
    assert t2 == 1 - t and ing.slot_info(t2)[0] == 1 << 20
    t3, ds, _ = ing.submit(np.tile(rec, 2), np.array([0], np.int32), 32000, st)       # slot t again, now too small: grown in place
    back = torch.empty(2 * len(rec), dtype=torch.int16, device="cuda")
    engine.lib.sdk_memcpy(engine.ctx, back.data_ptr(), ds, 4 * len(rec), 3, st)
    ing.release(t3, st)
    assert t3 == t and ing.slot_info(t)[0] == 3 << 20 and np.array_equal(back.cpu().numpy(), np.tile(rec, 2))
    ing.close()
    monkeypatch.setenv("SDK_INGEST_NO_PINNED", "1")                                   # what a host at its lock limit answers
    ing = Ingest(engine.lib, engine.ctx, depth=2)
    t, ds, dw = ing.submit(rec, np.array([0, 16000], np.int32), 32000, st)
    got = engine.fbank_windows(ds, len(rec), dw, 2, 32000)
    ing.release(t, st)
    want = engine.fbank(torch.from_numpy(wav.materialise_windows(rec, np.array([0, 16000], np.int32), 32000)).cuda())
    torch.cuda.synchronize()
    assert ing.slot_info(t) == (1 << 21, False) and torch.equal(got.view(torch.int16), want.view(torch.int16))
    ing.close()


def test_long_recording_goes_through_the_ring_in_pieces(engine, monkeypatch):
    """A recording longer than a staging slot may be ($SDK_INGEST_CHUNK) is uploaded piece by piece (ingest.plan_chunks): every window is
    computed from exactly the samples the one-piece form gives it - bit-identical per piece to embed_pcm on that piece's host-cut windows
    (a piece is its own batch), and equal to the one-piece result within the batch-invariance tolerance - and no slot exceeds the bound."""
    ingest = sub("ingest")
    rec = _recording(31.3, 12)
    s2, _, W = wav.window_starts(len(rec), None)
    s1 = np.arange(0, len(rec) - 100, 7000, dtype=np.int32)[::-1].copy()               # unsorted table, its last windows run past the end
    one = engine.embed_from_host(rec, {W: s2, 16000: s1})
    monkeypatch.setenv("SDK_INGEST_CHUNK", str(1 << 17))                               # 8.2-s pieces
    pieces = ingest.plan_chunks(len(rec), {W: s2, 16000: s1}, ingest.chunk_samples())
    assert len(pieces) == 6 and all(hi - lo <= 1 << 17 for lo, hi, _ in pieces)
    engine._ingest = None                                                              # fresh ring: its slots must stay at the bound
    got = engine.embed_from_host(rec, {W: s2, 16000: s1})
    torch.cuda.synchronize()
    assert max(engine.ingest().slot_info(i)[0] for i in range(2)) == 1 << 20
    for S, table in ((W, s2), (16000, s1)):
        assert got[S][0].shape == one[S][0].shape and float((got[S][0] - one[S][0]).abs().max()) < 1e-3
        for lo, hi, sub_ in pieces:
            if S in sub_:
                rows, local = sub_[S]
                assert np.array_equal(table[rows], local + lo)
                want = engine.embed_pcm(torch.from_numpy(wav.materialise_windows(rec, table[rows], S)).cuda())
                assert torch.equal(got[S][0][torch.from_numpy(rows).cuda()], want[0])
