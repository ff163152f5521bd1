"""sdk_rows_fc at the step's four shapes: time per launch (HIP events of the library's profiler) + a digest of the output
(the kernel is a fixed-order fp32 chain: a restructured fetch pipeline must leave the digest unchanged)."""
import hashlib, importlib, sys, torch
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
g = torch.Generator(device="cuda").manual_seed(0)
for B, Cin, Nout, act, sc in [(1000, 6144, 128, 0, False), (1000, 6144, 192, 0, True), (1000, 1024, 128, 1, False), (1000, 128, 1024, 2, False), (1000, 3000, 512, 0, False)]:
    x = torch.randn(B, Cin, device="cuda", generator=g)
    wt = torch.randn(Cin, Nout, device="cuda", generator=g) / Cin ** 0.5
    b = torch.randn(Nout, device="cuda", generator=g)
    isc = torch.rand(Cin, device="cuda", generator=g) + 0.5 if sc else None
    ish = torch.randn(Cin, device="cuda", generator=g) if sc else None
    for _ in range(3): out = eng.rows_fc(x, wt, b, isc, ish, act)
    eng.profile_begin()
    for _ in range(20): out = eng.rows_fc(x, wt, b, isc, ish, act)
    p = eng.profile_end()
    torch.cuda.synchronize()
    print((B, Cin, Nout, act, sc), {k: round(v["ms"] / 20 * 1e3, 2) for k, v in p.items()}, "us", hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
