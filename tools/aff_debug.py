#!/usr/bin/env python3
"""Debug aid for the k = 1 row/column-maxima path: decodes the coarse records of rows whose result differs from an fp64 scan."""
import importlib, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("speaker-diarization-toolkit_amd.ops")
eng = ops.get_engine(0)
N, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2000, 10000)
MAXP, SEGS = 3, 512
g = torch.Generator(device="cuda").manual_seed(11)
E, Eb, re = eng.l2norm(torch.randn(N, 192, device="cuda", generator=g))
Q, Qb, rq = eng.l2norm(torch.randn(P, 192, device="cuda", generator=g))
eng.set_option("affinity_fast_path", 1); eng.set_option("affinity_variant", 0)
idx, sc, cnt = eng.affinity_topk(E, Eb, re, Q, Qb, rq.max().reshape(1), k=1, want_count=True)
torch.cuda.synchronize()
ws = [v for k, v in eng._scratch.items() if k.startswith("affinity")][0]
al = lambda v: (v + 255) & ~255
o_stats = 0; o_pb = al(N * MAXP * 64); ng = (N + 255) // 256
o_pc = o_pb + al(ng * MAXP * 4); o_fc = o_pc + al(ng * 4); o_fr = o_fc + 256
stats = ws[o_stats:o_stats + N * MAXP * 64].view(torch.float32).view(N, MAXP, 2, 8).cpu().numpy()
pbase = ws[o_pb:o_pb + ng * MAXP * 4].view(torch.int32).view(ng, MAXP).cpu().numpy()
pcnt = ws[o_pc:o_pc + ng * 4].view(torch.int32).cpu().numpy()
fc = ws[o_fc:o_fc + 8].view(torch.int32).cpu().numpy()
frows = set(ws[o_fr:o_fr + 4 * int(fc[0])].view(torch.int32).cpu().numpy().tolist())
print("flag_count", fc, "cnt", int(cnt), "part_cnt", pcnt[:(N + SEGS - 1) // SEGS], "part_base", pbase[:(N + SEGS - 1) // SEGS].tolist())
X = (E.double() @ Q.double().t())
S = (Eb.float() @ Qb.float().t())                       # coarse model
top = X.max(1)
bad = torch.nonzero((top.values - X.gather(1, idx.long())[:, 0]) > 1e-6)[:, 0].cpu().numpy()
print("rows with a non-maximal winner:", len(bad), "of which flagged:", sum(int(b) in frows for b in bad))
Sn = S.cpu().numpy()
for n in bad[:6]:
    grp = n // SEGS
    print(f"row {n}: got idx {int(idx[n])} score {float(sc[n]):.7f}; true argmax {int(top.indices[n])} {float(top.values[n]):.7f}; flagged {int(n) in frows}")
    tp = int(top.indices[n]); tile = tp // 32; r_in = tp % 32; hh = (r_in >> 2) & 1; reg = (r_in & 3) + 4 * (r_in >> 3)
    print(f"   true best sits in tile {tile}, half {hh}, register {reg}; coarse {Sn[n, tp]:.6f}; coarse of returned {Sn[n, int(idx[n])]:.6f}")
    for p in range(int(pcnt[grp])):
        for h in range(2):
            rec = stats[n, p, h]; u = rec.view(np.uint32)
            T = [(float((u[q] & ~np.uint32(0x3ff)).view(np.float32)) if False else float(np.array(u[q] & 0xfffffc00, np.uint32).view(np.float32)), int(u[q] & 0x3ff) + int(pbase[grp, p])) for q in range(4)]
            Cc = [(float(np.array(u[4 + q] & 0xfffffff0, np.uint32).view(np.float32)), int(u[4 + q] & 0xf)) for q in range(4)]
            # model of the same record
            t0 = int(pbase[grp, p]); t1 = int(pbase[grp, p + 1]) if p + 1 < int(pcnt[grp]) else (P + 31) // 32
            pad = np.full(((t1 - t0) * 32,), -4.0, np.float32); seg = Sn[n, t0 * 32:min(P, t1 * 32)]; pad[:len(seg)] = seg
            A = pad.reshape(t1 - t0, 32)[:, [(r & 3) + 8 * (r >> 2) + 4 * h for r in range(16)]]
            Tm = np.sort(A.max(1))[::-1][:4]; Cm = np.sort(A.max(0))[::-1][:4]
            print(f"   part {p} half {h}: T {[(round(v, 5), t) for v, t in T]} model {np.round(Tm, 5).tolist()} argT {np.argsort(-A.max(1))[:4] + t0}")
            print(f"                  C {[(round(v, 5), r) for v, r in Cc]} model {np.round(Cm, 5).tolist()} argC {np.argsort(-A.max(0))[:4]}")
