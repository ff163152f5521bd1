#!/usr/bin/env python3
"""Does the memory side deliver 128-byte pieces at a 6-KB stride as fast as a sequential sweep?  (the ASP hidden-layer GEMM streams its A operand
[201 000, 3072] bf16 as 128-byte row pieces and stays at 3.4 TB/s.)  Pure torch copies, checker only: (a) the whole matrix, contiguous; (b) one 64-column block
at a time (128-byte pieces, row stride 6144 B); (c) 128- and 512-column blocks."""
import json, time
import torch
M, K = 201_000, 3072
h = torch.randn(M, K, device="cuda").to(torch.bfloat16)
out = {}
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
dst = torch.empty_like(h)
t = timed(lambda: dst.copy_(h))
out["contiguous_copy"] = {"ms": round(t * 1e3, 3), "read_TBps": round(h.numel() * 2 / t / 1e12, 2)}
for cols in (64, 128, 512):
    blk = torch.empty(M, cols, device="cuda", dtype=torch.bfloat16)
    def sweep():
        for k0 in range(0, K, cols):
            blk.copy_(h[:, k0:k0 + cols])
    t = timed(sweep)
    out[f"column_blocks_of_{cols}"] = {"piece_bytes": cols * 2, "ms": round(t * 1e3, 3), "read_TBps": round(h.numel() * 2 / t / 1e12, 2)}
print(json.dumps(out))
