"""k4 oracle: segments x profiles cosine affinity, top-k / argmax, threshold assignment.

TEST INFRASTRUCTURE - see oracle/__init__.py.  **Parity unpinned**: the reference's
`identify` path has no local scoring (speechmatics_backend.py:361-489 asks a cloud API);
what IS pinned is how the resulting rows are consumed (speaker_detection:1085-1127,
speaker-assign:296-322) - covered by tests/golden.  The arithmetic here is plain cosine
scoring on L2-normalised embeddings:

    S[n,p]   = sum_d E[n,d] * P[p,d]          (fp32 inputs, float64 accumulate)
    idx[n,:] = the k largest p by S[n,p], ties -> lowest p first
    assign   = idx[n,0] if S[n,idx[n,0]] >= threshold else -1
"""
from __future__ import annotations

import numpy as np


def affinity(E: np.ndarray, P: np.ndarray) -> np.ndarray:
    """Full [N,P] cosine score matrix (float64 accumulate, returned as float32)."""
    return (np.asarray(E, np.float64) @ np.asarray(P, np.float64).T).astype(np.float32)


def affinity_topk(E: np.ndarray, P: np.ndarray, k: int = 1, chunk: int = 8192):
    """Return (idx [N,k] int32, score [N,k] float32), sorted by descending score per row;
    ties broken by lowest profile index (stable).  float64 accumulate."""
    E = np.asarray(E, np.float32)
    P64 = np.asarray(P, np.float64)
    N = E.shape[0]
    k = min(k, P64.shape[0])
    idx = np.empty((N, k), np.int32)
    sc = np.empty((N, k), np.float32)
    for s in range(0, N, chunk):
        S = E[s:s + chunk].astype(np.float64) @ P64.T
        order = np.argsort(-S, axis=1, kind="stable")[:, :k]
        idx[s:s + chunk] = order
        sc[s:s + chunk] = np.take_along_axis(S, order, axis=1).astype(np.float32)
    return idx, sc


def assign(E: np.ndarray, P: np.ndarray, threshold: float):
    idx, sc = affinity_topk(E, P, 1)
    best = idx[:, 0].copy()
    best[sc[:, 0] < np.float32(threshold)] = -1
    return best, sc[:, 0]


def affinity_topk_fp32(E: np.ndarray, P: np.ndarray, k: int = 1):
    """Timed CPU baseline flavour (torch fp32 matmul + topk on all host cores)."""
    import torch
    S = torch.from_numpy(np.asarray(E, np.float32)) @ torch.from_numpy(np.asarray(P, np.float32)).T
    sc, idx = torch.topk(S, k, dim=1)
    return idx.numpy().astype(np.int32), sc.numpy()
