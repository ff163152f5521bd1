"""The whole sharded path in one call (BASELINE.json configs #2-#5), one process per GPU:

    local PCM windows --k1/k2/k3--> local embeddings --k4--> per-segment best profiles   (no collective:
                                                                                         profiles replicated)
                      --k5 all-gather (RCCL/xGMI)--> all embeddings --k6--> global cluster labels

`run_shard` is what a batch driver (the toolkit runs up to 4 `speaker-assign` processes at once,
speaker-process:627-629) would call once per GPU with its contiguous block of segments.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import cluster as scluster


@dataclass
class ShardResult:
    embeddings: torch.Tensor            # [n_local, 192] fp32 unit rows (device)
    best_profile: np.ndarray            # [n_local, k] int32 profile rows (-1 where below threshold for k = 1)
    best_score: np.ndarray              # [n_local, k] fp32 cosine
    cluster_labels: Optional[np.ndarray] = None      # [n_total] int32 canonical labels (same on every rank)
    eigenvalues: Optional[np.ndarray] = None


def run_shard(engine, pcm_local: torch.Tensor, profiles: torch.Tensor, n_total: Optional[int] = None, k: int = 1,
              threshold: Optional[float] = None, n_clusters: int = 0, cluster_iters: int = 30, group=None) -> ShardResult:
    """pcm_local [n_local, S] int16 on the engine's device (this rank's rows under dist.shard_bounds);
    profiles [P, 192] fp32 on the device (replicated).  n_clusters > 0 adds the global spectral clustering."""
    E, Eb, re = engine.embed_pcm(pcm_local)
    Pn, Pb, rp = engine.l2norm(profiles)
    idx, sc = engine.affinity_topk(E, Eb, re, Pn, Pb, rp.max().reshape(1), k=min(k, profiles.shape[0]))
    idx_h, sc_h = idx.cpu().numpy(), sc.cpu().numpy()
    if threshold is not None:
        idx_h = np.where(sc_h >= np.float32(threshold), idx_h, -1)
    out = ShardResult(E, idx_h, sc_h)
    if n_clusters > 0:
        n_total = n_total if n_total is not None else E.shape[0]
        res = scluster.spectral_cluster(engine, E, Eb, n_total, n_clusters, n_iter=cluster_iters, group=group)
        out.cluster_labels, out.eigenvalues = res.labels, res.eigenvalues
    return out
