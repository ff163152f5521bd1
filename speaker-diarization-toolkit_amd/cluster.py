"""Global segment clustering (BASELINE.json config #5; SURVEY.md §8 k5 + k6): spectral clustering of
N L2-normalised embeddings on the rectified cosine affinity, row-sharded over the ranks.

    E_all   = all_gather(E_local)                                  the ONE big exchange (RCCL over xGMI)
    deg     = A 1 ; S = D^-1/2 A D^-1/2                            A = max(E E^T, 0), recomputed per use
    V       = orth(seeded gaussian [N, k])                         CholeskyQR (Gram all-reduce, k x k)
    repeat  : V <- orth(S V)                                       subspace iteration; per step one
                                                                   all-gather of V [N, k] (6.4 MB at 100k x 16)
    Ritz    : H = V^T S V, eigh (k x k, host), U = V Q
    k-means : rows of U normalised; maximin init; Lloyd with centroid all-reduce
    labels  : canonical (order of first appearance over the global row order)

Every O(N^2) and O(N k^2) operation runs in libsdk_hip.so (ops.Engine), and so does the k x k Cholesky + triangular inverse
of every CholeskyQR pass (sdk_chol_inverse, float64): the subspace iteration, the maximin initialisation and the Lloyd passes never
synchronise with the host.  The one k x k eigh of the Ritz step runs on the host in float64, once.  The algorithm is restated for the CPU in
oracle/spectral.py; free choices (rectification, init, iteration counts) are identical.

`provider` is the object that executes the row-block primitives: an ops.Engine on a GPU.  The
distributed control flow itself is device-agnostic, so tests/test_dist_gloo.py drives it on CPU
ranks with an oracle-backed provider.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import dist as sdist


@dataclass
class SpectralResult:
    labels: np.ndarray            # [N] int32 canonical, identical on every rank
    eigenvalues: np.ndarray       # [k] descending (of S = D^-1/2 A D^-1/2)
    n_iter: int
    timing: Optional[dict] = None  # seconds per phase when spectral_cluster(..., trace=True) (each phase then ends with a device sync)
    rows: Optional[np.ndarray] = None  # spectral_cluster(..., keep_rows=True): the [N, k] unit rows of the Ritz vectors that k-means clustered (all ranks' rows)
    retried: bool = False          # the subspace iteration was repeated with shifted CholeskyQR (a pivot fell under the relative-pivot rule)


class _Comm:
    def __init__(self, group=None):
        import torch.distributed as d
        self.on = d.is_available() and d.is_initialized() and d.get_world_size(group) > 1
        self.group = group
        self.rank = d.get_rank(group) if self.on else 0
        self.world = d.get_world_size(group) if self.on else 1

    def gather_rows(self, local: torch.Tensor, n_total: int) -> torch.Tensor:
        return sdist.all_gather_rows(local, n_total, self.group) if self.on else local

    def sum_(self, x: torch.Tensor) -> torch.Tensor:
        return sdist.all_reduce_sum(x, self.group) if self.on else x


def canonical_labels(lab: np.ndarray) -> np.ndarray:
    lab = np.asarray(lab)
    _, first = np.unique(lab, return_index=True)
    remap = np.full(int(lab.max()) + 1, -1, dtype=np.int32)
    for new, old in enumerate(lab[np.sort(first)]):
        remap[int(old)] = new
    return remap[lab]


def spectral_cluster(provider, E_local: torch.Tensor, Eb_local: torch.Tensor, n_total: int, k: int, n_iter: int = 30,
                     n_kmeans: int = 20, seed: int = 0, group=None, trace: bool = False, keep_rows: bool = False) -> SpectralResult:
    """E_local / Eb_local: this rank's unit-norm embedding rows (fp32 / bf16) under
    dist.shard_bounds(n_total, world).  Returns identical results on every rank."""
    import time
    timing = {} if trace else None
    t_last = [time.perf_counter()]

    def mark(name):
        if trace:
            if Eb_local.is_cuda:
                torch.cuda.synchronize()
            now = time.perf_counter()
            timing[name] = timing.get(name, 0.0) + now - t_last[0]
            t_last[0] = now
    comm = _Comm(group)
    lo, hi = sdist.shard_bounds(n_total, comm.world)[comm.rank]
    n_loc = hi - lo
    dev = Eb_local.device
    assert Eb_local.shape[0] == n_loc, (Eb_local.shape, lo, hi)

    Eb_all = comm.gather_rows(Eb_local.contiguous(), n_total)                      # k5: the embedding all-gather
    ones = torch.ones((n_total, 1), dtype=torch.float32, device=dev)
    deg = provider.affinity_matvec(Eb_all, ones, lo, n_loc)[lo:hi, 0].contiguous()
    dinv_loc = torch.rsqrt(deg)
    dinv_all = comm.gather_rows(dinv_loc.reshape(-1, 1), n_total).reshape(-1).contiguous()
    mark("gather+degrees")

    # seeded start, same gaussian as the oracle (numpy PCG64), orthonormalised across ranks
    G0 = np.random.default_rng(seed).standard_normal((n_total, k))
    V = torch.from_numpy(G0[lo:hi].astype(np.float32)).to(dev)
    # not-positive-definite flag of every CholeskyQR pass (sticky, device side): read once, at the Ritz step's host synchronisation
    spd_flag = torch.zeros((1,), dtype=torch.int32, device=dev)
    V0 = V
    eye = torch.eye(k, dtype=torch.float32, device=dev)

    def apply_S(Vloc: torch.Tensor) -> torch.Tensor:
        Vall = comm.gather_rows(Vloc, n_total)
        Y = provider.affinity_matvec(Eb_all, Vall, lo, n_loc, xscale=dinv_all)[lo:hi].contiguous()
        return provider.rows_apply(Y, eye, scale=dinv_loc)

    def iterate(shifted: bool):
        Vc = _orth(provider, comm, V0, k, spd_flag, shifted)
        mark("init")
        for _ in range(n_iter):
            Y = apply_S(Vc)
            mark("apply_S")
            Vc = _orth(provider, comm, Y, k, spd_flag, shifted)
            mark("orth")
        SVc = apply_S(Vc)
        mark("apply_S")
        return Vc, comm.sum_(provider.rows_gram(Vc, SVc)).double().cpu().numpy()

    retried = False
    V, H = iterate(False)
    flagged = int(comm.sum_(spd_flag.float()).item()) > 0                # every rank takes the same branch (the Gram matrices are all-reduced: normally equal anyway)
    if flagged and np.isfinite(H).all() and hasattr(provider, "set_option"):
        # A pivot fell under the relative-pivot rule (sdk_chol_inverse: cond(Y) > ~1e3): over-clustered or near-duplicate input - lambda_k / lambda_1
        # near 1e-3 - where plain CholeskyQR2 is at the edge of what an fp32 Gram matrix carries.  One retry with SHIFTED CholeskyQR
        # (first pass on G + 1e-5 mean(diag) I, then two plain passes: "CholeskyQR3"), judged by the same rule on its plain passes.
        spd_flag.zero_()
        retried = True
        V, H = iterate(True)
        flagged = int(comm.sum_(spd_flag.float()).item()) > 0
    if flagged or not np.isfinite(H).all():
        # a Gram matrix of the iteration was not positive definite: k exceeds the rank of the embeddings (duplicated rows, fewer
        # distinct segments than clusters) or the input holds NaN.  The host-side CholeskyQR of round 1 raised here too.
        raise np.linalg.LinAlgError(f"spectral_cluster: the k = {k} subspace lost rank (Gram matrix not positive definite): "
                                    "fewer than k independent directions in the embeddings, or non-finite input")
    H = 0.5 * (H + H.T)
    lam, Q = np.linalg.eigh(H)
    order = np.argsort(-lam)
    Qd = torch.from_numpy(np.ascontiguousarray(Q[:, order]).astype(np.float32)).to(dev)
    U = provider.rows_apply(V, Qd)
    R = provider.rows_unit(U)
    mark("ritz")

    labels_loc = _kmeans(provider, comm, R, lo, n_total, k, n_kmeans)
    mark("kmeans")
    lab_all = comm.gather_rows(labels_loc.reshape(-1, 1), n_total).reshape(-1)
    labels = canonical_labels(lab_all.cpu().numpy())
    mark("labels")
    rows = comm.gather_rows(R, n_total).cpu().numpy() if keep_rows else None
    return SpectralResult(labels, lam[order], n_iter, timing, rows, retried)


def _orth(provider, comm: _Comm, Y: torch.Tensor, k: int, spd_flag: Optional[torch.Tensor] = None, shifted: bool = False) -> torch.Tensor:
    """CholeskyQR2: Q = Y R^-1 with R^T R = sum_ranks Y^T Y; applied twice for fp32 stability.  Nothing here synchronises with the
    host: Gram (two-stage, order-fixed) -> all-reduce -> k x k Cholesky + triangular inverse in float64 on the device -> apply.
    shifted: a first pass on G + 1e-5 mean(diag G) I (not judged by the pivot rule), then the two plain passes."""
    if shifted:
        provider.set_option("chol_shift_ppb", 10_000)
        try:
            G = comm.sum_(provider.rows_gram(Y, Y))
            Y = provider.rows_apply(Y, provider.chol_inverse(G, None))
        finally:
            provider.set_option("chol_shift_ppb", 0)
    for _ in range(2):
        G = comm.sum_(provider.rows_gram(Y, Y))
        Y = provider.rows_apply(Y, provider.chol_inverse(G, spd_flag))
    return Y


def _kmeans(provider, comm: _Comm, R: torch.Tensor, lo: int, n_total: int, k: int, n_iter: int) -> torch.Tensor:
    """Maximin initialisation + Lloyd, row-sharded.  Device-side selection throughout (no .item(), no host comparison): at 8 GPUs
    every host round trip would also be a collective everyone waits in."""
    dev = R.device
    n_loc = R.shape[0]
    kdim = R.shape[1]

    def fetch_row(grow: torch.Tensor) -> torch.Tensor:
        """The row with this GLOBAL index (0-d int64 tensor on the device), on every rank: the owner contributes it, the others
        zeros; summed."""
        if n_loc:
            loc = (grow - lo).clamp(0, n_loc - 1).reshape(1)
            mine = ((grow >= lo) & (grow < lo + n_loc)).to(torch.float32)
            v = R.index_select(0, loc)[0] * mine
        else:
            v = torch.zeros((kdim,), dtype=torch.float32, device=dev)
        return comm.sum_(v.contiguous())

    def global_argmax(val: torch.Tensor, grow: torch.Tensor) -> torch.Tensor:
        """Global row index of the largest value over the ranks (ties -> lowest row), as a device tensor."""
        if not comm.on:
            return grow
        pair = torch.stack([val.double(), grow.double()])
        out = torch.zeros((comm.world * 2,), dtype=torch.float64, device=dev)
        sdist._gather_into(out, pair, comm.group)
        pairs = out.reshape(comm.world, 2)
        best = pairs[:, 0].max()
        rows = torch.where(pairs[:, 0] == best, pairs[:, 1], torch.full_like(pairs[:, 1], float(n_total)))
        return rows.min().to(torch.int64)

    centres = [fetch_row(torch.zeros((), dtype=torch.int64, device=dev))]
    d2 = torch.empty((max(n_loc, 1),), dtype=torch.float32, device=dev)[:n_loc]
    for j in range(1, k):
        if n_loc:
            provider.kmeans_mindist(R, centres[-1], d2, first=(j == 1))
            val, arg = torch.max(d2, dim=0)                 # first maximum = lowest local row
            grow = arg.to(torch.int64) + lo
        else:
            val = torch.full((), -1.0, dtype=torch.float32, device=dev)
            grow = torch.full((), n_total, dtype=torch.int64, device=dev)
        centres.append(fetch_row(global_argmax(val, grow)))
    C = torch.stack(centres).contiguous()
    # Lloyd: a fixed number of passes.  Once the centres stop moving a pass reproduces them bit for bit (same sums, same counts),
    # so running past convergence changes nothing - and saves the host comparison every pass used to end with.
    for _ in range(n_iter):
        _, _, ps, pc = provider.kmeans_assign(R, C, want_sums=True)
        sums = comm.sum_(ps.double().sum(dim=0))
        cnts = comm.sum_(pc.sum(dim=0).double())
        C = torch.where(cnts[:, None] > 0, sums / cnts.clamp_min(1.0)[:, None], C.double()).float().contiguous()
    labels, _, _, _ = provider.kmeans_assign(R, C, want_sums=False)
    return labels
