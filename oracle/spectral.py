"""k6 oracle: segment-segment affinity -> normalised Laplacian -> subspace iteration -> k-means.

TEST INFRASTRUCTURE - see oracle/__init__.py.  **Parity unpinned**: the reference has no
clustering code at all; BASELINE.json config #5 names "spectral clustering
(Laplacian + power-iteration top-k eigvecs)".  The algorithm is the textbook one
(Ng-Jordan-Weiss), with every free choice fixed here so CPU and GPU can agree:

    A[i,j]  = max(0, <e_i, e_j>)                      rectified cosine affinity (diag = 1)
    deg[i]  = sum_j A[i,j];   S = D^-1/2 A D^-1/2     (L_sym = I - S; top-k of S = bottom-k of L)
    V0      = orth(G),  G ~ N(0,1) from numpy default_rng(seed), shape [N,k]
    repeat n_iter:  V = orth(S V)                     block power / subspace iteration
    Rayleigh-Ritz:  H = V^T S V,  H = Q diag(lam) Q^T (descending),  U = V Q
    rows:   R[i] = U[i] / ||U[i]||
    k-means on R: maximin init (first centre = row 0, next = farthest from chosen set),
            Lloyd for n_kmeans iterations, empty cluster keeps its centre
    labels relabelled in order of first appearance  -> canonical integer IDs

tests cross-check eigenvalues against scipy.linalg.eigh and labels against
sklearn.cluster.KMeans / ground truth on small N.
"""
from __future__ import annotations

import numpy as np


def rectified_affinity(E: np.ndarray) -> np.ndarray:
    E = np.asarray(E, np.float64)
    return np.maximum(E @ E.T, 0.0)


def canonical_labels(lab: np.ndarray) -> np.ndarray:
    lab = np.asarray(lab)
    _, first = np.unique(lab, return_index=True)
    order = lab[np.sort(first)]
    remap = {int(c): i for i, c in enumerate(order)}
    return np.array([remap[int(c)] for c in lab], dtype=np.int32)


def degrees(E: np.ndarray, block: int = 4096) -> np.ndarray:
    E = np.asarray(E, np.float64)
    N = E.shape[0]
    deg = np.zeros(N)
    for s in range(0, N, block):
        deg[s:s + block] = np.maximum(E[s:s + block] @ E.T, 0.0).sum(axis=1)
    return deg


def apply_S(E: np.ndarray, dinv: np.ndarray, V: np.ndarray, block: int = 4096) -> np.ndarray:
    """Y = D^-1/2 A D^-1/2 V with A recomputed tile by tile (never stored)."""
    E = np.asarray(E, np.float64)
    X = V * dinv[:, None]
    Y = np.empty_like(V)
    for s in range(0, E.shape[0], block):
        Y[s:s + block] = np.maximum(E[s:s + block] @ E.T, 0.0) @ X
    return Y * dinv[:, None]


def init_subspace(N: int, k: int, seed: int) -> np.ndarray:
    G = np.random.default_rng(seed).standard_normal((N, k))
    Q, _ = np.linalg.qr(G)
    return Q


def subspace_iteration(E: np.ndarray, k: int, n_iter: int = 30, seed: int = 0):
    """Return (eigenvalues desc [k], eigenvectors U [N,k]) of S = D^-1/2 A D^-1/2."""
    N = E.shape[0]
    dinv = 1.0 / np.sqrt(degrees(E))
    V = init_subspace(N, k, seed)
    for _ in range(n_iter):
        V, _ = np.linalg.qr(apply_S(E, dinv, V))
    H = V.T @ apply_S(E, dinv, V)
    H = 0.5 * (H + H.T)
    lam, Q = np.linalg.eigh(H)
    order = np.argsort(-lam)
    return lam[order], V @ Q[:, order]


def row_normalise(U: np.ndarray) -> np.ndarray:
    n = np.sqrt((U * U).sum(axis=1, keepdims=True))
    return U / np.maximum(n, 1e-12)


def kmeans_maximin(R: np.ndarray, k: int, n_iter: int = 20):
    """Deterministic k-means: maximin initialisation + Lloyd; returns (labels, centres)."""
    R = np.asarray(R, np.float64)
    centres = [R[0]]
    d2 = ((R - centres[0]) ** 2).sum(axis=1)
    for _ in range(1, k):
        j = int(np.argmax(d2))
        centres.append(R[j])
        d2 = np.minimum(d2, ((R - R[j]) ** 2).sum(axis=1))
    C = np.stack(centres)
    lab = np.zeros(R.shape[0], dtype=np.int64)
    for _ in range(n_iter):
        dist = ((R * R).sum(1)[:, None] - 2.0 * R @ C.T + (C * C).sum(1)[None, :])
        lab = np.argmin(dist, axis=1)
        for c in range(k):
            m = lab == c
            if m.any():
                C[c] = R[m].mean(axis=0)
    return lab, C


def spectral_cluster(E: np.ndarray, k: int, n_iter: int = 30, n_kmeans: int = 20, seed: int = 0):
    lam, U = subspace_iteration(E, k, n_iter, seed)
    lab, _ = kmeans_maximin(row_normalise(U), k, n_kmeans)
    return canonical_labels(lab), lam


def adjusted_rand_index(a: np.ndarray, b: np.ndarray) -> float:
    a = np.asarray(a); b = np.asarray(b)
    ua, ia = np.unique(a, return_inverse=True)
    ub, ib = np.unique(b, return_inverse=True)
    M = np.zeros((len(ua), len(ub)), dtype=np.int64)
    np.add.at(M, (ia, ib), 1)
    c2 = lambda x: x * (x - 1) / 2.0
    sij = c2(M).sum(); sa = c2(M.sum(1)).sum(); sb = c2(M.sum(0)).sum(); tot = c2(len(a))
    exp = sa * sb / tot
    mx = 0.5 * (sa + sb)
    return float((sij - exp) / (mx - exp)) if mx != exp else 1.0


def vmf_mixture(N: int, d: int, k: int, seed: int, noise: float = 0.35):
    """Synthetic config-#5 input: k seeded unit centroids + gaussian noise, re-normalised.
    Returns (E [N,d] float32 unit rows, labels [N] int32 canonical)."""
    rng = np.random.default_rng(seed)
    C = rng.standard_normal((k, d)); C /= np.linalg.norm(C, axis=1, keepdims=True)
    lab = rng.integers(0, k, size=N)
    X = C[lab] + (noise / np.sqrt(d)) * rng.standard_normal((N, d))   # |noise vector| ~ noise
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    return X.astype(np.float32), canonical_labels(lab)
