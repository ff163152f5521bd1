"""TEST INFRASTRUCTURE: a CPU stand-in for ops.Engine's row-block primitives (same method names and
semantics, torch-CPU / numpy float64 inside), so that the DISTRIBUTED control flow of cluster.py can
be exercised by world_size-2 gloo processes on a machine without GPUs.  Never imported by the product."""
import numpy as np
import torch


class CpuProvider:
    def affinity_matvec(self, Eb, X, row0=0, rows=None, xscale=None, out=None):
        E = Eb.float().double()
        N = E.shape[0]
        rows = N - row0 if rows is None else rows
        Xs = X.double() * (xscale.double()[:, None] if xscale is not None else 1.0)
        Y = torch.zeros((N, X.shape[1]), dtype=torch.float32)
        Y[row0:row0 + rows] = (torch.clamp_min(E[row0:row0 + rows] @ E.T, 0.0) @ Xs).float()
        return Y

    def rows_gram(self, X, Y):
        return (X.double().T @ Y.double()).float()

    def rows_apply(self, X, R, scale=None):
        Z = X.double() @ R.double()
        if scale is not None:
            Z = Z * scale.double()[:, None]
        return Z.float()

    def chol_inverse(self, G, flag=None):       # raises LinAlgError itself; the GPU provider sets `flag` instead and cluster.py raises
        g = G.double().numpy()
        L = np.linalg.cholesky(0.5 * (g + g.T) + 1e-30 * np.eye(g.shape[0]))
        if not (np.diag(L) ** 2 > 1e-6 * np.diag(g)).all():            # same rank rule as chol_inverse_kernel (csrc/spectral.hip)
            raise np.linalg.LinAlgError("Gram matrix numerically rank deficient")
        return torch.from_numpy(np.ascontiguousarray(np.linalg.inv(L.T)).astype(np.float32))

    def rows_unit(self, X):
        n = X.double().norm(dim=1, keepdim=True).clamp_min(1e-12)
        return (X.double() / n).float()

    def kmeans_mindist(self, R, centre, d2, first):
        d = ((R.double() - centre.double()[None]) ** 2).sum(1).float()
        d2.copy_(d if first else torch.minimum(d2, d))
        return d2

    def kmeans_assign(self, R, centres, want_sums=True):
        dist = ((R.double()[:, None, :] - centres.double()[None]) ** 2).sum(-1)
        d2, lab = dist.min(dim=1)
        lab = lab.to(torch.int32)
        if not want_sums:
            return lab, d2.float(), None, None
        kc, k = centres.shape
        ps = torch.zeros((1, kc, k), dtype=torch.float32)
        pc = torch.zeros((1, kc), dtype=torch.int32)
        for q in range(kc):
            m = lab == q
            ps[0, q] = R[m].double().sum(0).float()
            pc[0, q] = int(m.sum())
        return lab, d2.float(), ps, pc
