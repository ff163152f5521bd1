"""k2 oracle, second model family: the x-vector (plain TDNN) forward on the CPU.

TEST INFRASTRUCTURE - see oracle/__init__.py.  **Parity unpinned** (the reference has no embedding model; backends.yaml:22-31).  Restates the
published architecture (Snyder et al. 2018): frame layers  conv (reflect "same") -> ReLU -> BatchNorm(eval)  with bf16 GEMM operands and
bf16 layer-boundary storage (mode "bf16"), the same with fp16 operands and storage (mode "fp16": the single-plane precision 2), or no
rounding (mode "fp32"), statistics pooling (mean | sqrt(max(var, 1e-12)) over frames, fp32
as the GPU sweep computes it from the stored tensor), embedding layer in fp32.
"""
from __future__ import annotations

import torch

from .ecapa import BN_EPS, _bf16, reflect_index


def xvector_embed(weights, feats, kernels=(5, 3, 3, 1, 1), dilations=(1, 2, 3, 1, 1), mode="bf16", acc=torch.float64):
    """feats [B, T, n_feats] fp32 -> [B, embed_dim] fp32."""
    assert mode in ("bf16", "fp16", "fp32")
    q = _bf16 if mode == "bf16" else (lambda t: t.half().float()) if mode == "fp16" else (lambda t: t)
    x = q(torch.as_tensor(feats, dtype=torch.float32))
    B, T, _ = x.shape
    t = torch.arange(T)
    for l, (k, dil) in enumerate(zip(kernels, dilations)):
        W = torch.from_numpy(weights[f"frame{l}.conv.w"])                     # [cout, cin, k]
        Wq = q(W).to(acc)
        xa = x.to(acc)
        out = torch.zeros(B, T, W.shape[0], dtype=acc)
        for j in range(k):
            src = reflect_index(t + (j - (k - 1) // 2) * dil, T)
            out += xa[:, src, :] @ Wq[:, :, j].T
        pre = out.float() + torch.from_numpy(weights[f"frame{l}.conv.b"])
        g, b, m, v = (torch.from_numpy(weights[f"frame{l}.bn.{f}"]).double() for f in ("gamma", "beta", "mean", "var"))
        s = g / torch.sqrt(v + BN_EPS)
        x = q(torch.relu(pre) * s.float() + (b - m * s).float())
    xd = x.double()
    mu = xd.mean(dim=1)
    sd = torch.sqrt(((xd - mu[:, None, :]) ** 2).mean(dim=1).clamp_min(1e-12))
    stats = torch.cat([mu, sd], dim=-1).float()
    return (stats.double() @ torch.from_numpy(weights["embed.w"]).double().T).float() + torch.from_numpy(weights["embed.b"])
