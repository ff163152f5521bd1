"""k2/k3 oracle: ECAPA-TDNN (C=1024) forward + L2-normalise on the CPU.

TEST INFRASTRUCTURE - see oracle/__init__.py.  **Parity unpinned** (no reference
implementation exists).  Restates the published architecture
(Desplanques et al. 2020; layer table in SURVEY.md Appendix B) in channel-last form
[B, T, C], each conv as "gather shifted frames, then matmul", so that the structure is
the one the HIP kernels implement and not torch's conv1d (tests cross-check it against
torch.nn.functional.conv1d with reflect padding).

``mode``:
  "bf16"  the bf16 layer-boundary model the GPU implements: GEMM operands rounded to
          bf16 (RNE), accumulation in ``acc`` (float64 = ideal), fp32 epilogue, bf16 store.
  "fp32"  no rounding anywhere (the mathematical model; used to report how far the bf16
          model is from it, and as the timed CPU baseline with acc=float32).

``sites`` (optional) overrides the mode with PER-SITE rounding switches, for the error budget of
DESIGN.md section 3 (tools/error_budget.py, tests/test_error_budget_cpu.py): a dict
{site: significand bits} (8 = bf16, 11 = fp16's, 16 = a bf16 hi+lo pair, 22 = an fp16 hi+lo pair)
or an iterable of site names (bf16).  Sites = ROUNDING_SITES below: every place where the GPU
path rounds a value that the fp32 model does not.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

BN_EPS = 1e-5
STD_EPS = 1e-12


def _bf16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


# every place where the bf16 layer-boundary model rounds
WEIGHT_GROUPS = ("blk0", "tdnn1", "res2net", "tdnn2", "mfa", "asp")       # "w:<group>" overrides "w" for that group of layers
ROUNDING_SITES = (
    "w",            # weights of every MFMA GEMM (blk0, tdnn1, Res2Net convs, tdnn2, MFA, attention hidden, attention logits)
    "feats",        # the normalised log-mel features as the first conv's A operand
    "blk0",         # stored output of the first TDNN layer
    "tdnn1",        # stored output u of a block's first 1x1 layer
    "res2net",      # Res2Net: stored conv outputs y_c and the running sums bf16(u_c + y_{c-1})
    "tdnn2",        # stored output z of a block's second 1x1 layer
    "se_out",       # stored block output bf16(g * z + x)
    "mfa",          # stored output h of the 3072 x 3072 layer
    "attn_hidden",  # stored tanh output of the attention hidden layer
)


def round_significand(x: torch.Tensor, bits: int) -> torch.Tensor:
    """Round fp32 values to `bits` significand bits (implicit bit included), nearest-even, exponent range of fp32 kept:
    8 = bfloat16; 16 models a bf16 hi+lo operand pair, 22 an fp16 hi+lo pair; >= 24 is the identity."""
    if bits >= 24:
        return x
    if bits == 8:
        return _bf16(x)
    drop = 24 - bits
    u = x.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    r = (u + ((1 << (drop - 1)) - 1) + ((u >> drop) & 1)) >> drop << drop
    r = torch.where(r >= (1 << 31), r - (1 << 32), r).to(torch.int32)
    return r.view(torch.float32).reshape(x.shape)


def reflect_index(t: torch.Tensor, T: int) -> torch.Tensor:
    """Reflect (no edge repeat) an index into [0, T): -1 -> 1, T -> T-2."""
    t = torch.where(t < 0, -t, t)
    return torch.where(t >= T, 2 * (T - 1) - t, t)


class EcapaOracle:
    def __init__(self, weights: Dict[str, np.ndarray], mode: str = "bf16", acc=torch.float64,
                 n_dilations=(2, 3, 4), scale: int = 8, sites=None):
        assert mode in ("bf16", "fp32")
        self.mode = mode
        self.acc = acc
        self.dil = tuple(n_dilations)
        self.scale = scale
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}
        if sites is None:
            sites = {s: 8 for s in ROUNDING_SITES} if mode == "bf16" else {}
        elif not isinstance(sites, dict):
            sites = {s: 8 for s in sites}
        unknown = set(sites) - set(ROUNDING_SITES) - {f"w:{g}" for g in WEIGHT_GROUPS}
        assert not unknown, f"unknown rounding sites {sorted(unknown)}"
        self.sites = dict(sites)

    # -- helpers ---------------------------------------------------------------------
    def q(self, x: torch.Tensor, site: str) -> torch.Tensor:
        """Round at `site` (a member of ROUNDING_SITES) to that site's significand width; identity where the site is off."""
        bits = self.sites.get(site)
        return x if bits is None else round_significand(x, bits)

    def qw(self, W: torch.Tensor, group: str) -> torch.Tensor:
        """Round the weights of one layer group: "w:<group>" if given (24 = exact), else "w"."""
        bits = self.sites.get(f"w:{group}", self.sites.get("w"))
        return W if bits is None else round_significand(W, bits)

    @staticmethod
    def weight_group(name: str) -> str:
        if "res2net" in name:
            return "res2net"
        return "blk0" if name.startswith("blk0") else name.split(".")[-2]      # blk1.tdnn1.conv -> tdnn1, mfa.conv -> mfa

    def bn(self, name: str):
        g, b, m, v = (self.w[f"{name}.{f}"].double() for f in ("gamma", "beta", "mean", "var"))
        s = g / torch.sqrt(v + BN_EPS)
        return s.float(), (b - m * s).float()

    def conv(self, x: torch.Tensor, name: str, dilation: int = 1) -> torch.Tensor:
        """x [B,T,Cin] (already boundary-rounded) -> fp32 pre-activation [B,T,Cout] incl. bias."""
        W = self.w[f"{name}.w"]                      # [Cout, Cin, k]
        Cout, Cin, k = W.shape
        B, T, _ = x.shape
        Wq = self.qw(W, self.weight_group(name)).to(self.acc)
        xa = x.to(self.acc)
        out = torch.zeros(B, T, Cout, dtype=self.acc)
        t = torch.arange(T)
        for j in range(k):
            src = reflect_index(t + (j - (k - 1) // 2) * dilation, T)
            out += xa[:, src, :] @ Wq[:, :, j].T
        return out.float() + self.w[f"{name}.b"]

    def tdnn(self, x, name, dilation=1):
        """conv -> ReLU -> BN(eval), fp32 epilogue, no final rounding."""
        s, sh = self.bn(f"{name}.bn")
        return torch.relu(self.conv(x, f"{name}.conv", dilation)) * s + sh

    # -- blocks ------------------------------------------------------------------------
    def se_res2net(self, x: torch.Tensor, i: int) -> torch.Tensor:
        d = self.dil[i - 1]
        u = self.q(self.tdnn(x, f"blk{i}.tdnn1"), "tdnn1")
        C = u.shape[-1]
        s = C // self.scale
        chunks = [u[..., c * s:(c + 1) * s] for c in range(self.scale)]
        ys = [chunks[0]]
        prev = None
        for c in range(1, self.scale):
            inp = chunks[c] if c == 1 else self.q(chunks[c] + prev, "res2net")
            prev = self.q(self.tdnn(inp, f"blk{i}.res2net.{c - 1}", d), "res2net")
            ys.append(prev)
        r = torch.cat(ys, dim=-1)
        z = self.q(self.tdnn(r, f"blk{i}.tdnn2"), "tdnn2")
        # squeeze-excitation (fp32, per utterance)
        mean = z.double().mean(dim=1).float()                               # [B, C]
        w1 = self.w[f"blk{i}.se.conv1.w"][:, :, 0]; b1 = self.w[f"blk{i}.se.conv1.b"]
        w2 = self.w[f"blk{i}.se.conv2.w"][:, :, 0]; b2 = self.w[f"blk{i}.se.conv2.b"]
        h = torch.relu((mean.double() @ w1.double().T).float() + b1)
        g = torch.sigmoid((h.double() @ w2.double().T).float() + b2)        # [B, C]
        return self.q(g[:, None, :] * z + x, "se_out")

    def forward_pooled(self, feats: torch.Tensor):
        """feats [B,T,80] fp32 -> (pooled [B,6144] fp32, intermediates dict)."""
        inter = {}
        x0 = self.q(feats, "feats")
        x = self.q(self.tdnn(x0, "blk0"), "blk0")
        inter["blk0"] = x
        outs = []
        for i in range(1, len(self.dil) + 1):
            x = self.se_res2net(x, i)
            inter[f"blk{i}"] = x
            outs.append(x)
        cat = torch.cat(outs, dim=-1)                                       # [B,T,3072]
        h = self.q(self.tdnn(cat, "mfa"), "mfa")
        inter["mfa"] = h
        # attentive statistics pooling with global context
        hd = h.double()
        mu = hd.mean(dim=1)
        sd = torch.sqrt(((hd - mu[:, None, :]) ** 2).mean(dim=1).clamp_min(STD_EPS))
        Wt = self.w["asp.tdnn.conv.w"][:, :, 0]                              # [128, 9216]
        Cm = h.shape[-1]
        Wh = self.qw(Wt[:, :Cm], "asp").to(self.acc)
        ctx = torch.cat([mu, sd], dim=-1).float()                           # [B, 6144] fp32
        ubias = (ctx.double() @ Wt[:, Cm:].double().T).float() + self.w["asp.tdnn.conv.b"]
        s, sh = self.bn("asp.tdnn.bn")
        pre = (h.to(self.acc) @ Wh.T).float() + ubias[:, None, :]
        a = self.q(torch.tanh(torch.relu(pre) * s + sh), "attn_hidden")    # [B,T,128]
        inter["attn_hidden"] = a
        W2 = self.qw(self.w["asp.conv.w"][:, :, 0], "asp").to(self.acc)      # [3072,128]
        logits = (a.to(self.acc) @ W2.T).float() + self.w["asp.conv.b"]     # [B,T,3072] fp32
        wgt = torch.softmax(logits.double(), dim=1)
        wmu = (wgt * hd).sum(dim=1)
        wsd = torch.sqrt((wgt * (hd - wmu[:, None, :]) ** 2).sum(dim=1).clamp_min(STD_EPS))
        pooled = torch.cat([wmu, wsd], dim=-1).float()
        inter["pooled"] = pooled
        return pooled, inter

    def embed(self, feats, return_intermediates: bool = False):
        """feats [B,T,80] -> raw 192-d embeddings fp32 (before L2-normalise)."""
        feats = torch.as_tensor(feats, dtype=torch.float32)
        pooled, inter = self.forward_pooled(feats)
        s, sh = self.bn("asp_bn")
        p = pooled * s + sh
        emb = (p.double() @ self.w["fc.w"][:, :, 0].double().T).float() + self.w["fc.b"]
        if return_intermediates:
            inter["emb"] = emb
            return emb, inter
        return emb


def l2_normalise(x) -> np.ndarray:
    """k3: row-wise x / max(||x||, 1e-12), norm accumulated in float64, result fp32."""
    x = np.asarray(x, dtype=np.float32)
    n = np.sqrt((x.astype(np.float64) ** 2).sum(axis=-1, keepdims=True))
    return (x / np.maximum(n, 1e-12)).astype(np.float32)


def to_bf16_f32(x: np.ndarray) -> np.ndarray:
    """Round fp32 -> bf16 (RNE) and widen back; numpy-only (bit arithmetic)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(x))
