"""An independently composed ECAPA-TDNN (torch.nn modules, channel-first, torch's own Conv1d / BatchNorm1d) used ONLY to
cross-check oracle/ecapa.py.  Built from the public layer table (Desplanques et al. 2020): it
shares nothing with the oracle but the weight dictionary's key names.  Not part of the product path."""
import torch
import torch.nn as nn


class TDNN(nn.Module):
    """Conv1d (reflect padding, 'same' length) -> ReLU -> BatchNorm1d."""

    def __init__(self, cin, cout, k, dilation=1):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, k, dilation=dilation, padding=dilation * (k - 1) // 2, padding_mode="reflect")
        self.bn = nn.BatchNorm1d(cout, eps=1e-5)

    def forward(self, x):
        return self.bn(torch.relu(self.conv(x)))


class SERes2Net(nn.Module):
    def __init__(self, c, scale, se, dilation):
        super().__init__()
        self.scale = scale
        self.tdnn1 = TDNN(c, c, 1)
        self.res2net = nn.ModuleList(TDNN(c // scale, c // scale, 3, dilation) for _ in range(scale - 1))
        self.tdnn2 = TDNN(c, c, 1)
        self.se1 = nn.Conv1d(c, se, 1)
        self.se2 = nn.Conv1d(se, c, 1)

    def forward(self, x):
        u = self.tdnn1(x)
        parts = list(torch.chunk(u, self.scale, dim=1))
        y = [parts[0]]
        for i, blk in enumerate(self.res2net):
            inp = parts[i + 1] if i == 0 else parts[i + 1] + y[-1]
            y.append(blk(inp))
        z = self.tdnn2(torch.cat(y, dim=1))
        g = torch.sigmoid(self.se2(torch.relu(self.se1(z.mean(dim=2, keepdim=True)))))
        return g * z + x


class EcapaNN(nn.Module):
    def __init__(self, n_mels=80, c=1024, scale=8, se=128, attn=128, mfa=3072, emb=192, dilations=(2, 3, 4), k0=5):
        super().__init__()
        self.blk0 = TDNN(n_mels, c, k0)
        self.blocks = nn.ModuleList(SERes2Net(c, scale, se, d) for d in dilations)
        self.mfa = TDNN(mfa, mfa, 1)
        self.asp_tdnn = TDNN(3 * mfa, attn, 1)
        self.asp_conv = nn.Conv1d(attn, mfa, 1)
        self.asp_bn = nn.BatchNorm1d(2 * mfa, eps=1e-5)
        self.fc = nn.Conv1d(2 * mfa, emb, 1)

    def forward(self, feats):                      # feats [B, T, n_mels]
        x = self.blk0(feats.transpose(1, 2))
        outs = []
        for b in self.blocks:
            x = b(x)
            outs.append(x)
        h = self.mfa(torch.cat(outs, dim=1))       # [B, 3072, T]
        T = h.shape[2]
        mu = h.mean(dim=2, keepdim=True)
        sd = torch.sqrt(((h - mu) ** 2).mean(dim=2, keepdim=True).clamp_min(1e-12))
        a = torch.tanh(self.asp_tdnn(torch.cat([h, mu.expand(-1, -1, T), sd.expand(-1, -1, T)], dim=1)))
        w = torch.softmax(self.asp_conv(a), dim=2)
        wmu = (w * h).sum(dim=2)
        wsd = torch.sqrt((w * (h - wmu[:, :, None]) ** 2).sum(dim=2).clamp_min(1e-12))
        pooled = self.asp_bn(torch.cat([wmu, wsd], dim=1)[:, :, None])
        return self.fc(pooled)[:, :, 0]


def load_from_dict(model: EcapaNN, w) -> EcapaNN:
    """Copy a weights.py-style dictionary into the module tree (the only thing shared with the oracle)."""
    t = lambda k: torch.from_numpy(w[k]).double()

    def tdnn(mod, name):
        mod.conv.weight.data = t(f"{name}.conv.w"); mod.conv.bias.data = t(f"{name}.conv.b")
        bn(mod.bn, f"{name}.bn")

    def bn(mod, name):
        mod.weight.data = t(f"{name}.gamma"); mod.bias.data = t(f"{name}.beta")
        mod.running_mean.data = t(f"{name}.mean"); mod.running_var.data = t(f"{name}.var")

    model.double()
    tdnn(model.blk0, "blk0")
    for i, b in enumerate(model.blocks, start=1):
        tdnn(b.tdnn1, f"blk{i}.tdnn1"); tdnn(b.tdnn2, f"blk{i}.tdnn2")
        for j, r in enumerate(b.res2net):
            tdnn(r, f"blk{i}.res2net.{j}")
        b.se1.weight.data = t(f"blk{i}.se.conv1.w"); b.se1.bias.data = t(f"blk{i}.se.conv1.b")
        b.se2.weight.data = t(f"blk{i}.se.conv2.w"); b.se2.bias.data = t(f"blk{i}.se.conv2.b")
    tdnn(model.mfa, "mfa")
    tdnn(model.asp_tdnn, "asp.tdnn")
    model.asp_conv.weight.data = t("asp.conv.w"); model.asp_conv.bias.data = t("asp.conv.b")
    bn(model.asp_bn, "asp_bn")
    model.fc.weight.data = t("fc.w"); model.fc.bias.data = t("fc.b")
    return model.eval()


def public_state_dict(model: EcapaNN):
    """The module tree's own state_dict() re-keyed into the PUBLIC ECAPA-TDNN checkpoint naming (every Conv1d wrapped as `.conv`, every
    BatchNorm1d as `.norm`; blocks.0 = first TDNN, blocks.1-3 = SE-Res2Net blocks, ...), float32, including the `num_batches_tracked`
    buffers a real checkpoint carries.  Written from the public module structure, independently of weights._public_key_map."""
    sd = model.state_dict()
    out = {}

    def tdnn(src, dst):
        out[f"{dst}.conv.conv.weight"] = sd[f"{src}.conv.weight"]; out[f"{dst}.conv.conv.bias"] = sd[f"{src}.conv.bias"]
        norm(f"{src}.bn", f"{dst}.norm")

    def norm(src, dst):
        for a, b in (("weight", "weight"), ("bias", "bias"), ("running_mean", "running_mean"), ("running_var", "running_var"),
                     ("num_batches_tracked", "num_batches_tracked")):
            out[f"{dst}.norm.{b}"] = sd[f"{src}.{a}"]

    tdnn("blk0", "blocks.0")
    for i in range(len(model.blocks)):
        b = f"blocks.{i}"
        tdnn(f"{b}.tdnn1", f"blocks.{i + 1}.tdnn1")
        for j in range(len(model.blocks[i].res2net)):
            tdnn(f"{b}.res2net.{j}", f"blocks.{i + 1}.res2net_block.blocks.{j}")
        tdnn(f"{b}.tdnn2", f"blocks.{i + 1}.tdnn2")
        for k, name in ((1, "se1"), (2, "se2")):
            out[f"blocks.{i + 1}.se_block.conv{k}.conv.weight"] = sd[f"{b}.{name}.weight"]
            out[f"blocks.{i + 1}.se_block.conv{k}.conv.bias"] = sd[f"{b}.{name}.bias"]
    tdnn("mfa", "mfa")
    tdnn("asp_tdnn", "asp.tdnn")
    out["asp.conv.conv.weight"] = sd["asp_conv.weight"]; out["asp.conv.conv.bias"] = sd["asp_conv.bias"]
    norm("asp_bn", "asp_bn")
    out["fc.conv.weight"] = sd["fc.weight"]; out["fc.conv.bias"] = sd["fc.bias"]
    return {k: (v.float() if v.is_floating_point() else v) for k, v in out.items()}
