"""CPU oracle: functional restatement of the reference IF-Net path in stock torch CPU ops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows, call for call:

* coordinate prep + 7 displaced samples ........ model/ifnet.py:144-161 (128) / :82-97 (32)
* six (four) trilinear ``grid_sample`` calls ..... model/ifnet.py:162-193 / :98-115
* encoder conv -> ReLU (-> conv -> ReLU) -> BN -> pool  model/ifnet.py:164-192 / :100-114
* feature concat + reshape (row k = c*7 + j) .... model/ifnet.py:197, :43-45
* point MLP fc_0..fc_out ........................ model/ifnet.py:55-59
* loss  BCEWithLogits(none).sum(-1).mean() ...... trainer/trainer_ifnet.py:46
* Adam(lr) .......................................trainer/trainer_ifnet.py:29

Parameters are a flat ``dict[str, Tensor]`` keyed by the reference's state-dict
names so that the same name-seeded weights can be pushed
into the imported reference module, into this oracle and into the HIP module.

Parity: pinned by tests/golden/ifnet_*.npz (outputs of the imported reference,
made by oracle/gen_golden.py) -- tests/test_oracle_golden.py.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import torch
import torch.nn.functional as F

FX = "ifnet_feature_extractor."

# (conv names), bn name  per encoder stage; a 2x2x2 max-pool sits between stages.
ARCH = {
    128: dict(
        disp=0.0722, align_corners=False, hidden=(256, 256, 256),
        stages=[(("conv_in",), "conv_in_bn"),
                (("conv_0", "conv_0_1"), "conv0_1_bn"),
                (("conv_1", "conv_1_1"), "conv1_1_bn"),
                (("conv_2", "conv_2_1"), "conv2_1_bn"),
                (("conv_3", "conv_3_1"), "conv3_1_bn")],
        chans=[(1, 16), (16, 32, 32), (32, 64, 64), (64, 128, 128), (128, 128, 128)],
    ),
    32: dict(
        disp=0.035, align_corners=True, hidden=(512, 256, 256),
        stages=[(("conv_1", "conv_1_1"), "conv1_1_bn"),
                (("conv_2", "conv_2_1"), "conv2_1_bn"),
                (("conv_3", "conv_3_1"), "conv3_1_bn")],
        chans=[(1, 32, 64), (64, 128, 128), (128, 128, 128)],
    ),
}


def level_channels(net_res: int):
    """Channel count of every sampled level, level 0 = the raw input grid."""
    return [1] + [c[-1] for c in ARCH[net_res]["chans"]]


def feature_size(net_res: int) -> int:
    return 7 * sum(level_channels(net_res))


def param_shapes(net_res: int = 128) -> "OrderedDict[str, tuple]":
    """Reference state-dict entries (floating ones) and their shapes, in module order."""
    a = ARCH[net_res]
    out: "OrderedDict[str, tuple]" = OrderedDict()
    for (convs, bn), ch in zip(a["stages"], a["chans"]):
        for i, cname in enumerate(convs):
            out[FX + cname + ".weight"] = (ch[i + 1], ch[i], 3, 3, 3)
            out[FX + cname + ".bias"] = (ch[i + 1],)
    for (convs, bn), ch in zip(a["stages"], a["chans"]):
        c = ch[-1]
        for suffix in ("weight", "bias", "running_mean", "running_var"):
            out[FX + bn + "." + suffix] = (c,)
    h0, h1, h2 = a["hidden"]
    fs = feature_size(net_res)
    for name, (co, ci) in (("fc_0", (h0, fs)), ("fc_1", (h1, h0)), ("fc_2", (h2, h1)), ("fc_out", (1, h2))):
        out[name + ".weight"] = (co, ci, 1)
        out[name + ".bias"] = (co,)
    return out


def is_buffer(name: str) -> bool:
    return name.endswith("running_mean") or name.endswith("running_var")


def name_seeded_state(net_res: int = 128, gain: float = 3.0) -> "OrderedDict[str, torch.Tensor]":
    """Build convention of SURVEY.md App. A.7: every tensor is drawn from a generator
    seeded with crc32(name), so fixtures carry only inputs and outputs."""
    st: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in param_shapes(net_res).items():
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
        u = torch.rand(shape, generator=g, dtype=torch.float32)
        if name.endswith("running_mean"):
            t = torch.zeros(shape)
        elif name.endswith("running_var"):
            t = torch.ones(shape)
        elif "_bn.weight" in name:
            t = 0.5 + u
        elif "_bn.bias" in name:
            t = 0.4 * u - 0.2
        elif name.endswith(".bias"):
            t = 0.2 * u - 0.1
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = (2.0 * u - 1.0) * (gain / fan_in ** 0.5)
        st[name] = t
    return st


def displacements(net_res: int = 128) -> torch.Tensor:
    d = ARCH[net_res]["disp"]
    rows = [[0.0, 0.0, 0.0]]
    for axis in range(3):
        for sign in (-1.0, 1.0):
            r = [0.0, 0.0, 0.0]
            r[axis] = sign * d
            rows.append(r)
    return torch.tensor(rows, dtype=torch.float32)


def sample_grid(points: torch.Tensor, net_res: int = 128) -> torch.Tensor:
    """(B,N,3) points -> (B,1,7,N,3) grid_sample coordinates (x<-pts[2], z<-pts[0])."""
    g = torch.stack((2 * points[..., 2], 2 * points[..., 1], 2 * points[..., 0]), dim=-1)
    g = g[:, None, None]                                   # (B,1,1,N,3)
    return torch.cat([g + d for d in displacements(net_res).to(g.dtype)], dim=2)


def encoder_levels(st, x, net_res=128, training=True, momentum=0.1, eps=1e-5, keep=None):
    """Returns the list of sampled volumes [x, bn_1, bn_2, ...] (NCDHW).  BN buffers in
    ``st`` are updated in place in training mode, like nn.BatchNorm3d."""
    a = ARCH[net_res]
    levels = [x]
    net = x
    for si, (convs, bn) in enumerate(a["stages"]):
        if si > 0:
            net = F.max_pool3d(net, 2)
        for cname in convs:
            net = F.relu(F.conv3d(net, st[FX + cname + ".weight"], st[FX + cname + ".bias"], padding=1))
        if keep is not None:
            keep.append(net)
        net = F.batch_norm(net, st[FX + bn + ".running_mean"], st[FX + bn + ".running_var"],
                           st[FX + bn + ".weight"], st[FX + bn + ".bias"], training, momentum, eps)
        levels.append(net)
    return levels


def gather_features(levels, points, net_res=128):
    """(B, sum(C)*7, N) feature matrix, row k = c*7 + j."""
    g = sample_grid(points, net_res)
    ac = ARCH[net_res]["align_corners"]
    feats = [F.grid_sample(v, g.to(v.dtype), mode="bilinear", padding_mode="zeros", align_corners=ac)
             for v in levels]
    f = torch.cat(feats, dim=1)                            # (B, sumC, 1, 7, N)
    return f.reshape(f.shape[0], f.shape[1] * f.shape[3], f.shape[4])


def point_mlp(st, feats):
    net = feats
    for name in ("fc_0", "fc_1", "fc_2"):
        net = F.relu(F.conv1d(net, st[name + ".weight"], st[name + ".bias"]))
    return F.conv1d(net, st["fc_out.weight"], st["fc_out.bias"]).squeeze(1)


def ifnet_forward(st, x, points, net_res=128, training=True):
    levels = encoder_levels(st, x, net_res, training)
    return point_mlp(st, gather_features(levels, points, net_res))


def training_loss(logits, occupancies):
    return F.binary_cross_entropy_with_logits(logits, occupancies, reduction="none").sum(-1).mean()


def training_step(st, batch, net_res=128):
    """Mirror of ImplicitRefinementTrainer.training_step: returns {'loss': scalar}."""
    logits = ifnet_forward(st, batch["input"], batch["points"], net_res, training=True)
    return {"loss": training_loss(logits, batch["occupancies"]), "logits": logits}


def make_leaf_state(st):
    """Clone a state so the learnable entries are autograd leaves."""
    out = OrderedDict()
    for k, v in st.items():
        t = v.detach().clone()
        if not is_buffer(k):
            t.requires_grad_(True)
        out[k] = t
    return out


# --------------------------------------------------------------------------------------
# Explicit trilinear rule, independent of F.grid_sample: used for the
# bit-exact corner-index gate and to cross-check the HIP gather at sizes where storing
# golden vectors would be too large.
# --------------------------------------------------------------------------------------
def corner_indices(points, size_dhw, net_res=128):
    """Integer base corner (z0,y0,x0) of every (b, j, n) sample and the f32 source index.

    Every arithmetic step is a separately rounded fp32 op in the order grid_sample uses
    (torch/include/ATen/native/GridSampler.h:27-36): ((g+1)*S-1)/2 for align_corners=False,
    ((g+1)/2)*(S-1) for True."""
    Dd, H, W = size_dhw
    g = sample_grid(points.float(), net_res)[:, 0]          # (B,7,N,3): x,y,z
    ac = ARCH[net_res]["align_corners"]

    def unnorm(c, S):
        if ac:
            return ((c + 1.0) / 2.0) * float(S - 1)
        return ((c + 1.0) * float(S) - 1.0) / 2.0

    ix, iy, iz = unnorm(g[..., 0], W), unnorm(g[..., 1], H), unnorm(g[..., 2], Dd)
    x0, y0, z0 = ix.floor(), iy.floor(), iz.floor()
    idx = torch.stack((z0, y0, x0), dim=-1).to(torch.int32)
    return idx, torch.stack((iz, iy, ix), dim=-1)


def gather_manual(vol, points, net_res=128):
    """Explicit 8-corner trilinear gather of one NCDHW volume -> (B, C, 7, N)."""
    B, C, Dd, H, W = vol.shape
    idx, src = corner_indices(points, (Dd, H, W), net_res)
    z0, y0, x0 = (idx[..., k].long() for k in range(3))
    fz, fy, fx = (src[..., k] for k in range(3))
    out = torch.zeros(B, C, 7, points.shape[1], dtype=vol.dtype)
    bidx = torch.arange(B)[:, None, None].expand_as(z0)
    for dz in (0, 1):
        wz = (z0 + 1).to(fz.dtype) - fz if dz == 0 else fz - z0.to(fz.dtype)
        for dy in (0, 1):
            wy = (y0 + 1).to(fy.dtype) - fy if dy == 0 else fy - y0.to(fy.dtype)
            for dx in (0, 1):
                wx = (x0 + 1).to(fx.dtype) - fx if dx == 0 else fx - x0.to(fx.dtype)
                z, y, x = z0 + dz, y0 + dy, x0 + dx
                ok = (z >= 0) & (z < Dd) & (y >= 0) & (y < H) & (x >= 0) & (x < W)
                v = vol[bidx, :, z.clamp(0, Dd - 1), y.clamp(0, H - 1), x.clamp(0, W - 1)]  # (B,7,N,C)
                w = (wx * wy * wz) * ok.to(vol.dtype)
                out += (v * w[..., None]).permute(0, 3, 1, 2)
    return out
