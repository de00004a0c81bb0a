"""CPU oracle: the reference's end-to-end scene step (BASELINE config 5), restated functionally.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows:

* UNet depth regressor ............ model/unet.py:15-118 (Unet), :121-186 (UNetMini)
* resize / crop / sigmoid renorm .. trainer/trainer_scene_net.py:71-80
* unproject -> normalise -> voxelise/blur -> IF-Net .. trainer/trainer_scene_net.py:85-101
* loss = BCE(mean) + MSE(depth) ... trainer/trainer_scene_net.py:147-149,165-168

Parameters: flat dicts keyed by the reference state-dict names (`unet.*` names without prefix).
Pinned by tests/golden/scene_*.npz (the reference modules composed by oracle/gen_golden.py).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import ifnet_oracle as IO
from . import projection_oracle as PO

# encoder conv i (1-based) -> the BatchNorm applied to its output (None = no norm)
_UNET = {
    "full": dict(enc=8, enc_bn=[None, "batch_norm2_0", "batch_norm4_0", "batch_norm8_0", "batch_norm8_1",
                                "batch_norm8_2", "batch_norm8_3", None],
                 dec=[("dconv1", "batch_norm8_4"), ("dconv2", "batch_norm8_5"), ("dconv3", "batch_norm8_6"),
                      ("dconv4", "batch_norm8_7"), ("dconv5", "batch_norm4_1"), ("dconv6", "batch_norm2_1"),
                      ("dconv7", "batch_norm"), ("dconv8", None)]),
    "mini": dict(enc=4, enc_bn=[None, "batch_norm2_0", "batch_norm4_0", None],
                 dec=[("dconv5", "batch_norm4_1"), ("dconv6", "batch_norm2_1"), ("dconv7", "batch_norm"),
                      ("dconv8", None)]),
}


def name_seeded_like(state_dict, gain=1.0, prefix=""):
    """Name-seeded replacement for every floating entry of an nn.Module state dict."""
    out = OrderedDict()
    for name, t in state_dict.items():
        if not torch.is_floating_point(t):
            continue
        g = torch.Generator().manual_seed(zlib.crc32((prefix + name).encode()))
        u = torch.rand(t.shape, generator=g, dtype=torch.float32)
        if name.endswith("running_mean"):
            v = torch.zeros(t.shape)
        elif name.endswith("running_var"):
            v = torch.ones(t.shape)
        elif "batch_norm" in name and name.endswith("weight"):
            v = 0.5 + u
        elif "batch_norm" in name and name.endswith("bias"):
            v = 0.4 * u - 0.2
        elif name.endswith("bias"):
            v = 0.2 * u - 0.1
        else:
            fan_in = 1
            for s in t.shape[1:]:
                fan_in *= s
            v = (2.0 * u - 1.0) * (gain / fan_in ** 0.5)
        out[name] = v
    return out


def _bn(st, name, x, training):
    return F.batch_norm(x, st[name + ".running_mean"], st[name + ".running_var"], st[name + ".weight"],
                        st[name + ".bias"], training, 0.1, 1e-5)


def unet_forward(st, img, variant="full", training=True):
    a = _UNET[variant]
    skips = []
    x = img
    for i in range(1, a["enc"] + 1):
        if i > 1:
            x = F.leaky_relu(x, 0.2)
        x = F.conv2d(x, st[f"conv{i}.weight"], st[f"conv{i}.bias"], stride=2, padding=1)
        bn = a["enc_bn"][i - 1]
        if bn is not None:
            x = _bn(st, bn, x, training)
        skips.append(x)
    d = skips.pop()                                        # innermost code: no skip of itself
    for cname, bn in a["dec"]:
        d = F.interpolate(F.relu(d), scale_factor=2, mode="bilinear")
        d = F.conv2d(d, st[cname + ".weight"], st[cname + ".bias"], stride=1, padding=1)
        if bn is not None:
            d = _bn(st, bn, d, training)
            d = torch.cat((d, skips.pop()), 1)
    return d


def scene_forward(unet_st, ifnet_st, sigma, batch, dims, kernel_size, scale_factor=1, min_z=0.1953997164964676,
                  max_z=7.0, resize_input=True, net_res=128, training=True):
    """-> logits, renormalised depth map, normalised point cloud, voxel occupancy."""
    raw = unet_forward(unet_st, batch["rgb"], "full" if resize_input else "mini", training)
    if resize_input:
        z = F.interpolate(raw, size=320, mode="bilinear")[:, :, 40:280, :].squeeze(1)
    else:
        z = raw
    depth = torch.sigmoid(z) * (max_z - min_z) + min_z
    pc = PO.norm_grid_space(PO.depthmap_to_gridspace(depth, scale_factor), dims)
    vox = PO.project_forward(pc, dims, sigma, kernel_size)
    logits = IO.ifnet_forward(ifnet_st, vox, batch["points"], net_res, training)
    return logits, depth, pc, vox


def scene_loss(logits, depth, batch, no_depth_sup=False):
    ce = F.binary_cross_entropy_with_logits(logits, batch["occupancies"], reduction="mean")
    if no_depth_sup:
        return ce
    return ce + F.mse_loss(depth, batch["depthmap_target"], reduction="mean")
