"""Helpers shared by the parity tests: load tests/golden/*.npz (outputs of the reference itself,
made by oracle/gen_golden.py) and rebuild the seeded inputs / name-seeded weights."""
import os

import numpy as np
import torch

from oracle import ifnet_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
IFNET_CASES = ["cfg1", "odd", "b3"]
PROJECT_CASES = ["full", "half", "cube"]


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def ifnet_inputs(z):
    net_res, seed, B, d0, d1, d2, N = (int(v) for v in z["meta"])
    n = B * d0 * d1 * d2
    x = torch.from_numpy(np.unpackbits(z["x_bits"])[:n].astype(np.float32)).view(B, 1, d0, d1, d2)
    pts = torch.from_numpy(z["points"].copy())
    occ = torch.from_numpy(z["occupancies"].astype(np.float32))
    return net_res, x, pts, occ


def sample(t, cap=4096):
    f = t.detach().reshape(-1).cpu()
    stride = max(1, -(-f.numel() // cap))
    return f[::stride].contiguous().numpy()


def rel_err(a, b):
    """max|a-b| / max|b| -- the tolerance convention of SURVEY.md §7 hard part 1."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def project_inputs(z):
    seed, B, d0, d1, d2, k0, k1, k2, scale = (int(v) for v in z["meta"])
    g = torch.Generator(device="cpu").manual_seed(seed)
    depth = torch.rand(B, 240, 320, generator=g) * 5 + 0.5
    wsel = torch.rand(B, 1, d0, d1, d2, generator=g)
    assert np.array_equal(sample(depth), z["depth_s"]) and np.array_equal(sample(wsel), z["wsel_s"])
    return depth, wsel, (d0, d1, d2), (k0, k1, k2), torch.from_numpy(z["sigma"].copy()), scale


def state(net_res, gain=None, z=None):
    g = float(z["gain"]) if z is not None else (gain or 3.0)
    return O.name_seeded_state(net_res, g)


def scene_inputs(z):
    seed, B, scale, N, d0, d1, d2 = (int(v) for v in z["meta"])
    g = torch.Generator(device="cpu").manual_seed(seed)
    rgb = torch.rand(B, 3, 256, 256, generator=g) * 2 - 1
    target = torch.rand(B, 240, 320, generator=g) * 5 + 0.5
    assert np.array_equal(sample(rgb), z["rgb_s"]) and np.array_equal(sample(target), z["target_s"])
    batch = {"rgb": rgb, "depthmap_target": target, "points": torch.from_numpy(z["points"].copy()),
             "occupancies": torch.from_numpy(z["occupancies"].astype(np.float32))}
    return batch, (d0, d1, d2), scale
