"""Generate tests/golden/*.npz by running the REFERENCE itself on CPU (build container only).

TEST INFRASTRUCTURE ONLY.  Run from the repo root, once per architecture, in the container
that has /root/reference mounted (it does not exist on the GPU box; nothing at test time
reads it):

    python oracle/gen_golden.py --net_res 128
    python oracle/gen_golden.py --net_res 32

The reference parses sys.argv at import (model/ifnet.py:8) and imports visualisation-only
packages that are absent here (util/visualize.py:1-6), so argv is cleaned and those three
names are stubbed with empty modules before the import.  Only inputs/outputs are stored;
weights are name-seeded (oracle.ifnet_oracle.name_seeded_state) and rebuilt by the tests.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import ifnet_oracle as O  # noqa: E402

REF = os.environ.get("SVR_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
GAIN = 3.0


def sample(t, cap=4096):
    f = t.detach().reshape(-1)
    stride = max(1, -(-f.numel() // cap))
    return f[::stride].contiguous().numpy().copy()


def make_inputs(seed, B, dims, N, spread=1.0, density=0.05):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = (torch.rand(B, 1, *dims, generator=g) < density).float()
    pts = (torch.rand(B, N, 3, generator=g) - 0.5) * spread
    occ = (torch.rand(B, N, generator=g) < 0.5).float()
    return x, pts, occ


def ifnet_case(IFNet, net_res, tag, seed, B, dims, N, spread=1.0):
    torch.manual_seed(0)
    x, pts, occ = make_inputs(seed, B, dims, N, spread)
    st = O.name_seeded_state(net_res, GAIN)
    ref = IFNet()
    missing = ref.load_state_dict(st, strict=False)
    assert not missing.unexpected_keys and all("num_batches" in k for k in missing.missing_keys), missing
    ref.train()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-4)

    # intermediate features of the reference extractor (first 8 points only)
    feats = ref.ifnet_feature_extractor(x, pts)              # (B, sumC, 1, 7, N); also steps BN buffers
    feats8 = feats[..., :8].detach().clone()
    # reset BN buffers touched by the probe forward
    ref.load_state_dict(st, strict=False)

    logits = ref(x, pts)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, occ, reduction="none").sum(-1).mean()
    opt.zero_grad()
    loss.backward()
    out = {
        "meta": np.array([net_res, seed, B, dims[0], dims[1], dims[2], N], dtype=np.int64),
        "spread": np.float32(spread), "gain": np.float32(GAIN),
        "x_bits": np.packbits(x.numpy().astype(np.uint8)),
        "points": pts.numpy(), "occupancies": occ.numpy().astype(np.uint8),
        "logits": logits.detach().numpy(), "loss": np.float64(loss.item()),
        "features8": feats8.numpy(),
    }
    for name, p in ref.named_parameters():
        g = p.grad
        out["grad_norm/" + name] = np.float64(g.double().norm().item())
        out["grad/" + name] = sample(g)
    opt.step()
    sd = ref.state_dict()
    for name, t in sd.items():
        if "num_batches" in name:
            continue
        if O.is_buffer(name):
            out["buf/" + name] = t.numpy().copy()
        else:
            out["adam/" + name] = sample(t)
    # eval-mode logits with the updated parameters and buffers
    ref.eval()
    with torch.no_grad():
        out["logits_eval_after_step"] = ref(x, pts).numpy()
    path = os.path.join(OUT, f"ifnet_{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: loss={loss.item():.6f} logits std={logits.std().item():.4f} "
          f"absmax={logits.abs().max().item():.4f} size={os.path.getsize(path)/1024:.0f} KiB")


def ifnet_bf16_case(IFNet, tag, seed, B, dims, N, spread=1.0):
    """The reference run end to end in bfloat16 on the CPU (module.bfloat16(), bf16 inputs: every conv, BatchNorm,
    grid_sample and Conv1d in bf16 -- the precision util/arguments.py:30 --precision 16 asks Lightning for), next to
    the same module in float32: the two reference answers the bf16-storage throughput mode is reported against."""
    torch.manual_seed(0)
    x, pts, _ = make_inputs(seed, B, dims, N, spread)
    st = O.name_seeded_state(128, GAIN)
    out = {"meta": np.array([128, seed, B, dims[0], dims[1], dims[2], N], dtype=np.int64), "spread": np.float32(spread),
           "gain": np.float32(GAIN), "x_bits": np.packbits(x.numpy().astype(np.uint8)), "points": pts.numpy()}
    for mode in ("train", "eval"):
        for dt, name in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            ref = IFNet()
            ref.load_state_dict(st, strict=False)
            ref = ref.to(dt)
            ref.train(mode == "train")
            with torch.no_grad():
                z = ref(x.to(dt), pts.to(dt))
            out[f"logits_{name}_{mode}"] = z.float().numpy()
    for mode in ("train", "eval"):
        a, b = out[f"logits_bf16_{mode}"], out[f"logits_f32_{mode}"]
        print(f"  {mode}: reference bf16 vs reference f32: max|d| / max|f32| = {np.abs(a - b).max() / np.abs(b).max():.3e}")
    path = os.path.join(OUT, f"ifnet_{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: size={os.path.getsize(path)/1024:.0f} KiB")


def project_case(project, tag, seed, B, dims, kernel, sigma, scale):
    g = torch.Generator(device="cpu").manual_seed(seed)
    depth = torch.rand(B, 240, 320, generator=g) * 5 + 0.5
    wsel = torch.rand(B, 1, *dims, generator=g)
    depth.requires_grad_(True)
    mod = project(dims, list(kernel), torch.tensor(sigma, dtype=torch.float32))
    grid_pc = mod.depthmap_to_gridspace(depth, scale)
    grid_pc_keep = grid_pc.detach().clone()
    norm_pc = mod.norm_grid_space(grid_pc.clone())
    norm_keep = norm_pc.detach().clone()
    vox_raw = mod.pc_voxels(norm_pc)
    occ = mod(norm_pc)
    obj = (occ * wsel).sum()
    obj.backward()
    valid = torch.all((norm_keep < 0.5 - 1e-6) & (norm_keep > -0.5 + 1e-6), dim=-1)
    gi = ((norm_keep + 0.5) * (mod.vox_size - 1)).floor().to(torch.int16)
    stride = 37
    out = {
        "meta": np.array([seed, B, dims[0], dims[1], dims[2], kernel[0], kernel[1], kernel[2], scale], dtype=np.int64),
        "sigma": np.array(sigma, dtype=np.float32),
        "depth_s": sample(depth), "wsel_s": sample(wsel),
        "grid_pc_s": grid_pc_keep.reshape(-1, 3)[::stride].numpy().copy(),
        "norm_pc_s": norm_keep.reshape(-1, 3)[::stride].numpy().copy(),
        "valid_bits": np.packbits(valid.numpy().astype(np.uint8)),
        "vox_idx_s": gi.reshape(-1, 3)[::stride].numpy().copy(),
        "vox_idx_sum": np.int64(gi.long()[valid].sum().item()),
        "vox_raw_s": sample(vox_raw, 16384), "vox_raw_sum": np.float64(vox_raw.double().sum().item()),
        "occ_s": sample(occ, 16384), "occ_sum": np.float64(occ.double().sum().item()),
        "obj": np.float64(obj.item()),
        "sigma_grad": mod.sigma.grad.numpy().copy(),
        "depth_grad_s": sample(depth.grad, 16384),
        "depth_grad_norm": np.float64(depth.grad.double().norm().item()),
        "stride": np.int64(stride),
    }
    path = os.path.join(OUT, f"project_{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: occ_sum={out['occ_sum']:.3f} valid={valid.float().mean().item():.3f} "
          f"sigma_grad={out['sigma_grad']} size={os.path.getsize(path)/1024:.0f} KiB")


def scene_case(Unet, project, IFNet, tag, seed, B, scale, N):
    """The reference modules composed exactly like SceneNetTrainer.forward / losses_and_logging
    (trainer/trainer_scene_net.py:69-103,147-149; the LightningModule itself needs pytorch_lightning)."""
    from oracle import scene_oracle as S
    import torch.nn.functional as F
    g = torch.Generator(device="cpu").manual_seed(seed)
    rgb = torch.rand(B, 3, 256, 256, generator=g) * 2 - 1
    target = torch.rand(B, 240, 320, generator=g) * 5 + 0.5
    pts = torch.rand(B, N, 3, generator=g) - 0.5
    occ = (torch.rand(B, N, generator=g) < 0.5).float()
    dims = (torch.tensor([139, 104, 112]) / scale).round().long()
    min_z, max_z = 0.1953997164964676, 7.0
    unet = Unet(channels_in=3, channels_out=1)
    unet.load_state_dict(S.name_seeded_like(unet.state_dict(), 1.0, "unet."), strict=False)
    proj = project(dims, [3, 3, 3], torch.tensor([1.5, 1.5, 1.5]))
    ifnet = IFNet()
    ifnet.load_state_dict(O.name_seeded_state(128, GAIN), strict=False)
    unet.train(); ifnet.train()
    raw = unet(rgb)
    z = F.interpolate(raw, size=320, mode="bilinear")[:, :, 40:280, :].squeeze(1)
    depth = torch.sigmoid(z) * (max_z - min_z) + min_z
    pc = proj.depthmap_to_gridspace(depth, scale)
    pc = proj.norm_grid_space(pc)
    vox = proj(pc)
    logits = ifnet(vox, pts)
    ce = F.binary_cross_entropy_with_logits(logits, occ, reduction="mean")
    mse = F.mse_loss(depth, target, reduction="mean")
    loss = ce + mse
    loss.backward()
    out = {
        "meta": np.array([seed, B, scale, N, int(dims[0]), int(dims[1]), int(dims[2])], dtype=np.int64),
        "gain": np.float32(GAIN), "points": pts.numpy(), "occupancies": occ.numpy().astype(np.uint8),
        "rgb_s": sample(rgb), "target_s": sample(target),
        "depth_s": sample(depth, 8192), "pc_s": sample(pc, 8192), "vox_s": sample(vox, 16384),
        "vox_sum": np.float64(vox.double().sum().item()),
        "logits": logits.detach().numpy(), "ce": np.float64(ce.item()), "mse": np.float64(mse.item()),
        "loss": np.float64(loss.item()), "sigma_grad": proj.sigma.grad.numpy().copy(),
    }
    for prefix, mod in (("unet.", unet), ("ifnet.", ifnet)):
        for name, p in mod.named_parameters():
            out["grad_norm/" + prefix + name] = np.float64(p.grad.double().norm().item())
            out["grad/" + prefix + name] = sample(p.grad, 256)
    path = os.path.join(OUT, f"scene_{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: loss={loss.item():.6f} ce={ce.item():.6f} mse={mse.item():.6f} vox_sum={out['vox_sum']:.2f} "
          f"sigma_grad={out['sigma_grad']} size={os.path.getsize(path)/1024:.0f} KiB")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--net_res", type=int, default=128)
    ap.add_argument("--only-bf16", action="store_true", help="only (re)generate tests/golden/ifnet_bf16*.npz")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    os.chdir(REF)                       # project() reads data/raw/overfit/00000/intrinsic.txt relative to cwd
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    sys.argv = ["x", "--net_res", str(a.net_res)]
    for n in ("marching_cubes", "trimesh", "pyexr"):
        sys.modules.setdefault(n, types.ModuleType(n))
    import warnings
    warnings.filterwarnings("ignore")
    from model.ifnet import IFNet
    from model.projection import project
    torch.set_num_threads(8)
    if a.only_bf16:
        ifnet_bf16_case(IFNet, "bf16_cfg1", 101, 1, (32, 32, 32), 2048)            # BASELINE configs[0] inputs
        ifnet_bf16_case(IFNet, "bf16_b2", 141, 2, (32, 24, 40), 1500, spread=1.2)
        return
    if a.net_res == 128:
        ifnet_case(IFNet, 128, "cfg1", 101, 1, (32, 32, 32), 2048)                 # BASELINE configs[0]
        ifnet_case(IFNet, 128, "odd", 111, 2, (35, 26, 28), 600, spread=1.25)      # ragged dims, OOB points
        ifnet_case(IFNet, 128, "b3", 112, 3, (16, 16, 16), 257)                    # smallest legal pyramid
        project_case(project, "full", 105, 2, (139, 104, 112), (3, 3, 3), (1.5, 1.5, 1.5), 1)
        project_case(project, "half", 106, 1, (70, 52, 56), (11, 9, 9), (2.0, 1.5, 1.0), 2)
        project_case(project, "cube", 107, 2, (48, 40, 56), (5, 3, 3), (0.8, 1.2, 1.7), 2)
        from model.unet import Unet
        scene_case(Unet, project, IFNet, "cfg5small", 131, 2, 4, 300)
    else:
        ifnet_case(IFNet, 32, "res32", 121, 2, (16, 12, 20), 512, spread=1.1)


if __name__ == "__main__":
    main()
