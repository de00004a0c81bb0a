"""CPU oracle: restatement of the reference depth -> point cloud -> voxel grid path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows:

* pinhole unprojection ............ model/projection.py:199-206  (depth_to_camera)
* frustum + camera->grid affine ... model/projection.py:150-197  (depthmap_to_gridspace,
                                    generate_frustum, generate_frustum_volume)
* grid-space normalisation ........ model/projection.py:124-148
* trilinear splat (x8 alias quirk)  model/projection.py:39-80   (pc_voxels)
* separable learnable-sigma blur .. model/projection.py:82-117
* intrinsics ...................... data/raw/overfit/00000/intrinsic.txt via :208-218

Everything is stock torch CPU ops, so autograd gives the backward used by the tests
(grad wrt sigma, wrt the points, wrt depth).  Pinned by tests/golden/project_*.npz.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

FOCAL, CX, CY = 277.1281435, 159.5, 119.5          # the reference's intrinsic.txt
IMG_W, IMG_H = 320, 240                             # hard-coded frustum image (projection.py:156)
DEPTH_MIN, DEPTH_MAX = 0.4, 6.0


def intrinsic_matrix() -> torch.Tensor:
    return torch.tensor([[FOCAL, 0, CX, 0], [0, FOCAL, CY, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)


def camera_to_grid(scale_factor=1):
    """(dims xyz as floats, 4x4 camera->frustum-grid affine) -- projection.py:165-197."""
    Kinv = torch.inverse(intrinsic_matrix())
    corners = []
    for d in (DEPTH_MIN, DEPTH_MAX):
        for (u, v) in ((0, 0), (0, IMG_H), (IMG_W, IMG_H), (IMG_W, 0)):
            corners.append([u * d, v * d, d, 1.0])
    fr = torch.mm(Kinv, torch.tensor(corners, dtype=torch.float32).t()).t()[:, :3]
    vs = 0.05 * scale_factor
    hi = fr.max(dim=0).values / vs
    lo = fr.min(dim=0).values / vs
    dims = torch.ceil(hi - lo)
    c2f = torch.tensor([[1.0 / vs, 0, 0, -lo[0]], [0, 1.0 / vs, 0, -lo[1]],
                        [0, 0, 1.0 / vs, -lo[2]], [0, 0, 0, 1.0]], dtype=torch.float32)
    return dims, c2f


def depthmap_to_gridspace(depth: torch.Tensor, scale_factor=1) -> torch.Tensor:
    """depth (B,Hi,Wi) -> (B, Hi*Wi, 3) grid-space points.  Note the reference stacks the
    flattened X/Y/Z over the WHOLE batch before the 4x4 product (projection.py:160-161)."""
    B = depth.shape[0]
    K = intrinsic_matrix()
    f, cx, cy = K[0, 0], K[0, 2], K[1, 2]
    v, u = torch.meshgrid(torch.arange(depth.shape[-2]), torch.arange(depth.shape[-1]), indexing="ij")
    X = (torch.multiply(u, depth) - cx * depth) / f
    Y = -((torch.multiply(v, depth) - cy * depth) / f)
    Z = depth
    X, Y, Z = X.flatten(), Y.flatten(), Z.flatten()
    _, c2f = camera_to_grid(scale_factor)
    coords = torch.stack([X, Y, Z, torch.ones_like(X)])
    return (c2f @ coords)[:3, :].transpose(1, 0).reshape(B, -1, 3)


def norm_grid_space(pc: torch.Tensor, dims) -> torch.Tensor:
    d = torch.as_tensor(dims)
    half = d / 2
    return torch.stack([(pc[..., k] - half[k]) / d[k] for k in range(3)], dim=-1)


def un_norm_grid_space(pc: torch.Tensor, dims) -> torch.Tensor:
    d = torch.as_tensor(dims)
    half = d / 2
    return torch.stack([pc[..., k] * d[k] + half[k] for k in range(3)], dim=-1)


def splat_indices(points: torch.Tensor, dims, eps=1e-6):
    """valid mask (B,N) and int64 base voxel (B,N,3) -- the bit-exact gate of pc_voxels."""
    d = torch.as_tensor(dims, dtype=torch.int64)
    valid = torch.all((points < 0.5 - eps) & (points > -0.5 + eps), dim=-1)
    g = (points + 0.5) * (d - 1)
    return valid, g.floor().long(), g


def pc_voxels(points: torch.Tensor, dims, eps=1e-6) -> torch.Tensor:
    """(B,N,3) normalised points -> (B,D0,D1,D2) clamp(8 * splatted trilinear weights, 0, 1)."""
    B, N, _ = points.shape
    d0, d1, d2 = (int(v) for v in dims)
    valid, base, g = splat_indices(points, dims, eps)
    r = g - g.floor()
    w01 = (1.0 - r, r)
    acc = points.new_zeros(B * d0 * d1 * d2)
    b = torch.arange(B)[:, None].expand(B, N)
    vflat = valid.reshape(-1)
    for k in (0, 1):
        for j in (0, 1):
            for i in (0, 1):
                w = w01[k][..., 0] * w01[j][..., 1] * w01[i][..., 2]
                lin = ((b * d0 + base[..., 0] + k) * d1 + base[..., 1] + j) * d2 + base[..., 2] + i
                acc = acc.index_add(0, lin.reshape(-1)[vflat], w.reshape(-1)[vflat])
    # the reference sums 8 aliases of the one accumulated tensor (projection.py:75-80)
    return (8.0 * acc.view(B, d0, d1, d2)).clamp(0, 1)


def smoothing_taps(K: int) -> torch.Tensor:
    return torch.arange(-K // 2 + 1.0, K // 2 + 1.0)


def smoothing_kernels(sigma: torch.Tensor, kernel_size):
    ks = []
    for a in range(3):
        t = smoothing_taps(int(kernel_size[a]))
        k = torch.exp(-t ** 2 / (2.0 * sigma[a] ** 2))
        ks.append(k / k.sum())
    return ks          # ks[0] acts on the LAST spatial axis, ks[2] on the first


def voxels_smooth(vox: torch.Tensor, kernels) -> torch.Tensor:
    B = vox.shape[0]
    v = vox.unsqueeze(0)                                     # (1,B,D0,D1,D2), groups=B
    shapes = ((1, 1, 1, 1, -1), (1, 1, 1, -1, 1), (1, 1, -1, 1, 1))
    for k, shp in zip(kernels, shapes):
        pad = [0, 0, 0]
        pad[shp.index(-1) - 2] = k.numel() // 2
        v = F.conv3d(v, k.view(shp).repeat(B, 1, 1, 1, 1), stride=1, padding=pad, groups=B)
    return v.squeeze(0).clamp(0, 1)


def project_forward(points: torch.Tensor, dims, sigma: torch.Tensor, kernel_size) -> torch.Tensor:
    """project.forward: normalised point cloud -> (B,1,D0,D1,D2) smoothed occupancy."""
    return voxels_smooth(pc_voxels(points, dims), smoothing_kernels(sigma, kernel_size)).unsqueeze(1)
