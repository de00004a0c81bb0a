"""CPU oracle: the reference's sample loading restated with its own primitives (struct / np.load / Python lists).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows:

* BinaryReader / read_df ......... data_processing/volume_reader.py:21-45 (3 x UINT64 dims, dimX*dimY*dimZ floats
                                   through struct.unpack, reshape order='F'); down_sample = skimage block_reduce mean :47-51
* ImplicitDataset.__getitem__ .... dataset/implicit_dataset.py:24-56
* file formats written by ......... data_processing/process_sample.py:19-30 (np.savez_compressed(depth_grid, grid=float64
                                   grid); copy of the .df; np.savez(occupancy_<sigma>, points, occupancies, grid_coords))

Parity: PINNED by the reference itself.  oracle/gen_golden_dataset.py imports the reference's unmodified
dataset/implicit_dataset.py and data_processing/volume_reader.py in the build container (visualisation-only packages as
empty modules; skimage.measure.block_reduce -- only reachable with scale_factor != 1 -- as a function that raises) and
stores what ImplicitDataset.__getitem__ / read_df returned in tests/golden/dataset_{real_grid,synthetic}.npz;
tests/test_sample_io.py holds getitem / read_df below (and the product's mirror over the native readers) to those
outputs bit for bit.  The sample files are regenerated from the fixture's seeds by make_sample (process_sample.py's own
calls) around the reference's real depth_grid.npz (tests/golden/ref_depth_grid.npz: the only sample file the reference
ships -- target.df and the occupancy files are in .MISSING_LARGE_BLOBS).  Unpinned: down_sample (skimage absent).
"""
import struct
from pathlib import Path

import numpy as np
import torch


def read_df(filename):
    with open(filename, "rb") as f:
        dimX, dimY, dimZ = struct.unpack("QQQ", f.read(24))
        n = dimX * dimY * dimZ
        raw = f.read(4 * n)
        if len(raw) != 4 * n:
            raise Exception
        df = struct.unpack("f" * n, raw)
    return np.array(df, dtype=np.float32).reshape([dimX, dimY, dimZ], order="F")


def write_df(filename, vol):
    """The .df layout read_df expects (x fastest)."""
    with open(filename, "wb") as f:
        f.write(struct.pack("QQQ", *vol.shape))
        f.write(np.asarray(vol, dtype=np.float32).flatten(order="F").tobytes())


def getitem(sample_folder, item, num_points):
    sample_folder = Path(sample_folder)
    sample_input = torch.from_numpy(np.load(sample_folder / "depth_grid.npz")["grid"]).float()
    sample_target = torch.from_numpy(read_df(str(sample_folder / "target.df"))).float()
    points, occupancies, grids = [], [], []
    for sigma in ["0.10", "0.01"]:
        z = np.load(sample_folder / f"occupancy_{sigma}.npz")
        p, g, o = z["points"], z["grid_coords"], z["occupancies"]
        idx = np.random.randint(0, p.shape[0], num_points)
        points.extend(p[idx])
        grids.extend(g[idx])
        occupancies.extend(o[idx])
    return {"name": item, "grid": torch.from_numpy(np.array(grids, dtype=np.float32)),
            "points": torch.from_numpy(np.array(points, dtype=np.float32)), "input": sample_input.unsqueeze(0),
            "occupancies": torch.from_numpy(np.array(occupancies, dtype=np.float32)), "target": sample_target.unsqueeze(0)}


def make_sample(folder, dims=(139, 104, 112), n_pts=5000, seed=0, grid_from=None):
    """Write one processed sample with process_sample.py's own calls (:22,:26,:30)."""
    folder = Path(folder)
    folder.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(seed)
    if grid_from is not None:
        grid = np.load(grid_from)["grid"]
    else:
        grid = (rng.random(dims) < 0.03).astype(np.float64)
    np.savez_compressed(folder / "depth_grid", grid=grid)
    write_df(folder / "target.df", rng.standard_normal(grid.shape).astype(np.float32))
    for sigma in (0.01, 0.1):
        pts = rng.uniform(-0.5, 0.5, size=(n_pts, 3))
        gc = pts.copy()
        gc[:, 0], gc[:, 2] = pts[:, 2], pts[:, 0]
        np.savez(folder / f"occupancy_{sigma:.02f}", points=pts, occupancies=rng.random(n_pts) < 0.4, grid_coords=2 * gc)
    return grid
