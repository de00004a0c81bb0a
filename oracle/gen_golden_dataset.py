"""Generate tests/golden/dataset_*.npz by running the REFERENCE's own sample loader (build container only).

TEST INFRASTRUCTURE ONLY.  Run from the repo root in the container that has /root/reference mounted:

    python oracle/gen_golden_dataset.py

Imports /root/reference/dataset/implicit_dataset.py (ImplicitDataset.__getitem__, :24-56) and
data_processing/volume_reader.py (read_df, :36-45) unmodified.  Their import-time dependencies that are absent in this image
are stubbed exactly like oracle/gen_golden.py does for the model: the three visualisation-only packages of util/visualize.py
(marching_cubes, trimesh, pyexr) as empty modules, and skimage.measure.block_reduce -- used only by down_sample
(volume_reader.py:47-51, reached for scale_factor != 1, which neither the dataset nor this script passes) -- as a function
that RAISES, so a silently wrong answer is impossible.

The reference reads `data/splits/<splitsdir>/<split>.txt` relative to the working directory (:16) and ships only
depth_grid.npz of its one sample (target.df / occupancy_*.npz are in .MISSING_LARGE_BLOBS), so the script chdirs into a
temporary tree whose files are written by oracle.dataset_oracle.make_sample with process_sample.py's own calls -- sample
"00000" around the reference's REAL depth_grid.npz, sample "00001" fully synthetic on an odd non-cubic grid.  A fixture
stores the seeds that regenerate the files + the loader's outputs (fixtures are data, not the reference's code)."""
import os
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import dataset_oracle as DO  # noqa: E402

TSTRIDE = 97
REF = os.environ.get("SVR_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")

# (tag, item, dims or None = the reference's real grid, points per occupancy file, file seed, num_points, numpy seed)
CASES = [("dataset_real_grid", "00000", None, 3000, 11, 512, 1234),
         ("dataset_synthetic", "00001", (23, 17, 19), 700, 12, 300, 4321)]


def import_reference():
    for name in ("marching_cubes", "trimesh", "pyexr", "skimage"):
        sys.modules.setdefault(name, types.ModuleType(name))
    measure = types.ModuleType("skimage.measure")

    def block_reduce(*a, **k):
        raise RuntimeError("skimage.measure.block_reduce is not installed here (stub): scale_factor must stay 1")
    measure.block_reduce = block_reduce
    sys.modules["skimage.measure"] = measure
    sys.modules["skimage"].measure = measure
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from data_processing import volume_reader
    from dataset import implicit_dataset
    return implicit_dataset, volume_reader


def main():
    implicit_dataset, volume_reader = import_reference()
    real_grid = os.path.join(REF, "data", "processed", "overfit", "00000", "depth_grid.npz")
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "data", "splits", "overfit"))
        with open(os.path.join(tmp, "data", "splits", "overfit", "train.txt"), "w") as f:
            f.write("\n".join(c[1] for c in CASES) + "\n")
        for tag, item, dims, n_pts, fseed, num_points, nseed in CASES:
            DO.make_sample(os.path.join(tmp, "data", "processed", "overfit", item), dims=dims or (1, 1, 1), n_pts=n_pts, seed=fseed,
                           grid_from=real_grid if dims is None else None)
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            ds = implicit_dataset.ImplicitDataset("train", "data", 0, "overfit")
            assert len(ds) == 50 * len(CASES)          # ('overfit' in splitsdir and split == 'train': x 50, :18)
            for idx, (tag, item, dims, n_pts, fseed, num_points, nseed) in enumerate(CASES):
                ds.num_points = num_points
                np.random.seed(nseed)
                s = ds[idx]
                assert s["name"] == item
                folder = os.path.join("data", "processed", "overfit", item)
                df = volume_reader.read_df(os.path.join(folder, "target.df"))
                out = {"meta": np.array([n_pts, fseed, num_points, nseed, 1 if dims is None else 0], dtype=np.int64),
                       "dims": np.array(s["input"].shape[1:], dtype=np.int64),
                       "input_bits": np.packbits(s["input"].numpy().astype(np.uint8)),        # the grids are 0 / 1 valued
                       "input_sum": np.float64(s["input"].double().sum().item()),
                       # the distance field: complete for the small case; strided sample + f64 moments for the 1.6 M-voxel one
                       "target_shape": np.array(s["target"].shape, dtype=np.int64),
                       "target": s["target"].numpy() if s["target"].numel() < 100000 else s["target"].numpy().reshape(-1)[::TSTRIDE].copy(),
                       "target_moments": np.array([s["target"].double().sum().item(), (s["target"].double() ** 2).sum().item(),
                                                   (s["target"].double().reshape(-1) * np.arange(s["target"].numel())).sum().item()]),
                       "read_df_equals_target": np.array(bool(np.array_equal(df, s["target"].numpy()[0]))),
                       "points": s["points"].numpy(), "grid": s["grid"].numpy(), "occupancies": s["occupancies"].numpy()}
                assert np.array_equal(np.unique(s["input"].numpy()), [0.0, 1.0]) or s["input"].numpy().max() <= 1.0
                np.savez_compressed(os.path.join(OUT, tag + ".npz"), **out)
                print(tag, {k: (v.shape, str(v.dtype)) for k, v in out.items() if hasattr(v, "shape")})
        finally:
            os.chdir(cwd)


if __name__ == "__main__":
    main()
