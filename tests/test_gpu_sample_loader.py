"""GPU: DeviceSampleLoader (pinned staging, async H2D, on-device cast / transpose / row subset, device-resident cache)
returns exactly the tensors of the ORACLE's restatement of the reference's __getitem__ (oracle/dataset_oracle.py:getitem,
dataset/implicit_dataset.py:24-56) under the same numpy random state -- also without the cache while the consumer's
stream is still busy with the previous sample; and one IF-Net training step consumes its batch."""
import os

import numpy as np
import pytest
import torch

from oracle import dataset_oracle as DO

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_device_loader_equals_cpu_dataset_and_feeds_a_step(tmp_path):
    import svr_amd  # noqa: F401
    from oracle import ifnet_oracle as O
    from svr_amd.data_processing import sample_io
    from svr_amd.dataset import DeviceSampleLoader, ImplicitDataset
    from svr_amd.trainer import ImplicitRefinementTrainer
    root = tmp_path / "data"
    (tmp_path / "splits" / "overfit").mkdir(parents=True)
    (tmp_path / "splits" / "overfit" / "train.txt").write_text("00000\n00001\n")
    DO.make_sample(root / "processed" / "overfit" / "00000", grid_from=os.path.join(GOLD, "ref_depth_grid.npz"), seed=1)
    DO.make_sample(root / "processed" / "overfit" / "00001", seed=2)
    ds = ImplicitDataset("train", root, 400, "overfit", splits_root=tmp_path / "splits")
    loader = DeviceSampleLoader(ds, cache=True)
    for visit in range(2):                                   # second visit: served from the device-resident cache
        for idx in (0, 1):
            np.random.seed(7 + idx + 10 * visit)
            got = loader.get(idx)
            np.random.seed(7 + idx + 10 * visit)
            ref = DO.getitem(ds.sample_folder(idx), ds.data[idx], 400)          # the oracle, not the package's own dataset
            assert set(got) == set(ref) and got["name"] == ref["name"]
            for k in ("grid", "points", "input", "occupancies", "target"):
                assert got[k].is_cuda and got[k].dtype == ref[k].dtype and torch.equal(got[k].cpu(), ref[k]), (visit, idx, k)
            np.random.seed(7 + idx + 10 * visit)
            mine = ds[idx]                                                      # ... and the CPU mirror agrees with both
            assert all(torch.equal(mine[k], ref[k]) for k in ("grid", "points", "input", "occupancies", "target"))
    assert set(loader.cache) == {"00000", "00001"}
    # out-of-range indices are flagged, not read
    rows = torch.arange(12, dtype=torch.float64, device="cuda").view(4, 3)
    out, bad = sample_io.subsample_rows(rows, torch.tensor([0, 3, 4, -1]))
    assert int(bad) == 1 and torch.equal(out[:2].cpu(), rows[[0, 3]].float().cpu()) and float(out[2:].abs().sum()) == 0
    # a collated batch drives the trainer (BatchNorm needs > 1 value per channel: batch of 2)
    np.random.seed(0)
    batch = loader.batch([0, 1])
    assert tuple(batch["input"].shape) == (2, 1, 139, 104, 112) and tuple(batch["points"].shape) == (2, 800, 3)
    tr = ImplicitRefinementTrainer()
    tr.ifnet.load_state_dict(O.name_seeded_state(128), strict=False)
    tr = tr.cuda().train()
    loss = tr.training_step(batch, 0)["loss"]
    loss.backward()
    assert torch.isfinite(loss) and tr.ifnet.fc_out.weight.grad is not None


def test_uncached_loader_while_the_consumer_stream_is_busy(tmp_path):
    """cache=False: the decoded device arrays of sample A are dropped when get() returns, while A's subset kernels are
    still queued behind earlier work on the consumer's stream; decoding sample B must not overwrite them (the arrays are
    allocated under the copy stream: record_stream keeps their blocks out of its pool until the consumer has passed).
    Every sample equals the oracle's __getitem__ under the same numpy state."""
    import svr_amd  # noqa: F401
    from svr_amd.dataset import DeviceSampleLoader, ImplicitDataset
    root = tmp_path / "data"
    (tmp_path / "splits" / "overfit").mkdir(parents=True)
    (tmp_path / "splits" / "overfit" / "train.txt").write_text("00000\n00001\n00002\n")
    for i in range(3):
        DO.make_sample(root / "processed" / "overfit" / f"0000{i}", dims=(40, 36, 44), n_pts=60000, seed=10 + i)
    ds = ImplicitDataset("train", root, 20000, "overfit", splits_root=tmp_path / "splits")
    loader = DeviceSampleLoader(ds, cache=False)
    busy = torch.randn(6144, 6144, device="cuda")
    for rep in range(3):
        for _ in range(12):
            busy = (busy @ busy).clamp_(-1, 1)                # tens of ms of queued work in front of the subset kernels
        got = []
        for idx in (0, 1, 2):
            np.random.seed(100 * rep + idx)
            got.append(loader.get(idx))                       # no synchronisation between the samples
        torch.cuda.synchronize()
        for idx in (0, 1, 2):
            np.random.seed(100 * rep + idx)
            ref = DO.getitem(ds.sample_folder(idx), ds.data[idx], 20000)
            for k in ("grid", "points", "input", "occupancies", "target"):
                assert torch.equal(got[idx][k].cpu(), ref[k]), (rep, idx, k)
    assert loader.cache is None
