"""GPU: DeviceSampleLoader (pinned staging, async H2D, on-device cast / transpose / row subset, device-resident cache)
returns exactly the tensors of the CPU dataset mirror -- and so of the reference's __getitem__ -- under the same numpy
random state; and one IF-Net training step consumes its batch."""
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
            ref = ds[idx]
            for k in ("grid", "points", "input", "occupancies", "target"):
                assert got[k].is_cuda and got[k].dtype == ref[k].dtype and torch.equal(got[k].cpu(), ref[k]), (visit, idx, k)
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
