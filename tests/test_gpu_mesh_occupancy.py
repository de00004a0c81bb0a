"""GPU parity of the on-device occupancy labelling (C++ host triangle hash + HIP ray-parity kernel,
through the C ABI) against the reference's own outputs (tests/golden/mesh_*.npz) and the numpy oracle -- booleans, so
bit for bit -- plus the trainer branch that uses it (subsample_points != 0)."""
import os

import numpy as np
import pytest
import torch

from oracle import mesh_oracle as M
from tests.test_mesh_oracle_golden import CASES, load

pytestmark = pytest.mark.gpu


def _mesh(v, f):
    from types import SimpleNamespace
    return SimpleNamespace(vertices=v, faces=f)


@pytest.mark.parametrize("tag", CASES)
def test_check_mesh_contains_matches_reference_outputs(tag):
    import svr_amd  # noqa: F401
    from svr_amd.data_processing.libmesh.inside_mesh import MeshIntersector, check_mesh_contains
    z, contains, holes = load(tag)
    res = int(z["resolution"])
    c, h = check_mesh_contains(_mesh(z["vertices"], z["faces"]), z["points"], res)          # numpy in -> numpy out
    assert c.dtype == np.bool_ and np.array_equal(c, contains) and np.array_equal(h, holes)
    pts = torch.from_numpy(z["points"]).cuda()                                               # device in -> device out
    c2, h2 = check_mesh_contains(_mesh(z["vertices"], z["faces"]), pts, res)
    assert c2.is_cuda and np.array_equal(c2.cpu().numpy(), contains) and np.array_equal(h2.cpu().numpy(), holes)
    # the C++ host hash equals the reference's hash layout (cell = res * x + y, triangles in index order)
    mi = MeshIntersector(_mesh(z["vertices"], z["faces"]), res)
    oh = M.MeshIntersector(z["vertices"], z["faces"], res)
    assert np.array_equal(mi._cell_start.cpu().numpy(), oh._hash.start) and np.array_equal(mi._tri_ids.cpu().numpy()[:mi.n_entries], oh._hash.tris)
    assert np.array_equal(mi._tri.cpu().numpy(), oh._triangles) and np.array_equal(mi.scale, oh.scale) and np.array_equal(mi.translate, oh.translate)


@pytest.mark.parametrize("n_pts,res,dtype", [(200000, 512, np.float32), (50000, 256, np.float64), (1, 16, np.float32)])
def test_check_mesh_contains_large_random_vs_oracle(n_pts, res, dtype):
    import svr_amd  # noqa: F401
    from svr_amd.data_processing.libmesh.inside_mesh import check_mesh_contains
    rng = np.random.default_rng(n_pts)
    v1, f1 = M.icosphere(3, 0.3, (0.1, 0.0, -0.05))
    v2, f2 = M.torus(0.25, 0.08, 48, 20)
    v = np.concatenate([v1, v2 + [-0.1, 0.05, 0.2]])
    f = np.concatenate([f1, f2 + len(v1)])
    pts = rng.uniform(-0.55, 0.55, size=(n_pts, 3)).astype(dtype)
    pts[: n_pts // 50] = np.round(pts[: n_pts // 50], 2)              # lattice points: cell borders, edge-on rays
    c, h = check_mesh_contains(_mesh(v, f), torch.from_numpy(pts).cuda(), res)
    oc, oh = M.check_mesh_contains(v, f, pts, res)
    assert np.array_equal(c.cpu().numpy(), oc) and np.array_equal(h.cpu().numpy(), oh)
    if n_pts > 1:
        assert oc.any() and not oc.all()


def test_implicit_waterproofing_and_determine_occupancy_vs_oracle(tmp_path):
    import svr_amd  # noqa: F401
    from svr_amd.data_processing.implicit_waterproofing import implicit_waterproofing
    from svr_amd.data_processing.mesh_occupancies import determine_occupancy, load_obj
    rng = np.random.default_rng(21)
    v, f = M.box((-0.31, -0.2, -0.27), (0.3, 0.22, 0.25), drop_faces=2)        # open: the rotation rounds are exercised
    pts = rng.uniform(-0.45, 0.45, size=(20000, 3)).astype(np.float32)
    occ, holes = implicit_waterproofing(_mesh(v, f), torch.from_numpy(pts).cuda(), 128)
    o_occ, o_holes = M.implicit_waterproofing(v, f, pts, 128)
    assert o_holes.sum() < 0.2 * len(pts)
    # the rotated re-tests go through a float64 rotation whose summation order is not pinned (numpy BLAS vs elementwise):
    # compare away from the mesh's planes, where a last-bit difference cannot flip a test
    planes = np.concatenate([np.abs(pts - c).min(axis=1, keepdims=True) for c in ((-0.31, -0.2, -0.27), (0.3, 0.22, 0.25))], 1).min(1)
    safe = planes > 1e-4
    assert np.array_equal(occ.cpu().numpy()[safe], o_occ[safe]) and np.array_equal(holes.cpu().numpy()[safe], o_holes[safe])
    # determine_occupancy: .obj files on disk (batch['mesh'] is a list of paths, dataset/scene_net_data.py:87,97)
    sv, sf = M.icosphere(2, 30.0, (139 / 2, 104 / 2, 112 / 2))
    paths = []
    for i in range(2):
        p = tmp_path / f"mesh{i}.obj"
        with open(p, "w") as fh:
            for a in sv + i:
                fh.write(f"v {float(a[0])!r} {float(a[1])!r} {float(a[2])!r}\n")
            for t in sf:
                fh.write(f"f {t[0] + 1}//1 {t[1] + 1}//1 {t[2] + 1}//1\n")
        paths.append(str(p))
        m = load_obj(str(p))
        assert np.array_equal(m.vertices, sv + i) and np.array_equal(m.faces, sf)
    q = (rng.uniform(0, 1, size=(2, 3000, 3)) * [139, 104, 112]).astype(np.float32)
    pq, occ_q = determine_occupancy(paths, torch.from_numpy(q).cuda())
    o_pq, o_occ_q = M.determine_occupancy([(sv, sf), (sv + 1, sf)], q)
    assert np.array_equal(pq.cpu().numpy(), o_pq) and np.array_equal(occ_q.cpu().numpy(), o_occ_q) and not occ_q.any()
    _, occ_f = determine_occupancy(paths, torch.from_numpy(q).cuda(), reference_quirk=False)
    _, o_occ_f = M.determine_occupancy([(sv, sf), (sv + 1, sf)], q, reference_quirk=False)
    assert np.array_equal(occ_f.cpu().numpy(), o_occ_f) and 0.05 < o_occ_f.mean() < 0.95


def test_scene_trainer_with_subsampled_point_cloud(tmp_path):
    """trainer_scene_net.py:91-99,108-114 with subsample_points != 0: the whole projected point cloud is queried and
    labelled on the device; the step equals the default step on the concatenated points / occupancies."""
    import svr_amd  # noqa: F401
    from oracle import ifnet_oracle as O
    from oracle import scene_oracle as S
    from svr_amd.trainer import SceneNetTrainer, default_hparams
    scale, B, N = 4, 2, 200
    g = torch.Generator().manual_seed(9)
    rgb = torch.rand(B, 3, 256, 256, generator=g) * 2 - 1
    target = torch.rand(B, 240, 320, generator=g) * 5 + 0.5
    pts = torch.rand(B, N, 3, generator=g) - 0.5
    occ = (torch.rand(B, N, generator=g) < 0.5).float()
    sv, sf = M.icosphere(2, 45.0, (139 / 2, 104 / 2, 112 / 2))     # grid units, like the dataset's mesh.obj
    paths = []
    for i in range(B):
        p = tmp_path / f"m{i}.obj"
        with open(p, "w") as fh:
            fh.writelines(f"v {float(a[0])!r} {float(a[1])!r} {float(a[2])!r}\n" for a in sv)
            fh.writelines(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n" for t in sf)
        paths.append(str(p))
    batch = {"rgb": rgb.cuda(), "depthmap_target": target.cuda(), "points": pts.cuda(), "occupancies": occ.cuda(), "mesh": paths}

    def trainer(**kw):
        tr = SceneNetTrainer(default_hparams(scale_factor=scale, **kw))
        tr.unet.load_state_dict(S.name_seeded_like(tr.unet.state_dict(), 1.0, "unet."), strict=False)
        tr.ifnet.load_state_dict(O.name_seeded_state(128), strict=False)
        return tr.cuda().train()

    for quirk in (True, False):
        tr = trainer(subsample_points=1000, reference_occupancy_quirk=quirk)
        logits, depth, pc = tr(batch)
        assert tuple(logits.shape) == (B, 240 * 320 + N)            # the reference's condition never takes the subset branch
        occs = tr._occupancies(batch, pc)
        assert tuple(occs.shape) == (B, 240 * 320 + N) and torch.equal(occs[:, 240 * 320:], batch["occupancies"])
        o_pc, o_occ = M.determine_occupancy([(sv, sf)] * B, pc.detach().cpu().numpy(), reference_quirk=quirk, points_normalized=True)
        assert np.array_equal(occs[:, :240 * 320].cpu().numpy(), o_occ)
        assert (not quirk) == bool(o_occ.any())
        out = tr.training_step(batch, 0)
        out["loss"].backward()
        # same numbers as the default trainer fed with the concatenated points / labels
        tr0 = trainer()
        b0 = dict(batch, points=torch.cat((pc.detach(), batch["points"]), 1), occupancies=occs)
        l0 = tr0.training_step(b0, 0)["loss"]
        assert abs(out["loss"].item() - l0.item()) < 1e-5 * abs(l0.item())
        assert tr.last_log["train_mesh_ce_loss"].isfinite() and tr.unet.conv1.weight.grad is not None
