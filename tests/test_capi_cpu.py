"""CPU: the C-ABI library builds, loads, and exports every symbol include/svr_hip.h declares
(no compute calls without a GPU); host-side layout logic."""
import ctypes
import os
import re

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(REPO, "include", "svr_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(svr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    import svr_amd
    lib = ctypes.CDLL(svr_amd._lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/svr_hip.h but not exported"
    bound = set(svr_amd._lib.SIGNATURES)
    assert bound == set(names), bound ^ set(names)
    l = svr_amd._lib.lib()
    assert l.svr_version() >= 100
    assert l.svr_linear_bwd_weight_workspace(1000, 256, 2592) > 0


def test_feature_layout_is_a_permutation_of_the_reference_rows():
    import svr_amd  # noqa: F401
    from svr_amd.ops import FeatureLayout
    for chans in ([1, 16, 32, 64, 128, 128], [1, 64, 128, 128]):
        lay = FeatureLayout(chans)
        perm = lay.reference_permutation()
        used = perm[perm >= 0]
        assert lay.row_stride % 32 == 0 and lay.width == 7 * sum(chans)
        assert sorted(used.tolist()) == list(range(7 * sum(chans)))
        assert all(c % 16 == 0 for l, c in enumerate(lay.col) if chans[l] >= 4)


def test_module_state_dict_matches_reference_names():
    import svr_amd  # noqa: F401
    from svr_amd.model import IFNet
    from oracle import ifnet_oracle as O
    for res in (128, 32):
        m = IFNet(net_res=res)
        keys = {k for k in m.state_dict() if "num_batches" not in k}
        assert keys == set(O.param_shapes(res)), keys ^ set(O.param_shapes(res))
        for k, shp in O.param_shapes(res).items():
            assert tuple(m.state_dict()[k].shape) == tuple(shp), k
    assert sum(p.numel() for p in IFNet().parameters()) == 2550881
    assert sum(p.numel() for p in IFNet(net_res=32).parameters()) == 2954049


def test_hip_path_refuses_cpu_tensors():
    import pytest
    import svr_amd  # noqa: F401
    from svr_amd.model import IFNet
    with pytest.raises(RuntimeError):
        IFNet()(torch.zeros(1, 1, 16, 16, 16), torch.zeros(1, 4, 3))
