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


def _doc_stub_namespace():
    """Execute the ctypes struct definitions of INTEGRATION.md section 3 (the stub a maintainer would paste)."""
    txt = open(os.path.join(REPO, "INTEGRATION.md")).read()
    block = re.search(r"```python\nimport ctypes as C, torch\n(.*?)```", txt, flags=re.S).group(1)
    structs = block.split("lib.svr_gather_trilinear_fwd.restype")[0]
    structs = "\n".join(l for l in structs.splitlines() if not l.startswith("lib = "))
    ns = {"C": ctypes}
    exec(structs, ns)                                   # our own documentation, not reference code
    return ns


def test_header_structs_binding_and_doc_stub_cannot_drift(tmp_path):
    """include/svr_hip.h is compiled on its own (plain C, gcc) and its sizeof / offsetof of svr_level and
    svr_gather_desc are compared with (a) the ctypes mirrors in _lib.py, (b) the struct sizes the library itself
    reports (svr_sizeof_*), (c) the stub documented in INTEGRATION.md -- a field added to one of them fails here,
    on the CPU, instead of as a wild device read on the GPU."""
    import subprocess
    import svr_amd
    L, Gd = svr_amd._lib.Level, svr_amd._lib.GatherDesc
    src = tmp_path / "layout.c"
    lf = [n for n, _ in L._fields_]
    gf = [n for n, _ in Gd._fields_]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "svr_hip.h"', 'int main(void) {',
             '  printf("svr_level %zu\\n", sizeof(svr_level));', '  printf("svr_gather_desc %zu\\n", sizeof(svr_gather_desc));']
    lines += [f'  printf("svr_level.{f} %zu\\n", offsetof(svr_level, {f}));' for f in lf]
    lines += [f'  printf("svr_gather_desc.{f} %zu\\n", offsetof(svr_gather_desc, {f}));' for f in gf]
    lines += ['  printf("SVR_MAX_LEVELS %d\\n", SVR_MAX_LEVELS);', '  return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)],
                   check=True)
    out = dict(l.rsplit(" ", 1) for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    out = {k: int(v) for k, v in out.items()}
    assert out["SVR_MAX_LEVELS"] == svr_amd._lib.SVR_MAX_LEVELS
    assert out["svr_level"] == ctypes.sizeof(L) and out["svr_gather_desc"] == ctypes.sizeof(Gd)
    for f in lf:
        assert out[f"svr_level.{f}"] == getattr(L, f).offset, f
    for f in gf:
        assert out[f"svr_gather_desc.{f}"] == getattr(Gd, f).offset, f
    lib = svr_amd._lib.lib()
    assert lib.svr_sizeof_level() == out["svr_level"] and lib.svr_sizeof_gather_desc() == out["svr_gather_desc"]
    # the documented stub
    ns = _doc_stub_namespace()
    for name, mirror in (("Level", L), ("GatherDesc", Gd)):
        doc = ns[name]
        assert ctypes.sizeof(doc) == ctypes.sizeof(mirror), name
        assert [(n, getattr(doc, n).offset) for n, _ in doc._fields_] == \
               [(n, getattr(mirror, n).offset) for n, _ in mirror._fields_], name


def test_hand_written_kernels_do_not_spill():
    """Register report of the gfx950 build (hipcc -Rpass-analysis=kernel-resource-usage, kept per object file by build.py):
    no hand-written kernel may use scratch memory, except the two listed ones at their known sizes.  A source change that
    makes the register allocator spill shows up as a multiple of the kernel time on the GPU only (the fused gather -> fc_0
    kernel: 260 B / lane of scratch = 11.7 instead of 2.6 ms) -- this catches it where the code is compiled."""
    import importlib
    import __graft_entry__ as ge
    ge.build()
    b = importlib.import_module("single-view-3d-reconstruction_amd.build")
    usage = b.resource_usage()
    if not usage:                    # objects built before the report existed: rebuild once
        b.build(force=True)
        usage = b.resource_usage()
    own = {k: v for k, v in usage.items() if "rocprim" not in k and "hipcub" not in k}
    assert len(own) > 100 and any("gather_fc0_kernel" in k for k in own)
    known = {"linear_nt_h3_kernelILi64ELi0ELi2": 32, "conv3d_bwd_weight_x3_kernel": 16}
    bad = {}
    for k, v in own.items():
        limit = max([lim for name, lim in known.items() if name in k], default=0)
        if v.get("scratch", 0) > limit:
            bad[k] = v
    assert not bad, bad
    fc0 = [v for k, v in own.items() if "gather_fc0_kernel" in k][0]
    assert fc0["scratch"] == 0 and fc0["vgprs"] <= 128 and fc0["occupancy"] >= 4, fc0


def test_module_copies_and_pickles_after_a_step_left_scratch_state():
    """a training step leaves a torch.cuda.Event in ext._prepared.ready and GBs in ext._arena; the reference
    nn.Module can be deep-copied / pickled, so this one must be too -- the copy gets fresh, empty scratch state."""
    import copy
    import io
    import pickle

    import torch
    from svr_amd.arena import StepArena
    from svr_amd.model import IFNet
    m = IFNet(net_res=128)
    ext = m.ifnet_feature_extractor
    ext._prepared.ready = torch.cuda.Event()          # what _prepare_weights_async leaves behind
    ext._points_ready = torch.cuda.Event()
    ext._arena._bufs["x"] = torch.zeros(4)
    for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        e2 = clone.ifnet_feature_extractor
        assert isinstance(e2._arena, StepArena) and e2._arena is not ext._arena and e2._arena.nbytes() == 0
        assert e2._prepared is not ext._prepared and e2._prepared.ready is None
        # _stages / _param_list of the copy point at the copy's own modules
        assert e2._stages[0][0][0] is e2.conv_in and e2._param_list[0] is e2.conv_in.weight
        assert e2._param_list[0] is not ext._param_list[0]
        for (k, a), (k2, b) in zip(m.state_dict().items(), clone.state_dict().items()):
            assert k == k2 and torch.equal(a, b)
    buf = io.BytesIO()
    torch.save(m, buf)
    ext.release_step_buffers()
    assert ext._arena.nbytes() == 0
