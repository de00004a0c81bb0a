"""GPU parity of the whole IF-Net HIP path against the reference's own outputs
(tests/golden/ifnet_*.npz, generated from the imported reference) and the CPU oracle.

Tolerances (SURVEY.md §7 hard part 1: rel = max|a-b| / max|b|):
  logits 1e-4 (north_star), loss 1e-5, gradients 5e-4 (f32 atomics order), Adam step 1e-5."""
import numpy as np
import pytest
import torch

from oracle import ifnet_oracle as O
from tests import _golden as G

pytestmark = pytest.mark.gpu


def _model(net_res, z):
    import svr_amd  # noqa: F401
    from svr_amd.model import IFNet
    m = IFNet(net_res=net_res)
    missing = m.load_state_dict(G.state(net_res, z=z), strict=False)
    assert not missing.unexpected_keys and all("num_batches" in k for k in missing.missing_keys)
    return m.cuda().train()


@pytest.mark.parametrize("case", G.IFNET_CASES + ["res32"])
def test_training_step_matches_reference(case):
    from svr_amd.trainer import bce_with_logits_sum_mean
    z = G.load("ifnet_" + case)
    net_res, x, pts, occ = G.ifnet_inputs(z)
    m = _model(net_res, z)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    logits = m(x.cuda(), pts.cuda())
    assert logits.shape == (x.shape[0], pts.shape[1])
    e = G.rel_err(logits.detach().cpu().numpy(), z["logits"])
    print(case, "logits rel err", e)
    assert e < 1e-4
    loss = bce_with_logits_sum_mean(logits, occ.cuda())
    assert abs(loss.item() - float(z["loss"])) < 1e-5 * abs(float(z["loss"]))
    opt.zero_grad()
    loss.backward()
    for name, p in m.named_parameters():
        assert p.grad is not None, name
        e = G.rel_err(G.sample(p.grad), z["grad/" + name])
        n = abs(p.grad.double().norm().item() - float(z["grad_norm/" + name])) / (float(z["grad_norm/" + name]) + 1e-30)
        assert e < 5e-4 and n < 5e-4, (name, e, n)
    for name, b in m.named_buffers():
        if "running" in name:
            assert G.rel_err(b.cpu().numpy(), z["buf/" + name]) < 1e-5, name
    opt.step()
    for name, p in m.named_parameters():
        assert G.rel_err(G.sample(p), z["adam/" + name]) < 1e-5, name
    m.eval()
    with torch.no_grad():
        ev = m(x.cuda(), pts.cuda())
    assert G.rel_err(ev.cpu().numpy(), z["logits_eval_after_step"]) < 2e-4


def test_extractor_reference_layout_and_features8():
    z = G.load("ifnet_cfg1")
    net_res, x, pts, _ = G.ifnet_inputs(z)
    m = _model(net_res, z)
    with torch.no_grad():
        f = m.ifnet_feature_extractor(x.cuda(), pts[:, :8].cuda())
    assert tuple(f.shape) == tuple(z["features8"].shape)
    assert G.rel_err(f.cpu().numpy(), z["features8"]) < 1e-5


def test_input_and_point_gradients_match_oracle():
    """config-5 style: gradient flows into the input grid and (subsample_points>0) the points."""
    z = G.load("ifnet_b3")
    net_res, x, pts, occ = G.ifnet_inputs(z)
    m = _model(net_res, z)
    xg = (x * 0.7 + 0.1).cuda().requires_grad_(True)
    pg = pts.cuda().requires_grad_(True)
    logits = m(xg, pg)
    w = torch.linspace(-1, 1, logits.numel()).view_as(logits)
    (logits * w.cuda()).sum().backward()
    st = O.make_leaf_state(G.state(net_res, z=z))
    xc = (x * 0.7 + 0.1).requires_grad_(True)
    pc = pts.clone().requires_grad_(True)
    ref = O.ifnet_forward(st, xc, pc, net_res, training=True)
    (ref * w).sum().backward()
    assert G.rel_err(logits.detach().cpu().numpy(), ref.detach().numpy()) < 1e-4
    assert G.rel_err(xg.grad.cpu().numpy(), xc.grad.numpy()) < 5e-4
    assert G.rel_err(pg.grad.cpu().numpy(), pc.grad.numpy()) < 2e-3


def test_cpu_tensors_fail_loudly():
    import svr_amd  # noqa: F401
    from svr_amd.model import IFNet
    m = IFNet()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 16, 16, 16), torch.zeros(1, 4, 3))
