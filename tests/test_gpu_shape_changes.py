"""One model, a sequence of training steps whose shapes change from step to step (batch, grid, number of points) -- the
reference's loop does this whenever the last batch of an epoch is short (dataset/implicit_dataset.py + DataLoader) or the
validation grid differs.  Everything that is cached per module (step arena, prepared weight planes, scatter-form ring,
side streams, per-level events) must follow: every step is compared with the same step on a fresh model that uses none of
the caches (SVR_NO_ARENA-equivalent switches, serial backward)."""
import pytest
import torch

from oracle import ifnet_oracle as O

pytestmark = pytest.mark.gpu


def _fresh():
    import svr_amd  # noqa: F401
    from svr_amd.model import IFNet
    m = IFNet(net_res=128)
    m.load_state_dict(O.name_seeded_state(128), strict=False)
    return m.cuda().train()


def _batch(seed, B, D, N):
    g = torch.Generator().manual_seed(seed)
    dims = D if isinstance(D, tuple) else (D, D, D)
    x = (torch.rand(B, 1, *dims, generator=g) < 0.05).float().cuda()
    pts = (torch.rand(B, N, 3, generator=g) - 0.5).cuda()
    occ = (torch.rand(B, N, generator=g) < 0.5).float().cuda()
    return x, pts, occ


def test_steps_of_changing_shape_on_one_model():
    from svr_amd.model import ifnet as ifn
    from svr_amd.trainer import bce_with_logits_sum_mean
    shapes = [(2, 32, 3000), (2, 16, 500), (2, 32, 3000), (3, 24, 777), (2, (19, 18, 21), 1000), (2, 32, 3000), (2, 32, 3000),
              (2, 32, 3000), (2, 32, 3000)]          # (the repeated shape lets the fixed-lag scatter-form decision switch forms)
    m = _fresh()
    grown = []
    for i, (B, D, N) in enumerate(shapes):
        x, pts, occ = _batch(100 + i, B, D, N)
        for p in m.parameters():
            p.grad = None
        logits = m(x, pts)
        loss = bce_with_logits_sum_mean(logits, occ)
        loss.backward()
        torch.cuda.synchronize()
        grown.append(m.ifnet_feature_extractor._arena.grown)
        assert not m.ifnet_feature_extractor._arena._leased
        # the same step without any of the per-module caches / stream overlap
        saved = (ifn.USE_ARENA, ifn.OVERLAP_BACKWARD, ifn.PREPARE_WEIGHTS_AHEAD, ifn.SORT_ON_SIDE_STREAM)
        ifn.USE_ARENA = ifn.OVERLAP_BACKWARD = ifn.PREPARE_WEIGHTS_AHEAD = ifn.SORT_ON_SIDE_STREAM = False
        try:
            r = _fresh()
            for (_, b), (_, b2) in zip(m.named_buffers(), r.named_buffers()):
                pass
            rl = r(x, pts)
            rloss = bce_with_logits_sum_mean(rl, occ)
            rloss.backward()
            torch.cuda.synchronize()
        finally:
            ifn.USE_ARENA, ifn.OVERLAP_BACKWARD, ifn.PREPARE_WEIGHTS_AHEAD, ifn.SORT_ON_SIDE_STREAM = saved
        # training-mode BatchNorm: the forward does not depend on the running statistics the earlier steps moved
        assert torch.equal(logits, rl), (i, float((logits - rl).abs().max()))
        assert float(loss.detach()) == float(rloss.detach())
        for (n, p), (_, q) in zip(m.named_parameters(), r.named_parameters()):
            d = float((p.grad - q.grad).norm() / q.grad.norm().clamp_min(1e-30))
            assert d < 1e-4, (i, n, d)                 # float-atomic order and the scatter form taken; nothing else differs
    assert grown[-1] == grown[-2] == grown[-3]         # the arena has stopped growing once the shapes repeat
    ifn._pull_hint.clear()
