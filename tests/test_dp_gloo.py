"""CPU, world_size 2, gloo: the data-parallel plumbing (flat gradient bucket, one sum
all-reduce, mean, parameter broadcast) used by bench.py at N>1.  The HIP path cannot run
here, so a small torch module stands in for the trainer; the collective code is the same."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(6, 16), nn.ReLU(), nn.Linear(16, 1))
        self.hparams = type("H", (), {"lr": 1e-2})()

    def configure_optimizers(self):
        return [torch.optim.Adam(self.parameters(), lr=self.hparams.lr)], []

    def training_step(self, batch, batch_idx):
        z = self.net(batch["x"]).squeeze(-1)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(z, batch["y"], reduction="none").sum(-1).mean()
        return {"loss": loss}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import svr_amd  # noqa: F401
    from svr_amd.dp import DataParallelTrainer
    torch.manual_seed(100 + rank)                       # different init per rank: broadcast must fix it
    model = _Toy()
    dp = DataParallelTrainer(model)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 5, 6, generator=g)               # global batch 4 samples x 5 points
    y = (torch.rand(4, 5, generator=g) < 0.5).float()
    shard = slice(rank * 2, rank * 2 + 2)
    out = dp.step({"x": x[shard], "y": y[shard]})
    q.put((rank, dp.bucket.flat.numpy().copy(), [p.detach().numpy().copy() for p in model.parameters()], float(out["loss"].detach())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_global_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the global batch, starting from rank 0's init
    torch.manual_seed(100)
    ref = _Toy()
    opt = ref.configure_optimizers()[0][0]
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 5, 6, generator=g)
    y = (torch.rand(4, 5, generator=g) < 0.5).float()
    loss = ref.training_step({"x": x, "y": y}, 0)["loss"]
    loss.backward()
    flat_ref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    opt.step()
    for rank, flat, params, l in res:
        assert torch.allclose(torch.from_numpy(flat), flat_ref, rtol=1e-5, atol=1e-7), rank   # mean of shard means == global mean
        for a, b in zip(params, ref.parameters()):
            assert torch.allclose(torch.from_numpy(a), b.detach(), rtol=1e-5, atol=1e-7)
    assert abs(0.5 * (res[0][3] + res[1][3]) - float(loss)) < 1e-5


def test_grad_bucket_views_alias_one_flat_buffer():
    import svr_amd  # noqa: F401
    from svr_amd.dp import GradBucket
    m = _Toy()
    b = GradBucket(list(m.parameters()))
    assert b.numel == sum(p.numel() for p in m.parameters())
    m.training_step({"x": torch.randn(2, 3, 6), "y": torch.ones(2, 3)}, 0)["loss"].backward()
    off = 0
    for p in m.parameters():
        assert p.grad.data_ptr() == b.flat.data_ptr() + 4 * off            # still a view after backward
        assert torch.equal(p.grad.reshape(-1), b.flat[off:off + p.numel()])
        off += p.numel()
    b.zero()
    assert all(float(p.grad.abs().sum()) == 0 for p in m.parameters())


def _ifnet_worker(rank, world, port, q):
    """The REAL IF-Net parameter list (CPU tensors: only shapes matter for the plumbing) through the same bucket /
    broadcast / all-reduce / Adam code bench.py runs at N > 1; the gradients are synthetic (the HIP step needs a GPU)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import svr_amd  # noqa: F401
    from svr_amd.dp import DataParallelTrainer
    from svr_amd.trainer import ImplicitRefinementTrainer
    torch.manual_seed(500 + rank)                       # different init + different BN buffers per rank
    tr = ImplicitRefinementTrainer()
    with torch.no_grad():
        for b in tr.buffers():
            if b.dtype.is_floating_point:
                b.add_(float(rank))
    opt = torch.optim.Adam(tr.ifnet.parameters(), lr=1e-3)
    dp = DataParallelTrainer(tr, optimizer=opt)
    params = list(tr.parameters())
    # what a backward would leave behind: fresh gradient tensors, rank dependent
    dp.bucket.detach_grads()
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (1 + i % 3))
    dp.bucket.collect_grads()
    aliased = all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(params, dp.bucket.views))
    dp.bucket.all_reduce_mean()
    expect = torch.cat([torch.full((p.numel(),), 1.5 * (1 + i % 3)) for i, p in enumerate(params)])
    reduced_ok = bool(torch.equal(dp.bucket.flat, expect))
    opt.step()
    digest = torch.cat([p.detach().reshape(-1)[:7] for p in params] + [b.detach().float().reshape(-1)[:3] for b in tr.buffers()])
    q.put((rank, dp.bucket.numel, dp.bucket.flat.numel() * dp.bucket.flat.element_size(), aliased, reduced_ok,
           digest.numpy().copy(), len(params)))
    dist.barrier()
    dist.destroy_process_group()


def test_ifnet_parameter_bucket_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ifnet_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, numel, nbytes, aliased, reduced_ok, digest, nparams in res:
        assert numel == 2550881 and nbytes == 4 * 2550881        # one 10.2 MB f32 bucket = every IF-Net parameter
        assert nparams == 36                                     # 9 convs + 5 BN + 4 fc, weight + bias each
        assert aliased and reduced_ok, rank
    # broadcast made parameters AND BatchNorm buffers identical, and the identical reduced gradients keep them so
    assert (res[0][5] == res[1][5]).all()


def test_scene_trainer_bucket_is_the_config5_allreduce_size():
    """config 5 all-reduces UNet + project.sigma + IF-Net = 12 346 021 floats (49.4 MB) in ONE bucket."""
    import svr_amd  # noqa: F401
    from svr_amd.dp import GradBucket
    from svr_amd.trainer import SceneNetTrainer, default_hparams
    tr = SceneNetTrainer(default_hparams(miopen_benchmark=False))
    b = GradBucket(list(tr.parameters()), device="cpu")
    assert b.numel == 12346021 and b.flat.numel() * 4 == 49384084
    n_unet = sum(p.numel() for p in tr.unet.parameters())
    assert n_unet == 12346021 - 2550881 - 3          # 9 795 137 UNet parameters, 3 for project.sigma
    off = 0
    for p in tr.parameters():
        assert p.grad.data_ptr() == b.flat.data_ptr() + 4 * off
        off += p.numel()
