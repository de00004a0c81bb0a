"""Host (enqueue) time per training step vs wall time: how close is the step to being launch bound?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
import svr_amd  # noqa: F401
from svr_amd.dp import DataParallelTrainer
from svr_amd.trainer import ImplicitRefinementTrainer

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
trainer = ImplicitRefinementTrainer().to(dev).train()
opt = torch.optim.Adam(trainer.ifnet.parameters(), lr=trainer.hparams.lr, fused=True)
dp = DataParallelTrainer(trainer, optimizer=opt)
batch = B.synth_batch(103, 8, 128, 50000, dev)
for _ in range(3):
    dp.step(batch)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    dp.step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
