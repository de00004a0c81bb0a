"""Where does the HOST lose 40-80 ms in some eager config-5 steps?  cProfile over 20 steps with the autograd engine on the
calling thread (so the custom Functions' backward code is seen), cyclic GC off; top entries by own time."""
import cProfile
import gc
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import svr_amd  # noqa: F401
from oracle import ifnet_oracle as O
from oracle import scene_oracle as S
from svr_amd.trainer import SceneNetTrainer, default_hparams

torch.autograd.set_multithreading_enabled(False)
g = torch.Generator(device="cpu").manual_seed(105)
B, N = 4, 50000
rgb = torch.rand(B, 3, 256, 256, generator=g) * 2 - 1
target = torch.rand(B, 240, 320, generator=g) * 5 + 0.5
pts = torch.rand(B, N, 3, generator=g) - 0.5
occ = (torch.rand(B, N, generator=g) < 0.5).float()
tr = SceneNetTrainer(default_hparams(), dims=(128, 128, 128))
tr.unet.load_state_dict(S.name_seeded_like(tr.unet.state_dict(), 1.0, "unet."), strict=False)
tr.ifnet.load_state_dict(O.name_seeded_state(128), strict=False)
tr = tr.cuda().train()
opt = tr.configure_optimizers()[0][0]
batch = {"rgb": rgb.cuda(), "depthmap_target": target.cuda(), "points": pts.cuda(), "occupancies": occ.cuda()}
marks = []


def step():
    opt.zero_grad(set_to_none=True)
    loss = tr.training_step(batch, 0)["loss"]
    loss.backward()
    opt.step()
    e = torch.cuda.Event()
    e.record()
    marks.append(e)
    if len(marks) > 2:
        marks[-3].synchronize()
    return loss.detach()


for _ in range(8):
    step()
torch.cuda.synchronize()
gc.collect()
gc.disable()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
