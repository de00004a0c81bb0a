import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import svr_amd
from svr_amd.trainer import ImplicitRefinementTrainer
from svr_amd.graphs import GraphedStep
from oracle import ifnet_oracle as O
from bench import synth_batch
tr = ImplicitRefinementTrainer(); tr.ifnet.load_state_dict(O.name_seeded_state(128), strict=False); tr = tr.cuda().train()
opt = torch.optim.Adam(tr.ifnet.parameters(), lr=1e-4, capturable=True)
batch = synth_batch(103, 8, 128, 50000, "cuda")
gs = GraphedStep(tr, opt, batch)
for _ in range(3): gs.run(batch)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): l = gs.run(batch)
torch.cuda.synchronize(); print("graph ms/step", (time.perf_counter()-t0)/20*1e3, float(l))
