import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from tests import _golden as G
import svr_amd
from svr_amd.model import IFNet
from svr_amd.trainer import bce_with_logits_sum_mean
for case in ["cfg1","odd","b3"]:
    z = G.load("ifnet_"+case)
    net_res, x, pts, occ = G.ifnet_inputs(z)
    m = IFNet(net_res=net_res); m.load_state_dict(G.state(net_res, z=z), strict=False); m=m.cuda().train()
    logits = m(x.cuda(), pts.cuda())
    loss = bce_with_logits_sum_mean(logits, occ.cuda())
    loss.backward()
    print(case, "loss", loss.item(), float(z["loss"]))
    for name,p in m.named_parameters():
        e = G.rel_err(G.sample(p.grad), z["grad/"+name])
        n = abs(p.grad.double().norm().item()-float(z["grad_norm/"+name]))/(float(z["grad_norm/"+name])+1e-30)
        print(f"  {name:50s} {e:.2e} {n:.2e}")
