"""CPU-only evidence for the gradient tolerances in tests/test_gpu_ifnet_parity.py: the CPU oracle (= the
reference math) re-run with weights perturbed by 2e-7 relative moves gradient elements by up to 5e-3 of the
tensor max, 2.7e-3 in norm (ReLU-mask / max-pool arg-max flips at ~0 pre-activations); fc_out stays at 1e-6."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from tests import _golden as G
from oracle import ifnet_oracle as O
torch.manual_seed(0)
for case in ["cfg1","odd"]:
    z = G.load("ifnet_"+case)
    net_res, x, pts, occ = G.ifnet_inputs(z)
    res=[]
    for trial in range(3):
        base=G.state(net_res, z=z)
        if trial>0:
            base={k:(v*(1+2e-7*torch.randn_like(v)) if not O.is_buffer(k) else v) for k,v in base.items()}
        st = O.make_leaf_state(base)
        out = O.training_step(st, {"input":x,"points":pts,"occupancies":occ}, net_res)
        out["loss"].backward()
        res.append({k:v.grad.double() for k,v in st.items() if v.grad is not None})
    for k in res[0]:
        b=res[0][k]
        es=[]
        for t in (1,2):
            a=res[t][k]
            es.append(((a-b).abs().max().item()/b.abs().max().item(), abs(a.norm()-b.norm()).item()/b.norm().item(), ((a-b).abs().median()/b.abs().max()).item()))
        print(case, f"{k:48s}", " ".join(f"{e[0]:.1e}/{e[1]:.1e}/{e[2]:.1e}" for e in es))
