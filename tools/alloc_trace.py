"""Which allocations still reach hipMalloc in the steady state of the benched step?  (GPU box: python tools/alloc_trace.py)

Runs bench.py's set-up + warm-up steps, then records the caching allocator's history over 20 more steps and prints every
`segment_alloc` (= hipMalloc) with its size and the innermost frames of this package on its Python stack."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

import bench  # noqa: E402
import svr_amd  # noqa: F401,E402
from oracle import ifnet_oracle as O  # noqa: E402
from svr_amd.dp import DataParallelTrainer  # noqa: E402
from svr_amd.trainer import ImplicitRefinementTrainer  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    trainer = ImplicitRefinementTrainer()
    trainer.ifnet.load_state_dict(O.name_seeded_state(128), strict=False)
    trainer = trainer.to(dev).train()
    opt = torch.optim.Adam(trainer.ifnet.parameters(), lr=1e-4, fused=True)
    dp = DataParallelTrainer(trainer, optimizer=opt)
    batch = bench.synth_batch(103, 8, 128, 50000, dev)
    for _ in range(2):
        dp.step(batch)
    torch.cuda.synchronize()
    for _ in range(5):
        dp.step(batch)
    torch.cuda.synchronize()
    torch.cuda.memory._record_memory_history(max_entries=2000000, stacks="python")
    m0 = torch.cuda.memory_stats(dev)
    for _ in range(int(os.environ.get("STEPS", "20"))):
        dp.step(batch)
    torch.cuda.synchronize()
    m1 = torch.cuda.memory_stats(dev)
    snap = torch.cuda.memory._snapshot()
    torch.cuda.memory._record_memory_history(enabled=None)
    print("hipMalloc calls:", m1["num_device_alloc"] - m0["num_device_alloc"], "hipFree:", m1["num_device_free"] - m0["num_device_free"],
          "reserved growth:", m1["reserved_bytes.all.current"] - m0["reserved_bytes.all.current"])
    out = []
    for trace in snap["device_traces"]:
        for i, ev in enumerate(trace):
            if ev["action"] in ("segment_alloc", "segment_free"):
                # the request that caused it is the next 'alloc' event of the same stream
                frames = ev.get("frames", [])
                if not frames:
                    for ev2 in trace[i + 1:i + 4]:
                        if ev2["action"] == "alloc":
                            frames = ev2.get("frames", [])
                            break
                mine = [f"{os.path.relpath(f['filename'], ROOT)}:{f['line']} {f['name']}" for f in frames
                        if ROOT in f["filename"]][:6]
                out.append({"action": ev["action"], "size": ev["size"], "stream": ev.get("stream"), "frames": mine})
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
