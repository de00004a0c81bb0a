"""Calibration of rocprofv3's FETCH_SIZE for GATHER-shaped reads (MI355X_MICROARCH.md: the x2 correction of
FETCH_SIZE is calibrated for 16 B/lane streaming reads, "other access widths are uncalibrated").  Reads with a KNOWN byte count:
  stream : a 2 GiB f32 buffer read once, coalesced (torch sum)                             -> bytes = 2 GiB
  gather : rows of C floats (C = 16 / 32 / 64 / 128: the pyramid levels' corner vectors) fetched at random, each row ONCE
           (a random permutation of all rows of a 2 GiB table: > L2 + Infinity Cache, no reuse) -> bytes = 2 GiB + indices
Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (tools/exp/calibrate_fetch.sh); the script prints the kernels' names and
the bytes each one must have fetched, the shell script divides the counter by them."""
import json
import sys

import torch

torch.manual_seed(0)
N = 1 << 29                                   # 2 GiB of f32
buf = torch.randn(N, device="cuda")
out = {"stream_bytes": N * 4, "cases": []}
for _ in range(3):
    s = buf.sum()
torch.cuda.synchronize()
for C in (16, 32, 64, 128):
    rows = N // C
    table = buf.view(rows, C)
    idx = torch.randperm(rows, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        g = table.index_select(0, idx)        # one gather kernel: reads every row once in random order, writes 2 GiB
    torch.cuda.synchronize()
    out["cases"].append({"C": C, "row_bytes": C * 4, "gather_read_bytes": N * 4 + rows * 8, "rows": rows})
    del g, idx
print(json.dumps(out))
