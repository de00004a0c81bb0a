"""List kernels whose global loads are followed directly by s_waitcnt vmcnt(0) (a load inside a branch is
waited for at the end of the branch: serialised memory latency).  usage: python tools/scan_waits.py [file.hip ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "single-view-3d-reconstruction_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
for f in files:
    extra = ["-ffp-contract=off"] if f in ("gather.hip", "projection.hip") else []
    out = f"/tmp/{f}.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", *extra, "-I" + CSRC,
                    "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", os.path.join(CSRC, f), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    cur, stats = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1); stats[cur] = [0, 0]
        if cur and ("global_load" in l or "buffer_load" in l):
            stats[cur][0] += 1
            # a vmcnt(0) within the next few instructions and before any other load = this load is waited for alone
            seen = 0
            for j in range(i + 1, min(i + 40, len(lines))):
                t = lines[j].strip()
                if not t or t.startswith(";") or t.endswith(":") or t.startswith("."):
                    continue
                if "global_load" in t or "buffer_load" in t:
                    break
                if "vmcnt(0)" in t:
                    stats[cur][1] += 1
                    break
                seen += 1
                if seen >= 8:
                    break
    for k, (a, b) in stats.items():
        if b:
            print(f"{f}: {k[:90]} loads={a} immediately-waited={b}")
