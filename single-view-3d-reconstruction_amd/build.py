"""Build libsvr_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so sits next to the
sources so it travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsvr_hip.so")

# file -> extra flags.  Index/weight arithmetic that must round like ATen's CPU kernels is
# compiled with FP contraction off (see gather.hip / projection.hip headers).
SOURCES = {
    "capi.cpp": [],
    "gather.hip": ["-ffp-contract=off"] + ["-D" + d for d in os.environ.get("SVR_GATHER_DEFS", "").split()],   # experiments
    "gather_fc0.hip": ["-ffp-contract=off"] + [f"-D{k}={os.environ[e]}" for k, e in (("FC_TM", "SVR_FC_TM"), ("FC_DEPTH", "SVR_FC_DEPTH"), ("FC_FMA", "SVR_FC_FMA"))
                                               if os.environ.get(e)]    # tile-shape experiments
                      + (["-DSVR_FC0_MEASURE"] if os.environ.get("SVR_FC0_MEASURE") else [])   # role switches (SVR_FC0_DBG)
                      + ["-D" + d for d in os.environ.get("SVR_FC_DEFS", "").split()]   # experiments: SVR_FC_DEFS="FC_PRIO=1 FC_PK=0"
                      + (["-fno-slp-vectorize"] if "FC_PK=0" in os.environ.get("SVR_FC_DEFS", "").split() else []),
    "sort.hip": [],
    "gemm.hip": [],
    "gemm_bf16x3.hip": [],
    "gemm_bf16x6.hip": [],
    "gemm_f16x3.hip": [],
    "conv3d.hip": [],
    "conv3d_bf16.hip": ([f"-DSVR_CONV_EXP={os.environ['SVR_CONV_EXP']}"] if os.environ.get("SVR_CONV_EXP") else [])   # measurement builds
                       + ["-D" + d for d in os.environ.get("SVR_CONV_DEFS", "").split()],
    "conv3d_bwdw_bf16.hip": [f"-DSVR_WG_EXP={os.environ['SVR_WG_EXP']}"] if os.environ.get("SVR_WG_EXP") else [],   # measurement builds
    "bn_pool.hip": [],
    "stage1.hip": [f"-DS1_EXP={os.environ['SVR_S1_EXP']}"] if os.environ.get("SVR_S1_EXP") else [],   # measurement builds
    "projection.hip": ["-ffp-contract=off"],
    "bf16_path.hip": ["-ffp-contract=off"],
    "mesh_occupancy.hip": ["-ffp-contract=off"],
    "sample_io.hip": [],
    "conv2d.hip": [],
    "conv2d_igemm.hip": [],
}
COMMON = ["-O3", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-std=c++17", "-I" + INCLUDE, "-I" + CSRC,
          # per-kernel register / scratch report into build/<file>.log (resource_usage() parses it: a kernel that starts to
          # spill is caught on the CPU, tests/test_capi_cpu.py -- round 3 lost 4.5x on the fused gather to 260 B/lane of scratch)
          "-Rpass-analysis=kernel-resource-usage"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "svr_hip.h"))
    objs, procs, cmd_of = [], [], {}
    for src, extra in SOURCES.items():
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        cmd = [_hipcc()] + COMMON + extra + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", path, "-o", obj]
        # an object built with other flags (a measurement build, SVR_*_EXP) is stale even if the source did not change
        flagfile = obj[:-2] + ".flags"
        same_flags = os.path.exists(flagfile) and open(flagfile).read() == " ".join(cmd)
        if force or not same_flags or _stale(obj, [path] + headers):
            if os.path.exists(flagfile):
                os.remove(flagfile)
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            cmd_of[obj] = cmd
            procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        with open(obj[:-2] + ".log", "wb") as f:
            f.write(out)
        with open(obj[:-2] + ".flags", "w") as f:
            f.write(" ".join(cmd_of[obj]))
    if force or procs or _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-lz"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return LIB


def resource_usage():
    """{kernel name (mangled): {"vgprs", "agprs", "scratch", "occupancy", "file"}} from the compile logs of the last build."""
    import re
    objdir = os.path.join(HERE, "build")
    out = {}
    for src in SOURCES:
        log = os.path.join(objdir, src.rsplit(".", 1)[0] + ".log")
        if not os.path.exists(log):
            continue
        cur = None
        for line in open(log, errors="replace"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = out.setdefault(m.group(1), {"file": src})
                continue
            if cur is None:
                continue
            for key, pat in (("vgprs", r"remark:\s+VGPRs: (\d+)"), ("agprs", r"remark:\s+AGPRs: (\d+)"),
                             ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
                m = re.search(pat, line)
                if m:
                    cur[key] = int(m.group(1))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
