"""Recipe for oracle/_ref/: the reference's one native component, compiled from the source where it lies.

TEST INFRASTRUCTURE ONLY.  data_processing/libmesh/triangle_hash.pyx (86 lines of Cython: the 2-D triangle hash behind
check_mesh_contains) is translated by the Cython installed in this image and compiled with g++ directly -- the
reference's own setup.py is not run, nothing is copied into the repository, and the outputs (generated .cpp, .so) go
only to oracle/_ref/ (git-ignored; it still travels to the GPU box with the snapshot).  Needs /root/reference: a
no-op where that does not exist (the GPU box).  Used by oracle/gen_golden_mesh.py to run the reference's inside_mesh.py.
"""
import glob
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("SVR_REFERENCE", "/root/reference")
PYX = os.path.join(REF, "data_processing", "libmesh", "triangle_hash.pyx")


def built():
    so = glob.glob(os.path.join(OUT, "triangle_hash*.so"))
    return so[0] if so else None


def build(force=False):
    """-> path of oracle/_ref/triangle_hash*.so, or None when the reference tree is absent."""
    if not os.path.exists(PYX):
        return built()
    so = built()
    if so and not force and os.path.getmtime(so) >= os.path.getmtime(PYX):
        return so
    import numpy
    os.makedirs(OUT, exist_ok=True)
    cpp = os.path.join(OUT, "triangle_hash_ref.cpp")
    subprocess.run([sys.executable, "-m", "cython", "--cplus", "-3", "-o", cpp, PYX], check=True)
    so = os.path.join(OUT, "triangle_hash" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-w", "-I", sysconfig.get_paths()["include"], "-I", numpy.get_include(),
                    cpp, "-o", so], check=True)
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
