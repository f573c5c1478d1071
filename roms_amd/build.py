"""Build recipes for the native parts of roms_amd (in-tree, explicit hipcc).

libroms_hip.so   HIP/gfx950 kernels + C ABI (include/roms_hip.h)      <- csrc/*.cpp
libroms_host.so  Fortran host driver (ISO_C_BINDING -> libroms_hip.so) <- host/*.f90
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
INCLUDE = os.path.join(HERE, "..", "include")
LIB_HIP = os.path.join(HERE, "libroms_hip.so")
LIB_HOST = os.path.join(HERE, "libroms_host.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: no FMA contraction, so results are bit-comparable with the reference's
# plain IEEE arithmetic (memory-bound kernels: no measurable cost).
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "hip",
            "-Wno-unused-value", "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-Wno-unused-function"]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build_hip(force=False, verbose=False, defs=()):
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".cpp"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(INCLUDE, "roms_hip.h"))
    objdir = os.path.join(CSRC, "obj")
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(objdir, f[:-4] + ".o")
        objs.append(obj)
        if force or _newer([src] + hdrs, obj):
            jobs.append([HIPCC] + HIPFLAGS + list(defs) + ["-c", src, "-o", obj])
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _newer(objs, LIB_HIP):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_HIP] + objs)
    return LIB_HIP


def build_host(force=False, verbose=False):
    """Fortran host driver (needs amdflang; the image has it at /opt/rocm/bin/amdflang)."""
    mk = os.path.join(HOST, "Makefile")
    if not os.path.exists(mk):
        return None
    cmd = ["make", "-s", "-C", HOST] + (["-B"] if force else [])
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_HOST


if __name__ == "__main__":
    build_hip(force="-f" in sys.argv, verbose=True)
    build_host(force="-f" in sys.argv, verbose=True)
