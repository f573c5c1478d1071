"""GLS_MIXING soak: 400 steps of UPWELLING 128x64x20 (k-epsilon, Kantha-Clayson; HSIMT salinity) on the GPU against the
same run on the CPU-emulated kernels (tests/emu: bit-identical to the oracle, tests/test_kernels_emu.py): relative RMS
differences of the circulation and of the turbulent fields every 100 steps"""
import sys, os, subprocess
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np
from tests import util
from roms_amd import hostlib
EMU = os.path.join(ROOT, "tests", "emu")
if not os.path.exists(os.path.join(EMU, "libroms_host_emu.so")):
    subprocess.check_call(["bash", os.path.join(EMU, "build_emu.sh")])
cs = util.cases.upwelling_gls(Lm=128, Mm=64, N=20, hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT"))
cs["ninfo"] = 0
names = ("zeta", "u", "v", "t", "Akv", "tke", "gls")
snap = {}
for side in ("gpu", "emu"):
    H = hostlib.Host(params=cs) if side == "gpu" else hostlib.Host(params=cs, lib_path=os.path.join(EMU, "libroms_host_emu.so"), hip_lib_path=util.EMU_LIB)
    ctx = H.device_init()
    for blk in range(4):
        H.run(100)
        snap[side, blk] = {n: ctx.download(n).copy() for n in names}
    H.finalize()
for blk in range(4):
    a, b = snap["gpu", blk], snap["emu", blk]
    print("step", 100 * (blk + 1), " ".join("%s %.1e" % (n, util.relrms(a[n], b[n])) for n in names), "max Akv %.3e" % b["Akv"].max(), flush=True)
