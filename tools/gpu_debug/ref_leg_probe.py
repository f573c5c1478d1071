"""Probe (GPU box): run bench.reference_leg's child by hand with faulthandler, to see where the reference library stops."""
import faulthandler, sys, os, threading, time, json
faulthandler.enable()
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
def run():
    import bench
    from tests import refdrive as rd
    cs = bench.params_for("benchmark1", ntimes=400)
    print("configure", flush=True)
    R = rd.reference("benchmark", cs)
    print("initial done", flush=True)
    t0 = time.perf_counter(); R.main3d(1); print("step", time.perf_counter() - t0, flush=True)
threading.stack_size(1 << 30)
t = threading.Thread(target=run); t.start(); t.join()
