"""Worker of tests/test_tiles.py: one rank of a multi-tile run with the CPU-emulated kernels and the
torch.distributed (gloo) halo transport.  Started by torch.distributed.run; rank 0 writes the gathered
fields to the npz file given on the command line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)


def main():
    out, spec = sys.argv[1], json.loads(sys.argv[2])
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from roms_amd import tiling
    from tests import util
    if spec.get("workload"):                      # a BASELINE configuration at its own size (bench.py's parameter sets)
        import bench
        cs = bench.params_for(spec["workload"], ntimes=spec["steps"])
    else:
        cs = util.case_for(spec["tag"], **spec.get("kw", {}))
    cs["ninfo"] = 0
    cs.update(spec.get("case_update", {}))        # output keywords (NHIS, HISNAME, Hout ...), tests/test_output.py
    emu = os.path.join(ROOT, "tests", "emu")
    if spec.get("gpu"):      # real HIP build, all ranks on cuda:0, strips staged through the host
        run = tiling.TiledRun(cs, rank=rank, world=world, device=0, dist=dist, transport=spec.get("transport", "dist_staged"), weak=False,
                              tiles=tuple(spec["tiles"]))
    else:
        run = tiling.TiledRun(cs, rank=rank, world=world, dist=dist, transport=spec.get("transport_cpu", "dist"), weak=False,
                              tiles=tuple(spec["tiles"]), host_lib=os.path.join(emu, "libroms_host_emu.so"),
                              hip_lib=os.path.join(emu, "libroms_hip_emu.so"))
    if spec.get("probe"):                          # the transport's self-check before anything else moves
        ok, why = run.probe(3)
        assert ok, why
    nx0 = run.ctx.L.roms_hip_exchange_count(run.ctx.h)        # (behind the probe: what the steps themselves exchange)
    if spec.get("restart_from"):                   # every rank reads the restart file and uploads its window
        run.get_state(spec["restart_from"], 0)
    if spec.get("advance"):                        # steps with the history / restart records of output.F
        run.advance(spec["steps"], final=True)
        run.host.close_output()
    else:
        run.step(spec["steps"], kernels=spec.get("kernels", False))
    res = {n: run.gather(n) for n in spec["fields"]}
    d = run.diag()
    nx = run.ctx.L.roms_hip_exchange_count(run.ctx.h)
    if rank == 0:
        print("TRANSPORT", getattr(run, "transport", None), flush=True)
        np.savez(out, nexchanges=nx, nexchanges_steps=nx - nx0, diag=np.array([d["avgke"], d["avgpe"], d["volume"], d["maxspeed"]]), **res)
    run.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
