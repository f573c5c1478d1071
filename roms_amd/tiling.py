"""One ROMS tile per GPU/process: partition, halo transport, gathering.

The reference runs one MPI rank per tile (rank = tile = itile + jtile*NtileI, get_bounds.F:972) and
moves ghost strips with mp_exchange2d/3d/4d.  Here every rank builds the (cheap, host-side) global
set-up with the Fortran host, uploads only its tile's window to its GPU and installs a halo
transport on the device context:

  "peer"  the library's mailbox transport: neighbours' receive slots mapped over xGMI (hipIpc), two launches per
          exchange point, no send/receive calls
  "auto"  (default for world > 1; ROMS_HIP_TRANSPORT overrides) "peer" if every rank can map its neighbours and the
          index-coded probe exchange (roms_hip_exchange_probe) arrives intact everywhere, else "rccl"
  "rccl"  the library's built-in RCCL send/recv on its own HIP stream (multi-GPU runs); the RCCL
          unique id is created on rank 0 and broadcast through torch.distributed
  "dist"  a callback that moves the strips with torch.distributed isend/irecv on host tensors --
          used with the gloo backend and the CPU-emulated kernels by tests/test_tiles.py
  "dist_staged"  the same callback for the real HIP build: the device strips are staged through host
          memory (hipMemcpy) around the isend/irecv.  Slow; it exists so that the device-side
          pack/unpack kernels and the strip geometry can be tested with several ranks sharing ONE
          GPU (RCCL refuses two ranks on one device)

Weak scaling: `weak=True` replicates the case's Lm x Mm tile NtileI x NtileJ times, i.e. the global
grid is (Lm*NtileI) x (Mm*NtileJ); `weak=False` splits the case's own grid.
"""
import ctypes as C
import traceback

import numpy as np

from . import hiplib, hostlib


def partition(world):
    """NtileI x NtileJ for `world` ranks: as square as possible, the longer side along xi
    (1 -> 1x1, 2 -> 2x1, 4 -> 2x2, 8 -> 4x2)."""
    nj = int(np.floor(np.sqrt(world)))
    while world % nj:
        nj -= 1
    return world // nj, nj


class TiledRun:
    def __init__(self, cs, rank=0, world=1, device=0, dist=None, transport=None, weak=True,
                 host_lib=None, hip_lib=None, tiles=None, self_exchange=False):
        self.rank, self.world, self.dist = rank, world, dist
        self.NtileI, self.NtileJ = tiles or partition(world)
        assert self.NtileI * self.NtileJ == world
        g = dict(cs)
        if weak:
            g["Lm"], g["Mm"] = cs["Lm"] * self.NtileI, cs["Mm"] * self.NtileJ
        g["NtileI"], g["NtileJ"] = self.NtileI, self.NtileJ
        self.global_Lm, self.global_Mm = g["Lm"], g["Mm"]
        self.case = g
        self.host = hostlib.Host(params=g, lib_path=host_lib, hip_lib_path=hip_lib)
        self.nfast = self.host.dims["nfast"]
        # self_exchange (test aid, single tile): the periodic direction is closed through the halo
        # transport -- the tile is its own west and east neighbour -- instead of a local copy
        import os
        if self_exchange:
            assert world == 1
            os.environ["ROMS_HIP_SELF_EXCHANGE"] = "1"
        try:
            self.ctx = self.host.device_init(device, tile=rank, start=False)
        finally:
            os.environ.pop("ROMS_HIP_SELF_EXCHANGE", None)
        self._cb = None
        self.probe_log = []
        self._x0 = None
        if self_exchange:
            transport = transport or "rccl"
            if transport == "rccl":
                self._install_rccl()
            elif transport == "peer":
                self._install_peer()
            else:
                self._install_dist(staged=(transport == "dist_staged"))
        if world > 1:
            transport = transport or os.environ.get("ROMS_HIP_TRANSPORT", "auto")
            if transport == "auto":
                transport = self._install_auto()
            elif transport == "rccl":
                self._install_rccl()
            elif transport == "peer":
                self._install_peer()
            elif transport == "dist":
                self._install_dist()
            elif transport == "dist_staged":
                self._install_dist(staged=True)
            else:
                raise ValueError(transport)
        self.transport = transport
        if transport == "peer" and (world > 1 or self_exchange):
            self.rim_check()
        self.host.start()

    # ------------------------------------------------------------------ transports
    def _all_ok(self, ok):
        """True when every rank says so."""
        import torch
        dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def probe(self, reps=4, soak=200):
        """roms_hip_exchange_probe on every rank: index-coded planes through the installed transport; then (round 4)
        roms_hip_exchange_soak: `soak` exchange points back to back without a host synchronisation, each repetition coded
        and verified on the device -- ordering under a stream of exchanges, which the single probe exchanges cannot see."""
        L = self.ctx.L
        rc = L.roms_hip_exchange_probe(self.ctx.h, reps)
        if rc == 0 and soak:
            rc = L.roms_hip_exchange_soak(self.ctx.h, int(soak))
        return rc == 0, (L.roms_hip_last_error() or b"").decode() if rc else ""

    def rim_check(self, reps=4):
        """roms_hip_rim_probe on every rank (round 6): the rim planes through which the barotropic launches hand their rim to the
        neighbouring ranks themselves, index-coded and verified on the device.  If ANY rank fails, EVERY rank keeps the exchange
        launches instead (roms_hip_rim_disable) -- the decision is collective, the run goes on either way."""
        L = self.ctx.L
        if self.world > 1:
            self.dist.barrier()              # (every rank's slab is mapped and zeroed before anyone publishes)
        rc = L.roms_hip_rim_probe(self.ctx.h, reps)
        why = (L.roms_hip_last_error() or b"").decode() if rc else ""
        ok = self._all_ok(rc == 0) if self.world > 1 else rc == 0
        self.probe_log.append({"rim_planes": "ok" if ok else "failed: exchanges kept", **({"why": why} if why else {})})
        if not ok:
            L.roms_hip_rim_disable(self.ctx.h)
            if why:
                import sys
                print(f"[roms_amd] rank {self.rank}: {why}; the barotropic launches keep their exchanges", file=sys.stderr, flush=True)
        return ok

    def _install_auto(self):
        """The mailbox transport where it works -- every rank could map its neighbours' slabs and the index-coded probe
        planes arrived intact on all of them -- otherwise the RCCL send/recv groups, probed the same way.  Both are
        device-to-device transports of this library; which one runs is reported in `self.transport`."""
        import sys
        ok, why = True, ""
        try:
            self._install_peer()
        except hiplib.RomsHipError as e:
            ok, why = False, str(e)
        if self._all_ok(ok):
            ok, why = self.probe()
            if self._all_ok(ok):
                self.probe_log.append({"transport": "peer", "ok": True})
                return "peer"
        self.probe_log.append({"transport": "peer", "ok": False, "why": why or "another rank failed"})
        if self.rank == 0 or why:
            print(f"[roms_amd] rank {self.rank}: mailbox transport not usable ({why or 'another rank failed'}); using RCCL send/recv",
                  file=sys.stderr, flush=True)
        self.ctx._ck(self.ctx.L.roms_hip_comm_reset(self.ctx.h))
        self._install_rccl()
        ok, why = self.probe()
        if not self._all_ok(ok):
            self.probe_log.append({"transport": "rccl", "ok": False, "why": why or "another rank failed"})
            raise hiplib.RomsHipError("exit_flag=2: no usable halo transport: " + "; ".join(
                f"{p['transport']}: {p.get('why', 'ok')}" for p in self.probe_log))
        self.probe_log.append({"transport": "rccl", "ok": True})
        return "rccl"

    def rccl_ranks(self):
        """Ranks of the library's RCCL communicator (ncclCommCount), 0 when the built-in RCCL transport is not installed."""
        fn = getattr(self.ctx.L, "roms_hip_rccl_ranks", None)
        return int(fn(self.ctx.h)) if fn is not None else None

    def exchanges_per_step(self, nsteps):
        """exchange points per main3d pass since the first call of this function (roms_hip_exchange_count)"""
        n = int(self.ctx.L.roms_hip_exchange_count(self.ctx.h))
        if self._x0 is None:
            self._x0 = n
            return None
        return (n - self._x0) / max(nsteps, 1)

    def _install_rccl(self):
        L = self.ctx.L
        uid = (C.c_ubyte * 128)()
        if self.rank == 0:
            self.ctx._ck(L.roms_hip_rccl_unique_id(uid))
        if self.world > 1:
            import torch
            t = torch.tensor(list(uid), dtype=torch.uint8, device="cuda" if torch.cuda.is_available() else "cpu")
            self.dist.broadcast(t, 0)
            raw = bytes(t.cpu().tolist())
        else:
            raw = bytes(uid)
        self.ctx._ck(L.roms_hip_comm_rccl(self.ctx.h, raw, self.world, self.rank))

    def _install_peer(self):
        """The mailbox transport (include/roms_hip.h:roms_hip_comm_peer): every rank exports its slab, the 128-byte
        blobs travel through torch.distributed, each rank maps its neighbours' slabs."""
        L = self.ctx.L
        blob = (C.c_ubyte * 128)()
        rc = L.roms_hip_peer_export(self.ctx.h, blob)      # (a failure is raised after the collective below)
        err = (L.roms_hip_last_error() or b"").decode() if rc else ""
        if self.world > 1:
            import torch
            dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
            mine = torch.tensor(list(blob), dtype=torch.uint8, device=dev)
            every = [torch.empty_like(mine) for _ in range(self.world)]
            self.dist.all_gather(every, mine)
            raw = b"".join(bytes(t.cpu().tolist()) for t in every)
        else:
            raw = bytes(blob)
        if rc == 0:
            rc = L.roms_hip_comm_peer(self.ctx.h, raw, self.world, self.rank)
            err = (L.roms_hip_last_error() or b"").decode() if rc else ""
        if self.world > 1:
            self.dist.barrier()          # nobody starts before every slab is mapped
        if rc:
            raise hiplib.RomsHipError(f"exit_flag={rc}: {err}")

    def _install_dist(self, staged=False):
        import torch
        dist = self.dist
        hip = C.CDLL("libamdhip64.so") if staged else None
        if hip is not None:
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

        def view(ptr, n):
            return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(n,)))

        def cb(user, ns, sp, sb, sc, st, nr, rp, rb, rc, rt):
            try:
                if staged:      # device pointers: stage through host arrays
                    sh = [np.empty(sc[k]) for k in range(ns)]
                    rh = [np.empty(rc[k]) for k in range(nr)]
                    for k in range(ns):
                        if hip.hipMemcpy(sh[k].ctypes.data, sb[k], 8 * sc[k], 2) != 0:     # hipMemcpyDeviceToHost
                            return 2
                    reqs = [dist.irecv(torch.from_numpy(rh[k]), src=rp[k], tag=rt[k]) for k in range(nr)]
                    reqs += [dist.isend(torch.from_numpy(sh[k]), dst=sp[k], tag=st[k]) for k in range(ns)]
                    for r in reqs:
                        r.wait()
                    for k in range(nr):
                        if hip.hipMemcpy(rb[k], rh[k].ctypes.data, 8 * rc[k], 1) != 0:     # hipMemcpyHostToDevice
                            return 2
                    return 0
                reqs = [dist.irecv(view(rb[k], rc[k]), src=rp[k], tag=rt[k]) for k in range(nr)]
                reqs += [dist.isend(view(sb[k], sc[k]), dst=sp[k], tag=st[k]) for k in range(ns)]
                for r in reqs:
                    r.wait()
                return 0
            except Exception:       # a Python exception must not unwind through the C frames
                traceback.print_exc()
                return 1

        self._cb = hiplib.EXCHANGE_FN(cb)          # keep the trampoline alive
        self.ctx._ck(self.ctx.L.roms_hip_set_exchange(self.ctx.h, self._cb, None))

    # ------------------------------------------------------------------ running
    def step(self, n=1, kernels=False, check=False):
        """n main3d passes.  In a multi-tile run the library leaves the blow-up test (diag.F:510-540) to the
        globally reduced numbers: check=True makes every rank run it together after the n steps."""
        self.host.run(n, kernels=kernels)
        if check:
            self.check()

    def sync(self):
        self.ctx.sync()

    def step_times(self, n):
        """n more main3d passes with a HIP event at every step boundary (roms_hip_step_timing): their durations in ms."""
        L = self.ctx.L
        self.ctx._ck(L.roms_hip_step_timing(self.ctx.h, int(n)))
        self.host.run(n)
        buf = (C.c_double * n)()
        k = L.roms_hip_step_times(self.ctx.h, buf, n)
        L.roms_hip_step_timing(self.ctx.h, 0)
        return [float(buf[i]) for i in range(k)]

    def diag(self):
        """Global diagnostics of diag.F (energies, volume, maximum speed / Courant number)."""
        d = np.array(self.ctx.diag(raw=True))
        if self.world > 1:
            import torch
            dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
            sums = torch.tensor([d[3], d[12], d[13]], dtype=torch.float64, device=dev)
            maxs = torch.tensor([d[4], d[11]], dtype=torch.float64, device=dev)
            self.dist.all_reduce(sums)
            self.dist.all_reduce(maxs, op=self.dist.ReduceOp.MAX)
            vol, kes, pes = [float(x) for x in sums.cpu()]
            spd, C_ = [float(x) for x in maxs.cpu()]
        else:
            vol, kes, pes, spd, C_ = d[3], d[12], d[13], d[4], d[11]
        return {"avgke": kes / vol, "avgpe": pes / vol, "avgkp": (kes + pes) / vol, "volume": vol,
                "maxspeed": spd, "max_C": C_}

    def check(self):
        d = self.diag()
        if not (np.isfinite(d["avgke"]) and np.isfinite(d["avgpe"])) or d["maxspeed"] > 20.0:
            raise hiplib.RomsHipError(f"blow-up (exit_flag=1): {d}")
        return d

    def gather(self, name):
        """The global field `name` (interior + domain boundary points of every tile) on every rank,
        shaped (planes, nj_global, ni_global) with the reference's index origin (LBj, LBi)."""
        t = self.host.tile
        a = self.ctx.download(name)
        ni, nj = t["UBi"] - t["LBi"] + 1, t["UBj"] - t["LBj"] + 1
        a = a.reshape(-1, nj, ni)
        d = self.host.dims
        G = np.zeros((a.shape[0], d["UBj"] - d["LBj"] + 1, d["UBi"] - d["LBi"] + 1))
        it, jt = t["tile"] % self.NtileI, t["tile"] // self.NtileI
        # own range: interior, extended on domain edges to what the reference defines there -- the boundary points of a
        # closed edge, the ghost points -2:0 and Lm+1:Lm+Nghost of a periodic one -- not to the padding line of an even
        # Lm / Mm behind them (mod_param.F:1633-1636: allocated, never computed)
        ng = d["Nghost"]
        i0 = max(t["LBi"], -2 if d["EWper"] else 0) if it == 0 else t["Istr"]
        i1 = min(t["UBi"], d["Lm"] + (ng if d["EWper"] else 1)) if it == self.NtileI - 1 else t["Iend"]
        j0 = max(t["LBj"], -2 if d["NSper"] else 0) if jt == 0 else t["Jstr"]
        j1 = min(t["UBj"], d["Mm"] + (ng if d["NSper"] else 1)) if jt == self.NtileJ - 1 else t["Jend"]
        G[:, j0 - d["LBj"]:j1 - d["LBj"] + 1, i0 - d["LBi"]:i1 - d["LBi"] + 1] = \
            a[:, j0 - t["LBj"]:j1 - t["LBj"] + 1, i0 - t["LBi"]:i1 - t["LBi"] + 1]
        if self.world > 1:
            import torch
            dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
            T = torch.from_numpy(G).to(dev)
            self.dist.all_reduce(T)
            G = T.cpu().numpy()
        return G

    # ---- output: every rank calls the writer, rank 0 writes (wrt_his / wrt_rst over mp_gather2d/3d)
    def _install_gather(self):
        if getattr(self, "_gather_cb", None) is not None or self.world == 1:
            return
        GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_char_p, C.POINTER(C.c_double), C.c_long)

        def cb(name, buf, n):
            try:
                G = self.gather(name.decode())
                if self.rank == 0:
                    if G.size != n:
                        return 8
                    np.ctypeslib.as_array(buf, shape=(n,))[:] = G.ravel()
                return 0
            except Exception:
                traceback.print_exc()
                return 2

        self._gather_cb = GATHER_FN(cb)
        self.host.lib.roms_host_set_gather(C.cast(self._gather_cb, C.c_void_p))

    def write_his(self):
        self._install_gather()
        self.host.write_his()

    def write_rst(self):
        self._install_gather()
        self.host.write_rst()

    def advance(self, nsteps, final=False):
        """nsteps steps with the history / restart records NHIS and NRST of the case ask for"""
        self._install_gather()
        self.host.advance(nsteps, final=final)

    def get_state(self, path="", rec=0):
        """restart every tile from a restart file (each rank reads it and uploads its window)"""
        self.host.get_state(path, rec)

    def close(self):
        if self.world > 1 and getattr(self, "transport", None) == "peer":
            # the neighbours' kernels store into this rank's slab: nobody frees it while a neighbour may still be running
            self.ctx.sync()
            self.dist.barrier()
        self.host.finalize()
