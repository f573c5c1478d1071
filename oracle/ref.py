"""ctypes binding of oracle/_ref/libromsref_<app>.so (the reference's own Fortran,
built by oracle/ref/build_ref.sh in this container).  TEST INFRASTRUCTURE.

Only one application/configuration can live in a process (module-level Fortran
state), so fixture generators run one configuration per process.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def available(app):
    return os.path.exists(os.path.join(HERE, "_ref", f"libromsref_{app}.so")) and \
        os.path.isdir("/root/reference/ROMS")


class Ref:
    def __init__(self, app, ipar, rpar):
        self.L = C.CDLL(os.path.join(HERE, "_ref", f"libromsref_{app}.so"))
        self.L.ref_field.restype = C.c_long
        self.L.ref_call.restype = C.c_int
        ip = (C.c_int * 64)(*ipar)
        rp = (C.c_double * 96)(*rpar)
        self.L.ref_configure(ip, rp)
        b = self.bounds(0)
        self.LBi, self.UBi, self.LBj, self.UBj = b[:4]
        self.ni = self.UBi - self.LBi + 1
        self.nj = self.UBj - self.LBj + 1
        self._buf = np.zeros(1, dtype=np.float64)

    def initial(self):
        self.L.ref_initial()

    def bounds(self, tile=0):
        b = (C.c_int * 64)()
        self.L.ref_get_bounds(tile, b)
        return list(b)

    def table(self, which, n):
        a = np.zeros(max(n, 32))
        self.L.ref_get_table(C.c_int(which), a.ctypes.data_as(C.c_void_p))
        return a[:n].copy()

    def get(self, name, nmax=None):
        if nmax is None:
            nmax = 64 * self.ni * self.nj * 64
        if self._buf.size < nmax:
            self._buf = np.zeros(nmax)
        n = self.L.ref_field(name.encode(), C.c_int(0), self._buf.ctypes.data_as(C.c_void_p))
        if n < 0:
            raise KeyError(name)
        return self._buf[:n].copy()

    def has(self, name):
        """Fields behind cpp options the application does not define (CURVGRID, BULK_FLUXES ...) are absent."""
        try:
            self.get(name)
            return True
        except KeyError:
            return False

    def put(self, name, a):
        a = np.ascontiguousarray(a, dtype=np.float64).ravel()
        n = self.L.ref_field(name.encode(), C.c_int(1), a.ctypes.data_as(C.c_void_p))
        if n != a.size:
            raise ValueError(f"{name}: size {a.size} != {n}")

    def set_stepping(self, iic, iif, nstp, nnew, nrhs, kstp, knew, krhs, predictor, time, indx1=0):
        idx = (C.c_int * 16)(iic, iif, nstp, nnew, nrhs, kstp, knew, krhs, int(predictor), indx1)
        self.L.ref_set_stepping(idx, C.c_double(time))

    def get_stepping(self):
        """dict of the reference's mod_stepping indices and time(ng)."""
        idx = (C.c_int * 16)()
        tm = C.c_double()
        self.L.ref_get_stepping(idx, C.byref(tm))
        names = ["iic", "iif", "nstp", "nnew", "nrhs", "kstp", "knew", "krhs", "predictor", "indx1"]
        d = {n: int(idx[k]) for k, n in enumerate(names)}
        d["time"] = tm.value
        return d

    def main3d(self, nsteps=1):
        """nsteps passes of main3d's STEP_LOOP made of the reference's own kernels (ref_glue.F90:ref_main3d).
        Returns [avgke, avgpe, avgkp, volume, max_speed, iic, time, indx1] after the last step."""
        dg = (C.c_double * 16)()
        self.L.ref_main3d(C.c_int(nsteps), dg)
        return list(dg)[:8]

    def call(self, name):
        r = self.L.ref_call(name.encode())
        if r != 0:
            raise KeyError(name)
