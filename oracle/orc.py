"""ctypes binding of the CPU oracle (liborc.so).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product package roms_amd never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAXT, MAXW = 4, 512

# scheme codes / option bits: keep in sync with orc.h
A4, C2, C4, HSIMT, MPDATA, SPLINES, SPLIT_U3, U3 = range(1, 9)
SCHEMES = dict(A4=A4, C2=C2, C4=C4, HSIMT=HSIMT, MPDATA=MPDATA, SPLINES=SPLINES, SU3=SPLIT_U3, U3=U3)
UV_ADV, UV_COR, UV_VIS2, TS_DIF2, MIX_GEO_TS, CURVGRID, NONLIN_EOS, UV_QDRAG, LMD_MIXING, \
    BULK_FLUXES, SOLAR_SOURCE, ANA_VMIX, SALINITY, SPHERICAL, UV_LOGDRAG, MASKING = [1 << k for k in range(16)]
RADIATION_2D, PLAIN_VDIFF, PLAIN_VVISC, PRSGRD31 = 1 << 16, 1 << 17, 1 << 18, 1 << 19
WJ_GRADP = 1 << 27
GLS_MIXING = 1 << 25
PRSGRD40 = 1 << 26
MY25_MIXING = 1 << 28
MIX_ISO_TS = 1 << 29
APP_OVERFLOW = 1 << 30
GLS_FLAGS = {"CANUTO_A": 1, "CANUTO_B": 2, "KANTHA_CLAYSON": 4, "N2S2_HORAVG": 8, "RI_SPLINES": 16,
             "K_C2ADVECTION": 32, "K_C4ADVECTION": 64, "CHARNOK": 128, "CRAIG_BANNER": 256}
APP_UPWELLING, APP_BENCHMARK, APP_KELVIN, APP_SEAMOUNT, APP_GRAV_ADJ = 1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 24
# lateral boundary conditions (orc.h): edges, variables, kinds
IWEST, ISOUTH, IEAST, INORTH = range(4)
ISFSUR, ISUBAR, ISVBAR, ISUVEL, ISVVEL, ISTVAR = range(6)
NLBC = ISTVAR + MAXT
LBC_KINDS = dict(Clo=1, Per=2, Gra=3, Cla=4, Rad=5, RadNud=6, Che=7, Cha=8, Fla=9, Shc=10)


class Cfg(C.Structure):
    _fields_ = [
        ("Lm", C.c_int), ("Mm", C.c_int), ("N", C.c_int), ("NT", C.c_int), ("NAT", C.c_int),
        ("Nghost", C.c_int), ("LBi", C.c_int), ("UBi", C.c_int), ("LBj", C.c_int), ("UBj", C.c_int),
        ("NtileI", C.c_int), ("NtileJ", C.c_int), ("EWperiodic", C.c_int), ("NSperiodic", C.c_int),
        ("options", C.c_int), ("hadv", C.c_int * MAXT), ("vadv", C.c_int * MAXT),
        ("ntfirst", C.c_int), ("ntstart", C.c_int), ("ndtfast", C.c_int), ("nfast", C.c_int),
        ("dt", C.c_double), ("dtfast", C.c_double), ("weight", (C.c_double * (MAXW + 1)) * 2),
        ("rho0", C.c_double), ("g", C.c_double), ("lambda_", C.c_double), ("gamma2", C.c_double),
        ("Cp", C.c_double), ("R0", C.c_double), ("T0", C.c_double), ("S0", C.c_double),
        ("Tcoef", C.c_double), ("Scoef", C.c_double), ("hc", C.c_double), ("Vtransform", C.c_int),
        ("rdrg", C.c_double), ("rdrg2", C.c_double), ("Zob", C.c_double),
        ("Akt_bak", C.c_double * MAXT), ("Akv_bak", C.c_double), ("dstart", C.c_double),
        ("blk_ZQ", C.c_double), ("blk_ZT", C.c_double), ("blk_ZW", C.c_double),
        ("lmd_Jwt", C.c_int), ("cc1", C.c_double), ("cc2", C.c_double), ("cc3", C.c_double),
        ("lbc", (C.c_int * NLBC) * 4),
        ("FSobc_in", C.c_double * 4), ("FSobc_out", C.c_double * 4), ("M2obc_in", C.c_double * 4),
        ("M2obc_out", C.c_double * 4), ("M3obc_in", C.c_double * 4), ("M3obc_out", C.c_double * 4),
        ("Tobc_in", (C.c_double * 4) * MAXT), ("Tobc_out", (C.c_double * 4) * MAXT),
        ("gls_flags", C.c_int),
        ("gls_p", C.c_double), ("gls_m", C.c_double), ("gls_n", C.c_double), ("gls_Kmin", C.c_double),
        ("gls_Pmin", C.c_double), ("gls_cmu0", C.c_double), ("gls_c1", C.c_double), ("gls_c2", C.c_double),
        ("gls_c3m", C.c_double), ("gls_c3p", C.c_double), ("gls_sigk", C.c_double), ("gls_sigp", C.c_double),
        ("Akk_bak", C.c_double), ("Akp_bak", C.c_double), ("Zos", C.c_double), ("charnok_alpha", C.c_double),
        ("crgban_cw", C.c_double), ("obcfac", C.c_double), ("lbc_tke", C.c_int * 4), ("volcons", C.c_int),
    ]


class Step(C.Structure):
    _fields_ = [("iic", C.c_int), ("iif", C.c_int), ("nstp", C.c_int), ("nnew", C.c_int),
                ("nrhs", C.c_int), ("kstp", C.c_int), ("knew", C.c_int), ("krhs", C.c_int),
                ("indx1", C.c_int), ("predictor", C.c_int), ("time", C.c_double), ("tdays", C.c_double)]


def build(force=False):
    so = os.path.join(HERE, "liborc.so")
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        import fcntl
        with open(os.path.join(HERE, ".build.lock"), "w") as lock:      # one build at a time (pytest-xdist workers)
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-s", "-C", HERE, "-j4"])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(Cfg)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_field.restype = C.POINTER(C.c_double)
        L.orc_field.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_long)]
        L.orc_stepping.restype = C.POINTER(Step)
        L.orc_stepping.argtypes = [C.c_void_p]
        L.orc_config.restype = C.POINTER(Cfg)
        L.orc_config.argtypes = [C.c_void_p]
        L.orc_get_bounds.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_main3d_step.argtypes = [C.c_void_p]
        L.orc_start.argtypes = [C.c_void_p]
        L.orc_diag.argtypes = [C.c_void_p]
        L.orc_set_threads.argtypes = [C.c_void_p, C.c_int]
        _lib = L
    return _lib


TILE_KERNELS = ["set_depth", "set_massflux", "rho_eos", "set_vbc", "ana_vmix", "set_data", "omega",
                "set_zeta", "ini_zeta", "ini_fields", "pre_step3d", "prsgrd", "t3dmix2", "uv3dmix2",
                "rhs3d_tile", "rhs3d", "step2d", "step3d_uv", "step3d_t", "lmd_vmix", "bulk_flux", "set_diags"]


class Oracle:
    """One oracle state (global arrays, reference layout)."""

    def __init__(self, cfg: Cfg):
        self.L = lib()
        self.h = self.L.orc_create(C.byref(cfg))
        self.cfg = self.L.orc_config(self.h).contents
        self.step = self.L.orc_stepping(self.h).contents
        self.ni = self.cfg.UBi - self.cfg.LBi + 1
        self.nj = self.cfg.UBj - self.cfg.LBj + 1

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def field(self, name):
        """Flat numpy view (no copy) of a state array."""
        n = C.c_long()
        p = self.L.orc_field(self.h, name.encode(), C.byref(n))
        if n.value < 0:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,))

    def arr(self, name):
        """View shaped (..., nj, ni) (C order == Fortran (i,j,...) order reversed)."""
        a = self.field(name)
        if a.size % (self.ni * self.nj) == 0 and a.size >= self.ni * self.nj:
            return a.reshape(-1, self.nj, self.ni)
        return a

    def bounds(self, tile=0):
        b = (C.c_int * 64)()
        self.L.orc_get_bounds(self.h, tile, b)
        return list(b)[:54]

    def call(self, kernel, tile=None, *args):
        f = getattr(self.L, "orc_" + kernel)
        tiles = range(self.cfg.NtileI * self.cfg.NtileJ) if tile is None else [tile]
        for t in tiles:
            f(C.c_void_p(self.h), C.c_int(t), *[C.c_int(a) for a in args])

    def set_avg_window(self, nAVG, ntsAVG=1, nrrec=0, ntstart=1):
        """allocate the time-averaged fields ("avg_zeta" ... "avg_HvomT") and set the window of set_avg.F"""
        self.L.orc_set_avg_window(C.c_void_p(self.h), int(nAVG), int(ntsAVG), int(nrrec), int(ntstart))

    def set_dia_window(self, nDIA, ntsDIA=1, nrrec=0, ntstart=1, uv=False):
        """allocate the per-term tracer tendencies ("DiaTwrk", "DiaTrc", "dia_zeta") and set the window of set_diags.F;
        uv: the momentum terms too (DIAGNOSTICS_UV: "DiaU2wrk", "DiaU3wrk", "DiaRU", "DiaRUfrc", ... "DiaU2d", "DiaU3d")"""
        r = self.L.orc_set_dia_window(C.c_void_p(self.h), int(nDIA), int(ntsDIA), int(nrrec), int(ntstart))
        if r:
            raise ValueError("DIAGNOSTICS_TS with MPDATA tracers is not covered by the oracle")
        if uv:
            self.L.orc_set_diauv(C.c_void_p(self.h))

    def set_mix4(self, uv_vis4, ts_dif4):
        """biharmonic mixing along s-surfaces on (UV_VIS4 | TS_DIF4): fields "visc4_r", "visc4_p", "diff4" hold the square
        roots of the coefficients"""
        self.L.orc_set_mix4(C.c_void_p(self.h), int(uv_vis4), int(ts_dif4))

    def set_bkpp(self, on=True):
        """LMD_BKPP: the bottom boundary layer of the K-profile scheme (lmd_bkpp.F)"""
        self.L.orc_set_bkpp(C.c_void_p(self.h), int(bool(on)))

    def set_ddmix(self, on=True):
        """LMD_DDMIX: double-diffusive mixing (salt fingering, diffusive convection) added to Akt in lmd_vmix's interior scheme,
        lmd_vmix.F:360-428, with alfaobeta of rho_eos.F:454 | :794"""
        self.L.orc_set_ddmix(C.c_void_p(self.h), int(bool(on)))

    def set_prsgrd(self, scheme):
        """the pressure-gradient scheme of prsgrd.F:16-26 beyond the option bits: 42 = PJ_GRADPQ2 (prsgrd42.h), 44 = PJ_GRADPQ4
        (prsgrd44.h); 0 = what the ORC_PRSGRD* bits say"""
        self.L.orc_set_prsgrd(C.c_void_p(self.h), int(scheme))

    def set_clima(self, flags):
        """climatology nudging on: bit 0 = 3-D momentum (fields "uclm", "vclm", "M3nudgcof"), bit itrc = tracer itrc (fields
        "tclm", "Tnudgcof": N planes per tracer, tracer-major)"""
        self.L.orc_set_clima(C.c_void_p(self.h), int(flags))

    def set_geouv(self, on=True):
        """UV_VIS2 along geopotential surfaces (MIX_GEO_UV: uv3dmix2_geo.h in place of uv3dmix2_s.h)"""
        self.L.orc_set_geouv(C.c_void_p(self.h), int(bool(on)))

    def set_wetdry(self, Dcrit):
        """wetting and drying on (WET_DRY, wetdry.F): fields "rmask_wet", "umask_wet", "vmask_wet", "pmask_wet",
        "rmask_full" ..., "rmask_wet_avg"; call("wetdry_ini") sets the initial masks (initial.F:467)"""
        self.L.orc_set_wetdry.argtypes = [C.c_void_p, C.c_double]
        self.L.orc_set_wetdry(C.c_void_p(self.h), float(Dcrit))

    def start(self):
        self.L.orc_start(self.h)

    def set_threads(self, n):
        """n > 1: the tile loops of main3d_step run as n OpenMP threads (the reference's shared-memory mode)."""
        self.L.orc_set_threads(self.h, int(n))

    def main3d_step(self, n=1):
        for _ in range(n):
            self.L.orc_main3d_step(self.h)

    def diag(self):
        """Run orc_diag; returns [avgke, avgpe, avgkp, volume, maxspeed, Cu, Cv, Cw, Ci, Cj, Ck, C]."""
        self.L.orc_diag(C.c_void_p(self.h))
        out = (C.c_double * 16)()
        self.L.orc_get_diag(C.c_void_p(self.h), out)
        return list(out)[:12]
