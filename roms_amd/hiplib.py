"""ctypes binding of libroms_hip.so (C ABI: include/roms_hip.h)."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libroms_hip.so")
MAXT, MAXW = 4, 512

A4, C2, C4, HSIMT, MPDATA, SPLINES, SPLIT_U3, U3 = range(1, 9)
SCHEMES = dict(A4=A4, C2=C2, C4=C4, HSIMT=HSIMT, MPDATA=MPDATA, SPLINES=SPLINES, SU3=SPLIT_U3, U3=U3)
OPTIONS = {name: 1 << k for k, name in enumerate(
    ["UV_ADV", "UV_COR", "UV_VIS2", "TS_DIF2", "MIX_GEO_TS", "CURVGRID", "NONLIN_EOS", "UV_QDRAG",
     "LMD_MIXING", "BULK_FLUXES", "SOLAR_SOURCE", "ANA_VMIX", "SALINITY", "SPHERICAL", "UV_LOGDRAG", "MASKING"])}
OPTIONS.update(RADIATION_2D=1 << 16, PLAIN_VDIFF=1 << 17, PLAIN_VVISC=1 << 18, PRSGRD31=1 << 19, WJ_GRADP=1 << 27, APP_UPWELLING=1 << 20, APP_BENCHMARK=1 << 21, APP_KELVIN=1 << 22, APP_SEAMOUNT=1 << 23, APP_GRAV_ADJ=1 << 24, GLS_MIXING=1 << 25, PRSGRD40=1 << 26, MY25_MIXING=1 << 28, MIX_ISO_TS=1 << 29, APP_OVERFLOW=1 << 30,
               # the upper word (ABI version 4)
               UV_VIS4=1 << 32, TS_DIF4=1 << 33, WET_DRY=1 << 34, DIAGNOSTICS_UV=1 << 35, MIX_GEO_UV=1 << 36, NUDGE_M3CLM=1 << 37, NUDGE_M2CLM=1 << 42, PRSGRD42=1 << 43, PRSGRD44=1 << 44, LMD_DDMIX=1 << 45, LMD_BKPP=1 << 46,
               NUDGE_TCLM1=1 << 38, NUDGE_TCLM2=1 << 39, NUDGE_TCLM3=1 << 40, NUDGE_TCLM4=1 << 41)
ABI_VERSION = 5
# the compile-time forms of GLS_MIXING (roms_hip_config.gls_flags)
GLS_FLAGS = dict(CANUTO_A=1, CANUTO_B=2, KANTHA_CLAYSON=4, N2S2_HORAVG=8, RI_SPLINES=16, K_C2ADVECTION=32, K_C4ADVECTION=64,
                 CHARNOK=128, CRAIG_BANNER=256)
# lateral boundary conditions (include/roms_hip.h): lbc[edge][variable], ROMS_LBC_* kinds
NLBC = 5 + MAXT
LBC_KINDS = dict(Clo=1, Per=2, Gra=3, Cla=4, Rad=5, RadNud=6, Che=7, Cha=8, Fla=9, Shc=10)
BRY_FIELDS = [v + "_" + e for v in ("zeta", "ubar", "vbar", "u", "v", "t") for e in ("west", "east", "south", "north")]


class Config(C.Structure):
    """roms_hip_config"""
    _fields_ = [
        ("abi_version", C.c_int), ("device", C.c_int),
        ("Lm", C.c_int), ("Mm", C.c_int), ("N", C.c_int), ("NT", C.c_int), ("NAT", C.c_int),
        ("Nghost", C.c_int), ("LBi", C.c_int), ("UBi", C.c_int), ("LBj", C.c_int), ("UBj", C.c_int),
        ("NtileI", C.c_int), ("NtileJ", C.c_int), ("tile", C.c_int),
        ("EWperiodic", C.c_int), ("NSperiodic", C.c_int), ("options", C.c_ulonglong),
        ("hadv", C.c_int * MAXT), ("vadv", C.c_int * MAXT),
        ("Istr", C.c_int), ("Iend", C.c_int), ("Jstr", C.c_int), ("Jend", C.c_int),
        ("west_edge", C.c_int), ("east_edge", C.c_int), ("south_edge", C.c_int), ("north_edge", C.c_int),
        ("ntfirst", C.c_int), ("ntstart", C.c_int), ("ndtfast", C.c_int), ("nfast", C.c_int),
        ("ninfo", C.c_int),
        ("dt", C.c_double), ("dtfast", C.c_double), ("weight", (C.c_double * (MAXW + 1)) * 2),
        ("rho0", C.c_double), ("g", C.c_double), ("lambda_", C.c_double), ("gamma2", C.c_double),
        ("Cp", C.c_double), ("R0", C.c_double), ("T0", C.c_double), ("S0", C.c_double),
        ("Tcoef", C.c_double), ("Scoef", C.c_double), ("hc", C.c_double), ("Vtransform", C.c_int),
        ("rdrg", C.c_double), ("rdrg2", C.c_double), ("Zob", C.c_double),
        ("Akt_bak", C.c_double * MAXT), ("Akv_bak", C.c_double), ("dstart", C.c_double),
        ("blk_ZQ", C.c_double), ("blk_ZT", C.c_double), ("blk_ZW", C.c_double), ("lmd_Jwt", C.c_int),
        ("sc_r", C.c_double * 256), ("Cs_r", C.c_double * 256), ("sc_w", C.c_double * 257),
        ("Cs_w", C.c_double * 257),
        ("lbc", (C.c_int * NLBC) * 4),
        ("FSobc_in", C.c_double * 4), ("FSobc_out", C.c_double * 4), ("M2obc_in", C.c_double * 4),
        ("M2obc_out", C.c_double * 4), ("M3obc_in", C.c_double * 4), ("M3obc_out", C.c_double * 4),
        ("Tobc_in", (C.c_double * 4) * MAXT), ("Tobc_out", (C.c_double * 4) * MAXT),
        ("gls_flags", C.c_int),
        ("gls_p", C.c_double), ("gls_m", C.c_double), ("gls_n", C.c_double), ("gls_Kmin", C.c_double),
        ("gls_Pmin", C.c_double), ("gls_cmu0", C.c_double), ("gls_c1", C.c_double), ("gls_c2", C.c_double),
        ("gls_c3m", C.c_double), ("gls_c3p", C.c_double), ("gls_sigk", C.c_double), ("gls_sigp", C.c_double),
        ("Akk_bak", C.c_double), ("Akp_bak", C.c_double), ("Zos", C.c_double), ("charnok_alpha", C.c_double),
        ("crgban_cw", C.c_double), ("lbc_tke", C.c_int * 4),
        ("Dcrit", C.c_double), ("obcfac", C.c_double), ("volcons", C.c_int),
    ]


class Stepping(C.Structure):
    """roms_hip_stepping"""
    _fields_ = [("iic", C.c_int), ("iif", C.c_int), ("nstp", C.c_int), ("nnew", C.c_int),
                ("nrhs", C.c_int), ("kstp", C.c_int), ("knew", C.c_int), ("krhs", C.c_int),
                ("indx1", C.c_int), ("predictor", C.c_int), ("time", C.c_double)]


KERNELS = ["set_depth", "set_massflux", "rho_eos", "set_vbc", "ana_vmix", "set_data", "omega", "set_zeta",
           "ini_zeta", "ini_fields", "pre_step3d", "prsgrd", "t3dmix2", "uv3dmix2", "rhs3d_tile", "rhs3d",
           "step2d", "step2d_pair", "step2d_loop", "step3d_uv", "step3d_t", "lmd_vmix", "bulk_flux", "gls_prestep", "gls_corstep"]

EXPORTS = ["roms_hip_create", "roms_hip_destroy", "roms_hip_last_error", "roms_hip_abi_version",
           "roms_hip_field_size", "roms_hip_upload", "roms_hip_download", "roms_hip_sync",
           "roms_hip_set_stepping", "roms_hip_get_stepping", "roms_hip_wvelocity", "roms_hip_diag", "roms_hip_last_diag", "roms_hip_get_bounds", "roms_hip_output_point", "roms_hip_avg_config", "roms_hip_set_avg", "roms_hip_avg_time",
           "roms_hip_start", "roms_hip_main3d", "roms_hip_profile", "roms_hip_region_seconds",
           "roms_hip_kprof", "roms_hip_kprof_get", "roms_hip_kprof_stride", "roms_hip_kprof_window", "roms_hip_exchange_soak", "roms_hip_dia_config", "roms_hip_wetdry_ini", "roms_hip_set_diags", "roms_hip_dia_time", "roms_hip_kprof_batch", "roms_hip_set_exchange", "roms_hip_rccl_unique_id",
           "roms_hip_comm_rccl", "roms_hip_peer_export", "roms_hip_comm_peer", "roms_hip_exchange_probe", "roms_hip_comm_reset", "roms_hip_exchange_count", "roms_hip_rccl_ranks", "roms_hip_copy_probe", "roms_hip_step_timing", "roms_hip_step_times", "roms_hip_rim_probe", "roms_hip_rim_disable"] + \
          ["roms_hip_" + k for k in KERNELS]


# roms_hip_exchange_fn (include/roms_hip.h)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p),
                          C.POINTER(C.c_long), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int),
                          C.POINTER(C.c_void_p), C.POINTER(C.c_long), C.POINTER(C.c_int))


class RomsHipError(RuntimeError):
    pass


def load(path=None):
    """Load the native library.  No fallback: raises if it is missing."""
    path = path or os.environ.get("ROMS_HIP_LIB") or DEFAULT_LIB       # (ROMS_HIP_LIB: another build of the same library, for A/B timing)
    if not os.path.exists(path):
        raise RomsHipError(f"{path} not found: build it with roms_amd/build.py "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(path)
    L.roms_hip_last_error.restype = C.c_char_p
    L.roms_hip_field_size.restype = C.c_long
    L.roms_hip_field_size.argtypes = [C.c_void_p, C.c_char_p]
    L.roms_hip_upload.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_long]
    L.roms_hip_download.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_long]
    L.roms_hip_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.roms_hip_destroy.argtypes = [C.c_void_p]
    L.roms_hip_sync.argtypes = [C.c_void_p]
    L.roms_hip_set_stepping.argtypes = [C.c_void_p, C.POINTER(Stepping)]
    L.roms_hip_get_stepping.argtypes = [C.c_void_p, C.POINTER(Stepping)]
    L.roms_hip_wvelocity.argtypes = [C.c_void_p, C.c_int]
    L.roms_hip_diag.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.roms_hip_last_diag.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.roms_hip_get_bounds.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.roms_hip_output_point.argtypes = [C.c_void_p]
    L.roms_hip_avg_config.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint]
    L.roms_hip_set_avg.argtypes = [C.c_void_p]
    L.roms_hip_avg_time.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.roms_hip_start.argtypes = [C.c_void_p]
    L.roms_hip_main3d.argtypes = [C.c_void_p, C.c_int]
    L.roms_hip_profile.argtypes = [C.c_void_p, C.c_int]
    L.roms_hip_region_seconds.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.roms_hip_set_exchange.argtypes = [C.c_void_p, EXCHANGE_FN, C.c_void_p]
    L.roms_hip_rccl_unique_id.argtypes = [C.c_void_p]
    L.roms_hip_comm_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.roms_hip_peer_export.argtypes = [C.c_void_p, C.c_void_p]
    L.roms_hip_comm_peer.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.roms_hip_exchange_probe.argtypes = [C.c_void_p, C.c_int]
    L.roms_hip_comm_reset.argtypes = [C.c_void_p]
    L.roms_hip_exchange_count.argtypes = [C.c_void_p]
    L.roms_hip_exchange_count.restype = C.c_long
    L.roms_hip_rccl_ranks.argtypes = [C.c_void_p]
    L.roms_hip_rccl_ranks.restype = C.c_long
    L.roms_hip_copy_probe.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_long)]
    L.roms_hip_step_timing.argtypes = [C.c_void_p, C.c_int]
    L.roms_hip_step_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    L.roms_hip_kprof.argtypes = [C.c_int, C.c_char_p]
    L.roms_hip_kprof_get.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long)]
    for k in KERNELS:
        getattr(L, "roms_hip_" + k).argtypes = [C.c_void_p]
    return L


def kprof(mode, kernel=None, lib_path=None, stride=1, batch=1, window=None):
    """Per-kernel HIP-event timing: 0 off, 1 all launches (synchronous), 2 only `kernel` (asynchronous,
    every `stride`-th launch, or one event pair per run of `batch` consecutive launches), 3 only `kernel`, the event
    pair filled by the launch itself (the dispatch's begin / end timestamps: rocprofv3's kernel duration)."""
    L = load(lib_path)
    L.roms_hip_kprof(mode, kernel.encode() if kernel else None)
    L.roms_hip_kprof_stride(int(stride))
    L.roms_hip_kprof_batch(int(batch))
    if window:                       # (on, period): mode 3, `on` consecutive launches out of every `period`
        L.roms_hip_kprof_window(int(window[0]), int(window[1]))


def kprof_table(lib_path=None):
    """{kernel name: (seconds, launches)} accumulated since the last kprof() call."""
    L = load(lib_path)
    out, i = {}, 0
    name = C.create_string_buffer(64)
    sec, calls = C.c_double(), C.c_long()
    while L.roms_hip_kprof_get(i, name, 64, C.byref(sec), C.byref(calls)) == 0:
        out[name.value.decode()] = (sec.value, calls.value)
        i += 1
    return out


class Context:
    """One device context = one tile on one GPU (mirrors the reference's kernel(ng,tile) wrappers)."""

    def __init__(self, cfg: Config, lib_path=None):
        self.L = load(lib_path)
        self.cfg = cfg
        self.h = C.c_void_p()
        self._ck(self.L.roms_hip_create(C.byref(cfg), C.byref(self.h)))
        self.ni = cfg.UBi - cfg.LBi + 1
        self.nj = cfg.UBj - cfg.LBj + 1

    @classmethod
    def from_handle(cls, handle, lib_path=None):
        """View of a context created elsewhere (the Fortran host driver); close() is the owner's job."""
        self = cls.__new__(cls)
        self.L = load(lib_path)
        self.cfg = None
        self.h = C.c_void_p(handle)
        return self

    def _ck(self, r):
        if r != 0:
            msg = self.L.roms_hip_last_error()
            raise RomsHipError(f"exit_flag={r}: {msg.decode() if msg else ''}")

    def close(self):
        if self.h:
            self.L.roms_hip_destroy(self.h)
            self.h = C.c_void_p()

    def size(self, name):
        n = self.L.roms_hip_field_size(self.h, name.encode())
        if n < 0:
            raise KeyError(name)
        return n

    def upload(self, name, a):
        a = np.ascontiguousarray(a, dtype=np.float64).ravel()
        self._ck(self.L.roms_hip_upload(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))

    def download(self, name):
        a = np.empty(self.size(name), dtype=np.float64)
        self._ck(self.L.roms_hip_download(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        return a

    def set_stepping(self, **kw):
        s = self.get_stepping()
        for k, v in kw.items():
            setattr(s, k, v)
        self._ck(self.L.roms_hip_set_stepping(self.h, C.byref(s)))

    def get_stepping(self):
        s = Stepping()
        self._ck(self.L.roms_hip_get_stepping(self.h, C.byref(s)))
        return s

    def call(self, kernel, *args):
        if kernel == "wvelocity":
            self._ck(self.L.roms_hip_wvelocity(self.h, int(args[0])))
        else:
            self._ck(getattr(self.L, "roms_hip_" + kernel)(self.h))

    def sync(self):
        self._ck(self.L.roms_hip_sync(self.h))

    def start(self):
        self._ck(self.L.roms_hip_start(self.h))

    def main3d(self, nsteps=1):
        self._ck(self.L.roms_hip_main3d(self.h, int(nsteps)))

    def diag(self, raw=False):
        out = (C.c_double * 16)()
        self._ck(self.L.roms_hip_diag(self.h, out))
        return list(out) if raw else list(out)[:12]

    AVG_FIELDS = ["avg_zeta", "avg_ubar", "avg_vbar", "avg_u", "avg_v", "avg_omega", "avg_w", "avg_rho", "avg_t", "avg_ZZ",
                  "avg_U2", "avg_V2", "avg_UU", "avg_VV", "avg_UV", "avg_Huon", "avg_Hvom", "avg_TT", "avg_UT", "avg_VT",
                  "avg_HuonT", "avg_HvomT"]

    def avg_config(self, nAVG, ntsAVG=1, nrrec=0, ntstart=1, mask=(1 << 22) - 1):
        """time-averaged fields (set_avg.F): window of nAVG steps from step ntsAVG on; mask bit f = AVG_FIELDS[f]"""
        self._ck(self.L.roms_hip_avg_config(self.h, nAVG, ntsAVG, nrrec, ntstart, mask))

    def dia_config(self, nDIA, ntsDIA=1, nrrec=0, ntstart=1, uv=False):
        """DIAGNOSTICS_TS: allocate the per-term tracer tendencies and set the window of set_diags.F.  The per-term momentum
        tendencies ("DiaU2wrk" ... "DiaV3d") come with them when the context was created with the option bit DIAGNOSTICS_UV
        (ABI version 4); uv=True only asserts that it was."""
        if uv and self.cfg is not None and not (int(self.cfg.options) & OPTIONS["DIAGNOSTICS_UV"]):
            raise RomsHipError("dia_config(uv=True): create the context with OPTIONS['DIAGNOSTICS_UV'] in cfg.options")
        self._ck(self.L.roms_hip_dia_config(self.h, int(nDIA), int(ntsDIA), int(nrrec), int(ntstart)))


    def wetdry_ini(self):
        self._ck(self.L.roms_hip_wetdry_ini(self.h))

    def avg_time(self):
        t = C.c_double()
        self._ck(self.L.roms_hip_avg_time(self.h, C.byref(t)))
        return t.value

    def output_point(self):
        """derived fields of the step about to be taken as at main3d.F:591 (where the reference writes output)"""
        self._ck(self.L.roms_hip_output_point(self.h))

    def bounds(self):
        """The 54 BOUNDS/DOMAIN entries of this tile as the library derived them (order: include/roms_hip.h)."""
        out = (C.c_int * 54)()
        self._ck(self.L.roms_hip_get_bounds(self.h, out))
        return list(out)

    def last_diag(self):
        """(report, step) of the last diag that ran inside main3d (what the reference prints); step -1: none."""
        out = (C.c_double * 16)()
        self._ck(self.L.roms_hip_last_diag(self.h, out))
        return list(out)[:14], int(out[14])

    def copy_probe(self, reps=20):
        """Measured streaming-copy bandwidth in GB/s (read + write bytes / time) of k_copy_probe."""
        nb = C.c_long()
        self._ck(self.L.roms_hip_copy_probe(self.h, 3, C.byref(nb)))      # warm-up
        self.sync()
        kprof(1)
        self._ck(self.L.roms_hip_copy_probe(self.h, reps, C.byref(nb)))
        self.sync()
        sec, n = kprof_table().get("k_copy_probe", (0.0, 0))
        kprof(0)
        return nb.value * n / sec / 1e9 if sec > 0 else None

    def profile(self, enable=True):
        self._ck(self.L.roms_hip_profile(self.h, int(enable)))

    def region(self, rid):
        s, n = C.c_double(), C.c_long()
        self._ck(self.L.roms_hip_region_seconds(self.h, rid, C.byref(s), C.byref(n)))
        return s.value, n.value
