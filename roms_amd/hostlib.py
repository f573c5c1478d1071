"""ctypes binding of libroms_host.so, the Fortran host driver (roms_amd/host/*.f90).

The driver's input is a standard roms.in (the reference's own run-time interface, subset of keywords
read by ROMS/Utility/read_phypar.F); `write_roms_in` produces one from a parameter dict so that tests
and bench.py can run any grid size.
"""
import ctypes as C
import os
import tempfile

import numpy as np

from . import build as _build

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libroms_host.so")

# arrays the host set-up owns (roms_host_get names = the reference's mod_grid / mod_ocean / mod_mixing names)
HOST_FIELDS = ["h", "f", "fomn", "pm", "pn", "om_r", "on_r", "om_u", "on_u", "om_v", "on_v", "om_p", "on_p", "omn",
               "pmon_r", "pnom_r", "pmon_p", "pnom_p", "pmon_u", "pnom_u", "pmon_v", "pnom_v", "angler", "xr", "yr", "xp", "yp",
               "rdrag", "rdrag2", "visc2_r", "visc2_p", "diff2", "Hz", "z_r", "z_w", "zeta", "ubar", "vbar", "u", "v",
               "t", "Zt_avg1", "Akv", "Akt", "dmde", "dndx", "lonr", "latr", "sc_r", "Cs_r", "sc_w", "Cs_w"]

_libs = {}


def load(path=None):
    path = os.path.abspath(path or LIB)
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is not built (python -m roms_amd.build)")
        lib = C.CDLL(path)
        lib.roms_host_setup.argtypes = [C.c_char_p]
        lib.roms_host_device_init.argtypes = [C.c_int, C.c_int]
        lib.roms_host_tile.argtypes = [C.POINTER(C.c_int)]
        lib.roms_host_tile.restype = None
        lib.roms_host_run.argtypes = [C.c_int, C.c_int]
        lib.roms_host_ctx.restype = C.c_void_p
        lib.roms_host_dims.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_double)]
        lib.roms_host_dims.restype = None
        lib.roms_host_get.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_long]
        lib.roms_host_get.restype = C.c_long
        lib.roms_host_set_header.argtypes = [C.c_char_p]
        lib.roms_host_set_header.restype = None
        lib.roms_host_last_error.argtypes = [C.c_char_p, C.c_int]
        lib.roms_host_last_error.restype = None
        lib.roms_host_write.argtypes = [C.c_int]
        lib.roms_host_advance.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.roms_host_get_state.argtypes = [C.c_char_p, C.c_int]
        lib.roms_host_close_output.restype = None
        lib.roms_host_set_gather.argtypes = [C.c_void_p]
        lib.roms_host_set_gather.restype = None
        lib.roms_host_output_config.argtypes = [C.POINTER(C.c_int), C.c_char_p]
        lib.roms_host_output_config.restype = None
        _libs[path] = lib
    return _libs[path]


def write_roms_in(path, p):
    """p: dict with the reference's keyword names (see tests/cases.py)."""
    tr = lambda xs: " ".join(str(x) for x in xs)
    d = lambda x: repr(float(x)).replace("e", "d") if "e" in repr(float(x)) else repr(float(x)) + "d0"
    per = lambda f: "Per" if f else "Clo"
    lines = [
        # (kelvin: ROMS/Include/kelvin.h when the case asks for the plain vertical solvers, else oracle/ref/kelvin_splines.h)
        f"    MyAppCPP = {('KELVIN' if 'PLAIN_VDIFF' in p.get('options', ()) else 'KELVIN_SPLINES') if p['app'] == 'kelvin' else p['app'].upper()}",
        f"         NAT =  {p.get('NAT', 2)}",
        f"          Lm == {p['Lm']}", f"          Mm == {p['Mm']}", f"           N == {p['N']}",
        f"      NtileI == {p.get('NtileI', 1)}", f"      NtileJ == {p.get('NtileJ', 1)}",
        f"  Hadvection == {p['hadv'][0]} \\", f"                {p['hadv'][1]}",
        f"  Vadvection == {p['vadv'][0]} \\", f"                {p['vadv'][1]}",
    ]
    lbc = f"{per(p['EWperiodic'])} {per(p['NSperiodic'])} {per(p['EWperiodic'])} {per(p['NSperiodic'])}"   # W S E N
    # open boundaries: `lbc` of the case (variable -> the four keywords of its LBC line), else closed / periodic
    kinds = lambda v: " ".join(p.get("lbc", {}).get(v, ())) or lbc
    lines += [f" LBC({n}) == {kinds(v)}" for n, v in (("isFsur", "zeta"), ("isUbar", "ubar"), ("isVbar", "vbar"), ("isUvel", "u"),
                                                      ("isVvel", "v"), ("isMtke", "-"))]
    lines += [f" LBC(isTvar) == {kinds('temp')} \\", f"                {kinds('salt')}"]
    for key, name in (("Znudg", "ZNUDG"), ("M2nudg", "M2NUDG"), ("M3nudg", "M3NUDG"), ("obcfac", "OBCFAC")):
        if key in p:
            lines.append(f"{name:>12} == {d(p[key])}")
    if "Tnudg" in p:
        lines.append(f"       TNUDG == {tr(d(x) for x in p['Tnudg'])}")
    for bit, edge in enumerate(("west", "south", "east", "north")):       # VolCons(...) of roms.in from the case's bit mask
        if p.get("volcons", 0) & (1 << bit):
            lines.append(f" VolCons({edge}) == T")
    lines += [
        f"      NTIMES == {p.get('ntimes', 10)}", f"          DT == {d(p['dt'])}",
        f"     NDTFAST == {p['ndtfast']}", f"       NINFO == {p.get('ninfo', 1)}",
        f"        TNU2 == {tr(d(x) for x in p['tnu2'])}", f"       VISC2 == {d(p['visc2'])}",
        f"        TNU4 == {tr(d(x) for x in p.get('tnu4', (0.0, 0.0)))}", f"       VISC4 == {d(p.get('visc4', 0.0))}",
        f"       DCRIT == {d(p.get('Dcrit', 0.10))}",
        f"     AKT_BAK == {tr(d(x) for x in p['Akt_bak'])}", f"     AKV_BAK == {d(p['Akv_bak'])}",
        f"        RDRG == {d(p['rdrg'])}", f"       RDRG2 == {d(p['rdrg2'])}",
        f"         Zob == {d(p['Zob'])}", f"         Zos == {d(p['Zos'])}",
        f"      BLK_ZQ == {d(p.get('blk_ZQ', 10.0))}", f"      BLK_ZT == {d(p.get('blk_ZT', 10.0))}",
        f"      BLK_ZW == {d(p.get('blk_ZW', 10.0))}", f"       WTYPE == {p.get('lmd_Jwt', 1)}",
        f"  Vtransform == {p['Vtransform']}", f" Vstretching == {p['Vstretching']}",
        f"     THETA_S == {d(p['theta_s'])}", f"     THETA_B == {d(p['theta_b'])}",
        f"      TCLINE == {d(p['Tcline'])}", f"        RHO0 == {d(p['rho0'])}",
        f"      DSTART == {d(p.get('dstart', 0.0))}", "    TIME_REF == 0.0d0",
        f"          R0 == {d(p['R0'])}", f"          T0 == {d(p['T0'])}", f"          S0 == {d(p['S0'])}",
        f"       TCOEF == {d(p['Tcoef'])}", f"       SCOEF == {d(p['Scoef'])}",
        f"      GAMMA2 == {d(p['gamma2'])}",
    ]
    if "gls_flags" in p:       # the generic length-scale closure: its roms.in block (the cpp form comes from the header / ROMS_CPP_FLAGS)
        for key, name in (("Akk_bak", "AKK_BAK"), ("Akp_bak", "AKP_BAK"), ("gls_p", "GLS_P"), ("gls_m", "GLS_M"), ("gls_n", "GLS_N"),
                          ("gls_Kmin", "GLS_Kmin"), ("gls_Pmin", "GLS_Pmin"), ("gls_cmu0", "GLS_CMU0"), ("gls_c1", "GLS_C1"),
                          ("gls_c2", "GLS_C2"), ("gls_c3m", "GLS_C3M"), ("gls_c3p", "GLS_C3P"), ("gls_sigk", "GLS_SIGK"),
                          ("gls_sigp", "GLS_SIGP"), ("charnok_alpha", "CHARNOK_ALPHA"), ("crgban_cw", "CRGBAN_CW")):
            lines.append(f"{name:>14} == {d(p[key])}")
    # output (optional keys): NRREC, NRST, NHIS, LcycleRST, file names, Hout switches by their roms.in ids
    for key in ("NRREC", "NRST", "NHIS", "NAVG", "NTSAVG", "NDIA", "NTSDIA"):
        if key in p:
            lines.append(f"{key:>12} == {int(p[key])}")
    if "LcycleRST" in p:
        lines.append(f"   LcycleRST == {'T' if p['LcycleRST'] else 'F'}")
    for key in ("ININAME", "RSTNAME", "HISNAME", "AVGNAME", "DIANAME"):
        if key in p:
            lines.append(f"{key:>12} == {p[key]}")
    for sw in ("Hout", "Aout", "Dout"):
        for vid, val in p.get(sw, {}).items():
            val = val if isinstance(val, (tuple, list)) else (val,)
            lines.append(f"{sw}({vid}) == " + " ".join("T" if x else "F" for x in val))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


class HostError(RuntimeError):
    """set-up stopped with the reference's exit_flag code (5 = configuration, 2 = input file)"""

    def __init__(self, exit_flag, message):
        super().__init__(f"roms_host_setup failed, exit_flag={exit_flag}: {message}")
        self.exit_flag = exit_flag


class Host:
    """One ROMS run owned by the Fortran host (module state: one instance per process)."""

    def __init__(self, infile=None, params=None, lib_path=None, hip_lib_path=None, header=None):
        """infile: a roms.in; or params: a dict written out as one.  header: application header whose cpp
        options select the physics (ROMS/Include/<app>.h or a custom one); default: the built-in option
        lists of UPWELLING / BENCHMARK / UPWELLING_KPP (or $ROMS_APP_HEADER)."""
        self.lib = load(lib_path)
        self.hip_lib_path = hip_lib_path
        self.lib.roms_host_set_header((header or "").encode())
        tmp = None
        if infile is None:
            fd, tmp = tempfile.mkstemp(suffix=".in", prefix="roms_")
            os.close(fd)
            write_roms_in(tmp, params)
            infile = tmp
        try:
            r = self.lib.roms_host_setup(infile.encode())
        finally:
            if tmp:
                os.unlink(tmp)
        if r != 0:
            raise HostError(r, self.last_error())
        di = (C.c_int * 24)()
        dr = (C.c_double * 8)()
        self.lib.roms_host_dims(di, dr)
        names = ["Lm", "Mm", "N", "NT", "Nghost", "LBi", "UBi", "LBj", "UBj", "nfast", "ndtfast", "ntimes", "options",
                 "EWper", "NSper"]
        self.dims = dict(zip(names, list(di)[:15]))
        self.dims["hadv"], self.dims["vadv"], self.dims["ninfo"] = list(di)[15:19], list(di)[19:23], di[23]
        self.reals = dict(zip(["dt", "dtfast", "hc", "hmin", "hmax", "xl", "el", "dstart"], list(dr)))

    def last_error(self):
        buf = C.create_string_buffer(300)
        self.lib.roms_host_last_error(buf, 300)
        return buf.value.decode(errors="replace")

    def get(self, name):
        n = self.lib.roms_host_get(name.encode(), None, 0)
        if n <= 0:
            raise KeyError(name)
        a = np.empty(n)
        self.lib.roms_host_get(name.encode(), a.ctypes.data_as(C.POINTER(C.c_double)), n)
        return a

    def _fail(self, what, r):
        from . import hiplib
        msg = hiplib.load(self.hip_lib_path).roms_hip_last_error()
        raise hiplib.RomsHipError(f"{what}: exit_flag={r}: {msg.decode() if msg else ''}")

    def device_init(self, device=0, tile=0, start=True):
        """Create the device context of `tile` and upload its window of the state; with start=False
        the caller installs a halo transport on context() and then calls start()."""
        r = self.lib.roms_host_device_init(device, tile)
        if r != 0:
            self._fail("roms_host_device_init", r)
        b = (C.c_int * 12)()
        self.lib.roms_host_tile(b)
        self.tile = dict(zip(["tile", "Istr", "Iend", "Jstr", "Jend", "LBi", "UBi", "LBj", "UBj", "NtileI", "NtileJ"],
                             list(b)))
        if start:
            self.start()
        return self.context()

    def start(self):
        r = self.lib.roms_host_start()
        if r != 0:
            self._fail("roms_host_start", r)

    def context(self):
        """The device context as a roms_amd.hiplib.Context view (not owning)."""
        from . import hiplib
        return hiplib.Context.from_handle(self.lib.roms_host_ctx(), self.hip_lib_path)

    def run(self, nsteps, kernels=False):
        r = self.lib.roms_host_run(nsteps, 1 if kernels else 0)
        if r != 0:
            self._fail("roms_host_run", r)

    # ---- output and restart (roms_amd/host/roms_output.f90: wrt_his, wrt_rst, get_state of the reference)
    def output_config(self):
        """nrrec, nRST, nHIS, LcycleRST and the ININAME / RSTNAME / HISNAME of roms.in"""
        ints = (C.c_int * 6)()
        names = C.create_string_buffer(4 * 256)
        self.lib.roms_host_output_config(ints, names)
        raw = names.raw
        ini, rst, his, avg = [raw[k * 256:(k + 1) * 256].split(b"\0")[0].decode() for k in range(4)]
        return dict(nrrec=ints[0], nRST=ints[1], nHIS=ints[2], LcycleRST=bool(ints[3]), nAVG=ints[4], ntsAVG=ints[5],
                    ininame=ini, rstname=rst, hisname=his, avgname=avg)

    def _out(self, what, r):
        if r != 0:
            msg = self.last_error()
            if r in (2, 3, 5) and msg:
                raise HostError(r, msg)
            self._fail(what, r)

    def write_his(self):
        """one history record of the state between two steps (wrt_his at main3d.F:591 of the coming step)"""
        self._out("roms_host_write", self.lib.roms_host_write(1))

    def write_rst(self):
        """one restart record (wrt_rst, PERFECT_RESTART form)"""
        self._out("roms_host_write", self.lib.roms_host_write(2))

    def advance(self, nsteps, kernels=False, final=False):
        """nsteps steps with the history / restart records NHIS and NRST ask for (output.F)"""
        self._out("roms_host_advance", self.lib.roms_host_advance(nsteps, 1 if kernels else 0, 1 if final else 0))

    def get_state(self, path="", rec=0):
        """restart from record rec (1-based; <= 0: the latest) of a restart file (default: ININAME)"""
        self._out("roms_host_get_state", self.lib.roms_host_get_state(path.encode(), rec))

    def close_output(self):
        self.lib.roms_host_close_output()

    def finalize(self):
        self.lib.roms_host_finalize()
