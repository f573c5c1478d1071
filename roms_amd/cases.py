"""The BASELINE applications as parameter dicts (the values of the reference's ROMS/External/roms_upwelling.in
and roms_benchmark{1,2,3}.in; constants of ROMS/Modules/mod_scalars.F), the form roms_amd.hostlib.write_roms_in
turns into a roms.in for any grid size.  Used by bench.py, roms_amd.tiling and the tests."""
import numpy as np

SCHEME = dict(A4=1, C2=2, C4=3, HSIMT=4, MPDATA=5, SPLINES=6, SU3=7, U3=8)
# lateral boundary conditions: the LBC(isFsur) ... LBC(isTvar) lines of roms.in, one keyword per edge in the file's
# order west, south, east, north (`lbc` of a case dict: variable -> four keywords; absent = closed or periodic)
LBC_VARS = ("zeta", "ubar", "vbar", "u", "v", "temp", "salt")
LBC_KINDS = dict(Clo=1, Per=2, Gra=3, Cla=4, Rad=5, RadNud=6, Che=7, Cha=8, Fla=9, Shc=10)


def lbc_codes(cs):
    """[variable][edge] kind codes of include/roms_hip.h (ROMS_LBC_*), 0 where the case says nothing"""
    out = []
    for v in LBC_VARS:
        kw = cs.get("lbc", {}).get(v)
        out.append([LBC_KINDS[k] for k in kw] if kw else [0, 0, 0, 0])
    return out


def obc_scales(cs):
    """FSobc_in/out, M2obc_in/out, M3obc_in/out [edge] and Tobc_in/out [tracer][edge] in 1/s from the roms.in values
    ZNUDG, M2NUDG, M3NUDG, TNUDG (days) and OBCFAC, as Utility/inp_par.F:696-752 derives them (set on edges whose
    condition nudges, zero elsewhere)"""
    inv = lambda d: 1.0 / (d * 86400.0) if d > 0.0 else 0.0
    fac = cs.get("obcfac", 0.0)
    Z, M2, M3 = inv(cs.get("Znudg", 0.0)), inv(cs.get("M2nudg", 0.0)), inv(cs.get("M3nudg", 0.0))
    T = [inv(x) for x in cs.get("Tnudg", (0.0, 0.0))]
    code = lbc_codes(cs)
    nud = lambda v, e: code[v][e] == LBC_KINDS["RadNud"]
    o = dict(FSobc_in=[0.0] * 4, FSobc_out=[0.0] * 4, M2obc_in=[0.0] * 4, M2obc_out=[0.0] * 4, M3obc_in=[0.0] * 4,
             M3obc_out=[0.0] * 4, Tobc_in=[[0.0] * 4, [0.0] * 4], Tobc_out=[[0.0] * 4, [0.0] * 4])
    for e in range(4):
        if nud(0, e):
            o["FSobc_out"][e], o["FSobc_in"][e] = Z, fac * Z
        if nud(1, e) or nud(2, e):
            o["M2obc_out"][e], o["M2obc_in"][e] = M2, fac * M2
        if nud(3, e) or nud(4, e):
            o["M3obc_out"][e], o["M3obc_in"][e] = M3, fac * M3
        for it in range(2):
            if nud(5 + it, e):
                o["Tobc_out"][it][e], o["Tobc_in"][it][e] = T[it], fac * T[it]
    return o


def kelvin(Lm=50, Mm=30, N=10, NtileI=1, NtileJ=1, ntimes=96, lbc=None, plain=False):
    """roms_kelvin.in (ROMS/Include/kelvin.h): a Kelvin wave forced through the western boundary of a flat channel --
    Chapman / Flather conditions west, radiation east (RADIATION_2D), closed walls south and north.  The second tracer
    is carried passively (the library has two tracers; the reference application has NAT = 1)."""
    open_ = ("Rad", "Clo", "Rad", "Clo")
    return dict(
        app="kelvin", Lm=Lm, Mm=Mm, N=N, NtileI=NtileI, NtileJ=NtileJ, ndtfast=60, ntimes=ntimes,
        Vtransform=2, Vstretching=4, EWperiodic=0, NSperiodic=0, hadv=("U3", "U3"), vadv=("C4", "C4"), lmd_Jwt=1,
        dt=900.0, theta_s=0.0, theta_b=0.0, Tcline=1.0e16, rho0=1025.0, R0=1027.0, T0=10.0, S0=35.0,
        Tcoef=1.7e-4, Scoef=7.6e-4, visc2=0.0, tnu2=(20.0, 0.0), Akt_bak=(1.0e-6, 1.0e-6), Akv_bak=1.0e-5,
        rdrg=3.0e-4, rdrg2=3.0e-3, Zob=0.02, Zos=0.02, gamma2=1.0, dstart=0.0,
        blk_ZQ=10.0, blk_ZT=10.0, blk_ZW=10.0,
        # plain: kelvin.h as shipped, without SPLINES_VDIFF / SPLINES_VVISC (the plain tridiagonal vertical solvers)
        options=("UV_ADV", "UV_COR", "UV_QDRAG", "UV_VIS2", "TS_DIF2", "RADIATION_2D", "APP_KELVIN") +
                (("PLAIN_VDIFF", "PLAIN_VVISC") if plain else ()),
        lbc=lbc if lbc is not None else dict(zeta=("Cha", "Clo", "Rad", "Clo"), ubar=("Fla", "Clo", "Rad", "Clo"),
                                             vbar=("Fla", "Clo", "Rad", "Clo"), u=open_, v=open_, temp=open_, salt=open_),
    )


def seamount(Lm=49, Mm=48, N=13, NtileI=1, NtileJ=1, ntimes=100):
    """roms_seamount.in (ROMS/Include/seamount.h): the pressure-gradient test -- a resting stratified ocean over a tall
    Gaussian seamount in a periodic channel; no forcing, no-slip walls (GAMMA2 = -1), quadratic drag, Akima advection,
    harmonic mixing along geopotentials (zero coefficients), background vertical mixing."""
    return dict(
        app="seamount", Lm=Lm, Mm=Mm, N=N, NtileI=NtileI, NtileJ=NtileJ, ndtfast=20, ntimes=ntimes,
        Vtransform=2, Vstretching=4, EWperiodic=1, NSperiodic=0, hadv=("A4", "A4"), vadv=("A4", "A4"), lmd_Jwt=1,
        dt=60.0, theta_s=6.5, theta_b=2.0, Tcline=100.0, rho0=1025.0, R0=1027.0, T0=10.0, S0=32.0,
        Tcoef=1.7e-4, Scoef=7.6e-4, visc2=0.0, tnu2=(0.0, 0.0), Akt_bak=(1.0e-6, 1.0e-6), Akv_bak=1.0e-5,
        rdrg=3.0e-4, rdrg2=3.0e-3, Zob=0.02, Zos=0.02, gamma2=-1.0, dstart=0.0,
        blk_ZQ=10.0, blk_ZT=10.0, blk_ZW=10.0,
        options=("UV_ADV", "UV_COR", "UV_QDRAG", "UV_VIS2", "TS_DIF2", "MIX_GEO_TS", "APP_SEAMOUNT"),
    )


def grav_adj(Lm=128, Mm=4, N=40, NtileI=1, NtileJ=1, ntimes=100):
    """roms_grav_adj.in (ROMS/Include/grav_adj.h): the lock-exchange (gravitational adjustment) test -- warm water left of
    the middle of a closed flat channel, periodic across; MPDATA tracers, no rotation, no drag."""
    return dict(
        app="grav_adj", Lm=Lm, Mm=Mm, N=N, NtileI=NtileI, NtileJ=NtileJ, ndtfast=20, ntimes=ntimes,
        Vtransform=2, Vstretching=4, EWperiodic=0, NSperiodic=1, hadv=("MPDATA", "MPDATA"), vadv=("MPDATA", "MPDATA"), lmd_Jwt=1,
        dt=50.0, theta_s=0.0, theta_b=0.0, Tcline=1.0e16, rho0=1025.0, R0=1025.0, T0=5.0, S0=35.0,
        Tcoef=9.756e-4, Scoef=0.0, visc2=5.0, tnu2=(0.0, 0.0), Akt_bak=(1.0e-6, 1.0e-6), Akv_bak=1.0e-5,
        rdrg=0.0, rdrg2=0.0, Zob=0.02, Zos=0.02, gamma2=1.0, dstart=0.0,
        blk_ZQ=10.0, blk_ZT=10.0, blk_ZW=10.0,
        options=("UV_ADV", "UV_VIS2", "TS_DIF2", "APP_GRAV_ADJ"),
    )


def overflow(Lm=4, Mm=128, N=20, NtileI=1, NtileJ=1, ntimes=100):
    """roms_overflow.in (ROMS/Include/overflow.h): cold dense water released at the top of a slope in a closed channel four
    points wide; tracer mixing along isopycnic surfaces (MIX_ISO_TS), spline vertical advection, the 1994 s-coordinate
    (Vtransform 1, Vstretching 1), no rotation, no forcing, no closure."""
    return dict(
        app="overflow", Lm=Lm, Mm=Mm, N=N, NtileI=NtileI, NtileJ=NtileJ, ndtfast=20, ntimes=ntimes,
        Vtransform=1, Vstretching=1, EWperiodic=0, NSperiodic=0, hadv=("U3", "U3"), vadv=("SPLINES", "SPLINES"), lmd_Jwt=1,
        dt=20.0, theta_s=3.0, theta_b=1.0, Tcline=50.0, rho0=1025.0, R0=1030.0, T0=5.0, S0=0.0,
        Tcoef=1.7e-4, Scoef=0.0, visc2=5.0, tnu2=(5.0, 0.0), Akt_bak=(0.0, 0.0), Akv_bak=0.0,
        rdrg=0.0, rdrg2=0.0, Zob=0.0, Zos=0.0, gamma2=1.0, dstart=0.0,
        blk_ZQ=10.0, blk_ZT=10.0, blk_ZW=10.0,
        options=("UV_ADV", "UV_COR", "UV_QDRAG", "UV_VIS2", "TS_DIF2", "MIX_ISO_TS", "APP_OVERFLOW"),
    )


def upwelling(Lm=41, Mm=80, N=16, NtileI=1, NtileJ=1, hadv=("U3", "HSIMT"), vadv=("C4", "HSIMT"),
              ntimes=100):
    """roms_upwelling.in"""
    return dict(
        app="upwelling", Lm=Lm, Mm=Mm, N=N, NtileI=NtileI, NtileJ=NtileJ, ndtfast=30, ntimes=ntimes,
        Vtransform=2, Vstretching=4, EWperiodic=1, NSperiodic=0, hadv=hadv, vadv=vadv, lmd_Jwt=1,
        dt=300.0, theta_s=3.0, theta_b=0.0, Tcline=25.0, rho0=1025.0, R0=1027.0, T0=14.0, S0=35.0,
        Tcoef=1.7e-4, Scoef=0.0, visc2=5.0, tnu2=(0.0, 0.0), Akt_bak=(1.0e-6, 1.0e-6), Akv_bak=1.0e-5,
        rdrg=3.0e-4, rdrg2=3.0e-3, Zob=0.02, Zos=0.02, gamma2=1.0, dstart=0.0,
        blk_ZQ=10.0, blk_ZT=10.0, blk_ZW=10.0,
        # cpp options of ROMS/Include/upwelling.h
        options=("UV_ADV", "UV_COR", "UV_VIS2", "TS_DIF2", "ANA_VMIX", "SALINITY", "APP_UPWELLING"),
    )


def upwelling_kpp(**kw):
    """UPWELLING with the KPP closure (LMD_MIXING ... SOLAR_SOURCE, ANA_SRFLUX: 150 W/m2) instead of ANA_VMIX:
    the custom application header of BASELINE config 5, oracle/ref/upwelling_kpp.h (MPDATA tracers by default)."""
    kw.setdefault("hadv", ("MPDATA", "MPDATA"))
    kw.setdefault("vadv", ("MPDATA", "MPDATA"))
    cs = upwelling(**kw)
    cs["app"] = "upwelling_kpp"
    cs["options"] = ("UV_ADV", "UV_COR", "UV_VIS2", "TS_DIF2", "LMD_MIXING", "SOLAR_SOURCE", "SALINITY",
                     "APP_UPWELLING")
    return cs


def with_ddmix(cs):
    """... with double-diffusive mixing in the interior scheme of the LMD closure (LMD_DDMIX, lmd_vmix.F:360-428): the reference
    builds oracle/ref/upwelling_kpp_ddmix.h (linear EOS) and benchmark.h -DLMD_DDMIX (nonlinear EOS: alfaobeta of rho_eos.F:454)"""
    cs["app"] = cs["app"] + "_ddmix"
    cs["ddmix"] = 1
    return cs


def with_bkpp(cs):
    """LMD_BKPP: the bottom boundary layer of the K-profile scheme behind lmd_skpp (lmd_bkpp.F) -- the reference builds
    benchmark.h -DLMD_BKPP (nonlinear EOS, bulk fluxes, shortwave) and oracle/ref/upwelling_kpp.h -DLMD_BKPP (linear EOS)"""
    cs["app"] = cs["app"] + "_bkpp"
    cs["bkpp"] = 1
    return cs


def benchmark_bkpp(**kw):
    return with_bkpp(benchmark(**kw))


def upwelling_kpp_bkpp(**kw):
    return with_bkpp(upwelling_kpp(**kw))


def upwelling_kpp_ddmix(**kw):
    return with_ddmix(upwelling_kpp(**kw))


def benchmark_ddmix(**kw):
    return with_ddmix(benchmark(**kw))


def benchmark_wetdry_ddmix(**kw):
    return with_ddmix(benchmark_wetdry(**kw))


def ddmix_state(cs, t, LBi, UBi, LBj, UBj):
    """A temperature / salinity state that reaches every branch of LMD_DDMIX (the analytic initial salinity is uniform: no double
    diffusion at all).  West half of the domain: S = 35 + s (T - 14), warm and salty above cold and fresh with 1 < Rrho < 1.9
    (salt fingering).  East half: the temperature profile turned upside down, T' = Tmin + Tmax - T, and S = 35 + c (T' - 14) --
    cold and fresh above warm and salty, stable by its salinity, 0 < Rrho < 1 (diffusive convection), c chosen per half in eta
    so that Rrho lies on both sides of 0.5.  `t`: the flat tracer array (i, j, k, 3, NT) of the fixtures; returns the new one."""
    import numpy as np
    ni, nj = UBi - LBi + 1, UBj - LBj + 1
    N, Lm, Mm = cs["N"], cs["Lm"], cs["Mm"]
    a = np.array(t, dtype=np.float64).reshape(-1, 3, N, nj, ni).copy()
    T = a[0]
    ii = np.arange(LBi, UBi + 1)
    jj = np.arange(LBj, UBj + 1)
    if cs["EWperiodic"]:
        ii = (ii - 1) % Lm + 1
    if cs["NSperiodic"]:
        jj = (jj - 1) % Mm + 1
    west = (ii <= Lm // 2)[None, None, None, :]
    south = (jj <= Mm // 2)[None, None, :, None]
    nonlin = "NONLIN_EOS" in cs["options"]
    s, cA, cB = (0.1, 0.6, 0.3) if nonlin else (0.15, 0.5, 0.3)
    Tf = (T.min() + T.max()) - T
    Tn = np.where(west, T, Tf)
    S = np.where(west, 35.0 + s * (T - 14.0), 35.0 + np.where(south, cA, cB) * (Tf - 14.0))
    a[0] = Tn
    a[1] = S
    return a.reshape(np.asarray(t).shape)


def upwelling_logdrag(**kw):
    """UPWELLING with the logarithmic bottom drag (UV_LOGDRAG, Zob = 0.02 m) instead of UV_LDRAG: the custom
    application header oracle/ref/upwelling_logdrag.h"""
    cs = upwelling(**kw)
    cs["app"] = "upwelling_logdrag"
    cs["options"] = tuple(cs["options"]) + ("UV_LOGDRAG",)
    return cs


def upwelling_noadv(**kw):
    """UPWELLING with the option set of the reference's WINDBASIN application: no momentum advection, no horizontal mixing
    of momentum or tracers (oracle/ref/upwelling_noadv.h)"""
    cs = upwelling(**kw)
    cs["app"] = "upwelling_noadv"
    cs["options"] = tuple(o for o in cs["options"] if o not in ("UV_ADV", "UV_VIS2", "TS_DIF2"))
    return cs


def upwelling_prs40(**kw):
    """UPWELLING with the finite-volume pressure Jacobian of Lin (1997) (prsgrd40.h: PJ_GRADP): the custom application
    header oracle/ref/upwelling_prs40.h"""
    cs = upwelling(**kw)
    cs["app"] = "upwelling_prs40"
    cs["options"] = tuple(cs["options"]) + ("PRSGRD40",)
    return cs


def upwelling_prs4x(scheme=44, **kw):
    """UPWELLING with the finite-volume pressure Jacobians of Shchepetkin & McWilliams (2003): prsgrd42.h (PJ_GRADPQ2, scheme 42)
    or prsgrd44.h (PJ_GRADPQ4, scheme 44); the custom application headers oracle/ref/upwelling_prs42.h, upwelling_prs44.h"""
    cs = upwelling(**kw)
    cs["app"] = "upwelling_prs%d" % scheme
    cs["prsgrd"] = int(scheme)
    return cs


def upwelling_bih(visc4=4.0e8, tnu4=(2.0e7, 1.0e7), **kw):
    """UPWELLING with biharmonic mixing of momentum and tracers along s-surfaces (UV_VIS4, TS_DIF4: the custom application
    header oracle/ref/upwelling_bih.h) in place of the harmonic operators; VISC4, TNU4 [m4/s] as roms.in gives them"""
    cs = upwelling(**kw)
    cs["app"] = "upwelling_bih"
    cs["options"] = tuple(o for o in cs["options"] if o not in ("UV_VIS2", "TS_DIF2"))
    cs["mix4"], cs["visc4"], cs["tnu4"] = (1, 1), visc4, tuple(tnu4)
    return cs


def upwelling_bihgeo(**kw):
    """... with the tracers mixed along geopotential surfaces (TS_DIF4 + MIX_GEO_TS, t3dmix4_geo.h: the rotated operator twice);
    the custom application header oracle/ref/upwelling_bihgeo.h"""
    cs = upwelling_bih(**kw)
    cs["app"] = "upwelling_bihgeo"
    cs["options"] = tuple(cs["options"]) + ("MIX_GEO_TS",)
    return cs


def upwelling_bihiso(**kw):
    """... with the tracers mixed along isopycnic surfaces (TS_DIF4 + MIX_ISO_TS, t3dmix4_iso.h: the rotated operator twice);
    the custom application header oracle/ref/upwelling_bihiso.h"""
    cs = upwelling_bih(**kw)
    cs["app"] = "upwelling_bihiso"
    cs["options"] = tuple(cs["options"]) + ("MIX_ISO_TS",)
    return cs


def upwelling_prs31(wj=False, **kw):
    """UPWELLING with the standard density Jacobian (prsgrd31.h: no DJ_GRADPS; wj: WJ_GRADP, the weighted form): the custom
    application headers oracle/ref/upwelling_prs31.h, upwelling_wjgradp.h"""
    cs = upwelling(**kw)
    cs["app"] = "upwelling_wjgradp" if wj else "upwelling_prs31"
    cs["options"] = tuple(cs["options"]) + ("PRSGRD31",) + (("WJ_GRADP",) if wj else ())
    return cs


# the parameter sets of the generic length-scale closure suggested in the reference's input files
# (ROMS/External/roms_upwelling.in:1992-2006): p, m, n, Kmin, Pmin, cmu0, c1, c2, c3m, c3p, sigk, sigp
GLS_SETS = {
    "k-kl": (0.0, 1.0, 1.0, 5.0e-6, 5.0e-6, 0.5544, 0.9, 0.52, 2.5, 1.0, 1.96, 1.96),
    "k-epsilon": (3.0, 1.5, -1.0, 7.6e-6, 1.0e-12, 0.5477, 1.44, 1.92, -0.4, 1.0, 1.0, 1.30),
    "k-omega": (-1.0, 0.5, -1.0, 7.6e-6, 1.0e-12, 0.5477, 0.555, 0.833, -0.6, 1.0, 2.0, 2.0),
    "gen": (2.0, 1.0, -0.67, 1.0e-8, 1.0e-8, 0.5544, 1.00, 1.22, 0.1, 1.0, 0.8, 1.07),
}
GLS_NAMES = ("gls_p", "gls_m", "gls_n", "gls_Kmin", "gls_Pmin", "gls_cmu0", "gls_c1", "gls_c2", "gls_c3m", "gls_c3p",
             "gls_sigk", "gls_sigp")
# the compile-time forms: library name of oracle/ref/build_ref.sh -> cpp options beside GLS_MIXING
GLS_FORMS = {
    "upwelling_gls": ("KANTHA_CLAYSON", "N2S2_HORAVG", "RI_SPLINES"),          # upwelling.h as shipped, -DGLS_MIXING
    "upwelling_gls_ca": ("CANUTO_A", "N2S2_HORAVG", "RI_SPLINES"),             # + MASKING
    "upwelling_gls_cb": ("CANUTO_B", "K_C2ADVECTION", "CHARNOK", "CRAIG_BANNER"),
    "upwelling_gls_gal": ("K_C4ADVECTION", "RI_SPLINES"),
}


def upwelling_gls(form="upwelling_gls", closure="k-epsilon", **kw):
    """UPWELLING with the generic length-scale closure (GLS_MIXING) instead of ANA_VMIX: upwelling.h built with
    -DGLS_MIXING, or one of the custom headers oracle/ref/upwelling_gls_*.h; `closure`: a column of GLS_SETS"""
    cs = upwelling(**kw)
    cs["app"] = form
    cs["options"] = tuple(o for o in cs["options"] if o != "ANA_VMIX") + ("GLS_MIXING",) + \
        (("MASKING",) if form == "upwelling_gls_ca" else ())
    cs["gls_flags"] = GLS_FORMS[form]
    cs.update(dict(zip(GLS_NAMES, GLS_SETS[closure])))
    cs.update(Akk_bak=5.0e-6, Akp_bak=5.0e-6, charnok_alpha=1400.0, zos_hsig_alpha=0.5, sz_alpha=0.25, crgban_cw=100.0)
    return cs


def kelvin_geouv(**kw):
    """KELVIN (open boundaries) with the harmonic viscosity along geopotential surfaces (MIX_GEO_UV): oracle/ref/kelvin_geouv.h"""
    cs = kelvin(**kw)
    cs["app"] = "kelvin_geouv"
    cs["mix_geo_uv"] = 1
    cs["visc2"] = 50.0                   # (roms_kelvin.in has VISC2 = 0: the operator would add zeros)
    return cs


def benchmark_iso(**kw):
    """BENCHMARK with its tracer mixing along isopycnic instead of geopotential surfaces (MIX_ISO_TS with the nonlinear equation of
    state): oracle/ref/benchmark_iso.h"""
    cs = benchmark(**kw)
    cs["app"] = "benchmark_iso"
    cs["options"] = tuple("MIX_ISO_TS" if o == "MIX_GEO_TS" else o for o in cs["options"])
    return cs


def kelvin_gls(**kw):
    """KELVIN (open boundaries) with the generic length-scale closure (Canuto A, smoothing, spline shear; k-epsilon):
    the custom application header oracle/ref/kelvin_gls.h"""
    cs = kelvin(**kw)
    cs["app"] = "kelvin_gls"
    cs["options"] = tuple(cs["options"]) + ("GLS_MIXING",)
    cs["gls_flags"] = ("CANUTO_A", "N2S2_HORAVG", "RI_SPLINES")
    cs.update(dict(zip(GLS_NAMES, GLS_SETS["k-epsilon"])))
    cs.update(Akk_bak=5.0e-6, Akp_bak=5.0e-6, charnok_alpha=1400.0, zos_hsig_alpha=0.5, sz_alpha=0.25, crgban_cw=100.0)
    return cs


def upwelling_my25(form="upwelling_my25", **kw):
    """UPWELLING with the Mellor-Yamada level 2.5 closure (MY25_MIXING): upwelling.h built with -DMY25_MIXING
    (KANTHA_CLAYSON, N2S2_HORAVG, RI_SPLINES), or oracle/ref/upwelling_my25_gal.h (Galperin, K_C4ADVECTION).  The closure
    takes GLS_Kmin, GLS_Pmin (start values) and AKK_BAK from the roms.in block it shares with GLS_MIXING."""
    cs = upwelling(**kw)
    cs["app"] = form
    cs["options"] = tuple(o for o in cs["options"] if o != "ANA_VMIX") + ("MY25_MIXING",)
    cs["gls_flags"] = ("KANTHA_CLAYSON", "N2S2_HORAVG", "RI_SPLINES") if form == "upwelling_my25" else ("K_C4ADVECTION",)
    cs.update(dict(zip(GLS_NAMES, GLS_SETS["k-kl"])))
    cs.update(Akk_bak=5.0e-6, Akp_bak=5.0e-6, charnok_alpha=1400.0, zos_hsig_alpha=0.5, sz_alpha=0.25, crgban_cw=100.0)
    return cs


def upwelling_mask(**kw):
    """UPWELLING with land/sea masking (MASKING): an island and a headland on the southern wall (`land_mask`); the
    custom application header oracle/ref/upwelling_mask.h"""
    cs = upwelling(**kw)
    cs["app"] = "upwelling_mask"
    cs["options"] = tuple(cs["options"]) + ("MASKING",)
    return cs


def clima_arrays(cs, nij, seed=11):
    """The climatology and nudging-coefficient arrays of a test run with climatology nudging (cs["clima"]: bit 0 = 3-D momentum,
    bit itrc = tracer itrc): what set_data.F / ana_nudgcoef.h fill in a run of the reference is input data here.  Per tracer
    (N planes each, tracer-major): tclm around 10 +- 2 | 35 +- 0.5, Tnudgcof between 1/(20 d) and 1/(2 d); uclm, vclm +- 0.1 m/s,
    M3nudgcof likewise.  numpy's default generator: the same doubles on every machine."""
    import numpy as np
    rng = np.random.default_rng(seed)
    N, NT = cs["N"], 2
    day = 86400.0
    out = dict(
        tclm=np.concatenate([10.0 + 2.0 * rng.standard_normal(nij * N), 35.0 + 0.5 * rng.standard_normal(nij * N)]),
        Tnudgcof=(1.0 / (20.0 * day)) + (1.0 / (2.0 * day) - 1.0 / (20.0 * day)) * rng.random(nij * N * NT),
        uclm=0.1 * rng.standard_normal(nij * N), vclm=0.1 * rng.standard_normal(nij * N),
        M3nudgcof=(1.0 / (20.0 * day)) + (1.0 / (2.0 * day) - 1.0 / (20.0 * day)) * rng.random(nij * N))
    # (bit 5: the 2-D momentum, LnudgeM2CLM -- drawn behind the others, so that their doubles stay what the fixtures hold)
    out.update(ubarclm=0.05 * rng.standard_normal(nij), vbarclm=0.05 * rng.standard_normal(nij),
               M2nudgcof=(1.0 / (10.0 * day)) + (1.0 / (1.0 * day) - 1.0 / (10.0 * day)) * rng.random(nij))
    return out


def upwelling_geouv(**kw):
    """UPWELLING + MASKING with the harmonic viscosity along geopotential surfaces (UV_VIS2 + MIX_GEO_UV, uv3dmix2_geo.h: the
    rotated stress tensor); the custom application header oracle/ref/upwelling_geouv.h"""
    cs = upwelling_mask(**kw)
    cs["app"] = "upwelling_geouv"
    cs["mix_geo_uv"] = 1
    return cs


def upwelling_bihgeouv(visc4=4.0e7, **kw):
    """... with the BIHARMONIC viscosity along geopotential surfaces (UV_VIS4 + MIX_GEO_UV, uv3dmix4_geo.h: the rotated stress tensor
    twice) under MASKING, the tracers with their harmonic operator along s-surfaces; the custom application header
    oracle/ref/upwelling_bihgeouv.h.  (VISC4 = 4e7: with the 4e8 of the s-surface cases the masked channel blows up within ten
    steps, in the reference and bit for bit in the oracle.)"""
    cs = upwelling_geouv(**kw)
    cs["app"] = "upwelling_bihgeouv"
    cs["options"] = tuple(o for o in cs["options"] if o != "UV_VIS2")
    cs["mix4"], cs["visc4"], cs["tnu4"] = (1, 0), visc4, (0.0, 0.0)
    return cs


def upwelling_wetdry(Dcrit=0.1, **kw):
    """UPWELLING with land/sea masking and wetting and drying (MASKING + WET_DRY): the custom application header
    oracle/ref/upwelling_wetdry.h.  The land is `land_mask`; the bathymetry and the initial free surface are
    `wetdry_depth`: a beach that rises above the still water level towards the northern wall, and a ridge of water
    in the deep part that runs up it.  Dcrit: DCRIT of roms.in, the depth below which a cell counts as dry."""
    cs = upwelling(**kw)
    cs["app"] = "upwelling_wetdry"
    cs["options"] = tuple(cs["options"]) + ("MASKING",)
    cs["wet_dry"] = 1
    cs["Dcrit"] = Dcrit
    return cs


def upwelling_wetdry_x(variant, **kw):
    """WET_DRY together with what round 6 pinned it with (the custom application headers oracle/ref/upwelling_wetdry_<variant>.h):
    gls | my25 (the closures: gls_*.F, my25_*.F carry no WET_DRY statement of their own), geouv (uv3dmix2_geo.h's wet masks),
    prs31 | prs44 (ru, rv times the wet masks, prsgrd31.h:239,285,323,369, prsgrd44.h:466,500; PJ_GRADP + WET_DRY does not compile
    in the reference: prsgrd40.h:98 uses umask_wet, vmask_wet undeclared)"""
    cs = upwelling_wetdry(**kw)
    cs["app"] = "upwelling_wetdry_" + variant
    if variant in ("gls", "my25"):
        src = upwelling_gls() if variant == "gls" else upwelling_my25()
        cs["options"] = tuple(o for o in cs["options"] if o != "ANA_VMIX") + (("GLS_MIXING",) if variant == "gls" else ("MY25_MIXING",))
        for k in ("gls_flags",) + tuple(GLS_NAMES) + ("Akk_bak", "Akp_bak", "charnok_alpha", "zos_hsig_alpha", "sz_alpha", "crgban_cw"):
            cs[k] = src[k]
    elif variant == "geouv":
        cs["mix_geo_uv"] = 1
    elif variant == "prs31":
        cs["options"] = tuple(cs["options"]) + ("PRSGRD31",)
    elif variant == "prs44":
        cs["prsgrd"] = 44
    elif variant == "iso":
        cs["options"] = tuple(cs["options"]) + ("MIX_ISO_TS",)
        cs["tnu2"] = (20.0, 10.0)                    # (UPWELLING's TNU2 is zero: the operator would add zeros)
    else:
        raise ValueError(variant)
    return cs


def wetdry_depth(cs, LBi, UBi, LBj, UBj):
    """h and the initial zeta of the wetting/drying test case on arrays (LBi:UBi, LBj:UBj), [j, i] here: depth 10 m south
    of row Mm/3, from there a plane beach to -0.3 m (30 cm above the still water level) at the northern wall, roughened
    by 5 cm x ((7 i + 3 j) mod 5) so that the shore line is two-dimensional; zeta: a ridge 0.4 m high, 4 rows either
    side of row Mm/4.  Only +, -, *, / and min/max of exactly representable numbers: numpy and the Fortran host
    (roms_host.f90:wetdry_depths) produce the same doubles."""
    import numpy as np
    Lm, Mm = cs["Lm"], cs["Mm"]
    i = np.arange(LBi, UBi + 1)[None, :] * np.ones((UBj - LBj + 1, 1), dtype=int)
    j = np.arange(LBj, UBj + 1)[:, None] * np.ones((1, UBi - LBi + 1), dtype=int)
    if cs.get("EWperiodic"):
        i = (i - 1) % Lm + 1
    if cs.get("NSperiodic"):
        j = (j - 1) % Mm + 1
    j0 = Mm // 3
    ramp = np.minimum(1.0, np.maximum(0.0, (j - j0).astype(float) / float(Mm + 1 - j0)))
    h = 10.0 - 10.3 * ramp + 0.05 * ((7 * i + 3 * j) % 5).astype(float)
    js = Mm // 4
    zeta = 0.4 * np.maximum(0.0, 1.0 - np.abs(j - js).astype(float) / 4.0)
    return dict(h=h, zeta=zeta)


def benchmark_mask(**kw):
    """BENCHMARK (KPP, bulk fluxes, nonlinear EOS, geopotential mixing) with land/sea masking: the custom application
    header oracle/ref/benchmark_mask.h, land of `land_mask`"""
    cs = benchmark(**kw)
    cs["app"] = "benchmark_mask"
    cs["options"] = tuple(cs["options"]) + ("MASKING",)
    return cs


def benchmark_wetdry(Dcrit=0.1, **kw):
    """BENCHMARK (KPP, COARE bulk fluxes, solar source, nonlinear EOS, geopotential tracer mixing) with land/sea masking and
    wetting and drying: the custom application header oracle/ref/benchmark_wetdry.h, land of `land_mask`, bathymetry and
    initial free surface of `wetdry_depth` (the beach and the ridge of the UPWELLING wetting/drying case)."""
    cs = benchmark_mask(**kw)
    cs["app"] = "benchmark_wetdry"
    cs["wet_dry"] = 1
    cs["Dcrit"] = Dcrit
    return cs


def land_mask(cs, LBi, UBi, LBj, UBj):
    """rmask, umask, vmask, pmask of the masked test cases on arrays (LBi:UBi, LBj:UBj), Fortran order [j, i] here.
    Land: an island of 3 x 3 cells east of Lm/3 around Mm/2, and a headland two cells wide at 2 Lm/3 from the
    southern wall to Mm/4 (wall row included).  The xi direction wraps where the case is periodic.  u, v masks are the
    products of the two adjacent rho masks (ana_mask.h:224-233); the psi mask is the slipperiness mask metrics.F:527-583
    derives from rmask: 1 with at most one land cell among the four around the point, 2 (no-slip) where the two land
    cells share a side of it, 0 otherwise."""
    import numpy as np
    Lm, Mm = cs["Lm"], cs["Mm"]
    i = np.arange(LBi, UBi + 1)[None, :] * np.ones((UBj - LBj + 1, 1), dtype=int)
    j = np.arange(LBj, UBj + 1)[:, None] * np.ones((1, UBi - LBi + 1), dtype=int)
    if cs.get("EWperiodic"):
        i = (i - 1) % Lm + 1
    if cs.get("NSperiodic"):
        j = (j - 1) % Mm + 1
    i0, j0, i1 = Lm // 3 + 1, Mm // 2, 2 * Lm // 3 + 1
    land = ((i >= i0) & (i <= i0 + 2) & (j >= j0) & (j <= j0 + 2)) | ((i >= i1) & (i <= i1 + 1) & (j <= Mm // 4))
    r = np.where(land, 0.0, 1.0)
    u, v, p = np.ones_like(r), np.ones_like(r), np.ones_like(r)
    u[:, 1:] = r[:, :-1] * r[:, 1:]
    v[1:, :] = r[:-1, :] * r[1:, :]
    if cs.get("EWperiodic"):            # (the first line of the array: its lower neighbour is the periodic image)
        u[:, 0] = r[:, Lm - 1] * r[:, 0]
    if cs.get("NSperiodic"):
        v[0, :] = r[Mm - 1, :] * r[0, :]
    a, b, c, d = r[1:, :-1], r[1:, 1:], r[:-1, :-1], r[:-1, 1:]      # (i-1,j) (i,j) (i-1,j-1) (i,j-1)
    nland = 4 - (a + b + c + d)
    side = ((a + b == 0) | (c + d == 0) | (a + c == 0) | (b + d == 0)) & (nland == 2)
    p[1:, 1:] = np.where(nland <= 1, 1.0, np.where(side, 2.0, 0.0))
    # metrics.F computes psi points IstrP:IendP x JstrP:JendP and exchanges the periodic images; the outermost
    # line of the array on each side is never written (0, and never read)
    ii = np.arange(LBi, UBi + 1)
    jj = np.arange(LBj, UBj + 1)
    p[:, (ii == LBi) | (ii == UBi) if cs.get("EWperiodic") else (ii < 1) | (ii > Lm + 1)] = 0.0
    p[(jj == LBj) | (jj == UBj) if cs.get("NSperiodic") else (jj < 1) | (jj > Mm + 1), :] = 0.0
    return dict(rmask=r, umask=u, vmask=v, pmask=p)


def benchmark(Lm=512, Mm=64, N=30, NtileI=1, NtileJ=1, ntimes=200, hadv=("U3", "U3"), vadv=("C4", "C4")):
    """roms_benchmark1.in"""
    return dict(
        app="benchmark", Lm=Lm, Mm=Mm, N=N, NtileI=NtileI, NtileJ=NtileJ, ndtfast=20, ntimes=ntimes,
        Vtransform=2, Vstretching=4, EWperiodic=1, NSperiodic=0, hadv=tuple(hadv), vadv=tuple(vadv),
        lmd_Jwt=1, dt=150.0, theta_s=0.0, theta_b=0.0, Tcline=400.0, rho0=1025.0, R0=1027.0, T0=10.0,
        S0=35.0, Tcoef=1.7e-4, Scoef=7.6e-4, visc2=5000.0, tnu2=(500.0, 500.0),
        Akt_bak=(1.0e-5, 1.0e-5), Akv_bak=1.0e-4, rdrg=3.0e-4, rdrg2=3.0e-3, Zob=0.02, Zos=0.02,
        gamma2=1.0, dstart=0.0, blk_ZQ=10.0, blk_ZT=10.0, blk_ZW=10.0,
        options=("UV_ADV", "UV_COR", "UV_VIS2", "TS_DIF2", "MIX_GEO_TS", "CURVGRID", "NONLIN_EOS",
                 "UV_QDRAG", "LMD_MIXING", "BULK_FLUXES", "SOLAR_SOURCE", "SALINITY", "SPHERICAL",
                 "APP_BENCHMARK"),
    )


def gls_cfg(c, cs, flag_table):
    """the GLS_MIXING tail of a configuration record (roms_hip_config / orc_cfg)"""
    c.Zos = cs["Zos"]
    if "gls_flags" not in cs:
        return
    c.gls_flags = 0
    for n in cs["gls_flags"]:
        c.gls_flags |= flag_table[n]
    for n in GLS_NAMES + ("Akk_bak", "Akp_bak", "charnok_alpha", "crgban_cw"):
        setattr(c, n, cs[n])
    for e, k in enumerate(cs.get("lbc_tke", ())):        # LBC(isMtke) (west, south, east, north): "Clo", "Gra", "Rad", "Per"
        c.lbc_tke[e] = LBC_KINDS[k]


def hip_cfg(cs, hc, nfast, weight, sc_r, Cs_r, sc_w, Cs_w, device=0):
    """roms_hip_config for a single tile covering the domain (include/roms_hip.h)."""
    from . import hiplib
    c = hiplib.Config()
    c.abi_version, c.device = hiplib.ABI_VERSION, device
    c.Lm, c.Mm, c.N, c.NT, c.NAT = cs["Lm"], cs["Mm"], cs["N"], 2, 2
    hs = [SCHEME[x] for x in cs["hadv"]]
    vs = [SCHEME[x] for x in cs["vadv"]]
    c.Nghost = 3 if (hiplib.MPDATA in hs or hiplib.HSIMT in hs or cs.get("mix4", (0, 0))[0]) else 2     # inp_par.F:210-223
    Im = cs["Lm"] + ((cs["Lm"] + 2) // 2 - (cs["Lm"] + 1) // 2)
    Jm = cs["Mm"] + ((cs["Mm"] + 2) // 2 - (cs["Mm"] + 1) // 2)
    c.LBi, c.UBi = (-c.Nghost, Im + c.Nghost) if cs["EWperiodic"] else (0, Im + 1)
    c.LBj, c.UBj = (-c.Nghost, Jm + c.Nghost) if cs["NSperiodic"] else (0, Jm + 1)
    c.NtileI = c.NtileJ = 1
    c.tile = 0
    c.EWperiodic, c.NSperiodic = cs["EWperiodic"], cs["NSperiodic"]
    opt = 0
    for name in cs["options"]:
        opt |= hiplib.OPTIONS[name]
    if "mix4" in cs:        # biharmonic cases: the library keeps its harmonic operators (zero coefficients add exact zeros)
        opt |= hiplib.OPTIONS["UV_VIS2"] | hiplib.OPTIONS["TS_DIF2"]
        opt |= (hiplib.OPTIONS["UV_VIS4"] if cs["mix4"][0] else 0) | (hiplib.OPTIONS["TS_DIF4"] if cs["mix4"][1] else 0)
    if cs.get("mix_geo_uv"):
        opt |= hiplib.OPTIONS["MIX_GEO_UV"]
    if cs.get("ddmix"):     # LMD_DDMIX
        opt |= hiplib.OPTIONS["LMD_DDMIX"]
    if cs.get("bkpp"):      # LMD_BKPP
        opt |= hiplib.OPTIONS["LMD_BKPP"]
    if cs.get("prsgrd"):    # PJ_GRADPQ2 / PJ_GRADPQ4
        opt |= hiplib.OPTIONS["PRSGRD%d" % cs["prsgrd"]]
    if cs.get("clima"):     # climatology nudging: bit 0 the 3-D momentum, bit itrc tracer itrc
        opt |= hiplib.OPTIONS["NUDGE_M3CLM"] if cs["clima"] & 1 else 0
        opt |= hiplib.OPTIONS["NUDGE_M2CLM"] if cs["clima"] & 32 else 0
        for it in range(1, 5):
            opt |= hiplib.OPTIONS["NUDGE_TCLM%d" % it] if cs["clima"] & (1 << it) else 0
    if cs.get("wet_dry"):   # WET_DRY with DCRIT; the momentum diagnostics beside the tracer ones (ABI version 4: option bits)
        opt |= hiplib.OPTIONS["WET_DRY"]
        c.Dcrit = cs["Dcrit"]
    if cs.get("dia_uv"):
        opt |= hiplib.OPTIONS["DIAGNOSTICS_UV"]
    c.options = opt
    for i in range(2):
        c.hadv[i], c.vadv[i] = hs[i], vs[i]
    c.Istr, c.Iend, c.Jstr, c.Jend = 1, cs["Lm"], 1, cs["Mm"]
    c.west_edge = c.east_edge = c.south_edge = c.north_edge = 1
    c.ntfirst = c.ntstart = 1
    c.ndtfast, c.nfast, c.ninfo = cs["ndtfast"], nfast, 0
    c.dt = cs["dt"]
    c.dtfast = cs["dt"] / float(cs["ndtfast"])
    w = np.asarray(weight, dtype=np.float64).reshape(2, -1)
    for k in range(w.shape[1]):
        c.weight[0][k + 1] = w[0, k]
        c.weight[1][k + 1] = w[1, k]
    c.rho0, c.g, c.lambda_, c.gamma2, c.Cp = cs["rho0"], 9.81, 1.0, cs["gamma2"], 3985.0
    c.R0, c.T0, c.S0, c.Tcoef, c.Scoef = cs["R0"], cs["T0"], cs["S0"], cs["Tcoef"], cs["Scoef"]
    c.hc, c.Vtransform = hc, cs["Vtransform"]
    c.rdrg, c.rdrg2, c.Zob = cs["rdrg"], cs["rdrg2"], cs["Zob"]
    c.Akt_bak[0], c.Akt_bak[1], c.Akv_bak = cs["Akt_bak"][0], cs["Akt_bak"][1], cs["Akv_bak"]
    c.dstart = cs["dstart"]
    c.blk_ZQ, c.blk_ZT, c.blk_ZW, c.lmd_Jwt = cs["blk_ZQ"], cs["blk_ZT"], cs["blk_ZW"], cs["lmd_Jwt"]
    code, sc = lbc_codes(cs), obc_scales(cs)
    c.obcfac = cs.get("obcfac", 0.0)
    c.volcons = cs.get("volcons", 0)           # VolCons(west|south|east|north) of roms.in as bits 0..3 (obc_volcons.F)
    for e in range(4):
        for v in range(7):
            c.lbc[e][v] = code[v][e]
        for n in ("FSobc_in", "FSobc_out", "M2obc_in", "M2obc_out", "M3obc_in", "M3obc_out"):
            getattr(c, n)[e] = sc[n][e]
        for it in range(2):
            c.Tobc_in[it][e], c.Tobc_out[it][e] = sc["Tobc_in"][it][e], sc["Tobc_out"][it][e]
    gls_cfg(c, cs, hiplib.GLS_FLAGS)
    for k in range(cs["N"]):
        c.sc_r[k], c.Cs_r[k] = sc_r[k], Cs_r[k]
    for k in range(cs["N"] + 1):
        c.sc_w[k], c.Cs_w[k] = sc_w[k], Cs_w[k]
    return c
