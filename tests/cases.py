"""Test-side companions of roms_amd/cases.py: the same case dicts turned into the configuration records of
the oracle (oracle/orc.h) and of the reference glue (oracle/ref/ref_glue.F90).  TEST INFRASTRUCTURE."""
import numpy as np

from roms_amd.cases import LBC_KINDS, SCHEME, benchmark, hip_cfg, upwelling, upwelling_kpp, upwelling_logdrag, upwelling_noadv, upwelling_mask, upwelling_geouv, upwelling_bihgeouv, upwelling_bihiso, clima_arrays, upwelling_wetdry, upwelling_wetdry_x, wetdry_depth, benchmark_mask, benchmark_wetdry, land_mask, kelvin, seamount, grav_adj, overflow, upwelling_prs31, upwelling_bih, upwelling_bihgeo, upwelling_prs40, upwelling_prs4x, upwelling_kpp_ddmix, benchmark_ddmix, benchmark_bkpp, upwelling_kpp_bkpp, with_bkpp, benchmark_wetdry_ddmix, ddmix_state, upwelling_gls, upwelling_my25, kelvin_gls, kelvin_geouv, benchmark_iso, gls_cfg, GLS_NAMES, GLS_SETS, lbc_codes, obc_scales  # noqa: F401  (re-exported)


def ref_params(cs):
    """(ipar, rpar) for oracle/ref/ref_glue.F90:ref_configure."""
    ipar = [cs["Lm"], cs["Mm"], cs["N"], cs["NtileI"], cs["NtileJ"], cs["ndtfast"], cs["ntimes"],
            cs["Vtransform"], cs["Vstretching"], cs["EWperiodic"], cs["NSperiodic"],
            SCHEME[cs["hadv"][0]], SCHEME[cs["vadv"][0]], SCHEME[cs["hadv"][1]], SCHEME[cs["vadv"][1]],
            cs["lmd_Jwt"]]
    rpar = [cs["dt"], cs["theta_s"], cs["theta_b"], cs["Tcline"], cs["rho0"], cs["R0"], cs["T0"],
            cs["S0"], cs["Tcoef"], cs["Scoef"],
            # (biharmonic variants, cs["mix4"]: VISC4 and TNU4 of roms.in in the same slots, ref_glue.F90:ref_configure)
            cs["visc4"] if cs.get("mix4", (0, 0))[0] else cs["visc2"],
            cs["tnu4"][0] if cs.get("mix4", (0, 0))[1] else cs["tnu2"][0],
            cs["tnu4"][1] if cs.get("mix4", (0, 0))[1] else cs["tnu2"][1],
            cs["Akt_bak"][0], cs["Akt_bak"][1], cs["Akv_bak"], cs["rdrg"], cs["rdrg2"], cs["Zob"],
            cs["Zos"], cs["gamma2"], cs["dstart"], cs["blk_ZQ"], cs["blk_ZT"], cs["blk_ZW"]]
    ipar = ipar + [0] * (64 - len(ipar))
    rpar = rpar + [0.0] * (96 - len(rpar))
    # open boundaries (ref_glue.F90:ref_configure): kinds from ipar(17), nudging scales from rpar(26); ipar(45): all
    # of mod_boundary's arrays allocated (the test feeds boundary data)
    code = lbc_codes(cs)
    for v in range(7):
        for e in range(4):
            ipar[16 + 4 * v + e] = code[v][e]
    ipar[44] = 1 if cs.get("bry_all") else 0
    for e, k in enumerate(cs.get("lbc_tke", ())):        # LBC(isMtke): ipar(46:49)
        ipar[45 + e] = LBC_KINDS[k]
    sc = obc_scales(cs)
    for q, n in enumerate(["FSobc_in", "FSobc_out", "M2obc_in", "M2obc_out", "M3obc_in", "M3obc_out"]):
        for e in range(4):
            rpar[25 + 4 * q + e] = sc[n][e]
    for it in range(2):
        for e in range(4):
            rpar[49 + 8 * it + e] = sc["Tobc_in"][it][e]
            rpar[53 + 8 * it + e] = sc["Tobc_out"][it][e]
    ipar[49] = cs.get("volcons", 0)     # ref_glue.F90: ipar(50), VolCons(iwest..inorth) as bits 0..3 (round 6: obc_volcons.F)
    rpar[83] = cs.get("Dcrit", 0.0)     # ref_glue.F90: rpar(84), DCRIT (WET_DRY builds)
    rpar[84] = cs.get("obcfac", 0.0)    # ref_glue.F90: rpar(85), OBCFAC (round 6: the radiation conditions under climatology nudging)
    if "gls_flags" in cs:               # ref_glue.F90: rpar(66..83)
        for k, n in enumerate(GLS_NAMES + ("Akk_bak", "Akp_bak", "charnok_alpha", "zos_hsig_alpha", "sz_alpha", "crgban_cw")):
            rpar[65 + k] = cs[n]
    return ipar, rpar


def oracle_cfg(cs, hc, nfast, weight):
    """orc_cfg for oracle/orc.h.  hc, nfast and weight(2,2*ndtfast) come from the host set-up
    (set_scoord / set_weights) or from the golden fixture."""
    from oracle import orc
    c = orc.Cfg()
    c.Lm, c.Mm, c.N, c.NT, c.NAT = cs["Lm"], cs["Mm"], cs["N"], 2, 2
    hs = [SCHEME[x] for x in cs["hadv"]]
    vs = [SCHEME[x] for x in cs["vadv"]]
    c.Nghost = 3 if (orc.MPDATA in hs or orc.HSIMT in hs or cs.get("mix4", (0, 0))[0]) else 2     # inp_par.F:210-223
    c.NtileI, c.NtileJ = cs["NtileI"], cs["NtileJ"]
    c.EWperiodic, c.NSperiodic = cs["EWperiodic"], cs["NSperiodic"]
    opt = 0
    for name in cs["options"]:
        opt |= getattr(orc, name)
    c.options = opt
    for i in range(2):
        c.hadv[i], c.vadv[i] = hs[i], vs[i]
    c.ntfirst = c.ntstart = 1
    c.ndtfast, c.nfast = cs["ndtfast"], nfast
    c.dt = cs["dt"]
    c.dtfast = cs["dt"] / float(cs["ndtfast"])
    w = np.asarray(weight, dtype=np.float64).reshape(2, -1)
    for k in range(w.shape[1]):
        c.weight[0][k + 1] = w[0, k]
        c.weight[1][k + 1] = w[1, k]
    c.obcfac = cs.get("obcfac", 0.0)
    c.volcons = cs.get("volcons", 0)
    c.rho0, c.g, c.lambda_, c.gamma2, c.Cp = cs["rho0"], 9.81, 1.0, cs["gamma2"], 3985.0
    c.R0, c.T0, c.S0, c.Tcoef, c.Scoef = cs["R0"], cs["T0"], cs["S0"], cs["Tcoef"], cs["Scoef"]
    c.hc, c.Vtransform = hc, cs["Vtransform"]
    c.rdrg, c.rdrg2, c.Zob = cs["rdrg"], cs["rdrg2"], cs["Zob"]
    c.Akt_bak[0], c.Akt_bak[1], c.Akv_bak = cs["Akt_bak"][0], cs["Akt_bak"][1], cs["Akv_bak"]
    c.dstart = cs["dstart"]
    c.blk_ZQ, c.blk_ZT, c.blk_ZW, c.lmd_Jwt = cs["blk_ZQ"], cs["blk_ZT"], cs["blk_ZW"], cs["lmd_Jwt"]
    c.cc1, c.cc2, c.cc3 = 0.25, 0.5, 1.0 / 12.0
    code = lbc_codes(cs)
    sc = obc_scales(cs)
    for e in range(4):
        for v in range(5):
            c.lbc[e][v] = code[v][e]
        for it in range(2):
            c.lbc[e][5 + it] = code[5 + it][e]
            c.Tobc_in[it][e], c.Tobc_out[it][e] = sc["Tobc_in"][it][e], sc["Tobc_out"][it][e]
        for n in ("FSobc_in", "FSobc_out", "M2obc_in", "M2obc_out", "M3obc_in", "M3obc_out"):
            getattr(c, n)[e] = sc[n][e]
    gls_cfg(c, cs, orc.GLS_FLAGS)
    return c


