#!/usr/bin/env python3
"""Classify every keyword of the reference's roms.in reader (ROMS/Utility/read_phypar.F: one CASE per keyword) for
THIS build and write
    docs/ROMS_IN_KEYWORDS.md            the table (keyword, class, why)
    roms_amd/host/roms_in_inert.inc     the CASE lists the Fortran reader (roms_host.f90:read_roms_in) includes

Classes
  honoured   read and used by the host / the library
  checked    selects code this build does not have: the value must leave it switched off, else exit_flag 5
  inert      cannot change the forward time step of this build (output selection, file names, other drivers'
             parameters, parameters of cpp options the library does not carry -- an application header that
             defines such an option is stopped by options_from_defines before any keyword matters)
Run in the build container (reads /root/reference).  The keyword NAMES are the roms.in interface, not code."""
import os
import re
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = "/root/reference/ROMS/Utility/read_phypar.F"

HONOURED = """TITLE MyAppCPP Lm Mm N NAT NtileI NtileJ NTIMES DT NDTFAST NINFO Hadvection Vadvection NRREC LcycleRST NRST NHIS
ININAME RSTNAME HISNAME NAVG NTSAVG AVGNAME NDIA NTSDIA DIANAME TNU2 VISC2 TNU4 VISC4 DCRIT AKT_BAK AKV_BAK RDRG RDRG2 Zob Zos BLK_ZQ BLK_ZT BLK_ZW WTYPE
Vtransform Vstretching THETA_S THETA_B TCLINE RHO0 DSTART TIME_REF R0 T0 S0 TCOEF SCOEF GAMMA2
TNUDG ZNUDG M2NUDG M3NUDG OBCFAC
AKK_BAK AKP_BAK GLS_P GLS_M GLS_N GLS_Kmin GLS_Pmin GLS_CMU0 GLS_C1 GLS_C2 GLS_C3M GLS_C3P GLS_SIGK GLS_SIGP CHARNOK_ALPHA CRGBAN_CW""".split()
HONOURED += ["LBC(isFsur)", "LBC(isUbar)", "LBC(isVbar)", "LBC(isUvel)", "LBC(isVvel)", "LBC(isMtke)", "LBC(isTvar)"]
HONOURED_PREFIX = ["Aout(", "Hout("]        # the switches the averages / history writers know; the others are inert (below)
HONOURED_AOUT = "idFsur idUbar idVbar idUvel idVvel idOvel idWvel idDano idTvar idZZav idU2av idV2av idUUav idVVav idUVav idHUav idHVav idTTav idUTav idVTav iHUTav iHVTav".split()
HONOURED_HOUT = "idFsur idUbar idVbar idUvel idVvel idWvel idOvel idTvar idDano idVvis idTdif idSdif idHsbl idMtke idMtls".split()

CHECKED = {
    "Ngrids": "must be 1 (nesting is not built)", "NestLayers": "must be 1", "GridsInLayer": "must be 1",
    "NBT": "must be 0 (biology)", "NST": "must be 0 (sediment)", "NPT": "must be 0 (passive tracers)",
    "NCS": "must be 0 (cohesive sediment)", "NNS": "must be 0 (non-cohesive sediment)",
    "LuvSrc": "must be F (point sources)", "LwSrc": "must be F", "LtracerSrc": "must be F",
    "LuvSponge": "must be F (sponge layers)", "LtracerSponge": "must be F",
    "LsshCLM": "must be F (climatology)", "Lm2CLM": "must be F", "Lm3CLM": "must be F", "LtracerCLM": "must be F",
    "LnudgeM2CLM": "must be F (climatology nudging)", "LnudgeM3CLM": "must be F", "LnudgeTCLM": "must be F",
    "VolCons(west)": "must be F (volume conservation at open boundaries)", "VolCons(east)": "must be F",
    "VolCons(south)": "must be F", "VolCons(north)": "must be F",
}

INERT_RULES = [   # (regex, why)
    (r"^(Hout|Qout|Aout|Dout)\(", "output selection: which variables a history / quicksave / averages / diagnostics file holds"),
    (r"NAME$|NAM$|^(FCTnameA|FCTnameB|FOInameA|FOInameB|INP_LIB|OUT_LIB|NGCNAME|GRXNAME)$", "file name / I-O library choice"),
    (r"^PIO_|^NC_", "parallel / compressed NetCDF I-O settings"),
    (r"^(NDEF|LDEFOUT|NDIA|NSTA|NFLT|NQCK|NXTR|NTSDIA|ExtractFlag|NBCFILES|NCLMFILES|NFFILES|NUSER|USER|TITLE|VARNAME)", "output frequency / bookkeeping"),
    (r"^ad_|^(NADJ|NTLM|NSFF|NOBC|Nouter|Ninner|Nintervals|Nsaddle|NEV|NCV|Ritz_tol|MaxIterGST|LmultiGST|LrstGST|NGST|LcycleADJ|LcycleTLM|NTIMES_ANA|NTIMES_FCT|ERstr|ERend|DstrS|DendS|KstrS|KendS)$|^(Lstate|Fstate|SO_sdev|SO_decay)", "adjoint / tangent linear / 4D-Var / stability drivers: not the nonlinear forward step"),
    (r"^(TKENU2|TKENU4)$", "lateral mixing of the turbulent fields: no code of gls_prestep.F / gls_corstep.F reads them"),
    (r"^(ZOS_HSIG_ALPHA|SZ_ALPHA|WEC_ALPHA|AKT_LIMIT|AKV_LIMIT|BVF_BAK)$", "parameters of options the library does not carry (ZOS_HSIG, TKE_WAVEDISS, WEC, LIMIT_VDIFF / LIMIT_VVISC, BVF_MIXING): a header defining one is stopped (exit_flag 5)"),
    (r"^(DCRIT)$", "wetting and drying depth: WET_DRY is stopped by the header reader"),
    (r"^(LEVSFRC|LEVBFRC)$", "BODYFORCE levels: BODYFORCE is stopped by the header reader"),
    (r"^(Lnodal|TIDE_START)$", "tidal forcing: not built (SSH_TIDES / UV_TIDES stopped by the header reader)"),
    (r"^(Nbed)$", "sediment bed layers: SEDIMENT is not built"),
    (r"^LBC\(", "boundary conditions of variables this build does not step (Stokes drift, ...)"),
    (r"^ad_LBC\(|^ad_VolCons\(", "adjoint boundary conditions"),
]


def keywords():
    txt = open(REF).read()
    return sorted(set(re.findall(r"CASE \('([A-Za-z0-9_()%]*)'", txt)))


def classify(k):
    if k in HONOURED:
        return "honoured", "read by roms_host.f90:read_roms_in"
    m = re.match(r"^(Aout|Hout)\((\w+)\)$", k)
    if m and m.group(2) in (HONOURED_AOUT if m.group(1) == "Aout" else HONOURED_HOUT):
        return "honoured", "switch of the averages / history writer (roms_output.f90)"
    if re.match(r"^Dout\(iT(rate|hadv|xadv|yadv|vadv|hdif|xdif|ydif|sdif|vdif)\)$", k):
        return "honoured", "switch of the diagnostics writer, tracer terms (DIAGNOSTICS_TS; roms_output.f90)"
    if re.match(r"^Dout\(M2(rate|pgrd|fcor|hadv|xadv|yadv|hvis|xvis|yvis|sstr|bstr)\)$", k) or \
       re.match(r"^Dout\(M3(rate|pgrd|fcor|hadv|xadv|yadv|vadv|hvis|xvis|yvis|vvis)\)$", k):
        return "honoured", "switch of the diagnostics writer, momentum terms (DIAGNOSTICS_UV; roms_output.f90)"
    if k in CHECKED:
        return "checked", CHECKED[k]
    for rx, why in INERT_RULES:
        if re.search(rx, k):
            return "inert", why
    return "unclassified", ""


def main():
    ks = keywords()
    rows = [(k,) + classify(k) for k in ks]
    bad = [r for r in rows if r[1] == "unclassified"]
    if bad:
        sys.exit("unclassified keywords: " + " ".join(r[0] for r in bad))
    n = {c: sum(1 for r in rows if r[1] == c) for c in ("honoured", "checked", "inert")}
    with open(os.path.join(ROOT, "docs", "ROMS_IN_KEYWORDS.md"), "w") as f:
        f.write("# roms.in keywords (ROMS/Utility/read_phypar.F) and what this build does with each\n\n")
        f.write("Generated by `tools/roms_in_keywords.py` from the reference's reader (one `CASE` per keyword, "
                f"{len(ks)} distinct names): **{n['honoured']} honoured, {n['checked']} checked, {n['inert']} inert**. "
                "`honoured` = read and used; `checked` = selects code this build does not have, so the value must leave it "
                "switched off or the set-up stops with `exit_flag` 5 and the reason; `inert` = cannot change the forward time "
                "step of this build (the reader counts them, `roms_host_counts`). A keyword the reference's reader does not "
                "know either is counted as unknown; the `ref`-marked test feeds every `ROMS/External/roms_*.in` and "
                "requires zero unknown keywords and either a complete set-up or a stop with a reason.\n\n")
        f.write("| keyword | class | why |\n|---|---|---|\n")
        for k, c, why in rows:
            f.write(f"| `{k}` | {c} | {why} |\n")
    inert = [r[0] for r in rows if r[1] == "inert"]
    with open(os.path.join(ROOT, "roms_amd", "host", "roms_in_inert.inc"), "w") as f:
        f.write("!  generated by tools/roms_in_keywords.py: roms.in keywords that cannot change the forward time step of this build\n")
        f.write("!  (docs/ROMS_IN_KEYWORDS.md has the reason for each)\n")
        for a in range(0, len(inert), 4):
            f.write("          CASE (" + ", ".join("'%s'" % k for k in inert[a:a + 4]) + ")\n")
            f.write("            n_inert_keys=n_inert_keys+1\n")
    print(f"{len(ks)} keywords: {n}")


if __name__ == "__main__":
    main()
