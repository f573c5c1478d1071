// g_gls.cpp -- launch sequences of the generic length-scale closure (GLS_MIXING): gls_prestep (main3d.F:636) and
// gls_corstep (main3d.F:1021); the closure's derived constants (gls_corstep.F:250-340, mod_scalars.F:4715-4766).
#include "roms_host.h"
#include <cmath>
#include <cstdlib>
#include "k_gls.h"

static GlsArgs gls_args(roms_hip_ctx *c) {
  const roms_hip_config &cf = c->cfg;
  GlsArgs a = {};
  a.G = c->G;
  a.Fv = c->F;
  const int flags = cf.gls_flags;
  a.flags = flags;
  a.my25 = (cf.options & ROMS_MY25_MIXING) != 0;
  a.my_B1p2o3 = pow(16.6, 2.0 / 3.0);            // mod_scalars.F:4753
  const double vonKar = 0.41, gls_p = cf.gls_p, gls_m = cf.gls_m, gls_n = cf.gls_n, cmu0 = cf.gls_cmu0;
  a.gls_m = gls_m; a.gls_n = gls_n; a.Kmin = cf.gls_Kmin; a.Pmin = cf.gls_Pmin; a.cmu0 = cmu0;
  a.c1 = cf.gls_c1; a.c2 = cf.gls_c2; a.c3m = cf.gls_c3m; a.c3p = cf.gls_c3p; a.sigk = cf.gls_sigk; a.sigp = cf.gls_sigp;
  a.Akk_bak = cf.Akk_bak; a.Akp_bak = cf.Akp_bak;
  a.Zos_min = fmax(cf.Zos, 0.0001);
  a.Zob_min = fmax(cf.Zob, 0.0001);          // ZoBot = Zob (mod_grid.F:1380)
  a.charnok_alpha = cf.charnok_alpha; a.crgban_cw = cf.crgban_cw;
  a.Lmy25 = (gls_p == 0.0) && (gls_n == 1.0) && (gls_m == 1.0);
  a.L_sft = vonKar;
  if (flags & ROMS_GLS_CRAIG_BANNER) {
    const double cb_wallE = a.Lmy25 ? 1.25 : 1.0;
    const double cff1 = sqrt(1.5 * cf.gls_sigk) * cmu0 / a.L_sft;
    a.sigp_cb = (a.L_sft * a.L_sft) / ((cmu0 * cmu0) * cf.gls_c2 * cb_wallE) *
                ((gls_n * gls_n) - cff1 * gls_n / 3.0 * (4.0 * gls_m + 1.0) + (cff1 * cff1) * gls_m / 9.0 * (2.0 + 4.0 * gls_m));
  } else a.sigp_cb = cf.gls_sigp;
  a.ogls_sigp = 1.0 / a.sigp_cb;
  a.sqrt2 = sqrt(2.0);
  a.cmu_fac1 = pow(cmu0, -gls_p / gls_n);
  a.cmu_fac2 = pow(cmu0, 3.0 + gls_p / gls_n);
  a.cmu_fac3 = 1.0 / pow(cmu0, 2.0);
  a.cmu_fac4 = pow(1.5 * cf.gls_sigk, 1.0 / 3.0) / pow(cmu0, 4.0 / 3.0);
  a.cmu0p = pow(cmu0, gls_p);
  a.fac2 = a.cmu0p * gls_n * pow(vonKar, gls_n);
  a.fac3 = a.cmu0p * gls_n;
  a.fac4 = a.cmu0p;
  a.fac5 = pow(0.56, 0.5 * gls_n) * a.cmu0p;
  a.fac6 = 8.0 / pow(cmu0, 6.0);
  a.cmu0c = cmu0 * cmu0 * cmu0;
  a.crg23 = pow(cf.crgban_cw, 2.0 / 3.0);
  a.exp1 = 1.0 / gls_n; a.texp1 = gls_m / gls_n; a.texp2 = 0.5 + gls_m / gls_n; a.texp4 = gls_m + 0.5 * gls_n;
  // stability functions
  double L1 = 0, L2 = 0, L3 = 0, L4 = 0, L5 = 0, L6 = 0, L7 = 0, L8 = 0;
  if (flags & ROMS_GLS_CANUTO_A) {
    a.Gh0 = 0.0329; a.Ghcri = 0.03;
    L1 = 0.107; L2 = 0.0032; L3 = 0.0864; L4 = 0.12; L5 = 11.9; L6 = 0.4; L7 = 0.0; L8 = 0.48;
  } else if (flags & ROMS_GLS_CANUTO_B) {
    a.Gh0 = 0.0444; a.Ghcri = 0.0414;
    L1 = 0.127; L2 = 0.00336; L3 = 0.0906; L4 = 0.101; L5 = 11.2; L6 = 0.4; L7 = 0.0; L8 = 0.318;
  } else { a.Gh0 = 0.028; a.Ghcri = 0.02; }
  a.Ghmin = -0.28; a.E2 = 1.33;
  if (flags & (ROMS_GLS_CANUTO_A | ROMS_GLS_CANUTO_B)) {
    a.s0 = 3.0 / 2.0 * L1 * (L5 * L5);
    a.s1 = -L4 * (L6 + L7) + 2.0 * L4 * L5 * (L1 - 1.0 / 3.0 * L2 - L3) + 3.0 / 2.0 * L1 * L5 * L8;
    a.s2 = -3.0 / 8.0 * L1 * (L6 * L6 - L7 * L7);
    a.s4 = 2.0 * L5;
    a.s5 = 2.0 * L4;
    a.s6 = 2.0 / 3.0 * L5 * (3.0 * (L3 * L3) - L2 * L2) - 1.0 / 2.0 * L5 * L1 * (3.0 * L3 - L2) + 3.0 / 4.0 * L1 * (L6 - L7);
    a.b0 = 3.0 * (L5 * L5);
    a.b1 = L5 * (7.0 * L4 + 3.0 * L8);
    a.b2 = L5 * L5 * (3.0 * (L3 * L3) - L2 * L2) - 3.0 / 4.0 * (L6 * L6 - L7 * L7);
    a.b3 = L4 * (4.0 * L4 + 3.0 * L8);
    a.b5 = 1.0 / 4.0 * (L2 * L2 - 3.0 * (L3 * L3)) * (L6 * L6 - L7 * L7);
    a.b4 = L4 * (L2 * L6 - 3.0 * L3 * L7 - L5 * (L2 * L2 - L3 * L3)) + L5 * L8 * (3.0 * (L3 * L3) - L2 * L2);
  }
  const double A1 = 0.92, A2 = 0.74, B1 = 16.6, B2 = 10.1, C1 = 0.08, C2 = 0.7, C3 = 0.2;
  a.B1pm1o3 = 1.0 / pow(B1, 1.0 / 3.0);
  a.Sm2 = 9.0 * A1 * A2;
  a.Sh1 = A2 * (1.0 - 6.0 * A1 / B1);
  if (flags & ROMS_GLS_KANTHA_CLAYSON) {
    a.Sh2 = 3.0 * A2 * (6.0 * A1 + B2 * (1.0 - C3));
    a.Sm4 = 18.0 * A1 * A1 + 9.0 * A1 * A2 * (1.0 - C2);
  } else {
    a.Sh2 = 3.0 * A2 * (6.0 * A1 + B2);
    a.Sm3 = A1 * (1.0 - 3.0 * C1 - 6.0 * A1 / B1);
    a.Sm4 = 18.0 * A1 * A1 + 9.0 * A1 * A2;
  }
  return a;
}

static bool tke_radiates(const roms_hip_ctx *c) {
  for (int e = 0; e < 4; e++) if (c->cfg.lbc_tke[e] == ROMS_LBC_RAD) return true;
  return false;
}

// gls_prestep_tile gls_prestep.F:95; tkebc_tile (index 3) and the exchanges :441-466
int run_gls_prestep(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  const GlsArgs a = gls_args(c);
  LAUNCH_THREAD(k_gls_pre, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, G.N - 1, c->stream, a);
  const size_t lev = (size_t)G.nij * (size_t)(G.N + 1);
  const bool rad = tke_radiates(c);                   // LBC(isMtke) radiation on an edge: tkebc through k_obc.h, then only the exchange
  if (rad) { int r = run_obc_tke(c, 3); if (r) return r; }
  const int bc = rad ? (bc_rstate(c) & ~BC_KIND) : bc_rstate(c);
  HaloSpec sp[2] = {{c->F.tke + 2 * lev, G.N + 1, bc, 'r'}, {c->F.gls + 2 * lev, G.N + 1, bc, 'r'}};
  launch_halo_tail(c, sp, 2);
  return 0;
}

// gls_corstep_tile gls_corstep.F:114; the edge copies of Akv, Akt :1196-1280, tkebc_tile (nnew), the exchanges :1282-1320
int run_gls_corstep(roms_hip_ctx *c) {
  const DGrid &G = c->G;
  const TB &B = G.T;
  const GlsArgs a = gls_args(c);
  LAUNCH_THREAD(k_gls_shear, B.Iendp1 - B.Istrm1 + 1, B.Jendp1 - B.Jstrm1 + 1, 1, c->stream, a);
  LAUNCH_THREAD(k_gls_adv, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, G.N - 1, c->stream, a);
  // the sweeps in private memory: 412 us at 512x512x50; in LDS (COL launch, three waves per CU at N = 50): 691 us -- opt-in only
  static const char *ecl = getenv("ROMS_HIP_GLSLDS");
  if (ecl && ecl[0] == '1' && (size_t)2 * (G.N + 1) * 64 * sizeof(double) < 64 * 1024)
    LAUNCH_COL(k_gls_solve_l, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 1, 2 * (G.N + 1), c->stream, a);
  else LAUNCH_THREAD(k_gls_solve, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, 1, c->stream, a);
  LAUNCH_THREAD(k_gls_coef, B.Iend - B.Istr + 1, B.Jend - B.Jstr + 1, G.N - 1, c->stream, a);
  const size_t lev = (size_t)G.nij * (size_t)(G.N + 1);
  const bool my25 = a.my25 != 0;
  if (my25) {       // my25_corstep.F:774-850: its own edge copies (the eastern one lands on Iend-1), then only the exchanges
    KArgs e;
    e.G = c->G; e.Fv = c->F; e.p0 = 0; e.p2 = 0;
    const int np = (G.N + 1) * (1 + G.NAT), nx = B.Iend - B.Istr + 1, ny = B.Jend - B.Jstr + 1;
    e.p1 = 0; LAUNCH_THREAD(k_my25_edges, ny, 1, np, c->stream, e);
    e.p1 = 1; LAUNCH_THREAD(k_my25_edges, nx, 1, np, c->stream, e);
    e.p1 = 2; LAUNCH_THREAD(k_my25_edges, 1, 1, np, c->stream, e);
  }
  const bool rad = tke_radiates(c);
  if (rad) { int r = run_obc_tke(c, G.nnew); if (r) return r; }
  const int bck = rad ? (bc_rstate(c) & ~BC_KIND) : bc_rstate(c);
  HaloSpec sp[4] = {{c->F.tke + (size_t)(G.nnew - 1) * lev, G.N + 1, bck, 'r'},
                    {c->F.gls + (size_t)(G.nnew - 1) * lev, G.N + 1, bck, 'r'},
                    {c->F.Akv, G.N + 1, my25 ? BC_NONE : BC_R, 'r'},
                    {c->F.Akt, (G.N + 1) * G.NAT, my25 ? BC_NONE : BC_R, 'r'}};
  launch_halo_tail(c, sp, 4);
  return 0;
}
