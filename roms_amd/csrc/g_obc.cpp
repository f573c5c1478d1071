// g_obc.cpp -- launches of k_obc.h: the boundary conditions of the state variables of a context with open edges.
#include "roms_host.h"
#include "k_obc.h"

// one variable's table: kinds, nudging scales, boundary data.  bvar: index of the variable in Fields::bry (zeta ubar vbar
// u v t); level0: doubles to skip in a boundary line array before plane 0 (tracer itrc: (itrc-1)*N lines)
static void obc_item(roms_hip_ctx *c, ObcItem &it, double *Qo, const double *Qn, int nk, char grid, int is2d, int lbcvar, int bvar,
                     const double *in, const double *out, long level0) {
  const roms_hip_config &cf = c->cfg;
  it.Qo = Qo; it.Qn = Qn; it.nk = nk; it.grid = grid; it.is2d = is2d;
  it.cof = nullptr; it.obcfac = cf.obcfac;
  // Fields::bry order: west, east, south, north; edges here: ROMS_IWEST, ROMS_ISOUTH, ROMS_IEAST, ROMS_INORTH
  static const int slot[4] = {0, 2, 1, 3};
  for (int e = 0; e < 4; e++) {
    it.kind[e] = lbc_kind(cf, e, lbcvar);
    it.obc_in[e] = in[e]; it.obc_out[e] = out[e];
    const long nb = (e == ROMS_IWEST || e == ROMS_IEAST) ? c->cnj : c->cni;
    it.bry[e] = (const double *)c->F.bry[4 * bvar + slot[e]] + level0 * nb;
    it.bstride[e] = is2d ? 0 : nb;
  }
}

static void obc_common(roms_hip_ctx *c, ObcArgs &a) {
  a.G = c->G;
  for (int q = 0; q < OBC_MAXITEMS; q++) { a.it[q].Qo = nullptr; a.it[q].Qn = nullptr; a.it[q].nk = 0; a.it[q].grid = 'r'; a.it[q].cof = nullptr; a.it[q].obcfac = 0.0; }
  a.h = c->F.h; a.pm = c->F.pm; a.pn = c->F.pn;
  static const int slot[4] = {0, 2, 1, 3};
  for (int e = 0; e < 4; e++) a.zbry[e] = c->F.bry[slot[e]];
  a.zeta_n = a.zeta_o = c->F.zeta;
}

// zetabc_tile, u2dbc_tile, v2dbc_tile of level kout (vars: bit 0 zeta, 1 ubar, 2 vbar).  "Now" level and time step as the
// LF-AM3 kernel's conditions take them (zetabc.F:100-112): the first fast step and every predictor start from krhs, the
// predictor over 2*dtfast; a corrector from kstp.
int run_obc2d(roms_hip_ctx *c, int kout, unsigned vars) {
  const DGrid &G = c->G;
  const roms_hip_config &cf = c->cfg;
  ObcArgs a;
  obc_common(c, a);
  int know;
  if (G.iif == 1) { know = G.krhs; a.dtn = G.dtfast; }
  else if (G.predictor) { know = G.krhs; a.dtn = 2.0 * G.dtfast; }
  else { know = G.kstp; a.dtn = G.dtfast; }
  a.zeta_n = lev2d(c, c->F.zeta, know);
  a.zeta_o = lev2d(c, c->F.zeta, kout);
  int n = 0;
  if (vars & 1) obc_item(c, a.it[n++], lev2d(c, c->F.zeta, kout), lev2d(c, c->F.zeta, know), 1, 'r', 1, ROMS_ISFSUR, 0, cf.FSobc_in, cf.FSobc_out, 0);
  if (vars & 2) { obc_item(c, a.it[n], lev2d(c, c->F.ubar, kout), lev2d(c, c->F.ubar, know), 1, 'u', 1, ROMS_ISUBAR, 1, cf.M2obc_in, cf.M2obc_out, 0);
                  if (G.clima & 32) a.it[n].cof = c->F.M2nudgcof; n++; }       // LnudgeM2CLM: u2dbc_im.F:158
  if (vars & 4) { obc_item(c, a.it[n], lev2d(c, c->F.vbar, kout), lev2d(c, c->F.vbar, know), 1, 'v', 1, ROMS_ISVBAR, 2, cf.M2obc_in, cf.M2obc_out, 0);
                  if (G.clima & 32) a.it[n].cof = c->F.M2nudgcof; n++; }
  a.nitems = n;
  LAUNCH_COOP(k_obc, 1, 1, n, 256, 0, c->stream, a);
  return 0;
}

// obc_flux_tile of level kinp (VolCons)
int run_obc_flux(roms_hip_ctx *c, int kinp) {
  const DGrid &G = c->G;
  ObcFluxArgs a;
  a.G = G;
  a.zeta = lev2d(c, c->F.zeta, kinp); a.ubar = lev2d(c, c->F.ubar, kinp); a.vbar = lev2d(c, c->F.vbar, kinp);
  a.h = c->F.h; a.on_u = c->F.on_u; a.om_v = c->F.om_v;
  LAUNCH_COOP(k_obc_flux, 1, 1, 1, 256, 2 * OBCF_CHUNK, c->stream, a);
  return 0;
}

// u3dbc_tile, v3dbc_tile: levels nstp ("now") and nout
int run_obc3d_uv(roms_hip_ctx *c, int nout) {
  const DGrid &G = c->G;
  const roms_hip_config &cf = c->cfg;
  ObcArgs a;
  obc_common(c, a);
  a.dtn = G.dt;
  obc_item(c, a.it[0], uv_lev(c, c->F.u, nout), uv_lev(c, c->F.u, G.nstp), G.N, 'u', 0, ROMS_ISUVEL, 3, cf.M3obc_in, cf.M3obc_out, 0);
  obc_item(c, a.it[1], uv_lev(c, c->F.v, nout), uv_lev(c, c->F.v, G.nstp), G.N, 'v', 0, ROMS_ISVVEL, 4, cf.M3obc_in, cf.M3obc_out, 0);
  if (G.clima & 1) a.it[0].cof = a.it[1].cof = c->F.M3nudgcof;                 // LnudgeM3CLM: u3dbc_im.F:113, v3dbc_im.F:113
  a.nitems = 2;
  LAUNCH_COOP(k_obc, 1, 1, 2 * G.N, 256, 0, c->stream, a);
  return 0;
}

// t3dbc_tile of tracer itrc
int run_obc3d_t(roms_hip_ctx *c, int nout, int itrc) {
  const DGrid &G = c->G;
  const roms_hip_config &cf = c->cfg;
  ObcArgs a;
  obc_common(c, a);
  a.dtn = G.dt;
  obc_item(c, a.it[0], t_lev(c, nout, itrc), t_lev(c, G.nstp, itrc), G.N, 'r', 0, ROMS_ISTVAR + itrc - 1, 5, cf.Tobc_in[itrc - 1], cf.Tobc_out[itrc - 1],
           (long)(itrc - 1) * G.N);
  if (G.clima & (1 << itrc)) a.it[0].cof = (const double *)c->F.Tnudgcof + (size_t)(itrc - 1) * (size_t)G.N * (size_t)G.nij;      // LnudgeTCLM: t3dbc_im.F:120
  a.nitems = 1;
  LAUNCH_COOP(k_obc, 1, 1, G.N, 256, 0, c->stream, a);
  return 0;
}

// tkebc_tile (tkebc_im.F:46-700) of level nout when an edge radiates: tke and gls, W-points 0..N, kinds of LBC(isMtke)
// (radiation: the scheme of t3dbc without nudging, differences of level nstp; zero gradient at gradient and closed edges)
int run_obc_tke(roms_hip_ctx *c, int nout) {
  const DGrid &G = c->G;
  const roms_hip_config &cf = c->cfg;
  ObcArgs a;
  obc_common(c, a);
  a.dtn = G.dt;
  const size_t lev = (size_t)G.nij * (size_t)(G.N + 1);
  static const double zero[4] = {0.0, 0.0, 0.0, 0.0};
  for (int q = 0; q < 2; q++) {
    double *A = q == 0 ? (double *)c->F.tke : (double *)c->F.gls;
    obc_item(c, a.it[q], A + (size_t)(nout - 1) * lev, A + (size_t)(G.nstp - 1) * lev, G.N + 1, 'r', 0, ROMS_ISTVAR, 5, zero, zero, 0);
    for (int e = 0; e < 4; e++) {
      a.it[q].kind[e] = cf.lbc_tke[e] == ROMS_LBC_RAD ? ROMS_LBC_RAD : ROMS_LBC_GRA;
      a.it[q].bry[e] = nullptr; a.it[q].bstride[e] = 0;
    }
  }
  a.nitems = 2;
  LAUNCH_COOP(k_obc, 1, 1, 2 * (G.N + 1), 256, 0, c->stream, a);
  return 0;
}
