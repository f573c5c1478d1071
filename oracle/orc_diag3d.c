/*
 * orc_diag3d.c -- diagnostic-type 3-D kernels of the time step.
 * TEST INFRASTRUCTURE (see orc.h).
 *
 *   orc_set_depth     set_depth_tile     ROMS/Nonlinear/set_depth.F:76-278   pinned
 *   orc_set_massflux  set_massflux_tile  ROMS/Nonlinear/set_massflux.F:73-188 pinned
 *   orc_rho_eos       rho_eos_tile       ROMS/Nonlinear/rho_eos.F:111 (linear :688-880;
 *                                        nonlinear :247-560 in orc_eos.c)   pinned
 *   orc_set_vbc       set_vbc_tile       ROMS/Nonlinear/set_vbc.F:110       pinned
 *   orc_wvelocity     wvelocity_tile     ROMS/Nonlinear/wvelocity.F:64-289  pinned
 *   orc_set_zeta      set_zeta_tile      ROMS/Nonlinear/set_zeta.F:59-118   pinned
 *   orc_ini_zeta      ini_zeta_tile + set_zeta_timeavg_tile
 *                                        ROMS/Nonlinear/ini_fields.F:754,1017 pinned
 *   orc_ini_fields    ini_fields_tile    ROMS/Nonlinear/ini_fields.F:136    pinned
 *   orc_ana_vmix      ana_vmix_tile      ROMS/Functionals/ana_vmix.h (UPWELLING :200-207,
 *                                        :327-337)                          pinned
 *   orc_set_data      set_data_tile      ROMS/Nonlinear/set_data.F:255-564 ->
 *                                        ana_smflux.h:306-318, ana_stflux.h, ana_btflux.h pinned
 *   orc_omega         omega_tile         ROMS/Nonlinear/omega.F:96-377      pinned (round 2)
 *                                        (omega.F USEs mod_sources -> NetCDF)
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>

void orc_eos_nonlinear(orc_t *o, int tile);   /* orc_eos.c */
void orc_set_data_benchmark(orc_t *o, int tile); /* orc_bulk.c */

/* ------------------------------------------------------------- set_depth */
void orc_set_depth(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const double hc = o->c.hc;
  double *h = o->h, *Zt_avg1 = o->Zt_avg1, *Hz = o->Hz, *z_r = o->z_r, *z_w = o->z_w;
  if (o->c.Vtransform == 1) {
    for (int j = b->JstrT; j <= b->JendT; j++) {
      for (int i = b->IstrT; i <= b->IendT; i++) {
        if (o->wet_dry && h[X2(i, j)] == 0.0) h[X2(i, j)] = 1.0E-14;      /* set_depth.F:150-154,195-199 */
        z_w[XW(i, j, 0)] = -h[X2(i, j)];
      }
      for (int k = 1; k <= N; k++) {
        double cff_r = hc * (o->sc_r[k - 1] - o->Cs_r[k - 1]);
        double cff_w = hc * (o->sc_w[k] - o->Cs_w[k]);
        double cff1_r = o->Cs_r[k - 1], cff1_w = o->Cs_w[k];
        for (int i = b->IstrT; i <= b->IendT; i++) {
          double hwater = h[X2(i, j)];
          double hinv = 1.0 / hwater;
          double z_w0 = cff_w + cff1_w * hwater;
          z_w[XW(i, j, k)] = z_w0 + Zt_avg1[X2(i, j)] * (1.0 + z_w0 * hinv);
          double z_r0 = cff_r + cff1_r * hwater;
          z_r[X3(i, j, k)] = z_r0 + Zt_avg1[X2(i, j)] * (1.0 + z_r0 * hinv);
          Hz[X3(i, j, k)] = z_w[XW(i, j, k)] - z_w[XW(i, j, k - 1)];
        }
      }
    }
  } else {
    for (int j = b->JstrT; j <= b->JendT; j++) {
      for (int i = b->IstrT; i <= b->IendT; i++) {
        if (o->wet_dry && h[X2(i, j)] == 0.0) h[X2(i, j)] = 1.0E-14;      /* set_depth.F:150-154,195-199 */
        z_w[XW(i, j, 0)] = -h[X2(i, j)];
      }
      for (int k = 1; k <= N; k++) {
        double cff_r = hc * o->sc_r[k - 1];
        double cff_w = hc * o->sc_w[k];
        double cff1_r = o->Cs_r[k - 1], cff1_w = o->Cs_w[k];
        for (int i = b->IstrT; i <= b->IendT; i++) {
          double hwater = h[X2(i, j)];
          double hinv = 1.0 / (hc + hwater);
          double cff2_r = (cff_r + cff1_r * hwater) * hinv;
          double cff2_w = (cff_w + cff1_w * hwater) * hinv;
          z_w[XW(i, j, k)] = Zt_avg1[X2(i, j)] + (Zt_avg1[X2(i, j)] + hwater) * cff2_w;
          z_r[X3(i, j, k)] = Zt_avg1[X2(i, j)] + (Zt_avg1[X2(i, j)] + hwater) * cff2_r;
          Hz[X3(i, j, k)] = z_w[XW(i, j, k)] - z_w[XW(i, j, k - 1)];
        }
      }
    }
  }
  orc_exchange2d(o, b, 'r', h);
  orc_exchange3d(o, b, 'w', z_w, N + 1);
  orc_exchange3d(o, b, 'r', z_r, N);
  orc_exchange3d(o, b, 'r', Hz, N);
}

/* ---------------------------------------------------------- set_massflux */
void orc_set_massflux(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs;
  double *Hz = o->Hz, *u = o->u, *v = o->v, *Huon = o->Huon, *Hvom = o->Hvom;
  for (int k = 1; k <= N; k++) {
    for (int j = b->JstrT; j <= b->JendT; j++)
      for (int i = b->IstrP; i <= b->IendT; i++)
        Huon[X3(i, j, k)] = 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i - 1, j, k)]) * u[X4(i, j, k, nrhs)] *
                            o->on_u[X2(i, j)];
    for (int j = b->JstrP; j <= b->JendT; j++)
      for (int i = b->IstrT; i <= b->IendT; i++)
        Hvom[X3(i, j, k)] = 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i, j - 1, k)]) * v[X4(i, j, k, nrhs)] *
                            o->om_v[X2(i, j)];
  }
  orc_exchange3d(o, b, 'u', Huon, N);
  orc_exchange3d(o, b, 'v', Hvom, N);
}

/* --------------------------------------------------------------- rho_eos */
void orc_rho_eos(orc_t *o, int tile) {
  if (o->c.options & ORC_NONLIN_EOS) { orc_eos_nonlinear(o, tile); return; }
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs;
  const double R0 = o->c.R0, Tcoef = o->c.Tcoef, T0 = o->c.T0, Scoef = o->c.Scoef, S0 = o->c.S0;
  double *t = o->t, *rho = o->rho, *pden = o->pden, *Hz = o->Hz, *z_w = o->z_w;
  double *rhoA = o->rhoA, *rhoS = o->rhoS;
  for (int j = b->JstrT; j <= b->JendT; j++) {
    for (int k = 1; k <= N; k++)
      for (int i = b->IstrT; i <= b->IendT; i++) {
        double r = R0 - R0 * Tcoef * (t[XT(i, j, k, nrhs, 1)] - T0);
        if (o->c.options & ORC_SALINITY) r = r + R0 * Scoef * (t[XT(i, j, k, nrhs, 2)] - S0);
        r = r - 1000.0;
        if (o->c.options & ORC_MASKING) r = r * o->rmask[X2(i, j)];                         /* rho_eos.F:718 */
        rho[X3(i, j, k)] = r;
        pden[X3(i, j, k)] = r;
      }
    for (int i = b->IstrT; i <= b->IendT; i++) {
      double cff1 = rho[X3(i, j, N)] * Hz[X3(i, j, N)];
      rhoS[X2(i, j)] = 0.5 * cff1 * Hz[X3(i, j, N)];
      rhoA[X2(i, j)] = cff1;
    }
    for (int k = N - 1; k >= 1; k--)
      for (int i = b->IstrT; i <= b->IendT; i++) {
        double cff1 = rho[X3(i, j, k)] * Hz[X3(i, j, k)];
        rhoS[X2(i, j)] = rhoS[X2(i, j)] + Hz[X3(i, j, k)] * (rhoA[X2(i, j)] + 0.5 * cff1);
        rhoA[X2(i, j)] = rhoA[X2(i, j)] + cff1;
      }
    double cff2 = 1.0 / o->c.rho0;
    for (int i = b->IstrT; i <= b->IendT; i++) {
      double cff1 = 1.0 / (z_w[XW(i, j, N)] - z_w[XW(i, j, 0)]);
      rhoA[X2(i, j)] = cff2 * cff1 * rhoA[X2(i, j)];
      rhoS[X2(i, j)] = 2.0 * cff1 * cff1 * cff2 * rhoS[X2(i, j)];
    }
    if (o->c.options & (ORC_LMD_MIXING | ORC_GLS_MIXING | ORC_MY25_MIXING)) {
      /* BV_FREQUENCY (LMD_MIXING, GLS_MIXING: globaldefs.h) :751-764 of the linear EOS */
      const double gorho0 = o->c.g / o->c.rho0;
      for (int k = 1; k <= N - 1; k++)
        for (int i = b->IstrT; i <= b->IendT; i++)
          o->bvf[XW(i, j, k)] = -gorho0 * (rho[X3(i, j, k + 1)] - rho[X3(i, j, k)]) /
                                (o->z_r[X3(i, j, k + 1)] - o->z_r[X3(i, j, k)]);
    }
    if (o->c.options & ORC_LMD_MIXING) {
      /* LMD_SKPP expansion coefficients :766-780 */
      for (int i = b->IstrT; i <= b->IendT; i++) {
        o->alpha[X2(i, j)] = fabs(Tcoef);
        o->beta[X2(i, j)] = (o->c.options & ORC_SALINITY) ? fabs(Scoef) : 0.0;
      }
      if (o->ddmix) {                     /* LMD_DDMIX :782-796 */
        const double cff = Scoef == 0.0 ? 1.0 : 1.0 / Scoef;
        for (int k = 1; k <= N; k++)
          for (int i = b->IstrT; i <= b->IendT; i++) o->alfaobeta[XW(i, j, k)] = cff * Tcoef;
      }
    }
  }
  orc_exchange3d(o, b, 'r', rho, N);
  orc_exchange3d(o, b, 'r', pden, N);
  if (o->c.options & ORC_LMD_MIXING) {
    orc_exchange2d(o, b, 'r', o->alpha);
    orc_exchange2d(o, b, 'r', o->beta);
  }
  orc_exchange2d(o, b, 'r', rhoA);
  orc_exchange2d(o, b, 'r', rhoS);
  if (o->c.options & (ORC_LMD_MIXING | ORC_GLS_MIXING | ORC_MY25_MIXING)) orc_exchange3d(o, b, 'w', o->bvf, N + 1);
  if (o->ddmix && (o->c.options & ORC_LMD_MIXING)) orc_exchange3d(o, b, 'w', o->alfaobeta, N + 1);     /* :815-818 */
}

/* --------------------------------------------------------------- set_vbc */
void orc_set_vbc(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nrhs = o->s.nrhs;
  double *u = o->u, *v = o->v, *t = o->t;
  /* set_vbc.F: load kinematic surface/bottom tracer fluxes (same code with or
     without BULK_FLUXES: bulk_flux fills stflux) */
  for (int j = b->JstrR; j <= b->JendR; j++)
    for (int i = b->IstrR; i <= b->IendR; i++) {
      o->stflx[X2T(i, j, 1)] = o->stflux[X2T(i, j, 1)];
      o->btflx[X2T(i, j, 1)] = o->btflux[X2T(i, j, 1)];
      if (o->wet_dry) {                                                   /* set_vbc.F:307-308 */
        o->stflx[X2T(i, j, 1)] = o->stflx[X2T(i, j, 1)] * o->rmask_wet[X2(i, j)];
        o->btflx[X2T(i, j, 1)] = o->btflx[X2T(i, j, 1)] * o->rmask_wet[X2(i, j)];
      }
    }
  for (int j = b->JstrR; j <= b->JendR; j++)
    for (int i = b->IstrR; i <= b->IendR; i++) {
      double EmP = o->stflux[X2T(i, j, 2)];
      o->stflx[X2T(i, j, 2)] = EmP * t[XT(i, j, N, nrhs, 2)];
      if (o->wet_dry) o->stflx[X2T(i, j, 2)] = o->rmask_wet[X2(i, j)] * o->stflx[X2T(i, j, 2)];                /* set_vbc.F:397 */
      else if (o->c.options & ORC_MASKING) o->stflx[X2T(i, j, 2)] = o->rmask[X2(i, j)] * o->stflx[X2T(i, j, 2)];   /* set_vbc.F:399 */
      o->btflx[X2T(i, j, 2)] = o->btflx[X2T(i, j, 2)] * t[XT(i, j, 1, nrhs, 2)];
    }
  if (o->c.options & ORC_UV_LOGDRAG) {
    /* logarithmic bottom stress set_vbc.F:591-635; ZoBot = Zob (mod_grid.F:1380), Cdb_min, Cdb_max, vonKar of
       mod_scalars.F:469,772-773 */
    const double vonKar = 0.41, Cdb_min = 0.000001, Cdb_max = 0.5;
    double *wrk = (double *)calloc(nij, sizeof(double));
    for (int j = b->JstrV - 1; j <= b->Jend; j++)
      for (int i = b->IstrU - 1; i <= b->Iend; i++) {
        const double cff1 = 1.0 / log((o->z_r[X3(i, j, 1)] - o->z_w[XW(i, j, 0)]) / o->c.Zob);
        const double cff2 = vonKar * vonKar * cff1 * cff1;
        wrk[X2(i, j)] = fmin(Cdb_max, fmax(Cdb_min, cff2));
      }
    for (int j = b->Jstr; j <= b->Jend; j++)
      for (int i = b->IstrU; i <= b->Iend; i++) {
        const double cff1 = 0.25 * (v[X4(i, j, 1, nrhs)] + v[X4(i, j + 1, 1, nrhs)] +
                                    v[X4(i - 1, j, 1, nrhs)] + v[X4(i - 1, j + 1, 1, nrhs)]);
        const double cff2 = sqrt(u[X4(i, j, 1, nrhs)] * u[X4(i, j, 1, nrhs)] + cff1 * cff1);
        o->bustr[X2(i, j)] = 0.5 * (wrk[X2(i - 1, j)] + wrk[X2(i, j)]) * u[X4(i, j, 1, nrhs)] * cff2;
      }
    for (int j = b->JstrV; j <= b->Jend; j++)
      for (int i = b->Istr; i <= b->Iend; i++) {
        const double cff1 = 0.25 * (u[X4(i, j, 1, nrhs)] + u[X4(i + 1, j, 1, nrhs)] +
                                    u[X4(i, j - 1, 1, nrhs)] + u[X4(i + 1, j - 1, 1, nrhs)]);
        const double cff2 = sqrt(cff1 * cff1 + v[X4(i, j, 1, nrhs)] * v[X4(i, j, 1, nrhs)]);
        o->bvstr[X2(i, j)] = 0.5 * (wrk[X2(i, j - 1)] + wrk[X2(i, j)]) * v[X4(i, j, 1, nrhs)] * cff2;
      }
    free(wrk);
  } else if (o->c.options & ORC_UV_QDRAG) {
    /* quadratic bottom drag set_vbc.F:178 */
    for (int j = b->Jstr; j <= b->Jend; j++)
      for (int i = b->IstrU; i <= b->Iend; i++) {
        double cff1 = 0.25 * (v[X4(i, j, 1, nrhs)] + v[X4(i, j + 1, 1, nrhs)] +
                              v[X4(i - 1, j, 1, nrhs)] + v[X4(i - 1, j + 1, 1, nrhs)]);
        double cff2 = sqrt(u[X4(i, j, 1, nrhs)] * u[X4(i, j, 1, nrhs)] + cff1 * cff1);
        o->bustr[X2(i, j)] = 0.5 * (o->rdrag2[X2(i - 1, j)] + o->rdrag2[X2(i, j)]) *
                             u[X4(i, j, 1, nrhs)] * cff2;
      }
    for (int j = b->JstrV; j <= b->Jend; j++)
      for (int i = b->Istr; i <= b->Iend; i++) {
        double cff1 = 0.25 * (u[X4(i, j, 1, nrhs)] + u[X4(i + 1, j, 1, nrhs)] +
                              u[X4(i, j - 1, 1, nrhs)] + u[X4(i + 1, j - 1, 1, nrhs)]);
        double cff2 = sqrt(cff1 * cff1 + v[X4(i, j, 1, nrhs)] * v[X4(i, j, 1, nrhs)]);
        o->bvstr[X2(i, j)] = 0.5 * (o->rdrag2[X2(i, j - 1)] + o->rdrag2[X2(i, j)]) *
                             v[X4(i, j, 1, nrhs)] * cff2;
      }
  } else {
    for (int j = b->Jstr; j <= b->Jend; j++)
      for (int i = b->IstrU; i <= b->Iend; i++)
        o->bustr[X2(i, j)] = 0.5 * (o->rdrag[X2(i - 1, j)] + o->rdrag[X2(i, j)]) *
                             u[X4(i, j, 1, nrhs)];
    for (int j = b->JstrV; j <= b->Jend; j++)
      for (int i = b->Istr; i <= b->Iend; i++)
        o->bvstr[X2(i, j)] = 0.5 * (o->rdrag[X2(i, j - 1)] + o->rdrag[X2(i, j)]) *
                             v[X4(i, j, 1, nrhs)];
  }
  if (o->wet_dry) {
    /* LIMIT_BSTRESS (globaldefs.h:160 switches it on with WET_DRY), set_vbc.F:580-590 and :611-616, :649-654, :682-687:
       the bottom stress may slow the bottom layer down within 0.75 of a step but not reverse it */
    const double cff = 0.75 / o->c.dt;
    for (int j = b->Jstr; j <= b->Jend; j++)
      for (int i = b->IstrU; i <= b->Iend; i++) {
        const double cff3 = cff * 0.5 * (o->Hz[X3(i - 1, j, 1)] + o->Hz[X3(i, j, 1)]);
        o->bustr[X2(i, j)] = copysign(1.0, o->bustr[X2(i, j)]) * fmin(fabs(o->bustr[X2(i, j)]), fabs(u[X4(i, j, 1, nrhs)]) * cff3);
      }
    for (int j = b->JstrV; j <= b->Jend; j++)
      for (int i = b->Istr; i <= b->Iend; i++) {
        const double cff3 = cff * 0.5 * (o->Hz[X3(i, j - 1, 1)] + o->Hz[X3(i, j, 1)]);
        o->bvstr[X2(i, j)] = copysign(1.0, o->bvstr[X2(i, j)]) * fmin(fabs(o->bvstr[X2(i, j)]), fabs(v[X4(i, j, 1, nrhs)]) * cff3);
      }
  }
  orc_bc_u2d(o, b, o->bustr);
  orc_bc_v2d(o, b, o->bvstr);
}

/* -------------------------------------------------------------- ana_vmix */
void orc_ana_vmix(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  for (int k = 1; k <= N - 1; k++)
    for (int j = b->JstrT; j <= b->JendT; j++)
      for (int i = b->IstrT; i <= b->IendT; i++)
        o->Akv[XW(i, j, k)] = 2.0E-03 + 8.0E-03 * exp(o->z_w[XW(i, j, k)] / 150.0);
  orc_exchange3d(o, b, 'w', o->Akv, N + 1);
  for (int k = 1; k <= N - 1; k++)
    for (int j = b->JstrT; j <= b->JendT; j++)
      for (int i = b->IstrT; i <= b->IendT; i++) {
        o->Akt[XW4(i, j, k, 1)] = o->c.Akt_bak[0];
        o->Akt[XW4(i, j, k, 2)] = o->c.Akt_bak[1];
      }
  for (int it = 0; it < o->c.NAT; it++)
    orc_exchange3d(o, b, 'w', o->Akt + (size_t)it * nij * (N + 1), N + 1);
}

/* -------------------------------------------------------------- set_data */
void orc_set_data(orc_t *o, int tile) {
  if (o->c.options & ORC_APP_BENCHMARK) { orc_set_data_benchmark(o, tile); return; }
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const double pi = 3.14159265358979323846;
  /* set_data.F: ana_stflux(itemp), ana_btflux(itemp), ana_stflux(isalt),
     ana_btflux(isalt), ana_smflux -- UPWELLING branches */
  for (int it = 1; it <= 2; it++) {
    for (int j = b->JstrT; j <= b->JendT; j++)
      for (int i = b->IstrT; i <= b->IendT; i++) o->stflux[X2T(i, j, it)] = 0.0;
    orc_exchange2d(o, b, 'r', o->stflux + (size_t)(it - 1) * nij);
    for (int j = b->JstrT; j <= b->JendT; j++)
      for (int i = b->IstrT; i <= b->IendT; i++) o->btflux[X2T(i, j, it)] = 0.0;
  }
  if (o->c.options & ORC_SOLAR_SOURCE) {                 /* ana_srflux.h:270-277 (UPWELLING branch) */
    const double cff = 1.0 / (o->c.rho0 * o->c.Cp);
    for (int j = b->JstrT; j <= b->JendT; j++)
      for (int i = b->IstrT; i <= b->IendT; i++) o->srflx[X2(i, j)] = cff * 150.0;
    orc_exchange2d(o, b, 'r', o->srflx);
  }
  double windamp;
  if (!(o->c.options & ORC_APP_UPWELLING)) windamp = 0.0;      /* KELVIN: the default branch of ana_smflux.h, no wind */
  else if ((o->s.tdays - o->c.dstart) <= 2.0)
    windamp = -0.1 * sin(pi * (o->s.tdays - o->c.dstart) / 4.0) / o->c.rho0;
  else
    windamp = -0.1 / o->c.rho0;
  if (o->c.NSperiodic) {
    for (int j = b->JstrT; j <= b->JendT; j++)
      for (int i = b->IstrP; i <= b->IendT; i++) o->sustr[X2(i, j)] = 0.0;
    for (int j = b->JstrP; j <= b->JendT; j++)
      for (int i = b->IstrT; i <= b->IendT; i++) o->svstr[X2(i, j)] = windamp;
  } else if (o->c.EWperiodic) {
    for (int j = b->JstrT; j <= b->JendT; j++)
      for (int i = b->IstrP; i <= b->IendT; i++) o->sustr[X2(i, j)] = windamp;
    for (int j = b->JstrP; j <= b->JendT; j++)
      for (int i = b->IstrT; i <= b->IendT; i++) o->svstr[X2(i, j)] = 0.0;
  }
  orc_exchange2d(o, b, 'u', o->sustr);
  orc_exchange2d(o, b, 'v', o->svstr);
  if (o->c.options & ORC_APP_KELVIN) {
    /* set_data.F:881,1003: ana_fsobc.h:85-105 and ana_m2obc.h:169-200, KELVIN branches -- an M2 Kelvin wave of unit
       amplitude at the western edge, its image after one channel length at the eastern one (the eastern formulas index
       f, h and yp with this tile's Istr-1 / Iend exactly as the reference does) */
    const double g = o->c.g, time = o->s.time;
    const double fac = 1.0, omega = 2.0 * pi / (12.42 * 3600.0);
    const int Istr = b->Istr, Iend = b->Iend;
    if (orc_lbc_acquire(o, ORC_IWEST, ORC_ISFSUR) && b->west)
      for (int j = b->JstrT; j <= b->JendT; j++) {
        const double val = fac * exp(-o->f[X2(Istr - 1, j)] * o->yp[X2(Istr - 1, j)] / sqrt(g * o->h[X2(Istr - 1, j)]));
        o->zeta_west[j - LBj] = val * cos(omega * time);
      }
    if (orc_lbc_acquire(o, ORC_IEAST, ORC_ISFSUR) && b->east)
      for (int j = b->JstrT; j <= b->JendT; j++) {
        const double cff = 1.0 / sqrt(g * o->h[X2(Istr - 1, j)]);
        const double val = fac * exp(-o->f[X2(Istr - 1, j)] * o->yp[X2(Iend, j)] * cff);
        o->zeta_east[j - LBj] = val * cos(omega * o->xp[X2(Iend, j)] * cff - omega * time);
      }
    const double val0 = fac * sin(omega * time);
    if (orc_lbc_acquire(o, ORC_IWEST, ORC_ISUBAR) && orc_lbc_acquire(o, ORC_IWEST, ORC_ISVBAR) && b->west) {
      for (int j = b->JstrT; j <= b->JendT; j++) {
        const double cff = sqrt(g * o->h[X2(Istr - 1, j)]);
        o->ubar_west[j - LBj] = (val0 * cff / o->h[X2(Istr - 1, j)]) * exp(-o->f[X2(Istr - 1, j)] * o->yp[X2(Istr - 1, j)] / cff);
      }
      for (int j = b->JstrP; j <= b->JendT; j++) o->vbar_west[j - LBj] = 0.0;
    }
    if (orc_lbc_acquire(o, ORC_IEAST, ORC_ISUBAR) && orc_lbc_acquire(o, ORC_IEAST, ORC_ISVBAR) && b->east) {
      for (int j = b->JstrT; j <= b->JendT; j++) {
        const double cff = sqrt(g * o->h[X2(Iend, j)]);
        const double val = fac * exp(-o->f[X2(Iend, j)] * o->yp[X2(Istr - 1, j)] / cff);
        o->ubar_east[j - LBj] = (val * cff / o->h[X2(Iend, j)]) * sin(omega * o->xp[X2(Iend, j)] / cff - omega * time);
      }
      for (int j = b->JstrP; j <= b->JendT; j++) o->vbar_east[j - LBj] = 0.0;
    }
  }
}

/* ----------------------------------------------------------------- omega */
void orc_omega(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  double *W = o->W, *Huon = o->Huon, *Hvom = o->Hvom, *z_w = o->z_w;
  double *wrk = (double *)malloc(sizeof(double) * o->ni);
  for (int j = b->Jstr; j <= b->Jend; j++) {
    for (int i = b->Istr; i <= b->Iend; i++) W[XW(i, j, 0)] = 0.0;
    for (int k = 1; k <= N; k++)
      for (int i = b->Istr; i <= b->Iend; i++)
        W[XW(i, j, k)] = W[XW(i, j, k - 1)] - (Huon[X3(i + 1, j, k)] - Huon[X3(i, j, k)] +
                                               Hvom[X3(i, j + 1, k)] - Hvom[X3(i, j, k)]);
    for (int i = b->Istr; i <= b->Iend; i++)
      wrk[i - LBi] = W[XW(i, j, N)] / (z_w[XW(i, j, N)] - z_w[XW(i, j, 0)]);
    for (int k = N - 1; k >= 1; k--)
      for (int i = b->Istr; i <= b->Iend; i++)
        W[XW(i, j, k)] = W[XW(i, j, k)] - wrk[i - LBi] * (z_w[XW(i, j, k)] - z_w[XW(i, j, 0)]);
    for (int i = b->Istr; i <= b->Iend; i++) W[XW(i, j, N)] = 0.0;
  }
  free(wrk);
  orc_bc_w3d(o, b, W, N + 1);
}

/* ------------------------------------------------------------- wvelocity */
void orc_wvelocity(orc_t *o, int tile, int Ninp) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int Istr = b->Istr, Iend = b->Iend, Jstr = b->Jstr, Jend = b->Jend;
  double *u = o->u, *v = o->v, *z_r = o->z_r, *z_w = o->z_w, *pm = o->pm, *pn = o->pn;
  double *W = o->W, *wvel = o->wvel, *DU_avg1 = o->DU_avg1, *DV_avg1 = o->DV_avg1;
  double *vert = (double *)calloc(nij * (size_t)N, sizeof(double));
  double *wrk = (double *)calloc(nij, sizeof(double));
  orc_exchange2d(o, b, 'u', DU_avg1);
  orc_exchange2d(o, b, 'v', DV_avg1);
  for (int k = 1; k <= N; k++) {
    for (int j = Jstr; j <= Jend; j++) {
      for (int i = Istr; i <= Iend + 1; i++)
        wrk[X2(i, j)] = u[X4(i, j, k, Ninp)] * (z_r[X3(i, j, k)] - z_r[X3(i - 1, j, k)]) *
                        (pm[X2(i - 1, j)] + pm[X2(i, j)]);
      for (int i = Istr; i <= Iend; i++)
        vert[X3(i, j, k)] = 0.25 * (wrk[X2(i, j)] + wrk[X2(i + 1, j)]);
    }
    for (int j = Jstr; j <= Jend + 1; j++)
      for (int i = Istr; i <= Iend; i++)
        wrk[X2(i, j)] = v[X4(i, j, k, Ninp)] * (z_r[X3(i, j, k)] - z_r[X3(i, j - 1, k)]) *
                        (pn[X2(i, j - 1)] + pn[X2(i, j)]);
    for (int j = Jstr; j <= Jend; j++)
      for (int i = Istr; i <= Iend; i++)
        vert[X3(i, j, k)] = vert[X3(i, j, k)] + 0.25 * (wrk[X2(i, j)] + wrk[X2(i, j + 1)]);
  }
  const double cff1 = 3.0 / 8.0, cff2 = 3.0 / 4.0, cff3 = 1.0 / 8.0, cff4 = 9.0 / 16.0,
               cff5 = 1.0 / 16.0;
  for (int j = Jstr; j <= Jend; j++) {
    for (int i = Istr; i <= Iend; i++)
      wrk[X2(i, j)] = (DU_avg1[X2(i, j)] - DU_avg1[X2(i + 1, j)] + DV_avg1[X2(i, j)] -
                       DV_avg1[X2(i, j + 1)]) /
                      (z_w[XW(i, j, N)] - z_w[XW(i, j, 0)]);
    for (int i = Istr; i <= Iend; i++) {
      double slope = (z_r[X3(i, j, 1)] - z_w[XW(i, j, 0)]) / (z_r[X3(i, j, 2)] - z_r[X3(i, j, 1)]);
      wvel[XW(i, j, 0)] = cff1 * (vert[X3(i, j, 1)] - slope * (vert[X3(i, j, 2)] - vert[X3(i, j, 1)])) +
                          cff2 * vert[X3(i, j, 1)] - cff3 * vert[X3(i, j, 2)];
      wvel[XW(i, j, 1)] = pm[X2(i, j)] * pn[X2(i, j)] *
                              (W[XW(i, j, 1)] + wrk[X2(i, j)] * (z_w[XW(i, j, 1)] - z_w[XW(i, j, 0)])) +
                          cff1 * vert[X3(i, j, 1)] + cff2 * vert[X3(i, j, 2)] - cff3 * vert[X3(i, j, 3)];
    }
    for (int k = 2; k <= N - 2; k++)
      for (int i = Istr; i <= Iend; i++)
        wvel[XW(i, j, k)] = pm[X2(i, j)] * pn[X2(i, j)] *
                                (W[XW(i, j, k)] + wrk[X2(i, j)] * (z_w[XW(i, j, k)] - z_w[XW(i, j, 0)])) +
                            cff4 * (vert[X3(i, j, k)] + vert[X3(i, j, k + 1)]) -
                            cff5 * (vert[X3(i, j, k - 1)] + vert[X3(i, j, k + 2)]);
    for (int i = Istr; i <= Iend; i++) {
      double slope = (z_w[XW(i, j, N)] - z_r[X3(i, j, N)]) / (z_r[X3(i, j, N)] - z_r[X3(i, j, N - 1)]);
      wvel[XW(i, j, N)] = pm[X2(i, j)] * pn[X2(i, j)] * wrk[X2(i, j)] *
                              (z_w[XW(i, j, N)] - z_w[XW(i, j, 0)]) +
                          cff1 * (vert[X3(i, j, N)] + slope * (vert[X3(i, j, N)] - vert[X3(i, j, N - 1)])) +
                          cff2 * vert[X3(i, j, N)] - cff3 * vert[X3(i, j, N - 1)];
      wvel[XW(i, j, N - 1)] =
          pm[X2(i, j)] * pn[X2(i, j)] *
              (W[XW(i, j, N - 1)] + wrk[X2(i, j)] * (z_w[XW(i, j, N - 1)] - z_w[XW(i, j, 0)])) +
          cff1 * vert[X3(i, j, N)] + cff2 * vert[X3(i, j, N - 1)] - cff3 * vert[X3(i, j, N - 2)];
    }
  }
  free(vert);
  free(wrk);
  orc_bc_w3d(o, b, wvel, N + 1);
}

/* -------------------------------------------------------------- set_zeta */
void orc_set_zeta(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  for (int j = b->JstrR; j <= b->JendR; j++)
    for (int i = b->IstrR; i <= b->IendR; i++) {
      o->zeta[X2T(i, j, 1)] = o->Zt_avg1[X2(i, j)];
      o->zeta[X2T(i, j, 2)] = o->Zt_avg1[X2(i, j)];
    }
  orc_exchange2d(o, b, 'r', o->zeta);
  orc_exchange2d(o, b, 'r', o->zeta + nij);
}

/* ----------------------------------------- ini_zeta (+ set_zeta_timeavg) */
void orc_ini_zeta(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int kstp = o->s.kstp;
  /* the load zeta(kstp)=zeta(kstp) over IstrB:IendB is an identity without masks; with them :838-849 */
  /* radiation / Chapman conditions need the boundary values of the initial state: the load then covers them and no
     condition is applied (ini_fields.F:830-871) */
  int keep = 0;
  for (int e = 0; e < 4; e++) {
    const int k = orc_lbc(o, e, ORC_ISFSUR);
    keep |= k == ORC_LBC_RAD || k == ORC_LBC_RADNUD || k == ORC_LBC_CHE || k == ORC_LBC_CHI;
  }
  if ((o->c.options & ORC_MASKING) || o->wet_dry)
    for (int j = keep ? b->JstrT : b->JstrB; j <= (keep ? b->JendT : b->JendB); j++)
      for (int i = keep ? b->IstrT : b->IstrB; i <= (keep ? b->IendT : b->IendB); i++) {
        double cff1 = o->zeta[X2T(i, j, kstp)];
        if (o->c.options & ORC_MASKING) cff1 = cff1 * o->rmask[X2(i, j)];
        if (o->wet_dry && cff1 <= (o->Dcrit - o->h[X2(i, j)])) cff1 = o->Dcrit - o->h[X2(i, j)];   /* :850-851 */
        o->zeta[X2T(i, j, kstp)] = cff1;
      }
  if (!keep) orc_zetabc(o, b, kstp);
  orc_exchange2d(o, b, 'r', o->zeta + (size_t)(kstp - 1) * nij);
  for (int j = b->JstrT; j <= b->JendT; j++)
    for (int i = b->IstrT; i <= b->IendT; i++) o->Zt_avg1[X2(i, j)] = o->zeta[X2T(i, j, kstp)];
  orc_exchange2d(o, b, 'r', o->Zt_avg1);
}

/* ------------------------------------------------------------ ini_fields */
void orc_ini_fields(orc_t *o, int tile) {
  ORC_LOCALS(o);
  const orc_bounds *b = &o->b[tile];
  const int nstp = o->s.nstp, kstp = o->s.kstp;
  double *u = o->u, *v = o->v, *Hz = o->Hz;
  const int msk = (o->c.options & ORC_MASKING) != 0;
  if (msk || o->wet_dry)                       /* the loads u(nstp)=u(nstp), v(nstp)=v(nstp) with masks :286-312 */
    for (int j = b->JstrB; j <= b->JendB; j++)
      for (int k = 1; k <= N; k++) {
        for (int i = b->IstrM; i <= b->IendB; i++) {
          if (msk) u[X4(i, j, k, nstp)] = u[X4(i, j, k, nstp)] * o->umask[X2(i, j)];
          if (o->wet_dry) u[X4(i, j, k, nstp)] = u[X4(i, j, k, nstp)] * o->umask_wet[X2(i, j)];   /* :294 */
        }
        if (j >= b->JstrM)
          for (int i = b->IstrB; i <= b->IendB; i++) {
            if (msk) v[X4(i, j, k, nstp)] = v[X4(i, j, k, nstp)] * o->vmask[X2(i, j)];
            if (o->wet_dry) v[X4(i, j, k, nstp)] = v[X4(i, j, k, nstp)] * o->vmask_wet[X2(i, j)];   /* :308 */
          }
      }
  orc_u3dbc(o, b, nstp);
  orc_v3dbc(o, b, nstp);
  orc_exchange3d(o, b, 'u', u + (size_t)(nstp - 1) * nij * N, N);
  orc_exchange3d(o, b, 'v', v + (size_t)(nstp - 1) * nij * N, N);
  double *DC = (double *)malloc(sizeof(double) * o->ni * (size_t)(N + 1));
  double *CF = (double *)malloc(sizeof(double) * o->ni);
#define DCx(i, k) DC[(size_t)((i) - LBi) + (size_t)(k) * ni]
  for (int j = b->JstrB; j <= b->JendB; j++) {
    for (int i = b->IstrM; i <= b->IendB; i++) { DCx(i, 0) = 0.0; CF[i - LBi] = 0.0; }
    for (int k = 1; k <= N; k++)
      for (int i = b->IstrM; i <= b->IendB; i++) {
        DCx(i, k) = 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i - 1, j, k)]);
        DCx(i, 0) = DCx(i, 0) + DCx(i, k);
        CF[i - LBi] = CF[i - LBi] + DCx(i, k) * u[X4(i, j, k, nstp)];
      }
    for (int i = b->IstrM; i <= b->IendB; i++) {
      double cff1 = 1.0 / DCx(i, 0);
      double cff2 = CF[i - LBi] * cff1;
      if (msk) cff2 = cff2 * o->umask[X2(i, j)];                          /* :376 */
      if (o->wet_dry) cff2 = cff2 * o->umask_wet[X2(i, j)];               /* :379 */
      o->ubar[X2T(i, j, kstp)] = cff2;
    }
    if (j >= b->JstrM) {
      for (int i = b->IstrB; i <= b->IendB; i++) { DCx(i, 0) = 0.0; CF[i - LBi] = 0.0; }
      for (int k = 1; k <= N; k++)
        for (int i = b->IstrB; i <= b->IendB; i++) {
          DCx(i, k) = 0.5 * (Hz[X3(i, j, k)] + Hz[X3(i, j - 1, k)]);
          DCx(i, 0) = DCx(i, 0) + DCx(i, k);
          CF[i - LBi] = CF[i - LBi] + DCx(i, k) * v[X4(i, j, k, nstp)];
        }
      for (int i = b->IstrB; i <= b->IendB; i++) {
        double cff1 = 1.0 / DCx(i, 0);
        double cff2 = CF[i - LBi] * cff1;
        if (msk) cff2 = cff2 * o->vmask[X2(i, j)];                        /* :400 */
        if (o->wet_dry) cff2 = cff2 * o->vmask_wet[X2(i, j)];             /* :403 */
        o->vbar[X2T(i, j, kstp)] = cff2;
      }
    }
  }
#undef DCx
  free(DC);
  free(CF);
  {   /* not with radiation or Flather conditions on the barotropic momentum (ini_fields.F:412-425) */
    int keep = 0;
    for (int e = 0; e < 4; e++)
      for (int var = ORC_ISUBAR; var <= ORC_ISVBAR; var++) {
        const int k = orc_lbc(o, e, var);
        keep |= k == ORC_LBC_RAD || k == ORC_LBC_RADNUD || k == ORC_LBC_FLA;
      }
    if (!keep) {
      orc_u2dbc(o, b, kstp);
      orc_v2dbc(o, b, kstp);
    }
  }
  orc_exchange2d(o, b, 'u', o->ubar + (size_t)(kstp - 1) * nij);
  orc_exchange2d(o, b, 'v', o->vbar + (size_t)(kstp - 1) * nij);
  if (msk)                                                                /* t(nstp) * rmask :546-556 */
    for (int it = 1; it <= o->c.NT; it++)
      for (int k = 1; k <= N; k++)
        for (int j = b->JstrB; j <= b->JendB; j++)
          for (int i = b->IstrB; i <= b->IendB; i++)
            o->t[XT(i, j, k, nstp, it)] = o->t[XT(i, j, k, nstp, it)] * o->rmask[X2(i, j)];
  for (int it = 1; it <= o->c.NT; it++) orc_t3dbc(o, b, nstp, it);
  for (int it = 1; it <= o->c.NT; it++)
    orc_exchange3d(o, b, 'r', o->t + ((size_t)(nstp - 1) + 3 * (size_t)(it - 1)) * nij * N, N);
}
