// k_step2d.h -- barotropic LF-AM3 predictor/corrector sub-step as ONE kernel.
//
// Replaces step2d_tile, ROMS/Nonlinear/step2d_LF_AM3.h:163-3056 (options SOLVE3D, VAR_RHO_2D,
// UV_ADV 4th-order centred :1246-1395, UV_COR, CURVGRID, UV_VIS2).
//
// Mapping: one thread block = one ROMS sub-tile; the reference's private work arrays
// (IminS:ImaxS,JminS:JmaxS) live in LDS (12 arrays after aliasing), every loop nest of the
// reference is a block-strided loop and dependent nests are separated by a barrier.  All
// stencil reach (2 cells of DUon/DVom, 3 of Drhs) is satisfied from the 3-cell LDS halo, so the
// kernel reads each 2-D field once per sub-tile and writes zeta/ubar/vbar(knew), the r.h.s.
// history and the fast-time averages once.  Boundary fills and periodic copies follow in
// halo_multi (k_halo.h).
#pragma once
#include "roms_ctx.h"
#include "k_haloblock.h"

struct Step2dArgs {
  DGrid G;
  Fields F;
  double w1_m1;        // weight(1,iif-1)
  double w2_0, w2_p1;  // weight(2,iif), weight(2,iif+1)
};

#define STEP2D_NLDS 12

COOP_KERNEL(k_step2d, Step2dArgs) {
  (void)bz;
  const DGrid &G = a.G;
  const Fields &F = a.F;
  const TB B = block_bounds2(G, bx, by);
  const size_t sz = (size_t)(G.bw2 + 6) * (size_t)(G.bh2 + 6);
  double *Drhs = lds, *DUon = lds + sz, *DVom = lds + 2 * sz, *Dnew = lds + 3 * sz, *rhs_ubar = lds + 4 * sz,
         *rhs_vbar = lds + 5 * sz;
  double *zwrk = lds + 6 * sz, *gzeta = lds + 7 * sz, *gzeta2 = lds + 8 * sz, *gzetaSA = lds + 9 * sz;
  double *grad = lds + 6 * sz, *Dgrad = lds + 7 * sz, *UFx = lds + 8 * sz, *UFe = lds + 9 * sz, *VFx = lds + 10 * sz,
         *VFe = lds + 11 * sz;
  double *Drhs_p = grad;
  const int krhs = G.krhs, kstp = G.kstp, knew = G.knew, nstp = G.nstp, nnew = G.nnew, iif = G.iif, iic = G.iic;
  const bool PRED = G.predictor != 0;
  const int Istr = B.Istr, Iend = B.Iend, Jstr = B.Jstr, Jend = B.Jend, IstrU = B.IstrU, JstrV = B.JstrV;
  const int IstrR = B.IstrR, IendR = B.IendR, JstrR = B.JstrR, JendR = B.JendR;
  const int ptsk = 3 - kstp;
  const double dtfast = G.dtfast, g = G.g;
  const double *zk = F.zeta + (size_t)(krhs - 1) * G.nij, *zs = F.zeta + (size_t)(kstp - 1) * G.nij;
  const double *uk = F.ubar + (size_t)(krhs - 1) * G.nij, *us = F.ubar + (size_t)(kstp - 1) * G.nij;
  const double *vk = F.vbar + (size_t)(krhs - 1) * G.nij, *vs = F.vbar + (size_t)(kstp - 1) * G.nij;
  double *zn = F.zeta + (size_t)(knew - 1) * G.nij, *un = F.ubar + (size_t)(knew - 1) * G.nij,
         *vn = F.vbar + (size_t)(knew - 1) * G.nij;
  const double *h = F.h, *pm = F.pm, *pn = F.pn, *on_u = F.on_u, *om_v = F.om_v, *rhoA = F.rhoA, *rhoS = F.rhoS;

  // total depth and mass fluxes :600-700
  KLOOP2(i, j, B.IstrUm2 - 1, B.Iendp2, B.JstrVm2 - 1, B.Jendp2) Drhs[S2(i, j)] = zk[X2(i, j)] + h[X2(i, j)];
  KSYNC();
  KLOOP2(i, j, B.IstrUm2 - 1, B.Iendp2, B.JstrVm2 - 1, B.Jendp2) {
    if (i >= B.IstrUm2) {
      const double cff = 0.5 * on_u[X2(i, j)];
      const double cff1 = cff * (Drhs[S2(i, j)] + Drhs[S2(i - 1, j)]);
      DUon[S2(i, j)] = uk[X2(i, j)] * cff1;
    }
    if (j >= B.JstrVm2) {
      const double cff = 0.5 * om_v[X2(i, j)];
      const double cff1 = cff * (Drhs[S2(i, j)] + Drhs[S2(i, j - 1)]);
      DVom[S2(i, j)] = vk[X2(i, j)] * cff1;
    }
  }
  KSYNC();

  // fast-time averaging :739-880
  if (PRED) {
    if (iif == 1) {
      const double cff2 = (-1.0 / 12.0) * a.w2_p1;
      KLOOP2(i, j, KMIN(IstrR, Istr), IendR, KMIN(JstrR, Jstr), JendR) {
        if (i >= IstrR && j >= JstrR) F.Zt_avg1[X2(i, j)] = 0.0;
        if (i >= Istr && j >= JstrR) {
          F.DU_avg1[X2(i, j)] = 0.0;
          F.DU_avg2[X2(i, j)] = cff2 * DUon[S2(i, j)];
        }
        if (i >= IstrR && j >= Jstr) {
          F.DV_avg1[X2(i, j)] = 0.0;
          F.DV_avg2[X2(i, j)] = cff2 * DVom[S2(i, j)];
        }
      }
    } else {
      const double cff1 = a.w1_m1;
      const double cff2 = (8.0 / 12.0) * a.w2_0 - (1.0 / 12.0) * a.w2_p1;
      KLOOP2(i, j, KMIN(IstrR, Istr), IendR, KMIN(JstrR, Jstr), JendR) {
        if (i >= IstrR && j >= JstrR) F.Zt_avg1[X2(i, j)] = F.Zt_avg1[X2(i, j)] + cff1 * zk[X2(i, j)];
        if (i >= Istr && j >= JstrR) {
          F.DU_avg1[X2(i, j)] = F.DU_avg1[X2(i, j)] + cff1 * DUon[S2(i, j)];
          F.DU_avg2[X2(i, j)] = F.DU_avg2[X2(i, j)] + cff2 * DUon[S2(i, j)];
        }
        if (i >= IstrR && j >= Jstr) {
          F.DV_avg1[X2(i, j)] = F.DV_avg1[X2(i, j)] + cff1 * DVom[S2(i, j)];
          F.DV_avg2[X2(i, j)] = F.DV_avg2[X2(i, j)] + cff2 * DVom[S2(i, j)];
        }
      }
    }
  } else {
    const double cff2 = (iif == 1) ? a.w2_0 : (5.0 / 12.0) * a.w2_0;
    KLOOP2(i, j, KMIN(IstrR, Istr), IendR, KMIN(JstrR, Jstr), JendR) {
      if (i >= Istr && j >= JstrR) F.DU_avg2[X2(i, j)] = F.DU_avg2[X2(i, j)] + cff2 * DUon[S2(i, j)];
      if (i >= IstrR && j >= Jstr) F.DV_avg2[X2(i, j)] = F.DV_avg2[X2(i, j)] + cff2 * DVom[S2(i, j)];
    }
  }
  if (iif > G.nfast) {         // auxiliary last predictor call :883 (uniform over the grid)
    if (G.fuse_halo && PRED) { // final fast-time averages: exchange :821-883
      HaloBlockItems H;
      H.n = 3;
      H.A[0] = F.Zt_avg1; H.bc[0] = BC_NONE; H.gt[0] = 'r';
      H.A[1] = F.DU_avg1; H.bc[1] = BC_NONE; H.gt[1] = 'u';
      H.A[2] = F.DV_avg1; H.bc[2] = BC_NONE; H.gt[2] = 'v';
      halo_block(G, B, H);
    }
    return;
  }

  // free-surface step :886-1000
  {
    const double fac = 1000.0 / G.rho0;
    double cff1, cff2 = 0.0, cff3 = 0.0, cff4, cff5;
    int mode;
    if (iif == 1) { mode = 0; cff1 = dtfast; cff4 = 0.0; cff5 = 0.0; }
    else if (PRED) { mode = 1; cff1 = 2.0 * dtfast; cff4 = 4.0 / 25.0; cff5 = 1.0 - 2.0 * cff4; }
    else { mode = 2; cff1 = dtfast * 5.0 / 12.0; cff2 = dtfast * 8.0 / 12.0; cff3 = dtfast * 1.0 / 12.0; cff4 = 2.0 / 5.0; cff5 = 1.0 - cff4; }
    const double *rz_s = F.rzeta + (size_t)(kstp - 1) * G.nij, *rz_p = F.rzeta + (size_t)(ptsk - 1) * G.nij;
    double *rz_k = F.rzeta + (size_t)(krhs - 1) * G.nij;
    KLOOP2(i, j, IstrU - 1, Iend, JstrV - 1, Jend) {
      const double rhs_zeta = (DUon[S2(i, j)] - DUon[S2(i + 1, j)]) + (DVom[S2(i, j)] - DVom[S2(i, j + 1)]);
      double zeta_new, zw;
      if (mode == 0) {
        zeta_new = zs[X2(i, j)] + pm[X2(i, j)] * pn[X2(i, j)] * cff1 * rhs_zeta;
        zw = 0.5 * (zs[X2(i, j)] + zeta_new);
      } else if (mode == 1) {
        zeta_new = zs[X2(i, j)] + pm[X2(i, j)] * pn[X2(i, j)] * cff1 * rhs_zeta;
        zw = cff5 * zk[X2(i, j)] + cff4 * (zs[X2(i, j)] + zeta_new);
      } else {
        const double cff = cff1 * rhs_zeta;
        zeta_new = zs[X2(i, j)] + pm[X2(i, j)] * pn[X2(i, j)] * (cff + cff2 * rz_s[X2(i, j)] - cff3 * rz_p[X2(i, j)]);
        zw = cff5 * zeta_new + cff4 * zk[X2(i, j)];
      }
      Dnew[S2(i, j)] = zeta_new + h[X2(i, j)];
      zwrk[S2(i, j)] = zw;
      const double gz = (fac + rhoS[X2(i, j)]) * zw;
      gzeta[S2(i, j)] = gz;
      gzeta2[S2(i, j)] = gz * zw;
      gzetaSA[S2(i, j)] = zw * (rhoS[X2(i, j)] - rhoA[X2(i, j)]);
      if (i >= Istr && j >= Jstr) {
        zn[X2(i, j)] = zeta_new;
        if (PRED) rz_k[X2(i, j)] = rhs_zeta;
      }
    }
  }
  KSYNC();

  // pressure gradient (VAR_RHO_2D) :1080-1200
  {
    const double cff1 = 0.5 * g, cff2 = 1.0 / 3.0;
    KLOOP2(i, j, KMIN(IstrU, Istr), Iend, Jstr, Jend) {
      if (i >= IstrU)
        rhs_ubar[S2(i, j)] =
            cff1 * on_u[X2(i, j)] *
            ((h[X2(i - 1, j)] + h[X2(i, j)]) * (gzeta[S2(i - 1, j)] - gzeta[S2(i, j)]) +
             (h[X2(i - 1, j)] - h[X2(i, j)]) *
                 (gzetaSA[S2(i - 1, j)] + gzetaSA[S2(i, j)] +
                  cff2 * (rhoA[X2(i - 1, j)] - rhoA[X2(i, j)]) * (zwrk[S2(i - 1, j)] - zwrk[S2(i, j)])) +
             (gzeta2[S2(i - 1, j)] - gzeta2[S2(i, j)]));
      if (j >= JstrV)
        rhs_vbar[S2(i, j)] =
            cff1 * om_v[X2(i, j)] *
            ((h[X2(i, j - 1)] + h[X2(i, j)]) * (gzeta[S2(i, j - 1)] - gzeta[S2(i, j)]) +
             (h[X2(i, j - 1)] - h[X2(i, j)]) *
                 (gzetaSA[S2(i, j - 1)] + gzetaSA[S2(i, j)] +
                  cff2 * (rhoA[X2(i, j - 1)] - rhoA[X2(i, j)]) * (zwrk[S2(i, j - 1)] - zwrk[S2(i, j)])) +
             (gzeta2[S2(i, j - 1)] - gzeta2[S2(i, j)]));
    }
  }
  KSYNC();   // group-1 scratch is dead from here on (aliased by grad,Dgrad,UFx,UFe)

  if (G.options & ROMS_UV_ADV) {
    const double cff = 1.0 / 6.0;
    // ---- UFx :1249-1290
    KLOOP2(i, j, B.IstrUm1, B.Iendp1, Jstr, Jend) {
      grad[S2(i, j)] = uk[X2(i - 1, j)] - 2.0 * uk[X2(i, j)] + uk[X2(i + 1, j)];
      Dgrad[S2(i, j)] = DUon[S2(i - 1, j)] - 2.0 * DUon[S2(i, j)] + DUon[S2(i + 1, j)];
    }
    KSYNC();
    if (!G.ewp) {
      if (B.west) KLOOP1(j, Jstr, Jend) { grad[S2(Istr, j)] = grad[S2(Istr + 1, j)]; Dgrad[S2(Istr, j)] = Dgrad[S2(Istr + 1, j)]; }
      if (B.east) KLOOP1(j, Jstr, Jend) { grad[S2(Iend + 1, j)] = grad[S2(Iend, j)]; Dgrad[S2(Iend + 1, j)] = Dgrad[S2(Iend, j)]; }
    }
    KSYNC();
    KLOOP2(i, j, IstrU - 1, Iend, Jstr, Jend)
      UFx[S2(i, j)] = 0.25 * (uk[X2(i, j)] + uk[X2(i + 1, j)] - cff * (grad[S2(i, j)] + grad[S2(i + 1, j)])) *
                      (DUon[S2(i, j)] + DUon[S2(i + 1, j)] - cff * (Dgrad[S2(i, j)] + Dgrad[S2(i + 1, j)]));
    KSYNC();
    // ---- UFe :1292-1330
    KLOOP2(i, j, IstrU, Iend, B.Jstrm1, B.Jendp1)
      grad[S2(i, j)] = uk[X2(i, j - 1)] - 2.0 * uk[X2(i, j)] + uk[X2(i, j + 1)];
    KLOOP2(i, j, IstrU - 1, Iend, Jstr, Jend + 1)
      Dgrad[S2(i, j)] = DVom[S2(i - 1, j)] - 2.0 * DVom[S2(i, j)] + DVom[S2(i + 1, j)];
    KSYNC();
    if (!G.nsp) {
      if (B.south) KLOOP1(i, IstrU, Iend) grad[S2(i, Jstr - 1)] = grad[S2(i, Jstr)];
      if (B.north) KLOOP1(i, IstrU, Iend) grad[S2(i, Jend + 1)] = grad[S2(i, Jend)];
    }
    KSYNC();
    KLOOP2(i, j, IstrU, Iend, Jstr, Jend + 1)
      UFe[S2(i, j)] = 0.25 * (uk[X2(i, j)] + uk[X2(i, j - 1)] - cff * (grad[S2(i, j)] + grad[S2(i, j - 1)])) *
                      (DVom[S2(i, j)] + DVom[S2(i - 1, j)] - cff * (Dgrad[S2(i, j)] + Dgrad[S2(i - 1, j)]));
    KSYNC();
    // u-momentum advection r.h.s. (UFx,UFe complete)
    KLOOP2(i, j, IstrU, Iend, Jstr, Jend) {
      const double cff1 = UFx[S2(i, j)] - UFx[S2(i - 1, j)];
      const double cff2 = UFe[S2(i, j + 1)] - UFe[S2(i, j)];
      const double fac = cff1 + cff2;
      rhs_ubar[S2(i, j)] = rhs_ubar[S2(i, j)] - fac;
    }
    KSYNC();
    // ---- VFx :1332-1370
    KLOOP2(i, j, B.Istrm1, B.Iendp1, JstrV, Jend)
      grad[S2(i, j)] = vk[X2(i - 1, j)] - 2.0 * vk[X2(i, j)] + vk[X2(i + 1, j)];
    KLOOP2(i, j, Istr, Iend + 1, JstrV - 1, Jend)
      Dgrad[S2(i, j)] = DUon[S2(i, j - 1)] - 2.0 * DUon[S2(i, j)] + DUon[S2(i, j + 1)];
    KSYNC();
    if (!G.ewp) {
      if (B.west) KLOOP1(j, JstrV, Jend) grad[S2(Istr - 1, j)] = grad[S2(Istr, j)];
      if (B.east) KLOOP1(j, JstrV, Jend) grad[S2(Iend + 1, j)] = grad[S2(Iend, j)];
    }
    KSYNC();
    KLOOP2(i, j, Istr, Iend + 1, JstrV, Jend)
      VFx[S2(i, j)] = 0.25 * (vk[X2(i, j)] + vk[X2(i - 1, j)] - cff * (grad[S2(i, j)] + grad[S2(i - 1, j)])) *
                      (DUon[S2(i, j)] + DUon[S2(i, j - 1)] - cff * (Dgrad[S2(i, j)] + Dgrad[S2(i, j - 1)]));
    KSYNC();
    // ---- VFe :1372-1410
    KLOOP2(i, j, Istr, Iend, B.JstrVm1, B.Jendp1) {
      grad[S2(i, j)] = vk[X2(i, j - 1)] - 2.0 * vk[X2(i, j)] + vk[X2(i, j + 1)];
      Dgrad[S2(i, j)] = DVom[S2(i, j - 1)] - 2.0 * DVom[S2(i, j)] + DVom[S2(i, j + 1)];
    }
    KSYNC();
    if (!G.nsp) {
      if (B.south) KLOOP1(i, Istr, Iend) { grad[S2(i, Jstr)] = grad[S2(i, Jstr + 1)]; Dgrad[S2(i, Jstr)] = Dgrad[S2(i, Jstr + 1)]; }
      if (B.north) KLOOP1(i, Istr, Iend) { grad[S2(i, Jend + 1)] = grad[S2(i, Jend)]; Dgrad[S2(i, Jend + 1)] = Dgrad[S2(i, Jend)]; }
    }
    KSYNC();
    KLOOP2(i, j, Istr, Iend, JstrV - 1, Jend)
      VFe[S2(i, j)] = 0.25 * (vk[X2(i, j)] + vk[X2(i, j + 1)] - cff * (grad[S2(i, j)] + grad[S2(i, j + 1)])) *
                      (DVom[S2(i, j)] + DVom[S2(i, j + 1)] - cff * (Dgrad[S2(i, j)] + Dgrad[S2(i, j + 1)]));
    KSYNC();
    KLOOP2(i, j, Istr, Iend, JstrV, Jend) {
      const double cff1 = VFx[S2(i + 1, j)] - VFx[S2(i, j)];
      const double cff2 = VFe[S2(i, j)] - VFe[S2(i, j - 1)];
      const double fac = cff1 + cff2;
      rhs_vbar[S2(i, j)] = rhs_vbar[S2(i, j)] - fac;
    }
    KSYNC();
  }

  if (G.options & ROMS_UV_COR) {
    // Coriolis :1429-1490
    KLOOP2(i, j, IstrU - 1, Iend, JstrV - 1, Jend) {
      const double cff = 0.5 * Drhs[S2(i, j)] * F.fomn[X2(i, j)];
      UFx[S2(i, j)] = cff * (vk[X2(i, j)] + vk[X2(i, j + 1)]);
      VFe[S2(i, j)] = cff * (uk[X2(i, j)] + uk[X2(i + 1, j)]);
    }
    KSYNC();
    KLOOP2(i, j, KMIN(IstrU, Istr), Iend, KMIN(Jstr, JstrV), Jend) {
      if (i >= IstrU && j >= Jstr) {
        const double fac1 = 0.5 * (UFx[S2(i, j)] + UFx[S2(i - 1, j)]);
        rhs_ubar[S2(i, j)] = rhs_ubar[S2(i, j)] + fac1;
      }
      if (i >= Istr && j >= JstrV) {
        const double fac1 = 0.5 * (VFe[S2(i, j)] + VFe[S2(i, j - 1)]);
        rhs_vbar[S2(i, j)] = rhs_vbar[S2(i, j)] - fac1;
      }
    }
    KSYNC();
  }

  if ((G.options & ROMS_CURVGRID) && (G.options & ROMS_UV_ADV)) {
    // curvilinear metric terms :1494-1560
    KLOOP2(i, j, IstrU - 1, Iend, JstrV - 1, Jend) {
      const double cff1 = 0.5 * (vk[X2(i, j)] + vk[X2(i, j + 1)]);
      const double cff2 = 0.5 * (uk[X2(i, j)] + uk[X2(i + 1, j)]);
      const double cff3 = cff1 * F.dndx[X2(i, j)];
      const double cff4 = cff2 * F.dmde[X2(i, j)];
      const double cff = Drhs[S2(i, j)] * (cff3 - cff4);
      UFx[S2(i, j)] = cff * cff1;
      VFe[S2(i, j)] = cff * cff2;
    }
    KSYNC();
    KLOOP2(i, j, KMIN(IstrU, Istr), Iend, KMIN(Jstr, JstrV), Jend) {
      if (i >= IstrU && j >= Jstr) {
        const double fac1 = 0.5 * (UFx[S2(i, j)] + UFx[S2(i - 1, j)]);
        rhs_ubar[S2(i, j)] = rhs_ubar[S2(i, j)] + fac1;
      }
      if (i >= Istr && j >= JstrV) {
        const double fac1 = 0.5 * (VFe[S2(i, j)] + VFe[S2(i, j - 1)]);
        rhs_vbar[S2(i, j)] = rhs_vbar[S2(i, j)] - fac1;
      }
    }
    KSYNC();
  }

  if (G.options & ROMS_UV_VIS2) {
    // harmonic viscosity :1567-1660
    KLOOP2(i, j, Istr, Iend + 1, Jstr, Jend + 1)
      Drhs_p[S2(i, j)] = 0.25 * (Drhs[S2(i, j)] + Drhs[S2(i - 1, j)] + Drhs[S2(i, j - 1)] + Drhs[S2(i - 1, j - 1)]);
    KLOOP2(i, j, IstrU - 1, Iend, JstrV - 1, Jend) {
      const double cff = F.visc2_r[X2(i, j)] * Drhs[S2(i, j)] * 0.5 *
                         (F.pmon_r[X2(i, j)] * ((pn[X2(i, j)] + pn[X2(i + 1, j)]) * uk[X2(i + 1, j)] -
                                                (pn[X2(i - 1, j)] + pn[X2(i, j)]) * uk[X2(i, j)]) -
                          F.pnom_r[X2(i, j)] * ((pm[X2(i, j)] + pm[X2(i, j + 1)]) * vk[X2(i, j + 1)] -
                                                (pm[X2(i, j - 1)] + pm[X2(i, j)]) * vk[X2(i, j)]));
      UFx[S2(i, j)] = F.on_r[X2(i, j)] * F.on_r[X2(i, j)] * cff;
      VFe[S2(i, j)] = F.om_r[X2(i, j)] * F.om_r[X2(i, j)] * cff;
    }
    KSYNC();
    KLOOP2(i, j, Istr, Iend + 1, Jstr, Jend + 1) {
      const double cff = F.visc2_p[X2(i, j)] * Drhs_p[S2(i, j)] * 0.5 *
                         (F.pmon_p[X2(i, j)] * ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * vk[X2(i, j)] -
                                                (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)]) * vk[X2(i - 1, j)]) +
                          F.pnom_p[X2(i, j)] * ((pm[X2(i - 1, j)] + pm[X2(i, j)]) * uk[X2(i, j)] -
                                                (pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]) * uk[X2(i, j - 1)]));
      UFe[S2(i, j)] = F.om_p[X2(i, j)] * F.om_p[X2(i, j)] * cff;
      VFx[S2(i, j)] = F.on_p[X2(i, j)] * F.on_p[X2(i, j)] * cff;
    }
    KSYNC();
    KLOOP2(i, j, KMIN(IstrU, Istr), Iend, KMIN(Jstr, JstrV), Jend) {
      if (i >= IstrU && j >= Jstr) {
        const double cff1 = 0.5 * (pn[X2(i - 1, j)] + pn[X2(i, j)]) * (UFx[S2(i, j)] - UFx[S2(i - 1, j)]);
        const double cff2 = 0.5 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (UFe[S2(i, j + 1)] - UFe[S2(i, j)]);
        const double fac = cff1 + cff2;
        rhs_ubar[S2(i, j)] = rhs_ubar[S2(i, j)] + fac;
      }
      if (i >= Istr && j >= JstrV) {
        const double cff1 = 0.5 * (pn[X2(i, j - 1)] + pn[X2(i, j)]) * (VFx[S2(i + 1, j)] - VFx[S2(i, j)]);
        const double cff2 = 0.5 * (pm[X2(i, j - 1)] + pm[X2(i, j)]) * (VFe[S2(i, j)] - VFe[S2(i, j - 1)]);
        const double fac = cff1 - cff2;
        rhs_vbar[S2(i, j)] = rhs_vbar[S2(i, j)] + fac;
      }
    }
    KSYNC();
  }

  // coupling with the 3-D forcing :2225-2460, then the momentum step :2488-2670 -- point-wise
  {
    const int first = (iif == 1 && PRED);
    const int startup = (iic == G.ntfirst) ? 0 : ((iic == G.ntfirst + 1) ? 1 : 2);
    const double *rub_s = F.rubar + (size_t)(kstp - 1) * G.nij, *rub_p = F.rubar + (size_t)(ptsk - 1) * G.nij;
    const double *rvb_s = F.rvbar + (size_t)(kstp - 1) * G.nij, *rvb_p = F.rvbar + (size_t)(ptsk - 1) * G.nij;
    double *rub_k = F.rubar + (size_t)(krhs - 1) * G.nij, *rvb_k = F.rvbar + (size_t)(krhs - 1) * G.nij;
    double *ru0_stp = F.ru + (size_t)(nstp - 1) * G.nij * (size_t)(G.N + 1);   // ru(:,:,0,nstp)
    double *ru0_new = F.ru + (size_t)(nnew - 1) * G.nij * (size_t)(G.N + 1);
    double *rv0_stp = F.rv + (size_t)(nstp - 1) * G.nij * (size_t)(G.N + 1);
    double *rv0_new = F.rv + (size_t)(nnew - 1) * G.nij * (size_t)(G.N + 1);
    const int corr = (!PRED && iif != 1);
    const double c1 = (iif == 1) ? 0.5 * dtfast : dtfast;
    const double k1 = 0.5 * dtfast * 5.0 / 12.0, k2 = 0.5 * dtfast * 8.0 / 12.0, k3 = 0.5 * dtfast * 1.0 / 12.0;
    KLOOP2(i, j, KMIN(IstrU, Istr), Iend, KMIN(Jstr, JstrV), Jend) {
      if (i >= IstrU && j >= Jstr) {
        double r = rhs_ubar[S2(i, j)];
        if (first) {
          const double fr = F.rufrc[X2(i, j)] - r;
          F.rufrc[X2(i, j)] = fr;
          if (startup == 0) r = r + fr;
          else if (startup == 1) r = r + 1.5 * fr - 0.5 * ru0_new[X2(i, j)];
          else r = r + (23.0 / 12.0) * fr - (16.0 / 12.0) * ru0_new[X2(i, j)] + (5.0 / 12.0) * ru0_stp[X2(i, j)];
          ru0_stp[X2(i, j)] = fr;
        } else {
          r = r + F.rufrc[X2(i, j)];
        }
        const double cff = (pm[X2(i, j)] + pm[X2(i - 1, j)]) * (pn[X2(i, j)] + pn[X2(i - 1, j)]);
        const double fac = 1.0 / (Dnew[S2(i, j)] + Dnew[S2(i - 1, j)]);
        const double Dstp_i = zs[X2(i, j)] + h[X2(i, j)], Dstp_im = zs[X2(i - 1, j)] + h[X2(i - 1, j)];
        double ub;
        if (!corr) ub = (us[X2(i, j)] * (Dstp_i + Dstp_im) + cff * c1 * r) * fac;
        else ub = (us[X2(i, j)] * (Dstp_i + Dstp_im) + cff * (k1 * r + k2 * rub_s[X2(i, j)] - k3 * rub_p[X2(i, j)])) * fac;
        un[X2(i, j)] = ub;
        if (PRED) rub_k[X2(i, j)] = r;
      }
      if (i >= Istr && j >= JstrV) {
        double r = rhs_vbar[S2(i, j)];
        if (first) {
          const double fr = F.rvfrc[X2(i, j)] - r;
          F.rvfrc[X2(i, j)] = fr;
          if (startup == 0) r = r + fr;
          else if (startup == 1) r = r + 1.5 * fr - 0.5 * rv0_new[X2(i, j)];
          else r = r + (23.0 / 12.0) * fr - (16.0 / 12.0) * rv0_new[X2(i, j)] + (5.0 / 12.0) * rv0_stp[X2(i, j)];
          rv0_stp[X2(i, j)] = fr;
        } else {
          r = r + F.rvfrc[X2(i, j)];
        }
        const double cff = (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
        const double fac = 1.0 / (Dnew[S2(i, j)] + Dnew[S2(i, j - 1)]);
        const double Dstp_j = zs[X2(i, j)] + h[X2(i, j)], Dstp_jm = zs[X2(i, j - 1)] + h[X2(i, j - 1)];
        double vb;
        if (!corr) vb = (vs[X2(i, j)] * (Dstp_j + Dstp_jm) + cff * c1 * r) * fac;
        else vb = (vs[X2(i, j)] * (Dstp_j + Dstp_jm) + cff * (k1 * r + k2 * rvb_s[X2(i, j)] - k3 * rvb_p[X2(i, j)])) * fac;
        vn[X2(i, j)] = vb;
        if (PRED) rvb_k[X2(i, j)] = r;
      }
    }
  }
  // zetabc :1057 + exchange :1068, rzeta exchange :1030, u2dbc/v2dbc :2871-2876 + exchange :3043
  if (G.fuse_halo) {
    HaloBlockItems H;
    int n = 0;
    H.A[n] = zn; H.bc[n] = BC_R; H.gt[n] = 'r'; n++;
    if (PRED) { H.A[n] = F.rzeta + (size_t)(krhs - 1) * G.nij; H.bc[n] = BC_NONE; H.gt[n] = 'r'; n++; }
    H.A[n] = un; H.bc[n] = BC_U; H.gt[n] = 'u'; n++;
    H.A[n] = vn; H.bc[n] = BC_V; H.gt[n] = 'v'; n++;
    H.n = n;
    halo_block(G, B, H);
  }
}
COOP_GLOBAL(k_step2d, Step2dArgs)
