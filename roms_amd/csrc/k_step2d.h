// k_step2d.h -- barotropic LF-AM3 predictor/corrector sub-step as ONE kernel.
//
// Replaces step2d_tile, ROMS/Nonlinear/step2d_LF_AM3.h:163-3056 (options SOLVE3D, VAR_RHO_2D,
// UV_ADV 4th-order centred :1246-1395, UV_COR, CURVGRID, UV_VIS2).
//
// Mapping: one thread block = one ROMS sub-tile; the reference's private work arrays
// (IminS:ImaxS,JminS:JmaxS) live in LDS, every loop nest of the reference is a masked sweep over
// the sub-tile's (Istr-3:Iend+3, Jstr-3:Jend+3) rectangle and dependent nests are separated by a
// barrier.
//
// The kernel is a chain of ~25 short dependent phases, so its cost is latency, not bandwidth.  All
// global reads are therefore issued in ONE prologue: fields that are needed at neighbouring points
// (zeta+h, ubar, vbar, h, pm, pn, rhoA, Dstp) go to LDS tiles with the 3-cell halo, everything
// that is only needed at a thread's own points (metrics, r.h.s. history, running averages) stays
// in that thread's registers -- every sweep uses the same point -> thread mapping.  The phases then
// touch LDS and registers only; the epilogue stores zeta/ubar/vbar(knew), the r.h.s. history and
// the fast-time averages; in single-tile runs with a periodic direction every thread also stores
// the boundary and periodic images of its values (k_haloblock.h), so no halo launch follows.
#pragma once
#include "roms_ctx.h"
#include "k_haloblock.h"

struct Step2dArgs {
  DGrid G;
  const Fields *Fp;   // device-resident table of array pointers (roms_hip_ctx::d_F)
  double w1_m1;        // weight(1,iif-1)
  double w2_0, w2_p1;  // weight(2,iif), weight(2,iif+1)
};

#define STEP2D_NLDS 19
#define STEP2D_PTS 2      // tile points per thread: the launch uses >= tile/2 threads

// Sweep over the sub-tile rectangle with a fixed point -> thread mapping.
#ifdef ROMS_CPU_EMU
// serial emulation: one "thread" visits all points; own-point values are read where they are used
#define TLOOP(i, j)                                                                                                    \
  for (int m = 0, i = 0, j = 0, s0 = 0; m < NTILE && ((j = JT0 + m / TW), (i = IT0 + m - (m / TW) * TW), (s0 = m), true); m++) \
    for (long x0 = (long)X2(i, j), once_ = 1; once_; once_ = 0)
#define PWDECL(name)
#define PWLOAD(name, expr) ((void)0)
#define PW(name, expr) (expr)
#else
#define TLOOP(i, j)                                                                                                    \
  _Pragma("unroll") for (int m = 0; m < STEP2D_PTS; m++) if (tq[m])                                                    \
    for (int i = ti[m], j = tj[m], s0 = KTID + m * KNT, x0 = tx[m], once_ = 1; once_; once_ = 0)
#define PWDECL(name) double name[STEP2D_PTS]
#define PWLOAD(name, expr) name[m] = (expr)
#define PW(name, expr) name[m]
#endif
#define INR(i, j, i0, i1, j0, j1) ((i) >= (i0) && (i) <= (i1) && (j) >= (j0) && (j) <= (j1))

COOP_KERNEL(k_step2d, Step2dArgs) {
  (void)bz;
  const DGrid &G = a.G;
  const Fields &F = *a.Fp;
  const TB B = block_bounds2(G, bx, by);
  const size_t sz = (size_t)(G.bw2 + 6) * (size_t)(G.bh2 + 6);
  double *Drhs = lds, *DUon = lds + sz, *DVom = lds + 2 * sz, *Dnew = lds + 3 * sz, *rhs_ubar = lds + 4 * sz,
         *rhs_vbar = lds + 5 * sz;
  double *zwrk = lds + 6 * sz, *gzeta = lds + 7 * sz, *gzeta2 = lds + 8 * sz, *gzetaSA = lds + 9 * sz;
  double *grad = lds + 6 * sz, *Dgrad = lds + 7 * sz, *UFx = lds + 8 * sz, *UFe = lds + 9 * sz, *VFx = lds + 10 * sz,
         *VFe = lds + 11 * sz;
  double *Drhs_p = grad;
  double *sUk = lds + 12 * sz, *sVk = lds + 13 * sz, *sH = lds + 14 * sz, *sPm = lds + 15 * sz, *sPn = lds + 16 * sz,
         *sRhoA = lds + 17 * sz, *sDstp = lds + 18 * sz;
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
  const double *rz_s = F.rzeta + (size_t)(kstp - 1) * G.nij, *rz_p = F.rzeta + (size_t)(ptsk - 1) * G.nij;
  double *rz_k = F.rzeta + (size_t)(krhs - 1) * G.nij;
  const double *rub_s = F.rubar + (size_t)(kstp - 1) * G.nij, *rub_p = F.rubar + (size_t)(ptsk - 1) * G.nij;
  const double *rvb_s = F.rvbar + (size_t)(kstp - 1) * G.nij, *rvb_p = F.rvbar + (size_t)(ptsk - 1) * G.nij;
  double *rub_k = F.rubar + (size_t)(krhs - 1) * G.nij, *rvb_k = F.rvbar + (size_t)(krhs - 1) * G.nij;
  double *ru0_stp = F.ru + (size_t)(nstp - 1) * G.nij * (size_t)(G.N + 1);   // ru(:,:,0,nstp)
  double *ru0_new = F.ru + (size_t)(nnew - 1) * G.nij * (size_t)(G.N + 1);
  double *rv0_stp = F.rv + (size_t)(nstp - 1) * G.nij * (size_t)(G.N + 1);
  double *rv0_new = F.rv + (size_t)(nnew - 1) * G.nij * (size_t)(G.N + 1);
  const bool last = iif > G.nfast;                 // auxiliary last predictor call :883
  const bool fuse = G.fuse_halo != 0, fuse_last = fuse && last && PRED;
  const int first = (iif == 1 && PRED);
  const int corr = (!PRED && iif != 1);
  const int startup = (iic == G.ntfirst) ? 0 : ((iic == G.ntfirst + 1) ? 1 : 2);

  // sub-tile rectangle and the point -> thread mapping
  const int IT0 = Istr - 3, JT0 = Jstr - 3, TW = Iend - Istr + 7, TH = Jend - Jstr + 7, NTILE = TW * TH;
  const int UBi = G.LBi + G.ni - 1, UBj = G.LBj + G.nj - 1;
#ifndef ROMS_CPU_EMU
  int ti[STEP2D_PTS], tj[STEP2D_PTS], tx[STEP2D_PTS];   // point, and its offset in the global 2-D arrays
  bool tq[STEP2D_PTS];
#pragma unroll
  for (int m = 0; m < STEP2D_PTS; m++) {
    const int q = KTID + m * KNT;
    tq[m] = q < NTILE;
    tj[m] = JT0 + q / TW;
    ti[m] = IT0 + q - (q / TW) * TW;
    tx[m] = (ti[m] - G.LBi) + (tj[m] - G.LBj) * G.ni;   // may lie outside the array for masked points
  }
#endif
  PWDECL(r_zk); PWDECL(r_zs); PWDECL(r_on_u); PWDECL(r_om_v); PWDECL(r_rhoS); PWDECL(r_fomn); PWDECL(r_dndx);
  PWDECL(r_dmde); PWDECL(r_visc2_r); PWDECL(r_pmon_r); PWDECL(r_pnom_r); PWDECL(r_on_r); PWDECL(r_om_r);
  PWDECL(r_visc2_p); PWDECL(r_pmon_p); PWDECL(r_pnom_p); PWDECL(r_om_p); PWDECL(r_on_p);
  PWDECL(r_Zt); PWDECL(r_DU1); PWDECL(r_DU2); PWDECL(r_DV1); PWDECL(r_DV2);
  PWDECL(r_rz_s); PWDECL(r_rz_p);
  PWDECL(r_rub_s); PWDECL(r_rub_p); PWDECL(r_rvb_s); PWDECL(r_rvb_p); PWDECL(r_rufrc); PWDECL(r_rvfrc);
  PWDECL(r_ru0n); PWDECL(r_ru0s); PWDECL(r_rv0n); PWDECL(r_rv0s); PWDECL(r_us); PWDECL(r_vs);

  // ---- prologue: every global read of the kernel -------------------------------------------
  TLOOP(i, j) {
    if (INR(i, j, G.LBi, UBi, G.LBj, UBj)) {
      // each array is read only as far from the sub-tile as some phase below needs it
      const bool ring2 = INR(i, j, Istr - 2, Iend + 2, Jstr - 2, Jend + 2);
      const bool ring1 = INR(i, j, Istr - 1, Iend + 1, Jstr - 1, Jend + 1);
      const bool own = INR(i, j, Istr, Iend, Jstr, Jend);
      const double zkv = zk[x0], hv = F.h[x0];
      Drhs[s0] = zkv + hv;                                    // total depth :600
      sUk[s0] = uk[x0]; sVk[s0] = vk[x0]; sH[s0] = hv;
      PWLOAD(r_zk, zkv);
      PWLOAD(r_on_u, F.on_u[x0]); PWLOAD(r_om_v, F.om_v[x0]);
      if (ring2) { sPm[s0] = F.pm[x0]; sPn[s0] = F.pn[x0]; }
      if (ring1) {
        const double zsv = zs[x0];
        sDstp[s0] = zsv + hv;
        PWLOAD(r_zs, zsv);
        sRhoA[s0] = F.rhoA[x0];
        PWLOAD(r_Zt, F.Zt_avg1[x0]); PWLOAD(r_DU1, F.DU_avg1[x0]); PWLOAD(r_DU2, F.DU_avg2[x0]);
        PWLOAD(r_DV1, F.DV_avg1[x0]); PWLOAD(r_DV2, F.DV_avg2[x0]);
        if (!last) {
          PWLOAD(r_rhoS, F.rhoS[x0]); PWLOAD(r_fomn, F.fomn[x0]); PWLOAD(r_dndx, F.dndx[x0]); PWLOAD(r_dmde, F.dmde[x0]);
          PWLOAD(r_visc2_r, F.visc2_r[x0]); PWLOAD(r_pmon_r, F.pmon_r[x0]); PWLOAD(r_pnom_r, F.pnom_r[x0]);
          PWLOAD(r_on_r, F.on_r[x0]); PWLOAD(r_om_r, F.om_r[x0]);
          PWLOAD(r_visc2_p, F.visc2_p[x0]); PWLOAD(r_pmon_p, F.pmon_p[x0]); PWLOAD(r_pnom_p, F.pnom_p[x0]);
          PWLOAD(r_om_p, F.om_p[x0]); PWLOAD(r_on_p, F.on_p[x0]);
          if (corr) { PWLOAD(r_rz_s, rz_s[x0]); PWLOAD(r_rz_p, rz_p[x0]); }
        }
      }
      if (own && !last) {
        PWLOAD(r_us, us[x0]); PWLOAD(r_vs, vs[x0]);
        PWLOAD(r_rufrc, F.rufrc[x0]); PWLOAD(r_rvfrc, F.rvfrc[x0]);
        if (corr) {
          PWLOAD(r_rub_s, rub_s[x0]); PWLOAD(r_rub_p, rub_p[x0]); PWLOAD(r_rvb_s, rvb_s[x0]); PWLOAD(r_rvb_p, rvb_p[x0]);
        }
        if (first && startup >= 1) { PWLOAD(r_ru0n, ru0_new[x0]); PWLOAD(r_rv0n, rv0_new[x0]); }
        if (first && startup >= 2) { PWLOAD(r_ru0s, ru0_stp[x0]); PWLOAD(r_rv0s, rv0_stp[x0]); }
      }
    } else {
      Drhs[s0] = 0.0; sDstp[s0] = 0.0; sUk[s0] = 0.0; sVk[s0] = 0.0; sH[s0] = 0.0; sPm[s0] = 0.0; sPn[s0] = 0.0; sRhoA[s0] = 0.0;
    }
  }
  KSYNC();

  // mass fluxes :600-700
  TLOOP(i, j) {
    if (INR(i, j, B.IstrUm2 - 1, B.Iendp2, B.JstrVm2 - 1, B.Jendp2)) {
      if (i >= B.IstrUm2) {
        const double cff = 0.5 * PW(r_on_u, F.on_u[x0]);
        const double cff1 = cff * (Drhs[s0] + Drhs[(s0 - 1)]);
        DUon[s0] = sUk[s0] * cff1;
      }
      if (j >= B.JstrVm2) {
        const double cff = 0.5 * PW(r_om_v, F.om_v[x0]);
        const double cff1 = cff * (Drhs[s0] + Drhs[(s0 - TW)]);
        DVom[s0] = sVk[s0] * cff1;
      }
    }
  }
  KSYNC();

  // fast-time averaging :739-880
  if (PRED) {
    if (iif == 1) {
      const double cff2 = (-1.0 / 12.0) * a.w2_p1;
      TLOOP(i, j) {
        if (INR(i, j, KMIN(IstrR, Istr), IendR, KMIN(JstrR, Jstr), JendR)) {
          if (i >= IstrR && j >= JstrR) F.Zt_avg1[x0] = 0.0;
          if (i >= Istr && j >= JstrR) {
            F.DU_avg1[x0] = 0.0;
            F.DU_avg2[x0] = cff2 * DUon[s0];
          }
          if (i >= IstrR && j >= Jstr) {
            F.DV_avg1[x0] = 0.0;
            F.DV_avg2[x0] = cff2 * DVom[s0];
          }
        }
      }
    } else {
      const double cff1 = a.w1_m1;
      const double cff2 = (8.0 / 12.0) * a.w2_0 - (1.0 / 12.0) * a.w2_p1;
      TLOOP(i, j) {
        if (INR(i, j, KMIN(IstrR, Istr), IendR, KMIN(JstrR, Jstr), JendR)) {
          if (i >= IstrR && j >= JstrR) {
            const double v = PW(r_Zt, F.Zt_avg1[x0]) + cff1 * PW(r_zk, zk[x0]);
            if (fuse_last) hb_emit(G, B, F.Zt_avg1, BC_NONE, i, j, v);   // final averages: exchange :821-883
            else F.Zt_avg1[x0] = v;
          }
          if (i >= Istr && j >= JstrR) {
            const double v = PW(r_DU1, F.DU_avg1[x0]) + cff1 * DUon[s0];
            if (fuse_last) hb_emit(G, B, F.DU_avg1, BC_NONE, i, j, v);
            else F.DU_avg1[x0] = v;
            F.DU_avg2[x0] = PW(r_DU2, F.DU_avg2[x0]) + cff2 * DUon[s0];
          }
          if (i >= IstrR && j >= Jstr) {
            const double v = PW(r_DV1, F.DV_avg1[x0]) + cff1 * DVom[s0];
            if (fuse_last) hb_emit(G, B, F.DV_avg1, BC_NONE, i, j, v);
            else F.DV_avg1[x0] = v;
            F.DV_avg2[x0] = PW(r_DV2, F.DV_avg2[x0]) + cff2 * DVom[s0];
          }
        }
      }
    }
  } else {
    const double cff2 = (iif == 1) ? a.w2_0 : (5.0 / 12.0) * a.w2_0;
    TLOOP(i, j) {
      if (INR(i, j, KMIN(IstrR, Istr), IendR, KMIN(JstrR, Jstr), JendR)) {
        if (i >= Istr && j >= JstrR) F.DU_avg2[x0] = PW(r_DU2, F.DU_avg2[x0]) + cff2 * DUon[s0];
        if (i >= IstrR && j >= Jstr) F.DV_avg2[x0] = PW(r_DV2, F.DV_avg2[x0]) + cff2 * DVom[s0];
      }
    }
  }
  if (last) return;            // auxiliary last predictor call :883 (uniform over the grid)

  // free-surface step :886-1000
  {
    const double fac = 1000.0 / G.rho0;
    double cff1, cff2 = 0.0, cff3 = 0.0, cff4, cff5;
    int mode;
    if (iif == 1) { mode = 0; cff1 = dtfast; cff4 = 0.0; cff5 = 0.0; }
    else if (PRED) { mode = 1; cff1 = 2.0 * dtfast; cff4 = 4.0 / 25.0; cff5 = 1.0 - 2.0 * cff4; }
    else { mode = 2; cff1 = dtfast * 5.0 / 12.0; cff2 = dtfast * 8.0 / 12.0; cff3 = dtfast * 1.0 / 12.0; cff4 = 2.0 / 5.0; cff5 = 1.0 - cff4; }
    TLOOP(i, j) {
      if (INR(i, j, IstrU - 1, Iend, JstrV - 1, Jend)) {
            const double rhs_zeta = (DUon[s0] - DUon[(s0 + 1)]) + (DVom[s0] - DVom[(s0 + TW)]);
        const double zsv = PW(r_zs, zs[x0]), zkv = PW(r_zk, zk[x0]);
        double zeta_new, zw;
        if (mode == 0) {
          zeta_new = zsv + sPm[s0] * sPn[s0] * cff1 * rhs_zeta;
          zw = 0.5 * (zsv + zeta_new);
        } else if (mode == 1) {
          zeta_new = zsv + sPm[s0] * sPn[s0] * cff1 * rhs_zeta;
          zw = cff5 * zkv + cff4 * (zsv + zeta_new);
        } else {
          const double cff = cff1 * rhs_zeta;
          zeta_new = zsv + sPm[s0] * sPn[s0] * (cff + cff2 * PW(r_rz_s, rz_s[x0]) - cff3 * PW(r_rz_p, rz_p[x0]));
          zw = cff5 * zeta_new + cff4 * zkv;
        }
        const double rhoSv = PW(r_rhoS, F.rhoS[x0]);
        Dnew[s0] = zeta_new + sH[s0];
        zwrk[s0] = zw;
        const double gz = (fac + rhoSv) * zw;
        gzeta[s0] = gz;
        gzeta2[s0] = gz * zw;
        gzetaSA[s0] = zw * (rhoSv - sRhoA[s0]);
        if (i >= Istr && j >= Jstr) {
          // zetabc :1057 + exchange :1068, rzeta exchange :1030
          if (fuse) {
            hb_emit(G, B, zn, BC_R, i, j, zeta_new);
            if (PRED) hb_emit(G, B, rz_k, BC_NONE, i, j, rhs_zeta);
          } else {
            zn[x0] = zeta_new;
            if (PRED) rz_k[x0] = rhs_zeta;
          }
        }
      }
    }
  }
  KSYNC();

  // pressure gradient (VAR_RHO_2D) :1080-1200
  {
    const double cff1 = 0.5 * g, cff2 = 1.0 / 3.0;
    TLOOP(i, j) {
      if (INR(i, j, KMIN(IstrU, Istr), Iend, Jstr, Jend)) {
        if (i >= IstrU)
          rhs_ubar[s0] =
              cff1 * PW(r_on_u, F.on_u[x0]) *
              ((sH[(s0 - 1)] + sH[s0]) * (gzeta[(s0 - 1)] - gzeta[s0]) +
               (sH[(s0 - 1)] - sH[s0]) *
                   (gzetaSA[(s0 - 1)] + gzetaSA[s0] +
                    cff2 * (sRhoA[(s0 - 1)] - sRhoA[s0]) * (zwrk[(s0 - 1)] - zwrk[s0])) +
               (gzeta2[(s0 - 1)] - gzeta2[s0]));
        if (j >= JstrV)
          rhs_vbar[s0] =
              cff1 * PW(r_om_v, F.om_v[x0]) *
              ((sH[(s0 - TW)] + sH[s0]) * (gzeta[(s0 - TW)] - gzeta[s0]) +
               (sH[(s0 - TW)] - sH[s0]) *
                   (gzetaSA[(s0 - TW)] + gzetaSA[s0] +
                    cff2 * (sRhoA[(s0 - TW)] - sRhoA[s0]) * (zwrk[(s0 - TW)] - zwrk[s0])) +
               (gzeta2[(s0 - TW)] - gzeta2[s0]));
      }
    }
  }
  KSYNC();   // group-1 scratch is dead from here on (aliased by grad,Dgrad,UFx,UFe)

  if (G.options & ROMS_UV_ADV) {
    const double cff = 1.0 / 6.0;
    // ---- UFx :1249-1290
    TLOOP(i, j) {
      if (INR(i, j, B.IstrUm1, B.Iendp1, Jstr, Jend)) {
        grad[s0] = sUk[(s0 - 1)] - 2.0 * sUk[s0] + sUk[(s0 + 1)];
        Dgrad[s0] = DUon[(s0 - 1)] - 2.0 * DUon[s0] + DUon[(s0 + 1)];
      }
    }
    KSYNC();
    if (!G.ewp) {
      if (B.west) KLOOP1(j, Jstr, Jend) { grad[S2(Istr, j)] = grad[S2(Istr + 1, j)]; Dgrad[S2(Istr, j)] = Dgrad[S2(Istr + 1, j)]; }
      if (B.east) KLOOP1(j, Jstr, Jend) { grad[S2(Iend + 1, j)] = grad[S2(Iend, j)]; Dgrad[S2(Iend + 1, j)] = Dgrad[S2(Iend, j)]; }
      KSYNC();
    }
    TLOOP(i, j) {
      if (INR(i, j, IstrU - 1, Iend, Jstr, Jend))
        UFx[s0] = 0.25 * (sUk[s0] + sUk[(s0 + 1)] - cff * (grad[s0] + grad[(s0 + 1)])) *
                        (DUon[s0] + DUon[(s0 + 1)] - cff * (Dgrad[s0] + Dgrad[(s0 + 1)]));
    }
    KSYNC();
    // ---- UFe :1292-1330
    TLOOP(i, j) {
      if (INR(i, j, IstrU, Iend, B.Jstrm1, B.Jendp1))
        grad[s0] = sUk[(s0 - TW)] - 2.0 * sUk[s0] + sUk[(s0 + TW)];
      if (INR(i, j, IstrU - 1, Iend, Jstr, Jend + 1))
        Dgrad[s0] = DVom[(s0 - 1)] - 2.0 * DVom[s0] + DVom[(s0 + 1)];
    }
    KSYNC();
    if (!G.nsp) {
      if (B.south) KLOOP1(i, IstrU, Iend) grad[S2(i, Jstr - 1)] = grad[S2(i, Jstr)];
      if (B.north) KLOOP1(i, IstrU, Iend) grad[S2(i, Jend + 1)] = grad[S2(i, Jend)];
      KSYNC();
    }
    TLOOP(i, j) {
      if (INR(i, j, IstrU, Iend, Jstr, Jend + 1))
        UFe[s0] = 0.25 * (sUk[s0] + sUk[(s0 - TW)] - cff * (grad[s0] + grad[(s0 - TW)])) *
                        (DVom[s0] + DVom[(s0 - 1)] - cff * (Dgrad[s0] + Dgrad[(s0 - 1)]));
    }
    KSYNC();
    // u-momentum advection r.h.s. (UFx,UFe complete)
    TLOOP(i, j) {
      if (INR(i, j, IstrU, Iend, Jstr, Jend)) {
        const double cff1 = UFx[s0] - UFx[(s0 - 1)];
        const double cff2 = UFe[(s0 + TW)] - UFe[s0];
        const double fac = cff1 + cff2;
        rhs_ubar[s0] = rhs_ubar[s0] - fac;
      }
    }
    KSYNC();
    // ---- VFx :1332-1370
    TLOOP(i, j) {
      if (INR(i, j, B.Istrm1, B.Iendp1, JstrV, Jend))
        grad[s0] = sVk[(s0 - 1)] - 2.0 * sVk[s0] + sVk[(s0 + 1)];
      if (INR(i, j, Istr, Iend + 1, JstrV - 1, Jend))
        Dgrad[s0] = DUon[(s0 - TW)] - 2.0 * DUon[s0] + DUon[(s0 + TW)];
    }
    KSYNC();
    if (!G.ewp) {
      if (B.west) KLOOP1(j, JstrV, Jend) grad[S2(Istr - 1, j)] = grad[S2(Istr, j)];
      if (B.east) KLOOP1(j, JstrV, Jend) grad[S2(Iend + 1, j)] = grad[S2(Iend, j)];
      KSYNC();
    }
    TLOOP(i, j) {
      if (INR(i, j, Istr, Iend + 1, JstrV, Jend))
        VFx[s0] = 0.25 * (sVk[s0] + sVk[(s0 - 1)] - cff * (grad[s0] + grad[(s0 - 1)])) *
                        (DUon[s0] + DUon[(s0 - TW)] - cff * (Dgrad[s0] + Dgrad[(s0 - TW)]));
    }
    KSYNC();
    // ---- VFe :1372-1410
    TLOOP(i, j) {
      if (INR(i, j, Istr, Iend, B.JstrVm1, B.Jendp1)) {
        grad[s0] = sVk[(s0 - TW)] - 2.0 * sVk[s0] + sVk[(s0 + TW)];
        Dgrad[s0] = DVom[(s0 - TW)] - 2.0 * DVom[s0] + DVom[(s0 + TW)];
      }
    }
    KSYNC();
    if (!G.nsp) {
      if (B.south) KLOOP1(i, Istr, Iend) { grad[S2(i, Jstr)] = grad[S2(i, Jstr + 1)]; Dgrad[S2(i, Jstr)] = Dgrad[S2(i, Jstr + 1)]; }
      if (B.north) KLOOP1(i, Istr, Iend) { grad[S2(i, Jend + 1)] = grad[S2(i, Jend)]; Dgrad[S2(i, Jend + 1)] = Dgrad[S2(i, Jend)]; }
      KSYNC();
    }
    TLOOP(i, j) {
      if (INR(i, j, Istr, Iend, JstrV - 1, Jend))
        VFe[s0] = 0.25 * (sVk[s0] + sVk[(s0 + TW)] - cff * (grad[s0] + grad[(s0 + TW)])) *
                        (DVom[s0] + DVom[(s0 + TW)] - cff * (Dgrad[s0] + Dgrad[(s0 + TW)]));
    }
    KSYNC();
    TLOOP(i, j) {
      if (INR(i, j, Istr, Iend, JstrV, Jend)) {
        const double cff1 = VFx[(s0 + 1)] - VFx[s0];
        const double cff2 = VFe[s0] - VFe[(s0 - TW)];
        const double fac = cff1 + cff2;
        rhs_vbar[s0] = rhs_vbar[s0] - fac;
      }
    }
    KSYNC();
  }

  if (G.options & ROMS_UV_COR) {
    // Coriolis :1429-1490
    TLOOP(i, j) {
      if (INR(i, j, IstrU - 1, Iend, JstrV - 1, Jend)) {
        const double cff = 0.5 * Drhs[s0] * PW(r_fomn, F.fomn[x0]);
        UFx[s0] = cff * (sVk[s0] + sVk[(s0 + TW)]);
        VFe[s0] = cff * (sUk[s0] + sUk[(s0 + 1)]);
      }
    }
    KSYNC();
    TLOOP(i, j) {
      if (INR(i, j, KMIN(IstrU, Istr), Iend, KMIN(Jstr, JstrV), Jend)) {
        if (i >= IstrU && j >= Jstr) {
          const double fac1 = 0.5 * (UFx[s0] + UFx[(s0 - 1)]);
          rhs_ubar[s0] = rhs_ubar[s0] + fac1;
        }
        if (i >= Istr && j >= JstrV) {
          const double fac1 = 0.5 * (VFe[s0] + VFe[(s0 - TW)]);
          rhs_vbar[s0] = rhs_vbar[s0] - fac1;
        }
      }
    }
    KSYNC();
  }

  if ((G.options & ROMS_CURVGRID) && (G.options & ROMS_UV_ADV)) {
    // curvilinear metric terms :1494-1560
    TLOOP(i, j) {
      if (INR(i, j, IstrU - 1, Iend, JstrV - 1, Jend)) {
        const double cff1 = 0.5 * (sVk[s0] + sVk[(s0 + TW)]);
        const double cff2 = 0.5 * (sUk[s0] + sUk[(s0 + 1)]);
        const double cff3 = cff1 * PW(r_dndx, F.dndx[x0]);
        const double cff4 = cff2 * PW(r_dmde, F.dmde[x0]);
        const double cff = Drhs[s0] * (cff3 - cff4);
        UFx[s0] = cff * cff1;
        VFe[s0] = cff * cff2;
      }
    }
    KSYNC();
    TLOOP(i, j) {
      if (INR(i, j, KMIN(IstrU, Istr), Iend, KMIN(Jstr, JstrV), Jend)) {
        if (i >= IstrU && j >= Jstr) {
          const double fac1 = 0.5 * (UFx[s0] + UFx[(s0 - 1)]);
          rhs_ubar[s0] = rhs_ubar[s0] + fac1;
        }
        if (i >= Istr && j >= JstrV) {
          const double fac1 = 0.5 * (VFe[s0] + VFe[(s0 - TW)]);
          rhs_vbar[s0] = rhs_vbar[s0] - fac1;
        }
      }
    }
    KSYNC();
  }

  if (G.options & ROMS_UV_VIS2) {
    // harmonic viscosity :1567-1660
    TLOOP(i, j) {
      if (INR(i, j, Istr, Iend + 1, Jstr, Jend + 1))
        Drhs_p[s0] = 0.25 * (Drhs[s0] + Drhs[(s0 - 1)] + Drhs[(s0 - TW)] + Drhs[(s0 - 1 - TW)]);
      if (INR(i, j, IstrU - 1, Iend, JstrV - 1, Jend)) {
        const double cff = PW(r_visc2_r, F.visc2_r[x0]) * Drhs[s0] * 0.5 *
                           (PW(r_pmon_r, F.pmon_r[x0]) * ((sPn[s0] + sPn[(s0 + 1)]) * sUk[(s0 + 1)] -
                                                               (sPn[(s0 - 1)] + sPn[s0]) * sUk[s0]) -
                            PW(r_pnom_r, F.pnom_r[x0]) * ((sPm[s0] + sPm[(s0 + TW)]) * sVk[(s0 + TW)] -
                                                               (sPm[(s0 - TW)] + sPm[s0]) * sVk[s0]));
        UFx[s0] = PW(r_on_r, F.on_r[x0]) * PW(r_on_r, F.on_r[x0]) * cff;
        VFe[s0] = PW(r_om_r, F.om_r[x0]) * PW(r_om_r, F.om_r[x0]) * cff;
      }
    }
    KSYNC();
    TLOOP(i, j) {
      if (INR(i, j, Istr, Iend + 1, Jstr, Jend + 1)) {
        const double cff = PW(r_visc2_p, F.visc2_p[x0]) * Drhs_p[s0] * 0.5 *
                           (PW(r_pmon_p, F.pmon_p[x0]) * ((sPn[(s0 - TW)] + sPn[s0]) * sVk[s0] -
                                                               (sPn[(s0 - 1 - TW)] + sPn[(s0 - 1)]) * sVk[(s0 - 1)]) +
                            PW(r_pnom_p, F.pnom_p[x0]) * ((sPm[(s0 - 1)] + sPm[s0]) * sUk[s0] -
                                                               (sPm[(s0 - 1 - TW)] + sPm[(s0 - TW)]) * sUk[(s0 - TW)]));
        UFe[s0] = PW(r_om_p, F.om_p[x0]) * PW(r_om_p, F.om_p[x0]) * cff;
        VFx[s0] = PW(r_on_p, F.on_p[x0]) * PW(r_on_p, F.on_p[x0]) * cff;
      }
    }
    KSYNC();
    TLOOP(i, j) {
      if (INR(i, j, KMIN(IstrU, Istr), Iend, KMIN(Jstr, JstrV), Jend)) {
        if (i >= IstrU && j >= Jstr) {
          const double cff1 = 0.5 * (sPn[(s0 - 1)] + sPn[s0]) * (UFx[s0] - UFx[(s0 - 1)]);
          const double cff2 = 0.5 * (sPm[(s0 - 1)] + sPm[s0]) * (UFe[(s0 + TW)] - UFe[s0]);
          const double fac = cff1 + cff2;
          rhs_ubar[s0] = rhs_ubar[s0] + fac;
        }
        if (i >= Istr && j >= JstrV) {
          const double cff1 = 0.5 * (sPn[(s0 - TW)] + sPn[s0]) * (VFx[(s0 + 1)] - VFx[s0]);
          const double cff2 = 0.5 * (sPm[(s0 - TW)] + sPm[s0]) * (VFe[s0] - VFe[(s0 - TW)]);
          const double fac = cff1 - cff2;
          rhs_vbar[s0] = rhs_vbar[s0] + fac;
        }
      }
    }
    KSYNC();
  }

  // coupling with the 3-D forcing :2225-2460, then the momentum step :2488-2670 -- point-wise
  {
    const double c1 = (iif == 1) ? 0.5 * dtfast : dtfast;
    const double k1 = 0.5 * dtfast * 5.0 / 12.0, k2 = 0.5 * dtfast * 8.0 / 12.0, k3 = 0.5 * dtfast * 1.0 / 12.0;
    TLOOP(i, j) {
      if (INR(i, j, KMIN(IstrU, Istr), Iend, KMIN(Jstr, JstrV), Jend)) {
        if (i >= IstrU && j >= Jstr) {
          double r = rhs_ubar[s0];
          if (first) {
            const double fr = PW(r_rufrc, F.rufrc[x0]) - r;
            F.rufrc[x0] = fr;
            if (startup == 0) r = r + fr;
            else if (startup == 1) r = r + 1.5 * fr - 0.5 * PW(r_ru0n, ru0_new[x0]);
            else r = r + (23.0 / 12.0) * fr - (16.0 / 12.0) * PW(r_ru0n, ru0_new[x0]) +
                     (5.0 / 12.0) * PW(r_ru0s, ru0_stp[x0]);
            ru0_stp[x0] = fr;
          } else {
            r = r + PW(r_rufrc, F.rufrc[x0]);
          }
          const double cff = (sPm[s0] + sPm[(s0 - 1)]) * (sPn[s0] + sPn[(s0 - 1)]);
          const double fac = 1.0 / (Dnew[s0] + Dnew[(s0 - 1)]);
          const double Dstp_i = sDstp[s0], Dstp_im = sDstp[(s0 - 1)];
          double ub;
          if (!corr) ub = (PW(r_us, us[x0]) * (Dstp_i + Dstp_im) + cff * c1 * r) * fac;
          else ub = (PW(r_us, us[x0]) * (Dstp_i + Dstp_im) +
                     cff * (k1 * r + k2 * PW(r_rub_s, rub_s[x0]) - k3 * PW(r_rub_p, rub_p[x0]))) * fac;
          if (fuse) hb_emit(G, B, un, BC_U, i, j, ub);   // u2dbc :2871 + exchange :3043
          else un[x0] = ub;
          if (PRED) rub_k[x0] = r;
        }
        if (i >= Istr && j >= JstrV) {
          double r = rhs_vbar[s0];
          if (first) {
            const double fr = PW(r_rvfrc, F.rvfrc[x0]) - r;
            F.rvfrc[x0] = fr;
            if (startup == 0) r = r + fr;
            else if (startup == 1) r = r + 1.5 * fr - 0.5 * PW(r_rv0n, rv0_new[x0]);
            else r = r + (23.0 / 12.0) * fr - (16.0 / 12.0) * PW(r_rv0n, rv0_new[x0]) +
                     (5.0 / 12.0) * PW(r_rv0s, rv0_stp[x0]);
            rv0_stp[x0] = fr;
          } else {
            r = r + PW(r_rvfrc, F.rvfrc[x0]);
          }
          const double cff = (sPm[s0] + sPm[(s0 - TW)]) * (sPn[s0] + sPn[(s0 - TW)]);
          const double fac = 1.0 / (Dnew[s0] + Dnew[(s0 - TW)]);
          const double Dstp_j = sDstp[s0], Dstp_jm = sDstp[(s0 - TW)];
          double vb;
          if (!corr) vb = (PW(r_vs, vs[x0]) * (Dstp_j + Dstp_jm) + cff * c1 * r) * fac;
          else vb = (PW(r_vs, vs[x0]) * (Dstp_j + Dstp_jm) +
                     cff * (k1 * r + k2 * PW(r_rvb_s, rvb_s[x0]) - k3 * PW(r_rvb_p, rvb_p[x0]))) * fac;
          if (fuse) hb_emit(G, B, vn, BC_V, i, j, vb);   // v2dbc :2876 + exchange :3043
          else vn[x0] = vb;
          if (PRED) rvb_k[x0] = r;
        }
      }
    }
  }
}
COOP_GLOBAL_LB(k_step2d, Step2dArgs, 512)
#undef TLOOP
#undef PWDECL
#undef PWLOAD
#undef PW
#undef INR
