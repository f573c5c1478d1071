// k_wetdry.h -- WET_DRY: the time-dependent wet/dry masks (ROMS/Nonlinear/wetdry.F).
//
//   k_wetdry   mode 0  wetdry_tile in a fast step (wetdry.F:186-247): the rho mask from zeta(kstp) + h <= Dcrit, the u, v and
//                      psi masks of wetdry_mask_tile (:493-718), the rho mask added to rmask_wet_avg (started at the first
//                      predictor call)
//              mode 1  wetdry_tile behind the last fast step (:250-349): the rho mask = AINT(rmask_wet_avg / (2 nfast)), the
//                      masks of wetdry_avg_mask_tile (:723-900: a face between a wet and a dry cell is open in the direction
//                      of the time-averaged transport DU_avg1 | DV_avg1), then the wet x land masks *_full
//              mode 2  wetdry_ini_tile (:355-490, initial.F:467): the masks in the form of mode 1 from zeta(kstp), with ubar, vbar(kstp)
//                      for the direction of the flow (the SOLVE3D branch :466-472), and *_full
//   k_wd_scale3        ru | rv(:,:,k,nrhs) times umask_wet | vmask_wet (prsgrd32.h:362,426; step3d_uv.F:721,1188)
//
// One thread per point of the ARRAY (ghost points included): a mask at a ghost point is computed from the free surface at
// that ghost point, which the preceding exchange made equal to the neighbour's (or the periodic image's) value -- the same
// numbers the reference's exchange of the masks delivers (wetdry.F:605-623), without one.  A velocity or psi point whose
// lower neighbour lies outside the array takes the periodic image when the tile wraps onto itself, and is left alone
// otherwise (the reference never computes it either: Istr | Jstr is its first point at a wall).
// Pinned through the oracle (oracle/orc_wetdry.c, bit for bit against the reference built from oracle/ref/upwelling_wetdry.h).
#pragma once
#include "roms_ctx.h"
#include "k_diag3d.h"      // KArgs

struct WdArgs {
  DGrid G;
  Fields Fv;
  int mode, init;
};

// PSI-point mask from the four rho values around it (wetdry.F:545-600, 806-861)
KHD double wd_psi(double a, double b, double c, double d) {      // (i-1,j) (i,j) (i-1,j-1) (i,j-1)
  const int nw = (a > 0.5) + (b > 0.5) + (c > 0.5) + (d > 0.5), nd = (a < 0.5) + (b < 0.5) + (c < 0.5) + (d < 0.5);
  if (nw + nd != 4) return 0.0;                                   // (a value of exactly 0.5 matches no branch)
  if (nw >= 3) return 1.0;
  if (nw == 2) {
    // two dry cells that share a side of the point: no-slip (2); diagonal pairs match no branch (0)
    const bool da = a < 0.5, db = b < 0.5, dc = c < 0.5, dd = d < 0.5;
    if ((db && dd) || (da && dc) || (dc && dd) || (da && db)) return 2.0;
  }
  return 0.0;
}

// grid (ni, nj, 1)
THREAD_KERNEL(k_wetdry, WdArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const int i = G.LBi + gx, j = G.LBj + gy;
  if (gx >= G.ni || gy >= G.nj) return;
  {
    // the points an exchange fills, or the boundary line of a wall: Istr-3 .. Iend+Nghost (exchange_2d.F) | Istr-1 .. Iend+1;
    // the array may be wider (the first line with three ghost points, the padding Im > Lm of mod_param.F:1633)
    const TB &B = G.T;
    const int i0 = (B.west && !G.ewp) ? B.Istr - 1 : B.Istr - 3, i1 = (B.east && !G.ewp) ? B.Iend + 1 : B.Iend + G.Nghost;
    const int j0 = (B.south && !G.nsp) ? B.Jstr - 1 : B.Jstr - 3, j1 = (B.north && !G.nsp) ? B.Jend + 1 : B.Jend + G.Nghost;
    if (i < i0 || i > i1 || j < j0 || j > j1) return;
  }
  (void)gz;
  const int mode = a.mode;
  const size_t x = X2(i, j);
  const double *zeta = (const double *)F.zeta + (size_t)(G.kstp - 1) * (size_t)G.nij;
  const double cff = 1.0 / (double)(2 * G.nfast);
  int im1 = i - 1, jm1 = j - 1;
  bool hasx = true, hasy = true;
  {
    const TB &B = G.T;
    const int i0 = (B.west && !G.ewp) ? B.Istr - 1 : B.Istr - 3, j0 = (B.south && !G.nsp) ? B.Jstr - 1 : B.Jstr - 3;
    if (im1 < i0) { if (G.ewp && G.xloc) im1 += G.Lm; else hasx = false; }
    if (jm1 < j0) { if (G.nsp && G.yloc) jm1 += G.Mm; else hasy = false; }
  }
  // the local rho mask :186-204 (mode 1: :250-256)
  auto wd = [&](int ii, int jj) -> double {
    const size_t q = X2(ii, jj);
    if (mode == 1) return trunc(F.rmask_wet_avg[q] * cff);
    double w = 1.0;
    if (G.masking) w = w * F.rmask[q];
    if ((zeta[q] + F.h[q]) <= (G.Dcrit + 1.0E-10)) w = 0.0;
    return w;
  };
  const double w0 = wd(i, j);
  const double wx = hasx ? wd(im1, j) : 0.0, wy = hasy ? wd(i, jm1) : 0.0, wxy = (hasx && hasy) ? wd(im1, jm1) : 0.0;
  if (mode == 0) F.rmask_wet_avg[x] = a.init ? w0 : F.rmask_wet_avg[x] + w0;       // :220-233
  F.rmask_wet[x] = w0;
  if (hasx) {
    double m = wx + w0;
    if (m == 1.0) m = wx - w0;
    if (mode != 0) {                                                              // :765-778
      const double du = mode == 1 ? F.DU_avg1[x] : F.ubar[(size_t)(G.kstp - 1) * (size_t)G.nij + x];
      const double cff5 = fabs(fabs(m) - 1.0), cff6 = 0.5 + copysign(0.5, du) * m;
      m = 0.5 * m * cff5 + cff6 * (1.0 - cff5);
      if (du == 0.0 && (wx + w0) <= 1.0) m = 0.0;                                 // "catch lone ponds"
    }
    F.umask_wet[x] = m;
  }
  if (hasy) {
    double m = wy + w0;
    if (m == 1.0) m = wy - w0;
    if (mode != 0) {                                                              // :784-798
      const double dv = mode == 1 ? F.DV_avg1[x] : F.vbar[(size_t)(G.kstp - 1) * (size_t)G.nij + x];
      const double cff5 = fabs(fabs(m) - 1.0), cff6 = 0.5 + copysign(0.5, dv) * m;
      m = 0.5 * m * cff5 + cff6 * (1.0 - cff5);
      if (dv == 0.0 && (wy + w0) <= 1.0) m = 0.0;
    }
    F.vmask_wet[x] = m;
  }
  if (hasx && hasy) F.pmask_wet[x] = wd_psi(wx, w0, wxy, wy);
  if (mode != 0) {                                                                // :273-296, :428-453
    // (a periodic ghost point of a tile that wraps onto itself: the land mask of its image -- the reference exchanges the
    // products, and its psi land mask is never set on the array's outermost lines, metrics.F:527-583)
    int iw = i, jw = j;
    if (G.ewp && G.xloc) { if (iw < 1) iw += G.Lm; else if (iw > G.Lm) iw -= G.Lm; }
    if (G.nsp && G.yloc) { if (jw < 1) jw += G.Mm; else if (jw > G.Mm) jw -= G.Mm; }
    const size_t xs = X2(iw, jw);
    F.rmask_full[x] = F.rmask_wet[x] * F.rmask[xs];
    if (hasx) F.umask_full[x] = F.umask_wet[x] * F.umask[xs];
    if (hasy) F.vmask_full[x] = F.vmask_wet[x] * F.vmask[xs];
    if (hasx && hasy) { const double v = F.pmask_wet[x] * F.pmask[xs]; F.pmask_full[x] = v > 2.0 ? v : 2.0; }   // MAX(..., 2.0_r8) as written
  }
}
THREAD_GLOBAL(k_wetdry, WdArgs)

// grid (Iend-Istr+1, Jend-Jstr+1, N): ru, rv(i,j,k,nrhs) times the wet mask of the point
THREAD_KERNEL(k_wd_scale3, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int i = B.Istr + gx, j = B.Jstr + gy, k = gz + 1;
  const size_t o = (size_t)(G.nrhs - 1) * G.nij * (size_t)(G.N + 1) + XW(i, j, k), x = X2(i, j);
  if (i >= B.IstrU) F.ru[o] = F.ru[o] * F.umask_wet[x];
  if (j >= B.JstrV) F.rv[o] = F.rv[o] * F.vmask_wet[x];
}
THREAD_GLOBAL(k_wd_scale3, KArgs)

// grid (ni, nj, 1): the land mask times the wet mask at u and v points, for the kernels of step3d_uv, which multiply the new
// velocity by one and then by the other (step3d_uv.F:717-720, 1184-1187, 1400-1405 ...): (a*m)*w == a*(m*w) bit for bit
// for m in {0, 1} (land masks), whatever the finite w
THREAD_KERNEL(k_wd_eff, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  if (gx >= G.ni || gy >= G.nj) return;
  (void)gz;
  const size_t x = (size_t)gx + (size_t)gy * (size_t)G.ni;
  F.wd_eff[x] = F.umask[x] * F.umask_wet[x];
  F.wd_eff[(size_t)G.nij + x] = F.vmask[x] * F.vmask_wet[x];
}
THREAD_GLOBAL(k_wd_eff, KArgs)
