// k_uvmix_geo.h -- harmonic viscosity along geopotential surfaces (UV_VIS2 + MIX_GEO_UV): the rotated stress tensor of
//   uv3dmix2_geo_tile   ROMS/Nonlinear/uv3dmix2_geo.h:130-757
// The reference rolls two levels of fourteen 2-D work arrays through its K_LOOP.  Here every one of them is a 3-D work array
// (Fields::gwrk: 20 arrays of N+1 planes, allocated with ROMS_MIX_GEO_UV) and every loop nest of the routine is a point-wise
// kernel over its own index range for all levels at once -- the statements are the reference's, array for array:
//   k_uvg_slopes   :301-326 (slopes at u-, v-points), :415-458 (dUdz, dVdz at W-levels)
//   k_uvg_grads    :328-412 (slopes at psi-, rho-points; the four horizontal gradients)
//   k_uvg_flux     :464-543 (rotated fluxes UFx, VFe at rho-points, UFe, VFx at psi-points)
//   k_uvg_vflux    :548-690 (vertical fluxes due to the sloping surfaces, W-levels 1..N-1)
//   k_uvg_step     :693-740 (time step: rufrc, rvfrc summed bottom to top in the reference's order, u,v(nnew))
// A rho-type array holds level k in plane k (the reference's (:,:,k1) at loop index k, (:,:,k2) = plane k+1); a W-type array
// (dUdz, dVdz, UFs*) holds W-level k in plane k ((:,:,k1) = plane k-1, (:,:,k2) = plane k), planes 0 and N zero.
// No VISC_3DCOEF, no DIAGNOSTICS_UV statements (refused together).  Not a performance form: twenty work arrays where a
// marching kernel would keep two levels in registers -- none of BASELINE's configurations uses MIX_GEO_UV.
#pragma once
#include "roms_ctx.h"
#include "k_diag3d.h"

// Round 6: the BIHARMONIC form, uv3dmix4_geo_tile (uv3dmix4_geo.h:296-1478; UV_VIS4 + MIX_GEO_UV), is the same operator twice -- every
// kernel takes the mode in a.p0:
//   0  harmonic (above);
//   1  the FIRST operator :323-801: the same statements on the ranges widened by one point (the reference's Istrm1 ... Iendp2,
//      clamped at the domain edges: UgR below), no thickness in the horizontal fluxes, visc4 (the square root of the coefficient),
//      and the operator itself -- LapU, LapV, two more work arrays -- as its result (k_uvg_step); then k_uvg_lapbc, the closed /
//      gradient conditions and corner averages of :803-975 on LapU, LapV;
//   2  the SECOND operator :982-1476: the harmonic statements on LapU, LapV with visc4, subtracted from the sums and u, v(nnew).
enum { UG_SU = 0, UG_SV, UG_ZXP, UG_ZEP, UG_ZXR, UG_ZER, UG_NUX, UG_MUE, UG_NVX, UG_MVE, UG_UZ, UG_VZ, UG_UFX, UG_VFE, UG_UFE,
       UG_VFX, UG_USX, UG_USE, UG_VSX, UG_VSE, UG_LAPU, UG_LAPV, UG_NARR };
// loop bounds of the harmonic routine ("X-1" | "X" below, "Y+1" | "Y" above) and what the first biharmonic operator has in their place
struct UgR { int jS1, jS0, jV1, jV0, iS1, iS0, iU1, iU0, jE1, jE0, iE1, iE0; };
KHD UgR ug_ranges(const TB &B, int mode) {
  UgR r;
  const bool x = mode == 1;
  r.jS1 = x ? B.Jstrm2 : B.Jstr - 1; r.jS0 = x ? B.Jstrm1 : B.Jstr; r.jV1 = x ? B.JstrVm2 : B.JstrV - 1; r.jV0 = x ? B.JstrVm1 : B.JstrV;
  r.iS1 = x ? B.Istrm2 : B.Istr - 1; r.iS0 = x ? B.Istrm1 : B.Istr; r.iU1 = x ? B.IstrUm2 : B.IstrU - 1; r.iU0 = x ? B.IstrUm1 : B.IstrU;
  r.jE1 = x ? B.Jendp2 : B.Jend + 1; r.jE0 = x ? B.Jendp1 : B.Jend; r.iE1 = x ? B.Iendp2 : B.Iend + 1; r.iE0 = x ? B.Iendp1 : B.Iend;
  return r;
}
// the fields the operator acts on (level k at X3(i,j,k)): u, v(nrhs) | LapU, LapV
#define UG_INPUTS                                                                                                              \
  const int mode = a.p0;                                                                                                      \
  const UgR R = ug_ranges(B, mode);                                                                                           \
  const double *u = mode == 2 ? (const double *)UGA(UG_LAPU) : F.u + (size_t)(G.nrhs - 1) * nij * (size_t)N;                  \
  const double *v = mode == 2 ? (const double *)UGA(UG_LAPV) : F.v + (size_t)(G.nrhs - 1) * nij * (size_t)N;                  \
  const double *vis_r = mode ? (const double *)F.visc4_r : (const double *)F.visc2_r;                                        \
  const double *vis_p = mode ? (const double *)F.visc4_p : (const double *)F.visc2_p;                                        \
  (void)u; (void)v; (void)vis_r; (void)vis_p; (void)R
#define UGA(a_) (F.gwrk + (size_t)(a_) * nijw)
#define UGP(a_, i_, j_, k_) UGA(a_)[X2(i_, j_) + (size_t)(k_) * nij]

// index space (min(IstrU,Istr)-1 : Iend+1, min(Jstr,JstrV)-1 : Jend+1, 0:N)
THREAD_KERNEL(k_uvg_slopes, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, k = gz;
  const size_t nij = (size_t)G.nij, nijw = nij * (size_t)(N + 1);
  UG_INPUTS;
  const int i = KMIN(R.iU1, R.iS1) + gx, j = KMIN(R.jS1, R.jV1) + gy;
  const double *z_r = F.z_r, *pm = F.pm, *pn = F.pn;
  const bool inU = i >= R.iU1 && i <= R.iE1 && j >= R.jS1 && j <= R.jE1;
  const bool inV = i >= R.iS1 && i <= R.iE1 && j >= R.jV1 && j <= R.jE1;
  if (inU) {
    if (k >= 1) {                                              // :301-313 (the reference's level k+1 at loop index k)
      double cff = 0.5 * (pm[X2(i - 1, j)] + pm[X2(i, j)]);
      if (G.masking) cff = cff * F.umask[X2(i, j)];
      if (G.wet_dry) cff = cff * F.umask_wet[X2(i, j)];
      UGP(UG_SU, i, j, k) = cff * (z_r[X3(i, j, k)] - z_r[X3(i - 1, j, k)]);
    }
    if (k == 0 || k == N) UGP(UG_UZ, i, j, k) = 0.0;           // :415-420
    else {                                                     // :439-447
      const double cff = 1.0 / (0.5 * (z_r[X3(i - 1, j, k + 1)] - z_r[X3(i - 1, j, k)] + z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)]));
      UGP(UG_UZ, i, j, k) = cff * (u[X3(i, j, k + 1)] - u[X3(i, j, k)]);
    }
  }
  if (inV) {
    if (k >= 1) {                                              // :314-326
      double cff = 0.5 * (pn[X2(i, j - 1)] + pn[X2(i, j)]);
      if (G.masking) cff = cff * F.vmask[X2(i, j)];
      if (G.wet_dry) cff = cff * F.vmask_wet[X2(i, j)];
      UGP(UG_SV, i, j, k) = cff * (z_r[X3(i, j, k)] - z_r[X3(i, j - 1, k)]);
    }
    if (k == 0 || k == N) UGP(UG_VZ, i, j, k) = 0.0;
    else {                                                     // :449-457
      const double cff = 1.0 / (0.5 * (z_r[X3(i, j - 1, k + 1)] - z_r[X3(i, j - 1, k)] + z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)]));
      UGP(UG_VZ, i, j, k) = cff * (v[X3(i, j, k + 1)] - v[X3(i, j, k)]);
    }
  }
}
THREAD_GLOBAL(k_uvg_slopes, KArgs)

// index space (min(IstrU-1,Istr) : Iend+1, min(JstrV-1,Jstr) : Jend+1, 1:N)
THREAD_KERNEL(k_uvg_grads, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, k = gz + 1;
  const size_t nij = (size_t)G.nij, nijw = nij * (size_t)(N + 1);
  UG_INPUTS;
  const int i = KMIN(R.iU1, R.iS0) + gx, j = KMIN(R.jV1, R.jS0) + gy;
  const double *pm = F.pm, *pn = F.pn;
  const bool inP = i >= R.iS0 && i <= R.iE1 && j >= R.jS0 && j <= R.jE1;
  const bool inR = i >= R.iU1 && i <= R.iE0 && j >= R.jV1 && j <= R.jE0;
  if (inP) {
    UGP(UG_ZXP, i, j, k) = 0.5 * (UGP(UG_SU, i, j - 1, k) + UGP(UG_SU, i, j, k));          // :328-334
    UGP(UG_ZEP, i, j, k) = 0.5 * (UGP(UG_SV, i - 1, j, k) + UGP(UG_SV, i, j, k));
    double cff = 0.125 * (pn[X2(i - 1, j)] + pn[X2(i, j)] + pn[X2(i - 1, j - 1)] + pn[X2(i, j - 1)]);      // :363-378
    if (G.masking) cff = cff * F.pmask[X2(i, j)];
    if (G.wet_dry) cff = cff * F.pmask_wet[X2(i, j)];
    UGP(UG_MUE, i, j, k) = cff * ((pm[X2(i - 1, j)] + pm[X2(i, j)]) * u[X3(i, j, k)] - (pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]) * u[X3(i, j - 1, k)]);
    cff = 0.125 * (pm[X2(i - 1, j)] + pm[X2(i, j)] + pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]);             // :380-395
    if (G.masking) cff = cff * F.pmask[X2(i, j)];
    if (G.wet_dry) cff = cff * F.pmask_wet[X2(i, j)];
    UGP(UG_NVX, i, j, k) = cff * ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * v[X3(i, j, k)] - (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)]) * v[X3(i - 1, j, k)]);
  }
  if (inR) {
    UGP(UG_ZXR, i, j, k) = 0.5 * (UGP(UG_SU, i, j, k) + UGP(UG_SU, i + 1, j, k));          // :335-342
    UGP(UG_ZER, i, j, k) = 0.5 * (UGP(UG_SV, i, j, k) + UGP(UG_SV, i, j + 1, k));
    double cff = 0.5 * pm[X2(i, j)];                                                       // :346-361
    if (G.masking) cff = cff * F.rmask[X2(i, j)];
    if (G.wet_dry) cff = cff * F.rmask_wet[X2(i, j)];
    UGP(UG_NUX, i, j, k) = cff * ((pn[X2(i, j)] + pn[X2(i + 1, j)]) * u[X3(i + 1, j, k)] - (pn[X2(i - 1, j)] + pn[X2(i, j)]) * u[X3(i, j, k)]);
    cff = 0.5 * pn[X2(i, j)];                                                              // :397-412
    if (G.masking) cff = cff * F.rmask[X2(i, j)];
    if (G.wet_dry) cff = cff * F.rmask_wet[X2(i, j)];
    UGP(UG_MVE, i, j, k) = cff * ((pm[X2(i, j)] + pm[X2(i, j + 1)]) * v[X3(i, j + 1, k)] - (pm[X2(i, j - 1)] + pm[X2(i, j)]) * v[X3(i, j, k)]);
  }
}
THREAD_GLOBAL(k_uvg_grads, KArgs)

// index space (min(IstrU-1,Istr) : Iend+1, min(JstrV-1,Jstr) : Jend+1, 1:N)
THREAD_KERNEL(k_uvg_flux, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, k = gz + 1;
  const size_t nij = (size_t)G.nij, nijw = nij * (size_t)(N + 1);
  UG_INPUTS;
  const int i = KMIN(R.iU1, R.iS0) + gx, j = KMIN(R.jV1, R.jS0) + gy;
  const double *pm = F.pm, *pn = F.pn, *Hz = F.Hz;
  if (i >= R.iU1 && i <= R.iE0 && j >= R.jV1 && j <= R.jE0) {                              // :464-497
    const double cff1 = KMIN(UGP(UG_ZXR, i, j, k), 0.0), cff2 = KMAX(UGP(UG_ZXR, i, j, k), 0.0);
    const double cff3 = KMIN(UGP(UG_ZER, i, j, k), 0.0), cff4 = KMAX(UGP(UG_ZER, i, j, k), 0.0);
    double cff = (F.on_r[X2(i, j)] * (UGP(UG_NUX, i, j, k) -
                                      0.5 * pn[X2(i, j)] * (cff1 * (UGP(UG_UZ, i, j, k - 1) + UGP(UG_UZ, i + 1, j, k)) +
                                                             cff2 * (UGP(UG_UZ, i, j, k) + UGP(UG_UZ, i + 1, j, k - 1)))) -
                  F.om_r[X2(i, j)] * (UGP(UG_MVE, i, j, k) -
                                      0.5 * pm[X2(i, j)] * (cff3 * (UGP(UG_VZ, i, j, k - 1) + UGP(UG_VZ, i, j + 1, k)) +
                                                             cff4 * (UGP(UG_VZ, i, j, k) + UGP(UG_VZ, i, j + 1, k - 1)))));
    if (mode != 1) cff = Hz[X3(i, j, k)] * cff;                  // (the first biharmonic operator: no thickness, uv3dmix4_geo.h:501)
    if (G.masking) cff = cff * F.rmask[X2(i, j)];
    if (G.wet_dry) cff = cff * F.rmask_wet[X2(i, j)];
    UGP(UG_UFX, i, j, k) = F.on_r[X2(i, j)] * F.on_r[X2(i, j)] * vis_r[X2(i, j)] * cff;
    UGP(UG_VFE, i, j, k) = F.om_r[X2(i, j)] * F.om_r[X2(i, j)] * vis_r[X2(i, j)] * cff;
  }
  if (i >= R.iS0 && i <= R.iE1 && j >= R.jS0 && j <= R.jE1) {                              // :499-543
    const double pm_p = 0.25 * (pm[X2(i - 1, j - 1)] + pm[X2(i - 1, j)] + pm[X2(i, j - 1)] + pm[X2(i, j)]);
    const double pn_p = 0.25 * (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)] + pn[X2(i, j - 1)] + pn[X2(i, j)]);
    const double cff1 = KMIN(UGP(UG_ZXP, i, j, k), 0.0), cff2 = KMAX(UGP(UG_ZXP, i, j, k), 0.0);
    const double cff3 = KMIN(UGP(UG_ZEP, i, j, k), 0.0), cff4 = KMAX(UGP(UG_ZEP, i, j, k), 0.0);
    double cff = (F.on_p[X2(i, j)] * (UGP(UG_NVX, i, j, k) -
                                      0.5 * pn_p * (cff1 * (UGP(UG_VZ, i - 1, j, k - 1) + UGP(UG_VZ, i, j, k)) +
                                                    cff2 * (UGP(UG_VZ, i - 1, j, k) + UGP(UG_VZ, i, j, k - 1)))) +
                  F.om_p[X2(i, j)] * (UGP(UG_MUE, i, j, k) -
                                      0.5 * pm_p * (cff3 * (UGP(UG_UZ, i, j - 1, k - 1) + UGP(UG_UZ, i, j, k)) +
                                                    cff4 * (UGP(UG_UZ, i, j - 1, k) + UGP(UG_UZ, i, j, k - 1)))));
    if (mode != 1) cff = 0.25 * (Hz[X3(i - 1, j, k)] + Hz[X3(i, j, k)] + Hz[X3(i - 1, j - 1, k)] + Hz[X3(i, j - 1, k)]) * cff;   // (:540)
    if (G.masking) cff = cff * F.pmask[X2(i, j)];
    if (G.wet_dry) cff = cff * F.pmask_wet[X2(i, j)];
    UGP(UG_UFE, i, j, k) = F.om_p[X2(i, j)] * F.om_p[X2(i, j)] * vis_p[X2(i, j)] * cff;
    UGP(UG_VFX, i, j, k) = F.on_p[X2(i, j)] * F.on_p[X2(i, j)] * vis_p[X2(i, j)] * cff;
  }
}
THREAD_GLOBAL(k_uvg_flux, KArgs)

// the four-term brackets of :577-690: A at (k1: plane k) and (k2: plane k+1), B the gradient array, C the vertical gradient
#define UG_BR4(c1_, c2_, c3_, c4_, d1_, d2_, d3_, d4_, dz_, g1_, g2_, g3_, g4_) \
  ((c1_) * ((d1_) * (dz_) - (g1_)) + (c2_) * ((d2_) * (dz_) - (g2_)) + (c3_) * ((d3_) * (dz_) - (g3_)) + (c4_) * ((d4_) * (dz_) - (g4_)))

// index space (Istr:Iend, Jstr:Jend, 0:N) -- W-levels; planes 0 and N are zero (:426-437)
THREAD_KERNEL(k_uvg_vflux, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, k = gz;
  const size_t nij = (size_t)G.nij, nijw = nij * (size_t)(N + 1);
  UG_INPUTS;
  const int i = R.iS0 + gx, j = R.jS0 + gy;
  if (i > R.iE0 || j > R.jE0) return;
  const double *pm = F.pm, *pn = F.pn;
  if (i >= R.iU0) {
    if (k == 0 || k == N) { UGP(UG_USX, i, j, k) = 0.0; UGP(UG_USE, i, j, k) = 0.0; }
    else {                                                                                 // :549-612
      double cff = 0.25 * (vis_r[X2(i - 1, j)] + vis_r[X2(i, j)]);
      const double fac1 = cff * F.on_u[X2(i, j)], fac2 = cff * F.om_u[X2(i, j)];
      const double vz4 = UGP(UG_VZ, i - 1, j + 1, k) + UGP(UG_VZ, i, j + 1, k) + UGP(UG_VZ, i - 1, j, k) + UGP(UG_VZ, i, j, k);
      cff = 0.5 * (pn[X2(i - 1, j)] + pn[X2(i, j)]);
      const double dnUdz = cff * UGP(UG_UZ, i, j, k);
      const double dnVdz = cff * 0.25 * vz4;
      cff = 0.5 * (pm[X2(i - 1, j)] + pm[X2(i, j)]);
      const double dmUdz = cff * UGP(UG_UZ, i, j, k);
      const double dmVdz = cff * 0.25 * vz4;
      double cff1 = KMIN(UGP(UG_ZXR, i - 1, j, k), 0.0), cff2 = KMIN(UGP(UG_ZXR, i, j, k + 1), 0.0);
      double cff3 = KMAX(UGP(UG_ZXR, i - 1, j, k + 1), 0.0), cff4 = KMAX(UGP(UG_ZXR, i, j, k), 0.0);
      double usx = fac1 * UG_BR4(cff1, cff2, cff3, cff4, cff1, cff2, cff3, cff4, dnUdz, UGP(UG_NUX, i - 1, j, k), UGP(UG_NUX, i, j, k + 1),
                                 UGP(UG_NUX, i - 1, j, k + 1), UGP(UG_NUX, i, j, k));
      cff1 = KMIN(UGP(UG_ZEP, i, j, k), 0.0); cff2 = KMIN(UGP(UG_ZEP, i, j + 1, k + 1), 0.0);
      cff3 = KMAX(UGP(UG_ZEP, i, j, k + 1), 0.0); cff4 = KMAX(UGP(UG_ZEP, i, j + 1, k), 0.0);
      double use = fac2 * UG_BR4(cff1, cff2, cff3, cff4, cff1, cff2, cff3, cff4, dmUdz, UGP(UG_MUE, i, j, k), UGP(UG_MUE, i, j + 1, k + 1),
                                 UGP(UG_MUE, i, j, k + 1), UGP(UG_MUE, i, j + 1, k));
      {
        const double cff5 = KMIN(UGP(UG_ZXP, i, j, k), 0.0), cff6 = KMIN(UGP(UG_ZXP, i, j + 1, k + 1), 0.0);
        const double cff7 = KMAX(UGP(UG_ZXP, i, j, k + 1), 0.0), cff8 = KMAX(UGP(UG_ZXP, i, j + 1, k), 0.0);
        usx = usx + fac1 * UG_BR4(cff1, cff2, cff3, cff4, cff5, cff6, cff7, cff8, dnVdz, UGP(UG_NVX, i, j, k), UGP(UG_NVX, i, j + 1, k + 1),
                                  UGP(UG_NVX, i, j, k + 1), UGP(UG_NVX, i, j + 1, k));
      }
      cff1 = KMIN(UGP(UG_ZXR, i - 1, j, k), 0.0); cff2 = KMIN(UGP(UG_ZXR, i, j, k + 1), 0.0);
      cff3 = KMAX(UGP(UG_ZXR, i - 1, j, k + 1), 0.0); cff4 = KMAX(UGP(UG_ZXR, i, j, k), 0.0);
      {
        const double cff5 = KMIN(UGP(UG_ZER, i - 1, j, k), 0.0), cff6 = KMIN(UGP(UG_ZER, i, j, k + 1), 0.0);
        const double cff7 = KMAX(UGP(UG_ZER, i - 1, j, k + 1), 0.0), cff8 = KMAX(UGP(UG_ZER, i, j, k), 0.0);
        use = use - fac2 * UG_BR4(cff1, cff2, cff3, cff4, cff5, cff6, cff7, cff8, dmVdz, UGP(UG_MVE, i - 1, j, k), UGP(UG_MVE, i, j, k + 1),
                                  UGP(UG_MVE, i - 1, j, k + 1), UGP(UG_MVE, i, j, k));
      }
      UGP(UG_USX, i, j, k) = usx;
      UGP(UG_USE, i, j, k) = use;
    }
  }
  if (j >= R.jV0) {
    if (k == 0 || k == N) { UGP(UG_VSX, i, j, k) = 0.0; UGP(UG_VSE, i, j, k) = 0.0; }
    else {                                                                                 // :615-688
      double cff = 0.25 * (vis_r[X2(i, j - 1)] + vis_r[X2(i, j)]);
      const double fac1 = cff * F.on_v[X2(i, j)], fac2 = cff * F.om_v[X2(i, j)];
      const double uz4 = UGP(UG_UZ, i, j, k) + UGP(UG_UZ, i + 1, j, k) + UGP(UG_UZ, i, j - 1, k) + UGP(UG_UZ, i + 1, j - 1, k);
      cff = 0.5 * (pn[X2(i, j - 1)] + pn[X2(i, j)]);
      const double dnUdz = cff * 0.25 * uz4;
      const double dnVdz = cff * UGP(UG_VZ, i, j, k);
      cff = 0.5 * (pm[X2(i, j - 1)] + pm[X2(i, j)]);
      const double dmUdz = cff * 0.25 * uz4;
      const double dmVdz = cff * UGP(UG_VZ, i, j, k);
      double cff1 = KMIN(UGP(UG_ZXP, i, j, k), 0.0), cff2 = KMIN(UGP(UG_ZXP, i + 1, j, k + 1), 0.0);
      double cff3 = KMAX(UGP(UG_ZXP, i, j, k + 1), 0.0), cff4 = KMAX(UGP(UG_ZXP, i + 1, j, k), 0.0);
      double vsx = fac1 * UG_BR4(cff1, cff2, cff3, cff4, cff1, cff2, cff3, cff4, dnVdz, UGP(UG_NVX, i, j, k), UGP(UG_NVX, i + 1, j, k + 1),
                                 UGP(UG_NVX, i, j, k + 1), UGP(UG_NVX, i + 1, j, k));
      cff1 = KMIN(UGP(UG_ZER, i, j - 1, k), 0.0); cff2 = KMIN(UGP(UG_ZER, i, j, k + 1), 0.0);
      cff3 = KMAX(UGP(UG_ZER, i, j - 1, k + 1), 0.0); cff4 = KMAX(UGP(UG_ZER, i, j, k), 0.0);
      double vse = fac2 * UG_BR4(cff1, cff2, cff3, cff4, cff1, cff2, cff3, cff4, dmVdz, UGP(UG_MVE, i, j - 1, k), UGP(UG_MVE, i, j, k + 1),
                                 UGP(UG_MVE, i, j - 1, k + 1), UGP(UG_MVE, i, j, k));
      {
        const double cff5 = KMIN(UGP(UG_ZXR, i, j - 1, k), 0.0), cff6 = KMIN(UGP(UG_ZXR, i, j, k + 1), 0.0);
        const double cff7 = KMAX(UGP(UG_ZXR, i, j - 1, k + 1), 0.0), cff8 = KMAX(UGP(UG_ZXR, i, j, k), 0.0);
        vsx = vsx - fac1 * UG_BR4(cff1, cff2, cff3, cff4, cff5, cff6, cff7, cff8, dnUdz, UGP(UG_NUX, i, j - 1, k), UGP(UG_NUX, i, j, k + 1),
                                  UGP(UG_NUX, i, j - 1, k + 1), UGP(UG_NUX, i, j, k));
      }
      cff1 = KMIN(UGP(UG_ZXP, i, j, k), 0.0); cff2 = KMIN(UGP(UG_ZXP, i + 1, j, k + 1), 0.0);
      cff3 = KMAX(UGP(UG_ZXP, i, j, k + 1), 0.0); cff4 = KMAX(UGP(UG_ZXP, i + 1, j, k), 0.0);
      {
        const double cff5 = KMIN(UGP(UG_ZEP, i, j, k), 0.0), cff6 = KMIN(UGP(UG_ZEP, i + 1, j, k + 1), 0.0);
        const double cff7 = KMAX(UGP(UG_ZEP, i, j, k + 1), 0.0), cff8 = KMAX(UGP(UG_ZEP, i + 1, j, k), 0.0);
        vse = vse + fac2 * UG_BR4(cff1, cff2, cff3, cff4, cff5, cff6, cff7, cff8, dmUdz, UGP(UG_MUE, i, j, k), UGP(UG_MUE, i + 1, j, k + 1),
                                  UGP(UG_MUE, i, j, k + 1), UGP(UG_MUE, i + 1, j, k));
      }
      UGP(UG_VSX, i, j, k) = vsx;
      UGP(UG_VSE, i, j, k) = vse;
    }
  }
}
THREAD_GLOBAL(k_uvg_vflux, KArgs)

// index space (Istr:Iend, Jstr:Jend): one thread per column, bottom to top (:693-740)
THREAD_KERNEL(k_uvg_step, KArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, nnew = G.nnew;
  const size_t nij = (size_t)G.nij, nijw = nij * (size_t)(N + 1);
  UG_INPUTS;
  const int i = R.iS0 + gx, j = R.jS0 + gy;
  if (i > R.iE0 || j > R.jE0) return;
  const double *pm = F.pm, *pn = F.pn;
  const double dt = G.dt;
  double *un = F.u + (size_t)(nnew - 1) * nij * (size_t)N, *vn = F.v + (size_t)(nnew - 1) * nij * (size_t)N;
  if (mode == 1) {                                           // the first operator itself, uv3dmix4_geo.h:759-800
    const double *Hz = F.Hz;
    double *LU = UGA(UG_LAPU), *LV = UGA(UG_LAPV);
    if (i >= R.iU0) {
      const double cff = 0.125 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (pn[X2(i - 1, j)] + pn[X2(i, j)]);
      for (int k = 1; k <= N; k++) {
        const double cff1 = 1.0 / (0.5 * (Hz[X3(i - 1, j, k)] + Hz[X3(i, j, k)]));
        double l = cff * ((pn[X2(i - 1, j)] + pn[X2(i, j)]) * (UGP(UG_UFX, i, j, k) - UGP(UG_UFX, i - 1, j, k)) +
                          (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (UGP(UG_UFE, i, j + 1, k) - UGP(UG_UFE, i, j, k))) +
                   cff1 * ((UGP(UG_USX, i, j, k) + UGP(UG_USE, i, j, k)) - (UGP(UG_USX, i, j, k - 1) + UGP(UG_USE, i, j, k - 1)));
        if (G.masking) l = l * F.umask[X2(i, j)];
        if (G.wet_dry) l = l * F.umask_wet[X2(i, j)];
        LU[X3(i, j, k)] = l;
      }
    }
    if (j >= R.jV0) {
      const double cff = 0.125 * (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
      for (int k = 1; k <= N; k++) {
        const double cff1 = 1.0 / (0.5 * (Hz[X3(i, j - 1, k)] + Hz[X3(i, j, k)]));
        double l = cff * ((pn[X2(i, j - 1)] + pn[X2(i, j)]) * (UGP(UG_VFX, i + 1, j, k) - UGP(UG_VFX, i, j, k)) -
                          (pm[X2(i, j - 1)] + pm[X2(i, j)]) * (UGP(UG_VFE, i, j, k) - UGP(UG_VFE, i, j - 1, k))) +
                   cff1 * ((UGP(UG_VSX, i, j, k) + UGP(UG_VSE, i, j, k)) - (UGP(UG_VSX, i, j, k - 1) + UGP(UG_VSE, i, j, k - 1)));
        if (G.masking) l = l * F.vmask[X2(i, j)];
        if (G.wet_dry) l = l * F.vmask_wet[X2(i, j)];
        LV[X3(i, j, k)] = l;
      }
    }
    return;
  }
  const bool sub = mode == 2;                                // (the second biharmonic operator is subtracted, :1437-1462)
  if (i >= R.iU0) {
    const double cff = dt * 0.25 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (pn[X2(i - 1, j)] + pn[X2(i, j)]);
    double rf = F.rufrc[X2(i, j)];
    for (int k = 1; k <= N; k++) {
      const double cff1 = 0.5 * (pn[X2(i - 1, j)] + pn[X2(i, j)]) * (UGP(UG_UFX, i, j, k) - UGP(UG_UFX, i - 1, j, k));
      const double cff2 = 0.5 * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (UGP(UG_UFE, i, j + 1, k) - UGP(UG_UFE, i, j, k));
      const double cff3 = UGP(UG_USX, i, j, k) - UGP(UG_USX, i, j, k - 1);
      const double cff4 = UGP(UG_USE, i, j, k) - UGP(UG_USE, i, j, k - 1);
      const double cff5 = cff * (cff1 + cff2);
      const double cff6 = dt * (cff3 + cff4);
      if (sub) { rf = rf - cff1 - cff2 - cff3 - cff4; un[X3(i, j, k)] = un[X3(i, j, k)] - cff5 - cff6; }
      else { rf = rf + cff1 + cff2 + cff3 + cff4; un[X3(i, j, k)] = un[X3(i, j, k)] + cff5 + cff6; }
    }
    F.rufrc[X2(i, j)] = rf;
  }
  if (j >= R.jV0) {
    const double cff = dt * 0.25 * (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]);
    double rf = F.rvfrc[X2(i, j)];
    for (int k = 1; k <= N; k++) {
      const double cff1 = 0.5 * (pn[X2(i, j - 1)] + pn[X2(i, j)]) * (UGP(UG_VFX, i + 1, j, k) - UGP(UG_VFX, i, j, k));
      const double cff2 = 0.5 * (pm[X2(i, j - 1)] + pm[X2(i, j)]) * (UGP(UG_VFE, i, j, k) - UGP(UG_VFE, i, j - 1, k));
      const double cff3 = UGP(UG_VSX, i, j, k) - UGP(UG_VSX, i, j, k - 1);
      const double cff4 = UGP(UG_VSE, i, j, k) - UGP(UG_VSE, i, j, k - 1);
      const double cff5 = cff * (cff1 - cff2);
      const double cff6 = dt * (cff3 + cff4);
      if (sub) { rf = rf - cff1 + cff2 - cff3 - cff4; vn[X3(i, j, k)] = vn[X3(i, j, k)] - cff5 - cff6; }
      else { rf = rf + cff1 - cff2 + cff3 + cff4; vn[X3(i, j, k)] = vn[X3(i, j, k)] + cff5 + cff6; }
    }
    F.rvfrc[X2(i, j)] = rf;
  }
}
THREAD_GLOBAL(k_uvg_step, KArgs)

// ---- the conditions on the first biharmonic operator, uv3dmix4_geo.h:803-975 (the statements of uv3dmix4_s.h:335-524): a.p1 = 0 the
// four edges (no statement reads what another one writes), a.p1 = 1 the corner averages of a closed basin behind them.
// index space (Istrm1-1 : Iendp1+1, Jstrm1-1 : Jendp1+1, 1:N): every thread looks whether its point is a target
THREAD_KERNEL(k_uvg_lapbc, KArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, k = gz + 1;
  const size_t nij = (size_t)G.nij, nijw = nij * (size_t)(N + 1);
  const int i = B.Istrm1 - 1 + gx, j = B.Jstrm1 - 1 + gy;
  double *LU = UGA(UG_LAPU), *LV = UGA(UG_LAPV);
  const int clu = (int)((G.lbc_closed >> (4 * ROMS_ISUVEL)) & 15ull), clv = (int)((G.lbc_closed >> (4 * ROMS_ISVVEL)) & 15ull);
  const double g2 = G.gamma2;
#define LU_(ii, jj) LU[X3(ii, jj, k)]
#define LV_(ii, jj) LV[X3(ii, jj, k)]
  if (a.p1 == 0) {
    if (!G.ewp) {
      if (B.west) {
        if (i == B.Istr && j >= B.Jstrm1 && j <= B.Jendp1) LU_(i, j) = (clu & (1 << ROMS_IWEST)) ? 0.0 : LU_(B.Istr + 1, j);          // LapU(IstrU-1,j)
        if (i == B.Istr - 1 && j >= B.JstrVm1 && j <= B.Jendp1) LV_(i, j) = (clv & (1 << ROMS_IWEST)) ? g2 * LV_(B.Istr, j) : 0.0;
      }
      if (B.east && i == B.Iend + 1) {
        if (j >= B.Jstrm1 && j <= B.Jendp1) LU_(i, j) = (clu & (1 << ROMS_IEAST)) ? 0.0 : LU_(B.Iend, j);
        if (j >= B.JstrVm1 && j <= B.Jendp1) LV_(i, j) = (clv & (1 << ROMS_IEAST)) ? g2 * LV_(B.Iend, j) : 0.0;
      }
    }
    if (!G.nsp) {
      if (B.south) {
        if (j == B.Jstr - 1 && i >= B.IstrUm1 && i <= B.Iendp1) LU_(i, j) = (clu & (1 << ROMS_ISOUTH)) ? g2 * LU_(i, B.Jstr) : 0.0;
        if (j == B.Jstr && i >= B.Istrm1 && i <= B.Iendp1) LV_(i, j) = (clv & (1 << ROMS_ISOUTH)) ? 0.0 : LV_(i, B.Jstr + 1);         // LapV(i,JstrV-1)
      }
      if (B.north && j == B.Jend + 1) {
        if (i >= B.IstrUm1 && i <= B.Iendp1) LU_(i, j) = (clu & (1 << ROMS_INORTH)) ? g2 * LU_(i, B.Jend) : 0.0;
        if (i >= B.Istrm1 && i <= B.Iendp1) LV_(i, j) = (clv & (1 << ROMS_INORTH)) ? 0.0 : LV_(i, B.Jend);
      }
    }
  } else if (!(G.ewp || G.nsp)) {
    const int Istr = B.Istr, Iend = B.Iend, Jstr = B.Jstr, Jend = B.Jend;
    if (B.sw && i == Istr && j == Jstr) {
      LU_(Istr, Jstr - 1) = 0.5 * (LU_(Istr + 1, Jstr - 1) + LU_(Istr, Jstr));
      LV_(Istr - 1, Jstr) = 0.5 * (LV_(Istr - 1, Jstr + 1) + LV_(Istr, Jstr));
    }
    if (B.se && i == Iend && j == Jstr) {
      LU_(Iend + 1, Jstr - 1) = 0.5 * (LU_(Iend, Jstr - 1) + LU_(Iend + 1, Jstr));
      LV_(Iend + 1, Jstr) = 0.5 * (LV_(Iend, Jstr) + LV_(Iend + 1, Jstr + 1));
    }
    if (B.nw && i == Istr && j == Jend) {
      LU_(Istr, Jend + 1) = 0.5 * (LU_(Istr + 1, Jend + 1) + LU_(Istr, Jend));
      LV_(Istr - 1, Jend + 1) = 0.5 * (LV_(Istr, Jend + 1) + LV_(Istr - 1, Jend));
    }
    if (B.ne && i == Iend && j == Jend) {
      LU_(Iend + 1, Jend + 1) = 0.5 * (LU_(Iend, Jend + 1) + LU_(Iend + 1, Jend));
      LV_(Iend + 1, Jend + 1) = 0.5 * (LV_(Iend, Jend + 1) + LV_(Iend + 1, Jend));
    }
  }
#undef LU_
#undef LV_
}
THREAD_GLOBAL(k_uvg_lapbc, KArgs)
