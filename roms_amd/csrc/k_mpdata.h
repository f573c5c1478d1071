// k_mpdata.h -- MPDATA tracer advection of step3d_t (tracers with Hadvection = Vadvection = MPDATA).
//
// Replaces, for such tracers:
//   step3d_t_tile   ROMS/Nonlinear/step3d_t.F  first-order upstream fluxes on the extended range
//                   :451-470, horizontal step into Ta :873-890, vertical step :995-1010,1246-1290,
//                   corrected (anti-diffusive) fluxes :1399-1500, plain implicit vertical diffusion
//                   :1724-1790
//   mpdata_adiff_tile  ROMS/Nonlinear/mpdata_adiff.F:38-1227
//
// All kernels are point-wise (one thread per (i,j,k)); the reference's per-row temporaries C(i,k),
// Wm(i,k), odz and oHz are functions of the point and are recomputed where they are used.
// Work arrays (Fields::mp3): 0 Ta (N planes per tracer), 1 Ua, 2 Va, 3 Wa (w-levels), 4 beta_up,
// 5 beta_dn.
#pragma once
#include "roms_ctx.h"
#include "k_diag3d.h"

struct MpArgs {
  Fields Fv;         // the array pointers, by value (a table in device memory would cost every kernel one more dependent round trip)
  DGrid G;
  int itrc;
};

#define MP_TA(i, j, k) Ta[X3(i, j, k)]
#define MP_ODZ(i, j, k) (1.0 / (z_r[X3(i, j, (k) + 1)] - z_r[X3(i, j, k)]))
#define MP_OHZ(i, j, k) (1.0 / Hz[X3(i, j, k)])
#define MP_SIGN1(x) ((x) >= 0.0 ? 1.0 : -1.0)

// the third-order pseudo-velocity polynomial of mpdata_adiff.F (:405-440, :575-610, :790-830): P is
// the component being built, Q and R the other two, gp/gq/gr the normalised gradients.  The eta
// component multiplies sig_b with P*Q^2 and sig_c with P^2*Q (swapbc), as the reference does.
#define MP_SIGMA(P, Q, R, gp, gq, gr, out, swapbc)                                                          \
  do {                                                                                                       \
    const double PP = (gp) * (gp), QQ = (gq) * (gq), RR = (gr) * (gr), PQ = (gp) * (gq), PR = (gp) * (gr);    \
    const double sPP = (P) * (P), sQQ = (Q) * (Q), sRR = (R) * (R), sPQ = (P) * (Q), sPR = (P) * (R);         \
    const double s_alfa = 1.0 / (1.0 - fabs(gp) + eps);                                                      \
    const double s_beta = -(gp) / ((1.0 - fabs(gp)) * (1.0 - PP) + eps);                                     \
    const double s_gama = 2.0 * fabs(PP * (gp)) / ((1.0 - fabs(gp)) * (1.0 - PP) * (1.0 - fabs(PP * (gp))) + eps); \
    const double s_a = -(gq) / ((1.0 - fabs(gp)) * (1.0 - fabs(PQ)) + eps);                                  \
    const double s_b = PQ / ((1.0 - fabs(gp)) * (1.0 - PP * fabs(gq)) + eps) *                               \
                       (fabs(gq) / (1.0 - fabs(PQ) + eps) + 2.0 * (gp) / (1.0 - PP + eps));                  \
    const double s_c = fabs(gp) * QQ / ((1.0 - fabs(gp)) * (1.0 - QQ * fabs(gp)) * (1.0 - fabs(PQ)) + eps);  \
    const double s_d = -(gr) / ((1.0 - fabs(gp)) * (1.0 - fabs(PR)) + eps);                                  \
    const double s_e = PR / ((1.0 - fabs(gp)) * (1.0 - PP * fabs(gr)) + eps) *                               \
                       (fabs(gr) / (1.0 - fabs(PR) + eps) + 2.0 * (gp) / (1.0 - PP + eps));                  \
    const double s_f = fabs(gp) * RR / ((1.0 - fabs(gp)) * (1.0 - RR * fabs(gp)) * (1.0 - fabs(PR)) + eps);  \
    out = s_alfa * (P) + s_beta * sPP + s_gama * sPP * (P) + s_a * sPQ +                                     \
          ((swapbc) ? s_b * (P) * sQQ : s_b * sPP * (Q)) + ((swapbc) ? s_c * sPP * (Q) : s_c * (P) * sQQ) +  \
          s_d * sPR + s_e * sPP * (R) + s_f * (P) * sRR;                                                     \
  } while (0)

// ---- Ta: first-order upstream horizontal and vertical steps on the extended range ------------
// index space (IstrUm2:Iendp2i, JstrVm2:Jendp2i, 1:N); also the closed-edge values of Ta that
// mpdata_adiff fills (:224-290)
THREAD_KERNEL(k_mp_ta, MpArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int i = B.IstrUm2 + gx, j = B.JstrVm2 + gy, k = gz + 1, N = G.N, itrc = a.itrc;
  if (i > B.Iendp2i || j > B.Jendp2i) return;
  const double *T3 = F.t + XT(G.LBi, G.LBj, 1, 3, itrc), *tn = F.t + XT(G.LBi, G.LBj, 1, G.nnew, itrc);
  const double *Hu = F.Huon, *Hv = F.Hvom, *W = F.W, *Hz = F.Hz;
  double *Ta = F.mp3[0] + (size_t)(itrc - 1) * G.nij * N;
#define MP_FX(ii) (KMAX(Hu[X3(ii, j, k)], 0.0) * T3[X3((ii) - 1, j, k)] + KMIN(Hu[X3(ii, j, k)], 0.0) * T3[X3(ii, j, k)])
#define MP_FE(jj) (KMAX(Hv[X3(i, jj, k)], 0.0) * T3[X3(i, (jj) - 1, k)] + KMIN(Hv[X3(i, jj, k)], 0.0) * T3[X3(i, jj, k)])
#define MP_FC(kk) (((kk) <= 0 || (kk) >= N) ? 0.0 : KMAX(W[XW(i, j, kk)], 0.0) * T3[X3(i, j, kk)] + KMIN(W[XW(i, j, kk)], 0.0) * T3[X3(i, j, (kk) + 1)])
  const double cff = G.dt * F.pm[X2(i, j)] * F.pn[X2(i, j)];
  const double cff1 = cff * (MP_FX(i + 1) - MP_FX(i));
  const double cff2 = cff * (MP_FE(j + 1) - MP_FE(j));
  const double cff3 = cff1 + cff2;
  double ta = tn[X3(i, j, k)] - cff3;
  const double cv = cff * (MP_FC(k) - MP_FC(k - 1));
  ta = (ta - cv) * MP_OHZ(i, j, k);
#undef MP_FX
#undef MP_FE
#undef MP_FC
  Ta[X3(i, j, k)] = ta;
  const bool w = !G.ewp && B.west && i == B.Istr, e = !G.ewp && B.east && i == B.Iend;
  const bool s = !G.nsp && B.south && j == B.Jstr, n = !G.nsp && B.north && j == B.Jend;
  if (w) Ta[X3(i - 1, j, k)] = ta;
  if (e) Ta[X3(i + 1, j, k)] = ta;
  if (s) Ta[X3(i, j - 1, k)] = ta;
  if (n) Ta[X3(i, j + 1, k)] = ta;
  // corners (closed in both directions): 0.5*(edge + edge) of two copies of the same value
  if (w && s) Ta[X3(i - 1, j - 1, k)] = 0.5 * (ta + ta);
  if (e && s) Ta[X3(i + 1, j - 1, k)] = 0.5 * (ta + ta);
  if (w && n) Ta[X3(i - 1, j + 1, k)] = 0.5 * (ta + ta);
  if (e && n) Ta[X3(i + 1, j + 1, k)] = 0.5 * (ta + ta);
}
THREAD_GLOBAL(k_mp_ta, MpArgs)

// ---- Ua (dir 0; index space IstrU-1:Iendp2, JstrV-1:Jendp1) and Va (dir 1; IstrU-1:Iendp1,
//      min(JstrV-1,JstrVm1):Jendp2), mpdata_adiff.F:307-640, with the closed-wall values :642-720 ------------
THREAD_KERNEL(k_mp_uva, MpArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, itrc = a.itrc;
  const int dir = gz / N, k = gz % N + 1;
  // (Va: the reference's loop starts at JstrVm1 -- behind a closed southern wall that is Jstr+1 -- and its boundary statement then
  // sets Va(Istrm1:Iendp1,Jstr,:) = 0 (:726-733), which k_mp_beta reads at once: the wall row is part of this index space, its
  // value the wall rule at the end.  Round 6: until then that row kept the zeros of its allocation and of k_mp_limit's later
  // store -- right, but a read of scratch nobody had written in this step, which ROMS_HIP_POISON=1 turned into NaN)
  const int i = B.IstrU - 1 + gx, j = (dir == 0 ? B.JstrV - 1 : KMIN(B.JstrV - 1, B.JstrVm1)) + gy;
  if (i > (dir == 0 ? B.Iendp2 : B.Iendp1) || j > (dir == 0 ? B.Jendp1 : B.Jendp2)) return;
  const double eps = 1.0E-18, eps2 = 1.0E-10, fac = 1.0, dt = G.dt;
  const double *Ta = F.mp3[0] + (size_t)(itrc - 1) * G.nij * N;
  const double *pm = F.pm, *pn = F.pn, *z_r = F.z_r, *W = F.W, *Hz = F.Hz, *Huon = F.Huon, *Hvom = F.Hvom;
  const int di = dir == 0 ? 1 : 0, dj = dir == 0 ? 0 : 1;     // the neighbour across the face: (i-di, j-dj)
  const int im = i - di, jm = j - dj;
  double *Out = F.mp3[dir == 0 ? 1 : 2];
  double val;
  if (MP_TA(im, jm, k) <= 0.0 || MP_TA(i, j, k) <= 0.0 || fabs(MP_TA(im, jm, k) - MP_TA(i, j, k)) <= eps2) {
    val = 0.0;
  } else {
    // vertical gradient C and vertical Courant number Wm at the face (:311-370, :476-535)
    double Cc, Wmm;
    if (k == 1) {
      Cc = 0.25 * ((MP_TA(i, j, k + 1) - MP_TA(i, j, k)) * MP_ODZ(i, j, k) + (MP_TA(im, jm, k + 1) - MP_TA(im, jm, k)) * MP_ODZ(im, jm, k)) *
           (z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)] + z_r[X3(im, jm, k + 1)] - z_r[X3(im, jm, k)]) /
           (MP_TA(im, jm, k) + MP_TA(i, j, k) + eps);
      Wmm = 0.25 * dt * (W[XW(im, jm, k)] * MP_ODZ(im, jm, k) * pm[X2(im, jm)] * pn[X2(im, jm)] +
                         W[XW(i, j, k)] * MP_ODZ(i, j, k) * pm[X2(i, j)] * pn[X2(i, j)]);
    } else if (k < N) {
      Cc = 0.0625 * ((MP_TA(i, j, k + 1) - MP_TA(i, j, k)) * MP_ODZ(i, j, k) + (MP_TA(i, j, k) - MP_TA(i, j, k - 1)) * MP_ODZ(i, j, k - 1) +
                     (MP_TA(im, jm, k + 1) - MP_TA(im, jm, k)) * MP_ODZ(im, jm, k) +
                     (MP_TA(im, jm, k) - MP_TA(im, jm, k - 1)) * MP_ODZ(im, jm, k - 1)) *
           (z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k - 1)] + z_r[X3(im, jm, k + 1)] - z_r[X3(im, jm, k - 1)]) /
           (MP_TA(im, jm, k) + MP_TA(i, j, k) + eps);
      Wmm = 0.25 * dt * ((W[XW(im, jm, k - 1)] * MP_ODZ(im, jm, k - 1) + W[XW(im, jm, k)] * MP_ODZ(im, jm, k)) * pm[X2(im, jm)] *
                             pn[X2(im, jm)] +
                         (W[XW(i, j, k)] * MP_ODZ(i, j, k) + W[XW(i, j, k - 1)] * MP_ODZ(i, j, k - 1)) * pm[X2(i, j)] * pn[X2(i, j)]);
    } else {
      Cc = 0.25 * ((MP_TA(i, j, k) - MP_TA(i, j, k - 1)) * MP_ODZ(i, j, k - 1) + (MP_TA(im, jm, k) - MP_TA(im, jm, k - 1)) * MP_ODZ(im, jm, k - 1)) *
           (z_r[X3(i, j, k)] - z_r[X3(i, j, k - 1)] + z_r[X3(im, jm, k)] - z_r[X3(im, jm, k - 1)]) /
           (MP_TA(im, jm, k) + MP_TA(i, j, k) + eps);
      Wmm = 0.25 * dt * (W[XW(im, jm, k - 1)] * MP_ODZ(im, jm, k - 1) * pm[X2(im, jm)] * pn[X2(im, jm)] +
                         W[XW(i, j, k - 1)] * MP_ODZ(i, j, k - 1) * pm[X2(i, j)] * pn[X2(i, j)]);
    }
    double A, Bg, Um, Vm;
    if (dir == 0) {
      A = (MP_TA(i, j, k) - MP_TA(i - 1, j, k)) / (MP_TA(i, j, k) + MP_TA(i - 1, j, k) + eps);
      if (G.masking)        // mpdata_adiff.F:353-362: each cross difference carries the mask of the face it spans
        Bg = 0.03125 * ((MP_TA(i, j + 1, k) - MP_TA(i, j, k)) * (pn[X2(i, j)] + pn[X2(i, j + 1)]) * G.vmask[X2(i, j + 1)] +
                        (MP_TA(i, j, k) - MP_TA(i, j - 1, k)) * (pn[X2(i, j - 1)] + pn[X2(i, j)]) * G.vmask[X2(i, j)] +
                        (MP_TA(i - 1, j + 1, k) - MP_TA(i - 1, j, k)) * (pn[X2(i - 1, j)] + pn[X2(i - 1, j + 1)]) * G.vmask[X2(i - 1, j + 1)] +
                        (MP_TA(i - 1, j, k) - MP_TA(i - 1, j - 1, k)) * (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)]) * G.vmask[X2(i - 1, j)]);
      else
      Bg = 0.03125 * ((MP_TA(i, j + 1, k) - MP_TA(i, j, k)) * (pn[X2(i, j)] + pn[X2(i, j + 1)]) +
                      (MP_TA(i, j, k) - MP_TA(i, j - 1, k)) * (pn[X2(i, j - 1)] + pn[X2(i, j)]) +
                      (MP_TA(i - 1, j + 1, k) - MP_TA(i - 1, j, k)) * (pn[X2(i - 1, j)] + pn[X2(i - 1, j + 1)]) +
                      (MP_TA(i - 1, j, k) - MP_TA(i - 1, j - 1, k)) * (pn[X2(i - 1, j - 1)] + pn[X2(i - 1, j)]));
      Bg = Bg * (F.on_v[X2(i, j)] + F.on_v[X2(i, j + 1)] + F.on_v[X2(i - 1, j)] + F.on_v[X2(i - 1, j + 1)]) /
           (MP_TA(i - 1, j, k) + MP_TA(i, j, k) + eps);
      Um = 0.125 * Huon[X3(i, j, k)] * dt * (pm[X2(i, j)] + pm[X2(i - 1, j)]) * (pn[X2(i, j)] + pn[X2(i - 1, j)]) *
           (MP_OHZ(i - 1, j, k) + MP_OHZ(i, j, k));
      Vm = 0.03125 * dt *
           (Hvom[X3(i - 1, j, k)] * (pm[X2(i - 1, j)] + pm[X2(i - 1, j - 1)]) * (pn[X2(i - 1, j)] + pn[X2(i - 1, j - 1)]) *
                (MP_OHZ(i - 1, j, k) + MP_OHZ(i - 1, j - 1, k)) +
            Hvom[X3(i - 1, j + 1, k)] * (pm[X2(i - 1, j + 1)] + pm[X2(i - 1, j)]) * (pn[X2(i - 1, j + 1)] + pn[X2(i - 1, j)]) *
                (MP_OHZ(i - 1, j + 1, k) + MP_OHZ(i - 1, j, k)) +
            Hvom[X3(i, j, k)] * (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]) *
                (MP_OHZ(i, j, k) + MP_OHZ(i, j - 1, k)) +
            Hvom[X3(i, j + 1, k)] * (pm[X2(i, j + 1)] + pm[X2(i, j)]) * (pn[X2(i, j + 1)] + pn[X2(i, j)]) *
                (MP_OHZ(i, j + 1, k) + MP_OHZ(i, j, k)));
    } else {
      if (G.masking)        // :573-582
        A = 0.03125 * ((MP_TA(i + 1, j, k) - MP_TA(i, j, k)) * (pm[X2(i + 1, j)] + pm[X2(i, j)]) * G.umask[X2(i + 1, j)] +
                       (MP_TA(i, j, k) - MP_TA(i - 1, j, k)) * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * G.umask[X2(i, j)] +
                       (MP_TA(i + 1, j - 1, k) - MP_TA(i, j - 1, k)) * (pm[X2(i + 1, j - 1)] + pm[X2(i, j - 1)]) * G.umask[X2(i + 1, j - 1)] +
                       (MP_TA(i, j - 1, k) - MP_TA(i - 1, j - 1, k)) * (pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]) * G.umask[X2(i, j - 1)]);
      else
      A = 0.03125 * ((MP_TA(i + 1, j, k) - MP_TA(i, j, k)) * (pm[X2(i + 1, j)] + pm[X2(i, j)]) +
                     (MP_TA(i, j, k) - MP_TA(i - 1, j, k)) * (pm[X2(i - 1, j)] + pm[X2(i, j)]) +
                     (MP_TA(i + 1, j - 1, k) - MP_TA(i, j - 1, k)) * (pm[X2(i + 1, j - 1)] + pm[X2(i, j - 1)]) +
                     (MP_TA(i, j - 1, k) - MP_TA(i - 1, j - 1, k)) * (pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]));
      A = A * (F.om_u[X2(i, j)] + F.om_u[X2(i + 1, j)] + F.om_u[X2(i, j - 1)] + F.om_u[X2(i + 1, j - 1)]) /
          (MP_TA(i, j - 1, k) + MP_TA(i, j, k) + eps);
      Bg = (MP_TA(i, j, k) - MP_TA(i, j - 1, k)) / (MP_TA(i, j, k) + MP_TA(i, j - 1, k) + eps);
      Um = 0.03125 * dt *
           (Huon[X3(i + 1, j, k)] * (pm[X2(i + 1, j)] + pm[X2(i, j)]) * (pn[X2(i + 1, j)] + pn[X2(i, j)]) *
                (MP_OHZ(i + 1, j, k) + MP_OHZ(i, j, k)) +
            Huon[X3(i + 1, j - 1, k)] * (pm[X2(i + 1, j - 1)] + pm[X2(i, j - 1)]) * (pn[X2(i + 1, j - 1)] + pn[X2(i, j - 1)]) *
                (MP_OHZ(i + 1, j - 1, k) + MP_OHZ(i, j - 1, k)) +
            Huon[X3(i, j, k)] * (pm[X2(i - 1, j)] + pm[X2(i, j)]) * (pn[X2(i - 1, j)] + pn[X2(i, j)]) *
                (MP_OHZ(i - 1, j, k) + MP_OHZ(i, j, k)) +
            Huon[X3(i, j - 1, k)] * (pm[X2(i - 1, j - 1)] + pm[X2(i, j - 1)]) * (pn[X2(i - 1, j - 1)] + pn[X2(i, j - 1)]) *
                (MP_OHZ(i - 1, j - 1, k) + MP_OHZ(i, j - 1, k)));
      Vm = 0.125 * Hvom[X3(i, j, k)] * dt * (pn[X2(i, j - 1)] + pn[X2(i, j)]) * (pm[X2(i, j - 1)] + pm[X2(i, j)]) *
           (MP_OHZ(i, j - 1, k) + MP_OHZ(i, j, k));
    }
    const double X = (fabs(Um) - Um * Um) * A - Bg * Um * Vm - Cc * Um * Wmm;
    const double Y = (fabs(Vm) - Vm * Vm) * Bg - A * Um * Vm - Cc * Vm * Wmm;
    const double Z = (fabs(Wmm) - Wmm * Wmm) * Cc - A * Um * Wmm - Bg * Vm * Wmm;
    double r;
    if (dir == 0) {
      MP_SIGMA(X, Y, Z, A, Bg, Cc, r, 0);
      val = KMIN(fabs(r), fac * fabs(Um)) * MP_SIGN1(r);
      if (G.masking) val = val * G.umask[X2(i, j)];       // :460
      if (G.wet_dry) val = val * F.umask_wet[X2(i, j)];   // WET_DRY :464
    } else {
      MP_SIGMA(Y, X, Z, Bg, A, Cc, r, 1);
      val = KMIN(fabs(r), fac * fabs(Vm)) * MP_SIGN1(r);
      if (G.masking) val = val * G.vmask[X2(i, j)];       // :683
      if (G.wet_dry) val = val * F.vmask_wet[X2(i, j)];   // WET_DRY :687
    }
  }
  // closed walls :642-720 (the wall value replaces whatever was computed there)
  if (dir == 0) {
    if (!G.ewp && ((B.west && i == B.Istr) || (B.east && i == B.Iend + 1)) && j >= B.Jstrm1 && j <= B.Jendp1) val = 0.0;
  } else {
    if (!G.nsp && ((B.south && j == B.Jstr) || (B.north && j == B.Jend + 1)) && i >= B.Istrm1 && i <= B.Iendp1) val = 0.0;
  }
  Out[X3(i, j, k)] = val;
}
THREAD_GLOBAL(k_mp_uva, MpArgs)

// ---- Wa, mpdata_adiff.F:722-860; index space (IstrU-1:Iendp1, JstrV-1:Jendp1, 0:N) -----------
THREAD_KERNEL(k_mp_wa, MpArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, itrc = a.itrc;
  const int i = B.IstrU - 1 + gx, j = B.JstrV - 1 + gy, k = gz;
  if (i > B.Iendp1 || j > B.Jendp1) return;
  const double eps = 1.0E-18, eps2 = 1.0E-10, fac = 1.0, dt = G.dt;
  const double *Ta = F.mp3[0] + (size_t)(itrc - 1) * G.nij * N;
  const double *pm = F.pm, *pn = F.pn, *z_r = F.z_r, *W = F.W, *Hz = F.Hz, *Huon = F.Huon, *Hvom = F.Hvom;
  double *Wa = F.mp3[3];
  double val = 0.0;
  if (k >= 1 && k <= N - 1 &&
      !(MP_TA(i, j, k) <= 0.0 || MP_TA(i, j, k + 1) <= 0.0 || fabs(MP_TA(i, j, k) - MP_TA(i, j, k + 1)) <= eps2)) {
    const double Cc = (MP_TA(i, j, k + 1) - MP_TA(i, j, k)) / (MP_TA(i, j, k + 1) + MP_TA(i, j, k) + eps);
    double A, Bg;
    if (G.masking) {        // :777-794
      A = 0.0625 * ((MP_TA(i + 1, j, k + 1) - MP_TA(i, j, k + 1)) * (pm[X2(i + 1, j)] + pm[X2(i, j)]) * G.umask[X2(i + 1, j)] +
                    (MP_TA(i, j, k + 1) - MP_TA(i - 1, j, k + 1)) * (pm[X2(i, j)] + pm[X2(i - 1, j)]) * G.umask[X2(i, j)] +
                    (MP_TA(i + 1, j, k) - MP_TA(i, j, k)) * (pm[X2(i + 1, j)] + pm[X2(i, j)]) * G.umask[X2(i + 1, j)] +
                    (MP_TA(i, j, k) - MP_TA(i - 1, j, k)) * (pm[X2(i, j)] + pm[X2(i - 1, j)]) * G.umask[X2(i, j)]);
      Bg = 0.0625 * ((MP_TA(i, j + 1, k + 1) - MP_TA(i, j, k + 1)) * (pn[X2(i, j + 1)] + pn[X2(i, j)]) * G.vmask[X2(i, j + 1)] +
                     (MP_TA(i, j, k + 1) - MP_TA(i, j - 1, k + 1)) * (pn[X2(i, j)] + pn[X2(i, j - 1)]) * G.vmask[X2(i, j)] +
                     (MP_TA(i, j + 1, k) - MP_TA(i, j, k)) * (pn[X2(i, j + 1)] + pn[X2(i, j)]) * G.vmask[X2(i, j + 1)] +
                     (MP_TA(i, j, k) - MP_TA(i, j - 1, k)) * (pn[X2(i, j)] + pn[X2(i, j - 1)]) * G.vmask[X2(i, j)]);
    } else {
      A = 0.0625 * ((MP_TA(i + 1, j, k + 1) - MP_TA(i, j, k + 1)) * (pm[X2(i + 1, j)] + pm[X2(i, j)]) +
                    (MP_TA(i, j, k + 1) - MP_TA(i - 1, j, k + 1)) * (pm[X2(i, j)] + pm[X2(i - 1, j)]) +
                    (MP_TA(i + 1, j, k) - MP_TA(i, j, k)) * (pm[X2(i + 1, j)] + pm[X2(i, j)]) +
                    (MP_TA(i, j, k) - MP_TA(i - 1, j, k)) * (pm[X2(i, j)] + pm[X2(i - 1, j)]));
      Bg = 0.0625 * ((MP_TA(i, j + 1, k + 1) - MP_TA(i, j, k + 1)) * (pn[X2(i, j + 1)] + pn[X2(i, j)]) +
                     (MP_TA(i, j, k + 1) - MP_TA(i, j - 1, k + 1)) * (pn[X2(i, j)] + pn[X2(i, j - 1)]) +
                     (MP_TA(i, j + 1, k) - MP_TA(i, j, k)) * (pn[X2(i, j + 1)] + pn[X2(i, j)]) +
                     (MP_TA(i, j, k) - MP_TA(i, j - 1, k)) * (pn[X2(i, j)] + pn[X2(i, j - 1)]));
    }
    A = A * (F.om_u[X2(i + 1, j)] + F.om_u[X2(i, j)]) / (MP_TA(i, j, k + 1) + MP_TA(i, j, k) + eps);
    Bg = Bg * (F.on_v[X2(i, j + 1)] + F.on_v[X2(i, j)]) / (MP_TA(i, j, k + 1) + MP_TA(i, j, k) + eps);
    const double Um =
        0.03125 * dt *
        (Huon[X3(i, j, k)] * (pm[X2(i, j)] + pm[X2(i - 1, j)]) * (pn[X2(i, j)] + pn[X2(i - 1, j)]) *
             (MP_OHZ(i, j, k) + MP_OHZ(i - 1, j, k)) +
         Huon[X3(i, j, k + 1)] * (pm[X2(i, j)] + pm[X2(i - 1, j)]) * (pn[X2(i, j)] + pn[X2(i - 1, j)]) *
             (MP_OHZ(i, j, k + 1) + MP_OHZ(i - 1, j, k + 1)) +
         Huon[X3(i + 1, j, k)] * (pm[X2(i, j)] + pm[X2(i + 1, j)]) * (pn[X2(i, j)] + pn[X2(i + 1, j)]) *
             (MP_OHZ(i, j, k) + MP_OHZ(i + 1, j, k)) +
         Huon[X3(i + 1, j, k + 1)] * (pm[X2(i, j)] + pm[X2(i + 1, j)]) * (pn[X2(i, j)] + pn[X2(i + 1, j)]) *
             (MP_OHZ(i, j, k + 1) + MP_OHZ(i + 1, j, k + 1)));
    const double Vm =
        0.03125 * dt *
        (Hvom[X3(i, j, k)] * (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]) *
             (MP_OHZ(i, j, k) + MP_OHZ(i, j - 1, k)) +
         Hvom[X3(i, j, k + 1)] * (pm[X2(i, j)] + pm[X2(i, j - 1)]) * (pn[X2(i, j)] + pn[X2(i, j - 1)]) *
             (MP_OHZ(i, j, k + 1) + MP_OHZ(i, j - 1, k + 1)) +
         Hvom[X3(i, j + 1, k)] * (pm[X2(i, j)] + pm[X2(i, j + 1)]) * (pn[X2(i, j)] + pn[X2(i, j + 1)]) *
             (MP_OHZ(i, j, k) + MP_OHZ(i, j + 1, k)) +
         Hvom[X3(i, j + 1, k + 1)] * (pm[X2(i, j)] + pm[X2(i, j + 1)]) * (pn[X2(i, j)] + pn[X2(i, j + 1)]) *
             (MP_OHZ(i, j, k + 1) + MP_OHZ(i, j + 1, k + 1)));
    const double Wmm = W[XW(i, j, k)] * MP_ODZ(i, j, k) * pm[X2(i, j)] * pn[X2(i, j)] * dt;
    const double X = (fabs(Um) - Um * Um) * A - Bg * Um * Vm - Cc * Um * Wmm;
    const double Y = (fabs(Vm) - Vm * Vm) * Bg - A * Um * Vm - Cc * Vm * Wmm;
    const double Z = (fabs(Wmm) - Wmm * Wmm) * Cc - A * Um * Wmm - Bg * Vm * Wmm;
    double r;
    MP_SIGMA(Z, Y, X, Cc, Bg, A, r, 0);
    val = KMIN(fabs(r), fac * fabs(Wmm)) * MP_SIGN1(r);
    if (G.masking) val = val * G.rmask[X2(i, j)];         // :924
    if (G.wet_dry) val = val * F.rmask_wet[X2(i, j)];     // WET_DRY :928
  }
  Wa[XW(i, j, k)] = val;
}
THREAD_GLOBAL(k_mp_wa, MpArgs)

// ---- Ua, Va and Wa of a cell in one launch: the three evaluations above one behind the other in the same thread (they read
//      the same Ta, Huon, Hvom, W, Hz neighbourhood: the second and third find it in the L1); index space
//      (IstrU-1:Iendp2, JstrV-1:Jendp2, 0:N), every part keeps its own ranges.  Same expressions, same bits.
THREAD_KERNEL(k_mp_uvwa, MpArgs) {
  const TB &B = a.G.T;
  const int N = a.G.N;
  if (gz >= 1) {
    k_mp_uva_body(a, gx, gy, gz - 1);                            // Ua at level gz
    const int gy1 = gy - (KMIN(B.JstrV - 1, B.JstrVm1) - (B.JstrV - 1));      // (Va's rows: the wall row of a closed southern edge included)
    if (gy1 >= 0) k_mp_uva_body(a, gx, gy1, N + gz - 1);         // Va
  }
  k_mp_wa_body(a, gx, gy, gz);
}
THREAD_GLOBAL(k_mp_uvwa, MpArgs)

// ---- FCT ratios beta_up, beta_dn :862-1090; index space (IstrU-1:Iendp1, JstrV-1:Jendp1, 1:N) -
// the two ratios of one point (the expressions of :862-1090 for entry (i,j,k))
KDEV void mp_beta_pt(const DGrid &G, const Fields &F, int itrc, int i, int j, int k, double &b_up, double &b_dn) {
  const int N = G.N;
  const double eps = 1.0E-18;
  const double *Ta = F.mp3[0] + (size_t)(itrc - 1) * G.nij * N, *Ua = F.mp3[1], *Va = F.mp3[2], *Wa = F.mp3[3];
  const double *T3 = F.t + XT(G.LBi, G.LBj, 1, 3, itrc);
  // extrema over the point's neighbourhood :962-1100: Tmax over value * mask_up, Tmin over value * mask_dn, where
  // mask_up = rmask and mask_dn = MAX(1, MIN(Large, (1 - rmask) * Large)) keep land out of both (:945-960; both 1
  // without MASKING)
  const bool msk = G.masking != 0;
  const double Large = 1.0E+20;
#define MP_UP(ii, jj) (msk ? G.rmask[X2(ii, jj)] : 1.0)
#define MP_DN(ii, jj) (msk ? KMAX(1.0, KMIN(Large, (1.0 - G.rmask[X2(ii, jj)]) * Large)) : 1.0)
  double Tmax, Tmin;
  { const double q_ = MP_TA(i - 1, j, k); Tmax = msk ? q_ * MP_UP(i - 1, j) : q_; Tmin = msk ? q_ * MP_DN(i - 1, j) : q_; }
#define MP_MM(v_, ii, jj) do { const double q_ = (v_); Tmax = KMAX(Tmax, msk ? q_ * MP_UP(ii, jj) : q_); \
                               Tmin = KMIN(Tmin, msk ? q_ * MP_DN(ii, jj) : q_); } while (0)
  MP_MM(T3[X3(i - 1, j, k)], i - 1, j);
  MP_MM(MP_TA(i, j, k), i, j); MP_MM(T3[X3(i, j, k)], i, j);
  MP_MM(MP_TA(i + 1, j, k), i + 1, j); MP_MM(T3[X3(i + 1, j, k)], i + 1, j);
  MP_MM(MP_TA(i, j - 1, k), i, j - 1); MP_MM(T3[X3(i, j - 1, k)], i, j - 1);
  MP_MM(MP_TA(i, j + 1, k), i, j + 1); MP_MM(T3[X3(i, j + 1, k)], i, j + 1);
  if (k > 1) { MP_MM(MP_TA(i, j, k - 1), i, j); MP_MM(T3[X3(i, j, k - 1)], i, j); }
  if (k < N) { MP_MM(MP_TA(i, j, k + 1), i, j); MP_MM(T3[X3(i, j, k + 1)], i, j); }
#undef MP_MM
#undef MP_UP
#undef MP_DN
  double cff1 = MP_TA(i - 1, j, k) * KMAX(0.0, Ua[X3(i, j, k)]) - MP_TA(i + 1, j, k) * KMIN(0.0, Ua[X3(i + 1, j, k)]) +
                MP_TA(i, j - 1, k) * KMAX(0.0, Va[X3(i, j, k)]) - MP_TA(i, j + 1, k) * KMIN(0.0, Va[X3(i, j + 1, k)]);
  if (k > 1) cff1 = cff1 + MP_TA(i, j, k - 1) * KMAX(0.0, Wa[XW(i, j, k - 1)]);
  if (k < N) cff1 = cff1 - MP_TA(i, j, k + 1) * KMIN(0.0, Wa[XW(i, j, k)]);
  b_up = (Tmax - MP_TA(i, j, k)) / (cff1 + eps);
  double cff2 = MP_TA(i, j, k) * KMAX(0.0, Ua[X3(i + 1, j, k)]) - MP_TA(i, j, k) * KMIN(0.0, Ua[X3(i, j, k)]) +
                MP_TA(i, j, k) * KMAX(0.0, Va[X3(i, j + 1, k)]) - MP_TA(i, j, k) * KMIN(0.0, Va[X3(i, j, k)]);
  if (k < N) cff2 = cff2 + MP_TA(i, j, k) * KMAX(0.0, Wa[XW(i, j, k)]);
  if (k > 1) cff2 = cff2 - MP_TA(i, j, k) * KMIN(0.0, Wa[XW(i, j, k - 1)]);
  b_dn = (MP_TA(i, j, k) - Tmin) / (cff2 + eps);
}
THREAD_KERNEL(k_mp_beta, MpArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int i = B.IstrU - 1 + gx, j = B.JstrV - 1 + gy, k = gz + 1;
  if (i > B.Iendp1 || j > B.Jendp1) return;
  double bu, bd;
  mp_beta_pt(G, F, a.itrc, i, j, k, bu, bd);
  F.mp3[4][X3(i, j, k)] = bu;
  F.mp3[5][X3(i, j, k)] = bd;
}
THREAD_GLOBAL(k_mp_beta, MpArgs)

// ---- limited anti-diffusive velocities :1100-1220 (in place); index space (Istr:Iend+1, Jstr:Jend+1, 1:N)
THREAD_KERNEL(k_mp_limit, MpArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N;
  const int i = B.Istr + gx, j = B.Jstr + gy, k = gz + 1;
  double *Ua = F.mp3[1], *Va = F.mp3[2], *Wa = F.mp3[3];
  const double *bup = F.mp3[4], *bdn = F.mp3[5], *z_r = F.z_r;
  const double cff = 1.0 / G.dt;
  if (j <= B.Jend) {
    if (!G.ewp && ((B.west && i == B.Istr) || (B.east && i == B.Iend + 1))) {
      Ua[X3(i, j, k)] = 0.0;                                       // closed walls :1152-1185
    } else if (i >= B.IstrU && i <= B.Iendp1) {
      const double cff1 = KMIN(KMIN(bdn[X3(i - 1, j, k)], bup[X3(i, j, k)]), 1.0);
      const double cff2 = KMIN(KMIN(bup[X3(i - 1, j, k)], bdn[X3(i, j, k)]), 1.0);
      Ua[X3(i, j, k)] = (cff1 * KMAX(0.0, Ua[X3(i, j, k)]) + cff2 * KMIN(0.0, Ua[X3(i, j, k)])) * cff * F.om_u[X2(i, j)];
      if (G.masking) Ua[X3(i, j, k)] = Ua[X3(i, j, k)] * G.umask[X2(i, j)];     // :1114
      if (G.wet_dry) Ua[X3(i, j, k)] = Ua[X3(i, j, k)] * F.umask_wet[X2(i, j)];  // WET_DRY :1118
    }
  }
  if (i <= B.Iend) {
    if (!G.nsp && ((B.south && j == B.Jstr) || (B.north && j == B.Jend + 1))) {
      Va[X3(i, j, k)] = 0.0;                                       // :1187-1220
    } else if (j >= B.JstrV && j <= B.Jendp1) {
      const double cff1 = KMIN(KMIN(bdn[X3(i, j - 1, k)], bup[X3(i, j, k)]), 1.0);
      const double cff2 = KMIN(KMIN(bup[X3(i, j - 1, k)], bdn[X3(i, j, k)]), 1.0);
      Va[X3(i, j, k)] = (cff1 * KMAX(0.0, Va[X3(i, j, k)]) + cff2 * KMIN(0.0, Va[X3(i, j, k)])) * cff * F.on_v[X2(i, j)];
      if (G.masking) Va[X3(i, j, k)] = Va[X3(i, j, k)] * G.vmask[X2(i, j)];     // :1129
      if (G.wet_dry) Va[X3(i, j, k)] = Va[X3(i, j, k)] * F.vmask_wet[X2(i, j)];  // WET_DRY :1133
    }
  }
  if (i <= B.Iend && j <= B.Jend && k < N) {
    const double cff1 = KMIN(KMIN(bdn[X3(i, j, k)], bup[X3(i, j, k + 1)]), 1.0);
    const double cff2 = KMIN(KMIN(bup[X3(i, j, k)], bdn[X3(i, j, k + 1)]), 1.0);
    Wa[XW(i, j, k)] = (cff1 * KMAX(0.0, Wa[XW(i, j, k)]) + cff2 * KMIN(0.0, Wa[XW(i, j, k)])) * cff * F.omn[X2(i, j)] *
                      (z_r[X3(i, j, k + 1)] - z_r[X3(i, j, k)]);
    if (G.masking) Wa[XW(i, j, k)] = Wa[XW(i, j, k)] * G.rmask[X2(i, j)];       // :1145
    if (G.wet_dry) Wa[XW(i, j, k)] = Wa[XW(i, j, k)] * F.rmask_wet[X2(i, j)];    // WET_DRY :1149
  }
}
THREAD_GLOBAL(k_mp_limit, MpArgs)

// ---- corrected fluxes :1399-1500: t(nnew) = Ta*Hz - div(anti-diffusive fluxes); (Istr:Iend, Jstr:Jend, 1:N)
THREAD_KERNEL(k_mp_apply, MpArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, itrc = a.itrc;
  const int i = B.Istr + gx, j = B.Jstr + gy, k = gz + 1;
  const double *Ta = F.mp3[0] + (size_t)(itrc - 1) * G.nij * N, *Ua = F.mp3[1], *Va = F.mp3[2], *Wa = F.mp3[3];
  const double *Hz = F.Hz;
  double *tn = F.t + XT(G.LBi, G.LBj, 1, G.nnew, itrc);
#define MP_FX(ii) ((KMAX(Ua[X3(ii, j, k)], 0.0) * MP_TA((ii) - 1, j, k) + KMIN(Ua[X3(ii, j, k)], 0.0) * MP_TA(ii, j, k)) * 0.5 * \
                   (Hz[X3(ii, j, k)] + Hz[X3((ii) - 1, j, k)]) * F.on_u[X2(ii, j)])
#define MP_FE(jj) ((KMAX(Va[X3(i, jj, k)], 0.0) * MP_TA(i, (jj) - 1, k) + KMIN(Va[X3(i, jj, k)], 0.0) * MP_TA(i, jj, k)) * 0.5 * \
                   (Hz[X3(i, jj, k)] + Hz[X3(i, (jj) - 1, k)]) * F.om_v[X2(i, jj)])
#define MP_FC(kk) (((kk) <= 0 || (kk) >= N) ? 0.0 : KMAX(Wa[XW(i, j, kk)], 0.0) * MP_TA(i, j, kk) + KMIN(Wa[XW(i, j, kk)], 0.0) * MP_TA(i, j, (kk) + 1))
  const double cff = G.dt * F.pm[X2(i, j)] * F.pn[X2(i, j)];
  const double cff1 = cff * (MP_FX(i + 1) - MP_FX(i));
  const double cff2 = cff * (MP_FE(j + 1) - MP_FE(j));
  const double cff3 = cff1 + cff2;
  double t1 = MP_TA(i, j, k) * Hz[X3(i, j, k)] - cff3;
  const double cv = cff * (MP_FC(k) - MP_FC(k - 1));
  t1 = t1 - cv;
#undef MP_FX
#undef MP_FE
#undef MP_FC
  tn[X3(i, j, k)] = t1;
}
THREAD_GLOBAL(k_mp_apply, MpArgs)

// ---- k_mp_limit + k_mp_apply in one kernel: every thread limits the six anti-diffusive velocities
//      of its cell itself (a face velocity is limited by both cells that share it: same expression,
//      same bits) instead of a pass that rewrites Ua, Va, Wa in place and a second one that reads them;
//      Ua, Va, Wa stay unlimited in memory (nothing else reads them).  (Istr:Iend, Jstr:Jend, 1:N)
// BUP(ii, jj, kk), BDN(ii, jj, kk): the FCT ratios of a point -- the arrays of k_mp_beta, or the LDS levels of k_mp_limfused
template <class FU, class FD>
KDEV void mp_limapply_pt(const DGrid &G, const Fields &F, int itrc, int i, int j, int k, FU BUP, FD BDN) {
  const TB &B = G.T;
  const int N = G.N;
  const double *Ta = F.mp3[0] + (size_t)(itrc - 1) * G.nij * N, *Ua = F.mp3[1], *Va = F.mp3[2], *Wa = F.mp3[3];
  const double *z_r = F.z_r, *Hz = F.Hz;
  double *tn = F.t + XT(G.LBi, G.LBj, 1, G.nnew, itrc);
  const double odt = 1.0 / G.dt;
  const bool wc = !G.ewp && B.west, ec = !G.ewp && B.east, sc = !G.nsp && B.south, nc = !G.nsp && B.north;
  const double cff = G.dt * F.pm[X2(i, j)] * F.pn[X2(i, j)];
  // limited Ua at (ii,j,k) :1100-1185, Va at (i,jj,k) :1187-1220, Wa at w-level kk :1130-1150
#define MP_LU(ii)                                                                                         \
  (((wc && (ii) == B.Istr) || (ec && (ii) == B.Iend + 1))                                                 \
       ? 0.0                                                                                              \
       : (KMIN(KMIN(BDN((ii) - 1, j, k), BUP(ii, j, k)), 1.0) * KMAX(0.0, Ua[X3(ii, j, k)]) +     \
          KMIN(KMIN(BUP((ii) - 1, j, k), BDN(ii, j, k)), 1.0) * KMIN(0.0, Ua[X3(ii, j, k)])) *    \
             odt * F.om_u[X2(ii, j)] * (G.masking ? G.umask[X2(ii, j)] : 1.0) * (G.wet_dry ? F.umask_wet[X2(ii, j)] : 1.0))
#define MP_LV(jj)                                                                                         \
  (((sc && (jj) == B.Jstr) || (nc && (jj) == B.Jend + 1))                                                 \
       ? 0.0                                                                                              \
       : (KMIN(KMIN(BDN(i, (jj) - 1, k), BUP(i, jj, k)), 1.0) * KMAX(0.0, Va[X3(i, jj, k)]) +     \
          KMIN(KMIN(BUP(i, (jj) - 1, k), BDN(i, jj, k)), 1.0) * KMIN(0.0, Va[X3(i, jj, k)])) *    \
             odt * F.on_v[X2(i, jj)] * (G.masking ? G.vmask[X2(i, jj)] : 1.0) * (G.wet_dry ? F.vmask_wet[X2(i, jj)] : 1.0))
#define MP_LW(kk)                                                                                         \
  ((KMIN(KMIN(BDN(i, j, kk), BUP(i, j, (kk) + 1)), 1.0) * KMAX(0.0, Wa[XW(i, j, kk)]) +           \
    KMIN(KMIN(BUP(i, j, kk), BDN(i, j, (kk) + 1)), 1.0) * KMIN(0.0, Wa[XW(i, j, kk)])) *          \
   odt * F.omn[X2(i, j)] * (z_r[X3(i, j, (kk) + 1)] - z_r[X3(i, j, kk)]) * (G.masking ? G.rmask[X2(i, j)] : 1.0) * (G.wet_dry ? F.rmask_wet[X2(i, j)] : 1.0))
  const double ua0 = MP_LU(i), ua1 = MP_LU(i + 1), va0 = MP_LV(j), va1 = MP_LV(j + 1);
  const double ta = MP_TA(i, j, k);
#define MP_FX(ua_, ii) ((KMAX(ua_, 0.0) * MP_TA((ii) - 1, j, k) + KMIN(ua_, 0.0) * MP_TA(ii, j, k)) * 0.5 * \
                        (Hz[X3(ii, j, k)] + Hz[X3((ii) - 1, j, k)]) * F.on_u[X2(ii, j)])
#define MP_FE(va_, jj) ((KMAX(va_, 0.0) * MP_TA(i, (jj) - 1, k) + KMIN(va_, 0.0) * MP_TA(i, jj, k)) * 0.5 * \
                        (Hz[X3(i, jj, k)] + Hz[X3(i, (jj) - 1, k)]) * F.om_v[X2(i, jj)])
  const double cff1 = cff * (MP_FX(ua1, i + 1) - MP_FX(ua0, i));
  const double cff2 = cff * (MP_FE(va1, j + 1) - MP_FE(va0, j));
  const double cff3 = cff1 + cff2;
  double t1 = ta * Hz[X3(i, j, k)] - cff3;
  double fc1 = 0.0, fc0 = 0.0;                     // FC(k), FC(k-1); zero at the bottom and at the surface
  if (k < N) { const double w = MP_LW(k); fc1 = KMAX(w, 0.0) * ta + KMIN(w, 0.0) * MP_TA(i, j, k + 1); }
  if (k > 1) { const double w = MP_LW(k - 1); fc0 = KMAX(w, 0.0) * MP_TA(i, j, k - 1) + KMIN(w, 0.0) * ta; }
  const double cv = cff * (fc1 - fc0);
  t1 = t1 - cv;
  tn[X3(i, j, k)] = t1;
#undef MP_LU
#undef MP_LV
#undef MP_LW
#undef MP_FX
#undef MP_FE
}
THREAD_KERNEL(k_mp_limapply, MpArgs) {
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const double *bup = F.mp3[4], *bdn = F.mp3[5];
  mp_limapply_pt(G, F, a.itrc, G.T.Istr + gx, G.T.Jstr + gy, gz + 1,
                 [&](int ii, int jj, int kk) { return bup[X3(ii, jj, kk)]; },
                 [&](int ii, int jj, int kk) { return bdn[X3(ii, jj, kk)]; });
}
THREAD_GLOBAL(k_mp_limapply, MpArgs)

// ---- implicit vertical diffusion, plain tridiagonal :1724-1790; one thread per column ---------
THREAD_KERNEL(k_mp_vdiff, MpArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, itrc = a.itrc, ltrc = KMIN(G.NAT, itrc);
  const int i = B.Istr + gx, j = B.Jstr + gy;
  const double *Hz = F.Hz, *z_r = F.z_r;
  const double *Akt = F.Akt + (size_t)(ltrc - 1) * G.nij * (N + 1);
  double *tn = F.t + XT(G.LBi, G.LBj, 1, G.nnew, itrc);
  double *CF = F.mp3[4], *DC = F.mp3[5];            // beta arrays are free now
  const double cff = -G.dt * G.lambda;
#define MP_FCD(kk) (((kk) <= 0 || (kk) >= N) ? 0.0 : cff * (1.0 / (z_r[X3(i, j, (kk) + 1)] - z_r[X3(i, j, kk)])) * Akt[XW(i, j, kk)])
  {
    const double BC1 = Hz[X3(i, j, 1)] - MP_FCD(1) - MP_FCD(0);
    const double c = 1.0 / BC1;
    CF[X3(i, j, 1)] = c * MP_FCD(1);
    DC[X3(i, j, 1)] = c * tn[X3(i, j, 1)];
  }
  for (int k = 2; k <= N - 1; k++) {
    const double BCk = Hz[X3(i, j, k)] - MP_FCD(k) - MP_FCD(k - 1);
    const double c = 1.0 / (BCk - MP_FCD(k - 1) * CF[X3(i, j, k - 1)]);
    CF[X3(i, j, k)] = c * MP_FCD(k);
    DC[X3(i, j, k)] = c * (tn[X3(i, j, k)] - MP_FCD(k - 1) * DC[X3(i, j, k - 1)]);
  }
  {
    const double BCN = Hz[X3(i, j, N)] - MP_FCD(N) - MP_FCD(N - 1);
    const double d = (tn[X3(i, j, N)] - MP_FCD(N - 1) * DC[X3(i, j, N - 1)]) / (BCN - MP_FCD(N - 1) * CF[X3(i, j, N - 1)]);
    DC[X3(i, j, N)] = d;
    tn[X3(i, j, N)] = d;
  }
  for (int k = N - 1; k >= 1; k--) {
    const double d = DC[X3(i, j, k)] - CF[X3(i, j, k)] * DC[X3(i, j, k + 1)];
    DC[X3(i, j, k)] = d;
    tn[X3(i, j, k)] = d;
  }
#undef MP_FCD
}
THREAD_GLOBAL(k_mp_vdiff, MpArgs)

// COL form: CF/DC of the solve in LDS (2*(N+1) doubles per column) instead of two 3-D work arrays, the
// loads of the next six levels in flight while the recurrence runs on the current six; t(nnew) is read
// once and written once.
COL_KERNEL(k_mp_vdiff_l, MpArgs) {
  (void)gz;
  const DGrid &G = a.G;
  const Fields &F = a.Fv;
  const TB &B = G.T;
  const int N = G.N, itrc = a.itrc, ltrc = KMIN(G.NAT, itrc);
  const int i = B.Istr + gx, j = B.Jstr + gy;
  const double *Hz = F.Hz, *z_r = F.z_r;
  const double *Akt = F.Akt + (size_t)(ltrc - 1) * G.nij * (N + 1);
  double *tn = F.t + XT(G.LBi, G.LBj, 1, G.nnew, itrc);
  double *CF = lds, *DC = lds + (size_t)(N + 1) * KLS;
  const double cff = -G.dt * G.lambda;
  constexpr int CH = 6;
  double nh[CH], nt[CH], nz[CH + 1], na[CH];
#define VD_LOAD(kb)                                                                                    \
  do {                                                                                                 \
    _Pragma("unroll") for (int q = 0; q < CH; q++) {                                                   \
      const int k = KMIN((kb) + q, N);                                                                 \
      nh[q] = Hz[X3(i, j, k)]; nt[q] = tn[X3(i, j, k)]; na[q] = Akt[XW(i, j, k)];                      \
    }                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < CH + 1; q++) nz[q] = z_r[X3(i, j, KMIN((kb) + q, N))];       \
  } while (0)
  double FCm = 0.0, CFm = 0.0, DCm = 0.0;     // FCD(k-1), CF(k-1), DC(k-1)
  VD_LOAD(1);
  _Pragma("unroll 1") for (int k0 = 1; k0 <= N; k0 += CH) {
    double hz[CH], tt[CH], zr[CH + 1], ak[CH];
#pragma unroll
    for (int q = 0; q < CH; q++) { hz[q] = nh[q]; tt[q] = nt[q]; ak[q] = na[q]; }
#pragma unroll
    for (int q = 0; q < CH + 1; q++) zr[q] = nz[q];
    KSCHED_FENCE();
    if (k0 + CH <= N) VD_LOAD(k0 + CH);
    KSCHED_FENCE();
#pragma unroll
    for (int q = 0; q < CH; q++) {
      const int k = k0 + q;
      if (k > N) break;
      const double FCk = (k >= N) ? 0.0 : cff * (1.0 / (zr[q + 1] - zr[q])) * ak[q];
      const double BCk = hz[q] - FCk - FCm;
      if (k == 1) {
        const double c = 1.0 / BCk;
        CFm = c * FCk;
        DCm = c * tt[q];
      } else if (k <= N - 1) {
        const double c = 1.0 / (BCk - FCm * CFm);
        CFm = c * FCk;
        DCm = c * (tt[q] - FCm * DCm);
      } else {
        DCm = (tt[q] - FCm * DCm) / (BCk - FCm * CFm);
      }
      CF[k * KLS] = CFm;
      DC[k * KLS] = DCm;
      FCm = FCk;
    }
  }
#undef VD_LOAD
  double d = DC[N * KLS];
  tn[X3(i, j, N)] = d;
  _Pragma("unroll 4") for (int k = N - 1; k >= 1; k--) {
    d = DC[k * KLS] - CF[k * KLS] * d;
    tn[X3(i, j, k)] = d;
  }
}
COL_GLOBAL(k_mp_vdiff_l, MpArgs)
