// roms_ctx.h -- device-side descriptors and the host context of libroms_hip.so.
#pragma once
#include "kdefs.h"
#include "../../include/roms_hip.h"

// ---------------------------------------------------------------------------------------------
// Loop bounds of a (sub-)tile: BOUNDS(ng)%xxx(tile) of the reference, recomputed by the rules of
// var_bounds (ROMS/Utility/get_bounds.F:1044-1884).  The same rules are applied to the whole
// GPU tile and to the sub-tile owned by one thread block.
// ---------------------------------------------------------------------------------------------
struct TB {
  int Istr, Iend, Jstr, Jend;
  int IstrR, IendR, JstrR, JendR;
  int IstrU, JstrV;
  int IstrB, IendB, IstrM, JstrB, JendB, JstrM;
  int IstrP, IendP, JstrP, JendP;
  int IstrT, IendT, JstrT, JendT;
  int Istrm3, Istrm2, Istrm1, IstrUm2, IstrUm1;
  int Iendp1, Iendp2, Iendp2i, Iendp3;
  int Jstrm3, Jstrm2, Jstrm1, JstrVm2, JstrVm1;
  int Jendp1, Jendp2, Jendp2i, Jendp3;
  int west, east, south, north;   // DOMAIN%Western_Edge ... (of the global domain)
  int sw, se, nw, ne;
};

// Grid/time/physics descriptor passed BY VALUE to every kernel.
struct DGrid {
  // allocation
  int LBi, LBj, ni, nj;
  long nij;
  int N, NT, NAT, Lm, Mm, Nghost;
  int ewp, nsp, options;
  int hadv[ROMS_MAXT], vadv[ROMS_MAXT];
  TB T;                       // bounds of this GPU's tile
  int nbx, nby;               // thread-block decomposition of the tile for the 3-D COOP kernels
  int bw, bh;                 // max sub-tile extent (LDS scratch is (bw+6) x (bh+6))
  int dbg_stop;
  // rim / interior split of a 3-D producer in front of its strip exchange (multi-tile contexts, round 4): 0 = the whole
  // index space; 1 = only the points within `rimw` lines of the edge of the launch's (xi,eta) index space -- what the
  // exchange packs and its boundary fills read --, 2 = only the others (kdefs.h: THREAD / COL launches decide per thread,
  // the LDS-tiled kernels per block)
  int region, rimw;
  int fuse_halo;              // 1: single tile, k_step2d fills boundary/periodic ghost points itself (k_haloblock.h)
  int fuse3d;                 // 1: the 3-D producers do so too (emit_plan/emit_store); ROMS_HIP_FUSE3D=0 turns it off
  int xloc, yloc;             // 1: a periodic direction of this tile wraps onto itself by a local copy (no exchange partner)
  int xgl, xgh;               // ghost lines a strip exchange fills on the low | high side of a tile: 3 | Nghost (the reference's
                              // periodic layout), or B2D_GL | B2D_GH for the exchanges behind the barotropic pair kernel
  int xmap2;                  // barotropic launches: block -> sub-tile through xcd_remap2 (below)
  int nbx2, nby2, bw2, bh2;   // the same for the 2-D (barotropic) kernel: smaller sub-tiles, the
                              // 2-D grid alone cannot fill 256 CUs otherwise
  // stepping (mod_stepping)
  int iic, iif, nstp, nnew, nrhs, kstp, knew, krhs, predictor;
  int ntfirst, nfast;
  double time, tdays;
  // scalars
  double dt, dtfast, rho0, g, lambda, gamma2, Cp, R0, T0, S0, Tcoef, Scoef, hc, dstart;
  double Akt_bak[ROMS_MAXT], Akv_bak;
  double Zob;                 // bottom roughness (UV_LOGDRAG: GRID%ZoBot = Zob, mod_grid.F:1380)
  // MASKING (mod_grid.F rmask, umask, vmask, pmask): land/sea masks, 1 water / 0 land (pmask 2: no-slip); the arrays
  // are Fields::rmask ... (upload names "rmask", "umask", "vmask", "pmask"), here for the kernels that get no Fields
  int masking;
  const double *rmask, *umask, *vmask, *pmask;
  int Vtransform;
  // open boundaries: 1 if any edge of any variable is neither closed nor periodic (k_obc.h does the state's boundary
  // conditions then, and nothing is fused into the producers); bit (4*variable + edge) of lbc_closed set where
  // LBC(edge,variable)%closed (bc_2d.F:201, mpdata_adiff.F:698: closed, or else zero gradient)
  int obc;
  unsigned long long lbc_closed;     // (64 bits: 4 edges x (ROMS_ISTVAR + NT) variables)
  // ---- members of later rounds, kept BEHIND the ones above: a kernel fetches its arguments through the scalar cache, and the
  // members the hot kernels read stay packed in the cache lines they were measured in (profiles/)
  // DIAGNOSTICS_TS (mod_diags.F): 0 = off; else NDT, the number of tracer terms; dia_idx[term] = the reference's 1-based
  // index of the term (0 = absent for this option set), terms in the order of the enum below
  int dia_ts, dia_idx[10];
  // DIAGNOSTICS_UV (mod_diags.F:174-222): 0 = off; m2 / m3[term] = the reference's 1-based index of a 2-D / 3-D momentum
  // term for the option set (mod_scalars.F:4264-4377), 0 = absent; ndm2 = NDM2d, ndm3 = NDM3d, ndrhs = NDrhs
  int dia_uv;
  // biharmonic horizontal mixing along s-surfaces switched on (option bits ROMS_UV_VIS4, ROMS_TS_DIF4): UV_VIS4 + MIX_S_UV (uv3dmix4_s.h,
  // step2d_LF_AM3.h:1653-1920), TS_DIF4 + MIX_S_TS (t3dmix4_s.h); coefficient arrays visc4_r, visc4_p, diff4
  int uv_vis4, ts_dif4;
  int clima;           // climatology nudging: bit 0 LnudgeM3CLM (rhs3d.F:654), bit itrc LtracerCLM & LnudgeTCLM of tracer itrc (step3d_t.F:1866)
  int mix_geo_uv;      // UV_VIS2 along geopotential surfaces (option bit ROMS_MIX_GEO_UV; k_uvmix_geo.h, work arrays Fields::gwrk)
  signed char m2[12], m3[12];
  short ndm2, ndm3, ndrhs;
  // WET_DRY (wetdry.F; option bit ROMS_WET_DRY): time-dependent wet/dry masks rmask_wet ... (Fields::rmask_wet ...; here for
  // the kernels that get no Fields), Dcrit = DCRIT of roms.in, hbath = h
  int wet_dry;
  double Dcrit;
  const double *rmask_wet, *umask_wet, *vmask_wet, *pmask_wet, *hbath;
  // round 6, behind everything else (no member above moves)
  int prs4x;           // 44: PJ_GRADPQ4 (prsgrd44.h), 42: PJ_GRADPQ2 (prsgrd42.h); 0: the scheme the lower option bits name (k_prs4x.h)
  int ddmix;           // LMD_DDMIX (option bit ROMS_LMD_DDMIX): double-diffusive mixing in lmd_vmix's interior scheme (k_lmd.h), from ...
  double *alfaobeta;   // ... the ratio of the thermal expansion and saline contraction coefficients (i,j,0:N) rho_eos leaves (rho_eos.F:454, :794)
  double *vcons;       // VolCons: {bc_area, bc_flux, ubar_xs} of mod_scalars.F:1460-1462 on the device (k_obc.h: k_obc_flux; k_step2d.h reads ubar_xs)
  int volcons;         // ... bits by edge (obc_volcons.F)
  int bkpp;            // LMD_BKPP (option bit ROMS_LMD_BKPP): the bottom boundary layer behind lmd_skpp (k_lmd.h: k_lmd_bkpp, lmd_bkpp.F; Fields::hbbl, ksbl)
};

#ifdef ROMS_CPU_EMU
#define KHD inline
#else
#define KHD __host__ __device__ __forceinline__
#endif

// WET_DRY: the factor the barotropic step applies at a velocity point (step2d_LF_AM3.h:2208-2210, 2519-2521 ...): the reference's
// cff7 = 0.5*mask*cff5 + cff6*(1-cff5), cff5 = ABS(ABS(mask)-1), cff6 = 0.5 + DSIGN(0.5, value)*mask
KHD double wd_fac(double mw, double val) {
  const double cff5 = fabs(fabs(mw) - 1.0);
  const double cff6 = 0.5 + copysign(0.5, val) * mw;
  return 0.5 * mw * cff5 + cff6 * (1.0 - cff5);
}

// var_bounds for the rectangle [i0,i1]x[j0,j1]; w/e/s/n: the rectangle touches that edge of the
// global domain.
KHD TB make_bounds(int Lm, int Mm, int ewp, int nsp, int i0, int i1, int j0, int j1, int w, int e, int s,
                   int n) {
  TB b;
  b.west = w; b.east = e; b.south = s; b.north = n;
  b.sw = w && s; b.se = e && s; b.nw = w && n; b.ne = e && n;
  const int pw = w && !ewp, pe = e && !ewp, ps = s && !nsp, pn = n && !nsp;
  b.Istr = i0; b.IstrP = i0;
  b.IstrR = pw ? i0 - 1 : i0;
  b.IstrT = b.IstrR;
  b.IstrU = pw ? i0 + 1 : i0;
  b.IstrB = pw ? b.IstrT + 1 : i0;
  b.IstrM = pw ? b.IstrP + 1 : b.IstrU;
  b.Istrm3 = pw ? KMAX(0, i0 - 3) : i0 - 3;
  b.Istrm2 = pw ? KMAX(0, i0 - 2) : i0 - 2;
  b.Istrm1 = pw ? KMAX(1, i0 - 1) : i0 - 1;
  b.IstrUm2 = pw ? KMAX(1, b.IstrU - 2) : b.IstrU - 2;
  b.IstrUm1 = pw ? KMAX(2, b.IstrU - 1) : b.IstrU - 1;
  b.Iend = i1;
  b.IendR = pe ? i1 + 1 : i1;
  b.IendP = b.IendR; b.IendT = b.IendR;
  b.IendB = pe ? b.IendT - 1 : i1;
  b.Iendp1 = pe ? KMIN(i1 + 1, Lm) : i1 + 1;
  b.Iendp2i = pe ? KMIN(i1 + 2, Lm) : i1 + 2;
  b.Iendp2 = pe ? KMIN(i1 + 2, Lm + 1) : i1 + 2;
  b.Iendp3 = pe ? KMIN(i1 + 3, Lm + 1) : i1 + 3;
  b.Jstr = j0; b.JstrP = j0;
  b.JstrR = ps ? j0 - 1 : j0;
  b.JstrT = b.JstrR;
  b.JstrV = ps ? j0 + 1 : j0;
  b.JstrB = ps ? b.JstrT + 1 : j0;
  b.JstrM = ps ? b.JstrP + 1 : b.JstrV;
  b.Jstrm3 = ps ? KMAX(0, j0 - 3) : j0 - 3;
  b.Jstrm2 = ps ? KMAX(0, j0 - 2) : j0 - 2;
  b.Jstrm1 = ps ? KMAX(1, j0 - 1) : j0 - 1;
  b.JstrVm2 = ps ? KMAX(1, b.JstrV - 2) : b.JstrV - 2;
  b.JstrVm1 = ps ? KMAX(2, b.JstrV - 1) : b.JstrV - 1;
  b.Jend = j1;
  b.JendR = pn ? j1 + 1 : j1;
  b.JendP = b.JendR; b.JendT = b.JendR;
  b.JendB = pn ? b.JendT - 1 : j1;
  b.Jendp1 = pn ? KMIN(j1 + 1, Mm) : j1 + 1;
  b.Jendp2i = pn ? KMIN(j1 + 2, Mm) : j1 + 2;
  b.Jendp2 = pn ? KMIN(j1 + 2, Mm + 1) : j1 + 2;
  b.Jendp3 = pn ? KMIN(j1 + 3, Mm + 1) : j1 + 3;
  return b;
}

// Sub-tile of thread block (bx,by): the GPU tile is split into nbx x nby nearly equal
// rectangles with the reference's own partition rule (tile_bounds_2d, get_bounds.F:972-1042).
// Consecutive blockIdx.x values land on different XCDs (block b -> XCD b%8); the i-fastest
// linear order keeps eta-neighbouring sub-tiles of one XCD's L2 8 blocks apart, which is
// irrelevant for correctness.
KHD TB block_bounds_n(const DGrid &G, int nbx, int nby, int bx, int by) {
  const int LmT = G.T.Iend - G.T.Istr + 1, MmT = G.T.Jend - G.T.Jstr + 1;
  const int cI = (LmT + nbx - 1) / nbx, cJ = (MmT + nby - 1) / nby;
  const int mI = (nbx * cI - LmT) / 2, mJ = (nby * cJ - MmT) / 2;
  int i0 = 1 + bx * cI - mI, i1 = i0 + cI - 1;
  int j0 = 1 + by * cJ - mJ, j1 = j0 + cJ - 1;
  i0 = KMAX(i0, 1); i1 = KMIN(i1, LmT);
  j0 = KMAX(j0, 1); j1 = KMIN(j1, MmT);
  i0 += G.T.Istr - 1; i1 += G.T.Istr - 1;
  j0 += G.T.Jstr - 1; j1 += G.T.Jstr - 1;
  return make_bounds(G.Lm, G.Mm, G.ewp, G.nsp, i0, i1, j0, j1, G.T.west && bx == 0, G.T.east && bx == nbx - 1,
                     G.T.south && by == 0, G.T.north && by == nby - 1);
}
KHD TB block_bounds(const DGrid &G, int bx, int by) { return block_bounds_n(G, G.nbx, G.nby, bx, by); }
KHD TB block_bounds2(const DGrid &G, int bx, int by) { return block_bounds_n(G, G.nbx2, G.nby2, bx, by); }
// XCD-aware block order of the barotropic launches (round 4).  The hardware hands workgroup b = bx + nbx2*by to XCD
// b % 8, so with the plain order the xi-neighbours of a sub-tile sit on seven OTHER L2s and every block fetches the
// 128-byte lines at the two ends of its 38-point rows for itself.  Remapped, the sub-tiles (numbered xi-fastest) are
// cut into 8 contiguous segments -- whole rows of sub-tiles -- one per XCD: the blocks resident on an XCD at a time are
// neighbours in xi AND eta, and the rim lines they share are fetched into that L2 once.  Only the assignment of blocks
// to sub-tiles changes (same bits).  xmap2 is set when the number of sub-tiles is a multiple of 8.
KHD void xcd_remap2(const DGrid &G, int &bx, int &by) {
  if (!G.xmap2) return;
  if (G.xmap2 == 2) {
    // (round 6, the persistent loop: ROMS_HIP_LOOP_XCD=patch) the sub-tiles of an XCD form a (nbx2/4) x (nby2/2) PATCH instead of
    // whole rows: blocks that hand their rims to each other pair after pair then share an L2 on all four sides but the patch edge
    const int lin = bx + G.nbx2 * by, xcd = lin & 7, r = lin >> 3, pw = G.nbx2 >> 2, ph = G.nby2 >> 1;
    bx = (xcd & 3) * pw + r % pw;
    by = (xcd >> 2) * ph + r / pw;
    return;
  }
  const int lin = bx + G.nbx2 * by, seg = (G.nbx2 * G.nby2) >> 3;
  const int t = (lin & 7) * seg + (lin >> 3);
  by = t / G.nbx2;
  bx = t - by * G.nbx2;
}

// ghost lines the barotropic pair kernel reads beyond the tile (k_step2d_pair.h): each of its two step2d calls consumes
// three lines on the low side and two on the high side, so a pair needs 5 | 4 (2-D fields of a multi-tile run)
#define B2D_GL 5
#define B2D_GH 4

// boundary-fill kinds of the halo code (k_halo.h, k_haloblock.h)
enum { BC_NONE = 0, BC_R = 1, BC_U = 2, BC_V = 3 };
// MASKING, added to BC_R by the launch sequences of a masked run (never seen by the fused-halo paths, which a masked
// run does not take): BC_MASKF = the gradient value at a closed edge is multiplied by rmask of the boundary point
// (zetabc.F:264, t3dbc_im.F:214; bc_r2d/bc_w3d fills carry no mask), BC_MASKALL = after the fills the whole plane
// IstrR:IendR x JstrR:JendR is multiplied by rmask (step3d_t.F:1880-1890).  The slip values of BC_U / BC_V are
// always multiplied by umask / vmask of the boundary point in a masked run (u2dbc_im.F:989, bc_2d.F:252 ...).
enum { BC_MASKF = 16, BC_MASKALL = 32, BC_KIND = 15,
       BC_LBC2D = 64,    // bc_u2d / bc_v2d of a context with open edges: closed where LBC(:,isUbar / isVbar) is, else zero gradient
       // WET_DRY: BC_WET2 = the "wetting and drying conditions" at the end of zetabc.F:783-874 (BC_R), u2dbc_im.F:1190-1318
       // (BC_U), v2dbc_im.F:1239-1367 (BC_V); BC_WET3 = the slip value of u3dbc / v3dbc times the wet mask of the boundary
       // point (u3dbc_im.F:523,681, v3dbc_im.F:...)
       BC_WET2 = 128, BC_WET3 = 256 };   // bc_u2d / bc_v2d of a context with open edges: closed where LBC(:,isUbar / isVbar) is, else zero gradient

// ------------------------------------------------------------------ indexing (reference layout)
#define X2(i, j) ((size_t)((i) - G.LBi) + (size_t)((j) - G.LBj) * (size_t)G.ni)
#define X3(i, j, k) (X2(i, j) + (size_t)((k) - 1) * (size_t)G.nij)   /* rho levels 1..N */
#define XW(i, j, k) (X2(i, j) + (size_t)(k) * (size_t)G.nij)         /* w levels 0..N   */
// LDS scratch of a sub-tile b: (IminS:ImaxS,JminS:JmaxS) = (Istr-3:Iend+3, Jstr-3:Jend+3)
#define SW_(b) ((b).Iend - (b).Istr + 7)
#define S2(i, j) ((size_t)((i) - (B.Istr - 3)) + (size_t)((j) - (B.Jstr - 3)) * (size_t)SW_(B))

// Device array pointer as stored in the pointer table.  The table lives in device memory, so to the
// compiler a pointer loaded from it is a generic (flat) address: every access becomes flat_load /
// flat_store with a 64-bit VALU address computation.  Typing the stored pointer as global
// (address space 1) lets the address-space inference turn all accesses derived from it into
// global_load / global_store with scalar base + 32-bit offset addressing.  Converts to a plain
// double* everywhere, so kernel code is unaffected.
#ifdef ROMS_CPU_EMU
typedef double gdouble_t;
#else
typedef __attribute__((address_space(1))) double gdouble_t;
#endif
struct GPtr {
  gdouble_t *p;
  KHD operator double *() const { return (double *)p; }
  KHD GPtr &operator=(double *q) { p = (gdouble_t *)q; return *this; }
};

// terms of the momentum diagnostics (DIAGNOSTICS_UV), slots of DGrid::m2 / m3
enum { M2FCOR = 0, M2HADV, M2XADV, M2YADV, M2HVIS, M2XVIS, M2YVIS, M2PGRD, M2SSTR, M2BSTR, M2RATE, M2NTERMS };
enum { M3FCOR = 0, M3VADV, M3HADV, M3XADV, M3YADV, M3PGRD, M3VVIS, M3HVIS, M3XVIS, M3YVIS, M3RATE, M3NTERMS };
enum { DIA_HADV = 0, DIA_XADV, DIA_YADV, DIA_VADV, DIA_HDIF, DIA_XDIF, DIA_YDIF, DIA_SDIF, DIA_VDIF, DIA_RATE, DIA_NTERMS };

// All device arrays (reference component names).  Pointers only; passed to kernels through the
// small per-kernel argument structs.
struct Fields {
  // mod_grid
  GPtr h, f, fomn, pm, pn, om_r, on_r, om_u, on_u, om_v, on_v, om_p, on_p, omn, pmon_r, pnom_r,
      pmon_p, pnom_p, pmon_u, pnom_u, pmon_v, pnom_v, dmde, dndx, angler, xr, yr, xp, yp, lonr, latr, rdrag,
      rdrag2, rmask, umask, vmask, pmask;
  GPtr Hz, z_r, z_w, Huon, Hvom;
  // mod_ocean
  GPtr zeta, ubar, vbar, rzeta, rubar, rvbar, u, v, t, W, wvel, rho, pden, ru, rv;
  // mod_coupling
  GPtr rhoA, rhoS, rufrc, rvfrc, Zt_avg1, DU_avg1, DU_avg2, DV_avg1, DV_avg2;
  // mod_forces
  GPtr sustr, svstr, bustr, bvstr, stflx, btflx, stflux, btflux, srflx;
  GPtr Uwind, Vwind, Tair, Pair, Hair, rain, cloud, lhflx, shflx, lrflx, evap;
  // mod_mixing
  GPtr Akv, Akt, visc2_r, visc2_p, diff2, bvf, alpha, beta, hsbl, ghats;
  GPtr tke, gls, Lscale, Akk, Akp;   // GLS_MIXING: tke, gls (i,j,0:N,3); Lscale, Akk, Akp (i,j,0:N)
  // s-coordinate tables (device copies)
  GPtr sc_r, Cs_r, sc_w, Cs_w;
  // work space: private 3-D arrays of the reference kernels (P of prsgrd, vert of wvelocity,
  // oHz/Ta/Ua/Va/Wa of step3d_t, swdk of pre_step3d ...)
  GPtr wrk3[13];   // [1] P of prsgrd, [0..4] KPP, [3..4] spline fluxes, [5] swdk, [6..9] the four viscous terms of
                   // uv3dmix2, [10] wvelocity, [11..12] the old ru/rv bracket of the deferred momentum predictor
  GPtr wrk2[4];
  // MPDATA work arrays (allocated only when a tracer uses MPDATA): Ta (N planes per tracer), Ua, Va, Wa,
  // beta_up, beta_dn
  GPtr mp3[6];
  // packed metrics of the barotropic momentum stage, 8 doubles per grid point (k_step2d.h: M2Rec)
  GPtr m2r, m2p;
  // open-boundary data, BOUNDARY(ng)%zeta_west ... t_north of mod_boundary.F: [6 * edge-order + ...] = variable
  // (zeta, ubar, vbar, u, v, t) * 4 + (west, east, south, north); west/east lines (LBj:UBj [,N [,NT]]) in the caller's
  // bounds, south/north (LBi:UBi ...)
  GPtr bry[24];
  // ---- arrays of later rounds, behind the ones above (see DGrid)
  GPtr visc4_r, visc4_p, diff4;        // square roots of the biharmonic coefficients (inp_par.F:634, read_phypar.F:7840)
  // WET_DRY (wetdry.F): wet/dry masks of the fast steps / the 3-D step, wet x land masks for output, the sum of the rho
  // mask over the fast steps (allocated with the option bit ROMS_WET_DRY)
  GPtr rmask_wet, umask_wet, vmask_wet, pmask_wet, rmask_full, umask_full, vmask_full, pmask_full, rmask_wet_avg;
  GPtr wd_eff;                         // umask*umask_wet | vmask*vmask_wet as step3d_uv finds them (2 planes; k_wd_eff)
  GPtr tclm, Tnudgcof, uclm, vclm, M3nudgcof;   // climatology and nudging coefficients (mod_clima.F): input like the forcing; tclm, Tnudgcof per tracer
  GPtr ubarclm, vbarclm, M2nudgcof;             // ... of the 2-D momentum (LnudgeM2CLM, round 6)
  GPtr hbbl, ksbl;                              // LMD_BKPP: MIXING(ng)%hbbl; ksbl of lmd_skpp (as doubles) for lmd_bkpp.F:779
  GPtr gwrk;                           // the twenty 3-D work arrays of k_uvmix_geo.h (N+1 planes each; allocated with ROMS_MIX_GEO_UV)
  GPtr tmix;                           // harmonic tracer mixing as terms (N planes per tracer): t3dmix2 run ahead of pre_step3d stores
                                       // what it adds to t(nnew), k_pre_new adds it to the value it sets (allocated with TS_DIF2)
  GPtr lap4;                           // UV_VIS4: LapU | LapV of uv3dmix4_s.h (2 x N planes), allocated with the option bit ROMS_UV_VIS4
  // DIAGNOSTICS_TS: DIAGS(ng)%DiaTwrk, DiaTrc (i,j,k,itrc,idiag), avgzeta (allocated by roms_hip_dia_config)
  GPtr DiaTwrk, DiaTrc, dia_zeta;
  // DIAGNOSTICS_UV: ONE allocation holding DIAGS(ng)%DiaU2wrk ... DiaV3d in the order of duv_* below (option bit ROMS_DIAGNOSTICS_UV)
  GPtr duv;
};

// DIAGNOSTICS_UV: the arrays of mod_diags.F inside Fields::duv, each laid out as the reference's (dir 0 = U, 1 = V; id, lev
// 1-based).  2-D block, planes of nij: DiaU2wrk | DiaV2wrk (NDM2d each), DiaRUbar | DiaRVbar (2 x (NDM2d-1)), DiaU2int |
// DiaV2int (NDM2d), DiaRUfrc | DiaRVfrc (3 x (NDM2d-1)), DiaU2d | DiaV2d (NDM2d); then blocks of N planes: DiaU3wrk | DiaV3wrk
// (NDM3d), DiaRU | DiaRV (2 x NDrhs), DiaU3d | DiaV3d (NDM3d).
KHD size_t duv_planes2(const DGrid &G) { return (size_t)(6 * G.ndm2 + 10 * (G.ndm2 - 1)); }
KHD size_t duv_planes(const DGrid &G) { return duv_planes2(G) + (size_t)G.N * (size_t)(4 * G.ndm3 + 4 * G.ndrhs); }
template <class FT> KHD double *duv_2wrk(const DGrid &G, const FT &F, int dir, int id) { return (double *)F.duv + (size_t)(dir * G.ndm2 + id - 1) * G.nij; }
template <class FT> KHD double *duv_rbar(const DGrid &G, const FT &F, int dir, int lev, int id) {
  return (double *)F.duv + (size_t)(2 * G.ndm2 + dir * 2 * (G.ndm2 - 1) + (lev - 1) + 2 * (id - 1)) * G.nij;
}
template <class FT> KHD double *duv_2int(const DGrid &G, const FT &F, int dir, int id) {
  return (double *)F.duv + (size_t)(2 * G.ndm2 + 4 * (G.ndm2 - 1) + dir * G.ndm2 + id - 1) * G.nij;
}
template <class FT> KHD double *duv_rfrc(const DGrid &G, const FT &F, int dir, int lev, int id) {
  return (double *)F.duv + (size_t)(4 * G.ndm2 + 4 * (G.ndm2 - 1) + dir * 3 * (G.ndm2 - 1) + (lev - 1) + 3 * (id - 1)) * G.nij;
}
template <class FT> KHD double *duv_2d(const DGrid &G, const FT &F, int dir, int id) {
  return (double *)F.duv + (size_t)(4 * G.ndm2 + 10 * (G.ndm2 - 1) + dir * G.ndm2 + id - 1) * G.nij;
}
template <class FT> KHD double *duv_3wrk(const DGrid &G, const FT &F, int dir, int id) {
  return (double *)F.duv + (duv_planes2(G) + (size_t)G.N * (size_t)(dir * G.ndm3 + id - 1)) * G.nij;
}
template <class FT> KHD double *duv_r3(const DGrid &G, const FT &F, int dir, int lev, int id) {
  return (double *)F.duv + (duv_planes2(G) + (size_t)G.N * (size_t)(2 * G.ndm3 + dir * 2 * G.ndrhs + (lev - 1) + 2 * (id - 1))) * G.nij;
}
template <class FT> KHD double *duv_3d(const DGrid &G, const FT &F, int dir, int id) {
  return (double *)F.duv + (duv_planes2(G) + (size_t)G.N * (size_t)(2 * G.ndm3 + 4 * G.ndrhs + dir * G.ndm3 + id - 1)) * G.nij;
}
// DiaTwrk(:,:,:,itrc,term): level 1 of the term's block; nullptr when the diagnostics are off or the term is absent
KHD double *dia_wrk(const DGrid &G, const Fields &F, int term, int itrc) {
  if (!G.dia_ts || !G.dia_idx[term]) return nullptr;
  return (double *)F.DiaTwrk + ((size_t)(itrc - 1) + (size_t)G.NT * (size_t)(G.dia_idx[term] - 1)) * (size_t)G.N * (size_t)G.nij;
}
