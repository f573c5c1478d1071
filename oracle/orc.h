/*
 * orc.h -- CPU ORACLE for the ROMS nonlinear 3-D time step.  TEST INFRASTRUCTURE.
 *
 * Plain-C restatement of the reference algorithm (myroms/roms, ROMS/Nonlinear),
 * written loop-for-loop after the reference's serial (non-DISTRIBUTE) code so
 * that results can be compared bit-for-bit with the reference build in
 * oracle/_ref (gcc -O2 -ffp-contract=off, no -march => no FMA, like the
 * amdflang x86-64 baseline build of the reference).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library; the product (roms_amd/, libroms_hip.so) never links or calls it.
 *
 * PARITY STATUS: every function is PINNED -- checked bit for bit against the
 * reference routine itself (oracle/_ref built by oracle/ref/build_ref.sh;
 * tests/test_oracle_vs_ref.py: routine by routine on perturbed states and as
 * whole main3d passes over 100 steps) and against the fixtures the reference's
 * object code wrote (tests/golden/<case>_steps.npz, _kernels.npz, _sample.npz;
 * tests/test_golden_reference.py, which runs anywhere).
 *
 * Array layout = the reference's (mod_grid.F / mod_ocean.F): column-major,
 * i fastest, lower bounds LBi,LBj; rho-type levels 1..N, w-type levels 0..N.
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* tracer advection scheme codes (Hadvection/Vadvection, mod_param.F:324-335) */
enum { ORC_A4 = 1, ORC_C2 = 2, ORC_C4 = 3, ORC_HSIMT = 4, ORC_MPDATA = 5,
       ORC_SPLINES = 6, ORC_SPLIT_U3 = 7, ORC_U3 = 8 };

/* cpp-option bits (ROMS/Include/cppdefs.h names) */
enum {
  ORC_UV_ADV = 1 << 0, ORC_UV_COR = 1 << 1, ORC_UV_VIS2 = 1 << 2, ORC_TS_DIF2 = 1 << 3,
  ORC_MIX_GEO_TS = 1 << 4,   /* else MIX_S_TS */
  ORC_CURVGRID = 1 << 5, ORC_NONLIN_EOS = 1 << 6, ORC_UV_QDRAG = 1 << 7, /* else UV_LDRAG */
  ORC_LMD_MIXING = 1 << 8, ORC_BULK_FLUXES = 1 << 9, ORC_SOLAR_SOURCE = 1 << 10,
  ORC_ANA_VMIX = 1 << 11, ORC_SALINITY = 1 << 12, ORC_SPHERICAL = 1 << 13,
  ORC_UV_LOGDRAG = 1 << 14,  /* set_vbc.F:591-635 */
  ORC_MASKING = 1 << 15,     /* land/sea masks rmask, umask, vmask, pmask (mod_grid.F) */
  ORC_RADIATION_2D = 1 << 16, /* tangential phase speed in the radiation conditions (zetabc.F:157 ...) */
  ORC_PLAIN_VDIFF = 1 << 17,  /* SPLINES_VDIFF NOT defined: plain tridiagonal vertical diffusion for every tracer (step3d_t.F:1722-1790) */
  ORC_PLAIN_VVISC = 1 << 18,
  ORC_PRSGRD31 = 1 << 19,     /* DJ_GRADPS NOT defined: the standard density Jacobian, prsgrd31.h */
  ORC_WJ_GRADP = 1 << 27,     /* ... in its weighted form (Song 1998), prsgrd31.h:232-250 */  /* SPLINES_VVISC NOT defined: plain tridiagonal vertical viscosity (step3d_uv.F:436-500) */
  ORC_PRSGRD40 = 1 << 26,     /* PJ_GRADP: the finite-volume pressure Jacobian of Lin (1997), prsgrd40.h */
  ORC_MY25_MIXING = 1 << 28,  /* Mellor-Yamada level 2.5 closure: my25_prestep.F, my25_corstep.F (options KANTHA_CLAYSON, N2S2_HORAVG,
                                 RI_SPLINES, K_C2ADVECTION | K_C4ADVECTION in cfg.gls_flags; start values GLS_Kmin, GLS_Pmin; AKK_BAK) */
  ORC_MIX_ISO_TS = 1 << 29,   /* harmonic tracer mixing along isopycnic surfaces, t3dmix2_iso.h (else MIX_GEO_TS or MIX_S_TS) */
  ORC_APP_OVERFLOW = 1 << 30, /* the OVERFLOW application: unforced, like SEAMOUNT and GRAV_ADJ */
  ORC_GLS_MIXING = 1 << 25,   /* generic length-scale closure: gls_prestep.F, gls_corstep.F (its compile-time forms: cfg.gls_flags) */
  ORC_APP_UPWELLING = 1 << 20, ORC_APP_BENCHMARK = 1 << 21, ORC_APP_KELVIN = 1 << 22, ORC_APP_SEAMOUNT = 1 << 23, ORC_APP_GRAV_ADJ = 1 << 24   /* (no forcing: the default branches of ana_smflux.h ...) */
};

/* compile-time forms of GLS_MIXING (cppdefs.h names), cfg.gls_flags */
enum { ORC_GLS_CANUTO_A = 1, ORC_GLS_CANUTO_B = 2, ORC_GLS_KANTHA_CLAYSON = 4,   /* none of the three: Galperin */
       ORC_GLS_N2S2_HORAVG = 8, ORC_GLS_RI_SPLINES = 16,
       ORC_GLS_K_C2ADVECTION = 32, ORC_GLS_K_C4ADVECTION = 64,                   /* neither: third-order upstream */
       ORC_GLS_CHARNOK = 128, ORC_GLS_CRAIG_BANNER = 256 };

/* loop bounds of one tile: BOUNDS(ng)%xxx(tile), get_bounds.F:1044-1884 */
typedef struct {
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
  int west, east, south, north;          /* DOMAIN(ng)%Western_Edge(tile) ... */
  int sw, se, nw, ne;                    /* DOMAIN(ng)%SouthWest_Corner(tile) ... */
} orc_bounds;

#define ORC_MAXT 4
#define ORC_MAXW 512

/* lateral boundary conditions, LBC(ibry,ivar,ng) of mod_param.F / load_lbc (inp_decode.F): edge index 0..3 =
   iwest, isouth, ieast, inorth (mod_scalars.F); variable index isFsur .. isTvar(itrc) */
enum { ORC_IWEST = 0, ORC_ISOUTH = 1, ORC_IEAST = 2, ORC_INORTH = 3 };
enum { ORC_ISFSUR = 0, ORC_ISUBAR = 1, ORC_ISVBAR = 2, ORC_ISUVEL = 3, ORC_ISVVEL = 4, ORC_ISTVAR = 5 };
#define ORC_NLBC (ORC_ISTVAR + ORC_MAXT)
/* kinds (the roms.in keywords Clo Per Gra Cla Rad RadNud Che Cha Fla Shc); 0 = closed unless the direction is periodic */
enum { ORC_LBC_DEFAULT = 0, ORC_LBC_CLO = 1, ORC_LBC_PER = 2, ORC_LBC_GRA = 3, ORC_LBC_CLA = 4, ORC_LBC_RAD = 5,
       ORC_LBC_RADNUD = 6, ORC_LBC_CHE = 7, ORC_LBC_CHI = 8, ORC_LBC_FLA = 9, ORC_LBC_SHC = 10 };

typedef struct {
  /* sizes */
  int Lm, Mm, N, NT, NAT, Nghost;
  int LBi, UBi, LBj, UBj;               /* allocation bounds (global arrays) */
  int NtileI, NtileJ;
  int EWperiodic, NSperiodic;
  int options;                          /* ORC_* bits */
  int hadv[ORC_MAXT], vadv[ORC_MAXT];
  /* time stepping */
  int ntfirst, ntstart, ndtfast, nfast;
  double dt, dtfast;
  double weight[2][ORC_MAXW + 1];       /* weight(1:2,1:2*ndtfast), 1-based */
  /* physics */
  double rho0, g, lambda, gamma2, Cp;
  double R0, T0, S0, Tcoef, Scoef;
  double hc; int Vtransform;
  double rdrg, rdrg2, Zob;
  double Akt_bak[ORC_MAXT], Akv_bak;
  double dstart;
  double blk_ZQ, blk_ZT, blk_ZW;
  int lmd_Jwt;
  double cc1, cc2, cc3;                 /* HSIMT constants mod_scalars.F */
  /* open boundaries: kinds and the nudging time scales [1/s] of the radiation+nudging conditions
     (FSobc_in/out ... Tobc_in/out, inp_par.F after read_phypar: Znudg, M2nudg, M3nudg, Tnudg, obcfac) */
  int lbc[4][ORC_NLBC];
  double FSobc_in[4], FSobc_out[4], M2obc_in[4], M2obc_out[4], M3obc_in[4], M3obc_out[4];
  double Tobc_in[ORC_MAXT][4], Tobc_out[ORC_MAXT][4];
  /* GLS_MIXING: the GLS_* block of roms.in (read_phypar.F), Akk_bak, Akp_bak, Zos, the surface-flux constants */
  int gls_flags;
  double gls_p, gls_m, gls_n, gls_Kmin, gls_Pmin, gls_cmu0, gls_c1, gls_c2, gls_c3m, gls_c3p, gls_sigk, gls_sigp;
  double Akk_bak, Akp_bak, Zos, charnok_alpha, crgban_cw;
  double obcfac;                        /* OBCFAC: with climatology nudging the radiation conditions read the coefficient arrays, obc_in = obcfac * obc_out */
  int lbc_tke[4];                       /* LBC(isMtke) [iwest, isouth, ieast, inorth]: 0 = closed / periodic as the direction is; ORC_LBC_GRA, ORC_LBC_RAD (tkebc_im.F) */
  int volcons;                          /* VolCons(iwest..inorth) of roms.in: bit e = edge e (obc_volcons.F) */
} orc_cfg;

/* time-level state of main3d / mod_stepping */
typedef struct {
  int iic, iif;
  int nstp, nnew, nrhs;
  int kstp, knew, krhs, indx1;
  int predictor;                        /* PREDICTOR_2D_STEP */
  double time, tdays;
} orc_step;

typedef struct orc_s {
  orc_cfg c;
  orc_step s;
  orc_bounds *b;                        /* NtileI*NtileJ tiles */
  int ntiles;
  int nthreads;                          /* > 1: the tile loops of orc_main3d_step run as OpenMP threads */
  size_t ni, nj, nij;
  /* s-coordinate */
  double *sc_r, *Cs_r, *sc_w, *Cs_w;    /* sc_r[k-1], sc_w[k] */
  /* mod_grid 2-D */
  double *h, *f, *fomn, *pm, *pn, *om_r, *on_r, *om_u, *on_u, *om_v, *on_v, *om_p, *on_p,
      *omn, *pmon_r, *pnom_r, *pmon_p, *pnom_p, *pmon_u, *pnom_u, *pmon_v, *pnom_v,
      *dmde, *dndx, *angler, *xr, *yr, *xp, *yp, *lonr, *latr, *rdrag, *rdrag2,
      *rmask, *umask, *vmask, *pmask;      /* MASKING: 1 water, 0 land (pmask: 2 no-slip); all 1 otherwise */
  /* mod_grid 3-D */
  double *Hz, *z_r, *z_w, *Huon, *Hvom;
  /* mod_ocean */
  double *zeta, *ubar, *vbar, *rzeta, *rubar, *rvbar;
  double *u, *v, *t, *W, *wvel, *rho, *pden, *ru, *rv;
  /* mod_coupling */
  double *rhoA, *rhoS, *rufrc, *rvfrc, *Zt_avg1, *DU_avg1, *DU_avg2, *DV_avg1, *DV_avg2;
  /* mod_forces */
  double *sustr, *svstr, *bustr, *bvstr, *stflx, *btflx, *stflux, *btflux, *srflx;
  double *Uwind, *Vwind, *Tair, *Pair, *Hair, *rain, *cloud, *lhflx, *shflx, *lrflx, *evap;
  /* mod_mixing */
  double *Akv, *Akt, *visc2_r, *visc2_p, *diff2, *bvf, *alpha, *beta, *hsbl, *ghats;
  double *alfaobeta;                     /* LMD_DDMIX: ratio of the thermal expansion and saline contraction coefficients (i,j,0:N), rho_eos.F:454, :794 */
  int ddmix;                             /* LMD_DDMIX on (orc_set_ddmix): lmd_vmix.F:360-428 */
  double *hbbl; int *kbbl; int bkpp;     /* LMD_BKPP on (orc_set_bkpp): depth and level index of the bottom boundary layer, lmd_bkpp.F */
  /* WET_DRY (wetdry.F): time-dependent masks; rmask_wet_avg: sum of the rho mask over the fast steps; *_full: wet mask x land mask */
  double *rmask_wet, *umask_wet, *vmask_wet, *pmask_wet, *rmask_full, *umask_full, *vmask_full, *pmask_full, *rmask_wet_avg;
  /* obc_volcons.F / mod_scalars.F:1460-1462: cross-section and flux of the open edges summed over the tiles in calling order,
     the correction velocity; vc_count = tile_count of mod_parallel.F:89 */
  double bc_area, bc_flux, ubar_xs; int vc_count;
  int wet_dry; double Dcrit;             /* switched on by orc_set_wetdry (DCRIT of roms.in, read_phypar.F:1021) */
  double *visc4_r, *visc4_p, *diff4;     /* UV_VIS4 / TS_DIF4: square roots of the biharmonic coefficients (inp_par.F:634) */
  /* climatology nudging (mod_clima.F): tclm, Tnudgcof (i,j,k,itrc) -- per tracer, not the reference's compact index --, uclm, vclm,
     M3nudgcof (i,j,k); clima_flags: bit 0 LnudgeM3CLM, bit itrc LtracerCLM & LnudgeTCLM of tracer itrc (orc_set_clima) */
  double *tclm, *Tnudgcof, *uclm, *vclm, *M3nudgcof;
  /* ... bit 5 of clima_flags: LnudgeM2CLM (step2d_LF_AM3.h:2179-2203), towards ubarclm, vbarclm with M2nudgcof (i,j) */
  double *ubarclm, *vbarclm, *M2nudgcof;
  int clima_flags;
  int prs_scheme;                        /* 42: PJ_GRADPQ2 (prsgrd42.h), 44: PJ_GRADPQ4 (prsgrd44.h); 0: by the ORC_PRSGRD* bits (orc_set_prsgrd) */
  int uv_vis4, ts_dif4;                  /* biharmonic mixing along s-surfaces switched on (orc_set_mix4; orc_mix4.c) */
  int mix_geo_uv;                        /* UV_VIS2 along geopotential surfaces (MIX_GEO_UV; orc_set_geouv, orc_uvmix_geo.c) */
  double *tke, *gls, *Lscale, *Akk, *Akp;   /* GLS_MIXING: tke, gls(i,j,0:N,3); Lscale, Akk, Akp(i,j,0:N) */
  int *ksbl;
  /* mod_boundary: BOUNDARY(ng)%zeta_west(LBj:UBj) ... t_north(LBi:UBi,N,NT): the open-boundary data of this step (inputs) */
  double *zeta_west, *zeta_south, *zeta_east, *zeta_north, *ubar_west, *ubar_south, *ubar_east, *ubar_north,
      *vbar_west, *vbar_south, *vbar_east, *vbar_north, *u_west, *u_south, *u_east, *u_north,
      *v_west, *v_south, *v_east, *v_north, *t_west, *t_south, *t_east, *t_north;
  /* diag results: avgke, avgpe, avgkp, volume, max_speed, Cu_max ... */
  double diag[16];
  void *avg;                             /* time-averaged fields (orc_avg.c), NULL until orc_set_avg_window */
  void *dia;                             /* per-term tracer tendencies, DIAGNOSTICS_TS (orc_diags.c), NULL until orc_set_dia_window */
  struct orc_diauv *duv;                 /* per-term momentum tendencies, DIAGNOSTICS_UV (orc_diags_uv.c), NULL until orc_set_diauv */
} orc_t;

/* ---- index helpers (valid inside functions that define LBi,LBj,ni,nij,N) ---- */
#define X2(i, j) ((size_t)((i) - LBi) + (size_t)((j) - LBj) * ni)
#define X3(i, j, k) (X2(i, j) + (size_t)((k) - 1) * nij)                 /* rho levels 1..N */
#define XW(i, j, k) (X2(i, j) + (size_t)(k) * nij)                       /* w levels 0..N   */
#define X4(i, j, k, n) (X3(i, j, k) + (size_t)((n) - 1) * nij * N)       /* u,v(i,j,k,n)    */
#define XW4(i, j, k, n) (XW(i, j, k) + (size_t)((n) - 1) * nij * (N + 1))/* ru,rv,Akt(i,j,k,n) */
#define XT(i, j, k, n, it) (X3(i, j, k) + ((size_t)((n) - 1) + 3 * (size_t)((it) - 1)) * nij * N)
#define X2T(i, j, n) (X2(i, j) + (size_t)((n) - 1) * nij)                /* zeta(i,j,n) ... */

void orc_check_step(const orc_t *o, const char *who);          /* aborts on a stepping index outside its array extent */
#define ORC_LOCALS(o)                                                        \
  const int LBi = (o)->c.LBi, LBj = (o)->c.LBj, N = (o)->c.N;                \
  const size_t ni = (o)->ni, nij = (o)->nij;                                 \
  (void)LBi; (void)LBj; (void)N; (void)ni; (void)nij;                        \
  orc_check_step((o), __func__)

/* WET_DRY: the factor the barotropic step applies to a momentum point (step2d_LF_AM3.h:2208-2210 ...): the mask itself
   where it is 0 or +-2... in the reference's words "cff7": 0.5*mask*cff5 + cff6*(1-cff5) */
static inline double orc_wd_fac(double mw, double val) {
  const double cff5 = __builtin_fabs(__builtin_fabs(mw) - 1.0);
  const double cff6 = 0.5 + __builtin_copysign(0.5, val) * mw;
  return 0.5 * mw * cff5 + cff6 * (1.0 - cff5);
}

/* ---- API ---- */
orc_t *orc_create(const orc_cfg *cfg);
void orc_destroy(orc_t *o);
double *orc_field(orc_t *o, const char *name, long *nel);   /* pointer into the state */
orc_step *orc_stepping(orc_t *o);
orc_cfg *orc_config(orc_t *o);
void orc_get_bounds(orc_t *o, int tile, int *out);           /* same order as ref_get_bounds */

/* tiling: get_bounds.F */
void orc_tile_bounds(const orc_cfg *c, int tile, orc_bounds *b);

/* periodic copies: exchange_2d.F / exchange_3d.F ('r','u','v','p' grids) */
void orc_exchange2d(const orc_t *o, const orc_bounds *b, char grid, double *A);
void orc_exchange3d(const orc_t *o, const orc_bounds *b, char grid, double *A, int nk);
/* gradient / closed fills: bc_2d.F, bc_3d.F */
void orc_bc_r2d(const orc_t *o, const orc_bounds *b, double *A);
void orc_bc_u2d(const orc_t *o, const orc_bounds *b, double *A);
void orc_bc_v2d(const orc_t *o, const orc_bounds *b, double *A);
void orc_bc_u3d(const orc_t *o, const orc_bounds *b, double *A, int nk);
void orc_bc_v3d(const orc_t *o, const orc_bounds *b, double *A, int nk);
void orc_bc_w3d(const orc_t *o, const orc_bounds *b, double *A, int nk);
/* state BCs: zetabc.F u2dbc_im.F v2dbc_im.F t3dbc_im.F u3dbc_im.F v3dbc_im.F */
void orc_zetabc(const orc_t *o, const orc_bounds *b, int kout);
void orc_tkebc(const orc_t *o, const orc_bounds *b, int nout);   /* tkebc_im.F */
void orc_u2dbc(const orc_t *o, const orc_bounds *b, int kout);
void orc_v2dbc(const orc_t *o, const orc_bounds *b, int kout);
void orc_t3dbc(const orc_t *o, const orc_bounds *b, int nout, int itrc);
void orc_u3dbc(const orc_t *o, const orc_bounds *b, int nout);
void orc_v3dbc(const orc_t *o, const orc_bounds *b, int nout);
void orc_bc2d(orc_t *o, int tile, int kout);   /* zetabc, u2dbc, v2dbc of one tile (tests) */
void orc_bc3d(orc_t *o, int tile, int nout);   /* t3dbc of every tracer; u3dbc, v3dbc when nout <= 2 (tests) */

int orc_lbc(const orc_t *o, int edge, int var);              /* kind with the default resolved (periodic / closed) */
int orc_lbc_acquire(const orc_t *o, int edge, int var);     /* LBC(edge,var)%acquire as load_lbc sets it (inp_decode.F:1616-1660) */
int orc_lbc_open(const orc_t *o);                             /* any edge of any variable other than closed / periodic */

/* kernels (tile = 0..ntiles-1) */
void orc_set_wetdry(orc_t *o, double Dcrit);                 /* WET_DRY on (orc_wetdry.c) */
void orc_wetdry_ini(orc_t *o, int tile);                     /* wetdry_ini_tile, wetdry.F:355-490 (initial.F:467) */
void orc_wetdry_tile(orc_t *o, int tile);                    /* wetdry_tile, wetdry.F:93-351 (called by step2d) */
void orc_set_depth(orc_t *o, int tile);
void orc_set_massflux(orc_t *o, int tile);
void orc_rho_eos(orc_t *o, int tile);
void orc_set_vbc(orc_t *o, int tile);
void orc_ana_vmix(orc_t *o, int tile);
void orc_set_data(orc_t *o, int tile);       /* analytic forcing of this step */
void orc_omega(orc_t *o, int tile);
void orc_wvelocity(orc_t *o, int tile, int ninp);
void orc_set_zeta(orc_t *o, int tile);
void orc_ini_zeta(orc_t *o, int tile);
void orc_ini_fields(orc_t *o, int tile);
void orc_pre_step3d(orc_t *o, int tile);
void orc_prsgrd(orc_t *o, int tile);
void orc_t3dmix2(orc_t *o, int tile);
void orc_uv3dmix2(orc_t *o, int tile);
void orc_rhs3d_tile(orc_t *o, int tile);
void orc_rhs3d(orc_t *o, int tile);          /* pre_step3d, prsgrd, t3dmix2, rhs3d_tile, uv3dmix2 */
void orc_step2d(orc_t *o, int tile);
void orc_step3d_uv(orc_t *o, int tile);
void orc_step3d_t(orc_t *o, int tile);
void orc_diag(orc_t *o);
void orc_lmd_vmix(orc_t *o, int tile);
void orc_bulk_flux(orc_t *o, int tile);
void orc_gls_prestep(orc_t *o, int tile);
void orc_gls_corstep(orc_t *o, int tile);
void orc_my25_prestep(orc_t *o, int tile);
void orc_my25_corstep(orc_t *o, int tile);
void orc_mpdata_adiff(orc_t *o, int tile, int itrc, const double *Ta, double *Ua, double *Va,
                      double *Wa, const double *oHz);

/* time averages: set_avg.F (orc_avg.c); fields "avg_zeta" ... "avg_HvomT" through orc_avg_field */
void orc_set_avg_window(orc_t *o, int nAVG, int ntsAVG, int nrrec, int ntstart);
void orc_set_avg(orc_t *o, int tile);
double *orc_avg_field(orc_t *o, const char *name, long *nel);
double orc_avg_time(const orc_t *o);
void orc_avg_free(orc_t *o);

/* per-term tracer tendencies (DIAGNOSTICS_TS): mod_diags.F, set_diags.F (orc_diags.c); fields "DiaTwrk", "DiaTrc", "dia_zeta" */
enum { ORC_DIA_HADV = 0, ORC_DIA_XADV, ORC_DIA_YADV, ORC_DIA_VADV, ORC_DIA_HDIF, ORC_DIA_XDIF, ORC_DIA_YDIF, ORC_DIA_SDIF,
       ORC_DIA_VDIF, ORC_DIA_RATE, ORC_DIA_NTERMS };
int orc_set_dia_window(orc_t *o, int nDIA, int ntsDIA, int nrrec, int ntstart);   /* 0, or 5: MPDATA tracers are not covered */
void orc_set_diags(orc_t *o, int tile);
double *orc_dia_wrk(orc_t *o, int term, int itrc);       /* DiaTwrk(:,:,:,itrc,term), NULL when off / absent */
double *orc_dia_field(orc_t *o, const char *name, long *nel);
int orc_dia_ndt(const orc_t *o);
double orc_dia_time(const orc_t *o);
void orc_dia_free(orc_t *o);
/* per-term momentum tendencies (DIAGNOSTICS_UV): mod_diags.F:174-222, the term indices of mod_scalars.F:4264-4377 for the
   option set (1-based, 0 = the option set has no such term), arrays laid out as the reference's */
typedef struct orc_diauv {
  int NDM2d, NDM3d, NDrhs;
  int M2fcor, M2hadv, M2xadv, M2yadv, M2hvis, M2xvis, M2yvis, M2pgrd, M2sstr, M2bstr, M2rate;
  int M3fcor, M3vadv, M3hadv, M3xadv, M3yadv, M3pgrd, M3vvis, M3hvis, M3xvis, M3yvis, M3rate;
  double *U2wrk, *V2wrk;       /* (i,j,NDM2d) */
  double *RUbar, *RVbar;       /* (i,j,2,NDM2d-1) */
  double *U2int, *V2int;       /* (i,j,NDM2d) */
  double *RUfrc, *RVfrc;       /* (i,j,3,NDM2d-1) */
  double *U3wrk, *V3wrk;       /* (i,j,N,NDM3d) */
  double *RU, *RV;             /* (i,j,N,2,NDrhs) */
  double *U2d, *V2d, *U3d, *V3d;   /* the accumulated output of set_diags */
} orc_diauv;
/* biharmonic horizontal mixing along s-surfaces (UV_VIS4 + MIX_S_UV, TS_DIF4 + MIX_S_TS): orc_mix4.c */
void orc_set_mix4(orc_t *o, int uv_vis4, int ts_dif4);
void orc_prsgrd42(orc_t *o, int tile);                        /* orc_prs4x.c */
void orc_prsgrd44(orc_t *o, int tile);
void orc_set_bkpp(orc_t *o, int on);                          /* LMD_BKPP: the bottom boundary layer behind lmd_skpp (lmd_bkpp.F) */
void orc_set_ddmix(orc_t *o, int on);                         /* LMD_DDMIX: double-diffusive mixing in lmd_vmix's interior scheme */
void orc_set_prsgrd(orc_t *o, int scheme);                    /* prsgrd.F:16-19: PJ_GRADPQ4 -> prsgrd44.h, PJ_GRADPQ2 -> prsgrd42.h */
void orc_set_clima(orc_t *o, int flags);                       /* climatology nudging: step3d_t.F:1866-1878, rhs3d.F:654-680 */
void orc_set_geouv(orc_t *o, int on);                          /* MIX_GEO_UV: uv3dmix2_geo.h in place of uv3dmix2_s.h */
void orc_uv3dmix2_geo(orc_t *o, int tile);
void orc_t3dmix4(orc_t *o, int tile);
void orc_uv3dmix4(orc_t *o, int tile);
void orc_uv3dmix4_geo(orc_t *o, int tile);                    /* orc_uvmix_geo.c: UV_VIS4 + MIX_GEO_UV */
void orc_lap_bc(const orc_t *o, const orc_bounds *b, double *LapU, double *LapV, int isu, int isv, int iu0, int iu1, int ju0,
                int ju1, int iv0, int iv1, int jv0, int jv1);   /* conditions on the first harmonic operator of momentum (orc_mix4.c) */
void orc_step2d_vis4(orc_t *o, const orc_bounds *b, int krhs, const double *Drhs, double *rhs_ubar, double *rhs_vbar, double *U2rhs, double *V2rhs);
int orc_set_diauv(orc_t *o);                 /* allocate (the window is the one of orc_set_dia_window) */
void orc_diauv_free(orc_t *o);
void orc_set_diags_uv(orc_t *o, int tile, int init, int accum, int convert, double fac);
double *orc_diauv_field(orc_t *o, const char *name, long *nel);
/* element (i,j[,k]) of term `id` (1-based) of the arrays above; ORC_LOCALS in scope */
#define DU2(a, i, j, id) (a)[X2(i, j) + (size_t)((id) - 1) * nij]
#define DUB(a, i, j, lev, id) (a)[X2(i, j) + (size_t)((lev) - 1 + 2 * ((id) - 1)) * nij]
#define DUF(a, i, j, lev, id) (a)[X2(i, j) + (size_t)((lev) - 1 + 3 * ((id) - 1)) * nij]
#define DU3(a, i, j, k, id) (a)[X3(i, j, k) + (size_t)((id) - 1) * (size_t)N * nij]
#define DUR(a, i, j, k, lev, id) (a)[X3(i, j, k) + (size_t)((lev) - 1 + 2 * ((id) - 1)) * (size_t)N * nij]

/* one baroclinic step, main3d.F:216-1148 */
int orc_main3d_step(orc_t *o);
/* start-of-run: initial.F tail (iic=ntstart) */
void orc_start(orc_t *o);

#ifdef __cplusplus
}
#endif
#endif
