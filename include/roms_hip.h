/*
 * roms_hip.h -- C ABI of libroms_hip.so: the MI355X (gfx950) implementation of the
 * ROMS nonlinear 3-D time step (main3d -> step2d / rhs3d / step3d_uv / step3d_t ...).
 *
 * This is the drop-in boundary.  The reference has no FFI: the replaceable seam is
 * the set of module procedures that main3d USEs (ROMS/Nonlinear/main3d.F:108-157),
 * each of the form  kernel(ng, tile)  that reads loop bounds from BOUNDS(ng), time
 * indices from mod_stepping and the state from the mod_grid/mod_ocean/mod_coupling/
 * mod_forces/mod_mixing arrays.  Every entry point below names the reference
 * procedure it replaces.  A Fortran caller binds them with ISO_C_BINDING
 * (roms_amd/host/roms_hip_mod.f90; INTEGRATION.md shows the stub to add to the
 * reference's main3d.F).
 *
 * Conventions
 *   - plain C types only; all arrays are double precision (real(r8));
 *   - arrays cross the boundary in the reference's own layout: column-major,
 *     i fastest, bounds LBi:UBi x LBj:UBj, rho-levels 1:N, w-levels 0:N,
 *     time levels as in mod_ocean.F:386-454 (t(i,j,k,3,NT), u(i,j,k,2), zeta(i,j,3),
 *     ru(i,j,0:N,2) ...).  Sizes are implied by the dimensions given at create;
 *   - every function returns 0 on success; a non-zero value maps onto the
 *     reference's exit_flag (mod_scalars.F:548-560): 1 blow-up, 2 device/comm
 *     error, 5 bad configuration, 8 algorithm/usage error;
 *   - a context is bound to one HIP device and one stream; not re-entrant.
 */
#ifndef ROMS_HIP_H
#define ROMS_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ROMS_HIP_ABI_VERSION 5    /* 5 (round 6): obcfac appended -- with climatology nudging the radiation conditions take their time scales
                                        from the nudging coefficient arrays, obc_in = obcfac * obc_out (u3dbc_im.F:113-118);
                                     2: lateral boundary conditions (lbc ... Tobc_out) appended to roms_hip_config;
                                     3: the generic length-scale closure (gls_flags ... lbc_tke) appended;
                                     4: options is a 64-bit mask -- UV_VIS4, TS_DIF4, WET_DRY (+ Dcrit, appended),
                                        DIAGNOSTICS_UV are option bits like the others, not configuration calls */
#define ROMS_MAXT 4              /* max tracers handled (NT) */
#define ROMS_MAXW 512            /* max 2*ndtfast */

/* tracer advection schemes: Hadvection/Vadvection(itrc,ng)%..., mod_param.F:324-335 */
enum { ROMS_A4 = 1, ROMS_C2 = 2, ROMS_C4 = 3, ROMS_HSIMT = 4, ROMS_MPDATA = 5,
       ROMS_SPLINES = 6, ROMS_SPLIT_U3 = 7, ROMS_U3 = 8 };

/* cpp options of the application header (ROMS/Include/cppdefs.h names) */
enum {
  ROMS_UV_ADV = 1 << 0, ROMS_UV_COR = 1 << 1, ROMS_UV_VIS2 = 1 << 2, ROMS_TS_DIF2 = 1 << 3,
  ROMS_MIX_GEO_TS = 1 << 4, ROMS_CURVGRID = 1 << 5, ROMS_NONLIN_EOS = 1 << 6,
  ROMS_UV_QDRAG = 1 << 7, ROMS_LMD_MIXING = 1 << 8, ROMS_BULK_FLUXES = 1 << 9,
  ROMS_SOLAR_SOURCE = 1 << 10, ROMS_ANA_VMIX = 1 << 11, ROMS_SALINITY = 1 << 12,
  ROMS_SPHERICAL = 1 << 13,
  ROMS_UV_LOGDRAG = 1 << 14,        /* logarithmic bottom drag from Zob (set_vbc.F:591-635); else UV_QDRAG / UV_LDRAG */
  ROMS_MASKING = 1 << 15,           /* land/sea masks: arrays "rmask", "umask", "vmask", "pmask" (mod_grid.F), all water until
                                       uploaded; every physics option of the library carries its masked branches (MPDATA's
                                       mpdata_adiff.F blocks included) */
  ROMS_RADIATION_2D = 1 << 16,      /* tangential phase speed in the radiation conditions (zetabc.F:157, u2dbc_im.F:188 ...) */
  ROMS_PLAIN_VDIFF = 1 << 17,       /* SPLINES_VDIFF is NOT defined: plain tridiagonal vertical diffusion of every tracer (step3d_t.F:1722-1790) */
  ROMS_PLAIN_VVISC = 1 << 18,       /* SPLINES_VVISC is NOT defined: plain tridiagonal vertical viscosity (step3d_uv.F:436-500, :903-967) */
  ROMS_PRSGRD31 = 1 << 19,          /* DJ_GRADPS is NOT defined: the standard density Jacobian prsgrd31.h (prsgrd.F:22-26) */
  ROMS_WJ_GRADP = 1 << 27,          /* ... in its weighted form, prsgrd31.h:232-250 */
  ROMS_PRSGRD40 = 1 << 26,          /* PJ_GRADP: the finite-volume pressure Jacobian of Lin (1997), prsgrd40.h */
  ROMS_MY25_MIXING = 1 << 28,       /* Mellor-Yamada level 2.5 closure (my25_prestep.F, my25_corstep.F): the entries roms_hip_gls_prestep /
                                       _corstep run it; KANTHA_CLAYSON, N2S2_HORAVG, RI_SPLINES, K_C2/K_C4ADVECTION in gls_flags; GLS_Kmin, GLS_Pmin
                                       (start values) and AKK_BAK from the same roms.in block */
  ROMS_MIX_ISO_TS = 1 << 29,        /* harmonic tracer mixing along isopycnic surfaces, t3dmix2_iso.h (its default slope treatment) */
  ROMS_APP_OVERFLOW = 1 << 30,      /* the OVERFLOW application: unforced, like SEAMOUNT and GRAV_ADJ */
  ROMS_GLS_MIXING = 1 << 25,        /* generic length-scale vertical closure (gls_prestep.F, gls_corstep.F); its compile-time
                                       forms in roms_hip_config.gls_flags, its roms.in parameters beside them */
  ROMS_APP_UPWELLING = 1 << 20, ROMS_APP_BENCHMARK = 1 << 21,
  ROMS_APP_KELVIN = 1 << 22,        /* no wind, no surface fluxes (the default branches of ana_smflux.h, ana_stflux.h) */
  ROMS_APP_SEAMOUNT = 1 << 23, ROMS_APP_GRAV_ADJ = 1 << 24   /* likewise unforced (set_data has nothing to do) */
};
/* ... bits 32 and up of roms_hip_config.options (ABI version 4; rounds 1-4 switched these on through configuration calls) */
#define ROMS_UV_VIS4 (1ull << 32)         /* biharmonic viscosity along s-surfaces: uv3dmix4_s.h:119-627, step2d_LF_AM3.h:1653-1920 (MIX_S_UV) */
#define ROMS_TS_DIF4 (1ull << 33)         /* biharmonic tracer diffusion: t3dmix4_s.h:94-478 (MIX_S_TS); t3dmix4_geo.h:98-780 with ROMS_MIX_GEO_TS,
                                             t3dmix4_iso.h:98-812 with ROMS_MIX_ISO_TS (round 5: domains periodic in xi -- the wall conditions of :475-600 at iwest / ieast are refused, exit_flag 5) */
#define ROMS_WET_DRY (1ull << 34)         /* wetting and drying, wetdry.F and its branches (below); roms_hip_config.Dcrit = DCRIT of roms.in */
#define ROMS_MIX_GEO_UV (1ull << 36)      /* UV_VIS2 along geopotential surfaces: uv3dmix2_geo.h:130-757 (the rotated stress tensor) in place of
                                             uv3dmix2_s.h; with ROMS_UV_VIS4 (round 6): uv3dmix4_geo.h:296-1478, the operator twice, in place of
                                             uv3dmix4_s.h.  Refused (exit_flag 5) with DIAGNOSTICS_UV (and, in its biharmonic form, with WET_DRY) */
#define ROMS_NUDGE_M3CLM (1ull << 37)     /* LnudgeM3CLM of roms.in: nudging of u, v towards "uclm", "vclm" with "M3nudgcof" (rhs3d.F:654-680) */
#define ROMS_NUDGE_TCLM(itrc) (1ull << (37 + (itrc)))   /* LtracerCLM & LnudgeTCLM of tracer itrc = 1..4: towards "tclm" with "Tnudgcof"
                                             (step3d_t.F:1866-1878; N planes per tracer, tracer-major).  The arrays are inputs like the
                                             forcing: upload them before roms_hip_start and whenever set_data.F would refresh them.
                                             Refused (exit_flag 5) with open boundaries and DIAGNOSTICS_UV */
#define ROMS_NUDGE_TCLM_ALL (15ull << 38)
#define ROMS_NUDGE_M2CLM (1ull << 42)     /* LnudgeM2CLM of roms.in (round 6): nudging of ubar, vbar towards "ubarclm", "vbarclm" with "M2nudgcof"
                                             in every step2d call (step2d_LF_AM3.h:2179-2203); the per-call kernel carries it (no pair / loop launches) */
#define ROMS_PRSGRD42 (1ull << 43)        /* PJ_GRADPQ2 (round 6): the finite-volume pressure Jacobian with parabolic WENO reconstruction of density,
                                             prsgrd42.h:227-482.  A single tile only (its second pass reads rv(Iend+1,j), which no tile computes:
                                             what a partition gives depends on the partition, in the reference too) */
#define ROMS_PRSGRD44 (1ull << 44)        /* PJ_GRADPQ4 (round 6): ... with quartic reconstruction and power-law reconciliation, prsgrd44.h:224-508 */
#define ROMS_LMD_DDMIX (1ull << 45)       /* LMD_DDMIX (round 6, with ROMS_LMD_MIXING): double-diffusive mixing -- salt fingering, diffusive convection -- added to
                                             Akt in the interior scheme, lmd_vmix.F:360-428, from alfaobeta of rho_eos.F:435-455 | :782-796 */
#define ROMS_LMD_BKPP (1ull << 46)        /* LMD_BKPP (round 6, with ROMS_LMD_MIXING): the bottom boundary layer of the K-profile scheme behind lmd_skpp,
                                             lmd_bkpp.F:95-806 (RI_SPLINES, the file's own SASHA); MIXING(ng)%hbbl is the field "hbbl" */
#define ROMS_DIAGNOSTICS_UV (1ull << 35)  /* roms_hip_dia_config allocates and switches on the momentum terms too (mod_diags.F:174-222) */

/* GLS_MIXING: the cpp options that select a form of gls_prestep.F / gls_corstep.F (cppdefs.h names).  Stability
   functions: CANUTO_A | CANUTO_B | KANTHA_CLAYSON, none = Galperin (gls_corstep.F:1120-1165, mod_scalars.F:1764-1796,
   :4715-4766); advection of tke and gls: K_C2ADVECTION | K_C4ADVECTION, none = third-order upstream.  ZOS_HSIG and
   TKE_WAVEDISS (wave fields), LIMIT_VDIFF / LIMIT_VVISC are not built (the host stops on them). */
enum { ROMS_GLS_CANUTO_A = 1, ROMS_GLS_CANUTO_B = 2, ROMS_GLS_KANTHA_CLAYSON = 4, ROMS_GLS_N2S2_HORAVG = 8,
       ROMS_GLS_RI_SPLINES = 16, ROMS_GLS_K_C2ADVECTION = 32, ROMS_GLS_K_C4ADVECTION = 64, ROMS_GLS_CHARNOK = 128,
       ROMS_GLS_CRAIG_BANNER = 256 };

/* Lateral boundary conditions: LBC(ibry,ivar,ng) of mod_param.F, the LBC(isFsur) ... LBC(isTvar) lines of roms.in
   (load_lbc, Utility/inp_decode.F:1560-1680).  Edge index = the reference's iwest, isouth, ieast, inorth minus one;
   variable index isFsur ... isTvar(itrc).  Kind 0 = closed, or periodic where the direction is (what every caller of
   ABI version 1 had).  Built: zetabc.F, u2dbc_im.F, v2dbc_im.F, u3dbc_im.F, v3dbc_im.F, t3dbc_im.F with the kinds below;
   `Red`, `Nes`, `Mix` and nudging with climatology coefficients are not (roms_hip_create: exit_flag 5). */
enum { ROMS_IWEST = 0, ROMS_ISOUTH = 1, ROMS_IEAST = 2, ROMS_INORTH = 3 };
enum { ROMS_ISFSUR = 0, ROMS_ISUBAR = 1, ROMS_ISVBAR = 2, ROMS_ISUVEL = 3, ROMS_ISVVEL = 4, ROMS_ISTVAR = 5 };
#define ROMS_NLBC (ROMS_ISTVAR + ROMS_MAXT)
enum { ROMS_LBC_DEFAULT = 0, ROMS_LBC_CLO = 1, ROMS_LBC_PER = 2, ROMS_LBC_GRA = 3, ROMS_LBC_CLA = 4, ROMS_LBC_RAD = 5,
       ROMS_LBC_RADNUD = 6, ROMS_LBC_CHE = 7 /* Chapman explicit */, ROMS_LBC_CHI = 8 /* Chapman implicit, `Cha` */,
       ROMS_LBC_FLA = 9, ROMS_LBC_SHC = 10 };

/* Everything the kernels read from mod_param / mod_scalars / BOUNDS(ng) / DOMAIN(ng)
   for ONE tile (= one GPU).  Filled by the host (inp_par + get_bounds restatement). */
typedef struct roms_hip_config {
  int abi_version;               /* ROMS_HIP_ABI_VERSION */
  int device;                    /* HIP device ordinal */
  /* mod_param.F */
  int Lm, Mm, N, NT, NAT, Nghost;
  int LBi, UBi, LBj, UBj;        /* BOUNDS(ng)%LBi(tile) ... allocation bounds of this tile */
  int NtileI, NtileJ, tile;      /* tile = MyRank */
  int EWperiodic, NSperiodic;
  unsigned long long options;    /* ROMS_* option bits (64: ABI version 4) */
  int hadv[ROMS_MAXT], vadv[ROMS_MAXT];
  /* BOUNDS(ng)%xxx(tile): Istr Iend Jstr Jend; the derived ranges (IstrU, Istrm1 ...)
     are recomputed on the device by the rules of get_bounds.F:1044-1884 */
  int Istr, Iend, Jstr, Jend;
  /* DOMAIN(ng)%Western_Edge(tile) ... */
  int west_edge, east_edge, south_edge, north_edge;
  /* mod_scalars.F */
  int ntfirst, ntstart, ndtfast, nfast;
  int ninfo;                     /* diag every ninfo steps (NINFO); 0 = only on request */
  double dt, dtfast;
  double weight[2][ROMS_MAXW + 1];   /* weight(1:2,1:2*ndtfast,ng), index 0 unused */
  double rho0, g, lambda, gamma2, Cp;
  double R0, T0, S0, Tcoef, Scoef;
  double hc; int Vtransform;
  double rdrg, rdrg2, Zob;
  double Akt_bak[ROMS_MAXT], Akv_bak;
  double dstart;
  double blk_ZQ, blk_ZT, blk_ZW;
  int lmd_Jwt;
  double sc_r[256], Cs_r[256], sc_w[257], Cs_w[257];   /* SCALARS(ng)%sc_r(1:N) -> [k-1]; sc_w(0:N) -> [k] */
  /* open boundaries (ABI version 2).  lbc[edge][variable] = ROMS_LBC_*; the nudging time scales [1/s] of the
     radiation + nudging conditions as inp_par.F:726-752 derives them from ZNUDG, M2NUDG, M3NUDG, TNUDG and OBCFAC.
     The boundary data BOUNDARY(ng)%zeta_west(LBj:UBj) ... t_north(LBi:UBi,N,NT) of mod_boundary.F are uploaded under
     those names ("zeta_west", "ubar_east", "u_south", "t_north" ...) whenever the caller's set_data has new values. */
  int lbc[4][ROMS_NLBC];
  double FSobc_in[4], FSobc_out[4], M2obc_in[4], M2obc_out[4], M3obc_in[4], M3obc_out[4];
  double Tobc_in[ROMS_MAXT][4], Tobc_out[ROMS_MAXT][4];
  /* GLS_MIXING (ABI version 3): ROMS_GLS_* flags; GLS_P ... GLS_SIGP, AKK_BAK, AKP_BAK, Zos, CHARNOK_ALPHA, CRGBAN_CW of
     roms.in (read_phypar.F); LBC(isMtke) per edge: closed, gradient, radiation or periodic (tkebc_im.F:46-700; radiation with
     GLS_MIXING, not with MY25_MIXING).  State arrays "tke", "gls" (i,j,0:N,3), "Lscale", "Akk", "Akp" (i,j,0:N) of mod_mixing.F. */
  int gls_flags;
  double gls_p, gls_m, gls_n, gls_Kmin, gls_Pmin, gls_cmu0, gls_c1, gls_c2, gls_c3m, gls_c3p, gls_sigk, gls_sigp;
  double Akk_bak, Akp_bak, Zos, charnok_alpha, crgban_cw;
  int lbc_tke[4];
  /* WET_DRY (ABI version 4): DCRIT of roms.in (read_phypar.F:1021), the total depth below which a cell is dry */
  double Dcrit;
  /* ABI version 5: OBCFAC of roms.in (mod_scalars.F: obcfac) -- the ratio inflow / outflow nudging time scale of the radiation +
     nudging boundary conditions, used where ROMS_NUDGE_M3CLM / _TCLM / _M2CLM make them read "M3nudgcof" / "Tnudgcof" / "M2nudgcof" */
  double obcfac;
  /* ... and VolCons(west|south|east|north) of roms.in (mod_scalars.F: VolCons): bit ROMS_IWEST .. ROMS_INORTH set = the volume is
     conserved across that open edge -- obc_volcons.F:60 obc_flux_tile behind every barotropic call, :236 set_DUV_bc_tile in front
     of the next (step2d_LF_AM3.h:724, :2885).  One tile only: the sum over the tiles (mp_reduce) is not built */
  int volcons;
} roms_hip_config;

/* time indices of mod_stepping.F / mod_scalars.F that the kernel wrappers read */
typedef struct roms_hip_stepping {
  int iic, iif;
  int nstp, nnew, nrhs;
  int kstp, knew, krhs, indx1;
  int predictor;                 /* PREDICTOR_2D_STEP(ng) */
  double time;                   /* time(ng), seconds */
} roms_hip_stepping;

typedef struct roms_hip_ctx roms_hip_ctx;

/* life cycle (ROMS_allocate_arrays / ROMS_deallocate_arrays, mod_arrays.F) */
int roms_hip_create(const roms_hip_config *cfg, roms_hip_ctx **ctx);
int roms_hip_destroy(roms_hip_ctx *ctx);
const char *roms_hip_last_error(void);
int roms_hip_abi_version(void);

/* state transfer, whole arrays in the reference layout.  name = the reference's
   component name ("zeta","u","t","Hz","pm","Akv","DU_avg1",...).  n = element count
   (checked).  roms_hip_field_size returns the count, or -1 for an unknown name. */
long roms_hip_field_size(roms_hip_ctx *ctx, const char *name);
int roms_hip_upload(roms_hip_ctx *ctx, const char *name, const double *host, long n);
int roms_hip_download(roms_hip_ctx *ctx, const char *name, double *host, long n);
int roms_hip_sync(roms_hip_ctx *ctx);

/* time indices for the per-kernel entry points below */
int roms_hip_set_stepping(roms_hip_ctx *ctx, const roms_hip_stepping *s);
int roms_hip_get_stepping(roms_hip_ctx *ctx, roms_hip_stepping *s);

/* ---- one entry per reference procedure  kernel(ng,tile)  (asynchronous on the
        context's stream; state stays on the device) ---- */
int roms_hip_set_depth(roms_hip_ctx *ctx);     /* set_depth      set_depth.F:28      */
int roms_hip_set_massflux(roms_hip_ctx *ctx);  /* set_massflux   set_massflux.F:26   */
int roms_hip_rho_eos(roms_hip_ctx *ctx);       /* rho_eos        rho_eos.F:53        */
int roms_hip_set_vbc(roms_hip_ctx *ctx);       /* set_vbc        set_vbc.F:46        */
int roms_hip_ana_vmix(roms_hip_ctx *ctx);      /* ana_vmix       ana_vmix.h          */
int roms_hip_set_data(roms_hip_ctx *ctx);      /* set_data       set_data.F:19 (analytic forcing) */
int roms_hip_omega(roms_hip_ctx *ctx);         /* omega          omega.F:35          */
int roms_hip_wvelocity(roms_hip_ctx *ctx, int ninp); /* wvelocity wvelocity.F:27     */
int roms_hip_set_zeta(roms_hip_ctx *ctx);      /* set_zeta       set_zeta.F:23       */
int roms_hip_ini_zeta(roms_hip_ctx *ctx);      /* ini_zeta       ini_fields.F:700    */
int roms_hip_ini_fields(roms_hip_ctx *ctx);    /* ini_fields     ini_fields.F:47     */
int roms_hip_pre_step3d(roms_hip_ctx *ctx);    /* pre_step3d     pre_step3d.F:48     */
int roms_hip_prsgrd(roms_hip_ctx *ctx);        /* prsgrd         prsgrd32.h:33       */
int roms_hip_t3dmix2(roms_hip_ctx *ctx);       /* t3dmix2        t3dmix2_s.h / t3dmix2_geo.h */
int roms_hip_uv3dmix2(roms_hip_ctx *ctx);      /* uv3dmix2       uv3dmix2_s.h:40     */
int roms_hip_rhs3d_tile(roms_hip_ctx *ctx);    /* rhs3d_tile     rhs3d.F:196         */
int roms_hip_rhs3d(roms_hip_ctx *ctx);         /* rhs3d          rhs3d.F:25 (the five above in order) */
int roms_hip_step2d(roms_hip_ctx *ctx);        /* step2d         step2d_LF_AM3.h:18  */
/* step2d twice -- the predictor (main3d.F:839) and the corrector (:888) of ONE fast step 2 <= iif <= nfast -- as one
   launch.  The stepping is the PREDICTOR call's (predictor = 1, knew = 3, krhs = indx1, kstp = 3-indx1); afterwards the
   caller advances the indices as if it had made both calls.  The corrector's zeta/ubar/vbar(knew) stay in a staging
   level inside the library until the NEXT roms_hip_step2d_pair / roms_hip_step2d (a predictor call, krhs = that level:
   the auxiliary last call at the latest) commits them; roms_hip_main3d sequences this itself.  Same bits as two calls. */
int roms_hip_step2d_pair(roms_hip_ctx *ctx);
/* The fast steps iif = 2 .. nfast (main3d.F:810-918: 2*(nfast-1) calls of step2d) as ONE persistent launch: every block
   keeps its sub-tile in LDS / registers and exchanges the corrector's rim with its neighbours through arrival words
   (k_step2d_loop.h).  The stepping is the predictor call's of iif = 2; afterwards the caller sets the indices as the
   corrector of iif = nfast leaves them (indx1 flipped nfast-1 times) and makes the auxiliary call iif = nfast+1 with
   roms_hip_step2d, which commits the staged result.  Called with the stepping of the predictor call of iif = 1 instead,
   the launch covers the WHOLE loop of main3d.F:810-918 -- the first fast step (forward-Euler start, conversion of the 3-D
   forcing, step2d_LF_AM3.h:2225-2460) and the auxiliary call (final averages :821-883) too; the indices afterwards are
   those the auxiliary call leaves (iif = nfast+1, indx1 flipped nfast times, kstp = indx1, knew = 3-indx1, krhs = 3).  Single tile, at least one periodic direction, no land mask, up to
   64 K points (every sub-tile needs a compute unit of its own); exit_flag 8 elsewhere -- roms_hip_main3d decides itself
   (ROMS_HIP_LOOP=0: never).  Same bits as the calls it replaces. */
int roms_hip_step2d_loop(roms_hip_ctx *ctx);
int roms_hip_step3d_uv(roms_hip_ctx *ctx);     /* step3d_uv      step3d_uv.F:40      */
int roms_hip_step3d_t(roms_hip_ctx *ctx);      /* step3d_t       step3d_t.F:40       */
int roms_hip_lmd_vmix(roms_hip_ctx *ctx);      /* lmd_vmix       lmd_vmix.F:45       */
int roms_hip_bulk_flux(roms_hip_ctx *ctx);     /* bulk_flux      bulk_flux.F:100     */
int roms_hip_gls_prestep(roms_hip_ctx *ctx);   /* gls_prestep    gls_prestep.F:42  (main3d.F:636, behind rhs3d); MY25_MIXING: my25_prestep.F:42 (:634)  */
int roms_hip_gls_corstep(roms_hip_ctx *ctx);   /* gls_corstep    gls_corstep.F:52  (main3d.F:1021, behind omega); MY25_MIXING: my25_corstep.F:52 (:1019) */
/* diag diag.F:30 -- synchronises; out must hold 16 doubles: out[0..11] = avgke avgpe avgkp volume
   maxspeed max_Cu max_Cv max_Cw max_Ci max_Cj max_Ck max_C of this context's tile, out[12..13] =
   the un-normalised kinetic / potential energy sums (a multi-tile caller adds out[3], out[12],
   out[13] over the tiles and takes the maximum of out[4], out[11], as mp_reduce does in diag.F:331) */
int roms_hip_diag(roms_hip_ctx *ctx, double *out);
/* The report of the last diag that ran INSIDE roms_hip_main3d (main3d.F:355: at the start of every step
   with MOD(iic-1,ninfo) = 0), same layout; out[14] = the step count iic-1 it belongs to, or -1 if none
   ran yet.  Synchronises.  This is what the reference prints per NINFO steps (diag.F:473-500). */
int roms_hip_last_diag(roms_hip_ctx *ctx, double *out);

/* The BOUNDS(ng)/DOMAIN(ng) entries of this context's tile as the library derived them from the config
   (get_bounds.F:1044-1884; mod_param.F:88-175 lists the members): out must hold 54 ints,
     LBi UBi LBj UBj  Istr Iend Jstr Jend  IstrR IendR JstrR JendR  IstrU JstrV  IstrB IendB IstrM  JstrB JendB
     JstrM  IstrP IendP JstrP JendP  IstrT IendT JstrT JendT  Istrm3 Istrm2 Istrm1 IstrUm2 IstrUm1  Iendp1
     Iendp2 Iendp2i Iendp3  Jstrm3 Jstrm2 Jstrm1 JstrVm2 JstrVm1  Jendp1 Jendp2 Jendp2i Jendp3
     Western Eastern Southern Northern _Edge  SouthWest SouthEast NorthWest NorthEast _Corner (0/1).
   For a caller that wants to assert its own BOUNDS against the library's before the first step. */
int roms_hip_get_bounds(roms_hip_ctx *ctx, int *out);

/* Time-averaged fields, set_avg (ROMS/Nonlinear/set_avg.F:51, called at main3d.F:562; cpp option AVERAGES).
   roms_hip_avg_config: the averaging window -- nAVG steps per record (0: off), accumulation from step ntsAVG on,
   nrrec and ntstart of a restarted run as in set_avg.F:251-254 -- and the fields to average, bit f of mask =
   field f of:  0 zeta  1 ubar  2 vbar  3 u  4 v  5 omega (W*pm*pn)  6 w  7 rho  8 the tracers  9 zeta2  10 ubar2
   11 vbar2  12 uu  13 vv  14 uv  15 Huon  16 Hvom  17 <t*t>  18 <u*t>  19 <v*t>  20 <Huon*t>  21 <Hvom*t>
   (the Aout switches of roms.in).  With a window configured roms_hip_main3d calls set_avg in every step where
   main3d.F does; roms_hip_set_avg is the kernel(ng,tile) entry for a caller that sequences main3d itself.  The
   averages are read with roms_hip_download under the names avg_zeta avg_ubar avg_vbar avg_u avg_v avg_omega avg_w
   avg_rho avg_t avg_ZZ avg_U2 avg_V2 avg_UU avg_VV avg_UV avg_Huon avg_Hvom avg_TT avg_UT avg_VT avg_HuonT
   avg_HvomT (tracer terms: NT blocks of N levels); they are complete -- divided by nAVG, ghost points filled -- after
   the step iic with MOD(iic-1,nAVG) = 0.  roms_hip_avg_time: AVGtime of the last completed window (wrt_avg.F). */
int roms_hip_avg_config(roms_hip_ctx *ctx, int nAVG, int ntsAVG, int nrrec, int ntstart, unsigned mask);
int roms_hip_set_avg(roms_hip_ctx *ctx);
int roms_hip_avg_time(roms_hip_ctx *ctx, double *avgtime);
/* DIAGNOSTICS_TS (ROMS/Modules/mod_diags.F, ROMS/Utility/set_diags.F; main3d.F:559): per-term tracer tendencies.
   roms_hip_dia_config allocates DIAGS(ng)%DiaTwrk / DiaTrc (i,j,k,itrc,idiag; term order of mod_scalars.F:4246-4262:
   hadv, xadv, yadv, vadv, [hdif, xdif, ydif, [sdif,]] vdif, rate) and avgzeta, and sets the window nDIA / ntsDIA (nrrec,
   ntstart of a restart); from then on the kernels of pre_step3d, t3dmix2 and step3d_t store their terms
   (pre_step3d.F:925, t3dmix2_s.h:293, t3dmix2_geo.h:409, t3dmix2_iso.h:428, step3d_t.F:908, :1357, :1716, :1892),
   roms_hip_main3d / roms_hip_output_point call set_diags where main3d.F does, and "DiaTwrk", "DiaTrc", "dia_zeta" can be
   downloaded.  exit_flag 5: MPDATA tracers, applications without SPLINES_VDIFF, WET_DRY contexts (ROMS_WET_DRY in
   roms_hip_config.options; the refusal holds whichever comes first).  The momentum terms (DIAGNOSTICS_UV): the option bit
   ROMS_DIAGNOSTICS_UV of roms_hip_config.options (ABI version 4). */
int roms_hip_dia_config(roms_hip_ctx *ctx, int nDIA, int ntsDIA, int nrrec, int ntstart);
int roms_hip_set_diags(roms_hip_ctx *ctx);
/* Biharmonic horizontal mixing along s-surfaces (UV_VIS4 + MIX_S_UV: uv3dmix4_s.h:119-627 and step2d_LF_AM3.h:1653-1920;
   TS_DIF4 + MIX_S_TS: t3dmix4_s.h:94-478): between roms_hip_create and roms_hip_start.  Upload "visc4_r", "visc4_p", "diff4"
   (the SQUARE ROOTS of VISC4, TNU4: inp_par.F:634, read_phypar.F:7840) and leave "visc2_r", "visc2_p", "diff2" zero.  UV_VIS4
   needs Nghost = 3 (inp_par.F:214).
   Between the option bit and roms_hip_start the caller uploads the three coefficient arrays. */
/* -> ROMS_UV_VIS4, ROMS_TS_DIF4 in roms_hip_config.options (ABI version 4; roms_hip_mix4_config of version 3 is gone) */
/* Wetting and drying (WET_DRY: ROMS/Nonlinear/wetdry.F:93-900 and the WET_DRY branches of step2d_LF_AM3.h:863,992,1617,
   2205-2222,2518-2667, prsgrd32.h:362,426, rhs3d.F:1709-1910, t3dmix2_s.h:239,279, uv3dmix2_s.h:276, step3d_uv.F:720-721,
   1187-1188,1359,1579 and its boundary rows, set_vbc.F:307-308,397 with LIMIT_BSTRESS :611-699 (globaldefs.h:160),
   ini_fields.F:294-403,850, zetabc.F:783-874, u2dbc_im.F:1190-1318, v2dbc_im.F:1239-1367, u3dbc_im.F:523,681): between
   roms_hip_create and roms_hip_start; Dcrit = DCRIT of roms.in (read_phypar.F:1021), the total depth below which a cell is
   dry.  Needs ROMS_MASKING (globaldefs.h:152 defines it with WET_DRY).  The masks are the fields "rmask_wet", "umask_wet",
   "vmask_wet", "pmask_wet" (of the fast steps while they run, time-averaged for the 3-D step behind them), "rmask_full",
   "umask_full", "vmask_full", "pmask_full" (wet x land: what the output files mask with) and "rmask_wet_avg".
   roms_hip_wetdry_ini: the initial masks from zeta(kstp) (initial.F:467; wetdry.F:355-490), before the first step.
   Open boundaries take the WET_DRY forms of zetabc.F:190 (Chapman), u2dbc_im.F:339 (Shchepetkin), u3dbc_im.F:174 ...
   Round 5: also the wet masks of bulk_flux.F:637-1312, pre_step3d.F:903 (solar source), t3dmix2_geo.h:231,263 and
   mpdata_adiff.F:463,686,927,1117,1132,1148 (KPP has no WET_DRY statement of its own).
   exit_flag 5 where the reference's WET_DRY statements are not built on the device: the closures GLS / MY25,
   isopycnic / biharmonic mixing, prsgrd31 / prsgrd40, no SPLINES_VVISC, averages, diagnostics. */
/* -> ROMS_WET_DRY in roms_hip_config.options with roms_hip_config.Dcrit (ABI version 4; roms_hip_wetdry_config is gone) */
int roms_hip_wetdry_ini(roms_hip_ctx *ctx);
/* DIAGNOSTICS_UV (mod_diags.F:174-222; the DiaU2rhs / DiaRU / DiaU3wrk statements of step2d_LF_AM3.h, rhs3d.F, prsgrd32.h,
   uv3dmix2_s.h, pre_step3d.F, step3d_uv.F): per-term momentum tendencies.  After roms_hip_dia_config (whose window it shares):
   allocates DIAGS(ng)%DiaU2wrk, DiaV2wrk, DiaRUbar, DiaRVbar, DiaU2int, DiaV2int, DiaRUfrc, DiaRVfrc, DiaU3wrk, DiaV3wrk,
   DiaRU, DiaRV, DiaU2d, DiaV2d, DiaU3d, DiaV3d -- downloadable under these names, laid out as the reference's, term order
   of mod_scalars.F:4264-4377 -- and switches the term stores on; set_diags accumulates DiaU2d ... DiaV3d. */
/* -> ROMS_DIAGNOSTICS_UV in roms_hip_config.options: roms_hip_dia_config then does this too (ABI version 4) */
int roms_hip_dia_time(roms_hip_ctx *ctx, double *DIAtime);

/* The reference writes its history and restart records in the middle of a step (CALL output, main3d.F:591,
   behind set_zeta): before a caller between two roms_hip_main3d calls downloads fields for output it brings
   the derived ones -- rho, Huon/Hvom, W, the surface fluxes, Akv/Akt/hsbl, zeta(1:2) = Zt_avg1 -- to that point
   of the step about to be taken.  The step itself recomputes all of them and is not changed by the call.
   wvelocity's result is left in the download-only field "w_out" (wvel keeps the previous step's values,
   which diag.F reads). */
int roms_hip_output_point(roms_hip_ctx *ctx);

/* initial.F:549-577 tail (set_massflux, omega, rho_eos at iic = ntstart) */
int roms_hip_start(roms_hip_ctx *ctx);
/* nsteps passes of main3d's STEP_LOOP (main3d.F:216-1148) with the state resident on
   the device; the stepping indices advance exactly as in the reference. */
int roms_hip_main3d(roms_hip_ctx *ctx, int nsteps);

/* per-region device timing (the reference's wclock regions, mod_strings.F:39-135):
   seconds accumulated since create / the last reset, measured with HIP events when
   enabled with roms_hip_profile(ctx,1).  region ids as in the reference (9 = 2D kernel,
   21 = rhs3d, 22 = pre_step3d, 23 = prsgrd, 34 = step3d_uv, 35 = step3d_t ...). */
int roms_hip_profile(roms_hip_ctx *ctx, int enable);
int roms_hip_region_seconds(roms_hip_ctx *ctx, int region, double *seconds, long *calls);
/* ---- one tile per GPU: halo exchange between contexts of different processes -------------
   A context created with NtileI*NtileJ > 1 owns tile `tile` (= itile + jtile*NtileI, the
   reference's tile/rank numbering, get_bounds.F:972) with arrays allocated LBi:UBi x LBj:UBj
   around [Istr,Iend]x[Jstr,Jend]: three ghost columns/rows on the low side, Nghost on the high
   side (the layout of the reference's periodic arrays), or the boundary points of the domain
   edge.  Wherever the reference calls mp_exchange2d/3d/4d (mp_exchange.F:28,1025,1755) the
   library applies the boundary conditions and packs the strips on the device (one launch),
   moves them -- full-height xi strips, full-width eta strips and the corner blocks of the
   diagonal neighbours in ONE message phase, which leaves the ghost zone exactly as the
   reference's xi phase followed by its eta phase -- and unpacks (one launch).  Two transports,
   one of which must be installed before roms_hip_start:
     roms_hip_comm_rccl      built-in RCCL send/recv on the context's stream (rank = tile), one
                             group per exchange point; the 128-byte unique id comes from
                             roms_hip_rccl_unique_id on rank 0 and is distributed by the caller
                             (MPI_Bcast, torch.distributed ...)
     roms_hip_set_exchange   a caller-supplied function: it receives DEVICE pointers (host
                             pointers in the CPU-emulated test build) after the stream has been
                             synchronised, must complete all sends and receives before it
                             returns, and returns 0 on success.  Message m is matched by
                             (peer, tag), tag = direction of travel: 0/1 = eastward/westward,
                             2/3 = northward/southward, 4..7 = NE, NW, SE, SW; messages are
                             listed in ascending tag order on both sides.
     roms_hip_comm_peer      built-in mailbox transport (round 2): every rank owns a slab of uncached
                             device memory holding, per direction, two receive buffers and an arrival
                             word; the neighbours map it (hipIpcOpenMemHandle: xGMI peer access) and
                             their PACK kernel stores the strips straight into it, the last block
                             releasing the arrival words; the receiver's UNPACK kernel waits for its
                             eight words and copies the strips into the ghost zone.  An exchange point
                             is two launches -- no send/receive calls, no staging copy.  The 128-byte
                             blob of roms_hip_peer_export of every rank (rank order) is distributed
                             by the caller and handed to roms_hip_comm_peer.  A message that does not
                             arrive within ROMS_HIP_PEER_TIMEOUT seconds (default 20) fails the next
                             call with exit_flag 2. */
typedef int (*roms_hip_exchange_fn)(void *user, int nsend, const int *send_peer, double *const *send_buf,
                                    const long *send_count, const int *send_tag, int nrecv,
                                    const int *recv_peer, double *const *recv_buf, const long *recv_count,
                                    const int *recv_tag);
int roms_hip_set_exchange(roms_hip_ctx *ctx, roms_hip_exchange_fn fn, void *user);
int roms_hip_rccl_unique_id(void *id128);
int roms_hip_comm_rccl(roms_hip_ctx *ctx, const void *id128, int nranks, int rank);
int roms_hip_peer_export(roms_hip_ctx *ctx, void *blob128);
int roms_hip_comm_peer(roms_hip_ctx *ctx, const void *blobs128, int nranks, int rank);
/* self-check of the installed transport: `reps` exchanges of a work plane coded with the global indices of its
   points; every ghost point received from a neighbour must hold the code of the point it images.  0, or exit_flag 2
   with the first wrong point in roms_hip_last_error.  roms_hip_comm_reset removes the installed transport (after a
   failed probe the caller may install another one). */
int roms_hip_exchange_probe(roms_hip_ctx *ctx, int reps);
/* soak test: `reps` exchange points back to back without a host synchronisation in between (narrow, tail and wide
   strips in turn), every plane coded with its repetition and verified on the device: an exchange that delivers the
   PREVIOUS repetition's strips -- a slot read too early, a stale line -- is counted.  0, or exit_flag 2. */
int roms_hip_exchange_soak(roms_hip_ctx *ctx, int reps);
int roms_hip_comm_reset(roms_hip_ctx *ctx);
/* round 6: self-check of the RIM PLANES -- the part of the mailbox slab through which the barotropic launches of a multi-tile run
   hand their rim to the neighbouring ranks themselves (k_step2d_loop.h, k_step2d_pair.h) instead of through exchange launches:
   index-coded values of every rank's own points into its neighbours' planes, all eight directions, verified on the device.
   Call it on every rank behind roms_hip_exchange_probe; if it fails on ANY rank, call roms_hip_rim_disable on every rank
   (the exchanges of rounds 3-5 stay in place of the rim hand-off).  Returns 0 where there is nothing to check. */
int roms_hip_rim_probe(roms_hip_ctx *ctx, int reps);
int roms_hip_rim_disable(roms_hip_ctx *ctx);
/* number of halo exchanges performed so far (0 for a single-tile context) */
long roms_hip_exchange_count(roms_hip_ctx *ctx);
/* ranks of the built-in RCCL communicator of this context (ncclCommCount); 0: roms_hip_comm_rccl was not called */
long roms_hip_rccl_ranks(roms_hip_ctx *ctx);

/* measurement aid (round 6): the duration of every baroclinic step of the following roms_hip_main3d calls, from HIP events
   recorded on the main stream at the step boundaries (main3d.F:216-1148, one pass each): roms_hip_step_timing(ctx, nmax)
   arms it for up to nmax steps (0: off), roms_hip_step_times synchronises and copies the milliseconds of the steps
   recorded so far (returns their number, at most cap).  bench.py reports min / median / max from a pass of its own. */
int roms_hip_step_timing(roms_hip_ctx *ctx, int nmax);
int roms_hip_step_times(roms_hip_ctx *ctx, double *ms, int cap);

/* measurement aid: `reps` launches of a plain streaming copy (kernel k_copy_probe) between two 3-D
   work arrays; *bytes_per_launch = bytes read + written by one launch.  Timed by the caller with
   roms_hip_kprof: the measured HBM ceiling quoted next to the roofline fractions. */
int roms_hip_copy_probe(roms_hip_ctx *ctx, int reps, long *bytes_per_launch);

/* per-kernel device timing with HIP events on the library's stream (process-wide table):
   mode 0 off; 1 = every launch, synchronous (breakdown pass); 2 = only launches of `kernel`,
   asynchronous event pairs resolved when the table is read (usable inside a timed region);
   3 = only launches of `kernel`, the pair filled by the launch itself (hipExtLaunchKernel's start /
   stop events: the dispatch's own begin and end timestamps, the duration rocprofv3 reports -- no
   marker packets, the gaps between back-to-back launches are not counted); 4 = every launch of every kernel
   that way, asynchronous, the side streams off as in mode 1 (each kernel's own duration, as a serial
   rocprofv3 --kernel-trace run reports it).
   roms_hip_kprof resets the table; roms_hip_kprof_get enumerates it (returns 8 past the end). */
int roms_hip_kprof(int mode, const char *kernel);
/* modes 2 and 3: time every `every`-th launch of the selected kernel (keeps the event overhead out
   of a timed region) */
int roms_hip_kprof_stride(int every);
/* mode 3: time `on` consecutive launches of the selected kernel out of every `period` (e.g. all launches of one step
   in ten: every position of a loop is sampled equally, at a tenth of the cost) */
int roms_hip_kprof_window(int on, int period);
/* mode 2 only: one event pair around `n` consecutive launches of the selected kernel (back-to-back
   launches, e.g. the barotropic loop): the event markers then do not inflate a short kernel.  A run
   interrupted by another launch is discarded and the library falls back to one pair per launch. */
int roms_hip_kprof_batch(int n);
int roms_hip_kprof_get(int index, char *name, int name_len, double *seconds, long *calls);

#ifdef __cplusplus
}
#endif
#endif
