!  ref_glue.F90 -- TEST INFRASTRUCTURE, not product code.
!
!  bind(C) driver that is linked against the reference's OWN Fortran sources
!  (compiled unmodified from /root/reference by oracle/ref/build_ref.sh) so
!  that tests and fixture generators can execute the reference kernels that
!  build in this container and read back their results.
!
!  It plays the role that inp_par/read_phypar (Utility/inp_par.F:138-226,
!  Utility/read_phypar.F) and ROMS_initialize/initial (Drivers/nl_roms.h:61,
!  Nonlinear/initial.F) play in the real program: those files depend on the
!  NetCDF Fortran module, which this image lacks, so they are NOT built and
!  nothing here stands in for NetCDF -- the glue only fills the reference's
!  module variables by hand and calls reference procedures.
!
!  step2d, omega, pre_step3d, rhs3d, step3d_uv, step3d_t ARE built (see
!  build_ref.sh for how mod_sources.F is compiled without NetCDF).  main3d.F,
!  set_data.F, initial.F, post_initial.F cannot be (NetCDF readers/writers);
!  ref_main3d / ref_set_data below call the reference kernels in the order
!  those files do, citing their lines.
!
#include "cppdefs.h"
#if defined UPWELLING
# define REF_APP 1
#elif defined BENCHMARK
# define REF_APP 2
#endif
      MODULE ref_glue
      USE, INTRINSIC :: iso_c_binding
      USE mod_kinds
      USE mod_param
      USE mod_parallel
      USE mod_scalars
      USE mod_stepping
      USE mod_iounits
      USE mod_ncparam
      USE mod_grid
      USE mod_ocean
      USE mod_coupling
      USE mod_forces
      USE mod_mixing
      USE mod_boundary
      USE mod_clima
#ifdef AVERAGES
      USE mod_average
#endif
#ifdef DIAGNOSTICS
      USE mod_diags
#endif
      implicit none
      integer, parameter :: ng = 1
      integer, save :: clima_flags = 0, clima_stat = 0
      character(len=16), save :: clima_env = ' '
      CONTAINS
!
!=======================================================================
!  ipar: 1 Lm, 2 Mm, 3 N, 4 NtileI, 5 NtileJ, 6 ndtfast, 7 ntimes,
!        8 Vtransform, 9 Vstretching, 10 EWperiodic, 11 NSperiodic,
!        12 Hadv(temp), 13 Vadv(temp), 14 Hadv(salt), 15 Vadv(salt)
!        (scheme codes: 1 A4, 2 C2, 3 C4, 4 HSIMT, 5 MPDATA, 6 SPLINES,
!         7 SPLIT_U3, 8 U3), 16 lmd_Jwt (water type)
!  rpar: 1 dt, 2 theta_s, 3 theta_b, 4 Tcline, 5 rho0, 6 R0, 7 T0, 8 S0,
!        9 Tcoef, 10 Scoef, 11 visc2, 12 tnu2(temp), 13 tnu2(salt),
!        14 Akt_bak(temp), 15 Akt_bak(salt), 16 Akv_bak, 17 rdrg,
!        18 rdrg2, 19 Zob, 20 Zos, 21 gamma2, 22 dstart, 23 blk_ZQ,
!        24 blk_ZT, 25 blk_ZW
!=======================================================================
!
      SUBROUTINE ref_configure (ipar, rpar) bind(C, name="ref_configure")
      USE tile_indices_mod, ONLY : tile_indices
      USE dateclock_mod,    ONLY : ref_clock
      integer(c_int), intent(in) :: ipar(*)
      real(c_double), intent(in) :: rpar(*)
      integer :: itrc, ibry, ivar, tile
      integer :: LBi, UBi, LBj, UBj, LBij, UBij

      Ngrids=1
      CALL initialize_parallel
      CALL allocate_param
      CALL allocate_parallel (Ngrids)
      CALL allocate_iounits (Ngrids)
      CALL allocate_stepping (Ngrids)
      IF (.not.allocated(GridsInLayer)) THEN
        allocate ( GridsInLayer(NestLayers) )
        GridsInLayer=1
      END IF
      IF (.not.allocated(GridNumber)) THEN
        allocate ( GridNumber(Ngrids,NestLayers) )
        GridNumber=1
      END IF
      Lm(ng)=ipar(1)
      Mm(ng)=ipar(2)
      N(ng)=ipar(3)
      NAT=2
      NtileI(ng)=ipar(4)
      NtileJ(ng)=ipar(5)
      NtileX(ng)=NtileI(ng)
      NtileE(ng)=NtileJ(ng)
      CALL initialize_param
      CALL allocate_scalars
      CALL initialize_scalars
!
!  Climatology nudging (read_phypar.F: LnudgeM3CLM, LtracerCLM, LnudgeTCLM, Lm3CLM of roms.in): switched on by the test
!  through the environment -- ROMS_REF_CLIMA = bit 0: 3-D momentum, bit itrc: tracer itrc.  The climatology and
!  coefficient arrays (set_data.F / ana_nudgcoef.h in a run) are then data the test puts through ref_field.
!
      clima_flags=0
      CALL GET_ENVIRONMENT_VARIABLE ('ROMS_REF_CLIMA', clima_env, STATUS=clima_stat)
      IF (clima_stat.eq.0) READ (clima_env,*,IOSTAT=clima_stat) clima_flags
      IF (clima_stat.ne.0) clima_flags=0
      stdout=6
      Master=.TRUE.
!
!  Variable metadata (mod_ncparam): the analytical routines print field
!  names from Vname(:,idXXXX); the table is read from the reference's own
!  ROMS/External/varinfo.yaml, as read_phypar does for keyword VARNAME.
!
      varname=ROOT_DIR//'/ROMS/External/varinfo.yaml'
      CALL allocate_ncparam
      CALL initialize_ncparam
!
!  Advection schemes (load_tadv, Utility/inp_decode.F).
!
      DO itrc=1,NT(ng)
        CALL set_adv (Hadvection(itrc,ng), ipar(12+2*(MIN(itrc,2)-1)))
        CALL set_adv (Vadvection(itrc,ng), ipar(13+2*(MIN(itrc,2)-1)))
      END DO
!
!  Lateral boundary conditions (load_lbc): periodic or closed only.
!
      EWperiodic(ng)=ipar(10).ne.0
      NSperiodic(ng)=ipar(11).ne.0
      DO ivar=1,nLBCvar
        DO ibry=1,4
          IF ((ibry.eq.iwest).or.(ibry.eq.ieast)) THEN
            LBC(ibry,ivar,ng)%periodic=EWperiodic(ng)
            LBC(ibry,ivar,ng)%closed=.not.EWperiodic(ng)
          ELSE
            LBC(ibry,ivar,ng)%periodic=NSperiodic(ng)
            LBC(ibry,ivar,ng)%closed=.not.NSperiodic(ng)
          END IF
        END DO
      END DO
!
!  VolCons(iwest:inorth): bits 0..3 of ipar(50) (round 6: obc_volcons.F); the running values start from their declarations
      DO ibry=1,4
        VolCons(ibry,ng)=BTEST(ipar(50),ibry-1)
      END DO
      bc_area=0.0_dp
      bc_flux=0.0_dp
      ubar_xs=0.0_dp
!
#if defined GLS_MIXING || defined MY25_MIXING
!  LBC(isMtke): ipar(46:49) at iwest, isouth, ieast, inorth (0 = the default above, 1 Clo, 3 Gra, 5 Rad: tkebc_im.F)
      DO ibry=1,4
        IF (ipar(45+ibry).ne.0) THEN
          LBC(ibry,isMtke,ng)%closed=ipar(45+ibry).eq.1
          LBC(ibry,isMtke,ng)%periodic=.FALSE.
          LBC(ibry,isMtke,ng)%gradient=ipar(45+ibry).eq.3
          LBC(ibry,isMtke,ng)%radiation=ipar(45+ibry).eq.5
        END IF
      END DO
#endif
!
!  Open boundaries: ipar(17+4*(v-1)+(ibry-1)) = kind of variable v (1 isFsur, 2 isUbar, 3 isVbar, 4 isUvel,
!  5 isVvel, 6.. isTvar) at edge ibry (iwest, isouth, ieast, inorth), coded as oracle/orc.h does (0 = leave the
!  default above; 1 Clo 2 Per 3 Gra 4 Cla 5 Rad 6 RadNud 7 Che 8 Cha 9 Fla 10 Shc); flags as load_lbc sets them
!  for the roms.in keywords (Utility/inp_decode.F:1616-1660).
!
      DO itrc=1,5+NT(ng)
        SELECT CASE (itrc)
          CASE (1); ivar=isFsur
          CASE (2); ivar=isUbar
          CASE (3); ivar=isVbar
          CASE (4); ivar=isUvel
          CASE (5); ivar=isVvel
          CASE DEFAULT; ivar=isTvar(itrc-5)
        END SELECT
        DO ibry=1,4
          tile=ipar(17+4*(MIN(itrc,7)-1)+(ibry-1))
          IF (tile.ne.0) THEN
            LBC(ibry,ivar,ng)%closed=.FALSE.
            LBC(ibry,ivar,ng)%periodic=.FALSE.
          END IF
          SELECT CASE (tile)
            CASE (1); LBC(ibry,ivar,ng)%closed=.TRUE.
            CASE (2); LBC(ibry,ivar,ng)%periodic=.TRUE.
            CASE (3); LBC(ibry,ivar,ng)%gradient=.TRUE.
            CASE (4); LBC(ibry,ivar,ng)%clamped=.TRUE.
                      LBC(ibry,ivar,ng)%acquire=.TRUE.
            CASE (5); LBC(ibry,ivar,ng)%radiation=.TRUE.
            CASE (6); LBC(ibry,ivar,ng)%radiation=.TRUE.
                      LBC(ibry,ivar,ng)%nudging=.TRUE.
                      LBC(ibry,ivar,ng)%acquire=.TRUE.
            CASE (7); LBC(ibry,ivar,ng)%Chapman_explicit=.TRUE.
            CASE (8); LBC(ibry,ivar,ng)%Chapman_implicit=.TRUE.
            CASE (9); LBC(ibry,ivar,ng)%Flather=.TRUE.
                      LBC(ibry,ivar,ng)%acquire=.TRUE.
                      LBC(ibry,isFsur,ng)%acquire=.TRUE.
            CASE (10); LBC(ibry,ivar,ng)%Shchepetkin=.TRUE.
                      LBC(ibry,ivar,ng)%acquire=.TRUE.
                      LBC(ibry,isFsur,ng)%acquire=.TRUE.
          END SELECT
        END DO
      END DO
!  the test feeds boundary data of every variable: have all of mod_boundary's arrays allocated
      IF (ipar(45).ne.0) THEN
        DO ivar=1,nLBCvar
          DO ibry=1,4
            LBC(ibry,ivar,ng)%acquire=.TRUE.
          END DO
        END DO
      END IF
!  nudging time scales (1/s) as inp_par.F:726-752 derives them: rpar(26+(ibry-1)+4*q), q = 0 FSobc_in, 1 FSobc_out,
!  2 M2obc_in, 3 M2obc_out, 4 M3obc_in, 5 M3obc_out, 6 Tobc_in(1), 7 Tobc_out(1), 8 Tobc_in(2), 9 Tobc_out(2)
      DO ibry=1,4
        FSobc_in (ng,ibry)=rpar(26+(ibry-1))
        FSobc_out(ng,ibry)=rpar(30+(ibry-1))
        M2obc_in (ng,ibry)=rpar(34+(ibry-1))
        M2obc_out(ng,ibry)=rpar(38+(ibry-1))
        M3obc_in (ng,ibry)=rpar(42+(ibry-1))
        M3obc_out(ng,ibry)=rpar(46+(ibry-1))
        obcfac(ng)=rpar(85)           ! OBCFAC itself: the conditions read it where climatology nudging gives them their time scales (u3dbc_im.F:117)
        DO itrc=1,NT(ng)
          Tobc_in (itrc,ng,ibry)=rpar(50+8*(MIN(itrc,2)-1)+(ibry-1))
          Tobc_out(itrc,ng,ibry)=rpar(54+8*(MIN(itrc,2)-1)+(ibry-1))
        END DO
      END DO
!
!  Physical parameters (read_phypar).
!
      ntimes(ng)=ipar(7)
      dt(ng)=rpar(1)
      ndtfast(ng)=ipar(6)
      nrrec(ng)=0
      ninfo(ng)=1
      Vtransform(ng)=ipar(8)
      Vstretching(ng)=ipar(9)
      theta_s(ng)=rpar(2)
      theta_b(ng)=rpar(3)
      Tcline(ng)=rpar(4)
      rho0=rpar(5)
      R0(ng)=rpar(6)
      T0(ng)=rpar(7)
      S0(ng)=rpar(8)
      Tcoef(ng)=rpar(9)
      Scoef(ng)=rpar(10)
      nl_visc2(ng)=rpar(11)
      nl_tnu2(1,ng)=rpar(12)
      nl_tnu2(2,ng)=rpar(13)
#ifdef UV_VIS4
!  (biharmonic variants: rpar(11:13) are VISC4 and TNU4 of roms.in; inp_par.F:634 and read_phypar.F:7840 take their square roots)
      nl_visc4(ng)=SQRT(ABS(rpar(11)))
#endif
#ifdef TS_DIF4
      nl_tnu4(1,ng)=SQRT(ABS(rpar(12)))
      nl_tnu4(2,ng)=SQRT(ABS(rpar(13)))
#endif
      Akt_bak(1,ng)=rpar(14)
      Akt_bak(2,ng)=rpar(15)
      Akv_bak(ng)=rpar(16)
      rdrg(ng)=rpar(17)
      rdrg2(ng)=rpar(18)
      Zob(ng)=rpar(19)
      Zos(ng)=rpar(20)
      gamma2(ng)=rpar(21)
      dstart=rpar(22)
      time_ref=0.0_dp
      CALL ref_clock (time_ref)
#ifdef WET_DRY
!  DCRIT of roms.in (read_phypar.F:1021)
      Dcrit(ng)=rpar(84)
#endif
#ifdef BULK_FLUXES
      blk_ZQ(ng)=rpar(23)
      blk_ZT(ng)=rpar(24)
      blk_ZW(ng)=rpar(25)
#endif
#if defined LMD_SKPP || defined SOLAR_SOURCE
      lmd_Jwt(ng)=ipar(16)
#endif
#if defined GLS_MIXING || defined MY25_MIXING
!  the GLS_* block of roms.in (read_phypar.F): rpar(66..)
      gls_p(ng)=rpar(66)
      gls_m(ng)=rpar(67)
      gls_n(ng)=rpar(68)
      gls_Kmin(ng)=rpar(69)
      gls_Pmin(ng)=rpar(70)
      gls_cmu0(ng)=rpar(71)
      gls_c1(ng)=rpar(72)
      gls_c2(ng)=rpar(73)
      gls_c3m(ng)=rpar(74)
      gls_c3p(ng)=rpar(75)
      gls_sigk(ng)=rpar(76)
      gls_sigp(ng)=rpar(77)
      Akk_bak(ng)=rpar(78)
      Akp_bak(ng)=rpar(79)
      charnok_alpha(ng)=rpar(80)
      zos_hsig_alpha(ng)=rpar(81)
      sz_alpha(ng)=rpar(82)
      crgban_cw(ng)=rpar(83)
#endif
!
!  What inp_par does after read_phypar (Utility/inp_par.F:210-226,...).
!
      ThreeGhostPoints=ANY(Hadvection(:,:)%MPDATA).or.                  &
     &                 ANY(Hadvection(:,:)%HSIMT)
#ifdef UV_VIS4
      ThreeGhostPoints=.TRUE.                        ! inp_par.F:214-216
#endif
      IF (ThreeGhostPoints) THEN
        NghostPoints=3
      ELSE
        NghostPoints=2
      END IF
      LprocessOBC(ng)=.TRUE.
      CALL tile_indices (iNLM, Im, Jm, Lm, Mm, BOUNDS, DOMAIN, IOBOUNDS)
      gorho0=g/rho0
      dtfast(ng)=dt(ng)/REAL(ndtfast(ng),r8)
      numthreads=1
      MyThread=0
      first_tile(ng)=0
      last_tile(ng)=NtileI(ng)*NtileJ(ng)-1
!
!  ROMS_allocate_arrays / ROMS_initialize_arrays (Modules/mod_arrays.F).
!
      tile=0
      LBi=BOUNDS(ng)%LBi(tile)
      UBi=BOUNDS(ng)%UBi(tile)
      LBj=BOUNDS(ng)%LBj(tile)
      UBj=BOUNDS(ng)%UBj(tile)
      LBij=BOUNDS(ng)%LBij
      UBij=BOUNDS(ng)%UBij
      CALL allocate_boundary (ng)
      IF (clima_flags.ne.0) THEN
        NTCLM(ng)=0
        IF (IAND(clima_flags,1).ne.0) THEN
          Lm3CLM(ng)=.TRUE.
          LnudgeM3CLM(ng)=.TRUE.
        END IF
        IF (IAND(clima_flags,32).ne.0) THEN
          Lm2CLM(ng)=.TRUE.
          LnudgeM2CLM(ng)=.TRUE.
        END IF
        DO itrc=1,NT(ng)
          IF (IAND(clima_flags,ISHFT(1,itrc)).ne.0) THEN
            LtracerCLM(itrc,ng)=.TRUE.
            LnudgeTCLM(itrc,ng)=.TRUE.
            NTCLM(ng)=NTCLM(ng)+1
          END IF
        END DO
      END IF
      CALL allocate_clima (ng, LBi, UBi, LBj, UBj)
      CALL allocate_coupling (ng, LBi, UBi, LBj, UBj)
      CALL allocate_forces (ng, LBi, UBi, LBj, UBj)
      CALL allocate_grid (ng, ExtractFlag(ng),                          &
     &                    LBi, UBi, LBj, UBj, LBij, UBij)
      CALL allocate_mixing (ng, LBi, UBi, LBj, UBj)
      CALL allocate_ocean (ng, LBi, UBi, LBj, UBj)
#ifdef AVERAGES
!  the Aout switches of roms_upwelling.in (:786-899), then the arrays they ask for
      Aout(idFsur,ng)=.TRUE.; Aout(idUbar,ng)=.TRUE.; Aout(idVbar,ng)=.TRUE.
      Aout(idUvel,ng)=.TRUE.; Aout(idVvel,ng)=.TRUE.; Aout(idWvel,ng)=.TRUE.; Aout(idOvel,ng)=.TRUE.
      Aout(idDano,ng)=.TRUE.
      Aout(idHUav,ng)=.TRUE.; Aout(idHVav,ng)=.TRUE.; Aout(idUUav,ng)=.TRUE.; Aout(idUVav,ng)=.TRUE.
      Aout(idVVav,ng)=.TRUE.; Aout(idU2av,ng)=.TRUE.; Aout(idV2av,ng)=.TRUE.; Aout(idZZav,ng)=.TRUE.
      DO tile=1,NT(ng)
        Aout(idTvar(tile),ng)=.TRUE.; Aout(idTTav(tile),ng)=.TRUE.; Aout(idUTav(tile),ng)=.TRUE.
        Aout(idVTav(tile),ng)=.TRUE.; Aout(iHUTav(tile),ng)=.TRUE.; Aout(iHVTav(tile),ng)=.TRUE.
      END DO
      CALL allocate_average (ng, LBi, UBi, LBj, UBj)
      DO tile=first_tile(ng),last_tile(ng)
        CALL initialize_average (ng, tile)
      END DO
#endif
#ifdef DIAGNOSTICS
!  the per-term tendencies of mod_diags.F (upwelling.h as shipped: DIAGNOSTICS_TS, DIAGNOSTICS_UV)
      CALL allocate_diags (ng, LBi, UBi, LBj, UBj)
      DO tile=first_tile(ng),last_tile(ng)
        CALL initialize_diags (ng, tile)
      END DO
#endif
      DO tile=first_tile(ng),last_tile(ng)
        CALL initialize_boundary (ng, tile, 0)
        CALL initialize_coupling (ng, tile, 0)
        CALL initialize_forces (ng, tile, 0)
        CALL initialize_grid (ng, tile, 0)
        CALL initialize_mixing (ng, tile, 0)
        CALL initialize_ocean (ng, tile, 0)
      END DO
!
!  Time-stepping state as set at the top of initial (Nonlinear/initial.F).
!
      iif(ng)=1
      indx1(ng)=1
      kstp(ng)=1
      krhs(ng)=1
      knew(ng)=1
      PREDICTOR_2D_STEP(ng)=.FALSE.
      iic(ng)=0
      nstp(ng)=1
      nrhs(ng)=1
      nnew(ng)=1
      tdays(ng)=dstart
      time(ng)=tdays(ng)*day2sec
      ntstart(ng)=INT((time(ng)-dstart*day2sec)/dt(ng))+1
      ntend(ng)=ntstart(ng)+ntimes(ng)-1
      ntfirst(ng)=ntstart(ng)
      END SUBROUTINE ref_configure

      SUBROUTINE set_adv (A, code)
      TYPE (T_ADV), intent(inout) :: A
      integer, intent(in) :: code
      A%AKIMA4=code.eq.1
      A%CENTERED2=code.eq.2
      A%CENTERED4=code.eq.3
      A%HSIMT=code.eq.4
      A%MPDATA=code.eq.5
      A%SPLINES=code.eq.6
      A%SPLIT_U3=code.eq.7
      A%UPSTREAM3=code.eq.8
      END SUBROUTINE set_adv
!
!=======================================================================
!  The part of "initial" (Nonlinear/initial.F:277-577) that builds here:
!  set_grid (ana_grid, set_scoord, set_weights, metrics), ini_hmixcoef,
!  set_depth, ana_initial, set_depth0, set_zeta_timeavg, set_depth,
!  set_massflux, omega, rho_eos.
!=======================================================================
!
      SUBROUTINE ref_initial () bind(C, name="ref_initial")
      USE analytical_mod
      USE metrics_mod,       ONLY : metrics
      USE ini_hmixcoef_mod,  ONLY : ini_hmixcoef
      USE ini_fields_mod,    ONLY : set_zeta_timeavg
      USE set_depth_mod,     ONLY : set_depth0, set_depth
      USE set_massflux_mod,  ONLY : set_massflux
      USE rho_eos_mod,       ONLY : rho_eos
      USE omega_mod,         ONLY : omega
      USE dateclock_mod,     ONLY : time_string
#ifdef WET_DRY
      USE wetdry_mod,        ONLY : wetdry
#endif
      integer :: tile
      DO tile=first_tile(ng),last_tile(ng)
        CALL ana_grid (ng, tile, iNLM)
      END DO
      CALL set_scoord (ng)
      CALL set_weights (ng)
      DO tile=first_tile(ng),last_tile(ng)
        CALL metrics (ng, tile, iNLM)
      END DO
      DO tile=first_tile(ng),last_tile(ng)
        CALL ini_hmixcoef (ng, tile, iNLM)
      END DO
      DO tile=first_tile(ng),last_tile(ng)
        CALL set_depth (ng, tile, iNLM)
      END DO
      DO tile=first_tile(ng),last_tile(ng)
        CALL ana_initial (ng, tile, iNLM)
      END DO
#ifdef WET_DRY
      DO tile=first_tile(ng),last_tile(ng)
        CALL wetdry (ng, tile, 1, .TRUE.)                  ! initial.F:467
      END DO
#endif
      DO tile=first_tile(ng),last_tile(ng)
        CALL set_depth0 (ng, tile, iNLM)
        CALL set_zeta_timeavg (ng, tile, iNLM)
        CALL set_depth (ng, tile, iNLM)
      END DO
      DO tile=first_tile(ng),last_tile(ng)
        CALL set_massflux (ng, tile, iNLM)
      END DO
      DO tile=first_tile(ng),last_tile(ng)
        CALL omega (ng, tile, iNLM)
        CALL rho_eos (ng, tile, iNLM)
      END DO
      iic(ng)=ntstart(ng)
      CALL time_string (time(ng), time_code(ng))           ! initial.F:862
      END SUBROUTINE ref_initial
!
!=======================================================================
!  The analytical branches of set_data_tile (Nonlinear/set_data.F:187-640)
!  in that file's order; set_data.F itself USEs the NetCDF field readers.
!=======================================================================
!
      SUBROUTINE ref_set_data (tile)
      USE analytical_mod
      integer, intent(in) :: tile
#ifdef ANA_CLOUD
      CALL ana_cloud (ng, tile, iNLM)                    ! :197
#endif
#ifdef ANA_TAIR
      CALL ana_tair (ng, tile, iNLM)                     ! :217
#endif
#ifdef ANA_HUMIDITY
      CALL ana_humid (ng, tile, iNLM)                    ! :237
#endif
#if defined SHORTWAVE && defined ANA_SRFLUX
      CALL ana_srflux (ng, tile, iNLM)                   ! :255
#endif
#if defined BULK_FLUXES && defined ANA_WINDS
      CALL ana_winds (ng, tile, iNLM)                    ! :329
#endif
#if defined BULK_FLUXES && defined ANA_RAIN
      CALL ana_rain (ng, tile, iNLM)                     ! :394
#endif
#if !defined BULK_FLUXES && defined ANA_STFLUX
      CALL ana_stflux (ng, tile, iNLM, itemp)            ! :412
#endif
#ifdef ANA_BTFLUX
      CALL ana_btflux (ng, tile, iNLM, itemp)            ! :455
#endif
#if defined SALINITY && defined ANA_SSFLUX
      CALL ana_stflux (ng, tile, iNLM, isalt)            ! :470
#endif
#if defined SALINITY && defined ANA_BSFLUX
      CALL ana_btflux (ng, tile, iNLM, isalt)            ! :519
#endif
#if !defined BULK_FLUXES && defined ANA_SMFLUX
      CALL ana_smflux (ng, tile, iNLM)                   ! :564
#endif
#if defined BULK_FLUXES && defined ANA_PAIR
      CALL ana_pair (ng, tile, iNLM)                     ! :628
#endif
#ifdef ANA_FSOBC
      CALL ana_fsobc (ng, tile, iNLM)                    ! :881
#endif
#ifdef ANA_M2OBC
      CALL ana_m2obc (ng, tile, iNLM)                    ! :1003
#endif
      END SUBROUTINE ref_set_data
!
!=======================================================================
!  nsteps passes of main3d's STEP_LOOP (Nonlinear/main3d.F:216-1148) made
!  of the reference's own kernels: same calls, same tile order.  Returns
!  the diag numbers of the last step in dg(1:8).
!=======================================================================
!
      SUBROUTINE ref_main3d (nsteps, dg) bind(C, name="ref_main3d")
      USE set_depth_mod,     ONLY : set_depth
      USE set_massflux_mod,  ONLY : set_massflux
      USE rho_eos_mod,       ONLY : rho_eos
      USE set_vbc_mod,       ONLY : set_vbc
      USE set_zeta_mod,      ONLY : set_zeta
      USE wvelocity_mod,     ONLY : wvelocity
      USE diag_mod,          ONLY : diag
      USE ini_fields_mod,    ONLY : ini_fields, ini_zeta
      USE omega_mod,         ONLY : omega
      USE rhs3d_mod,         ONLY : rhs3d
      USE step2d_mod,        ONLY : step2d
      USE step3d_uv_mod,     ONLY : step3d_uv
      USE step3d_t_mod,      ONLY : step3d_t
      USE analytical_mod
#ifdef LMD_MIXING
      USE lmd_vmix_mod,      ONLY : lmd_vmix
#endif
#ifdef GLS_MIXING
      USE gls_prestep_mod,   ONLY : gls_prestep
      USE gls_corstep_mod,   ONLY : gls_corstep
#endif
#ifdef MY25_MIXING
      USE my25_prestep_mod,  ONLY : my25_prestep
      USE my25_corstep_mod,  ONLY : my25_corstep
#endif
#ifdef BULK_FLUXES
      USE bulk_flux_mod,     ONLY : bulk_flux
#endif
      USE dateclock_mod,     ONLY : time_string
      integer(c_int), value :: nsteps
      real(c_double), intent(out) :: dg(*)
      integer :: istep, tile, my_iif, next_indx1
      DO istep=1,nsteps
        nstp(ng)=1+MOD(iic(ng)-ntstart(ng),2)              ! :222-229
        nnew(ng)=3-nstp(ng)
        nrhs(ng)=nstp(ng)
        tdays(ng)=time(ng)*sec2day
        DO tile=first_tile(ng),last_tile(ng),+1            ! :257-259
          CALL ref_set_data (tile)
        END DO
        IF (iic(ng).eq.ntstart(ng)) THEN                   ! :334 post_initial
          DO tile=first_tile(ng),last_tile(ng),+1          ! post_initial.F:55-58
            CALL ini_zeta (ng, tile, iNLM)
            CALL set_depth (ng, tile, iNLM)
          END DO
          DO tile=last_tile(ng),first_tile(ng),-1          ! post_initial.F:65-67
            CALL ini_fields (ng, tile, iNLM)
          END DO
        END IF
        DO tile=first_tile(ng),last_tile(ng),+1            ! :347-359
          CALL set_massflux (ng, tile, iNLM)
          CALL rho_eos (ng, tile, iNLM)
          CALL diag (ng, tile)
        END DO
        DO tile=first_tile(ng),last_tile(ng),+1            ! :431-449
#ifdef BULK_FLUXES
          CALL bulk_flux (ng, tile)
#endif
          CALL set_vbc (ng, tile)
        END DO
        DO tile=last_tile(ng),first_tile(ng),-1            ! :523-539
#if defined ANA_VMIX
          CALL ana_vmix (ng, tile, iNLM)
#elif defined LMD_MIXING
          CALL lmd_vmix (ng, tile)
#endif
          CALL omega (ng, tile, iNLM)
          CALL wvelocity (ng, tile, nstp(ng))
        END DO
        DO tile=first_tile(ng),last_tile(ng),+1            ! :554-564
          CALL set_zeta (ng, tile)
        END DO
        DO tile=last_tile(ng),first_tile(ng),-1            ! :630-639
          CALL rhs3d (ng, tile)
#ifdef MY25_MIXING
          CALL my25_prestep (ng, tile)
#elif defined GLS_MIXING
          CALL gls_prestep (ng, tile)
#endif
        END DO
        LOOP_2D : DO my_iif=1,nfast(ng)+1                  ! :810-918
          next_indx1=3-indx1(ng)
          IF (.not.PREDICTOR_2D_STEP(ng).and.                           &
     &        my_iif.le.(nfast(ng)+1)) THEN
            PREDICTOR_2D_STEP(ng)=.TRUE.
            iif(ng)=my_iif
            IF (iif(ng).eq.1) THEN
              kstp(ng)=indx1(ng)
            ELSE
              kstp(ng)=3-indx1(ng)
            END IF
            knew(ng)=3
            krhs(ng)=indx1(ng)
          END IF
          IF (my_iif.le.(nfast(ng)+1)) THEN                ! :853-859
            DO tile=last_tile(ng),first_tile(ng),-1
              CALL step2d (ng, tile)
            END DO
          END IF
          IF (PREDICTOR_2D_STEP(ng)) THEN                  ! :876-884
            PREDICTOR_2D_STEP(ng)=.FALSE.
            knew(ng)=next_indx1
            kstp(ng)=3-knew(ng)
            krhs(ng)=3
            IF (iif(ng).lt.(nfast(ng)+1)) indx1(ng)=next_indx1
          END IF
          IF (iif(ng).lt.(nfast(ng)+1)) THEN               ! :894-900
            DO tile=first_tile(ng),last_tile(ng),+1
              CALL step2d (ng, tile)
            END DO
          END IF
        END DO LOOP_2D
        DO tile=last_tile(ng),first_tile(ng),-1            ! :961-965
          CALL set_depth (ng, tile, iNLM)
        END DO
        DO tile=last_tile(ng),first_tile(ng),-1            ! :988-992
          CALL step3d_uv (ng, tile)
        END DO
        DO tile=first_tile(ng),last_tile(ng),+1            ! :1015-1023
          CALL omega (ng, tile, iNLM)
#ifdef MY25_MIXING
          CALL my25_corstep (ng, tile)
#elif defined GLS_MIXING
          CALL gls_corstep (ng, tile)
#endif
        END DO
        DO tile=last_tile(ng),first_tile(ng),-1            ! :1043-1047
          CALL step3d_t (ng, tile)
        END DO
        iic(ng)=iic(ng)+1                                  ! :1145-1148
        time(ng)=time(ng)+dt(ng)
        CALL time_string (time(ng), time_code(ng))
      END DO
      dg(1)=avgke
      dg(2)=avgpe
      dg(3)=avgkp
      dg(4)=volume
      dg(5)=max_speed
      dg(6)=REAL(iic(ng),r8)
      dg(7)=time(ng)
      dg(8)=REAL(indx1(ng),r8)
      END SUBROUTINE ref_main3d
!
!  idx out: 1 iic 2 iif 3 nstp 4 nnew 5 nrhs 6 kstp 7 knew 8 krhs
!           9 PREDICTOR_2D_STEP 10 indx1
      SUBROUTINE ref_get_stepping (idx, tm) bind(C, name="ref_get_stepping")
      integer(c_int), intent(out) :: idx(*)
      real(c_double), intent(out) :: tm
      idx(1)=iic(ng)
      idx(2)=iif(ng)
      idx(3)=nstp(ng)
      idx(4)=nnew(ng)
      idx(5)=nrhs(ng)
      idx(6)=kstp(ng)
      idx(7)=knew(ng)
      idx(8)=krhs(ng)
      idx(9)=MERGE(1,0,PREDICTOR_2D_STEP(ng))
      idx(10)=indx1(ng)
      tm=time(ng)
      END SUBROUTINE ref_get_stepping
!
!=======================================================================
!  Set the time-stepping indices the kernel wrappers read from
!  mod_stepping / mod_scalars.
!  idx: 1 iic, 2 iif, 3 nstp, 4 nnew, 5 nrhs, 6 kstp, 7 knew, 8 krhs,
!       9 PREDICTOR_2D_STEP
!=======================================================================
!
      SUBROUTINE ref_set_avg_window (n_avg, nts_avg, nrrec_in, ntstart_in) bind(C, name="ref_set_avg_window")
      integer(c_int), value :: n_avg, nts_avg, nrrec_in, ntstart_in
      nAVG(ng)=n_avg
      ntsAVG(ng)=nts_avg
      nrrec(ng)=nrrec_in
      ntstart(ng)=ntstart_in
      END SUBROUTINE ref_set_avg_window

      SUBROUTINE ref_set_dia_window (n_dia, nts_dia, nrrec_in, ntstart_in) bind(C, name="ref_set_dia_window")
      integer(c_int), value :: n_dia, nts_dia, nrrec_in, ntstart_in
      nDIA(ng)=n_dia
      ntsDIA(ng)=nts_dia
      nrrec(ng)=nrrec_in
      ntstart(ng)=ntstart_in
      END SUBROUTINE ref_set_dia_window

      SUBROUTINE ref_set_stepping (idx, tm) bind(C, name="ref_set_stepping")
      USE dateclock_mod,     ONLY : time_string
      integer(c_int), intent(in) :: idx(*)
      real(c_double), value :: tm
      iic(ng)=idx(1)
      iif(ng)=idx(2)
      IF (idx(10).gt.0) indx1(ng)=idx(10)
      nstp(ng)=idx(3)
      nnew(ng)=idx(4)
      nrhs(ng)=idx(5)
      kstp(ng)=idx(6)
      knew(ng)=idx(7)
      krhs(ng)=idx(8)
      PREDICTOR_2D_STEP(ng)=idx(9).ne.0
      time(ng)=tm
      tdays(ng)=time(ng)*sec2day
      CALL time_string (time(ng), time_code(ng))
      END SUBROUTINE ref_set_stepping
!
!=======================================================================
!  Call a reference kernel wrapper on every tile.  Returns 0, or -1 for
!  an unknown name.
!=======================================================================
!
      FUNCTION ref_call (cname) bind(C, name="ref_call") RESULT (ierr)
      USE analytical_mod
      USE set_depth_mod,     ONLY : set_depth
      USE set_massflux_mod,  ONLY : set_massflux
      USE rho_eos_mod,       ONLY : rho_eos
      USE prsgrd_mod,        ONLY : prsgrd
#ifdef TS_DIF2
      USE t3dmix2_mod,       ONLY : t3dmix2
#endif
#ifdef UV_VIS2
      USE uv3dmix2_mod,      ONLY : uv3dmix2
#endif
      USE set_vbc_mod,       ONLY : set_vbc
      USE set_zeta_mod,      ONLY : set_zeta
      USE wvelocity_mod,     ONLY : wvelocity
      USE diag_mod,          ONLY : diag
      USE ini_fields_mod,    ONLY : ini_fields, ini_zeta
      USE omega_mod,         ONLY : omega
      USE pre_step3d_mod,    ONLY : pre_step3d
      USE rhs3d_mod,         ONLY : rhs3d
      USE step2d_mod,        ONLY : step2d
      USE step3d_uv_mod,     ONLY : step3d_uv
      USE step3d_t_mod,      ONLY : step3d_t
#ifdef WET_DRY
      USE wetdry_mod,        ONLY : wetdry
#endif
#ifdef LMD_MIXING
      USE lmd_vmix_mod,      ONLY : lmd_vmix
#endif
#ifdef GLS_MIXING
      USE gls_prestep_mod,   ONLY : gls_prestep
      USE gls_corstep_mod,   ONLY : gls_corstep
#endif
#ifdef MY25_MIXING
      USE my25_prestep_mod,  ONLY : my25_prestep
      USE my25_corstep_mod,  ONLY : my25_corstep
#endif
#ifdef BULK_FLUXES
      USE bulk_flux_mod,     ONLY : bulk_flux
#endif
#ifdef AVERAGES
      USE set_avg_mod,       ONLY : set_avg
#endif
#ifdef DIAGNOSTICS
      EXTERNAL set_diags                         ! (set_diags.F is not a module)
#endif
      character(kind=c_char), intent(in) :: cname(*)
      integer(c_int) :: ierr
      character(len=32) :: name
      integer :: i, tile
      name=' '
      DO i=1,32
        IF (cname(i).eq.c_null_char) EXIT
        name(i:i)=cname(i)
      END DO
      ierr=0
      DO tile=first_tile(ng),last_tile(ng)
        SELECT CASE (TRIM(name))
#ifdef AVERAGES
          CASE ('set_avg')
            CALL set_avg (ng, tile)
#endif
#ifdef DIAGNOSTICS
          CASE ('set_diags')
            CALL set_diags (ng, tile)
#endif
#ifdef WET_DRY
          CASE ('wetdry')
!  initial.F:467 -- the initial wet/dry masks
            CALL wetdry (ng, tile, kstp(ng), .TRUE.)
#endif
          CASE ('set_depth')
            CALL set_depth (ng, tile, iNLM)
          CASE ('set_massflux')
            CALL set_massflux (ng, tile, iNLM)
          CASE ('rho_eos')
            CALL rho_eos (ng, tile, iNLM)
          CASE ('prsgrd')
            CALL prsgrd (ng, tile)
#ifdef TS_DIF2
          CASE ('t3dmix2')
            CALL t3dmix2 (ng, tile)
#endif
#ifdef UV_VIS2
          CASE ('uv3dmix2')
            CALL uv3dmix2 (ng, tile)
#endif
          CASE ('set_vbc')
            CALL set_vbc (ng, tile)
          CASE ('set_zeta')
            CALL set_zeta (ng, tile)
          CASE ('wvelocity')
            CALL wvelocity (ng, tile, nstp(ng))
          CASE ('diag')
            CALL diag (ng, tile)
          CASE ('omega')
            CALL omega (ng, tile, iNLM)
          CASE ('pre_step3d')
            CALL pre_step3d (ng, tile)
          CASE ('rhs3d')
            CALL rhs3d (ng, tile)
          CASE ('step2d')
            CALL step2d (ng, tile)
          CASE ('step3d_uv')
            CALL step3d_uv (ng, tile)
          CASE ('step3d_t')
            CALL step3d_t (ng, tile)
          CASE ('set_data')
            CALL ref_set_data (tile)
          CASE ('ini_zeta')
            CALL ini_zeta (ng, tile, iNLM)
          CASE ('ini_fields')
            CALL ini_fields (ng, tile, iNLM)
#ifdef ANA_VMIX
          CASE ('ana_vmix')
            CALL ana_vmix (ng, tile, iNLM)
#endif
#ifdef ANA_SMFLUX
          CASE ('ana_smflux')
            CALL ana_smflux (ng, tile, iNLM)
#endif
#ifdef ANA_STFLUX
          CASE ('ana_stflux')
            CALL ana_stflux (ng, tile, iNLM, itemp)
            CALL ana_stflux (ng, tile, iNLM, isalt)
#endif
#ifdef ANA_BTFLUX
          CASE ('ana_btflux')
            CALL ana_btflux (ng, tile, iNLM, itemp)
            CALL ana_btflux (ng, tile, iNLM, isalt)
#endif
#ifdef ANA_SRFLUX
          CASE ('ana_srflux')
            CALL ana_srflux (ng, tile, iNLM)
#endif
#ifdef BULK_FLUXES
          CASE ('ana_atm')
            CALL ana_winds (ng, tile, iNLM)
            CALL ana_tair (ng, tile, iNLM)
            CALL ana_pair (ng, tile, iNLM)
            CALL ana_humid (ng, tile, iNLM)
            CALL ana_rain (ng, tile, iNLM)
            CALL ana_cloud (ng, tile, iNLM)
          CASE ('bulk_flux')
            CALL bulk_flux (ng, tile)
#endif
#ifdef LMD_MIXING
          CASE ('lmd_vmix')
            CALL lmd_vmix (ng, tile)
#endif
#ifdef GLS_MIXING
          CASE ('gls_prestep')
            CALL gls_prestep (ng, tile)
          CASE ('gls_corstep')
            CALL gls_corstep (ng, tile)
#endif
          CASE DEFAULT
            ierr=-1
        END SELECT
      END DO
      END FUNCTION ref_call
!
!=======================================================================
!  mpdata_adiff_tile on caller-supplied Ta (scratch extents
!  IminS:ImaxS,JminS:JmaxS,N); returns Ua,Va,Wa.  Single tile only.
!=======================================================================
!
      SUBROUTINE ref_mpdata_adiff (itrc, Ta, Ua, Va, Wa)                &
     &                            bind(C, name="ref_mpdata_adiff")
      USE mpdata_adiff_mod
      integer(c_int), value :: itrc
      real(c_double), intent(in) :: Ta(*)
      real(c_double), intent(out) :: Ua(*), Va(*), Wa(*)
      integer :: tile, LBi, UBi, LBj, UBj
      integer :: IminS, ImaxS, JminS, JmaxS, n2, i, j, k
      real(r8), allocatable :: oHz(:,:,:), Tw(:,:,:)
      real(r8), allocatable :: Uw(:,:,:), Vw(:,:,:), Ww(:,:,:)
      tile=0
      LBi=BOUNDS(ng)%LBi(tile)
      UBi=BOUNDS(ng)%UBi(tile)
      LBj=BOUNDS(ng)%LBj(tile)
      UBj=BOUNDS(ng)%UBj(tile)
      IminS=BOUNDS(ng)%Istr(tile)-3
      ImaxS=BOUNDS(ng)%Iend(tile)+3
      JminS=BOUNDS(ng)%Jstr(tile)-3
      JmaxS=BOUNDS(ng)%Jend(tile)+3
      allocate ( oHz(IminS:ImaxS,JminS:JmaxS,N(ng)) )
      allocate ( Tw(IminS:ImaxS,JminS:JmaxS,N(ng)) )
      allocate ( Uw(IminS:ImaxS,JminS:JmaxS,N(ng)) )
      allocate ( Vw(IminS:ImaxS,JminS:JmaxS,N(ng)) )
      allocate ( Ww(IminS:ImaxS,JminS:JmaxS,0:N(ng)) )
      oHz=0.0_r8
      Uw=0.0_r8
      Vw=0.0_r8
      Ww=0.0_r8
      n2=(ImaxS-IminS+1)*(JmaxS-JminS+1)
      DO k=1,N(ng)
        DO j=JminS,JmaxS
          DO i=IminS,ImaxS
            Tw(i,j,k)=Ta(1+(i-IminS)+(j-JminS)*(ImaxS-IminS+1)+(k-1)*n2)
          END DO
        END DO
        DO j=BOUNDS(ng)%Jstrm2(tile),BOUNDS(ng)%Jendp2(tile)
          DO i=BOUNDS(ng)%Istrm2(tile),BOUNDS(ng)%Iendp2(tile)
            oHz(i,j,k)=1.0_r8/GRID(ng)%Hz(i,j,k)
          END DO
        END DO
      END DO
      CALL mpdata_adiff_tile (ng, tile,                                 &
     &                        LBi, UBi, LBj, UBj,                       &
     &                        IminS, ImaxS, JminS, JmaxS,               &
#ifdef MASKING
     &                        GRID(ng)%rmask, GRID(ng)%umask,           &
     &                        GRID(ng)%vmask,                           &
#endif
#ifdef WET_DRY
     &                        GRID(ng)%rmask_wet, GRID(ng)%umask_wet,   &
     &                        GRID(ng)%vmask_wet,                       &
#endif
     &                        GRID(ng)%pm, GRID(ng)%pn, GRID(ng)%omn,   &
     &                        GRID(ng)%om_u, GRID(ng)%on_v,             &
     &                        GRID(ng)%z_r, oHz,                        &
     &                        GRID(ng)%Huon, GRID(ng)%Hvom,             &
     &                        OCEAN(ng)%W,                              &
     &                        OCEAN(ng)%t(:,:,:,3,itrc),                &
     &                        Tw, Uw, Vw, Ww)
      DO k=1,N(ng)
        DO j=JminS,JmaxS
          DO i=IminS,ImaxS
            Ua(1+(i-IminS)+(j-JminS)*(ImaxS-IminS+1)+(k-1)*n2)=Uw(i,j,k)
            Va(1+(i-IminS)+(j-JminS)*(ImaxS-IminS+1)+(k-1)*n2)=Vw(i,j,k)
          END DO
        END DO
      END DO
      DO k=0,N(ng)
        DO j=JminS,JmaxS
          DO i=IminS,ImaxS
            Wa(1+(i-IminS)+(j-JminS)*(ImaxS-IminS+1)+k*n2)=Ww(i,j,k)
          END DO
        END DO
      END DO
      END SUBROUTINE ref_mpdata_adiff
!
!=======================================================================
!  Lateral boundary condition routines with explicit output index.
!=======================================================================
!
      SUBROUTINE ref_bc2d (kout) bind(C, name="ref_bc2d")
      USE zetabc_mod, ONLY : zetabc_tile
      USE u2dbc_mod,  ONLY : u2dbc_tile
      USE v2dbc_mod,  ONLY : v2dbc_tile
      integer(c_int), value :: kout
      integer :: tile, LBi, UBi, LBj, UBj
      integer :: IminS, ImaxS, JminS, JmaxS
      DO tile=first_tile(ng),last_tile(ng)
        LBi=BOUNDS(ng)%LBi(tile)
        UBi=BOUNDS(ng)%UBi(tile)
        LBj=BOUNDS(ng)%LBj(tile)
        UBj=BOUNDS(ng)%UBj(tile)
        IminS=BOUNDS(ng)%Istr(tile)-3
        ImaxS=BOUNDS(ng)%Iend(tile)+3
        JminS=BOUNDS(ng)%Jstr(tile)-3
        JmaxS=BOUNDS(ng)%Jend(tile)+3
        CALL zetabc_tile (ng, tile, LBi, UBi, LBj, UBj,                 &
     &                    IminS, ImaxS, JminS, JmaxS,                   &
     &                    krhs(ng), kstp(ng), kout, OCEAN(ng)%zeta)
        CALL u2dbc_tile (ng, tile, LBi, UBi, LBj, UBj,                  &
     &                   IminS, ImaxS, JminS, JmaxS,                    &
     &                   krhs(ng), kstp(ng), kout,                      &
     &                   OCEAN(ng)%ubar, OCEAN(ng)%vbar, OCEAN(ng)%zeta)
        CALL v2dbc_tile (ng, tile, LBi, UBi, LBj, UBj,                  &
     &                   IminS, ImaxS, JminS, JmaxS,                    &
     &                   krhs(ng), kstp(ng), kout,                      &
     &                   OCEAN(ng)%ubar, OCEAN(ng)%vbar, OCEAN(ng)%zeta)
      END DO
      END SUBROUTINE ref_bc2d

      SUBROUTINE ref_bc3d (nout) bind(C, name="ref_bc3d")
      USE t3dbc_mod, ONLY : t3dbc_tile
      USE u3dbc_mod, ONLY : u3dbc_tile
      USE v3dbc_mod, ONLY : v3dbc_tile
      integer(c_int), value :: nout
      integer :: tile, LBi, UBi, LBj, UBj, itrc, ic
      integer :: IminS, ImaxS, JminS, JmaxS
      DO tile=first_tile(ng),last_tile(ng)
        LBi=BOUNDS(ng)%LBi(tile)
        UBi=BOUNDS(ng)%UBi(tile)
        LBj=BOUNDS(ng)%LBj(tile)
        UBj=BOUNDS(ng)%UBj(tile)
        IminS=BOUNDS(ng)%Istr(tile)-3
        ImaxS=BOUNDS(ng)%Iend(tile)+3
        JminS=BOUNDS(ng)%Jstr(tile)-3
        JmaxS=BOUNDS(ng)%Jend(tile)+3
        ic=0                          ! (the compact index of the nudged tracers, as step3d_t.F:1845-1854 counts it)
        DO itrc=1,NT(ng)
          IF (LtracerCLM(itrc,ng).and.LnudgeTCLM(itrc,ng)) ic=ic+1
          CALL t3dbc_tile (ng, tile, itrc, ic, LBi, UBi, LBj, UBj,      &
     &                     N(ng), NT(ng), IminS, ImaxS, JminS, JmaxS,   &
     &                     nstp(ng), nout, OCEAN(ng)%t)
        END DO
        IF (nout.le.2) THEN
          CALL u3dbc_tile (ng, tile, LBi, UBi, LBj, UBj, N(ng),         &
     &                     IminS, ImaxS, JminS, JmaxS,                  &
     &                     nstp(ng), nout, OCEAN(ng)%u)
          CALL v3dbc_tile (ng, tile, LBi, UBi, LBj, UBj, N(ng),         &
     &                     IminS, ImaxS, JminS, JmaxS,                  &
     &                     nstp(ng), nout, OCEAN(ng)%v)
        END IF
      END DO
      END SUBROUTINE ref_bc3d
!
!=======================================================================
!  Integer tables: BOUNDS/DOMAIN of one tile (get_bounds.F / tile_indices).
!=======================================================================
!
      SUBROUTINE ref_get_bounds (tile, b) bind(C, name="ref_get_bounds")
      integer(c_int), value :: tile
      integer(c_int), intent(out) :: b(*)
      b(1)=BOUNDS(ng)%LBi(tile)
      b(2)=BOUNDS(ng)%UBi(tile)
      b(3)=BOUNDS(ng)%LBj(tile)
      b(4)=BOUNDS(ng)%UBj(tile)
      b(5)=BOUNDS(ng)%Istr(tile)
      b(6)=BOUNDS(ng)%Iend(tile)
      b(7)=BOUNDS(ng)%Jstr(tile)
      b(8)=BOUNDS(ng)%Jend(tile)
      b(9)=BOUNDS(ng)%IstrR(tile)
      b(10)=BOUNDS(ng)%IendR(tile)
      b(11)=BOUNDS(ng)%JstrR(tile)
      b(12)=BOUNDS(ng)%JendR(tile)
      b(13)=BOUNDS(ng)%IstrU(tile)
      b(14)=BOUNDS(ng)%JstrV(tile)
      b(15)=BOUNDS(ng)%IstrB(tile)
      b(16)=BOUNDS(ng)%IendB(tile)
      b(17)=BOUNDS(ng)%IstrM(tile)
      b(18)=BOUNDS(ng)%JstrB(tile)
      b(19)=BOUNDS(ng)%JendB(tile)
      b(20)=BOUNDS(ng)%JstrM(tile)
      b(21)=BOUNDS(ng)%IstrP(tile)
      b(22)=BOUNDS(ng)%IendP(tile)
      b(23)=BOUNDS(ng)%JstrP(tile)
      b(24)=BOUNDS(ng)%JendP(tile)
      b(25)=BOUNDS(ng)%IstrT(tile)
      b(26)=BOUNDS(ng)%IendT(tile)
      b(27)=BOUNDS(ng)%JstrT(tile)
      b(28)=BOUNDS(ng)%JendT(tile)
      b(29)=BOUNDS(ng)%Istrm3(tile)
      b(30)=BOUNDS(ng)%Istrm2(tile)
      b(31)=BOUNDS(ng)%Istrm1(tile)
      b(32)=BOUNDS(ng)%IstrUm2(tile)
      b(33)=BOUNDS(ng)%IstrUm1(tile)
      b(34)=BOUNDS(ng)%Iendp1(tile)
      b(35)=BOUNDS(ng)%Iendp2(tile)
      b(36)=BOUNDS(ng)%Iendp2i(tile)
      b(37)=BOUNDS(ng)%Iendp3(tile)
      b(38)=BOUNDS(ng)%Jstrm3(tile)
      b(39)=BOUNDS(ng)%Jstrm2(tile)
      b(40)=BOUNDS(ng)%Jstrm1(tile)
      b(41)=BOUNDS(ng)%JstrVm2(tile)
      b(42)=BOUNDS(ng)%JstrVm1(tile)
      b(43)=BOUNDS(ng)%Jendp1(tile)
      b(44)=BOUNDS(ng)%Jendp2(tile)
      b(45)=BOUNDS(ng)%Jendp2i(tile)
      b(46)=BOUNDS(ng)%Jendp3(tile)
      b(47)=MERGE(1,0,DOMAIN(ng)%Western_Edge(tile))
      b(48)=MERGE(1,0,DOMAIN(ng)%Eastern_Edge(tile))
      b(49)=MERGE(1,0,DOMAIN(ng)%Southern_Edge(tile))
      b(50)=MERGE(1,0,DOMAIN(ng)%Northern_Edge(tile))
      b(51)=MERGE(1,0,DOMAIN(ng)%SouthWest_Corner(tile))
      b(52)=MERGE(1,0,DOMAIN(ng)%SouthEast_Corner(tile))
      b(53)=MERGE(1,0,DOMAIN(ng)%NorthWest_Corner(tile))
      b(54)=MERGE(1,0,DOMAIN(ng)%NorthEast_Corner(tile))
      b(55)=NghostPoints
      b(56)=Im(ng)
      b(57)=Jm(ng)
      b(58)=NT(ng)
      b(59)=nfast(ng)
      b(60)=0
      END SUBROUTINE ref_get_bounds
!
!=======================================================================
!  Scalars and 1-D tables: s-coordinate, barotropic filter weights.
!  which: 1 sc_r(N) 2 Cs_r(N) 3 sc_w(0:N) 4 Cs_w(0:N)
!         5 weight(1,1:2*ndtfast) 6 weight(2,1:2*ndtfast)
!         7 {hc, hmin, hmax, xl, el, dtfast, avgke, avgpe, avgkp, volume,
!            max_speed, max_Cu, max_Cv, max_Cw, Cu_i, Cu_j, Cu_k}
!=======================================================================
!
      SUBROUTINE ref_get_table (which, a) bind(C, name="ref_get_table")
      integer(c_int), value :: which
      real(c_double), intent(out) :: a(*)
      integer :: k
      SELECT CASE (which)
        CASE (1)
          DO k=1,N(ng)
            a(k)=SCALARS(ng)%sc_r(k)
          END DO
        CASE (2)
          DO k=1,N(ng)
            a(k)=SCALARS(ng)%Cs_r(k)
          END DO
        CASE (3)
          DO k=0,N(ng)
            a(k+1)=SCALARS(ng)%sc_w(k)
          END DO
        CASE (4)
          DO k=0,N(ng)
            a(k+1)=SCALARS(ng)%Cs_w(k)
          END DO
        CASE (5)
          DO k=1,2*ndtfast(ng)
            a(k)=weight(1,k,ng)
          END DO
        CASE (6)
          DO k=1,2*ndtfast(ng)
            a(k)=weight(2,k,ng)
          END DO
        CASE (7)
          a(1)=hc(ng)
          a(2)=hmin(ng)
          a(3)=hmax(ng)
          a(4)=xl(ng)
          a(5)=el(ng)
          a(6)=dtfast(ng)
          a(7)=avgke
          a(8)=avgpe
          a(9)=avgkp
          a(10)=volume
          a(11)=max_speed
          a(12)=max_Cu
          a(13)=max_Cv
          a(14)=max_Cw
      END SELECT
      END SUBROUTINE ref_get_table
!
!=======================================================================
!  Field access by name.  dir=0: copy reference array -> buf;
!  dir=1: buf -> reference array.  Returns the element count, or -1.
!  Arrays travel whole, in the reference's own (column-major) layout.
!=======================================================================
!
      FUNCTION ref_field (cname, dir, buf) bind(C, name="ref_field")    &
     &                   RESULT (nel)
      character(kind=c_char), intent(in) :: cname(*)
      integer(c_int), value :: dir
      real(c_double), intent(inout) :: buf(*)
      integer(c_long) :: nel
      character(len=32) :: name
      integer :: i
      name=' '
      DO i=1,32
        IF (cname(i).eq.c_null_char) EXIT
        name(i:i)=cname(i)
      END DO
      nel=-1
      SELECT CASE (TRIM(name))
#define F2(nm,arr) CASE (nm); nel=SIZE(arr); CALL cp2(arr,SIZE(arr),dir,buf)
        F2('h',GRID(ng)%h)
        F2('f',GRID(ng)%f)
        F2('fomn',GRID(ng)%fomn)
        F2('pm',GRID(ng)%pm)
        F2('pn',GRID(ng)%pn)
        F2('om_r',GRID(ng)%om_r)
        F2('on_r',GRID(ng)%on_r)
        F2('om_u',GRID(ng)%om_u)
        F2('on_u',GRID(ng)%on_u)
        F2('om_v',GRID(ng)%om_v)
        F2('on_v',GRID(ng)%on_v)
        F2('om_p',GRID(ng)%om_p)
        F2('on_p',GRID(ng)%on_p)
        F2('omn',GRID(ng)%omn)
        F2('pmon_r',GRID(ng)%pmon_r)
        F2('pnom_r',GRID(ng)%pnom_r)
        F2('pmon_p',GRID(ng)%pmon_p)
        F2('pnom_p',GRID(ng)%pnom_p)
        F2('pmon_u',GRID(ng)%pmon_u)
        F2('pnom_u',GRID(ng)%pnom_u)
        F2('pmon_v',GRID(ng)%pmon_v)
        F2('pnom_v',GRID(ng)%pnom_v)
#ifdef MASKING
        F2('rmask',GRID(ng)%rmask)
        F2('umask',GRID(ng)%umask)
        F2('vmask',GRID(ng)%vmask)
        F2('pmask',GRID(ng)%pmask)
#endif
#ifdef WET_DRY
        F2('rmask_wet',GRID(ng)%rmask_wet)
        F2('umask_wet',GRID(ng)%umask_wet)
        F2('vmask_wet',GRID(ng)%vmask_wet)
        F2('pmask_wet',GRID(ng)%pmask_wet)
        F2('rmask_full',GRID(ng)%rmask_full)
        F2('umask_full',GRID(ng)%umask_full)
        F2('vmask_full',GRID(ng)%vmask_full)
        F2('pmask_full',GRID(ng)%pmask_full)
        F2('rmask_wet_avg',GRID(ng)%rmask_wet_avg)
#endif
        F2('grdscl',GRID(ng)%grdscl)
        F2('xr',GRID(ng)%xr)
        F2('xp',GRID(ng)%xp)
        F2('yp',GRID(ng)%yp)
        F2('yr',GRID(ng)%yr)
        F2('angler',GRID(ng)%angler)
#ifdef CURVGRID
        F2('dmde',GRID(ng)%dmde)
        F2('dndx',GRID(ng)%dndx)
#endif
#ifdef SPHERICAL
        F2('lonr',GRID(ng)%lonr)
        F2('latr',GRID(ng)%latr)
#endif
        F2('Hz',GRID(ng)%Hz)
        F2('z_r',GRID(ng)%z_r)
        F2('z_w',GRID(ng)%z_w)
        F2('z0_r',GRID(ng)%z0_r)
        F2('z0_w',GRID(ng)%z0_w)
        F2('Huon',GRID(ng)%Huon)
        F2('Hvom',GRID(ng)%Hvom)
        F2('zeta',OCEAN(ng)%zeta)
        F2('ubar',OCEAN(ng)%ubar)
        F2('vbar',OCEAN(ng)%vbar)
        F2('rzeta',OCEAN(ng)%rzeta)
        F2('rubar',OCEAN(ng)%rubar)
        F2('rvbar',OCEAN(ng)%rvbar)
        F2('u',OCEAN(ng)%u)
        F2('v',OCEAN(ng)%v)
        F2('t',OCEAN(ng)%t)
        F2('W',OCEAN(ng)%W)
        F2('wvel',OCEAN(ng)%wvel)
        F2('rho',OCEAN(ng)%rho)
        F2('pden',OCEAN(ng)%pden)
        F2('ru',OCEAN(ng)%ru)
        F2('rv',OCEAN(ng)%rv)
        F2('rhoA',COUPLING(ng)%rhoA)
        F2('rhoS',COUPLING(ng)%rhoS)
        F2('rufrc',COUPLING(ng)%rufrc)
        F2('rvfrc',COUPLING(ng)%rvfrc)
        F2('Zt_avg1',COUPLING(ng)%Zt_avg1)
        F2('DU_avg1',COUPLING(ng)%DU_avg1)
        F2('DU_avg2',COUPLING(ng)%DU_avg2)
        F2('DV_avg1',COUPLING(ng)%DV_avg1)
        F2('DV_avg2',COUPLING(ng)%DV_avg2)
        F2('sustr',FORCES(ng)%sustr)
        F2('svstr',FORCES(ng)%svstr)
        F2('bustr',FORCES(ng)%bustr)
        F2('bvstr',FORCES(ng)%bvstr)
        F2('stflx',FORCES(ng)%stflx)
        F2('btflx',FORCES(ng)%btflx)
        F2('stflux',FORCES(ng)%stflux)
        F2('btflux',FORCES(ng)%btflux)
        F2('Akv',MIXING(ng)%Akv)
        F2('Akt',MIXING(ng)%Akt)
        CASE ('tclm'); IF (clima_flags.gt.1) THEN; nel=SIZE(CLIMA(ng)%tclm); CALL cp2(CLIMA(ng)%tclm,SIZE(CLIMA(ng)%tclm),dir,buf); END IF
        CASE ('Tnudgcof'); IF (clima_flags.gt.1) THEN; nel=SIZE(CLIMA(ng)%Tnudgcof); CALL cp2(CLIMA(ng)%Tnudgcof,SIZE(CLIMA(ng)%Tnudgcof),dir,buf); END IF
        CASE ('ubarclm'); IF (IAND(clima_flags,32).ne.0) THEN; nel=SIZE(CLIMA(ng)%ubarclm); CALL cp2(CLIMA(ng)%ubarclm,SIZE(CLIMA(ng)%ubarclm),dir,buf); END IF
        CASE ('vbarclm'); IF (IAND(clima_flags,32).ne.0) THEN; nel=SIZE(CLIMA(ng)%vbarclm); CALL cp2(CLIMA(ng)%vbarclm,SIZE(CLIMA(ng)%vbarclm),dir,buf); END IF
        CASE ('M2nudgcof'); IF (IAND(clima_flags,32).ne.0) THEN; nel=SIZE(CLIMA(ng)%M2nudgcof); CALL cp2(CLIMA(ng)%M2nudgcof,SIZE(CLIMA(ng)%M2nudgcof),dir,buf); END IF
        CASE ('uclm'); IF (IAND(clima_flags,1).ne.0) THEN; nel=SIZE(CLIMA(ng)%uclm); CALL cp2(CLIMA(ng)%uclm,SIZE(CLIMA(ng)%uclm),dir,buf); END IF
        CASE ('vclm'); IF (IAND(clima_flags,1).ne.0) THEN; nel=SIZE(CLIMA(ng)%vclm); CALL cp2(CLIMA(ng)%vclm,SIZE(CLIMA(ng)%vclm),dir,buf); END IF
        CASE ('M3nudgcof'); IF (IAND(clima_flags,1).ne.0) THEN; nel=SIZE(CLIMA(ng)%M3nudgcof); CALL cp2(CLIMA(ng)%M3nudgcof,SIZE(CLIMA(ng)%M3nudgcof),dir,buf); END IF
#ifdef UV_VIS2
        F2('visc2_r',MIXING(ng)%visc2_r)
        F2('visc2_p',MIXING(ng)%visc2_p)
#endif
#ifdef UV_VIS4
        F2('visc4_r',MIXING(ng)%visc4_r)
        F2('visc4_p',MIXING(ng)%visc4_p)
#endif
#ifdef TS_DIF4
        F2('diff4',MIXING(ng)%diff4)
#endif
#ifdef TS_DIF2
        F2('diff2',MIXING(ng)%diff2)
#endif
        F2('rdrag',GRID(ng)%rdrag)
#ifdef UV_QDRAG
        F2('rdrag2',GRID(ng)%rdrag2)
#endif
#ifdef BV_FREQUENCY
        F2('bvf',MIXING(ng)%bvf)
#endif
#ifdef NONLIN_EOS
        F2('alpha',MIXING(ng)%alpha)
        F2('beta',MIXING(ng)%beta)
#endif
#if defined GLS_MIXING || defined MY25_MIXING
        F2('tke',MIXING(ng)%tke)
        F2('gls',MIXING(ng)%gls)
        F2('Lscale',MIXING(ng)%Lscale)
        F2('Akk',MIXING(ng)%Akk)
#endif
#ifdef GLS_MIXING
        F2('Akp',MIXING(ng)%Akp)
#endif
#ifdef LMD_BKPP
        F2('hbbl',MIXING(ng)%hbbl)
#endif
#ifdef LMD_SKPP
        F2('hsbl',MIXING(ng)%hsbl)
        F2('ghats',MIXING(ng)%ghats)
#endif
#ifdef SHORTWAVE
        F2('srflx',FORCES(ng)%srflx)
#endif
#ifdef DIAGNOSTICS_TS
        F2('DiaTwrk',DIAGS(ng)%DiaTwrk)
        F2('DiaTrc',DIAGS(ng)%DiaTrc)
        F2('dia_zeta',DIAGS(ng)%avgzeta)
#endif
#ifdef DIAGNOSTICS_UV
        F2('DiaU2wrk',DIAGS(ng)%DiaU2wrk)
        F2('DiaV2wrk',DIAGS(ng)%DiaV2wrk)
        F2('DiaRUbar',DIAGS(ng)%DiaRUbar)
        F2('DiaRVbar',DIAGS(ng)%DiaRVbar)
        F2('DiaU2int',DIAGS(ng)%DiaU2int)
        F2('DiaV2int',DIAGS(ng)%DiaV2int)
        F2('DiaRUfrc',DIAGS(ng)%DiaRUfrc)
        F2('DiaRVfrc',DIAGS(ng)%DiaRVfrc)
        F2('DiaU3wrk',DIAGS(ng)%DiaU3wrk)
        F2('DiaV3wrk',DIAGS(ng)%DiaV3wrk)
        F2('DiaRU',DIAGS(ng)%DiaRU)
        F2('DiaRV',DIAGS(ng)%DiaRV)
        F2('DiaU2d',DIAGS(ng)%DiaU2d)
        F2('DiaV2d',DIAGS(ng)%DiaV2d)
        F2('DiaU3d',DIAGS(ng)%DiaU3d)
        F2('DiaV3d',DIAGS(ng)%DiaV3d)
#endif
#ifdef AVERAGES
        F2('avg_zeta',AVERAGE(ng)%avgzeta)
        F2('avg_ubar',AVERAGE(ng)%avgu2d)
        F2('avg_vbar',AVERAGE(ng)%avgv2d)
        F2('avg_u',AVERAGE(ng)%avgu3d)
        F2('avg_v',AVERAGE(ng)%avgv3d)
        F2('avg_omega',AVERAGE(ng)%avgw3d)
        F2('avg_w',AVERAGE(ng)%avgwvel)
        F2('avg_rho',AVERAGE(ng)%avgrho)
        F2('avg_t',AVERAGE(ng)%avgt)
        F2('avg_ZZ',AVERAGE(ng)%avgZZ)
        F2('avg_U2',AVERAGE(ng)%avgU2)
        F2('avg_V2',AVERAGE(ng)%avgV2)
        F2('avg_UU',AVERAGE(ng)%avgUU)
        F2('avg_VV',AVERAGE(ng)%avgVV)
        F2('avg_UV',AVERAGE(ng)%avgUV)
        F2('avg_Huon',AVERAGE(ng)%avgHuon)
        F2('avg_Hvom',AVERAGE(ng)%avgHvom)
        F2('avg_TT',AVERAGE(ng)%avgTT)
        F2('avg_UT',AVERAGE(ng)%avgUT)
        F2('avg_VT',AVERAGE(ng)%avgVT)
        F2('avg_HuonT',AVERAGE(ng)%avgHuonT)
        F2('avg_HvomT',AVERAGE(ng)%avgHvomT)
#endif
#ifdef BULK_FLUXES
        F2('Uwind',FORCES(ng)%Uwind)
        F2('Vwind',FORCES(ng)%Vwind)
        F2('Tair',FORCES(ng)%Tair)
        F2('Pair',FORCES(ng)%Pair)
        F2('Hair',FORCES(ng)%Hair)
        F2('rain',FORCES(ng)%rain)
        F2('cloud',FORCES(ng)%cloud)
        F2('lhflx',FORCES(ng)%lhflx)
        F2('shflx',FORCES(ng)%shflx)
        F2('lrflx',FORCES(ng)%lrflx)
#ifdef EMINUSP
        F2('evap',FORCES(ng)%evap)
#endif
#endif
!  open-boundary data (mod_boundary.F; pointers, associated where LBC(...)%acquire)
#define FB(nm,arr) CASE (nm); IF (associated(arr)) THEN; nel=SIZE(arr); CALL cp2(arr,SIZE(arr),dir,buf); END IF
        FB('zeta_west',BOUNDARY(ng)%zeta_west)
        FB('zeta_east',BOUNDARY(ng)%zeta_east)
        FB('zeta_south',BOUNDARY(ng)%zeta_south)
        FB('zeta_north',BOUNDARY(ng)%zeta_north)
        FB('ubar_west',BOUNDARY(ng)%ubar_west)
        FB('ubar_east',BOUNDARY(ng)%ubar_east)
        FB('ubar_south',BOUNDARY(ng)%ubar_south)
        FB('ubar_north',BOUNDARY(ng)%ubar_north)
        FB('vbar_west',BOUNDARY(ng)%vbar_west)
        FB('vbar_east',BOUNDARY(ng)%vbar_east)
        FB('vbar_south',BOUNDARY(ng)%vbar_south)
        FB('vbar_north',BOUNDARY(ng)%vbar_north)
        FB('u_west',BOUNDARY(ng)%u_west)
        FB('u_east',BOUNDARY(ng)%u_east)
        FB('u_south',BOUNDARY(ng)%u_south)
        FB('u_north',BOUNDARY(ng)%u_north)
        FB('v_west',BOUNDARY(ng)%v_west)
        FB('v_east',BOUNDARY(ng)%v_east)
        FB('v_south',BOUNDARY(ng)%v_south)
        FB('v_north',BOUNDARY(ng)%v_north)
        FB('t_west',BOUNDARY(ng)%t_west)
        FB('t_east',BOUNDARY(ng)%t_east)
        FB('t_south',BOUNDARY(ng)%t_south)
        FB('t_north',BOUNDARY(ng)%t_north)
      END SELECT
      END FUNCTION ref_field

      SUBROUTINE cp2 (arr, n, dir, buf)
      integer, intent(in) :: n
      real(r8), intent(inout) :: arr(n)
      integer(c_int), intent(in) :: dir
      real(c_double), intent(inout) :: buf(*)
      integer :: i
      IF (dir.eq.0) THEN
        DO i=1,n
          buf(i)=arr(i)
        END DO
      ELSE
        DO i=1,n
          arr(i)=buf(i)
        END DO
      END IF
      END SUBROUTINE cp2

      END MODULE ref_glue
