!  roms_host.f90 -- Fortran host driver of the MI355X ROMS time step.
!
!  It does, on the host and once, what the reference does before its time loop -- read roms.in
!  (Utility/read_phypar.F keyword subset), partition (Utility/get_bounds.F), build the vertical
!  coordinate (Utility/set_scoord.F), the barotropic filter (Utility/set_weights.F), the analytic
!  grid/initial fields (Functionals/ana_grid.h, ana_initial.h), the metrics (Utility/metrics.F) and
!  mixing coefficients (Utility/ini_hmixcoef.F, Modules/mod_mixing.F) -- keeps the arrays in the
!  mod_grid/mod_ocean layout, hands them to the device through the C ABI (roms_hip_mod.f90) and
!  then drives main3d (Nonlinear/main3d.F) either kernel by kernel or through roms_hip_main3d.
!
      MODULE roms_host
      USE, INTRINSIC :: iso_c_binding
      USE roms_hip
      implicit none
      integer, parameter :: r8 = selected_real_kind(12,300), dp = r8
      integer, parameter :: r16 = selected_real_kind(15,300)   ! mod_kinds.F:50 (r16 is 64-bit on Linux builds)
      real(dp), parameter :: pi = 3.14159265358979323846_dp
      real(dp), parameter :: deg2rad = pi/180.0_dp
      real(dp), parameter :: Eradius = 6371315.0_dp, g = 9.81_dp, Cp = 3985.0_dp
      real(dp), parameter :: Falpha = 2.0_dp, Fbeta = 4.0_dp, Fgamma = 0.284_dp

!  ---- run-time parameters (roms.in) ----
      character(len=32) :: MyAppCPP = 'UPWELLING'
      integer :: Lm = 41, Mm = 80, N = 16, NAT = 2, NT = 2, NtileI = 1, NtileJ = 1
      integer :: ntimes = 100, ndtfast = 30, ninfo = 1, Vtransform = 2, Vstretching = 4, lmd_Jwt = 1
      integer :: hadv(ROMS_MAXT) = ROMS_U3, vadv(ROMS_MAXT) = ROMS_C4
      logical :: EWperiodic = .TRUE., NSperiodic = .FALSE.
      real(dp) :: dt = 300.0_dp, theta_s = 3.0_dp, theta_b = 0.0_dp, Tcline = 25.0_dp, rho0 = 1025.0_dp
      real(dp) :: R0 = 1027.0_dp, T0 = 14.0_dp, S0 = 35.0_dp, Tcoef = 1.7E-4_dp, Scoef = 0.0_dp
!  UV_VIS4 / TS_DIF4 (biharmonic mixing along s-surfaces): VISC4, TNU4 of roms.in [m4/s]; mix4(1:2): the header defines them
      real(dp) :: visc4 = 0.0_dp, tnu4(ROMS_MAXT) = 0.0_dp
      logical :: mix4(2) = .FALSE.
      real(dp) :: Dcrit = 0.10_dp              ! DCRIT of roms.in (WET_DRY; read_phypar.F:1021)
      logical :: wet_dry = .FALSE.
      logical :: ddmix = .FALSE.                ! LMD_DDMIX: double-diffusive mixing in lmd_vmix's interior scheme
      logical :: bkpp = .FALSE.                 ! LMD_BKPP: the bottom boundary layer behind lmd_skpp (lmd_bkpp.F)
      integer :: prs4x = 0                      ! 44: PJ_GRADPQ4 (prsgrd44.h), 42: PJ_GRADPQ2 (prsgrd42.h), 0: the scheme in `options`
      logical :: mix_geo_uv = .FALSE.           ! UV_VIS2 along geopotential surfaces (MIX_GEO_UV: uv3dmix2_geo.h)
      real(dp) :: visc2 = 5.0_dp, tnu2(ROMS_MAXT) = 0.0_dp, Akt_bak(ROMS_MAXT) = 1.0E-6_dp, Akv_bak = 1.0E-5_dp
      real(dp) :: rdrg = 3.0E-4_dp, rdrg2 = 3.0E-3_dp, Zob = 0.02_dp, Zos = 0.02_dp, gamma2 = 1.0_dp
      real(dp) :: dstart = 0.0_dp, time_ref = 0.0_dp, blk_ZQ = 10.0_dp, blk_ZT = 10.0_dp, blk_ZW = 10.0_dp
      integer :: options = 0
!  GLS_MIXING: the compile-time form (ROMS_GLS_* flags from the application header), the GLS_* block of roms.in
!  (defaults: the k-epsilon column of roms_upwelling.in), LBC(isMtke) per edge (west south east north)
      integer :: gls_flags = 0, lbc_tke(4) = 0
      real(dp) :: gls_p = 3.0_dp, gls_m = 1.5_dp, gls_n = -1.0_dp, gls_Kmin = 7.6E-6_dp, gls_Pmin = 1.0E-12_dp
      real(dp) :: gls_cmu0 = 0.5477_dp, gls_c1 = 1.44_dp, gls_c2 = 1.92_dp, gls_c3m = -0.4_dp, gls_c3p = 1.0_dp
      real(dp) :: gls_sigk = 1.0_dp, gls_sigp = 1.30_dp, Akk_bak = 5.0E-6_dp, Akp_bak = 5.0E-6_dp
      real(dp) :: charnok_alpha = 1400.0_dp, crgban_cw = 100.0_dp
!  output (read_phypar.F: NRREC ... NHIS, the Hout switches, the file names); written by roms_output.f90
      integer :: nrrec = 0, nRST = 0, nHIS = 0
      logical :: LcycleRST = .TRUE.
      character(len=256) :: ininame = 'roms_ini.nc', rstname = 'roms_rst.nc', hisname = 'roms_his.nc'
      integer, parameter :: idFsur = 1, idUbar = 2, idVbar = 3, idUvel = 4, idVvel = 5, idWvel = 6, idOvel = 7,    &
     &                      idTemp = 8, idSalt = 9, idDano = 10, idVvis = 11, idTdif = 12, idSdif = 13,            &
     &                      idHsbl = 14, nHout = 14
      logical :: Hout(nHout) = .FALSE.
      logical :: HoutMtke = .FALSE., HoutMtls = .FALSE.   ! Hout(idMtke), Hout(idMtls): tke; gls and Lscale (GLS_MIXING, MY25_MIXING)
!  time-averaged output (AVERAGES: set_avg.F, def_avg.F/wrt_avg.F): window, file, the Aout switches in the order of
!  roms_hip_avg_config's mask bits; tracer terms (bits 8, 17..21) per tracer
      integer :: nAVG = 0, ntsAVG = 1
      character(len=256) :: avgname = 'roms_avg.nc'
!  per-term tracer tendencies (DIAGNOSTICS_TS: mod_diags.F, set_diags.F, def_diags.F/wrt_diags.F): window, file, the
!  Dout(iT...) switches per tracer in the library's term order (hadv xadv yadv vadv hdif xdif ydif sdif vdif rate)
      integer :: nDIA = 0, ntsDIA = 1
      character(len=256) :: dianame = 'roms_dia.nc'
      logical :: DoutT(0:9,ROMS_MAXT) = .FALSE.
!  Dout(M2...), Dout(M3...): the momentum terms (DIAGNOSTICS_UV), in the library's order (roms_ctx.h: M2FCOR ... M2RATE,
!  M3FCOR ... M3RATE); diag_uv: the application header defines DIAGNOSTICS_UV
      logical :: DoutM2(0:10) = .FALSE., DoutM3(0:10) = .FALSE., diag_uv = .FALSE.
      integer, parameter :: nAout = 22
      logical :: Aout(0:nAout-1) = .FALSE., AoutT(0:nAout-1,ROMS_MAXT) = .FALSE.
      character(len=512) :: app_header = ' '      ! application header to read the cpp options from (optional)
      character(len=256) :: host_message = ' '    ! why the last set-up step returned a non-zero exit_flag
      character(len=256) :: title = ' '            ! TITLE of roms.in (echoed beside the application name, checkdefs.F:61)
      integer :: n_unused_keys = 0                ! roms.in keywords this build has no use for = n_inert_keys + n_unknown_keys
      integer :: n_inert_keys = 0                 ! ... that the reference's reader knows and that cannot change this build's time
                                                  !     step (docs/ROMS_IN_KEYWORDS.md; roms_in_inert.inc)
      integer :: n_unknown_keys = 0               ! ... that the reference's reader (read_phypar.F) does not know either
      character(len=64) :: first_unknown_key = ' ' 
      integer :: lbc_seen = 0
!  lateral boundary conditions: lbc(variable, edge) = ROMS_LBC_* (variable 1 isFsur, 2 isUbar, 3 isVbar, 4 isUvel, 5 isVvel,
!  6.. isTvar; edge 1 west, 2 south, 3 east, 4 north: the order of a roms.in LBC line), 0 = nothing said; the nudging
!  time scales of roms.in in days (ZNUDG M2NUDG M3NUDG TNUDG) and OBCFAC
      integer :: lbc(ROMS_NLBC,4) = 0
      real(dp) :: Znudg = 0.0_dp, M2nudg = 0.0_dp, M3nudg = 0.0_dp, Tnudg(ROMS_MAXT) = 0.0_dp, obcfac = 0.0_dp
      integer :: volcons = 0
      character(len=32) :: defs(256)              ! cpp options defined by the application header
      integer :: ndefs = 0
      character(len=64) :: xtok(64)               ! tokens of the #if condition being evaluated
      integer :: nxtok = 0, xpos = 1

!  ---- derived ----
      integer :: Nghost, Im, Jm, LBi, UBi, LBj, UBj, nfast
      integer :: Istr, Iend, Jstr, Jend, IstrR, IendR, JstrR, JendR, IstrT, IendT, JstrT, JendT
      integer :: IstrP, JstrP
      integer :: my_tile = 0, tIstr, tIend, tJstr, tJend, tLBi, tUBi, tLBj, tUBj    ! this process' tile
      real(dp) :: dtfast, hc, hmin, hmax, xl, el
      real(dp), allocatable :: weight(:,:), sc_r(:), Cs_r(:), sc_w(:), Cs_w(:)

!  ---- mod_grid / mod_ocean / mod_mixing / mod_coupling arrays owned by the host ----
      real(r8), allocatable, target :: h(:,:), f(:,:), fomn(:,:), pm(:,:), pn(:,:), om_r(:,:), on_r(:,:)
      real(r8), allocatable, target :: om_u(:,:), on_u(:,:), om_v(:,:), on_v(:,:), om_p(:,:), on_p(:,:), omn(:,:)
      real(r8), allocatable, target :: pmon_r(:,:), pnom_r(:,:), pmon_p(:,:), pnom_p(:,:), pmon_u(:,:)
      real(r8), allocatable, target :: pnom_u(:,:), pmon_v(:,:), pnom_v(:,:), dmde(:,:), dndx(:,:), angler(:,:)
      real(r8), allocatable, target :: xr(:,:), yr(:,:), lonr(:,:), latr(:,:), rdrag(:,:), rdrag2(:,:)
      real(r8), allocatable, target :: xp(:,:), yp(:,:)          ! psi-point coordinates (the KELVIN boundary data read them)
      real(r8), allocatable, target :: visc2_r(:,:), visc2_p(:,:), diff2(:,:,:)
      real(r8), allocatable, target :: rmask(:,:), umask(:,:), vmask(:,:), pmask(:,:)     ! MASKING (all water otherwise)
      real(r8), allocatable, target :: Hz(:,:,:), z_r(:,:,:), z_w(:,:,:)
      real(r8), allocatable, target :: zeta(:,:,:), ubar(:,:,:), vbar(:,:,:), u(:,:,:,:), v(:,:,:,:), t(:,:,:,:,:)
      real(r8), allocatable, target :: Akv(:,:,:), Akt(:,:,:,:), Zt_avg1(:,:)

      TYPE (c_ptr) :: ctx = c_null_ptr
      TYPE (roms_hip_stepping) :: step
      integer :: exit_flag = 0

      CONTAINS
!
!=======================================================================
!  roms.in reader: free-format "KEYWORD == value ... ! comment" with "\" continuation, d-exponents
!  and n*value repeats (subset of Utility/read_phypar.F + inp_decode.F).
!=======================================================================
!
      SUBROUTINE read_roms_in (fname, ierr)
      character(len=*), intent(in) :: fname
      integer, intent(out) :: ierr
      integer :: iu, ios, ic, ieq, nv, itr
      character(len=512) :: line, val
      character(len=64) :: key
      character(len=64) :: tok(16)
      logical :: cont
      ierr=0
      CALL set_defaults ()
      host_message=' '
      n_unused_keys=0
      n_inert_keys=0
      n_unknown_keys=0
      first_unknown_key=' '
      lbc_seen=0
      open (newunit=iu, file=TRIM(fname), status='old', action='read', iostat=ios)
      IF (ios.ne.0) THEN
        host_message='cannot open '//TRIM(fname)
        ierr=2
        RETURN
      END IF
      itr=0
      DO
        read (iu,'(a)',iostat=ios) line
        IF (ios.ne.0) EXIT
        ic=INDEX(line,'!')
        IF (ic.gt.0) line(ic:)=' '
        ieq=INDEX(line,'=')
        IF (ieq.le.1) CYCLE
        key=ADJUSTL(line(1:ieq-1))
        val=line(ieq+1:)
        IF (val(1:1).eq.'=') val=val(2:)
!  continuation lines
        cont=INDEX(val,'\').gt.0
        DO WHILE (cont)
          val(INDEX(val,'\'):)=' '
          read (iu,'(a)',iostat=ios) line
          IF (ios.ne.0) EXIT
          ic=INDEX(line,'!')
          IF (ic.gt.0) line(ic:)=' '
          cont=INDEX(line,'\').gt.0
          IF (cont) line(INDEX(line,'\'):)=' '
          val=TRIM(val)//' '//TRIM(ADJUSTL(line))
        END DO
        CALL split (val, tok, nv)
        IF (nv.eq.0) CYCLE
        SELECT CASE (TRIM(key))
          CASE ('MyAppCPP');    MyAppCPP=tok(1)
          CASE ('TITLE');       title=ADJUSTL(val)
          CASE ('Lm');          Lm=toint(tok(1))
          CASE ('Mm');          Mm=toint(tok(1))
          CASE ('N');           N=toint(tok(1))
          CASE ('NAT');         NAT=toint(tok(1))
          CASE ('NtileI');      NtileI=toint(tok(1))
          CASE ('NtileJ');      NtileJ=toint(tok(1))
          CASE ('NTIMES');      ntimes=toint(tok(1))
          CASE ('DT');          dt=toreal(tok(1))
          CASE ('NDTFAST');     ndtfast=toint(tok(1))
          CASE ('NINFO');       ninfo=toint(tok(1))
          CASE ('Hadvection');  CALL load_tadv (tok, nv, hadv)
          CASE ('Vadvection');  CALL load_tadv (tok, nv, vadv)
          CASE ('LBC(isFsur)', 'LBC(isUbar)', 'LBC(isVbar)', 'LBC(isUvel)', 'LBC(isVvel)', 'LBC(isMtke)',    &
     &          'LBC(isTvar)')
            CALL load_lbc (TRIM(key), tok, nv, ierr)
            IF (ierr.ne.0) EXIT
          CASE ('Ngrids', 'NestLayers', 'GridsInLayer')
            IF (toint(tok(1)).ne.1) CALL unsupported (TRIM(key)//' > 1: nested grids are not built', ierr)
            IF (ierr.ne.0) EXIT
          CASE ('NBT', 'NST', 'NPT', 'NCS', 'NNS')
            IF (toint(tok(1)).ne.0) CALL unsupported (TRIM(key)//' > 0: only the active tracers are built', ierr)
            IF (ierr.ne.0) EXIT
          CASE ('NRREC');       nrrec=toint(tok(1))
          CASE ('LcycleRST');   LcycleRST=tok(1)(1:1).eq.'T'.or.tok(1)(1:1).eq.'t'
          CASE ('NRST');        nRST=toint(tok(1))
          CASE ('NHIS');        nHIS=toint(tok(1))
          CASE ('ININAME');     ininame=ADJUSTL(val)
          CASE ('RSTNAME');     rstname=ADJUSTL(val)
          CASE ('HISNAME');     hisname=ADJUSTL(val)
          CASE ('NAVG');        nAVG=toint(tok(1))
          CASE ('NTSAVG');      ntsAVG=toint(tok(1))
          CASE ('AVGNAME');     avgname=ADJUSTL(val)
          CASE ('NDIA');        nDIA=toint(tok(1))
          CASE ('NTSDIA');      ntsDIA=toint(tok(1))
          CASE ('DIANAME');     dianame=ADJUSTL(val)
          CASE ('Dout(iThadv)'); CALL load_dout (0, tok, nv)
          CASE ('Dout(iTxadv)'); CALL load_dout (1, tok, nv)
          CASE ('Dout(iTyadv)'); CALL load_dout (2, tok, nv)
          CASE ('Dout(iTvadv)'); CALL load_dout (3, tok, nv)
          CASE ('Dout(iThdif)'); CALL load_dout (4, tok, nv)
          CASE ('Dout(iTxdif)'); CALL load_dout (5, tok, nv)
          CASE ('Dout(iTydif)'); CALL load_dout (6, tok, nv)
          CASE ('Dout(iTsdif)'); CALL load_dout (7, tok, nv)
          CASE ('Dout(iTvdif)'); CALL load_dout (8, tok, nv)
          CASE ('Dout(iTrate)'); CALL load_dout (9, tok, nv)
          CASE ('Dout(M2fcor)'); DoutM2(0)=istrue(tok(1))
          CASE ('Dout(M2hadv)'); DoutM2(1)=istrue(tok(1))
          CASE ('Dout(M2xadv)'); DoutM2(2)=istrue(tok(1))
          CASE ('Dout(M2yadv)'); DoutM2(3)=istrue(tok(1))
          CASE ('Dout(M2hvis)'); DoutM2(4)=istrue(tok(1))
          CASE ('Dout(M2xvis)'); DoutM2(5)=istrue(tok(1))
          CASE ('Dout(M2yvis)'); DoutM2(6)=istrue(tok(1))
          CASE ('Dout(M2pgrd)'); DoutM2(7)=istrue(tok(1))
          CASE ('Dout(M2sstr)'); DoutM2(8)=istrue(tok(1))
          CASE ('Dout(M2bstr)'); DoutM2(9)=istrue(tok(1))
          CASE ('Dout(M2rate)'); DoutM2(10)=istrue(tok(1))
          CASE ('Dout(M3fcor)'); DoutM3(0)=istrue(tok(1))
          CASE ('Dout(M3vadv)'); DoutM3(1)=istrue(tok(1))
          CASE ('Dout(M3hadv)'); DoutM3(2)=istrue(tok(1))
          CASE ('Dout(M3xadv)'); DoutM3(3)=istrue(tok(1))
          CASE ('Dout(M3yadv)'); DoutM3(4)=istrue(tok(1))
          CASE ('Dout(M3pgrd)'); DoutM3(5)=istrue(tok(1))
          CASE ('Dout(M3vvis)'); DoutM3(6)=istrue(tok(1))
          CASE ('Dout(M3hvis)'); DoutM3(7)=istrue(tok(1))
          CASE ('Dout(M3xvis)'); DoutM3(8)=istrue(tok(1))
          CASE ('Dout(M3yvis)'); DoutM3(9)=istrue(tok(1))
          CASE ('Dout(M3rate)'); DoutM3(10)=istrue(tok(1))
          CASE ('Aout(idFsur)'); Aout(0)=istrue(tok(1))
          CASE ('Aout(idUbar)'); Aout(1)=istrue(tok(1))
          CASE ('Aout(idVbar)'); Aout(2)=istrue(tok(1))
          CASE ('Aout(idUvel)'); Aout(3)=istrue(tok(1))
          CASE ('Aout(idVvel)'); Aout(4)=istrue(tok(1))
          CASE ('Aout(idOvel)'); Aout(5)=istrue(tok(1))
          CASE ('Aout(idWvel)'); Aout(6)=istrue(tok(1))
          CASE ('Aout(idDano)'); Aout(7)=istrue(tok(1))
          CASE ('Aout(idTvar)'); CALL load_aout (8, tok, nv)
          CASE ('Aout(idZZav)'); Aout(9)=istrue(tok(1))
          CASE ('Aout(idU2av)'); Aout(10)=istrue(tok(1))
          CASE ('Aout(idV2av)'); Aout(11)=istrue(tok(1))
          CASE ('Aout(idUUav)'); Aout(12)=istrue(tok(1))
          CASE ('Aout(idVVav)'); Aout(13)=istrue(tok(1))
          CASE ('Aout(idUVav)'); Aout(14)=istrue(tok(1))
          CASE ('Aout(idHUav)'); Aout(15)=istrue(tok(1))
          CASE ('Aout(idHVav)'); Aout(16)=istrue(tok(1))
          CASE ('Aout(idTTav)'); CALL load_aout (17, tok, nv)
          CASE ('Aout(idUTav)'); CALL load_aout (18, tok, nv)
          CASE ('Aout(idVTav)'); CALL load_aout (19, tok, nv)
          CASE ('Aout(iHUTav)'); CALL load_aout (20, tok, nv)
          CASE ('Aout(iHVTav)'); CALL load_aout (21, tok, nv)
          CASE ('Hout(idFsur)'); Hout(idFsur)=istrue(tok(1))
          CASE ('Hout(idUbar)'); Hout(idUbar)=istrue(tok(1))
          CASE ('Hout(idVbar)'); Hout(idVbar)=istrue(tok(1))
          CASE ('Hout(idUvel)'); Hout(idUvel)=istrue(tok(1))
          CASE ('Hout(idVvel)'); Hout(idVvel)=istrue(tok(1))
          CASE ('Hout(idWvel)'); Hout(idWvel)=istrue(tok(1))
          CASE ('Hout(idOvel)'); Hout(idOvel)=istrue(tok(1))
          CASE ('Hout(idTvar)')
            Hout(idTemp)=istrue(tok(1))
            IF (nv.ge.2) Hout(idSalt)=istrue(tok(2))
          CASE ('Hout(idDano)'); Hout(idDano)=istrue(tok(1))
          CASE ('Hout(idVvis)'); Hout(idVvis)=istrue(tok(1))
          CASE ('Hout(idTdif)'); Hout(idTdif)=istrue(tok(1))
          CASE ('Hout(idSdif)'); Hout(idSdif)=istrue(tok(1))
          CASE ('Hout(idHsbl)'); Hout(idHsbl)=istrue(tok(1))
          CASE ('Hout(idMtke)'); HoutMtke=istrue(tok(1))
          CASE ('Hout(idMtls)'); HoutMtls=istrue(tok(1))
!  VolCons(west|south|east|north) (read_phypar.F: volume conservation across an open edge, obc_volcons.F): bits 0..3 in the order of
!  iwest, isouth, ieast, inorth
          CASE ('VolCons(west)');  IF (istrue(tok(1))) volcons=IOR(volcons, 1)
          CASE ('VolCons(south)'); IF (istrue(tok(1))) volcons=IOR(volcons, 2)
          CASE ('VolCons(east)');  IF (istrue(tok(1))) volcons=IOR(volcons, 4)
          CASE ('VolCons(north)'); IF (istrue(tok(1))) volcons=IOR(volcons, 8)
          CASE ('LuvSrc', 'LwSrc', 'LtracerSrc', 'LuvSponge', 'LtracerSponge', 'LsshCLM', 'Lm2CLM', 'Lm3CLM',     &
     &          'LtracerCLM', 'LnudgeM2CLM', 'LnudgeM3CLM', 'LnudgeTCLM')
            DO itr=1,nv
              IF (tok(itr)(1:1).eq.'T'.or.tok(itr)(1:1).eq.'t') THEN
                CALL unsupported (TRIM(key)//' == T: point sources, sponges and climatology input '//           &
     &                            'are not built', ierr)
              END IF
            END DO
            IF (ierr.ne.0) EXIT
          CASE ('TNU2');        CALL load_r (tok, nv, tnu2)
          CASE ('VISC2');       visc2=toreal(tok(1))
          CASE ('TNU4');        CALL load_r (tok, nv, tnu4)
          CASE ('VISC4');       visc4=toreal(tok(1))
          CASE ('DCRIT');       Dcrit=toreal(tok(1))
          CASE ('AKT_BAK');     CALL load_r (tok, nv, Akt_bak)
          CASE ('AKV_BAK');     Akv_bak=toreal(tok(1))
          CASE ('RDRG');        rdrg=toreal(tok(1))
          CASE ('RDRG2');       rdrg2=toreal(tok(1))
          CASE ('Zob');         Zob=toreal(tok(1))
          CASE ('Zos');         Zos=toreal(tok(1))
          CASE ('AKK_BAK');     Akk_bak=toreal(tok(1))
          CASE ('AKP_BAK');     Akp_bak=toreal(tok(1))
          CASE ('GLS_P');       gls_p=toreal(tok(1))
          CASE ('GLS_M');       gls_m=toreal(tok(1))
          CASE ('GLS_N');       gls_n=toreal(tok(1))
          CASE ('GLS_Kmin');    gls_Kmin=toreal(tok(1))
          CASE ('GLS_Pmin');    gls_Pmin=toreal(tok(1))
          CASE ('GLS_CMU0');    gls_cmu0=toreal(tok(1))
          CASE ('GLS_C1');      gls_c1=toreal(tok(1))
          CASE ('GLS_C2');      gls_c2=toreal(tok(1))
          CASE ('GLS_C3M');     gls_c3m=toreal(tok(1))
          CASE ('GLS_C3P');     gls_c3p=toreal(tok(1))
          CASE ('GLS_SIGK');    gls_sigk=toreal(tok(1))
          CASE ('GLS_SIGP');    gls_sigp=toreal(tok(1))
          CASE ('CHARNOK_ALPHA'); charnok_alpha=toreal(tok(1))
          CASE ('CRGBAN_CW');   crgban_cw=toreal(tok(1))
          CASE ('BLK_ZQ');      blk_ZQ=toreal(tok(1))
          CASE ('BLK_ZT');      blk_ZT=toreal(tok(1))
          CASE ('BLK_ZW');      blk_ZW=toreal(tok(1))
          CASE ('WTYPE');       lmd_Jwt=toint(tok(1))
          CASE ('Vtransform');  Vtransform=toint(tok(1))
          CASE ('Vstretching'); Vstretching=toint(tok(1))
          CASE ('THETA_S');     theta_s=toreal(tok(1))
          CASE ('THETA_B');     theta_b=toreal(tok(1))
          CASE ('TCLINE');      Tcline=toreal(tok(1))
          CASE ('RHO0');        rho0=toreal(tok(1))
          CASE ('DSTART');      dstart=toreal(tok(1))
          CASE ('TIME_REF');    time_ref=toreal(tok(1))
          CASE ('R0');          R0=toreal(tok(1))
          CASE ('T0');          T0=toreal(tok(1))
          CASE ('S0');          S0=toreal(tok(1))
          CASE ('TCOEF');       Tcoef=toreal(tok(1))
          CASE ('SCOEF');       Scoef=toreal(tok(1))
          CASE ('GAMMA2');      gamma2=toreal(tok(1))
          CASE ('ZNUDG');       Znudg=toreal(tok(1))
          CASE ('M2NUDG');      M2nudg=toreal(tok(1))
          CASE ('M3NUDG');      M3nudg=toreal(tok(1))
          CASE ('TNUDG');       CALL load_r (tok, nv, Tnudg)
          CASE ('OBCFAC');      obcfac=toreal(tok(1))
!  Every other keyword of the reference's reader (read_phypar.F has 612): classified keyword by keyword as unable to
!  change the forward time step of this build -- output selection, file names, the parameters of other drivers and of
!  cpp options the library does not carry (a header that defines one is stopped by options_from_defines).  The list
!  is generated (tools/roms_in_keywords.py -> docs/ROMS_IN_KEYWORDS.md, roms_in_inert.inc).
          INCLUDE 'roms_in_inert.inc'
          CASE DEFAULT
!  not a keyword of the reference's reader either (it skips such lines silently, read_phypar.F has no CASE DEFAULT).
!  Its own input files carry a few (Aout(idBath), Qout(idUVwc), Dout(M2fsgr) ..., ad_AKT_BAK): members of the output
!  selection and adjoint families, inert by family; anything else is counted as unknown.
            IF (key(1:5).eq.'Hout('.or.key(1:5).eq.'Qout('.or.key(1:5).eq.'Aout('.or.key(1:5).eq.'Dout('.or.        &
     &          key(1:3).eq.'ad_') THEN
              n_inert_keys=n_inert_keys+1
            ELSE
              n_unknown_keys=n_unknown_keys+1
              IF (LEN_TRIM(first_unknown_key).eq.0) first_unknown_key=key
            END IF
        END SELECT
      END DO
      n_unused_keys=n_inert_keys+n_unknown_keys
      close (iu)
      END SUBROUTINE read_roms_in
!
!  A recognised setting that selects code this build does not have: stop as checkdefs.F / inp_par.F
!  do for illegal configurations (exit_flag = 5) instead of running with different physics.
!
      SUBROUTINE unsupported (what, ierr)
      character(len=*), intent(in) :: what
      integer, intent(inout) :: ierr
      host_message=what
      ierr=5
      END SUBROUTINE unsupported
!
!  Lateral boundary conditions, load_lbc (Utility/inp_decode.F:1560-1680): four keywords per variable in the order west
!  south east north (isTvar: one set per tracer on continuation lines).  Built: Per Clo Gra Cla Rad RadNud, Che / Cha for
!  the free surface, Fla / Shc for the 2-D momentum (the library checks the pairing); every variable must share the
!  periodicity of the grid.  isMtke: the turbulent fields of GLS_MIXING (closed, gradient or periodic: tkebc_im.F).
!
      SUBROUTINE load_lbc (key, tok, nv, ierr)
      character(len=*), intent(in) :: key
      character(len=64), intent(in) :: tok(16)
      integer, intent(in) :: nv
      integer, intent(inout) :: ierr
      integer :: k, side, ivar, code
      logical :: per(4)
      character(len=64) :: val
      IF (nv.lt.4.or.MOD(nv,4).ne.0) THEN
        CALL unsupported (key//': expected west south east north', ierr)
        RETURN
      END IF
      SELECT CASE (key)
        CASE ('LBC(isFsur)'); ivar=1
        CASE ('LBC(isUbar)'); ivar=2
        CASE ('LBC(isVbar)'); ivar=3
        CASE ('LBC(isUvel)'); ivar=4
        CASE ('LBC(isVvel)'); ivar=5
        CASE ('LBC(isTvar)'); ivar=6
        CASE DEFAULT;         ivar=0
      END SELECT
      DO k=1,nv
        val=upper(tok(k))
        SELECT CASE (TRIM(val))
          CASE ('CLO');    code=ROMS_LBC_CLO
          CASE ('PER');    code=ROMS_LBC_PER
          CASE ('GRA');    code=ROMS_LBC_GRA
          CASE ('CLA');    code=ROMS_LBC_CLA
          CASE ('RAD');    code=ROMS_LBC_RAD
          CASE ('RADNUD'); code=ROMS_LBC_RADNUD
          CASE ('CHE');    code=ROMS_LBC_CHE
          CASE ('CHA');    code=ROMS_LBC_CHI
          CASE ('FLA');    code=ROMS_LBC_FLA
          CASE ('SHC');    code=ROMS_LBC_SHC
          CASE DEFAULT
            CALL unsupported (key//' = '//TRIM(tok(k))//': built are Per Clo Gra Cla Rad RadNud Che Cha Fla Shc '//   &
     &                        '(no Red, Nes, Mix)', ierr)
            RETURN
        END SELECT
        side=MOD(k-1,4)+1
        IF (k.le.4) THEN
          per(side)=code.eq.ROMS_LBC_PER
        ELSE IF (per(side).neqv.(code.eq.ROMS_LBC_PER)) THEN
          CALL unsupported (key//': tracers disagree on periodicity', ierr)
          RETURN
        END IF
        IF (ivar.gt.0.and.ivar+(k-1)/4.le.ROMS_NLBC) lbc(ivar+(k-1)/4,side)=code
        IF (key.eq.'LBC(isMtke)'.and.k.le.4) lbc_tke(side)=code
      END DO
      IF ((per(1).neqv.per(3)).or.(per(2).neqv.per(4))) THEN
        CALL unsupported (key//': a periodic edge needs its opposite edge periodic too', ierr)
        RETURN
      END IF
      IF (lbc_seen.eq.0) THEN
        EWperiodic=per(1)
        NSperiodic=per(2)
      ELSE IF ((per(1).neqv.EWperiodic).or.(per(2).neqv.NSperiodic)) THEN
        CALL unsupported (key//': periodicity differs from the other state variables', ierr)
        RETURN
      END IF
      lbc_seen=lbc_seen+1
      END SUBROUTINE load_lbc

      SUBROUTINE set_defaults ()          ! roms_upwelling.in values, so that a partial file is usable
      MyAppCPP='UPWELLING'
      Lm=41; Mm=80; N=16; NAT=2; NT=2; NtileI=1; NtileJ=1
      ntimes=100; ndtfast=30; ninfo=1; Vtransform=2; Vstretching=4; lmd_Jwt=1
      hadv=ROMS_U3; vadv=ROMS_C4
      EWperiodic=.TRUE.; NSperiodic=.FALSE.
      lbc=0; Znudg=0.0_dp; M2nudg=0.0_dp; M3nudg=0.0_dp; Tnudg=0.0_dp; obcfac=0.0_dp; volcons=0
      dt=300.0_dp; theta_s=3.0_dp; theta_b=0.0_dp; Tcline=25.0_dp; rho0=1025.0_dp
      R0=1027.0_dp; T0=14.0_dp; S0=35.0_dp; Tcoef=1.7E-4_dp; Scoef=0.0_dp
      visc2=5.0_dp; tnu2=0.0_dp; Akt_bak=1.0E-6_dp; Akv_bak=1.0E-5_dp
      visc4=0.0_dp; tnu4=0.0_dp; mix4=.FALSE.
      Dcrit=0.10_dp; wet_dry=.FALSE.; mix_geo_uv=.FALSE.
      rdrg=3.0E-4_dp; rdrg2=3.0E-3_dp; Zob=0.02_dp; Zos=0.02_dp; gamma2=1.0_dp
      dstart=0.0_dp; time_ref=0.0_dp; blk_ZQ=10.0_dp; blk_ZT=10.0_dp; blk_ZW=10.0_dp
      gls_flags=0; lbc_tke=0; gls_p=3.0_dp; gls_m=1.5_dp; gls_n=-1.0_dp; gls_Kmin=7.6E-6_dp; gls_Pmin=1.0E-12_dp
      gls_cmu0=0.5477_dp; gls_c1=1.44_dp; gls_c2=1.92_dp; gls_c3m=-0.4_dp; gls_c3p=1.0_dp; gls_sigk=1.0_dp; gls_sigp=1.30_dp
      Akk_bak=5.0E-6_dp; Akp_bak=5.0E-6_dp; charnok_alpha=1400.0_dp; crgban_cw=100.0_dp
      nrrec=0; nRST=0; nHIS=0; LcycleRST=.TRUE.; Hout=.FALSE.; HoutMtke=.FALSE.; HoutMtls=.FALSE.
      nAVG=0; ntsAVG=1; avgname='roms_avg.nc'; Aout=.FALSE.; AoutT=.FALSE.
      nDIA=0; ntsDIA=1; dianame='roms_dia.nc'; DoutT=.FALSE.; DoutM2=.FALSE.; DoutM3=.FALSE.; diag_uv=.FALSE.
      ininame='roms_ini.nc'; rstname='roms_rst.nc'; hisname='roms_his.nc'
      END SUBROUTINE set_defaults

      SUBROUTINE split (s, tok, n)
      character(len=*), intent(in) :: s
      character(len=64), intent(out) :: tok(16)
      integer, intent(out) :: n
      integer :: i, i0, L, ist, k, nrep
      character(len=64) :: w
      n=0
      L=LEN_TRIM(s)
      i=1
      DO WHILE (i.le.L)
        IF (s(i:i).eq.' '.or.s(i:i).eq.ACHAR(9)) THEN
          i=i+1
          CYCLE
        END IF
        i0=i
        DO WHILE (i.le.L)
          IF (s(i:i).eq.' '.or.s(i:i).eq.ACHAR(9)) EXIT
          i=i+1
        END DO
        w=s(i0:i-1)
        ist=INDEX(w,'*')
        IF (ist.gt.1) THEN            ! n*value
          read (w(1:ist-1),*) nrep
          DO k=1,nrep
            IF (n.lt.16) THEN
              n=n+1
              tok(n)=w(ist+1:)
            END IF
          END DO
        ELSE IF (n.lt.16) THEN
          n=n+1
          tok(n)=w
        END IF
      END DO
      END SUBROUTINE split

      SUBROUTINE load_dout (term, tok, nv)        ! Dout(iT...): one value per tracer
      integer, intent(in) :: term, nv
      character(len=64), intent(in) :: tok(16)
      integer :: it
      DO it=1,MIN(nv,ROMS_MAXT)
        DoutT(term,it)=istrue(tok(it))
      END DO
      END SUBROUTINE load_dout

      SUBROUTINE load_aout (bit, tok, nv)         ! a per-tracer switch line: one value per tracer
      integer, intent(in) :: bit, nv
      character(len=64), intent(in) :: tok(16)
      integer :: it
      DO it=1,MIN(nv,ROMS_MAXT)
        AoutT(bit,it)=istrue(tok(it))
      END DO
      Aout(bit)=ANY(AoutT(bit,:))
      END SUBROUTINE load_aout

      LOGICAL FUNCTION istrue (s)
      character(len=*), intent(in) :: s
      istrue=s(1:1).eq.'T'.or.s(1:1).eq.'t'
      END FUNCTION istrue

      INTEGER FUNCTION toint (s)
      character(len=*), intent(in) :: s
      read (s,*) toint
      END FUNCTION toint

      REAL(dp) FUNCTION toreal (s)
      character(len=*), intent(in) :: s
      read (s,*) toreal
      END FUNCTION toreal

      SUBROUTINE load_r (tok, nv, a)
      character(len=64), intent(in) :: tok(16)
      integer, intent(in) :: nv
      real(dp), intent(inout) :: a(:)
      integer :: k
      DO k=1,MIN(nv,SIZE(a))
        a(k)=toreal(tok(k))
      END DO
      END SUBROUTINE load_r

      FUNCTION upper (w) RESULT (u_)
      character(len=*), intent(in) :: w
      character(len=LEN(w)) :: u_
      integer :: k, c
      u_=w
      DO k=1,LEN_TRIM(w)
        c=IACHAR(w(k:k))
        IF (c.ge.IACHAR('a').and.c.le.IACHAR('z')) u_(k:k)=ACHAR(c-32)
      END DO
      END FUNCTION upper

      SUBROUTINE load_tadv (tok, nv, adv)       ! load_tadv, Utility/inp_decode.F
      character(len=64), intent(in) :: tok(16)
      integer, intent(in) :: nv
      integer, intent(inout) :: adv(:)
      integer :: k
      DO k=1,MIN(nv,SIZE(adv))
        SELECT CASE (TRIM(upper(tok(k))))            ! the spellings inp_decode.F:2557-2571 accepts
          CASE ('A4', 'AKIMA4');          adv(k)=ROMS_A4
          CASE ('C2', 'CENTERED2');       adv(k)=ROMS_C2
          CASE ('C4', 'CENTERED4');       adv(k)=ROMS_C4
          CASE ('HS', 'HSIMT');           adv(k)=ROMS_HSIMT
          CASE ('MP', 'MPDATA');          adv(k)=ROMS_MPDATA
          CASE ('SP', 'SPLINES');         adv(k)=ROMS_SPLINES
          CASE ('SU', 'SU3', 'SPLIT_U3'); adv(k)=ROMS_SPLIT_U3
          CASE ('U3', 'UPSTREAM3');       adv(k)=ROMS_U3
          CASE DEFAULT;                   adv(k)=-1          ! reported by host_setup
        END SELECT
      END DO
      END SUBROUTINE load_tadv
!
!=======================================================================
!  cpp options.  The reference fixes them at compile time in ROMS/Include/<application>.h (through
!  cppdefs.h); this build carries every option's code and selects at run time, so the SAME header
!  is read here: read_app_header understands the directives those files use (#define, #undef,
!  #ifdef, #ifndef, #if/#elif with defined, !, &&, ||, parentheses, #else, #endif, C comments,
!  "\" continuation) and options_from_defines turns the resulting set into the option mask of
!  include/roms_hip.h -- or stops with exit_flag 5 naming the option whose code is not built, the
!  way checkdefs.F stops on an illegal combination.  Without a header file the option lists of the
!  three BASELINE applications below are used.
!=======================================================================
!
      LOGICAL FUNCTION is_defined (name)
      character(len=*), intent(in) :: name
      integer :: k
      is_defined=.FALSE.
      DO k=1,ndefs
        IF (TRIM(defs(k)).eq.TRIM(name)) is_defined=.TRUE.
      END DO
      END FUNCTION is_defined

      SUBROUTINE define (name)
      character(len=*), intent(in) :: name
      IF (is_defined(name).or.LEN_TRIM(name).eq.0) RETURN
      IF (ndefs.lt.SIZE(defs)) THEN
        ndefs=ndefs+1
        defs(ndefs)=name
      END IF
      END SUBROUTINE define

      SUBROUTINE undefine (name)
      character(len=*), intent(in) :: name
      integer :: k, m
      m=0
      DO k=1,ndefs
        IF (TRIM(defs(k)).ne.TRIM(name)) THEN
          m=m+1
          defs(m)=defs(k)
        END IF
      END DO
      ndefs=m
      END SUBROUTINE undefine
!
!  #if expression:  or := and { "||" and } ;  and := unary { "&&" unary } ;
!                   unary := "!" unary | "(" or ")" | "defined" [ "(" ] NAME [ ")" ] | NAME | integer
!
      SUBROUTINE lex_condition (text)
      character(len=*), intent(in) :: text
      integer :: i, i0, L
      nxtok=0
      xpos=1
      L=LEN_TRIM(text)
      i=1
      DO WHILE (i.le.L.and.nxtok.lt.SIZE(xtok))
        SELECT CASE (text(i:i))
          CASE (' ', ACHAR(9))
            i=i+1
          CASE ('(', ')', '!')
            nxtok=nxtok+1
            xtok(nxtok)=text(i:i)
            i=i+1
          CASE ('|', '&')
            nxtok=nxtok+1
            xtok(nxtok)=text(i:i)//text(i:i)
            i=i+2
          CASE DEFAULT
            i0=i
            DO WHILE (i.le.L)
              IF (INDEX(' ()!|&'//ACHAR(9), text(i:i)).gt.0) EXIT
              i=i+1
            END DO
            nxtok=nxtok+1
            xtok(nxtok)=text(i0:i-1)
        END SELECT
      END DO
      END SUBROUTINE lex_condition

      RECURSIVE FUNCTION cond_or () RESULT (val)
      logical :: val, rhs
      val=cond_and()
      DO WHILE (xpos.le.nxtok)
        IF (xtok(xpos).ne.'||') EXIT
        xpos=xpos+1
        rhs=cond_and()
        val=val.or.rhs
      END DO
      END FUNCTION cond_or

      RECURSIVE FUNCTION cond_and () RESULT (val)
      logical :: val, rhs
      val=cond_unary()
      DO WHILE (xpos.le.nxtok)
        IF (xtok(xpos).ne.'&&') EXIT
        xpos=xpos+1
        rhs=cond_unary()
        val=val.and.rhs
      END DO
      END FUNCTION cond_and

      RECURSIVE FUNCTION cond_unary () RESULT (val)
      logical :: val, paren
      integer :: number, ios
      val=.FALSE.
      IF (xpos.gt.nxtok) RETURN
      IF (xtok(xpos).eq.'!') THEN
        xpos=xpos+1
        val=.not.cond_unary()
      ELSE IF (xtok(xpos).eq.'(') THEN
        xpos=xpos+1
        val=cond_or()
        IF (xpos.le.nxtok) xpos=xpos+1            ! ")"
      ELSE IF (xtok(xpos).eq.'defined') THEN
        xpos=xpos+1
        paren=.FALSE.
        IF (xpos.le.nxtok) paren=xtok(xpos).eq.'('
        IF (paren) xpos=xpos+1
        IF (xpos.le.nxtok) val=is_defined(xtok(xpos))
        xpos=xpos+1
        IF (paren) xpos=xpos+1
      ELSE
        read (xtok(xpos),*,iostat=ios) number      ! "#if 0"
        IF (ios.eq.0) THEN
          val=number.ne.0
        ELSE
          val=is_defined(xtok(xpos))                ! a bare macro name counts as defined-and-non-zero
        END IF
        xpos=xpos+1
      END IF
      END FUNCTION cond_unary

      SUBROUTINE read_app_header (fname, ierr)
      character(len=*), intent(in) :: fname
      integer, intent(inout) :: ierr
      integer, parameter :: maxdepth = 32
      logical :: live(0:maxdepth), done(0:maxdepth), in_comment, more
      integer :: iu, ios, depth, i, L
      character(len=1024) :: raw, code, stmt
      character(len=64) :: word, name
      open (newunit=iu, file=TRIM(fname), status='old', action='read', iostat=ios)
      IF (ios.ne.0) THEN
        host_message='cannot open the application header '//TRIM(fname)
        ierr=2
        RETURN
      END IF
      depth=0
      live(0)=.TRUE.
      done(0)=.TRUE.
      in_comment=.FALSE.
      stmt=' '
      DO
        read (iu,'(a)',iostat=ios) raw
        IF (ios.ne.0) EXIT
        code=' '                                   ! the line without its C comments
        L=LEN_TRIM(raw)
        i=1
        DO WHILE (i.le.L)
          IF (in_comment) THEN
            IF (raw(i:MIN(i+1,L)).eq.'*/') THEN
              in_comment=.FALSE.
              i=i+2
            ELSE
              i=i+1
            END IF
          ELSE IF (raw(i:MIN(i+1,L)).eq.'/*') THEN
            in_comment=.TRUE.
            i=i+2
          ELSE
            code(i:i)=raw(i:i)
            i=i+1
          END IF
        END DO
        more=.FALSE.
        L=LEN_TRIM(code)
        IF (L.gt.0) more=code(L:L).eq.ACHAR(92)              ! a backslash (one character: no escape processing)
        IF (more) code(L:L)=' '
        stmt=TRIM(stmt)//' '//TRIM(ADJUSTL(code))
        IF (more) CYCLE
        stmt=ADJUSTL(stmt)
        IF (stmt(1:1).eq.'#') THEN
          stmt=ADJUSTL(stmt(2:))
          i=SCAN(stmt,' ')
          word=stmt(1:i-1)
          stmt=ADJUSTL(stmt(i:))
          i=SCAN(stmt,' ')
          name=stmt(1:MAX(i-1,1))
          SELECT CASE (TRIM(word))
            CASE ('define')
              IF (live(depth)) CALL define (TRIM(name))
            CASE ('undef')
              IF (live(depth)) CALL undefine (TRIM(name))
            CASE ('ifdef', 'ifndef', 'if')
              IF (depth.ge.maxdepth) THEN
                CALL unsupported (TRIM(fname)//': conditionals nested too deep', ierr)
                EXIT
              END IF
              depth=depth+1
              IF (TRIM(word).eq.'ifdef') THEN
                live(depth)=is_defined(TRIM(name))
              ELSE IF (TRIM(word).eq.'ifndef') THEN
                live(depth)=.not.is_defined(TRIM(name))
              ELSE
                CALL lex_condition (stmt)
                live(depth)=cond_or()
              END IF
              live(depth)=live(depth).and.live(depth-1)
              done(depth)=live(depth).or..not.live(depth-1)
            CASE ('elif')
              IF (depth.gt.0) THEN
                IF (done(depth)) THEN
                  live(depth)=.FALSE.
                ELSE
                  CALL lex_condition (stmt)
                  live(depth)=cond_or()
                  done(depth)=live(depth)
                END IF
              END IF
            CASE ('else')
              IF (depth.gt.0) THEN
                live(depth)=.not.done(depth)
                done(depth)=.TRUE.
              END IF
            CASE ('endif')
              depth=MAX(depth-1,0)
          END SELECT
        END IF
        stmt=' '
      END DO
      close (iu)
      END SUBROUTINE read_app_header
!
!  The option lists of the BASELINE applications (what ROMS/Include/upwelling.h and benchmark.h
!  define for a forward run without output options; UPWELLING_KPP = BASELINE config 5's custom
!  header: UPWELLING with the LMD/KPP closure in place of the analytic mixing).
!
      SUBROUTINE builtin_defines (ierr)
      integer, intent(inout) :: ierr
      integer :: k, nl
      logical :: with_ddmix, with_bkpp
      character(len=16), parameter :: common(9) = [ character(len=16) :: 'SOLVE3D', 'SALINITY', 'UV_ADV',        &
     &    'UV_COR', 'UV_VIS2', 'MIX_S_UV', 'TS_DIF2', 'DJ_GRADPS', 'ANA_GRID' ]
      character(len=16), parameter :: kpp(7) = [ character(len=16) :: 'LMD_MIXING', 'LMD_RIMIX', 'LMD_CONVEC',    &
     &    'LMD_SKPP', 'LMD_NONLOCAL', 'RI_SPLINES', 'SOLAR_SOURCE' ]
      character(len=16), parameter :: flux0(5) = [ character(len=16) :: 'ANA_SMFLUX', 'ANA_STFLUX', 'ANA_SSFLUX', &
     &    'ANA_BTFLUX', 'ANA_BSFLUX' ]
      character(len=16), parameter :: bulk(9) = [ character(len=16) :: 'BULK_FLUXES', 'LONGWAVE', 'ANA_WINDS',    &
     &    'ANA_TAIR', 'ANA_PAIR', 'ANA_HUMIDITY', 'ANA_RAIN', 'ANA_CLOUD', 'ALBEDO' ]
!  (<application>_DDMIX = the same application with LMD_DDMIX: oracle/ref/upwelling_kpp_ddmix.h, benchmark.h -DLMD_DDMIX)
!  (<application>_BKPP = the same application with LMD_BKPP: benchmark.h / oracle/ref/upwelling_kpp.h -DLMD_BKPP)
      nl=LEN_TRIM(MyAppCPP)
      with_bkpp=.FALSE.
      IF (nl.gt.5) THEN
        IF (MyAppCPP(nl-4:nl).eq.'_BKPP') THEN
          with_bkpp=.TRUE.
          MyAppCPP=MyAppCPP(1:nl-5)
        END IF
      END IF
      nl=LEN_TRIM(MyAppCPP)
      with_ddmix=.FALSE.
      IF (nl.gt.6) THEN
        IF (MyAppCPP(nl-5:nl).eq.'_DDMIX') THEN
          with_ddmix=.TRUE.
          MyAppCPP=MyAppCPP(1:nl-6)
        END IF
      END IF
      DO k=1,SIZE(common)
!  (UPWELLING_BIH = oracle/ref/upwelling_bih.h: biharmonic mixing along s-surfaces in place of the harmonic operators)
!  (UPWELLING_BIHGEO = oracle/ref/upwelling_bihgeo.h: ... the tracers along geopotentials; _BIHISO: along isopycnals; UPWELLING_GEOUV =
!  oracle/ref/upwelling_geouv.h: MASKING and the harmonic viscosity along geopotentials, MIX_GEO_UV)
!  (UPWELLING_BIHGEOUV = oracle/ref/upwelling_bihgeouv.h: the BIHARMONIC viscosity along geopotentials, uv3dmix4_geo.h, under MASKING;
!  the tracers keep their harmonic operator)
        IF (TRIM(MyAppCPP).eq.'UPWELLING_BIHGEOUV') THEN
          IF (TRIM(common(k)).eq.'UV_VIS2'.or.TRIM(common(k)).eq.'MIX_S_UV') CYCLE
        ELSE IF (MyAppCPP(1:13).eq.'UPWELLING_BIH'.and.(TRIM(common(k)).eq.'UV_VIS2'.or.TRIM(common(k)).eq.'TS_DIF2')) THEN
          CYCLE
        END IF
        IF (TRIM(MyAppCPP).eq.'UPWELLING_GEOUV'.and.TRIM(common(k)).eq.'MIX_S_UV') CYCLE
!  (UPWELLING_PRS40 / _PRS42 / _PRS44 = oracle/ref/upwelling_prs40.h ...: PJ_GRADP / PJ_GRADPQ2 / PJ_GRADPQ4 in DJ_GRADPS' place)
        IF (MyAppCPP(1:14).eq.'UPWELLING_PRS4'.and.TRIM(common(k)).eq.'DJ_GRADPS') CYCLE
        CALL define (TRIM(common(k)))
      END DO
      IF (TRIM(MyAppCPP).eq.'UPWELLING_PRS40') CALL define ('PJ_GRADP')
      IF (TRIM(MyAppCPP).eq.'UPWELLING_PRS42') CALL define ('PJ_GRADPQ2')
      IF (TRIM(MyAppCPP).eq.'UPWELLING_PRS44') CALL define ('PJ_GRADPQ4')
      IF (TRIM(MyAppCPP).eq.'UPWELLING_BIHGEOUV') THEN
        CALL define ('UV_VIS4'); CALL define ('MIX_GEO_UV')
      ELSE IF (MyAppCPP(1:13).eq.'UPWELLING_BIH') THEN
        CALL define ('UV_VIS4'); CALL define ('TS_DIF4')
      END IF
      IF (TRIM(MyAppCPP).eq.'UPWELLING_GEOUV') CALL define ('MIX_GEO_UV')
      IF (TRIM(MyAppCPP).eq.'SEAMOUNT'.or.TRIM(MyAppCPP).eq.'GRAV_ADJ') THEN
!  ROMS/Include/seamount.h, grav_adj.h (their output options AVERAGES / DIAGNOSTICS_* / ANA_DIAG select no time-stepping code)
        ndefs=0
        CALL define ('UV_ADV'); CALL define ('UV_VIS2'); CALL define ('MIX_S_UV'); CALL define ('DJ_GRADPS')
        CALL define ('SPLINES_VDIFF'); CALL define ('SPLINES_VVISC'); CALL define ('TS_DIF2'); CALL define ('SOLVE3D')
        CALL define ('ANA_GRID'); CALL define ('ANA_INITIAL'); CALL define ('ANA_SMFLUX'); CALL define ('ANA_STFLUX')
        CALL define ('ANA_BTFLUX')
        IF (TRIM(MyAppCPP).eq.'SEAMOUNT') THEN
          CALL define ('UV_COR'); CALL define ('UV_QDRAG'); CALL define ('MIX_GEO_TS')
        ELSE
          CALL define ('UV_LDRAG'); CALL define ('MIX_S_TS'); CALL define ('OUT_DOUBLE')
        END IF
        RETURN
      END IF
      IF (TRIM(MyAppCPP).eq.'OVERFLOW') THEN
!  ROMS/Include/overflow.h (its output option AVERAGES selects no time-stepping code)
        ndefs=0
        CALL define ('UV_ADV'); CALL define ('UV_COR'); CALL define ('UV_QDRAG'); CALL define ('UV_VIS2')
        CALL define ('MIX_S_UV'); CALL define ('DJ_GRADPS'); CALL define ('SPLINES_VDIFF'); CALL define ('SPLINES_VVISC')
        CALL define ('TS_DIF2'); CALL define ('MIX_ISO_TS'); CALL define ('SOLVE3D'); CALL define ('ANA_GRID')
        CALL define ('ANA_INITIAL'); CALL define ('ANA_SMFLUX'); CALL define ('ANA_STFLUX'); CALL define ('ANA_BTFLUX')
        RETURN
      END IF
      IF (TRIM(MyAppCPP).eq.'KELVIN'.or.TRIM(MyAppCPP).eq.'KELVIN_SPLINES') THEN
!  ROMS/Include/kelvin.h (plain tridiagonal vertical solvers); KELVIN_SPLINES = oracle/ref/kelvin_splines.h: the same with
!  the spline vertical solvers of UPWELLING and BENCHMARK
        ndefs=0
        CALL define ('UV_ADV'); CALL define ('UV_COR'); CALL define ('UV_QDRAG'); CALL define ('UV_VIS2')
        CALL define ('MIX_S_UV'); CALL define ('DJ_GRADPS'); CALL define ('TS_DIF2'); CALL define ('MIX_S_TS')
        CALL define ('SOLVE3D'); CALL define ('RADIATION_2D'); CALL define ('ANA_GRID'); CALL define ('ANA_INITIAL')
        CALL define ('ANA_FSOBC'); CALL define ('ANA_M2OBC'); CALL define ('ANA_SMFLUX'); CALL define ('ANA_STFLUX')
        CALL define ('ANA_SRFLUX'); CALL define ('ANA_BTFLUX')
        IF (TRIM(MyAppCPP).eq.'KELVIN_SPLINES') THEN
          CALL define ('SPLINES_VDIFF'); CALL define ('SPLINES_VVISC')
        END IF
        RETURN
      END IF
      CALL define ('ANA_INITIAL'); CALL define ('SPLINES_VDIFF'); CALL define ('SPLINES_VVISC')
      SELECT CASE (TRIM(MyAppCPP))
        CASE ('UPWELLING', 'UPWELLING_KPP', 'UPWELLING_LOGDRAG', 'UPWELLING_MASK', 'UPWELLING_BIH', 'UPWELLING_WETDRY',  &  ! (oracle/ref/upwelling_logdrag.h, _mask.h, _bih.h, _wetdry.h)
     &        'UPWELLING_BIHGEO', 'UPWELLING_BIHISO', 'UPWELLING_GEOUV', 'UPWELLING_BIHGEOUV', 'UPWELLING_PRS40', 'UPWELLING_PRS42',          &
     &        'UPWELLING_PRS44', 'UPWELLING_GLS', 'UPWELLING_GLS_CA', 'UPWELLING_GLS_CB', 'UPWELLING_GLS_GAL', 'UPWELLING_MY25',         &
     &        'UPWELLING_MY25_GAL')
!  UPWELLING_GLS = upwelling.h built with -DGLS_MIXING; _CA, _CB, _GAL = oracle/ref/upwelling_gls_ca.h, _cb.h, _gal.h: the
!  other compile-time forms of the closure the library is pinned in
          IF (MyAppCPP(1:13).eq.'UPWELLING_GLS') CALL define ('GLS_MIXING')
          IF (MyAppCPP(1:14).eq.'UPWELLING_MY25') CALL define ('MY25_MIXING')     ! (upwelling.h -DMY25_MIXING; _GAL: oracle/ref/upwelling_my25_gal.h)
          IF (TRIM(MyAppCPP).eq.'UPWELLING_MY25_GAL') CALL define ('K_C4ADVECTION')
          IF (TRIM(MyAppCPP).eq.'UPWELLING_GLS_CA') THEN
            CALL define ('MASKING'); CALL define ('CANUTO_A'); CALL define ('N2S2_HORAVG'); CALL define ('RI_SPLINES')
          ELSE IF (TRIM(MyAppCPP).eq.'UPWELLING_GLS_CB') THEN
            CALL define ('CANUTO_B'); CALL define ('K_C2ADVECTION'); CALL define ('CHARNOK'); CALL define ('CRAIG_BANNER')
          ELSE IF (TRIM(MyAppCPP).eq.'UPWELLING_GLS_GAL') THEN
            CALL define ('K_C4ADVECTION'); CALL define ('RI_SPLINES')
          END IF
          IF (TRIM(MyAppCPP).eq.'UPWELLING_MASK'.or.TRIM(MyAppCPP).eq.'UPWELLING_GEOUV'.or.                               &
     &        TRIM(MyAppCPP).eq.'UPWELLING_BIHGEOUV') CALL define ('MASKING')
          IF (TRIM(MyAppCPP).eq.'UPWELLING_WETDRY') THEN
            CALL define ('MASKING'); CALL define ('WET_DRY')
          END IF
          CALL define (TRIM(MERGE('UV_LOGDRAG', 'UV_LDRAG  ', TRIM(MyAppCPP).eq.'UPWELLING_LOGDRAG')))
          IF (TRIM(MyAppCPP).eq.'UPWELLING_BIHISO') THEN
            CALL define ('MIX_ISO_TS')                                   ! (oracle/ref/upwelling_bihiso.h: t3dmix4_iso.h)
          ELSE
            CALL define (TRIM(MERGE('MIX_GEO_TS', 'MIX_S_TS  ', TRIM(MyAppCPP).eq.'UPWELLING_BIHGEO')))
          END IF
          DO k=1,SIZE(flux0)
            CALL define (TRIM(flux0(k)))
          END DO
          IF (is_defined('GLS_MIXING').or.is_defined('MY25_MIXING')) THEN   ! upwelling.h:57-63 (-DGLS_MIXING | -DMY25_MIXING: ROMS_CPP_FLAGS)
            IF (TRIM(MyAppCPP).eq.'UPWELLING'.or.TRIM(MyAppCPP).eq.'UPWELLING_GLS'.or.TRIM(MyAppCPP).eq.'UPWELLING_MY25') THEN
              CALL define ('KANTHA_CLAYSON'); CALL define ('N2S2_HORAVG'); CALL define ('RI_SPLINES')
            END IF
          ELSE IF (TRIM(MyAppCPP).ne.'UPWELLING_KPP') THEN
            CALL define ('ANA_VMIX')
          ELSE
            DO k=1,SIZE(kpp)
              CALL define (TRIM(kpp(k)))
            END DO
            CALL define ('ANA_SRFLUX')
          END IF
        CASE ('BENCHMARK', 'BENCHMARK_MASK')                          ! (_MASK: oracle/ref/benchmark_mask.h)
          IF (TRIM(MyAppCPP).eq.'BENCHMARK_MASK') CALL define ('MASKING')
          CALL define ('UV_QDRAG'); CALL define ('MIX_GEO_TS'); CALL define ('NONLIN_EOS')
          CALL define ('CURVGRID'); CALL define ('SPHERICAL'); CALL define ('ANA_SRFLUX')
          CALL define ('ANA_SSFLUX'); CALL define ('ANA_BSFLUX'); CALL define ('ANA_BTFLUX')
          DO k=1,SIZE(kpp)
            CALL define (TRIM(kpp(k)))
          END DO
          DO k=1,SIZE(bulk)
            CALL define (TRIM(bulk(k)))
          END DO
        CASE DEFAULT
          CALL unsupported ('MyAppCPP = '//TRIM(MyAppCPP)//': no built-in option list; give the '//             &
     &                      'application header (ROMS_APP_HEADER / second argument of romsM)', ierr)
      END SELECT
      IF (with_ddmix) CALL define ('LMD_DDMIX')
      IF (with_bkpp) CALL define ('LMD_BKPP')
      END SUBROUTINE builtin_defines
!
!  defined options -> option mask, or exit_flag 5 with the reason.
!
      SUBROUTINE options_from_defines (ierr)
      integer, intent(inout) :: ierr
      integer :: k
      logical :: upw, bench, kelv, seam, grav, ovf
!  options with a bit in the mask (include/roms_hip.h)
      character(len=16), parameter :: bitname(20) = [ character(len=16) :: 'UV_ADV', 'UV_COR', 'UV_VIS2',       &
     &    'TS_DIF2', 'MIX_GEO_TS', 'CURVGRID', 'NONLIN_EOS', 'UV_QDRAG', 'LMD_MIXING', 'BULK_FLUXES',           &
     &    'SOLAR_SOURCE', 'ANA_VMIX', 'SALINITY', 'SPHERICAL', 'UV_LOGDRAG', 'MASKING', 'RADIATION_2D', 'GLS_MIXING', 'MY25_MIXING', 'MIX_ISO_TS' ]
      integer, parameter :: bitval(20) = [ ROMS_UV_ADV, ROMS_UV_COR, ROMS_UV_VIS2, ROMS_TS_DIF2, ROMS_MIX_GEO_TS, &
     &    ROMS_CURVGRID, ROMS_NONLIN_EOS, ROMS_UV_QDRAG, ROMS_LMD_MIXING, ROMS_BULK_FLUXES, ROMS_SOLAR_SOURCE,    &
     &    ROMS_ANA_VMIX, ROMS_SALINITY, ROMS_SPHERICAL, ROMS_UV_LOGDRAG, ROMS_MASKING, ROMS_RADIATION_2D, ROMS_GLS_MIXING, ROMS_MY25_MIXING, ROMS_MIX_ISO_TS ]
!  options whose code is the only form built (accepted, nothing to select) or that only affect output
      character(len=16), parameter :: inherent(39) = [ character(len=16) :: 'LMD_BKPP', 'ANA_FSOBC', 'ANA_M2OBC', 'WJ_GRADP', 'PJ_GRADP', &
     &    'PJ_GRADPQ2', 'PJ_GRADPQ4', 'LMD_DDMIX', &
     &    'SOLVE3D', 'ANA_GRID', 'ANA_INITIAL', &
     &    'DJ_GRADPS', 'MIX_S_UV', 'MIX_S_TS', 'SPLINES_VDIFF', 'SPLINES_VVISC', 'UV_LDRAG', 'ANA_SMFLUX',        &
     &    'ANA_STFLUX', 'ANA_SSFLUX', 'ANA_BTFLUX', 'ANA_BSFLUX', 'ANA_SRFLUX', 'LMD_RIMIX', 'LMD_CONVEC',        &
     &    'LMD_SKPP', 'LMD_NONLOCAL', 'RI_SPLINES', 'LONGWAVE', 'ANA_WINDS', 'ANA_TAIR', 'ANA_PAIR',             &
     &    'ANA_HUMIDITY', 'ANA_RAIN', 'ANA_CLOUD', 'ALBEDO', 'OUT_DOUBLE', 'PERFECT_RESTART', 'NONLINEAR' ]
      character(len=16), parameter :: output_only(3) = [ character(len=16) :: 'AVERAGES', 'DIAGNOSTICS_TS',      &
     &    'DIAGNOSTICS_UV' ]
!  the compile-time forms of GLS_MIXING (gls_prestep.F, gls_corstep.F): flags of roms_hip_config%gls_flags
      character(len=16), parameter :: glsname(9) = [ character(len=16) :: 'CANUTO_A', 'CANUTO_B', 'KANTHA_CLAYSON',  &
     &    'N2S2_HORAVG', 'RI_SPLINES', 'K_C2ADVECTION', 'K_C4ADVECTION', 'CHARNOK', 'CRAIG_BANNER' ]
      integer, parameter :: glsval(9) = [ ROMS_GLS_CANUTO_A, ROMS_GLS_CANUTO_B, ROMS_GLS_KANTHA_CLAYSON,              &
     &    ROMS_GLS_N2S2_HORAVG, ROMS_GLS_RI_SPLINES, ROMS_GLS_K_C2ADVECTION, ROMS_GLS_K_C4ADVECTION,                  &
     &    ROMS_GLS_CHARNOK, ROMS_GLS_CRAIG_BANNER ]
      IF (ierr.ne.0) RETURN
      upw=TRIM(MyAppCPP).eq.'UPWELLING'.or.TRIM(MyAppCPP).eq.'UPWELLING_KPP'.or.                                &
     &    TRIM(MyAppCPP).eq.'UPWELLING_LOGDRAG'.or.TRIM(MyAppCPP).eq.'UPWELLING_MASK'.or.is_defined('UPWELLING').or.    &
     &    MyAppCPP(1:13).eq.'UPWELLING_BIH'.or.MyAppCPP(1:16).eq.'UPWELLING_WETDRY'.or.                             &
     &    TRIM(MyAppCPP).eq.'UPWELLING_GEOUV'.or.MyAppCPP(1:14).eq.'UPWELLING_PRS4'.or.                                &
     &    MyAppCPP(1:13).eq.'UPWELLING_GLS'.or.MyAppCPP(1:14).eq.'UPWELLING_MY25'
      bench=TRIM(MyAppCPP).eq.'BENCHMARK'.or.TRIM(MyAppCPP).eq.'BENCHMARK_MASK'.or.is_defined('BENCHMARK')
      kelv=TRIM(MyAppCPP).eq.'KELVIN'.or.TRIM(MyAppCPP).eq.'KELVIN_SPLINES'.or.is_defined('KELVIN')
      seam=TRIM(MyAppCPP).eq.'SEAMOUNT'.or.is_defined('SEAMOUNT')
      grav=TRIM(MyAppCPP).eq.'GRAV_ADJ'.or.is_defined('GRAV_ADJ')
      ovf=TRIM(MyAppCPP).eq.'OVERFLOW'.or.is_defined('OVERFLOW')
      IF (COUNT((/ upw, bench, kelv, seam, grav, ovf /)).ne.1) THEN
        CALL unsupported ('MyAppCPP = '//TRIM(MyAppCPP)//': the analytic grid, initial state and forcing '//   &
     &                    'exist for UPWELLING, BENCHMARK, KELVIN, SEAMOUNT, GRAV_ADJ and OVERFLOW', ierr)
        RETURN
      END IF
!  WET_DRY switches MASKING on (globaldefs.h:152-154) and, inside set_vbc, LIMIT_BSTRESS (:160-162)
      wet_dry=is_defined('WET_DRY')
      IF (wet_dry.and..not.is_defined('MASKING')) CALL define ('MASKING')
      gls_flags=0
      IF (upw) options=ROMS_APP_UPWELLING
      IF (bench) options=ROMS_APP_BENCHMARK
      IF (kelv) options=ROMS_APP_KELVIN
      IF (seam) options=ROMS_APP_SEAMOUNT
      IF (grav) options=ROMS_APP_GRAV_ADJ
      IF (ovf) options=ROMS_APP_OVERFLOW
      DO k=1,ndefs
        IF (ANY(bitname.eq.defs(k))) THEN
          options=IOR(options, bitval(FINDLOC(bitname, defs(k), 1)))
        ELSE IF (ANY(glsname.eq.defs(k)).and.(is_defined('GLS_MIXING').or.is_defined('MY25_MIXING').or.          &
     &           TRIM(defs(k)).eq.'RI_SPLINES')) THEN
          IF (is_defined('GLS_MIXING').or.is_defined('MY25_MIXING'))                                            &
     &      gls_flags=IOR(gls_flags, glsval(FINDLOC(glsname, defs(k), 1)))
        ELSE IF (ANY(inherent.eq.defs(k)).or.ANY(output_only.eq.defs(k))) THEN
          CONTINUE
        ELSE IF (TRIM(defs(k)).eq.'UPWELLING'.or.TRIM(defs(k)).eq.'BENCHMARK'.or.TRIM(defs(k)).eq.'KELVIN'.or.            &
     &           TRIM(defs(k)).eq.'SEAMOUNT'.or.TRIM(defs(k)).eq.'GRAV_ADJ'.or.TRIM(defs(k)).eq.'OVERFLOW') THEN
          CONTINUE
        ELSE IF (TRIM(defs(k)).eq.'ANA_DIAG') THEN        ! the user diagnostics hook (ana_diag.h): output of its own, not built
          CONTINUE
        ELSE IF (TRIM(defs(k)).eq.'UV_VIS4'.or.TRIM(defs(k)).eq.'TS_DIF4') THEN   ! biharmonic mixing: ROMS_UV_VIS4 / ROMS_TS_DIF4 of cfg%options (below)
          CONTINUE
        ELSE IF (TRIM(defs(k)).eq.'MIX_GEO_UV') THEN      ! viscosity along geopotentials: ROMS_MIX_GEO_UV of cfg%options (below)
          CONTINUE
        ELSE IF (TRIM(defs(k)).eq.'WET_DRY'.or.TRIM(defs(k)).eq.'LIMIT_BSTRESS') THEN   ! wetting and drying: ROMS_WET_DRY of cfg%options
          IF (.not.wet_dry) CALL unsupported ('LIMIT_BSTRESS is built as part of WET_DRY only (globaldefs.h:160)', ierr)
        ELSE
          CALL unsupported ('cpp option '//TRIM(defs(k))//' is not built into this library', ierr)
          RETURN
        END IF
      END DO
!  what the kernels assume (the combinations the BASELINE headers use)
      IF (.not.is_defined('SOLVE3D')) CALL unsupported ('SOLVE3D is required (3-D baroclinic step)', ierr)
!  the pressure-gradient scheme (prsgrd.F:16-26): DJ_GRADPS -> prsgrd32.h; none of the four options -> prsgrd31.h, WJ_GRADP its
!  weighted form
!  (prsgrd.F tests PJ_GRADPQ4, PJ_GRADPQ2, PJ_GRADP, DJ_GRADPS in this order)
      ddmix=is_defined('LMD_DDMIX')               ! (upper word of cfg%options: ROMS_LMD_DDMIX, below)
      bkpp=is_defined('LMD_BKPP')
      IF (bkpp.and..not.(is_defined('LMD_MIXING').and.is_defined('SALINITY'))) CALL unsupported ('LMD_BKPP needs LMD_MIXING and SALINITY', ierr)
      IF (bkpp.and.(is_defined('MASKING').or.is_defined('WET_DRY').or.ddmix)) CALL unsupported ('LMD_BKPP together with MASKING, '//  &
     &    'WET_DRY or LMD_DDMIX: not pinned against the reference, not built', ierr)
      IF (ddmix.and..not.is_defined('LMD_MIXING')) CALL unsupported ('LMD_DDMIX without LMD_MIXING', ierr)
      IF (ddmix.and..not.is_defined('SALINITY')) CALL unsupported ('LMD_DDMIX needs SALINITY', ierr)
      prs4x=0
      IF (volcons.ne.0.and.NtileI*NtileJ.ne.1) CALL unsupported ('VolCons on more than one tile: obc_flux_tile sums over the '//      &
     &    'tiles (mp_reduce), a reduction across ranks that is not built', ierr)
      IF (is_defined('PJ_GRADPQ4')) THEN
        prs4x=44                                  ! (upper word of cfg%options: ROMS_PRSGRD44, below)
      ELSE IF (is_defined('PJ_GRADPQ2')) THEN
        prs4x=42
        IF (NtileI*NtileJ.ne.1) CALL unsupported ('PJ_GRADPQ2 (prsgrd42.h) on more than one tile: its second pass reads '//  &
     &    'rv(Iend+1,j,k), which no tile computes (prsgrd42.h:449) -- a single tile, or PJ_GRADPQ4', ierr)
      ELSE IF (is_defined('PJ_GRADP')) THEN
        options=IOR(options, ROMS_PRSGRD40)
      ELSE IF (.not.is_defined('DJ_GRADPS')) THEN
        options=IOR(options, ROMS_PRSGRD31)
        IF (is_defined('WJ_GRADP')) options=IOR(options, ROMS_WJ_GRADP)
      END IF
      IF (COUNT((/ is_defined('UV_LDRAG'), is_defined('UV_QDRAG'), is_defined('UV_LOGDRAG') /)).ne.1)           &
     &  CALL unsupported ('exactly one of UV_LDRAG, UV_QDRAG, UV_LOGDRAG is required', ierr)
!  UV_VIS4 / TS_DIF4 (round 4): the biharmonic operators along s-surfaces IN PLACE of the harmonic ones -- the library keeps
!  its harmonic kernels with zero coefficients (they add exact zeros) and runs uv3dmix4_s.h / t3dmix4_s.h / the UV_VIS4
!  block of step2d behind the option bits ROMS_UV_VIS4 / ROMS_TS_DIF4
      mix4(1)=is_defined('UV_VIS4')
      mix4(2)=is_defined('TS_DIF4')
!  WET_DRY (round 4): wetdry.F and its branches in the barotropic step, prsgrd32, rhs3d, the harmonic mixing along s-surfaces,
!  step3d_uv, set_vbc with LIMIT_BSTRESS, the closed-boundary routines (option bit ROMS_WET_DRY); round 5: bulk_flux, the
!  solar source of pre_step3d, t3dmix2_geo, mpdata_adiff.  The combinations whose WET_DRY statements the library does not
!  carry stop here
!  (round 6: the closures GLS_MIXING / MY25_MIXING, MIX_GEO_UV and the Jacobians prsgrd31 / 40 / 44 are pinned under WET_DRY:
!  oracle/ref/upwelling_wetdry_*.h)
      IF (wet_dry.and.(IAND(options, ROMS_PRSGRD40).ne.0.or.ANY(mix4).or.prs4x.eq.42))                                  &
     &  CALL unsupported ('WET_DRY is not built with UV_VIS4, TS_DIF4, PJ_GRADP (which the reference does not '//          &
     &                    'compile with WET_DRY either) or PJ_GRADPQ2', ierr)
      IF ((mix4(1).and.is_defined('UV_VIS2')).or.(mix4(2).and.is_defined('TS_DIF2')))                          &
     &  CALL unsupported ('harmonic and biharmonic mixing of the same field together (UV_VIS2 + UV_VIS4, TS_DIF2 + TS_DIF4) '// &
     &                    'are not built', ierr)
      IF (mix4(1).and.COUNT((/ is_defined('MIX_S_UV'), is_defined('MIX_GEO_UV') /)).ne.1)                               &
     &  CALL unsupported ('UV_VIS4 needs exactly one of MIX_S_UV, MIX_GEO_UV', ierr)
      IF (mix4(1)) options=IOR(options, ROMS_UV_VIS2)
      IF (mix4(2)) options=IOR(options, ROMS_TS_DIF2)
!  (an application without UV_ADV, UV_VIS2 or TS_DIF2 -- the reference's WINDBASIN option set -- runs since round 5: the
!  library is pinned to a reference build without them, oracle/ref/upwelling_noadv.h)
      IF (is_defined('UV_VIS2').and.COUNT((/ is_defined('MIX_S_UV'), is_defined('MIX_GEO_UV') /)).ne.1)                &
     &  CALL unsupported ('UV_VIS2 needs exactly one of MIX_S_UV, MIX_GEO_UV', ierr)
      mix_geo_uv=(is_defined('UV_VIS2').or.mix4(1)).and.is_defined('MIX_GEO_UV')         ! (uv3dmix2_geo.h | uv3dmix4_geo.h, round 6)
      IF ((is_defined('TS_DIF2').or.mix4(2)).and.COUNT((/ is_defined('MIX_S_TS'), is_defined('MIX_GEO_TS'), is_defined('MIX_ISO_TS') /)).ne.1) &
     &  CALL unsupported ('TS_DIF2 needs exactly one of MIX_S_TS, MIX_GEO_TS, MIX_ISO_TS', ierr)
      IF (is_defined('MIX_ISO_TS').and.(is_defined('TS_MIX_MAX_SLOPE').or.is_defined('TS_MIX_MIN_STRAT').or.          &
     &    is_defined('TS_MIX_STABILITY').or.is_defined('TS_MIX_CLIMA').or.is_defined('DIFF_3DCOEF')))                 &
     &  CALL unsupported ('MIX_ISO_TS is built in its default slope treatment (no TS_MIX_*, no DIFF_3DCOEF)', ierr)
!  without SPLINES_VDIFF / SPLINES_VVISC: the plain tridiagonal vertical solvers (step3d_t.F:1722-1790, step3d_uv.F:436-500)
      IF (.not.is_defined('SPLINES_VDIFF')) options=IOR(options, ROMS_PLAIN_VDIFF)
      IF (.not.is_defined('SPLINES_VVISC')) options=IOR(options, ROMS_PLAIN_VVISC)
      IF (.not.is_defined('SPLINES_VDIFF').and.ANY(hadv(1:NAT).eq.ROMS_MPDATA))                                &
     &  CALL unsupported ('MPDATA is built with SPLINES_VDIFF only', ierr)
      IF (COUNT((/ is_defined('ANA_VMIX'), is_defined('LMD_MIXING'), is_defined('GLS_MIXING'),                   &
     &             is_defined('MY25_MIXING') /)).gt.1)                                                         &
     &  CALL unsupported ('at most one vertical mixing closure: ANA_VMIX, LMD_MIXING, GLS_MIXING or MY25_MIXING (none: the '//      &
     &                    'background coefficients AKV_BAK, AKT_BAK, as in KELVIN)', ierr)
      IF (.not.(kelv.or.seam.or.grav.or.ovf).and..not.(is_defined('ANA_VMIX').or.is_defined('LMD_MIXING').or.           &
     &    is_defined('GLS_MIXING').or.is_defined('MY25_MIXING')))                                                                            &
     &  CALL unsupported ('a vertical mixing closure is required: ANA_VMIX, LMD_MIXING or GLS_MIXING', ierr)
!  GLS_MIXING: the stability functions, the smoothing, the shear form, the advection of the turbulent fields and the two
!  surface-flux options are run-time flags of the library; the wave-dependent forms and the limiters are not built
      IF (is_defined('MY25_MIXING').and.(is_defined('CANUTO_A').or.is_defined('CANUTO_B').or.is_defined('CHARNOK').or.  &
     &    is_defined('CRAIG_BANNER').or.is_defined('LIMIT_VDIFF').or.is_defined('LIMIT_VVISC')))                  &
     &  CALL unsupported ('MY25_MIXING takes KANTHA_CLAYSON, N2S2_HORAVG, RI_SPLINES, K_C2ADVECTION, K_C4ADVECTION', ierr)
      IF (is_defined('GLS_MIXING').and.(is_defined('ZOS_HSIG').or.is_defined('TKE_WAVEDISS').or.                 &
     &    is_defined('LIMIT_VDIFF').or.is_defined('LIMIT_VVISC')))                                              &
     &  CALL unsupported ('GLS_MIXING: ZOS_HSIG, TKE_WAVEDISS (wave fields) and LIMIT_VDIFF / LIMIT_VVISC are not built', ierr)
      IF (is_defined('GLS_MIXING').and.COUNT((/ is_defined('CANUTO_A'), is_defined('CANUTO_B'),                  &
     &    is_defined('KANTHA_CLAYSON') /)).gt.1)                                                                &
     &  CALL unsupported ('GLS_MIXING: at most one of CANUTO_A, CANUTO_B, KANTHA_CLAYSON (none: Galperin)', ierr)
      IF (is_defined('GLS_MIXING').and.is_defined('K_C2ADVECTION').and.is_defined('K_C4ADVECTION'))              &
     &  CALL unsupported ('GLS_MIXING: at most one of K_C2ADVECTION, K_C4ADVECTION', ierr)
      IF ((is_defined('ANA_FSOBC').or.is_defined('ANA_M2OBC')).and..not.kelv)                                   &
     &  CALL unsupported ('ANA_FSOBC / ANA_M2OBC (analytic boundary data) are built for KELVIN only', ierr)
      IF (is_defined('LMD_MIXING').and..not.(is_defined('LMD_RIMIX').and.is_defined('LMD_CONVEC').and.          &
     &    is_defined('LMD_SKPP').and.is_defined('LMD_NONLOCAL').and.is_defined('RI_SPLINES').and.               &
     &    is_defined('SOLAR_SOURCE')))                                                                         &
     &  CALL unsupported ('LMD_MIXING is built with LMD_RIMIX LMD_CONVEC LMD_SKPP LMD_NONLOCAL RI_SPLINES '//   &
     &                    'SOLAR_SOURCE', ierr)
      IF (is_defined('BULK_FLUXES').and..not.(bench.and.is_defined('LONGWAVE').and.is_defined('ALBEDO')))       &
     &  CALL unsupported ('BULK_FLUXES is built with the BENCHMARK analytic atmosphere (LONGWAVE, ALBEDO)', ierr)
      IF (is_defined('ALBEDO').and..not.bench)                                                                 &
     &  CALL unsupported ('ALBEDO (ana_srflux zenith-angle formula) is built for BENCHMARK only', ierr)
      IF ((is_defined('SPHERICAL').or.is_defined('CURVGRID').or.is_defined('NONLIN_EOS')).and..not.bench)       &
     &  CALL unsupported ('SPHERICAL / CURVGRID / NONLIN_EOS are built for BENCHMARK only', ierr)
!  MASKING: the analytic land of this host (island + headland, SUBROUTINE analytic_masks) goes with the UPWELLING grid; the
!  masked branches exist for all of its physics, MPDATA's included (mpdata_adiff.F's 13 masked blocks)
      END SUBROUTINE options_from_defines

!
!  " Activated C-preprocessing Options:" of the reference's run report (checkdefs.F:56-59, FORMAT 20 = (1x,a,t27,a)): the
!  application and its title, then every active option checkdefs.F has a line for, in its order -- the options of the
!  application header (or the built-in list) plus what globaldefs.h derives for any forward run of this kind
!  (ASSUMED_SHAPE, DOUBLE_PRECISION, NONLINEAR, POWER_LAW, PROFILE; RHO_SURF and VAR_RHO_2D with SOLVE3D; the analytic
!  momentum / heat fluxes drop out under BULK_FLUXES, globaldefs.h:993-1001).
!
      SUBROUTINE echo_cppdefs (iu)
      integer, intent(in) :: iu
      INCLUDE 'checkdefs_text.inc'
      integer :: k
      logical :: on
      write (iu,'(/,a,/)') ' Activated C-preprocessing Options:'
      write (iu,'(1x,a,t27,a)') TRIM(ADJUSTL(MyAppCPP)), TRIM(ADJUSTL(title))
      DO k=1,n_cpp_text
        SELECT CASE (TRIM(cpp_text_name(k)))
          CASE ('ASSUMED_SHAPE', 'DOUBLE_PRECISION', 'NONLINEAR', 'POWER_LAW', 'PROFILE')
            on=.TRUE.
          CASE ('RHO_SURF', 'VAR_RHO_2D')
            on=is_defined('SOLVE3D')
          CASE ('ANA_SMFLUX', 'ANA_STFLUX')
            on=is_defined(TRIM(cpp_text_name(k))).and..not.is_defined('BULK_FLUXES')
          CASE DEFAULT
            on=is_defined(TRIM(cpp_text_name(k)))
        END SELECT
        IF (on) write (iu,'(1x,a,t27,a)') TRIM(cpp_text_name(k)), TRIM(cpp_text(k))
      END DO
      END SUBROUTINE echo_cppdefs

      SUBROUTINE set_cppdefs (ierr)
      integer, intent(out) :: ierr
      integer :: L, k, nf
      character(len=512) :: flags
      character(len=64) :: ftok(16)
      ierr=0
      ndefs=0
      IF (LEN_TRIM(app_header).eq.0) THEN
        CALL GET_ENVIRONMENT_VARIABLE ('ROMS_APP_HEADER', app_header, L)
      END IF
!  options given on the compiler command line of a reference build (MY_CPP_FLAGS of its build script, e.g. "-DGLS_MIXING"
!  to switch the closure of upwelling.h): ROMS_CPP_FLAGS, defined before the header is read
      CALL GET_ENVIRONMENT_VARIABLE ('ROMS_CPP_FLAGS', flags, L)
      IF (L.gt.0) THEN
        CALL split (flags, ftok, nf)
        DO k=1,nf
          IF (ftok(k)(1:2).eq.'-D'.and.LEN_TRIM(ftok(k)).gt.2) CALL define (TRIM(ftok(k)(3:)))
        END DO
      END IF
      IF (LEN_TRIM(app_header).gt.0) THEN
        CALL read_app_header (TRIM(app_header), ierr)
!  time-averaged output is a cpp option of the application (AVERAGES, e.g. the stock upwelling.h has it,
!  benchmark.h has not): a header without it switches NAVG of roms.in off, as the reference build would
        IF (ierr.eq.0.and..not.is_defined('AVERAGES')) nAVG=0
!  ... and so are the per-term tracer tendencies (DIAGNOSTICS_TS, stock upwelling.h:32): NDIA counts only with it
        IF (ierr.eq.0.and..not.is_defined('DIAGNOSTICS_TS')) nDIA=0
        diag_uv=is_defined('DIAGNOSTICS_UV')
      ELSE
        CALL builtin_defines (ierr)
        diag_uv=ANY(DoutM2).or.ANY(DoutM3)           ! (the built-in option lists: the switches of roms.in decide)
      END IF
      CALL options_from_defines (ierr)
!  (tracers advected with MPDATA: their Dhadv / Dvadv work arrays are not built -- no diagnostics file for such a run)
      IF (ANY(hadv(1:NT).eq.ROMS_MPDATA)) nDIA=0
!  WET_DRY: the wet/dry statements of set_diags.F are not built -- a run that would write diagnostics files stops here (the
!  library refuses the same: roms_hip_dia_config), instead of accumulating them without the masks.  (AVERAGES with WET_DRY:
!  built in round 6 -- the full masks and wet-point counters of set_avg.F, k_avg.h.)
      IF (ierr.eq.0.and.wet_dry.and.nDIA.gt.0)                                                                            &
     &  CALL unsupported ('WET_DRY together with DIAGNOSTICS_TS / DIAGNOSTICS_UV output (NDIA > 0): '//                    &
     &                    'the wet/dry masks of set_diags.F are not built; set NDIA = 0', ierr)
      END SUBROUTINE set_cppdefs
!
!=======================================================================
!  Partition and array bounds (single tile = the whole domain on one GPU):
!  get_bounds.F:232-283, var_bounds :1044-1884, inp_par.F:210-226 (ghost points),
!  mod_param.F:1633-1636 (padding).
!=======================================================================
!
      SUBROUTINE set_bounds ()
      IF (ANY(hadv(1:NT).eq.ROMS_MPDATA).or.ANY(hadv(1:NT).eq.ROMS_HSIMT).or.mix4(1)) THEN      ! (UV_VIS4: inp_par.F:214-216)
        Nghost=3
      ELSE
        Nghost=2
      END IF
      Im=Lm+((Lm+2)/2-(Lm+1)/2)
      Jm=Mm+((Mm+2)/2-(Mm+1)/2)
      IF (EWperiodic) THEN
        LBi=-Nghost
        UBi=Im+Nghost
      ELSE
        LBi=0
        UBi=Im+1
      END IF
      IF (NSperiodic) THEN
        LBj=-Nghost
        UBj=Jm+Nghost
      ELSE
        LBj=0
        UBj=Jm+1
      END IF
      Istr=1
      Iend=Lm
      Jstr=1
      Jend=Mm
      IF (EWperiodic) THEN
        IstrR=Istr
        IendR=Iend
      ELSE
        IstrR=Istr-1
        IendR=Iend+1
      END IF
      IF (NSperiodic) THEN
        JstrR=Jstr
        JendR=Jend
      ELSE
        JstrR=Jstr-1
        JendR=Jend+1
      END IF
      IstrT=IstrR
      IendT=IendR
      JstrT=JstrR
      JendT=JendR
      IstrP=Istr
      JstrP=Jstr
      END SUBROUTINE set_bounds
!
!  periodic ghost copies, exchange_2d.F (single tile)
!
      SUBROUTINE exchange2d (A, gtype)
      real(r8), intent(inout) :: A(LBi:,LBj:)
      character(len=1), intent(in) :: gtype
      integer :: i, j, Jmin, Jmax, Imin, Imax
      IF (NSperiodic) THEN
        Jmin=Jstr
        Jmax=Jend
      ELSE
        Jmin=MERGE(JstrR,Jstr,gtype.eq.'r'.or.gtype.eq.'u')
        Jmax=JendR
      END IF
      IF (EWperiodic) THEN
        Imin=Istr
        Imax=Iend
      ELSE
        Imin=MERGE(IstrR,Istr,gtype.eq.'r'.or.gtype.eq.'v')
        Imax=IendR
      END IF
      IF (EWperiodic) THEN
        DO j=Jmin,Jmax
          A(Lm+1,j)=A(1,j)
          A(Lm+2,j)=A(2,j)
          IF (Nghost.eq.3) A(Lm+3,j)=A(3,j)
          A(-2,j)=A(Lm-2,j)
          A(-1,j)=A(Lm-1,j)
          A( 0,j)=A(Lm  ,j)
        END DO
      END IF
      IF (NSperiodic) THEN
        DO i=Imin,Imax
          A(i,Mm+1)=A(i,1)
          A(i,Mm+2)=A(i,2)
          IF (Nghost.eq.3) A(i,Mm+3)=A(i,3)
          A(i,-2)=A(i,Mm-2)
          A(i,-1)=A(i,Mm-1)
          A(i, 0)=A(i,Mm  )
        END DO
      END IF
      IF (EWperiodic.and.NSperiodic) THEN
        DO j=1,MERGE(3,2,Nghost.eq.3)
          DO i=1,MERGE(3,2,Nghost.eq.3)
            A(Lm+i,Mm+j)=A(i,j)
          END DO
          DO i=-2,0
            A(i,Mm+j)=A(Lm+i,j)
          END DO
        END DO
        DO j=-2,0
          DO i=1,MERGE(3,2,Nghost.eq.3)
            A(Lm+i,j)=A(i,Mm+j)
          END DO
          DO i=-2,0
            A(i,j)=A(Lm+i,Mm+j)
          END DO
        END DO
      END IF
      END SUBROUTINE exchange2d
!
!=======================================================================
!  Host set-up: what a caller that embeds the library gets from the reference itself (mod_grid,
!  mod_scalars, mod_ocean after ROMS_initialize); the stand-alone driver romsM builds it here.
!  Every routine below is the closed form of the quantity it fills, evaluated with whole-array
!  expressions; the arithmetic of each element is the one the reference performs (set_scoord.F,
!  set_weights.F, ana_grid.h, metrics.F, set_depth.F, ana_initial.h), so the arrays are identical
!  to the reference's, which tests/test_host.py asserts bit for bit against tests/golden/*_init.npz.
!=======================================================================
!
!  ---- vertical coordinate: stretching curve of A. Shchepetkin (2010), Vstretching = 4 ----------
!       C(s) = [1-cosh(theta_s s)]/[cosh(theta_s)-1], then C <- [exp(theta_b C)-1]/[1-exp(-theta_b)]
!       (set_scoord.F:433-475); s on the w-levels k/N-1 and on the rho-levels (k-1/2)/N-1.
!
      ELEMENTAL FUNCTION stretch_curve (s) RESULT (C)
      real(dp), intent(in) :: s
      real(dp) :: C
      IF (theta_s.gt.0.0_dp) THEN
        C=(1.0_dp-COSH(theta_s*s))/(COSH(theta_s)-1.0_dp)
      ELSE
        C=-s**2
      END IF
      IF (theta_b.gt.0.0_dp) C=(EXP(theta_b*C)-1.0_dp)/(1.0_dp-EXP(-theta_b))
      END FUNCTION stretch_curve

!  level index rk (k on w-levels, k - 1/2 on rho-levels) -> s and C(s) of Vstretching 2, 3, 5
      SUBROUTINE stretch_level (rk, s, C)
      real(dp), intent(in) :: rk
      real(dp), intent(out) :: s, C
      real(dp) :: rN, Csur, Cbot, wgt, Hscale
      rN=REAL(N,dp)
      IF (Vstretching.eq.5) THEN
        s=-(rk*rk-2.0_dp*rk*rN+rk+rN*rN-rN)/(rN*rN-rN)-0.01_dp*(rk*rk-rk*rN)/(1.0_dp-rN)
      ELSE
        s=(1.0_dp/rN)*(rk-rN)
      END IF
      SELECT CASE (Vstretching)
        CASE (2)
          IF (theta_s.gt.0.0_dp) THEN
            Csur=(1.0_dp-COSH(theta_s*s))/(COSH(theta_s)-1.0_dp)
            IF (theta_b.gt.0.0_dp) THEN
              Cbot=SINH(theta_b*(s+1.0_dp))/SINH(theta_b)-1.0_dp
              wgt=(s+1.0_dp)**1.0_dp*(1.0_dp+(1.0_dp/1.0_dp)*(1.0_dp-(s+1.0_dp)**1.0_dp))
              C=wgt*Csur+(1.0_dp-wgt)*Cbot
            ELSE
              C=Csur
            END IF
          ELSE
            C=s
          END IF
        CASE (3)
          Hscale=3.0_dp
          Cbot=LOG(COSH(Hscale*(s+1.0_dp)**theta_b))/LOG(COSH(Hscale))-1.0_dp
          Csur=-LOG(COSH(Hscale*ABS(s)**theta_s))/LOG(COSH(Hscale))
          wgt=0.5_dp*(1.0_dp-TANH(Hscale*(s+0.5_dp)))
          C=wgt*Cbot+(1.0_dp-wgt)*Csur
        CASE DEFAULT
          IF (theta_s.gt.0.0_dp) THEN
            Csur=(1.0_dp-COSH(theta_s*s))/(COSH(theta_s)-1.0_dp)
          ELSE
            Csur=-s**2
          END IF
          IF (theta_b.gt.0.0_dp) THEN
            C=(EXP(theta_b*Csur)-1.0_dp)/(1.0_dp-EXP(-theta_b))
          ELSE
            C=Csur
          END IF
      END SELECT
      END SUBROUTINE stretch_level

      SUBROUTINE vertical_coordinate (ierr)
      integer, intent(out) :: ierr
      integer :: k
      real(dp) :: dsig, c1, c2
      ierr=0
      hc=MERGE(MIN(hmin,Tcline), Tcline, Vtransform.eq.1)
      IF (Vstretching.eq.1) THEN          ! Song and Haidvogel (1994), set_scoord.F:184-240 (the OVERFLOW application)
        IF (theta_s.ne.0.0_dp) THEN
          c1=1.0_dp/SINH(theta_s)
          c2=0.5_dp/TANH(0.5_dp*theta_s)
        END IF
        sc_w(0)=-1.0_dp;  Cs_w(0)=-1.0_dp
        dsig=1.0_dp/REAL(N,dp)
        DO k=1,N
          sc_w(k)=dsig*REAL(k-N,dp)
          sc_r(k)=dsig*(REAL(k-N,dp)-0.5_dp)
          IF (theta_s.ne.0.0_dp) THEN
            Cs_w(k)=(1.0_dp-theta_b)*c1*SINH(theta_s*sc_w(k))+theta_b*(c2*TANH(theta_s*(sc_w(k)+0.5_dp))-0.5_dp)
            Cs_r(k)=(1.0_dp-theta_b)*c1*SINH(theta_s*sc_r(k))+theta_b*(c2*TANH(theta_s*(sc_r(k)+0.5_dp))-0.5_dp)
          ELSE
            Cs_w(k)=sc_w(k)
            Cs_r(k)=sc_r(k)
          END IF
        END DO
        RETURN
      END IF
      IF (Vstretching.eq.2.or.Vstretching.eq.3.or.Vstretching.eq.5) THEN
!  (round 6) the other curves of set_scoord.F: 2 = Shchepetkin (2005), :240-292; 3 = Geyer's bottom-boundary-layer
!  function, :298-337; 5 = Souza's quadratic Legendre levels under the double curve of 4, :478-526.  The surface and bottom
!  w-levels are exact by definition; every interior level is one evaluation of stretch_level
        sc_w(N)=0.0_dp;   Cs_w(N)=0.0_dp
        sc_w(0)=-1.0_dp;  Cs_w(0)=-1.0_dp
        DO k=1,N
          IF (k.lt.N) CALL stretch_level (REAL(k,dp), sc_w(k), Cs_w(k))
          CALL stretch_level (REAL(k,dp)-0.5_dp, sc_r(k), Cs_r(k))
        END DO
        RETURN
      END IF
      IF (Vstretching.ne.4) THEN          ! the BASELINE applications use 4
        CALL unsupported ('Vstretching: built are 1 (Song and Haidvogel 1994), 2 (Shchepetkin 2005), 3 (Geyer), '//            &
     &                    '4 (Shchepetkin 2010) and 5 (Souza)', ierr)
        RETURN
      END IF
      dsig=1.0_dp/REAL(N,dp)
      sc_w(1:N-1)=dsig*REAL([(k-N, k=1,N-1)],dp)
      Cs_w(1:N-1)=stretch_curve(sc_w(1:N-1))
      sc_w(0)=-1.0_dp;  Cs_w(0)=-1.0_dp          ! bottom and surface are exact by definition
      sc_w(N)=0.0_dp;   Cs_w(N)=0.0_dp
      sc_r(1:N)=dsig*(REAL([(k-N, k=1,N)],dp)-0.5_dp)
      Cs_r(1:N)=stretch_curve(sc_r(1:N))
      END SUBROUTINE vertical_coordinate
!
!  ---- barotropic time filter (set_weights.F:56-230) -------------------------------------------
!       primary weights  a_m ~ B(m s),  B(x) = x**alpha - x**(alpha+beta) - gamma x,  m = 1..2 ndtfast,
!       s iterated (16 sweeps) until the centroid of the positive lobe sits at ndtfast; the lobe is then
!       moved by fractions of a sub-step until its centroid is exact; the secondary weights are the
!       suffix sums b_m = sum_{m' >= m} a_m'; both sets are normalised to unit sum.
!
      PURE FUNCTION lobe_moments (a, n) RESULT (mom)      ! (sum a_m, sum m a_m), m ascending
      real(dp), intent(in) :: a(:)
      integer, intent(in) :: n
      real(r16) :: mom(2)
      integer :: m
      mom=0.0_r16
      DO m=1,n
        mom(1)=mom(1)+a(m)
        mom(2)=mom(2)+a(m)*REAL(m,dp)
      END DO
      END FUNCTION lobe_moments

      SUBROUTINE barotropic_filter ()
      real(dp), allocatable :: a(:), b(:)
      real(dp) :: gam, s
      real(r16) :: mom(2), x, lag, keep
      integer :: m, n, sweep, last
      logical :: in_lobe
      allocate ( a(2*ndtfast), b(2*ndtfast) )
      s=(Falpha+1.0_dp)*(Falpha+Fbeta+1.0_dp)/((Falpha+2.0_dp)*(Falpha+Fbeta+2.0_dp)*REAL(ndtfast,dp))
      gam=Fgamma*MAX(0.0_dp, 1.0_dp-10.0_dp/REAL(ndtfast,dp))
      n=0
      DO sweep=1,16
        in_lobe=.FALSE.
        last=0
        DO m=1,2*ndtfast
          x=s*REAL(m,dp)
          a(m)=x**Falpha-x**(Falpha+Fbeta)-gam*x
          IF (a(m).gt.0.0_dp) THEN
            in_lobe=.TRUE.
            last=m
          ELSE IF (in_lobe.and.(a(m).lt.0.0_dp)) THEN
            a(m)=0.0_dp                   ! beyond the lobe
          END IF
        END DO
        n=last
        mom=lobe_moments(a, n)
        s=s*mom(2)/(mom(1)*REAL(ndtfast,dp))
      END DO
      DO sweep=1,ndtfast
        mom=lobe_moments(a, n)
        lag=REAL(ndtfast,dp)-mom(2)/mom(1)          ! sub-steps the centroid is short of ndtfast
        IF (lag.gt.1.0_r16) THEN                    ! whole sub-step later
          n=n+1
          a(2:n)=a(1:n-1)
          a(1)=0.0_dp
        ELSE IF (lag.gt.0.0_r16) THEN               ! fraction of a sub-step later
          keep=1.0_r16-lag
          a(2:n)=keep*a(2:n)+lag*a(1:n-1)
          a(1)=keep*a(1)
        ELSE IF (lag.lt.-1.0_r16) THEN              ! whole sub-step earlier
          n=n-1
          a(1:n)=a(2:n+1)
          a(n+1)=0.0_dp
        ELSE IF (lag.lt.0.0_r16) THEN               ! fraction of a sub-step earlier
          keep=1.0_r16+lag
          a(1:n-1)=keep*a(1:n-1)-lag*a(2:n)
          a(n)=keep*a(n)
        END IF
      END DO
      b=0.0_dp
      DO m=1,n                                      ! b_m = a_m + a_{m+1} + ... + a_n, added in that order
        x=0.0_r16
        DO last=m,n
          x=x+a(last)
        END DO
        b(m)=x
      END DO
      mom(1)=0.0_r16
      mom(2)=0.0_r16
      DO m=1,n
        mom(1)=mom(1)+a(m)
        mom(2)=mom(2)+b(m)
      END DO
      a(1:n)=(1.0_r16/mom(1))*a(1:n)
      b(1:n)=(1.0_r16/mom(2))*b(1:n)
      nfast=n
      weight(1,:)=a
      weight(2,:)=b
      deallocate ( a, b )
      END SUBROUTINE barotropic_filter
!
!  ---- analytic grids (ana_grid.h: UPWELLING :1058-1082, BENCHMARK :462-482,677-690,931-936) ---
!
      ELEMENTAL FUNCTION shelf_depth (rows_from_wall) RESULT (d)   ! UPWELLING: tanh shelf, 150 m cap
      real(r8), intent(in) :: rows_from_wall
      real(r8) :: d
      d=MIN(150.0_r8, 84.5_r8+66.526_r8*TANH((rows_from_wall-10.0_r8)/7.0_r8))
      END FUNCTION shelf_depth

      SUBROUTINE periodic_fill_grid ()
      CALL exchange2d (pm, 'r')
      CALL exchange2d (pn, 'r')
      CALL exchange2d (angler, 'r')
      CALL exchange2d (f, 'r')
      CALL exchange2d (h, 'r')
      END SUBROUTINE periodic_fill_grid

      SUBROUTINE grid_upwelling ()
!  Cartesian channel, 1 km cells, f-plane, shelf on both walls of the non-periodic direction.
      real(r8) :: dx, dy
      real(r8), allocatable :: across(:)
      integer :: i, j, nacross, c0, c1
      xl=1000.0_r8*REAL(Lm,r8)
      el=1000.0_r8*REAL(Mm,r8)
      dx=xl/REAL(Lm,r8)
      dy=el/REAL(Mm,r8)
      FORALL (i=Istr-1:Iend+1, j=Jstr-1:Jend+1)
        xr(i,j)=dx*(REAL(i-1,r8)+0.5_r8)
        yr(i,j)=dy*(REAL(j-1,r8)+0.5_r8)
      END FORALL
      pm(IstrT:IendT,JstrT:JendT)=1.0_r8/dx
      pn(IstrT:IendT,JstrT:JendT)=1.0_r8/dy
      angler(IstrT:IendT,JstrT:JendT)=0.0_r8
      f(IstrT:IendT,JstrT:JendT)=-8.26E-05_r8
!  depth depends on the distance (in rows) to the nearer wall of the cross-channel direction
      IF (NSperiodic) THEN
        nacross=Lm; c0=IstrT; c1=IendT
      ELSE
        nacross=Mm; c0=JstrT; c1=JendT
      END IF
      allocate ( across(c0:c1) )
      across=shelf_depth(REAL([(MERGE(j, nacross+1-j, j.le.nacross/2), j=c0,c1)],r8))
      IF (NSperiodic) THEN
        h(IstrT:IendT,JstrT:JendT)=SPREAD(across, 2, JendT-JstrT+1)
      ELSE IF (EWperiodic) THEN
        h(IstrT:IendT,JstrT:JendT)=SPREAD(across, 1, IendT-IstrT+1)
      END IF
      deallocate ( across )
      END SUBROUTINE grid_upwelling

      SUBROUTINE grid_benchmark ()
!  Spherical zonal channel 360 deg x (70S..50S), cells of equal angle: pm varies with latitude only.
      real(r8) :: dlon, dlat, per_rad_x, inv_dy, omega2
      real(r8), allocatable :: lat(:), inv_dx(:)
      integer :: i, j, j0, j1
      xl=360.0_r8
      el=20.0_r8
      dlon=xl/REAL(Lm,r8)
      dlat=el/REAL(Mm,r8)
      j0=MIN(JstrT,Jstr-1)
      j1=MAX(Jend+1,JendT)
      allocate ( lat(j0:j1), inv_dx(j0:j1) )
      lat=-70.0_r8+dlat*(REAL([(j, j=j0,j1)],r8)-0.5_r8)
      FORALL (i=Istr-1:Iend+1, j=Jstr-1:Jend+1)
        lonr(i,j)=dlon*(REAL(i,r8)-0.5_r8)
        latr(i,j)=lat(j)
      END FORALL
      per_rad_x=REAL(Lm,r8)/(2.0_r8*pi*Eradius)
      inv_dy=REAL(Mm,r8)*360.0_r8/(2.0_r8*pi*Eradius*el)
      inv_dx=per_rad_x*(1.0_r8/COS(lat*deg2rad))
      pm(IstrT:IendT,JstrT:JendT)=SPREAD(inv_dx(JstrT:JendT), 1, IendT-IstrT+1)
      pn(IstrT:IendT,JstrT:JendT)=inv_dy
!  d(1/pn)/dxi vanishes identically; d(1/pm)/deta is a centred difference of the row values
      dndx(Istr:Iend,Jstr:Jend)=0.5_r8*((1.0_r8/inv_dy)-(1.0_r8/inv_dy))
      dmde(Istr:Iend,Jstr:Jend)=SPREAD(0.5_r8*((1.0_r8/inv_dx(Jstr+1:Jend+1))-(1.0_r8/inv_dx(Jstr-1:Jend-1))),  &
     &                                 1, Iend-Istr+1)
      CALL exchange2d (dndx, 'r')
      CALL exchange2d (dmde, 'r')
      omega2=2.0_r8*(2.0_r8*pi*366.25_r8/365.25_r8)/86400.0_r8
      angler(IstrT:IendT,JstrT:JendT)=0.0_r8
      f(IstrT:IendT,JstrT:JendT)=omega2*SIN(latr(IstrT:IendT,JstrT:JendT)*deg2rad)
      h(IstrT:IendT,JstrT:JendT)=500.0_r8+1750.0_r8*(1.0+TANH((68.0_r8+latr(IstrT:IendT,JstrT:JendT))/dlat))
      deallocate ( lat, inv_dx )
      END SUBROUTINE grid_benchmark

      SUBROUTINE grid_kelvin ()
!  ana_grid.h:286-291 with the generic Cartesian branch :515-533, :886-891, :1130-1136: a flat channel of 20 km cells,
!  100 m deep, f-plane
      real(r8) :: dx, dy
      integer :: i, j
      xl=20000.0_r8*REAL(Lm,r8)
      el=20000.0_r8*REAL(Mm,r8)
      dx=xl/REAL(Lm,r8)
      dy=el/REAL(Mm,r8)
      FORALL (i=Istr-1:Iend+1, j=Jstr-1:Jend+1)
        xp(i,j)=dx*REAL(i-1,r8)
        xr(i,j)=dx*(REAL(i-1,r8)+0.5_r8)
        yp(i,j)=dy*REAL(j-1,r8)
        yr(i,j)=dy*(REAL(j-1,r8)+0.5_r8)
      END FORALL
      pm(IstrT:IendT,JstrT:JendT)=1.0_r8/dx
      pn(IstrT:IendT,JstrT:JendT)=1.0_r8/dy
      angler(IstrT:IendT,JstrT:JendT)=0.0_r8
      f(IstrT:IendT,JstrT:JendT)=1.0E-04_r8
      h(IstrT:IendT,JstrT:JendT)=100.0_r8
      END SUBROUTINE grid_kelvin

      SUBROUTINE grid_cartesian (Xsize, Esize, depth, f0)
!  the generic Cartesian branch of ana_grid.h (:515-533 coordinates, :886-891 f-plane, :1130-1136 flat bottom)
      real(r8), intent(in) :: Xsize, Esize, depth, f0
      real(r8) :: dx, dy
      integer :: i, j
      xl=Xsize
      el=Esize
      dx=Xsize/REAL(Lm,r8)
      dy=Esize/REAL(Mm,r8)
      FORALL (i=Istr-1:Iend+1, j=Jstr-1:Jend+1)
        xp(i,j)=dx*REAL(i-1,r8)
        xr(i,j)=dx*(REAL(i-1,r8)+0.5_r8)
        yp(i,j)=dy*REAL(j-1,r8)
        yr(i,j)=dy*(REAL(j-1,r8)+0.5_r8)
      END FORALL
      pm(IstrT:IendT,JstrT:JendT)=1.0_r8/dx
      pn(IstrT:IendT,JstrT:JendT)=1.0_r8/dy
      angler(IstrT:IendT,JstrT:JendT)=0.0_r8
      f(IstrT:IendT,JstrT:JendT)=f0
      h(IstrT:IendT,JstrT:JendT)=depth
      END SUBROUTINE grid_cartesian

      ELEMENTAL FUNCTION seamount_depth (x, y) RESULT (d)      ! ana_grid.h:1032-1039: a Gaussian seamount 4500 m tall, 40 km wide
      real(r8), intent(in) :: x, y
      real(r8) :: d, val1, val2
      val1=(x-0.5_r8*320.0E+03_r8)/40000.0_r8
      val2=(y-0.5_r8*320.0E+03_r8)/40000.0_r8
      d=5000.0_r8-4500.0_r8*EXP(-(val1*val1+val2*val2))
      END FUNCTION seamount_depth

      ELEMENTAL FUNCTION overflow_depth (y) RESULT (d)         ! ana_grid.h:1004-1011
      real(r8), intent(in) :: y
      real(r8) :: d
      d=200.0_r8+0.5_r8*(4000.0_r8-200.0_r8)*(1.0_r8+TANH((y-100000.0_r8)/20000.0_r8))
      END FUNCTION overflow_depth

      SUBROUTINE analytic_grid (ierr)
      integer, intent(out) :: ierr
      ierr=0
      IF (IAND(options,ROMS_APP_SEAMOUNT).ne.0) THEN            ! ana_grid.h:346-352
        CALL grid_cartesian (320.0E+03_r8, 320.0E+03_r8, 5000.0_r8, 0.0_r8)
        h(IstrT:IendT,JstrT:JendT)=seamount_depth(xr(IstrT:IendT,JstrT:JendT), yr(IstrT:IendT,JstrT:JendT))
      ELSE IF (IAND(options,ROMS_APP_OVERFLOW).ne.0) THEN       ! :328-333, :1004-1011: a tanh slope from 200 m down to 4000 m
        CALL grid_cartesian (4.0E+03_r8, 200.0E+03_r8, 4000.0_r8, 0.0_r8)
        h(IstrT:IendT,JstrT:JendT)=overflow_depth(yr(IstrT:IendT,JstrT:JendT))
      ELSE IF (IAND(options,ROMS_APP_GRAV_ADJ).ne.0) THEN       ! :298-304
        CALL grid_cartesian (64.0E+03_r8, REAL(Mm,r8)*64.0E+03_r8/REAL(Lm,r8), 20.0_r8, 0.0_r8)
      ELSE IF (IAND(options,ROMS_APP_KELVIN).ne.0) THEN
        CALL grid_kelvin ()
      ELSE IF (IAND(options,ROMS_APP_UPWELLING).ne.0) THEN
        CALL grid_upwelling ()
      ELSE IF (IAND(options,ROMS_APP_BENCHMARK).ne.0) THEN
        CALL grid_benchmark ()
      ELSE
        ierr=5
        RETURN
      END IF
      hmin=MINVAL(h(IstrT:IendT,JstrT:JendT))
      hmax=MAXVAL(h(IstrT:IendT,JstrT:JendT))
      CALL periodic_fill_grid ()
      END SUBROUTINE analytic_grid
!
!  ---- derived metrics (metrics.F:23-250): cell sizes and their ratios at rho, u, v, psi points --
!       a u-/v-/psi-point value is built from the SUM of pm (pn) over its 2 / 2 / 4 surrounding
!       rho points: size = (number of points)/sum, ratio = sum/sum.
!
      SUBROUTINE grid_metrics ()
      real(r8), allocatable :: sm(:,:), sn(:,:)
      integer :: i0, i1, j0, j1
      i0=IstrT; i1=IendT; j0=JstrT; j1=JendT
      om_r(i0:i1,j0:j1)=1.0_r8/pm(i0:i1,j0:j1)
      on_r(i0:i1,j0:j1)=1.0_r8/pn(i0:i1,j0:j1)
      omn(i0:i1,j0:j1)=1.0_r8/(pm(i0:i1,j0:j1)*pn(i0:i1,j0:j1))
      fomn(i0:i1,j0:j1)=f(i0:i1,j0:j1)*omn(i0:i1,j0:j1)
      pnom_r(i0:i1,j0:j1)=pn(i0:i1,j0:j1)/pm(i0:i1,j0:j1)
      pmon_r(i0:i1,j0:j1)=pm(i0:i1,j0:j1)/pn(i0:i1,j0:j1)
      allocate ( sm(LBi:UBi,LBj:UBj), sn(LBi:UBi,LBj:UBj) )
!  u points (i-1,j)+(i,j)
      i0=IstrP
      sm(i0:i1,j0:j1)=pm(i0-1:i1-1,j0:j1)+pm(i0:i1,j0:j1)
      sn(i0:i1,j0:j1)=pn(i0-1:i1-1,j0:j1)+pn(i0:i1,j0:j1)
      pmon_u(i0:i1,j0:j1)=sm(i0:i1,j0:j1)/sn(i0:i1,j0:j1)
      pnom_u(i0:i1,j0:j1)=sn(i0:i1,j0:j1)/sm(i0:i1,j0:j1)
      om_u(i0:i1,j0:j1)=2.0_r8/sm(i0:i1,j0:j1)
      on_u(i0:i1,j0:j1)=2.0_r8/sn(i0:i1,j0:j1)
!  v points (i,j-1)+(i,j)
      i0=IstrT; j0=JstrP
      sm(i0:i1,j0:j1)=pm(i0:i1,j0-1:j1-1)+pm(i0:i1,j0:j1)
      sn(i0:i1,j0:j1)=pn(i0:i1,j0-1:j1-1)+pn(i0:i1,j0:j1)
      pmon_v(i0:i1,j0:j1)=sm(i0:i1,j0:j1)/sn(i0:i1,j0:j1)
      pnom_v(i0:i1,j0:j1)=sn(i0:i1,j0:j1)/sm(i0:i1,j0:j1)
      om_v(i0:i1,j0:j1)=2.0_r8/sm(i0:i1,j0:j1)
      on_v(i0:i1,j0:j1)=2.0_r8/sn(i0:i1,j0:j1)
!  psi points (i-1,j-1)+(i-1,j)+(i,j-1)+(i,j), added in that order
      i0=IstrP
      sm(i0:i1,j0:j1)=pm(i0-1:i1-1,j0-1:j1-1)+pm(i0-1:i1-1,j0:j1)+pm(i0:i1,j0-1:j1-1)+pm(i0:i1,j0:j1)
      sn(i0:i1,j0:j1)=pn(i0-1:i1-1,j0-1:j1-1)+pn(i0-1:i1-1,j0:j1)+pn(i0:i1,j0-1:j1-1)+pn(i0:i1,j0:j1)
      pnom_p(i0:i1,j0:j1)=sn(i0:i1,j0:j1)/sm(i0:i1,j0:j1)
      pmon_p(i0:i1,j0:j1)=sm(i0:i1,j0:j1)/sn(i0:i1,j0:j1)
      om_p(i0:i1,j0:j1)=4.0_r8/sm(i0:i1,j0:j1)
      on_p(i0:i1,j0:j1)=4.0_r8/sn(i0:i1,j0:j1)
      deallocate ( sm, sn )
      CALL exchange2d (om_r, 'r');   CALL exchange2d (on_r, 'r');   CALL exchange2d (omn, 'r')
      CALL exchange2d (fomn, 'r');   CALL exchange2d (pnom_r, 'r'); CALL exchange2d (pmon_r, 'r')
      CALL exchange2d (pmon_u, 'u'); CALL exchange2d (pnom_u, 'u'); CALL exchange2d (om_u, 'u')
      CALL exchange2d (on_u, 'u');   CALL exchange2d (pmon_v, 'v'); CALL exchange2d (pnom_v, 'v')
      CALL exchange2d (om_v, 'v');   CALL exchange2d (on_v, 'v');   CALL exchange2d (pnom_p, 'p')
      CALL exchange2d (pmon_p, 'p'); CALL exchange2d (om_p, 'p');   CALL exchange2d (on_p, 'p')
      END SUBROUTINE grid_metrics
!
!  ---- horizontal mixing, drag and background vertical mixing (ini_hmixcoef.F:29 with uniform
!       coefficients; allocation defaults of mod_mixing.F / mod_grid.F) ---------------------------
!
      SUBROUTINE ini_mixing ()
      integer :: itrc, k
!  single tile: the tile range IstrT:IendT plus the periodic exchange covers the whole array,
!  and the allocation defaults of mod_grid/mod_mixing are whole-array assignments.
      visc2_r=visc2
      visc2_p=visc2
      DO itrc=1,NT
        diff2(:,:,itrc)=tnu2(itrc)
      END DO
!  biharmonic mixing in place of the harmonic one: zero harmonic coefficients (the biharmonic ones are uploaded in
!  device_init as the square roots inp_par.F:634 / read_phypar.F:7840 take)
      IF (mix4(1)) THEN
        visc2_r=0.0_dp
        visc2_p=0.0_dp
      END IF
      IF (mix4(2)) diff2=0.0_dp
      rdrag=rdrg
      rdrag2=rdrg2
      Akv=0.0_r8
      Akt=0.0_r8
      DO k=1,N-1
        Akv(:,:,k)=Akv_bak
        DO itrc=1,NAT
          Akt(:,:,k,itrc)=Akt_bak(itrc)
        END DO
      END DO
      END SUBROUTINE ini_mixing
!
!  ---- depths of the s-levels for a given free surface (set_depth.F:76-278) ---------------------
!       Vtransform 1:  z = z0 + zeta (1 + z0/h),          z0 = hc (s - C) + C h
!       Vtransform 2:  z = zeta + (zeta + h) S,           S  = (hc s + C h)/(hc + h)
!
      SUBROUTINE level_depths (zsurf)
      real(r8), intent(in) :: zsurf(LBi:,LBj:)
      real(r8), allocatable :: hw(:,:), zs(:,:), rest(:,:)
      integer :: k, i0, i1, j0, j1
      i0=IstrT; i1=IendT; j0=JstrT; j1=JendT
      allocate ( hw(i0:i1,j0:j1), zs(i0:i1,j0:j1), rest(i0:i1,j0:j1) )
      hw=h(i0:i1,j0:j1)
      zs=zsurf(i0:i1,j0:j1)
      z_w(i0:i1,j0:j1,0)=-hw
      DO k=1,N
        IF (Vtransform.eq.1) THEN
          rest=hc*(sc_w(k)-Cs_w(k))+Cs_w(k)*hw
          z_w(i0:i1,j0:j1,k)=rest+zs*(1.0_r8+rest*(1.0_r8/hw))
          rest=hc*(sc_r(k)-Cs_r(k))+Cs_r(k)*hw
          z_r(i0:i1,j0:j1,k)=rest+zs*(1.0_r8+rest*(1.0_r8/hw))
        ELSE
          rest=(hc*sc_w(k)+Cs_w(k)*hw)*(1.0_r8/(hc+hw))
          z_w(i0:i1,j0:j1,k)=zs+(zs+hw)*rest
          rest=(hc*sc_r(k)+Cs_r(k)*hw)*(1.0_r8/(hc+hw))
          z_r(i0:i1,j0:j1,k)=zs+(zs+hw)*rest
        END IF
        Hz(i0:i1,j0:j1,k)=z_w(i0:i1,j0:j1,k)-z_w(i0:i1,j0:j1,k-1)
      END DO
      deallocate ( hw, zs, rest )
      DO k=0,N
        CALL exchange2d (z_w(:,:,k), 'r')
      END DO
      DO k=1,N
        CALL exchange2d (z_r(:,:,k), 'r')
        CALL exchange2d (Hz(:,:,k), 'r')
      END DO
      END SUBROUTINE level_depths
!
!  ---- analytic initial state: ocean at rest, temperature a function of depth only, uniform salt
!       (ana_initial.h: UPWELLING :828-849, BENCHMARK :545-560) ---------------------------------
!
      ELEMENTAL FUNCTION temp_upwelling (z) RESULT (temp)
      real(r8), intent(in) :: z
      real(r8) :: temp
      temp=T0+8.0_r8*EXP(z/50.0_r8)
      END FUNCTION temp_upwelling

      ELEMENTAL FUNCTION temp_benchmark (z, amp) RESULT (temp)
      real(r8), intent(in) :: z, amp
      real(r8) :: temp
      temp=amp*EXP(z/800.0_r8)*(0.6_r8-0.4_r8*TANH(z/800.0_r8))
      END FUNCTION temp_benchmark

      SUBROUTINE initial_state ()
      real(r8) :: ratio2, amp
      integer :: i0, i1, j0, j1, k
      i0=IstrT; i1=IendT; j0=JstrT; j1=JendT
      zeta=0.0_r8;  ubar=0.0_r8;  vbar=0.0_r8
      u=0.0_r8;     v=0.0_r8;     t=0.0_r8
      IF (IAND(options,ROMS_APP_BENCHMARK).ne.0) THEN
        ratio2=(44.69_r8/39.382_r8)**2
        amp=ratio2*(rho0*800.0_r8/g)*(5.0E-05_r8/((42.689_r8/44.69_r8)**2))
        t(i0:i1,j0:j1,:,1,1)=temp_benchmark(z_r(i0:i1,j0:j1,:), amp)
        t(i0:i1,j0:j1,:,1,2)=35.0_r8
      ELSE IF (IAND(options,ROMS_APP_SEAMOUNT).ne.0) THEN      ! ana_initial.h:809-816
        t(i0:i1,j0:j1,:,1,1)=T0+7.5_r8*EXP(z_r(i0:i1,j0:j1,:)/1000.0_r8)
      ELSE IF (IAND(options,ROMS_APP_OVERFLOW).ne.0) THEN      ! :778-787: cold water on the shelf, T0 -> 0 across y = 60 km
        DO k=1,N
          t(i0:i1,j0:j1,k,1,1)=T0-0.5_r8*T0*(1.0_r8+TANH((yr(i0:i1,j0:j1)-60000.0_r8)/2000.0_r8))
        END DO
      ELSE IF (IAND(options,ROMS_APP_GRAV_ADJ).ne.0) THEN      ! :672-686: warm water in the left half
        t(i0:MIN((Lm+1)/2,i1),j0:j1,:,1,1)=T0+5.0_r8
        t(MAX((Lm+1)/2+1,i0):i1,j0:j1,:,1,1)=T0
      ELSE IF (IAND(options,ROMS_APP_KELVIN).ne.0) THEN
!  ana_initial.h: the default branch, T0 everywhere; no SALINITY: the second tracer (carried passively) stays zero
        t(i0:i1,j0:j1,:,1,1)=T0
      ELSE
        t(i0:i1,j0:j1,:,1,1)=temp_upwelling(z_r(i0:i1,j0:j1,:))
        t(i0:i1,j0:j1,:,1,2)=S0
      END IF
      END SUBROUTINE initial_state
!
!=======================================================================
!  Allocate (mod_arrays.F), set up and initialise the host state.
!=======================================================================
!
      SUBROUTINE host_setup (ierr)
      integer, intent(out) :: ierr
      IF (NAT.eq.1) THEN
!  (an application without salinity, NAT = 1: the library steps two tracers; the second one is carried passively -- zero,
!  the advection schemes, diffusivities and boundary conditions of the first)
        NAT=2
        hadv(2)=hadv(1); vadv(2)=vadv(1); tnu2(2)=0.0_dp
        lbc(7,:)=lbc(6,:); Tnudg(2)=Tnudg(1)
      END IF
      NT=NAT
      IF (ANY(hadv(1:NT).lt.0).or.ANY(vadv(1:NT).lt.0)) THEN
        CALL unsupported ('Hadvection/Vadvection: unknown scheme (A4 C2 C4 HSIMT MPDATA SP SU3 U3)', ierr)
        RETURN
      END IF
      IF (ANY(hadv(1:NT).eq.ROMS_MPDATA.neqv.vadv(1:NT).eq.ROMS_MPDATA)) THEN
        CALL unsupported ('MPDATA must be chosen for both Hadvection and Vadvection of a tracer', ierr)
        RETURN
      END IF
      CALL set_cppdefs (ierr)
      IF (ierr.ne.0) RETURN
      CALL set_bounds ()
      dtfast=dt/REAL(ndtfast,r8)
      IF (allocated(h)) CALL host_free ()
      allocate ( weight(2,2*ndtfast), sc_r(N), Cs_r(N), sc_w(0:N), Cs_w(0:N) )
      allocate ( h(LBi:UBi,LBj:UBj), f(LBi:UBi,LBj:UBj), fomn(LBi:UBi,LBj:UBj), pm(LBi:UBi,LBj:UBj),           &
     &           pn(LBi:UBi,LBj:UBj), om_r(LBi:UBi,LBj:UBj), on_r(LBi:UBi,LBj:UBj), om_u(LBi:UBi,LBj:UBj),     &
     &           on_u(LBi:UBi,LBj:UBj), om_v(LBi:UBi,LBj:UBj), on_v(LBi:UBi,LBj:UBj), om_p(LBi:UBi,LBj:UBj),   &
     &           on_p(LBi:UBi,LBj:UBj), omn(LBi:UBi,LBj:UBj), pmon_r(LBi:UBi,LBj:UBj), pnom_r(LBi:UBi,LBj:UBj),&
     &           pmon_p(LBi:UBi,LBj:UBj), pnom_p(LBi:UBi,LBj:UBj), pmon_u(LBi:UBi,LBj:UBj),                    &
     &           pnom_u(LBi:UBi,LBj:UBj), pmon_v(LBi:UBi,LBj:UBj), pnom_v(LBi:UBi,LBj:UBj),                    &
     &           dmde(LBi:UBi,LBj:UBj), dndx(LBi:UBi,LBj:UBj), angler(LBi:UBi,LBj:UBj), xr(LBi:UBi,LBj:UBj),   &
     &           xp(LBi:UBi,LBj:UBj), yp(LBi:UBi,LBj:UBj),                                                     &
     &           yr(LBi:UBi,LBj:UBj), lonr(LBi:UBi,LBj:UBj), latr(LBi:UBi,LBj:UBj), rdrag(LBi:UBi,LBj:UBj),    &
     &           rdrag2(LBi:UBi,LBj:UBj), visc2_r(LBi:UBi,LBj:UBj), visc2_p(LBi:UBi,LBj:UBj),                  &
     &           diff2(LBi:UBi,LBj:UBj,NT), Zt_avg1(LBi:UBi,LBj:UBj), rmask(LBi:UBi,LBj:UBj),                  &
     &           umask(LBi:UBi,LBj:UBj), vmask(LBi:UBi,LBj:UBj), pmask(LBi:UBi,LBj:UBj) )
      allocate ( Hz(LBi:UBi,LBj:UBj,N), z_r(LBi:UBi,LBj:UBj,N), z_w(LBi:UBi,LBj:UBj,0:N),                     &
     &           zeta(LBi:UBi,LBj:UBj,3), ubar(LBi:UBi,LBj:UBj,3), vbar(LBi:UBi,LBj:UBj,3),                    &
     &           u(LBi:UBi,LBj:UBj,N,2), v(LBi:UBi,LBj:UBj,N,2), t(LBi:UBi,LBj:UBj,N,3,NT),                    &
     &           Akv(LBi:UBi,LBj:UBj,0:N), Akt(LBi:UBi,LBj:UBj,0:N,NAT) )
      h=0.0_r8; f=0.0_r8; fomn=0.0_r8; pm=0.0_r8; pn=0.0_r8; om_r=0.0_r8; on_r=0.0_r8; om_u=0.0_r8
      on_u=0.0_r8; om_v=0.0_r8; on_v=0.0_r8; om_p=0.0_r8; on_p=0.0_r8; omn=0.0_r8; pmon_r=0.0_r8
      pnom_r=0.0_r8; pmon_p=0.0_r8; pnom_p=0.0_r8; pmon_u=0.0_r8; pnom_u=0.0_r8; pmon_v=0.0_r8
      pnom_v=0.0_r8; dmde=0.0_r8; dndx=0.0_r8; angler=0.0_r8; xr=0.0_r8; yr=0.0_r8; lonr=0.0_r8; xp=0.0_r8; yp=0.0_r8
      latr=0.0_r8; Zt_avg1=0.0_r8; Hz=0.0_r8; z_r=0.0_r8; z_w=0.0_r8
!  the order of the reference's set_grid (Utility/set_grid.F) and initial (Nonlinear/initial.F:293-358)
      CALL analytic_grid (ierr)
      IF (ierr.ne.0) RETURN
      CALL vertical_coordinate (ierr)
      IF (ierr.ne.0) RETURN
      CALL barotropic_filter ()
      CALL analytic_masks ()
      CALL grid_metrics ()
      CALL ini_mixing ()
      CALL level_depths (Zt_avg1)               ! Zt_avg1 = 0: depths of the resting ocean
      CALL initial_state ()
      IF (MyAppCPP(1:16).eq.'UPWELLING_WETDRY') CALL wetdry_depths ()     ! (UPWELLING_WETDRY_GLS ...: the same case under another header)
      END SUBROUTINE host_setup
!
!  The wetting/drying test application UPWELLING_WETDRY (oracle/ref/upwelling_wetdry.h; roms_amd/cases.py:wetdry_depth holds the
!  same numbers): depth 10 m south of row Mm/3, from there a plane beach to -0.3 m (above the still water level) at the northern
!  wall, roughened by 5 cm x MOD(7 i + 3 j, 5); initial free surface: a ridge 0.4 m high, 4 rows either side of row Mm/4.  The
!  reference would read such a bathymetry and initial state from its grid and initial NetCDF files.  (The temperature profile
!  stays the one of the UPWELLING depths, level by level.)
      SUBROUTINE wetdry_depths ()
      integer :: i, j, iw, jw, j0, js
      real(r8) :: ramp
      j0=Mm/3; js=Mm/4
      DO j=LBj,UBj
        DO i=LBi,UBi
          iw=MERGE(MODULO(i-1,Lm)+1, i, EWperiodic)
          jw=MERGE(MODULO(j-1,Mm)+1, j, NSperiodic)
          ramp=MIN(1.0_r8, MAX(0.0_r8, REAL(jw-j0,r8)/REAL(Mm+1-j0,r8)))
          h(i,j)=10.0_r8-10.3_r8*ramp+0.05_r8*REAL(MODULO(7*iw+3*jw,5),r8)
          zeta(i,j,:)=0.4_r8*MAX(0.0_r8, 1.0_r8-ABS(REAL(jw-js,r8))/4.0_r8)
        END DO
      END DO
      END SUBROUTINE wetdry_depths

!
!  Land/sea masks of a MASKING run (all water otherwise).  The reference reads them from its grid file (or
!  ana_mask.h of an application); this host has the land of its masked test case: an island of 3 x 3 cells east of
!  Lm/3 around Mm/2 and a headland two cells wide at 2 Lm/3 from the southern wall (wall row included) to Mm/4,
!  xi wrapped where periodic.  u, v masks: product of the two rho masks across the face (ana_mask.h:224-233).
!  psi mask: the slipperiness mask metrics.F:527-583 derives -- 1 with at most one land cell among the four
!  around the point, 2 (no-slip) where the two land cells share a side of it, 0 otherwise -- on the points that
!  routine computes (IstrP:IendP, JstrP:JendP and their periodic images); the outermost lines keep 0.
!
      SUBROUTINE analytic_masks ()
      integer :: i, j, iw, jw, i0, j0, i1, nl
      real(r8) :: a, b, c, d
      rmask=1.0_r8; umask=1.0_r8; vmask=1.0_r8; pmask=1.0_r8
      IF (.not.is_defined('MASKING')) RETURN
      i0=Lm/3+1; j0=Mm/2; i1=2*Lm/3+1
      DO j=LBj,UBj
        DO i=LBi,UBi
          iw=MERGE(MODULO(i-1,Lm)+1, i, EWperiodic)
          jw=MERGE(MODULO(j-1,Mm)+1, j, NSperiodic)
          IF ((iw.ge.i0.and.iw.le.i0+2.and.jw.ge.j0.and.jw.le.j0+2).or.                                       &
     &        (iw.ge.i1.and.iw.le.i1+1.and.jw.le.Mm/4)) rmask(i,j)=0.0_r8
        END DO
      END DO
      umask(LBi+1:UBi,:)=rmask(LBi:UBi-1,:)*rmask(LBi+1:UBi,:)
      vmask(:,LBj+1:UBj)=rmask(:,LBj:UBj-1)*rmask(:,LBj+1:UBj)
!  (the first line of the array in a periodic direction: its lower neighbour is the periodic image)
      IF (EWperiodic) umask(LBi,:)=rmask(LBi+Lm-1,:)*rmask(LBi,:)
      IF (NSperiodic) vmask(:,LBj)=rmask(:,LBj+Mm-1)*rmask(:,LBj)
      pmask=0.0_r8
      DO j=LBj+1,UBj
        DO i=LBi+1,UBi
          IF (MERGE(i.eq.UBi, i.lt.1.or.i.gt.Lm+1, EWperiodic)) CYCLE
          IF (MERGE(j.eq.UBj, j.lt.1.or.j.gt.Mm+1, NSperiodic)) CYCLE
          a=rmask(i-1,j); b=rmask(i,j); c=rmask(i-1,j-1); d=rmask(i,j-1)
          nl=4-NINT(a+b+c+d)
          IF (nl.le.1) THEN
            pmask(i,j)=1.0_r8
          ELSE IF (nl.eq.2.and.(a+b.eq.0.0_r8.or.c+d.eq.0.0_r8.or.a+c.eq.0.0_r8.or.b+d.eq.0.0_r8)) THEN
            pmask(i,j)=2.0_r8
          END IF
        END DO
      END DO
      END SUBROUTINE analytic_masks

      SUBROUTINE host_free ()
      deallocate ( weight, sc_r, Cs_r, sc_w, Cs_w )
      deallocate ( h, f, fomn, pm, pn, om_r, on_r, om_u, on_u, om_v, on_v, om_p, on_p, omn, pmon_r, pnom_r,    &
     &             pmon_p, pnom_p, pmon_u, pnom_u, pmon_v, pnom_v, dmde, dndx, angler, xr, yr, xp, yp, lonr, latr, &
     &             rdrag, rdrag2, visc2_r, visc2_p, diff2, Zt_avg1, Hz, z_r, z_w, zeta, ubar, vbar, u, v, t,  &
     &             Akv, Akt, rmask, umask, vmask, pmask )
      END SUBROUTINE host_free
!
!=======================================================================
!  Device context: fill roms_hip_config, upload the state, run the tail of "initial".
!=======================================================================
!
!  lbc and the nudging time scales (1/s) of the radiation + nudging conditions, as inp_par.F:696-752 derives them from
!  ZNUDG, M2NUDG, M3NUDG, TNUDG (days) and OBCFAC: on the edges whose condition nudges, zero elsewhere
      SUBROUTINE boundary_config (cfg)
      TYPE (roms_hip_config), intent(inout) :: cfg
      integer :: e, itrc
      real(dp) :: Zn, M2n, M3n, Tn(ROMS_MAXT)
      cfg%lbc=lbc
      Zn=0.0_dp; M2n=0.0_dp; M3n=0.0_dp; Tn=0.0_dp
      IF (Znudg.gt.0.0_dp) Zn=1.0_dp/(Znudg*86400.0_dp)
      IF (M2nudg.gt.0.0_dp) M2n=1.0_dp/(M2nudg*86400.0_dp)
      IF (M3nudg.gt.0.0_dp) M3n=1.0_dp/(M3nudg*86400.0_dp)
      DO itrc=1,NT
        IF (Tnudg(itrc).gt.0.0_dp) Tn(itrc)=1.0_dp/(Tnudg(itrc)*86400.0_dp)
      END DO
      cfg%FSobc_in=0.0_dp; cfg%FSobc_out=0.0_dp; cfg%M2obc_in=0.0_dp; cfg%M2obc_out=0.0_dp
      cfg%M3obc_in=0.0_dp; cfg%M3obc_out=0.0_dp; cfg%Tobc_in=0.0_dp; cfg%Tobc_out=0.0_dp
      DO e=1,4
        IF (lbc(1,e).eq.ROMS_LBC_RADNUD) THEN
          cfg%FSobc_out(e)=Zn; cfg%FSobc_in(e)=obcfac*Zn
        END IF
        IF (lbc(2,e).eq.ROMS_LBC_RADNUD.or.lbc(3,e).eq.ROMS_LBC_RADNUD) THEN
          cfg%M2obc_out(e)=M2n; cfg%M2obc_in(e)=obcfac*M2n
        END IF
        IF (lbc(4,e).eq.ROMS_LBC_RADNUD.or.lbc(5,e).eq.ROMS_LBC_RADNUD) THEN
          cfg%M3obc_out(e)=M3n; cfg%M3obc_in(e)=obcfac*M3n
        END IF
        DO itrc=1,NT
          IF (lbc(5+itrc,e).eq.ROMS_LBC_RADNUD) THEN
            cfg%Tobc_out(e,itrc)=Tn(itrc); cfg%Tobc_in(e,itrc)=obcfac*Tn(itrc)
          END IF
        END DO
      END DO
      END SUBROUTINE boundary_config

      SUBROUTINE device_init (device, tile, ierr)
      integer, intent(in) :: device, tile
      integer, intent(out) :: ierr
      TYPE (roms_hip_config) :: cfg
      integer :: i, itile, jtile, chunk, margin
      logical :: tw, te, ts, tn
      real(r8), allocatable :: gw(:,:,:)
      ierr=0
      IF (tile.lt.0.or.tile.ge.NtileI*NtileJ) THEN
        ierr=5
        RETURN
      END IF
!
!  Tile partition, tile_bounds_2d (get_bounds.F:972-1042); rank = tile = itile + jtile*NtileI.
!
      my_tile=tile
      itile=MOD(tile,NtileI)
      jtile=tile/NtileI
      chunk=(Lm+NtileI-1)/NtileI
      margin=(NtileI*chunk-Lm)/2
      tIstr=MAX(1+itile*chunk-margin,1)
      tIend=MIN(1+itile*chunk-margin+chunk-1,Lm)
      chunk=(Mm+NtileJ-1)/NtileJ
      margin=(NtileJ*chunk-Mm)/2
      tJstr=MAX(1+jtile*chunk-margin,1)
      tJend=MIN(1+jtile*chunk-margin+chunk-1,Mm)
      tw=itile.eq.0
      te=itile.eq.NtileI-1
      ts=jtile.eq.0
      tn=jtile.eq.NtileJ-1
!
!  Tile arrays: the domain's own bounds at a domain edge, otherwise three ghost lines on the low
!  side and Nghost on the high side (the periodic layout of mod_param.F:1640-1665 around the tile).
!
      tLBi=MERGE(LBi, tIstr-1-Nghost, tw)
      tUBi=MERGE(UBi, tIend+Nghost, te)
      tLBj=MERGE(LBj, tJstr-1-Nghost, ts)
      tUBj=MERGE(UBj, tJend+Nghost, tn)
      cfg%abi_version=5
      cfg%obcfac=obcfac
      cfg%volcons=volcons
      cfg%device=device
      cfg%Lm=Lm; cfg%Mm=Mm; cfg%N=N; cfg%NT=NT; cfg%NAT=NAT; cfg%Nghost=Nghost
      cfg%LBi=tLBi; cfg%UBi=tUBi; cfg%LBj=tLBj; cfg%UBj=tUBj
      cfg%NtileI=NtileI; cfg%NtileJ=NtileJ; cfg%tile=tile
      cfg%EWperiodic=MERGE(1,0,EWperiodic); cfg%NSperiodic=MERGE(1,0,NSperiodic)
      cfg%options=INT(options, c_int64_t)
!  ABI version 4: the options of the upper word -- biharmonic mixing along s-surfaces, wetting and drying (with DCRIT), the
!  momentum diagnostics beside the tracer ones (decided below)
      IF (mix4(1)) cfg%options=IOR(cfg%options, ROMS_UV_VIS4)
      IF (mix4(2)) cfg%options=IOR(cfg%options, ROMS_TS_DIF4)
      IF (wet_dry) cfg%options=IOR(cfg%options, ROMS_WET_DRY)
      IF (mix_geo_uv) cfg%options=IOR(cfg%options, ROMS_MIX_GEO_UV)
      IF (ddmix) cfg%options=IOR(cfg%options, ROMS_LMD_DDMIX)
      IF (bkpp) cfg%options=IOR(cfg%options, ROMS_LMD_BKPP)
      IF (prs4x.eq.44) cfg%options=IOR(cfg%options, ROMS_PRSGRD44)
      IF (prs4x.eq.42) cfg%options=IOR(cfg%options, ROMS_PRSGRD42)
      cfg%Dcrit=Dcrit
      IF (nDIA.gt.0.and.diag_uv.and.(ANY(DoutM2).or.ANY(DoutM3)).and.IAND(options,ROMS_PLAIN_VVISC).eq.0) THEN
        cfg%options=IOR(cfg%options, ROMS_DIAGNOSTICS_UV)
      ELSE
        diag_uv=.FALSE.                             ! (without SPLINES_VVISC the library has no such terms: tracer terms only)
      END IF
      cfg%hadv=hadv; cfg%vadv=vadv
      cfg%Istr=tIstr; cfg%Iend=tIend; cfg%Jstr=tJstr; cfg%Jend=tJend
      cfg%west_edge=MERGE(1,0,tw); cfg%east_edge=MERGE(1,0,te)
      cfg%south_edge=MERGE(1,0,ts); cfg%north_edge=MERGE(1,0,tn)
      cfg%ntfirst=1; cfg%ntstart=1; cfg%ndtfast=ndtfast; cfg%nfast=nfast; cfg%ninfo=ninfo
      cfg%dt=dt; cfg%dtfast=dtfast
      cfg%weight=0.0_dp
      DO i=1,2*ndtfast
        cfg%weight(i,1)=weight(1,i)
        cfg%weight(i,2)=weight(2,i)
      END DO
      cfg%rho0=rho0; cfg%g=g; cfg%lambda=1.0_dp; cfg%gamma2=gamma2; cfg%Cp=Cp
      cfg%R0=R0; cfg%T0=T0; cfg%S0=S0; cfg%Tcoef=Tcoef; cfg%Scoef=Scoef
      cfg%hc=hc; cfg%Vtransform=Vtransform
      cfg%rdrg=rdrg; cfg%rdrg2=rdrg2; cfg%Zob=Zob
      cfg%Akt_bak=Akt_bak; cfg%Akv_bak=Akv_bak
      cfg%dstart=dstart
      cfg%blk_ZQ=blk_ZQ; cfg%blk_ZT=blk_ZT; cfg%blk_ZW=blk_ZW; cfg%lmd_Jwt=lmd_Jwt
      cfg%gls_flags=gls_flags; cfg%lbc_tke=lbc_tke
      cfg%gls_p=gls_p; cfg%gls_m=gls_m; cfg%gls_n=gls_n; cfg%gls_Kmin=gls_Kmin; cfg%gls_Pmin=gls_Pmin
      cfg%gls_cmu0=gls_cmu0; cfg%gls_c1=gls_c1; cfg%gls_c2=gls_c2; cfg%gls_c3m=gls_c3m; cfg%gls_c3p=gls_c3p
      cfg%gls_sigk=gls_sigk; cfg%gls_sigp=gls_sigp; cfg%Akk_bak=Akk_bak; cfg%Akp_bak=Akp_bak
      cfg%Zos=Zos; cfg%charnok_alpha=charnok_alpha; cfg%crgban_cw=crgban_cw
      cfg%sc_r=0.0_dp; cfg%Cs_r=0.0_dp; cfg%sc_w=0.0_dp; cfg%Cs_w=0.0_dp
      CALL boundary_config (cfg)
      cfg%sc_r(1:N)=sc_r; cfg%Cs_r(1:N)=Cs_r; cfg%sc_w(0:N)=sc_w; cfg%Cs_w(0:N)=Cs_w
      ierr=roms_hip_create(cfg, ctx)
      IF (ierr.ne.0) RETURN
      IF (nAVG.gt.0.and.ANY(Aout)) THEN           ! AVERAGES: mod_average.F allocate_average
        ierr=roms_hip_avg_config(ctx, nAVG, ntsAVG, 0, 1, aout_mask())
        IF (ierr.ne.0) RETURN
      END IF
      IF (nDIA.gt.0) THEN                          ! DIAGNOSTICS_TS (and _UV: the option bit above): mod_diags.F allocate_diags
        ierr=roms_hip_dia_config(ctx, nDIA, ntsDIA, 0, 1)
        IF (ierr.ne.0) RETURN
      END IF
      CALL up ('h', h, 1, ierr); CALL up ('f', f, 1, ierr); CALL up ('fomn', fomn, 1, ierr)
      CALL up ('pm', pm, 1, ierr); CALL up ('pn', pn, 1, ierr); CALL up ('om_r', om_r, 1, ierr)
      CALL up ('on_r', on_r, 1, ierr); CALL up ('om_u', om_u, 1, ierr); CALL up ('on_u', on_u, 1, ierr)
      CALL up ('om_v', om_v, 1, ierr); CALL up ('on_v', on_v, 1, ierr); CALL up ('om_p', om_p, 1, ierr)
      CALL up ('on_p', on_p, 1, ierr); CALL up ('omn', omn, 1, ierr); CALL up ('pmon_r', pmon_r, 1, ierr)
      CALL up ('pnom_r', pnom_r, 1, ierr); CALL up ('pmon_p', pmon_p, 1, ierr)
      CALL up ('pnom_p', pnom_p, 1, ierr); CALL up ('pmon_u', pmon_u, 1, ierr)
      CALL up ('pnom_u', pnom_u, 1, ierr); CALL up ('pmon_v', pmon_v, 1, ierr)
      CALL up ('pnom_v', pnom_v, 1, ierr); CALL up ('dmde', dmde, 1, ierr); CALL up ('dndx', dndx, 1, ierr)
      CALL up ('angler', angler, 1, ierr); CALL up ('xr', xr, 1, ierr); CALL up ('yr', yr, 1, ierr)
      CALL up ('xp', xp, 1, ierr); CALL up ('yp', yp, 1, ierr)
      CALL up ('lonr', lonr, 1, ierr); CALL up ('latr', latr, 1, ierr); CALL up ('rdrag', rdrag, 1, ierr)
      CALL up ('rdrag2', rdrag2, 1, ierr); CALL up ('visc2_r', visc2_r, 1, ierr)
      CALL up ('visc2_p', visc2_p, 1, ierr); CALL up ('diff2', diff2, NT, ierr)
      IF (ANY(mix4)) CALL up_mix4 (ierr)
      CALL up ('Zt_avg1', Zt_avg1, 1, ierr)
      IF (is_defined('MASKING')) THEN
        CALL up ('rmask', rmask, 1, ierr); CALL up ('umask', umask, 1, ierr)
        CALL up ('vmask', vmask, 1, ierr); CALL up ('pmask', pmask, 1, ierr)
      END IF
      CALL up ('Hz', Hz, N, ierr); CALL up ('z_r', z_r, N, ierr); CALL up ('z_w', z_w, N+1, ierr)
      CALL up ('zeta', zeta, 3, ierr); CALL up ('ubar', ubar, 3, ierr); CALL up ('vbar', vbar, 3, ierr)
      CALL up ('u', u, 2*N, ierr); CALL up ('v', v, 2*N, ierr); CALL up ('t', t, 3*N*NT, ierr)
      CALL up ('Akv', Akv, N+1, ierr); CALL up ('Akt', Akt, (N+1)*NAT, ierr)
      IF (IAND(options,IOR(ROMS_GLS_MIXING,ROMS_MY25_MIXING)).ne.0) THEN       ! initialize_mixing, mod_mixing.F:1490-1515
        allocate ( gw(LBi:UBi,LBj:UBj,3*(N+1)) )
        gw=gls_Kmin
        CALL up ('tke', gw, 3*(N+1), ierr)
        gw=gls_Pmin
        CALL up ('gls', gw, 3*(N+1), ierr)
        gw=0.0_r8
        CALL up ('Lscale', gw(:,:,1:N+1), N+1, ierr)
        gw(:,:,2:N)=Akk_bak
        CALL up ('Akk', gw(:,:,1:N+1), N+1, ierr)
        gw(:,:,2:N)=Akp_bak
        CALL up ('Akp', gw(:,:,1:N+1), N+1, ierr)
        deallocate ( gw )
      END IF
      IF (wet_dry.and.ierr.eq.0) ierr=roms_hip_wetdry_ini(ctx)       ! the initial wet/dry masks from zeta(1) (initial.F:467)
      END SUBROUTINE device_init
!
!  Upload the tile's window (tLBi:tUBi,tLBj:tUBj) of a host array with np horizontal planes.
!
      INTEGER FUNCTION aout_mask ()
      integer :: k
      aout_mask=0
      DO k=0,nAout-1
        IF (Aout(k)) aout_mask=IBSET(aout_mask,k)
      END DO
      END FUNCTION aout_mask

!  UV_VIS4 / TS_DIF4: visc4_r = visc4_p = SQRT(ABS(VISC4)), diff4(:,:,itrc) = SQRT(ABS(TNU4(itrc))) everywhere (inp_par.F:634,
!  read_phypar.F:7840, ini_hmixcoef.F:270-296: uniform coefficients, no VISC_GRID / DIFF_GRID scaling)
      SUBROUTINE up_mix4 (ierr)
      integer, intent(inout) :: ierr
      real(r8), allocatable :: W(:,:,:)
      integer :: itrc
      allocate ( W(LBi:UBi,LBj:UBj,NT) )
      W(:,:,1)=SQRT(ABS(visc4))
      CALL up ('visc4_r', W(:,:,1:1), 1, ierr)
      CALL up ('visc4_p', W(:,:,1:1), 1, ierr)
      DO itrc=1,NT
        W(:,:,itrc)=SQRT(ABS(tnu4(itrc)))
      END DO
      CALL up ('diff4', W, NT, ierr)
      deallocate ( W )
      END SUBROUTINE up_mix4

      SUBROUTINE up (name, A, np, ierr)
      character(len=*), intent(in) :: name
      integer, intent(in) :: np
      real(r8), intent(in) :: A(LBi:UBi,LBj:UBj,np)
      integer, intent(inout) :: ierr
      real(r8), allocatable :: win(:,:,:)
      integer(c_long) :: n
      integer :: r
      IF (ierr.ne.0) RETURN
      allocate ( win(tLBi:tUBi,tLBj:tUBj,np) )
      win=A(tLBi:tUBi,tLBj:tUBj,:)
      n=roms_hip_field_size(ctx, TRIM(name)//c_null_char)
      IF (n.ne.SIZE(win,KIND=c_long)) THEN
        ierr=8
      ELSE
        r=roms_hip_upload(ctx, TRIM(name)//c_null_char, win, n)
        IF (r.ne.0) ierr=r
      END IF
      deallocate ( win )
      END SUBROUTINE up
!
!  Tail of "initial" on the device: set_massflux, omega, rho_eos (initial.F:562-577).  In a multi-tile
!  run the halo transport (roms_hip_comm_rccl / roms_hip_set_exchange) must be installed first.
!
      SUBROUTINE device_start (ierr)
      integer, intent(out) :: ierr
      ierr=roms_hip_start(ctx)
      END SUBROUTINE device_start
!
!=======================================================================
!  main3d, Nonlinear/main3d.F:216-1148, kernel by kernel through the C ABI (the seam a maintainer
!  would use inside the reference).  host_run uses the fused roms_hip_main3d instead.
!=======================================================================
!
      SUBROUTINE main3d_kernels (nsteps, ierr)
      integer, intent(in) :: nsteps
      integer, intent(out) :: ierr
      integer :: istep, my_iif, next_indx1
      ierr=roms_hip_get_stepping(ctx, step)
      DO istep=1,nsteps
        step%nstp=1+MOD(step%iic-1,2)
        step%nnew=3-step%nstp
        step%nrhs=step%nstp
        ierr=roms_hip_set_stepping(ctx, step)
        ierr=roms_hip_set_data(ctx);                      IF (ierr.ne.0) RETURN
        IF (step%iic.eq.1) THEN                            ! post_initial
          ierr=roms_hip_ini_zeta(ctx);                    IF (ierr.ne.0) RETURN
          ierr=roms_hip_set_depth(ctx);                   IF (ierr.ne.0) RETURN
          ierr=roms_hip_ini_fields(ctx);                  IF (ierr.ne.0) RETURN
        END IF
        ierr=roms_hip_set_massflux(ctx);                  IF (ierr.ne.0) RETURN
        ierr=roms_hip_rho_eos(ctx);                       IF (ierr.ne.0) RETURN
        IF (IAND(options,ROMS_BULK_FLUXES).ne.0) ierr=roms_hip_bulk_flux(ctx)
        IF (ierr.ne.0) RETURN
        ierr=roms_hip_set_vbc(ctx);                       IF (ierr.ne.0) RETURN
        IF (IAND(options,ROMS_ANA_VMIX).ne.0) THEN
          ierr=roms_hip_ana_vmix(ctx)
        ELSE IF (IAND(options,ROMS_LMD_MIXING).ne.0) THEN
          ierr=roms_hip_lmd_vmix(ctx)
        END IF
        IF (ierr.ne.0) RETURN
        ierr=roms_hip_omega(ctx);                         IF (ierr.ne.0) RETURN
        ierr=roms_hip_wvelocity(ctx, step%nstp);          IF (ierr.ne.0) RETURN
        ierr=roms_hip_set_zeta(ctx);                      IF (ierr.ne.0) RETURN
        ierr=roms_hip_set_avg(ctx);                       IF (ierr.ne.0) RETURN      ! :562 (AVERAGES; no-op when off)
        ierr=roms_hip_rhs3d(ctx);                         IF (ierr.ne.0) RETURN
        IF (IAND(options,IOR(ROMS_GLS_MIXING,ROMS_MY25_MIXING)).ne.0) ierr=roms_hip_gls_prestep(ctx)         ! :634-636
        IF (ierr.ne.0) RETURN
        DO my_iif=1,nfast+1                                ! LF-AM3 barotropic loop :810-918
          next_indx1=3-step%indx1
          IF (step%predictor.eq.0) THEN
            step%predictor=1
            step%iif=my_iif
            IF (step%iif.eq.1) THEN
              step%kstp=step%indx1
            ELSE
              step%kstp=3-step%indx1
            END IF
            step%knew=3
            step%krhs=step%indx1
          END IF
          ierr=roms_hip_set_stepping(ctx, step)
          ierr=roms_hip_step2d(ctx);                      IF (ierr.ne.0) RETURN
          IF (step%predictor.ne.0) THEN
            step%predictor=0
            step%knew=next_indx1
            step%kstp=3-step%knew
            step%krhs=3
            IF (step%iif.lt.(nfast+1)) step%indx1=next_indx1
          END IF
          ierr=roms_hip_set_stepping(ctx, step)
          IF (step%iif.lt.(nfast+1)) THEN
            ierr=roms_hip_step2d(ctx);                    IF (ierr.ne.0) RETURN
          END IF
        END DO
        ierr=roms_hip_set_depth(ctx);                     IF (ierr.ne.0) RETURN
        ierr=roms_hip_step3d_uv(ctx);                     IF (ierr.ne.0) RETURN
        ierr=roms_hip_omega(ctx);                         IF (ierr.ne.0) RETURN
        IF (IAND(options,IOR(ROMS_GLS_MIXING,ROMS_MY25_MIXING)).ne.0) ierr=roms_hip_gls_corstep(ctx)         ! :1019-1021
        IF (ierr.ne.0) RETURN
        ierr=roms_hip_step3d_t(ctx);                      IF (ierr.ne.0) RETURN
        step%iic=step%iic+1
        step%time=step%time+dt
        ierr=roms_hip_set_stepping(ctx, step)
      END DO
      END SUBROUTINE main3d_kernels

      END MODULE roms_host
