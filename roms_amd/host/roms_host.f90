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
      real(dp) :: visc2 = 5.0_dp, tnu2(ROMS_MAXT) = 0.0_dp, Akt_bak(ROMS_MAXT) = 1.0E-6_dp, Akv_bak = 1.0E-5_dp
      real(dp) :: rdrg = 3.0E-4_dp, rdrg2 = 3.0E-3_dp, Zob = 0.02_dp, Zos = 0.02_dp, gamma2 = 1.0_dp
      real(dp) :: dstart = 0.0_dp, time_ref = 0.0_dp, blk_ZQ = 10.0_dp, blk_ZT = 10.0_dp, blk_ZW = 10.0_dp
      integer :: options = 0

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
      real(r8), allocatable, target :: visc2_r(:,:), visc2_p(:,:), diff2(:,:,:)
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
      open (newunit=iu, file=TRIM(fname), status='old', action='read', iostat=ios)
      IF (ios.ne.0) THEN
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
          CASE ('LBC(isFsur)')
            EWperiodic=(tok(1)(1:3).eq.'Per')
            IF (nv.ge.2) NSperiodic=(tok(2)(1:3).eq.'Per')
          CASE ('TNU2');        CALL load_r (tok, nv, tnu2)
          CASE ('VISC2');       visc2=toreal(tok(1))
          CASE ('AKT_BAK');     CALL load_r (tok, nv, Akt_bak)
          CASE ('AKV_BAK');     Akv_bak=toreal(tok(1))
          CASE ('RDRG');        rdrg=toreal(tok(1))
          CASE ('RDRG2');       rdrg2=toreal(tok(1))
          CASE ('Zob');         Zob=toreal(tok(1))
          CASE ('Zos');         Zos=toreal(tok(1))
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
          CASE DEFAULT
        END SELECT
      END DO
      close (iu)
      END SUBROUTINE read_roms_in

      SUBROUTINE set_defaults ()          ! roms_upwelling.in values, so that a partial file is usable
      MyAppCPP='UPWELLING'
      Lm=41; Mm=80; N=16; NAT=2; NT=2; NtileI=1; NtileJ=1
      ntimes=100; ndtfast=30; ninfo=1; Vtransform=2; Vstretching=4; lmd_Jwt=1
      hadv=ROMS_U3; vadv=ROMS_C4
      EWperiodic=.TRUE.; NSperiodic=.FALSE.
      dt=300.0_dp; theta_s=3.0_dp; theta_b=0.0_dp; Tcline=25.0_dp; rho0=1025.0_dp
      R0=1027.0_dp; T0=14.0_dp; S0=35.0_dp; Tcoef=1.7E-4_dp; Scoef=0.0_dp
      visc2=5.0_dp; tnu2=0.0_dp; Akt_bak=1.0E-6_dp; Akv_bak=1.0E-5_dp
      rdrg=3.0E-4_dp; rdrg2=3.0E-3_dp; Zob=0.02_dp; Zos=0.02_dp; gamma2=1.0_dp
      dstart=0.0_dp; time_ref=0.0_dp; blk_ZQ=10.0_dp; blk_ZT=10.0_dp; blk_ZW=10.0_dp
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

      SUBROUTINE load_tadv (tok, nv, adv)       ! load_tadv, Utility/inp_decode.F
      character(len=64), intent(in) :: tok(16)
      integer, intent(in) :: nv
      integer, intent(inout) :: adv(:)
      integer :: k
      DO k=1,MIN(nv,SIZE(adv))
        SELECT CASE (TRIM(tok(k)))
          CASE ('A4');     adv(k)=ROMS_A4
          CASE ('C2');     adv(k)=ROMS_C2
          CASE ('C4');     adv(k)=ROMS_C4
          CASE ('HSIMT');  adv(k)=ROMS_HSIMT
          CASE ('MPDATA'); adv(k)=ROMS_MPDATA
          CASE ('SP');     adv(k)=ROMS_SPLINES
          CASE ('SU3');    adv(k)=ROMS_SPLIT_U3
          CASE ('U3');     adv(k)=ROMS_U3
        END SELECT
      END DO
      END SUBROUTINE load_tadv
!
!=======================================================================
!  cpp options of the application header (ROMS/Include/upwelling.h, benchmark.h + globaldefs.h)
!=======================================================================
!
      SUBROUTINE set_cppdefs (ierr)
      integer, intent(out) :: ierr
      ierr=0
      SELECT CASE (TRIM(MyAppCPP))
        CASE ('UPWELLING')
          options=ROMS_UV_ADV+ROMS_UV_COR+ROMS_UV_VIS2+ROMS_TS_DIF2+ROMS_ANA_VMIX+ROMS_SALINITY+              &
     &            ROMS_APP_UPWELLING
        CASE ('UPWELLING_KPP')
!  UPWELLING with the KPP closure instead of ANA_VMIX: the "custom application header" of BASELINE
!  config 5 (upwelling.h with LMD_MIXING, LMD_RIMIX, LMD_CONVEC, LMD_SKPP, LMD_NONLOCAL, RI_SPLINES,
!  SOLAR_SOURCE and ANA_SRFLUX (zero) in place of ANA_VMIX).
          options=ROMS_UV_ADV+ROMS_UV_COR+ROMS_UV_VIS2+ROMS_TS_DIF2+ROMS_LMD_MIXING+ROMS_SOLAR_SOURCE+         &
     &            ROMS_SALINITY+ROMS_APP_UPWELLING
        CASE ('BENCHMARK')
          options=ROMS_UV_ADV+ROMS_UV_COR+ROMS_UV_VIS2+ROMS_TS_DIF2+ROMS_MIX_GEO_TS+ROMS_CURVGRID+            &
     &            ROMS_NONLIN_EOS+ROMS_UV_QDRAG+ROMS_LMD_MIXING+ROMS_BULK_FLUXES+ROMS_SOLAR_SOURCE+            &
     &            ROMS_SALINITY+ROMS_SPHERICAL+ROMS_APP_BENCHMARK
        CASE DEFAULT
          ierr=5                        ! unknown application (checkdefs.F would stop too)
      END SELECT
      END SUBROUTINE set_cppdefs
!
!=======================================================================
!  Partition and array bounds (single tile = the whole domain on one GPU):
!  get_bounds.F:232-283, var_bounds :1044-1884, inp_par.F:210-226 (ghost points),
!  mod_param.F:1633-1636 (padding).
!=======================================================================
!
      SUBROUTINE set_bounds ()
      IF (ANY(hadv(1:NT).eq.ROMS_MPDATA).or.ANY(hadv(1:NT).eq.ROMS_HSIMT)) THEN
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
!  set_scoord, Utility/set_scoord.F (Vstretching 4: A. Shchepetkin 2010 double stretching)
!=======================================================================
!
      SUBROUTINE set_scoord (ierr)
      integer, intent(out) :: ierr
      integer :: k
      real(dp) :: Cbot, Csur, ds, scr, scw
      ierr=0
      IF (Vtransform.eq.1) THEN
        hc=MIN(hmin,Tcline)
      ELSE
        hc=Tcline
      END IF
      IF (Vstretching.ne.4) THEN
        ierr=5
        RETURN
      END IF
      ds=1.0_dp/REAL(N,dp)
      sc_w(N)=0.0_dp
      Cs_w(N)=0.0_dp
      DO k=N-1,1,-1
        scw=ds*REAL(k-N,dp)
        sc_w(k)=scw
        IF (theta_s.gt.0.0_dp) THEN
          Csur=(1.0_dp-COSH(theta_s*scw))/(COSH(theta_s)-1.0_dp)
        ELSE
          Csur=-scw**2
        END IF
        IF (theta_b.gt.0.0_dp) THEN
          Cbot=(EXP(theta_b*Csur)-1.0_dp)/(1.0_dp-EXP(-theta_b))
          Cs_w(k)=Cbot
        ELSE
          Cs_w(k)=Csur
        END IF
      END DO
      sc_w(0)=-1.0_dp
      Cs_w(0)=-1.0_dp
      DO k=1,N
        scr=ds*(REAL(k-N,dp)-0.5_dp)
        sc_r(k)=scr
        IF (theta_s.gt.0.0_dp) THEN
          Csur=(1.0_dp-COSH(theta_s*scr))/(COSH(theta_s)-1.0_dp)
        ELSE
          Csur=-scr**2
        END IF
        IF (theta_b.gt.0.0_dp) THEN
          Cbot=(EXP(theta_b*Csur)-1.0_dp)/(1.0_dp-EXP(-theta_b))
          Cs_r(k)=Cbot
        ELSE
          Cs_r(k)=Csur
        END IF
      END DO
      END SUBROUTINE set_scoord
!
!=======================================================================
!  set_weights, Utility/set_weights.F:56-230: power-law barotropic filter (r16 accumulators, see mod_kinds.F:50)
!=======================================================================
!
      SUBROUTINE set_weights ()
      integer :: i, j, iter
      real(dp) :: gamma, scale
      real(r16) :: wsum, shift, cff
      nfast=0
      weight=0.0_dp
      scale=(Falpha+1.0_dp)*(Falpha+Fbeta+1.0_dp)/                                                            &
     &      ((Falpha+2.0_dp)*(Falpha+Fbeta+2.0_dp)*REAL(ndtfast,dp))
      gamma=Fgamma*MAX(0.0_dp, 1.0_dp-10.0_dp/REAL(ndtfast,dp))
      DO iter=1,16
        nfast=0
        DO i=1,2*ndtfast
          cff=scale*REAL(i,dp)
          weight(1,i)=cff**Falpha-cff**(Falpha+Fbeta)-gamma*cff
          IF (weight(1,i).gt.0.0_dp) nfast=i
          IF ((nfast.gt.0).and.(weight(1,i).lt.0.0_dp)) THEN
            weight(1,i)=0.0_dp
          END IF
        END DO
        wsum=0.0_r16
        shift=0.0_r16
        DO i=1,nfast
          wsum=wsum+weight(1,i)
          shift=shift+weight(1,i)*REAL(i,dp)
        END DO
        scale=scale*shift/(wsum*REAL(ndtfast,dp))
      END DO
      DO iter=1,ndtfast
        wsum=0.0_r16
        shift=0.0_r16
        DO i=1,nfast
          wsum=wsum+weight(1,i)
          shift=shift+REAL(i,dp)*weight(1,i)
        END DO
        shift=shift/wsum
        cff=REAL(ndtfast,dp)-shift
        IF (cff.gt.1.0_r16) THEN
          nfast=nfast+1
          DO i=nfast,2,-1
            weight(1,i)=weight(1,i-1)
          END DO
          weight(1,1)=0.0_dp
        ELSE IF (cff.gt.0.0_r16) THEN
          wsum=1.0_r16-cff
          DO i=nfast,2,-1
            weight(1,i)=wsum*weight(1,i)+cff*weight(1,i-1)
          END DO
          weight(1,1)=wsum*weight(1,1)
        ELSE IF (cff.lt.-1.0_r16) THEN
          nfast=nfast-1
          DO i=1,nfast,+1
            weight(1,i)=weight(1,i+1)
          END DO
          weight(1,nfast+1)=0.0_dp
        ELSE IF (cff.lt.0.0_r16) THEN
          wsum=1.0_r16+cff
          DO i=1,nfast-1,+1
            weight(1,i)=wsum*weight(1,i)-cff*weight(1,i+1)
          END DO
          weight(1,nfast)=wsum*weight(1,nfast)
        END IF
      END DO
      DO j=1,nfast
        cff=weight(1,j)
        DO i=1,j
          weight(2,i)=weight(2,i)+cff
        END DO
      END DO
      wsum=0.0_r16
      cff=0.0_r16
      DO i=1,nfast
        wsum=wsum+weight(1,i)
        cff=cff+weight(2,i)
      END DO
      wsum=1.0_r16/wsum
      cff=1.0_r16/cff
      DO i=1,nfast
        weight(1,i)=wsum*weight(1,i)
        weight(2,i)=cff*weight(2,i)
      END DO
      END SUBROUTINE set_weights
!
!=======================================================================
!  ana_grid, Functionals/ana_grid.h: UPWELLING :1058-1082 (+ Cartesian set-up), BENCHMARK
!  :462-482,677-690,931-936 (spherical)
!=======================================================================
!
      SUBROUTINE ana_grid (ierr)
      integer, intent(out) :: ierr
      integer :: i, j
      real(r8) :: Esize, Xsize, beta, cff, depth, dx, dy, f0, val1, val2
      real(r8), allocatable :: wrkX(:,:), wrkY(:,:)
      ierr=0
      allocate ( wrkX(LBi:UBi,LBj:UBj), wrkY(LBi:UBi,LBj:UBj) )
      IF (IAND(options,ROMS_APP_UPWELLING).ne.0) THEN
        Xsize=1000.0_r8*REAL(Lm,r8)
        Esize=1000.0_r8*REAL(Mm,r8)
        depth=150.0_r8
        f0=-8.26E-05_r8
        beta=0.0_r8
        xl=Xsize
        el=Esize
        dx=Xsize/REAL(Lm,r8)
        dy=Esize/REAL(Mm,r8)
        DO j=Jstr-1,Jend+1
          DO i=Istr-1,Iend+1
            xr(i,j)=dx*(REAL(i-1,r8)+0.5_r8)
            yr(i,j)=dy*(REAL(j-1,r8)+0.5_r8)
          END DO
        END DO
        DO j=MIN(JstrT,Jstr-1),MAX(Jend+1,JendT)
          DO i=MIN(IstrT,Istr-1),MAX(Iend+1,IendT)
            wrkX(i,j)=1.0_r8/dx
            wrkY(i,j)=1.0_r8/dy
          END DO
        END DO
        DO j=JstrT,JendT
          DO i=IstrT,IendT
            pm(i,j)=wrkX(i,j)
            pn(i,j)=wrkY(i,j)
            angler(i,j)=0.0_r8
            f(i,j)=f0
          END DO
        END DO
        IF (NSperiodic) THEN
          DO i=IstrT,IendT
            IF (i.le.Lm/2) THEN
              val1=REAL(i,r8)
            ELSE
              val1=REAL(Lm+1-i,r8)
            END IF
            val2=MIN(depth,84.5_r8+66.526_r8*TANH((val1-10.0_r8)/7.0_r8))
            DO j=JstrT,JendT
              h(i,j)=val2
            END DO
          END DO
        ELSE IF (EWperiodic) THEN
          DO j=JstrT,JendT
            IF (j.le.Mm/2) THEN
              val1=REAL(j,r8)
            ELSE
              val1=REAL(Mm+1-j,r8)
            END IF
            val2=MIN(depth,84.5_r8+66.526_r8*TANH((val1-10.0_r8)/7.0_r8))
            DO i=IstrT,IendT
              h(i,j)=val2
            END DO
          END DO
        END IF
      ELSE IF (IAND(options,ROMS_APP_BENCHMARK).ne.0) THEN
        Xsize=360.0_r8
        Esize=20.0_r8
        xl=Xsize
        el=Esize
        dx=Xsize/REAL(Lm,r8)
        dy=Esize/REAL(Mm,r8)
        DO j=Jstr-1,Jend+1
          val1=-70.0_r8+dy*(REAL(j,r8)-0.5_r8)
          DO i=Istr-1,Iend+1
            lonr(i,j)=dx*(REAL(i,r8)-0.5_r8)
            latr(i,j)=val1
          END DO
        END DO
        val1=REAL(Lm,r8)/(2.0_r8*pi*Eradius)
        val2=REAL(Mm,r8)*360.0_r8/(2.0_r8*pi*Eradius*Esize)
        DO j=MIN(JstrT,Jstr-1),MAX(Jend+1,JendT)
          cff=1.0_r8/COS((-70.0_r8+dy*(REAL(j,r8)-0.5_r8))*deg2rad)
          DO i=MIN(IstrT,Istr-1),MAX(Iend+1,IendT)
            wrkX(i,j)=val1*cff
            wrkY(i,j)=val2
          END DO
        END DO
        DO j=JstrT,JendT
          DO i=IstrT,IendT
            pm(i,j)=wrkX(i,j)
            pn(i,j)=wrkY(i,j)
          END DO
        END DO
        DO j=Jstr,Jend
          DO i=Istr,Iend
            dndx(i,j)=0.5_r8*((1.0_r8/wrkY(i+1,j  ))-(1.0_r8/wrkY(i-1,j  )))
            dmde(i,j)=0.5_r8*((1.0_r8/wrkX(i  ,j+1))-(1.0_r8/wrkX(i  ,j-1)))
          END DO
        END DO
        CALL exchange2d (dndx, 'r')
        CALL exchange2d (dmde, 'r')
        val1=2.0_r8*(2.0_r8*pi*366.25_r8/365.25_r8)/86400.0_r8
        DO j=JstrT,JendT
          DO i=IstrT,IendT
            angler(i,j)=0.0_r8
            f(i,j)=val1*SIN(latr(i,j)*deg2rad)
            h(i,j)=500.0_r8+1750.0_r8*(1.0+TANH((68.0_r8+latr(i,j))/dy))
          END DO
        END DO
      ELSE
        ierr=5
        RETURN
      END IF
      hmin=MINVAL(h(IstrT:IendT,JstrT:JendT))
      hmax=MAXVAL(h(IstrT:IendT,JstrT:JendT))
      CALL exchange2d (pm, 'r')
      CALL exchange2d (pn, 'r')
      CALL exchange2d (angler, 'r')
      CALL exchange2d (f, 'r')
      CALL exchange2d (h, 'r')
      deallocate ( wrkX, wrkY )
      END SUBROUTINE ana_grid
!
!=======================================================================
!  metrics, Utility/metrics.F:23-250
!=======================================================================
!
      SUBROUTINE metrics ()
      integer :: i, j
      DO j=JstrT,JendT
        DO i=IstrT,IendT
          om_r(i,j)=1.0_r8/pm(i,j)
          on_r(i,j)=1.0_r8/pn(i,j)
          omn(i,j)=1.0_r8/(pm(i,j)*pn(i,j))
          fomn(i,j)=f(i,j)*omn(i,j)
          pnom_r(i,j)=pn(i,j)/pm(i,j)
          pmon_r(i,j)=pm(i,j)/pn(i,j)
        END DO
      END DO
      CALL exchange2d (om_r, 'r')
      CALL exchange2d (on_r, 'r')
      CALL exchange2d (omn, 'r')
      CALL exchange2d (fomn, 'r')
      CALL exchange2d (pnom_r, 'r')
      CALL exchange2d (pmon_r, 'r')
      DO j=JstrT,JendT
        DO i=IstrP,IendT
          pmon_u(i,j)=(pm(i-1,j)+pm(i,j))/(pn(i-1,j)+pn(i,j))
          pnom_u(i,j)=(pn(i-1,j)+pn(i,j))/(pm(i-1,j)+pm(i,j))
          om_u(i,j)=2.0_r8/(pm(i-1,j)+pm(i,j))
          on_u(i,j)=2.0_r8/(pn(i-1,j)+pn(i,j))
        END DO
      END DO
      CALL exchange2d (pmon_u, 'u')
      CALL exchange2d (pnom_u, 'u')
      CALL exchange2d (om_u, 'u')
      CALL exchange2d (on_u, 'u')
      DO j=JstrP,JendT
        DO i=IstrT,IendT
          pmon_v(i,j)=(pm(i,j-1)+pm(i,j))/(pn(i,j-1)+pn(i,j))
          pnom_v(i,j)=(pn(i,j-1)+pn(i,j))/(pm(i,j-1)+pm(i,j))
          om_v(i,j)=2.0_r8/(pm(i,j-1)+pm(i,j))
          on_v(i,j)=2.0_r8/(pn(i,j-1)+pn(i,j))
        END DO
      END DO
      CALL exchange2d (pmon_v, 'v')
      CALL exchange2d (pnom_v, 'v')
      CALL exchange2d (om_v, 'v')
      CALL exchange2d (on_v, 'v')
      DO j=JstrP,JendT
        DO i=IstrP,IendT
          pnom_p(i,j)=(pn(i-1,j-1)+pn(i-1,j)+pn(i,j-1)+pn(i,j))/(pm(i-1,j-1)+pm(i-1,j)+pm(i,j-1)+pm(i,j))
          pmon_p(i,j)=(pm(i-1,j-1)+pm(i-1,j)+pm(i,j-1)+pm(i,j))/(pn(i-1,j-1)+pn(i-1,j)+pn(i,j-1)+pn(i,j))
          om_p(i,j)=4.0_r8/(pm(i-1,j-1)+pm(i-1,j)+pm(i,j-1)+pm(i,j))
          on_p(i,j)=4.0_r8/(pn(i-1,j-1)+pn(i-1,j)+pn(i,j-1)+pn(i,j))
        END DO
      END DO
      CALL exchange2d (pnom_p, 'p')
      CALL exchange2d (pmon_p, 'p')
      CALL exchange2d (om_p, 'p')
      CALL exchange2d (on_p, 'p')
      END SUBROUTINE metrics
!
!=======================================================================
!  ini_hmixcoef (Utility/ini_hmixcoef.F:29) + initialize_mixing/initialize_grid defaults
!  (Modules/mod_mixing.F, mod_grid.F): uniform viscosity/diffusivity, background Akv/Akt, drag.
!=======================================================================
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
!=======================================================================
!  set_depth on the host for the initial state (Nonlinear/set_depth.F:76-278, Zt_avg1 = 0)
!=======================================================================
!
      SUBROUTINE set_depth_host ()
      integer :: i, j, k
      real(r8) :: cff_r, cff1_r, cff2_r, cff_w, cff1_w, cff2_w, hinv, hwater, z_r0, z_w0
      DO j=JstrT,JendT
        DO i=IstrT,IendT
          z_w(i,j,0)=-h(i,j)
        END DO
        DO k=1,N
          IF (Vtransform.eq.1) THEN
            cff_r=hc*(sc_r(k)-Cs_r(k))
            cff_w=hc*(sc_w(k)-Cs_w(k))
          ELSE
            cff_r=hc*sc_r(k)
            cff_w=hc*sc_w(k)
          END IF
          cff1_r=Cs_r(k)
          cff1_w=Cs_w(k)
          DO i=IstrT,IendT
            hwater=h(i,j)
            IF (Vtransform.eq.1) THEN
              hinv=1.0_r8/hwater
              z_w0=cff_w+cff1_w*hwater
              z_w(i,j,k)=z_w0+Zt_avg1(i,j)*(1.0_r8+z_w0*hinv)
              z_r0=cff_r+cff1_r*hwater
              z_r(i,j,k)=z_r0+Zt_avg1(i,j)*(1.0_r8+z_r0*hinv)
            ELSE
              hinv=1.0_r8/(hc+hwater)
              cff2_r=(cff_r+cff1_r*hwater)*hinv
              cff2_w=(cff_w+cff1_w*hwater)*hinv
              z_w(i,j,k)=Zt_avg1(i,j)+(Zt_avg1(i,j)+hwater)*cff2_w
              z_r(i,j,k)=Zt_avg1(i,j)+(Zt_avg1(i,j)+hwater)*cff2_r
            END IF
            Hz(i,j,k)=z_w(i,j,k)-z_w(i,j,k-1)
          END DO
        END DO
      END DO
      DO k=0,N
        CALL exchange2d (z_w(:,:,k), 'r')
      END DO
      DO k=1,N
        CALL exchange2d (z_r(:,:,k), 'r')
        CALL exchange2d (Hz(:,:,k), 'r')
      END DO
      END SUBROUTINE set_depth_host
!
!=======================================================================
!  ana_initial, Functionals/ana_initial.h: UPWELLING :828-849, BENCHMARK :545-560
!=======================================================================
!
      SUBROUTINE ana_initial ()
      integer :: i, j, k
      real(r8) :: val1, val2
      zeta=0.0_r8
      ubar=0.0_r8
      vbar=0.0_r8
      u=0.0_r8
      v=0.0_r8
      t=0.0_r8
      IF (IAND(options,ROMS_APP_BENCHMARK).ne.0) THEN
        val1=(44.69_r8/39.382_r8)**2
        val2=val1*(rho0*800.0_r8/g)*(5.0E-05_r8/((42.689_r8/44.69_r8)**2))
        DO k=1,N
          DO j=JstrT,JendT
            DO i=IstrT,IendT
              t(i,j,k,1,1)=val2*EXP(z_r(i,j,k)/800.0_r8)*(0.6_r8-0.4_r8*TANH(z_r(i,j,k)/800.0_r8))
              t(i,j,k,1,2)=35.0_r8
            END DO
          END DO
        END DO
      ELSE
        DO k=1,N
          DO j=JstrT,JendT
            DO i=IstrT,IendT
              t(i,j,k,1,1)=T0+8.0_r8*EXP(z_r(i,j,k)/50.0_r8)
              t(i,j,k,1,2)=S0
            END DO
          END DO
        END DO
      END IF
      END SUBROUTINE ana_initial
!
!=======================================================================
!  Allocate (mod_arrays.F), set up and initialise the host state.
!=======================================================================
!
      SUBROUTINE host_setup (ierr)
      integer, intent(out) :: ierr
      NT=NAT
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
     &           yr(LBi:UBi,LBj:UBj), lonr(LBi:UBi,LBj:UBj), latr(LBi:UBi,LBj:UBj), rdrag(LBi:UBi,LBj:UBj),    &
     &           rdrag2(LBi:UBi,LBj:UBj), visc2_r(LBi:UBi,LBj:UBj), visc2_p(LBi:UBi,LBj:UBj),                  &
     &           diff2(LBi:UBi,LBj:UBj,NT), Zt_avg1(LBi:UBi,LBj:UBj) )
      allocate ( Hz(LBi:UBi,LBj:UBj,N), z_r(LBi:UBi,LBj:UBj,N), z_w(LBi:UBi,LBj:UBj,0:N),                     &
     &           zeta(LBi:UBi,LBj:UBj,3), ubar(LBi:UBi,LBj:UBj,3), vbar(LBi:UBi,LBj:UBj,3),                    &
     &           u(LBi:UBi,LBj:UBj,N,2), v(LBi:UBi,LBj:UBj,N,2), t(LBi:UBi,LBj:UBj,N,3,NT),                    &
     &           Akv(LBi:UBi,LBj:UBj,0:N), Akt(LBi:UBi,LBj:UBj,0:N,NAT) )
      h=0.0_r8; f=0.0_r8; fomn=0.0_r8; pm=0.0_r8; pn=0.0_r8; om_r=0.0_r8; on_r=0.0_r8; om_u=0.0_r8
      on_u=0.0_r8; om_v=0.0_r8; on_v=0.0_r8; om_p=0.0_r8; on_p=0.0_r8; omn=0.0_r8; pmon_r=0.0_r8
      pnom_r=0.0_r8; pmon_p=0.0_r8; pnom_p=0.0_r8; pmon_u=0.0_r8; pnom_u=0.0_r8; pmon_v=0.0_r8
      pnom_v=0.0_r8; dmde=0.0_r8; dndx=0.0_r8; angler=0.0_r8; xr=0.0_r8; yr=0.0_r8; lonr=0.0_r8
      latr=0.0_r8; Zt_avg1=0.0_r8; Hz=0.0_r8; z_r=0.0_r8; z_w=0.0_r8
      CALL ana_grid (ierr)                      ! set_grid, Utility/set_grid.F
      IF (ierr.ne.0) RETURN
      CALL set_scoord (ierr)
      IF (ierr.ne.0) RETURN
      CALL set_weights ()
      CALL metrics ()
      CALL ini_mixing ()                        ! initial, Nonlinear/initial.F:293
      CALL set_depth_host ()                    ! :341
      CALL ana_initial ()                       ! :358
      END SUBROUTINE host_setup

      SUBROUTINE host_free ()
      deallocate ( weight, sc_r, Cs_r, sc_w, Cs_w )
      deallocate ( h, f, fomn, pm, pn, om_r, on_r, om_u, on_u, om_v, on_v, om_p, on_p, omn, pmon_r, pnom_r,    &
     &             pmon_p, pnom_p, pmon_u, pnom_u, pmon_v, pnom_v, dmde, dndx, angler, xr, yr, lonr, latr,    &
     &             rdrag, rdrag2, visc2_r, visc2_p, diff2, Zt_avg1, Hz, z_r, z_w, zeta, ubar, vbar, u, v, t,  &
     &             Akv, Akt )
      END SUBROUTINE host_free
!
!=======================================================================
!  Device context: fill roms_hip_config, upload the state, run the tail of "initial".
!=======================================================================
!
      SUBROUTINE device_init (device, tile, ierr)
      integer, intent(in) :: device, tile
      integer, intent(out) :: ierr
      TYPE (roms_hip_config) :: cfg
      integer :: i, itile, jtile, chunk, margin
      logical :: tw, te, ts, tn
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
      cfg%abi_version=1
      cfg%device=device
      cfg%Lm=Lm; cfg%Mm=Mm; cfg%N=N; cfg%NT=NT; cfg%NAT=NAT; cfg%Nghost=Nghost
      cfg%LBi=tLBi; cfg%UBi=tUBi; cfg%LBj=tLBj; cfg%UBj=tUBj
      cfg%NtileI=NtileI; cfg%NtileJ=NtileJ; cfg%tile=tile
      cfg%EWperiodic=MERGE(1,0,EWperiodic); cfg%NSperiodic=MERGE(1,0,NSperiodic)
      cfg%options=options
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
      cfg%sc_r=0.0_dp; cfg%Cs_r=0.0_dp; cfg%sc_w=0.0_dp; cfg%Cs_w=0.0_dp
      cfg%sc_r(1:N)=sc_r; cfg%Cs_r(1:N)=Cs_r; cfg%sc_w(0:N)=sc_w; cfg%Cs_w(0:N)=Cs_w
      ierr=roms_hip_create(cfg, ctx)
      IF (ierr.ne.0) RETURN
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
      CALL up ('lonr', lonr, 1, ierr); CALL up ('latr', latr, 1, ierr); CALL up ('rdrag', rdrag, 1, ierr)
      CALL up ('rdrag2', rdrag2, 1, ierr); CALL up ('visc2_r', visc2_r, 1, ierr)
      CALL up ('visc2_p', visc2_p, 1, ierr); CALL up ('diff2', diff2, NT, ierr)
      CALL up ('Zt_avg1', Zt_avg1, 1, ierr)
      CALL up ('Hz', Hz, N, ierr); CALL up ('z_r', z_r, N, ierr); CALL up ('z_w', z_w, N+1, ierr)
      CALL up ('zeta', zeta, 3, ierr); CALL up ('ubar', ubar, 3, ierr); CALL up ('vbar', vbar, 3, ierr)
      CALL up ('u', u, 2*N, ierr); CALL up ('v', v, 2*N, ierr); CALL up ('t', t, 3*N*NT, ierr)
      CALL up ('Akv', Akv, N+1, ierr); CALL up ('Akt', Akt, (N+1)*NAT, ierr)
      END SUBROUTINE device_init
!
!  Upload the tile's window (tLBi:tUBi,tLBj:tUBj) of a host array with np horizontal planes.
!
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
        ierr=roms_hip_rhs3d(ctx);                         IF (ierr.ne.0) RETURN
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
        ierr=roms_hip_step3d_t(ctx);                      IF (ierr.ne.0) RETURN
        step%iic=step%iic+1
        step%time=step%time+dt
        ierr=roms_hip_set_stepping(ctx, step)
      END DO
      END SUBROUTINE main3d_kernels

      END MODULE roms_host
