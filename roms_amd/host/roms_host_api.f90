!  roms_host_api.f90 -- C-callable surface of the Fortran host driver (libroms_host.so), used by
!  romsM below, by roms_amd/hostlib.py and by bench.py.
!
      MODULE roms_host_api
      USE, INTRINSIC :: iso_c_binding
      USE roms_hip
      USE roms_host
      USE roms_output
      implicit none
      CONTAINS

      FUNCTION c2f (s) RESULT (f_)
      character(kind=c_char), intent(in) :: s(*)
      character(len=:), allocatable :: f_
      integer :: n
      n=0
      DO WHILE (s(n+1).ne.c_null_char)
        n=n+1
      END DO
      allocate (character(len=n) :: f_)
      f_=TRANSFER(s(1:n), f_)
      END FUNCTION c2f
!
!  Name the application header whose cpp options select the physics (ROMS/Include/<app>.h or a custom
!  one); an empty string returns to the built-in lists / the ROMS_APP_HEADER environment variable.
!
      SUBROUTINE roms_host_set_header (path) bind(C, name='roms_host_set_header')
      character(kind=c_char), intent(in) :: path(*)
      app_header=c2f(path)
      END SUBROUTINE roms_host_set_header
!
!  Why the last call returned a non-zero exit_flag (host side); at most n-1 characters + NUL.
!
      SUBROUTINE roms_host_last_error (buf, n) bind(C, name='roms_host_last_error')
      integer(c_int), value :: n
      character(kind=c_char), intent(out) :: buf(n)
      integer :: k, L
      L=MIN(LEN_TRIM(host_message), n-1)
      DO k=1,L
        buf(k)=host_message(k:k)
      END DO
      buf(L+1)=c_null_char
      END SUBROUTINE roms_host_last_error
!
!  Number of roms.in keywords the last roms_host_setup had no use for (read_phypar.F skips them too).
!
      FUNCTION roms_host_unused_keys () bind(C, name='roms_host_unused_keys') RESULT (n)
      integer(c_int) :: n
      n=n_unused_keys
      END FUNCTION roms_host_unused_keys
!
!  ... of those, the ones the reference's reader does not know either (the others are classified inert,
!  docs/ROMS_IN_KEYWORDS.md); buf receives the first such keyword.
!
      FUNCTION roms_host_unknown_keys (buf, n) bind(C, name='roms_host_unknown_keys') RESULT (nk)
      integer(c_int), value :: n
      character(kind=c_char), intent(out) :: buf(n)
      integer(c_int) :: nk
      integer :: k, L
      L=MIN(LEN_TRIM(first_unknown_key), n-1)
      DO k=1,L
        buf(k)=first_unknown_key(k:k)
      END DO
      buf(L+1)=c_null_char
      nk=n_unknown_keys
      END FUNCTION roms_host_unknown_keys
!
!  The " Activated C-preprocessing Options:" block of the run report (checkdefs.F:56-59) of the last set-up, into a file.
!
      FUNCTION roms_host_echo_cppdefs (path) bind(C, name='roms_host_echo_cppdefs') RESULT (ierr)
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int) :: ierr
      integer :: iu, ios
      open (newunit=iu, file=c2f(path), status='replace', action='write', iostat=ios)
      ierr=MERGE(0, 2, ios.eq.0)
      IF (ios.ne.0) RETURN
      CALL echo_cppdefs (iu)
      close (iu)
      END FUNCTION roms_host_echo_cppdefs
!
!  Read roms.in and build the host state only (no device needed).
!
      FUNCTION roms_host_setup (infile) bind(C, name='roms_host_setup') RESULT (ierr)
      character(kind=c_char), intent(in) :: infile(*)
      integer(c_int) :: ierr
      integer :: e
      CALL read_roms_in (c2f(infile), e)
      IF (e.eq.0) CALL host_setup (e)
      exit_flag=e
      ierr=e
      END FUNCTION roms_host_setup
!
!  Create the device context, upload the state, finish "initial" on the device.
!
      FUNCTION roms_host_device_init (device, tile) bind(C, name='roms_host_device_init') RESULT (ierr)
      integer(c_int), value :: device, tile
      integer(c_int) :: ierr
      integer :: e
      IF (.not.allocated(h)) THEN
        ierr=8
        RETURN
      END IF
      CALL device_init (INT(device), INT(tile), e)
      exit_flag=e
      ierr=e
      END FUNCTION roms_host_device_init
!
!  Finish "initial" on the device (after the halo transport of a multi-tile run is installed).
!
      FUNCTION roms_host_start () bind(C, name='roms_host_start') RESULT (ierr)
      integer(c_int) :: ierr
      integer :: e
      IF (.not.c_associated(ctx)) THEN
        ierr=8
        RETURN
      END IF
      CALL device_start (e)
      exit_flag=e
      ierr=e
      END FUNCTION roms_host_start
!
!  b(1:12) = tile, Istr, Iend, Jstr, Jend, LBi, UBi, LBj, UBj of the tile arrays, NtileI, NtileJ, 0
!
      SUBROUTINE roms_host_tile (b) bind(C, name='roms_host_tile')
      integer(c_int), intent(out) :: b(12)
      b=(/ my_tile, tIstr, tIend, tJstr, tJend, tLBi, tUBi, tLBj, tUBj, NtileI, NtileJ, 0 /)
      END SUBROUTINE roms_host_tile
!
!  Advance nsteps baroclinic steps: mode 0 = fused roms_hip_main3d, 1 = kernel-by-kernel main3d.
!
      FUNCTION roms_host_run (nsteps, mode) bind(C, name='roms_host_run') RESULT (ierr)
      integer(c_int), value :: nsteps, mode
      integer(c_int) :: ierr
      integer :: e
      IF (.not.c_associated(ctx)) THEN
        ierr=8
        RETURN
      END IF
      IF (mode.eq.0) THEN
        e=roms_hip_main3d(ctx, nsteps)
      ELSE
        CALL main3d_kernels (INT(nsteps), e)
      END IF
      exit_flag=e
      ierr=e
      END FUNCTION roms_host_run

      FUNCTION roms_host_ctx () bind(C, name='roms_host_ctx') RESULT (p)
      TYPE (c_ptr) :: p
      p=ctx
      END FUNCTION roms_host_ctx

      FUNCTION roms_host_finalize () bind(C, name='roms_host_finalize') RESULT (ierr)
      integer(c_int) :: ierr
      ierr=0
      CALL out_close ()
      gather_cb => NULL()
      IF (c_associated(ctx)) ierr=roms_hip_destroy(ctx)
      ctx=c_null_ptr
      IF (allocated(h)) CALL host_free ()
      END FUNCTION roms_host_finalize
!
!  Output (roms_output.f90).  kind 1: a history record (wrt_his), 2: a restart record (wrt_rst) of the state
!  between two steps, as the reference writes it at main3d.F:591 of the step about to be taken.
!
      FUNCTION roms_host_write (kind) bind(C, name='roms_host_write') RESULT (ierr)
      integer(c_int), value :: kind
      integer(c_int) :: ierr
      integer :: e
      IF (.not.c_associated(ctx).or.kind.lt.1.or.kind.gt.2) THEN
        ierr=8
        RETURN
      END IF
      CALL out_record (INT(kind), e)
      exit_flag=e
      ierr=e
      END FUNCTION roms_host_write
!
!  nsteps steps with the history / restart records NHIS, NRST of roms.in ask for (output.F); final /= 0: also
!  those of the step after the last (main3d.F:595).  mode as roms_host_run.
!
      FUNCTION roms_host_advance (nsteps, mode, final) bind(C, name='roms_host_advance') RESULT (ierr)
      integer(c_int), value :: nsteps, mode, final
      integer(c_int) :: ierr
      integer :: e
      IF (.not.c_associated(ctx)) THEN
        ierr=8
        RETURN
      END IF
      CALL advance (INT(nsteps), INT(mode), final.ne.0, e)
      exit_flag=e
      ierr=e
      END FUNCTION roms_host_advance
!
!  Restart: record rec (1-based; <= 0: the latest) of restart file `path` (empty: ININAME of roms.in) into the
!  device state and the stepping indices (get_state.F + initial.F for nrrec /= 0).
!
      FUNCTION roms_host_get_state (path, rec) bind(C, name='roms_host_get_state') RESULT (ierr)
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int), value :: rec
      integer(c_int) :: ierr
      integer :: e
      character(len=:), allocatable :: p
      IF (.not.c_associated(ctx)) THEN
        ierr=8
        RETURN
      END IF
      p=c2f(path)
      IF (LEN_TRIM(p).eq.0) p=TRIM(ininame)
      CALL get_state (p, INT(rec), e)
      exit_flag=e
      ierr=e
      END FUNCTION roms_host_get_state
!
!  Close the output files (the run report of the reference lists them at ROMS_finalize).
!
      SUBROUTINE roms_host_close_output () bind(C, name='roms_host_close_output')
      CALL out_close ()
      END SUBROUTINE roms_host_close_output
!
!  Multi-tile runs: the call-back that assembles a global array on rank 0 (collective over the ranks).
!
      SUBROUTINE roms_host_set_gather (cb) bind(C, name='roms_host_set_gather')
      TYPE (c_funptr), value :: cb
      IF (c_associated(cb)) THEN
        CALL c_f_procpointer (cb, gather_cb)
      ELSE
        gather_cb => NULL()
      END IF
      END SUBROUTINE roms_host_set_gather
!
!  nrrec, nRST, nHIS, LcycleRST, nAVG, ntsAVG of roms.in and the file names (ININAME, RSTNAME, HISNAME, AVGNAME),
!  NUL-terminated, 256 bytes each
!
      SUBROUTINE roms_host_output_config (ints, names) bind(C, name='roms_host_output_config')
      integer(c_int), intent(out) :: ints(6)
      character(kind=c_char), intent(out) :: names(256,4)
      integer :: k, j
      character(len=256) :: w(4)
      ints=(/ nrrec, nRST, nHIS, MERGE(1,0,LcycleRST), nAVG, ntsAVG /)
      w(1)=ininame; w(2)=rstname; w(3)=hisname; w(4)=avgname
      DO k=1,4
        names(:,k)=c_null_char
        DO j=1,MIN(LEN_TRIM(w(k)),255)
          names(j,k)=w(k)(j:j)
        END DO
      END DO
      END SUBROUTINE roms_host_output_config
!
!  dims(1:24): Lm Mm N NT Nghost LBi UBi LBj UBj nfast ndtfast ntimes options EW NS hadv(1:4) vadv(1:4) ninfo
!  reals(1:8): dt dtfast hc hmin hmax xl el dstart
!
      SUBROUTINE roms_host_dims (dims, reals) bind(C, name='roms_host_dims')
      integer(c_int), intent(out) :: dims(24)
      real(c_double), intent(out) :: reals(8)
      dims(1:13)=(/ Lm, Mm, N, NT, Nghost, LBi, UBi, LBj, UBj, nfast, ndtfast, ntimes, options /)
      dims(14)=MERGE(1,0,EWperiodic)
      dims(15)=MERGE(1,0,NSperiodic)
      dims(16:19)=hadv
      dims(20:23)=vadv
      dims(24)=ninfo
      reals=(/ dt, dtfast, hc, hmin, hmax, xl, el, dstart /)
      END SUBROUTINE roms_host_dims
!
!  Copy a host array out by its reference name; returns the number of values (0 = unknown name).
!
      FUNCTION roms_host_get (name, buf, nmax) bind(C, name='roms_host_get') RESULT (n)
      character(kind=c_char), intent(in) :: name(*)
      integer(c_long), value :: nmax
      real(c_double), intent(out) :: buf(*)
      integer(c_long) :: n
      n=0
      IF (.not.allocated(h)) RETURN
      SELECT CASE (c2f(name))
        CASE ('h');       CALL put (RESHAPE(h,(/SIZE(h)/)))
        CASE ('f');       CALL put (RESHAPE(f,(/SIZE(f)/)))
        CASE ('fomn');    CALL put (RESHAPE(fomn,(/SIZE(fomn)/)))
        CASE ('pm');      CALL put (RESHAPE(pm,(/SIZE(pm)/)))
        CASE ('pn');      CALL put (RESHAPE(pn,(/SIZE(pn)/)))
        CASE ('om_r');    CALL put (RESHAPE(om_r,(/SIZE(om_r)/)))
        CASE ('on_r');    CALL put (RESHAPE(on_r,(/SIZE(on_r)/)))
        CASE ('om_u');    CALL put (RESHAPE(om_u,(/SIZE(om_u)/)))
        CASE ('on_u');    CALL put (RESHAPE(on_u,(/SIZE(on_u)/)))
        CASE ('om_v');    CALL put (RESHAPE(om_v,(/SIZE(om_v)/)))
        CASE ('on_v');    CALL put (RESHAPE(on_v,(/SIZE(on_v)/)))
        CASE ('om_p');    CALL put (RESHAPE(om_p,(/SIZE(om_p)/)))
        CASE ('on_p');    CALL put (RESHAPE(on_p,(/SIZE(on_p)/)))
        CASE ('omn');     CALL put (RESHAPE(omn,(/SIZE(omn)/)))
        CASE ('pmon_r');  CALL put (RESHAPE(pmon_r,(/SIZE(pmon_r)/)))
        CASE ('pnom_r');  CALL put (RESHAPE(pnom_r,(/SIZE(pnom_r)/)))
        CASE ('pmon_p');  CALL put (RESHAPE(pmon_p,(/SIZE(pmon_p)/)))
        CASE ('pnom_p');  CALL put (RESHAPE(pnom_p,(/SIZE(pnom_p)/)))
        CASE ('pmon_u');  CALL put (RESHAPE(pmon_u,(/SIZE(pmon_u)/)))
        CASE ('pnom_u');  CALL put (RESHAPE(pnom_u,(/SIZE(pnom_u)/)))
        CASE ('pmon_v');  CALL put (RESHAPE(pmon_v,(/SIZE(pmon_v)/)))
        CASE ('pnom_v');  CALL put (RESHAPE(pnom_v,(/SIZE(pnom_v)/)))
        CASE ('dmde');    CALL put (RESHAPE(dmde,(/SIZE(dmde)/)))
        CASE ('dndx');    CALL put (RESHAPE(dndx,(/SIZE(dndx)/)))
        CASE ('angler');  CALL put (RESHAPE(angler,(/SIZE(angler)/)))
        CASE ('xr');      CALL put (RESHAPE(xr,(/SIZE(xr)/)))
        CASE ('yr');      CALL put (RESHAPE(yr,(/SIZE(yr)/)))
        CASE ('xp');      CALL put (RESHAPE(xp,(/SIZE(xp)/)))
        CASE ('yp');      CALL put (RESHAPE(yp,(/SIZE(yp)/)))
        CASE ('lonr');    CALL put (RESHAPE(lonr,(/SIZE(lonr)/)))
        CASE ('latr');    CALL put (RESHAPE(latr,(/SIZE(latr)/)))
        CASE ('rdrag');   CALL put (RESHAPE(rdrag,(/SIZE(rdrag)/)))
        CASE ('rdrag2');  CALL put (RESHAPE(rdrag2,(/SIZE(rdrag2)/)))
        CASE ('rmask');   CALL put (RESHAPE(rmask,(/SIZE(rmask)/)))
        CASE ('umask');   CALL put (RESHAPE(umask,(/SIZE(umask)/)))
        CASE ('vmask');   CALL put (RESHAPE(vmask,(/SIZE(vmask)/)))
        CASE ('pmask');   CALL put (RESHAPE(pmask,(/SIZE(pmask)/)))
        CASE ('visc2_r'); CALL put (RESHAPE(visc2_r,(/SIZE(visc2_r)/)))
        CASE ('visc2_p'); CALL put (RESHAPE(visc2_p,(/SIZE(visc2_p)/)))
        CASE ('diff2');   CALL put (RESHAPE(diff2,(/SIZE(diff2)/)))
        CASE ('Zt_avg1'); CALL put (RESHAPE(Zt_avg1,(/SIZE(Zt_avg1)/)))
        CASE ('Hz');      CALL put (RESHAPE(Hz,(/SIZE(Hz)/)))
        CASE ('z_r');     CALL put (RESHAPE(z_r,(/SIZE(z_r)/)))
        CASE ('z_w');     CALL put (RESHAPE(z_w,(/SIZE(z_w)/)))
        CASE ('zeta');    CALL put (RESHAPE(zeta,(/SIZE(zeta)/)))
        CASE ('ubar');    CALL put (RESHAPE(ubar,(/SIZE(ubar)/)))
        CASE ('vbar');    CALL put (RESHAPE(vbar,(/SIZE(vbar)/)))
        CASE ('u');       CALL put (RESHAPE(u,(/SIZE(u)/)))
        CASE ('v');       CALL put (RESHAPE(v,(/SIZE(v)/)))
        CASE ('t');       CALL put (RESHAPE(t,(/SIZE(t)/)))
        CASE ('Akv');     CALL put (RESHAPE(Akv,(/SIZE(Akv)/)))
        CASE ('Akt');     CALL put (RESHAPE(Akt,(/SIZE(Akt)/)))
        CASE ('sc_r');    CALL put (sc_r)
        CASE ('Cs_r');    CALL put (Cs_r)
        CASE ('sc_w');    CALL put (sc_w)
        CASE ('Cs_w');    CALL put (Cs_w)
        CASE ('weight1'); CALL put (weight(1,:))
        CASE ('weight2'); CALL put (weight(2,:))
      END SELECT
      CONTAINS
        SUBROUTINE put (A)
        real(r8), intent(in) :: A(:)
        n=SIZE(A)
        IF (n.le.nmax) buf(1:n)=A
        END SUBROUTINE put
      END FUNCTION roms_host_get

      END MODULE roms_host_api
