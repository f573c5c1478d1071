!  romsM -- stand-alone driver: romsM roms.in [kernels]
!  Mirrors ROMS/Drivers/nl_roms.h (ROMS_initialize / ROMS_run / ROMS_finalize) for the forward
!  nonlinear model with the time step executed on the MI355X.
      PROGRAM romsM
      USE, INTRINSIC :: iso_c_binding
      USE roms_hip
      USE roms_host
      implicit none
      character(len=512) :: infile, arg2
      integer :: ierr, istep, nchunk, mode
      real(c_double) :: d(16)
      integer(8) :: c0, c1, crate
      IF (COMMAND_ARGUMENT_COUNT().lt.1) THEN
        print '(a)', ' usage: romsM roms.in [kernels]'
        STOP 8
      END IF
      CALL GET_COMMAND_ARGUMENT (1, infile)
      mode=0
      IF (COMMAND_ARGUMENT_COUNT().ge.2) THEN
        CALL GET_COMMAND_ARGUMENT (2, arg2)
        IF (TRIM(arg2).eq.'kernels') mode=1
      END IF
      CALL read_roms_in (TRIM(infile), ierr)
      IF (ierr.ne.0) THEN
        print '(a,a)', ' romsM: cannot read ', TRIM(infile)
        STOP 2
      END IF
      CALL host_setup (ierr)
      IF (ierr.ne.0) THEN
        print '(a,i0)', ' romsM: set-up failed, exit_flag = ', ierr
        STOP 5
      END IF
      print '(1x,a,a,3(1x,i0),a,i0,a,f8.2)', TRIM(MyAppCPP), ':', Lm, Mm, N, '  nfast = ', nfast, '  dt = ', dt
      IF (NtileI*NtileJ.ne.1) THEN
        print '(a)', ' romsM: one process drives one GPU tile; multi-GPU runs are launched through roms_amd.tiling'
        STOP 5
      END IF
      CALL device_init (0, 0, ierr)
      IF (ierr.eq.0) CALL device_start (ierr)
      IF (ierr.ne.0) THEN
        print '(a,i0)', ' romsM: device initialisation failed, exit_flag = ', ierr
        STOP 2
      END IF
      print '(a)', '   STEP   time[DAYS]  KINETIC_ENRG   POTEN_ENRG    TOTAL_ENRG    NET_VOLUME'
      CALL SYSTEM_CLOCK (c0, crate)
      nchunk=MAX(1,ninfo)
      istep=0
      DO WHILE (istep.lt.ntimes)
        nchunk=MIN(nchunk, ntimes-istep)
        IF (mode.eq.0) THEN
          ierr=roms_hip_main3d(ctx, nchunk)
        ELSE
          CALL main3d_kernels (nchunk, ierr)
        END IF
        istep=istep+nchunk
        IF (ierr.ne.0) EXIT
        ierr=roms_hip_diag(ctx, d)
        print '(i7,f12.5,4(1pe14.6))', istep, (dstart*86400.0_dp+REAL(istep,dp)*dt)/86400.0_dp, d(1), d(2), d(3), d(4)
        IF (ierr.ne.0) EXIT
      END DO
      ierr=MAX(ierr, roms_hip_sync(ctx))
      CALL SYSTEM_CLOCK (c1)
      IF (ierr.ne.0) THEN
        print '(a,i0)', ' romsM: blowing-up or device error, exit_flag = ', ierr
        STOP 1
      END IF
      print '(a,f10.3,a,1pe12.4,a)', ' Elapsed wall time: ', REAL(c1-c0,dp)/REAL(crate,dp), ' s   (',             &
     &      REAL(Lm,dp)*REAL(Mm,dp)*REAL(N,dp)*REAL(ntimes,dp)*REAL(crate,dp)/REAL(MAX(c1-c0,1_8),dp),           &
     &      ' grid-cell-updates/sec)'
      ierr=roms_hip_destroy(ctx)
      print '(a)', ' ROMS: DONE'
      END PROGRAM romsM
