!  romsM -- stand-alone driver:  romsM roms.in [application header] [kernels]
!  The shape of ROMS/Drivers/nl_roms.h (ROMS_initialize / ROMS_run / ROMS_finalize) for the forward
!  nonlinear model with the time step executed on the MI355X; standard output carries the run
!  report of diag.F:446-500 (one entry per NINFO steps) in the reference's own layout.
      PROGRAM romsM
      USE, INTRINSIC :: iso_c_binding
      USE roms_hip
      USE roms_host
      USE roms_output
      implicit none
      character(len=512) :: infile, arg
      integer :: ierr, done, chunk, mode, k, first
      real(c_double) :: d(16)
      integer(8) :: c0, c1, crate
      IF (COMMAND_ARGUMENT_COUNT().lt.1) THEN
        print '(a)', ' usage: romsM roms.in [application header.h] [kernels]'
        STOP 8
      END IF
      CALL GET_COMMAND_ARGUMENT (1, infile)
      mode=0
      DO k=2,COMMAND_ARGUMENT_COUNT()
        CALL GET_COMMAND_ARGUMENT (k, arg)
        IF (TRIM(arg).eq.'kernels') THEN
          mode=1                                  ! main3d kernel by kernel through the C ABI
        ELSE
          app_header=arg
        END IF
      END DO
      CALL read_roms_in (TRIM(infile), ierr)
      IF (ierr.eq.0) CALL host_setup (ierr)
      IF (ierr.ne.0) THEN
        print '(a,i0,2a)', ' romsM: set-up stopped, exit_flag = ', ierr, ': ', TRIM(host_message)
        STOP 5
      END IF
      CALL echo_cppdefs (6)                       ! checkdefs.F:56-59
      print '(a)', ' '
      print '(1x,a,a,3(1x,i0),a,i0,a,f8.2,a,i0,a)', TRIM(MyAppCPP), ':', Lm, Mm, N, '  nfast = ', nfast,           &
     &      '  dt = ', dt, '  (', n_inert_keys, ' roms.in keywords inert for the time step)'
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
      first=0
      IF (nrrec.ne.0) THEN                        ! initial.F: nrrec /= 0 -> get_state (< 0: the latest record)
        CALL get_state (TRIM(ininame), nrrec, ierr)
        IF (ierr.ne.0) THEN
          print '(a,i0,2a)', ' romsM: restart failed, exit_flag = ', ierr, ': ', TRIM(host_message)
          STOP 2
        END IF
        first=step%iic-1
        print '(a,i0,2a)', ' romsM: restarted at time-step ', first, ' from ', TRIM(ininame)
      END IF
      print '(/,1x,a,1x,a,2x,a,3x,a,4x,a,4x,a)', 'TIME-STEP', 'YYYY-MM-DD hh:mm:ss.ss', 'KINETIC_ENRG',           &
     &      'POTEN_ENRG', 'TOTAL_ENRG', 'NET_VOLUME'
      print '(21x,a,7x,a,12x,a,10x,a,7x,a,/)', 'C => (i,j,k)', 'Cu', 'Cv', '  Cw  ', 'Max Speed'
      CALL SYSTEM_CLOCK (c0, crate)
      chunk=MAX(1,ninfo)
      done=0
      DO WHILE (done.lt.ntimes)
        chunk=MIN(chunk, ntimes-done)
        IF (mode.eq.0) THEN
          CALL advance (chunk, 0, .FALSE., ierr)  ! roms_hip_main3d + the records of output.F; the first step is
          IF (ierr.eq.0) ierr=roms_hip_last_diag(ctx, d)      ! a NINFO point: diag ran inside it
        ELSE
          ierr=roms_hip_get_stepping(ctx, step)   ! kernel by kernel: diag where main3d.F:355 has it
          step%nstp=1+MOD(step%iic-1,2)
          step%nnew=3-step%nstp                   ! main3d.F:222-229
          step%nrhs=step%nstp
          ierr=roms_hip_set_stepping(ctx, step)
          ierr=roms_hip_set_massflux(ctx)         ! main3d.F:348-355: diag follows set_massflux and rho_eos of the pass
          IF (ierr.eq.0) ierr=roms_hip_rho_eos(ctx)      ! (the step repeats them: same inputs, same results)
          IF (ierr.eq.0) ierr=roms_hip_diag(ctx, d)
          d(15)=REAL(first+done,c_double)
          IF (ierr.eq.0) CALL advance (chunk, 1, .FALSE., ierr)
        END IF
        IF (ierr.ne.0) EXIT
        CALL report (NINT(d(15)), d)
        done=done+chunk
      END DO
      IF (ierr.eq.0) THEN                         ! the entry of the final state (the reference's last pass)
        ierr=roms_hip_get_stepping(ctx, step)
        step%nstp=1+MOD(step%iic-1,2)
        step%nnew=3-step%nstp
        step%nrhs=step%nstp
        ierr=roms_hip_set_stepping(ctx, step)
        ierr=roms_hip_set_massflux(ctx)           ! the reference's last pass: set_massflux, rho_eos, diag (main3d.F:348-355)
        IF (ierr.eq.0) ierr=roms_hip_rho_eos(ctx)
        IF (ierr.eq.0) ierr=roms_hip_diag(ctx, d)
        IF (ierr.eq.0) CALL report (first+done, d)
        IF (ierr.eq.0) CALL output (ierr)        ! main3d.F:591-595 at iic = ntend+1: the final records
      END IF
      CALL out_close ()
      ierr=MAX(ierr, roms_hip_sync(ctx))
      CALL SYSTEM_CLOCK (c1)
      IF (ierr.ne.0) THEN
        print '(a,i0)', ' romsM: blowing-up or device error, exit_flag = ', ierr
        STOP 1
      END IF
      print '(/,a,f10.3,a,1pe12.4,a)', ' Elapsed wall time: ', REAL(c1-c0,dp)/REAL(crate,dp), ' s   (',           &
     &      REAL(Lm,dp)*REAL(Mm,dp)*REAL(N,dp)*REAL(ntimes,dp)*REAL(crate,dp)/REAL(MAX(c1-c0,1_8),dp),           &
     &      ' grid-cell-updates/sec)'
      ierr=roms_hip_destroy(ctx)
      print '(a)', ' ROMS: DONE'

      CONTAINS
!
!  One entry of the run report, diag.F FORMAT 30 and the (i,j,k) Courant-number line built at :476-484.
!
      SUBROUTINE report (istep, dg)
      integer, intent(in) :: istep
      real(c_double), intent(in) :: dg(16)
      character(len=80) :: frmt
      integer :: di, dj, dk
      di=INT(LOG10(REAL(MAX(Lm,1),dp)))+1
      dj=INT(LOG10(REAL(MAX(Mm,1),dp)))+1
      dk=INT(LOG10(REAL(MAX(N,1),dp)))+1
      print '(i10,1x,a,4(1pe14.6))', istep, model_date(dstart*86400.0_dp+REAL(istep,dp)*dt), dg(1:4)
      write (frmt,'(a,i2.2,a,3(a,i1,a,i1),a)') '(', 35-(6+di+dj+dk), 'x,"("', ',i', di, '.', di, ',",",i', dj,      &
     &       '.', dj, ',",",i', dk, '.', dk, ',")",t35,4(1pe13.6,1x))'
      print frmt, NINT(dg(9)), NINT(dg(10)), NINT(dg(11)), dg(6), dg(7), dg(8), dg(5)
      END SUBROUTINE report
!
!  "YYYY-MM-DD hh:mm:ss.ss" of model time t (seconds) for TIME_REF = 0: days counted from 0001-01-01
!  in the proleptic Gregorian calendar (what time_string / caldate give for that reference date).
!
      FUNCTION model_date (tsec) RESULT (str)
      real(dp), intent(in) :: tsec
      character(len=22) :: str
      integer(8) :: days, era, doe, yoe, doy, mp, y, m, dd
      real(dp) :: sod
      integer :: hh, mi
      days=FLOOR(tsec/86400.0_dp, 8)
      sod=tsec-REAL(days,dp)*86400.0_dp
      days=days+306_8                              ! civil-from-days with the epoch moved to 0000-03-01
      era=days/146097_8
      doe=days-era*146097_8
      yoe=(doe-doe/1460_8+doe/36524_8-doe/146096_8)/365_8
      y=yoe+era*400_8
      doy=doe-(365_8*yoe+yoe/4_8-yoe/100_8)
      mp=(5_8*doy+2_8)/153_8
      dd=doy-(153_8*mp+2_8)/5_8+1_8
      m=MERGE(mp+3_8, mp-9_8, mp.lt.10_8)
      IF (m.le.2_8) y=y+1_8
      hh=INT(sod/3600.0_dp)
      mi=INT((sod-3600.0_dp*hh)/60.0_dp)
      write (str,'(i4.4,a,i2.2,a,i2.2,1x,i2.2,a,i2.2,a,f5.2)') y, '-', m, '-', dd, hh, ':', mi, ':',               &
     &       sod-3600.0_dp*hh-60.0_dp*mi
      IF (str(18:18).eq.' ') str(18:18)='0'
      END FUNCTION model_date
      END PROGRAM romsM
