!  roms_hip_mod.f90 -- ISO_C_BINDING interface to libroms_hip.so (include/roms_hip.h).
!
!  This is the thin Fortran side of the drop-in boundary: every kernel(ng,tile) procedure that the
!  reference's main3d USEs (ROMS/Nonlinear/main3d.F:108-157) has a bind(C) counterpart here.
!  INTEGRATION.md shows how the reference's own main3d.F would call them.
!
      MODULE roms_hip
      USE, INTRINSIC :: iso_c_binding
      implicit none

      integer(c_int), parameter :: ROMS_MAXT = 4, ROMS_MAXW = 512
!  tracer advection scheme codes
      integer(c_int), parameter :: ROMS_A4 = 1, ROMS_C2 = 2, ROMS_C4 = 3, ROMS_HSIMT = 4, ROMS_MPDATA = 5,    &
     &                             ROMS_SPLINES = 6, ROMS_SPLIT_U3 = 7, ROMS_U3 = 8
!  cpp option bits
      integer(c_int), parameter :: ROMS_UV_ADV = 1, ROMS_UV_COR = 2, ROMS_UV_VIS2 = 4, ROMS_TS_DIF2 = 8,      &
     &   ROMS_MIX_GEO_TS = 16, ROMS_CURVGRID = 32, ROMS_NONLIN_EOS = 64, ROMS_UV_QDRAG = 128,                  &
     &   ROMS_LMD_MIXING = 256, ROMS_BULK_FLUXES = 512, ROMS_SOLAR_SOURCE = 1024, ROMS_ANA_VMIX = 2048,        &
     &   ROMS_SALINITY = 4096, ROMS_SPHERICAL = 8192, ROMS_UV_LOGDRAG = 16384, ROMS_MASKING = 32768,                                 &
     &   ROMS_RADIATION_2D = 65536, ROMS_PLAIN_VDIFF = 131072, ROMS_PLAIN_VVISC = 262144, ROMS_PRSGRD31 = 524288, ROMS_WJ_GRADP = 134217728,                                   &
     &   ROMS_APP_UPWELLING = 1048576, ROMS_APP_BENCHMARK = 2097152, ROMS_APP_KELVIN = 4194304, ROMS_APP_SEAMOUNT = 8388608,   &
     &   ROMS_APP_GRAV_ADJ = 16777216, ROMS_GLS_MIXING = 33554432, ROMS_PRSGRD40 = 67108864, ROMS_MY25_MIXING = 268435456, ROMS_MIX_ISO_TS = 536870912,          &
     &   ROMS_APP_OVERFLOW = 1073741824
!  ... and the upper word of roms_hip_config%options (ABI version 4)
      integer(c_int64_t), parameter :: ROMS_UV_VIS4 = 4294967296_c_int64_t, ROMS_TS_DIF4 = 8589934592_c_int64_t,            &
     &   ROMS_WET_DRY = 17179869184_c_int64_t, ROMS_DIAGNOSTICS_UV = 34359738368_c_int64_t,                               &
     &   ROMS_MIX_GEO_UV = 68719476736_c_int64_t,                                                                        &
     &   ROMS_PRSGRD42 = 8796093022208_c_int64_t, ROMS_PRSGRD44 = 17592186044416_c_int64_t,                               &
     &   ROMS_LMD_DDMIX = 35184372088832_c_int64_t,                                                                      &
     &   ROMS_LMD_BKPP = 70368744177664_c_int64_t,                                                                      &
     &   ROMS_NUDGE_M3CLM = 137438953472_c_int64_t, ROMS_NUDGE_TCLM1 = 274877906944_c_int64_t       ! (tracer itrc: ISHFT(ROMS_NUDGE_TCLM1, itrc-1))
      integer(c_int), parameter :: ROMS_GLS_CANUTO_A = 1, ROMS_GLS_CANUTO_B = 2, ROMS_GLS_KANTHA_CLAYSON = 4,              &
     &   ROMS_GLS_N2S2_HORAVG = 8, ROMS_GLS_RI_SPLINES = 16, ROMS_GLS_K_C2ADVECTION = 32, ROMS_GLS_K_C4ADVECTION = 64,     &
     &   ROMS_GLS_CHARNOK = 128, ROMS_GLS_CRAIG_BANNER = 256
      integer(c_int), parameter :: ROMS_NLBC = 5+ROMS_MAXT
      integer(c_int), parameter :: ROMS_LBC_CLO = 1, ROMS_LBC_PER = 2, ROMS_LBC_GRA = 3, ROMS_LBC_CLA = 4, ROMS_LBC_RAD = 5,      &
     &   ROMS_LBC_RADNUD = 6, ROMS_LBC_CHE = 7, ROMS_LBC_CHI = 8, ROMS_LBC_FLA = 9, ROMS_LBC_SHC = 10

      TYPE, bind(C) :: roms_hip_config
        integer(c_int) :: abi_version, device
        integer(c_int) :: Lm, Mm, N, NT, NAT, Nghost
        integer(c_int) :: LBi, UBi, LBj, UBj
        integer(c_int) :: NtileI, NtileJ, tile
        integer(c_int) :: EWperiodic, NSperiodic
        integer(c_int64_t) :: options                 ! (64 bits: ABI version 4)
        integer(c_int) :: hadv(ROMS_MAXT), vadv(ROMS_MAXT)
        integer(c_int) :: Istr, Iend, Jstr, Jend
        integer(c_int) :: west_edge, east_edge, south_edge, north_edge
        integer(c_int) :: ntfirst, ntstart, ndtfast, nfast
        integer(c_int) :: ninfo
        real(c_double) :: dt, dtfast
        real(c_double) :: weight(0:ROMS_MAXW,2)
        real(c_double) :: rho0, g, lambda, gamma2, Cp
        real(c_double) :: R0, T0, S0, Tcoef, Scoef
        real(c_double) :: hc
        integer(c_int) :: Vtransform
        real(c_double) :: rdrg, rdrg2, Zob
        real(c_double) :: Akt_bak(ROMS_MAXT), Akv_bak
        real(c_double) :: dstart
        real(c_double) :: blk_ZQ, blk_ZT, blk_ZW
        integer(c_int) :: lmd_Jwt
        real(c_double) :: sc_r(256), Cs_r(256), sc_w(0:256), Cs_w(0:256)
!  open boundaries: lbc(variable, edge) = C's lbc[edge][variable] (variable 1 isFsur ... 6.. isTvar; edge 1 west, 2 south,
!  3 east, 4 north), nudging time scales (edge) and (edge, tracer)
        integer(c_int) :: lbc(ROMS_NLBC,4)
        real(c_double) :: FSobc_in(4), FSobc_out(4), M2obc_in(4), M2obc_out(4), M3obc_in(4), M3obc_out(4)
        real(c_double) :: Tobc_in(4,ROMS_MAXT), Tobc_out(4,ROMS_MAXT)
!  GLS_MIXING (ABI version 3): ROMS_GLS_* flags, the GLS_* block of roms.in, LBC(isMtke) per edge
        integer(c_int) :: gls_flags
        real(c_double) :: gls_p, gls_m, gls_n, gls_Kmin, gls_Pmin, gls_cmu0, gls_c1, gls_c2, gls_c3m, gls_c3p, gls_sigk,    &
     &                    gls_sigp
        real(c_double) :: Akk_bak, Akp_bak, Zos, charnok_alpha, crgban_cw
        integer(c_int) :: lbc_tke(4)
!  WET_DRY (ABI version 4): DCRIT of roms.in
        real(c_double) :: Dcrit
        real(c_double) :: obcfac
        integer(c_int) :: volcons
      END TYPE roms_hip_config

      TYPE, bind(C) :: roms_hip_stepping
        integer(c_int) :: iic, iif
        integer(c_int) :: nstp, nnew, nrhs
        integer(c_int) :: kstp, knew, krhs, indx1
        integer(c_int) :: predictor
        real(c_double) :: time
      END TYPE roms_hip_stepping

      INTERFACE
        FUNCTION roms_hip_create (cfg, ctx) bind(C, name='roms_hip_create') RESULT (ierr)
          IMPORT :: c_int, c_ptr, roms_hip_config
          TYPE (roms_hip_config), intent(in) :: cfg
          TYPE (c_ptr), intent(out) :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_destroy (ctx) bind(C, name='roms_hip_destroy') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_last_error () bind(C, name='roms_hip_last_error') RESULT (msg)
          IMPORT :: c_ptr
          TYPE (c_ptr) :: msg
        END FUNCTION
        FUNCTION roms_hip_field_size (ctx, name) bind(C, name='roms_hip_field_size') RESULT (n)
          IMPORT :: c_long, c_ptr, c_char
          TYPE (c_ptr), value :: ctx
          character(kind=c_char), intent(in) :: name(*)
          integer(c_long) :: n
        END FUNCTION
        FUNCTION roms_hip_upload (ctx, name, host, n) bind(C, name='roms_hip_upload') RESULT (ierr)
          IMPORT :: c_int, c_long, c_ptr, c_char, c_double
          TYPE (c_ptr), value :: ctx
          character(kind=c_char), intent(in) :: name(*)
          real(c_double), intent(in) :: host(*)
          integer(c_long), value :: n
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_download (ctx, name, host, n) bind(C, name='roms_hip_download') RESULT (ierr)
          IMPORT :: c_int, c_long, c_ptr, c_char, c_double
          TYPE (c_ptr), value :: ctx
          character(kind=c_char), intent(in) :: name(*)
          real(c_double), intent(out) :: host(*)
          integer(c_long), value :: n
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_sync (ctx) bind(C, name='roms_hip_sync') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_set_stepping (ctx, s) bind(C, name='roms_hip_set_stepping') RESULT (ierr)
          IMPORT :: c_int, c_ptr, roms_hip_stepping
          TYPE (c_ptr), value :: ctx
          TYPE (roms_hip_stepping), intent(in) :: s
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_get_stepping (ctx, s) bind(C, name='roms_hip_get_stepping') RESULT (ierr)
          IMPORT :: c_int, c_ptr, roms_hip_stepping
          TYPE (c_ptr), value :: ctx
          TYPE (roms_hip_stepping), intent(out) :: s
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_wvelocity (ctx, ninp) bind(C, name='roms_hip_wvelocity') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int), value :: ninp
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_diag (ctx, out) bind(C, name='roms_hip_diag') RESULT (ierr)
          IMPORT :: c_int, c_ptr, c_double
          TYPE (c_ptr), value :: ctx
          real(c_double), intent(out) :: out(*)
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_last_diag (ctx, out) bind(C, name='roms_hip_last_diag') RESULT (ierr)
          IMPORT :: c_int, c_ptr, c_double
          TYPE (c_ptr), value :: ctx
          real(c_double), intent(out) :: out(*)
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_get_bounds (ctx, out) bind(C, name='roms_hip_get_bounds') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int), intent(out) :: out(54)
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_avg_config (ctx, nAVG, ntsAVG, nrrec, ntstart, mask) bind(C, name='roms_hip_avg_config')  &
     &           RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int), value :: nAVG, ntsAVG, nrrec, ntstart, mask
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_set_avg (ctx) bind(C, name='roms_hip_set_avg') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_dia_config (ctx, nDIA, ntsDIA, nrrec, ntstart) bind(C, name='roms_hip_dia_config')      &
     &                               RESULT (ierr)
          IMPORT :: c_ptr, c_int
          type(c_ptr), value :: ctx
          integer(c_int), value :: nDIA, ntsDIA, nrrec, ntstart
          integer(c_int) :: ierr
        END FUNCTION roms_hip_dia_config
        FUNCTION roms_hip_wetdry_ini (ctx) bind(C, name='roms_hip_wetdry_ini') RESULT (ierr)
          IMPORT :: c_ptr, c_int
          type(c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION roms_hip_wetdry_ini
        FUNCTION roms_hip_dia_time (ctx, diatime) bind(C, name='roms_hip_dia_time') RESULT (ierr)
          IMPORT :: c_ptr, c_int, c_double
          type(c_ptr), value :: ctx
          real(c_double), intent(out) :: diatime
          integer(c_int) :: ierr
        END FUNCTION roms_hip_dia_time
        FUNCTION roms_hip_avg_time (ctx, avgtime) bind(C, name='roms_hip_avg_time') RESULT (ierr)
          IMPORT :: c_int, c_ptr, c_double
          TYPE (c_ptr), value :: ctx
          real(c_double), intent(out) :: avgtime
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_output_point (ctx) bind(C, name='roms_hip_output_point') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_start (ctx) bind(C, name='roms_hip_start') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_main3d (ctx, nsteps) bind(C, name='roms_hip_main3d') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int), value :: nsteps
          integer(c_int) :: ierr
        END FUNCTION
      END INTERFACE
!
!  One interface per reference procedure kernel(ng,tile) with the plain (ctx) signature.
!
      INTERFACE
        FUNCTION roms_hip_set_depth (ctx) bind(C, name='roms_hip_set_depth') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_set_massflux (ctx) bind(C, name='roms_hip_set_massflux') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_rho_eos (ctx) bind(C, name='roms_hip_rho_eos') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_set_vbc (ctx) bind(C, name='roms_hip_set_vbc') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_ana_vmix (ctx) bind(C, name='roms_hip_ana_vmix') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_set_data (ctx) bind(C, name='roms_hip_set_data') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_omega (ctx) bind(C, name='roms_hip_omega') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_set_zeta (ctx) bind(C, name='roms_hip_set_zeta') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_ini_zeta (ctx) bind(C, name='roms_hip_ini_zeta') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_ini_fields (ctx) bind(C, name='roms_hip_ini_fields') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_rhs3d (ctx) bind(C, name='roms_hip_rhs3d') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_step2d (ctx) bind(C, name='roms_hip_step2d') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_step3d_uv (ctx) bind(C, name='roms_hip_step3d_uv') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_step3d_t (ctx) bind(C, name='roms_hip_step3d_t') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_lmd_vmix (ctx) bind(C, name='roms_hip_lmd_vmix') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
        FUNCTION roms_hip_gls_prestep (ctx) bind(C, name='roms_hip_gls_prestep') RESULT (ierr)
          IMPORT :: c_ptr, c_int
          type(c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION roms_hip_gls_prestep
        FUNCTION roms_hip_gls_corstep (ctx) bind(C, name='roms_hip_gls_corstep') RESULT (ierr)
          IMPORT :: c_ptr, c_int
          type(c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION roms_hip_gls_corstep
        FUNCTION roms_hip_bulk_flux (ctx) bind(C, name='roms_hip_bulk_flux') RESULT (ierr)
          IMPORT :: c_int, c_ptr
          TYPE (c_ptr), value :: ctx
          integer(c_int) :: ierr
        END FUNCTION
      END INTERFACE
      END MODULE roms_hip
