!
!  oracle/ref/functionals/ana_mask.h -- TEST INFRASTRUCTURE (user analytical file of the custom application
!  oracle/ref/upwelling_mask.h, found through the include path the way the reference finds a user's
!  MY_ANALYTICAL_DIR files, makefile:230-238).
!
!  analytical.F includes <ana_mask.h> when ANA_GRID and MASKING are both defined and the stock file has no
!  branch for UPWELLING.  The masks of this test are DATA: tests/refdrive.py writes rmask, umask, vmask into
!  GRID(ng) through ref_field before ref_initial (metrics.F then derives the slipperiness mask pmask itself),
!  exactly as a grid NetCDF file would deliver them.  Nothing to compute here.
!
      SUBROUTINE ana_mask (ng, tile, model)
      USE mod_param
      integer, intent(in) :: ng, tile, model
      RETURN
      END SUBROUTINE ana_mask
