/* orc_stubs.c -- placeholders for oracle parts not restated yet (abort loudly). */
#include "orc.h"
#include <stdio.h>
#include <stdlib.h>
#define NI(name) do { fprintf(stderr, "oracle: %s not implemented\n", name); abort(); } while (0)
#ifndef HAVE_EOS
void orc_eos_nonlinear(orc_t *o, int tile) { (void)o; (void)tile; NI("rho_eos NONLIN_EOS"); }
#endif
#ifndef HAVE_BULK
void orc_set_data_benchmark(orc_t *o, int tile) { (void)o; (void)tile; NI("set_data BENCHMARK"); }
void orc_bulk_flux(orc_t *o, int tile) { (void)o; (void)tile; NI("bulk_flux"); }
#endif
#ifndef HAVE_GEO
void orc_t3dmix2_geo(orc_t *o, int tile) { (void)o; (void)tile; NI("t3dmix2_geo"); }
#endif
#ifndef HAVE_LMD
void orc_lmd_swfrac(const orc_t *o, const orc_bounds *b, double Zscale, const double *Z, double *swdk) {
  (void)o; (void)b; (void)Zscale; (void)Z; (void)swdk; NI("lmd_swfrac"); }
void orc_lmd_vmix(orc_t *o, int tile) { (void)o; (void)tile; NI("lmd_vmix"); }
#endif
#ifndef HAVE_MPDATA
void orc_mpdata_adiff(orc_t *o, int tile, int itrc, const double *Ta, double *Ua, double *Va, double *Wa,
                      const double *oHz) {
  (void)o; (void)tile; (void)itrc; (void)Ta; (void)Ua; (void)Va; (void)Wa; (void)oHz; NI("mpdata_adiff"); }
#endif
