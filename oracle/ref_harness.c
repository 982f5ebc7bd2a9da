/*
 * TEST INFRASTRUCTURE ONLY -- never linked into or called from the product.
 *
 * Harness that compiles the FFTW-free part of the reference hot path from the
 * reference's own sources WHERE THEY LIE under /root/reference (nothing is
 * copied into this repo) and exports the file-static inner loops under
 * "ref_*" names so tests/ can call them through ctypes:
 *
 *   fftw_convfuns.h   mixnscale / convolve / convolve_inplace / convolve_add /
 *                     dirac_convolve(_inplace)     (float and double)
 *   raw2real.h        raw sample -> real
 *   real2raw.h + dither.h/dither_funs.h  real -> raw, overflow accounting,
 *                     HP-TPDF dither
 *   convolver_xmm.c   the SSE / SSE2 convolve_add (separate object)
 *   dither.c          Tausworthe table + randmap (separate object)
 *   emalloc.c         (separate object, needed by dither.c)
 *
 * delay.c / firwindow.c live in a library of their own (ref_delay_harness.c ->
 * oracle/_ref/libbfref_delay.so): delay.c references three FFTW-backed symbols
 * that have to be closed at link time, and the library that pins the hot loop
 * carries no such placeholder.
 *
 * The headers are "templates": fftw_convolver.c:128-168 and :435-479 in the
 * reference instantiate them by defining the macro names below and including
 * them; this file does the same instantiation, because the macro names are
 * the headers' interface.  The file-static globals n_fft/n_fft2 that the
 * headers read (fftw_convolver.c:44) and the two host globals they touch
 * (`bfconf` for safety_limit/quiet, `bf_exit`) are defined here.
 *
 * What is NOT here: anything that needs FFTW3 (absent from this image and not
 * vendored by the reference): convolver_time2freq/freq2time/coeffs2cbuf/
 * crossfade/convolve_eval and the whole brutefir binary.  No stand-in for
 * FFTW is written; those functions are pinned by mathematical definition
 * (numpy.fft) and analytic known-answer tests instead -- see DESIGN.md.
 */
#include <stdbool.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <inttypes.h>
#include <math.h>

#include "defs.h"
#include "convolver.h"
#include "bfconf.h"
#include "bfrun.h"
#include "dai.h"
#include "swap.h"
#include "dither.h"
#include "numunion.h"
#include "asmprot.h"

static int n_fft, n_fft2;

static struct bfconf ref_bfconf_storage;
struct bfconf *bfconf = &ref_bfconf_storage;
static int ref_exit_called;

void
bf_exit(int status)
{
    /* the reference kills every process; the harness records and returns so
       the python test can assert on it */
    ref_exit_called = status ? status : -1;
}

#define real_t float
#define REALSIZE 4
#define RAW2REAL_NAME raw2realf
#define MIXNSCALE_NAME mixnscalef
#define CONVOLVE_INPLACE_NAME convolve_inplacef
#define CONVOLVE_NAME convolvef
#define CONVOLVE_ADD_NAME convolve_addf
#define DIRAC_CONVOLVE_INPLACE_NAME dirac_convolve_inplacef
#define DIRAC_CONVOLVE_NAME dirac_convolvef
#include "raw2real.h"
#include "fftw_convfuns.h"
#undef real_t
#undef REALSIZE
#undef RAW2REAL_NAME
#undef MIXNSCALE_NAME
#undef CONVOLVE_INPLACE_NAME
#undef CONVOLVE_NAME
#undef CONVOLVE_ADD_NAME
#undef DIRAC_CONVOLVE_INPLACE_NAME
#undef DIRAC_CONVOLVE_NAME

#define real_t double
#define REALSIZE 8
#define RAW2REAL_NAME raw2reald
#define MIXNSCALE_NAME mixnscaled
#define CONVOLVE_INPLACE_NAME convolve_inplaced
#define CONVOLVE_NAME convolved
#define CONVOLVE_ADD_NAME convolve_addd
#define DIRAC_CONVOLVE_INPLACE_NAME dirac_convolve_inplaced
#define DIRAC_CONVOLVE_NAME dirac_convolved
#include "raw2real.h"
#include "fftw_convfuns.h"
#undef real_t
#undef REALSIZE
#undef RAW2REAL_NAME
#undef MIXNSCALE_NAME
#undef CONVOLVE_INPLACE_NAME
#undef CONVOLVE_NAME
#undef CONVOLVE_ADD_NAME
#undef DIRAC_CONVOLVE_INPLACE_NAME
#undef DIRAC_CONVOLVE_NAME

/* real2raw instantiations: the same four the reference makes
   (fftw_convolver.c:435-479); note that the reference's float no-dither
   variant calls the DOUBLE real2int (":448"), kept as is. */
#define real_t float
#define REALSIZE 4
#define REAL2RAW_NAME real2rawf_hp_tpdf
#define REAL2INT_CALL ditherf_real2int_hp_tpdf(((float *)realbuf)[n], rmin,    \
                                               rmax, imin, imax, overflow,     \
                                               dither_state, n)
#define REAL2RAW_EXTRA_PARAMS , struct dither_state *dither_state
#include "real2raw.h"
#undef REAL2RAW_NAME
#undef REAL2INT_CALL
#undef REAL2RAW_EXTRA_PARAMS

#define REAL2RAW_NAME real2rawf_no_dither
#define REAL2INT_CALL ditherd_real2int_no_dither(((float *)realbuf)[n], rmin,  \
                                                 rmax, imin, imax, overflow)
#define REAL2RAW_EXTRA_PARAMS
#include "real2raw.h"
#undef REAL2RAW_NAME
#undef REAL2INT_CALL
#undef REAL2RAW_EXTRA_PARAMS
#undef REALSIZE
#undef real_t

#define real_t double
#define REALSIZE 8
#define REAL2RAW_NAME real2rawd_hp_tpdf
#define REAL2INT_CALL ditherd_real2int_hp_tpdf(((double *)realbuf)[n], rmin,   \
                                               rmax, imin, imax, overflow,     \
                                               dither_state, n)
#define REAL2RAW_EXTRA_PARAMS , struct dither_state *dither_state
#include "real2raw.h"
#undef REAL2RAW_NAME
#undef REAL2INT_CALL
#undef REAL2RAW_EXTRA_PARAMS

#define REAL2RAW_NAME real2rawd_no_dither
#define REAL2INT_CALL ditherd_real2int_no_dither(((double *)realbuf)[n], rmin, \
                                                 rmax, imin, imax, overflow)
#define REAL2RAW_EXTRA_PARAMS
#include "real2raw.h"
#undef REAL2RAW_NAME
#undef REAL2INT_CALL
#undef REAL2RAW_EXTRA_PARAMS
#undef REALSIZE
#undef real_t

/* ------------------------------------------------------------------------
 * exported wrappers (plain C ABI for ctypes)
 * ---------------------------------------------------------------------- */

void
ref_set_length(int length, double safety_limit)
{
    n_fft = 2 * length;
    n_fft2 = length;
    bfconf->quiet = true;
    bfconf->safety_limit = safety_limit;
    ref_exit_called = 0;
}

int
ref_exit_status(void)
{
    return ref_exit_called;
}

void
ref_mixnscale(int realsize, void *in[], void *out, double scales[], int n,
              int mode)
{
    if (realsize == 4) {
        mixnscalef(in, out, scales, n, mode);
    } else {
        mixnscaled(in, out, scales, n, mode);
    }
}

void
ref_convolve(int realsize, void *b, void *c, void *d)
{
    if (realsize == 4) {
        convolvef(b, c, d);
    } else {
        convolved(b, c, d);
    }
}

void
ref_convolve_inplace(int realsize, void *b, void *c)
{
    if (realsize == 4) {
        convolve_inplacef(b, c);
    } else {
        convolve_inplaced(b, c);
    }
}

void
ref_convolve_add(int realsize, void *b, void *c, void *d)
{
    if (realsize == 4) {
        convolve_addf(b, c, d);
    } else {
        convolve_addd(b, c, d);
    }
}

/* the SSE kernels the reference intends for Intel hosts
   (fftw_convolver.c:268-279, loop_counter = n_fft >> 3) */
void
ref_convolve_add_simd(int realsize, void *b, void *c, void *d)
{
    if (realsize == 4) {
        convolver_sse_convolve_add(b, c, d, n_fft >> 3);
    } else {
        convolver_sse2_convolve_add(b, c, d, n_fft >> 3);
    }
}

void
ref_dirac_convolve(int realsize, void *in, void *out)
{
    if (realsize == 4) {
        dirac_convolvef(in, out);
    } else {
        dirac_convolved(in, out);
    }
}

void
ref_dirac_convolve_inplace(int realsize, void *buf)
{
    if (realsize == 4) {
        dirac_convolve_inplacef(buf);
    } else {
        dirac_convolve_inplaced(buf);
    }
}

void
ref_raw2real(int realsize, void *realbuf, void *rawbuf, int bytes, int isfloat,
             int spacing, int swap, int n_samples)
{
    if (realsize == 4) {
        raw2realf(realbuf, rawbuf, bytes, isfloat, spacing, swap, n_samples);
    } else {
        raw2reald(realbuf, rawbuf, bytes, isfloat, spacing, swap, n_samples);
    }
}

/* overflow: {n_overflows u32, intlargest i32, largest f64, max f64} =
   struct bfoverflow (bfmod.h:99-104) */
void
ref_real2raw(int realsize, void *rawbuf, void *realbuf, int bits, int bytes,
             int isfloat, int spacing, int swap, int n_samples,
             struct bfoverflow *overflow, int dither_channel)
{
    if (dither_channel >= 0) {
        struct dither_state *ds = bfconf->dither_state[dither_channel];
        /* what convolver_cbuf2raw does (fftw_convolver.c:491-496) */
        dither_preloop_real2int_hp_tpdf(ds, n_samples);
        if (realsize == 4) {
            real2rawf_hp_tpdf(rawbuf, realbuf, bits, bytes, isfloat, spacing,
                              swap, n_samples, overflow, ds);
        } else {
            real2rawd_hp_tpdf(rawbuf, realbuf, bits, bytes, isfloat, spacing,
                              swap, n_samples, overflow, ds);
        }
    } else if (realsize == 4) {
        real2rawf_no_dither(rawbuf, realbuf, bits, bytes, isfloat, spacing,
                            swap, n_samples, overflow);
    } else {
        real2rawd_no_dither(rawbuf, realbuf, bits, bytes, isfloat, spacing,
                            swap, n_samples, overflow);
    }
}

/* dither_init (dither.c:75-139) with harness-owned state array */
int
ref_dither_init(int n_channels, int sample_rate, int realsize, int max_size,
                int max_samples_per_loop)
{
    bfconf->dither_state = calloc(n_channels, sizeof(struct dither_state *));
    return dither_init(n_channels, sample_rate, realsize, max_size,
                       max_samples_per_loop, bfconf->dither_state);
}

int
ref_dither_table(const int8_t **table)
{
    *table = dither_randtab;
    return dither_randtab_size;
}

int
ref_dither_randtab_ptr(int channel)
{
    return bfconf->dither_state[channel]->randtab_ptr;
}
