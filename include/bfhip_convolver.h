/*
 * bfhip_convolver.h -- the 22 link-time symbols of BruteFIR's convolver.h, exported by
 * libbfhip.so with the reference's names, argument order, host-memory semantics and error
 * behaviour, so that the host links libbfhip.so in place of fftw_convolver.o +
 * convolver_xmm.o (reference Makefile:43-44,130-131) without touching a caller.
 *
 * Every buffer is caller-owned HOST memory in the reference's layouts (time domain: 2L reals;
 * halfcomplex from time2freq; "4 re / 4 im" reordered cbufs everywhere else -- SURVEY A.2/A.3).
 *
 * Two kinds of entry points (csrc/convolver_abi.hip and csrc/host_ops.cpp):
 *   - the per-block ops (raw2cbuf, time2freq, mixnscale, convolve*, dirac*, crossfade, eval,
 *     freq2time, cbuf2raw, td_convolve) stage their operands to the device, run the HIP
 *     kernel(s) and copy the result back: correct and self-contained, but PCIe-bound -- the
 *     per-block loop should use the fused API of bfhip.h instead; they are the
 *     unfused fallback (e.g. for hosts whose modules hook the per-buffer events);
 *   - what the host calls in the PARENT before it forks (bfconf.c: init, cbufsize, coeffs2cbuf,
 *     verify; delay.c: td_new) or from module processes (bflogic_eq: runtime_coeffs2cbuf,
 *     fftplan) and the debug dump are pure host code (a small host FFT where the reference runs
 *     FFTW outside its block loop): HIP state does not survive fork(), and a module process has
 *     no business owning a GPU context.  The device is initialised by the first per-block op, in
 *     the process that makes it; a per-block op in a fork()ed child of such a process fails with
 *     fatal code 106.
 *
 * The arithmetic of the FFT-free ops (mixnscale, convolve*, dirac, raw2cbuf, cbuf2raw) is
 * done without FMA contraction in the reference's operation order: results are bit-identical
 * to the reference's C loops (tests/test_gpu_ops.py against tests/golden/ref_*.npz).
 *
 * Citations: convolver.h:16-152 (declarations), fftw_convolver.c (definitions).
 */
#ifndef BFHIP_CONVOLVER_H
#define BFHIP_CONVOLVER_H

#include <stdint.h>

#include "bfhip.h"

#ifdef __cplusplus
extern "C" {
#endif

#ifndef CONVOLVER_MIXMODE_INPUT
#define CONVOLVER_MIXMODE_INPUT      1
#define CONVOLVER_MIXMODE_INPUT_ADD  2   /* declared by the reference, never implemented */
#define CONVOLVER_MIXMODE_OUTPUT     3
#endif

/* struct sample_format + struct buffer_format exactly as dai.h:21-34 lays them out; the host
   passes its own `struct buffer_format *` */
struct bfhip_sample_format {
    int isfloat;
    int swap;
    int bytes;
    int sbytes;
    double scale;
    int format;
};
struct bfhip_buffer_format {
    struct bfhip_sample_format sf;
    int sample_spacing;
    int byte_offset;
};

/* struct dither_state, dither.h:17-22 */
struct bfhip_dither_state {
    int randtab_ptr;
    int8_t *randtab;
    float sf[2];
    double sd[2];
};

typedef struct _td_conv_t_ td_conv_t;

/* convolver.h:149-152 / fftw_convolver.c:784-851.  config_filename (FFTW wisdom in the
   reference) is accepted and ignored.  Returns 1 (true) / 0 with the reference's messages. */
int convolver_init(const char config_filename[], int length, int realsize);
int convolver_cbufsize(void);                                           /* :520-524 */

void convolver_raw2cbuf(void *rawbuf, void *cbuf, void *next_cbuf,      /* :170-194 */
                        struct bfhip_buffer_format *bf,
                        void (*postprocess)(void *realbuf, int n_samples, void *arg),
                        void *pp_arg);
void convolver_time2freq(void *input_cbuf, void *output_cbuf);          /* :196-214 */
void convolver_mixnscale(void *input_cbufs[], void *output_cbuf,        /* :216-228 */
                         double scales[], int n_bufs, int mixmode);
void convolver_convolve_inplace(void *cbuf, void *coeffs);              /* :230-239 */
void convolver_convolve(void *input_cbuf, void *coeffs, void *output_cbuf);     /* :241-251 */
void convolver_crossfade_inplace(void *input_cbuf, void *crossfade_cbuf,        /* :330-368 */
                                 void *buffer_cbuf);
void convolver_convolve_add(void *input_cbuf, void *coeffs, void *output_cbuf); /* :253-328 */
void convolver_dirac_convolve(void *input_cbuf, void *output_cbuf);     /* :380-389 */
void convolver_dirac_convolve_inplace(void *cbuf);                      /* :370-378 */
void convolver_freq2time(void *input_cbuf, void *output_cbuf);          /* :391-409 */
void convolver_convolve_eval(void *input_cbuf, void *buffer_cbuf,       /* :411-433 */
                             void *output_cbuf);
void convolver_cbuf2raw(void *cbuf, void *outbuf, struct bfhip_buffer_format *bf,   /* :482-518 */
                        int apply_dither, void *dither_state, struct bfhip_overflow *overflow);
void *convolver_coeffs2cbuf(void *coeffs, int n_coeffs, double scale,   /* :526-573 */
                            void *optional_dest);
void convolver_runtime_coeffs2cbuf(void *src, void *dest);              /* :575-596 */
int convolver_verify_cbuf(void *cbufs[], int n_cbufs);                  /* :598-622 */
void convolver_debug_dump_cbuf(const char filename[], void *cbufs[], int n_cbufs);  /* :624-660 */

/* :662-680.  The reference returns an FFTW plan that bflogic_eq hands to fftw[f]_execute_r2r
   (rendereq.h:66-70).  Here the handle is opaque and is executed with bfhip_fftplan_execute()
   (same in/out conventions: R2HC when invert == 0, HC2R otherwise; in may equal out). */
void *convolver_fftplan(int order, int invert, int inplace);
void bfhip_fftplan_execute(void *plan, void *in, void *out);

int convolver_td_block_length(int n_coeffs);                            /* :689-696 */
td_conv_t *convolver_td_new(void *coeffs, int n_coeffs);                /* :698-736 */
void convolver_td_convolve(td_conv_t *tdc, void *overlap_block);        /* :767-782 */

/* Set by the last failed convolver_* call that cannot return an error (the reference calls
   bf_exit() / abort() there): 0 = none.  The host may also install a handler that is called
   instead (default: print to stderr and exit(BF_EXIT_OTHER = 1) like bf_exit). */
int bfhip_convolver_last_fatal(void);
void bfhip_convolver_set_fatal_handler(void (*handler)(int code, const char *message));

#ifdef __cplusplus
}
#endif
#endif
