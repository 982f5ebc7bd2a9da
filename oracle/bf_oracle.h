/*
 * bf_oracle -- CPU restatement of BruteFIR's partitioned-convolution filter path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under brutefir_amd/ (the product) may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg do, and only as the checker.
 *
 * Parity status: the non-FFT inner loops (mix/scale+reorder, convolve, convolve_add,
 * dirac, raw<->real conversion, quantiser, dither, overflow accounting) are pinned
 * BIT-EXACTLY against the reference's own code compiled into oracle/_ref (see
 * ref_harness.c, tests/test_oracle_vs_ref.py, fixtures tests/golden/ref_ops_*.npz).
 * The FFT (FFTW3 in the reference: third-party, unpinned version, absent from this
 * image, `Makefile:21`) cannot be built here, so the transforms are "parity unpinned" by
 * reference output: they are pinned by the mathematical definition of FFTW's R2HC/HC2R
 * transforms (checked against numpy.fft), by analytic known-answer tests (dirac, delayed
 * dirac, cascades) and by an independent numpy/scipy linear convolution of the same inputs.
 * The filter_process() CONTROL FLOW (ring slots, delay clamp, cblocks, warm-up, coefficient
 * switch with cross-fade, cascades) is pinned by the reference's own filter_process():
 * bfrun.c compiled unchanged into oracle/_ref/ref_filter_process over the product's
 * convolver.h symbols (ref_filter_process_harness.c), whose outputs this restatement has to
 * match on random networks (tests/test_gpu_refloop.py; needs the GPU, the ops being the
 * product's).
 *
 * Layouts are the reference's: halfcomplex spectra from the FFT,
 * "4 re / 4 im" reordered spectra in rings / coefficient partitions / filter outputs.
 */
#ifndef BF_ORACLE_H
#define BF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BFO_MIX_INPUT  1   /* convolver.h:39 */
#define BFO_MIX_OUTPUT 3   /* convolver.h:41 */

/* same field order and types as struct bfoverflow, bfmod.h:99-104 */
typedef struct {
    unsigned int n_overflows;
    int32_t intlargest;
    double largest;
    double max;
} bfo_overflow;

/* struct sample_format + struct buffer_format, dai.h:21-34, flattened */
typedef struct {
    int isfloat;
    int swap;
    int bytes;
    int sbytes;
    double scale;
    int sample_spacing; /* in samples */
    int byte_offset;    /* in bytes */
} bfo_format;

typedef struct bfo_ctx bfo_ctx;
typedef struct bfo_engine bfo_engine;

/* ---- op level: one function per convolver.h entry on the path ------------------ */

/* convolver_init, fftw_convolver.c:784-851.  NULL on invalid length / realsize. */
bfo_ctx *bfo_ctx_new(int length, int realsize);
void bfo_ctx_free(bfo_ctx *c);
int bfo_cbufsize(const bfo_ctx *c);                       /* fftw_convolver.c:520-524 */

void bfo_raw2real(const bfo_ctx *c, void *real, const void *raw, int bytes, int isfloat,
                  int spacing, int swap, int n_samples);  /* raw2real.h:7-160 */
void bfo_raw2cbuf(const bfo_ctx *c, const void *rawbuf, void *cbuf, void *next_cbuf,
                  const bfo_format *bf);                  /* fftw_convolver.c:170-194 */
void bfo_time2freq(const bfo_ctx *c, const void *in, void *out);   /* :196-214, R2HC */
void bfo_freq2time(const bfo_ctx *c, const void *in, void *out);   /* :391-409, HC2R */
void bfo_mixnscale(const bfo_ctx *c, void *const in[], void *out, const double scales[],
                   int n_bufs, int mode);                 /* fftw_convfuns.h:7-501 */
void bfo_convolve(const bfo_ctx *c, const void *b, const void *h, void *d);   /* :534-561 */
void bfo_convolve_inplace(const bfo_ctx *c, void *b, const void *h);          /* :503-532 */
void bfo_convolve_add(const bfo_ctx *c, const void *b, const void *h, void *d); /* :564-590 */
void bfo_dirac_convolve(const bfo_ctx *c, const void *in, void *out);         /* :606-619 */
void bfo_dirac_convolve_inplace(const bfo_ctx *c, void *buf);                 /* :592-604 */
/* fftw_convolver.c:330-368, float-branch semantics for both precisions */
void bfo_crossfade_inplace(const bfo_ctx *c, void *input, void *crossfade, void *buffer);
/* fftw_convolver.c:411-433; buffer is 1.5 cbufs, zeroed before first call */
void bfo_convolve_eval(const bfo_ctx *c, const void *in, void *buffer, void *out);
/* fftw_convolver.c:526-573; returns 0 on NaN/Inf (reference returns NULL) */
int bfo_coeffs2cbuf(const bfo_ctx *c, const void *coeffs, int n_coeffs, double scale,
                    void *dest);
void bfo_runtime_coeffs2cbuf(const bfo_ctx *c, const void *src, void *dest); /* :575-596 */
int bfo_verify_cbuf(const bfo_ctx *c, void *const cbufs[], int n);          /* :598-622 */

/* dither.c:75-139 / dither.h:28-38.  One table per context. */
int bfo_dither_init(bfo_ctx *c, int n_channels, int sample_rate, int max_size,
                    int max_samples_per_loop);
const int8_t *bfo_dither_table(const bfo_ctx *c, int *size);
int bfo_dither_randtab_ptr(const bfo_ctx *c, int channel);

/* convolver_cbuf2raw, fftw_convolver.c:482-518 (+ real2raw.h, dither_funs.h).
   dither_channel < 0: no dither.  Returns 0 ok, 1 NaN/Inf (reference abort()s),
   2 safety limit exceeded (reference bf_exit()s). */
int bfo_cbuf2raw(bfo_ctx *c, const void *cbuf, void *outbuf, const bfo_format *bf,
                 int dither_channel, bfo_overflow *overflow, double safety_limit);

/* ---- integer sample delay, delay.c:78-340 (contiguous buffers: the way filter_process()
   uses it for N:1 mapped channels, bfrun.c:1517-1521,1948).  Pinned bit-exactly against the
   reference's delay.c incl. run-time delay changes (tests/test_oracle_delay.py). */
typedef struct bfo_delay bfo_delay;
bfo_delay *bfo_delay_new(int fragment_size, int initdelay, int maxdelay, int sample_size);
void bfo_delay_free(bfo_delay *d);
void bfo_delay_update(bfo_delay *d, void *buf, int delay);   /* buf: fragment_size samples */
/* firwindow_kaiser, firwindow.c:80-160 (incl. its double application for offset != 0) */
void bfo_firwindow_kaiser(void *target, int len, double offset, double beta, int realsize);

/* ---- block level: the contract of one filter_process() iteration --------------- */
/* (bfrun.c:1420-2083).  1:1 virtual/physical channels, no sub-sample
   delay, no powersave (it changes no sample), events not modelled. */

bfo_engine *bfo_engine_new(int length, int n_blocks, int realsize, int n_in, int n_out);
void bfo_engine_free(bfo_engine *e);
void bfo_engine_set_format(bfo_engine *e, int io, int channel, const bfo_format *bf);
void bfo_engine_set_safety_limit(bfo_engine *e, double limit);
/* `powersave:` (bfconf.c:1549-1561): 0 = off; >= 1.0 = `true` (windows that are exactly zero);
   10^(dB/20) < 1.0 = noise floor in full-scale units (bfrun.c:721-771, 1541-1553) */
void bfo_engine_set_powersave(bfo_engine *e, double analog_powersave);
/* bfconf.c:3170-3230 decides which outputs dither; here the caller says which */
int bfo_engine_enable_dither(bfo_engine *e, const int out_channels[], int n,
                             int sample_rate, int max_size);
/* N:1 virtual -> physical channel mapping (`mapping:` in the config; bfconf->virt2phys).
   Any mapping is legal (bench4_config: `mapping: 0,1,0,1,0,1`); outputs are processed grouped by
   physical channel, members in ascending virtual order (bfrun.c:2322-2323).  After this call
   set_format / enable_dither address PHYSICAL channels.  For channels that share a physical
   one, integer delay and mute are applied inside the block (bfrun.c:1509-1531,1938-2003);
   for 1:1 channels they are dai.c's business and ignored here.  Returns 1 / 0. */
int bfo_engine_map_channels(bfo_engine *e, int io, int n_phys, const int virt2phys[]);
void bfo_engine_set_delay(bfo_engine *e, int io, int virt_channel, int delay);
void bfo_engine_set_maxdelay(bfo_engine *e, int io, int virt_channel, int maxdelay);
void bfo_engine_set_mute(bfo_engine *e, int io, int virt_channel, int muted);
/* sub-sample delay (`sdf_length`, `sdf_beta`, per-channel `subdelay:`; delay.c:416-505 on
   top of convolver_td_*).  subdelay in (-100, 100) hundredths of a sample; -100 = channel has
   no sub-sample filter (BF_UNDEFINED_SUBDELAY) and, if it shares a physical channel, is delayed
   by sdf_length whole samples instead.  Whether a channel has a filter is fixed at the first
   block.  Returns 1 / 0. */
int bfo_engine_enable_subdelay(bfo_engine *e, int sdf_length, double beta);
void bfo_engine_set_subdelay(bfo_engine *e, int io, int virt_channel, int subdelay);
/* load_coeff, bfconf.c:1867-2030: split n_taps into n_blocks partitions of L
   (n_blocks <= 0: ceil(n_taps / L), capped to N).  Returns coeff index or -1. */
int bfo_engine_add_coeff(bfo_engine *e, const void *taps, int n_taps, double scale,
                         int n_blocks);
/* struct bffilter + initial struct bffilter_control (bfmod.h:113-133).
   Filters must be added in an order where from_filters precede their users
   (the reference sorts them so, bfconf.c:2933-2964).  Returns filter index. */
int bfo_engine_add_filter(bfo_engine *e,
                          int n_in_ch, const int in_ch[], const double in_scale[],
                          int n_in_f, const int in_f[], const double in_fscale[],
                          int n_out_ch, const int out_ch[], const double out_scale[],
                          int coeff, int delayblocks, int crossfade);
/* run-time control = what bflogic_cli writes into icomm->fctrl under the mutex */
void bfo_engine_set_coeff(bfo_engine *e, int filter, int coeff);
void bfo_engine_set_delayblocks(bfo_engine *e, int filter, int delayblocks);
void bfo_engine_set_scale(bfo_engine *e, int filter, int io, int index, double scale);
void bfo_engine_set_fscale(bfo_engine *e, int filter, int index, double scale);
/* one block: raw in -> raw out.  0 ok, 1 NaN/Inf, 2 safety limit. */
int bfo_engine_block(bfo_engine *e, const void *rawin, void *rawout);
void bfo_engine_get_overflow(const bfo_engine *e, int out_channel, bfo_overflow *of);
unsigned int bfo_engine_blockcounter(const bfo_engine *e);
/* the filter's current output spectrum (reordered layout), for spot checks */
const void *bfo_engine_filter_output(const bfo_engine *e, int filter);
/* the mixed output spectrum of a channel (halfcomplex), before the inverse FFT */
const void *bfo_engine_output_spectrum(const bfo_engine *e, int out_channel);

#ifdef __cplusplus
}
#endif
#endif
