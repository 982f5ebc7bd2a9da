/*
 * bf_oracle.c -- CPU restatement of BruteFIR's filter path.  TEST INFRASTRUCTURE ONLY:
 * see bf_oracle.h for who may use it and for the parity-pinning status.
 *
 * Follows (file:line in /root/reference):
 *   fftw_convolver.c:170-194, 330-368, 411-433, 482-596, 784-851   op wrappers
 *   fftw_convfuns.h:7-619                                          inner loops
 *   raw2real.h:7-160, real2raw.h:24-250, dither_funs.h:7-114        sample conversion
 *   dither.c:37-138, dither.h:28-38                                 dither table walk
 *   bfconf.c:1979-2019                                              partition split
 *   bfrun.c:1420-2034                                               one block
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bf_oracle.h"

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define R float
#define FN(x) CAT(x, _f)
#define BFO_IS_FLOAT 1
#include "bf_oracle_ops.inc"
#undef R
#undef FN
#undef BFO_IS_FLOAT

#define R double
#define FN(x) CAT(x, _d)
#define BFO_IS_FLOAT 0
#include "bf_oracle_ops.inc"
#undef R
#undef FN
#undef BFO_IS_FLOAT

struct bfo_ctx {
    int L, rs;
    plan_f pf;
    plan_d pd;
    /* dither */
    int8_t *randtab;
    int randtab_size;
    void *randmap_base, *randmap;   /* 511 entries, index -256..254 */
    int n_dstates;
    dstate_f *dsf;
    dstate_d *dsd;
    void *scratch;                   /* 3L reals */
};

/* ================================================================ op level */

bfo_ctx *
bfo_ctx_new(int length, int realsize)
{
    bfo_ctx *c;
    if (realsize != 4 && realsize != 8) {
        fprintf(stderr, "Invalid real size %d.\n", realsize);
        return NULL;
    }
    if (length < 4 || (length & (length - 1)) != 0) {
        fprintf(stderr, "Invalid length %d.\n", length);
        return NULL;
    }
    c = calloc(1, sizeof(*c));
    c->L = length;
    c->rs = realsize;
    if (realsize == 4) plan_init_f(&c->pf, length); else plan_init_d(&c->pd, length);
    c->scratch = calloc(3 * (size_t)length, realsize);
    return c;
}

void
bfo_ctx_free(bfo_ctx *c)
{
    if (c == NULL) return;
    if (c->rs == 4) plan_free_f(&c->pf); else plan_free_d(&c->pd);
    free(c->randtab); free(c->randmap_base); free(c->dsf); free(c->dsd); free(c->scratch);
    free(c);
}

int
bfo_cbufsize(const bfo_ctx *c)
{
    return 2 * c->L * c->rs;
}

void
bfo_raw2real(const bfo_ctx *c, void *real, const void *raw, int bytes, int isfloat,
             int spacing, int swap, int n)
{
    if (c->rs == 4) raw2real_f(real, raw, bytes, isfloat, spacing, swap, n);
    else raw2real_d(real, raw, bytes, isfloat, spacing, swap, n);
}

void
bfo_raw2cbuf(const bfo_ctx *c, const void *rawbuf, void *cbuf, void *next_cbuf,
             const bfo_format *bf)
{
    const size_t half = (size_t)c->L * c->rs;
    bfo_raw2real(c, next_cbuf, (const uint8_t *)rawbuf + bf->byte_offset, bf->bytes,
                 bf->isfloat, bf->sample_spacing, bf->swap, c->L);
    memcpy((uint8_t *)cbuf + half, next_cbuf, half);
}

void
bfo_time2freq(const bfo_ctx *c, const void *in, void *out)
{
    if (c->rs == 4) r2hc_f(&c->pf, in, out); else r2hc_d(&c->pd, in, out);
}

void
bfo_freq2time(const bfo_ctx *c, const void *in, void *out)
{
    if (c->rs == 4) hc2r_f(&c->pf, in, out); else hc2r_d(&c->pd, in, out);
}

void
bfo_mixnscale(const bfo_ctx *c, void *const in[], void *out, const double scales[],
              int n_bufs, int mode)
{
    if (c->rs == 4) mixnscale_f(c->L, in, out, scales, n_bufs, mode);
    else mixnscale_d(c->L, in, out, scales, n_bufs, mode);
}

void
bfo_convolve(const bfo_ctx *c, const void *b, const void *h, void *d)
{
    if (c->rs == 4) convolve_f(c->L, b, h, d); else convolve_d(c->L, b, h, d);
}

void
bfo_convolve_inplace(const bfo_ctx *c, void *b, const void *h)
{
    /* every output element depends only on inputs of its own group of 8, read before
       written, so the out-of-place body is also the in-place one */
    bfo_convolve(c, b, h, b);
}

void
bfo_convolve_add(const bfo_ctx *c, const void *b, const void *h, void *d)
{
    if (c->rs == 4) convolve_add_f(c->L, b, h, d); else convolve_add_d(c->L, b, h, d);
}

void
bfo_dirac_convolve(const bfo_ctx *c, const void *in, void *out)
{
    if (c->rs == 4) dirac_convolve_f(c->L, in, out); else dirac_convolve_d(c->L, in, out);
}

void
bfo_dirac_convolve_inplace(const bfo_ctx *c, void *buf)
{
    bfo_dirac_convolve(c, buf, buf);
}

void
bfo_crossfade_inplace(const bfo_ctx *c, void *input, void *crossfade, void *buffer)
{
    double one = 1.0, inv = 1.0 / (double)(2 * c->L);
    void *p;
    p = crossfade;
    bfo_mixnscale(c, &p, buffer, &one, 1, BFO_MIX_OUTPUT);
    bfo_freq2time(c, buffer, crossfade);
    p = input;
    bfo_mixnscale(c, &p, buffer, &one, 1, BFO_MIX_OUTPUT);
    bfo_freq2time(c, buffer, buffer);
    if (c->rs == 4) fade_f(c->L, crossfade, buffer); else fade_d(c->L, crossfade, buffer);
    bfo_time2freq(c, buffer, buffer);
    p = buffer;
    bfo_mixnscale(c, &p, input, &inv, 1, BFO_MIX_INPUT);
}

void
bfo_convolve_eval(const bfo_ctx *c, const void *in, void *buffer, void *out)
{
    const size_t half = (size_t)c->L * c->rs;
    bfo_freq2time(c, in, (uint8_t *)buffer + half);
    bfo_time2freq(c, buffer, out);
    memcpy(buffer, (uint8_t *)buffer + half, half);
}

int
bfo_coeffs2cbuf(const bfo_ctx *c, const void *coeffs, int n_coeffs, double scale,
                void *dest)
{
    const int L = c->L;
    const int len = n_coeffs > L ? L : n_coeffs;
    double inv = 1.0 / (double)(2 * L);
    void *tmp = calloc(2 * (size_t)L, c->rs), *p = tmp;
    int n, ok = 1;
    if (c->rs == 4) {
        for (n = 0; n < len; n++) {
            float v = ((const float *)coeffs)[n] * (float)scale;
            ((float *)tmp)[L + n] = v;
            if (!isfinite(v)) ok = 0;
        }
    } else {
        for (n = 0; n < len; n++) {
            double v = ((const double *)coeffs)[n] * scale;
            ((double *)tmp)[L + n] = v;
            if (!isfinite(v)) ok = 0;
        }
    }
    if (ok) {
        bfo_time2freq(c, tmp, tmp);
        bfo_mixnscale(c, &p, dest, &inv, 1, BFO_MIX_INPUT);
    } else {
        fprintf(stderr, "NaN or Inf value among coefficients.\n");
    }
    free(tmp);
    return ok;
}

void
bfo_runtime_coeffs2cbuf(const bfo_ctx *c, const void *src, void *dest)
{
    const size_t half = (size_t)c->L * c->rs;
    double inv = 1.0 / (double)(2 * c->L);
    void *tmp = malloc(2 * half), *p = tmp;
    memset(dest, 0, half);
    memcpy((uint8_t *)dest + half, src, half);
    bfo_time2freq(c, dest, tmp);
    bfo_mixnscale(c, &p, dest, &inv, 1, BFO_MIX_INPUT);
    free(tmp);
}

int
bfo_verify_cbuf(const bfo_ctx *c, void *const cbufs[], int n)
{
    int i, k;
    for (i = 0; i < n; i++) {
        for (k = 0; k < 2 * c->L; k++) {
            double v = c->rs == 4 ? (double)((float *)cbufs[i])[k] : ((double *)cbufs[i])[k];
            if (!isfinite(v)) {
                fprintf(stderr, "NaN or Inf value among coefficients.\n");
                return 0;
            }
        }
    }
    return 1;
}

/* ---- dither: Tausworthe generator as published in GSL ("taus"), which is what
   dither.c:37-71 uses, seeded with 1 and warmed up six draws */

static uint32_t
taus_next(uint32_t s[3])
{
    s[0] = ((s[0] & 4294967294U) << 12) ^ (((s[0] << 13) ^ s[0]) >> 19);
    s[1] = ((s[1] & 4294967288U) << 4) ^ (((s[1] << 2) ^ s[1]) >> 25);
    s[2] = ((s[2] & 4294967280U) << 17) ^ (((s[2] << 3) ^ s[2]) >> 11);
    return s[0] ^ s[1] ^ s[2];
}

int
bfo_dither_init(bfo_ctx *c, int n_channels, int sample_rate, int max_size,
                int max_samples_per_loop)
{
    int spacing = 10 * sample_rate, minspacing, n;
    uint32_t s[3];
    minspacing = (sample_rate > max_samples_per_loop) ? sample_rate : max_samples_per_loop;
    if (spacing < minspacing) spacing = minspacing;
    if (max_size > 0 && n_channels * spacing > max_size) spacing = max_size / n_channels;
    if (spacing < minspacing) return 0;
    c->randtab_size = n_channels * spacing + 1;
    c->randtab = malloc(c->randtab_size);
    s[0] = (69069u * 1u) & 0xFFFFFFFFu;
    s[1] = (69069u * s[0]) & 0xFFFFFFFFu;
    s[2] = (69069u * s[1]) & 0xFFFFFFFFu;
    for (n = 0; n < 6; n++) taus_next(s);
    for (n = 0; n < c->randtab_size; n++) c->randtab[n] = (int8_t)(taus_next(s) & 0xFF);

    /* 511 entries in the reference, indexed with int8 - int8 which can be +255: one past
       the table (undefined there).  Entry 255 is defined here by continuing the formula. */
    c->randmap_base = malloc(512 * c->rs);
    c->randmap = (uint8_t *)c->randmap_base + 256 * c->rs;
    if (c->rs == 4) {
        float *m = c->randmap;
        m[-256] = -0.5;
        for (n = -255; n < 256; n++) m[n] = 0.5 + 1.0 / 255.0 + 1.0 / 255.0 * (float)n;
        m[254] = 1.5;
    } else {
        double *m = c->randmap;
        m[-256] = -0.5;
        for (n = -255; n < 256; n++) m[n] = 0.5 + 1.0 / 255.0 + 1.0 / 255.0 * (double)n;
        m[254] = 1.5;
    }
    c->n_dstates = n_channels;
    c->dsf = calloc(n_channels, sizeof(dstate_f));
    c->dsd = calloc(n_channels, sizeof(dstate_d));
    for (n = 0; n < n_channels; n++) {
        c->dsf[n].randtab_ptr = c->dsd[n].randtab_ptr = n * spacing + 1;
    }
    return 1;
}

const int8_t *
bfo_dither_table(const bfo_ctx *c, int *size)
{
    *size = c->randtab_size;
    return c->randtab;
}

int
bfo_dither_randtab_ptr(const bfo_ctx *c, int channel)
{
    return c->rs == 4 ? c->dsf[channel].randtab_ptr : c->dsd[channel].randtab_ptr;
}

int
bfo_cbuf2raw(bfo_ctx *c, const void *cbuf, void *outbuf, const bfo_format *bf,
             int dither_channel, bfo_overflow *of, double safety_limit)
{
    uint8_t *raw = (uint8_t *)outbuf + bf->byte_offset;
    const int bits = bf->sbytes << 3, L = c->L;
    if (dither_channel >= 0 && !bf->isfloat) {
        /* dither_preloop_real2int_hp_tpdf, dither.h:28-38 */
        int *ptr = c->rs == 4 ? &c->dsf[dither_channel].randtab_ptr
                              : &c->dsd[dither_channel].randtab_ptr;
        const int8_t *tab;
        if (*ptr + L >= c->randtab_size) {
            c->randtab[0] = c->randtab[*ptr - 1];
            *ptr = 1;
        }
        tab = &c->randtab[*ptr];
        *ptr += L;
        if (c->rs == 4) {
            c->dsf[dither_channel].randtab = tab;
            return real2raw_f(raw, cbuf, bits, bf->bytes, bf->isfloat, bf->sample_spacing,
                              bf->swap, L, of, safety_limit, &c->dsf[dither_channel],
                              c->randmap);
        }
        c->dsd[dither_channel].randtab = tab;
        return real2raw_d(raw, cbuf, bits, bf->bytes, bf->isfloat, bf->sample_spacing,
                          bf->swap, L, of, safety_limit, &c->dsd[dither_channel], c->randmap);
    }
    if (c->rs == 4) {
        return real2raw_f(raw, cbuf, bits, bf->bytes, bf->isfloat, bf->sample_spacing,
                          bf->swap, L, of, safety_limit, NULL, NULL);
    }
    return real2raw_d(raw, cbuf, bits, bf->bytes, bf->isfloat, bf->sample_spacing, bf->swap,
                      L, of, safety_limit, NULL, NULL);
}

/* ================================================ integer delay (delay.c:78-340) */

struct bfo_delay {
    int F, ss;            /* fragment size in samples, bytes per sample */
    int maxdelay, curdelay, cur, n_full, n_full_cap, n_rest;
    uint8_t **full;       /* whole-fragment buffers */
    uint8_t *rest;        /* rest buffer */
    uint8_t *shrt[2];     /* the two "short" buffers used while delay <= fragment */
};

static uint8_t *
zbytes(size_t n)
{
    return calloc(n ? n : 1, 1);
}

bfo_delay *
bfo_delay_new(int fragment_size, int initdelay, int maxdelay, int sample_size)
{
    bfo_delay *d = calloc(1, sizeof(*d));
    const size_t frag = (size_t)fragment_size * sample_size;
    int delay, n;
    d->F = fragment_size; d->ss = sample_size;
    delay = maxdelay <= 0 ? initdelay : maxdelay;                 /* delay.c:357-360 */
    if (maxdelay >= 0 && delay > maxdelay) delay = initdelay = maxdelay;
    d->curdelay = initdelay;
    d->maxdelay = maxdelay;
    if (delay == 0) return d;
    if (delay <= fragment_size) {                                 /* :365-374 */
        d->n_rest = initdelay;
        d->shrt[0] = zbytes((size_t)delay * sample_size);
        d->shrt[1] = zbytes((size_t)delay * sample_size);
        return d;
    }
    if (maxdelay > 0) { d->shrt[0] = zbytes(frag); d->shrt[1] = zbytes(frag); }
    d->n_rest = initdelay % fragment_size;
    d->n_full = initdelay / fragment_size + 1;
    if (d->n_full == 1) d->n_full = 0;
    d->n_full_cap = delay / fragment_size + 1;
    d->full = calloc(d->n_full_cap, sizeof(uint8_t *));
    for (n = 0; n < d->n_full_cap; n++) d->full[n] = zbytes(frag);
    if (maxdelay > 0) d->rest = zbytes(frag);
    else if (d->n_rest != 0) d->rest = zbytes((size_t)d->n_rest * sample_size);
    return d;
}

void
bfo_delay_free(bfo_delay *d)
{
    int n;
    if (d == NULL) return;
    for (n = 0; n < d->n_full_cap; n++) free(d->full[n]);
    free(d->full); free(d->rest); free(d->shrt[0]); free(d->shrt[1]); free(d);
}

static void
delay_retarget(bfo_delay *d, int newdelay)                        /* change_delay, :283-318 */
{
    int i;
    if (newdelay == d->curdelay || newdelay > d->maxdelay) return;
    if (newdelay <= d->F) {
        d->n_rest = newdelay;
        if (d->curdelay > d->F || d->curdelay < newdelay) {
            memset(d->shrt[0], 0, (size_t)newdelay * d->ss);
            memset(d->shrt[1], 0, (size_t)newdelay * d->ss);
        }
        d->n_full = 0; d->cur = 0; d->curdelay = newdelay;
        return;
    }
    d->n_rest = newdelay % d->F;
    d->n_full = newdelay / d->F + 1;
    if (d->curdelay < newdelay) {
        for (i = 0; i < d->n_full; i++) memset(d->full[i], 0, (size_t)d->F * d->ss);
        if (d->n_rest != 0) memset(d->rest, 0, (size_t)d->n_rest * d->ss);
    }
    d->cur = 0; d->curdelay = newdelay;
}

void
bfo_delay_update(bfo_delay *d, void *buf_, int delay)
{
    uint8_t *buf = buf_;
    const size_t ss = d->ss, F = d->F, r = d->n_rest;
    delay_retarget(d, delay);
    if (d->n_full > 0) {                                          /* update_delay_buffer */
        uint8_t *last = d->cur == d->n_full - 1 ? d->full[0] : d->full[d->cur + 1];
        const size_t rr = d->n_rest;
        memcpy(d->full[d->cur], buf, F * ss);
        if (rr != 0) {
            memcpy(buf, d->rest, rr * ss);
            memcpy(d->rest, last + (F - rr) * ss, rr * ss);
        }
        memcpy(buf + rr * ss, last, (F - rr) * ss);
        if (++d->cur == d->n_full) d->cur = 0;
    } else if (d->n_rest > 0) {                                   /* update_delay_short_buffer */
        const size_t rr = d->n_rest;
        memcpy(d->shrt[d->cur], buf + (F - rr) * ss, rr * ss);
        memmove(buf + rr * ss, buf, (F - rr) * ss);
        d->cur = !d->cur;
        memcpy(buf, d->shrt[d->cur], rr * ss);
    }
    (void)r;
}

/* ================================================ Kaiser window (firwindow.c) */

static double
bessel_i0(double x)
{
    double n = 1.0, a = 1.0, sum = 1.0;
    const double h = x / 2.0;
    do {
        a *= h; a /= n;
        sum += a * a;
        n += 1.0;
    } while (a != 0.0 && isfinite(sum));
    return sum;
}

static double
kaiser_at(double x, double beta, double inv)
{
    if (x < -1.0) x = -1.0;
    if (x > 1.0) x = 1.0;
    return bessel_i0(beta * sqrt(1.0 - x * x)) * inv;
}

static void
wmul(void *t, int realsize, int n, double y)
{
    if (realsize == 4) ((float *)t)[n] *= y; else ((double *)t)[n] *= y;
}

void
bfo_firwindow_kaiser(void *target, int len, double offset, double beta, int realsize)
{
    const int half = len >> 1;
    const double inv = 1.0 / bessel_i0(beta);
    int n, max;
    double step;
    if (offset != 0.0) {
        max = half + (int)floor(offset);
        offset -= floor(offset);
        if (fabs(offset) < 1e-20) offset = 0.0;
        step = 1.0 / ((double)max + offset);
        if (offset == 0.0) max -= 1;
        for (n = 0; n <= max; n++) {
            const double y = kaiser_at(-1.0 + (double)n * step, beta, inv);
            wmul(target, realsize, n, y);         /* applied twice in the reference (:105-110) */
            wmul(target, realsize, n, y);
        }
        if (offset == 0.0) max += 1;
        step = 1.0 / ((double)(len - max - 1) - offset);
        for (; n < len; n++) {
            const double y = kaiser_at(((double)(n - max) - offset) * step, beta, inv);
            wmul(target, realsize, n, y);
            wmul(target, realsize, n, y);
        }
    } else if (len & 1) {
        step = 1.0 / (double)half;
        for (n = 1; n <= half; n++) {
            const double y = kaiser_at((double)n * step, beta, inv);
            wmul(target, realsize, half + n, y);
            wmul(target, realsize, half - n, y);
        }
    } else {
        step = 1.0 / (double)half;
        step *= (double)half / ((double)half - 0.5);
        for (n = 1; n <= half; n++) {
            const double y = kaiser_at(((double)n - 0.5) * step, beta, inv);
            wmul(target, realsize, half + n - 1, y);
            wmul(target, realsize, half - n, y);
        }
    }
}

/* ============================ td_conv + sub-sample delay (fftw_convolver.c:682-782, delay.c:416-505) */

typedef struct {
    bfo_ctx *fc;          /* FFT context of length blocklen (transforms of 2*blocklen reals) */
    int blocklen;
    void *coeffs;         /* halfcomplex, scaled by 1/(2*blocklen) */
} otd;

static int
td_block_length(int n)
{
    int o = 0;
    if (n < 1) return -1;
    while ((1 << o) < n) o++;
    return 1 << o;
}

static otd *
td_new(const void *coeffs, int n_coeffs, int rs)
{
    otd *t = calloc(1, sizeof(*t));
    int n;
    t->blocklen = td_block_length(n_coeffs);
    t->fc = bfo_ctx_new(t->blocklen < 4 ? 4 : t->blocklen, rs);
    t->coeffs = calloc(2 * (size_t)t->blocklen, rs);
    memcpy((uint8_t *)t->coeffs + (size_t)t->blocklen * rs, coeffs, (size_t)n_coeffs * rs);
    bfo_time2freq(t->fc, t->coeffs, t->coeffs);
    for (n = 0; n < 2 * t->blocklen; n++) {
        if (rs == 4) ((float *)t->coeffs)[n] *= 1.0 / (float)(t->blocklen << 1);   /* scalef, :724 */
        else ((double *)t->coeffs)[n] *= 1.0 / (double)(t->blocklen << 1);
    }
    return t;
}

static void
td_convolve(const otd *t, void *blk, int rs)
{
    const int size = t->blocklen << 1, half = size >> 1;
    int n;
    bfo_time2freq(t->fc, blk, blk);
    if (rs == 4) {                                   /* convolve_inplace_ordered, :738-765 */
        float *b = blk; const float *c = t->coeffs;
        b[0] *= c[0];
        for (n = 1; n < half; n++) {
            const float a = b[n];
            b[n] = a * c[n] - b[size - n] * c[size - n];
            b[size - n] = a * c[size - n] + b[size - n] * c[n];
        }
        b[half] *= c[half];
    } else {
        double *b = blk; const double *c = t->coeffs;
        b[0] *= c[0];
        for (n = 1; n < half; n++) {
            const double a = b[n];
            b[n] = a * c[n] - b[size - n] * c[size - n];
            b[size - n] = a * c[size - n] + b[size - n] * c[n];
        }
        b[half] *= c[half];
    }
    bfo_freq2time(t->fc, blk, blk);
}

typedef struct {
    int steps, flen, fbsize, rs;
    otd **bank;           /* index -steps+1 .. steps-1 via bank[steps + i] */
} osubdelay;

static osubdelay *
subdelay_new(int step_count, int half_len, double beta, int fragment, int rs)
{
    osubdelay *sd = calloc(1, sizeof(*sd));
    void *f;
    int i, n;
    (void)beta;                                      /* the reference passes 9, not beta (delay.c:79) */
    sd->steps = step_count; sd->rs = rs;
    sd->flen = 2 * half_len + 1;
    sd->fbsize = td_block_length(sd->flen);
    if (fragment % sd->fbsize != 0) { free(sd); return NULL; }
    sd->bank = calloc(2 * step_count + 1, sizeof(otd *));
    f = calloc(sd->flen, rs);
    if (rs == 4) ((float *)f)[sd->flen >> 1] = 1.0f; else ((double *)f)[sd->flen >> 1] = 1.0;
    sd->bank[step_count] = td_new(f, sd->flen, rs);
    for (i = -step_count + 1; i < step_count; i++) {
        const double offset = (double)i / step_count;
        if (i == 0) continue;
        for (n = 0; n < sd->flen; n++) {             /* sample_sinc, delay.c:56-76 */
            const double x = M_PI * ((double)(n - half_len) - offset);
            const double v = x == 0.0 ? 1.0 : sin(x) / x;
            if (rs == 4) ((float *)f)[n] = (float)v; else ((double *)f)[n] = v;
        }
        bfo_firwindow_kaiser(f, sd->flen, offset, 9, rs);
        sd->bank[step_count + i] = td_new(f, sd->flen, rs);
    }
    free(f);
    return sd;
}

/* delay_subsample_update, delay.c:416-442 */
static void
subdelay_update(const osubdelay *sd, void *buf, void *rest, int subdelay, int fragment)
{
    const size_t bs = (size_t)sd->fbsize * sd->rs;
    uint8_t *cb = malloc(2 * bs);
    size_t i;
    if (subdelay <= -sd->steps || subdelay >= sd->steps) { free(cb); return; }
    for (i = 0; i < (size_t)fragment * sd->rs; i += bs) {
        memcpy(cb, rest, bs);
        memcpy(cb + bs, (uint8_t *)buf + i, bs);
        memcpy(rest, cb + bs, bs);
        td_convolve(sd->bank[sd->steps + subdelay], cb, sd->rs);
        memcpy((uint8_t *)buf + i, cb, bs);
    }
    free(cb);
}

/* ============================================================== block level */

typedef struct {
    int n_in_ch, *in_ch;
    int n_in_f, *in_f;
    int n_out_ch, *out_ch;
    int crossfade;
    /* struct bffilter_control */
    int coeff, delayblocks;
    double *scale_in, *scale_out, *fscale;
    /* filter_process() state */
    void **ring;        /* cbuf[n][N]   */
    void *ocbuf;        /* ocbuf[n]     */
    void *evalbuf;      /* 1.5 cbuf     */
    int prevcoeff;
    int procblocks;     /* bfrun.c:1084, 1567-1571 */
} ofilter;

typedef struct {
    int n_blocks;
    void **part;        /* coeffs_data[c][i] */
} ocoeff;

struct bfo_engine {
    bfo_ctx *c;
    int L, N, rs, n_ch[2];
    bfo_format *fmt[2];
    int *dither_ch;                 /* per output: dither state index or -1 */
    bfo_overflow *overflow;
    double safety_limit;
    double powersave;              /* 0 = off, >= 1.0 exact zero test, else linear noise floor */
    int n_filters, n_coeffs;
    ofilter *f;
    ocoeff *co;
    void **in_time[2];              /* input_timecbuf[ch][2] */
    void **in_freq, **out_freq;     /* shm spectra in the reference */
    void *static_eval, *xfade[2], *tmp_out;
    int curbuf;
    unsigned int blockcounter;
    /* virtual -> physical channel mapping (bfconf->virt2phys, n_virtperphys); 1:1 by default.
       fmt[], dither_ch[] are indexed by PHYSICAL channel, everything else by virtual */
    int n_phys[2];
    int *v2p[2], *n_vpp[2];
    int *delay[2], *maxdelay[2], *muted[2];
    bfo_delay **db[2];              /* input_db / output_db, bfrun.c:1059-1060 */
    void *incopy, *mixbuf;
    /* sub-sample delay: bfconf->use_subdelay / subdelay[][] / sdf_length */
    osubdelay *sd;
    int sdf_length;
    int *subdelay[2];
    void **sd_rest[2];
};

static void *
zalloc(const bfo_engine *e, int halves)
{
    return calloc((size_t)halves * e->L, e->rs);
}

bfo_engine *
bfo_engine_new(int length, int n_blocks, int realsize, int n_in, int n_out)
{
    bfo_engine *e;
    bfo_ctx *c = bfo_ctx_new(length, realsize);
    int io, n;
    if (c == NULL || n_blocks < 1) return NULL;
    e = calloc(1, sizeof(*e));
    e->c = c; e->L = length; e->N = n_blocks; e->rs = realsize;
    e->n_ch[0] = n_in; e->n_ch[1] = n_out;
    for (io = 0; io < 2; io++) {
        e->fmt[io] = calloc(e->n_ch[io], sizeof(bfo_format));
        for (n = 0; n < e->n_ch[io]; n++) {
            /* default: non-interleaved native float of the working precision */
            bfo_format *b = &e->fmt[io][n];
            b->isfloat = 1; b->bytes = b->sbytes = realsize; b->scale = 1.0;
            b->sample_spacing = 1; b->byte_offset = n * length * realsize;
        }
    }
    for (io = 0; io < 2; io++) {
        e->n_phys[io] = e->n_ch[io];
        e->v2p[io] = malloc(e->n_ch[io] * sizeof(int));
        e->n_vpp[io] = malloc(e->n_ch[io] * sizeof(int));
        e->delay[io] = calloc(e->n_ch[io], sizeof(int));
        e->maxdelay[io] = calloc(e->n_ch[io], sizeof(int));
        e->muted[io] = calloc(e->n_ch[io], sizeof(int));
        e->db[io] = calloc(e->n_ch[io], sizeof(bfo_delay *));
        e->subdelay[io] = malloc(e->n_ch[io] * sizeof(int));
        e->sd_rest[io] = calloc(e->n_ch[io], sizeof(void *));
        { int q; for (q = 0; q < e->n_ch[io]; q++) e->subdelay[io][q] = -100; }   /* BF_UNDEFINED_SUBDELAY */
        for (n = 0; n < e->n_ch[io]; n++) { e->v2p[io][n] = n; e->n_vpp[io][n] = 1; }
    }
    e->incopy = calloc((size_t)length, 8);
    e->mixbuf = calloc((size_t)length, realsize);
    e->dither_ch = malloc(n_out * sizeof(int));
    e->overflow = calloc(n_out, sizeof(bfo_overflow));
    for (n = 0; n < n_out; n++) { e->dither_ch[n] = -1; e->overflow[n].max = 1.0; }
    e->in_time[0] = calloc(n_in, sizeof(void *));
    e->in_time[1] = calloc(n_in, sizeof(void *));
    e->in_freq = calloc(n_in, sizeof(void *));
    e->out_freq = calloc(n_out, sizeof(void *));
    for (n = 0; n < n_in; n++) {
        e->in_time[0][n] = zalloc(e, 2); e->in_time[1][n] = zalloc(e, 2);
        e->in_freq[n] = zalloc(e, 2);
    }
    for (n = 0; n < n_out; n++) e->out_freq[n] = zalloc(e, 2);
    e->static_eval = zalloc(e, 2);
    e->xfade[0] = zalloc(e, 2); e->xfade[1] = zalloc(e, 2);
    e->tmp_out = zalloc(e, 2);
    return e;
}

void
bfo_engine_free(bfo_engine *e)
{
    int n, i;
    if (e == NULL) return;
    for (n = 0; n < e->n_filters; n++) {
        ofilter *f = &e->f[n];
        if (e->N > 1) { for (i = 0; i < e->N; i++) free(f->ring[i]); }
        free(f->ocbuf); free(f->ring); free(f->evalbuf);
        free(f->in_ch); free(f->in_f); free(f->out_ch);
        free(f->scale_in); free(f->scale_out); free(f->fscale);
    }
    for (n = 0; n < e->n_coeffs; n++) {
        for (i = 0; i < e->co[n].n_blocks; i++) free(e->co[n].part[i]);
        free(e->co[n].part);
    }
    for (n = 0; n < e->n_ch[0]; n++) {
        free(e->in_time[0][n]); free(e->in_time[1][n]); free(e->in_freq[n]);
    }
    for (n = 0; n < e->n_ch[1]; n++) free(e->out_freq[n]);
    free(e->in_time[0]); free(e->in_time[1]); free(e->in_freq); free(e->out_freq);
    free(e->static_eval); free(e->xfade[0]); free(e->xfade[1]); free(e->tmp_out);
    for (n = 0; n < 2; n++) {
        for (i = 0; i < e->n_ch[n]; i++) bfo_delay_free(e->db[n][i]);
        free(e->v2p[n]); free(e->n_vpp[n]); free(e->delay[n]); free(e->maxdelay[n]);
        free(e->muted[n]); free(e->db[n]);
    }
    free(e->incopy); free(e->mixbuf);
    free(e->fmt[0]); free(e->fmt[1]); free(e->dither_ch); free(e->overflow);
    free(e->f); free(e->co);
    bfo_ctx_free(e->c);
    free(e);
}

void
bfo_engine_set_format(bfo_engine *e, int io, int ch, const bfo_format *bf)
{
    int v;
    e->fmt[io][ch] = *bf;                       /* ch is a PHYSICAL channel */
    if (io == 1) {
        /* bfrun.c:2270-2277: every virtual channel of that physical one */
        for (v = 0; v < e->n_ch[1]; v++) {
            if (e->v2p[1][v] != ch) continue;
            memset(&e->overflow[v], 0, sizeof(bfo_overflow));
            e->overflow[v].max = bf->isfloat ? 1.0
                : (double)((uint64_t)1 << ((bf->sbytes << 3) - 1)) - 1;
        }
    }
}

/* test_silent, bfrun.c:721-771 (n_reals = the whole 2L window).  analog_powersave >= 1.0: the
   window is silent iff every byte is zero (memiszero, :696-719: -0.0 is NOT zero); else iff
   scale * max|x| < analog_powersave, and then the window is made truly zero. */
static int
test_silent(void *buf, int n_reals, int rs, double analog_powersave, double scale)
{
    int n;
    double dmax;
    if (analog_powersave >= 1.0) {
        const uint8_t *b = buf;
        uint8_t acc = 0;
        size_t i;
        for (i = 0; i < (size_t)n_reals * rs; i++) acc |= b[i];
        return acc == 0;
    }
    if (rs == 4) {
        float fmax = 0;
        for (n = 0; n < n_reals; n++) {
            const float v = ((float *)buf)[n];
            if (v < 0) { if (-v > fmax) fmax = -v; } else { if (v > fmax) fmax = v; }
        }
        dmax = fmax;
    } else {
        dmax = 0;
        for (n = 0; n < n_reals; n++) {
            const double v = ((double *)buf)[n];
            if (v < 0) { if (-v > dmax) dmax = -v; } else { if (v > dmax) dmax = v; }
        }
    }
    if (scale * dmax >= analog_powersave) return 0;
    memset(buf, 0, (size_t)n_reals * rs);
    return 1;
}

void bfo_engine_set_powersave(bfo_engine *e, double analog_powersave) { e->powersave = analog_powersave; }

void
bfo_engine_set_safety_limit(bfo_engine *e, double limit)
{
    e->safety_limit = limit;
}

int
bfo_engine_enable_dither(bfo_engine *e, const int out_channels[], int n, int sample_rate,
                         int max_size)
{
    int i;
    if (!bfo_dither_init(e->c, n, sample_rate, max_size, e->L)) return 0;
    for (i = 0; i < n; i++) e->dither_ch[out_channels[i]] = i;
    return 1;
}

int
bfo_engine_map_channels(bfo_engine *e, int io, int n_phys, const int virt2phys[])
{
    int v;
    if (n_phys < 1 || n_phys > e->n_ch[io]) return 0;
    for (v = 0; v < e->n_phys[io]; v++) e->n_vpp[io][v] = 0;
    for (v = 0; v < n_phys; v++) e->n_vpp[io][v] = 0;
    for (v = 0; v < e->n_ch[io]; v++) {
        if (virt2phys[v] < 0 || virt2phys[v] >= n_phys) return 0;
        e->v2p[io][v] = virt2phys[v];
        e->n_vpp[io][virt2phys[v]]++;
    }
    e->n_phys[io] = n_phys;
    return 1;
}

void bfo_engine_set_delay(bfo_engine *e, int io, int ch, int delay) { e->delay[io][ch] = delay; }
void bfo_engine_set_maxdelay(bfo_engine *e, int io, int ch, int maxdelay) { e->maxdelay[io][ch] = maxdelay; }
void bfo_engine_set_mute(bfo_engine *e, int io, int ch, int muted) { e->muted[io][ch] = muted; }

int
bfo_engine_enable_subdelay(bfo_engine *e, int sdf_length, double beta)
{
    if (sdf_length <= 0 || 2 * sdf_length + 1 > e->L) return 0;       /* bfconf.c:2796-2805 */
    e->sd = subdelay_new(100, sdf_length, beta, e->L, e->rs);         /* BF_SAMPLE_SLOTS */
    e->sdf_length = sdf_length;
    return e->sd != NULL;
}

void bfo_engine_set_subdelay(bfo_engine *e, int io, int ch, int subdelay) { e->subdelay[io][ch] = subdelay; }

/* What filter_process() hands delay_allocate_buffer() for a channel that shares a physical one
   (bfrun.c:1152-1162, 1185-1197): delay + extra and maxdelay + extra, extra = the sub-sample filter's
   integer part on channels without a filter of their own.  Where that leaves the reference's own
   buffer too small -- maxdelay -1 becomes the limit extra - 1, below the delay; delay.c:357-374 then
   allocates for the limit and fills for the delay: a heap overrun in the reference -- the engines
   agree on: a negative maxdelay stays negative (fixed delay), a delay above a positive limit starts
   at the limit. */
static void
vdelay_limits(int delay, int maxdelay, int extra, int *init_eff, int *max_eff)
{
    *max_eff = maxdelay < 0 ? maxdelay : maxdelay + extra;
    *init_eff = delay + extra;
    if (*max_eff > 0 && *init_eff > *max_eff) *init_eff = *max_eff;
}

static int
side_uses_subdelay(const bfo_engine *e, int io)
{
    int n;
    if (e->sd == NULL) return 0;
    for (n = 0; n < e->n_ch[io]; n++) if (e->subdelay[io][n] != -100) return 1;
    return 0;
}

static void
apply_subdelay(bfo_engine *e, int io, int ch, void *buf)
{
    if (e->sd == NULL || e->subdelay[io][ch] == -100) {
        if (e->sd_rest[io][ch] == NULL) return;
    }
    if (e->sd_rest[io][ch] == NULL) {
        if (e->sd == NULL || e->subdelay[io][ch] == -100) return;     /* bfrun.c:1133-1142: decided at start */
        e->sd_rest[io][ch] = calloc(e->sd->fbsize, e->rs);
    }
    subdelay_update(e->sd, buf, e->sd_rest[io][ch], e->subdelay[io][ch], e->L);
}

int
bfo_engine_add_coeff(bfo_engine *e, const void *taps, int n_taps, double scale, int n_blocks)
{
    const int L = e->L;
    ocoeff *co;
    int n;
    if (n_blocks <= 0) n_blocks = (n_taps + L - 1) / L;
    if (n_blocks < 1) n_blocks = 1;
    if (n_blocks > e->N) return -1;        /* bfconf.c:2827-2832 */
    e->co = realloc(e->co, (e->n_coeffs + 1) * sizeof(ocoeff));
    co = &e->co[e->n_coeffs];
    co->n_blocks = n_blocks;
    co->part = calloc(n_blocks, sizeof(void *));
    if (n_taps > n_blocks * L) n_taps = n_blocks * L;
    for (n = 0; n < n_blocks; n++) {
        int len = n_taps - n * L;
        if (len < 0) len = 0;
        if (len > L) len = L;
        co->part[n] = zalloc(e, 2);
        if (!bfo_coeffs2cbuf(e->c, (const uint8_t *)taps + (size_t)n * L * e->rs, len, scale,
                             co->part[n])) {
            return -1;
        }
    }
    return e->n_coeffs++;
}

static double *
dupd(const double *src, int n)
{
    double *d = malloc((n > 0 ? n : 1) * sizeof(double));
    if (n > 0) memcpy(d, src, n * sizeof(double));
    return d;
}

static int *
dupi(const int *src, int n)
{
    int *d = malloc((n > 0 ? n : 1) * sizeof(int));
    if (n > 0) memcpy(d, src, n * sizeof(int));
    return d;
}

int
bfo_engine_add_filter(bfo_engine *e,
                      int n_in_ch, const int in_ch[], const double in_scale[],
                      int n_in_f, const int in_f[], const double in_fscale[],
                      int n_out_ch, const int out_ch[], const double out_scale[],
                      int coeff, int delayblocks, int crossfade)
{
    ofilter *f;
    int i;
    for (i = 0; i < n_in_f; i++) {
        if (in_f[i] < 0 || in_f[i] >= e->n_filters) return -1;   /* must precede */
    }
    e->f = realloc(e->f, (e->n_filters + 1) * sizeof(ofilter));
    f = &e->f[e->n_filters];
    memset(f, 0, sizeof(*f));
    f->n_in_ch = n_in_ch; f->in_ch = dupi(in_ch, n_in_ch); f->scale_in = dupd(in_scale, n_in_ch);
    f->n_in_f = n_in_f; f->in_f = dupi(in_f, n_in_f); f->fscale = dupd(in_fscale, n_in_f);
    f->n_out_ch = n_out_ch; f->out_ch = dupi(out_ch, n_out_ch);
    f->scale_out = dupd(out_scale, n_out_ch);
    f->coeff = coeff; f->delayblocks = delayblocks; f->crossfade = crossfade;
    f->prevcoeff = coeff;                                   /* bfrun.c:1326 */
    f->ring = calloc(e->N, sizeof(void *));
    f->ocbuf = zalloc(e, 2);
    if (e->N > 1) {
        for (i = 0; i < e->N; i++) f->ring[i] = zalloc(e, 2);
    } else {
        f->ring[0] = f->ocbuf;                              /* bfrun.c:1290 */
    }
    if (n_in_f > 0) f->evalbuf = zalloc(e, 3);
    return e->n_filters++;
}

void bfo_engine_set_coeff(bfo_engine *e, int filter, int coeff) { e->f[filter].coeff = coeff; }
void bfo_engine_set_delayblocks(bfo_engine *e, int filter, int d) { e->f[filter].delayblocks = d; }

void
bfo_engine_set_scale(bfo_engine *e, int filter, int io, int index, double scale)
{
    if (io == 0) e->f[filter].scale_in[index] = scale;
    else e->f[filter].scale_out[index] = scale;
}

void
bfo_engine_set_fscale(bfo_engine *e, int filter, int index, double scale)
{
    e->f[filter].fscale[index] = scale;
}

static int
blocks_of(const bfo_engine *e, int coeff, int delay)
{
    /* bfrun.c:1585-1598 */
    if (coeff < 0 || e->co[coeff].n_blocks > e->N - delay) return e->N - delay;
    return e->co[coeff].n_blocks;
}

int
bfo_engine_block(bfo_engine *e, const void *rawin, void *rawout)
{
    const bfo_ctx *c = e->c;
    const unsigned int N = (unsigned int)e->N, t = e->blockcounter;
    const int cur = e->curbuf;
    int n, i, status = 0;

    /* bfrun.c:1494-1560: raw -> sliding window -> spectrum, per input channel */
    for (n = 0; n < e->n_ch[0]; n++) {
        const int ph = e->v2p[0][n];
        const bfo_format *bf = &e->fmt[0][ph];
        const size_t halfb = (size_t)e->L * e->rs;
        if (e->n_vpp[0][ph] == 1) {
            /* convolver_raw2cbuf with the apply_subdelay post-process (bfrun.c:1503-1508) */
            bfo_raw2real(c, e->in_time[!cur][n], (const uint8_t *)rawin + bf->byte_offset, bf->bytes,
                         bf->isfloat, bf->sample_spacing, bf->swap, e->L);
            apply_subdelay(e, 0, n, e->in_time[!cur][n]);
            memcpy((uint8_t *)e->in_time[cur][n] + halfb, e->in_time[!cur][n], halfb);
        } else {
            /* :1509-1531: several virtual inputs share a physical one: private copy of the raw
               samples, integer delay or mute applied here (dai.c does it for 1:1 channels) */
            bfo_format cf = *bf;
            cf.sample_spacing = 1; cf.byte_offset = 0;
            /* channels without a sub-sample filter are delayed by its integer part
               (bfrun.c:1152-1162, 1512-1516) */
            const int extra = (side_uses_subdelay(e, 0) && e->subdelay[0][n] == -100) ? e->sdf_length : 0;
            if (e->db[0][n] == NULL) {
                /* the delay buffer exists from the start, with the delay configured then (bfrun.c:1128-1166),
                   muted or not -- not from the first block the channel is heard in */
                int d0, m0;
                vdelay_limits(e->delay[0][n], e->maxdelay[0][n], extra, &d0, &m0);
                e->db[0][n] = bfo_delay_new(e->L, d0, m0, bf->bytes);
            }
            if (!e->muted[0][n]) {
                const uint8_t *src = (const uint8_t *)rawin + bf->byte_offset;
                const size_t st = (size_t)bf->sample_spacing * bf->bytes;
                for (i = 0; i < e->L; i++) memcpy((uint8_t *)e->incopy + (size_t)i * bf->bytes, src + i * st, bf->bytes);
                bfo_delay_update(e->db[0][n], e->incopy, e->delay[0][n] + extra);
            } else {
                memset(e->incopy, 0, (size_t)e->L * bf->bytes);
            }
            bfo_raw2real(c, e->in_time[!cur][n], e->incopy, cf.bytes, cf.isfloat, 1, cf.swap, e->L);
            apply_subdelay(e, 0, n, e->in_time[!cur][n]);
            memcpy((uint8_t *)e->in_time[cur][n] + halfb, e->in_time[!cur][n], halfb);
        }
        /* bfrun.c:1541-1553 + test_silent (:721-771): with powersave a silent 2L window is made
           truly zero and its spectrum is zero instead of transformed */
        if (e->powersave > 0.0 &&
            test_silent(e->in_time[cur][n], 2 * e->L, e->rs, e->powersave, e->fmt[0][e->v2p[0][n]].scale)) {
            memset(e->in_freq[n], 0, (size_t)2 * e->L * e->rs);
        } else {
            bfo_time2freq(c, e->in_time[cur][n], e->in_freq[n]);
        }
    }

    /* bfrun.c:1566-1844: every filter, in order */
    for (n = 0; n < e->n_filters; n++) {
        ofilter *f = &e->f[n];
        const int coeff = f->coeff, prev = f->prevcoeff;
        const int fading = f->crossfade && prev != coeff;
        int delay = f->delayblocks, cblocks, prevcblocks, wslot, rslot, nmix;
        void *mix_in[f->n_in_ch + 1];
        double scales[f->n_in_ch + 1];

        if (f->procblocks < e->N) f->procblocks++;            /* :1567-1569 */
        if (delay < 0) delay = 0; else if (delay > e->N - 1) delay = e->N - 1;
        cblocks = blocks_of(e, coeff, delay);
        prevcblocks = blocks_of(e, prev, delay);
        wslot = (int)((t + (unsigned int)delay) % N);

        nmix = f->n_in_ch;
        for (i = 0; i < f->n_in_ch; i++) {
            mix_in[i] = e->in_freq[f->in_ch[i]];
            scales[i] = f->scale_in[i] * e->fmt[0][e->v2p[0][f->in_ch[i]]].scale;   /* :1641, virtscales */
        }
        if (f->n_in_f > 0) {
            /* :1603-1649 filter inputs: mix upstream outputs, re-window in time domain */
            void *up[f->n_in_f];
            for (i = 0; i < f->n_in_f; i++) up[i] = e->f[f->in_f[i]].ocbuf;
            bfo_mixnscale(c, up, e->static_eval, f->fscale, f->n_in_f, BFO_MIX_OUTPUT);
            bfo_convolve_eval(c, e->static_eval, f->evalbuf, e->static_eval);
            mix_in[nmix] = e->static_eval;
            scales[nmix] = 1.0;
            nmix++;
        }
        bfo_mixnscale(c, mix_in, f->ring[wslot], scales, nmix, BFO_MIX_INPUT);

        rslot = (int)(t % N);
        if (e->N == 1) {
            /* :1692-1723 and :1780-1800, ring[0] aliases ocbuf */
            if (fading) {
                if (prev < 0) bfo_dirac_convolve(c, f->ring[0], e->xfade[0]);
                else bfo_convolve(c, f->ring[0], e->co[prev].part[0], e->xfade[0]);
            }
            if (coeff >= 0) bfo_convolve_inplace(c, f->ring[0], e->co[coeff].part[0]);
            else bfo_dirac_convolve_inplace(c, f->ring[0]);
            if (fading) bfo_crossfade_inplace(c, f->ring[0], e->xfade[0], e->xfade[1]);
        } else {
            /* :1725-1777 and :1802-1835 */
            if (fading) {
                if (prev < 0) bfo_dirac_convolve(c, f->ring[rslot], e->xfade[0]);
                else bfo_convolve(c, f->ring[rslot], e->co[prev].part[0], e->xfade[0]);
            }
            if (coeff >= 0) {
                bfo_convolve(c, f->ring[rslot], e->co[coeff].part[0], f->ocbuf);
                /* :1745: never reach further back than blocks processed so far (with
                   N not a power of two the unsigned wrap of t - i would otherwise land
                   on a live slot) */
                for (i = 1; i < cblocks && i < f->procblocks; i++) {
                    const int j = (int)((t - (unsigned int)i) % N);
                    bfo_convolve_add(c, f->ring[j], e->co[coeff].part[i], f->ocbuf);
                }
            } else {
                bfo_dirac_convolve(c, f->ring[rslot], f->ocbuf);
            }
            if (fading && prev >= 0) {
                for (i = 1; i < prevcblocks && i < f->procblocks; i++) {      /* :1758 */
                    const int j = (int)((t - (unsigned int)i) % N);
                    bfo_convolve_add(c, f->ring[j], e->co[prev].part[i], e->xfade[0]);
                }
            }
            if (fading) bfo_crossfade_inplace(c, f->ocbuf, e->xfade[0], e->xfade[1]);
        }
        f->prevcoeff = coeff;
    }

    /* :1847-1868 output mix, then :1877-2003 inverse FFT, (delay / mute / N:1 mix), quantise */
    {
        int filled = 0, seen = 0, idx;
        /* the process's output list is built physical channel by physical channel, members in
           phys2virt order (bfrun.c:2322-2323); :1981 counts the members as they come */
        int order[e->n_ch[1]];
        {
            int ph2, k = 0;
            for (ph2 = 0; ph2 < e->n_phys[1]; ph2++)
                for (n = 0; n < e->n_ch[1]; n++) if (e->v2p[1][n] == ph2) order[k++] = n;
        }
        for (idx = 0; idx < e->n_ch[1]; idx++) {
            void *src[e->n_filters > 0 ? e->n_filters : 1];
            double scales[e->n_filters > 0 ? e->n_filters : 1];
            const int ph = e->v2p[1][(n = order[idx])];
            const bfo_format *bf = &e->fmt[1][ph];
            int cnt = 0, j, r = 0;
            for (i = 0; i < e->n_filters; i++) {
                for (j = 0; j < e->f[i].n_out_ch; j++) {
                    if (e->f[i].out_ch[j] == n) {
                        src[cnt] = e->f[i].ocbuf;
                        scales[cnt] = e->f[i].scale_out[j] / bf->scale;       /* :1850 */
                        cnt++;
                        break;
                    }
                }
            }
            if (cnt == 0) {
                memset(e->out_freq[n], 0, (size_t)2 * e->L * e->rs);
            } else {
                bfo_mixnscale(c, src, e->out_freq[n], scales, cnt, BFO_MIX_OUTPUT);
            }
            bfo_freq2time(c, e->out_freq[n], e->tmp_out);
            apply_subdelay(e, 1, n, e->tmp_out);                              /* :1921-1925 */
            if (e->n_vpp[1][ph] == 1) {
                /* :1926-1936; the probe of sample 0 (:1903-1911) is subsumed by cbuf2raw's test */
                r = bfo_cbuf2raw(e->c, e->tmp_out, rawout, bf, e->dither_ch[ph], &e->overflow[n],
                                 e->safety_limit);
            } else {
                /* :1938-2003 */
                const int extra = (side_uses_subdelay(e, 1) && e->subdelay[1][n] == -100) ? e->sdf_length : 0;
                if (e->db[1][n] == NULL) {
                    int d1, m1;
                    vdelay_limits(e->delay[1][n], e->maxdelay[1][n], extra, &d1, &m1);
                    e->db[1][n] = bfo_delay_new(e->L, d1, m1, e->rs);
                }
                bfo_delay_update(e->db[1][n], e->tmp_out, e->delay[1][n] + extra);
                if (!e->muted[1][n]) {
                    if (!filled) {
                        memcpy(e->mixbuf, e->tmp_out, (size_t)e->L * e->rs);
                    } else if (e->rs == 4) {
                        for (i = 0; i < e->L; i++) ((float *)e->mixbuf)[i] += ((float *)e->tmp_out)[i];
                    } else {
                        for (i = 0; i < e->L; i++) ((double *)e->mixbuf)[i] += ((double *)e->tmp_out)[i];
                    }
                    filled = 1;
                }
                if (++seen == e->n_vpp[1][ph]) {
                    bfo_overflow of = e->overflow[n];
                    if (!filled) memset(e->mixbuf, 0, (size_t)e->L * e->rs);
                    r = bfo_cbuf2raw(e->c, e->mixbuf, rawout, bf, e->dither_ch[ph], &of, e->safety_limit);
                    for (i = 0; i < e->n_ch[1]; i++) {
                        if (e->v2p[1][i] == ph) e->overflow[i] = of;
                    }
                    seen = 0; filled = 0;
                }
            }
            if (r != 0 && status == 0) status = r;
        }
    }

    e->curbuf = !cur;
    e->blockcounter++;
    return status;
}

void
bfo_engine_get_overflow(const bfo_engine *e, int ch, bfo_overflow *of)
{
    *of = e->overflow[ch];
}

unsigned int
bfo_engine_blockcounter(const bfo_engine *e)
{
    return e->blockcounter;
}

const void *
bfo_engine_filter_output(const bfo_engine *e, int filter)
{
    return e->f[filter].ocbuf;
}

const void *
bfo_engine_output_spectrum(const bfo_engine *e, int ch)
{
    return e->out_freq[ch];
}
