/*
 * bflatency -- round-trip latency of ONE block through the C ABI from a plain C host, the
 * number that matters for callback I/O (bfio_jack.c:132-200 -> bf_callback_ready,
 * bfrun.c:2086-2131: the sound server hands over a period and blocks until it is filtered).
 *
 *   gcc -O2 -Iinclude examples/bflatency.c -o examples/bflatency -Lbrutefir_amd -lbfhip \
 *       -Wl,-rpath,'$ORIGIN/../brutefir_amd' -lm
 *   examples/bflatency [blocks]
 *
 * For every shape it times bfhip_engine_block (pageable buffers, stream launches) and
 * bfhip_engine_rt_block (pinned double buffer, HIP-graph replay; with and without
 * BFHIP_RT_SPIN, in-place buffers) and prints one JSON line per shape with median / p99 / max
 * microseconds and the period the block lasts at 48 kHz.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bfhip.h"

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : x > y;
}

static uint32_t rng_state = 12345u;
static double rnd(void)          /* uniform (-1, 1), xorshift */
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 17;
    rng_state ^= rng_state << 5;
    return (double)(int32_t)rng_state / 2147483648.0;
}

struct shape { int n_in, n_out, length, n_blocks, realsize; };

static void die(const char *what)
{
    fprintf(stderr, "bflatency: %s: %s\n", what, bfhip_last_error());
    exit(1);
}

static void stats(double *t, int n, double *med, double *p99, double *mx)
{
    qsort(t, n, sizeof(double), cmp_double);
    *med = t[n / 2];
    *p99 = t[(int)(n * 0.99)];
    *mx = t[n - 1];
}

static bfhip_engine *make_engine(const struct shape *s)
{
    bfhip_engine *e = bfhip_engine_create(0, s->length, s->n_blocks, s->realsize, s->n_in, s->n_out);
    if (e == NULL) die("create");
    for (int io = 0; io < 2; io++) {
        int n = io == 0 ? s->n_in : s->n_out;
        for (int c = 0; c < n; c++) {
            bfhip_format f;
            memset(&f, 0, sizeof(f));
            f.isfloat = 0; f.swap = 0; f.bytes = 4; f.sbytes = 4;       /* S32_LE, interleaved */
            f.scale = 1.0 / 2147483648.0;
            f.sample_spacing = n; f.byte_offset = 4 * c;
            if (bfhip_engine_set_format(e, io, c, &f) < 0) die("set_format");
        }
    }
    const int taps = s->length * s->n_blocks;
    void *h = malloc((size_t)taps * s->realsize);
    for (int o = 0; o < s->n_out; o++) {
        for (int i = 0; i < s->n_in; i++) {
            for (int k = 0; k < taps; k++) {
                double v = rnd() * exp(-6.0 * k / taps) / (4.0 * s->n_in * sqrt((double)taps));
                if (s->realsize == 4) ((float *)h)[k] = (float)v; else ((double *)h)[k] = v;
            }
            int c = bfhip_engine_add_coeff(e, h, taps, 1.0, 0);
            if (c < 0) die("add_coeff");
            double one = 1.0;
            if (bfhip_engine_add_filter(e, 1, &i, &one, 0, NULL, NULL, 1, &o, &one, c, 0, 0) < 0) die("add_filter");
        }
    }
    free(h);
    if (bfhip_engine_finalize(e) < 0) die("finalize");
    return e;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 2000, warm = 200;
    const struct shape shapes[] = {
        {2, 2, 64, 64, 4}, {2, 2, 256, 64, 4}, {8, 8, 256, 64, 4}, {8, 8, 1024, 64, 4},
        {8, 8, 1024, 64, 8}, {64, 64, 1024, 16, 4}, {8, 8, 8192, 8, 4},
    };
    double *t = malloc(sizeof(double) * n);
    for (size_t si = 0; si < sizeof(shapes) / sizeof(shapes[0]); si++) {
        const struct shape *s = &shapes[si];
        const size_t inb = (size_t)s->n_in * s->length * 4, outb = (size_t)s->n_out * s->length * 4;
        int32_t *in = malloc(inb), *out = malloc(outb), *ref = malloc(outb * (size_t)(n + warm > 64 ? 64 : n + warm));
        int32_t *in0 = malloc(inb);
        for (size_t k = 0; k < inb / 4; k++) in0[k] = (int32_t)(rnd() * 0.25 * 2147483647.0);
        double med[4], p99[4], mx[4];
        int identical = 1;
        for (int mode = 0; mode < 4; mode++) {
            /* 0: bfhip_engine_block, 1: rt_block with copy-engine nodes, 2: rt_block (copy kernels) + SPIN,
               3: SPIN + in-place pinned buffers */
            rng_state = 4242u + (uint32_t)si;          /* same taps and periods in every mode */
            bfhip_engine *e = make_engine(s);
            memcpy(in, in0, inb);
            if (mode >= 1 && bfhip_engine_rt_begin(e, mode >= 2 ? BFHIP_RT_SPIN : BFHIP_RT_COPY_ENGINE) < 0) die("rt_begin");
            for (int k = 0; k < n + warm; k++) {
                in[(k * 7) % (inb / 4)] = (int32_t)(rnd() * 0.25 * 2147483647.0);     /* inputs differ per block */
                double t0 = now_us();
                int st;
                if (mode == 0) {
                    st = bfhip_engine_block(e, in, out, NULL);
                } else if (mode < 3) {
                    st = bfhip_engine_rt_block(e, in, out, NULL);
                } else {
                    void *pin = bfhip_engine_rt_buffer(e, 0, k & 1), *pout = bfhip_engine_rt_buffer(e, 1, k & 1);
                    memcpy(pin, in, inb);            /* stands for the sound server writing the period */
                    st = bfhip_engine_rt_submit(e, NULL);
                    if (st >= 0) st = bfhip_engine_rt_wait(e, NULL, NULL);
                    if (k < 64) memcpy(out, pout, outb);
                }
                double t1 = now_us();
                if (st < 0) die("block");
                if (k >= warm) t[k - warm] = t1 - t0;
                if (k < 64) {
                    if (mode == 0) memcpy((char *)ref + (size_t)k * outb, out, outb);
                    else if (memcmp((char *)ref + (size_t)k * outb, out, outb) != 0) identical = 0;
                }
            }
            stats(t, n, &med[mode], &p99[mode], &mx[mode]);
            bfhip_engine_destroy(e);
        }
        printf("{\"shape\": \"%dx%d L=%d N=%d f%d\", \"period_us_48k\": %.1f, "
               "\"block\": [%.1f, %.1f, %.1f], \"rt_copy_engine\": [%.1f, %.1f, %.1f], \"rt_spin\": [%.1f, %.1f, %.1f], "
               "\"rt_spin_inplace\": [%.1f, %.1f, %.1f], \"columns\": \"median, p99, max us\", \"outputs_identical\": %s}\n",
               s->n_in, s->n_out, s->length, s->n_blocks, s->realsize * 8, s->length / 48000.0 * 1e6,
               med[0], p99[0], mx[0], med[1], p99[1], mx[1], med[2], p99[2], mx[2], med[3], p99[3], mx[3],
               identical ? "true" : "false");
        fflush(stdout);
        free(in); free(in0); free(out); free(ref);
    }
    free(t);
    return 0;
}
