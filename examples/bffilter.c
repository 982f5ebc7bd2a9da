/*
 * bffilter.c -- a minimal C host for libbfhip.so: the part of BruteFIR's run loop that matters
 * for the filter path, in plain C99, linked against the C ABI only (no HIP, no C++, no Python).
 *
 * It does what `brutefir` does with bfio_file devices on both sides (bfio_file.c) and a
 * crossbar of filters: read interleaved raw PCM blocks from a file, push every block through
 * bfhip_engine_block() -- the call that replaces bfrun.c:1493-2008 in a patched
 * filter_process() -- and write the interleaved raw output, truncating the
 * last block like dai.c does at EOF (dai.c:1423-1439).
 *
 *   bffilter L N n_in n_out infmt outfmt coeffs.f32 in.raw out.raw [dither_rate [benchmark]]
 *
 *   benchmark: print the reference's `benchmark: true` stage table (bfrun.c:2035-2078) every ten
 *   periods, its columns filled from bfhip_engine_stage_times() (device milliseconds)
 *
 *   coeffs.f32: n_out * n_in impulse responses of L*N float32 taps, output-major
 *   infmt/outfmt: S16_LE S24_LE S24_4LE S32_LE FLOAT_LE FLOAT64_LE
 *
 * Build:  gcc -std=c99 -O2 -Iinclude examples/bffilter.c -o examples/bffilter \
 *             -Lbrutefir_amd -lbfhip -Wl,-rpath,$PWD/brutefir_amd
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bfhip.h"

struct fmtdesc { const char *name; int bytes, sbytes, isfloat; };
static const struct fmtdesc formats[] = {
    {"S16_LE", 2, 2, 0}, {"S24_LE", 3, 3, 0}, {"S24_4LE", 4, 3, 0}, {"S32_LE", 4, 4, 0},
    {"FLOAT_LE", 4, 4, 1}, {"FLOAT64_LE", 8, 8, 1},
};

static const struct fmtdesc *find_format(const char *name)
{
    size_t i;
    for (i = 0; i < sizeof(formats) / sizeof(formats[0]); i++) {
        if (strcmp(formats[i].name, name) == 0) return &formats[i];
    }
    fprintf(stderr, "Unknown sample format.\n");
    exit(2);
}

static void die(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, bfhip_last_error());
    exit(1);                                    /* BF_EXIT_OTHER */
}

int main(int argc, char *argv[])
{
    int L, N, n_in, n_out, io, c, o, i, st, dither_rate = 0, benchmark = 0;
    const struct fmtdesc *fd[2];
    bfhip_engine *e;
    bfhip_overflow *of;
    FILE *fc, *fi, *fo;
    float *taps;
    unsigned char *inbuf, *outbuf;
    size_t inbytes, outbytes, got, frames;
    unsigned long blocks = 0;

    if (argc < 10) {
        fprintf(stderr, "usage: %s L N n_in n_out infmt outfmt coeffs.f32 in.raw out.raw [dither_rate]\n", argv[0]);
        return 2;
    }
    L = atoi(argv[1]); N = atoi(argv[2]); n_in = atoi(argv[3]); n_out = atoi(argv[4]);
    fd[BFHIP_IN] = find_format(argv[5]);
    fd[BFHIP_OUT] = find_format(argv[6]);
    if (argc > 10) dither_rate = atoi(argv[10]);
    if (argc > 11) benchmark = 1;

    /* bfconf_init() + filter_process() set-up */
    e = bfhip_engine_create(0, L, N, 4, n_in, n_out);
    if (e == NULL) die("bfhip_engine_create");
    for (io = 0; io < 2; io++) {
        const int n = io == BFHIP_IN ? n_in : n_out;
        for (c = 0; c < n; c++) {
            /* an interleaved device with all channels open: dai.c:537-576 */
            bfhip_format f;
            f.isfloat = fd[io]->isfloat; f.swap = 0;
            f.bytes = fd[io]->bytes; f.sbytes = fd[io]->sbytes;
            f.scale = f.isfloat ? 1.0 : 1.0 / (double)(1ULL << (8 * f.sbytes - 1));
            f.sample_spacing = n; f.byte_offset = c * f.bytes;
            if (bfhip_engine_set_format(e, io, c, &f) < 0) die("bfhip_engine_set_format");
        }
    }
    if (dither_rate > 0) {
        int *chs = malloc(n_out * sizeof(int));
        for (c = 0; c < n_out; c++) chs[c] = c;
        if (bfhip_engine_enable_dither(e, chs, n_out, dither_rate, 0) < 0) die("bfhip_engine_enable_dither");
        free(chs);
    }
    if ((fc = fopen(argv[7], "rb")) == NULL) { perror(argv[7]); return 1; }
    taps = malloc((size_t)L * N * sizeof(float));
    for (o = 0; o < n_out; o++) {
        for (i = 0; i < n_in; i++) {
            double one = 1.0;
            int coeff;
            if (fread(taps, sizeof(float), (size_t)L * N, fc) != (size_t)L * N) {
                fprintf(stderr, "Length mismatch of file \"%s\".\n", argv[7]);
                return 2;
            }
            if ((coeff = bfhip_engine_add_coeff(e, taps, L * N, 1.0, N)) < 0) die("bfhip_engine_add_coeff");
            if (bfhip_engine_add_filter(e, 1, &i, &one, 0, NULL, NULL, 1, &o, &one, coeff, 0, 0) < 0)
                die("bfhip_engine_add_filter");
        }
    }
    fclose(fc);
    free(taps);
    if (bfhip_engine_finalize(e) < 0) die("bfhip_engine_finalize");
    if (benchmark && bfhip_engine_enable_timing(e, 1) < 0) die("bfhip_engine_enable_timing");

    /* the run loop: input_process / filter_process / output_process collapsed into one */
    inbytes = (size_t)L * n_in * fd[BFHIP_IN]->bytes;
    outbytes = (size_t)L * n_out * fd[BFHIP_OUT]->bytes;
    inbuf = malloc(inbytes);
    outbuf = malloc(outbytes);
    of = calloc(n_out, sizeof(bfhip_overflow));
    for (c = 0; c < n_out; c++) bfhip_engine_get_overflow(e, c, &of[c]);
    if ((fi = fopen(argv[8], "rb")) == NULL) { perror(argv[8]); return 1; }
    if ((fo = fopen(argv[9], "wb")) == NULL) { perror(argv[9]); return 1; }
    while ((got = fread(inbuf, 1, inbytes, fi)) > 0) {
        if (got < inbytes) memset(inbuf + got, 0, inbytes - got);
        st = bfhip_engine_block(e, inbuf, outbuf, of);
        if (st < 0) die("bfhip_engine_block");
        if (st & BFHIP_ST_NONFINITE) { fprintf(stderr, "NaN or Inf values in the output! Bad output. Aborting.\n"); return 1; }
        if (st & BFHIP_ST_SAFETY) { fprintf(stderr, "Safety limit exceeded on output. Aborting.\n"); return 1; }
        frames = got / ((size_t)n_in * fd[BFHIP_IN]->bytes);
        fwrite(outbuf, (size_t)n_out * fd[BFHIP_OUT]->bytes, frames, fo);
        blocks++;
        if (benchmark && blocks % 10 == 0) {
            double ms[8];
            if (bfhip_engine_stage_times(e, ms) > 0) {
                if (blocks == 10)
                    fprintf(stderr, "  raw2real | time2freq | mixscale1 |  convolve | mixscale2 | freq2time |  real2raw |     total | periods\n");
                fprintf(stderr, " %9.3f | %9.3f | %9.3f | %9.3f | %9.3f | %9.3f | %9.3f | %9.3f | %7lu\n",
                        ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], ms[6], ms[7], blocks);
            }
        }
    }
    fclose(fi);
    fclose(fo);
    for (c = 0; c < n_out; c++) {
        double peak = of[c].largest > (double)of[c].intlargest ? of[c].largest : (double)of[c].intlargest;
        fprintf(stderr, "output %d: %u overflows, peak %.6g of %.6g\n", c, of[c].n_overflows, peak, of[c].max);
    }
    fprintf(stderr, "%lu blocks, blockcounter %u\n", blocks, bfhip_engine_blockcounter(e));
    bfhip_engine_destroy(e);
    return 0;
}
