/* bfunfused.c -- the UNFUSED block loop: a filter process that keeps the reference's own per-block
 * structure and calls the 22 convolver.h symbols one buffer at a time (include/bfhip_convolver.h),
 * the way an unpatched bfrun.c does when it is merely linked against libbfhip.so.
 *
 * This is the path a configuration keeps when a logic module registers one of the per-buffer
 * events of struct bfevents (bfmod.h:192-215: input_timed, input_freqd, pre_convolve,
 * post_convolve, output_freqd, output_timed): those hand the module a HOST buffer in the middle of
 * the block (call sites bfrun.c:1533-1535, 1554-1557, 1688-1690, 1839-1841, 1882-1884,
 * 1918-1920), which the fused device path (bfhip_engine_block, one call per block) has no room
 * for.  Every op stages its operands over PCIe, so this loop is slow -- it exists for
 * completeness, and tests/test_gpu_unfused.py checks it against the fused engine and the oracle.
 *
 * Topology: I inputs x O outputs, one filter per (output, input) pair, filter f = o * I + i,
 * N partitions of L taps, S24_4LE interleaved frames in and out.  Buffer roles follow
 * filter_process(): per input a double-buffered 2L window and a spectrum; per filter a ring of N
 * mixed-and-scaled input spectra and one output spectrum; per output a mixed spectrum.
 *
 * usage: bfunfused <realsize> <L> <N> <I> <O> <coeffs.bin> <in.raw> <out.raw> [event:index:factor ...]
 *   coeffs.bin : O*I impulse responses of L*N reals, `realsize` bytes each
 *   event      : one of the six names above; the hook multiplies the 2L reals it is handed by
 *                `factor` when it is called for channel / filter `index`
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bfhip_convolver.h"

enum { EV_INPUT_TIMED, EV_INPUT_FREQD, EV_PRE_CONVOLVE, EV_POST_CONVOLVE, EV_OUTPUT_FREQD, EV_OUTPUT_TIMED, N_EVENTS };
static const char *const event_name[N_EVENTS] = {"input_timed", "input_freqd", "pre_convolve", "post_convolve",
                                                 "output_freqd", "output_timed"};

/* what a logic module would register: void (*)(void *buf, int channel_or_filter), bfmod.h:201-214 */
typedef void (*buffer_event)(void *buf, int index);
#define MAX_HOOKS 16
static struct { buffer_event fn[MAX_HOOKS]; int n; } events[N_EVENTS];

/* the test module: one table of (event, index, factor) */
static struct { int event, index; double factor; } rules[MAX_HOOKS];
static int n_rules, g_realsize, g_len;
static unsigned long hook_calls[N_EVENTS];

static void scale_buffer(int event, void *buf, int index)
{
    int r, k;
    hook_calls[event]++;
    for (r = 0; r < n_rules; r++) {
        if (rules[r].event != event || rules[r].index != index) continue;
        if (g_realsize == 4) for (k = 0; k < 2 * g_len; k++) ((float *)buf)[k] *= (float)rules[r].factor;
        else for (k = 0; k < 2 * g_len; k++) ((double *)buf)[k] *= rules[r].factor;
    }
}
static void on_input_timed(void *b, int i) { scale_buffer(EV_INPUT_TIMED, b, i); }
static void on_input_freqd(void *b, int i) { scale_buffer(EV_INPUT_FREQD, b, i); }
static void on_pre_convolve(void *b, int i) { scale_buffer(EV_PRE_CONVOLVE, b, i); }
static void on_post_convolve(void *b, int i) { scale_buffer(EV_POST_CONVOLVE, b, i); }
static void on_output_freqd(void *b, int i) { scale_buffer(EV_OUTPUT_FREQD, b, i); }
static void on_output_timed(void *b, int i) { scale_buffer(EV_OUTPUT_TIMED, b, i); }
static const buffer_event module_hooks[N_EVENTS] = {on_input_timed, on_input_freqd, on_pre_convolve,
                                                    on_post_convolve, on_output_freqd, on_output_timed};

static void fire(int event, void *buf, int index)
{
    int k;
    for (k = 0; k < events[event].n; k++) events[event].fn[k](buf, index);
}

static void die(const char *m) { fprintf(stderr, "bfunfused: %s\n", m); exit(2); }
static void *zalloc(size_t n) { void *p = calloc(1, n); if (!p) die("out of memory"); return p; }

int main(int argc, char **argv)
{
    int rs, L, N, I, O, F, a, c, f, i, o, p, curbuf = 0;
    unsigned int blockcounter = 0;
    size_t cbufsize, frame_in, frame_out, got;
    void ***in_time, **in_freq, ***ring, **ocbuf, **out_freq, *tbuf, ***coeffs, **mix_src;
    double *mix_scale;
    int *procblocks;
    uint8_t *rawin, *rawout, *taps;
    struct bfhip_buffer_format *bf_in, *bf_out;
    struct bfhip_overflow *overflow;
    FILE *fin, *fout, *fc;
    unsigned long n_blocks = 0;

    if (argc < 9) die("usage: bfunfused realsize L N I O coeffs.bin in.raw out.raw [event:index:factor ...]");
    rs = atoi(argv[1]); L = atoi(argv[2]); N = atoi(argv[3]); I = atoi(argv[4]); O = atoi(argv[5]);
    if ((rs != 4 && rs != 8) || L < 4 || N < 1 || I < 1 || O < 1) die("bad shape");
    F = I * O;
    g_realsize = rs; g_len = L;
    for (a = 9; a < argc; a++) {
        char name[32];
        int e;
        if (n_rules == MAX_HOOKS) die("too many hooks");
        if (sscanf(argv[a], "%31[^:]:%d:%lf", name, &rules[n_rules].index, &rules[n_rules].factor) != 3) die("bad hook");
        for (e = 0; e < N_EVENTS; e++) if (strcmp(name, event_name[e]) == 0) break;
        if (e == N_EVENTS) die("unknown event");
        rules[n_rules++].event = e;
        /* a module registers a function per event once, however many channels it cares about */
        if (events[e].n == 0) events[e].fn[events[e].n++] = module_hooks[e];
    }

    if (!convolver_init(NULL, L, rs)) die("convolver_init failed");
    cbufsize = (size_t)convolver_cbufsize();

    /* coefficients: one convolver_coeffs2cbuf() per partition, like load_coeff (bfconf.c:1979-2019) */
    taps = zalloc((size_t)L * N * rs);
    coeffs = zalloc(F * sizeof(*coeffs));
    if ((fc = fopen(argv[6], "rb")) == NULL) die("cannot open coefficients");
    for (f = 0; f < F; f++) {
        if (fread(taps, rs, (size_t)L * N, fc) != (size_t)L * N) die("short coefficient file");
        coeffs[f] = zalloc(N * sizeof(void *));
        for (p = 0; p < N; p++)
            if ((coeffs[f][p] = convolver_coeffs2cbuf(taps + (size_t)p * L * rs, L, 1.0, NULL)) == NULL) die("coeffs2cbuf failed");
    }
    fclose(fc);

    in_time = zalloc(I * sizeof(*in_time));
    in_freq = zalloc(I * sizeof(*in_freq));
    bf_in = zalloc(I * sizeof(*bf_in));
    for (c = 0; c < I; c++) {
        in_time[c] = zalloc(2 * sizeof(void *));
        in_time[c][0] = zalloc(cbufsize); in_time[c][1] = zalloc(cbufsize);
        in_freq[c] = zalloc(cbufsize);
        bf_in[c].sf.isfloat = 0; bf_in[c].sf.swap = 0; bf_in[c].sf.bytes = 4; bf_in[c].sf.sbytes = 3;
        bf_in[c].sf.scale = 1.0 / 8388608.0;
        bf_in[c].sample_spacing = I; bf_in[c].byte_offset = 4 * c;
    }
    ring = zalloc(F * sizeof(*ring));
    ocbuf = zalloc(F * sizeof(*ocbuf));
    procblocks = zalloc(F * sizeof(int));
    for (f = 0; f < F; f++) {
        ring[f] = zalloc(N * sizeof(void *));
        for (p = 0; p < N; p++) ring[f][p] = zalloc(cbufsize);
        ocbuf[f] = zalloc(cbufsize);
    }
    out_freq = zalloc(O * sizeof(*out_freq));
    bf_out = zalloc(O * sizeof(*bf_out));
    overflow = zalloc(O * sizeof(*overflow));
    for (c = 0; c < O; c++) {
        out_freq[c] = zalloc(cbufsize);
        bf_out[c] = bf_in[0];
        bf_out[c].sample_spacing = O; bf_out[c].byte_offset = 4 * c;
        overflow[c].max = 8388607.0;
    }
    tbuf = zalloc(cbufsize);
    mix_src = zalloc(I * sizeof(void *));
    mix_scale = zalloc(I * sizeof(double));
    frame_in = (size_t)4 * I; frame_out = (size_t)4 * O;
    rawin = zalloc(frame_in * L);
    rawout = zalloc(frame_out * L);

    if ((fin = fopen(argv[7], "rb")) == NULL) die("cannot open input");
    if ((fout = fopen(argv[8], "wb")) == NULL) die("cannot open output");
    while ((got = fread(rawin, frame_in, L, fin)) > 0) {
        const int curblock = (int)(blockcounter % (unsigned int)N);
        if (got < (size_t)L) memset(rawin + got * frame_in, 0, (L - got) * frame_in);

        /* inputs: raw -> sliding 2L window -> spectrum */
        for (c = 0; c < I; c++) {
            convolver_raw2cbuf(rawin, in_time[c][curbuf], in_time[c][!curbuf], &bf_in[c], NULL, NULL);
            fire(EV_INPUT_TIMED, in_time[c][curbuf], c);
            convolver_time2freq(in_time[c][curbuf], in_freq[c]);
            fire(EV_INPUT_FREQD, in_freq[c], c);
        }
        /* filters: scaled input into the ring, then the partitions that exist so far */
        for (f = 0; f < F; f++) {
            void *src[1];
            double scale[1];
            i = f % I;
            src[0] = in_freq[i];
            scale[0] = 1.0 * bf_in[i].sf.scale;
            if (procblocks[f] < N) procblocks[f]++;
            convolver_mixnscale(src, ring[f][curblock], scale, 1, CONVOLVER_MIXMODE_INPUT);
            fire(EV_PRE_CONVOLVE, ring[f][curblock], f);
            convolver_convolve(ring[f][curblock], coeffs[f][0], ocbuf[f]);
            for (p = 1; p < N && p < procblocks[f]; p++)
                convolver_convolve_add(ring[f][(int)((blockcounter - (unsigned int)p) % (unsigned int)N)], coeffs[f][p], ocbuf[f]);
            fire(EV_POST_CONVOLVE, ring[f][curblock], f);
        }
        /* outputs: mix, back to time, requantise */
        for (o = 0; o < O; o++) {
            for (i = 0; i < I; i++) { mix_src[i] = ocbuf[o * I + i]; mix_scale[i] = 1.0 / bf_out[o].sf.scale; }
            convolver_mixnscale(mix_src, out_freq[o], mix_scale, I, CONVOLVER_MIXMODE_OUTPUT);
            fire(EV_OUTPUT_FREQD, out_freq[o], o);
            convolver_freq2time(out_freq[o], tbuf);
            fire(EV_OUTPUT_TIMED, tbuf, o);
            convolver_cbuf2raw(tbuf, rawout, &bf_out[o], 0, NULL, &overflow[o]);
        }
        if (fwrite(rawout, frame_out, got, fout) != got) die("short write");
        blockcounter++;
        curbuf = !curbuf;
        n_blocks++;
    }
    fclose(fin);
    fclose(fout);
    fprintf(stderr, "bfunfused: %lu blocks;", n_blocks);
    for (a = 0; a < N_EVENTS; a++) fprintf(stderr, " %s %lu", event_name[a], hook_calls[a]);
    fprintf(stderr, "; overflows");
    for (c = 0; c < O; c++) fprintf(stderr, " %u", overflow[c].n_overflows);
    fprintf(stderr, "\n");
    return bfhip_convolver_last_fatal() ? 1 : 0;
}
