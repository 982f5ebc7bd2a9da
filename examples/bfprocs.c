/* bfprocs.c -- BruteFIR's MULTI-PROCESS filter topology (bfrun.c:2312-2328) against libbfhip.so, in
 * plain C99: n forked filter processes, one engine each (on GPU process_index % device_count -- on
 * a one-GPU machine they share it), every process running the filters bfconf would have given it
 * (bfconf.c:2227-2318: a filter lives in the process that mixes its outputs) and writing ITS
 * outputs into the one raw output buffer all of them share.  It is what patches/bfrun-bfhip.diff
 * makes filter_process() do, without the rest of bfrun around it:
 *
 *   parent ("bfconf_init" + input/output process)   convolver_init(), every coefficient partition
 *        through convolver_coeffs2cbuf() -- host code, the parent never touches the GPU --, shared
 *        input / output / overflow memory, one pipe pair per filter process; fork()s; then per
 *        period: reads a block from the input file into the shared input buffer, wakes every filter
 *        process (bl_input_2_filter), waits for all of them (filter_2_bl_output), writes the
 *        shared output buffer to the output file
 *   filter process k                                 describes the WHOLE configuration to its engine,
 *        marks the other processes' filters inactive (bfhip_engine_set_filter_active), registers
 *        every coefficient set lazily (BFHIP_COEFF_LAZY: its GPU loads its own share), and per
 *        period calls bfhip_engine_rt_block(shared in, shared out, shared overflow): the engine
 *        writes only the samples and overflow entries of the outputs it owns
 *
 * The filter network: a full n_in x n_out crossbar (filter o*n_in + i: input i -> output o, delayed
 * by (o + i) % 2 blocks), plus one two-input mix into the last output.  Output o belongs to process
 * o % n_procs ("interleaved": the engine's groups of eight outputs are split between the
 * processes) or o * n_procs / n_out ("blocked").
 *
 * The point: the output file of `bfprocs 2 ...` (or 3, 4 ...) is BYTE-IDENTICAL to that of
 * `bfprocs 1 ...` -- tests/test_gpu_chost.py -- as the reference's own multi-process mode is bit-equal
 * to its single-process mode (SURVEY B.5 iii).
 *
 *   bfprocs n_procs split L N n_in n_out outfmt coeffs.f32 in.s24 out.raw [benchmark]
 *     split: interleaved | blocked;  outfmt: S16_LE | S24_4LE | FLOAT_LE
 *     coeffs.f32: n_out*n_in + 1 impulse responses of L*N float taps;  in.s24: S24_4LE frames
 *     benchmark: every process prints the reference's stage table (bfrun.c:2035-2078) every 10 periods
 *
 * Build:  gcc -std=c99 -O2 -Iinclude examples/bfprocs.c -o examples/bfprocs \
 *             -Lbrutefir_amd -lbfhip -Wl,-rpath,$PWD/brutefir_amd
 */
#define _DEFAULT_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include "bfhip.h"
#include "bfhip_convolver.h"

#define MAXPROCS 16

static void die(const char *m) { fprintf(stderr, "bfprocs[%d]: %s (%s)\n", (int)getpid(), m, bfhip_last_error()); exit(2); }

static void *shared(size_t bytes)
{
    void *p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) die("mmap");
    memset(p, 0, bytes);
    return p;
}

static int n_procs, blocked, L, N, n_in, n_out, n_filters, benchmark;
static int out_bytes, out_sbytes, out_isfloat;
static void ***cbufs;                     /* [n_filters][N]: "bfconf->coeffs_data" */
static unsigned char *inbuf, *outbuf;     /* the buffers the processes share ("buffers[IN/OUT]") */
static bfhip_overflow *overflow;          /* "icomm->overflow" */

static int owner_of_output(int o) { return blocked ? (int)((long)o * n_procs / n_out) : o % n_procs; }

/* the body of a forked filter process */
static void filter_process(int k, int wake_fd, int done_fd)
{
    bfhip_engine *e;
    int c, o, i, f, io, st, n_dev, flags;
    unsigned long cc = 0;
    char token;

    if ((n_dev = bfhip_device_count()) < 1) die("no device");
    if ((e = bfhip_engine_create(k % n_dev, L, N, 4, n_in, n_out)) == NULL) die("engine_create");
    for (io = 0; io < 2; io++) {
        const int n = io == BFHIP_IN ? n_in : n_out;
        for (c = 0; c < n; c++) {
            bfhip_format fm;
            memset(&fm, 0, sizeof(fm));
            if (io == BFHIP_IN) { fm.bytes = 4; fm.sbytes = 3; }
            else { fm.isfloat = out_isfloat; fm.bytes = out_bytes; fm.sbytes = out_sbytes; }
            fm.scale = fm.isfloat ? 1.0 : 1.0 / (double)(1ULL << (8 * fm.sbytes - 1));
            fm.sample_spacing = n; fm.byte_offset = c * fm.bytes;
            if (bfhip_engine_set_format(e, io, c, &fm) < 0) die("set_format");
        }
    }
    flags = n_procs > 1 ? BFHIP_COEFF_LAZY : 0;
    for (f = 0; f < n_filters; f++)
        if (bfhip_engine_add_coeff_processed_blocks(e, cbufs[f], N, flags) != f) die("add_coeff");
    /* every filter of the configuration, in ANY order (here: this process's own first): the engine
       orders its work by the filters' names */
    for (int pass = 0; pass < 2; pass++) {
        for (f = 0; f < n_filters; f++) {
            int in_ch[2], out_ch, idx, n_inputs = 1;
            double sc_in[2] = {1.0, -0.5}, sc_out = 1.0;
            if (f < n_in * n_out) { o = f / n_in; i = f % n_in; in_ch[0] = i; }
            else { o = n_out - 1; i = 1; in_ch[0] = 0; in_ch[1] = n_in - 1; n_inputs = n_in > 1 ? 2 : 1; }
            if ((owner_of_output(o) == k) != (pass == 0)) continue;
            out_ch = o;
            idx = bfhip_engine_add_filter(e, n_inputs, in_ch, sc_in, 0, NULL, NULL, 1, &out_ch, &sc_out, f, (o + i) % 2, 0);
            if (idx < 0 || bfhip_engine_set_filter_name(e, idx, f) < 0 ||
                bfhip_engine_set_filter_active(e, idx, owner_of_output(o) == k) < 0)
                die("add_filter");
        }
    }
    if (bfhip_engine_finalize(e) < 0) die("finalize");
    if (benchmark && bfhip_engine_enable_timing(e, 1) < 0) die("enable_timing");
    if (bfhip_engine_rt_begin(e, BFHIP_RT_SPIN | (benchmark ? BFHIP_RT_NO_GRAPH : 0)) < 0) die("rt_begin");

    while (read(wake_fd, &token, 1) == 1 && token == 'b') {
        st = bfhip_engine_rt_block(e, inbuf, outbuf, overflow);
        if (st < 0) die("rt_block");
        if (st != 0) { fprintf(stderr, "NaN or Inf values in the system, or safety limit exceeded. Aborting.\n"); exit(1); }
        if (benchmark && ++cc % 10 == 0) {
            double ms[8];
            if (bfhip_engine_stage_times(e, ms) > 0) {
                if (cc == 10 && k == 0)
                    fprintf(stderr, "  pid |  raw2real | time2freq | mixscale1 |  convolve | mixscale2 | freq2time |  real2raw |"
                            "     total | periods\n");
                fprintf(stderr, "%5d | %9.3f | %9.3f | %9.3f | %9.3f | %9.3f | %9.3f | %9.3f | %9.3f | %7lu\n", (int)getpid(),
                        ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], ms[6], ms[7], cc);
            }
        }
        if (write(done_fd, &token, 1) != 1) die("signal");
    }
    bfhip_engine_rt_end(e);
    bfhip_engine_destroy(e);
    _exit(0);
}

int main(int argc, char **argv)
{
    int k, f, b, o, wake[MAXPROCS][2], done[MAXPROCS][2], st = 0;
    pid_t pid[MAXPROCS];
    size_t in_block, out_block, got;
    unsigned long blocks = 0;
    float *taps;
    FILE *fc, *fi, *fo;
    char token;

    if (argc < 11) {
        fprintf(stderr, "usage: %s n_procs interleaved|blocked L N n_in n_out outfmt coeffs.f32 in.s24 out.raw [benchmark]\n", argv[0]);
        return 2;
    }
    n_procs = atoi(argv[1]); blocked = strcmp(argv[2], "blocked") == 0;
    L = atoi(argv[3]); N = atoi(argv[4]); n_in = atoi(argv[5]); n_out = atoi(argv[6]);
    benchmark = argc > 11;
    if (n_procs < 1 || n_procs > MAXPROCS || n_procs > n_out) die("bad process count");
    if (strcmp(argv[7], "S16_LE") == 0) { out_bytes = out_sbytes = 2; }
    else if (strcmp(argv[7], "S24_4LE") == 0) { out_bytes = 4; out_sbytes = 3; }
    else if (strcmp(argv[7], "FLOAT_LE") == 0) { out_bytes = out_sbytes = 4; out_isfloat = 1; }
    else die("unknown output format");
    n_filters = n_in * n_out + 1;

    /* ---- parent = bfconf_init(): coefficients prepared on the host, before any fork, no HIP */
    if (!convolver_init(NULL, L, 4)) die("convolver_init");
    if ((fc = fopen(argv[8], "rb")) == NULL) die(argv[8]);
    taps = malloc((size_t)L * N * sizeof(float));
    cbufs = malloc(n_filters * sizeof(void **));
    for (f = 0; f < n_filters; f++) {
        if (fread(taps, sizeof(float), (size_t)L * N, fc) != (size_t)L * N) die("coefficient file too short");
        cbufs[f] = malloc(N * sizeof(void *));
        for (b = 0; b < N; b++)
            if ((cbufs[f][b] = convolver_coeffs2cbuf(taps + (size_t)b * L, L, 1.0, NULL)) == NULL) die("coeffs2cbuf");
    }
    fclose(fc);
    free(taps);
    in_block = (size_t)L * n_in * 4;
    out_block = (size_t)L * n_out * out_bytes;
    inbuf = shared(in_block);
    outbuf = shared(out_block);
    overflow = shared(n_out * sizeof(bfhip_overflow));
    for (o = 0; o < n_out; o++)          /* bfrun.c:2263-2277: the parent initialises the overflow structs */
        overflow[o].max = out_isfloat ? 1.0 : (double)((1ULL << (8 * out_sbytes - 1)) - 1);

    /* ---- fork the filter processes (bfrun.c:2312-2328) */
    for (k = 0; k < n_procs; k++) {
        if (pipe(wake[k]) != 0 || pipe(done[k]) != 0) die("pipe");
        if ((pid[k] = fork()) < 0) die("fork");
        if (pid[k] == 0) {
            close(wake[k][1]); close(done[k][0]);
            filter_process(k, wake[k][0], done[k][1]);
        }
        close(wake[k][0]); close(done[k][1]);
    }

    /* ---- parent = input + output process */
    if ((fi = fopen(argv[9], "rb")) == NULL) die(argv[9]);
    if ((fo = fopen(argv[10], "wb")) == NULL) die(argv[10]);
    while ((got = fread(inbuf, 1, in_block, fi)) > 0) {
        if (got < in_block) memset(inbuf + got, 0, in_block - got);
        memset(outbuf, 0xEE, out_block);                 /* whatever is not written by its owner would show */
        token = 'b';
        for (k = 0; k < n_procs; k++) if (write(wake[k][1], &token, 1) != 1) die("wake");
        for (k = 0; k < n_procs; k++) if (read(done[k][0], &token, 1) != 1) die("a filter process died");
        fwrite(outbuf, (size_t)n_out * out_bytes, got / ((size_t)n_in * 4), fo);
        blocks++;
    }
    fclose(fi);
    fclose(fo);
    token = 'q';
    for (k = 0; k < n_procs; k++) if (write(wake[k][1], &token, 1) != 1) die("quit");
    for (k = 0; k < n_procs; k++)
        if (waitpid(pid[k], &st, 0) != pid[k] || !WIFEXITED(st) || WEXITSTATUS(st) != 0) die("a filter process failed");
    for (o = 0; o < n_out; o++) {
        const double peak = overflow[o].largest > (double)overflow[o].intlargest ? overflow[o].largest : (double)overflow[o].intlargest;
        printf("output %d: %u overflows, peak %.9g\n", o, overflow[o].n_overflows, peak);     /* (stdout: equal for every n_procs) */
    }
    fprintf(stderr, "bfprocs: %lu blocks through %d filter process%s\n", blocks, n_procs, n_procs == 1 ? "" : "es");
    return 0;
}
