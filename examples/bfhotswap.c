/* bfhotswap.c -- the process topology of BruteFIR with an equaliser module, in 150 lines of C99
 * against libbfhip.so: who calls what, in which process, and why nothing but the filter process
 * ever touches the GPU.
 *
 *   parent  ("bfconf_init")  convolver_init(), coefficient memory in a shared mapping, every
 *                            partition prepared with convolver_coeffs2cbuf() -- all pure host
 *                            code in libbfhip.so, the parent never initialises HIP -- then fork()s
 *   child F ("filter process", bfrun.c:2312-2328)  builds the engine from the prepared cbufs
 *                            (bfhip_engine_add_coeff_processed_blocks, watch = 1), runs the block
 *                            loop with bfhip_engine_block()
 *   child M ("bflogic_eq", rendereq.h:87-91)  at some point renders new taps for one partition into
 *                            the shared coefficient memory with convolver_runtime_coeffs2cbuf() --
 *                            host code again; the call leaves a change notice, and F's engine
 *                            re-uploads that partition at the start of its next block
 *
 * usage: bfhotswap L N n_blocks switch_block taps0.f32 newpart.f32 in.f32 out.f32
 *   taps0: L*N float taps; newpart: L float taps that replace partition 1 from `switch_block` on;
 *   in: n_blocks*L float samples (one channel); out: n_blocks*L filtered float samples
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

static void die(const char *m) { fprintf(stderr, "bfhotswap: %s (%s)\n", m, bfhip_last_error()); exit(2); }

static float *slurp(const char *path, size_t n)
{
    FILE *f = fopen(path, "rb");
    float *p = malloc(n * sizeof(float));
    if (f == NULL || p == NULL || fread(p, sizeof(float), n, f) != n) die(path);
    fclose(f);
    return p;
}

int main(int argc, char **argv)
{
    int L, N, n_blocks, switch_block, b, to_m[2], to_f[2];
    float *taps0, *newpart, *in, *cmem;
    void **cbufs;
    pid_t pid_f, pid_m;
    char token = 'x';
    int st = 0;

    if (argc != 9) die("usage: bfhotswap L N n_blocks switch_block taps0 newpart in out");
    L = atoi(argv[1]); N = atoi(argv[2]); n_blocks = atoi(argv[3]); switch_block = atoi(argv[4]);
    taps0 = slurp(argv[5], (size_t)L * N);
    newpart = slurp(argv[6], (size_t)L);
    in = slurp(argv[7], (size_t)n_blocks * L);

    /* ---- parent: what bfconf_init() does (bfconf.c:2786, 1979-2019).  No HIP anywhere here. */
    if (!convolver_init(NULL, L, 4)) die("convolver_init");
    cmem = mmap(NULL, (size_t)N * convolver_cbufsize(), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (cmem == MAP_FAILED) die("mmap");
    cbufs = malloc(N * sizeof(void *));
    for (b = 0; b < N; b++) {
        cbufs[b] = (char *)cmem + (size_t)b * convolver_cbufsize();
        if (convolver_coeffs2cbuf(taps0 + (size_t)b * L, L, 1.0, cbufs[b]) == NULL) die("coeffs2cbuf");
    }
    if (pipe(to_m) != 0 || pipe(to_f) != 0) die("pipe");

    pid_f = fork();
    if (pid_f == 0) {
        /* ---- filter process: the only one that owns a GPU context */
        bfhip_engine *e = bfhip_engine_create(0, L, N, 4, 1, 1);
        int in_ch = 0, out_ch = 0, c;
        double one = 1.0;
        float *out = malloc((size_t)n_blocks * L * sizeof(float));
        FILE *f;
        if (e == NULL) die("engine_create");
        if (bfhip_engine_reserve_coeffs(e, (double)N * convolver_cbufsize()) < 0) die("reserve");
        if ((c = bfhip_engine_add_coeff_processed_blocks(e, cbufs, N, 1 /* shared memory: watch */)) < 0) die("add_coeff");
        if (bfhip_engine_add_filter(e, 1, &in_ch, &one, 0, NULL, NULL, 1, &out_ch, &one, c, 0, 0) < 0) die("add_filter");
        if (bfhip_engine_finalize(e) < 0) die("finalize");
        for (b = 0; b < n_blocks; b++) {
            if (b == switch_block) {
                /* let the module render, wait until it has (the real host needs no such handshake:
                   whenever the notice arrives, the NEXT block uses the new partition) */
                if (write(to_m[1], &token, 1) != 1 || read(to_f[0], &token, 1) != 1) die("handshake");
            }
            if (bfhip_engine_block(e, in + (size_t)b * L, out + (size_t)b * L, NULL) != 0) die("block");
        }
        if ((f = fopen(argv[8], "wb")) == NULL || fwrite(out, sizeof(float), (size_t)n_blocks * L, f) != (size_t)n_blocks * L) die("write");
        fclose(f);
        bfhip_engine_destroy(e);
        _exit(0);
    }
    pid_m = fork();
    if (pid_m == 0) {
        /* ---- module process (bflogic_eq): host code only, through what bfaccess->convolver_coeffs2cbuf
           points at */
        if (read(to_m[0], &token, 1) != 1) die("handshake");
        convolver_runtime_coeffs2cbuf(newpart, cbufs[1]);
        if (write(to_f[1], &token, 1) != 1) die("handshake");
        _exit(0);
    }
    if (waitpid(pid_m, &st, 0) != pid_m || !WIFEXITED(st) || WEXITSTATUS(st) != 0) die("module process failed");
    if (waitpid(pid_f, &st, 0) != pid_f || !WIFEXITED(st) || WEXITSTATUS(st) != 0) die("filter process failed");
    fprintf(stderr, "bfhotswap: %d blocks, partition 1 replaced from block %d on by another process\n", n_blocks, switch_block);
    return 0;
}
