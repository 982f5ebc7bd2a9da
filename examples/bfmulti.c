/*
 * bfmulti.c -- the multi-GPU crossbar from a plain C host with RCCL's own C API: what a patched
 * bfrun would do in place of its n_processes filter processes and their two pipe barriers per
 * block (bfrun.c:986-1006, 1563, 1873, 2312-2328).
 *
 * One process drives G GPUs (ncclCommInitAll).  GPU g owns input channels [g*I/G, (g+1)*I/G),
 * their rings and coefficient sets; per block it
 *     1. transforms its inputs                      bfhip_engine_inputs_dev
 *     2. computes partial spectra of ALL outputs    bfhip_engine_mac_dev
 *     3. reduce-scatters them (sum) over xGMI       ncclReduceScatter, O/G outputs stay local
 *     4. inverse-transforms and requantises those   bfhip_engine_outputs_dev
 * The same sequence bench.py --gpus N runs with torch.distributed standing in for the host.
 *
 *   bfmulti [n_gpus] [blocks] [n_in] [n_out] [L] [N]      (defaults: all GPUs, 50, 64, 64, 8192, 32)
 *
 * It prints the aggregate rate and, with one GPU, checks the result byte for byte against the
 * unsharded bfhip_engine_block_dev path (the reduce-scatter over one rank is a copy).
 *
 * Build: gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/bfmulti.c \
 *            -o examples/bfmulti -Lbrutefir_amd -lbfhip -L/opt/rocm/lib -lamdhip64 -lrccl -lm \
 *            -Wl,-rpath,$PWD/brutefir_amd -Wl,-rpath,/opt/rocm/lib
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "bfhip.h"

#define MAXG 16
#define HCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define NCK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); exit(1); } } while (0)
#define BCK(x) do { if ((x) < 0) { fprintf(stderr, "%s: %s\n", #x, bfhip_last_error()); exit(1); } } while (0)

static uint32_t rng_state = 2463534242u;
static double rnd(void)
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5;
    return (double)(int32_t)rng_state / 2147483648.0;
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static void set_formats(bfhip_engine *e, int io, int n_local, int first, int n_total)
{
    /* S24_4LE, interleaved over ALL channels of the side: a rank reads its own columns of the
       shared input block and writes its own columns of the shared output block */
    for (int c = 0; c < n_local; c++) {
        bfhip_format f;
        memset(&f, 0, sizeof(f));
        f.bytes = 4; f.sbytes = 3; f.scale = 1.0 / 8388608.0;
        f.sample_spacing = n_total; f.byte_offset = 4 * (first + c);
        BCK(bfhip_engine_set_format(e, io, c, &f));
    }
}

int main(int argc, char **argv)
{
    int n_dev = 0;
    HCK(hipGetDeviceCount(&n_dev));
    const int G = argc > 1 && atoi(argv[1]) > 0 ? atoi(argv[1]) : n_dev;
    const int blocks = argc > 2 ? atoi(argv[2]) : 50;
    const int I = argc > 3 ? atoi(argv[3]) : 64, O = argc > 4 ? atoi(argv[4]) : 64;
    const int L = argc > 5 ? atoi(argv[5]) : 8192, N = argc > 6 ? atoi(argv[6]) : 32;
    if (G < 1 || G > n_dev || G > MAXG || I % G || O % G) {
        fprintf(stderr, "bfmulti: %d GPUs requested, %d present; channels must divide evenly\n", G, n_dev);
        return 2;
    }
    const int Il = I / G, Ol = O / G, taps = L * N;
    int devs[MAXG];
    ncclComm_t comm[MAXG];
    hipStream_t stream[MAXG];
    bfhip_engine *eng[MAXG];
    void *d_in[MAXG], *d_out[MAXG], *z_part[MAXG], *z_loc[MAXG];
    for (int g = 0; g < G; g++) devs[g] = g;
    NCK(ncclCommInitAll(comm, G, devs));

    /* one shared input block (all ranks see the same interleaved frames, like dai's shm buffer) */
    const size_t in_bytes = (size_t)L * I * 4, out_bytes = (size_t)L * O * 4;
    int32_t *h_in = malloc(in_bytes);
    for (size_t k = 0; k < in_bytes / 4; k++) h_in[k] = (int32_t)(rnd() * 0.1 * 8388608.0);
    float *h = malloc((size_t)taps * sizeof(float));

    for (int g = 0; g < G; g++) {
        HCK(hipSetDevice(g));
        HCK(hipStreamCreateWithFlags(&stream[g], hipStreamNonBlocking));
        eng[g] = bfhip_engine_create(g, L, N, 4, Il, O);
        if (!eng[g]) { fprintf(stderr, "create: %s\n", bfhip_last_error()); return 1; }
        set_formats(eng[g], BFHIP_IN, Il, g * Il, I);
        set_formats(eng[g], BFHIP_OUT, O, 0, O);
        for (int o = 0; o < O; o++) {
            for (int i = 0; i < Il; i++) {
                rng_state = 77u + (uint32_t)(o * I + g * Il + i);         /* IR of (global in, out) */
                for (int k = 0; k < taps; k++) h[k] = (float)(rnd() * exp(-6.0 * k / taps) / (2.0 * I * sqrt((double)taps)));
                const int c = bfhip_engine_add_coeff(eng[g], h, taps, 1.0, 0);
                const double one = 1.0;
                BCK(c);
                BCK(bfhip_engine_add_filter(eng[g], 1, &i, &one, 0, NULL, NULL, 1, &o, &one, c, 0, 0));
            }
        }
        BCK(bfhip_engine_finalize(eng[g]));
        BCK(bfhip_engine_set_stream(eng[g], stream[g]));
        BCK(bfhip_engine_prewarm(eng[g]));
        HCK(hipMalloc(&d_in[g], in_bytes));
        HCK(hipMalloc(&d_out[g], out_bytes));
        HCK(hipMemset(d_out[g], 0, out_bytes));
        HCK(hipMalloc(&z_part[g], (size_t)O * L * 8));
        HCK(hipMalloc(&z_loc[g], (size_t)Ol * L * 8));
        HCK(hipMemcpy(d_in[g], h_in, in_bytes, hipMemcpyHostToDevice));
    }

    double t0 = 0;
    const int warm = 5;
    for (int b = 0; b < warm + blocks; b++) {
        if (b == warm) {
            for (int g = 0; g < G; g++) { HCK(hipSetDevice(g)); HCK(hipStreamSynchronize(stream[g])); }
            t0 = now_s();
        }
        for (int g = 0; g < G; g++) {
            HCK(hipSetDevice(g));
            BCK(bfhip_engine_inputs_dev(eng[g], d_in[g]));
            BCK(bfhip_engine_mac_dev(eng[g], z_part[g]));
            BCK(bfhip_engine_advance(eng[g]));
        }
        NCK(ncclGroupStart());
        for (int g = 0; g < G; g++)
            NCK(ncclReduceScatter(z_part[g], z_loc[g], (size_t)Ol * L * 2, ncclFloat, ncclSum, comm[g], stream[g]));
        NCK(ncclGroupEnd());
        for (int g = 0; g < G; g++) {
            HCK(hipSetDevice(g));
            /* rank g writes columns [g*Ol, (g+1)*Ol) of the interleaved output block */
            BCK(bfhip_engine_outputs_dev(eng[g], z_loc[g], g * Ol, Ol, d_out[g]));
        }
    }
    int status = 0;
    for (int g = 0; g < G; g++) {
        HCK(hipSetDevice(g));
        const int st = bfhip_engine_sync(eng[g]);
        BCK(st);
        status |= st;
    }
    const double el = now_s() - t0;
    printf("{\"host\": \"C + RCCL (ncclCommInitAll)\", \"n_gpus\": %d, \"workload\": \"%d-in/%d-out, %d taps (%d x %d)\", "
           "\"blocks\": %d, \"ms_per_block\": %.4f, \"samples_per_s\": %.1f, \"status_bits\": %d",
           G, I, O, taps, L, N, blocks, el * 1e3 / blocks, (double)O * L * blocks / el, status);

    if (G == 1) {
        /* reference run: the same engine layout through the fused block call */
        bfhip_engine *ref = bfhip_engine_create(0, L, N, 4, I, O);
        set_formats(ref, BFHIP_IN, I, 0, I);
        set_formats(ref, BFHIP_OUT, O, 0, O);
        for (int o = 0; o < O; o++)
            for (int i = 0; i < I; i++) {
                rng_state = 77u + (uint32_t)(o * I + i);
                for (int k = 0; k < taps; k++) h[k] = (float)(rnd() * exp(-6.0 * k / taps) / (2.0 * I * sqrt((double)taps)));
                const int c = bfhip_engine_add_coeff(ref, h, taps, 1.0, 0);
                const double one = 1.0;
                BCK(bfhip_engine_add_filter(ref, 1, &i, &one, 0, NULL, NULL, 1, &o, &one, c, 0, 0));
            }
        BCK(bfhip_engine_finalize(ref));
        BCK(bfhip_engine_prewarm(ref));
        void *d_ref;
        HCK(hipMalloc(&d_ref, out_bytes));
        for (int b = 0; b < warm + blocks; b++) BCK(bfhip_engine_block_dev(ref, d_in[0], d_ref));
        BCK(bfhip_engine_sync(ref));
        int32_t *a = malloc(out_bytes), *r = malloc(out_bytes);
        HCK(hipMemcpy(a, d_out[0], out_bytes, hipMemcpyDeviceToHost));
        HCK(hipMemcpy(r, d_ref, out_bytes, hipMemcpyDeviceToHost));
        long maxdiff = 0, maxabs = 0;
        for (size_t k = 0; k < out_bytes / 4; k++) {
            const long d = labs((long)a[k] - (long)r[k]);
            if (d > maxdiff) maxdiff = d;
            if (labs((long)r[k]) > maxabs) maxabs = labs((long)r[k]);
        }
        printf(", \"max_abs_difference_vs_block_dev\": %ld, \"max_abs_output\": %ld", maxdiff, maxabs);
        bfhip_engine_destroy(ref);
        free(a); free(r);
    }
    printf("}\n");
    for (int g = 0; g < G; g++) { bfhip_engine_destroy(eng[g]); ncclCommDestroy(comm[g]); }
    free(h); free(h_in);
    return 0;
}
