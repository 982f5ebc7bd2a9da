// hbm_piece_probe -- does the SHAPE of a read stream matter at equal volume?  Development tool for
// the one-to-one filter plans (config D): there a MAC workgroup (bin tile t of output group g) reads,
// for each of its 8 entries and 16 partitions, one 4 KiB piece of a ring slot and one 4 KiB piece
// of a coefficient partition -- pieces 64 KiB apart -- where a crossbar workgroup reads one long
// sequential stream.  Same bytes, same kernel, three address maps:
//   seq     workgroup w reads [w * span, (w+1) * span) front to back
//   piece   workgroup (g, t) reads piece t of rows g*R .. g*R + R-1 (rows of `row` bytes, pieces of 4 KiB):
//           what the set-major / slot-major layouts give a one-to-one plan
//   tile    the same pieces after a tile-major re-layout: workgroup (g, t) reads R pieces back to back
//           (= seq with span = R * 4 KiB, listed for the launch geometry)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hbm_piece_probe.hip -o tools/hbm_piece_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const v4f *gp;

// two streams (a "ring" region and a "coefficient" region), U pieces of each in flight per wave
template <int U, bool NT>
__global__ __launch_bounds__(256) void piece_kernel(const v4f *__restrict__ a, const v4f *__restrict__ b, int rows_per_wg,
                                                     size_t row_vec, int tiles, int piece_major, float *__restrict__ sink) {
    const int g = blockIdx.x / tiles, t = blockIdx.x % tiles;
    v4f acc = {0, 0, 0, 0};
    for (int r0 = 0; r0 < rows_per_wg; r0 += U) {
        v4f qa[U], qb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t row = (size_t)g * rows_per_wg + r0 + u;
            // piece-major: the workgroup's pieces are contiguous; else piece t of row `row`
            const size_t off = piece_major ? ((size_t)blockIdx.x * rows_per_wg + r0 + u) * 256 + threadIdx.x
                                           : row * row_vec + (size_t)t * 256 + threadIdx.x;
            qa[u] = *(gp)(const void *)(a + off);
            if (NT) qb[u] = __builtin_nontemporal_load((gp)(const void *)(b + off));
            else qb[u] = *(gp)(const void *)(b + off);
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += qa[u] * qb[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int U, bool NT>
void run(const char *what, const v4f *a, const v4f *b, int groups, int tiles, int rows_per_wg, size_t row_bytes, int piece_major, float *sink) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int wgs = groups * tiles;
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((piece_kernel<U, NT>), dim3(wgs), dim3(256), 0, 0, a, b, rows_per_wg, row_bytes / 16, tiles, piece_major, sink);
    CK(hipDeviceSynchronize());
    const int iters = 20;
    CK(hipEventRecord(e0, 0));
    for (int w = 0; w < iters; w++) hipLaunchKernelGGL((piece_kernel<U, NT>), dim3(wgs), dim3(256), 0, 0, a, b, rows_per_wg, row_bytes / 16, tiles, piece_major, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 2.0 * wgs * rows_per_wg * 4096.0;
    printf("%-6s wgs %5d  pieces in flight per wave 2x%-2d %s: %.4f ms  %.0f GB/s  (%.0f MiB)\n", what, wgs, U, NT ? "nt" : "  ", ms / iters,
           bytes / (ms / iters * 1e-3) / 1e9, bytes / 1048576.0);
}

int main() {
    // config D: 256 one-to-one filters x 16 partitions x 64 KiB: 256 MiB of ring + 256 MiB of coefficients;
    // 32 groups of 8 filters x 16 bin tiles; a workgroup reads 8 * 16 = 128 rows' pieces of each region
    const size_t region = (size_t)256 << 20;
    v4f *a, *b; float *sink;
    CK(hipMalloc(&a, region)); CK(hipMalloc(&b, region)); CK(hipMemset(a, 1, region)); CK(hipMemset(b, 1, region)); CK(hipMalloc(&sink, 4));
    for (int rep = 0; rep < 2; rep++) {
        run<4, false>("piece", a, b, 32, 16, 128, 65536, 0, sink);
        run<4, false>("tile", a, b, 32, 16, 128, 65536, 1, sink);
        run<8, false>("piece", a, b, 32, 16, 128, 65536, 0, sink);
        run<8, false>("tile", a, b, 32, 16, 128, 65536, 1, sink);
        run<8, true>("piece", a, b, 32, 16, 128, 65536, 0, sink);
        run<8, true>("tile", a, b, 32, 16, 128, 65536, 1, sink);
        run<16, true>("piece", a, b, 32, 16, 128, 65536, 0, sink);
        run<16, true>("tile", a, b, 32, 16, 128, 65536, 1, sink);
        // twice the workgroups, half the rows each (S = 2 chunks)
        run<8, true>("piece", a, b, 64, 16, 64, 65536, 0, sink);
        run<8, true>("tile", a, b, 64, 16, 64, 65536, 1, sink);
    }
    return 0;
}
