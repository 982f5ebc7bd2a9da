// hbm_policy_probe -- does any cache-policy combination of global_load beat the plain non-temporal
// hint for a pure 8 GiB read stream?  (gfx950 load modifiers: sc0, sc1, nt.)  Development tool.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hbm_policy_probe.hip -o tools/hbm_policy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

#define LOADX4(MODS)                                                                         \
    asm volatile("global_load_dwordx4 %0, %1, off " MODS : "=v"(q[u]) : "v"(p + i + (size_t)u * 256) : "memory")

template <int POLICY, int U>
__global__ __launch_bounds__(256) void read_kernel(const v4f *__restrict__ src, size_t n_vec_per_wg, float *__restrict__ sink) {
    const v4f *p = src + (size_t)blockIdx.x * n_vec_per_wg + threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    for (size_t i = 0; i + (size_t)U * 256 <= n_vec_per_wg; i += (size_t)U * 256) {
        v4f q[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (POLICY == 0) LOADX4("");
            else if (POLICY == 1) LOADX4("nt");
            else if (POLICY == 2) LOADX4("sc0");
            else if (POLICY == 3) LOADX4("sc1");
            else if (POLICY == 4) LOADX4("sc0 sc1");
            else if (POLICY == 5) LOADX4("sc0 nt");
            else if (POLICY == 6) LOADX4("sc1 nt");
            else LOADX4("sc0 sc1 nt");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; u++) acc += q[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int POLICY, int U>
void run(const char *name, const v4f *d, size_t bytes, int wgs, float *sink) {
    const size_t n_vec_per_wg = bytes / 16 / wgs;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((read_kernel<POLICY, U>), dim3(wgs), dim3(256), 0, 0, d, n_vec_per_wg, sink);
    CK(hipDeviceSynchronize());
    const int iters = 10;
    CK(hipEventRecord(e0, 0));
    for (int w = 0; w < iters; w++) hipLaunchKernelGGL((read_kernel<POLICY, U>), dim3(wgs), dim3(256), 0, 0, d, n_vec_per_wg, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-12s wgs %4d U %2d: %.3f ms  %.0f GB/s\n", name, wgs, U, ms / iters, (double)(n_vec_per_wg * 16 * wgs) / (ms / iters * 1e-3) / 1e9);
}

int main() {
    const size_t bytes = (size_t)8 << 30;
    v4f *d; float *sink;
    CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes)); CK(hipMalloc(&sink, 4));
    for (int wgs : {256, 1024}) {
        run<0, 18>("plain", d, bytes, wgs, sink);
        run<1, 18>("nt", d, bytes, wgs, sink);
        run<2, 18>("sc0", d, bytes, wgs, sink);
        run<3, 18>("sc1", d, bytes, wgs, sink);
        run<4, 18>("sc0 sc1", d, bytes, wgs, sink);
        run<5, 18>("sc0 nt", d, bytes, wgs, sink);
        run<6, 18>("sc1 nt", d, bytes, wgs, sink);
        run<7, 18>("sc0 sc1 nt", d, bytes, wgs, sink);
    }
    return 0;
}
