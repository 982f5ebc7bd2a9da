// hbm_read_probe -- what does a PURE read stream reach on this GPU?  Development tool: the
// ceiling against which the MAC kernel's achieved bytes/s are to be read (the 8 TB/s in bench.py's
// roofline is the specification figure; this is what the memory system delivers to the simplest
// possible kernel with the same access shape: 16 bytes per lane, contiguous per wave, U loads in
// flight per wave, one or two workgroups per CU).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hbm_read_probe.hip -o tools/hbm_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const v4f *gp;

// every workgroup walks its own contiguous slice in steps of U * 4 KiB
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const v4f *__restrict__ src, size_t n_vec_per_wg, float *__restrict__ sink) {
    const v4f *p = src + (size_t)blockIdx.x * n_vec_per_wg + threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    for (size_t i = 0; i + (size_t)U * 256 <= n_vec_per_wg; i += (size_t)U * 256) {
        v4f q[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT) q[u] = __builtin_nontemporal_load((gp)(const void *)(p + i + (size_t)u * 256));
            else q[u] = *(gp)(const void *)(p + i + (size_t)u * 256);
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += q[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;     // never true: keeps the loads
}

template <int U, bool NT>
void run(const v4f *d, size_t bytes, int wgs, float *sink) {
    const size_t n_vec_per_wg = bytes / 16 / wgs;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((read_kernel<U, NT>), dim3(wgs), dim3(256), 0, 0, d, n_vec_per_wg, sink);
    CK(hipDeviceSynchronize());
    const int iters = 10;
    CK(hipEventRecord(e0, 0));
    for (int w = 0; w < iters; w++) hipLaunchKernelGGL((read_kernel<U, NT>), dim3(wgs), dim3(256), 0, 0, d, n_vec_per_wg, sink);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("wgs %5d  loads in flight per wave %2d  %s: %.3f ms  %.0f GB/s\n", wgs, U, NT ? "nt" : "  ", ms / iters,
           (double)(n_vec_per_wg * 16 * wgs) / (ms / iters * 1e-3) / 1e9);
}

int main() {
    const size_t bytes = (size_t)8 << 30;
    v4f *d; float *sink;
    CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes)); CK(hipMalloc(&sink, 4));
    for (int wgs : {256, 512, 1024, 2048}) {
        run<9, true>(d, bytes, wgs, sink);
        run<18, true>(d, bytes, wgs, sink);
        run<27, true>(d, bytes, wgs, sink);
        run<27, false>(d, bytes, wgs, sink);
        run<36, true>(d, bytes, wgs, sink);
    }
    return 0;
}
