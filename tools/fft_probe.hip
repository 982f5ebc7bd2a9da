// fft_probe -- where does the time of the forward-FFT kernel (K1) go?  Development tool, not part
// of the product: it compiles kernels.h with BF_PROBE turned into cycle stamps taken by wave 0
// of workgroup 0 and prints the phase deltas next to the kernel's event-timed duration.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fft_probe.hip -o tools/fft_probe && tools/fft_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ unsigned long long g_probe[16];
#define BF_PROBE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_probe[i] = wall_clock64(); } while (0)
#include "../brutefir_amd/csrc/kernels.h"

using namespace bfhip;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// WAVE: the wave-FFT kernel (fft_wave.h) instead of the plain LDS Stockham one
template <int LOG2L, bool WAVE = false>
void run(int n_ch, int spacing_is_interleaved) {
    constexpr int L = 1 << LOG2L;
    constexpr int NT = WAVE ? L / 16 : fft_threads<float>(LOG2L);
    const size_t lds = lds_fft_bytes(LOG2L, sizeof(c2<float>));
    std::vector<DevFormat> fmt(n_ch);
    for (int c = 0; c < n_ch; c++) {
        DevFormat f{};
        f.isfloat = 0; f.swap = 0; f.bytes = 4; f.sbytes = 3;
        f.sample_spacing = spacing_is_interleaved ? n_ch : 1;
        f.byte_offset = spacing_is_interleaved ? 4 * c : 4 * c * L;
        f.alt = nullptr;
        fmt[c] = f;
    }
    DevFormat *d_fmt; uint8_t *d_raw; float *d_prev; c2<float> *d_ring, *d_tw;
    const int R = 4;
    CK(hipMalloc(&d_fmt, n_ch * sizeof(DevFormat)));
    CK(hipMemcpy(d_fmt, fmt.data(), n_ch * sizeof(DevFormat), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_raw, (size_t)n_ch * L * 4)); CK(hipMemset(d_raw, 1, (size_t)n_ch * L * 4));
    CK(hipMalloc(&d_prev, (size_t)n_ch * L * 4)); CK(hipMemset(d_prev, 0, (size_t)n_ch * L * 4));
    CK(hipMalloc(&d_ring, (size_t)n_ch * R * L * 8));
    const std::vector<unsigned char> tw = WAVE ? make_wave_twiddle_table(LOG2L, 4) : make_twiddle_table(LOG2L, 4, NT);
    CK(hipMalloc(&d_tw, tw.size())); CK(hipMemcpy(d_tw, tw.data(), tw.size(), hipMemcpyHostToDevice));
    void (*k)(const uint8_t *, const DevFormat *, float *, c2<float> *, const c2<float> *, int, int, const BlockState *, PowerSave);
    if constexpr (WAVE) k = fft_in_wave_kernel<float, LOG2L>; else k = fft_in_kernel<float, LOG2L>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 50;
    for (int w = 0; w < 5; w++) hipLaunchKernelGGL(k, dim3(n_ch), dim3(NT), lds, 0, d_raw, d_fmt, d_prev, d_ring, d_tw, R, w % R, (const BlockState *)nullptr, PowerSave{0.0, nullptr, nullptr, nullptr});
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int w = 0; w < iters; w++) hipLaunchKernelGGL(k, dim3(n_ch), dim3(NT), lds, 0, d_raw, d_fmt, d_prev, d_ring, d_tw, R, w % R, (const BlockState *)nullptr, PowerSave{0.0, nullptr, nullptr, nullptr});
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long p[16];
    CK(hipMemcpyFromSymbol(p, HIP_SYMBOL(g_probe), sizeof(p)));
    { unsigned long long zero[16] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_probe), zero, sizeof(zero))); }
    printf("%s L=%d n_ch=%d %s threads=%d: %.2f us per launch back-to-back; phases (us, 100 MHz clock): ", WAVE ? "wave" : "lds ", L, n_ch,
           spacing_is_interleaved ? "interleaved" : "planar", NT, ms * 1e3 / iters);
    const int idx[] = {0, 1, 2, 4, 5, 6, 7, 8, 10, 11};
    const char *nm_lds[] = {"start", "loads issued", "lds filled", "pass0", "pass1", "pass2", "pass3", "pass4", "fft done", "stores issued"};
    const char *nm_wave[] = {"start", "loads issued", "pass0 from registers", "-", "-", "-", "-", "passes 1-3 (in-wave exchange)", "spectrum in lds", "stores issued"};
    const char **nm = WAVE ? nm_wave : nm_lds;
    for (int i = 1; i < 10; i++) {
        if (p[idx[i]] == 0 || p[idx[i]] < p[0]) continue;
        printf("%s +%.2f | ", nm[i], (double)(p[idx[i]] - p[0]) * 0.01);
    }
    printf("\n");
    hipFree(d_fmt); hipFree(d_raw); hipFree(d_prev); hipFree(d_ring); hipFree(d_tw);
}

// K3 (wave FFT): spectrum partials -> inverse transform -> quantise -> raw
template <int LOG2L>
void run_k3(int n_ch, int interleaved, int n_chunks) {
    constexpr int L = 1 << LOG2L, NT = L / 16;
    const size_t lds = lds_fft_bytes(LOG2L, sizeof(c2<float>));
    std::vector<DevFormat> fmt(n_ch);
    std::vector<DevOverflow> ov(n_ch);
    for (int c = 0; c < n_ch; c++) {
        DevFormat f{};
        f.isfloat = 0; f.swap = 0; f.bytes = 4; f.sbytes = 3;
        f.sample_spacing = interleaved ? n_ch : 1;
        f.byte_offset = interleaved ? 4 * c : 4 * c * L;
        f.alt = nullptr;
        fmt[c] = f;
        ov[c] = DevOverflow{0, 0, 0.0, 8388607.0};
    }
    DevFormat *d_fmt; DevOverflow *d_ov; uint8_t *d_raw; c2<float> *d_z, *d_tw; int *d_st;
    CK(hipMalloc(&d_fmt, n_ch * sizeof(DevFormat)));
    CK(hipMemcpy(d_fmt, fmt.data(), n_ch * sizeof(DevFormat), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_ov, n_ch * sizeof(DevOverflow)));
    CK(hipMemcpy(d_ov, ov.data(), n_ch * sizeof(DevOverflow), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_raw, (size_t)n_ch * L * 4));
    CK(hipMalloc(&d_z, (size_t)n_chunks * n_ch * L * 8)); CK(hipMemset(d_z, 0, (size_t)n_chunks * n_ch * L * 8));
    CK(hipMalloc(&d_st, 4)); CK(hipMemset(d_st, 0, 4));
    const std::vector<unsigned char> tw = make_wave_twiddle_table(LOG2L, 4);
    CK(hipMalloc(&d_tw, tw.size())); CK(hipMemcpy(d_tw, tw.data(), tw.size(), hipMemcpyHostToDevice));
    auto k = ifft_out_wave_kernel<float, LOG2L>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 50;
    auto launch = [&]() {
        hipLaunchKernelGGL(k, dim3(n_ch), dim3(NT), lds, 0, (const c2<float> *)d_z, (size_t)n_ch * L, n_chunks, 0,
                           (const DevFormat *)d_fmt, d_ov, (const unsigned char *)nullptr, d_raw, (float *)nullptr,
                           (const c2<float> *)d_tw, 0.0, d_st);
    };
    for (int w = 0; w < 5; w++) launch();
    CK(hipDeviceSynchronize());
    { unsigned long long zero[16] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_probe), zero, sizeof(zero))); }
    CK(hipEventRecord(e0, 0));
    for (int w = 0; w < iters; w++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long p[16];
    CK(hipMemcpyFromSymbol(p, HIP_SYMBOL(g_probe), sizeof(p)));
    printf("K3 wave L=%d n_ch=%d %s chunks=%d threads=%d: %.2f us per launch back-to-back; phases: ", L, n_ch,
           interleaved ? "interleaved" : "planar", n_chunks, NT, ms * 1e3 / iters);
    const int idx[] = {0, 1, 2, 4, 8, 10, 11};
    const char *nm[] = {"start", "loads issued", "tangled into lds", "pass0", "passes 1-3", "quantised + stores issued", "reduced"};
    for (int i = 1; i < 7; i++) if (p[idx[i]] >= p[0] && p[idx[i]] != 0) printf("%s +%.2f | ", nm[i], (double)(p[idx[i]] - p[0]) * 0.01);
    printf("\n");
    hipFree(d_fmt); hipFree(d_ov); hipFree(d_raw); hipFree(d_z); hipFree(d_tw); hipFree(d_st);
}

int main() {
    run_k3<13>(64, 1, 2);
    run_k3<13>(64, 0, 2);
    run_k3<13>(8, 1, 1);
    run<12>(64, 1);
    run<12, true>(64, 1);
    run<11>(64, 1);
    run<11, true>(64, 1);
    run<13>(64, 1);
    run<13, true>(64, 1);
    run<13>(64, 0);
    run<13, true>(64, 0);
    run<13>(8, 1);
    run<13, true>(8, 1);
    run<13>(1, 1);
    run<13, true>(1, 1);
    run<10>(64, 1);
    run<10, true>(64, 1);
    run<8>(8, 1);
    return 0;
}
