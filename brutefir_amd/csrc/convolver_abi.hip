// convolver_abi.hip -- the per-block symbols of the reference's convolver.h
// (include/bfhip_convolver.h), host-memory semantics, executed on the device.  The symbols
// that run before the fork or in module processes (init, coeffs2cbuf, runtime_coeffs2cbuf,
// verify, debug dump, fftplan, td_new) are pure host code in host_ops.cpp.
//
// Layouts here are the REFERENCE's (halfcomplex, "4 re / 4 im" reordered), not the engine's
// packed spectra: these entry points exchange buffers with unmodified host code
// (bfconf.c, delay.c, bflogic_eq, "processed" coefficient files).  The FFT-free ops use
// explicitly rounded multiplies/adds (no FMA contraction) in the reference's order, so they
// return the same bits as the reference's C loops.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/bfhip_convolver.h"
#include "conv_shared.h"
#include "kernels.h"
#include "bigfft.h"

using namespace bfhip;

// The FFT-free ops below must round like the reference's C loops (gcc, no FMA): HIP's
// __fmul_rn/__fadd_rn are plain operators that clang may still fuse, so contraction is
// switched off for everything defined in this file.
#pragma clang fp contract(off)

// the host's dither tables (dither.c:20-22); absent when the library is used stand-alone
extern "C" {
__attribute__((weak)) extern int8_t *dither_randtab;
__attribute__((weak)) extern int dither_randtab_size;
__attribute__((weak)) extern void *dither_randmap;
}

namespace {

// ---------------------------------------------------------------- exact (uncontracted) arithmetic
// plain operators: compiled under `fp contract(off)` above they carry no contract flag, so the
// backend cannot fuse them (HIP's __fmul_rn & co. are inline functions from a header compiled
// with contraction on -- they DO get fused)
template <typename T> __device__ __forceinline__ T nmul(T a, T b) { return a * b; }
template <typename T> __device__ __forceinline__ T nadd(T a, T b) { return a + b; }
template <typename T> __device__ __forceinline__ T nsub(T a, T b) { return a - b; }

// ---------------------------------------------------------------- kernels on reference layouts

// mixnscale (fftw_convfuns.h:7-501): one lane per bin, both halves of the bin
template <typename T>
__global__ void k_mix(const T *const *in, T *out, const T *scales, int n, int mode, int L) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= L) return;
    const int hre = k, him = (k == 0) ? L : 2 * L - k;
    const int qre = 8 * (k >> 2) + (k & 3), qim = qre + 4;
    const int sre = mode == CONVOLVER_MIXMODE_INPUT ? hre : qre;
    const int sim = mode == CONVOLVER_MIXMODE_INPUT ? him : qim;
    T a = nmul(in[0][sre], scales[0]), b = nmul(in[0][sim], scales[0]);
    for (int i = 1; i < n; i++) {
        a = nadd(a, nmul(in[i][sre], scales[i]));
        b = nadd(b, nmul(in[i][sim], scales[i]));
    }
    if (mode == CONVOLVER_MIXMODE_INPUT) { out[qre] = a; out[qim] = b; }
    else { out[hre] = a; out[him] = b; }
}

// convolve / convolve_add (fftw_convfuns.h:503-590); add = 0 assigns, 1 accumulates
template <typename T>
__global__ void k_conv(const T *b, const T *h, T *d, int add, int L) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= L) return;
    const int qre = 8 * (k >> 2) + (k & 3), qim = qre + 4;
    const T br = b[qre], bi = b[qim], hr = h[qre], hi = h[qim];
    T re, im;
    if (k == 0) { re = nmul(br, hr); im = nmul(bi, hi); }           // DC and Nyquist slots
    else { re = nsub(nmul(br, hr), nmul(bi, hi)); im = nadd(nmul(br, hi), nmul(bi, hr)); }
    if (add) { re = nadd(d[qre], re); im = nadd(d[qim], im); }
    d[qre] = re;
    d[qim] = im;
}

template <typename T>
__global__ void k_dirac(const T *in, T *out, int L) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= 2 * L) return;
    const T f = (T)1.0 / (T)(2 * L);
    out[n] = nmul(in[n], (n & 1) ? -f : f);
}

// plain halfcomplex product of convolve_inplace_ordered (fftw_convolver.c:738-765)
template <typename T>
__global__ void k_conv_ordered(T *b, const T *c, int size) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int size2 = size >> 1;
    if (n > size2) return;
    if (n == 0 || n == size2) { b[n] = nmul(b[n], c[n]); return; }
    const T a = b[n], bi = b[size - n];
    b[n] = nsub(nmul(a, c[n]), nmul(bi, c[size - n]));
    b[size - n] = nadd(nmul(a, c[size - n]), nmul(bi, c[n]));
}

// FFTW R2HC of 2L reals (in may equal out)
template <typename T, int LOG2L>
__global__ __launch_bounds__(fft_threads<T>(LOG2L)) void
k_r2hc(const T *in, T *out, const c2<T> *__restrict__ tw) {
    constexpr int L = 1 << LOG2L, NT = fft_threads<T>(LOG2L);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsArr<T> s{reinterpret_cast<c2<T> *>(smem)};
    const int tid = threadIdx.x;
    TwRegs<T, LOG2L, NT> twr;
    twr.prefetch(tw);
    for (int n = tid; n < L; n += NT) s[n] = mk<T>(in[2 * n], in[2 * n + 1]);
    __syncthreads();
    lds_fft<T, LOG2L, NT, false>(s, twr);
    for (int k = tid; k <= L / 2; k += NT) {
        if (k == 0) {
            const c2<T> z = s[0];
            out[0] = z.x + z.y;
            out[L] = z.x - z.y;
        } else {
            c2<T> x, y;
            untangle(s[k], conj(s[L - k]), tw[k], x, y);
            out[k] = x.x; out[2 * L - k] = x.y;
            if (k != L - k) { out[L - k] = y.x; out[L + k] = y.y; }
        }
    }
}

// FFTW HC2R, unnormalised (in may equal out)
template <typename T, int LOG2L>
__global__ __launch_bounds__(fft_threads<T>(LOG2L)) void
k_hc2r(const T *in, T *out, const c2<T> *__restrict__ tw) {
    constexpr int L = 1 << LOG2L, NT = fft_threads<T>(LOG2L);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsArr<T> s{reinterpret_cast<c2<T> *>(smem)};
    const int tid = threadIdx.x;
    TwRegs<T, LOG2L, NT> twr;
    twr.prefetch(tw);
    for (int k = tid; k <= L / 2; k += NT) {
        if (k == 0) {
            s[0] = mk<T>(in[0] + in[L], in[0] - in[L]);
        } else {
            const c2<T> a = mk<T>(in[k], in[2 * L - k]);
            const c2<T> b = (k == L - k) ? conj(a) : mk<T>(in[L - k], -in[L + k]);
            c2<T> zk, zlk;
            tangle(a, b, tw[k], zk, zlk);
            s[k] = zk;
            if (k != L - k) s[L - k] = zlk;
        }
    }
    __syncthreads();
    lds_fft<T, LOG2L, NT, true>(s, twr);
    for (int n = tid; n < L; n += NT) { const c2<T> z = s[n]; out[2 * n] = z.x; out[2 * n + 1] = z.y; }
}

// ---- transforms above the LDS limit (bigfft.h): pack / unpack around the two-stage complex FFT
template <typename T>
__global__ void k_big_pack(const T *in, c2<T> *zin, int L) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < L) zin[n] = mk<T>(in[2 * n], in[2 * n + 1]);
}
template <typename T>
__global__ void k_big_unpack(const c2<T> *zout, T *out, int L) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < L) { const c2<T> z = zout[n]; out[2 * n] = z.x; out[2 * n + 1] = z.y; }
}
template <typename T>
__global__ void k_big_r2hc_post(const c2<T> *s, T *out, const c2<T> *__restrict__ tw, int L) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > L / 2) return;
    if (k == 0) { const c2<T> z = s[0]; out[0] = z.x + z.y; out[L] = z.x - z.y; return; }
    c2<T> x, y;
    untangle(s[k], conj(s[L - k]), tw[k], x, y);
    out[k] = x.x; out[2 * L - k] = x.y;
    if (k != L - k) { out[L - k] = y.x; out[L + k] = y.y; }
}
template <typename T>
__global__ void k_big_hc2r_pre(const T *in, c2<T> *s, const c2<T> *__restrict__ tw, int L) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > L / 2) return;
    if (k == 0) { s[0] = mk<T>(in[0] + in[L], in[0] - in[L]); return; }
    const c2<T> a = mk<T>(in[k], in[2 * L - k]);
    const c2<T> b = (k == L - k) ? conj(a) : mk<T>(in[L - k], -in[L + k]);
    c2<T> zk, zlk;
    tangle(a, b, tw[k], zk, zlk);
    s[k] = zk;
    if (k != L - k) s[L - k] = zlk;
}

// the ramp of convolver_crossfade_inplace, float-branch arithmetic (fftw_convolver.c:349-355)
template <typename T>
__global__ void k_fade(const T *oldt, T *newt, int L) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= L) return;
    if constexpr (sizeof(T) == 4) {
        const float f = 1.0f / (float)(L - 1);
        const float fn = f * (float)n;
        const double a = (double)oldt[n] * (1.0 - (double)fn);
        const double b = (double)((newt[n] * f) * (float)n);
        newt[n] = (float)(a + b);
    } else {
        const double d = 1.0 / (double)(L - 1);
        const double a = oldt[n] * (1.0 - d * (double)n);
        const double b = (newt[n] * d) * (double)n;
        newt[n] = a + b;
    }
}

template <typename T>
__global__ void k_raw2real(const uint8_t *raw, DevFormat f, T *out, int n_samples) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_samples) return;
    out[n] = load_raw<T>(raw + f.byte_offset + (size_t)n * f.sample_spacing * f.bytes, f);
}

// real2raw without dither (real2raw.h:61-250 + dither_funs.h:71-114), one workgroup
template <typename T>
__global__ __launch_bounds__(256) void
k_real2raw(const T *real, uint8_t *raw, DevFormat f, int n_samples, DevOverflow *over,
           double safety_limit, int *status) {
    const int tid = threadIdx.x;
    DevOverflow of = *over;
    Quantiser<T> qz;
    qz.init(f, of, safety_limit);
    uint8_t *base = raw + f.byte_offset;
    const size_t stride = (size_t)f.sample_spacing * f.bytes;
    for (int n = tid; n < n_samples; n += 256) qz.put(real[n], base + (size_t)n * stride);
    qz.reduce(tid, 256);
    if (tid == 0) {
        qz.commit(of);
        *over = of;
        if (qz.st) atomicOr(status, qz.st);
    }
}

// HP-TPDF pass on host-owned dither state: the table bytes of this block travel with the call
template <typename T>
__global__ void k_real2raw_dither(const T *real, uint8_t *raw, DevFormat f, int n_samples,
                                  const int8_t *tab /* [-1 .. n) shifted by one */, const T *randmap,
                                  T *fb /* s0, s1 */, DevOverflow *over, double safety_limit, int *status) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    DevOverflow of = *over;
    const int bits = f.sbytes << 3;
    const int32_t imin = (int32_t)(-((uint64_t)1 << (bits - 1)));
    const int32_t imax = (int32_t)(((uint64_t)1 << (bits - 1)) - 1);
    const T rmin = (T)imin, rmax = (T)imax;
    T s0 = fb[0], s1 = fb[1];
    int flags = 0;
    uint8_t *base = raw + f.byte_offset;
    const size_t stride = (size_t)f.sample_spacing * f.bytes;
    for (int n = 0; n < n_samples; n++) {
        T v = real[n];
        if (!isfinite(v)) { flags |= 1; continue; }
        if (safety_limit != 0.0 && ((double)v < -safety_limit * of.max || (double)v > safety_limit * of.max)) { flags |= 2; continue; }
        v = nadd(v, nsub(s0, s1));
        s1 = s0;
        const T dv = nadd(v, randmap[(int)tab[n + 1] - (int)tab[n]]);
        int32_t q;
        if (dv < 0) {
            if (dv <= rmin) { q = imin; of.n_overflows++; if ((double)v < -of.largest) of.largest = (double)-dv; }
            else { q = (int32_t)dv; q--; if (q < -of.intlargest) of.intlargest = -q; }
        } else {
            if (dv > rmax) { q = imax; of.n_overflows++; if ((double)v > of.largest) of.largest = (double)dv; }
            else { q = (int32_t)dv; if (q > of.intlargest) of.intlargest = q; }
        }
        s0 = nsub(v, (T)q);
        const uint32_t u = (uint32_t)q;
        store_raw_word(base + (size_t)n * stride, (uint64_t)u, f.bytes, f.swap);
    }
    fb[0] = s0; fb[1] = s1;
    *over = of;
    if (flags) atomicOr(status, flags);
}

// ---------------------------------------------------------------- process-wide state

struct State {
    int L = 0, rs = 0, log2L = -1;
    bool inited = false;
    pid_t pid = 0;
    hipStream_t stream = nullptr;
    std::map<int, void *> tw;          // log2(complex length) -> device twiddles
    void *d_big[3] = {nullptr, nullptr, nullptr};   // scratch of the transforms above the LDS limit
    size_t big_bytes[3] = {0, 0, 0};
    std::vector<void *> buf;
    std::vector<size_t> cap;
    int *d_flag = nullptr;
    DevOverflow *d_over = nullptr;
} G;

void fatal(int code, const char *fmt, ...) {
    char msg[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof(msg), fmt, ap);
    va_end(ap);
    bfhip_conv_fatal(code, msg);              /* handler, or print + exit(BF_EXIT_OTHER) */
}

#define DCHK(expr)                                                                     \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) { fatal(100, "bfhip: %s failed: %s", #expr, hipGetErrorString(_e)); return false; } \
    } while (0)

bool ensure_device() {
    if (!bfhip_conv_g.inited) { fatal(101, "convolver_init() has not been called."); return false; }
    // convolver_init() may have been called again (tests do): the sizes live in host_ops.cpp
    if (G.rs != bfhip_conv_g.rs && !G.tw.empty()) {
        // the twiddle tables are per precision
        if (G.stream) (void)hipStreamSynchronize(G.stream);
        for (auto &kv : G.tw) (void)hipFree(kv.second);
        G.tw.clear();
    }
    G.L = bfhip_conv_g.L; G.rs = bfhip_conv_g.rs; G.log2L = bfhip_conv_g.log2L; G.inited = true;
    const pid_t me = getpid();
    if (G.pid == me && G.stream) return true;
    if (G.pid != 0 && G.pid != me) {
        // HIP state does not survive fork(): a child of a process that already ran a device op
        // through this library inherits a dead runtime.  Nothing the host does before the fork
        // reaches a device op (those entry points are pure host code, host_ops.cpp), so this is a
        // host bug: say so instead of hanging in the runtime.
        fatal(106, "bfhip: the HIP runtime was initialised in process %d before fork(); device "
                   "calls are only valid in the process that made the first one", (int)G.pid);
        return false;
    }
    // first device use in this process
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fatal(102, "bfhip: no HIP device available (there is no CPU fallback)");
        return false;
    }
    G.pid = me;
    DCHK(hipStreamCreateWithFlags(&G.stream, hipStreamNonBlocking));
    DCHK(hipMalloc((void **)&G.d_flag, sizeof(int)));
    DCHK(hipMemset(G.d_flag, 0, sizeof(int)));
    DCHK(hipMalloc((void **)&G.d_over, sizeof(DevOverflow)));
    return true;
}

void *scratch(int i, size_t bytes) {
    if ((int)G.buf.size() <= i) { G.buf.resize(i + 1, nullptr); G.cap.resize(i + 1, 0); }
    if (G.cap[i] < bytes) {
        if (G.buf[i]) { (void)hipStreamSynchronize(G.stream); (void)hipFree(G.buf[i]); }
        if (hipMalloc(&G.buf[i], bytes) != hipSuccess) { fatal(103, "bfhip: out of device memory"); return nullptr; }
        G.cap[i] = bytes;
    }
    return G.buf[i];
}

const void *twiddles(int log2c) {
    auto it = G.tw.find(log2c);
    if (it != G.tw.end()) return it->second;
    const std::vector<unsigned char> h = make_twiddle_table(log2c, G.rs,
        G.rs == 4 ? fft_threads<float>(log2c) : fft_threads<double>(log2c));
    void *d = nullptr;
    if (hipMalloc(&d, h.size()) != hipSuccess || hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice) != hipSuccess) {
        fatal(103, "bfhip: twiddle upload failed");
        return nullptr;
    }
    G.tw[log2c] = d;
    return d;
}

size_t csz() { return (size_t)2 * G.L * G.rs; }          // one cbuf in bytes

bool up(void *dst, const void *src, size_t n) { DCHK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, G.stream)); return true; }
bool down(void *dst, const void *src, size_t n) {
    DCHK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, G.stream));
    DCHK(hipStreamSynchronize(G.stream));
    return true;
}

template <typename T, int LOG2>
void fft_launch(bool inverse, const void *in, void *out) {
    constexpr int NT = fft_threads<T>(LOG2);
    const size_t lds = lds_fft_bytes(LOG2, sizeof(c2<T>));
    const c2<T> *tw = (const c2<T> *)twiddles(LOG2);
    if (inverse) {
        auto k = k_hc2r<T, LOG2>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(1), dim3(NT), lds, G.stream, (const T *)in, (T *)out, tw);
    } else {
        auto k = k_r2hc<T, LOG2>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(1), dim3(NT), lds, G.stream, (const T *)in, (T *)out, tw);
    }
}

// 2^log2c complex points, log2c in 14..16: the input may alias the output (everything goes
// through the scratch buffers)
template <typename T>
bool big_fft_launch(int log2c, bool inverse, const void *in, void *out) {
    const int L = 1 << log2c;
    const size_t bytes = (size_t)L * sizeof(c2<T>);
    for (int i = 0; i < 3; i++) {
        if (G.big_bytes[i] < bytes) {
            if (G.d_big[i]) (void)hipFree(G.d_big[i]);
            G.d_big[i] = nullptr;
            if (hipMalloc(&G.d_big[i], bytes) != hipSuccess) { fatal(105, "bfhip: FFT scratch allocation failed"); return false; }
            G.big_bytes[i] = bytes;
        }
    }
    c2<T> *zin = (c2<T> *)G.d_big[0], *zmid = (c2<T> *)G.d_big[1], *zout = (c2<T> *)G.d_big[2];
    const c2<T> *twL = (const c2<T> *)twiddles(log2c), *tw13 = (const c2<T> *)twiddles(BIG_LOG2M);
    if (!twL || !tw13) return false;
    const dim3 gh((unsigned)(L / 2 / 256 + 1)), gf((unsigned)(L / 256));
    hipError_t ferr;
    if (inverse) {
        hipLaunchKernelGGL(k_big_hc2r_pre<T>, gh, dim3(256), 0, G.stream, (const T *)in, zin, twL, L);
        ferr = big_fft_run<T, true>((const c2<T> *)zin, zmid, zout, log2c, 1, tw13, twL, G.stream);
        hipLaunchKernelGGL(k_big_unpack<T>, gf, dim3(256), 0, G.stream, (const c2<T> *)zout, (T *)out, L);
    } else {
        hipLaunchKernelGGL(k_big_pack<T>, gf, dim3(256), 0, G.stream, (const T *)in, zin, L);
        ferr = big_fft_run<T, false>((const c2<T> *)zin, zmid, zout, log2c, 1, tw13, twL, G.stream);
        hipLaunchKernelGGL(k_big_r2hc_post<T>, gh, dim3(256), 0, G.stream, (const c2<T> *)zout, (T *)out, twL, L);
    }
    if (ferr != hipSuccess) return false;
    return hipGetLastError() == hipSuccess;
}

// device-side real FFT of 2^(log2c+1) reals, buffers on the device
bool dev_fft(int log2c, bool inverse, const void *in, void *out) {
    if (log2c > BIG_LOG2M && log2c <= 20)
        return G.rs == 4 ? big_fft_launch<float>(log2c, inverse, in, out) : big_fft_launch<double>(log2c, inverse, in, out);
#define CASE(n) case n: if (G.rs == 4) fft_launch<float, n>(inverse, in, out); else fft_launch<double, n>(inverse, in, out); break;
    switch (log2c) {
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
    default:
        fatal(104, "bfhip: FFT size 2^%d is not supported on the device", log2c + 1);
        return false;
    }
#undef CASE
    return hipGetLastError() == hipSuccess;
}

template <typename F> void per_type(F f) { if (G.rs == 4) f((float)0); else f((double)0); }

int grid(int n, int b = 256) { return (n + b - 1) / b; }

bool dev_mix(void *const d_in[], int n, void *d_out, const double scales[], int mode) {
    std::vector<unsigned char> sc((size_t)n * G.rs);
    for (int i = 0; i < n; i++) {
        if (G.rs == 4) ((float *)sc.data())[i] = (float)scales[i]; else ((double *)sc.data())[i] = scales[i];
    }
    void *d_ptrs = scratch(10, n * sizeof(void *)), *d_sc = scratch(11, sc.size());
    if (!d_ptrs || !d_sc) return false;
    if (!up(d_ptrs, d_in, n * sizeof(void *)) || !up(d_sc, sc.data(), sc.size())) return false;
    DCHK(hipStreamSynchronize(G.stream));     // sc / d_in are stack temporaries
    per_type([&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(k_mix<T>, dim3(grid(G.L)), dim3(256), 0, G.stream, (const T *const *)d_ptrs, (T *)d_out, (const T *)d_sc, n, mode, G.L);
    });
    return true;
}

void dev_conv(const void *b, const void *h, void *d, int add) {
    per_type([&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(k_conv<T>, dim3(grid(G.L)), dim3(256), 0, G.stream, (const T *)b, (const T *)h, (T *)d, add, G.L);
    });
}

DevFormat devfmt(const bfhip_buffer_format *bf, bool with_offset) {
    DevFormat f;
    f.isfloat = bf->sf.isfloat; f.swap = bf->sf.swap; f.bytes = bf->sf.bytes; f.sbytes = bf->sf.sbytes;
    f.sample_spacing = bf->sample_spacing; f.byte_offset = with_offset ? bf->byte_offset : 0;
    return f;
}

size_t raw_span(const bfhip_buffer_format *bf, int n) {
    return ((size_t)(n - 1) * bf->sample_spacing + 1) * bf->sf.bytes;
}

bool format_ok(const bfhip_buffer_format *bf) {
    const int b = bf->sf.bytes;
    if (bf->sf.isfloat ? (b != 4 && b != 8) : (b < 1 || b > 4)) {
        fatal(1, "Sample byte size %d is not supported.", b);      // raw2real.h:155-158
        return false;
    }
    return true;
}

}  // namespace

// ==================================================================== the 22 symbols

extern "C" {

void convolver_raw2cbuf(void *rawbuf, void *cbuf, void *next_cbuf, struct bfhip_buffer_format *bf,
                        void (*postprocess)(void *, int, void *), void *pp_arg) {
    if (!ensure_device() || !format_ok(bf)) return;
    const size_t span = raw_span(bf, G.L), half = (size_t)G.L * G.rs;
    void *d_raw = scratch(0, span), *d_real = scratch(1, half);
    if (!d_raw || !d_real) return;
    if (!up(d_raw, (const uint8_t *)rawbuf + bf->byte_offset, span)) return;
    const DevFormat f = devfmt(bf, false);
    per_type([&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(k_raw2real<T>, dim3(grid(G.L)), dim3(256), 0, G.stream, (const uint8_t *)d_raw, f, (T *)d_real, G.L);
    });
    if (!down(next_cbuf, d_real, half)) return;
    if (postprocess != NULL) postprocess(next_cbuf, G.L, pp_arg);
    memcpy((uint8_t *)cbuf + half, next_cbuf, half);        /* fftw_convolver.c:193 */
}

static void fft_host(int log2c, bool inverse, void *in, void *out) {
    if (!ensure_device()) return;
    const size_t bytes = ((size_t)2 << log2c) * G.rs;
    void *d = scratch(0, bytes);
    if (!d || !up(d, in, bytes)) return;
    if (!dev_fft(log2c, inverse, d, d)) return;
    down(out, d, bytes);
}

void convolver_time2freq(void *input_cbuf, void *output_cbuf) { fft_host(bfhip_conv_g.log2L, false, input_cbuf, output_cbuf); }
void convolver_freq2time(void *input_cbuf, void *output_cbuf) { fft_host(bfhip_conv_g.log2L, true, input_cbuf, output_cbuf); }

void convolver_mixnscale(void *input_cbufs[], void *output_cbuf, double scales[], int n_bufs, int mixmode) {
    if (mixmode != CONVOLVER_MIXMODE_INPUT && mixmode != CONVOLVER_MIXMODE_OUTPUT) {
        fatal(1, "Invalid mixmode: %d.", mixmode);            /* fftw_convfuns.h:496-499 */
        return;
    }
    if (!ensure_device() || n_bufs < 1) return;
    void *d_all = scratch(0, csz() * (n_bufs + 1));
    if (!d_all) return;
    std::vector<void *> ptrs(n_bufs);
    for (int i = 0; i < n_bufs; i++) {
        ptrs[i] = (uint8_t *)d_all + csz() * i;
        if (!up(ptrs[i], input_cbufs[i], csz())) return;
    }
    void *d_out = (uint8_t *)d_all + csz() * n_bufs;
    if (!dev_mix(ptrs.data(), n_bufs, d_out, scales, mixmode)) return;
    down(output_cbuf, d_out, csz());
}

static void conv_host(void *b, void *h, void *d, int add, bool dirac) {
    if (!ensure_device()) return;
    void *db = scratch(0, csz()), *dh = scratch(1, csz()), *dd = scratch(2, csz());
    if (!db || !dh || !dd) return;
    if (!up(db, b, csz())) return;
    if (!dirac && !up(dh, h, csz())) return;
    if (add && !up(dd, d, csz())) return;
    if (dirac) {
        per_type([&](auto t) {
            using T = decltype(t);
            hipLaunchKernelGGL(k_dirac<T>, dim3(grid(2 * G.L)), dim3(256), 0, G.stream, (const T *)db, (T *)dd, G.L);
        });
    } else {
        dev_conv(db, dh, dd, add);
    }
    down(d, dd, csz());
}

void convolver_convolve_inplace(void *cbuf, void *coeffs) { conv_host(cbuf, coeffs, cbuf, 0, false); }
void convolver_convolve(void *input_cbuf, void *coeffs, void *output_cbuf) { conv_host(input_cbuf, coeffs, output_cbuf, 0, false); }
void convolver_convolve_add(void *input_cbuf, void *coeffs, void *output_cbuf) { conv_host(input_cbuf, coeffs, output_cbuf, 1, false); }
void convolver_dirac_convolve(void *input_cbuf, void *output_cbuf) { conv_host(input_cbuf, NULL, output_cbuf, 0, true); }
void convolver_dirac_convolve_inplace(void *cbuf) { conv_host(cbuf, NULL, cbuf, 0, true); }

void convolver_crossfade_inplace(void *input_cbuf, void *crossfade_cbuf, void *buffer_cbuf) {
    if (!ensure_device()) return;
    void *d_in = scratch(0, csz()), *d_cf = scratch(1, csz()), *d_buf = scratch(2, csz());
    if (!d_in || !d_cf || !d_buf) return;
    if (!up(d_in, input_cbuf, csz()) || !up(d_cf, crossfade_cbuf, csz())) return;
    const double one = 1.0, inv = 1.0 / (double)(2 * G.L);
    void *p;
    p = d_cf;
    if (!dev_mix(&p, 1, d_buf, &one, CONVOLVER_MIXMODE_OUTPUT)) return;
    if (!dev_fft(G.log2L, true, d_buf, d_cf)) return;
    p = d_in;
    if (!dev_mix(&p, 1, d_buf, &one, CONVOLVER_MIXMODE_OUTPUT)) return;
    if (!dev_fft(G.log2L, true, d_buf, d_buf)) return;
    per_type([&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(k_fade<T>, dim3(grid(G.L)), dim3(256), 0, G.stream, (const T *)d_cf, (T *)d_buf, G.L);
    });
    if (!dev_fft(G.log2L, false, d_buf, d_buf)) return;
    p = d_buf;
    if (!dev_mix(&p, 1, d_in, &inv, CONVOLVER_MIXMODE_INPUT)) return;
    /* the reference leaves the time-domain old result in crossfade_cbuf and the re-FFT'd
       fade in buffer_cbuf; callers only use input_cbuf, but keep the side effects */
    if (!down(crossfade_cbuf, d_cf, csz()) || !down(buffer_cbuf, d_buf, csz())) return;
    down(input_cbuf, d_in, csz());
}

void convolver_convolve_eval(void *input_cbuf, void *buffer_cbuf, void *output_cbuf) {
    if (!ensure_device()) return;
    const size_t half = (size_t)G.L * G.rs;
    void *d_in = scratch(0, csz()), *d_buf = scratch(1, 3 * half), *d_out = scratch(2, csz());
    if (!d_in || !d_buf || !d_out) return;
    if (!up(d_in, input_cbuf, csz()) || !up(d_buf, buffer_cbuf, half)) return;
    if (!dev_fft(G.log2L, true, d_in, (uint8_t *)d_buf + half)) return;
    if (!dev_fft(G.log2L, false, d_buf, d_out)) return;
    if (!down(output_cbuf, d_out, csz())) return;
    /* buffer keeps [valid half | garbage half] shifted down, as fftw_convolver.c:431-432 */
    if (!down((uint8_t *)buffer_cbuf + half, (uint8_t *)d_buf + half, 2 * half)) return;
    memcpy(buffer_cbuf, (uint8_t *)buffer_cbuf + half, half);
}

void convolver_cbuf2raw(void *cbuf, void *outbuf, struct bfhip_buffer_format *bf, int apply_dither,
                        void *dither_state, struct bfhip_overflow *overflow) {
    if (!ensure_device() || !format_ok(bf)) return;
    const size_t span = raw_span(bf, G.L), half = (size_t)G.L * G.rs;
    void *d_real = scratch(0, half), *d_raw = scratch(1, span);
    if (!d_real || !d_raw) return;
    uint8_t *hraw = (uint8_t *)outbuf + bf->byte_offset;
    /* interleaved neighbours share the span: keep their bytes */
    if (!up(d_real, cbuf, half) || !up(d_raw, hraw, span)) return;
    if (!up(G.d_over, overflow, sizeof(DevOverflow))) return;
    const DevFormat f = devfmt(bf, false);
    /* bfconf->safety_limit is host state; the host sets it through the fused API.  At op
       level the NaN/Inf test is kept (abort in the reference), the safety test is the
       caller's */
    const double safety = 0.0;
    if (apply_dither && !bf->sf.isfloat) {
        bfhip_dither_state *ds = (bfhip_dither_state *)dither_state;
        if (&dither_randtab == NULL || &dither_randmap == NULL || dither_randtab == NULL) {
            fatal(105, "bfhip: dither requested but the host's dither tables are not linked in");
            return;
        }
        /* dither_preloop_real2int_hp_tpdf, dither.h:28-38 (host-owned integer bookkeeping) */
        if (ds->randtab_ptr + G.L >= dither_randtab_size) {
            dither_randtab[0] = dither_randtab[ds->randtab_ptr - 1];
            ds->randtab_ptr = 1;
        }
        ds->randtab = &dither_randtab[ds->randtab_ptr];
        ds->randtab_ptr += G.L;
        void *d_tab = scratch(2, G.L + 1), *d_map = scratch(3, (size_t)511 * G.rs), *d_fb = scratch(4, 2 * G.rs);
        if (!d_tab || !d_map || !d_fb) return;
        if (!up(d_tab, ds->randtab - 1, G.L + 1)) return;
        if (!up(d_map, (uint8_t *)dither_randmap - (size_t)256 * G.rs, (size_t)511 * G.rs)) return;
        if (!up(d_fb, G.rs == 4 ? (void *)ds->sf : (void *)ds->sd, 2 * G.rs)) return;
        per_type([&](auto t) {
            using T = decltype(t);
            hipLaunchKernelGGL(k_real2raw_dither<T>, dim3(1), dim3(64), 0, G.stream, (const T *)d_real, (uint8_t *)d_raw, f, G.L,
                               (const int8_t *)d_tab, (const T *)d_map + 256, (T *)d_fb, G.d_over, safety, G.d_flag);
        });
        if (!down(G.rs == 4 ? (void *)ds->sf : (void *)ds->sd, d_fb, 2 * G.rs)) return;
    } else {
        per_type([&](auto t) {
            using T = decltype(t);
            hipLaunchKernelGGL(k_real2raw<T>, dim3(1), dim3(256), 0, G.stream, (const T *)d_real, (uint8_t *)d_raw, f, G.L, G.d_over, safety, G.d_flag);
        });
    }
    int flag = 0;
    /* Only THIS channel's samples may reach the host buffer: with several filter processes the
       channels of one interleaved frame are converted by different processes at the same time
       (bfrun.c:1875-2003 behind the second barrier), and writing the whole span back would put a
       neighbour's stale bytes over what the other process has just written -- found by running the
       reference's own multi-process filter_process() over these symbols (tests/test_gpu_refloop.py).
       real2raw writes its own samples only (real2raw.h); so does this. */
    static thread_local std::vector<uint8_t> stage;
    stage.resize(span);
    if (!down(stage.data(), d_raw, span) || !down(overflow, G.d_over, sizeof(DevOverflow)) || !down(&flag, G.d_flag, sizeof(int))) return;
    {
        const size_t sb = (size_t)bf->sf.bytes, step = (size_t)bf->sample_spacing * sb;
        if (step == sb) memcpy(hraw, stage.data(), sb * (size_t)G.L);
        else for (int n = 0; n < G.L; n++) memcpy(hraw + (size_t)n * step, stage.data() + (size_t)n * step, sb);
    }
    if (flag) {
        (void)hipMemsetAsync(G.d_flag, 0, sizeof(int), G.stream);
        fatal(2, "NaN or Inf values in the output! Bad output. Aborting.");   /* real2raw.h:27-30 */
    }
}

void convolver_td_convolve(td_conv_t *tdc, void *overlap_block) {
    if (tdc == NULL || !ensure_device()) return;
    const int size = tdc->blocklen << 1;
    int lg = 0;
    while ((1 << lg) < tdc->blocklen) lg++;
    const size_t bytes = (size_t)size * G.rs;
    if (tdc->d_coeffs == nullptr || tdc->d_pid != G.pid) {
        /* the filter's spectrum was computed on the host by convolver_td_new() (delay.c builds
           its filters in the parent, before the fork): first use in this process uploads it */
        void *d = nullptr;
        if (hipMalloc(&d, bytes) != hipSuccess) { fatal(103, "bfhip: out of device memory"); return; }
        if (!up(d, tdc->h_coeffs, bytes)) return;
        tdc->d_coeffs = d;
        tdc->d_pid = G.pid;
    }
    void *d = scratch(0, bytes);
    if (!d || !up(d, overlap_block, bytes)) return;
    if (!dev_fft(lg, false, d, d)) return;
    per_type([&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL(k_conv_ordered<T>, dim3(grid(size / 2 + 1)), dim3(256), 0, G.stream, (T *)d, (const T *)tdc->d_coeffs, size);
    });
    if (!dev_fft(lg, true, d, d)) return;
    down(overlap_block, d, bytes);
}

}  // extern "C"
