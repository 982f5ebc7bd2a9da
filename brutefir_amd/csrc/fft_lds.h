// fft_lds.h -- in-LDS Stockham FFT for one workgroup (gfx950, wave64).
//
// A complex FFT of L = 2^LOG2L points lives entirely in LDS (L <= 8192 complex:
// 68 KiB f32 / 136 KiB f64 of the CU's 160 KiB, padding included) and is transformed in
// place by radix-8 passes (plus one radix-4 or radix-2 tail pass).  Every pass is the
// autosort form
//     u[r]  = s[j + r*T] * w^(r*k),   T = L/R, k = j mod Ns, w = exp(-+2 pi i/(Ns*R))
//     v     = DFT_R(u)
//     s[(j-k)*R + k + r*Ns] = v[r]
// so the result is in natural order with no bit reversal.  All data of a pass is
// held in registers across the read -> barrier -> write hand-over, which is what
// makes the in-place update legal; butterfly counts per thread are compile-time
// so nothing is runtime-indexed (no scratch).
//
// Latency matters more than throughput here (one workgroup per channel, and on 8 GPUs
// only 8 workgroups per launch), so:
//  * the twiddles a thread needs for ALL passes depend only on its thread id: they are
//    fetched into registers up front (TwRegs::prefetch) while the kernel is still loading
//    its input, instead of one dependent L2/HBM round trip per pass; per radix-8 butterfly
//    only w, w^2, w^4 are fetched and w^3, w^5, w^6, w^7 are products;
//  * the LDS array is padded by one element per 16 (LdsArr) so that the stride-R stores of
//    the first pass do not all hit the same banks.
//
// The real transforms the convolver needs (FFTW R2HC / HC2R of 2L reals,
// fftw_convolver.c:113-119) are built on it with the usual even/odd packing.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

// phase markers for tools/fft_probe.hip (cycle stamps of one wave); nothing in the product build
#ifndef BF_PROBE
#define BF_PROBE(i)
#endif

namespace bfhip {

template <typename T> struct alignas(2 * sizeof(T)) c2 { T x, y; };

template <typename T> __device__ __forceinline__ c2<T> mk(T x, T y) { c2<T> r; r.x = x; r.y = y; return r; }
template <typename T> __device__ __forceinline__ c2<T> operator+(c2<T> a, c2<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T> __device__ __forceinline__ c2<T> operator-(c2<T> a, c2<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <typename T> __device__ __forceinline__ c2<T> cmul(c2<T> a, c2<T> b) { return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <typename T> __device__ __forceinline__ c2<T> conj(c2<T> a) { return mk<T>(a.x, -a.y); }
// multiply by -i (forward) / +i (inverse)
template <typename T, bool INV> __device__ __forceinline__ c2<T> rot90(c2<T> a) { return INV ? mk<T>(-a.y, a.x) : mk<T>(a.y, -a.x); }

// LDS array with one pad element per 16: element i lives at i + (i >> 4)
template <typename T> struct LdsArr {
    c2<T> *p;
    __device__ __forceinline__ c2<T> &operator[](int i) const { return p[i + (i >> 4)]; }
};
constexpr size_t lds_fft_bytes(int log2l, size_t elem) { return (((size_t)1 << log2l) + (((size_t)1 << log2l) >> 4) + 1) * elem; }

template <typename T, bool INV> __device__ __forceinline__ void dft2(c2<T> *u) {
    c2<T> a = u[0] + u[1], b = u[0] - u[1];
    u[0] = a; u[1] = b;
}

template <typename T, bool INV> __device__ __forceinline__ void dft4(c2<T> *u) {
    c2<T> b0 = u[0] + u[2], b2 = u[0] - u[2];
    c2<T> b1 = u[1] + u[3], b3 = rot90<T, INV>(u[1] - u[3]);
    u[0] = b0 + b1; u[2] = b0 - b1; u[1] = b2 + b3; u[3] = b2 - b3;
}

template <typename T, bool INV> __device__ __forceinline__ void dft8(c2<T> *u) {
    const T h = (T)0.70710678118654752440;
    c2<T> a0 = u[0] + u[4], a4 = u[0] - u[4];
    c2<T> a1 = u[1] + u[5], a5 = u[1] - u[5];
    c2<T> a2 = u[2] + u[6], a6 = u[2] - u[6];
    c2<T> a3 = u[3] + u[7], a7 = u[3] - u[7];
    // a5 *= w8, a6 *= w8^2, a7 *= w8^3   (w8 = exp(-+i pi/4))
    a5 = INV ? mk<T>(h * (a5.x - a5.y), h * (a5.x + a5.y)) : mk<T>(h * (a5.x + a5.y), h * (a5.y - a5.x));
    a6 = rot90<T, INV>(a6);
    a7 = INV ? mk<T>(-h * (a7.x + a7.y), h * (a7.x - a7.y)) : mk<T>(h * (a7.y - a7.x), -h * (a7.x + a7.y));
    c2<T> b0 = a0 + a2, b2 = a0 - a2, b1 = a1 + a3, b3 = rot90<T, INV>(a1 - a3);
    c2<T> b4 = a4 + a6, b6 = a4 - a6, b5 = a5 + a7, b7 = rot90<T, INV>(a5 - a7);
    u[0] = b0 + b1; u[4] = b0 - b1; u[2] = b2 + b3; u[6] = b2 - b3;
    u[1] = b4 + b5; u[5] = b4 - b5; u[3] = b6 + b7; u[7] = b6 - b7;
}

template <typename T, bool INV, int R> __device__ __forceinline__ void dftR(c2<T> *u) {
    if constexpr (R == 8) dft8<T, INV>(u);
    else if constexpr (R == 4) dft4<T, INV>(u);
    else dft2<T, INV>(u);
}

// ---- pass geometry, shared by the prefetch and the passes themselves
constexpr int pass_log2r(int log2l, int log2ns) { return (log2l - log2ns >= 3) ? 3 : (log2l - log2ns); }
constexpr int pass_b(int log2l, int nt, int log2ns) {       // butterflies per thread
    return (((1 << log2l) >> pass_log2r(log2l, log2ns)) + nt - 1) / nt;
}
constexpr int pass_ntw(int log2l, int log2ns) {             // fetched twiddles per butterfly
    return log2ns == 0 ? 0 : pass_log2r(log2l, log2ns);    // w, w^2, w^4 -> log2(R) of them
}
constexpr int tw_total(int log2l, int nt, int log2ns) {
    return log2ns >= log2l ? 0
        : pass_b(log2l, nt, log2ns) * pass_ntw(log2l, log2ns) + tw_total(log2l, nt, log2ns + pass_log2r(log2l, log2ns));
}

// The twiddles of every pass for this thread, in registers.
// Twiddle table in global memory (make_twiddle_table below): entries [0, 2L) are
// exp(-2 pi i m / (2L)); behind them the same values again in THREAD ORDER -- entry
// 2L + q*NT + tid is register q of thread tid -- so that a wave fetches each register with one
// contiguous 512-byte load instead of a 64-line gather (the gathers of 16 waves kept the CU's
// texture path busy for ~3 us of a 17 us transform at L = 8192).
// LAZY (more than 64 VGPRs of twiddles, e.g. float64 at L = 8192: 32 of them = 128 VGPRs, which
// made the kernels spill; or a narrow workgroup with many butterflies per thread): nothing is
// held in registers, every pass reads its twiddles from the thread-ordered table when it needs
// them -- still one coalesced load per value.
template <typename T, int LOG2L, int NT, bool FORCE_LAZY = false> struct TwRegs {
    // FORCE_LAZY: kernels that run several transforms and keep other state across them
    static constexpr bool LAZY = FORCE_LAZY || (size_t)tw_total(LOG2L, NT, 0) * sizeof(c2<T>) > 256;     // > 64 VGPRs
    static constexpr int N = (LAZY || tw_total(LOG2L, NT, 0) == 0) ? 1 : tw_total(LOG2L, NT, 0);
    c2<T> r[N];
    const c2<T> *g;                 // thread-ordered table, this thread's column

    __device__ __forceinline__ c2<T> get(int q) const { return LAZY ? g[q * NT] : r[q]; }

    template <int LOG2NS, int OFF>
    __device__ __forceinline__ void fetch_from(const c2<T> *__restrict__ tw) {
        if constexpr (LOG2NS < LOG2L) {
            constexpr int LOG2R = pass_log2r(LOG2L, LOG2NS);
            constexpr int L = 1 << LOG2L, R = 1 << LOG2R, Ns = 1 << LOG2NS, TT = L / R;
            constexpr int B = pass_b(LOG2L, NT, LOG2NS), NTW = pass_ntw(LOG2L, LOG2NS);
            constexpr int TWSTEP = (2 * L) / (Ns * R);
            if constexpr (NTW > 0) {
#pragma unroll
                for (int b = 0; b < B; b++) {
#pragma unroll
                    for (int i = 0; i < NTW; i++) r[OFF + b * NTW + i] = tw[2 * L + (OFF + b * NTW + i) * NT + (int)threadIdx.x];
                }
                (void)TT; (void)Ns; (void)TWSTEP;
            }
            fetch_from<LOG2NS + LOG2R, OFF + B * NTW>(tw);
        }
    }
    __device__ __forceinline__ void prefetch(const c2<T> *__restrict__ tw) {
        g = tw + 2 * (1 << LOG2L) + (int)threadIdx.x;
        if constexpr (!LAZY) fetch_from<0, 0>(tw);
    }
};

// one Stockham pass on the padded LDS array with prefetched twiddles
template <typename T, int LOG2L, int NT, bool INV, int LOG2NS, int OFF, typename TW>
__device__ __forceinline__ void fft_pass(LdsArr<T> s, const TW &tw) {
    constexpr int LOG2R = pass_log2r(LOG2L, LOG2NS);
    constexpr int L = 1 << LOG2L, R = 1 << LOG2R, Ns = 1 << LOG2NS, TT = L / R;
    constexpr int B = pass_b(LOG2L, NT, LOG2NS), NTW = pass_ntw(LOG2L, LOG2NS);
    c2<T> u[B][R];
    const int tid = threadIdx.x;
#pragma unroll
    for (int b = 0; b < B; b++) {
        const int j = tid + b * NT;
        if (TT % NT == 0 || j < TT) {
#pragma unroll
            for (int r = 0; r < R; r++) u[b][r] = s[j + r * TT];
            if constexpr (NTW > 0) {
                c2<T> w[R];
                w[1] = tw.get(OFF + b * NTW);
                if (INV) w[1].y = -w[1].y;
                if constexpr (R >= 4) {
                    w[2] = tw.get(OFF + b * NTW + 1);
                    if (INV) w[2].y = -w[2].y;
                    w[3] = cmul(w[1], w[2]);
                }
                if constexpr (R == 8) {
                    w[4] = tw.get(OFF + b * NTW + 2);
                    if (INV) w[4].y = -w[4].y;
                    w[5] = cmul(w[4], w[1]); w[6] = cmul(w[4], w[2]); w[7] = cmul(w[4], w[3]);
                }
#pragma unroll
                for (int r = 1; r < R; r++) u[b][r] = cmul(u[b][r], w[r]);
            }
            dftR<T, INV, R>(u[b]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < B; b++) {
        const int j = tid + b * NT;
        if (TT % NT == 0 || j < TT) {
            const int k = j & (Ns - 1);
            const int base = (j - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; r++) s[base + r * Ns] = u[b][r];
        }
    }
    __syncthreads();
}

template <typename T, int LOG2L, int NT, bool INV, int LOG2NS, int OFF, typename TW>
__device__ __forceinline__ void fft_passes(LdsArr<T> s, const TW &tw) {
    if constexpr (LOG2NS < LOG2L) {
        constexpr int LOG2R = pass_log2r(LOG2L, LOG2NS);
        fft_pass<T, LOG2L, NT, INV, LOG2NS, OFF>(s, tw);
        BF_PROBE(4 + LOG2NS / 3);
        fft_passes<T, LOG2L, NT, INV, LOG2NS + LOG2R, OFF + pass_b(LOG2L, NT, LOG2NS) * pass_ntw(LOG2L, LOG2NS)>(s, tw);
    }
}

// Complex FFT of the L values in s (padded LDS array), in place, natural order in and out.
// The caller has synchronised after filling s; on return all threads see the result.
template <typename T, int LOG2L, int NT, bool INV, typename TW>
__device__ __forceinline__ void lds_fft(LdsArr<T> s, const TW &tw) {
    fft_passes<T, LOG2L, NT, INV, 0, 0>(s, tw);
}

// Host: the table TwRegs::prefetch and the real-transform (un)tangling read, for the NT the
// kernels of this precision use.  Values are computed in double and rounded once.
inline std::vector<unsigned char> make_twiddle_table(int log2l, int realsize, int nt) {
    const size_t L = (size_t)1 << log2l;
    // lengths above the LDS limit (bigfft.h) only use the base table
    const size_t n_regs = log2l > 13 ? 0 : (size_t)tw_total(log2l, nt, 0);
    const size_t total = 2 * L + n_regs * (size_t)nt;
    std::vector<unsigned char> out(total * 2 * (size_t)realsize);
    auto put = [&](size_t idx, size_t m) {
        const double a = -M_PI * (double)m / (double)L;
        if (realsize == 4) { ((float *)out.data())[2 * idx] = (float)std::cos(a); ((float *)out.data())[2 * idx + 1] = (float)std::sin(a); }
        else { ((double *)out.data())[2 * idx] = std::cos(a); ((double *)out.data())[2 * idx + 1] = std::sin(a); }
    };
    for (size_t m = 0; m < 2 * L; m++) put(m, m);
    size_t q = 0;
    for (int log2ns = 0; n_regs > 0 && log2ns < log2l; log2ns += pass_log2r(log2l, log2ns)) {
        const int log2r = pass_log2r(log2l, log2ns);
        const int R = 1 << log2r, Ns = 1 << log2ns, TT = (int)L / R;
        const int B = pass_b(log2l, nt, log2ns), NTW = pass_ntw(log2l, log2ns);
        const int twstep = (int)(2 * L) / (Ns * R);
        for (int b = 0; b < B; b++)
            for (int i = 0; i < NTW; i++)
                for (int tid = 0; tid < nt; tid++) {
                    const int j = tid + b * nt;
                    const int k = (TT % nt == 0 || j < TT) ? (j & (Ns - 1)) : 0;
                    put(2 * L + (q + (size_t)b * NTW + i) * (size_t)nt + tid, (size_t)(k << i) * twstep);
                }
        q += (size_t)B * NTW;
    }
    return out;
}

// threads per workgroup used for a transform of 2^LOG2L complex points: one radix-8 butterfly
// per thread, at most 1024 threads (float) / 512 (double: twice the registers per value, and
// 1024 threads would cap the kernel at 128 VGPRs)
template <typename T> constexpr int fft_threads(int log2l) {
    return (1 << log2l) / 8 < 64 ? 64 : ((1 << log2l) / 8 > (sizeof(T) == 8 ? 512 : 1024) ? (sizeof(T) == 8 ? 512 : 1024) : (1 << log2l) / 8);
}

}  // namespace bfhip
