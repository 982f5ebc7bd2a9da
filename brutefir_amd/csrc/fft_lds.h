// fft_lds.h -- in-LDS Stockham FFT for one workgroup (gfx950, wave64).
//
// A complex FFT of L = 2^LOG2L points lives entirely in LDS (L <= 8192 complex:
// 64 KiB f32 / 128 KiB f64 of the CU's 160 KiB) and is transformed in place by
// radix-8 passes (plus one radix-4 or radix-2 tail pass).  Every pass is the
// autosort form
//     u[r]  = s[j + r*T] * w^(r*k),   T = L/R, k = j mod Ns, w = exp(-+2 pi i/(Ns*R))
//     v     = DFT_R(u)
//     s[(j-k)*R + k + r*Ns] = v[r]
// so the result is in natural order with no bit reversal.  All data of a pass is
// held in registers across the read -> barrier -> write hand-over, which is what
// makes the in-place update legal; butterfly counts per thread are compile-time
// so nothing is runtime-indexed (no scratch).
//
// The real transforms the convolver needs (FFTW R2HC / HC2R of 2L reals,
// fftw_convolver.c:113-119) are built on it with the usual even/odd packing; see
// untangle_forward / tangle_inverse in kernels.h.
#pragma once
#include <hip/hip_runtime.h>

namespace bfhip {

template <typename T> struct alignas(2 * sizeof(T)) c2 { T x, y; };

template <typename T> __device__ __forceinline__ c2<T> mk(T x, T y) { c2<T> r; r.x = x; r.y = y; return r; }
template <typename T> __device__ __forceinline__ c2<T> operator+(c2<T> a, c2<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T> __device__ __forceinline__ c2<T> operator-(c2<T> a, c2<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <typename T> __device__ __forceinline__ c2<T> cmul(c2<T> a, c2<T> b) { return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <typename T> __device__ __forceinline__ c2<T> conj(c2<T> a) { return mk<T>(a.x, -a.y); }
// multiply by -i (forward) / +i (inverse)
template <typename T, bool INV> __device__ __forceinline__ c2<T> rot90(c2<T> a) { return INV ? mk<T>(-a.y, a.x) : mk<T>(a.y, -a.x); }

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

// one Stockham pass; tw = exp(-2 pi i m / (2L)), m in [0, 2L), in global memory
template <typename T, int LOG2L, int NT, bool INV, int LOG2NS, int LOG2R>
__device__ __forceinline__ void fft_pass(c2<T> *s, const c2<T> *__restrict__ tw) {
    constexpr int L = 1 << LOG2L, R = 1 << LOG2R, Ns = 1 << LOG2NS, TT = L / R;
    constexpr int B = (TT + NT - 1) / NT;
    constexpr int TWSTEP = (2 * L) / (Ns * R);
    c2<T> u[B][R];
    const int tid = threadIdx.x;
#pragma unroll
    for (int b = 0; b < B; b++) {
        const int j = tid + b * NT;
        if (TT % NT == 0 || j < TT) {
#pragma unroll
            for (int r = 0; r < R; r++) u[b][r] = s[j + r * TT];
            if constexpr (LOG2NS > 0) {
                const int k = j & (Ns - 1);
#pragma unroll
                for (int r = 1; r < R; r++) {
                    c2<T> w = tw[r * k * TWSTEP];
                    if (INV) w.y = -w.y;
                    u[b][r] = cmul(u[b][r], w);
                }
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

template <typename T, int LOG2L, int NT, bool INV, int LOG2NS>
__device__ __forceinline__ void fft_passes(c2<T> *s, const c2<T> *__restrict__ tw) {
    constexpr int REM = LOG2L - LOG2NS;
    if constexpr (REM >= 3) {
        fft_pass<T, LOG2L, NT, INV, LOG2NS, 3>(s, tw);
        fft_passes<T, LOG2L, NT, INV, LOG2NS + 3>(s, tw);
    } else if constexpr (REM == 2) {
        fft_pass<T, LOG2L, NT, INV, LOG2NS, 2>(s, tw);
    } else if constexpr (REM == 1) {
        fft_pass<T, LOG2L, NT, INV, LOG2NS, 1>(s, tw);
    }
}

// Complex FFT of the L values in s (LDS), in place, natural order in and out.
// The caller has synchronised after filling s; on return all threads see the result.
template <typename T, int LOG2L, int NT, bool INV>
__device__ __forceinline__ void lds_fft(c2<T> *s, const c2<T> *__restrict__ tw) {
    fft_passes<T, LOG2L, NT, INV, 0>(s, tw);
}

// threads per workgroup used for a transform of 2^LOG2L complex points
constexpr int fft_threads(int log2l) { return (1 << log2l) / 8 < 64 ? 64 : ((1 << log2l) / 8 > 1024 ? 1024 : (1 << log2l) / 8); }

}  // namespace bfhip
