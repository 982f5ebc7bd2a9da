// fft_wave.h -- the block path's complex FFT for L = 1024 .. 8192 (gfx950, wave64): fewer LDS
// round trips than the plain Stockham of fft_lds.h, and the last two radix-8 stages joined by
// WAVE-LEVEL exchanges (DPP row rotate + gfx950 permlane swaps) instead of an LDS pass.
//
// Every thread owns 16 points; NT = L / 16 threads (512 at L = 8192).  Autosort (Stockham)
// passes, natural order in and out:
//
//   P0  radix R0 = L / 512 (2, 4, 8, 16), Ns = 1.  No twiddles.  The forward input transform
//       feeds it straight from the registers the global loads landed in (no LDS fill).
//       Writes s[j*R0 + r].
//   P1  radix 8, Ns = R0: the ordinary LDS pass of fft_lds.h (read, barrier, write, barrier).
//   P2  radix 8, Ns = 8 R0 = L/64.   j = k + Ns q with the lane digit q = lane bits 3..5.
//   P3  radix 8, Ns = 64 R0 = L/8, the last pass (in place: element j + r*L/8 stays where it is).
//
// In the autosort scheme the element that P2's thread (k, q) produces in register r is the one
// P3's thread (k, q' = r) consumes in register r' = q: a TRANSPOSE between the register index
// and a 3-bit lane digit.  The digit sits in lane bits 3..5, where gfx950 has an instruction for
// exactly this exchange: the transpose is three butterfly steps (half of the registers change
// hands with lane ^ 8, ^ 16, ^ 32): ^8 is a DPP row rotate (v_mov_b32 .. row_ror:8, bank masks
// pick the receiving half), ^16 and ^32 are v_permlane16_swap / v_permlane32_swap, which swap two
// registers between lane halves in ONE instruction -- no selects, no LDS, no memory, no barrier.
// P2 -> P3 therefore costs no LDS traffic at all,
// and P3's result is written once, for the consumer that needs another thread's bins
// (the real-transform untangle), or not at all (the inverse transform hands its samples to the
// quantiser from registers).
//
// LDS round trips per transform: 3 (forward) / 2.5 (inverse) instead of 6; barriers 4-5 instead
// of 11.  Same arithmetic as fft_lds.h (dft8 / twiddles w, w^2, w^4 fetched, the rest
// products), so results agree with it to rounding.
#pragma once
#include "fft_lds.h"

namespace bfhip {

// ---- lane exchange -------------------------------------------------------------------------
// One butterfly step of the register <-> lane-digit transpose on a dword pair (a = v[i],
// b = v[i | m]): lanes whose digit bit is 0 receive the partner's a into b, lanes whose bit is 1
// receive the partner's b into a.
template <int LANE_BIT> __device__ __forceinline__ void swap_step(unsigned int &a, unsigned int &b) {
    static_assert(LANE_BIT == 3 || LANE_BIT == 4 || LANE_BIT == 5, "digit lives in lane bits 3..5");
    if constexpr (LANE_BIT == 5) {
        const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);     // a[32..63] <-> b[0..31]
        a = r[0]; b = r[1];
    } else if constexpr (LANE_BIT == 4) {
        const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);     // odd rows of a <-> even rows of b
        a = r[0]; b = r[1];
    } else {
        // row_ror:8 = lane ^ 8 inside a row of 16; bank_mask 0x3 = lanes with bit 3 clear, 0xC = set
        const unsigned int t = b;
        b = (unsigned int)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x128, 0xf, 0x3, false);
        a = (unsigned int)__builtin_amdgcn_update_dpp((int)a, (int)t, 0x128, 0xf, 0xC, false);
    }
}
template <int LANE_BIT> __device__ __forceinline__ void swap_step(float &a, float &b) {
    unsigned int ua = __float_as_uint(a), ub = __float_as_uint(b);
    swap_step<LANE_BIT>(ua, ub);
    a = __uint_as_float(ua); b = __uint_as_float(ub);
}
template <int LANE_BIT> __device__ __forceinline__ void swap_step(double &a, double &b) {
    const unsigned long long qa = (unsigned long long)__double_as_longlong(a), qb = (unsigned long long)__double_as_longlong(b);
    unsigned int alo = (unsigned int)qa, ahi = (unsigned int)(qa >> 32), blo = (unsigned int)qb, bhi = (unsigned int)(qb >> 32);
    swap_step<LANE_BIT>(alo, blo);
    swap_step<LANE_BIT>(ahi, bhi);
    a = __longlong_as_double((long long)(((unsigned long long)ahi << 32) | alo));
    b = __longlong_as_double((long long)(((unsigned long long)bhi << 32) | blo));
}

// v[i] of the lane with digit q  <-  v[q] of the lane with digit i (digit = lane bits 3..5)
template <typename T> __device__ __forceinline__ void octet_transpose(c2<T> (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) if (!(i & 1)) { swap_step<3>(v[i].x, v[i | 1].x); swap_step<3>(v[i].y, v[i | 1].y); }
#pragma unroll
    for (int i = 0; i < 8; i++) if (!(i & 2)) { swap_step<4>(v[i].x, v[i | 2].x); swap_step<4>(v[i].y, v[i | 2].y); }
#pragma unroll
    for (int i = 0; i < 8; i++) if (!(i & 4)) { swap_step<5>(v[i].x, v[i | 4].x); swap_step<5>(v[i].y, v[i | 4].y); }
}

// ---- LDS addressing --------------------------------------------------------------------------
// LdsArr pads one element per 16: element i lives at i + (i >> 4).  For a run of accesses
// i = i0 + r*S with S a multiple of 16 that is i0 + (i0 >> 4) + r * (S + S/16): one address
// computation, the rest are immediate offsets of the ds instruction.
template <typename T> __device__ __forceinline__ c2<T> *lds_at(LdsArr<T> s, int i0) { return s.p + i0 + (i0 >> 4); }
constexpr int lds_stride(int S) { return S + S / 16; }

// ---- radix 16 ------------------------------------------------------------------------------
template <typename T, bool INV> __device__ __forceinline__ void dft16(c2<T> *u) {
    // n = n1 + 4 n2, k = k1 + 4 k2:  X[k1 + 4 k2] = sum_n1 W4^(n1 k2) W16^(n1 k1) sum_n2 x[n1 + 4 n2] W4^(n2 k1)
    const T c1 = (T)0.92387953251128675613, s1 = (T)0.38268343236508977173, h = (T)0.70710678118654752440;
    c2<T> y[4][4];
#pragma unroll
    for (int n1 = 0; n1 < 4; n1++) {
        c2<T> t[4] = {u[n1], u[n1 + 4], u[n1 + 8], u[n1 + 12]};
        dft4<T, INV>(t);
#pragma unroll
        for (int k1 = 0; k1 < 4; k1++) y[n1][k1] = t[k1];
    }
    // W16^m = (cos, -+sin)(pi m / 8), m = n1 * k1
    auto tw = [&](c2<T> a, int m) -> c2<T> {
        T cr, ci;
        switch (m) {
        case 0: return a;
        case 1: cr = c1; ci = s1; break;
        case 2: cr = h; ci = h; break;
        case 3: cr = s1; ci = c1; break;
        case 4: return rot90<T, INV>(a);
        case 6: cr = -h; ci = h; break;
        default: cr = -c1; ci = -s1; break;        // m = 9
        }
        if (INV) ci = -ci;
        // a * (cr - i ci)
        return mk<T>(a.x * cr + a.y * ci, a.y * cr - a.x * ci);
    };
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) {
        c2<T> t[4] = {y[0][k1], tw(y[1][k1], k1), tw(y[2][k1], 2 * k1), tw(y[3][k1], 3 * k1)};
        dft4<T, INV>(t);
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) u[k1 + 4 * k2] = t[k2];
    }
}

template <typename T, bool INV, int R> __device__ __forceinline__ void dftR2(c2<T> *u) {
    if constexpr (R == 16) dft16<T, INV>(u);
    else dftR<T, INV, R>(u);
}

// ---- geometry --------------------------------------------------------------------------------
template <int LOG2L> struct WaveGeo {
    static_assert(LOG2L >= 10 && LOG2L <= 13, "wave FFT covers L = 1024 .. 8192");
    static constexpr int L = 1 << LOG2L, NT = L / 16;
    static constexpr int LOG2R0 = LOG2L - 9, R0 = 1 << LOG2R0, B0 = 16 / R0;
    static constexpr int T8 = L / 8;          // stride of the radix-8 passes
    static constexpr int NS2 = L / 64;        // sub-transform length entering P2
    static constexpr int NTW = 18;            // twiddle registers per thread: 3 passes x 2 butterflies x (w, w^2, w^4)
};
constexpr bool wave_fft_ok(int log2l, int realsize) { return (realsize == 4 || realsize == 8) && log2l >= 10 && log2l <= 13; }

// this thread's butterfly b of the lane-remapped passes P2 / P3: j = k + NS2 * q with the digit
// q = lane bits 3..5 and k = the remaining bits of tid (+ NT/8 for the second butterfly)
template <int LOG2L> __device__ __forceinline__ int wave_j(int tid, int b) {
    using G = WaveGeo<LOG2L>;
    return (tid & 7) + ((tid >> 6) << 3) + (G::NT / 8) * b + G::NS2 * ((tid >> 3) & 7);
}

// Twiddle table: [0, 2L) = exp(-2 pi i m / (2L)) like fft_lds.h (the real-transform untangle reads
// it); behind it the 18 per-thread registers in thread order: entry 2L + q*NT + tid.
inline std::vector<unsigned char> make_wave_twiddle_table(int log2l, int realsize) {
    const int L = 1 << log2l, NT = L / 16, R0 = 1 << (log2l - 9), NS2 = L / 64;
    const size_t total = (size_t)2 * L + (size_t)18 * NT;
    std::vector<unsigned char> out(total * 2 * (size_t)realsize);
    auto put = [&](size_t idx, double turns) {             // exp(-2 pi i turns)
        const double a = -2.0 * M_PI * turns;
        if (realsize == 4) { ((float *)out.data())[2 * idx] = (float)std::cos(a); ((float *)out.data())[2 * idx + 1] = (float)std::sin(a); }
        else { ((double *)out.data())[2 * idx] = std::cos(a); ((double *)out.data())[2 * idx + 1] = std::sin(a); }
    };
    for (int m = 0; m < 2 * L; m++) put((size_t)m, (double)m / (double)(2 * L));
    for (int tid = 0; tid < NT; tid++) {
        for (int b = 0; b < 2; b++) {
            const int j1 = tid + b * NT, k1 = j1 & (R0 - 1);                       // P1: Ns = R0
            const int k2 = (tid & 7) + ((tid >> 6) << 3) + (NT / 8) * b;            // P2: Ns = NS2 (wave_j)
            const int j3 = k2 + NS2 * ((tid >> 3) & 7);                             // P3: Ns = L/8, k = j
            for (int i = 0; i < 3; i++) {
                put((size_t)2 * L + (size_t)(0 + b * 3 + i) * NT + tid, (double)((long)k1 << i) / (double)(R0 * 8));
                put((size_t)2 * L + (size_t)(6 + b * 3 + i) * NT + tid, (double)((long)k2 << i) / (double)(NS2 * 8));
                put((size_t)2 * L + (size_t)(12 + b * 3 + i) * NT + tid, (double)((long)j3 << i) / (double)L);
            }
        }
    }
    return out;
}

template <typename T, int LOG2L> struct WaveTw {
    c2<T> r[18];
    __device__ __forceinline__ void prefetch(const c2<T> *__restrict__ tw) {
        constexpr int L = 1 << LOG2L, NT = L / 16;
#pragma unroll
        for (int q = 0; q < 18; q++) r[q] = tw[2 * L + q * NT + (int)threadIdx.x];
    }
};

// u[1..7] *= w^1..w^7 given w, w^2, w^4 (conjugated for the inverse transform)
template <typename T, bool INV> __device__ __forceinline__ void twiddle8(c2<T> *u, c2<T> w1, c2<T> w2, c2<T> w4) {
    if (INV) { w1.y = -w1.y; w2.y = -w2.y; w4.y = -w4.y; }
    const c2<T> w3 = cmul(w1, w2), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
    u[1] = cmul(u[1], w1); u[2] = cmul(u[2], w2); u[3] = cmul(u[3], w3); u[4] = cmul(u[4], w4);
    u[5] = cmul(u[5], w5); u[6] = cmul(u[6], w6); u[7] = cmul(u[7], w7);
}

// ---- the passes --------------------------------------------------------------------------------

// P0 on values already in registers: z[b][r] = element (tid + b*NT) + r * (L/R0).  Writes LDS.
// The caller synchronises afterwards (wave_p1 starts with LDS reads).
template <typename T, int LOG2L, bool INV>
__device__ __forceinline__ void wave_p0_regs(LdsArr<T> s, c2<T> (&z)[WaveGeo<LOG2L>::B0][WaveGeo<LOG2L>::R0]) {
    using G = WaveGeo<LOG2L>;
    const int tid = threadIdx.x;
#pragma unroll
    for (int b = 0; b < G::B0; b++) {
        dftR2<T, INV, G::R0>(z[b]);
        const int j = tid + b * G::NT;
        // R0 consecutive elements from j*R0 on never straddle a pad (R0 divides 16)
        c2<T> *p = lds_at(s, j * G::R0);
#pragma unroll
        for (int r = 0; r < G::R0; r++) p[r] = z[b][r];
    }
}

// P0 from LDS (the caller has synchronised after filling s)
template <typename T, int LOG2L, bool INV>
__device__ __forceinline__ void wave_p0_lds(LdsArr<T> s) {
    using G = WaveGeo<LOG2L>;
    const int tid = threadIdx.x;
    c2<T> z[G::B0][G::R0];
#pragma unroll
    for (int b = 0; b < G::B0; b++) {
        const c2<T> *p = lds_at(s, tid + b * G::NT);
#pragma unroll
        for (int r = 0; r < G::R0; r++) z[b][r] = p[r * lds_stride(G::L / G::R0)];
    }
    __syncthreads();
    wave_p0_regs<T, LOG2L, INV>(s, z);
}

// P1 (LDS pass), P2, register/lane transpose, P3.  On return x[b][r] = X[wave_j(tid, b) + r * L/8]
// (natural order), nothing of it in LDS yet.  Entry: P0's writes are NOT yet synchronised.
template <typename T, int LOG2L, bool INV>
__device__ __forceinline__ void wave_p123(LdsArr<T> s, const WaveTw<T, LOG2L> &tw, c2<T> (&x)[2][8]) {
    using G = WaveGeo<LOG2L>;
    const int tid = threadIdx.x;
    __syncthreads();
    {   // P1: radix 8, Ns = R0, standard mapping j = tid + b*NT
        c2<T> u[2][8];
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const c2<T> *p = lds_at(s, tid + b * G::NT);
#pragma unroll
            for (int r = 0; r < 8; r++) u[b][r] = p[r * lds_stride(G::T8)];
            twiddle8<T, INV>(u[b], tw.r[b * 3], tw.r[b * 3 + 1], tw.r[b * 3 + 2]);
            dft8<T, INV>(u[b]);
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int j = tid + b * G::NT;
            const int k = j & (G::R0 - 1);
            const int base = (j - k) * 8 + k;
            if constexpr (G::R0 == 16) {
                c2<T> *p = lds_at(s, base);
#pragma unroll
                for (int r = 0; r < 8; r++) p[r * lds_stride(16)] = u[b][r];
            } else {
#pragma unroll
                for (int r = 0; r < 8; r++) s[base + r * G::R0] = u[b][r];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < 2; b++) {
        // P2: radix 8, Ns = NS2, k = j mod NS2 (the twiddles know it)
        const c2<T> *p = lds_at(s, wave_j<LOG2L>(tid, b));
#pragma unroll
        for (int r = 0; r < 8; r++) x[b][r] = p[r * lds_stride(G::T8)];
        twiddle8<T, INV>(x[b], tw.r[6 + b * 3], tw.r[6 + b * 3 + 1], tw.r[6 + b * 3 + 2]);
        dft8<T, INV>(x[b]);
        // what P3's thread (k, q') reads in register r' is what P2's thread (k, q = r') left in
        // register r = q': wave-level exchange, no LDS, no barrier
        octet_transpose<T>(x[b]);
        // P3: radix 8, Ns = L/8, k = j
        twiddle8<T, INV>(x[b], tw.r[12 + b * 3], tw.r[12 + b * 3 + 1], tw.r[12 + b * 3 + 2]);
        dft8<T, INV>(x[b]);
    }
}

// the spectrum back into s (natural order) for consumers that need other threads' bins;
// synchronises first (P2's reads of other waves) and afterwards
template <typename T, int LOG2L>
__device__ __forceinline__ void wave_store(LdsArr<T> s, const c2<T> (&x)[2][8]) {
    using G = WaveGeo<LOG2L>;
    const int tid = threadIdx.x;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; b++) {
        c2<T> *p = lds_at(s, wave_j<LOG2L>(tid, b));
#pragma unroll
        for (int r = 0; r < 8; r++) p[r * lds_stride(G::T8)] = x[b][r];
    }
    __syncthreads();
}

}  // namespace bfhip
