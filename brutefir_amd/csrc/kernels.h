// kernels.h -- the device side of the filter path (gfx950 only).
//
// Spectrum layout in HBM ("packed spectrum"): L complex numbers per 2L-point real
// transform; element k (1 <= k < L) is bin X[k]; element 0 carries the two purely real
// bins as (Re X[0], Re X[L]).  This is the device-internal counterpart of the reference's
// "4 re / 4 im" cbuf layout (fftw_convfuns.h:25-43) -- same information, bins contiguous
// so that one lane's 16-byte access is two whole bins and a wave reads 1 KiB contiguous.
//
//   K1 fft_in_kernel     raw2cbuf + time2freq     fftw_convolver.c:170-214, raw2real.h
//   K2 mac_xbar_kernel   mixnscale(INPUT) + convolve + convolve_add* + dirac +
//                        mixnscale(OUTPUT)        fftw_convfuns.h:7-619, bfrun.c:1566-1868
//   K3 ifft_out_kernel   freq2time + cbuf2raw     fftw_convolver.c:391-409, 482-518,
//                                                 real2raw.h, dither_funs.h:71-114
//   K7 coeff_prep_kernel coeffs2cbuf              fftw_convolver.c:526-573
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_lds.h"
#include "fft_wave.h"

namespace bfhip {

// ------------------------------------------------------------------ shared structs

struct DevFormat {          // bfhip_format, device copy
    int isfloat, swap, bytes, sbytes;
    int sample_spacing, byte_offset;
    const uint8_t *alt;     // input side: read from here instead of the raw buffer (a virtual
                            // channel's private, delayed copy: N:1 mapped inputs), else null
};

struct DevOverflow {        // struct bfoverflow (bfmod.h:99-104)
    unsigned int n_overflows;
    int32_t intlargest;
    double largest;
    double max;
};

constexpr int OG = 8;       // outputs accumulated per MAC workgroup

enum { TERM_NONE = -1, TERM_COEFF = 0, TERM_DIRAC = 1, TERM_IDENT = 2 };

template <typename T> struct MacTerm {
    const c2<T> *H;     // coefficient set: [n_blocks][L] packed spectra (already / n_fft)
    T scale;            // input scale * sf.scale * output scale / sf_out.scale
    int P;              // partitions to accumulate (cblocks, bfrun.c:1585-1591)
    int kind;           // TERM_*
};

template <typename T> struct MacEntry {
    const c2<T> *ring;  // [R][L] packed spectra of past blocks
    int R;              // ring depth
    int delay;          // filter delay in blocks (bfrun.c:1579-1584)
    int p0;             // first partition this entry covers (entries may be split along p)
    int maxP;           // one past the last partition: max over terms of P, or the split point
    int dense;          // 1: all OG terms are coefficient terms with P >= maxP (crossbar path);
                        // 16: the same for the subset `mask` of the terms, the rest empty;
                        // 2 + j: term j is the only one; 0: generic per-term path
    int mask;           // dense == 16: bit j set = term j is active
    const int *live;    // powersave: how many of the ring's slots hold a non-silent block (null: n/a)
    MacTerm<T> term[OG];
};

struct ChunkRange { int begin, end; };

// Stream-ordered copy of the coefficients of a uniform crossbar plan (every entry: OG coefficient
// terms of the same length P, every chunk the same number of entries).  Workgroup `bid` of the MAC
// grid owns the contiguous slice [bid * slice, (bid + 1) * slice): inside it, for each of its
// entries q and partitions p, the OG tiles it multiplies with (one per output of its group), one
// after the other:   base + bid*slice + ((q*P + p)*OG + j) * chunk + lane*16,   chunk = 256 * 16 B.
// Every workgroup then reads ONE sequential stream front to back -- the access shape of a plain
// read benchmark -- whatever order the host loaded its coefficient sets in and wherever their
// allocations ended up (both move the set-major layout between 0.59 and 0.89 of peak).
struct StreamLayout {
    const unsigned char *base;      // null: coefficients are read set by set (MacTerm::H)
    unsigned long long slice;       // bytes per workgroup
    unsigned int entry_bytes;       // P * OG * chunk
    unsigned int chunk;             // bytes of one tile of one set: blockDim.x * 16
};

// Per-block counters in device memory, for launch sequences replayed from a HIP graph (their
// kernel arguments are frozen at capture time): t = blockcounter (bfrun.c:2034), age =
// min(blocks processed + 1, N) (the procblocks guard, bfrun.c:1745).  Kernels take an optional
// pointer to it; null = use the by-value arguments.  rt_tail_kernel advances it.
// The counter does not run up to 2^32 like the reference's: slots are `t mod ring depth`, and a
// depth that is not a power of two (N = 13, or the spare slot of the pipelined schedules, N + 1)
// would jump at the wrap -- after 66 days of 64-frame blocks.  It wraps from wrap_at by wrap_by, a
// multiple of every ring depth in the engine, instead (wrap_by = 0: plain unsigned wrap).
struct BlockState { unsigned int t; int age; unsigned int n_blocks; unsigned int wrap_at, wrap_by; int pad; };

// ------------------------------------------------------------------ raw sample access

template <typename T>
__device__ __forceinline__ T load_raw(const uint8_t *p, const DevFormat &f) {
    // word-sized fast paths for naturally aligned samples (the common S16/S24_4/S32/FLOAT
    // layouts); everything else is assembled byte by byte
    uint32_t lo = 0, hi = 0;
    const uintptr_t a = (uintptr_t)p;
    if (f.bytes == 4 && (a & 3) == 0) {
        lo = *reinterpret_cast<const uint32_t *>(p);
        if (f.swap) lo = __builtin_bswap32(lo);
    } else if (f.bytes == 2 && (a & 1) == 0) {
        lo = *reinterpret_cast<const uint16_t *>(p);
        if (f.swap) lo = __builtin_bswap16((uint16_t)lo);
    } else if (f.bytes == 8 && (a & 7) == 0) {
        uint64_t q = *reinterpret_cast<const uint64_t *>(p);
        if (f.swap) q = __builtin_bswap64(q);
        lo = (uint32_t)q; hi = (uint32_t)(q >> 32);
    } else {
        uint8_t t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = 0;
        if (f.swap && f.bytes != 3) {
            for (int i = 0; i < f.bytes; i++) t[i] = p[f.bytes - 1 - i];
        } else {
            for (int i = 0; i < f.bytes; i++) t[i] = p[i];
        }
        if (f.bytes == 3) {
            // packed 24 bit: into the top of an int32, arithmetic shift down (raw2real.h:106-142)
            const uint32_t u = f.swap ? ((uint32_t)t[2] << 8 | (uint32_t)t[1] << 16 | (uint32_t)t[0] << 24)
                                      : ((uint32_t)t[0] << 8 | (uint32_t)t[1] << 16 | (uint32_t)t[2] << 24);
            return (T)((int32_t)u >> 8);
        }
        lo = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
        hi = (uint32_t)t[4] | (uint32_t)t[5] << 8 | (uint32_t)t[6] << 16 | (uint32_t)t[7] << 24;
    }
    if (f.isfloat) {
        if (f.bytes == 4) return (T)__uint_as_float(lo);
        return (T)__longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
    }
    switch (f.bytes) {
    case 1: return (T)(int8_t)(lo & 0xff);
    case 2: return (T)(int16_t)(lo & 0xffff);
    default: return (T)(int32_t)lo;
    }
}

// A naturally aligned sample already loaded as one word (what load_raw's fast paths yield).
template <typename T> __device__ __forceinline__ T word_to_real(uint32_t w, const DevFormat &f) {
    if (f.swap) w = __builtin_bswap32(w);
    return f.isfloat ? (T)__uint_as_float(w) : (T)(int32_t)w;
}
template <typename T> __device__ __forceinline__ T word_to_real(uint16_t w, const DevFormat &f) {
    if (f.swap) w = __builtin_bswap16(w);
    return (T)(int16_t)w;
}
template <typename T> __device__ __forceinline__ T word_to_real(uint64_t w, const DevFormat &f) {
    if (f.swap) w = __builtin_bswap64(w);
    return f.isfloat ? (T)__longlong_as_double((long long)w) : (T)(int32_t)(uint32_t)w;
}

// Sample pairs (2n, 2n+1), n = tid + i*NT, of one channel as words: ALL loads are issued before
// the first conversion waits on one.  (load_raw's format branches would otherwise put a full
// memory round trip between consecutive loads: 6.5 of the kernel's 17.6 us at L = 8192.)
template <typename T, typename W, int QP, int NT, int HALF>
__device__ __forceinline__ void load_pairs_words(const uint8_t *__restrict__ base, size_t stride, const DevFormat &f,
                                                 int tid, c2<T> (&cur)[QP]) {
    W w0[QP], w1[QP];
#pragma unroll
    for (int i = 0; i < QP; i++) {
        const int n = tid + i * NT;
        w0[i] = 0; w1[i] = 0;
        if (HALF % NT == 0 || n < HALF) {
            // 32-bit byte offsets from the (uniform) channel base: scalar base + one VGPR offset per
            // load instead of 64-bit multiplies (a block's raw extent is far below 4 GiB)
            const uint32_t o = (uint32_t)(2 * n) * (uint32_t)stride;
            w0[i] = *reinterpret_cast<const W *>(base + o);
            w1[i] = *reinterpret_cast<const W *>(base + (o + (uint32_t)stride));
        }
    }
#pragma unroll
    for (int i = 0; i < QP; i++) cur[i] = mk<T>(word_to_real<T>(w0[i], f), word_to_real<T>(w1[i], f));
}

// w holds the sample's bytes, little end first
__device__ __forceinline__ void store_raw_word(uint8_t *p, uint64_t w, int bytes, int swap) {
    const uintptr_t a = (uintptr_t)p;
    if (bytes == 4 && (a & 3) == 0) {
        uint32_t u = (uint32_t)w;
        if (swap) u = __builtin_bswap32(u);
        *reinterpret_cast<uint32_t *>(p) = u;
    } else if (bytes == 2 && (a & 1) == 0) {
        uint16_t u = (uint16_t)w;
        if (swap) u = __builtin_bswap16(u);
        *reinterpret_cast<uint16_t *>(p) = u;
    } else {
        if (swap) w = __builtin_bswap64(w) >> (8 * (8 - bytes));
        for (int i = 0; i < bytes; i++) p[i] = (uint8_t)(w >> (8 * i));
    }
}

// ------------------------------------------------------------------ K1: raw -> spectrum

// helpers shared by the kernels that run the in-LDS real FFTs --------------------------

// bins this thread untangles: k = 1 + tid + i*NT (i < QU) covers 1..L/2; k = 0 is thread 0's
template <typename T, int LOG2L, int NTP = fft_threads<T>(LOG2L)> struct UT {
    static constexpr int L = 1 << LOG2L, NT = NTP;
    static constexpr int QU = (L / 2 + NT - 1) / NT;     // bins (pairs k, L-k) per thread
    static constexpr int QP = (L / 2 + NT - 1) / NT;     // sample pairs per thread and half
};

// forward untangle of bin pair (k, L-k): X[k] = E + w^k O, X[L-k] = conj(E - w^k O)
template <typename T>
__device__ __forceinline__ void untangle(c2<T> a, c2<T> bconj, c2<T> w, c2<T> &xk, c2<T> &xlk) {
    const c2<T> e = mk<T>((T)0.5 * (a.x + bconj.x), (T)0.5 * (a.y + bconj.y));
    const c2<T> d = mk<T>((T)0.5 * (a.x - bconj.x), (T)0.5 * (a.y - bconj.y));
    const c2<T> wo = cmul(mk<T>(d.y, -d.x), w);           // (d / i) * w^k
    xk = e + wo;
    xlk = conj(e - wo);
}

// inverse pre-pass of bin pair: Z'[k] = 2E + i 2O, Z'[L-k] = conj(2E) + i conj(2O)
template <typename T>
__device__ __forceinline__ void tangle(c2<T> a, c2<T> bconj, c2<T> w, c2<T> &zk, c2<T> &zlk) {
    const c2<T> e = a + bconj, d = a - bconj;
    const c2<T> o = cmul(d, conj(w));
    zk = mk<T>(e.x - o.y, e.y + o.x);
    zlk = mk<T>(e.x + o.y, -e.y + o.x);
}

// `powersave:` on the device.  thr = 0: off.  thr >= 1: a 2L window whose reals are all zero bits
// is silent (memiszero, bfrun.c:696-719); thr < 1: a window with scale * max|x| < thr is silent
// and counts as zero (test_silent, :721-771).  A silent window's spectrum is zero (:1541-1553);
// flags[ch * R + slot] remembers it and live[ch] counts the non-silent slots of the ring so that
// the MAC can skip inputs that have been silent for a whole filter length.
struct PowerSave {
    double thr;
    const double *scale;    // [n_in] sf.scale of the input's sample format
    int *flags;             // [n_in][R], 1 = silent
    int *live;              // [n_in]
};

// Which channel a transform workgroup takes.  Workgroups b, b + 8, b + 16 ... are dispatched to the same
// XCD and share its L2; channels that are neighbours in an interleaved frame share cache lines (32
// four-byte samples per 128 bytes).  Every XCD therefore gets ONE contiguous run of channels: a line
// of the raw buffer is fetched into one L2 instead of eight, and the 4-byte pieces K3 scatters into a
// line meet in one L2, which writes whole sectors back instead of eight masked fragments.
__device__ __forceinline__ int xcd_channel(int b, int n) {
    return (n & 7) ? b : (b & 7) * (n >> 3) + (b >> 3);
}

// One workgroup per input channel.  Window = [previous L samples | new L samples]
// (fftw_convolver.c:181-193); z[n] = x[2n] + i x[2n+1]; complex FFT; untangle; write the
// packed spectrum into ring slot `slot` of that channel.  All global loads (twiddles,
// previous block, raw samples) are issued before anything waits on them.
template <typename T, int LOG2L, int NTP = fft_threads<T>(LOG2L)>
__device__ __forceinline__ void
fft_in_body(int ch, unsigned char *smem, const uint8_t *__restrict__ raw, const DevFormat *__restrict__ fmt,
            T *__restrict__ prev,            // [n_in][L] last block's samples
            c2<T> *__restrict__ ring,        // [n_in][R][L]
            const c2<T> *__restrict__ tw, int R, int slot, PowerSave ps) {
    constexpr int L = 1 << LOG2L, NT = NTP;
    constexpr int QP = UT<T, LOG2L, NT>::QP, QU = UT<T, LOG2L, NT>::QU;
    LdsArr<T> s{reinterpret_cast<c2<T> *>(smem)};
    const int tid = threadIdx.x;
    const DevFormat f = fmt[ch];
    c2<T> *pv = reinterpret_cast<c2<T> *>(prev + (size_t)ch * L);
    const uint8_t *base = f.alt ? f.alt : raw + f.byte_offset;
    const size_t stride = (size_t)f.sample_spacing * f.bytes;

    BF_PROBE(0);
    // samples first, twiddles second: loads return in order, so the LDS fill and the first pass
    // (which needs no twiddles) only wait for the samples while the twiddles are still in flight
    c2<T> old[QP], cur[QP];
#pragma unroll
    for (int i = 0; i < QP; i++) {
        const int n = tid + i * NT;
        if (L / 2 % NT == 0 || n < L / 2) old[i] = pv[n];
    }
    // the channel's format is uniform over the workgroup: pick the load width once
    const uintptr_t ba = (uintptr_t)base;
    if (f.bytes == 4 && ((ba | stride) & 3) == 0) {
        load_pairs_words<T, uint32_t, QP, NT, L / 2>(base, stride, f, tid, cur);
    } else if (f.bytes == 2 && ((ba | stride) & 1) == 0) {
        load_pairs_words<T, uint16_t, QP, NT, L / 2>(base, stride, f, tid, cur);
    } else if (f.bytes == 8 && ((ba | stride) & 7) == 0) {
        load_pairs_words<T, uint64_t, QP, NT, L / 2>(base, stride, f, tid, cur);
    } else {
#pragma unroll
        for (int i = 0; i < QP; i++) {
            const int n = tid + i * NT;
            if (L / 2 % NT == 0 || n < L / 2)
                cur[i] = mk<T>(load_raw<T>(base + (size_t)(2 * n) * stride, f),
                               load_raw<T>(base + (size_t)(2 * n + 1) * stride, f));
        }
    }
    c2<T> *out = ring + ((size_t)ch * R + slot) * L;
    if (ps.thr > 0.0) {                                   // uniform
        __shared__ T ps_red[16];
        const bool exact = ps.thr >= 1.0;
        T m = (T)0;                                        // exact: 1 if any bit is set; else max |x|
#pragma unroll
        for (int i = 0; i < QP; i++) {
            const int n = tid + i * NT;
            if (L / 2 % NT == 0 || n < L / 2) {
                const T v4[4] = {old[i].x, old[i].y, cur[i].x, cur[i].y};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const T v = v4[q];
                    if (exact) {
                        bool nz;
                        if constexpr (sizeof(T) == 4) nz = __float_as_uint(v) != 0u;
                        else nz = __double_as_longlong(v) != 0ll;
                        if (nz) m = (T)1;
                    } else if (v < (T)0) { if (-v > m) m = -v; }
                    else { if (v > m) m = v; }
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const T o = __shfl_xor(m, off); m = o > m ? o : m; }
        if ((tid & 63) == 0) ps_red[tid >> 6] = m;
        __syncthreads();
        m = ps_red[0];
        for (int w = 1; w < NT / 64; w++) m = ps_red[w] > m ? ps_red[w] : m;
        const bool silent = exact ? (m == (T)0) : !(ps.scale[ch] * (double)m >= ps.thr);
        if (tid == 0) {
            const int was = ps.flags[ch * R + slot], now = silent ? 1 : 0;
            ps.flags[ch * R + slot] = now;
            if (was != now) ps.live[ch] += was - now;
        }
        if (silent) {
            // the block still becomes "the previous block" of the next window (the reference zeroes
            // its transform buffer, not the copy kept for the next call: fftw_convolver.c:181-193)
#pragma unroll
            for (int i = 0; i < QP; i++) {
                const int n = tid + i * NT;
                if (L / 2 % NT == 0 || n < L / 2) pv[n] = cur[i];
            }
            for (int k = tid; k < L; k += NT) out[k] = mk<T>((T)0, (T)0);
            return;
        }
    }
    TwRegs<T, LOG2L, NT> twr;
    twr.prefetch(tw);
    c2<T> uw[QU];
#pragma unroll
    for (int i = 0; i < QU; i++) { const int k = 1 + tid + i * NT; uw[i] = tw[k <= L / 2 ? k : 0]; }
    BF_PROBE(1);
#pragma unroll
    for (int i = 0; i < QP; i++) {
        const int n = tid + i * NT;
        if (L / 2 % NT == 0 || n < L / 2) {
            s[n] = old[i];
            s[L / 2 + n] = cur[i];
            pv[n] = cur[i];                      // same thread read it above: no hazard
        }
    }
    __syncthreads();
    BF_PROBE(2);
    lds_fft<T, LOG2L, NT, false>(s, twr);
    BF_PROBE(10);

    if (tid == 0) out[0] = mk<T>(s[0].x + s[0].y, s[0].x - s[0].y);
#pragma unroll
    for (int i = 0; i < QU; i++) {
        const int k = 1 + tid + i * NT;
        if (k <= L / 2) {
            c2<T> xk, xlk;
            untangle(s[k], conj(s[L - k]), uw[i], xk, xlk);
            out[k] = xk;
            if (k != L - k) out[L - k] = xlk;
        }
    }
    BF_PROBE(11);
}

template <typename T, int LOG2L, int NTP = fft_threads<T>(LOG2L)>
__global__ __launch_bounds__(NTP) void
fft_in_kernel(const uint8_t *__restrict__ raw, const DevFormat *__restrict__ fmt, T *__restrict__ prev,
              c2<T> *__restrict__ ring, const c2<T> *__restrict__ tw, int R, int slot,
              const BlockState *__restrict__ bs, PowerSave ps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (bs) slot = (int)(bs->t % (unsigned int)R);
    fft_in_body<T, LOG2L, NTP>(xcd_channel(blockIdx.x, gridDim.x), smem, raw, fmt, prev, ring, tw, R, slot, ps);
}

// K1 on the wave FFT (fft_wave.h; L = 1024 .. 8192, NT = L/16 threads): same statement as
// fft_in_body -- window [previous L | new L], packed pairs, complex FFT, untangle into the ring
// slot -- but the first radix pass runs on the registers the global loads landed in, and the last
// two radix-8 passes are joined by wave-level exchanges.
template <typename T, int LOG2L>
__device__ __forceinline__ void
fft_in_wave_body(int ch, unsigned char *smem, const uint8_t *__restrict__ raw, const DevFormat *__restrict__ fmt,
                 T *__restrict__ prev, c2<T> *__restrict__ ring, const c2<T> *__restrict__ tw, int R, int slot,
                 PowerSave ps) {
    using G = WaveGeo<LOG2L>;
    constexpr int L = G::L, NT = G::NT, QP = 8, QU = 8;
    LdsArr<T> s{reinterpret_cast<c2<T> *>(smem)};
    const int tid = threadIdx.x;
    const DevFormat f = fmt[ch];
    c2<T> *pv = reinterpret_cast<c2<T> *>(prev + (size_t)ch * L);
    const uint8_t *base = f.alt ? f.alt : raw + f.byte_offset;
    const size_t stride = (size_t)f.sample_spacing * f.bytes;

    BF_PROBE(0);
    // pair e of this thread is pair index tid + e*NT of its half of the window: the very elements
    // its first-pass butterflies need (L/R0 = B0 * NT)
    c2<T> old[QP], cur[QP];
#pragma unroll
    for (int e = 0; e < QP; e++) old[e] = pv[tid + e * NT];
    const uintptr_t ba = (uintptr_t)base;
    if (f.bytes == 4 && ((ba | stride) & 3) == 0) {
        load_pairs_words<T, uint32_t, QP, NT, L / 2>(base, stride, f, tid, cur);
    } else if (f.bytes == 2 && ((ba | stride) & 1) == 0) {
        load_pairs_words<T, uint16_t, QP, NT, L / 2>(base, stride, f, tid, cur);
    } else if (f.bytes == 8 && ((ba | stride) & 7) == 0) {
        load_pairs_words<T, uint64_t, QP, NT, L / 2>(base, stride, f, tid, cur);
    } else {
#pragma unroll
        for (int e = 0; e < QP; e++) {
            const int n = tid + e * NT;
            cur[e] = mk<T>(load_raw<T>(base + (size_t)(2 * n) * stride, f),
                           load_raw<T>(base + (size_t)(2 * n + 1) * stride, f));
        }
    }
    c2<T> *out = ring + ((size_t)ch * R + slot) * L;
    if (ps.thr > 0.0) {                                   // uniform
        __shared__ T ps_red[16];
        const bool exact = ps.thr >= 1.0;
        T m = (T)0;
#pragma unroll
        for (int e = 0; e < QP; e++) {
            const T v4[4] = {old[e].x, old[e].y, cur[e].x, cur[e].y};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const T v = v4[q];
                if (exact) {
                    bool nz;
                    if constexpr (sizeof(T) == 4) nz = __float_as_uint(v) != 0u;
                    else nz = __double_as_longlong(v) != 0ll;
                    if (nz) m = (T)1;
                } else if (v < (T)0) { if (-v > m) m = -v; }
                else { if (v > m) m = v; }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const T o = __shfl_xor(m, off); m = o > m ? o : m; }
        if ((tid & 63) == 0) ps_red[tid >> 6] = m;
        __syncthreads();
        m = ps_red[0];
        for (int w = 1; w < NT / 64; w++) m = ps_red[w] > m ? ps_red[w] : m;
        const bool silent = exact ? (m == (T)0) : !(ps.scale[ch] * (double)m >= ps.thr);
        if (tid == 0) {
            const int was = ps.flags[ch * R + slot], now = silent ? 1 : 0;
            ps.flags[ch * R + slot] = now;
            if (was != now) ps.live[ch] += was - now;
        }
        if (silent) {
#pragma unroll
            for (int e = 0; e < QP; e++) pv[tid + e * NT] = cur[e];
            for (int k = tid; k < L; k += NT) out[k] = mk<T>((T)0, (T)0);
            return;
        }
    }
    WaveTw<T, LOG2L> twr;
    twr.prefetch(tw);
    c2<T> uw[QU];
#pragma unroll
    for (int i = 0; i < QU; i++) { const int k = 1 + tid + i * NT; uw[i] = tw[k <= L / 2 ? k : 0]; }
    BF_PROBE(1);
    // first pass straight from the loaded registers: butterfly b, input r = z[(tid + b*NT) + r*L/R0];
    // the first R0/2 inputs come from the previous block, the others from the new one
    c2<T> z[G::B0][G::R0];
#pragma unroll
    for (int b = 0; b < G::B0; b++) {
#pragma unroll
        for (int r = 0; r < G::R0; r++)
            z[b][r] = r < G::R0 / 2 ? old[b + r * G::B0] : cur[b + (r - G::R0 / 2) * G::B0];
    }
#pragma unroll
    for (int e = 0; e < QP; e++) pv[tid + e * NT] = cur[e];       // same thread read it above: no hazard
    wave_p0_regs<T, LOG2L, false>(s, z);
    BF_PROBE(2);
    c2<T> x[2][8];
    wave_p123<T, LOG2L, false>(s, twr, x);
    BF_PROBE(8);
    wave_store<T, LOG2L>(s, x);
    BF_PROBE(10);

    if (tid == 0) out[0] = mk<T>(s[0].x + s[0].y, s[0].x - s[0].y);
    {
        // bins k = 1 + tid + i*NT and L - k: NT is a multiple of 16, so both runs are one LDS
        // address plus immediate offsets
        const c2<T> *pk = lds_at(s, 1 + tid), *pl = lds_at(s, L - 1 - tid);
#pragma unroll
        for (int i = 0; i < QU; i++) {
            const int k = 1 + tid + i * NT;
            c2<T> xk, xlk;
            untangle(pk[i * lds_stride(NT)], conj(pl[-i * lds_stride(NT)]), uw[i], xk, xlk);
            out[k] = xk;
            if (k != L - k) out[L - k] = xlk;
        }
    }
    BF_PROBE(11);
}

template <typename T, int LOG2L>
__global__ __launch_bounds__(WaveGeo<LOG2L>::NT) void
fft_in_wave_kernel(const uint8_t *__restrict__ raw, const DevFormat *__restrict__ fmt, T *__restrict__ prev,
                   c2<T> *__restrict__ ring, const c2<T> *__restrict__ tw, int R, int slot,
                   const BlockState *__restrict__ bs, PowerSave ps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (bs) slot = (int)(bs->t % (unsigned int)R);
    fft_in_wave_body<T, LOG2L>(xcd_channel(blockIdx.x, gridDim.x), smem, raw, fmt, prev, ring, tw, R, slot, ps);
}

// ------------------------------------------------------------------ K7: taps -> coefficient partition

// One workgroup per partition: [0..L) zero, [L..2L) = taps * scale, real FFT, * 1/n_fft.
// (fftw_convolver.c:535-569).  bad[0] is set if a scaled tap is not finite.
template <typename T, int LOG2L>
__global__ __launch_bounds__(fft_threads<T>(LOG2L)) void
coeff_prep_kernel(const T *__restrict__ taps, int n_taps, T scale, c2<T> *__restrict__ H,
                  const c2<T> *__restrict__ tw, int *__restrict__ bad) {
    constexpr int L = 1 << LOG2L, NT = fft_threads<T>(LOG2L);
    constexpr int QU = UT<T, LOG2L>::QU;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsArr<T> s{reinterpret_cast<c2<T> *>(smem)};
    const int part = blockIdx.x, tid = threadIdx.x;
    const int first = part * L;
    TwRegs<T, LOG2L, NT> twr;
    twr.prefetch(tw);
    c2<T> uw[QU];
#pragma unroll
    for (int i = 0; i < QU; i++) { const int k = 1 + tid + i * NT; uw[i] = tw[k <= L / 2 ? k : 0]; }
    int notfinite = 0;
    for (int n = tid; n < L / 2; n += NT) {
        s[n] = mk<T>((T)0, (T)0);
        const int i0 = first + 2 * n, i1 = i0 + 1;
        const T a = i0 < n_taps ? taps[i0] * scale : (T)0;
        const T b = i1 < n_taps ? taps[i1] * scale : (T)0;
        if (!isfinite(a) || !isfinite(b)) notfinite = 1;
        s[L / 2 + n] = mk<T>(a, b);
    }
    if (notfinite) atomicOr(bad, 1);
    __syncthreads();
    lds_fft<T, LOG2L, NT, false>(s, twr);
    const T inv = (T)1.0 / (T)(2 * L);
    c2<T> *out = H + (size_t)part * L;
    if (tid == 0) out[0] = mk<T>((s[0].x + s[0].y) * inv, (s[0].x - s[0].y) * inv);
#pragma unroll
    for (int i = 0; i < QU; i++) {
        const int k = 1 + tid + i * NT;
        if (k <= L / 2) {
            c2<T> xk, xlk;
            untangle(s[k], conj(s[L - k]), uw[i], xk, xlk);
            out[k] = mk<T>(xk.x * inv, xk.y * inv);
            if (k != L - k) out[L - k] = mk<T>(xlk.x * inv, xlk.y * inv);
        }
    }
}

// 16-byte global load of V = 16/sizeof(c2<T>) spectrum elements.  The pointer comes out of a
// plan struct, so the compiler cannot know its address space and would emit flat_load (which
// also ties up lgkmcnt); the explicit global address space gives global_load_dwordx4.
// NT: non-temporal hint for data that is streamed exactly once per block (coefficients).
template <typename T, bool NT> struct Load16;
template <bool NT> struct Load16<float, NT> {
    static __device__ __forceinline__ void get(const c2<float> *p, c2<float> *o) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(1))) const v4f *gp;
        v4f q;
        if (NT) q = __builtin_nontemporal_load((gp)(const void *)p); else q = *(gp)(const void *)p;
        o[0] = mk<float>(q.x, q.y); o[1] = mk<float>(q.z, q.w);
    }
};
template <bool NT> struct Load16<double, NT> {
    static __device__ __forceinline__ void get(const c2<double> *p, c2<double> *o) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(1))) const v2d *gp;
        v2d q;
        if (NT) q = __builtin_nontemporal_load((gp)(const void *)p); else q = *(gp)(const void *)p;
        o[0] = mk<double>(q.x, q.y);
    }
};

// reference cbuf layout ("4 re / 4 im", Nyquist in slot 4; fftw_convfuns.h:25-43) <-> packed spectrum.
// Pure permutation: what `format: "processed"` coefficient files and shared-memory coefficient
// sets hold (bfconf.c:1924-1971) and what bfaccess->coeffs_data exposes (bfmod.h:139).
template <typename T>
__global__ void reorder_kernel(const T *__restrict__ q, c2<T> *__restrict__ packed, int L, int to_packed,
                               T *__restrict__ q_out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int blk = blockIdx.y;
    if (k >= L) return;
    const size_t base = (size_t)blk * 2 * L;
    const int qre = 8 * (k >> 2) + (k & 3), qim = qre + 4;
    if (to_packed) {
        packed[(size_t)blk * L + k] = mk<T>(q[base + qre], q[base + qim]);
    } else {
        const c2<T> v = packed[(size_t)blk * L + k];
        q_out[base + qre] = v.x;
        q_out[base + qim] = v.y;
    }
}

// set-major coefficient partitions -> the stream-ordered copy (StreamLayout).  One workgroup per
// (partition, tile, entry); where[] names the chunk slot of every entry: (group, chunk, q).
struct StreamWhere { int group, chunk, q, pad; };

template <typename T>
__global__ __launch_bounds__(256) void
stream_relayout_kernel(const MacEntry<T> *__restrict__ entries, const StreamWhere *__restrict__ where,
                       const int *__restrict__ which, int p_first, int L, int n_groups, int n_chunks,
                       StreamLayout sl) {
    const int p = p_first + (int)blockIdx.x, tile = blockIdx.y;
    const int e = which ? which[blockIdx.z] : (int)blockIdx.z;
    const StreamWhere w = where[e];
    // the MAC's own block decode, inverted: tc = tile * n_chunks + chunk, xcd = tc & 7
    const int tc = tile * n_chunks + w.chunk;
    const unsigned long long bid = (unsigned long long)((tc >> 3) * n_groups + w.group) * 8ull + (unsigned long long)(tc & 7);
    const int P = (int)(sl.entry_bytes / ((unsigned int)OG * sl.chunk));
    if (p >= P) return;
    unsigned char *dst = const_cast<unsigned char *>(sl.base) + bid * sl.slice + (unsigned int)w.q * sl.entry_bytes +
                         (unsigned int)p * ((unsigned int)OG * sl.chunk) + (unsigned int)threadIdx.x * 16u;
    const size_t soff = ((size_t)p * L) * sizeof(c2<T>) + (size_t)tile * sl.chunk + (size_t)threadIdx.x * 16u;
    if ((size_t)tile * sl.chunk + (size_t)threadIdx.x * 16u >= (size_t)L * sizeof(c2<T>)) return;
#pragma unroll
    for (int j = 0; j < OG; j++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(entries[e].term[j].H) + soff);
        *reinterpret_cast<uint4 *>(dst + (unsigned int)j * sl.chunk) = v;
    }
}

// ------------------------------------------------------------------ K2: crossbar MAC

// acc += x * h, complex, as four explicit fused multiply-adds in a fixed order.  Written out
// (rather than left to -ffp-contract) so that the unrolled body and the remainder of the
// partition loop round identically: the result must not depend on how many partitions exist
// yet during warm-up.  `am` is 0 for the lane that holds element 0 = (DC, Nyquist), whose two
// components are real*real products (fftw_convfuns.h:545-559), else 1.
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T>
__device__ __forceinline__ void cmac_first(T &re, T &im, T xr, T xi, c2<T> h, T am, bool dc) {
    const T hsel = dc ? h.y : h.x;
    re = fma_t(xr, h.x, re);
    re = fma_t(-(am * xi), h.y, re);
    im = fma_t(am * xr, h.y, im);
    im = fma_t(xi, hsel, im);
}
template <typename T>
__device__ __forceinline__ void cmac(T &re, T &im, T xr, T xi, c2<T> h) {
    re = fma_t(xr, h.x, re);
    re = fma_t(-xi, h.y, re);
    im = fma_t(xr, h.y, im);
    im = fma_t(xi, h.x, im);
}


// Z[o][k] (+)= sum over entries (ring, delay) and partitions p of
//              scale * ring[(t - p - delay) mod R][k] * H[p][k]
// Work decomposition: workgroup = (bin tile, output group of OG, chunk of the group's
// entries).  A lane owns V consecutive bins (16 bytes); the ring value is loaded once per
// (entry, p) and reused in registers for the OG outputs; each coefficient element is read
// exactly once per block, as a 1 KiB-per-wave contiguous load.  The 8 output groups that
// need the same ring tile get the same blockIdx%8 (one XCD, adjacent dispatch) so that the
// ring re-reads hit that XCD's L2.  Partial sums of the chunks go to Zp[chunk][o][k] and
// are added up by the consumer (K3 or sum_partials_kernel): deterministic, no atomics.
template <typename T, bool NT, int UNROLL = 2>
__global__ __launch_bounds__(256) void
mac_xbar_kernel(const MacEntry<T> *__restrict__ entries, const ChunkRange *__restrict__ chunks,
                c2<T> *__restrict__ Zp, int L, int n_out_padded, int n_groups, int n_chunks,
                int n_tc, unsigned int t, int age, const BlockState *__restrict__ bs, StreamLayout sl) {
    constexpr int V = 16 / sizeof(c2<T>);          // bins per lane: 2 (f32) / 1 (f64)
    if (bs) { t = bs->t; age = bs->age; }
    // XCD-aware decode (blocks b and b+8 share an XCD)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int group = local % n_groups;
    const int tc = (local / n_groups) * 8 + xcd;
    if (tc >= n_tc) return;
    const int tile = tc / n_chunks, chunk = tc % n_chunks;
    const int k0 = (tile * (int)blockDim.x + (int)threadIdx.x) * V;
    if (k0 >= L) return;
    const bool dc = (k0 == 0);
    const T am = dc ? (T)0 : (T)1;

    T acc[OG][2 * V];
#pragma unroll
    for (int j = 0; j < OG; j++)
#pragma unroll
        for (int v = 0; v < 2 * V; v++) acc[j][v] = (T)0;

    const ChunkRange cr = chunks[group * n_chunks + chunk];
    for (int e = cr.begin; e < cr.end; e++) {
        const MacEntry<T> *E = &entries[e];
        const c2<T> *ring = E->ring;
        const int R = E->R, delay = E->delay;
        const int p0 = E->p0;
        int maxP = E->maxP;
        if (maxP > age - delay) maxP = age - delay;     // blocks that exist yet (procblocks)
        // powersave (bfrun.c:1541-1553, 1694-1770): every block this input still has in its ring
        // is silence -> nothing to add, nothing to read
        if (E->live != nullptr && *E->live == 0) continue;
        if (E->dense == 1 || E->dense == 16) {
            const unsigned int mask = E->dense == 1 ? 0xffu : (unsigned int)E->mask;
            // The crossbar case: OG coefficient terms of equal length (or, for the last group of
            // a crossbar whose output count is not a multiple of OG and for small matrices, the
            // `mask` subset of them: wave-uniform branches around the absent ones).  Per partition one ring
            // load and OG coefficient loads (1 KiB per wave each) are issued back to back from
            // wave-uniform bases + one shared lane offset, then accumulated as they land.
            const c2<T> *Hs[OG];
            T sc[OG];
#pragma unroll
            for (int j = 0; j < OG; j++) { Hs[j] = E->term[j].H; sc[j] = E->term[j].scale; }
            constexpr int UR = UNROLL > 0 ? UNROLL : 2;
            // where partition p of term j starts for this lane: Hs[j] + p * hstep + hlane
            unsigned int hstep = (unsigned int)L * (unsigned int)sizeof(c2<T>);
            unsigned int hlane = (unsigned int)k0 * (unsigned int)sizeof(c2<T>);
            if (UNROLL == 0 && sl.base != nullptr) {
                const unsigned char *wg = sl.base + (unsigned long long)bid * sl.slice + (unsigned int)(e - cr.begin) * sl.entry_bytes;
#pragma unroll
                for (int j = 0; j < OG; j++) Hs[j] = (const c2<T> *)(wg + (unsigned int)j * sl.chunk);
                hstep = (unsigned int)OG * sl.chunk;
                hlane = (unsigned int)threadIdx.x * 16u;
            }
            if (mask != 0xffu) {
#pragma unroll 2
                for (int p = p0; p < maxP; p++) {
                    const unsigned int slot = (t - (unsigned int)p - (unsigned int)delay) % (unsigned int)R;
                    const unsigned int xoff = (slot * (unsigned int)L + (unsigned int)k0) * (unsigned int)sizeof(c2<T>);
                    const unsigned int hoff = ((unsigned int)p * (unsigned int)L + (unsigned int)k0) * (unsigned int)sizeof(c2<T>);
                    c2<T> x[V], h[OG][V];
                    Load16<T, false>::get((const c2<T> *)((const char *)ring + xoff), x);
#pragma unroll
                    for (int j = 0; j < OG; j++)
                        if (mask & (1u << j)) Load16<T, NT>::get((const c2<T> *)((const char *)Hs[j] + hoff), h[j]);
#pragma unroll
                    for (int j = 0; j < OG; j++) {
                        if (!(mask & (1u << j))) continue;
                        {
                            const T xr = x[0].x * sc[j], xi = x[0].y * sc[j];
                            cmac_first(acc[j][0], acc[j][1], xr, xi, h[j][0], am, dc);
                        }
                        if constexpr (V == 2) {
                            const T xr = x[1].x * sc[j], xi = x[1].y * sc[j];
                            cmac(acc[j][2], acc[j][3], xr, xi, h[j][1]);
                        }
                    }
                }
                continue;
            }
            if constexpr (UNROLL == 0) {
                // Rotating three-stage pipeline: while one partition's products are accumulated
                // the loads of the next two are in flight and the one after is being issued --
                // the memory pipe never drains inside an entry.
                struct Stage { c2<T> x[V]; c2<T> h[OG][V]; };
                auto issue = [&](Stage &st, int p) {
                    const unsigned int slot = (t - (unsigned int)p - (unsigned int)delay) % (unsigned int)R;
                    const unsigned int xoff = (slot * (unsigned int)L + (unsigned int)k0) * (unsigned int)sizeof(c2<T>);
                    const unsigned int hoff = (unsigned int)p * hstep + hlane;
                    Load16<T, false>::get((const c2<T> *)((const char *)ring + xoff), st.x);
#pragma unroll
                    for (int j = 0; j < OG; j++) Load16<T, NT>::get((const c2<T> *)((const char *)Hs[j] + hoff), st.h[j]);
                };
                auto consume = [&](const Stage &st) {
#pragma unroll
                    for (int j = 0; j < OG; j++) {
                        {
                            const T xr = st.x[0].x * sc[j], xi = st.x[0].y * sc[j];
                            cmac_first(acc[j][0], acc[j][1], xr, xi, st.h[j][0], am, dc);
                        }
                        if constexpr (V == 2) {
                            const T xr = st.x[1].x * sc[j], xi = st.x[1].y * sc[j];
                            cmac(acc[j][2], acc[j][3], xr, xi, st.h[j][1]);
                        }
                    }
                };
                const int n = maxP - p0;
                Stage s0, s1, s2;
                if (n >= 1) issue(s0, p0);
                if (n >= 2) issue(s1, p0 + 1);
                if (n >= 3) issue(s2, p0 + 2);
                int i = 0;
                for (; i + 6 <= n; i += 3) {                 // steady state: no conditions
                    consume(s0); issue(s0, p0 + i + 3);
                    consume(s1); issue(s1, p0 + i + 4);
                    consume(s2); issue(s2, p0 + i + 5);
                }
                for (; i + 3 <= n; i += 3) {                 // at most two trips
                    consume(s0); if (i + 3 < n) issue(s0, p0 + i + 3);
                    consume(s1); if (i + 4 < n) issue(s1, p0 + i + 4);
                    consume(s2); if (i + 5 < n) issue(s2, p0 + i + 5);
                }
                if (i < n) consume(s0);
                if (i + 1 < n) consume(s1);
                continue;
            }
#pragma unroll UR
            for (int p = p0; p < maxP; p++) {
                // byte offsets kept in 32 bits (N * L * 16 < 4 GiB) so that the loads take the
                // scalar-base + 32-bit lane-offset form
                const unsigned int slot = (t - (unsigned int)p - (unsigned int)delay) % (unsigned int)R;
                const unsigned int xoff = (slot * (unsigned int)L + (unsigned int)k0) * (unsigned int)sizeof(c2<T>);
                const unsigned int hoff = ((unsigned int)p * (unsigned int)L + (unsigned int)k0) * (unsigned int)sizeof(c2<T>);
                c2<T> x[V], h[OG][V];
                Load16<T, false>::get((const c2<T> *)((const char *)ring + xoff), x);
#pragma unroll
                for (int j = 0; j < OG; j++) Load16<T, NT>::get((const c2<T> *)((const char *)Hs[j] + hoff), h[j]);
#pragma unroll
                for (int j = 0; j < OG; j++) {
                    {
                        const T xr = x[0].x * sc[j], xi = x[0].y * sc[j];
                        cmac_first(acc[j][0], acc[j][1], xr, xi, h[j][0], am, dc);
                    }
                    if constexpr (V == 2) {
                        const T xr = x[1].x * sc[j], xi = x[1].y * sc[j];
                        cmac(acc[j][2], acc[j][3], xr, xi, h[j][1]);
                    }
                }
            }
            continue;
        }
        if (E->dense >= 2 && E->dense < 2 + OG) {
            // One coefficient term only (one-to-one filters): nothing to reuse, so the depth
            // comes from four partitions (ring + coefficient tile each) in flight per wave.
            const int js = E->dense - 2;
            const c2<T> *H = E->term[js].H;
            const T sc = E->term[js].scale;
            if (maxP > E->term[js].P) maxP = E->term[js].P;
            T ta[2 * V];
#pragma unroll
            for (int v = 0; v < 2 * V; v++) ta[v] = (T)0;
#pragma unroll 4
            for (int p = p0; p < maxP; p++) {
                const unsigned int slot = (t - (unsigned int)p - (unsigned int)delay) % (unsigned int)R;
                const unsigned int xoff = (slot * (unsigned int)L + (unsigned int)k0) * (unsigned int)sizeof(c2<T>);
                const unsigned int hoff = ((unsigned int)p * (unsigned int)L + (unsigned int)k0) * (unsigned int)sizeof(c2<T>);
                c2<T> x[V], h[V];
                Load16<T, false>::get((const c2<T> *)((const char *)ring + xoff), x);
                Load16<T, false>::get((const c2<T> *)((const char *)H + hoff), h);
                {
                    const T xr = x[0].x * sc, xi = x[0].y * sc;
                    cmac_first(ta[0], ta[1], xr, xi, h[0], am, dc);
                }
                if constexpr (V == 2) {
                    const T xr = x[1].x * sc, xi = x[1].y * sc;
                    cmac(ta[2], ta[3], xr, xi, h[1]);
                }
            }
#pragma unroll
            for (int j = 0; j < OG; j++)
                if (j == js) {
#pragma unroll
                    for (int v = 0; v < 2 * V; v++) acc[j][v] += ta[v];
                }
            continue;
        }
        for (int p = p0; p < maxP; p++) {
            const unsigned int slot = (t - (unsigned int)p - (unsigned int)delay) % (unsigned int)R;
            c2<T> x[V];
            {
                const c2<T> *xp = ring + (size_t)slot * L + k0;
                if constexpr (V == 2) {
                    const float4 q = *reinterpret_cast<const float4 *>(xp);
                    x[0] = mk<T>(q.x, q.y); x[1] = mk<T>(q.z, q.w);
                } else {
                    x[0] = *xp;
                }
            }
#pragma unroll
            for (int j = 0; j < OG; j++) {
                const int kind = E->term[j].kind;
                if (kind == TERM_NONE || p >= E->term[j].P) continue;
                const T sc = E->term[j].scale;
                c2<T> h[V];
                if (kind == TERM_COEFF) {
                    const c2<T> *hp = E->term[j].H + (size_t)p * L + k0;
                    if constexpr (V == 2) {
                        const float4 q = *reinterpret_cast<const float4 *>(hp);
                        h[0] = mk<T>(q.x, q.y); h[1] = mk<T>(q.z, q.w);
                    } else {
                        h[0] = *hp;
                    }
                } else if (kind == TERM_DIRAC) {
                    // spectrum of a unit pulse at sample L over n_fft: (-1)^k / n_fft
                    // (fftw_convfuns.h:592-619); bins k0 (even) and k0+1 (odd)
                    const T f = (T)1.0 / (T)(2 * L);
                    if constexpr (V == 2) { h[0] = mk<T>(f, (T)0); h[1] = mk<T>(-f, (T)0); }
                    else { h[0] = mk<T>((k0 & 1) ? -f : f, (T)0); }
                    if (dc) h[0] = mk<T>(f, f);
                } else {
                    h[0] = mk<T>((T)1, (T)0);
                    if constexpr (V == 2) h[1] = mk<T>((T)1, (T)0);
                    if (dc) h[0] = mk<T>((T)1, (T)1);
                }
                // first bin of the lane: element 0 of the spectrum is (DC, Nyquist), a
                // real*real product per component (fftw_convfuns.h:545-559)
                {
                    const T xr = x[0].x * sc, xi = x[0].y * sc;
                    cmac_first(acc[j][0], acc[j][1], xr, xi, h[0], am, dc);
                }
                if constexpr (V == 2) {
                    const T xr = x[1].x * sc, xi = x[1].y * sc;
                    cmac(acc[j][2], acc[j][3], xr, xi, h[1]);
                }
            }
        }
    }
    c2<T> *zp = Zp + ((size_t)chunk * n_out_padded + (size_t)group * OG) * L + k0;
#pragma unroll
    for (int j = 0; j < OG; j++) {
        if constexpr (V == 2) {
            *reinterpret_cast<float4 *>(zp + (size_t)j * L) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
        } else {
            zp[(size_t)j * L] = mk<T>(acc[j][0], acc[j][1]);
        }
    }
}

// K2 over TWO consecutive blocks in one pass over the coefficients (uniform crossbars only: every
// entry OG coefficient terms of full length).  The MAC of a large crossbar is nothing but the
// coefficient stream -- config C reads 8 GiB per block to produce 2 MiB -- so a host that has two
// periods in hand (the blocking-I/O topology, which keeps two in flight anyway) gets two blocks for
// the traffic of one:   Z_t[o] += X_(t-p) H_p   and   Z_(t+1)[o] += X_(t+1-p) H_p   share H_p, and
// X_(t+1-p) is the ring value the previous partition used for block t (one ring load per
// partition, kept in registers).  Per block the products are accumulated in exactly the order of
// mac_xbar_kernel: the two outputs are bit-identical to two single-block launches.
template <typename T, bool NT>
__global__ __launch_bounds__(256) void
mac_xbar2_kernel(const MacEntry<T> *__restrict__ entries, const ChunkRange *__restrict__ chunks,
                 c2<T> *__restrict__ Zp0, c2<T> *__restrict__ Zp1, int L, int n_out_padded, int n_groups, int n_chunks,
                 int n_tc, unsigned int t /* counter of the FIRST block; the rings hold t + 1 already */, StreamLayout sl) {
    constexpr int V = 16 / sizeof(c2<T>);
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int group = local % n_groups;
    const int tc = (local / n_groups) * 8 + xcd;
    if (tc >= n_tc) return;
    const int tile = tc / n_chunks, chunk = tc % n_chunks;
    const int k0 = (tile * (int)blockDim.x + (int)threadIdx.x) * V;
    if (k0 >= L) return;
    const bool dc = (k0 == 0);
    const T am = dc ? (T)0 : (T)1;

    T acc0[OG][2 * V], acc1[OG][2 * V];
#pragma unroll
    for (int j = 0; j < OG; j++)
#pragma unroll
        for (int v = 0; v < 2 * V; v++) { acc0[j][v] = (T)0; acc1[j][v] = (T)0; }

    const ChunkRange cr = chunks[group * n_chunks + chunk];
    for (int e = cr.begin; e < cr.end; e++) {
        const MacEntry<T> *E = &entries[e];
        const c2<T> *ring = E->ring;
        const int R = E->R, delay = E->delay, p0 = E->p0, maxP = E->maxP;
        if (E->live != nullptr && *E->live == 0) continue;           // powersave: nothing but silence in this ring
        const c2<T> *Hs[OG];
        T sc[OG];
#pragma unroll
        for (int j = 0; j < OG; j++) { Hs[j] = E->term[j].H; sc[j] = E->term[j].scale; }
        unsigned int hstep = (unsigned int)L * (unsigned int)sizeof(c2<T>);
        unsigned int hlane = (unsigned int)k0 * (unsigned int)sizeof(c2<T>);
        if (sl.base != nullptr) {
            const unsigned char *wg = sl.base + (unsigned long long)bid * sl.slice + (unsigned int)(e - cr.begin) * sl.entry_bytes;
#pragma unroll
            for (int j = 0; j < OG; j++) Hs[j] = (const c2<T> *)(wg + (unsigned int)j * sl.chunk);
            hstep = (unsigned int)OG * sl.chunk;
            hlane = (unsigned int)threadIdx.x * 16u;
        }
        struct Stage { c2<T> x[V]; c2<T> h[OG][V]; };
        auto ring_at = [&](unsigned int blk, c2<T> *x) {              // X_blk: ring slot (blk - delay) mod R
            const unsigned int slot = (blk - (unsigned int)delay) % (unsigned int)R;
            Load16<T, false>::get((const c2<T> *)((const char *)ring + (slot * (unsigned int)L + (unsigned int)k0) * (unsigned int)sizeof(c2<T>)), x);
        };
        auto issue = [&](Stage &st, int p) {
            ring_at(t - (unsigned int)p, st.x);
            const unsigned int hoff = (unsigned int)p * hstep + hlane;
#pragma unroll
            for (int j = 0; j < OG; j++) Load16<T, NT>::get((const c2<T> *)((const char *)Hs[j] + hoff), st.h[j]);
        };
        c2<T> xn[V];                                                    // X_(t+1-p): what block t+1 multiplies H_p with
        auto consume = [&](const Stage &st) {
#pragma unroll
            for (int j = 0; j < OG; j++) {
                {
                    const T xr = st.x[0].x * sc[j], xi = st.x[0].y * sc[j];
                    cmac_first(acc0[j][0], acc0[j][1], xr, xi, st.h[j][0], am, dc);
                    const T yr = xn[0].x * sc[j], yi = xn[0].y * sc[j];
                    cmac_first(acc1[j][0], acc1[j][1], yr, yi, st.h[j][0], am, dc);
                }
                if constexpr (V == 2) {
                    const T xr = st.x[1].x * sc[j], xi = st.x[1].y * sc[j];
                    cmac(acc0[j][2], acc0[j][3], xr, xi, st.h[j][1]);
                    const T yr = xn[1].x * sc[j], yi = xn[1].y * sc[j];
                    cmac(acc1[j][2], acc1[j][3], yr, yi, st.h[j][1]);
                }
            }
#pragma unroll
            for (int v = 0; v < V; v++) xn[v] = st.x[v];               // X_(t-p) is X_(t+1-(p+1))
        };
        const int n = maxP - p0;
        Stage s0, s1, s2;
        if (n >= 1) { ring_at(t + 1u - (unsigned int)p0, xn); issue(s0, p0); }
        if (n >= 2) issue(s1, p0 + 1);
        if (n >= 3) issue(s2, p0 + 2);
        int i = 0;
        for (; i + 6 <= n; i += 3) {
            consume(s0); issue(s0, p0 + i + 3);
            consume(s1); issue(s1, p0 + i + 4);
            consume(s2); issue(s2, p0 + i + 5);
        }
        for (; i + 3 <= n; i += 3) {
            consume(s0); if (i + 3 < n) issue(s0, p0 + i + 3);
            consume(s1); if (i + 4 < n) issue(s1, p0 + i + 4);
            consume(s2); if (i + 5 < n) issue(s2, p0 + i + 5);
        }
        if (i < n) consume(s0);
        if (i + 1 < n) consume(s1);
    }
    const size_t zoff = ((size_t)chunk * n_out_padded + (size_t)group * OG) * L + k0;
#pragma unroll
    for (int j = 0; j < OG; j++) {
        if constexpr (V == 2) {
            *reinterpret_cast<float4 *>(Zp0 + zoff + (size_t)j * L) = make_float4(acc0[j][0], acc0[j][1], acc0[j][2], acc0[j][3]);
            *reinterpret_cast<float4 *>(Zp1 + zoff + (size_t)j * L) = make_float4(acc1[j][0], acc1[j][1], acc1[j][2], acc1[j][3]);
        } else {
            Zp0[zoff + (size_t)j * L] = mk<T>(acc0[j][0], acc0[j][1]);
            Zp1[zoff + (size_t)j * L] = mk<T>(acc1[j][0], acc1[j][1]);
        }
    }
}

// K2 for ONE-TO-ONE plans (every output fed by at most one entry of a single coefficient term per
// chunk: massive_config, BASELINE configs[3]): there is nothing to share between outputs, so the
// crossbar kernel's decomposition -- a workgroup owns a 4 KiB bin tile and walks entries and
// partitions -- turns into 4 KiB pieces of hundreds of separate ring and coefficient streams
// (tools/hbm_piece_probe.hip: 5.3 - 5.6 TB/s for that shape against 6.9 for the same bytes read in
// long runs).  Here a workgroup owns one (chunk, output) job and walks the WHOLE spectrum of every
// partition of its entry: a ring slot and a coefficient partition are each one contiguous run of
// L complex numbers (64 KiB at L = 8192), read front to back; the L bins live in registers across the
// partition loop (acc[tile]), one store per bin at the end.  Same per-bin arithmetic, in the same order
// over p, as the crossbar kernel's single-term path.
template <typename T, bool NT, int TILES /* 4 KiB bin tiles per spectrum: L / (256 * V), at least 1 */>
__global__ __launch_bounds__(256, 2) void          // two workgroups per CU: 256 registers per lane
mac_diag_kernel(const MacEntry<T> *__restrict__ entries, const int *__restrict__ jobs /* [n_chunks * n_out_padded]: entry or -1 */,
                c2<T> *__restrict__ Zp, int L, int n_out_padded, unsigned int t, int age, const BlockState *__restrict__ bs,
                int tsplit /* workgroups per job: each takes TILES consecutive tiles of the spectrum */) {
    constexpr int V = 16 / sizeof(c2<T>);          // bins per lane and tile: 2 (f32) / 1 (f64)
    constexpr int TB = 256 * V;                    // bins per tile (4 KiB)
    constexpr int HT = TILES >= 2 ? TILES / 2 : 1; // tiles per half: the two halves of a partition rotate through two register sets
    if (bs) { t = bs->t; age = bs->age; }
    const int job = blockIdx.x / tsplit, part = blockIdx.x % tsplit;
    const int o = job % n_out_padded;
    const int e = jobs[job];
    const int tid = threadIdx.x;
    const int kl = part * TILES * TB + tid * V;    // this lane's first bin: inside tile 0 of this workgroup's share
    if (kl >= L) return;                           // (L < one tile: the lanes beyond the spectrum have nothing to do)
    c2<T> *zp = Zp + (size_t)job * L + kl;         // job = chunk * n_out_padded + o: Zp[chunk][o][.]
    T acc[TILES][2 * V];
#pragma unroll
    for (int tt = 0; tt < TILES; tt++)
#pragma unroll
        for (int v = 0; v < 2 * V; v++) acc[tt][v] = (T)0;
    if (e >= 0) {
        const MacEntry<T> *E = &entries[e];
        const int js = o % OG;
        const c2<T> *ring = E->ring + kl, *H = E->term[js].H + kl;
        const int R = E->R, delay = E->delay, p0 = E->p0;
        const T sc = E->term[js].scale;
        int maxP = E->maxP;
        if (maxP > age - delay) maxP = age - delay;                 // blocks that exist yet (procblocks)
        if (maxP > E->term[js].P) maxP = E->term[js].P;
        if (E->live != nullptr && *E->live == 0) maxP = p0;         // powersave: nothing but silence in the ring
        struct Half { c2<T> x[HT][V], h[HT][V]; };
        // one half of partition p in flight: 2 x HT loads of 16 bytes per lane out of two sequential runs
        auto issue = [&](Half &b, int p, int half) {
            const unsigned int slot = (t - (unsigned int)p - (unsigned int)delay) % (unsigned int)R;
            const c2<T> *xb = ring + (size_t)slot * L + half * HT * TB, *hb = H + (size_t)p * L + half * HT * TB;
#pragma unroll
            for (int i = 0; i < HT; i++) {
                Load16<T, false>::get(xb + i * TB, b.x[i]);
                Load16<T, NT>::get(hb + i * TB, b.h[i]);
            }
        };
        auto consume = [&](const Half &b, int half) {
#pragma unroll
            for (int i = 0; i < HT; i++) {
                const int tt = half * HT + i;
                const bool dc = tt == 0 && kl == 0;                 // element 0 = (DC, Nyquist): real * real per component
                const T am = dc ? (T)0 : (T)1;
                {
                    const T xr = b.x[i][0].x * sc, xi = b.x[i][0].y * sc;
                    cmac_first(acc[tt][0], acc[tt][1], xr, xi, b.h[i][0], am, dc);
                }
                if constexpr (V == 2) {
                    const T xr = b.x[i][1].x * sc, xi = b.x[i][1].y * sc;
                    cmac(acc[tt][2], acc[tt][3], xr, xi, b.h[i][1]);
                }
            }
        };
        Half a, b;
        const int n = maxP - p0;
        if (n > 0) {
            issue(a, p0, 0);
            if constexpr (TILES >= 2) issue(b, p0, 1);
            for (int i = 0; i + 1 < n; i++) {                       // steady state: the next partition's half is requested
                consume(a, 0); issue(a, p0 + i + 1, 0);             // as soon as its registers are free
                if constexpr (TILES >= 2) { consume(b, 1); issue(b, p0 + i + 1, 1); }
            }
            consume(a, 0);
            if constexpr (TILES >= 2) consume(b, 1);
        }
    }
#pragma unroll
    for (int tt = 0; tt < TILES; tt++) {
        // (0 + acc: what the crossbar kernel's single-term path leaves -- bit-identical, sign of zero included)
        if constexpr (V == 2) {
            *reinterpret_cast<float4 *>(zp + tt * TB) =
                make_float4((T)0 + acc[tt][0], (T)0 + acc[tt][1], (T)0 + acc[tt][2], (T)0 + acc[tt][3]);
        } else {
            zp[tt * TB] = mk<T>((T)0 + acc[tt][0], (T)0 + acc[tt][1]);
        }
    }
}

// Z[o][k] = sum_c Zp[c][o][k]   (only used when the spectra leave the engine: multi-GPU)
template <typename T>
__global__ __launch_bounds__(256) void
sum_partials_kernel(const c2<T> *Zp, c2<T> *Z /* may be Zp: in place */, size_t n_per_chunk,
                    size_t n_valid, int n_chunks) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_valid) return;
    c2<T> a = Zp[i];
    for (int c = 1; c < n_chunks; c++) a = a + Zp[(size_t)c * n_per_chunk + i];
    Z[i] = a;
}

// ------------------------------------------------------------------ K4/K5: ring fill (N-way mix + cascade eval)

// A filter that mixes several inputs, or takes other filters' outputs ("from_filters"), owns a
// private ring exactly like the reference's cbuf[n][N] (bfrun.c:1273-1287).  One workgroup per
// such filter computes, for block t,
//     E      = rfft([ y_{t-1} | y_t ]),  y_t = first L samples of irfft( sum_g fscale_g * Y_g )
//                                         (convolver_convolve_eval, fftw_convolver.c:411-433)
//     ring[(t + delay) mod N] = sum_i scale_i * X_i[t]  +  1.0 * E      (bfrun.c:1603-1680)
// with the summation order of mixnscale(INPUT): channel inputs first, the evaluated buffer last.
template <typename T> struct FillJob {
    c2<T> *ring;            // [N][L]
    T *evalprev;            // [L] previous valid half, nullptr when there are no filter inputs
    int n_in, in_off;       // channel inputs: src[in_off .. in_off + n_in)
    int n_up, up_off;       // filter inputs
    int delay, pad;
};

template <typename T> struct MixSrc {
    const c2<T> *spec;      // packed spectrum (an input ring's base, or an upstream Y)
    T scale;
    int R;                  // ring depth of spec (slot = t mod R), 1 for a Y buffer
};

template <typename T, int LOG2L>
__global__ __launch_bounds__(fft_threads<T>(LOG2L)) void
ring_fill_kernel(const FillJob<T> *__restrict__ jobs, const MixSrc<T> *__restrict__ src,
                 const c2<T> *__restrict__ tw, int N, unsigned int t, const BlockState *__restrict__ bs) {
    constexpr int L = 1 << LOG2L, NT = fft_threads<T>(LOG2L);
    constexpr int QP = UT<T, LOG2L>::QP, QU = UT<T, LOG2L>::QU;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsArr<T> s{reinterpret_cast<c2<T> *>(smem)};
    const int tid = threadIdx.x;
    if (bs) t = bs->t;
    const FillJob<T> job = jobs[blockIdx.x];
    c2<T> *dst = job.ring + (size_t)((t + (unsigned int)job.delay) % (unsigned int)N) * L;
    const MixSrc<T> *in = src + job.in_off;
    const MixSrc<T> *up = src + job.up_off;
    TwRegs<T, LOG2L, NT, true> twr;
    c2<T> uw[QU];
#pragma unroll
    for (int i = 0; i < QU; i++) { const int k = 1 + tid + i * NT; uw[i] = tw[k <= L / 2 ? k : 0]; }

    if (job.n_up > 0) {
        twr.prefetch(tw);
        // M = sum_g fscale_g Y_g (mixnscale OUTPUT order), straight into the C2R pre-pass
#pragma unroll
        for (int i = -1; i < QU; i++) {
            const int k = i < 0 ? 0 : 1 + tid + i * NT;
            if ((i < 0 && tid != 0) || k > L / 2) continue;
            c2<T> a = mk<T>(up[0].spec[k].x * up[0].scale, up[0].spec[k].y * up[0].scale);
            c2<T> b = mk<T>((T)0, (T)0);
            if (k != 0) b = mk<T>(up[0].spec[L - k].x * up[0].scale, up[0].spec[L - k].y * up[0].scale);
            for (int g = 1; g < job.n_up; g++) {
                const c2<T> v = up[g].spec[k];
                a = mk<T>(a.x + v.x * up[g].scale, a.y + v.y * up[g].scale);
                if (k != 0) {
                    const c2<T> w = up[g].spec[L - k];
                    b = mk<T>(b.x + w.x * up[g].scale, b.y + w.y * up[g].scale);
                }
            }
            if (k == 0) {
                s[0] = mk<T>(a.x + a.y, a.x - a.y);
            } else {
                c2<T> zk, zlk;
                tangle(a, conj(b), uw[i < 0 ? 0 : i], zk, zlk);
                s[k] = zk;
                if (k != L - k) s[L - k] = zlk;
            }
        }
        __syncthreads();
        lds_fft<T, LOG2L, NT, true>(s, twr);
        // slide: z'[n] = prev pairs (n < L/2), z'[L/2 + n] = new valid pairs; prev <- new
        c2<T> v[QP];
        c2<T> *ep = reinterpret_cast<c2<T> *>(job.evalprev);
#pragma unroll
        for (int b = 0; b < QP; b++) {
            const int n = tid + b * NT;
            if (n < L / 2) v[b] = s[n];
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < QP; b++) {
            const int n = tid + b * NT;
            if (n < L / 2) {
                s[L / 2 + n] = v[b];
                s[n] = ep[n];
                ep[n] = v[b];
            }
        }
        __syncthreads();
        lds_fft<T, LOG2L, NT, false>(s, twr);
    }

#pragma unroll
    for (int i = -1; i < QU; i++) {
        const int k = i < 0 ? 0 : 1 + tid + i * NT;
        if ((i < 0 && tid != 0) || k > L / 2) continue;
        const int k2 = (k == 0 || k == L - k) ? -1 : L - k;
        c2<T> a = mk<T>((T)0, (T)0), b = mk<T>((T)0, (T)0);
        bool first = true;
        for (int q = 0; q < job.n_in; q++) {
            const c2<T> *sp = in[q].spec + (size_t)(t % (unsigned int)in[q].R) * L;
            const T sc = in[q].scale;
            const c2<T> v = sp[k];
            if (first) a = mk<T>(v.x * sc, v.y * sc); else a = mk<T>(a.x + v.x * sc, a.y + v.y * sc);
            if (k2 >= 0) {
                const c2<T> w = sp[k2];
                if (first) b = mk<T>(w.x * sc, w.y * sc); else b = mk<T>(b.x + w.x * sc, b.y + w.y * sc);
            }
            first = false;
        }
        if (job.n_up > 0) {
            c2<T> ek, ek2 = mk<T>((T)0, (T)0);
            if (k == 0) ek = mk<T>(s[0].x + s[0].y, s[0].x - s[0].y);
            else untangle(s[k], conj(s[L - k]), uw[i < 0 ? 0 : i], ek, ek2);
            if (first) { a = ek; b = ek2; } else { a = a + ek; b = b + ek2; }
        }
        dst[k] = a;
        if (k2 >= 0) dst[k2] = b;
    }
}

// Turn a filter that shared its input's ring into a ring owner (first run-time change of its
// input scale or delay): rebuild what the reference's private cbuf[n][] holds right now, i.e.
// slot (t' + delay_old) mod N = scale_old * X[t'] for the last N blocks t' (bfrun.c:1600,1651-1656).
template <typename T>
__global__ __launch_bounds__(256) void
promote_ring_kernel(const c2<T> *__restrict__ src, int R, c2<T> *__restrict__ dst, int N, int L,
                    unsigned int t, int delay_old, T scale_old, int n_valid) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;                     // how many blocks back (0 = most recent)
    if (k >= L || j >= n_valid) return;
    const unsigned int tb = t - 1u - (unsigned int)j;
    const c2<T> v = src[(size_t)(tb % (unsigned int)R) * L + k];
    dst[(size_t)((tb + (unsigned int)delay_old) % (unsigned int)N) * L + k] = mk<T>(v.x * scale_old, v.y * scale_old);
}

// ------------------------------------------------------------------ per-filter MAC (materialised outputs)

// Y_f[k] = sum_p scale * ring[(t - p - delay) mod R][k] * H[p][k]   (or the dirac spectrum)
// for filters whose output spectrum must exist on its own: cascade sources (from_filters of
// another filter) and filters in a cross-fade block (old and new coefficient results).
template <typename T> struct FilterJob {
    const c2<T> *ring;
    const c2<T> *H;
    c2<T> *Y;
    T scale;
    int R, delay, P, kind;
};

template <typename T>
__global__ __launch_bounds__(256) void
mac_filter_kernel(const FilterJob<T> *__restrict__ jobs, int L, unsigned int t, int age,
                  const BlockState *__restrict__ bs) {
    constexpr int V = 16 / sizeof(c2<T>);
    if (bs) { t = bs->t; age = bs->age; }
    const FilterJob<T> job = jobs[blockIdx.y];
    const int k0 = ((int)blockIdx.x * (int)blockDim.x + (int)threadIdx.x) * V;
    if (k0 >= L) return;
    T acc[2 * V];
#pragma unroll
    for (int v = 0; v < 2 * V; v++) acc[v] = (T)0;
    int P = job.P;
    if (P > age - job.delay) P = age - job.delay;
    for (int p = 0; p < P; p++) {
        const unsigned int slot = (t - (unsigned int)p - (unsigned int)job.delay) % (unsigned int)job.R;
        const c2<T> *xp = job.ring + (size_t)slot * L + k0;
#pragma unroll
        for (int v = 0; v < V; v++) {
            const int k = k0 + v;
            const c2<T> x = mk<T>(xp[v].x * job.scale, xp[v].y * job.scale);
            c2<T> h;
            if (job.kind == TERM_COEFF) h = job.H[(size_t)p * L + k];
            else { const T f = (T)1.0 / (T)(2 * L); h = mk<T>((k & 1) ? -f : f, (T)0); if (k == 0) h.y = f; }
            if (k == 0) { acc[0] = fma_t(x.x, h.x, acc[0]); acc[1] = fma_t(x.y, h.y, acc[1]); }
            else cmac(acc[2 * v], acc[2 * v + 1], x.x, x.y, h);
        }
    }
#pragma unroll
    for (int v = 0; v < V; v++) job.Y[k0 + v] = mk<T>(acc[2 * v], acc[2 * v + 1]);
}

// ------------------------------------------------------------------ K6: crossfade

// convolver_crossfade_inplace (fftw_convolver.c:330-368): both results to the time domain, linear
// ramp over the first L samples (the float branch's arithmetic, used for both precisions: the
// reference's double branch reads out of bounds), back to the frequency domain, / n_fft.
template <typename T> struct FadeJob { c2<T> *Ynew; const c2<T> *Yold; };

template <typename T, int LOG2L>
__device__ __forceinline__ void load_c2r(LdsArr<T> s, const c2<T> *__restrict__ z,
                                         const c2<T> *uw /* [QU] untangle twiddles */) {
    constexpr int L = 1 << LOG2L, NT = fft_threads<T>(LOG2L), QU = UT<T, LOG2L>::QU;
    const int tid = threadIdx.x;
    if (tid == 0) { const c2<T> a = z[0]; s[0] = mk<T>(a.x + a.y, a.x - a.y); }
#pragma unroll
    for (int i = 0; i < QU; i++) {
        const int k = 1 + tid + i * NT;
        if (k <= L / 2) {
            c2<T> zk, zlk;
            tangle(z[k], conj(z[L - k]), uw[i], zk, zlk);
            s[k] = zk;
            if (k != L - k) s[L - k] = zlk;
        }
    }
}

template <typename T, int LOG2L>
__global__ __launch_bounds__(fft_threads<T>(LOG2L)) void
crossfade_kernel(const FadeJob<T> *__restrict__ jobs, const c2<T> *__restrict__ tw) {
    constexpr int L = 1 << LOG2L, NT = fft_threads<T>(LOG2L);
    constexpr int QP = UT<T, LOG2L>::QP, QU = UT<T, LOG2L>::QU;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsArr<T> s{reinterpret_cast<c2<T> *>(smem)};
    const int tid = threadIdx.x;
    const FadeJob<T> job = jobs[blockIdx.x];
    TwRegs<T, LOG2L, NT, true> twr;                  // three transforms and the old result live at once
    twr.prefetch(tw);
    c2<T> uw[QU];
#pragma unroll
    for (int i = 0; i < QU; i++) { const int k = 1 + tid + i * NT; uw[i] = tw[k <= L / 2 ? k : 0]; }
    c2<T> oldv[QP];

    load_c2r<T, LOG2L>(s, job.Yold, uw);
    __syncthreads();
    lds_fft<T, LOG2L, NT, true>(s, twr);
#pragma unroll
    for (int b = 0; b < QP; b++) {
        const int n = tid + b * NT;
        if (n < L / 2) oldv[b] = s[n];
    }
    __syncthreads();
    load_c2r<T, LOG2L>(s, job.Ynew, uw);
    __syncthreads();
    lds_fft<T, LOG2L, NT, true>(s, twr);
#pragma unroll
    for (int b = 0; b < QP; b++) {
        const int n = tid + b * NT;
        if (n < L / 2) {
            c2<T> nv = s[n];
            if constexpr (sizeof(T) == 4) {
                const float f = 1.0f / (float)(L - 1);
                const float m0 = (float)(2 * n), m1 = (float)(2 * n + 1);
                nv.x = (float)((double)oldv[b].x * (1.0 - (double)(f * m0)) + (double)(nv.x * f * m0));
                nv.y = (float)((double)oldv[b].y * (1.0 - (double)(f * m1)) + (double)(nv.y * f * m1));
            } else {
                const double d = 1.0 / (double)(L - 1);
                const double m0 = (double)(2 * n), m1 = (double)(2 * n + 1);
                nv.x = oldv[b].x * (1.0 - d * m0) + nv.x * d * m0;
                nv.y = oldv[b].y * (1.0 - d * m1) + nv.y * d * m1;
            }
            s[n] = nv;
        }
    }
    __syncthreads();
    lds_fft<T, LOG2L, NT, false>(s, twr);
    const T inv = (T)1.0 / (T)(2 * L);
    if (tid == 0) job.Ynew[0] = mk<T>((s[0].x + s[0].y) * inv, (s[0].x - s[0].y) * inv);
#pragma unroll
    for (int i = 0; i < QU; i++) {
        const int k = 1 + tid + i * NT;
        if (k <= L / 2) {
            c2<T> xk, xlk;
            untangle(s[k], conj(s[L - k]), uw[i], xk, xlk);
            job.Ynew[k] = mk<T>(xk.x * inv, xk.y * inv);
            if (k != L - k) job.Ynew[L - k] = mk<T>(xlk.x * inv, xlk.y * inv);
        }
    }
}

// ------------------------------------------------------------------ K3: spectrum -> raw

// the no-dither requantiser, literally (dither_funs.h:71-114); both precisions go through
// the double version in the reference (fftw_convolver.c:448, :471)
__device__ __forceinline__ int32_t
real2int_no_dither(double v, double rmin, double rmax, int32_t imin, int32_t imax,
                   unsigned int &n_over, int32_t &intlargest, double &largest) {
    int32_t s;
    v += 0.5;
    if (v < 0) {
        if (v <= rmin) {
            s = imin; n_over++;
            if (v < -largest) largest = -v;
        } else {
            s = (int32_t)v; s--;
            if (s < -intlargest) intlargest = -s;
        }
    } else {
        if (v > rmax) {
            s = imax; n_over++;
            if (v > largest) largest = v;
        } else {
            s = (int32_t)v;
            if (s > intlargest) intlargest = s;
        }
    }
    return s;
}

// One output channel's requantiser without dither, the way convolver_cbuf2raw drives it
// (fftw_convolver.c:482-518, real2raw.h:24-250): NaN and safety-limit screening, clipping with the
// overflow bookkeeping of struct bfoverflow, the store in the channel's sample format.  Every
// thread keeps its own counters; reduce() leaves the workgroup's totals with thread 0.
template <typename T> struct Quantiser {
    DevFormat f;
    double of_max, safety, rmin_i, rmax_i, largest;
    int32_t imin, imax, intlargest;
    T rmin_f, rmax_f;
    unsigned int n_over;
    int st;

    __device__ __forceinline__ void init(const DevFormat &fmt, const DevOverflow &of, double safety_limit) {
        f = fmt; of_max = of.max; safety = safety_limit;
        const int bits = f.sbytes << 3;
        imin = (int32_t)(-((uint64_t)1 << (bits - 1)));
        imax = (int32_t)(((uint64_t)1 << (bits - 1)) - 1);
        rmin_i = (double)(T)imin; rmax_i = (double)(T)imax;
        rmin_f = (T)(-of.max); rmax_f = (T)of.max;
        n_over = 0; intlargest = of.intlargest; largest = of.largest; st = 0;
    }
    // false: the sample is left unwritten and a status bit says why (bfrun.c:1925-1935)
    __device__ __forceinline__ bool screen(T x) {
        if (!isfinite(x)) { st |= 1; return false; }
        if (safety != 0.0 && ((double)x < -safety * of_max || (double)x > safety * of_max)) { st |= 2; return false; }
        return true;
    }
    __device__ __forceinline__ int32_t to_int(T x) {
        return real2int_no_dither((double)x, rmin_i, rmax_i, imin, imax, n_over, intlargest, largest);
    }
    // the sample's bytes, little end first
    __device__ __forceinline__ uint64_t encode(T x) {
        if (!f.isfloat) return (uint64_t)(uint32_t)to_int(x);
        if (x < (T)0) {
            if (x < rmin_f) n_over++;
            if ((double)x < -largest) largest = -(double)x;
        } else {
            if (x > rmax_f) n_over++;
            if ((double)x > largest) largest = (double)x;
        }
        return f.bytes == 4 ? (uint64_t)__float_as_uint((float)x) : (uint64_t)__double_as_longlong((double)x);
    }
    __device__ __forceinline__ void put(T x, uint8_t *p) {
        if (screen(x)) store_raw_word(p, encode(x), f.bytes, f.swap);
    }
    // workgroup totals (a count, two maxima, status bits: order independent) into thread 0; every
    // thread of the workgroup (at most 1024) has to call it
    __device__ __forceinline__ void reduce(int tid, int n_threads) {
        __shared__ unsigned int red_n[16];
        __shared__ int32_t red_i[16];
        __shared__ double red_l[16];
        __shared__ int red_s[16];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            n_over += __shfl_down(n_over, off);
            const int32_t oi = __shfl_down(intlargest, off);
            intlargest = oi > intlargest ? oi : intlargest;
            const double ol = __shfl_down(largest, off);
            largest = ol > largest ? ol : largest;
            st |= __shfl_down(st, off);
        }
        const int wave = tid >> 6;
        if ((tid & 63) == 0) { red_n[wave] = n_over; red_i[wave] = intlargest; red_l[wave] = largest; red_s[wave] = st; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < (n_threads + 63) / 64; w++) {
                n_over += red_n[w];
                intlargest = red_i[w] > intlargest ? red_i[w] : intlargest;
                largest = red_l[w] > largest ? red_l[w] : largest;
                st |= red_s[w];
            }
        }
    }
    // thread 0 after reduce(): fold the block's totals into the channel's struct
    __device__ __forceinline__ void commit(DevOverflow &of) const {
        of.n_overflows += n_over; of.intlargest = intlargest; of.largest = largest;
    }
};

// One workgroup per output channel: sum the chunk partials, build Z' = E + iO, inverse
// complex FFT; samples x[2n], x[2n+1] = Re, Im z[n]; the first L samples are the block's
// output (fftw_convolver.c:493-515).  Then cbuf2raw: finite / safety tests, quantise or
// float store, peak + overflow accounting, strided interleaved write.
// `timeout` (may be NULL): if given, the samples are ALSO stored there as T [count][L]
// (used by the dither pass and by debug taps).
template <typename T, int LOG2L, int NTP = fft_threads<T>(LOG2L)>
__device__ __forceinline__ void
ifft_out_body(int zi /* index into Zp's channel axis */, unsigned char *smem,
              const c2<T> *__restrict__ Zp, size_t chunk_stride, int n_chunks,
              int first_channel, const DevFormat *__restrict__ fmt,
              DevOverflow *__restrict__ over, const unsigned char *__restrict__ skip_quant,
              uint8_t *__restrict__ raw, T *__restrict__ timeout,
              const c2<T> *__restrict__ tw, double safety_limit, int *__restrict__ status) {
    constexpr int L = 1 << LOG2L, NT = NTP;
    constexpr int QU = UT<T, LOG2L, NT>::QU;
    LdsArr<T> s{reinterpret_cast<c2<T> *>(smem)};
    const int tid = threadIdx.x;
    const int ch = first_channel + zi;         // output channel
    const c2<T> *z = Zp + (size_t)zi * L;

    // Everything the kernel will need from global memory is requested up front, the spectra
    // first (loads return in order: the sum and the LDS fill then only wait for those), the
    // channel's format and overflow state last (they are needed after the transform).
    c2<T> uw[QU], za[QU], zb[QU], ta[QU], tb[QU];
    c2<T> z0 = mk<T>((T)0, (T)0), t0 = mk<T>((T)0, (T)0);
#pragma unroll
    for (int i = 0; i < QU; i++) {
        const int k = 1 + tid + i * NT;
        if (k <= L / 2) { za[i] = z[k]; zb[i] = z[L - k]; }
    }
    if (tid == 0) z0 = z[0];
    // the second chunk rides along only where the registers allow it (8 bins per thread in
    // float64 at L = 8192 would spill)
    constexpr bool PRELOAD2 = QU * sizeof(c2<T>) <= 64;
    if (PRELOAD2 && n_chunks > 1) {
        const c2<T> *zc = z + chunk_stride;
#pragma unroll
        for (int i = 0; i < QU; i++) {
            const int k = 1 + tid + i * NT;
            if (k <= L / 2) { ta[i] = zc[k]; tb[i] = zc[L - k]; }
        }
        if (tid == 0) t0 = zc[0];
    }
    TwRegs<T, LOG2L, NT> twr;
    twr.prefetch(tw);
#pragma unroll
    for (int i = 0; i < QU; i++) { const int k = 1 + tid + i * NT; uw[i] = tw[k <= L / 2 ? k : 0]; }
    const DevFormat f = fmt[ch];
    DevOverflow of = over[ch];
    const bool quant = skip_quant == nullptr || !skip_quant[ch];
    // chunk partials add up in chunk order (deterministic)
    if (PRELOAD2 && n_chunks > 1) {
        z0 = z0 + t0;
#pragma unroll
        for (int i = 0; i < QU; i++) { za[i] = za[i] + ta[i]; zb[i] = zb[i] + tb[i]; }
    }
    for (int c = PRELOAD2 ? 2 : 1; c < n_chunks; c++) {
        const c2<T> *zc = z + (size_t)c * chunk_stride;
#pragma unroll
        for (int i = 0; i < QU; i++) {
            const int k = 1 + tid + i * NT;
            if (k <= L / 2) { ta[i] = zc[k]; tb[i] = zc[L - k]; }
        }
        if (tid == 0) { t0 = zc[0]; z0 = z0 + t0; }
#pragma unroll
        for (int i = 0; i < QU; i++) { za[i] = za[i] + ta[i]; zb[i] = zb[i] + tb[i]; }
    }
    if (tid == 0) s[0] = mk<T>(z0.x + z0.y, z0.x - z0.y);
#pragma unroll
    for (int i = 0; i < QU; i++) {
        const int k = 1 + tid + i * NT;
        if (k <= L / 2) {
            c2<T> zk, zlk;
            tangle(za[i], conj(zb[i]), uw[i], zk, zlk);
            s[k] = zk;
            if (k != L - k) s[L - k] = zlk;
        }
    }
    __syncthreads();
    lds_fft<T, LOG2L, NT, true>(s, twr);

    uint8_t *base = raw + f.byte_offset;
    const size_t stride = (size_t)f.sample_spacing * f.bytes;
    Quantiser<T> qz;
    qz.init(f, of, safety_limit);

    for (int n = tid; n < L / 2; n += NT) {
        const c2<T> zz = s[n];
        T xs[2] = {zz.x, zz.y};
        if (timeout != nullptr) {
            timeout[(size_t)zi * L + 2 * n] = xs[0];
            timeout[(size_t)zi * L + 2 * n + 1] = xs[1];
        }
        if (!quant) continue;
#pragma unroll
        for (int q = 0; q < 2; q++) qz.put(xs[q], base + (size_t)(2 * n + q) * stride);
    }

    qz.reduce(tid, NT);
    if (tid == 0 && quant) {
        qz.commit(of);
        over[ch] = of;
        if (qz.st) atomicOr(status, qz.st);
    }
}

// Wave-wide reductions for the overflow bookkeeping: four DPP steps (quad permutes, half-row and
// row mirror: VALU only) leave every lane of a 16-lane row with the row's result, two shuffles
// combine the four rows.  Every lane ends up with the wave's result.
__device__ __forceinline__ int dpp_i32(int v, int ctrl_b1_4e_141_140) {
    switch (ctrl_b1_4e_141_140) {
    case 0: return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
    case 1: return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);      // quad_perm [2,3,0,1]
    case 2: return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false);     // row_half_mirror
    default: return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false);    // row_mirror
    }
}
template <typename OP> __device__ __forceinline__ int wave_all_i32(int v, OP op) {
#pragma unroll
    for (int st = 0; st < 4; st++) v = op(v, dpp_i32(v, st));
    v = op(v, __shfl_xor(v, 16));
    return op(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int st = 0; st < 4; st++) {
        const long long q = __double_as_longlong(v);
        const int lo = dpp_i32((int)(q & 0xffffffffll), st), hi = dpp_i32((int)(q >> 32), st);
        const double o = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        v = o > v ? o : v;
    }
    double o = __shfl_xor(v, 16);
    v = o > v ? o : v;
    o = __shfl_xor(v, 32);
    return o > v ? o : v;
}

// K3 on the wave FFT (fft_wave.h; L = 1024 .. 8192): the statement of ifft_out_body with
// the inverse transform's last two passes joined in-wave and the samples quantised straight from
// the registers they end up in.
template <typename T, int LOG2L>
__device__ __forceinline__ void
ifft_out_wave_body(int zi /* index into Zp's channel axis */, unsigned char *smem,
              const c2<T> *__restrict__ Zp, size_t chunk_stride, int n_chunks,
              int first_channel, const DevFormat *__restrict__ fmt,
              DevOverflow *__restrict__ over, const unsigned char *__restrict__ skip_quant,
              uint8_t *__restrict__ raw, T *__restrict__ timeout,
              const c2<T> *__restrict__ tw, double safety_limit, int *__restrict__ status) {
    using G = WaveGeo<LOG2L>;
    constexpr int L = G::L, NT = G::NT, QU = 8;
    LdsArr<T> s{reinterpret_cast<c2<T> *>(smem)};
    const int tid = threadIdx.x;
    const int ch = first_channel + zi;         // output channel
    const c2<T> *z = Zp + (size_t)zi * L;

    // Everything the kernel will need from global memory is requested up front, the spectra
    // first (loads return in order: the sum and the LDS fill then only wait for those), the
    // channel's format and overflow state last (they are needed after the transform).
    c2<T> uw[QU], za[QU], zb[QU], ta[QU], tb[QU];
    c2<T> z0 = mk<T>((T)0, (T)0), t0 = mk<T>((T)0, (T)0);
    BF_PROBE(0);
#pragma unroll
    for (int i = 0; i < QU; i++) {
        const int k = 1 + tid + i * NT;
        if (k <= L / 2) { za[i] = z[k]; zb[i] = z[L - k]; }
    }
    if (tid == 0) z0 = z[0];
    // the second chunk rides along only where the registers allow it (8 bins per thread in
    // float64 at L = 8192 would spill)
    constexpr bool PRELOAD2 = QU * sizeof(c2<T>) <= 64;
    if (PRELOAD2 && n_chunks > 1) {
        const c2<T> *zc = z + chunk_stride;
#pragma unroll
        for (int i = 0; i < QU; i++) {
            const int k = 1 + tid + i * NT;
            if (k <= L / 2) { ta[i] = zc[k]; tb[i] = zc[L - k]; }
        }
        if (tid == 0) t0 = zc[0];
    }
    WaveTw<T, LOG2L> twr;
    // (float64: the 18 pass twiddles are 72 registers -- requested only once the spectra have been
    // folded into LDS, or the kernel spills; their latency then hides behind the barrier and pass 0)
    if constexpr (sizeof(T) == 4) twr.prefetch(tw);
#pragma unroll
    for (int i = 0; i < QU; i++) { const int k = 1 + tid + i * NT; uw[i] = tw[k <= L / 2 ? k : 0]; }
    const DevFormat f = fmt[ch];
    DevOverflow of = over[ch];
    const bool quant = skip_quant == nullptr || !skip_quant[ch];
    BF_PROBE(1);
    // chunk partials add up in chunk order (deterministic)
    if (PRELOAD2 && n_chunks > 1) {
        z0 = z0 + t0;
#pragma unroll
        for (int i = 0; i < QU; i++) { za[i] = za[i] + ta[i]; zb[i] = zb[i] + tb[i]; }
    }
    for (int c = PRELOAD2 ? 2 : 1; c < n_chunks; c++) {
        const c2<T> *zc = z + (size_t)c * chunk_stride;
#pragma unroll
        for (int i = 0; i < QU; i++) {
            const int k = 1 + tid + i * NT;
            if (k <= L / 2) { ta[i] = zc[k]; tb[i] = zc[L - k]; }
        }
        if (tid == 0) { t0 = zc[0]; z0 = z0 + t0; }
#pragma unroll
        for (int i = 0; i < QU; i++) { za[i] = za[i] + ta[i]; zb[i] = zb[i] + tb[i]; }
    }
    if (tid == 0) s[0] = mk<T>(z0.x + z0.y, z0.x - z0.y);
    {
        c2<T> *pk = lds_at(s, 1 + tid), *pl = lds_at(s, L - 1 - tid);     // k and L - k: see fft_in_wave_body
#pragma unroll
        for (int i = 0; i < QU; i++) {
            const int k = 1 + tid + i * NT;
            c2<T> zk, zlk;
            tangle(za[i], conj(zb[i]), uw[i], zk, zlk);
            pk[i * lds_stride(NT)] = zk;
            if (k != L - k) pl[-i * lds_stride(NT)] = zlk;
        }
    }
    BF_PROBE(2);
    if constexpr (sizeof(T) == 8) twr.prefetch(tw);
    __syncthreads();
    wave_p0_lds<T, LOG2L, true>(s);
    BF_PROBE(4);
    c2<T> xr[2][8];
    wave_p123<T, LOG2L, true>(s, twr, xr);
    BF_PROBE(8);
    // xr[b][r] = z[wave_j(tid, b) + r * L/8]; the block's output samples are z[n], n < L/2: r < 4.
    // They go to the quantiser from registers -- no LDS write, no barrier.

    uint8_t *base = raw + f.byte_offset;
    const size_t stride = (size_t)f.sample_spacing * f.bytes;
    Quantiser<T> qz;
    qz.init(f, of, safety_limit);

    // the common raw layout -- integer samples in naturally aligned 32-bit words (S24_4LE, S32_LE,
    // their byte-swapped twins) -- is decided once for the channel: no byte assembly per sample
    const bool word32 = quant && !f.isfloat && f.bytes == 4 && ((((uintptr_t)base) | stride) & 3) == 0;
    const uint32_t stride32 = (uint32_t)stride;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int n = wave_j<LOG2L>(tid, it >> 2) + (it & 3) * G::T8;
        const c2<T> zz = xr[it >> 2][it & 3];
        T xs[2] = {zz.x, zz.y};
        if (timeout != nullptr) {
            timeout[(size_t)zi * L + 2 * n] = xs[0];
            timeout[(size_t)zi * L + 2 * n + 1] = xs[1];
        }
        if (!quant) continue;
        const uint32_t o0 = (uint32_t)(2 * n) * stride32;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const T x = xs[q];
            uint8_t *p = base + (o0 + (q ? stride32 : 0u));
            if (word32) {
                if (!qz.screen(x)) continue;
                uint32_t u = (uint32_t)qz.to_int(x);
                if (f.swap) u = __builtin_bswap32(u);
                *reinterpret_cast<uint32_t *>(p) = u;
            } else qz.put(x, p);
        }
    }

    BF_PROBE(10);
    // The overflow bookkeeping is order independent (a count, two maxima, status bits): every wave
    // reduces its own share in registers and its first lane folds it into the channel's struct
    // with atomics -- no LDS, no barrier behind the stores, and in the steady state (peaks already
    // recorded, nothing clipped) no memory operation at all.
    const unsigned int n_over = (unsigned int)wave_all_i32((int)qz.n_over, [](int a, int b) { return a + b; });
    const int32_t intlargest = wave_all_i32(qz.intlargest, [](int a, int b) { return a > b ? a : b; });
    const double largest = wave_max_f64(qz.largest);
    const int st = wave_all_i32(qz.st, [](int a, int b) { return a | b; });
    if ((tid & 63) == 0 && quant) {
        if (n_over) atomicAdd(&over[ch].n_overflows, n_over);
        if (intlargest > of.intlargest) atomicMax(&over[ch].intlargest, intlargest);
        if (largest > of.largest) atomicMax(&over[ch].largest, largest);
        if (st) atomicOr(status, st);
    }
    BF_PROBE(11);
}

template <typename T, int LOG2L>
__global__ __launch_bounds__(WaveGeo<LOG2L>::NT) void
ifft_out_wave_kernel(const c2<T> *__restrict__ Zp, size_t chunk_stride, int n_chunks, int first_channel,
                     const DevFormat *__restrict__ fmt, DevOverflow *__restrict__ over,
                     const unsigned char *__restrict__ skip_quant, uint8_t *__restrict__ raw,
                     T *__restrict__ timeout, const c2<T> *__restrict__ tw, double safety_limit,
                     int *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ifft_out_wave_body<T, LOG2L>(xcd_channel(blockIdx.x, gridDim.x), smem, Zp, chunk_stride, n_chunks, first_channel, fmt, over,
                                 skip_quant, raw, timeout, tw, safety_limit, status);
}

// K3 of one block and K1 of a later block in ONE launch on the wave FFT (see io_kernel)
template <typename T, int LOG2L>
__global__ __launch_bounds__(WaveGeo<LOG2L>::NT) void
io_wave_kernel(int n_k3,
               const c2<T> *__restrict__ Zp, size_t chunk_stride, int n_chunks, int first_channel,
               const DevFormat *__restrict__ fmt_out, DevOverflow *__restrict__ over,
               const unsigned char *__restrict__ skip_quant, uint8_t *__restrict__ rawout,
               T *__restrict__ timeout, double safety_limit, int *__restrict__ status,
               const uint8_t *__restrict__ rawin, const DevFormat *__restrict__ fmt_in, T *__restrict__ prev,
               c2<T> *__restrict__ ring, int R, int slot, const c2<T> *__restrict__ tw, PowerSave ps,
               const BlockState *__restrict__ bs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (bs) slot = (int)(bs->t % (unsigned int)R);
    if ((int)blockIdx.x < n_k3)
        ifft_out_wave_body<T, LOG2L>(xcd_channel(blockIdx.x, n_k3), smem, Zp, chunk_stride, n_chunks, first_channel, fmt_out, over,
                                     skip_quant, rawout, timeout, tw, safety_limit, status);
    else
        fft_in_wave_body<T, LOG2L>(xcd_channel((int)blockIdx.x - n_k3, (int)gridDim.x - n_k3), smem, rawin, fmt_in, prev, ring, tw, R, slot, ps);
}

template <typename T, int LOG2L, int NTP = fft_threads<T>(LOG2L)>
__global__ __launch_bounds__(NTP) void
ifft_out_kernel(const c2<T> *__restrict__ Zp, size_t chunk_stride, int n_chunks, int first_channel,
                const DevFormat *__restrict__ fmt, DevOverflow *__restrict__ over,
                const unsigned char *__restrict__ skip_quant, uint8_t *__restrict__ raw,
                T *__restrict__ timeout, const c2<T> *__restrict__ tw, double safety_limit,
                int *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ifft_out_body<T, LOG2L, NTP>(xcd_channel(blockIdx.x, gridDim.x), smem, Zp, chunk_stride, n_chunks, first_channel, fmt, over,
                            skip_quant, raw, timeout, tw, safety_limit, status);
}

// K3 of one block and K1 of a later block in ONE launch: the first n_k3 workgroups do the
// inverse transforms, the rest the forward ones.  For a host that pipelines blocks (multi-GPU:
// the mix-down of block t is in flight while block t+1 is computed) the two are independent
// and each is only a handful of workgroups, so running them side by side hides one launch.
template <typename T, int LOG2L>
__global__ __launch_bounds__(fft_threads<T>(LOG2L)) void
io_kernel(int n_k3,
          const c2<T> *__restrict__ Zp, int first_channel, const DevFormat *__restrict__ fmt_out,
          DevOverflow *__restrict__ over, const unsigned char *__restrict__ skip_quant,
          uint8_t *__restrict__ rawout, T *__restrict__ timeout, double safety_limit, int *__restrict__ status,
          const uint8_t *__restrict__ rawin, const DevFormat *__restrict__ fmt_in, T *__restrict__ prev,
          c2<T> *__restrict__ ring, int R, int slot, const c2<T> *__restrict__ tw, PowerSave ps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x < n_k3)
        ifft_out_body<T, LOG2L>(xcd_channel(blockIdx.x, n_k3), smem, Zp, 0, 1, first_channel, fmt_out, over, skip_quant,
                                rawout, timeout, tw, safety_limit, status);
    else
        fft_in_body<T, LOG2L>(xcd_channel((int)blockIdx.x - n_k3, (int)gridDim.x - n_k3), smem, rawin, fmt_in, prev, ring, tw, R, slot, ps);
}

// ------------------------------------------------------------------ N:1 virtual channels: delay, mute, mix

// The reference's integer delay (delay.c:78-340) is a little machine of memcpy/memset between a
// ring of whole-fragment buffers, a rest buffer and two short buffers; which copies happen
// depends on the delay history.  The host mirrors that machine (DelayLine in bfhip.hip) and
// emits, per update, the list of byte moves; the device only executes them, in order.
struct ByteOp {
    uint8_t *dst;
    const uint8_t *src;     // null: zero fill
    unsigned int n;
    unsigned int pad;
};

__device__ __forceinline__ void run_byte_ops(const ByteOp *__restrict__ ops, int n_ops) {
    for (int i = 0; i < n_ops; i++) {
        const ByteOp op = ops[i];
        __syncthreads();                       // previous op's stores are visible (global memory,
        __threadfence_block();                 // same workgroup)
        for (unsigned int b = threadIdx.x; b < op.n; b += blockDim.x) op.dst[b] = op.src ? op.src[b] : (uint8_t)0;
    }
    __syncthreads();
    __threadfence_block();
}

// Input side (bfrun.c:1509-1531): one workgroup per virtual input that shares a physical one.
// Gather its samples from the raw buffer into the private copy (or zero it when muted, in
// which case the delay line is NOT advanced), then run the delay line's moves on the copy.
struct VInJob {
    uint8_t *copy;          // [L * bytes] private contiguous copy, what K1 then reads
    int byte_offset, sample_spacing, bytes, muted;
    int ops_off, n_ops;
};

template <int UNUSED>
__global__ __launch_bounds__(256) void
vchan_in_kernel(const VInJob *__restrict__ jobs, const ByteOp *__restrict__ ops,
                const uint8_t *__restrict__ raw, int L) {
    const VInJob job = jobs[blockIdx.x];
    const unsigned int total = (unsigned int)L * job.bytes;
    if (job.muted) {
        for (unsigned int b = threadIdx.x; b < total; b += blockDim.x) job.copy[b] = 0;
        return;
    }
    const size_t stride = (size_t)job.sample_spacing * job.bytes;
    for (unsigned int b = threadIdx.x; b < total; b += blockDim.x) {
        const unsigned int smp = b / job.bytes, k = b % job.bytes;
        job.copy[b] = raw[job.byte_offset + smp * stride + k];
    }
    run_byte_ops(ops + job.ops_off, job.n_ops);
}

// Output side (bfrun.c:1938-2003): one workgroup per physical output that several virtual
// outputs mix into.  Every member's time samples (from K3) go through its delay line; the
// un-muted ones are added up in channel order (float adds, first one copied); the sum is
// requantised once with the group's shared overflow struct, which is then copied to every
// member.
struct VOutMember { int channel, muted, ops_off, n_ops; };
struct VOutJob { int first_member, n_members, fmt_channel, dither; };   // dither: leave the mix as
                                                                        // reals in row fmt_channel for the dither pass

template <typename T>
__global__ __launch_bounds__(256) void
vchan_out_kernel(const VOutJob *__restrict__ jobs, const VOutMember *__restrict__ members,
                 const ByteOp *__restrict__ ops, T *__restrict__ samples /* [n_out][L] */,
                 const DevFormat *__restrict__ fmt, DevOverflow *__restrict__ over,
                 uint8_t *__restrict__ raw, int L, double safety_limit, int *__restrict__ status) {
    const VOutJob job = jobs[blockIdx.x];
    const VOutMember *mem = members + job.first_member;
    const int tid = threadIdx.x;
    for (int m = 0; m < job.n_members; m++) run_byte_ops(ops + mem[m].ops_off, mem[m].n_ops);

    const int last = mem[job.n_members - 1].channel;
    const DevFormat f = fmt[job.fmt_channel];
    DevOverflow of = over[last];
    uint8_t *base = raw + f.byte_offset;
    const size_t stride = (size_t)f.sample_spacing * f.bytes;
    Quantiser<T> qz;
    qz.init(f, of, safety_limit);
    for (int n = tid; n < L; n += 256) {
        T x = (T)0;
        bool filled = false;
        for (int m = 0; m < job.n_members; m++) {
            if (mem[m].muted) continue;
            const T v = samples[(size_t)mem[m].channel * L + n];
            x = filled ? x + v : v;
            filled = true;
        }
        if (job.dither) { samples[(size_t)job.fmt_channel * L + n] = x; continue; }     // tests + quantiser: dither_kernel
        qz.put(x, base + (size_t)n * stride);
    }
    qz.reduce(tid, 256);
    if (tid == 0) {
        qz.commit(of);
        if (!job.dither) for (int m = 0; m < job.n_members; m++) over[mem[m].channel] = of;     // bfrun.c:1999-2001
        if (qz.st) atomicOr(status, qz.st);
    }
}

// after the dither pass of a shared output: every member carries the group's overflow struct
// (bfrun.c:1999-2001)
template <int UNUSED>
__global__ void vout_spread_overflow_kernel(const VOutJob *__restrict__ jobs, const VOutMember *__restrict__ members,
                                            DevOverflow *__restrict__ over) {
    const VOutJob job = jobs[blockIdx.x];
    if (!job.dither || threadIdx.x != 0) return;
    const DevOverflow of = over[job.fmt_channel];
    for (int m = 0; m < job.n_members; m++) over[members[job.first_member + m].channel] = of;
}

// ------------------------------------------------------------------ sub-sample delay

// delay_subsample_update (delay.c:416-442) runs a small FFT overlap-save (convolver_td_*) over
// the block; what it computes is the causal FIR  y[n] = sum_k h[k] x[n - k]  with h one of 199
// Kaiser-windowed sinc filters of 2*sdf_length+1 taps and the history carried in a `rest`
// buffer.  Here it is evaluated directly: one workgroup per channel, block + history in LDS.
template <typename T> struct SdJob {
    const uint8_t *raw;     // raw samples (input side, converted on the fly) or null
    DevFormat fmt;          // of raw
    const T *src;           // real samples when raw is null (may equal dst)
    T *dst;                 // [L] filtered block
    T *rest;                // [bs] last bs unfiltered samples of the previous block
    const T *taps;          // [flen], null: sub-delay out of range -> block passes unchanged,
                            // history NOT updated (the reference returns early)
};

template <typename T>
__global__ __launch_bounds__(256) void
subdelay_fir_kernel(const SdJob<T> *__restrict__ jobs, int L, int bs, int flen) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T *xx = reinterpret_cast<T *>(smem);          // [bs + L] = [history | block]
    const SdJob<T> job = jobs[blockIdx.x];
    const int tid = threadIdx.x;
    for (int n = tid; n < L; n += 256) {
        T v;
        if (job.raw) v = load_raw<T>(job.raw + job.fmt.byte_offset + (size_t)n * job.fmt.sample_spacing * job.fmt.bytes, job.fmt);
        else v = job.src[n];
        xx[bs + n] = v;
    }
    if (job.taps) for (int n = tid; n < bs; n += 256) xx[n] = job.rest[n];
    __syncthreads();
    if (job.taps == nullptr) {
        for (int n = tid; n < L; n += 256) job.dst[n] = xx[bs + n];
        return;
    }
    for (int n = tid; n < L; n += 256) {
        T acc = (T)0;
        for (int k = 0; k < flen; k++) acc += job.taps[k] * xx[bs + n - k];
        job.dst[n] = acc;
    }
    for (int n = tid; n < bs; n += 256) job.rest[n] = xx[L + n];
}

// ------------------------------------------------------------------ K3d: HP-TPDF dithered requantiser

// dither_funs.h:7-69 + dither.h:28-38.  The error feedback {1,-1} makes sample n depend on
// the quantised sample n-1, so one channel is one sequential chain: one wave per dithered
// channel, lane 0 walks the block, the wave stages samples and table bytes through LDS in
// coalesced chunks.  The reference wraps its table walk by copying the last used byte to
// table[0]; only the wrapping channel ever reads that byte again (as the predecessor of its
// first sample), so the device keeps the table immutable and carries the byte in a register.
template <typename T> struct DitherState { int ptr; int pad; T s0, s1; };

// wave-uniform value of lane `i` (i is uniform): v_readlane, no LDS round trip
__device__ __forceinline__ float lane_value(float v, int i) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i));
}
__device__ __forceinline__ double lane_value(double v, int i) {
    const long long b = __double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), i);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(b >> 32), i);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// The quantiser is an error-feedback loop (dither_funs.h:7-68: sample n needs the errors of n-1
// and n-2), so a channel is a serial chain; everything that is NOT part of the chain is done 64
// samples at a time by the wave: loading the samples, looking up the dither values
// randmap[r[n] - r[n-1]] (their indices depend on n only), the finite / safety tests, and the
// byte stores.  The chain itself runs on wave-uniform values fetched with v_readlane -- no
// memory access inside it -- and lane i keeps sample i's integer.  (The first version walked
// the samples with one lane and a table load per sample: 6 ms per 8192-sample block.)
template <typename T>
__global__ __launch_bounds__(64) void
dither_kernel(const T *__restrict__ samples,          // [n_out][L] from ifft_out_kernel
              const int *__restrict__ channels,       // output channel of each dither slot
              DitherState<T> *__restrict__ state, const int8_t *__restrict__ table, int table_size,
              const T *__restrict__ randmap,          // index -256..255 (centre pointer)
              const DevFormat *__restrict__ fmt, DevOverflow *__restrict__ over,
              uint8_t *__restrict__ raw, int L, double safety_limit, int *__restrict__ status) {
    __shared__ T rmap[512];
    const int slot = blockIdx.x, lane = threadIdx.x;
    const int ch = channels[slot];
    const DevFormat f = fmt[ch];
    const T *x = samples + (size_t)ch * L;
    uint8_t *base = raw + f.byte_offset;
    const size_t stride = (size_t)f.sample_spacing * f.bytes;
    const int bits = f.sbytes << 3;
    const int32_t imin = (int32_t)(-((uint64_t)1 << (bits - 1)));
    const int32_t imax = (int32_t)(((uint64_t)1 << (bits - 1)) - 1);
    const T rmin = (T)imin, rmax = (T)imax;
    for (int i = lane; i < 512; i += 64) rmap[i] = randmap[i - 256];
    __syncthreads();

    const DitherState<T> st = state[slot];
    DevOverflow of = over[ch];
    int ptr = st.ptr;
    const int8_t pred = table[ptr - 1];
    if (ptr + L >= table_size) ptr = 1;                 // dither_preloop_real2int_hp_tpdf
    const int8_t *tab = table + ptr;
    T s0 = st.s0, s1 = st.s1;
    unsigned int n_over = of.n_overflows;
    int32_t intlargest = of.intlargest;
    double largest = of.largest;
    const double lim = safety_limit * of.max;
    int flags = 0;

    for (int c0 = 0; c0 < L; c0 += 64) {
        const int n = c0 + lane;
        const bool valid = n < L;
        const T xv = valid ? x[n] : (T)0;
        const int r = valid ? (int)tab[n] : 0;
        const int rp = n == 0 ? (int)pred : (valid ? (int)tab[n - 1] : 0);
        const T dth = rmap[256 + r - rp];
        int skip = valid ? 0 : 1;
        if (valid && !isfinite(xv)) { flags |= 1; skip = 1; }
        else if (valid && safety_limit != 0.0 && ((double)xv < -lim || (double)xv > lim)) { flags |= 2; skip = 1; }
        int32_t myq = 0;
        T myfb = (T)0;
#pragma unroll 8
        for (int i = 0; i < 64; i++) {
            const T xi = lane_value(xv, i), di = lane_value(dth, i);
            const bool sk = __builtin_amdgcn_readlane(skip, i) != 0;
            // dither_funs.h:21-66 on uniform values; only what the next sample needs is in here
            const T fb = s0 - s1;
            const T v = xi + fb;
            const T dv = v + di;
            const bool neg = dv < (T)0;
            const bool clip = neg ? (dv <= rmin) : (dv > rmax);
            int32_t q = (int32_t)(clip ? (T)0 : dv);
            q = neg ? q - 1 : q;
            q = clip ? (neg ? imin : imax) : q;
            const T e0 = v - (T)q;
            s1 = sk ? s1 : s0;
            s0 = sk ? s0 : e0;
            if (lane == i) { myq = q; myfb = fb; }
        }
        // the bookkeeping of dither_funs.h:33-60 from each lane's own sample: the overflow count
        // and the largest integer are order independent; `largest` (compares the undithered,
        // stores the dithered value) is walked in sample order over the clipped samples only
        {
            const T v = xv + myfb;
            const T dv = v + dth;
            const bool neg = dv < (T)0;
            const bool clip = !skip && (neg ? (dv <= rmin) : (dv > rmax));
            int32_t mag = (skip || clip) ? 0 : (neg ? -myq : myq);
            for (int off = 32; off > 0; off >>= 1) { const int32_t o = __shfl_xor(mag, off); mag = o > mag ? o : mag; }
            if (mag > intlargest) intlargest = mag;
            unsigned long long cm = __ballot(clip);
            n_over += (unsigned int)__popcll(cm);
            while (cm) {
                const int i = __ffsll((long long)cm) - 1;
                cm &= cm - 1;
                const double vi = (double)lane_value(v, i), dvi = (double)lane_value(dv, i);
                if (dvi < 0) { if (vi < -largest) largest = -dvi; }
                else { if (vi > largest) largest = dvi; }
            }
        }
        if (!skip) {
            const uint32_t u = (uint32_t)myq;
            store_raw_word(base + (size_t)n * stride, (uint64_t)u, f.bytes, f.swap);
        }
    }
    for (int off = 32; off > 0; off >>= 1) flags |= __shfl_down(flags, off);
    if (lane == 0) {
        DitherState<T> out = st;
        out.ptr = ptr + L;
        out.s0 = s0; out.s1 = s1;
        state[slot] = out;
        over[ch].n_overflows = n_over;
        over[ch].intlargest = intlargest;
        over[ch].largest = largest;
        if (flags) atomicOr(status, flags);
    }
}

// ------------------------------------------------------------------ real-time tail

// Host <-> device staging of a period as kernels (16 bytes per lane, whole PCIe bursts) rather
// than copy-engine nodes: inside a replayed graph a kernel node costs less than a memcpy node,
// and the last one doubles as the block's tail.
struct RtCopy {
    uint4 *dst;
    const uint4 *src;
    unsigned int n16;        // 16-byte words (buffers are padded to a multiple of 16 bytes)
    unsigned int pad;
};

// Wide interleaved sides (hundreds of channels in one frame): a transform workgroup owns ONE channel
// and would gather its L samples at the stride of a whole frame -- every 4-byte sample in a cache line
// of its own, every line fetched by as many workgroups as it holds channels (256 channels of S24_4LE:
// 2 M line requests for 8 MiB of samples).  Instead the frames are transposed once, coalesced both ways
// through a 64 x 64 tile in LDS, into a planar copy the transforms read (and, on the output side, write)
// contiguously.  Words are moved, not converted: the sample conversion stays where it is.
//   to_planar:      src = frames [rows][cols]  ->  dst = planar [cols][rows]
//   else:           src = planar [cols][rows]  ->  dst = frames [rows][cols], columns [first, first+count)
//                   whose mask byte (may be null) is 0 only: an engine that runs a shard of the
//                   configuration leaves the other engines' channels alone
template <typename W>
__global__ __launch_bounds__(256) void
transpose_words_kernel(const W *__restrict__ src, W *__restrict__ dst, int rows, int cols, int to_planar,
                       const unsigned char *__restrict__ skip, int first, int count) {
    // 16 bytes per lane on both sides of the tile: Q words in a row of the frame (read) or in a row of
    // the planar copy (write), whenever the tile is whole and both pointers allow it
    constexpr int Q = 16 / sizeof(W), TR = 64, TC = 64;
    __shared__ W tile[TR][TC + 1];
    const int c0 = blockIdx.x * TC, r0 = blockIdx.y * TR;
    const int tid = threadIdx.x;
    const bool whole = r0 + TR <= rows && c0 + TC <= cols && (rows % Q) == 0 && (cols % Q) == 0 &&
                       ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0;
    struct alignas(16) Vec { W w[Q]; };
    if (to_planar) {
        if (whole) {
            // frames -> tile[r][c]: a lane takes Q consecutive channels of one frame
            for (int i = tid; i < TR * (TC / Q); i += 256) {
                const int r = i / (TC / Q), cq = (i % (TC / Q)) * Q;
                const Vec v = *reinterpret_cast<const Vec *>(src + (size_t)(r0 + r) * cols + c0 + cq);
#pragma unroll
                for (int q = 0; q < Q; q++) tile[r][cq + q] = v.w[q];
            }
            __syncthreads();
            // tile -> planar: a lane takes Q consecutive frames of one channel
            for (int i = tid; i < TC * (TR / Q); i += 256) {
                const int c = i / (TR / Q), rq = (i % (TR / Q)) * Q;
                Vec v;
#pragma unroll
                for (int q = 0; q < Q; q++) v.w[q] = tile[rq + q][c];
                *reinterpret_cast<Vec *>(dst + (size_t)(c0 + c) * rows + r0 + rq) = v;
            }
            return;
        }
        for (int i = tid; i < TR * TC; i += 256) {
            const int r = r0 + i / TC, c = c0 + i % TC;
            if (r < rows && c < cols) tile[i / TC][i % TC] = src[(size_t)r * cols + c];
        }
        __syncthreads();
        for (int i = tid; i < TR * TC; i += 256) {
            const int c = c0 + i / TR, r = r0 + i % TR;
            if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[i % TR][i / TR];
        }
        return;
    }
    // planar -> frames; only columns [first, first + count) that are not another engine's
    bool all_mine = whole && c0 >= first && c0 + TC <= first + count;
    if (all_mine && skip != nullptr)
        for (int c = 0; c < TC; c++) all_mine = all_mine && !skip[c0 + c];
    if (all_mine) {
        for (int i = tid; i < TC * (TR / Q); i += 256) {
            const int c = i / (TR / Q), rq = (i % (TR / Q)) * Q;
            const Vec v = *reinterpret_cast<const Vec *>(src + (size_t)(c0 + c) * rows + r0 + rq);
#pragma unroll
            for (int q = 0; q < Q; q++) tile[rq + q][c] = v.w[q];
        }
        __syncthreads();
        for (int i = tid; i < TR * (TC / Q); i += 256) {
            const int r = i / (TC / Q), cq = (i % (TC / Q)) * Q;
            Vec v;
#pragma unroll
            for (int q = 0; q < Q; q++) v.w[q] = tile[r][cq + q];
            *reinterpret_cast<Vec *>(dst + (size_t)(r0 + r) * cols + c0 + cq) = v;
        }
        return;
    }
    for (int i = tid; i < TR * TC; i += 256) {
        const int c = c0 + i / TR, r = r0 + i % TR;
        if (r < rows && c < cols) tile[i % TR][i / TR] = src[(size_t)c * rows + r];
    }
    __syncthreads();
    for (int i = tid; i < TR * TC; i += 256) {
        const int r = r0 + i / TC, c = c0 + i % TC;
        const bool mine = c < cols && c >= first && c < first + count && (skip == nullptr || !skip[c]);
        if (r < rows && mine) dst[(size_t)r * cols + c] = tile[i / TC][i % TC];
    }
}

template <int UNUSED>
__global__ __launch_bounds__(256) void rt_copy_in_kernel(RtCopy c) {
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    if (i < c.n16) c.dst[i] = c.src[i];
}

// Last node of a replayed block: copy the raw output to pinned host memory, publish the
// overflow structs and the status word there too (so the host needs no further copy after the
// sync) and advance the per-block counters.  The completion word is written by the last
// workgroup to finish, after a system-scope fence.
template <int UNUSED>
__global__ __launch_bounds__(256) void
rt_tail_kernel(RtCopy c, BlockState *__restrict__ bs, int N, const DevOverflow *__restrict__ over,
               DevOverflow *__restrict__ host_over, int n_out, int *__restrict__ status,
               int *__restrict__ host_status, unsigned int *__restrict__ arrive) {
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    if (i < c.n16) c.dst[i] = c.src[i];
    if (blockIdx.x == 0) {
        for (int ch = threadIdx.x; ch < n_out; ch += 256) host_over[ch] = over[ch];
        if (threadIdx.x == 0) {
            host_status[0] = *status;
            *status = 0;
            unsigned int tn = bs->t + 1u;                    // bfrun.c:2034
            if (bs->wrap_by != 0u && tn >= bs->wrap_at) tn -= bs->wrap_by;
            bs->t = tn;
            bs->age = bs->age < N ? bs->age + 1 : N;
            bs->n_blocks = bs->n_blocks + 1u;
        }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int done = atomicAdd(arrive, 1u) + 1u;
        if (done == gridDim.x) {
            *arrive = 0;
            __hip_atomic_store(host_status + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // "period done"
        }
    }
}

}  // namespace bfhip
