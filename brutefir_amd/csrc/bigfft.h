// bigfft.h -- partition lengths above the LDS limit (L = 16384 ... 1048576).
//
// BruteFIR's stock configuration is ONE partition of 65536 taps (`filter_length: 65536;`,
// bfconf.c:197, bench3_config) and any power of two is legal (bfconf.c:1512-1514).  A complex
// FFT of more than 8192 points does not fit a CU's LDS, so for these lengths every
// FFT-bearing kernel of kernels.h is replaced by a short sequence over global scratch buffers:
//
//     pre   (elementwise: window / tangle / mix ...  -> zin[transform][L])
//     A     R = L/8192 LDS transforms of M = 8192 points per transform, on the decimated
//           subsequences z[r + R m]                        -> [transform][r][k]
//     B     combine passes of radix <= 8 (one for R <= 8, up to three for R = 128), each
//           merging Rp sub-transforms of length Mc into one of length Mc Rp with twiddles
//           W_(Mc Rp)^(j k)                                 -> zout[transform][L]
//     post  (elementwise: untangle / quantise / ramp ...)
//
// (decimation in time: X[k + Mc q] = sum_j W_Rp^(j q) (W_(Mc Rp)^(j k) Y_j[k]); with r = r1 + R1 r2
// the first pass merges over the high digit r2, the next over r1, ...).
// The block period at these lengths is a third of a second and more; the extra launches and
// the global round trips do not matter, the MAC (kernels.h, any L) still dominates.
// Arithmetic and bookkeeping are the same statements as in the LDS kernels, so every feature
// (cascades, cross-fades, dither, N:1 channels ...) keeps working unchanged.
#pragma once
#include "kernels.h"

namespace bfhip {

constexpr int BIG_LOG2M = 13;
constexpr int BIG_M = 1 << BIG_LOG2M;

// ---- stage A: R x n_tr workgroups, each one M-point LDS FFT of a decimated subsequence
template <typename T, bool INV>
__global__ __launch_bounds__(fft_threads<T>(BIG_LOG2M)) void
big_fft_a(const c2<T> *__restrict__ zin, c2<T> *__restrict__ zmid, int R, const c2<T> *__restrict__ tw13) {
    constexpr int NT = fft_threads<T>(BIG_LOG2M), M = BIG_M;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsArr<T> s{reinterpret_cast<c2<T> *>(smem)};
    const int r = blockIdx.x, tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.y * M * R;
    TwRegs<T, BIG_LOG2M, NT> twr;
    for (int m = tid; m < M; m += NT) s[m] = zin[base + (size_t)r + (size_t)R * m];
    twr.prefetch(tw13);
    __syncthreads();
    lds_fft<T, BIG_LOG2M, NT, INV>(s, twr);
    for (int k = tid; k < M; k += NT) zmid[base + (size_t)r * M + k] = s[k];
}

// ---- stage B, one combine pass: S = Gp * RP sub-transforms of length Mc (sub s at s * Mc) become Gp
// of length Mc * RP; sub-transform g of the result merges the inputs s = g + Gp * j, j < RP.
// One thread per (g, k).  tw_stride = L / (Mc * RP): W_(Mc RP)^(j k) = twL[2 j k tw_stride].
template <typename T, bool INV, int RP>
__global__ __launch_bounds__(256) void
big_fft_b(const c2<T> *__restrict__ zin, c2<T> *__restrict__ zout, const c2<T> *__restrict__ twL,
          int Mc, int Gp, int tw_stride, size_t L) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // < Gp * Mc
    const int g = i / Mc, k = i - g * Mc;
    const size_t base = (size_t)blockIdx.y * L;
    c2<T> u[RP];
#pragma unroll
    for (int j = 0; j < RP; j++) {
        u[j] = zin[base + (size_t)(g + Gp * j) * Mc + k];
        if (j > 0) {
            c2<T> w = twL[(size_t)2 * j * k * tw_stride];
            if (INV) w.y = -w.y;
            u[j] = cmul(u[j], w);
        }
    }
    dftR<T, INV, RP>(u);
#pragma unroll
    for (int q = 0; q < RP; q++) zout[base + (size_t)g * Mc * RP + k + (size_t)Mc * q] = u[q];
}

// ---- host: the complete complex FFT of 2^log2L points for n_tr transforms, zin -> zout.
// zmid and zout are used alternately by the combine passes so that the last one lands in zout.
template <typename T, bool INV, int RP>
inline void big_fft_launch_b(const c2<T> *in, c2<T> *out, const c2<T> *twL, int Mc, int Gp, int tw_stride,
                             size_t L, int n_tr, hipStream_t st) {
    hipLaunchKernelGGL((big_fft_b<T, INV, RP>), dim3((unsigned)((size_t)Mc * Gp / 256), (unsigned)n_tr), dim3(256), 0, st,
                       in, out, twL, Mc, Gp, tw_stride, L);
}

template <typename T, bool INV>
inline hipError_t big_fft_run(const c2<T> *zin, c2<T> *zmid, c2<T> *zout, int log2L, int n_tr,
                              const c2<T> *tw13, const c2<T> *twL, hipStream_t st) {
    constexpr int NT = fft_threads<T>(BIG_LOG2M);
    const size_t lds = lds_fft_bytes(BIG_LOG2M, sizeof(c2<T>));
    const size_t L = (size_t)1 << log2L;
    const int R = (int)(L / BIG_M);
    int bits = log2L - BIG_LOG2M, passes = (bits + 2) / 3;
    auto ka = big_fft_a<T, INV>;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(ka), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    c2<T> *buf[2] = {zout, zmid};
    int cur = passes & 1;                              // after `passes` flips the data is in buf[0] = zout
    hipLaunchKernelGGL(ka, dim3(R, n_tr), dim3(NT), lds, st, zin, buf[cur], R, tw13);
    int Mc = BIG_M, S = R;
    while (bits > 0) {
        const int b = bits >= 3 ? 3 : bits, RP = 1 << b, Gp = S / RP;
        const int tw_stride = (int)(L / ((size_t)Mc * RP));
        if (RP == 8) big_fft_launch_b<T, INV, 8>(buf[cur], buf[cur ^ 1], twL, Mc, Gp, tw_stride, L, n_tr, st);
        else if (RP == 4) big_fft_launch_b<T, INV, 4>(buf[cur], buf[cur ^ 1], twL, Mc, Gp, tw_stride, L, n_tr, st);
        else big_fft_launch_b<T, INV, 2>(buf[cur], buf[cur ^ 1], twL, Mc, Gp, tw_stride, L, n_tr, st);
        cur ^= 1; Mc *= RP; S = Gp; bits -= b;
    }
    return hipGetLastError();
}

// ---- K1 ------------------------------------------------------------------------------------

// window [previous L | new L] packed as z[n] = x[2n] + i x[2n+1] (fft_in_body's first half)
// `powersave:` at these lengths: the window of a channel is spread over many workgroups, so the
// silence test (PowerSave, kernels.h) is an atomic maximum over the samples' bit patterns -- all
// bits for the exact test (-0.0 counts as non-zero, like memiszero), the magnitude for the noise
// floor -- into acc[parity][ch]; the other parity's word is cleared for the next block.
template <typename T>
__global__ __launch_bounds__(256) void
big_in_pre(const uint8_t *__restrict__ raw, const DevFormat *__restrict__ fmt, T *__restrict__ prev,
           c2<T> *__restrict__ zin, int L, unsigned long long *__restrict__ ps_acc, int n_in, int parity,
           int ps_exact) {
    const int n = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
    unsigned long long m = 0;
    if (n < L / 2) {
        const DevFormat f = fmt[ch];
        c2<T> *pv = reinterpret_cast<c2<T> *>(prev + (size_t)ch * L);
        const uint8_t *base = f.alt ? f.alt : raw + f.byte_offset;
        const size_t stride = (size_t)f.sample_spacing * f.bytes;
        const c2<T> cur = mk<T>(load_raw<T>(base + (size_t)(2 * n) * stride, f),
                                load_raw<T>(base + (size_t)(2 * n + 1) * stride, f));
        const c2<T> old = pv[n];
        zin[(size_t)ch * L + n] = old;
        zin[(size_t)ch * L + L / 2 + n] = cur;
        pv[n] = cur;
        if (ps_acc) {
            const T v4[4] = {old.x, old.y, cur.x, cur.y};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                unsigned long long b;
                if constexpr (sizeof(T) == 4) b = __float_as_uint(v4[q]) & (ps_exact ? 0xffffffffu : 0x7fffffffu);
                else b = (unsigned long long)__double_as_longlong(v4[q]) & (ps_exact ? ~0ull : 0x7fffffffffffffffull);
                m = b > m ? b : m;
            }
        }
    }
    if (ps_acc) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(m, off); m = o > m ? o : m; }
        if ((threadIdx.x & 63) == 0 && m) atomicMax(ps_acc + (size_t)parity * n_in + ch, m);
        if (n == 0) ps_acc[(size_t)(parity ^ 1) * n_in + ch] = 0;
    }
}

// after the spectrum has been written: a silent window's spectrum is zero (bfrun.c:1541-1553)
template <typename T>
__global__ __launch_bounds__(256) void
big_ps_finish(const unsigned long long *__restrict__ ps_acc, int n_in, int parity, PowerSave ps,
              c2<T> *__restrict__ ring, int R, int slot, int L) {
    const int k = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.y;
    const unsigned long long m = ps_acc[(size_t)parity * n_in + ch];
    bool silent;
    if (ps.thr >= 1.0) silent = m == 0;
    else {
        double mag;
        if constexpr (sizeof(T) == 4) mag = (double)__uint_as_float((unsigned int)m);
        else mag = __longlong_as_double((long long)m);
        silent = !(ps.scale[ch] * mag >= ps.thr);
    }
    if (k == 0) {
        const int was = ps.flags[ch * R + slot], now = silent ? 1 : 0;
        ps.flags[ch * R + slot] = now;
        if (was != now) ps.live[ch] += was - now;
    }
    if (silent && k < L) ring[((size_t)ch * R + slot) * L + k] = mk<T>((T)0, (T)0);
}

// complex FFT of the packed window -> packed spectrum of the 2L real samples, times `scale`;
// transform t goes to dst0 + t * dst_stride (elements)
template <typename T>
__global__ __launch_bounds__(256) void
big_untangle(const c2<T> *__restrict__ zout, c2<T> *__restrict__ dst0, size_t dst_stride,
             const c2<T> *__restrict__ twL, int L, T scale) {
    const int k = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
    if (k > L / 2) return;
    const c2<T> *s = zout + (size_t)t * L;
    c2<T> *out = dst0 + (size_t)t * dst_stride;
    if (k == 0) { out[0] = mk<T>((s[0].x + s[0].y) * scale, (s[0].x - s[0].y) * scale); return; }
    c2<T> xk, xlk;
    untangle(s[k], conj(s[L - k]), twL[k], xk, xlk);
    out[k] = mk<T>(xk.x * scale, xk.y * scale);
    if (k != L - k) out[L - k] = mk<T>(xlk.x * scale, xlk.y * scale);
}

// ---- K7 ------------------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(256) void
big_coeff_pre(const T *__restrict__ taps, int n_taps, T scale, c2<T> *__restrict__ zin, int L,
              int *__restrict__ bad) {
    const int n = blockIdx.x * 256 + threadIdx.x, part = blockIdx.y;
    if (n >= L / 2) return;
    const long i0 = (long)part * L + 2 * n, i1 = i0 + 1;
    const T a = i0 < n_taps ? taps[i0] * scale : (T)0;
    const T b = i1 < n_taps ? taps[i1] * scale : (T)0;
    if (!isfinite(a) || !isfinite(b)) atomicOr(bad, 1);
    zin[(size_t)part * L + n] = mk<T>((T)0, (T)0);
    zin[(size_t)part * L + L / 2 + n] = mk<T>(a, b);
}

// ---- K3 ------------------------------------------------------------------------------------

// sum of the chunk partials (in chunk order) + C2R pre-pass (ifft_out_body's first half)
template <typename T>
__global__ __launch_bounds__(256) void
big_out_pre(const c2<T> *__restrict__ Zp, size_t chunk_stride, int n_chunks, c2<T> *__restrict__ zin,
            const c2<T> *__restrict__ twL, int L) {
    const int k = blockIdx.x * 256 + threadIdx.x, zi = blockIdx.y;
    if (k > L / 2) return;
    const c2<T> *z = Zp + (size_t)zi * L;
    c2<T> *s = zin + (size_t)zi * L;
    c2<T> a = z[k], b = z[k == 0 ? 0 : L - k];
    for (int c = 1; c < n_chunks; c++) {
        const c2<T> *zc = z + (size_t)c * chunk_stride;
        a = a + zc[k];
        b = b + zc[k == 0 ? 0 : L - k];
    }
    if (k == 0) { s[0] = mk<T>(a.x + a.y, a.x - a.y); return; }
    c2<T> zk, zlk;
    tangle(a, conj(b), twL[k], zk, zlk);
    s[k] = zk;
    if (k != L - k) s[L - k] = zlk;
}

// cbuf2raw of the first L samples (pairs zout[n], n < L/2): the statements of ifft_out_body's
// second half, one workgroup per output channel
template <typename T>
__global__ __launch_bounds__(1024) void
big_out_post(const c2<T> *__restrict__ zout, int first_channel, const DevFormat *__restrict__ fmt,
             DevOverflow *__restrict__ over, const unsigned char *__restrict__ skip_quant,
             uint8_t *__restrict__ raw, T *__restrict__ timeout, int L, double safety_limit,
             int *__restrict__ status) {
    constexpr int NT = 1024;
    const int zi = blockIdx.x, tid = threadIdx.x, ch = first_channel + zi;
    const c2<T> *s = zout + (size_t)zi * L;
    const DevFormat f = fmt[ch];
    DevOverflow of = over[ch];
    const bool quant = skip_quant == nullptr || !skip_quant[ch];
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

// ---- K4/K5: ring fill (see ring_fill_kernel) ---------------------------------------------------

// M = sum_g fscale_g Y_g, tangled, for the jobs that have filter inputs
template <typename T>
__global__ __launch_bounds__(256) void
big_fill_pre(const FillJob<T> *__restrict__ jobs, const MixSrc<T> *__restrict__ src,
             c2<T> *__restrict__ zin, const c2<T> *__restrict__ twL, int L) {
    const int k = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (k > L / 2) return;
    const FillJob<T> job = jobs[j];
    c2<T> *s = zin + (size_t)j * L;
    if (job.n_up <= 0) {                                    // nothing to evaluate: keep the buffers defined
        s[k] = mk<T>((T)0, (T)0);
        if (k != 0 && k != L - k) s[L - k] = mk<T>((T)0, (T)0);
        return;
    }
    const MixSrc<T> *up = src + job.up_off;
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
    if (k == 0) { s[0] = mk<T>(a.x + a.y, a.x - a.y); return; }
    c2<T> zk, zlk;
    tangle(a, conj(b), twL[k], zk, zlk);
    s[k] = zk;
    if (k != L - k) s[L - k] = zlk;
}

// convolver_convolve_eval's slide: window = [previous valid half | new valid half]
template <typename T>
__global__ __launch_bounds__(256) void
big_fill_slide(const FillJob<T> *__restrict__ jobs, const c2<T> *__restrict__ zout, c2<T> *__restrict__ zin, int L) {
    const int n = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (n >= L / 2) return;
    const FillJob<T> job = jobs[j];
    if (job.n_up <= 0) return;
    c2<T> *ep = reinterpret_cast<c2<T> *>(job.evalprev);
    const c2<T> v = zout[(size_t)j * L + n];
    zin[(size_t)j * L + L / 2 + n] = v;
    zin[(size_t)j * L + n] = ep[n];
    ep[n] = v;
}

// ring[(t + delay) mod N] = sum_i scale_i X_i[t] + E   (channel inputs first, evaluated buffer last)
template <typename T>
__global__ __launch_bounds__(256) void
big_fill_post(const FillJob<T> *__restrict__ jobs, const MixSrc<T> *__restrict__ src,
              const c2<T> *__restrict__ zout, const c2<T> *__restrict__ twL, int N, unsigned int t,
              int L, const BlockState *__restrict__ bs) {
    const int k = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (k > L / 2) return;
    if (bs) t = bs->t;
    const FillJob<T> job = jobs[j];
    c2<T> *dst = job.ring + (size_t)((t + (unsigned int)job.delay) % (unsigned int)N) * L;
    const MixSrc<T> *in = src + job.in_off;
    const c2<T> *s = zout + (size_t)j * L;
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
        else untangle(s[k], conj(s[L - k]), twL[k], ek, ek2);
        if (first) { a = ek; b = ek2; } else { a = a + ek; b = b + ek2; }
    }
    dst[k] = a;
    if (k2 >= 0) dst[k2] = b;
}

// ---- K6: crossfade (see crossfade_kernel) --------------------------------------------------------

// transform 2j = old-coefficient result, 2j + 1 = new-coefficient result, both tangled
template <typename T>
__global__ __launch_bounds__(256) void
big_fade_pre(const FadeJob<T> *__restrict__ jobs, c2<T> *__restrict__ zin, const c2<T> *__restrict__ twL, int L) {
    const int k = blockIdx.x * 256 + threadIdx.x, tr = blockIdx.y;
    if (k > L / 2) return;
    const FadeJob<T> job = jobs[tr >> 1];
    const c2<T> *z = (tr & 1) ? (const c2<T> *)job.Ynew : job.Yold;
    c2<T> *s = zin + (size_t)tr * L;
    if (k == 0) { const c2<T> a = z[0]; s[0] = mk<T>(a.x + a.y, a.x - a.y); return; }
    c2<T> zk, zlk;
    tangle(z[k], conj(z[L - k]), twL[k], zk, zlk);
    s[k] = zk;
    if (k != L - k) s[L - k] = zlk;
}

// the linear ramp over the first L samples, float-branch arithmetic (fftw_convolver.c:349-355)
template <typename T>
__global__ __launch_bounds__(256) void
big_fade_mix(const c2<T> *__restrict__ zout, c2<T> *__restrict__ zin, int L) {
    const int n = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (n >= L) return;
    if (n >= L / 2) { zin[(size_t)j * L + n] = zout[(size_t)(2 * j + 1) * L + n]; return; }   // second half: as computed
    const c2<T> ov = zout[(size_t)(2 * j) * L + n];
    c2<T> nv = zout[(size_t)(2 * j + 1) * L + n];
    if constexpr (sizeof(T) == 4) {
        const float f = 1.0f / (float)(L - 1);
        const float m0 = (float)(2 * n), m1 = (float)(2 * n + 1);
        nv.x = (float)((double)ov.x * (1.0 - (double)(f * m0)) + (double)(nv.x * f * m0));
        nv.y = (float)((double)ov.y * (1.0 - (double)(f * m1)) + (double)(nv.y * f * m1));
    } else {
        const double d = 1.0 / (double)(L - 1);
        const double m0 = (double)(2 * n), m1 = (double)(2 * n + 1);
        nv.x = ov.x * (1.0 - d * m0) + nv.x * d * m0;
        nv.y = ov.y * (1.0 - d * m1) + nv.y * d * m1;
    }
    zin[(size_t)j * L + n] = nv;
}

// back in the frequency domain: untangle, / n_fft, into the job's Ynew
template <typename T>
__global__ __launch_bounds__(256) void
big_fade_post(const FadeJob<T> *__restrict__ jobs, const c2<T> *__restrict__ zout,
              const c2<T> *__restrict__ twL, int L) {
    const int k = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (k > L / 2) return;
    const c2<T> *s = zout + (size_t)j * L;
    c2<T> *out = jobs[j].Ynew;
    const T inv = (T)1.0 / (T)(2 * L);
    if (k == 0) { out[0] = mk<T>((s[0].x + s[0].y) * inv, (s[0].x - s[0].y) * inv); return; }
    c2<T> xk, xlk;
    untangle(s[k], conj(s[L - k]), twL[k], xk, xlk);
    out[k] = mk<T>(xk.x * inv, xk.y * inv);
    if (k != L - k) out[L - k] = mk<T>(xlk.x * inv, xlk.y * inv);
}

}  // namespace bfhip
