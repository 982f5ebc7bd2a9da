// host_fft.h -- a small real FFT for the HOST side of the convolver.h boundary.
//
// Used only by host_ops.cpp, i.e. by the entry points that run outside the per-block loop in
// processes that must not own a HIP context: convolver_coeffs2cbuf in bfconf's parent before the
// fork (bfconf.c:1979-2019), convolver_runtime_coeffs2cbuf / the fftplan handle in bflogic_eq's
// process (rendereq.h:66-91), convolver_td_new from delay.c at start-up, debug dumps.  The
// reference runs FFTW there (fftw_convolver.c:526-596, 624-680, 698-736); this is the same
// transform pair by definition -- FFTW's R2HC / HC2R (halfcomplex: r0..r_n/2, i_(n/2-1)..i_1),
// unnormalised, forward sign e^{-j} -- computed in the working precision with twiddles rounded
// once from double.  The block loop never comes here: its transforms are the LDS kernels.
#pragma once
#include <cmath>
#include <complex>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace bfhost {

template <typename T> struct CFFT {
    int log2n = 0;
    size_t n = 1;
    std::vector<std::complex<T>> w;      // exp(-2 pi i k / n), k < n/2
    std::vector<std::complex<T>> wr;     // exp(-2 pi i k / (2n)), k <= n/2: real-transform (un)tangling
    std::vector<uint32_t> rev;

    explicit CFFT(int lg) : log2n(lg), n((size_t)1 << lg) {
        w.resize(n / 2 ? n / 2 : 1);
        for (size_t k = 0; k < n / 2; k++) {
            const double a = -2.0 * M_PI * (double)k / (double)n;
            w[k] = std::complex<T>((T)std::cos(a), (T)std::sin(a));
        }
        wr.resize(n / 2 + 1);
        for (size_t k = 0; k <= n / 2; k++) {
            const double a = -M_PI * (double)k / (double)n;
            wr[k] = std::complex<T>((T)std::cos(a), (T)std::sin(a));
        }
        rev.resize(n);
        for (size_t i = 0; i < n; i++) {
            uint32_t r = 0;
            for (int b = 0; b < lg; b++) if (i & ((size_t)1 << b)) r |= 1u << (lg - 1 - b);
            rev[i] = r;
        }
    }

    // in place, natural order in and out; inv: conjugate twiddles, no 1/n
    void run(std::complex<T> *a, bool inv) const {
        for (size_t i = 0; i < n; i++) if (rev[i] > i) std::swap(a[i], a[rev[i]]);
        for (size_t half = 1; half < n; half <<= 1) {
            const size_t step = n / (2 * half);
            for (size_t base = 0; base < n; base += 2 * half) {
                for (size_t j = 0; j < half; j++) {
                    std::complex<T> tw = w[j * step];
                    if (inv) tw = std::conj(tw);
                    const std::complex<T> u = a[base + j];
                    const std::complex<T> x = a[base + j + half];
                    const std::complex<T> v(x.real() * tw.real() - x.imag() * tw.imag(),
                                            x.real() * tw.imag() + x.imag() * tw.real());
                    a[base + j] = u + v;
                    a[base + j + half] = u - v;
                }
            }
        }
    }
};

template <typename T> const CFFT<T> &cfft(int lg) {
    static std::mutex mu;
    static std::map<int, std::unique_ptr<CFFT<T>>> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto &p = cache[lg];
    if (!p) p.reset(new CFFT<T>(lg));
    return *p;
}

// FFTW_R2HC of n = 2^order reals (order >= 1); in may equal out
template <typename T> void r2hc(int order, const T *in, T *out) {
    const size_t n = (size_t)1 << order, L = n / 2;
    if (L == 1) { const T a = in[0], b = in[1]; out[0] = a + b; out[1] = a - b; return; }
    const CFFT<T> &f = cfft<T>(order - 1);
    std::vector<std::complex<T>> z(L);
    for (size_t j = 0; j < L; j++) z[j] = std::complex<T>(in[2 * j], in[2 * j + 1]);
    f.run(z.data(), false);
    out[0] = z[0].real() + z[0].imag();
    out[L] = z[0].real() - z[0].imag();
    for (size_t k = 1; k <= L / 2; k++) {
        const std::complex<T> a = z[k], b = std::conj(z[L - k]);
        const std::complex<T> e((T)0.5 * (a.real() + b.real()), (T)0.5 * (a.imag() + b.imag()));
        const std::complex<T> d((T)0.5 * (a.real() - b.real()), (T)0.5 * (a.imag() - b.imag()));
        const std::complex<T> q(d.imag(), -d.real());                 // d / i
        const std::complex<T> wk = f.wr[k];
        const std::complex<T> wo(q.real() * wk.real() - q.imag() * wk.imag(), q.real() * wk.imag() + q.imag() * wk.real());
        const std::complex<T> xk = e + wo, xlk = std::conj(e - wo);
        out[k] = xk.real(); out[n - k] = xk.imag();
        if (k != L - k) { out[L - k] = xlk.real(); out[L + k] = xlk.imag(); }
    }
}

// FFTW_HC2R of n = 2^order reals, unnormalised (r2hc then hc2r multiplies by n); in may equal out
template <typename T> void hc2r(int order, const T *in, T *out) {
    const size_t n = (size_t)1 << order, L = n / 2;
    if (L == 1) { const T a = in[0], b = in[1]; out[0] = a + b; out[1] = a - b; return; }
    const CFFT<T> &f = cfft<T>(order - 1);
    std::vector<std::complex<T>> z(L);
    z[0] = std::complex<T>(in[0] + in[L], in[0] - in[L]);
    for (size_t k = 1; k <= L / 2; k++) {
        const std::complex<T> a(in[k], in[n - k]);
        const std::complex<T> b = (k == L - k) ? std::conj(a) : std::complex<T>(in[L - k], -in[L + k]);
        const std::complex<T> e = a + b, d = a - b;
        const std::complex<T> wk = std::conj(f.wr[k]);
        const std::complex<T> o(d.real() * wk.real() - d.imag() * wk.imag(), d.real() * wk.imag() + d.imag() * wk.real());
        z[k] = std::complex<T>(e.real() - o.imag(), e.imag() + o.real());
        if (k != L - k) z[L - k] = std::complex<T>(e.real() + o.imag(), -e.imag() + o.real());
    }
    f.run(z.data(), true);
    for (size_t j = 0; j < L; j++) { out[2 * j] = z[j].real(); out[2 * j + 1] = z[j].imag(); }
}

}  // namespace bfhost
