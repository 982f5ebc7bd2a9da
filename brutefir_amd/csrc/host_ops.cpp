// host_ops.cpp -- the pure HOST half of the convolver.h boundary (include/bfhip_convolver.h).
//
// BruteFIR initialises its convolver and prepares every coefficient set in the PARENT process,
// before bfrun fork()s the filter / input / output / module processes (bfconf.c:2786,
// 1979-2019; delay.c:416-505 for the sub-sample filters), and bflogic_eq renders new
// coefficients in a process of its own (rendereq.h:66-91, bflogic_eq.c:105-120, 587-589).
// HIP state does not survive fork(), and a module process has no business owning a GPU context.
// So everything those callers reach is implemented here without a single HIP call -- this file
// is compiled by g++, not hipcc, which is the proof -- with the small host FFT of host_fft.h
// (the reference runs FFTW at the same places).  None of it is on the per-block path: the block
// loop's transforms and multiply-accumulates are the HIP kernels of kernels.h.
//
// Also here: the cross-process coefficient change notices (bfhip_coeff_mark_dirty) that let the
// filter process re-upload a partition bflogic_eq has just rewritten in shared memory.
#include <sys/mman.h>
#include <unistd.h>

#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/bfhip_convolver.h"
#include "conv_shared.h"
#include "host_fft.h"

extern "C" {
struct bfhip_conv_globals bfhip_conv_g = {0, 0, -1, 0};
}

namespace {

int g_last_fatal = 0;
void (*g_handler)(int, const char *) = nullptr;
bfhip_dirty_table *g_dirty = nullptr;
std::vector<void *> g_coeff_allocs;          // "never freed", like the reference's emallocaligned

void fatalf(int code, const char *fmt, ...) {
    char msg[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof(msg), fmt, ap);
    va_end(ap);
    bfhip_conv_fatal(code, msg);
}

bool ready() {
    if (!bfhip_conv_g.inited) { fatalf(101, "convolver_init() has not been called."); return false; }
    return true;
}

inline int L() { return bfhip_conv_g.L; }
inline int RS() { return bfhip_conv_g.rs; }
inline size_t cbuf_bytes() { return (size_t)2 * L() * RS(); }

// mixnscale of ONE buffer (fftw_convfuns.h:22-60, 268-300): scale + layout change between
// halfcomplex and the "4 re / 4 im" cbuf order (Nyquist in slot 4 of the first group)
template <typename T> void reorder_in(const T *hc, T *cbuf, T s, int len) {      // MIXMODE_INPUT
    for (int k = 0; k < len; k++) {
        const int hre = k, him = k == 0 ? len : 2 * len - k;
        const int qre = 8 * (k >> 2) + (k & 3), qim = qre + 4;
        cbuf[qre] = hc[hre] * s;
        cbuf[qim] = hc[him] * s;
    }
}
template <typename T> void reorder_out(const T *cbuf, T *hc, T s, int len) {     // MIXMODE_OUTPUT
    for (int k = 0; k < len; k++) {
        const int hre = k, him = k == 0 ? len : 2 * len - k;
        const int qre = 8 * (k >> 2) + (k & 3), qim = qre + 4;
        hc[hre] = cbuf[qre] * s;
        hc[him] = cbuf[qim] * s;
    }
}

template <typename T>
void *coeffs2cbuf_t(const void *coeffs, int n_coeffs, double scale, void *optional_dest) {
    const int len = n_coeffs > L() ? L() : (n_coeffs < 0 ? 0 : n_coeffs);
    std::vector<T> r((size_t)2 * L(), (T)0);
    for (int n = 0; n < len; n++) {                               // fftw_convolver.c:535-547
        r[L() + n] = ((const T *)coeffs)[n] * (T)scale;
        if (!std::isfinite((double)r[L() + n])) {
            fprintf(stderr, "NaN or Inf value among coefficients.\n");
            return nullptr;
        }
    }
    bfhost::r2hc<T>(bfhip_conv_g.log2L + 1, r.data(), r.data());
    void *dest = optional_dest;
    if (dest == nullptr) {
        if (posix_memalign(&dest, 32, cbuf_bytes()) != 0) { fatalf(3, "Could not allocate memory."); return nullptr; }
        g_coeff_allocs.push_back(dest);
    }
    reorder_in<T>(r.data(), (T *)dest, (T)(1.0 / (double)(2 * L())), L());
    return dest;
}

struct Plan { int order, invert; };

unsigned slot_of(uintptr_t a) {
    uint64_t h = (uint64_t)a * 0x9E3779B97F4A7C15ull;
    return (unsigned)(h >> 40) % BFHIP_DIRTY_SLOTS;
}

}  // namespace

extern "C" {

void bfhip_conv_fatal(int code, const char *message) {
    g_last_fatal = code;
    if (g_handler) { g_handler(code, message); return; }
    fprintf(stderr, "%s\n", message);
    exit(1);                                  /* BF_EXIT_OTHER, what bf_exit() passes on */
}

int bfhip_convolver_last_fatal(void) { return g_last_fatal; }
void bfhip_convolver_set_fatal_handler(void (*handler)(int, const char *)) { g_handler = handler; g_last_fatal = 0; }

int convolver_init(const char config_filename[], int length, int realsize) {
    (void)config_filename;                     /* FFTW wisdom: nothing to tune here */
    if (realsize != 4 && realsize != 8) { fprintf(stderr, "Invalid real size %d.\n", realsize); return 0; }
    int order = 0;
    while ((1 << order) < length) order++;
    /* (the cbuf layout works in groups of 4 bins, fftw_convfuns.h:25-43: 4 is the shortest partition) */
    if (length < 4 || (1 << order) != length) { fprintf(stderr, "Invalid length %d.\n", length); return 0; }
    if (order > 20) { fprintf(stderr, "Invalid length %d (the device path supports up to 1048576).\n", length); return 0; }
    bfhip_conv_g.L = length; bfhip_conv_g.rs = realsize; bfhip_conv_g.log2L = order; bfhip_conv_g.inited = 1;
    g_last_fatal = 0;
    if (g_dirty == nullptr) {
        /* shared with every process fork()ed from here on: the change notices of
           bfhip_coeff_mark_dirty travel through it */
        void *p = mmap(nullptr, sizeof(bfhip_dirty_table), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
        if (p != MAP_FAILED) g_dirty = (bfhip_dirty_table *)p;      /* zero-filled by the kernel */
    }
    return 1;
}

int convolver_cbufsize(void) { return 2 * bfhip_conv_g.L * bfhip_conv_g.rs; }

struct bfhip_dirty_table *bfhip_dirty_table_get(void) { return g_dirty; }

void bfhip_coeff_mark_dirty(const void *cbuf) {
    if (g_dirty == nullptr || cbuf == nullptr) return;
    const uintptr_t a = (uintptr_t)cbuf;
    unsigned s = slot_of(a);
    for (int probe = 0; probe < BFHIP_DIRTY_SLOTS; probe++, s = (s + 1) % BFHIP_DIRTY_SLOTS) {
        uintptr_t cur = __atomic_load_n(&g_dirty->slot[s].addr, __ATOMIC_ACQUIRE);
        if (cur == 0) {
            uintptr_t expect = 0;
            if (__atomic_compare_exchange_n(&g_dirty->slot[s].addr, &expect, a, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) cur = a;
            else cur = expect;
        }
        if (cur == a) {
            __atomic_add_fetch(&g_dirty->slot[s].gen, 1, __ATOMIC_RELEASE);
            __atomic_add_fetch(&g_dirty->seq, 1, __ATOMIC_RELEASE);
            return;
        }
    }
    /* table full (more than 8192 distinct partitions rewritten at run time): count the notice
       as lost -- watchers that see the counter move re-read everything they watch */
    __atomic_add_fetch(&g_dirty->lost, 1, __ATOMIC_RELEASE);
    __atomic_add_fetch(&g_dirty->seq, 1, __ATOMIC_RELEASE);
}

uint64_t bfhip_dirty_lost(void) {
    return g_dirty ? __atomic_load_n(&g_dirty->lost, __ATOMIC_ACQUIRE) : 0;
}

uint64_t bfhip_dirty_generation(const void *cbuf) {
    if (g_dirty == nullptr || cbuf == nullptr) return 0;
    const uintptr_t a = (uintptr_t)cbuf;
    unsigned s = slot_of(a);
    for (int probe = 0; probe < BFHIP_DIRTY_SLOTS; probe++, s = (s + 1) % BFHIP_DIRTY_SLOTS) {
        const uintptr_t cur = __atomic_load_n(&g_dirty->slot[s].addr, __ATOMIC_ACQUIRE);
        if (cur == 0) return 0;
        if (cur == a) return __atomic_load_n(&g_dirty->slot[s].gen, __ATOMIC_ACQUIRE);
    }
    return 0;
}

unsigned long long bfhip_coeff_dirty_sequence(void) {
    return g_dirty ? (unsigned long long)__atomic_load_n(&g_dirty->seq, __ATOMIC_ACQUIRE) : 0ull;
}

void *convolver_coeffs2cbuf(void *coeffs, int n_coeffs, double scale, void *optional_dest) {
    if (!ready()) return NULL;
    return RS() == 4 ? coeffs2cbuf_t<float>(coeffs, n_coeffs, scale, optional_dest)
                     : coeffs2cbuf_t<double>(coeffs, n_coeffs, scale, optional_dest);
}

void convolver_runtime_coeffs2cbuf(void *src, void *dest) {
    if (!ready()) return;
    const size_t half = (size_t)L() * RS();
    /* the reference zero-pads in `dest` itself and transforms out of place into a static
       temporary (fftw_convolver.c:586-595); a local temporary keeps this re-entrant */
    std::vector<unsigned char> tmp(cbuf_bytes(), 0);
    memcpy(tmp.data() + half, src, half);
    if (RS() == 4) {
        bfhost::r2hc<float>(bfhip_conv_g.log2L + 1, (const float *)tmp.data(), (float *)tmp.data());
        reorder_in<float>((const float *)tmp.data(), (float *)dest, (float)(1.0 / (double)(2 * L())), L());
    } else {
        bfhost::r2hc<double>(bfhip_conv_g.log2L + 1, (const double *)tmp.data(), (double *)tmp.data());
        reorder_in<double>((const double *)tmp.data(), (double *)dest, 1.0 / (double)(2 * L()), L());
    }
    /* tell the filter process(es) that this partition changed (no-op for anyone not watching) */
    bfhip_coeff_mark_dirty(dest);
}

int convolver_verify_cbuf(void *cbufs[], int n_cbufs) {
    for (int n = 0; n < n_cbufs; n++) {
        for (int i = 0; i < 2 * L(); i++) {
            const double v = RS() == 4 ? (double)((float *)cbufs[n])[i] : ((double *)cbufs[n])[i];
            if (!std::isfinite(v)) { fprintf(stderr, "NaN or Inf value among coefficients.\n"); return 0; }
        }
    }
    return 1;
}

void convolver_debug_dump_cbuf(const char filename[], void *cbufs[], int n_cbufs) {
    if (!ready()) return;
    FILE *stream = fopen(filename, "wt");
    if (stream == NULL) { fprintf(stderr, "Could not open \"%s\" for writing: %s", filename, strerror(errno)); return; }
    std::vector<unsigned char> tmp(cbuf_bytes());
    for (int n = 0; n < n_cbufs; n++) {
        if (RS() == 4) {
            reorder_out<float>((const float *)cbufs[n], (float *)tmp.data(), 1.0f, L());
            bfhost::hc2r<float>(bfhip_conv_g.log2L + 1, (const float *)tmp.data(), (float *)tmp.data());
            for (int i = 0; i < L(); i++) fprintf(stream, "%.16e\n", ((float *)tmp.data())[L() + i]);
        } else {
            reorder_out<double>((const double *)cbufs[n], (double *)tmp.data(), 1.0, L());
            bfhost::hc2r<double>(bfhip_conv_g.log2L + 1, (const double *)tmp.data(), (double *)tmp.data());
            for (int i = 0; i < L(); i++) fprintf(stream, "%.16e\n", ((double *)tmp.data())[L() + i]);
        }
    }
    fclose(stream);
}

void *convolver_fftplan(int order, int invert, int inplace) {
    (void)inplace;
    static std::mutex mu;
    static std::map<std::pair<int, int>, Plan *> plans;     /* "Do not free it" (convolver.h:128) */
    std::lock_guard<std::mutex> lock(mu);
    auto key = std::make_pair(order, invert ? 1 : 0);
    auto it = plans.find(key);
    if (it != plans.end()) return it->second;
    Plan *p = new Plan{order, invert ? 1 : 0};
    plans[key] = p;
    return p;
}

void bfhip_fftplan_execute(void *plan, void *in, void *out) {
    const Plan *p = (const Plan *)plan;
    if (p == NULL || p->order < 1 || p->order > 24) { fatalf(104, "bfhip: invalid FFT plan"); return; }
    if (!ready()) return;
    if (RS() == 4) {
        if (p->invert) bfhost::hc2r<float>(p->order, (const float *)in, (float *)out);
        else bfhost::r2hc<float>(p->order, (const float *)in, (float *)out);
    } else {
        if (p->invert) bfhost::hc2r<double>(p->order, (const double *)in, (double *)out);
        else bfhost::r2hc<double>(p->order, (const double *)in, (double *)out);
    }
}

int convolver_td_block_length(int n_coeffs) {
    if (n_coeffs < 1) return -1;
    int o = 0;
    while ((1 << o) < n_coeffs) o++;                          /* 1 << log2_roof(n) */
    return 1 << o;
}

td_conv_t *convolver_td_new(void *coeffs, int n_coeffs) {
    const int blocklen = convolver_td_block_length(n_coeffs);
    if (blocklen == -1 || !ready()) return NULL;
    int lg = 0;
    while ((1 << lg) < blocklen) lg++;
    const size_t bytes = (size_t)2 * blocklen * RS();
    td_conv_t *tdc = new td_conv_t();
    tdc->blocklen = blocklen;
    tdc->d_coeffs = nullptr;
    tdc->d_pid = 0;
    tdc->h_coeffs = calloc(1, bytes);                          /* [blocklen zeros | coeffs | zeros] */
    if (tdc->h_coeffs == nullptr) { fatalf(3, "Could not allocate memory."); delete tdc; return NULL; }
    memcpy((unsigned char *)tdc->h_coeffs + (size_t)blocklen * RS(), coeffs, (size_t)n_coeffs * RS());
    if (RS() == 4) {
        float *c = (float *)tdc->h_coeffs;
        bfhost::r2hc<float>(lg + 1, c, c);
        const float s = 1.0f / (float)(blocklen << 1);        /* fftw_convolver.c:720-724 */
        for (int n = 0; n < blocklen << 1; n++) c[n] *= s;
    } else {
        double *c = (double *)tdc->h_coeffs;
        bfhost::r2hc<double>(lg + 1, c, c);
        const double s = 1.0 / (double)(blocklen << 1);
        for (int n = 0; n < blocklen << 1; n++) c[n] *= s;
    }
    return tdc;
}

}  // extern "C"
