// nupc.hip -- non-uniform partitioned convolution (low-latency first block) on top of the
// uniform engines.  An EXTENSION: the reference only has uniform partitions
// (`filter_length: L,N`, bfconf.c:1495-1520); BASELINE.json's room-correction
// config asks for "non-uniform partition sizes (low-latency first block)".  Results are the
// same linear convolution a uniform run computes (tests compare against the oracle's uniform
// engine on the same stream); only the I/O block -- the latency -- shrinks from L to the
// smallest segment length.
//
// The impulse response is cut into segments; segment k is a uniform partitioned convolver
// (a bfhip_engine) with partition length L_k (ascending powers of two) and N_k partitions,
// covering taps [off_k, off_k + N_k * L_k), off_0 = 0.  I/O happens in blocks of L_0 frames.
// After input block b, every segment whose block is complete ((b+1) * L_0 multiple of L_k)
// runs on the last L_k frames; its L_k output frames belong at absolute sample
// (b+1) * L_0 - L_k + off_k and are added into a time-domain accumulator ring; then the L_0
// frames of output block b are requantised out of the ring.  A contribution is in time iff
//     off_k >= L_k - L_0
// (checked at create); e.g. 2 x 64, 2 x 128, ... doubling satisfies it with equality + L_0.
// A segment that starts later than it has to (slack = off_k - (L_k - L_0) > 0 frames; one period
// for every segment but the first in the "two blocks per size, doubling" schedule, L_k - L_0
// at most) does not have to be finished within the period it is launched in: it runs on its
// own low-priority stream beside the periods that follow, and the main stream only waits for
// it in the period its first output frame is due.  So the worst period costs about as much
// as the common one, which is what real-time use needs.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bfhip_nupc.h"
#include "alloc.h"
#include "kernels.h"

using namespace bfhip;

namespace {

// acc[(pos + j) mod A][o] += seg[j][o] for an L_k x n_out block of a segment's output
template <typename T>
__global__ __launch_bounds__(256) void
nupc_accumulate_kernel(T *__restrict__ acc, const T *__restrict__ seg, unsigned long long pos,
                       int A, int n_out, int n_frames) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_frames * n_out) return;
    const size_t j = i / n_out, o = i % n_out;
    acc[(size_t)((pos + j) % (unsigned long long)A) * n_out + o] += seg[i];
}

// output block b: L_0 frames out of the ring, scaled into output units, requantised like
// convolver_cbuf2raw (real2raw.h / dither_funs.h:71-114), ring region cleared.  One workgroup
// per output channel.
template <typename T>
__global__ __launch_bounds__(256) void
nupc_emit_kernel(T *__restrict__ acc, unsigned long long pos, int A, int n_out, int L0,
                 const DevFormat *__restrict__ fmt, const double *__restrict__ inv_scale,
                 DevOverflow *__restrict__ over, uint8_t *__restrict__ raw, double safety_limit,
                 int *__restrict__ status, unsigned int *__restrict__ arrive, int *__restrict__ host_status) {
    const int ch = blockIdx.x, tid = threadIdx.x;
    const DevFormat f = fmt[ch];
    DevOverflow of = over[ch];
    uint8_t *base = raw + f.byte_offset;
    const size_t stride = (size_t)f.sample_spacing * f.bytes;
    const T sc = (T)inv_scale[ch];
    Quantiser<T> qz;
    qz.init(f, of, safety_limit);
    for (int n = tid; n < L0; n += 256) {
        T *cell = &acc[(size_t)((pos + n) % (unsigned long long)A) * n_out + ch];
        const T x = *cell * sc;
        *cell = (T)0;
        qz.put(x, base + (size_t)n * stride);
    }
    qz.reduce(tid, 256);
    if (tid == 0) {
        qz.commit(of);
        over[ch] = of;
        if (qz.st) atomicOr(status, qz.st);
        // the last channel to finish hands the status bits of all segments (they OR into the
        // same word) to the host's pinned word: no device-to-host copy after the sync
        __threadfence();
        if (atomicAdd(arrive, 1u) + 1u == gridDim.x) {
            *arrive = 0;
            const int all_bits = atomicExch(status, 0);
            if (all_bits) *host_status = *host_status | all_bits;
            __threadfence_system();
        }
    }
}

thread_local std::string n_err;

// (a runtime failure's sticky error is cleared; argument / state errors make no HIP call)
int nfail(int code, const std::string &msg) { n_err = msg; if (code == BFHIP_EHIP || code == BFHIP_ENOMEM) (void)hipGetLastError(); return code; }

#define NCHK(expr)                                                                          \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) return nfail(BFHIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define OWNER(n)                                                                            \
    do {                                                                                    \
        if ((n)->owner != getpid())                                                         \
            return nfail(BFHIP_ESTATE, "this convolver belongs to another process: HIP state does not survive fork()"); \
    } while (0)

#define ECHK(expr)                                                                          \
    do {                                                                                    \
        int _r = (expr);                                                                    \
        if (_r < 0) return nfail(_r, std::string(#expr) + ": " + bfhip_last_error());       \
    } while (0)

struct Seg {
    int L = 0, N = 0;
    long off = 0;               // first tap this segment covers
    bfhip_engine *eng = nullptr;
    void *d_out = nullptr;      // [L][n_out] reals
    // background execution (delay_steps > 0): launched on `stream`, added to the accumulator
    // by the main stream delay_steps periods later
    int delay_steps = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_done = nullptr, ev_consumed = nullptr;
    bool pending = false, consumed_once = false;
    unsigned long long pending_pos = 0, due_block = 0;
};

}  // namespace

struct bfhip_nupc {
    int device = 0, rs = 4, n_in = 0, n_out = 0;
    pid_t owner = 0;                       // like an engine, it lives in the process that created it
    std::vector<Seg> seg;
    std::vector<bfhip_format> fmt[2];
    double safety_limit = 0;
    bool finalized = false, finalize_failed = false;
    unsigned long long block = 0;          // L0-blocks processed
    hipStream_t stream = nullptr;          // main stream: I/O, zero-slack segments, accumulate, emit
    hipEvent_t ev_in = nullptr;            // this period's frames are in the input ring
    int in_frames = 0;                     // input ring length: 2 * Lmax
    unsigned int *d_arrive = nullptr;
    int *h_status = nullptr;               // pinned
    uint8_t *h_in = nullptr, *h_out = nullptr;   // pinned staging of one period (bfhip_nupc_block)
    DevOverflow *h_over = nullptr;         // pinned
    bool background = true;
    int A = 0;                             // accumulator ring length in frames
    void *d_acc = nullptr;                 // [A][n_out] reals
    uint8_t *d_in = nullptr;               // raw input ring: Lmax frames
    size_t frame_bytes[2] = {0, 0};        // interleaved raw frame size in / out
    uint8_t *d_rawout = nullptr;
    DevFormat *d_fmt_out = nullptr;
    double *d_inv_scale = nullptr;
    DevOverflow *d_over = nullptr;
    int *d_status = nullptr;
};

extern "C" {

const char *bfhip_nupc_last_error(void) { return n_err.c_str(); }

bfhip_nupc *bfhip_nupc_create(int device, int realsize, int n_in, int n_out, int n_segments,
                              const int seg_length[], const int seg_blocks[]) {
    if (n_segments < 1 || !seg_length || !seg_blocks || n_in < 1 || n_out < 1) { nfail(BFHIP_EINVAL, "nupc_create: bad argument"); return nullptr; }
    bfhip_nupc *n = new bfhip_nupc();
    n->device = device; n->rs = realsize; n->n_in = n_in; n->n_out = n_out;
    n->owner = getpid();
    long off = 0;
    for (int k = 0; k < n_segments; k++) {
        Seg s;
        s.L = seg_length[k]; s.N = seg_blocks[k]; s.off = off;
        if (s.N < 1 || (k > 0 && (s.L <= seg_length[k - 1] || s.L % seg_length[k - 1] != 0))) {
            nfail(BFHIP_EINVAL, "nupc_create: segment lengths must ascend and divide each other");
            delete n; return nullptr;
        }
        if (off < (long)s.L - seg_length[0]) {
            char buf[200];
            snprintf(buf, sizeof(buf), "nupc_create: segment %d (length %d) starts at tap %ld, before %d: its output "
                     "would not be ready in time", k, s.L, off, s.L - seg_length[0]);
            nfail(BFHIP_EINVAL, buf);
            delete n; return nullptr;
        }
        off += (long)s.L * s.N;
        n->seg.push_back(s);
    }
    for (auto &s : n->seg) {
        s.eng = bfhip_engine_create(device, s.L, s.N, realsize, n_in, n_out);
        if (!s.eng) { nfail(BFHIP_EINVAL, std::string("nupc_create: ") + bfhip_last_error()); bfhip_nupc_destroy(n); return nullptr; }
    }
    for (int io = 0; io < 2; io++) {
        const int c = io ? n_out : n_in;
        n->fmt[io].resize(c);
        for (int ch = 0; ch < c; ch++) {
            bfhip_format &f = n->fmt[io][ch];
            f.isfloat = 1; f.swap = 0; f.bytes = f.sbytes = realsize; f.scale = 1.0;
            f.sample_spacing = c; f.byte_offset = ch * realsize;         // interleaved frames
        }
    }
    return n;
}

void bfhip_nupc_destroy(bfhip_nupc *n) {
    if (!n) return;
    if (n->owner != getpid()) { for (auto &sg : n->seg) if (sg.eng) bfhip_engine_destroy(sg.eng); delete n; return; }   // a forked child
    (void)hipSetDevice(n->device);
    if (n->stream) (void)hipStreamSynchronize(n->stream);
    for (auto &s : n->seg) if (s.stream) (void)hipStreamSynchronize(s.stream);
    for (auto &s : n->seg) {
        if (s.eng) bfhip_engine_destroy(s.eng);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
        if (s.ev_consumed) (void)hipEventDestroy(s.ev_consumed);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    void *p[] = {n->d_acc, n->d_in, n->d_rawout, n->d_fmt_out, n->d_inv_scale, n->d_over, n->d_status, n->d_arrive};
    for (void *q : p) if (q) (void)hipFree(q);
    if (n->h_status) (void)hipHostFree(n->h_status);
    if (n->h_in) (void)hipHostFree(n->h_in);
    if (n->h_out) (void)hipHostFree(n->h_out);
    if (n->h_over) (void)hipHostFree(n->h_over);
    if (n->ev_in) (void)hipEventDestroy(n->ev_in);
    if (n->stream) (void)hipStreamDestroy(n->stream);
    delete n;
}

long bfhip_nupc_taps(const bfhip_nupc *n) { return n ? n->seg.back().off + (long)n->seg.back().L * n->seg.back().N : 0; }
int bfhip_nupc_latency(const bfhip_nupc *n) { return n ? n->seg[0].L : 0; }

// raw formats of the L0-frame I/O buffers; frames must be interleaved (every channel of a side
// has the same sample_spacing = frame size in samples), the way dai.c lays out an interleaved
// device, because segment k reads L_k consecutive frames of the input ring
int bfhip_nupc_set_format(bfhip_nupc *n, int io, int ch, const bfhip_format *f) {
    if (!n || !f || io < 0 || io > 1 || ch < 0 || ch >= (io ? n->n_out : n->n_in)) return nfail(BFHIP_EINVAL, "nupc_set_format: bad argument");
    if (n->finalized) return nfail(BFHIP_ESTATE, "nupc_set_format after finalize");
    n->fmt[io][ch] = *f;
    return BFHIP_OK;
}

int bfhip_nupc_set_safety_limit(bfhip_nupc *n, double limit) { if (!n) return BFHIP_EINVAL; n->safety_limit = limit; return BFHIP_OK; }

// one filter = one impulse response from an input to an output (taps in host memory, realsize
// wide); it is cut along the segment boundaries and loaded into every segment's engine
int bfhip_nupc_add_filter(bfhip_nupc *n, int in_ch, int out_ch, const void *taps, long n_taps,
                          double in_scale, double out_scale) {
    if (!n || !taps || n_taps < 1 || in_ch < 0 || in_ch >= n->n_in || out_ch < 0 || out_ch >= n->n_out) return nfail(BFHIP_EINVAL, "nupc_add_filter: bad argument");
    if (n->finalized) return nfail(BFHIP_ESTATE, "nupc_add_filter after finalize");
    for (auto &s : n->seg) {
        const long avail = n_taps - s.off;
        const long cap = (long)s.L * s.N;
        const long take = avail < 0 ? 0 : (avail > cap ? cap : avail);
        std::vector<unsigned char> zero;
        const void *src = (const unsigned char *)taps + (size_t)s.off * n->rs;
        if (take == 0) { zero.assign((size_t)n->rs, 0); src = zero.data(); }
        const int c = bfhip_engine_add_coeff(s.eng, src, take == 0 ? 1 : (int)take, 1.0, take == 0 ? 1 : 0);
        if (c < 0) return nfail(c, std::string("nupc_add_filter: ") + bfhip_last_error());
        // the engines emit plain reals: the output format's 1/scale is applied at the emit step
        const int r = bfhip_engine_add_filter(s.eng, 1, &in_ch, &in_scale, 0, nullptr, nullptr, 1, &out_ch, &out_scale, c, 0, 0);
        if (r < 0) return nfail(r, std::string("nupc_add_filter: ") + bfhip_last_error());
    }
    return BFHIP_OK;
}

static int nupc_finalize_impl(bfhip_nupc *n);

int bfhip_nupc_finalize(bfhip_nupc *n) {
    if (!n) return nfail(BFHIP_EINVAL, "null");
    if (n->finalized) return BFHIP_OK;
    // half-built state is cleaned up by bfhip_nupc_destroy only: a failed finalize is not retried
    if (n->finalize_failed) return nfail(BFHIP_ESTATE, "nupc_finalize failed before: destroy this convolver");
    const int r = nupc_finalize_impl(n);
    if (r != BFHIP_OK) { n->finalize_failed = true; n->finalized = false; }
    return r;
}

static int nupc_finalize_impl(bfhip_nupc *n) {
    OWNER(n);
    NCHK(hipSetDevice(n->device));
    int prio_least = 0, prio_greatest = 0;
    NCHK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    NCHK(hipStreamCreateWithPriority(&n->stream, hipStreamNonBlocking, prio_greatest));
    NCHK(hipEventCreateWithFlags(&n->ev_in, hipEventDisableTiming));
    if (const char *bg = getenv("BFHIP_NUPC_BACKGROUND")) n->background = atoi(bg) != 0;
    for (int io = 0; io < 2; io++) {
        const int c = io ? n->n_out : n->n_in;
        const int spacing = n->fmt[io][0].sample_spacing, bytes = n->fmt[io][0].bytes;
        for (int ch = 0; ch < c; ch++)
            if (n->fmt[io][ch].sample_spacing != spacing || n->fmt[io][ch].bytes != bytes)
                return nfail(BFHIP_EINVAL, "nupc: all channels of a side must share one interleaved frame layout");
        n->frame_bytes[io] = (size_t)spacing * bytes;
    }
    const int L0 = n->seg[0].L, Lmax = n->seg.back().L;
    long reach = 0;
    for (auto &s : n->seg) reach = std::max(reach, s.off + 2L * s.L);
    int A = 1;
    while (A < reach + L0) A <<= 1;
    n->A = A;
    NCHK(bfhip_internal_dev_alloc((void **)&n->d_acc, (size_t)A * n->n_out * n->rs));
    NCHK(hipMemset(n->d_acc, 0, (size_t)A * n->n_out * n->rs));
    // two periods of the longest segment: its forward transform may still be reading one while
    // the next is being filled
    n->in_frames = 2 * Lmax;
    NCHK(bfhip_internal_dev_alloc((void **)&n->d_in, (size_t)n->in_frames * n->frame_bytes[0]));
    NCHK(hipMemset(n->d_in, 0, (size_t)n->in_frames * n->frame_bytes[0]));
    NCHK(bfhip_internal_dev_alloc((void **)&n->d_rawout, (size_t)L0 * n->frame_bytes[1]));
    NCHK(hipMemset(n->d_rawout, 0, (size_t)L0 * n->frame_bytes[1]));
    std::vector<DevFormat> df(n->n_out);
    std::vector<double> inv(n->n_out);
    std::vector<DevOverflow> ov(n->n_out);
    for (int ch = 0; ch < n->n_out; ch++) {
        const bfhip_format &f = n->fmt[1][ch];
        df[ch].isfloat = f.isfloat; df[ch].swap = f.swap; df[ch].bytes = f.bytes; df[ch].sbytes = f.sbytes;
        df[ch].sample_spacing = f.sample_spacing; df[ch].byte_offset = f.byte_offset; df[ch].alt = nullptr;
        inv[ch] = 1.0 / f.scale;                                         // bfrun.c:1850
        memset(&ov[ch], 0, sizeof(DevOverflow));
        ov[ch].max = f.isfloat ? 1.0 : (double)((uint64_t)1 << ((f.sbytes << 3) - 1)) - 1;
    }
    NCHK(bfhip_internal_dev_alloc((void **)&n->d_fmt_out, df.size() * sizeof(DevFormat)));
    NCHK(hipMemcpy(n->d_fmt_out, df.data(), df.size() * sizeof(DevFormat), hipMemcpyHostToDevice));
    NCHK(bfhip_internal_dev_alloc((void **)&n->d_inv_scale, inv.size() * sizeof(double)));
    NCHK(hipMemcpy(n->d_inv_scale, inv.data(), inv.size() * sizeof(double), hipMemcpyHostToDevice));
    NCHK(bfhip_internal_dev_alloc((void **)&n->d_over, ov.size() * sizeof(DevOverflow)));
    NCHK(hipMemcpy(n->d_over, ov.data(), ov.size() * sizeof(DevOverflow), hipMemcpyHostToDevice));
    NCHK(bfhip_internal_dev_alloc((void **)&n->d_status, sizeof(int)));
    NCHK(hipMemset(n->d_status, 0, sizeof(int)));
    NCHK(bfhip_internal_dev_alloc((void **)&n->d_arrive, sizeof(unsigned int)));
    NCHK(hipMemset(n->d_arrive, 0, sizeof(unsigned int)));
    NCHK(bfhip_internal_pin_alloc((void **)&n->h_status, sizeof(int), hipHostMallocDefault));
    NCHK(bfhip_internal_pin_alloc((void **)&n->h_in, (size_t)L0 * n->frame_bytes[0], hipHostMallocDefault));
    NCHK(bfhip_internal_pin_alloc((void **)&n->h_out, (size_t)L0 * n->frame_bytes[1], hipHostMallocDefault));
    NCHK(bfhip_internal_pin_alloc((void **)&n->h_over, (size_t)n->n_out * sizeof(DevOverflow), hipHostMallocDefault));
    *n->h_status = 0;
    for (auto &s : n->seg) {
        for (int ch = 0; ch < n->n_in; ch++) ECHK(bfhip_engine_set_format(s.eng, BFHIP_IN, ch, &n->fmt[0][ch]));
        for (int ch = 0; ch < n->n_out; ch++) {
            bfhip_format f;
            f.isfloat = 1; f.swap = 0; f.bytes = f.sbytes = n->rs; f.scale = 1.0;
            f.sample_spacing = n->n_out; f.byte_offset = ch * n->rs;
            ECHK(bfhip_engine_set_format(s.eng, BFHIP_OUT, ch, &f));
        }
        ECHK(bfhip_engine_set_overlap(s.eng, 0));          // a segment's kernels are ordered on ONE stream
        ECHK(bfhip_engine_finalize(s.eng));
        const long slack = s.off - ((long)s.L - L0);       // >= 0, checked at create
        const long room = (long)s.L - L0;                  // its next block is launched L frames later
        s.delay_steps = n->background ? (int)(std::min(slack, room) / L0) : 0;
        if (s.delay_steps > 0) {
            NCHK(hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, prio_least));
            NCHK(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
            NCHK(hipEventCreateWithFlags(&s.ev_consumed, hipEventDisableTiming));
        }
        ECHK(bfhip_engine_set_stream(s.eng, s.delay_steps > 0 ? s.stream : n->stream));
        ECHK(bfhip_engine_set_status_dev(s.eng, n->d_status));
        NCHK(bfhip_internal_dev_alloc((void **)&s.d_out, (size_t)s.L * n->n_out * n->rs));
    }
    n->finalized = true;
    return BFHIP_OK;
}

}  // extern "C"

namespace {

int nupc_accumulate(bfhip_nupc *n, const Seg &s, unsigned long long pos) {
    const size_t cnt = (size_t)s.L * n->n_out;
    if (n->rs == 4)
        hipLaunchKernelGGL(nupc_accumulate_kernel<float>, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, n->stream,
                           (float *)n->d_acc, (const float *)s.d_out, pos, n->A, n->n_out, s.L);
    else
        hipLaunchKernelGGL(nupc_accumulate_kernel<double>, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, n->stream,
                           (double *)n->d_acc, (const double *)s.d_out, pos, n->A, n->n_out, s.L);
    NCHK(hipGetLastError());
    return BFHIP_OK;
}

}  // namespace

extern "C" {

// one I/O block of L_0 frames, device-resident raw buffers, asynchronous on the nupc's stream
int bfhip_nupc_block_dev(bfhip_nupc *n, const void *rawin_dev, void *rawout_dev) {
    if (!n || !n->finalized) return nfail(BFHIP_ESTATE, "nupc not finalized");
    if (!rawin_dev || !rawout_dev) return nfail(BFHIP_EINVAL, "nupc_block_dev: null buffer");
    OWNER(n);
    NCHK(hipSetDevice(n->device));
    const int L0 = n->seg[0].L;
    const unsigned long long end = (n->block + 1) * (unsigned long long)L0;       // samples received
    // append to the input ring (segment blocks are aligned: never split by the wrap)
    const size_t wpos = (size_t)((end - L0) % (unsigned long long)n->in_frames);
    if ((const uint8_t *)rawin_dev != n->d_in + wpos * n->frame_bytes[0])     // block() uploads straight into the slot
        NCHK(hipMemcpyAsync(n->d_in + wpos * n->frame_bytes[0], rawin_dev, (size_t)L0 * n->frame_bytes[0], hipMemcpyDeviceToDevice, n->stream));
    bool ev_in_recorded = false;
    for (auto &s : n->seg) {
        if (end % (unsigned long long)s.L != 0) continue;
        const size_t rpos = (size_t)((end - s.L) % (unsigned long long)n->in_frames);
        const unsigned long long pos = end - s.L + (unsigned long long)s.off;
        if (s.delay_steps == 0) {
            ECHK(bfhip_engine_block_dev(s.eng, n->d_in + rpos * n->frame_bytes[0], s.d_out));
            { const int r = nupc_accumulate(n, s, pos); if (r < 0) return r; }
            continue;
        }
        if (s.pending) return nfail(BFHIP_ESTATE, "nupc: a background segment block was never collected");
        if (!ev_in_recorded) { NCHK(hipEventRecord(n->ev_in, n->stream)); ev_in_recorded = true; }
        NCHK(hipStreamWaitEvent(s.stream, n->ev_in, 0));
        if (s.consumed_once) NCHK(hipStreamWaitEvent(s.stream, s.ev_consumed, 0));   // d_out is free again
        ECHK(bfhip_engine_block_dev(s.eng, n->d_in + rpos * n->frame_bytes[0], s.d_out));
        NCHK(hipEventRecord(s.ev_done, s.stream));
        s.pending = true;
        s.pending_pos = pos;
        s.due_block = n->block + (unsigned long long)s.delay_steps;
    }
    for (auto &s : n->seg) {
        if (!s.pending || s.due_block != n->block) continue;
        NCHK(hipStreamWaitEvent(n->stream, s.ev_done, 0));
        { const int r = nupc_accumulate(n, s, s.pending_pos); if (r < 0) return r; }
        NCHK(hipEventRecord(s.ev_consumed, n->stream));
        s.pending = false;
        s.consumed_once = true;
    }
    const unsigned long long opos = end - L0;
    if (n->rs == 4)
        hipLaunchKernelGGL(nupc_emit_kernel<float>, dim3(n->n_out), dim3(256), 0, n->stream, (float *)n->d_acc, opos, n->A, n->n_out, L0,
                           n->d_fmt_out, n->d_inv_scale, n->d_over, (uint8_t *)rawout_dev, n->safety_limit, n->d_status,
                           n->d_arrive, n->h_status);
    else
        hipLaunchKernelGGL(nupc_emit_kernel<double>, dim3(n->n_out), dim3(256), 0, n->stream, (double *)n->d_acc, opos, n->A, n->n_out, L0,
                           n->d_fmt_out, n->d_inv_scale, n->d_over, (uint8_t *)rawout_dev, n->safety_limit, n->d_status,
                           n->d_arrive, n->h_status);
    NCHK(hipGetLastError());
    n->block++;
    return BFHIP_OK;
}

// waits for the periods handed in so far (NOT for background segment blocks that are not due
// yet) and returns the status bits collected since the last call
int bfhip_nupc_sync(bfhip_nupc *n) {
    if (!n || !n->finalized) return nfail(BFHIP_ESTATE, "nupc not finalized");
    OWNER(n);
    NCHK(hipSetDevice(n->device));
    NCHK(hipStreamSynchronize(n->stream));
    const int st = *(volatile int *)n->h_status;
    *n->h_status = 0;
    return st;
}

// host buffers: copies in, runs, copies out, waits; returns status bits
int bfhip_nupc_block(bfhip_nupc *n, const void *rawin, void *rawout, bfhip_overflow overflow[]) {
    if (!n || !n->finalized || !rawin || !rawout) return nfail(BFHIP_ESTATE, "nupc_block: bad state or argument");
    OWNER(n);
    NCHK(hipSetDevice(n->device));
    const int L0 = n->seg[0].L;
    // upload straight into this block's slot of the input ring
    const unsigned long long end = (n->block + 1) * (unsigned long long)L0;
    const size_t wpos = (size_t)((end - L0) % (unsigned long long)n->in_frames);
    uint8_t *slot = n->d_in + wpos * n->frame_bytes[0];
    // through pinned staging buffers: copies from pageable memory stall in the runtime every so often
    memcpy(n->h_in, rawin, (size_t)L0 * n->frame_bytes[0]);
    NCHK(hipMemcpyAsync(slot, n->h_in, (size_t)L0 * n->frame_bytes[0], hipMemcpyHostToDevice, n->stream));
    if (overflow) {
        memcpy(n->h_over, overflow, n->n_out * sizeof(DevOverflow));
        NCHK(hipMemcpyAsync(n->d_over, n->h_over, n->n_out * sizeof(DevOverflow), hipMemcpyHostToDevice, n->stream));
    }
    int r = bfhip_nupc_block_dev(n, slot, n->d_rawout);
    if (r < 0) return r;
    NCHK(hipMemcpyAsync(n->h_out, n->d_rawout, (size_t)L0 * n->frame_bytes[1], hipMemcpyDeviceToHost, n->stream));
    if (overflow) NCHK(hipMemcpyAsync(n->h_over, n->d_over, n->n_out * sizeof(DevOverflow), hipMemcpyDeviceToHost, n->stream));
    r = bfhip_nupc_sync(n);
    if (r < 0) return r;
    memcpy(rawout, n->h_out, (size_t)L0 * n->frame_bytes[1]);
    if (overflow) memcpy(overflow, n->h_over, n->n_out * sizeof(DevOverflow));
    return r;
}

int bfhip_nupc_get_overflow(bfhip_nupc *n, int ch, bfhip_overflow *of) {
    if (!n || !n->finalized || !of || ch < 0 || ch >= n->n_out) return nfail(BFHIP_EINVAL, "nupc_get_overflow: bad argument");
    OWNER(n);
    NCHK(hipSetDevice(n->device));
    NCHK(hipStreamSynchronize(n->stream));
    NCHK(hipMemcpy(of, n->d_over + ch, sizeof(DevOverflow), hipMemcpyDeviceToHost));
    return BFHIP_OK;
}

}  // extern "C"
