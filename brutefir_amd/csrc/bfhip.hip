// bfhip.hip -- host side of the engine + the C ABI of include/bfhip.h.
//
// What the reference's filter_process() does with host buffers and a chain of convolver_*
// calls per filter (bfrun.c:1493-2008) is turned into a static device plan here:
//   * every input channel owns one ring of its last N spectra in HBM (the reference keeps one
//     ring per filter, bfrun.c:1273-1287; filters that only scale a single input share the
//     input's ring and carry their scale in the plan -> the ring is read once per output
//     group instead of once per filter);
//   * every output group owns a list of (ring, delay) entries with up to 8 filter terms each;
//   * a block is three launches: fft_in (K1), mac_xbar (K2), ifft_out (K3).
// The integer bookkeeping (slot = blockcounter mod N, delay clamp, cblocks truncation,
// warm-up count) follows bfrun.c:1566-1600,1745-1746 literally.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <string>
#include <vector>
#include <mutex>
#include <set>
#include <tuple>

#include "../../include/bfhip.h"
#include "alloc.h"
#include "conv_shared.h"
#include "kernels.h"
#include "bigfft.h"

using namespace bfhip;

// Every device / pinned allocation of the engines goes through these (alloc.h), so that the tests
// can make the n-th one fail (bfhip_selftest_fail_alloc) and walk every out-of-memory path there is.
static int g_fail_alloc = 0;           // countdown: the allocation that takes it to zero fails
static inline bool alloc_injected() { return g_fail_alloc > 0 && --g_fail_alloc == 0; }
extern "C" hipError_t bfhip_internal_dev_alloc(void **p, size_t bytes) {
    if (alloc_injected()) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipMalloc(p, bytes);
}
extern "C" hipError_t bfhip_internal_pin_alloc(void **p, size_t bytes, unsigned int flags) {
    if (alloc_injected()) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipHostMalloc(p, bytes, flags);
}
static inline hipError_t dev_alloc(void **p, size_t bytes) { return bfhip_internal_dev_alloc(p, bytes); }
template <typename P> static hipError_t dev_alloc(P **p, size_t bytes) { return bfhip_internal_dev_alloc((void **)p, bytes); }
static inline hipError_t pin_alloc(void **p, size_t bytes, unsigned int flags) { return bfhip_internal_pin_alloc(p, bytes, flags); }

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    // a failed runtime call (an allocation that did not fit, say) leaves its code behind as the
    // "last error"; reported here, it must not be found again by the launch check of a later,
    // unrelated call.  Only the two codes a runtime failure is reported under reach the runtime:
    // argument and state errors (a NULL handle, the fork()ed-child refusal of check_owner) return
    // without a single HIP call.
    if (code == BFHIP_EHIP || code == BFHIP_ENOMEM) (void)hipGetLastError();
    return code;
}

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return fail(BFHIP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                        __FILE__, __LINE__);                                              \
    } while (0)

int ilog2(int v) {
    int o = 0;
    while ((1 << o) < v) o++;
    return ((1 << o) == v) ? o : -1;
}

struct Coeff {
    int n_blocks = 0;
    void *d_H = nullptr;     // [n_blocks][L] packed spectra
    // a set the host keeps in (shared) memory in the reference's cbuf layout and that another
    // process may rewrite at run time (bflogic_eq through bfaccess->convolver_coeffs2cbuf):
    // where each block lives on the host, and the change-notice generation last uploaded
    std::vector<const void *> watch_src;
    std::vector<uint64_t> watch_gen;
    // BFHIP_COEFF_LAZY: the host blocks are remembered (watch_src) and go to the device the first
    // time an ACTIVE filter refers to the set -- a filter process of a multi-process host holds the
    // sets of its own filters, not the whole configuration's (d_H == nullptr until then)
    bool lazy = false, watched = false;
};

// Pinned staging for the small per-block tables (N:1 channel jobs, delay-line moves, sub-sample
// delay jobs) that go to the device with every block: K slots, one per block in flight, so the
// upload is truly asynchronous and the host copy outlives the call that queued it.
struct Stage {
    static constexpr int K = 4;
    uint8_t *base = nullptr;
    size_t slot = 0;
    unsigned int turn = 0;             // (unsigned: a stream of 64-frame blocks passes 2^31 in a month)
    hipEvent_t done[K] = {nullptr, nullptr, nullptr, nullptr};
    bool pending[K] = {false, false, false, false};

    hipError_t init(size_t slot_bytes) {
        slot = (slot_bytes + 63) & ~(size_t)63;
        hipError_t r = pin_alloc((void **)&base, slot * K, hipHostMallocDefault);
        for (int i = 0; i < K && r == hipSuccess; i++) r = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
        return r;
    }
    void release() {
        for (int i = 0; i < K; i++) if (done[i]) { (void)hipEventDestroy(done[i]); done[i] = nullptr; }
        if (base) { (void)hipHostFree(base); base = nullptr; }
    }
    // next slot, free again once the copy queued from it K blocks ago has run
    hipError_t take(uint8_t **p) {
        const int i = (int)(turn++ % (unsigned int)K);
        if (pending[i]) { const hipError_t r = hipEventSynchronize(done[i]); if (r != hipSuccess) return r; pending[i] = false; }
        *p = base + (size_t)i * slot;
        return hipSuccess;
    }
    // the copies out of the slot handed out last have been queued on st
    hipError_t queued(hipStream_t st) {
        const int i = (int)((turn - 1u) % (unsigned int)K);
        const hipError_t r = hipEventRecord(done[i], st);
        pending[i] = r == hipSuccess;
        return r;
    }
};

struct Filter {
    std::vector<int> in_ch, in_f, out_ch;
    std::vector<double> in_scale, in_fscale, out_scale;
    int coeff = -1, delayblocks = 0, crossfade = 0;
    int prevcoeff = -1;
    // false: another engine (another filter process of the host, bfrun.c:2312-2328) runs this filter.
    // It still takes part in the PLAN -- groups, entry order, chunk boundaries are those of the whole
    // configuration, so every output is summed in exactly the order a single engine would use --
    // but none of its work is launched here and its coefficients are not loaded.
    bool active = true;
    int name = -1;                     // the host's own number for the filter (intname); -1: its index here
};

constexpr int MAX_TIMED = 4096;
// event pairs of a timed block: 0 K1 (input transforms), 1 K2 (MAC), 2 K3 (output pass, everything
// of it), 3 the part of the output pass behind the inverse transforms (dither, N:1 mix, sub-sample
// delay), 4 the per-filter level kernels in front of the MAC (N-way input mixes, cascades, cross-fades)
constexpr int EV_PAIRS = 5, EV_PER_BLOCK = 2 * EV_PAIRS;

// Host mirror of the reference's integer delay buffer (delay.c:29-45, 229-340, 346-411): same
// state variables, same decisions; the byte moves themselves are emitted as ByteOps that a
// device kernel executes on buffers in HBM.  Contiguous buffers only (how filter_process() uses
// it for channels that share a physical channel, bfrun.c:1517-1521, 1948).
struct DelayLine {
    int F = 0, ss = 0, maxdelay = 0, curdelay = 0, cur = 0, n_full = 0, n_full_cap = 0, n_rest = 0;
    uint8_t *arena = nullptr;
    std::vector<uint8_t *> full;
    uint8_t *rest = nullptr, *shrt[2] = {nullptr, nullptr}, *tmp = nullptr;

    size_t frag() const { return (size_t)F * ss; }

    bool host = false;                 // self test: arena in host memory, moves executed on the host

    int init(int fragment, int initdelay, int maxd, int sample_size) {
        F = fragment; ss = sample_size;
        int delay = maxd <= 0 ? initdelay : maxd;                      // delay.c:357-360
        if (maxd >= 0 && delay > maxd) delay = initdelay = maxd;
        if (maxd > 0 && initdelay > maxd) initdelay = maxd;            // (the reference overruns its buffer here)
        curdelay = initdelay; maxdelay = maxd;
        n_full_cap = delay > F ? delay / F + 1 : 0;
        const size_t total = (size_t)(n_full_cap + 4) * frag();
        if (host) {
            arena = (uint8_t *)calloc(total, 1);
            if (!arena) return BFHIP_ENOMEM;
        } else {
            if (dev_alloc((void **)&arena, total) != hipSuccess) return BFHIP_ENOMEM;
            if (hipMemset(arena, 0, total) != hipSuccess) return BFHIP_EHIP;
        }
        uint8_t *p = arena;
        for (int i = 0; i < n_full_cap; i++) { full.push_back(p); p += frag(); }
        rest = p; p += frag();
        shrt[0] = p; p += frag();
        shrt[1] = p; p += frag();
        tmp = p;
        if (delay == 0) return BFHIP_OK;
        if (delay <= F) { n_rest = initdelay; return BFHIP_OK; }       // :365-374
        n_rest = initdelay % F;
        n_full = initdelay / F + 1;
        if (n_full == 1) n_full = 0;
        return BFHIP_OK;
    }

    static void op(std::vector<ByteOp> &ops, uint8_t *dst, const uint8_t *src, size_t n) {
        if (n == 0) return;
        ByteOp o; o.dst = dst; o.src = src; o.n = (unsigned int)n; o.pad = 0;
        ops.push_back(o);
    }

    void retarget(int newdelay, std::vector<ByteOp> &ops) {           // change_delay, :283-318
        if (newdelay == curdelay || newdelay > maxdelay) return;
        if (newdelay <= F) {
            n_rest = newdelay;
            if (curdelay > F || curdelay < newdelay) {
                op(ops, shrt[0], nullptr, (size_t)newdelay * ss);
                op(ops, shrt[1], nullptr, (size_t)newdelay * ss);
            }
            n_full = 0; cur = 0; curdelay = newdelay;
            return;
        }
        n_rest = newdelay % F;
        n_full = newdelay / F + 1;
        if (curdelay < newdelay) {
            for (int i = 0; i < n_full; i++) op(ops, full[i], nullptr, frag());
            if (n_rest != 0) op(ops, rest, nullptr, (size_t)n_rest * ss);
        }
        cur = 0; curdelay = newdelay;
    }

    void update(uint8_t *buf, int delay, std::vector<ByteOp> &ops) {  // delay_update, :320-340
        retarget(delay, ops);
        const size_t rr = (size_t)n_rest * ss;
        if (n_full > 0) {                                              // update_delay_buffer
            uint8_t *last = cur == n_full - 1 ? full[0] : full[cur + 1];
            op(ops, full[cur], buf, frag());
            if (rr != 0) {
                op(ops, buf, rest, rr);
                op(ops, rest, last + (frag() - rr), rr);
            }
            op(ops, buf + rr, last, frag() - rr);
            if (++cur == n_full) cur = 0;
        } else if (n_rest > 0) {                                       // update_delay_short_buffer
            op(ops, shrt[cur], buf + (frag() - rr), rr);
            op(ops, tmp, buf, frag() - rr);                            // shift_samples via a
            op(ops, buf + rr, tmp, frag() - rr);                       // scratch copy
            cur = !cur;
            op(ops, buf, shrt[cur], rr);
        }
    }
};

}  // namespace

struct bfhip_engine {
    int device = 0;
    pid_t owner = 0;               // the process that created the engine (HIP state does not survive fork())
    int L = 0, N = 0, rs = 4, log2L = 0;
    int n_ch[2] = {0, 0};
    std::vector<bfhip_format> fmt[2];
    double safety_limit = 0.0;
    // `powersave:` (bfconf.c:1549-1561): 0 off, >= 1 exact-zero windows, < 1 linear noise floor
    double powersave = 0.0;
    int *d_ps_flags = nullptr, *d_ps_live = nullptr;
    unsigned long long *d_ps_acc = nullptr;    // partition lengths above 8192: [2][n_in] running maxima
    double *d_ps_scale = nullptr;
    std::vector<Coeff> coeffs;
    // Coefficient sets are carved out of a few large slabs instead of one hipMalloc each: the MAC
    // streams all of them at once (config C: 4096 sets, 8 GiB), and thousands of separate 2 MiB
    // allocations cost it 5-10 % (one hipMalloc per set: 0.79 - 0.86 of peak box to box, one slab
    // 0.886).  Where inside the slab a set starts matters as much -- the stream-ordered
    // copy below takes that out of the host's hands.
    struct Slab { void *base = nullptr; size_t cap = 0, used = 0; };
    std::vector<Slab> slabs;
    bool coeff_arena = true;               // BFHIP_COEFF_ARENA=0: one hipMalloc per set
    unsigned int skew_state = 12345u;
    // Stream-ordered copy of the coefficients for uniform crossbar plans (StreamLayout, kernels.h):
    // every MAC workgroup reads one sequential slice.  Built / refreshed by build_plan.
    int stream_wanted = 1;                 // BFHIP_COEFF_STREAM: 0 off, 1 from 64 MiB of coefficients up, 2 always
    StreamLayout hstream = {nullptr, 0, 0, 0};
    void *d_stream = nullptr;
    size_t stream_cap = 0;
    StreamWhere *d_where = nullptr;
    int *d_which = nullptr;
    size_t where_cap = 0;
    std::vector<std::array<const void *, OG>> stream_keys;     // per flat entry: the sets laid out there
    int stream_geom[5] = {0, 0, 0, 0, 0};                          // entries, E_c, P, n_tc, n_groups
    bool any_watched = false;
    unsigned long long watch_seq = 0;       // bfhip_coeff_dirty_sequence() at the last poll
    unsigned long long watch_lost = 0;      // bfhip_dirty_lost() at the last poll
    std::vector<Filter> filters;
    // shard of a configuration (multi-process host): which outputs this engine converts and writes.
    // -1 = derive at finalize (fed by an active filter, or by no filter at all), 0 / 1 = set by the host
    std::vector<signed char> out_active_set;
    std::vector<char> out_active;          // after finalize
    bool sharded = false;                  // some output belongs to another engine
    struct OwnedRun { size_t offset, len, stride; };       // bytes: per frame `len` at `offset`, frames `stride` apart
    std::vector<OwnedRun> owned_runs;      // the raw output samples this engine owns (merged channel runs)
    std::vector<unsigned char> h_stage;    // host staging for bfhip_engine_block of a sharded engine
    bool finalized = false, finalize_failed = false, plan_dirty = true;
    unsigned int blockcounter = 0;
    // ... which wraps by a multiple of every ring depth (N, and N + 1 when a spare slot exists)
    // long before 2^32, so that `blockcounter mod depth` never jumps (BlockState, kernels.h)
    unsigned int wrap_at = 0, wrap_by = 0;
    unsigned long long blocks_done = 0;      // since creation (procblocks analogue)

    hipStream_t stream = nullptr;      // main stream: per-filter kernels and the crossbar MAC
    bool own_stream = false;
    hipStream_t ls = nullptr;          // stream the next launch goes to
    // Pipelined block (bfhip_engine_block[_dev]): the input FFT of block t+1 and the inverse
    // FFT / requantiser of block t-1 run on their own streams beside the HBM-bound MAC of
    // block t, the way the reference overlaps its input, filter and output processes
    // (bfrun.c:2312-2616).  Needs one spare ring slot (R = N + 1) and two Zp buffers.
    bool pipelined = false;            // decided at finalize (or BFHIP_OVERLAP=0/1)
    int overlap_mode = -1;             // -1 auto, 0 off, 1 on (bfhip_engine_set_overlap)
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_mac[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    int R = 0;                         // depth of the input rings
    void *d_Zp2 = nullptr;             // second partial-spectra buffer
    // Deferred output (large crossbars on one stream): the inverse transforms of block t run in
    // ONE launch with the forward transforms of block t+1 (io_wave_kernel) -- both are a handful
    // of workgroups that would otherwise run back to back, each alone on the chip, in front of
    // and behind the millisecond MAC.  The output of a block is therefore written during the
    // NEXT bfhip_engine_block_dev call, or by bfhip_engine_sync (which flushes it).
    bool defer_out = false;            // decided at finalize (BFHIP_DEFER=0/1 forces)
    // Two-stream ping-pong (small crossbars with the wave FFT): [K3 of t-2 | K1 of t] in one launch
    // on the side stream s_in while the MAC of t-1 runs on the main stream; the MAC of t waits for
    // that launch.  Two launches and four event calls per block instead of three launches and
    // seven, and the two transforms run side by side.  Outputs are owed for two calls.
    bool pipe2 = false;                // decided at finalize (BFHIP_PIPE2=0 keeps the three-stream schedule)
    void *d_Zp3 = nullptr;
    hipEvent_t ev_io[2] = {nullptr, nullptr}, ev_mac3[3] = {nullptr, nullptr, nullptr};
    struct Pending {
        void *Zp = nullptr; size_t chunk_stride = 0; int n_chunks = 0;
        void *rawout = nullptr; hipEvent_t out_done = nullptr;
        hipEvent_t mac_done = nullptr;     // pipe2: the side stream waits for it before the inverse transforms
    };
    std::deque<Pending> pendq;         // outputs owed, oldest first (deferred: at most 1, pipe2: at most 2)

    // partition lengths above the LDS limit (bigfft.h): global scratch for [transform][L] complex
    bool big = false;
    int big_R = 1;                 // L / 8192
    void *d_big[3] = {nullptr, nullptr, nullptr};   // zin, zmid, zout
    size_t big_cap = 0;            // transforms the scratch holds
    void *d_tw13 = nullptr;        // twiddle table of the 8192-point LDS transform
    // K1 / K3 on the wave FFT (fft_wave.h: L = 1024 .. 8192, default from 4096 up; BFHIP_FFT_WAVE=0/1)
    bool wave = false;
    void *d_tww = nullptr;         // its twiddle table

    // device state
    void *d_tw = nullptr;          // [2L] complex
    void *d_prev = nullptr;        // [n_in][L] real
    void *d_ring = nullptr;        // [n_in][N][L] complex
    DevFormat *d_fmt[2] = {nullptr, nullptr};
    DevOverflow *d_over = nullptr;
    int *d_status = nullptr;
    int *d_status_own = nullptr;   // the engine's allocation while d_status points at a caller's word
    int *d_bad = nullptr;
    void *d_Zp = nullptr;          // [n_chunks][n_out_padded][L] complex
    size_t zp_bytes = 0;
    void *d_entries = nullptr;
    size_t entries_cap = 0;
    ChunkRange *d_chunks = nullptr;
    bool zp_is_sum = false;            // chunk 0 of the last block's partial-sum buffer holds the sum over all chunks (launch_sum in place)
    size_t chunks_cap = 0;
    uint8_t *d_rawin = nullptr, *d_rawout = nullptr;
    size_t raw_bytes[2] = {0, 0};
    void *d_taps = nullptr;
    size_t taps_cap = 0;

    // filters that need more than the shared-ring fast path (SURVEY A3 N-way mix, A7, A8)
    std::vector<int> level, owner_index, y_index, sink_index, fade_index;
    std::vector<char> is_source;
    std::vector<void *> promoted;      // private ring given to a shared-ring filter at run time
    int n_levels = 0, n_owners = 0, n_y = 0, n_sinks = 0, n_fadeable = 0;
    void *d_fring = nullptr;       // [n_owners][N][L] complex: private rings
    void *d_Y = nullptr;           // [n_y][L] complex: materialised filter outputs
    void *d_Yold = nullptr;        // [n_fadeable][L] complex: old-coefficient result in a fade block
    void *d_evalprev = nullptr;    // [n_sinks][L] real: previous valid half (convolve_eval state)
    void *d_jobs = nullptr;        // fill jobs | mix sources | filter jobs | fade jobs
    size_t jobs_cap = 0;
    struct LevelJobs { size_t fill_off = 0; int n_fill = 0; size_t filt_off = 0; int n_filt = 0;
                       size_t fade_off = 0; int n_fade = 0; };
    std::vector<LevelJobs> level_jobs;
    size_t src_off = 0;
    bool any_fading = false;

    // N:1 virtual -> physical channels (bfconf->virt2phys): fmt[] is indexed by PHYSICAL channel
    int n_phys[2] = {0, 0};
    std::vector<int> v2p[2], n_vpp[2], vdelay[2], vmaxdelay[2], vmuted[2];
    std::vector<DelayLine> vline[2];
    std::vector<int> vin_list;             // virtual inputs that share a physical one
    std::vector<std::vector<int>> vout_groups;   // members of every shared physical output
    uint8_t *d_incopy = nullptr;           // [vin_list.size()][L * 8]
    void *d_vjobs = nullptr;               // per-block job/op tables (device): input half | output half
    size_t vjobs_slot = 0;
    Stage st_vin, st_vout, st_sd[2];       // their pinned staging, one slot per block in flight
    bool has_vchan = false;

    // sub-sample delay (sdf_length / sdf_beta / per-channel subdelay; delay.c:416-505)
    int sdf_length = 0, sd_flen = 0, sd_bs = 0;
    std::vector<int> subdelay[2];          // current value per virtual channel (-100 = undefined)
    std::vector<int> sd_slot[2];           // filter slot of a channel, -1 = none (fixed at finalize)
    void *d_sd_bank = nullptr;             // [199][flen] taps, index 99 + subdelay
    void *d_sd_rest[2] = {nullptr, nullptr};
    void *d_sdin = nullptr;                // [n filtered inputs][L] reals, what K1 reads
    void *d_sdjobs[2] = {nullptr, nullptr};

    // HP-TPDF dither (dither.c, dither.h)
    std::vector<int> dither_channels;      // output channel of each dither slot (virtual, ascending, after finalize)
    std::vector<int> dither_rank;          // its rank among the dithered PHYSICAL outputs: where its table walk starts
    std::vector<char> dither_late;         // slot is the head of a shared / sub-sample-filtered output: dithered after the mix
    std::vector<int8_t> dither_table;
    int dither_spacing = 0;
    int *d_dither_ch = nullptr;
    void *d_dither_state = nullptr;
    int8_t *d_dither_table = nullptr;
    void *d_randmap = nullptr;             // 512 entries; centre at +256
    unsigned char *d_skip_quant = nullptr; // [n_out] 1 = the dither pass writes this channel
    void *d_timeout = nullptr;             // [n_out][L] time samples for the dither pass

    // plan geometry
    int n_groups = 0, n_out_padded = 0, n_chunks = 1, n_tiles = 1, mac_threads = 256;
    int n_entries = 0;
    bool mac_nt = true;            // non-temporal coefficient loads (BFHIP_MAC_NT=0 turns them off)
    int mac_unroll = 0;            // 0: rotating three-stage pipeline; 1..4: plain unroll (BFHIP_MAC_UNROLL, tools/tune_mac.py)
    // wide interleaved sides (>= 128 channels in one uniform frame; BFHIP_WIDE_IO=0/1): the raw frames are
    // transposed to / from a planar copy by a coalesced pass of their own (transpose_words_kernel)
    bool wide[2] = {false, false};
    uint8_t *d_planar[2] = {nullptr, nullptr};     // [n_phys][L] words of the side's sample size
    unsigned char *d_phys_skip = nullptr;          // [n_phys out] 1 = another engine's channel (shards)
    // two blocks per pass over the coefficients (bfhip_engine_enable_pairs / bfhip_engine_block_pair_dev)
    bool pairs = false;
    unsigned long long n_pair_launches = 0;
    bool all_dense = false;        // every MAC entry takes the crossbar path
    // one-to-one plans (every output fed by at most one single-term entry per chunk): mac_diag_kernel,
    // one workgroup per (chunk, output) walking whole spectra (BFHIP_MAC_DIAG=0 keeps the crossbar kernel)
    bool mac_diag = false;
    const int *d_diag_jobs = nullptr;      // [n_chunks * n_out_padded] entry index or -1 (lives behind d_chunks)
    double alg_bytes_total = 0, alg_bytes_mac = 0;

    // real-time mode (bfhip_engine_rt_*): pinned host double buffer, the block's launch
    // sequence replayed from a HIP graph, completion signalled through pinned memory
    struct Rt {
        bool on = false;
        int flags = 0;
        void *h_in[2] = {nullptr, nullptr}, *h_out[2] = {nullptr, nullptr};
        DevOverflow *h_over[2] = {nullptr, nullptr};
        int *h_status[2] = {nullptr, nullptr};      // [0] status bits, [1] sequence number
        hipGraphExec_t exec[2] = {nullptr, nullptr};
        bool valid[2] = {false, false};
        bool primed = false;                        // one direct block ran since the last plan change
        hipEvent_t done[2] = {nullptr, nullptr};
        unsigned long long submitted = 0, waited = 0;
        unsigned int bs_t = 0;                      // what d_bs->t holds
        bool bs_synced = false;
        unsigned long long n_graph = 0, n_direct = 0, n_capture = 0;
        // what rt_wait last wrote into the caller's overflow array: an entry that differs from it
        // next time was changed by the HOST (bf_reset_peak, bfrun.c: the CLI's peak reset) and
        // becomes the device's state for that output
        std::vector<DevOverflow> last_over;
        std::vector<unsigned long long> over_from;
        bool last_valid = false;
        // BFHIP_RT_OVERLAP: copies of neighbouring periods run on the copy engines beside compute
        hipStream_t s_h2d = nullptr, s_d2h = nullptr;
        hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_cmp[2] = {nullptr, nullptr};
        uint8_t *d_in[2] = {nullptr, nullptr}, *d_out[2] = {nullptr, nullptr};
    } rt;
    BlockState *d_bs = nullptr;
    unsigned int *d_rt_arrive = nullptr;
    BlockState *bs_arg = nullptr;      // non-null while a graph is being captured

    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;    // EV_PER_BLOCK per block: start/stop of the EV_PAIRS stages on their streams
    int ev_used = 0;
    int timing_stride = 1;         // time every n-th block (the event records cost ~3 us each)
    bool timed_now = false;
    int timed_mask = 0;            // which of the three kernel pairs of the block in progress were recorded
    unsigned long long timed_for = ~0ull;   // blocks_done the decision above was taken for
    std::vector<unsigned char> ev_mask;     // [MAX_TIMED] timed_mask of every timed block

    size_t csize() const { return (size_t)2 * rs; }     // bytes per complex
};

namespace {

int sync_all(bfhip_engine *e) {
    if (e->s_in) HIPCHK(hipStreamSynchronize(e->s_in));
    if (e->stream) HIPCHK(hipStreamSynchronize(e->stream));
    if (e->s_out) HIPCHK(hipStreamSynchronize(e->s_out));
    return BFHIP_OK;
}

// ---------------------------------------------------------------- coefficient memory

void *coeff_alloc(bfhip_engine *e, size_t bytes) {
    if (!e->coeff_arena) {
        void *p = nullptr;
        return dev_alloc(&p, bytes) == hipSuccess ? p : nullptr;
    }
    size_t need = (bytes + 65535) & ~(size_t)65535;                // sets start on 64 KiB
    size_t skew = 0;
    if (const char *env = getenv("BFHIP_COEFF_PAD_B")) {
        if (env[0] == 'r') {
            // pseudo-random start inside an extra 64 KiB (256-byte steps)
            e->skew_state = e->skew_state * 1664525u + 1013904223u;
            skew = (size_t)((e->skew_state >> 16) & 255u) * 256;
            need = ((bytes + 255) & ~(size_t)255) + 65536;
        } else need = (bytes + (size_t)atol(env) + 255) & ~(size_t)255;
    }
    if (!e->slabs.empty()) {
        auto &sl = e->slabs.back();
        if (sl.cap - sl.used >= need) { void *p = (unsigned char *)sl.base + sl.used + skew; sl.used += need; return p; }
    }
    // a first slab of 64 MiB keeps small engines small; whoever needs more gets 2 GiB pieces
    // (config C, MAC alone, one box: one hipMalloc per set 0.83 of peak, 512 MiB slabs 0.86,
    // >= 2 GiB slabs 0.883-0.886).  A host that knows the total calls bfhip_engine_reserve_coeffs.
    size_t cap = e->slabs.empty() ? ((size_t)64 << 20) : ((size_t)2 << 30);
    if (const char *env = getenv("BFHIP_COEFF_SLAB_MB")) cap = (size_t)std::max(1, atoi(env)) << 20;
    cap = std::max(cap, need);
    bfhip_engine::Slab sl;
    while (dev_alloc(&sl.base, cap) != hipSuccess) {
        (void)hipGetLastError();
        if (cap <= need) return nullptr;
        cap = std::max(need, cap / 2);
    }
    sl.cap = cap; sl.used = need;
    e->slabs.push_back(sl);
    return (unsigned char *)sl.base + skew;
}

// give back a set that was allocated last (error paths); anything else stays with its slab
void coeff_release(bfhip_engine *e, void *p, size_t bytes) {
    if (!p) return;
    if (!e->coeff_arena) { (void)hipFree(p); return; }
    if (e->slabs.empty()) return;
    auto &sl = e->slabs.back();
    const size_t need = (bytes + 65535) & ~(size_t)65535;
    if (sl.used >= need && (unsigned char *)sl.base + sl.used - need == (unsigned char *)p) sl.used -= need;
}

// ---------------------------------------------------------------- template dispatch

#define DISPATCH_LOG2L(T, FN, ...)                        \
    switch (e->log2L) {                                   \
    case 2: FN<T, 2>(__VA_ARGS__); break;                 \
    case 3: FN<T, 3>(__VA_ARGS__); break;                 \
    case 4: FN<T, 4>(__VA_ARGS__); break;                 \
    case 5: FN<T, 5>(__VA_ARGS__); break;                 \
    case 6: FN<T, 6>(__VA_ARGS__); break;                 \
    case 7: FN<T, 7>(__VA_ARGS__); break;                 \
    case 8: FN<T, 8>(__VA_ARGS__); break;                 \
    case 9: FN<T, 9>(__VA_ARGS__); break;                 \
    case 10: FN<T, 10>(__VA_ARGS__); break;               \
    case 11: FN<T, 11>(__VA_ARGS__); break;               \
    case 12: FN<T, 12>(__VA_ARGS__); break;               \
    case 13: FN<T, 13>(__VA_ARGS__); break;               \
    default: break;                                       \
    }

#define DISPATCH(FN, ...)                                                   \
    do {                                                                    \
        if (e->rs == 4) { DISPATCH_LOG2L(float, FN, __VA_ARGS__) }          \
        else { DISPATCH_LOG2L(double, FN, __VA_ARGS__) }                    \
    } while (0)

// raise a kernel's dynamic-LDS limit once per (device, kernel, size): the attribute call costs a
// few microseconds, which is real money in a 35 us block
inline PowerSave ps_arg(const bfhip_engine *e) {
    PowerSave ps;
    ps.thr = e->d_ps_flags ? e->powersave : 0.0;
    ps.scale = e->d_ps_scale; ps.flags = e->d_ps_flags; ps.live = e->d_ps_live;
    return ps;
}

template <typename K> hipError_t allow_lds(K kernel, size_t bytes) {
    static std::mutex mu;
    static std::set<std::tuple<int, const void *, size_t>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const auto key = std::make_tuple(dev, reinterpret_cast<const void *>(kernel), bytes);
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(key)) return hipSuccess;
    const hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (r == hipSuccess) done.insert(key);
    return r;
}

template <typename T, int LOG2L>
void launch_fft_in(bfhip_engine *e, const uint8_t *raw, int slot, hipError_t *err) {
    constexpr int NT = fft_threads<T>(LOG2L);
    const size_t lds = lds_fft_bytes(LOG2L, sizeof(c2<T>));
    auto k = fft_in_kernel<T, LOG2L>;
    *err = allow_lds(k, lds);
    if (*err != hipSuccess) return;
    hipLaunchKernelGGL(k, dim3(e->n_ch[0]), dim3(NT), lds, e->ls, raw, e->d_fmt[0],
                       (T *)e->d_prev, (c2<T> *)e->d_ring, (const c2<T> *)e->d_tw, e->R, slot,
                       (const BlockState *)e->bs_arg, ps_arg(e));
    *err = hipGetLastError();
}

#define DISPATCH_WAVE_T(T, FN, ...)                        \
    switch (e->log2L) {                                    \
    case 10: FN<T, 10>(__VA_ARGS__); break;                \
    case 11: FN<T, 11>(__VA_ARGS__); break;                \
    case 12: FN<T, 12>(__VA_ARGS__); break;                \
    case 13: FN<T, 13>(__VA_ARGS__); break;                \
    default: break;                                        \
    }
#define DISPATCH_WAVE(FN, ...)                                              \
    do {                                                                    \
        if (e->rs == 4) { DISPATCH_WAVE_T(float, FN, __VA_ARGS__) }         \
        else { DISPATCH_WAVE_T(double, FN, __VA_ARGS__) }                   \
    } while (0);

template <typename T, int LOG2L>
void launch_fft_in_wave(bfhip_engine *e, const uint8_t *raw, int slot, hipError_t *err) {
    constexpr int NT = WaveGeo<LOG2L>::NT;
    const size_t lds = lds_fft_bytes(LOG2L, sizeof(c2<T>));
    auto k = fft_in_wave_kernel<T, LOG2L>;
    *err = allow_lds(k, lds);
    if (*err != hipSuccess) return;
    hipLaunchKernelGGL(k, dim3(e->n_ch[0]), dim3(NT), lds, e->ls, raw, e->d_fmt[0],
                       (T *)e->d_prev, (c2<T> *)e->d_ring, (const c2<T> *)e->d_tww, e->R, slot,
                       (const BlockState *)e->bs_arg, ps_arg(e));
    *err = hipGetLastError();
}

template <typename T, int LOG2L>
void launch_ifft_out_wave(bfhip_engine *e, const void *Zp, size_t chunk_stride, int n_chunks,
                          int first, int count, uint8_t *raw, hipError_t *err) {
    constexpr int NT = WaveGeo<LOG2L>::NT;
    const size_t lds = lds_fft_bytes(LOG2L, sizeof(c2<T>));
    auto k = ifft_out_wave_kernel<T, LOG2L>;
    *err = allow_lds(k, lds);
    if (*err != hipSuccess) return;
    hipLaunchKernelGGL(k, dim3(count), dim3(NT), lds, e->ls, (const c2<T> *)Zp, chunk_stride,
                       n_chunks, first, e->d_fmt[1], e->d_over, (const unsigned char *)e->d_skip_quant,
                       raw, e->d_timeout ? (T *)e->d_timeout + (size_t)first * e->L : (T *)nullptr,
                       (const c2<T> *)e->d_tww, e->safety_limit, e->d_status);
    *err = hipGetLastError();
}

// K3 of an earlier block (count channels from Zp) + K1 of the current block in one launch
template <typename T, int LOG2L>
void launch_io_wave(bfhip_engine *e, const void *Zp, size_t chunk_stride, int n_chunks, int first, int count,
                    uint8_t *rawout, const uint8_t *rawin, int slot, hipError_t *err) {
    constexpr int NT = WaveGeo<LOG2L>::NT;
    const size_t lds = lds_fft_bytes(LOG2L, sizeof(c2<T>));
    auto k = io_wave_kernel<T, LOG2L>;
    *err = allow_lds(k, lds);
    if (*err != hipSuccess) return;
    hipLaunchKernelGGL(k, dim3(count + e->n_ch[0]), dim3(NT), lds, e->ls, count,
                       (const c2<T> *)Zp, chunk_stride, n_chunks, first, e->d_fmt[1], e->d_over,
                       (const unsigned char *)e->d_skip_quant, rawout,
                       e->d_timeout ? (T *)e->d_timeout + (size_t)first * e->L : (T *)nullptr,
                       e->safety_limit, e->d_status,
                       rawin, e->d_fmt[0], (T *)e->d_prev, (c2<T> *)e->d_ring, e->R, slot,
                       (const c2<T> *)e->d_tww, ps_arg(e), (const BlockState *)e->bs_arg);
    *err = hipGetLastError();
}

template <typename T, int LOG2L>
void launch_coeff_prep(bfhip_engine *e, const void *taps, int n_taps, double scale, void *H,
                       int n_blocks, hipError_t *err) {
    constexpr int NT = fft_threads<T>(LOG2L);
    const size_t lds = lds_fft_bytes(LOG2L, sizeof(c2<T>));
    auto k = coeff_prep_kernel<T, LOG2L>;
    *err = allow_lds(k, lds);
    if (*err != hipSuccess) return;
    hipLaunchKernelGGL(k, dim3(n_blocks), dim3(NT), lds, e->stream, (const T *)taps, n_taps,
                       (T)scale, (c2<T> *)H, (const c2<T> *)e->d_tw, e->d_bad);
    *err = hipGetLastError();
}

// the HP-TPDF pass over the channels of [first, first+count) that dither (after K3)
template <typename T>
void launch_dither(bfhip_engine *e, int first, int count, uint8_t *raw, hipError_t *err, bool late = false) {
    if (e->dither_channels.empty()) return;
    // dither slots whose channel lies in [first, first+count) (the slots are sorted by channel), in
    // runs of the requested phase: straight after K3, or after the N:1 mix / sub-sample filter
    const int n = (int)e->dither_channels.size();
    for (int s0 = 0; s0 < n;) {
        const bool in = e->dither_channels[s0] >= first && e->dither_channels[s0] < first + count &&
                        (bool)e->dither_late[s0] == late;
        if (!in) { s0++; continue; }
        int s1 = s0 + 1;
        while (s1 < n && e->dither_channels[s1] < first + count && (bool)e->dither_late[s1] == late) s1++;
        hipLaunchKernelGGL(dither_kernel<T>, dim3(s1 - s0), dim3(64), 0, e->ls,
                           (const T *)e->d_timeout, (const int *)e->d_dither_ch + s0,
                           (DitherState<T> *)e->d_dither_state + s0, (const int8_t *)e->d_dither_table,
                           (int)e->dither_table.size(), (const T *)e->d_randmap + 256,
                           e->d_fmt[1], e->d_over, raw, e->L, e->safety_limit, e->d_status);
        if ((*err = hipGetLastError()) != hipSuccess) return;
        s0 = s1;
    }
}

template <typename T, int LOG2L>
void launch_ifft_out(bfhip_engine *e, const void *Zp, size_t chunk_stride, int n_chunks,
                     int first, int count, uint8_t *raw, hipError_t *err) {
    constexpr int NT = fft_threads<T>(LOG2L);
    const size_t lds = lds_fft_bytes(LOG2L, sizeof(c2<T>));
    auto k = ifft_out_kernel<T, LOG2L>;
    *err = allow_lds(k, lds);
    if (*err != hipSuccess) return;
    hipLaunchKernelGGL(k, dim3(count), dim3(NT), lds, e->ls, (const c2<T> *)Zp, chunk_stride,
                       n_chunks, first, e->d_fmt[1], e->d_over, (const unsigned char *)e->d_skip_quant,
                       raw, e->d_timeout ? (T *)e->d_timeout + (size_t)first * e->L : (T *)nullptr,
                       (const c2<T> *)e->d_tw, e->safety_limit, e->d_status);
    *err = hipGetLastError();          // (the dither pass behind it: do_outputs)
}

template <typename T, int LOG2L>
void launch_io(bfhip_engine *e, const void *z, int first, int count, uint8_t *rawout,
               const uint8_t *rawin, int slot, hipError_t *err) {
    constexpr int NT = fft_threads<T>(LOG2L);
    const size_t lds = lds_fft_bytes(LOG2L, sizeof(c2<T>));
    auto k = io_kernel<T, LOG2L>;
    *err = allow_lds(k, lds);
    if (*err != hipSuccess) return;
    hipLaunchKernelGGL(k, dim3(count + e->n_ch[0]), dim3(NT), lds, e->ls, count,
                       (const c2<T> *)z, first, e->d_fmt[1], e->d_over, (const unsigned char *)e->d_skip_quant,
                       rawout, e->d_timeout ? (T *)e->d_timeout + (size_t)first * e->L : (T *)nullptr,
                       e->safety_limit, e->d_status,
                       rawin, e->d_fmt[0], (T *)e->d_prev, (c2<T> *)e->d_ring, e->R, slot, (const c2<T> *)e->d_tw, ps_arg(e));
    *err = hipGetLastError();
}

template <typename T>
void launch_mac(bfhip_engine *e, void *Zp, hipError_t *err) {
    const int n_tc = e->n_tiles * e->n_chunks;
    const int tc8 = (n_tc + 7) / 8;
    const int grid = tc8 * e->n_groups * 8;
    const unsigned long long age64 = std::min<unsigned long long>(e->blocks_done + 1, (unsigned long long)e->N);
#define BFHIP_LAUNCH_MAC(NTFLAG, U)                                                                   \
    hipLaunchKernelGGL((mac_xbar_kernel<T, NTFLAG, U>), dim3(grid), dim3(e->mac_threads), 0, e->ls,      \
                       (const MacEntry<T> *)e->d_entries, (const ChunkRange *)e->d_chunks,               \
                       (c2<T> *)Zp, e->L, e->n_out_padded, e->n_groups, e->n_chunks, n_tc,               \
                       e->blockcounter, (int)age64, (const BlockState *)e->bs_arg,                       \
                       (NTFLAG && U == 0) ? e->hstream : StreamLayout{nullptr, 0, 0, 0})
    if (e->mac_diag) {
        const int n_jobs = e->n_chunks * e->n_out_padded;
        // the spectrum of a job goes to `tsplit` workgroups (consecutive runs of tiles): ~512 workgroups
        // in all, no partial sums -- runs of 16 - 64 KiB either way (BFHIP_DIAG_TSPLIT: 1, 2, 4 ...)
        const int all_tiles = std::max(1, e->L / (256 * (int)(16 / sizeof(c2<T>))));
        int tsplit = 1;
        while (tsplit < all_tiles && tsplit < 8 && n_jobs * tsplit < 512) tsplit *= 2;
        if (const char *env = getenv("BFHIP_DIAG_TSPLIT")) {
            tsplit = 1;
            while (tsplit * 2 <= std::min(all_tiles, atoi(env))) tsplit *= 2;
        }
        while (all_tiles / tsplit > 16) tsplit *= 2;            // float64 at L = 8192: 32 tiles, 16 per workgroup
        const int tiles = all_tiles / tsplit;
#define BFHIP_LAUNCH_DIAG(NTFLAG, TL)                                                                                  \
        hipLaunchKernelGGL((mac_diag_kernel<T, NTFLAG, TL>), dim3(n_jobs * tsplit), dim3(256), 0, e->ls,               \
                           (const MacEntry<T> *)e->d_entries, e->d_diag_jobs, (c2<T> *)Zp, e->L, e->n_out_padded,      \
                           e->blockcounter, (int)age64, (const BlockState *)e->bs_arg, tsplit)
#define BFHIP_DIAG_TILES(NTFLAG)                                              \
        switch (tiles) {                                                      \
        case 1: BFHIP_LAUNCH_DIAG(NTFLAG, 1); break;                          \
        case 2: BFHIP_LAUNCH_DIAG(NTFLAG, 2); break;                          \
        case 4: BFHIP_LAUNCH_DIAG(NTFLAG, 4); break;                          \
        case 8: BFHIP_LAUNCH_DIAG(NTFLAG, 8); break;                          \
        default: BFHIP_LAUNCH_DIAG(NTFLAG, 16); break;                        \
        }
        if (e->mac_nt) { BFHIP_DIAG_TILES(true) } else { BFHIP_DIAG_TILES(false) }
#undef BFHIP_DIAG_TILES
#undef BFHIP_LAUNCH_DIAG
    }
    else if (!e->mac_nt) BFHIP_LAUNCH_MAC(false, 2);
    // the pipelined variant needs ~290 VGPRs: worth it for the pure crossbar, a loss of occupancy
    // for plans that (also) run the latency-bound per-term paths
    else if (e->mac_unroll == 0 && e->all_dense) BFHIP_LAUNCH_MAC(true, 0);
    else if (e->mac_unroll == 0) BFHIP_LAUNCH_MAC(true, 2);
    else if (e->mac_unroll == 4) BFHIP_LAUNCH_MAC(true, 4);
    else if (e->mac_unroll == 3) BFHIP_LAUNCH_MAC(true, 3);
    else if (e->mac_unroll == 1) BFHIP_LAUNCH_MAC(true, 1);
    else BFHIP_LAUNCH_MAC(true, 2);
#undef BFHIP_LAUNCH_MAC
    *err = hipGetLastError();
}

template <typename T>
void launch_mac2(bfhip_engine *e, void *Zp0, void *Zp1, hipError_t *err) {
    const int n_tc = e->n_tiles * e->n_chunks;
    e->zp_is_sum = false;
    const int grid = ((n_tc + 7) / 8) * e->n_groups * 8;
    hipLaunchKernelGGL((mac_xbar2_kernel<T, true>), dim3(grid), dim3(e->mac_threads), 0, e->ls,
                       (const MacEntry<T> *)e->d_entries, (const ChunkRange *)e->d_chunks, (c2<T> *)Zp0, (c2<T> *)Zp1,
                       e->L, e->n_out_padded, e->n_groups, e->n_chunks, n_tc, e->blockcounter, e->hstream);
    *err = hipGetLastError();
}

template <typename T>
void launch_sum(bfhip_engine *e, const void *Zp, void *Z, hipError_t *err) {
    const size_t n_per_chunk = (size_t)e->n_out_padded * e->L;
    const size_t n_valid = (size_t)e->n_ch[1] * e->L;
    const int grid = (int)((n_valid + 255) / 256);
    hipLaunchKernelGGL(sum_partials_kernel<T>, dim3(grid), dim3(256), 0, e->ls,
                       (const c2<T> *)Zp, (c2<T> *)Z, n_per_chunk, n_valid, e->n_chunks);
    if (Z == Zp) e->zp_is_sum = true;
    *err = hipGetLastError();
}

template <typename T, int LOG2L>
void launch_levels(bfhip_engine *e, hipError_t *err) {
    constexpr int NT = fft_threads<T>(LOG2L);
    const size_t lds = lds_fft_bytes(LOG2L, sizeof(c2<T>));
    auto kf = ring_fill_kernel<T, LOG2L>;
    auto kx = crossfade_kernel<T, LOG2L>;
    const unsigned long long age64 = std::min<unsigned long long>(e->blocks_done + 1, (unsigned long long)e->N);
    const unsigned char *base = (const unsigned char *)e->d_jobs;
    const int V = 16 / (int)sizeof(c2<T>);
    const int threads = e->mac_threads;
    const int tiles = (e->L + threads * V - 1) / (threads * V);
    for (auto &lj : e->level_jobs) {
        if (lj.n_fill > 0) {
            if ((*err = allow_lds(kf, lds)) != hipSuccess) return;
            hipLaunchKernelGGL(kf, dim3(lj.n_fill), dim3(NT), lds, e->ls,
                               (const FillJob<T> *)(base + lj.fill_off),
                               (const MixSrc<T> *)(base + e->src_off), (const c2<T> *)e->d_tw,
                               e->N, e->blockcounter, (const BlockState *)e->bs_arg);
        }
        if (lj.n_filt > 0) {
            hipLaunchKernelGGL(mac_filter_kernel<T>, dim3(tiles, lj.n_filt), dim3(threads), 0, e->ls,
                               (const FilterJob<T> *)(base + lj.filt_off), e->L, e->blockcounter, (int)age64,
                               (const BlockState *)e->bs_arg);
        }
        if (lj.n_fade > 0) {
            if ((*err = allow_lds(kx, lds)) != hipSuccess) return;
            hipLaunchKernelGGL(kx, dim3(lj.n_fade), dim3(NT), lds, e->ls,
                               (const FadeJob<T> *)(base + lj.fade_off), (const c2<T> *)e->d_tw);
        }
        if ((*err = hipGetLastError()) != hipSuccess) return;
    }
}


// ---------------------------------------------------------------- block lengths above 8192 (bigfft.h)

// scratch for n_tr transforms of L complex points each (zin, zmid, zout)
int big_reserve(bfhip_engine *e, size_t n_tr) {
    if (n_tr <= e->big_cap) return BFHIP_OK;
    { int r = sync_all(e); if (r != BFHIP_OK) return r; }
    for (int i = 0; i < 3; i++) {
        if (e->d_big[i]) (void)hipFree(e->d_big[i]);
        e->d_big[i] = nullptr;
        HIPCHK(dev_alloc(&e->d_big[i], n_tr * (size_t)e->L * e->csize()));
    }
    e->big_cap = n_tr;
    return BFHIP_OK;
}

// zin -> zout: complex FFT of L points for n_tr transforms on `st`
template <typename T>
void big_fft(bfhip_engine *e, int n_tr, bool inv, hipStream_t st, hipError_t *err) {
    const c2<T> *zin = (const c2<T> *)e->d_big[0];
    c2<T> *zmid = (c2<T> *)e->d_big[1], *zout = (c2<T> *)e->d_big[2];
    const c2<T> *tw13 = (const c2<T> *)e->d_tw13, *twL = (const c2<T> *)e->d_tw;
    *err = inv ? big_fft_run<T, true>(zin, zmid, zout, e->log2L, n_tr, tw13, twL, st)
               : big_fft_run<T, false>(zin, zmid, zout, e->log2L, n_tr, tw13, twL, st);
}

inline dim3 big_grid_half(const bfhip_engine *e, int n_tr) { return dim3((unsigned)(e->L / 2 / 256 + 1), (unsigned)n_tr); }

template <typename T>
void launch_fft_in_big(bfhip_engine *e, const uint8_t *raw, int slot, hipError_t *err) {
    const int n = e->n_ch[0];
    const int parity = (int)(e->blocks_done & 1);
    const bool ps_on = e->d_ps_flags != nullptr;
    hipLaunchKernelGGL(big_in_pre<T>, big_grid_half(e, n), dim3(256), 0, e->ls, raw, e->d_fmt[0], (T *)e->d_prev,
                       (c2<T> *)e->d_big[0], e->L, ps_on ? e->d_ps_acc : (unsigned long long *)nullptr, n, parity,
                       e->powersave >= 1.0 ? 1 : 0);
    big_fft<T>(e, n, false, e->ls, err);
    if (*err != hipSuccess) return;
    hipLaunchKernelGGL(big_untangle<T>, big_grid_half(e, n), dim3(256), 0, e->ls, (const c2<T> *)e->d_big[2],
                       (c2<T> *)e->d_ring + (size_t)slot * e->L, (size_t)e->R * e->L, (const c2<T> *)e->d_tw, e->L, (T)1);
    if (ps_on)
        hipLaunchKernelGGL(big_ps_finish<T>, dim3((unsigned)(e->L / 256), (unsigned)n), dim3(256), 0, e->ls,
                           (const unsigned long long *)e->d_ps_acc, n, parity, ps_arg(e), (c2<T> *)e->d_ring, e->R, slot, e->L);
    *err = hipGetLastError();
}

template <typename T>
void launch_coeff_prep_big(bfhip_engine *e, const void *taps, int n_taps, double scale, void *H,
                           int n_blocks, hipError_t *err) {
    hipStream_t keep = e->ls;
    e->ls = e->stream;
    hipLaunchKernelGGL(big_coeff_pre<T>, big_grid_half(e, n_blocks), dim3(256), 0, e->stream, (const T *)taps, n_taps,
                       (T)scale, (c2<T> *)e->d_big[0], e->L, e->d_bad);
    big_fft<T>(e, n_blocks, false, e->stream, err);
    if (*err == hipSuccess) {
        hipLaunchKernelGGL(big_untangle<T>, big_grid_half(e, n_blocks), dim3(256), 0, e->stream, (const c2<T> *)e->d_big[2],
                           (c2<T> *)H, (size_t)e->L, (const c2<T> *)e->d_tw, e->L, (T)1.0 / (T)(2 * e->L));
        *err = hipGetLastError();
    }
    e->ls = keep;
}

template <typename T>
void launch_ifft_out_big(bfhip_engine *e, const void *Zp, size_t chunk_stride, int n_chunks,
                         int first, int count, uint8_t *raw, hipError_t *err) {
    hipLaunchKernelGGL(big_out_pre<T>, big_grid_half(e, count), dim3(256), 0, e->ls, (const c2<T> *)Zp, chunk_stride,
                       n_chunks, (c2<T> *)e->d_big[0], (const c2<T> *)e->d_tw, e->L);
    big_fft<T>(e, count, true, e->ls, err);
    if (*err != hipSuccess) return;
    hipLaunchKernelGGL(big_out_post<T>, dim3(count), dim3(1024), 0, e->ls, (const c2<T> *)e->d_big[2], first, e->d_fmt[1],
                       e->d_over, (const unsigned char *)e->d_skip_quant, raw,
                       e->d_timeout ? (T *)e->d_timeout + (size_t)first * e->L : (T *)nullptr, e->L, e->safety_limit,
                       e->d_status);
    *err = hipGetLastError();          // (the dither pass behind it: do_outputs)
}

template <typename T>
void launch_levels_big(bfhip_engine *e, hipError_t *err) {
    const unsigned long long age64 = std::min<unsigned long long>(e->blocks_done + 1, (unsigned long long)e->N);
    const unsigned char *base = (const unsigned char *)e->d_jobs;
    const int V = 16 / (int)sizeof(c2<T>);
    const int threads = e->mac_threads;
    const int tiles = (e->L + threads * V - 1) / (threads * V);
    const c2<T> *twL = (const c2<T> *)e->d_tw;
    for (auto &lj : e->level_jobs) {
        if (lj.n_fill > 0) {
            const FillJob<T> *jobs = (const FillJob<T> *)(base + lj.fill_off);
            const MixSrc<T> *src = (const MixSrc<T> *)(base + e->src_off);
            hipLaunchKernelGGL(big_fill_pre<T>, big_grid_half(e, lj.n_fill), dim3(256), 0, e->ls, jobs, src,
                               (c2<T> *)e->d_big[0], twL, e->L);
            big_fft<T>(e, lj.n_fill, true, e->ls, err);
            if (*err != hipSuccess) return;
            hipLaunchKernelGGL(big_fill_slide<T>, big_grid_half(e, lj.n_fill), dim3(256), 0, e->ls, jobs,
                               (const c2<T> *)e->d_big[2], (c2<T> *)e->d_big[0], e->L);
            big_fft<T>(e, lj.n_fill, false, e->ls, err);
            if (*err != hipSuccess) return;
            hipLaunchKernelGGL(big_fill_post<T>, big_grid_half(e, lj.n_fill), dim3(256), 0, e->ls, jobs, src,
                               (const c2<T> *)e->d_big[2], twL, e->N, e->blockcounter, e->L, (const BlockState *)e->bs_arg);
        }
        if (lj.n_filt > 0) {
            hipLaunchKernelGGL(mac_filter_kernel<T>, dim3(tiles, lj.n_filt), dim3(threads), 0, e->ls,
                               (const FilterJob<T> *)(base + lj.filt_off), e->L, e->blockcounter, (int)age64,
                               (const BlockState *)e->bs_arg);
        }
        if (lj.n_fade > 0) {
            const FadeJob<T> *jobs = (const FadeJob<T> *)(base + lj.fade_off);
            hipLaunchKernelGGL(big_fade_pre<T>, big_grid_half(e, 2 * lj.n_fade), dim3(256), 0, e->ls, jobs,
                               (c2<T> *)e->d_big[0], twL, e->L);
            big_fft<T>(e, 2 * lj.n_fade, true, e->ls, err);
            if (*err != hipSuccess) return;
            hipLaunchKernelGGL(big_fade_mix<T>, dim3((unsigned)(e->L / 256), (unsigned)lj.n_fade), dim3(256), 0, e->ls,
                               (const c2<T> *)e->d_big[2], (c2<T> *)e->d_big[0], e->L);
            big_fft<T>(e, lj.n_fade, false, e->ls, err);
            if (*err != hipSuccess) return;
            hipLaunchKernelGGL(big_fade_post<T>, big_grid_half(e, lj.n_fade), dim3(256), 0, e->ls, jobs,
                               (const c2<T> *)e->d_big[2], twL, e->L);
        }
        if ((*err = hipGetLastError()) != hipSuccess) return;
    }
}

#define DISPATCH_BIG(FN, ...)                                               \
    do {                                                                    \
        if (e->rs == 4) FN<float>(__VA_ARGS__); else FN<double>(__VA_ARGS__); \
    } while (0)

// ---------------------------------------------------------------- plan

int clamp_delay(const bfhip_engine *e, int d) {          // bfrun.c:1579-1584
    if (d < 0) return 0;
    if (d > e->N - 1) return e->N - 1;
    return d;
}

int cblocks_of(const bfhip_engine *e, int coeff, int delay) {   // bfrun.c:1585-1591
    if (coeff < 0 || e->coeffs[coeff].n_blocks > e->N - delay) return e->N - delay;
    return e->coeffs[coeff].n_blocks;
}

// (Re)build the stream-ordered coefficient copy for the plan just uploaded, if the plan is a
// uniform crossbar: every entry OG coefficient terms of the same length, no split entries, every
// chunk the same number of entries, no idle blocks in the grid.  Entries whose OG sets did not
// change since the last build are left alone (scale changes rebuild the plan, not the data).
const void *const STREAM_KEY_STALE = (const void *)(uintptr_t)1;     // never a set's address
void *const PROMOTED_ELSEWHERE = (void *)(uintptr_t)1;               // promoted[] of an inactive filter: no ring here
int coeff_make_resident(bfhip_engine *e, int ci);

template <typename T>
int build_stream_layout(bfhip_engine *e, const std::vector<MacEntry<T>> &flat, const std::vector<ChunkRange> &chunks,
                        int S, double bytes_H) {
    e->hstream = StreamLayout{nullptr, 0, 0, 0};
    if (!e->stream_wanted || !e->all_dense || flat.empty()) return BFHIP_OK;
    if (e->stream_wanted == 1 && bytes_H < 64.0 * 1048576.0) return BFHIP_OK;      // cache resident anyway
    const int n_tc = e->n_tiles * S;
    if (n_tc % 8 != 0 || e->mac_threads != 256 || !e->mac_nt || e->mac_unroll != 0) return BFHIP_OK;
    const int P = flat[0].maxP, E_c = chunks[0].end - chunks[0].begin;
    for (auto &cr : chunks) if (cr.end - cr.begin != E_c) return BFHIP_OK;
    for (auto &en : flat) if (en.dense != 1 || en.p0 != 0 || en.maxP != P) return BFHIP_OK;
    if (E_c < 1 || P < 1) return BFHIP_OK;
    const unsigned int chunk = 256u * 16u;
    const unsigned long long entry_bytes = (unsigned long long)P * OG * chunk;
    const unsigned long long slice = (unsigned long long)E_c * entry_bytes;
    if (entry_bytes >= (1ull << 31) || slice >= (1ull << 32)) return BFHIP_OK;       // 32-bit offsets inside a slice
    const unsigned long long grid = (unsigned long long)n_tc * e->n_groups;
    const size_t total = (size_t)(grid * slice);
    const int geom[5] = {(int)flat.size(), E_c, P, n_tc, e->n_groups};
    const bool same_geom = e->d_stream != nullptr && memcmp(geom, e->stream_geom, sizeof(geom)) == 0;
    if (!same_geom) {
        if (total > e->stream_cap) {
            if (e->d_stream) (void)hipFree(e->d_stream);
            e->d_stream = nullptr; e->stream_cap = 0;
            if (dev_alloc(&e->d_stream, total) != hipSuccess) {
                (void)hipGetLastError();
                return BFHIP_OK;                 // no room for the second copy: the set-major path still works
            }
            e->stream_cap = total;
        }
        e->stream_keys.clear();
        memcpy(e->stream_geom, geom, sizeof(geom));
    }
    // which entries hold other sets than last time?
    std::vector<int> changed;
    std::vector<StreamWhere> where(flat.size());
    e->stream_keys.resize(flat.size(), std::array<const void *, OG>{});
    for (int g = 0; g < e->n_groups; g++) {
        for (int c = 0; c < S; c++) {
            const ChunkRange cr = chunks[(size_t)g * S + c];
            for (int q = 0; q < E_c; q++) {
                const int idx = cr.begin + q;
                where[idx] = StreamWhere{g, c, q, 0};
                std::array<const void *, OG> key;
                for (int j = 0; j < OG; j++) key[j] = flat[idx].term[j].H;
                if (key != e->stream_keys[idx]) { changed.push_back(idx); e->stream_keys[idx] = key; }
            }
        }
    }
    StreamLayout sl;
    sl.base = (const unsigned char *)e->d_stream; sl.slice = slice; sl.entry_bytes = (unsigned int)entry_bytes; sl.chunk = chunk;
    if (!changed.empty()) {
        const size_t wb = where.size() * sizeof(StreamWhere), cb = changed.size() * sizeof(int);
        if (wb + cb > e->where_cap) {
            if (e->d_where) (void)hipFree(e->d_where);
            e->d_where = nullptr;
            HIPCHK(dev_alloc((void **)&e->d_where, wb + flat.size() * sizeof(int)));
            e->where_cap = wb + flat.size() * sizeof(int);
        }
        e->d_which = (int *)((unsigned char *)e->d_where + wb);
        HIPCHK(hipMemcpy(e->d_where, where.data(), wb, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_which, changed.data(), cb, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(stream_relayout_kernel<T>, dim3((unsigned)P, (unsigned)e->n_tiles, (unsigned)changed.size()), dim3(256), 0,
                           e->stream, (const MacEntry<T> *)e->d_entries, (const StreamWhere *)e->d_where, (const int *)e->d_which,
                           0, e->L, e->n_groups, S, sl);
        HIPCHK(hipGetLastError());
        { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    }
    e->hstream = sl;
    return BFHIP_OK;
}

// one partition of a coefficient set changed in place (update_coeff_block, refresh of a watched
// set): bring the stream-ordered copy up to date
template <typename T>
int stream_refresh_block(bfhip_engine *e, const void *H, int block) {
    if (e->hstream.base == nullptr || e->plan_dirty) {
        // no copy now (the plan is about to be rebuilt, or this block runs without the stream-ordered
        // copy: a cross-fade block).  The rebuild compares entries by their set POINTERS, which an
        // in-place rewrite does not change: forget that these entries hold H, so that the next
        // build_stream_layout lays them out again from the new data.
        for (auto &k : e->stream_keys)
            for (int j = 0; j < OG; j++) if (k[j] == H) k[j] = STREAM_KEY_STALE;
        return BFHIP_OK;
    }
    std::vector<int> hit;
    for (size_t i = 0; i < e->stream_keys.size(); i++)
        for (int j = 0; j < OG; j++) if (e->stream_keys[i][j] == H) { hit.push_back((int)i); break; }
    if (hit.empty()) return BFHIP_OK;
    const int P = e->stream_geom[2];
    if (block >= P) return BFHIP_OK;
    HIPCHK(hipMemcpy(e->d_which, hit.data(), hit.size() * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(stream_relayout_kernel<T>, dim3(1u, (unsigned)e->n_tiles, (unsigned)hit.size()), dim3(256), 0, e->stream,
                       (const MacEntry<T> *)e->d_entries, (const StreamWhere *)e->d_where, (const int *)e->d_which,
                       block, e->L, e->n_groups, e->n_chunks, e->hstream);
    HIPCHK(hipGetLastError());
    return sync_all(e);
}

template <typename T>
int build_plan_t(bfhip_engine *e) {
    const int O = e->n_ch[1], I = e->n_ch[0], F = (int)e->filters.size();
    const size_t L = e->L;
    e->n_groups = (O + OG - 1) / OG;
    e->n_out_padded = e->n_groups * OG;
    std::vector<std::vector<MacEntry<T>>> per_group(e->n_groups);
    std::vector<std::map<std::pair<long, int>, std::vector<int>>> index(e->n_groups);
    double bytes_H = 0, bytes_ring = 0;
    // ring id: 0..I-1 input rings, I+f private ring of filter f, I+F+f materialised Y_f
    std::map<long, std::vector<char>> ring_used;

    std::vector<std::vector<FillJob<T>>> fills(e->n_levels);
    std::vector<std::vector<FilterJob<T>>> filts(e->n_levels);
    std::vector<std::vector<FadeJob<T>>> fades(e->n_levels);
    std::vector<MixSrc<T>> srcs;
    e->any_fading = false;

    auto Yptr = [&](int f) { return (c2<T> *)e->d_Y + (size_t)e->y_index[f] * L; };
    // ring ids order the entries of a group: filters go by the host's own numbering where it gave one,
    // so that the order does not depend on the order the filters were added in
    long NAMES = F;
    for (auto &f : e->filters) NAMES = std::max(NAMES, (long)f.name + 1);
    auto fname = [&](int fi) { return (long)(e->filters[fi].name >= 0 ? e->filters[fi].name : fi); };

    // lazily loaded coefficient sets: whatever an active filter refers to now has to be on the device
    for (int fi = 0; fi < F; fi++) {
        const Filter &f = e->filters[fi];
        if (!f.active) continue;
        for (int c : {f.coeff, (f.crossfade && f.prevcoeff != f.coeff) ? f.prevcoeff : -1}) {
            if (c < 0 || c >= (int)e->coeffs.size() || e->coeffs[c].d_H != nullptr) continue;
            const int r = coeff_make_resident(e, c);
            if (r != BFHIP_OK) return r;
        }
    }
    // which terms belong to filters another engine runs: [group][entry] bit j
    std::vector<std::vector<unsigned int>> foreign(e->n_groups);
    double bytes_H_plan = 0;           // the whole configuration's: what the launch geometry is chosen for

    for (int fi = 0; fi < F; fi++) {
        const Filter &f = e->filters[fi];
        if (f.coeff >= (int)e->coeffs.size()) return fail(BFHIP_EINVAL, "filter %d: bad coeff", fi);
        const int delay = clamp_delay(e, f.delayblocks);
        const int P = f.coeff < 0 ? 1 : cblocks_of(e, f.coeff, delay);
        // (an inactive filter is classified like an active one -- the plan's shape must not depend
        // on who runs what -- but owns no memory here)
        const bool owner_kind = f.in_ch.size() != 1 || !f.in_f.empty();
        const bool owner = f.active ? (e->owner_index[fi] >= 0 || e->promoted[fi] != nullptr)
                                    : (owner_kind || e->promoted[fi] != nullptr);
        const bool fading = f.crossfade && f.prevcoeff != f.coeff;
        const bool needY = e->is_source[fi] || fading;
        if (fading && f.active) e->any_fading = true;
        if (f.active && needY && e->y_index[fi] < 0) return fail(BFHIP_ESTATE, "filter %d: no output buffer reserved", fi);

        // where this filter's history lives and how to index it
        const c2<T> *ring;
        long ring_id;
        int rdelay;
        double rscale;
        if (owner) {
            ring = !f.active ? nullptr
                 : e->promoted[fi] ? (const c2<T> *)e->promoted[fi]
                                   : (const c2<T> *)e->d_fring + (size_t)e->owner_index[fi] * e->N * L;
            ring_id = I + fname(fi); rdelay = 0; rscale = 1.0;
            if (f.active) {
            FillJob<T> job;
            memset(&job, 0, sizeof(job));
            job.ring = (c2<T> *)ring;
            job.delay = delay;
            job.n_in = (int)f.in_ch.size();
            job.in_off = (int)srcs.size();
            for (size_t i = 0; i < f.in_ch.size(); i++) {
                MixSrc<T> m;
                m.spec = (const c2<T> *)e->d_ring + (size_t)f.in_ch[i] * e->R * L;
                m.scale = (T)(f.in_scale[i] * e->fmt[0][e->v2p[0][f.in_ch[i]]].scale);      // bfrun.c:1641 (virtscales)
                m.R = e->R;
                srcs.push_back(m);
            }
            job.n_up = (int)f.in_f.size();
            job.up_off = (int)srcs.size();
            for (size_t i = 0; i < f.in_f.size(); i++) {
                MixSrc<T> m;
                m.spec = Yptr(f.in_f[i]);
                m.scale = (T)f.in_fscale[i];
                m.R = 1;
                srcs.push_back(m);
            }
            job.evalprev = job.n_up > 0 ? (T *)e->d_evalprev + (size_t)e->sink_index[fi] * L : nullptr;
            fills[e->level[fi]].push_back(job);
            }
        } else {
            const int ch = f.in_ch[0];
            ring = (const c2<T> *)e->d_ring + (size_t)ch * e->R * L;
            ring_id = ch; rdelay = delay;
            rscale = f.in_scale[0] * e->fmt[0][e->v2p[0][ch]].scale;                         // bfrun.c:1664
        }
        if (f.active) {
            auto &u = ring_used[ring_id];
            u.resize(e->N, 0);
            for (int p = 0; p < P; p++) u[(p + rdelay) % e->N] = 1;
        }

        if (needY && f.active) {
            FilterJob<T> job;
            job.ring = ring; job.R = owner ? e->N : e->R; job.delay = rdelay; job.scale = (T)rscale;
            job.H = f.coeff < 0 ? nullptr : (const c2<T> *)e->coeffs[f.coeff].d_H;
            job.P = P; job.kind = f.coeff < 0 ? TERM_DIRAC : TERM_COEFF;
            job.Y = Yptr(fi);
            filts[e->level[fi]].push_back(job);
            if (f.coeff >= 0) bytes_H += (double)P * (double)L * (double)sizeof(c2<T>);
            if (fading) {
                // old-coefficient result for the fade (bfrun.c:1726-1769, 1803-1827)
                FilterJob<T> old = job;
                const int pc = f.prevcoeff;
                old.H = pc < 0 ? nullptr : (const c2<T> *)e->coeffs[pc].d_H;
                old.P = pc < 0 ? 1 : cblocks_of(e, pc, delay);
                old.kind = pc < 0 ? TERM_DIRAC : TERM_COEFF;
                old.Y = (c2<T> *)e->d_Yold + (size_t)e->fade_index[fi] * L;
                filts[e->level[fi]].push_back(old);
                FadeJob<T> fj;
                fj.Ynew = job.Y; fj.Yold = old.Y;
                fades[e->level[fi]].push_back(fj);
            }
        }
        if (needY && f.coeff >= 0) bytes_H_plan += (double)P * (double)L * (double)sizeof(c2<T>);

        for (size_t oi = 0; oi < f.out_ch.size(); oi++) {
            const int o = f.out_ch[oi];
            const int g = o / OG, j = o % OG;
            const double s_out = f.out_scale[oi] / e->fmt[1][e->v2p[1][o]].scale;            // bfrun.c:1850
            const std::pair<long, int> key = needY ? std::make_pair((long)(I + NAMES + fname(fi)), 0)
                                                   : std::make_pair(ring_id, rdelay);
            auto &slots = index[g][key];
            int ei = -1;
            for (int cand : slots) {
                if (per_group[g][cand].term[j].kind == TERM_NONE) { ei = cand; break; }
            }
            if (ei < 0) {
                MacEntry<T> ne;
                memset(&ne, 0, sizeof(ne));
                ne.ring = needY ? (f.active ? Yptr(fi) : nullptr) : ring;
                ne.R = needY ? 1 : (owner ? e->N : e->R);
                ne.delay = needY ? 0 : rdelay;
                ne.live = (!needY && !owner && e->d_ps_live) ? e->d_ps_live + ring_id : nullptr;
                for (int q = 0; q < OG; q++) ne.term[q].kind = TERM_NONE;
                per_group[g].push_back(ne);
                foreign[g].push_back(0u);
                ei = (int)per_group[g].size() - 1;
                slots.push_back(ei);
            }
            MacTerm<T> &tm = per_group[g][ei].term[j];
            if (!f.active) foreign[g][ei] |= 1u << j;
            if (needY) {
                tm.kind = TERM_IDENT; tm.H = nullptr; tm.P = 1; tm.scale = (T)s_out;
            } else {
                tm.kind = f.coeff < 0 ? TERM_DIRAC : TERM_COEFF;
                tm.H = f.coeff < 0 ? nullptr : (const c2<T> *)e->coeffs[f.coeff].d_H;
                tm.P = P;
                tm.scale = (T)(rscale * s_out);
                if (f.coeff >= 0) {
                    bytes_H_plan += (double)P * (double)L * (double)sizeof(c2<T>);
                    if (f.active) bytes_H += (double)P * (double)L * (double)sizeof(c2<T>);
                }
            }
            per_group[g][ei].maxP = std::max(per_group[g][ei].maxP, tm.P);
        }
    }

    // Canonical entry order inside a group: by (ring, delay), then by the order in which the same
    // key was needed again (an output fed twice from one ring).  An output's terms are then summed
    // in an order that depends on its own filters only -- not on which other outputs share its
    // group, nor on the order the host listed its filters in.
    for (int g = 0; g < e->n_groups; g++) {
        std::vector<std::pair<std::pair<std::pair<long, int>, int>, int>> order;     // ((key, n-th use), old index)
        for (auto &kv : index[g])
            for (size_t n = 0; n < kv.second.size(); n++) order.push_back({{kv.first, (int)n}, kv.second[n]});
        std::sort(order.begin(), order.end());
        std::vector<MacEntry<T>> sorted;
        std::vector<unsigned int> fsorted;
        for (auto &o : order) { sorted.push_back(per_group[g][o.second]); fsorted.push_back(foreign[g][o.second]); }
        per_group[g].swap(sorted);
        foreign[g].swap(fsorted);
    }

    // launch geometry: one fat workgroup per CU measured best on MI355X (tools/tune_mac.py): each
    // wave keeps 18 KiB of loads in flight, so 4 waves per CU already saturate HBM, and fewer
    // chunks mean fewer partial sums to write and re-read
    e->mac_threads = std::min(256, std::max(64, e->L / (int)(16 / sizeof(c2<T>))));
    const int bins_per_wg = e->mac_threads * (int)(16 / sizeof(c2<T>));
    e->n_tiles = (e->L + bins_per_wg - 1) / bins_per_wg;
    int target_wgs = 256;
    if (const char *env = getenv("BFHIP_MAC_TARGET_WGS")) target_wgs = std::max(1, atoi(env));
    if (const char *env = getenv("BFHIP_MAC_NT")) e->mac_nt = atoi(env) != 0;
    if (const char *env = getenv("BFHIP_MAC_UNROLL")) e->mac_unroll = atoi(env);
    const int s_full = std::max(1, std::min(64, (target_wgs + e->n_tiles * e->n_groups - 1) / (e->n_tiles * e->n_groups)));
    int S = s_full;
    if (!getenv("BFHIP_MAC_TARGET_WGS")) {
        // every chunk costs the output pass one more partial spectrum to read (~2 us measured);
        // a workgroup streams ~1.35 GB/s per KiB it keeps in flight per wave (2 partitions x
        // (1 ring + n coefficient loads) KiB; 25 GB/s at the crossbar's 18 KiB), the chip
        // ~6.4 TB/s, and a launch is never shorter than ~13 us.  Pick the split that minimises
        // MAC + output-pass time (matters for small crossbars; config C: S = 2).
        double n_terms = 0, n_ent = 0;
        for (auto &v : per_group)
            for (auto &en : v) {
                n_ent += 1;
                for (int q = 0; q < OG; q++) n_terms += en.term[q].kind == TERM_COEFF ? 1 : 0;
            }
        const double rate = 1.35e9 * 2.0 * (1.0 + (n_ent > 0 ? n_terms / n_ent : 1.0));
        double best = 1e30;
        for (int c = 1; c <= s_full; c++) {
            const double wgs = std::min(256.0, (double)e->n_tiles * e->n_groups * c);
            const double t_mac = std::max(std::max(13e-6, bytes_H_plan / 6.4e12), bytes_H_plan / (wgs * rate));
            const double t_out = 2e-6 * c;
            if (t_mac + t_out < best - 1e-9) { best = t_mac + t_out; S = c; }
        }
    }

    // One-to-one plans (massive_config, BASELINE configs[3]): every entry a single coefficient term,
    // every output fed by one entry.  They get mac_diag_kernel -- a workgroup per (chunk, output)
    // that walks whole spectra -- and their parallelism from splitting the PARTITIONS of every entry
    // into S parts (part c of every entry = chunk c), not from bin tiles.
    // (a workgroup holds up to 16 tiles of the spectrum in registers, a job is split over up to 8: 128 tiles)
    bool diag_plan = !e->big && (e->L + bins_per_wg - 1) / bins_per_wg <= 128;
    if (const char *env = getenv("BFHIP_MAC_DIAG")) diag_plan = diag_plan && atoi(env) != 0;
    {
        std::vector<char> fed(e->n_out_padded, 0);
        int n_ent = 0;
        for (int g = 0; g < e->n_groups && diag_plan; g++)
            for (auto &en : per_group[g]) {
                int n_terms = 0, only = -1;
                for (int q = 0; q < OG; q++) if (en.term[q].kind != TERM_NONE) { n_terms++; only = q; }
                if (n_terms != 1 || en.term[only].kind != TERM_COEFF || fed[g * OG + only]) { diag_plan = false; break; }
                fed[g * OG + only] = 1;
                n_ent++;
            }
        if (n_ent == 0) diag_plan = false;
        if (diag_plan) {
            // ~512 workgroups keep every CU busy with two; every part costs the output pass one more
            // partial spectrum per channel
            int maxlen = 1;
            for (auto &v : per_group) for (auto &en : v) maxlen = std::max(maxlen, en.maxP);
            // (workgroups come from splitting a job's SPECTRUM first -- up to 8 runs of tiles, no partial
            // sums -- and only then from parts of the partition axis: config D 256 entries x 2 tile runs)
            const int all_tiles = std::max(1, e->L / bins_per_wg);
            const int per_entry = std::min(all_tiles, 8);
            S = std::max(1, std::min(std::min(4, maxlen), (256 + n_ent * per_entry - 1) / (n_ent * per_entry)));
            if (const char *env = getenv("BFHIP_DIAG_SPLIT")) S = std::max(1, std::min(maxlen, atoi(env)));
        }
    }

    std::vector<std::vector<int>> part_of(e->n_groups);     // diag plans: which part an entry is (= its chunk)
    // few filters with many partitions (room correction): split entries along p until every
    // group has S work items
    for (int g = 0; g < e->n_groups; g++) {
        auto &v = per_group[g];
        if (v.empty()) continue;
        if (!diag_plan && (int)v.size() >= S) continue;
        if (diag_plan && S == 1) continue;
        const int parts = diag_plan ? S : (S + (int)v.size() - 1) / (int)v.size();
        std::vector<MacEntry<T>> split;
        std::vector<unsigned int> fsplit;
        std::vector<int> psplit;
        auto emit = [&](size_t i, int q) {
            const MacEntry<T> &en = v[i];
            const int len = en.maxP;
            const int np = std::max(1, std::min(parts, len));
            if (q >= np) return;
            MacEntry<T> sub = en;
            sub.p0 = (int)((long)len * q / np);
            sub.maxP = (int)((long)len * (q + 1) / np);
            if (sub.maxP > sub.p0) { split.push_back(sub); fsplit.push_back(foreign[g][i]); psplit.push_back(q); }
        };
        if (diag_plan) {
            for (int q = 0; q < parts; q++) for (size_t i = 0; i < v.size(); i++) emit(i, q);      // part-major: chunk c = part c
        } else {
            for (size_t i = 0; i < v.size(); i++) for (int q = 0; q < parts; q++) emit(i, q);
        }
        v.swap(split);
        foreign[g].swap(fsplit);
        if (diag_plan) part_of[g].swap(psplit);
    }
    size_t max_entries = 1;
    for (auto &v : per_group) max_entries = std::max(max_entries, v.size());
    S = std::max(1, std::min<int>(S, (int)max_entries));
    e->n_chunks = S;

    e->all_dense = true;
    for (auto &v : per_group) {
        for (auto &en : v) {
            bool dense = en.maxP > en.p0;
            for (int q = 0; q < OG; q++) dense = dense && en.term[q].kind == TERM_COEFF && en.term[q].P >= en.maxP;
            en.dense = dense ? 1 : 0;
            // exactly one coefficient term (one-to-one filters, massive_config style): its own
            // pipelined path, 2 + index of the term
            if (en.dense != 1) e->all_dense = false;
            int n_active = 0, only = -1;
            for (int q = 0; q < OG; q++) if (en.term[q].kind != TERM_NONE) { n_active++; only = q; }
            if (!dense && n_active == 1 && en.term[only].kind == TERM_COEFF && en.maxP > en.p0) en.dense = 2 + only;
            // several (not all OG) coefficient terms of full length: the crossbar path over a subset
            en.mask = 0;
            if (!dense && n_active >= 2 && en.maxP > en.p0) {
                bool ok = true;
                int mask = 0;
                for (int q = 0; q < OG; q++) {
                    if (en.term[q].kind == TERM_NONE) continue;
                    ok = ok && en.term[q].kind == TERM_COEFF && en.term[q].P >= en.maxP;
                    mask |= 1 << q;
                }
                if (ok) { en.dense = 16; en.mask = mask; }
            }
        }
    }

    // chunk boundaries from the WHOLE plan; then the terms of filters another engine runs are taken
    // out (an entry's path -- crossbar, single term, generic -- stays the one the whole plan gave it:
    // the single-term path rounds differently from the others), entries left empty are dropped
    std::vector<MacEntry<T>> flat;
    std::vector<ChunkRange> chunks((size_t)e->n_groups * S);
    for (int g = 0; g < e->n_groups; g++) {
        const auto &v = per_group[g];
        long total = 0;
        for (auto &en : v) total += en.maxP - en.p0;
        size_t pos = 0;
        long acc = 0;
        for (int c = 0; c < S; c++) {
            ChunkRange cr;
            cr.begin = (int)flat.size();
            const long want = (total * (c + 1) + S - 1) / S;
            // (a one-to-one plan split along p: chunk c is part c of every entry, whatever their lengths)
            const bool by_part = diag_plan && part_of[g].size() == v.size();
            while (pos < v.size() && (by_part ? part_of[g][pos] == c : (acc < want || c == S - 1))) {
                acc += v[pos].maxP - v[pos].p0;
                MacEntry<T> en = v[pos];
                const unsigned int fm = foreign[g][pos];
                pos++;
                if (fm) {
                    unsigned int present = 0;
                    for (int q = 0; q < OG; q++) if (en.term[q].kind != TERM_NONE) present |= 1u << q;
                    const unsigned int keep = present & ~fm;
                    if (keep == 0) continue;                              // nothing of ours in it
                    if (en.dense == 1) { en.dense = 16; en.mask = (int)keep; }
                    else if (en.dense == 16) en.mask &= (int)keep;
                    for (int q = 0; q < OG; q++)
                        if (fm & (1u << q)) { en.term[q].kind = TERM_NONE; en.term[q].H = nullptr; }
                }
                flat.push_back(en);
            }
            cr.end = (int)flat.size();
            chunks[(size_t)g * S + c] = cr;
        }
    }
    e->all_dense = !flat.empty();
    for (auto &en : flat) if (en.dense != 1) e->all_dense = false;
    // mac_diag_kernel's job table: the one entry behind every (chunk, output), or -1.  Checked on the
    // plan as it came out (unequal filter lengths can put two parts of one entry into one chunk):
    // anything unexpected keeps the crossbar kernel, which takes every plan.
    std::vector<int> diag_jobs;
    e->mac_diag = diag_plan && !flat.empty();
    if (e->mac_diag) {
        diag_jobs.assign((size_t)S * e->n_out_padded, -1);
        for (int g = 0; g < e->n_groups && e->mac_diag; g++)
            for (int c = 0; c < S && e->mac_diag; c++) {
                const ChunkRange cr = chunks[(size_t)g * S + c];
                for (int idx = cr.begin; idx < cr.end; idx++) {
                    const MacEntry<T> &en = flat[idx];
                    if (en.dense < 2 || en.dense >= 2 + OG) { e->mac_diag = false; break; }
                    int &slot = diag_jobs[(size_t)c * e->n_out_padded + g * OG + (en.dense - 2)];
                    if (slot != -1) { e->mac_diag = false; break; }
                    slot = idx;
                }
            }
    }
    e->n_entries = (int)flat.size();

    // job arrays for the levelled (non fast-path) filters, one device blob
    std::vector<unsigned char> blob;
    auto append = [&](const void *p, size_t n) {
        const size_t off = (blob.size() + 15) & ~(size_t)15;
        blob.resize(off + n);
        if (n) memcpy(blob.data() + off, p, n);
        return off;
    };
    e->src_off = append(srcs.data(), srcs.size() * sizeof(MixSrc<T>));
    e->level_jobs.assign(e->n_levels, bfhip_engine::LevelJobs());
    for (int lv = 0; lv < e->n_levels; lv++) {
        auto &lj = e->level_jobs[lv];
        lj.n_fill = (int)fills[lv].size();
        lj.fill_off = append(fills[lv].data(), fills[lv].size() * sizeof(FillJob<T>));
        lj.n_filt = (int)filts[lv].size();
        lj.filt_off = append(filts[lv].data(), filts[lv].size() * sizeof(FilterJob<T>));
        lj.n_fade = (int)fades[lv].size();
        lj.fade_off = append(fades[lv].data(), fades[lv].size() * sizeof(FadeJob<T>));
    }

    // upload
    const size_t eb = std::max<size_t>(flat.size(), 1) * sizeof(MacEntry<T>);
    if (eb > e->entries_cap) {
        if (e->d_entries) (void)hipFree(e->d_entries);
        HIPCHK(dev_alloc(&e->d_entries, eb));
        e->entries_cap = eb;
    }
    const size_t cb_chunks = chunks.size() * sizeof(ChunkRange);
    const size_t cb = cb_chunks + (e->mac_diag ? diag_jobs.size() * sizeof(int) : 0);
    if (cb > e->chunks_cap) {
        if (e->d_chunks) (void)hipFree(e->d_chunks);
        HIPCHK(dev_alloc((void **)&e->d_chunks, cb));
        e->chunks_cap = cb;
    }
    if (blob.size() > e->jobs_cap) {
        if (e->d_jobs) (void)hipFree(e->d_jobs);
        HIPCHK(dev_alloc(&e->d_jobs, blob.size()));
        e->jobs_cap = blob.size();
    }
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    if (!flat.empty()) HIPCHK(hipMemcpy(e->d_entries, flat.data(), flat.size() * sizeof(MacEntry<T>), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_chunks, chunks.data(), cb_chunks, hipMemcpyHostToDevice));
    e->d_diag_jobs = nullptr;
    if (e->mac_diag) {
        e->d_diag_jobs = (const int *)((const unsigned char *)e->d_chunks + cb_chunks);
        HIPCHK(hipMemcpy((void *)e->d_diag_jobs, diag_jobs.data(), diag_jobs.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    if (!blob.empty()) HIPCHK(hipMemcpy(e->d_jobs, blob.data(), blob.size(), hipMemcpyHostToDevice));
    {   // stream-ordered coefficient copy (StreamLayout) for uniform crossbar plans
        const int r = build_stream_layout<T>(e, flat, chunks, S, bytes_H);
        if (r != BFHIP_OK) return r;
    }
    const size_t zb = (size_t)S * e->n_out_padded * L * sizeof(c2<T>);
    if (zb > e->zp_bytes) {
        if (e->d_Zp) (void)hipFree(e->d_Zp);
        if (e->d_Zp2) (void)hipFree(e->d_Zp2);
        if (e->d_Zp3) (void)hipFree(e->d_Zp3);
        e->d_Zp2 = nullptr; e->d_Zp3 = nullptr;
        HIPCHK(dev_alloc(&e->d_Zp, zb));
        if (e->pipelined || e->defer_out || e->pairs) HIPCHK(dev_alloc(&e->d_Zp2, zb));
        if (e->pipe2) HIPCHK(dev_alloc(&e->d_Zp3, zb));
        e->zp_bytes = zb;
    }

    // algorithmic bytes per block, SURVEY 8(d): C*(F*P + U*P + U + O) + (I+O)*L*s_raw
    const double C = (double)L * sizeof(c2<T>);
    for (auto &kv : ring_used) for (char u : kv.second) bytes_ring += u ? C : 0;
    double raw = 0;
    for (int io = 0; io < 2; io++)
        for (int c = 0; c < e->n_phys[io]; c++) raw += (double)L * e->fmt[io][c].bytes;
    int O_here = 0;
    for (int o = 0; o < O; o++) O_here += e->out_active.empty() || e->out_active[o] ? 1 : 0;
    e->alg_bytes_mac = bytes_H + bytes_ring + C * O_here;
    e->alg_bytes_total = e->alg_bytes_mac + C * (I + e->n_owners) + raw;
    e->plan_dirty = false;
    return BFHIP_OK;
}

int build_plan(bfhip_engine *e) {
    return e->rs == 4 ? build_plan_t<float>(e) : build_plan_t<double>(e);
}

size_t raw_extent(const std::vector<bfhip_format> &v, int n, int L) {
    size_t m = 0;
    for (int i = 0; i < n; i++) {
        const bfhip_format &f = v[i];
        const size_t end = (size_t)f.byte_offset + ((size_t)(L - 1) * f.sample_spacing + 1) * f.bytes;
        m = std::max(m, end);
    }
    return m;
}

DevFormat to_dev(const bfhip_format &f) {
    DevFormat d;
    d.isfloat = f.isfloat; d.swap = f.swap; d.bytes = f.bytes; d.sbytes = f.sbytes;
    d.sample_spacing = f.sample_spacing; d.byte_offset = f.byte_offset;
    d.alt = nullptr;
    return d;
}

double overflow_max(const bfhip_format &f) {             // bfrun.c:2270-2277
    return f.isfloat ? 1.0 : (double)((uint64_t)1 << ((f.sbytes << 3) - 1)) - 1;
}

int upload_formats(bfhip_engine *e) {
    for (int io = 0; io < 2; io++) {
        // the device tables are indexed by VIRTUAL channel and hold the physical channel's format
        std::vector<DevFormat> d;
        for (int v = 0; v < e->n_ch[io]; v++) {
            DevFormat f = to_dev(e->fmt[io][e->v2p[io][v]]);
            if (io == 0 && e->n_vpp[0][e->v2p[0][v]] > 1) {
                // a private, delayed copy of the samples (bfrun.c:1509-1531), contiguous
                size_t k = 0;
                while (e->vin_list[k] != v) k++;
                f.alt = e->d_incopy + k * (size_t)e->L * 8;
                f.sample_spacing = 1;
                f.byte_offset = 0;
            }
            if (e->wide[io]) {
                // the transforms read / write the planar copy: channel p's L words, contiguous
                const int ph = e->v2p[io][v];
                f.sample_spacing = 1;
                if (io == 0) { f.alt = e->d_planar[0] + (size_t)ph * e->L * f.bytes; f.byte_offset = 0; }
                else f.byte_offset = (int)((size_t)ph * e->L * f.bytes);       // relative to d_planar[1], the `raw` of the output kernels
            }
            if (io == 0 && e->sd_slot[0][v] >= 0) {
                // sub-sample filtered: K1 reads the filtered reals (the conversion from the
                // raw format happened in the filter pass)
                f.alt = (const uint8_t *)e->d_sdin + (size_t)e->sd_slot[0][v] * e->L * e->rs;
                f.isfloat = 1; f.swap = 0; f.bytes = e->rs; f.sbytes = e->rs;
                f.sample_spacing = 1; f.byte_offset = 0;
            }
            d.push_back(f);
        }
        if (!d.empty()) HIPCHK(hipMemcpy(e->d_fmt[io], d.data(), d.size() * sizeof(DevFormat), hipMemcpyHostToDevice));
    }
    return BFHIP_OK;
}

int check_format(const bfhip_format *f) {
    if (f->isfloat) {
        if (f->bytes != 4 && f->bytes != 8) return 0;
    } else if (f->bytes < 1 || f->bytes > 4 || f->sbytes < 1 || f->sbytes > f->bytes) {
        return 0;
    }
    return f->sample_spacing >= 1 && f->byte_offset >= 0;
}

int record(bfhip_engine *e, int idx) {
    if (!e->timed_now) return BFHIP_OK;
    if (e->bs_arg != nullptr) return BFHIP_OK;           // a graph is being captured: replayed blocks are not timed
    HIPCHK(hipEventRecord(e->ev[(size_t)e->ev_used * EV_PER_BLOCK + idx], e->ls));
    if (idx & 1) e->timed_mask |= 1 << (idx >> 1);
    return BFHIP_OK;
}

// is the block in progress one of the timed ones?  Decided once per block, by whichever entry
// point touches it first (block_dev, or the phase calls of a multi-GPU host).
void timing_begin(bfhip_engine *e) {
    if (e->timed_for == e->blocks_done) return;
    e->timed_for = e->blocks_done;
    e->timed_mask = 0;
    e->timed_now = e->timing && e->ev_used < MAX_TIMED && e->blocks_done % (unsigned long long)e->timing_stride == 0;
}

int poll_coeff_changes(bfhip_engine *e);
int do_outputs(bfhip_engine *e, const void *Zp, size_t chunk_stride, int n_chunks, int first,
               int count, void *rawout_dev);

// restores "the next launch goes to the main stream" on every return path of a schedule that sends
// launches to a side stream (an error in the middle must not leave e->ls pointing there: the do_*
// helpers of the phase entry points launch on e->ls)
struct LsGuard {
    bfhip_engine *e;
    ~LsGuard() { e->ls = e->stream; }
};

// the inverse transforms of the blocks whose output is still owed (deferred output: one, ping-pong
// schedule: up to two), oldest first, each as a launch of its own
int flush_pending(bfhip_engine *e) {
    LsGuard guard{e};                  // every return path leaves the launch stream on the main stream
    const bool side = e->pipe2 && !e->pendq.empty();
    while (!e->pendq.empty()) {
        // (popped only once its launch is in the stream: a failed launch leaves the owed output
        // queued -- the caller sees the error, and the entry's event is never silently dropped)
        const bfhip_engine::Pending p = e->pendq.front();
        // the ping-pong schedule keeps every output pass on the side stream (their overflow state
        // and the dither chains are sequential), behind the MAC that produced the spectra
        e->ls = e->pipe2 ? e->s_in : e->stream;
        if (p.mac_done) HIPCHK(hipStreamWaitEvent(e->ls, p.mac_done, 0));
        const int r = do_outputs(e, p.Zp, p.chunk_stride, p.n_chunks, 0, e->n_ch[1], p.rawout);
        if (r != BFHIP_OK) return r;
        e->pendq.pop_front();
        if (p.out_done) HIPCHK(hipEventRecord(p.out_done, e->ls));
    }
    e->ls = e->stream;
    if (side) {
        // whatever the main stream does next (a phase call's MAC writes the partial-sum buffer these
        // output passes read) comes behind them
        HIPCHK(hipEventRecord(e->ev_io[0], e->s_in));
        HIPCHK(hipStreamWaitEvent(e->stream, e->ev_io[0], 0));
    }
    return BFHIP_OK;
}

// An engine is usable in the process that created it only: a fork()ed child inherits the handle but
// not a working HIP runtime, and its first device call would hang.  Checked before anything else.
int check_owner(const bfhip_engine *e) {
    if (e->owner != getpid())
        return fail(BFHIP_ESTATE, "this engine belongs to process %d: HIP state does not survive fork(); create the "
                    "engine in the process that runs the blocks", (int)e->owner);
    return BFHIP_OK;
}

int ensure_ready(bfhip_engine *e) {
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    if (!e->finalized) return fail(BFHIP_ESTATE, "engine not finalized");
    HIPCHK(hipSetDevice(e->device));
    if (e->plan_dirty && !e->pendq.empty()) {
        // the owed output belongs to the old plan's geometry and buffers
        const int r = flush_pending(e);
        if (r != BFHIP_OK) return r;
    }
    if (e->any_watched) {
        // one load of a shared counter per block; a partition another process rewrote since the
        // last block is re-uploaded now, before this block's plan / coefficient switch is applied
        const int r = poll_coeff_changes(e);
        if (r < 0) return r;
    }
    if (e->plan_dirty) {
        e->rt.valid[0] = e->rt.valid[1] = false;     // captured launches point into the old plan
        e->rt.primed = false;
        return build_plan(e);
    }
    return BFHIP_OK;
}

bool side_uses_subdelay(const bfhip_engine *e, int io) {
    if (e->sdf_length <= 0) return false;
    for (int s : e->sd_slot[io]) if (s >= 0) return true;
    return false;
}

// sub-sample delay of the channels that have a filter (bfrun.c:1503-1508 apply_subdelay, :1921-1925)
template <typename T>
int do_subdelay_t(bfhip_engine *e, int io, const void *rawin_dev) {
    std::vector<SdJob<T>> jobs;
    for (int v = 0; v < e->n_ch[io]; v++) {
        const int slot = e->sd_slot[io][v];
        if (slot < 0) continue;
        // an output another engine converts is not filtered here: nobody mixes or quantises it in this
        // engine, and an engine that owns no such output at all has no time-sample buffer either
        if (io == 1 && (!e->out_active[v] || e->d_timeout == nullptr)) continue;
        SdJob<T> j;
        memset(&j, 0, sizeof(j));
        const int sd = e->subdelay[io][v];
        j.taps = (sd > -100 && sd < 100) ? (const T *)e->d_sd_bank + (size_t)(99 + sd) * e->sd_flen : nullptr;
        j.rest = (T *)e->d_sd_rest[io] + (size_t)slot * e->sd_bs;
        if (io == 0) {
            const bfhip_format &f = e->fmt[0][e->v2p[0][v]];
            j.fmt = to_dev(f);
            if (e->n_vpp[0][e->v2p[0][v]] > 1) {              // the delayed private copy
                size_t k = 0;
                while (e->vin_list[k] != v) k++;
                j.raw = e->d_incopy + k * (size_t)e->L * 8;
                j.fmt.sample_spacing = 1; j.fmt.byte_offset = 0;
            } else {
                j.raw = (const uint8_t *)rawin_dev;
            }
            j.dst = (T *)e->d_sdin + (size_t)slot * e->L;
        } else {
            j.src = (const T *)e->d_timeout + (size_t)v * e->L;
            j.dst = (T *)e->d_timeout + (size_t)v * e->L;
        }
        jobs.push_back(j);
    }
    if (jobs.empty()) return BFHIP_OK;
    uint8_t *pin = nullptr;
    HIPCHK(e->st_sd[io].take(&pin));
    memcpy(pin, jobs.data(), jobs.size() * sizeof(SdJob<T>));
    HIPCHK(hipMemcpyAsync(e->d_sdjobs[io], pin, jobs.size() * sizeof(SdJob<T>), hipMemcpyHostToDevice, e->ls));
    HIPCHK(e->st_sd[io].queued(e->ls));
    const size_t lds = (size_t)(e->sd_bs + e->L) * sizeof(T);
    auto k = subdelay_fir_kernel<T>;
    HIPCHK(allow_lds(k, lds));
    hipLaunchKernelGGL(k, dim3((unsigned)jobs.size()), dim3(256), lds, e->ls, (const SdJob<T> *)e->d_sdjobs[io], e->L, e->sd_bs, e->sd_flen);
    HIPCHK(hipGetLastError());
    return BFHIP_OK;
}

int do_subdelay(bfhip_engine *e, int io, const void *rawin_dev) {
    if (e->sdf_length <= 0) return BFHIP_OK;
    return e->rs == 4 ? do_subdelay_t<float>(e, io, rawin_dev) : do_subdelay_t<double>(e, io, rawin_dev);
}

// N:1 inputs: gather + mute + integer delay into the private copies K1 reads (bfrun.c:1509-1531)
int do_vin(bfhip_engine *e, const void *rawin_dev) {
    if (e->vin_list.empty()) return BFHIP_OK;
    std::vector<VInJob> jobs;
    std::vector<ByteOp> ops;
    for (size_t k = 0; k < e->vin_list.size(); k++) {
        const int v = e->vin_list[k];
        const bfhip_format &f = e->fmt[0][e->v2p[0][v]];
        VInJob j;
        j.copy = e->d_incopy + k * (size_t)e->L * 8;
        j.byte_offset = f.byte_offset; j.sample_spacing = f.sample_spacing; j.bytes = f.bytes;
        j.muted = e->vmuted[0][v];
        j.ops_off = (int)ops.size();
        const int extra = (side_uses_subdelay(e, 0) && e->sd_slot[0][v] < 0) ? e->sdf_length : 0;   // bfrun.c:1512-1516
        if (!j.muted) e->vline[0][v].update(j.copy, e->vdelay[0][v] + extra, ops);   // not advanced when muted
        j.n_ops = (int)ops.size() - j.ops_off;
        jobs.push_back(j);
    }
    const size_t jb = jobs.size() * sizeof(VInJob), ob = ops.size() * sizeof(ByteOp);
    if (jb + ob > e->vjobs_slot) return fail(BFHIP_ESTATE, "virtual-channel job table overflow");
    unsigned char *dev = (unsigned char *)e->d_vjobs;
    uint8_t *pin = nullptr;
    HIPCHK(e->st_vin.take(&pin));
    memcpy(pin, jobs.data(), jb);
    if (ob) memcpy(pin + jb, ops.data(), ob);
    HIPCHK(hipMemcpyAsync(dev, pin, jb + ob, hipMemcpyHostToDevice, e->ls));
    HIPCHK(e->st_vin.queued(e->ls));
    hipLaunchKernelGGL(vchan_in_kernel<0>, dim3((unsigned)jobs.size()), dim3(256), 0, e->ls,
                       (const VInJob *)dev, (const ByteOp *)(dev + jb), (const uint8_t *)rawin_dev, e->L);
    HIPCHK(hipGetLastError());
    return BFHIP_OK;
}

// N:1 outputs: integer delay of every member, mix of the un-muted ones, one requantisation
// (bfrun.c:1938-2003); runs after K3 has left the members' time samples in d_timeout
int do_vout(bfhip_engine *e, void *rawout_dev) {
    if (e->vout_groups.empty()) return BFHIP_OK;
    std::vector<VOutJob> jobs;
    std::vector<VOutMember> mem;
    std::vector<ByteOp> ops;
    for (auto &g : e->vout_groups) {
        VOutJob j;
        j.first_member = (int)mem.size(); j.n_members = (int)g.size(); j.fmt_channel = g[0]; j.dither = 0;
        for (size_t d = 0; d < e->dither_channels.size(); d++)
            if (e->dither_channels[d] == g[0] && e->dither_late[d]) j.dither = 1;
        for (int v : g) {
            VOutMember m;
            // a 1:1 output is here for its sub-sample filter only: its delay and mute are dai.c's business
            // on the raw buffer (bfrun.c:1926-1936 converts it whatever icomm says)
            const bool shared = g.size() > 1;
            m.channel = v; m.muted = shared ? e->vmuted[1][v] : 0; m.ops_off = (int)ops.size();
            uint8_t *row = (uint8_t *)e->d_timeout + (size_t)v * e->L * e->rs;
            const int extra = (side_uses_subdelay(e, 1) && e->sd_slot[1][v] < 0 && shared) ? e->sdf_length : 0;
            e->vline[1][v].update(row, shared ? e->vdelay[1][v] + extra : 0, ops);        // always advanced (:1948)
            m.n_ops = (int)ops.size() - m.ops_off;
            mem.push_back(m);
        }
        jobs.push_back(j);
    }
    const size_t jb = jobs.size() * sizeof(VOutJob), mb = mem.size() * sizeof(VOutMember), ob = ops.size() * sizeof(ByteOp);
    if (jb + mb + ob > e->vjobs_slot) return fail(BFHIP_ESTATE, "virtual-channel job table overflow");
    unsigned char *dev = (unsigned char *)e->d_vjobs + e->vjobs_slot;      // second half: output side
    uint8_t *pin = nullptr;
    HIPCHK(e->st_vout.take(&pin));
    memcpy(pin, jobs.data(), jb);
    memcpy(pin + jb, mem.data(), mb);
    if (ob) memcpy(pin + jb + mb, ops.data(), ob);
    HIPCHK(hipMemcpyAsync(dev, pin, jb + mb + ob, hipMemcpyHostToDevice, e->ls));
    HIPCHK(e->st_vout.queued(e->ls));
    if (e->rs == 4)
        hipLaunchKernelGGL(vchan_out_kernel<float>, dim3((unsigned)jobs.size()), dim3(256), 0, e->ls,
                           (const VOutJob *)dev, (const VOutMember *)(dev + jb), (const ByteOp *)(dev + jb + mb),
                           (float *)e->d_timeout, e->d_fmt[1], e->d_over, (uint8_t *)rawout_dev, e->L, e->safety_limit, e->d_status);
    else
        hipLaunchKernelGGL(vchan_out_kernel<double>, dim3((unsigned)jobs.size()), dim3(256), 0, e->ls,
                           (const VOutJob *)dev, (const VOutMember *)(dev + jb), (const ByteOp *)(dev + jb + mb),
                           (double *)e->d_timeout, e->d_fmt[1], e->d_over, (uint8_t *)rawout_dev, e->L, e->safety_limit, e->d_status);
    HIPCHK(hipGetLastError());
    bool any_late = false;
    for (char c : e->dither_late) any_late = any_late || c;
    if (any_late) {
        hipError_t err = hipSuccess;
        if (e->rs == 4) launch_dither<float>(e, 0, e->n_ch[1], (uint8_t *)rawout_dev, &err, true);
        else launch_dither<double>(e, 0, e->n_ch[1], (uint8_t *)rawout_dev, &err, true);
        if (err != hipSuccess) return fail(BFHIP_EHIP, "dither launch: %s", hipGetErrorString(err));
        hipLaunchKernelGGL(vout_spread_overflow_kernel<0>, dim3((unsigned)jobs.size()), dim3(64), 0, e->ls,
                           (const VOutJob *)dev, (const VOutMember *)(dev + jb), e->d_over);
        HIPCHK(hipGetLastError());
    }
    return BFHIP_OK;
}

// ---- wide interleaved sides: frames <-> planar copy (transpose_words_kernel)
int launch_transpose(bfhip_engine *e, int io, const void *src, void *dst, int to_planar, int first, int count) {
    const int rows = e->L, cols = e->n_phys[io], bytes = e->fmt[io][0].bytes;
    const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    const unsigned char *skip = io == 1 ? e->d_phys_skip : nullptr;
    if (bytes == 4)
        hipLaunchKernelGGL(transpose_words_kernel<uint32_t>, grid, dim3(256), 0, e->ls, (const uint32_t *)src, (uint32_t *)dst, rows, cols, to_planar, skip, first, count);
    else if (bytes == 2)
        hipLaunchKernelGGL(transpose_words_kernel<uint16_t>, grid, dim3(256), 0, e->ls, (const uint16_t *)src, (uint16_t *)dst, rows, cols, to_planar, skip, first, count);
    else
        hipLaunchKernelGGL(transpose_words_kernel<uint64_t>, grid, dim3(256), 0, e->ls, (const uint64_t *)src, (uint64_t *)dst, rows, cols, to_planar, skip, first, count);
    HIPCHK(hipGetLastError());
    return BFHIP_OK;
}
// in front of every launch that transforms inputs
int pre_inputs(bfhip_engine *e, const void *rawin_dev) {
    return e->wide[0] ? launch_transpose(e, 0, rawin_dev, e->d_planar[0], 1, 0, e->n_phys[0]) : BFHIP_OK;
}
// what the output kernels get as `raw`, and the pass behind them (virtual channels [first, first+count))
uint8_t *k3_target(bfhip_engine *e, void *rawout_dev) { return e->wide[1] ? e->d_planar[1] : (uint8_t *)rawout_dev; }
int post_outputs(bfhip_engine *e, void *rawout_dev, int first, int count) {
    if (!e->wide[1]) return BFHIP_OK;
    int p0 = e->n_phys[1], p1 = -1;
    for (int v = first; v < first + count; v++) { p0 = std::min(p0, e->v2p[1][v]); p1 = std::max(p1, e->v2p[1][v]); }
    return p1 < p0 ? BFHIP_OK : launch_transpose(e, 1, e->d_planar[1], rawout_dev, 0, p0, p1 - p0 + 1);
}

int do_inputs(bfhip_engine *e, const void *rawin_dev, unsigned int ahead = 0u /* blocks ahead of the counter (block pairs) */) {
    { int rv = pre_inputs(e, rawin_dev); if (rv != BFHIP_OK) return rv; }
    { int rv = do_vin(e, rawin_dev); if (rv != BFHIP_OK) return rv; }
    { int rv = do_subdelay(e, 0, rawin_dev); if (rv != BFHIP_OK) return rv; }
    hipError_t err = hipSuccess;
    const int slot = (int)((e->blockcounter + ahead) % (unsigned int)e->R);
    if (e->big) { int rr = big_reserve(e, (size_t)e->n_ch[0]); if (rr != BFHIP_OK) return rr; }
    if (e->big) DISPATCH_BIG(launch_fft_in_big, e, (const uint8_t *)rawin_dev, slot, &err);
    else if (e->wave) { DISPATCH_WAVE(launch_fft_in_wave, e, (const uint8_t *)rawin_dev, slot, &err) }
    else DISPATCH(launch_fft_in, e, (const uint8_t *)rawin_dev, slot, &err);
    if (err != hipSuccess) return fail(BFHIP_EHIP, "fft_in launch: %s", hipGetErrorString(err));
    return BFHIP_OK;
}

int do_levels(bfhip_engine *e) {
    bool any = false;
    for (auto &lj : e->level_jobs) any = any || lj.n_fill || lj.n_filt || lj.n_fade;
    if (!any) return BFHIP_OK;
    hipError_t err = hipSuccess;
    { const int rr = record(e, 8); if (rr != BFHIP_OK) return rr; }
    if (e->big) {
        size_t need = 1;
        for (auto &lj : e->level_jobs) need = std::max(need, std::max((size_t)lj.n_fill, (size_t)2 * lj.n_fade));
        int rr = big_reserve(e, need);
        if (rr != BFHIP_OK) return rr;
    }
    if (e->big) DISPATCH_BIG(launch_levels_big, e, &err);
    else DISPATCH(launch_levels, e, &err);
    if (err != hipSuccess) return fail(BFHIP_EHIP, "level kernels: %s", hipGetErrorString(err));
    return record(e, 9);
}

int do_mac(bfhip_engine *e, void *Zp) {
    hipError_t err = hipSuccess;
    e->zp_is_sum = false;
    if (e->rs == 4) launch_mac<float>(e, Zp, &err); else launch_mac<double>(e, Zp, &err);
    if (err != hipSuccess) return fail(BFHIP_EHIP, "mac launch: %s", hipGetErrorString(err));
    return BFHIP_OK;
}

int do_outputs(bfhip_engine *e, const void *Zp, size_t chunk_stride, int n_chunks, int first,
               int count, void *rawout_dev) {
    hipError_t err = hipSuccess;
    if (count <= 0) return BFHIP_OK;
    if (!e->vout_groups.empty() && (first != 0 || count != e->n_ch[1]))
        return fail(BFHIP_EINVAL, "outputs that share a physical channel cannot be split over several calls");
    if (n_chunks > 2 && n_chunks == e->n_chunks && first == 0 && count == e->n_ch[1] &&
        chunk_stride == (size_t)e->n_out_padded * e->L) {
        // many partials (few long filters): the one-workgroup-per-channel output pass would walk
        // them one dependent load after the other; add them up with the whole chip first
        // (in place, same order) and hand over a single spectrum per channel
        if (e->rs == 4) launch_sum<float>(e, Zp, const_cast<void *>(Zp), &err);
        else launch_sum<double>(e, Zp, const_cast<void *>(Zp), &err);
        if (err != hipSuccess) return fail(BFHIP_EHIP, "sum_partials launch: %s", hipGetErrorString(err));
        n_chunks = 1;
    }
    void *const user_out = rawout_dev;
    rawout_dev = k3_target(e, rawout_dev);
    if (e->big) { int rr = big_reserve(e, (size_t)count); if (rr != BFHIP_OK) return rr; }
    // what follows the inverse transforms as launches of its own (dither chains, N:1 mix, sub-sample
    // delay) is timed apart: the reference's real2raw column
    const bool post = !e->dither_channels.empty() || !e->vout_groups.empty() || side_uses_subdelay(e, 1);
    if (e->big) DISPATCH_BIG(launch_ifft_out_big, e, Zp, chunk_stride, n_chunks, first, count, (uint8_t *)rawout_dev, &err);
    else if (e->wave) { DISPATCH_WAVE(launch_ifft_out_wave, e, Zp, chunk_stride, n_chunks, first, count, (uint8_t *)rawout_dev, &err) }
    else DISPATCH(launch_ifft_out, e, Zp, chunk_stride, n_chunks, first, count, (uint8_t *)rawout_dev, &err);
    if (err != hipSuccess) return fail(BFHIP_EHIP, "ifft_out launch: %s", hipGetErrorString(err));
    if (post) { const int rr = record(e, 6); if (rr != BFHIP_OK) return rr; }
    if (e->rs == 4) launch_dither<float>(e, first, count, (uint8_t *)rawout_dev, &err);
    else launch_dither<double>(e, first, count, (uint8_t *)rawout_dev, &err);
    if (err != hipSuccess) return fail(BFHIP_EHIP, "dither launch: %s", hipGetErrorString(err));
    { int rv = do_subdelay(e, 1, nullptr); if (rv != BFHIP_OK) return rv; }
    { int rv = do_vout(e, rawout_dev); if (rv != BFHIP_OK) return rv; }
    { int rv = post_outputs(e, user_out, first, count); if (rv != BFHIP_OK) return rv; }
    return post ? record(e, 7) : BFHIP_OK;
}

void advance(bfhip_engine *e) {
    if (e->timed_now) {
        if (e->timed_mask) { e->ev_mask[e->ev_used] = (unsigned char)e->timed_mask; e->ev_used++; }
        e->timed_now = false;
    }
    e->blockcounter++;                                   // bfrun.c:2034
    if (e->wrap_by != 0u && e->blockcounter >= e->wrap_at) e->blockcounter -= e->wrap_by;
    e->blocks_done++;
    for (auto &f : e->filters) f.prevcoeff = f.coeff;    // bfrun.c:1838
    if (e->any_fading) e->plan_dirty = true;             // the fade lasts exactly one block
}

// Re-upload the watched partitions whose change notice moved (bfhip_coeff_mark_dirty, written by
// convolver_runtime_coeffs2cbuf in whichever process renders new coefficients).  Returns the
// number of partitions refreshed.
int poll_coeff_changes(bfhip_engine *e) {
    const unsigned long long seq = bfhip_coeff_dirty_sequence();
    if (seq == e->watch_seq) return 0;
    e->watch_seq = seq;
    const unsigned long long lost = bfhip_dirty_lost();
    const bool all = lost != e->watch_lost;             // a notice found no slot: trust nothing
    e->watch_lost = lost;
    int n = 0;
    for (size_t ci = 0; ci < e->coeffs.size(); ci++) {
        Coeff &c = e->coeffs[ci];
        if (c.lazy && !c.watched) continue;                     // host pointers kept for the first load only
        for (size_t b = 0; b < c.watch_src.size(); b++) {
            const uint64_t gen = bfhip_dirty_generation(c.watch_src[b]);
            if (gen == c.watch_gen[b] && !all) continue;
            c.watch_gen[b] = gen;
            const int r = bfhip_engine_refresh_coeff_processed(e, (int)ci, (int)b, c.watch_src[b]);
            if (r != BFHIP_OK) return r;
            n++;
        }
    }
    return n;
}

// ---------------------------------------------------------------- real-time mode

void rt_release(bfhip_engine *e) {
    auto &rt = e->rt;
    for (int p = 0; p < 2; p++) {
        if (rt.exec[p]) (void)hipGraphExecDestroy(rt.exec[p]);
        if (rt.done[p]) (void)hipEventDestroy(rt.done[p]);
        if (rt.h_in[p]) (void)hipHostFree(rt.h_in[p]);
        if (rt.h_out[p]) (void)hipHostFree(rt.h_out[p]);
        if (rt.h_over[p]) (void)hipHostFree(rt.h_over[p]);
        if (rt.h_status[p]) (void)hipHostFree(rt.h_status[p]);
        if (rt.ev_h2d[p]) (void)hipEventDestroy(rt.ev_h2d[p]);
        if (rt.ev_cmp[p]) (void)hipEventDestroy(rt.ev_cmp[p]);
        if (rt.d_in[p]) (void)hipFree(rt.d_in[p]);
        if (rt.d_out[p]) (void)hipFree(rt.d_out[p]);
    }
    if (rt.s_h2d) (void)hipStreamDestroy(rt.s_h2d);
    if (rt.s_d2h) (void)hipStreamDestroy(rt.s_d2h);
    if (e->d_bs) (void)hipFree(e->d_bs);
    if (e->d_rt_arrive) (void)hipFree(e->d_rt_arrive);
    e->d_bs = nullptr;
    e->d_rt_arrive = nullptr;
    rt = bfhip_engine::Rt();
}

// can this plan's launch sequence be replayed unchanged block after block?
bool rt_graphable(const bfhip_engine *e) {
    if ((e->rt.flags & (BFHIP_RT_NO_GRAPH | BFHIP_RT_OVERLAP)) || e->big) return false;
    // N:1 channels and sub-sample delays upload a per-block job table; a cross-fade lasts one block
    return !e->has_vchan && !side_uses_subdelay(e, 0) && !side_uses_subdelay(e, 1) && !e->any_fading;
}

// Throughput with host buffers: upload of period t+1 and download of period t-1 on the copy
// engines (their own streams) while period t computes -- the reference's input / filter / output
// process pipeline over its two buffer halves (bfrun.c:2031, 2312-2616).  Two periods must be
// in flight for the overlap to happen (rt_submit before rt_wait of the previous one).
int rt_enqueue_overlap(bfhip_engine *e, int p) {
    auto &rt = e->rt;
    int r;
    HIPCHK(hipMemcpyAsync(rt.d_in[p], rt.h_in[p], e->raw_bytes[0], hipMemcpyHostToDevice, rt.s_h2d));
    HIPCHK(hipEventRecord(rt.ev_h2d[p], rt.s_h2d));
    e->ls = e->stream;
    HIPCHK(hipStreamWaitEvent(e->stream, rt.ev_h2d[p], 0));
    if ((r = record(e, 0)) != BFHIP_OK) return r;
    if ((r = do_inputs(e, rt.d_in[p])) != BFHIP_OK) return r;
    if ((r = record(e, 1)) != BFHIP_OK) return r;
    if ((r = do_levels(e)) != BFHIP_OK) return r;
    if ((r = record(e, 2)) != BFHIP_OK) return r;
    if ((r = do_mac(e, e->d_Zp)) != BFHIP_OK) return r;
    if ((r = record(e, 3)) != BFHIP_OK) return r;
    if ((r = record(e, 4)) != BFHIP_OK) return r;
    if ((r = do_outputs(e, e->d_Zp, (size_t)e->n_out_padded * e->L, e->n_chunks, 0, e->n_ch[1], rt.d_out[p])) != BFHIP_OK) return r;
    if ((r = record(e, 5)) != BFHIP_OK) return r;
    RtCopy none;
    none.dst = nullptr; none.src = nullptr; none.n16 = 0; none.pad = 0;
    hipLaunchKernelGGL(rt_tail_kernel<0>, dim3(1), dim3(256), 0, e->stream, none, e->d_bs, e->N,
                       (const DevOverflow *)e->d_over, rt.h_over[p], e->n_ch[1], e->d_status, rt.h_status[p], e->d_rt_arrive);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(rt.ev_cmp[p], e->stream));
    HIPCHK(hipStreamWaitEvent(rt.s_d2h, rt.ev_cmp[p], 0));
    HIPCHK(hipMemcpyAsync(rt.h_out[p], rt.d_out[p], e->raw_bytes[1], hipMemcpyDeviceToHost, rt.s_d2h));
    HIPCHK(hipEventRecord(rt.done[p], rt.s_d2h));
    return BFHIP_OK;
}

// the launch sequence of one block on the main stream: pinned in -> device -> pinned out
int rt_enqueue(bfhip_engine *e, int p) {
    auto &rt = e->rt;
    int r;
    if (rt.flags & BFHIP_RT_OVERLAP) return rt_enqueue_overlap(e, p);
    e->ls = e->stream;
    const bool nodes = (rt.flags & BFHIP_RT_COPY_ENGINE) != 0;
    RtCopy cin, cout;
    cin.dst = (uint4 *)e->d_rawin; cin.src = (const uint4 *)rt.h_in[p];
    cin.n16 = (unsigned int)((e->raw_bytes[0] + 15) / 16); cin.pad = 0;
    cout.dst = (uint4 *)rt.h_out[p]; cout.src = (const uint4 *)e->d_rawout;
    cout.n16 = nodes ? 0u : (unsigned int)((e->raw_bytes[1] + 15) / 16); cout.pad = 0;
    if (nodes) {
        HIPCHK(hipMemcpyAsync(e->d_rawin, rt.h_in[p], e->raw_bytes[0], hipMemcpyHostToDevice, e->stream));
    } else {
        hipLaunchKernelGGL(rt_copy_in_kernel<0>, dim3((cin.n16 + 255) / 256), dim3(256), 0, e->stream, cin);
        HIPCHK(hipGetLastError());
    }
    // (plain launches of a timed engine -- bfhip_engine_enable_timing; `benchmark: true` hosts ask for
    // BFHIP_RT_NO_GRAPH -- are bracketed by events like bfhip_engine_block_dev's; captured ones are not)
    if ((r = record(e, 0)) != BFHIP_OK) return r;
    if ((r = do_inputs(e, e->d_rawin)) != BFHIP_OK) return r;
    if ((r = record(e, 1)) != BFHIP_OK) return r;
    if ((r = do_levels(e)) != BFHIP_OK) return r;
    if ((r = record(e, 2)) != BFHIP_OK) return r;
    if ((r = do_mac(e, e->d_Zp)) != BFHIP_OK) return r;
    if ((r = record(e, 3)) != BFHIP_OK) return r;
    if ((r = record(e, 4)) != BFHIP_OK) return r;
    if ((r = do_outputs(e, e->d_Zp, (size_t)e->n_out_padded * e->L, e->n_chunks, 0, e->n_ch[1], e->d_rawout)) != BFHIP_OK) return r;
    if ((r = record(e, 5)) != BFHIP_OK) return r;
    if (nodes) HIPCHK(hipMemcpyAsync(rt.h_out[p], e->d_rawout, e->raw_bytes[1], hipMemcpyDeviceToHost, e->stream));
    hipLaunchKernelGGL(rt_tail_kernel<0>, dim3(nodes ? 1u : (cout.n16 + 255) / 256), dim3(256), 0, e->stream, cout,
                       e->d_bs, e->N, (const DevOverflow *)e->d_over, rt.h_over[p], e->n_ch[1], e->d_status,
                       rt.h_status[p], e->d_rt_arrive);
    HIPCHK(hipGetLastError());
    return BFHIP_OK;
}

int rt_capture(bfhip_engine *e, int p) {
    auto &rt = e->rt;
    if (rt.exec[p]) { (void)hipGraphExecDestroy(rt.exec[p]); rt.exec[p] = nullptr; }
    HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeRelaxed));
    e->bs_arg = e->d_bs;
    const int r = rt_enqueue(e, p);
    e->bs_arg = nullptr;
    hipGraph_t g = nullptr;
    const hipError_t ce = hipStreamEndCapture(e->stream, &g);
    if (r != BFHIP_OK) { if (g) (void)hipGraphDestroy(g); return r; }
    if (ce != hipSuccess) return fail(BFHIP_EHIP, "graph capture: %s", hipGetErrorString(ce));
    const hipError_t ie = hipGraphInstantiate(&rt.exec[p], g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ie != hipSuccess) { rt.exec[p] = nullptr; return fail(BFHIP_EHIP, "graph instantiate: %s", hipGetErrorString(ie)); }
    rt.valid[p] = true;
    rt.n_capture++;
    return BFHIP_OK;
}

}  // namespace

namespace {


// Kaiser window exactly as the reference applies it to its sub-sample filters
// (firwindow.c:12-160 via delay.c:56-76; note the window is applied twice when offset != 0)
double sd_bessel_i0(double x) {
    double n = 1.0, a = 1.0, sum = 1.0;
    const double h = x / 2.0;
    do { a *= h; a /= n; sum += a * a; n += 1.0; } while (a != 0.0 && std::isfinite(sum));
    return sum;
}

template <typename T>
void sd_make_filter(std::vector<T> &f, int half, double offset, double beta) {
    const int len = 2 * half + 1;
    f.assign(len, (T)0);
    if (offset == 0.0) { f[half] = (T)1; return; }            // delay.c:476-483: a unit pulse
    for (int n = 0; n < len; n++) {
        const double x = M_PI * ((double)(n - half) - offset);
        f[n] = (T)(x == 0.0 ? 1.0 : sin(x) / x);
    }
    const double inv = 1.0 / sd_bessel_i0(beta);
    auto kaiser = [&](double x) {
        if (x < -1.0) x = -1.0;
        if (x > 1.0) x = 1.0;
        return sd_bessel_i0(beta * sqrt(1.0 - x * x)) * inv;
    };
    int max = half + (int)floor(offset);
    offset -= floor(offset);
    if (fabs(offset) < 1e-20) offset = 0.0;
    double step = 1.0 / ((double)max + offset);
    if (offset == 0.0) max -= 1;
    int n = 0;
    for (; n <= max; n++) { const double y = kaiser(-1.0 + (double)n * step); f[n] = (T)(f[n] * y); f[n] = (T)(f[n] * y); }
    if (offset == 0.0) max += 1;
    step = 1.0 / ((double)(len - max - 1) - offset);
    for (; n < len; n++) { const double y = kaiser(((double)(n - max) - offset) * step); f[n] = (T)(f[n] * y); f[n] = (T)(f[n] * y); }
}

int subdelay_setup(bfhip_engine *e) {
    if (e->sdf_length <= 0) return BFHIP_OK;
    int n_slots[2] = {0, 0};
    for (int io = 0; io < 2; io++)
        for (int v = 0; v < e->n_ch[io]; v++)
            e->sd_slot[io][v] = e->subdelay[io][v] != -100 ? n_slots[io]++ : -1;      // bfrun.c:1133-1142
    if (n_slots[0] + n_slots[1] == 0) { e->sdf_length = 0; return BFHIP_OK; }
    // bank: index 99 + subdelay, subdelay in (-100, 100) hundredths of a sample (BF_SAMPLE_SLOTS)
    std::vector<unsigned char> bank((size_t)199 * e->sd_flen * e->rs);
    for (int sd = -99; sd <= 99; sd++) {
        if (e->rs == 4) {
            std::vector<float> f;
            sd_make_filter(f, e->sdf_length, (double)sd / 100, 9.0);
            memcpy(bank.data() + (size_t)(99 + sd) * e->sd_flen * 4, f.data(), f.size() * 4);
        } else {
            std::vector<double> f;
            sd_make_filter(f, e->sdf_length, (double)sd / 100, 9.0);
            memcpy(bank.data() + (size_t)(99 + sd) * e->sd_flen * 8, f.data(), f.size() * 8);
        }
    }
    HIPCHK(dev_alloc(&e->d_sd_bank, bank.size()));
    HIPCHK(hipMemcpy(e->d_sd_bank, bank.data(), bank.size(), hipMemcpyHostToDevice));
    for (int io = 0; io < 2; io++) {
        if (n_slots[io] == 0) continue;
        HIPCHK(dev_alloc(&e->d_sd_rest[io], (size_t)n_slots[io] * e->sd_bs * e->rs));
        HIPCHK(hipMemset(e->d_sd_rest[io], 0, (size_t)n_slots[io] * e->sd_bs * e->rs));
        HIPCHK(dev_alloc(&e->d_sdjobs[io], (size_t)n_slots[io] * 128));
        HIPCHK(e->st_sd[io].init((size_t)n_slots[io] * 128));
    }
    if (n_slots[0] > 0) {
        HIPCHK(dev_alloc(&e->d_sdin, (size_t)n_slots[0] * e->L * e->rs));
        HIPCHK(hipMemset(e->d_sdin, 0, (size_t)n_slots[0] * e->L * e->rs));
    }
    return BFHIP_OK;
}

}  // namespace

// ==================================================================== C ABI

extern "C" {

const char *bfhip_last_error(void) { return g_err.c_str(); }
const char *bfhip_version(void) { return "bfhip 0.1 (gfx950)"; }

int bfhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

bfhip_engine *bfhip_engine_create(int device, int length, int n_blocks, int realsize,
                                  int n_in, int n_out) {
    if (realsize != 4 && realsize != 8) { fail(BFHIP_EINVAL, "Invalid real size %d.", realsize); return nullptr; }
    const int lg = ilog2(length);
    if (lg < 2 || lg > 20) {
        fail(BFHIP_EINVAL, "Invalid length %d (power of two in 4..1048576 required).", length);
        return nullptr;
    }
    if (n_blocks < 1 || n_in < 1 || n_out < 1) { fail(BFHIP_EINVAL, "bad n_blocks/n_in/n_out"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fail(BFHIP_ENODEV, "no HIP device available (there is no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { fail(BFHIP_ENODEV, "device %d out of range (%d devices)", device, ndev); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fail(BFHIP_ENODEV, "hipSetDevice(%d) failed", device); return nullptr; }

    bfhip_engine *e = new bfhip_engine();
    e->device = device; e->L = length; e->N = n_blocks; e->rs = realsize; e->log2L = lg;
    e->owner = getpid();
    e->big = lg > BIG_LOG2M;
    e->big_R = e->big ? length / BIG_M : 1;
    e->n_ch[0] = n_in; e->n_ch[1] = n_out;
    for (int io = 0; io < 2; io++) {
        e->fmt[io].resize(e->n_ch[io]);
        for (int c = 0; c < e->n_ch[io]; c++) {
            bfhip_format &f = e->fmt[io][c];
            f.isfloat = 1; f.swap = 0; f.bytes = f.sbytes = realsize; f.scale = 1.0;
            f.sample_spacing = 1; f.byte_offset = c * length * realsize;
        }
    }
    for (int io = 0; io < 2; io++) {
        e->n_phys[io] = e->n_ch[io];
        e->v2p[io].resize(e->n_ch[io]);
        for (int c = 0; c < e->n_ch[io]; c++) e->v2p[io][c] = c;
        e->n_vpp[io].assign(e->n_ch[io], 1);
        e->vdelay[io].assign(e->n_ch[io], 0);
        e->vmaxdelay[io].assign(e->n_ch[io], 0);
        e->vmuted[io].assign(e->n_ch[io], 0);
        e->subdelay[io].assign(e->n_ch[io], -100);       // BF_UNDEFINED_SUBDELAY
        e->sd_slot[io].assign(e->n_ch[io], -1);
    }
    bool ok = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) == hipSuccess;
    e->own_stream = ok;
    // twiddles exp(-2 pi i m / (2L)), computed in double, + their thread-ordered copy (fft_lds.h)
    int log2l = 0;
    while ((1 << log2l) < length) log2l++;
    const std::vector<unsigned char> tw = make_twiddle_table(log2l, realsize,
        realsize == 4 ? fft_threads<float>(log2l) : fft_threads<double>(log2l));
    ok = ok && dev_alloc(&e->d_tw, tw.size()) == hipSuccess;
    ok = ok && hipMemcpy(e->d_tw, tw.data(), tw.size(), hipMemcpyHostToDevice) == hipSuccess;
    if (e->big) {
        const std::vector<unsigned char> tw13 = make_twiddle_table(BIG_LOG2M, realsize,
            realsize == 4 ? fft_threads<float>(BIG_LOG2M) : fft_threads<double>(BIG_LOG2M));
        ok = ok && dev_alloc(&e->d_tw13, tw13.size()) == hipSuccess;
        ok = ok && hipMemcpy(e->d_tw13, tw13.data(), tw13.size(), hipMemcpyHostToDevice) == hipSuccess;
    }
    // default from L = 4096 up: below that a workgroup of L/16 threads is one or two waves and the
    // plain LDS transform with twice the threads has the shorter critical path (tools/fft_probe:
    // L = 2048 5.6 vs 6.3 us, L = 1024 4.5 vs 5.4 us); BFHIP_FFT_WAVE=1 forces it on, =0 off
    if (const char *env = getenv("BFHIP_COEFF_ARENA")) e->coeff_arena = atoi(env) != 0;
    if (const char *env = getenv("BFHIP_COEFF_STREAM")) e->stream_wanted = atoi(env);
    e->wave = wave_fft_ok(lg, realsize) && lg >= 12;
    if (const char *env = getenv("BFHIP_FFT_WAVE")) e->wave = wave_fft_ok(lg, realsize) && atoi(env) != 0;
    if (e->wave) {
        const std::vector<unsigned char> tww = make_wave_twiddle_table(lg, realsize);
        ok = ok && dev_alloc(&e->d_tww, tww.size()) == hipSuccess;
        ok = ok && hipMemcpy(e->d_tww, tww.data(), tww.size(), hipMemcpyHostToDevice) == hipSuccess;
    }
    ok = ok && dev_alloc((void **)&e->d_bad, sizeof(int)) == hipSuccess;
    ok = ok && hipMemset(e->d_bad, 0, sizeof(int)) == hipSuccess;
    if (!ok) {
        const hipError_t le = hipGetLastError();
        fail(BFHIP_EHIP, "device set-up failed: %s", le == hipSuccess ? "out of device memory" : hipGetErrorString(le));
        bfhip_engine_destroy(e);
        return nullptr;
    }
    return e;
}

void bfhip_engine_destroy(bfhip_engine *e) {
    if (e == nullptr) return;
    if (e->owner != getpid()) { delete e; return; }     // a forked child: the device objects are the parent's
    (void)hipSetDevice(e->device);
    e->pendq.clear();
    (void)sync_all(e);
    if (e->coeff_arena) { for (auto &sl : e->slabs) if (sl.base) (void)hipFree(sl.base); }
    else { for (auto &c : e->coeffs) if (c.d_H) (void)hipFree(c.d_H); }
    for (void *p : e->promoted) if (p && p != PROMOTED_ELSEWHERE) (void)hipFree(p);
    for (int io = 0; io < 2; io++) for (auto &dl : e->vline[io]) if (dl.arena) (void)hipFree(dl.arena);
    if (e->d_incopy) (void)hipFree(e->d_incopy);
    { void *sp[] = {e->d_sd_bank, e->d_sd_rest[0], e->d_sd_rest[1], e->d_sdin, e->d_sdjobs[0], e->d_sdjobs[1]};
      for (void *q : sp) if (q) (void)hipFree(q); }
    rt_release(e);
    if (e->d_vjobs) (void)hipFree(e->d_vjobs);
    e->st_vin.release(); e->st_vout.release(); e->st_sd[0].release(); e->st_sd[1].release();
    if (e->d_Zp2) (void)hipFree(e->d_Zp2);
    for (int i = 0; i < 2; i++) {
        if (e->ev_in[i]) (void)hipEventDestroy(e->ev_in[i]);
        if (e->ev_mac[i]) (void)hipEventDestroy(e->ev_mac[i]);
        if (e->ev_out[i]) (void)hipEventDestroy(e->ev_out[i]);
        if (e->ev_io[i]) (void)hipEventDestroy(e->ev_io[i]);
    }
    for (int i = 0; i < 3; i++) if (e->ev_mac3[i]) (void)hipEventDestroy(e->ev_mac3[i]);
    if (e->d_Zp3) (void)hipFree(e->d_Zp3);
    if (e->s_in) (void)hipStreamDestroy(e->s_in);
    if (e->s_out) (void)hipStreamDestroy(e->s_out);
    if (e->d_status_own) e->d_status = e->d_status_own;
    void *ptrs[] = {e->d_tw, e->d_prev, e->d_ring, e->d_fmt[0], e->d_fmt[1], e->d_over, e->d_status,
                    e->d_bad, e->d_Zp, e->d_entries, e->d_chunks, e->d_rawin, e->d_rawout, e->d_taps,
                    e->d_fring, e->d_Y, e->d_Yold, e->d_evalprev, e->d_jobs,
                    e->d_dither_ch, e->d_dither_state, e->d_dither_table, e->d_randmap, e->d_skip_quant, e->d_timeout,
                    e->d_big[0], e->d_big[1], e->d_big[2], e->d_tw13, e->d_tww,
                    e->d_ps_flags, e->d_ps_live, e->d_ps_scale, e->d_ps_acc, e->d_stream, e->d_where,
                    e->d_planar[0], e->d_planar[1], e->d_phys_skip};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto ev : e->ev) (void)hipEventDestroy(ev);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int bfhip_engine_set_format(bfhip_engine *e, int io, int ch, const bfhip_format *bf) {
    if (!e || !bf || io < 0 || io > 1 || ch < 0 || ch >= e->n_phys[io]) return fail(BFHIP_EINVAL, "set_format: bad argument");
    if (!check_format(bf)) return fail(BFHIP_EINVAL, "Sample byte size %d is not supported.", bf->bytes);
    if (e->finalized) return fail(BFHIP_ESTATE, "set_format after finalize");
    e->fmt[io][ch] = *bf;
    return BFHIP_OK;
}

int bfhip_engine_set_overlap(bfhip_engine *e, int mode) {
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    if (e->finalized) return fail(BFHIP_ESTATE, "set_overlap after finalize");
    e->overlap_mode = mode;
    return BFHIP_OK;
}

int bfhip_engine_set_safety_limit(bfhip_engine *e, double limit) {
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    e->safety_limit = limit;
    return BFHIP_OK;
}

int bfhip_engine_set_powersave(bfhip_engine *e, double analog_powersave) {
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    if (e->finalized) return fail(BFHIP_ESTATE, "set_powersave after finalize");
    if (analog_powersave < 0.0) return fail(BFHIP_EINVAL, "set_powersave: negative noise floor");
    e->powersave = analog_powersave;
    return BFHIP_OK;
}

int bfhip_engine_map_channels(bfhip_engine *e, int io, int n_phys, const int virt2phys[]) {
    if (!e || io < 0 || io > 1 || !virt2phys || n_phys < 1 || n_phys > e->n_ch[io]) return fail(BFHIP_EINVAL, "map_channels: bad argument");
    if (e->finalized) return fail(BFHIP_ESTATE, "map_channels after finalize");
    std::vector<int> cnt(n_phys, 0);
    for (int v = 0; v < e->n_ch[io]; v++) {
        if (virt2phys[v] < 0 || virt2phys[v] >= n_phys) return fail(BFHIP_EINVAL, "map_channels: physical channel %d", virt2phys[v]);
        cnt[virt2phys[v]]++;
    }
    for (int c = 0; c < n_phys; c++) if (cnt[c] == 0) return fail(BFHIP_EINVAL, "map_channels: physical channel %d unused", c);
    e->n_phys[io] = n_phys;
    e->v2p[io].assign(virt2phys, virt2phys + e->n_ch[io]);
    e->n_vpp[io] = cnt;
    return BFHIP_OK;
}

int bfhip_engine_set_delay(bfhip_engine *e, int io, int ch, int delay) {
    if (!e || io < 0 || io > 1 || ch < 0 || ch >= e->n_ch[io] || delay < 0) return fail(BFHIP_EINVAL, "set_delay: bad argument");
    e->vdelay[io][ch] = delay;
    return BFHIP_OK;
}

int bfhip_engine_set_maxdelay(bfhip_engine *e, int io, int ch, int maxdelay) {
    if (!e || io < 0 || io > 1 || ch < 0 || ch >= e->n_ch[io]) return fail(BFHIP_EINVAL, "set_maxdelay: bad argument");
    if (e->finalized) return fail(BFHIP_ESTATE, "set_maxdelay after finalize");
    e->vmaxdelay[io][ch] = maxdelay;
    return BFHIP_OK;
}

int bfhip_engine_set_mute(bfhip_engine *e, int io, int ch, int muted) {
    if (!e || io < 0 || io > 1 || ch < 0 || ch >= e->n_ch[io]) return fail(BFHIP_EINVAL, "set_mute: bad argument");
    e->vmuted[io][ch] = muted != 0;
    return BFHIP_OK;
}

int bfhip_engine_enable_subdelay(bfhip_engine *e, int sdf_length, double kaiser_beta) {
    (void)kaiser_beta;       // parsed by the reference (sdf_beta) but its filters are built with 9 (delay.c:73)
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    if (e->finalized) return fail(BFHIP_ESTATE, "enable_subdelay after finalize");
    if (sdf_length < 1) return fail(BFHIP_EINVAL, "Invalid half filter length %d.", sdf_length);
    if (2 * sdf_length + 1 > e->L) return fail(BFHIP_EINVAL, "The filter_length must be at least 2 x sdf_length + 1.");
    int bs = 1;
    while (bs < 2 * sdf_length + 1) bs <<= 1;
    if (e->L % bs != 0) return fail(BFHIP_EINVAL, "Incompatible fragment/filter sizes (%d/%d).", e->L, 2 * sdf_length + 1);
    e->sdf_length = sdf_length; e->sd_flen = 2 * sdf_length + 1; e->sd_bs = bs;
    return BFHIP_OK;
}

int bfhip_engine_set_subdelay(bfhip_engine *e, int io, int ch, int subdelay) {
    if (!e || io < 0 || io > 1 || ch < 0 || ch >= e->n_ch[io]) return fail(BFHIP_EINVAL, "set_subdelay: bad argument");
    e->subdelay[io][ch] = subdelay;
    return BFHIP_OK;
}

int bfhip_engine_enable_dither(bfhip_engine *e, const int out_channels[], int n,
                               int sample_rate, int max_size) {
    if (!e || !out_channels || n < 1 || sample_rate < 1) return fail(BFHIP_EINVAL, "enable_dither: bad argument");
    if (e->finalized) return fail(BFHIP_ESTATE, "enable_dither after finalize");
    std::vector<int> chs(out_channels, out_channels + n);
    for (int i = 0; i < n; i++) {
        if (chs[i] < 0 || chs[i] >= e->n_phys[1]) return fail(BFHIP_EINVAL, "enable_dither: output channel %d", chs[i]);
        if (i > 0 && chs[i] <= chs[i - 1]) return fail(BFHIP_EINVAL, "enable_dither: channels must be ascending");
        if (e->fmt[1][chs[i]].isfloat) return fail(BFHIP_EINVAL, "cannot dither floating point format (output %d)", chs[i]);
    }
    // dither_init (dither.c:75-139): table spacing, Tausworthe bytes
    int spacing = 10 * sample_rate;
    const int minspacing = sample_rate > e->L ? sample_rate : e->L;
    if (spacing < minspacing) spacing = minspacing;
    if (max_size > 0 && n * spacing > max_size) spacing = max_size / n;
    if (spacing < minspacing)
        return fail(BFHIP_EINVAL, "Maximum dither table size %d bytes is too small, must at least be %d bytes.",
                    max_size, n * sample_rate * minspacing);
    e->dither_spacing = spacing;
    e->dither_table.resize((size_t)n * spacing + 1);
    uint32_t st[3];
    auto lcg = [](uint32_t v) { return (uint32_t)(69069u * v); };
    st[0] = lcg(1); st[1] = lcg(st[0]); st[2] = lcg(st[1]);
    auto taus = [&]() {
        st[0] = ((st[0] & 4294967294u) << 12) ^ (((st[0] << 13) ^ st[0]) >> 19);
        st[1] = ((st[1] & 4294967288u) << 4) ^ (((st[1] << 2) ^ st[1]) >> 25);
        st[2] = ((st[2] & 4294967280u) << 17) ^ (((st[2] << 3) ^ st[2]) >> 11);
        return st[0] ^ st[1] ^ st[2];
    };
    for (int i = 0; i < 6; i++) taus();
    for (auto &b : e->dither_table) b = (int8_t)(taus() & 0xFF);
    e->dither_channels = chs;
    return BFHIP_OK;
}

static int dither_upload(bfhip_engine *e) {
    const int n = (int)e->dither_channels.size();
    if (n == 0) return BFHIP_OK;
    HIPCHK(dev_alloc((void **)&e->d_dither_ch, n * sizeof(int)));
    HIPCHK(hipMemcpy(e->d_dither_ch, e->dither_channels.data(), n * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(dev_alloc((void **)&e->d_dither_table, e->dither_table.size()));
    HIPCHK(hipMemcpy(e->d_dither_table, e->dither_table.data(), e->dither_table.size(), hipMemcpyHostToDevice));
    // randmap[d] = 0.5 + (d + 1)/255 for d in -255..253, [-256] = -0.5, [254] = 1.5
    // (dither.c:115-131).  The reference indexes it with int8 - int8, which can be +255: one
    // element past its table (undefined there); defined here by continuing the formula.
    std::vector<unsigned char> map(512 * e->rs);
    for (int d = -256; d < 256; d++) {
        if (e->rs == 4) {
            float v = d == -256 ? -0.5f : (d == 254 ? 1.5f : (float)(0.5 + 1.0 / 255.0 + 1.0 / 255.0 * (float)d));
            ((float *)map.data())[d + 256] = v;
        } else {
            double v = d == -256 ? -0.5 : (d == 254 ? 1.5 : 0.5 + 1.0 / 255.0 + 1.0 / 255.0 * (double)d);
            ((double *)map.data())[d + 256] = v;
        }
    }
    HIPCHK(dev_alloc(&e->d_randmap, map.size()));
    HIPCHK(hipMemcpy(e->d_randmap, map.data(), map.size(), hipMemcpyHostToDevice));
    // per-slot state: ptr = n*spacing + 1, error feedback zero (dither.c:133-137)
    const size_t ssz = e->rs == 4 ? sizeof(DitherState<float>) : sizeof(DitherState<double>);
    std::vector<unsigned char> stv(ssz * n, 0);
    for (int i = 0; i < n; i++) *(int *)(stv.data() + ssz * i) = e->dither_rank[i] * e->dither_spacing + 1;
    HIPCHK(dev_alloc(&e->d_dither_state, stv.size()));
    HIPCHK(hipMemcpy(e->d_dither_state, stv.data(), stv.size(), hipMemcpyHostToDevice));
    return BFHIP_OK;             // d_skip_quant / d_timeout: allocated with the channel set-up in finalize
}

static int add_coeff_common(bfhip_engine *e, const void *taps, bool on_device, int n_taps,
                            double scale, int n_blocks) {
    if (!e || (!taps && n_taps > 0) || n_taps < 0) return fail(BFHIP_EINVAL, "add_coeff: bad argument");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    const int L = e->L;
    if (n_blocks <= 0) n_blocks = (n_taps + L - 1) / L;
    if (n_blocks < 1) n_blocks = 1;
    if (n_blocks > e->N) return fail(BFHIP_EINVAL, "coefficient set needs %d blocks, engine has %d", n_blocks, e->N);
    if (n_taps > n_blocks * L) n_taps = n_blocks * L;
    const void *src = taps;
    // Device taps: whatever produced them ran on a stream this engine knows nothing about (its own
    // stream is non-blocking: not even the legacy default stream orders with it).  Loading is not
    // the hot path: wait for the device before reading them, and for this engine's kernel before
    // returning -- the buffer is the caller's again when the call returns.
    if (on_device) HIPCHK(hipDeviceSynchronize());
    if (!on_device && n_taps > 0) {
        const size_t bytes = (size_t)n_taps * e->rs;
        if (bytes > e->taps_cap) {
            if (e->d_taps) { { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; } (void)hipFree(e->d_taps); e->d_taps = nullptr; }
            HIPCHK(dev_alloc(&e->d_taps, bytes));
            e->taps_cap = bytes;
        }
        { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
        HIPCHK(hipMemcpy(e->d_taps, taps, bytes, hipMemcpyHostToDevice));
        src = e->d_taps;
    }
    Coeff c;
    c.n_blocks = n_blocks;
    const size_t h_bytes = (size_t)n_blocks * L * e->csize();
    if ((c.d_H = coeff_alloc(e, h_bytes)) == nullptr)
        return fail(BFHIP_ENOMEM, "out of device memory for coefficient set");
    hipError_t err = hipSuccess;
    if (e->big) {
        { int rr = big_reserve(e, (size_t)n_blocks); if (rr != BFHIP_OK) { coeff_release(e, c.d_H, h_bytes); return rr; } }
        DISPATCH_BIG(launch_coeff_prep_big, e, src, n_taps, scale, c.d_H, n_blocks, &err);
    } else DISPATCH(launch_coeff_prep, e, src, n_taps, scale, c.d_H, n_blocks, &err);
    if (err != hipSuccess) { coeff_release(e, c.d_H, h_bytes); return fail(BFHIP_EHIP, "coeff_prep launch: %s", hipGetErrorString(err)); }
    if (on_device) { int _r = sync_all(e); if (_r != BFHIP_OK) { coeff_release(e, c.d_H, h_bytes); return _r; } }
    if (!on_device) {
        // host-taps path is synchronous, like convolver_coeffs2cbuf: report NaN/Inf now
        int bad = 0;
        { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
        HIPCHK(hipMemcpy(&bad, e->d_bad, sizeof(int), hipMemcpyDeviceToHost));
        if (bad) {
            HIPCHK(hipMemset(e->d_bad, 0, sizeof(int)));
            coeff_release(e, c.d_H, h_bytes);
            return fail(BFHIP_EINVAL, "NaN or Inf value among coefficients.");
        }
    }
    e->coeffs.push_back(c);
    return (int)e->coeffs.size() - 1;
}

int bfhip_engine_reserve_coeffs(bfhip_engine *e, double total_bytes) {
    if (!e || total_bytes < 0) return fail(BFHIP_EINVAL, "reserve_coeffs: bad argument");
    if (!e->coeff_arena || total_bytes == 0) return BFHIP_OK;
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    // every set is rounded up to 64 KiB: leave room for that
    const size_t want = ((size_t)(total_bytes * 1.05) + ((size_t)1 << 20) + 65535) & ~(size_t)65535;
    if (!e->slabs.empty() && e->slabs.back().cap - e->slabs.back().used >= want) return BFHIP_OK;
    bfhip_engine::Slab sl;
    if (dev_alloc(&sl.base, want) != hipSuccess) {
        (void)hipGetLastError();
        return fail(BFHIP_ENOMEM, "out of device memory reserving %.0f bytes of coefficient memory", total_bytes);
    }
    sl.cap = want; sl.used = 0;
    e->slabs.push_back(sl);
    return BFHIP_OK;
}

int bfhip_engine_add_coeff(bfhip_engine *e, const void *taps, int n_taps, double scale, int n_blocks) {
    return add_coeff_common(e, taps, false, n_taps, scale, n_blocks);
}

int bfhip_engine_add_coeff_dev(bfhip_engine *e, const void *taps_dev, int n_taps, double scale, int n_blocks) {
    return add_coeff_common(e, taps_dev, true, n_taps, scale, n_blocks);
}

// one cbuf (2L reals, the reference's "4 re / 4 im" layout) -> packed partition `block` of set c
static int upload_processed_block(bfhip_engine *e, Coeff &c, int block, const void *cbuf) {
    const size_t n = (size_t)2 * e->L;
    for (size_t i = 0; i < n; i++) {           // convolver_verify_cbuf (fftw_convolver.c:598-622)
        const double v = e->rs == 4 ? (double)((const float *)cbuf)[i] : ((const double *)cbuf)[i];
        if (!std::isfinite(v)) return fail(BFHIP_EINVAL, "NaN or Inf value among coefficients.");
    }
    const size_t bytes = n * e->rs;
    if (bytes > e->taps_cap) {
        if (e->d_taps) { { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; } (void)hipFree(e->d_taps); e->d_taps = nullptr; }
        HIPCHK(dev_alloc(&e->d_taps, bytes));
        e->taps_cap = bytes;
    }
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }      // d_taps is reused; nothing may still read H
    HIPCHK(hipMemcpy(e->d_taps, cbuf, bytes, hipMemcpyHostToDevice));
    void *H = (unsigned char *)c.d_H + (size_t)block * e->L * e->csize();
    const dim3 grid((e->L + 255) / 256, 1);
    if (e->rs == 4)
        hipLaunchKernelGGL(reorder_kernel<float>, grid, dim3(256), 0, e->stream, (const float *)e->d_taps, (c2<float> *)H, e->L, 1, (float *)nullptr);
    else
        hipLaunchKernelGGL(reorder_kernel<double>, grid, dim3(256), 0, e->stream, (const double *)e->d_taps, (c2<double> *)H, e->L, 1, (double *)nullptr);
    HIPCHK(hipGetLastError());
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    return e->rs == 4 ? stream_refresh_block<float>(e, c.d_H, block) : stream_refresh_block<double>(e, c.d_H, block);
}

namespace {
// a lazily registered set goes to the device (the first time an active filter refers to it)
int coeff_make_resident(bfhip_engine *e, int ci) {
    Coeff &c = e->coeffs[ci];
    if (c.d_H != nullptr) return BFHIP_OK;
    if ((int)c.watch_src.size() != c.n_blocks) return fail(BFHIP_ESTATE, "coefficient set %d has no data", ci);
    const size_t h_bytes = (size_t)c.n_blocks * e->L * e->csize();
    if ((c.d_H = coeff_alloc(e, h_bytes)) == nullptr)
        return fail(BFHIP_ENOMEM, "out of device memory for coefficient set");
    for (int b = 0; b < c.n_blocks; b++) {
        // generation first, data second: a notice that arrives in between is seen by the next poll
        const uint64_t gen = c.watched ? bfhip_dirty_generation(c.watch_src[b]) : 0;
        const int r = upload_processed_block(e, c, b, c.watch_src[b]);
        if (r != BFHIP_OK) { coeff_release(e, c.d_H, h_bytes); c.d_H = nullptr; return r; }
        if (c.watched) c.watch_gen[b] = gen;
    }
    return BFHIP_OK;
}
}  // namespace

int bfhip_engine_add_coeff_processed_blocks(bfhip_engine *e, void *const cbufs[], int n_blocks, int flags) {
    if (!e || !cbufs || n_blocks < 1) return fail(BFHIP_EINVAL, "add_coeff_processed_blocks: bad argument");
    if (n_blocks > e->N) return fail(BFHIP_EINVAL, "coefficient set has %d blocks, engine has %d", n_blocks, e->N);
    for (int b = 0; b < n_blocks; b++) if (!cbufs[b]) return fail(BFHIP_EINVAL, "add_coeff_processed_blocks: block %d is NULL", b);
    if (flags & ~(BFHIP_COEFF_WATCH | BFHIP_COEFF_LAZY)) return fail(BFHIP_EINVAL, "add_coeff_processed_blocks: unknown flag");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    const bool watch = (flags & BFHIP_COEFF_WATCH) != 0, lazy = (flags & BFHIP_COEFF_LAZY) != 0;
    Coeff c;
    c.n_blocks = n_blocks;
    c.lazy = lazy; c.watched = watch;
    if (lazy) {
        // nothing goes to the device yet: the host blocks (which the host keeps for its whole life,
        // bfconf->coeffs_data) are loaded when a filter this engine runs first refers to the set
        for (int b = 0; b < n_blocks; b++) { c.watch_src.push_back(cbufs[b]); c.watch_gen.push_back(0); }
        if (watch) e->any_watched = true;
        e->coeffs.push_back(c);
        if (e->finalized) e->plan_dirty = true;
        return (int)e->coeffs.size() - 1;
    }
    HIPCHK(hipSetDevice(e->device));
    const size_t h_bytes = (size_t)n_blocks * e->L * e->csize();
    if ((c.d_H = coeff_alloc(e, h_bytes)) == nullptr)
        return fail(BFHIP_ENOMEM, "out of device memory for coefficient set");
    for (int b = 0; b < n_blocks; b++) {
        // generation first, data second: a notice that arrives in between is seen by the next poll
        const uint64_t gen = watch ? bfhip_dirty_generation(cbufs[b]) : 0;
        const int r = upload_processed_block(e, c, b, cbufs[b]);
        if (r != BFHIP_OK) { coeff_release(e, c.d_H, h_bytes); return r; }
        if (watch) { c.watch_src.push_back(cbufs[b]); c.watch_gen.push_back(gen); }
    }
    if (watch) {
        e->any_watched = true;
        // (watch_seq stays where it is: a notice older than this call costs one extra scan)
    }
    e->coeffs.push_back(c);
    return (int)e->coeffs.size() - 1;
}

int bfhip_engine_coeff_is_resident(const bfhip_engine *e, int coeff) {
    if (!e || coeff < 0 || coeff >= (int)e->coeffs.size()) return 0;
    return e->coeffs[coeff].d_H != nullptr ? 1 : 0;
}

int bfhip_engine_refresh_coeff_processed(bfhip_engine *e, int coeff, int block, const void *cbuf) {
    if (!e || coeff < 0 || coeff >= (int)e->coeffs.size() || block < 0 || block >= e->coeffs[coeff].n_blocks)
        return fail(BFHIP_EINVAL, "refresh_coeff_processed: bad argument");
    Coeff &c = e->coeffs[coeff];
    if (cbuf == nullptr) {
        if (c.watch_src.empty()) return fail(BFHIP_EINVAL, "refresh_coeff_processed: no host buffer known for this set");
        cbuf = c.watch_src[block];
    }
    if (c.d_H == nullptr) {
        // registered lazily and not loaded yet: its host block is read when the set is first needed
        if (cbuf != c.watch_src[block]) return fail(BFHIP_ESTATE, "refresh_coeff_processed: set %d is not on the device yet", coeff);
        return BFHIP_OK;
    }
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    return upload_processed_block(e, c, block, cbuf);
}

int bfhip_engine_poll_coeff_changes(bfhip_engine *e) {
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    return poll_coeff_changes(e);
}

int bfhip_engine_add_coeff_processed(bfhip_engine *e, const void *cbufs, int n_blocks) {
    if (!e || !cbufs || n_blocks < 1) return fail(BFHIP_EINVAL, "add_coeff_processed: bad argument");
    if (n_blocks > e->N) return fail(BFHIP_EINVAL, "coefficient set has %d blocks, engine has %d", n_blocks, e->N);
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    const size_t n = (size_t)n_blocks * 2 * e->L;
    // convolver_verify_cbuf (fftw_convolver.c:598-622) as load_coeff applies it (bfconf.c:1958)
    for (size_t i = 0; i < n; i++) {
        const double v = e->rs == 4 ? (double)((const float *)cbufs)[i] : ((const double *)cbufs)[i];
        if (!std::isfinite(v)) return fail(BFHIP_EINVAL, "NaN or Inf value among coefficients.");
    }
    const size_t bytes = n * e->rs;
    if (bytes > e->taps_cap) {
        if (e->d_taps) { { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; } (void)hipFree(e->d_taps); e->d_taps = nullptr; }
        HIPCHK(dev_alloc(&e->d_taps, bytes));
        e->taps_cap = bytes;
    }
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    HIPCHK(hipMemcpy(e->d_taps, cbufs, bytes, hipMemcpyHostToDevice));
    Coeff c;
    c.n_blocks = n_blocks;
    if ((c.d_H = coeff_alloc(e, (size_t)n_blocks * e->L * e->csize())) == nullptr)
        return fail(BFHIP_ENOMEM, "out of device memory for coefficient set");
    const dim3 grid((e->L + 255) / 256, n_blocks);
    if (e->rs == 4)
        hipLaunchKernelGGL(reorder_kernel<float>, grid, dim3(256), 0, e->stream, (const float *)e->d_taps, (c2<float> *)c.d_H, e->L, 1, (float *)nullptr);
    else
        hipLaunchKernelGGL(reorder_kernel<double>, grid, dim3(256), 0, e->stream, (const double *)e->d_taps, (c2<double> *)c.d_H, e->L, 1, (double *)nullptr);
    HIPCHK(hipGetLastError());
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    e->coeffs.push_back(c);
    return (int)e->coeffs.size() - 1;
}

int bfhip_engine_read_coeff_processed(bfhip_engine *e, int coeff, void *cbufs) {
    if (!e || !cbufs || coeff < 0 || coeff >= (int)e->coeffs.size()) return fail(BFHIP_EINVAL, "read_coeff_processed: bad argument");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    { const int rr = coeff_make_resident(e, coeff); if (rr != BFHIP_OK) return rr; }
    const int nb = e->coeffs[coeff].n_blocks;
    const size_t bytes = (size_t)nb * 2 * e->L * e->rs;
    if (bytes > e->taps_cap) {
        if (e->d_taps) { { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; } (void)hipFree(e->d_taps); e->d_taps = nullptr; }
        HIPCHK(dev_alloc(&e->d_taps, bytes));
        e->taps_cap = bytes;
    }
    const dim3 grid((e->L + 255) / 256, nb);
    if (e->rs == 4)
        hipLaunchKernelGGL(reorder_kernel<float>, grid, dim3(256), 0, e->stream, (const float *)nullptr, (c2<float> *)e->coeffs[coeff].d_H, e->L, 0, (float *)e->d_taps);
    else
        hipLaunchKernelGGL(reorder_kernel<double>, grid, dim3(256), 0, e->stream, (const double *)nullptr, (c2<double> *)e->coeffs[coeff].d_H, e->L, 0, (double *)e->d_taps);
    HIPCHK(hipGetLastError());
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    HIPCHK(hipMemcpy(cbufs, e->d_taps, bytes, hipMemcpyDeviceToHost));
    return nb;
}

int bfhip_engine_update_coeff_block(bfhip_engine *e, int coeff, int block, const void *taps) {
    if (!e || coeff < 0 || coeff >= (int)e->coeffs.size() || block < 0 ||
        block >= e->coeffs[coeff].n_blocks || !taps)
        return fail(BFHIP_EINVAL, "update_coeff_block: bad argument");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    { const int rr = coeff_make_resident(e, coeff); if (rr != BFHIP_OK) return rr; }
    const size_t bytes = (size_t)e->L * e->rs;
    if (bytes > e->taps_cap) {
        if (e->d_taps) { { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; } (void)hipFree(e->d_taps); e->d_taps = nullptr; }
        HIPCHK(dev_alloc(&e->d_taps, bytes));
        e->taps_cap = bytes;
    }
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    HIPCHK(hipMemcpy(e->d_taps, taps, bytes, hipMemcpyHostToDevice));
    void *H = (unsigned char *)e->coeffs[coeff].d_H + (size_t)block * e->L * e->csize();
    hipError_t err = hipSuccess;
    if (e->big) {
        { int rr = big_reserve(e, 1); if (rr != BFHIP_OK) return rr; }
        DISPATCH_BIG(launch_coeff_prep_big, e, e->d_taps, e->L, 1.0, H, 1, &err);
    } else DISPATCH(launch_coeff_prep, e, e->d_taps, e->L, 1.0, H, 1, &err);
    if (err != hipSuccess) return fail(BFHIP_EHIP, "coeff_prep launch: %s", hipGetErrorString(err));
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    return e->rs == 4 ? stream_refresh_block<float>(e, e->coeffs[coeff].d_H, block)
                      : stream_refresh_block<double>(e, e->coeffs[coeff].d_H, block);
}

int bfhip_engine_add_filter(bfhip_engine *e,
                            int n_in_ch, const int in_ch[], const double in_scale[],
                            int n_in_f, const int in_f[], const double in_fscale[],
                            int n_out_ch, const int out_ch[], const double out_scale[],
                            int coeff, int delayblocks, int crossfade) {
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    if (e->finalized) return fail(BFHIP_ESTATE, "add_filter after finalize");
    if (n_in_ch < 0 || n_in_f < 0 || n_out_ch < 0) return fail(BFHIP_EINVAL, "add_filter: negative count");
    if ((n_in_ch > 0 && (!in_ch || !in_scale)) || (n_in_f > 0 && (!in_f || !in_fscale)) ||
        (n_out_ch > 0 && (!out_ch || !out_scale)))
        return fail(BFHIP_EINVAL, "add_filter: null array");
    if (coeff < -1) return fail(BFHIP_EINVAL, "add_filter: coeff %d (-1 = none)", coeff);
    Filter f;
    for (int i = 0; i < n_in_ch; i++) {
        if (in_ch[i] < 0 || in_ch[i] >= e->n_ch[0]) return fail(BFHIP_EINVAL, "add_filter: input channel %d", in_ch[i]);
        f.in_ch.push_back(in_ch[i]); f.in_scale.push_back(in_scale[i]);
    }
    for (int i = 0; i < n_in_f; i++) {
        if (in_f[i] < 0 || in_f[i] >= (int)e->filters.size()) return fail(BFHIP_EINVAL, "add_filter: from_filter %d not defined yet", in_f[i]);
        f.in_f.push_back(in_f[i]); f.in_fscale.push_back(in_fscale[i]);
    }
    for (int i = 0; i < n_out_ch; i++) {
        if (out_ch[i] < 0 || out_ch[i] >= e->n_ch[1]) return fail(BFHIP_EINVAL, "add_filter: output channel %d", out_ch[i]);
        f.out_ch.push_back(out_ch[i]); f.out_scale.push_back(out_scale[i]);
    }
    if (coeff >= (int)e->coeffs.size()) return fail(BFHIP_EINVAL, "add_filter: coeff %d not loaded", coeff);
    f.coeff = coeff; f.delayblocks = delayblocks; f.crossfade = crossfade; f.prevcoeff = coeff;
    e->filters.push_back(f);
    return (int)e->filters.size() - 1;
}

int bfhip_engine_set_filter_active(bfhip_engine *e, int filter, int active) {
    if (!e || filter < 0 || filter >= (int)e->filters.size()) return fail(BFHIP_EINVAL, "set_filter_active: bad argument");
    if (e->finalized) return fail(BFHIP_ESTATE, "set_filter_active after finalize");
    e->filters[filter].active = active != 0;
    return BFHIP_OK;
}

int bfhip_engine_set_filter_name(bfhip_engine *e, int filter, int name) {
    if (!e || filter < 0 || filter >= (int)e->filters.size() || name < 0) return fail(BFHIP_EINVAL, "set_filter_name: bad argument");
    if (e->finalized) return fail(BFHIP_ESTATE, "set_filter_name after finalize");
    for (size_t i = 0; i < e->filters.size(); i++)
        if ((int)i != filter && e->filters[i].name == name) return fail(BFHIP_EINVAL, "set_filter_name: %d is taken", name);
    e->filters[filter].name = name;
    return BFHIP_OK;
}

int bfhip_engine_set_output_active(bfhip_engine *e, int ch, int active) {
    if (!e || ch < 0 || ch >= e->n_ch[1]) return fail(BFHIP_EINVAL, "set_output_active: bad argument");
    if (e->finalized) return fail(BFHIP_ESTATE, "set_output_active after finalize");
    e->out_active_set.resize(e->n_ch[1], -1);
    e->out_active_set[ch] = active ? 1 : 0;
    return BFHIP_OK;
}

int bfhip_engine_output_is_active(const bfhip_engine *e, int ch) {
    if (!e || ch < 0 || ch >= e->n_ch[1]) return 0;
    return e->out_active.empty() ? 1 : (e->out_active[ch] ? 1 : 0);
}

static int finalize_impl(bfhip_engine *e);

int bfhip_engine_finalize(bfhip_engine *e) {
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    if (e->finalized) return BFHIP_OK;
    // a finalize that failed half way (out of device memory, a bad coefficient set) leaves partial
    // state behind that only bfhip_engine_destroy cleans up: it is not retried
    if (e->finalize_failed) return fail(BFHIP_ESTATE, "finalize failed before: destroy this engine");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    const int r = finalize_impl(e);
    if (r != BFHIP_OK) { e->finalize_failed = true; e->finalized = false; }
    return r;
}

// Which outputs does this engine convert and write?  (bfconf.c:2893-2931: every output is mixed
// inside one filter process, connected filters stay in one process -- the same rules, checked.)
static int resolve_shard(bfhip_engine *e) {
    const int O = e->n_ch[1], F = (int)e->filters.size();
    e->out_active_set.resize(O, -1);
    std::vector<char> fed(O, 0), fed_here(O, 0), fed_else(O, 0);
    for (int fi = 0; fi < F; fi++) {
        const Filter &f = e->filters[fi];
        for (int o : f.out_ch) { fed[o] = 1; (f.active ? fed_here : fed_else)[o] = 1; }
        for (int g : f.in_f)
            if (e->filters[g].active != f.active)
                return fail(BFHIP_EINVAL, "filters %d and %d are connected: they must be run by the same engine", g, fi);
    }
    e->out_active.assign(O, 1);
    e->sharded = false;
    for (int o = 0; o < O; o++) {
        if (fed_here[o] && fed_else[o]) return fail(BFHIP_EINVAL, "output %d is mixed from filters of two engines", o);
        const bool derived = fed[o] ? fed_here[o] != 0 : true;
        const bool act = e->out_active_set[o] < 0 ? derived : e->out_active_set[o] != 0;
        if (!act && fed_here[o]) return fail(BFHIP_EINVAL, "output %d is fed by this engine's filters but marked inactive", o);
        if (act && fed_else[o]) return fail(BFHIP_EINVAL, "output %d is fed by another engine's filters but marked active", o);
        e->out_active[o] = act ? 1 : 0;
        if (!act) e->sharded = true;
    }
    // virtual outputs that share a physical channel are mixed in the time domain by ONE engine
    for (int o = 0; o < O; o++)
        for (int q = o + 1; q < O; q++)
            if (e->v2p[1][o] == e->v2p[1][q] && e->out_active[o] != e->out_active[q])
                return fail(BFHIP_EINVAL, "outputs %d and %d share a physical channel: one engine must own both", o, q);
    return BFHIP_OK;
}

// the raw output samples a sharded engine owns, as byte runs per frame (neighbouring channels of an
// interleaved frame merge into one run)
static void build_owned_runs(bfhip_engine *e) {
    e->owned_runs.clear();
    if (!e->sharded) return;
    std::vector<bfhip_engine::OwnedRun> runs;
    std::vector<char> seen(e->n_phys[1], 0);
    for (int o = 0; o < e->n_ch[1]; o++) {
        const int p = e->v2p[1][o];
        if (!e->out_active[o] || seen[p]) continue;
        seen[p] = 1;
        const bfhip_format &f = e->fmt[1][p];
        runs.push_back({(size_t)f.byte_offset, (size_t)f.bytes, (size_t)f.sample_spacing * f.bytes});
    }
    std::sort(runs.begin(), runs.end(), [](const bfhip_engine::OwnedRun &a, const bfhip_engine::OwnedRun &b) { return a.offset < b.offset; });
    for (auto &r : runs) {
        if (!e->owned_runs.empty()) {
            auto &b = e->owned_runs.back();
            if (b.stride == r.stride && b.offset + b.len == r.offset && b.len + r.len <= b.stride) { b.len += r.len; continue; }
        }
        e->owned_runs.push_back(r);
    }
}

// dst <- src for the samples this engine owns; everything else in dst belongs to other engines
static void copy_owned(const bfhip_engine *e, void *dst, const void *src) {
    if (!e->sharded) { memcpy(dst, src, e->raw_bytes[1]); return; }
    for (auto &r : e->owned_runs) {
        unsigned char *d = (unsigned char *)dst + r.offset;
        const unsigned char *s = (const unsigned char *)src + r.offset;
        if (r.len == r.stride) { memcpy(d, s, r.len * (size_t)e->L); continue; }
        for (int n = 0; n < e->L; n++, d += r.stride, s += r.stride) memcpy(d, s, r.len);
    }
}

static void copy_owned_overflow(const bfhip_engine *e, bfhip_overflow dst[], const void *src) {
    const DevOverflow *s = (const DevOverflow *)src;
    for (int o = 0; o < e->n_ch[1]; o++)
        if (!e->sharded || e->out_active[o]) memcpy(&dst[o], &s[o], sizeof(DevOverflow));
}

static int finalize_impl(bfhip_engine *e) {
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    { const int rsh = resolve_shard(e); if (rsh != BFHIP_OK) return rsh; }
    HIPCHK(hipSetDevice(e->device));
    const size_t L = e->L;
    const size_t prev_b = (size_t)e->n_ch[0] * L * e->rs;
    // Run the FFT kernels of neighbouring blocks beside the MAC?  Pays when the MAC is short
    // (small crossbars: config B 84 -> 58 us per block) and costs when it streams for a
    // millisecond (config C 1.40 -> 1.77 ms): decide from the coefficient bytes per block.
    {
        double bytes = 0;
        for (auto &f : e->filters) {
            if (f.coeff < 0) continue;
            const int d = clamp_delay(e, f.delayblocks);
            bytes += (double)cblocks_of(e, f.coeff, d) * (double)e->L * (double)e->csize() * (double)std::max<size_t>(1, f.out_ch.size());
        }
        // ... and only while the transforms leave most CUs to the MAC: with hundreds of channels the
        // FFT workgroups fill the chip themselves (config D, 256 + 256: 0.178 ms piped, 0.168 plain)
        e->pipelined = bytes / 6.4e12 < 100e-6 && e->n_ch[0] + e->n_ch[1] <= 128;
        // (Beside a long MAC the transforms only get in the way: a MAC workgroup takes 320 of a SIMD's
        // 512 registers per lane, no transform workgroup fits on the same CU, and on a side stream
        // they either wait for the MAC's tail or push its workgroups together on fewer CUs --
        // config C 1.40 -> 1.74 ms.  A 256-thread variant that did fit next to round 1's MAC hid
        // them but slowed the MAC by as much)
        if (e->overlap_mode >= 0) e->pipelined = e->overlap_mode != 0;
        // (the environment only moves the AUTOMATIC choice: an explicit bfhip_engine_set_overlap wins --
        // the non-uniform convolver depends on its segment engines running strictly in order)
        if (const char *env = e->overlap_mode < 0 ? getenv("BFHIP_OVERLAP") : nullptr) e->pipelined = atoi(env) != 0;
        for (int io = 0; io < 2; io++) for (int c : e->n_vpp[io]) if (c > 1) e->pipelined = false;   // one job table per side
        if (e->big) e->pipelined = false;          // one FFT scratch
        // one stream and a MAC that fills the chip for a long time: fuse the output pass of a block
        // with the input pass of the next one (deferred output).  Needs the plain 1:1 raw path.
        bool plain = e->wave && !e->big && e->sdf_length <= 0 && e->dither_channels.empty();
        for (int io = 0; io < 2; io++) for (int c : e->n_vpp[io]) if (c > 1) plain = false;
        e->defer_out = !e->pipelined && plain && e->overlap_mode != 0;
        // (an explicit bfhip_engine_set_overlap(e, 0) -- "strictly in order" -- beats the environment:
        // the non-uniform convolver relies on it for its segment engines)
        if (const char *env = getenv("BFHIP_DEFER")) e->defer_out = atoi(env) != 0 && !e->pipelined && plain && e->overlap_mode != 0;
        // small MACs with the wave FFT: the two-stream ping-pong instead of three streams
        e->pipe2 = e->pipelined && plain;
        if (const char *env = getenv("BFHIP_PIPE2")) e->pipe2 = e->pipe2 && atoi(env) != 0;
    }
    if (e->pairs) { e->pipelined = false; e->pipe2 = false; }       // block pairs run on the one main stream
    // (a pair needs the spare ring slot too: block t + 1 is transformed before the MAC reads block t - N + 1)
    e->R = (e->pipelined || e->pairs) ? e->N + 1 : e->N;
    {
        // the shared rings are R deep, the private ones N: the counter wraps by a multiple of both
        const unsigned long long period = e->R == e->N ? (unsigned long long)e->N : (unsigned long long)e->N * e->R;
        if (period <= (1ull << 30)) {
            e->wrap_by = (unsigned int)((0x7fffffffull / period) * period);
            e->wrap_at = e->wrap_by + 2u * (unsigned int)e->R;           // t - p - delay stays >= 0
        }
        if (const char *env = getenv("BFHIP_TEST_WRAP_PERIODS")) {       // tests: wrap after a few ring periods
            e->wrap_by = (unsigned int)(std::max(1, atoi(env)) * period);
            // (... and BFHIP_TEST_WRAP_SKEW: by a few blocks more, i.e. NOT a multiple of the depths --
            // what a wrap at 2^32 is for such depths; the tests want to see that go wrong)
            if (const char *skew = getenv("BFHIP_TEST_WRAP_SKEW")) e->wrap_by += (unsigned int)atoi(skew);
            e->wrap_at = e->wrap_by + 2u * (unsigned int)e->R;
        }
    }
    // the MAC addresses a ring / a coefficient set with 32-bit byte offsets from its base
    if ((double)(e->N + 1) * (double)L * (double)e->csize() >= 4294967296.0)
        return fail(BFHIP_EINVAL, "%d partitions of %d taps: a coefficient set would exceed 4 GiB", e->N, e->L);
    const size_t ring_b = (size_t)e->n_ch[0] * e->R * L * e->csize();
    if (dev_alloc(&e->d_prev, prev_b) != hipSuccess || dev_alloc(&e->d_ring, ring_b) != hipSuccess)
        return fail(BFHIP_ENOMEM, "out of device memory for the spectrum rings");
    HIPCHK(hipMemset(e->d_prev, 0, prev_b));       // bfrun.c:1388: everything starts zeroed
    HIPCHK(hipMemset(e->d_ring, 0, ring_b));
    if (e->powersave > 0.0) {
        if (e->big) {
            HIPCHK(dev_alloc((void **)&e->d_ps_acc, 2 * e->n_ch[0] * sizeof(unsigned long long)));
            HIPCHK(hipMemset(e->d_ps_acc, 0, 2 * e->n_ch[0] * sizeof(unsigned long long)));
        }
        {
            std::vector<int> ones((size_t)e->n_ch[0] * e->R, 1);     // the zeroed rings are silence
            std::vector<double> sc(e->n_ch[0]);
            for (int c = 0; c < e->n_ch[0]; c++) sc[c] = e->fmt[0][e->v2p[0][c]].scale;
            HIPCHK(dev_alloc((void **)&e->d_ps_flags, ones.size() * sizeof(int)));
            HIPCHK(hipMemcpy(e->d_ps_flags, ones.data(), ones.size() * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(dev_alloc((void **)&e->d_ps_live, e->n_ch[0] * sizeof(int)));
            HIPCHK(hipMemset(e->d_ps_live, 0, e->n_ch[0] * sizeof(int)));
            HIPCHK(dev_alloc((void **)&e->d_ps_scale, sc.size() * sizeof(double)));
            HIPCHK(hipMemcpy(e->d_ps_scale, sc.data(), sc.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    for (int io = 0; io < 2; io++) HIPCHK(dev_alloc((void **)&e->d_fmt[io], e->n_ch[io] * sizeof(DevFormat)));
    { int rs_ = subdelay_setup(e); if (rs_ != BFHIP_OK) return rs_; }
    // channels that share a physical channel: private copies, delay lines, job tables
    {
        e->vin_list.clear(); e->vout_groups.clear();
        for (int v = 0; v < e->n_ch[0]; v++) if (e->n_vpp[0][e->v2p[0][v]] > 1) e->vin_list.push_back(v);
        // the members of a shared physical output are mixed in ascending virtual order, the order
        // in which the reference walks phys2virt (bfrun.c:2322-2323, 1938-2003); any mapping is
        // legal (bench4_config: `mapping: 0,1,0,1,0,1`)
        std::map<int, std::vector<int>> shared;
        for (int v = 0; v < e->n_ch[1]; v++) {
            if (e->n_vpp[1][e->v2p[1][v]] <= 1) {
                // a 1:1 output with a sub-sample filter is requantised after the filter: a group of one
                if (e->sd_slot[1][v] >= 0) e->vout_groups.push_back({v});
                continue;
            }
            shared[e->v2p[1][v]].push_back(v);
        }
        for (auto &kv : shared) e->vout_groups.push_back(kv.second);
        // (groups of outputs another engine converts: not ours to mix -- resolve_shard made sure a
        // group is owned as a whole)
        e->vout_groups.erase(std::remove_if(e->vout_groups.begin(), e->vout_groups.end(),
                                            [&](const std::vector<int> &g) { return !e->out_active[g[0]]; }),
                             e->vout_groups.end());
        e->has_vchan = !e->vin_list.empty() || !e->vout_groups.empty() || e->sdf_length > 0;
        // enable_dither named PHYSICAL outputs (bfconf->dither_state[physch], bfrun.c:1933): from
        // here on the list holds the virtual channel behind each of them
        {
            std::vector<std::pair<int, int>> byvirt;            // (virtual channel, rank in the physical list)
            for (size_t i = 0; i < e->dither_channels.size(); i++) {
                const int c = e->dither_channels[i];
                if (c < 0 || c >= e->n_phys[1]) return fail(BFHIP_EINVAL, "dither: output %d does not exist", c);
                int v = 0;
                while (e->v2p[1][v] != c) v++;          // the first (lowest) virtual channel of the physical one
                byvirt.push_back({v, (int)i});
            }
            std::sort(byvirt.begin(), byvirt.end());
            e->dither_rank.clear(); e->dither_late.clear();
            e->dither_channels.clear();
            for (size_t j = 0; j < byvirt.size(); j++) {
                const int v = byvirt[j].first;
                // an output another engine converts keeps its rank (= where its walk through the
                // random table starts, dither.c) but has no slot here
                if (!e->out_active[v]) continue;
                e->dither_channels.push_back(v);
                e->dither_rank.push_back(byvirt[j].second);
                // shared outputs and outputs with a sub-sample filter are requantised by the N:1
                // pass (bfrun.c:1938-2003): their dither runs on what that pass leaves behind
                e->dither_late.push_back(e->n_vpp[1][e->v2p[1][v]] > 1 || e->sd_slot[1][v] >= 0);
            }
        }
        if (e->has_vchan) {
            size_t n_ops_max = 0;
            e->vline[0].assign(e->n_ch[0], DelayLine());
            e->vline[1].assign(e->n_ch[1], DelayLine());
            // The reference adds the sub-sample filter's integer part to the delay AND to maxdelay
            // (bfrun.c:1152-1162, 1185-1197) -- also to maxdelay -1 ("cannot be changed"), which turns it
            // into the limit sdf_length - 1, below the delay it allocates for: delay.c:357-374 then sizes
            // the buffer for the limit and fills it with the delay (a heap overrun in the reference, found
            // by tests/test_gpu_refloop.py).  Here a negative maxdelay stays negative (fixed delay), and a
            // delay above a positive limit starts at the limit -- the clamp delay.c:358-360 means to make.
            auto limits = [](int delay, int maxd, int extra, int *init_eff, int *max_eff) {
                *max_eff = maxd < 0 ? maxd : maxd + extra;
                *init_eff = delay + extra;
                if (*max_eff > 0 && *init_eff > *max_eff) *init_eff = *max_eff;
            };
            for (int v : e->vin_list) {
                const int extra = (side_uses_subdelay(e, 0) && e->sd_slot[0][v] < 0) ? e->sdf_length : 0;    // bfrun.c:1152-1162
                int d0, m0;
                limits(e->vdelay[0][v], e->vmaxdelay[0][v], extra, &d0, &m0);
                int rr = e->vline[0][v].init(e->L, d0, m0, e->fmt[0][e->v2p[0][v]].bytes);
                if (rr != BFHIP_OK) return fail(rr, "delay buffer allocation failed");
                n_ops_max += e->vline[0][v].n_full_cap + 10;
            }
            for (auto &g : e->vout_groups) for (int v : g) {
                const int extra = (side_uses_subdelay(e, 1) && e->sd_slot[1][v] < 0 && g.size() > 1) ? e->sdf_length : 0;
                int d1, m1;
                limits(e->vdelay[1][v], e->vmaxdelay[1][v], extra, &d1, &m1);
                int rr = g.size() > 1 ? e->vline[1][v].init(e->L, d1, m1, e->rs)
                                      : e->vline[1][v].init(e->L, 0, 0, e->rs);     // 1:1: dai.c delays it
                if (rr != BFHIP_OK) return fail(rr, "delay buffer allocation failed");
                n_ops_max += e->vline[1][v].n_full_cap + 10;
            }
            if (!e->vin_list.empty()) {
                HIPCHK(dev_alloc((void **)&e->d_incopy, e->vin_list.size() * (size_t)e->L * 8));
                HIPCHK(hipMemset(e->d_incopy, 0, e->vin_list.size() * (size_t)e->L * 8));
            }
            e->vjobs_slot = (n_ops_max + 8) * sizeof(ByteOp) + (e->n_ch[0] + e->n_ch[1] + 8) * 64;
            HIPCHK(dev_alloc(&e->d_vjobs, 2 * e->vjobs_slot));
            HIPCHK(e->st_vin.init(e->vjobs_slot));
            HIPCHK(e->st_vout.init(e->vjobs_slot));
            e->vline[0].resize(e->n_ch[0]); e->vline[1].resize(e->n_ch[1]);
        }
        // outputs K3 does not requantise itself (the dither pass or the N:1 mix does, from the time
        // samples K3 leaves in d_timeout)
        // ... and outputs another engine converts: nobody here writes them or their overflow state
        if (!e->vout_groups.empty() || !e->dither_channels.empty() || e->sharded) {
            std::vector<unsigned char> skip(e->n_ch[1], 0);
            for (auto &g : e->vout_groups) for (int v : g) skip[v] = 1;
            for (int c : e->dither_channels) skip[c] = 1;
            for (int o = 0; o < e->n_ch[1]; o++) if (!e->out_active[o]) skip[o] = 1;
            HIPCHK(dev_alloc((void **)&e->d_skip_quant, skip.size()));
            HIPCHK(hipMemcpy(e->d_skip_quant, skip.data(), skip.size(), hipMemcpyHostToDevice));
        }
        if (!e->vout_groups.empty() || !e->dither_channels.empty()) {
            HIPCHK(dev_alloc(&e->d_timeout, (size_t)e->n_ch[1] * e->L * e->rs));
            HIPCHK(hipMemset(e->d_timeout, 0, (size_t)e->n_ch[1] * e->L * e->rs));
        }
    }
    // wide interleaved sides: one uniform frame of >= 128 channels (words of 2, 4 or 8 bytes, channel p
    // at byte p * bytes of the frame), nothing on that side that reads its samples from a private
    // copy already (N:1 inputs, sub-sample delayed inputs); BFHIP_WIDE_IO=1 takes every eligible
    // side, 0 none
    for (int io = 0; io < 2; io++) {
        const int n = e->n_phys[io];
        bool ok = n >= 1 && !e->big;
        for (int p = 0; p < n && ok; p++) {
            const bfhip_format &f = e->fmt[io][p];
            ok = (f.bytes == 2 || f.bytes == 4 || f.bytes == 8) && f.bytes == e->fmt[io][0].bytes &&
                 f.sample_spacing == n && f.byte_offset == p * f.bytes;
        }
        if (io == 0) ok = ok && e->vin_list.empty() && !side_uses_subdelay(e, 0);
        int want = 0;          // (measured on config D, 256 + 256 channels: no gain, DESIGN 6 -- kept as an option)
        if (const char *env = getenv("BFHIP_WIDE_IO")) want = atoi(env) != 0;
        e->wide[io] = ok && want;
        if (e->wide[io]) {
            const size_t bytes = (size_t)n * e->L * e->fmt[io][0].bytes;
            HIPCHK(dev_alloc((void **)&e->d_planar[io], bytes));
            HIPCHK(hipMemset(e->d_planar[io], 0, bytes));
        }
    }
    if (e->wide[1] && e->sharded) {
        std::vector<unsigned char> skip(e->n_phys[1], 0);
        for (int v = 0; v < e->n_ch[1]; v++) if (!e->out_active[v]) skip[e->v2p[1][v]] = 1;
        HIPCHK(dev_alloc((void **)&e->d_phys_skip, skip.size()));
        HIPCHK(hipMemcpy(e->d_phys_skip, skip.data(), skip.size(), hipMemcpyHostToDevice));
    }
    int r = upload_formats(e);
    if (r != BFHIP_OK) return r;
    HIPCHK(dev_alloc((void **)&e->d_over, e->n_ch[1] * sizeof(DevOverflow)));
    HIPCHK(dev_alloc((void **)&e->d_status, sizeof(int)));
    HIPCHK(hipMemset(e->d_status, 0, sizeof(int)));
    e->raw_bytes[0] = raw_extent(e->fmt[0], e->n_phys[0], e->L);
    e->raw_bytes[1] = raw_extent(e->fmt[1], e->n_phys[1], e->L);
    build_owned_runs(e);
    HIPCHK(dev_alloc((void **)&e->d_rawin, e->raw_bytes[0] + 16));       // +16: staged in 16-byte words
    HIPCHK(dev_alloc((void **)&e->d_rawout, e->raw_bytes[1] + 16));
    HIPCHK(hipMemset(e->d_rawout, 0, e->raw_bytes[1]));
    // classify filters: who owns a private ring, who must materialise its output
    {
        const int F = (int)e->filters.size();
        e->level.assign(F, 0); e->owner_index.assign(F, -1); e->y_index.assign(F, -1);
        e->sink_index.assign(F, -1); e->fade_index.assign(F, -1); e->is_source.assign(F, 0);
        e->promoted.assign(F, nullptr);
        e->n_owners = e->n_y = e->n_sinks = e->n_fadeable = 0;
        int maxlevel = 0;
        for (int fi = 0; fi < F; fi++) {
            const Filter &f = e->filters[fi];
            for (int g : f.in_f) { e->is_source[g] = 1; e->level[fi] = std::max(e->level[fi], e->level[g] + 1); }
            maxlevel = std::max(maxlevel, e->level[fi]);
            if (!f.active) continue;               // run by another engine: no ring, no buffers here
            if (f.in_ch.size() != 1 || !f.in_f.empty()) e->owner_index[fi] = e->n_owners++;
            if (!f.in_f.empty()) e->sink_index[fi] = e->n_sinks++;
            if (f.crossfade) e->fade_index[fi] = e->n_fadeable++;
        }
        for (int fi = 0; fi < F; fi++)
            if (e->filters[fi].active && (e->is_source[fi] || e->filters[fi].crossfade)) e->y_index[fi] = e->n_y++;
        e->n_levels = maxlevel + 1;
        auto zalloc = [&](void **p, size_t bytes) -> int {
            if (bytes == 0) return BFHIP_OK;
            if (dev_alloc(p, bytes) != hipSuccess) return fail(BFHIP_ENOMEM, "out of device memory (%zu bytes)", bytes);
            HIPCHK(hipMemset(*p, 0, bytes));
            return BFHIP_OK;
        };
        if ((r = zalloc(&e->d_fring, (size_t)e->n_owners * e->N * L * e->csize())) != BFHIP_OK) return r;
        if ((r = zalloc(&e->d_Y, (size_t)e->n_y * L * e->csize())) != BFHIP_OK) return r;
        if ((r = zalloc(&e->d_Yold, (size_t)e->n_fadeable * L * e->csize())) != BFHIP_OK) return r;
        if ((r = zalloc(&e->d_evalprev, (size_t)e->n_sinks * L * e->rs)) != BFHIP_OK) return r;
    }
    if ((r = dither_upload(e)) != BFHIP_OK) return r;
    if (e->pipelined) {
        HIPCHK(hipStreamCreateWithFlags(&e->s_in, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&e->s_out, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            HIPCHK(hipEventCreateWithFlags(&e->ev_in[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&e->ev_mac[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&e->ev_out[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&e->ev_io[i], hipEventDisableTiming));
        }
        for (int i = 0; i < 3; i++) HIPCHK(hipEventCreateWithFlags(&e->ev_mac3[i], hipEventDisableTiming));
    }
    e->ls = e->stream;
    e->finalized = true;
    r = bfhip_engine_reset_overflow(e);
    if (r != BFHIP_OK) return r;
    // coefficient sets loaded from device memory are checked here
    int bad = 0;
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    HIPCHK(hipMemcpy(&bad, e->d_bad, sizeof(int), hipMemcpyDeviceToHost));
    if (bad) return fail(BFHIP_EINVAL, "NaN or Inf value among coefficients.");
    return build_plan(e);
}

// Exact history semantics for run-time changes (SURVEY 7 "hard parts"): the reference scales
// and delays a block when it ENTERS the filter's ring, so a change must not touch the blocks
// already in it.  A filter that shares its input's ring gets its own ring at that moment.
static int promote_filter(bfhip_engine *e, int fi) {
    if (!e->finalized || e->owner_index[fi] >= 0 || e->promoted[fi]) return BFHIP_OK;
    const Filter &f = e->filters[fi];
    if (!f.active) {
        // run by another engine, which gives it its private ring there; here only the plan's shape follows
        if (f.in_ch.size() == 1 && f.in_f.empty()) { e->promoted[fi] = PROMOTED_ELSEWHERE; e->plan_dirty = true; }
        return BFHIP_OK;
    }
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    const size_t bytes = (size_t)e->N * e->L * e->csize();
    void *ring = nullptr;
    if (dev_alloc(&ring, bytes) != hipSuccess) return fail(BFHIP_ENOMEM, "out of device memory for a private ring");
    HIPCHK(hipMemsetAsync(ring, 0, bytes, e->stream));
    const int ch = f.in_ch[0];
    const int delay = clamp_delay(e, f.delayblocks);
    const double sc = f.in_scale[0] * e->fmt[0][e->v2p[0][ch]].scale;
    const int n_valid = (int)std::min<unsigned long long>(e->blocks_done, (unsigned long long)e->N);
    if (n_valid > 0) {
        const dim3 grid((e->L + 255) / 256, n_valid);
        const unsigned char *src = (const unsigned char *)e->d_ring + (size_t)ch * e->R * e->L * e->csize();
        if (e->rs == 4)
            hipLaunchKernelGGL(promote_ring_kernel<float>, grid, dim3(256), 0, e->stream, (const c2<float> *)src, e->R,
                               (c2<float> *)ring, e->N, e->L, e->blockcounter, delay, (float)sc, n_valid);
        else
            hipLaunchKernelGGL(promote_ring_kernel<double>, grid, dim3(256), 0, e->stream, (const c2<double> *)src, e->R,
                               (c2<double> *)ring, e->N, e->L, e->blockcounter, delay, sc, n_valid);
        HIPCHK(hipGetLastError());
    }
    e->promoted[fi] = ring;
    e->plan_dirty = true;
    return BFHIP_OK;
}

int bfhip_engine_set_coeff(bfhip_engine *e, int filter, int coeff) {
    if (!e || filter < 0 || filter >= (int)e->filters.size() || coeff >= (int)e->coeffs.size())
        return fail(BFHIP_EINVAL, "set_coeff: bad argument");
    if (e->filters[filter].coeff != coeff) { e->filters[filter].coeff = coeff; e->plan_dirty = true; }
    return BFHIP_OK;
}

int bfhip_engine_set_delayblocks(bfhip_engine *e, int filter, int blocks) {
    if (!e || filter < 0 || filter >= (int)e->filters.size()) return fail(BFHIP_EINVAL, "set_delayblocks: bad argument");
    if (e->filters[filter].delayblocks != blocks) {
        int r = promote_filter(e, filter);
        if (r != BFHIP_OK) return r;
        e->filters[filter].delayblocks = blocks; e->plan_dirty = true;
    }
    return BFHIP_OK;
}

int bfhip_engine_set_scale(bfhip_engine *e, int filter, int io, int index, double scale) {
    if (!e || filter < 0 || filter >= (int)e->filters.size() || io < 0 || io > 1) return fail(BFHIP_EINVAL, "set_scale: bad argument");
    auto &v = io == 0 ? e->filters[filter].in_scale : e->filters[filter].out_scale;
    if (index < 0 || index >= (int)v.size()) return fail(BFHIP_EINVAL, "set_scale: bad index");
    if (v[index] != scale) {
        if (io == 0) {
            int r = promote_filter(e, filter);
            if (r != BFHIP_OK) return r;
        }
        v[index] = scale; e->plan_dirty = true;
    }
    return BFHIP_OK;
}

int bfhip_engine_set_fscale(bfhip_engine *e, int filter, int index, double scale) {
    if (!e || filter < 0 || filter >= (int)e->filters.size()) return fail(BFHIP_EINVAL, "set_fscale: bad argument");
    auto &v = e->filters[filter].in_fscale;
    if (index < 0 || index >= (int)v.size()) return fail(BFHIP_EINVAL, "set_fscale: bad index");
    if (v[index] != scale) { v[index] = scale; e->plan_dirty = true; }
    return BFHIP_OK;
}

int bfhip_engine_inputs_dev(bfhip_engine *e, const void *rawin_dev) {
    int r = ensure_ready(e);
    if (r != BFHIP_OK) return r;
    if (!rawin_dev) return fail(BFHIP_EINVAL, "inputs: null buffer");
    if ((r = flush_pending(e)) != BFHIP_OK) return r;
    e->ls = e->stream;
    timing_begin(e);
    if ((r = record(e, 0)) != BFHIP_OK) return r;
    if ((r = do_inputs(e, rawin_dev)) != BFHIP_OK) return r;
    return record(e, 1);
}

int bfhip_engine_mac_dev(bfhip_engine *e, void *z_dev) {
    int r = ensure_ready(e);
    if (r != BFHIP_OK) return r;
    if (!z_dev) return fail(BFHIP_EINVAL, "mac: null buffer");
    if ((r = flush_pending(e)) != BFHIP_OK) return r;      // an output owed by bfhip_engine_block_dev goes first
    e->ls = e->stream;
    timing_begin(e);
    if ((r = do_levels(e)) != BFHIP_OK) return r;
    if ((r = record(e, 2)) != BFHIP_OK) return r;
    if (e->n_chunks == 1 && e->n_out_padded == e->n_ch[1]) {
        if ((r = do_mac(e, z_dev)) != BFHIP_OK) return r;
        return record(e, 3);
    }
    r = do_mac(e, e->d_Zp);
    if (r != BFHIP_OK) return r;
    if ((r = record(e, 3)) != BFHIP_OK) return r;
    hipError_t err = hipSuccess;
    if (e->rs == 4) launch_sum<float>(e, e->d_Zp, z_dev, &err); else launch_sum<double>(e, e->d_Zp, z_dev, &err);
    if (err != hipSuccess) return fail(BFHIP_EHIP, "sum_partials launch: %s", hipGetErrorString(err));
    return BFHIP_OK;
}

int bfhip_engine_outputs_dev(bfhip_engine *e, const void *z_dev, int first, int count, void *rawout_dev) {
    int r = ensure_ready(e);
    if (r != BFHIP_OK) return r;
    if ((r = flush_pending(e)) != BFHIP_OK) return r;      // an output owed by bfhip_engine_block_dev goes first
    if (first < 0 || count < 0 || first + count > e->n_ch[1]) return fail(BFHIP_EINVAL, "outputs: channel range");
    if (!z_dev || !rawout_dev) return fail(BFHIP_EINVAL, "outputs: null buffer");
    e->ls = e->stream;
    timing_begin(e);
    if ((r = record(e, 4)) != BFHIP_OK) return r;
    if ((r = do_outputs(e, z_dev, 0, 1, first, count, rawout_dev)) != BFHIP_OK) return r;
    return record(e, 5);
}

int bfhip_engine_outputs_inputs_dev(bfhip_engine *e, const void *z_dev, int first, int count,
                                    void *rawout_dev, const void *rawin_dev) {
    int r = ensure_ready(e);
    if (r != BFHIP_OK) return r;
    if ((r = flush_pending(e)) != BFHIP_OK) return r;      // an output owed by bfhip_engine_block_dev goes first
    if (first < 0 || count < 0 || first + count > e->n_ch[1]) return fail(BFHIP_EINVAL, "outputs: channel range");
    if (!z_dev || !rawout_dev || !rawin_dev) return fail(BFHIP_EINVAL, "outputs_inputs: null buffer");
    if (!e->dither_channels.empty() || e->has_vchan || count == 0 || e->big) {
        // the dither pass follows the inverse transforms: keep the two launches apart
        if ((r = bfhip_engine_outputs_dev(e, z_dev, first, count, rawout_dev)) != BFHIP_OK) return r;
        return bfhip_engine_inputs_dev(e, rawin_dev);
    }
    e->ls = e->stream;
    timing_begin(e);
    if ((r = record(e, 0)) != BFHIP_OK) return r;      // the fused launch is timed in the input slot
    hipError_t err = hipSuccess;
    const int slot = (int)(e->blockcounter % (unsigned int)e->R);
    if ((r = pre_inputs(e, rawin_dev)) != BFHIP_OK) return r;
    if (e->wave) { DISPATCH_WAVE(launch_io_wave, e, z_dev, (size_t)0, 1, first, count, k3_target(e, rawout_dev), (const uint8_t *)rawin_dev, slot, &err) }
    else DISPATCH(launch_io, e, z_dev, first, count, k3_target(e, rawout_dev), (const uint8_t *)rawin_dev, slot, &err);
    if (err != hipSuccess) return fail(BFHIP_EHIP, "io launch: %s", hipGetErrorString(err));
    if ((r = post_outputs(e, rawout_dev, first, count)) != BFHIP_OK) return r;
    return record(e, 1);
}

int bfhip_engine_advance(bfhip_engine *e) {
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    advance(e);
    return BFHIP_OK;
}

static int block_dev_impl(bfhip_engine *e, const void *rawin_dev, void *rawout_dev,
                          hipEvent_t in_ready, hipEvent_t out_done);

int bfhip_engine_block_dev(bfhip_engine *e, const void *rawin_dev, void *rawout_dev) {
    return block_dev_impl(e, rawin_dev, rawout_dev, nullptr, nullptr);
}

int bfhip_engine_block_dev_ev(bfhip_engine *e, const void *rawin_dev, void *rawout_dev,
                              void *in_ready_event, void *out_done_event) {
    return block_dev_impl(e, rawin_dev, rawout_dev, (hipEvent_t)in_ready_event, (hipEvent_t)out_done_event);
}

static int block_dev_impl(bfhip_engine *e, const void *rawin_dev, void *rawout_dev,
                          hipEvent_t in_ready, hipEvent_t out_done) {
    int r = ensure_ready(e);
    if (r != BFHIP_OK) return r;
    if (!rawin_dev || !rawout_dev) return fail(BFHIP_EINVAL, "block_dev: null buffer");
    const bool pipe = e->pipelined;
    const int buf = (int)(e->blocks_done & 1);
    void *Zp = ((pipe || e->defer_out) && buf) ? e->d_Zp2 : e->d_Zp;
    timing_begin(e);
    LsGuard guard{e};

    if (e->pipe2) {
        // side stream: [K3 of block t-2 | K1 of block t]; main stream: MAC of block t behind it.  The
        // MAC of t-1 (main) and this launch (side) overlap: K1 fills the spare ring slot, K3 reads the
        // third Zp buffer.
        const int zi = (int)(e->blocks_done % 3ull), par = (int)(e->blocks_done & 1ull);
        void *Zq = zi == 0 ? e->d_Zp : (zi == 1 ? e->d_Zp2 : e->d_Zp3);
        e->ls = e->s_in;
        if (in_ready) HIPCHK(hipStreamWaitEvent(e->s_in, in_ready, 0));
        if ((r = record(e, 0)) != BFHIP_OK) return r;
        if (e->pendq.size() >= 2) {
            const bfhip_engine::Pending p = e->pendq.front();     // popped once its launch is in the stream
            HIPCHK(hipStreamWaitEvent(e->s_in, p.mac_done, 0));
            if (p.n_chunks <= 2) {
                hipError_t err = hipSuccess;
                const int slot = (int)(e->blockcounter % (unsigned int)e->R);
                if ((r = pre_inputs(e, rawin_dev)) != BFHIP_OK) return r;
                DISPATCH_WAVE(launch_io_wave, e, p.Zp, p.chunk_stride, p.n_chunks, 0, e->n_ch[1],
                              k3_target(e, p.rawout), (const uint8_t *)rawin_dev, slot, &err)
                if (err != hipSuccess) return fail(BFHIP_EHIP, "io launch: %s", hipGetErrorString(err));
                e->pendq.pop_front();
                if ((r = post_outputs(e, p.rawout, 0, e->n_ch[1])) != BFHIP_OK) return r;
            } else {
                if ((r = do_outputs(e, p.Zp, p.chunk_stride, p.n_chunks, 0, e->n_ch[1], p.rawout)) != BFHIP_OK) return r;
                e->pendq.pop_front();
                if ((r = do_inputs(e, rawin_dev)) != BFHIP_OK) return r;
            }
            if (p.out_done) HIPCHK(hipEventRecord(p.out_done, e->s_in));
        } else {
            if ((r = do_inputs(e, rawin_dev)) != BFHIP_OK) return r;
        }
        if ((r = record(e, 1)) != BFHIP_OK) return r;
        HIPCHK(hipEventRecord(e->ev_io[par], e->s_in));
        e->ls = e->stream;
        HIPCHK(hipStreamWaitEvent(e->stream, e->ev_io[par], 0));
        if ((r = do_levels(e)) != BFHIP_OK) return r;
        if ((r = record(e, 2)) != BFHIP_OK) return r;
        if ((r = do_mac(e, Zq)) != BFHIP_OK) return r;
        if ((r = record(e, 3)) != BFHIP_OK) return r;
        bfhip_engine::Pending np;
        np.Zp = Zq; np.chunk_stride = (size_t)e->n_out_padded * e->L; np.n_chunks = e->n_chunks;
        if (e->n_chunks > 2) {
            // many partials (few long filters): add them up with the whole chip here, in place, so that
            // the fused transform launch reads one spectrum per channel
            hipError_t err = hipSuccess;
            if (e->rs == 4) launch_sum<float>(e, Zq, Zq, &err); else launch_sum<double>(e, Zq, Zq, &err);
            if (err != hipSuccess) return fail(BFHIP_EHIP, "sum_partials launch: %s", hipGetErrorString(err));
            np.n_chunks = 1;
        }
        HIPCHK(hipEventRecord(e->ev_mac3[zi], e->stream));
        np.rawout = rawout_dev; np.out_done = out_done; np.mac_done = e->ev_mac3[zi];
        e->pendq.push_back(np);
        advance(e);
        return BFHIP_OK;
    }

    if (e->defer_out && !pipe) {
        // [K3 of block t-1 | K1 of block t] in one launch, then the MAC of block t; K3 of block t is
        // owed to the next call (or to sync).  Same kernels, same order per channel: same bits.
        e->ls = e->stream;
        if (in_ready) HIPCHK(hipStreamWaitEvent(e->stream, in_ready, 0));
        if ((r = record(e, 0)) != BFHIP_OK) return r;
        if (!e->pendq.empty() && e->pendq.front().n_chunks <= 2) {
            const bfhip_engine::Pending p = e->pendq.front();     // popped once its launch is in the stream
            hipError_t err = hipSuccess;
            const int slot = (int)(e->blockcounter % (unsigned int)e->R);
            if ((r = pre_inputs(e, rawin_dev)) != BFHIP_OK) return r;
            DISPATCH_WAVE(launch_io_wave, e, p.Zp, p.chunk_stride, p.n_chunks, 0, e->n_ch[1],
                          k3_target(e, p.rawout), (const uint8_t *)rawin_dev, slot, &err)
            if (err != hipSuccess) return fail(BFHIP_EHIP, "io launch: %s", hipGetErrorString(err));
            e->pendq.pop_front();
            if ((r = post_outputs(e, p.rawout, 0, e->n_ch[1])) != BFHIP_OK) return r;
            if (p.out_done) HIPCHK(hipEventRecord(p.out_done, e->stream));
        } else {
            if ((r = flush_pending(e)) != BFHIP_OK) return r;
            if ((r = do_inputs(e, rawin_dev)) != BFHIP_OK) return r;
        }
        if ((r = record(e, 1)) != BFHIP_OK) return r;
        if ((r = do_levels(e)) != BFHIP_OK) return r;
        if ((r = record(e, 2)) != BFHIP_OK) return r;
        if ((r = do_mac(e, Zp)) != BFHIP_OK) return r;
        if ((r = record(e, 3)) != BFHIP_OK) return r;
        bfhip_engine::Pending np;
        np.Zp = Zp; np.chunk_stride = (size_t)e->n_out_padded * e->L; np.n_chunks = e->n_chunks;
        if (e->n_chunks > 2) {
            // many partials (few outputs, many inputs: an output-sharded rank): add them up with the
            // whole chip here, in place, same order -- the next call's fused [K3 | K1] launch then
            // reads one spectrum per channel (and exists: it takes at most two)
            hipError_t err = hipSuccess;
            if (e->rs == 4) launch_sum<float>(e, Zp, Zp, &err); else launch_sum<double>(e, Zp, Zp, &err);
            if (err != hipSuccess) return fail(BFHIP_EHIP, "sum_partials launch: %s", hipGetErrorString(err));
            np.n_chunks = 1;
        }
        np.rawout = rawout_dev; np.out_done = out_done; np.mac_done = nullptr;
        e->pendq.push_back(np);
        advance(e);
        return BFHIP_OK;
    }

    // K1 on the input stream.  It overwrites the ring slot of block t-R, last read by the MAC
    // of block t-2 (the MAC of t-1 reaches back only N = R-1 blocks).
    e->ls = pipe ? e->s_in : e->stream;
    if (pipe && e->blocks_done >= 2) HIPCHK(hipStreamWaitEvent(e->s_in, e->ev_mac[buf], 0));
    // the caller's producer of rawin_dev (any stream): K1 must not start before it is done
    if (in_ready) HIPCHK(hipStreamWaitEvent(e->ls, in_ready, 0));
    if ((r = record(e, 0)) != BFHIP_OK) return r;
    if ((r = do_inputs(e, rawin_dev)) != BFHIP_OK) return r;
    if ((r = record(e, 1)) != BFHIP_OK) return r;
    if (pipe) HIPCHK(hipEventRecord(e->ev_in[buf], e->s_in));

    // per-filter kernels and the crossbar MAC on the main stream; Zp[buf] is free once the
    // output pass of block t-2 has read it
    e->ls = e->stream;
    if (pipe) {
        HIPCHK(hipStreamWaitEvent(e->stream, e->ev_in[buf], 0));
        if (e->blocks_done >= 2) HIPCHK(hipStreamWaitEvent(e->stream, e->ev_out[buf], 0));
    }
    if ((r = do_levels(e)) != BFHIP_OK) return r;
    if ((r = record(e, 2)) != BFHIP_OK) return r;
    if ((r = do_mac(e, Zp)) != BFHIP_OK) return r;
    if ((r = record(e, 3)) != BFHIP_OK) return r;
    if (pipe) HIPCHK(hipEventRecord(e->ev_mac[buf], e->stream));

    // K3 (+ dither pass) on the output stream
    e->ls = pipe ? e->s_out : e->stream;
    if (pipe) HIPCHK(hipStreamWaitEvent(e->s_out, e->ev_mac[buf], 0));
    if ((r = record(e, 4)) != BFHIP_OK) return r;
    if ((r = do_outputs(e, Zp, (size_t)e->n_out_padded * e->L, e->n_chunks, 0, e->n_ch[1], rawout_dev)) != BFHIP_OK) return r;
    if ((r = record(e, 5)) != BFHIP_OK) return r;
    if (pipe) HIPCHK(hipEventRecord(e->ev_out[buf], e->s_out));
    // rawout_dev of THIS block is complete (and rawin_dev no longer needed) once this fires
    if (out_done) HIPCHK(hipEventRecord(out_done, e->ls));
    e->ls = e->stream;
    advance(e);
    return BFHIP_OK;
}

int bfhip_engine_enable_pairs(bfhip_engine *e, int on) {
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    if (e->finalized) return fail(BFHIP_ESTATE, "enable_pairs after finalize");
    e->pairs = on != 0;
    return BFHIP_OK;
}

// Two consecutive blocks, ONE pass over the coefficients (mac_xbar2_kernel).  Falls back to two
// single blocks whenever the plan is not a plain uniform crossbar or its rings are not full yet:
// the outputs are the same bits either way.
int bfhip_engine_block_pair_dev(bfhip_engine *e, const void *rawin0_dev, void *rawout0_dev,
                                const void *rawin1_dev, void *rawout1_dev) {
    int r = ensure_ready(e);
    if (r != BFHIP_OK) return r;
    if (!rawin0_dev || !rawout0_dev || !rawin1_dev || !rawout1_dev) return fail(BFHIP_EINVAL, "block_pair_dev: null buffer");
    bool plain = e->pairs && e->all_dense && !e->mac_diag && e->mac_nt && e->mac_unroll == 0 && !e->any_fading &&
                 !e->has_vchan && !e->big && !e->rt.on && e->d_Zp2 != nullptr && e->blocks_done >= (unsigned long long)e->N;
    for (auto &lj : e->level_jobs) plain = plain && !lj.n_fill && !lj.n_filt && !lj.n_fade;
    if (!plain) {
        if ((r = bfhip_engine_block_dev(e, rawin0_dev, rawout0_dev)) != BFHIP_OK) return r;
        return bfhip_engine_block_dev(e, rawin1_dev, rawout1_dev);
    }
    if ((r = flush_pending(e)) != BFHIP_OK) return r;          // an output owed by a single block goes first
    LsGuard guard{e};
    e->ls = e->stream;
    timing_begin(e);
    if ((r = record(e, 0)) != BFHIP_OK) return r;
    if ((r = do_inputs(e, rawin0_dev, 0u)) != BFHIP_OK) return r;
    if ((r = do_inputs(e, rawin1_dev, 1u)) != BFHIP_OK) return r;
    if ((r = record(e, 1)) != BFHIP_OK) return r;
    if ((r = record(e, 2)) != BFHIP_OK) return r;
    hipError_t err = hipSuccess;
    if (e->rs == 4) launch_mac2<float>(e, e->d_Zp, e->d_Zp2, &err); else launch_mac2<double>(e, e->d_Zp, e->d_Zp2, &err);
    if (err != hipSuccess) return fail(BFHIP_EHIP, "mac2 launch: %s", hipGetErrorString(err));
    if ((r = record(e, 3)) != BFHIP_OK) return r;
    if ((r = record(e, 4)) != BFHIP_OK) return r;
    const size_t stride = (size_t)e->n_out_padded * e->L;
    if ((r = do_outputs(e, e->d_Zp, stride, e->n_chunks, 0, e->n_ch[1], rawout0_dev)) != BFHIP_OK) return r;
    if ((r = do_outputs(e, e->d_Zp2, stride, e->n_chunks, 0, e->n_ch[1], rawout1_dev)) != BFHIP_OK) return r;
    if ((r = record(e, 5)) != BFHIP_OK) return r;
    e->n_pair_launches++;
    advance(e);
    advance(e);
    return BFHIP_OK;
}

unsigned long long bfhip_engine_pair_launches(const bfhip_engine *e) { return e ? e->n_pair_launches : 0ull; }

int bfhip_engine_flush(bfhip_engine *e) {
    if (!e || !e->finalized) return fail(BFHIP_ESTATE, "engine not finalized");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    return flush_pending(e);
}

int bfhip_engine_output_lag(const bfhip_engine *e) {
    if (!e || !e->finalized) return 0;
    return e->pipe2 ? 2 : ((e->defer_out && !e->pipelined) ? 1 : 0);
}

int bfhip_engine_sync(bfhip_engine *e) {
    if (!e || !e->finalized) return fail(BFHIP_ESTATE, "engine not finalized");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    { int _r = flush_pending(e); if (_r != BFHIP_OK) return _r; }
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    int st = 0;
    HIPCHK(hipMemcpy(&st, e->d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (st) HIPCHK(hipMemset(e->d_status, 0, sizeof(int)));
    return st;
}

int bfhip_engine_block(bfhip_engine *e, const void *rawin, void *rawout, bfhip_overflow overflow[]) {
    int r = ensure_ready(e);
    if (r != BFHIP_OK) return r;
    if (!rawin || !rawout) return fail(BFHIP_EINVAL, "block: null buffer");
    static_assert(sizeof(bfhip_overflow) == sizeof(DevOverflow), "overflow struct layout");
    hipStream_t sin = e->pipelined ? e->s_in : e->stream, sout = e->pipelined ? e->s_out : e->stream;
    if (e->pipe2) sout = e->s_in;              // the ping-pong schedule keeps the output passes on the side stream
    if (overflow) HIPCHK(hipMemcpyAsync(e->d_over, overflow, e->n_ch[1] * sizeof(DevOverflow), hipMemcpyHostToDevice, sout));
    HIPCHK(hipMemcpyAsync(e->d_rawin, rawin, e->raw_bytes[0], hipMemcpyHostToDevice, sin));
    if ((r = bfhip_engine_block_dev(e, e->d_rawin, e->d_rawout)) != BFHIP_OK) return r;
    if ((r = flush_pending(e)) != BFHIP_OK) return r;          // host buffers: this block's output is due now
    if (e->sharded) {
        // rawout is shared with the engines of the other filter processes: only this engine's
        // samples (and overflow entries) may land in it
        e->h_stage.resize(e->raw_bytes[1] + (size_t)e->n_ch[1] * sizeof(DevOverflow));
        HIPCHK(hipMemcpyAsync(e->h_stage.data(), e->d_rawout, e->raw_bytes[1], hipMemcpyDeviceToHost, sout));
        if (overflow) HIPCHK(hipMemcpyAsync(e->h_stage.data() + e->raw_bytes[1], e->d_over, e->n_ch[1] * sizeof(DevOverflow), hipMemcpyDeviceToHost, sout));
        const int st = bfhip_engine_sync(e);
        if (st < 0) return st;
        copy_owned(e, rawout, e->h_stage.data());
        if (overflow) copy_owned_overflow(e, overflow, e->h_stage.data() + e->raw_bytes[1]);
        return st;
    }
    HIPCHK(hipMemcpyAsync(rawout, e->d_rawout, e->raw_bytes[1], hipMemcpyDeviceToHost, sout));
    if (overflow) HIPCHK(hipMemcpyAsync(overflow, e->d_over, e->n_ch[1] * sizeof(DevOverflow), hipMemcpyDeviceToHost, sout));
    return bfhip_engine_sync(e);
}

int bfhip_engine_rt_begin(bfhip_engine *e, int flags) {
    int r = ensure_ready(e);
    if (r != BFHIP_OK) return r;
    if ((r = flush_pending(e)) != BFHIP_OK) return r;
    if (e->rt.on) return fail(BFHIP_ESTATE, "rt_begin: already in real-time mode");
    if ((r = sync_all(e)) != BFHIP_OK) return r;
    e->pipelined = false;                      // one stream: a period is needed back as soon as possible
    e->pipe2 = false;
    auto &rt = e->rt;
    rt.flags = flags;
    for (int p = 0; p < 2; p++) {
        HIPCHK(pin_alloc(&rt.h_in[p], e->raw_bytes[0] + 16, hipHostMallocDefault));
        HIPCHK(pin_alloc(&rt.h_out[p], e->raw_bytes[1] + 16, hipHostMallocDefault));
        HIPCHK(pin_alloc((void **)&rt.h_over[p], e->n_ch[1] * sizeof(DevOverflow), hipHostMallocDefault));
        HIPCHK(pin_alloc((void **)&rt.h_status[p], 2 * sizeof(int), hipHostMallocDefault));
        memset(rt.h_in[p], 0, e->raw_bytes[0]);
        memset(rt.h_out[p], 0, e->raw_bytes[1]);
        memset(rt.h_over[p], 0, e->n_ch[1] * sizeof(DevOverflow));
        rt.h_status[p][0] = rt.h_status[p][1] = 0;
        HIPCHK(hipEventCreateWithFlags(&rt.done[p], hipEventDisableTiming));
        if (flags & BFHIP_RT_OVERLAP) {
            HIPCHK(hipEventCreateWithFlags(&rt.ev_h2d[p], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&rt.ev_cmp[p], hipEventDisableTiming));
            HIPCHK(dev_alloc((void **)&rt.d_in[p], e->raw_bytes[0] + 16));
            HIPCHK(dev_alloc((void **)&rt.d_out[p], e->raw_bytes[1] + 16));
            HIPCHK(hipMemset(rt.d_out[p], 0, e->raw_bytes[1] + 16));
        }
    }
    if (flags & BFHIP_RT_OVERLAP) {
        HIPCHK(hipStreamCreateWithFlags(&rt.s_h2d, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&rt.s_d2h, hipStreamNonBlocking));
    }
    HIPCHK(dev_alloc((void **)&e->d_bs, sizeof(BlockState)));
    HIPCHK(dev_alloc((void **)&e->d_rt_arrive, sizeof(unsigned int)));
    HIPCHK(hipMemset(e->d_rt_arrive, 0, sizeof(unsigned int)));
    rt.on = true;
    rt.bs_synced = false;
    return BFHIP_OK;
}

int bfhip_engine_rt_end(bfhip_engine *e) {
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    if (!e->rt.on) return BFHIP_OK;
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    { int r = sync_all(e); if (r != BFHIP_OK) return r; }
    rt_release(e);
    return BFHIP_OK;
}

void *bfhip_engine_rt_buffer(bfhip_engine *e, int io, int index) {
    if (!e || !e->rt.on || io < 0 || io > 1 || index < 0 || index > 1) return nullptr;
    return io == 0 ? e->rt.h_in[index] : e->rt.h_out[index];
}

int bfhip_engine_rt_submit(bfhip_engine *e, const void *rawin) {
    if (!e || !e->rt.on) return fail(BFHIP_ESTATE, "rt_submit: not in real-time mode");
    auto &rt = e->rt;
    if (rt.submitted - rt.waited >= 2) return fail(BFHIP_ESTATE, "rt_submit: two periods already in flight");
    int r = ensure_ready(e);                   // control changes since the last block rebuild the plan
    if (r != BFHIP_OK) return r;
    const int p = (int)(rt.submitted & 1);
    if (rawin && rawin != rt.h_in[p]) memcpy(rt.h_in[p], rawin, e->raw_bytes[0]);
    if (!rt.bs_synced || rt.bs_t != e->blockcounter) {
        // blocks went through the other entry points (or this is the first period)
        BlockState bs;
        bs.t = e->blockcounter;
        bs.age = (int)std::min<unsigned long long>(e->blocks_done + 1, (unsigned long long)e->N);
        bs.n_blocks = 0; bs.pad = 0;
        bs.wrap_at = e->wrap_at; bs.wrap_by = e->wrap_by;
        HIPCHK(hipMemcpyAsync(e->d_bs, &bs, sizeof(bs), hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        rt.bs_synced = true;
    }
    rt.h_status[p][1] = 0;
    if (rt_graphable(e) && rt.primed) {
        if (!rt.valid[p] && (r = rt_capture(e, p)) != BFHIP_OK) return r;
        HIPCHK(hipGraphLaunch(rt.exec[p], e->stream));
        rt.n_graph++;
    } else {
        // first block after a plan change (also sets the kernels' LDS attributes outside any
        // capture), cross-fade blocks, per-block job tables: plain launches
        timing_begin(e);
        if ((r = rt_enqueue(e, p)) != BFHIP_OK) return r;
        rt.primed = true;
        rt.n_direct++;
    }
    if (!(rt.flags & BFHIP_RT_OVERLAP)) HIPCHK(hipEventRecord(rt.done[p], e->stream));   // else: after the download
    rt.submitted++;
    advance(e);
    rt.bs_t = e->blockcounter;
    return BFHIP_OK;
}

int bfhip_engine_rt_wait(bfhip_engine *e, void *rawout, bfhip_overflow overflow[]) {
    if (!e || !e->rt.on) return fail(BFHIP_ESTATE, "rt_wait: not in real-time mode");
    auto &rt = e->rt;
    if (rt.submitted == rt.waited) return fail(BFHIP_ESTATE, "rt_wait: nothing in flight");
    const int p = (int)(rt.waited & 1);
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    if ((rt.flags & BFHIP_RT_SPIN) && !(rt.flags & BFHIP_RT_OVERLAP)) {
        // the tail kernel's last store is the sequence word: watch it from the CPU; after 2 ms
        // by the clock (checked every 256 polls) fall back to the runtime, so that a device
        // error cannot hang the caller and a long block does not burn the core for its whole length
        volatile int *seq = rt.h_status[p] + 1;
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned spin = 1; *seq == 0; spin++) {
            __builtin_ia32_pause();
            if ((spin & 255u) == 0) {
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 2000000L) break;
            }
        }
        // seen: every store of the block (the tail kernel's copy-out included) is visible
        if (*seq == 0 || (rt.flags & BFHIP_RT_COPY_ENGINE)) HIPCHK(hipEventSynchronize(rt.done[p]));
    } else {
        HIPCHK(hipEventSynchronize(rt.done[p]));
    }
    rt.waited++;
    if (rawout && rawout != rt.h_out[p]) copy_owned(e, rawout, rt.h_out[p]);      // (a sharded engine: its own samples only)
    if (overflow) {
        // The reference reads icomm->overflow[ch] at every block and writes it back (bfrun.c:1929-1936),
        // so a reset by another process (bf_reset_peak: the CLI's peak reset) is picked up by the next
        // block.  Here the structs live on the device between periods: an entry the HOST changed since
        // this engine last wrote it becomes the device's state (queued behind whatever is in flight),
        // and the host's value stands until a period that started after the change reports.
        const int O = e->n_ch[1];
        const DevOverflow *dev = rt.h_over[p];
        const unsigned long long period = rt.waited - 1;                 // the period this call collected
        if ((int)rt.last_over.size() != O) { rt.last_over.assign(O, DevOverflow()); rt.over_from.assign(O, 0ull); rt.last_valid = false; }
        for (int o = 0; o < O; o++) {
            if (e->sharded && !e->out_active[o]) continue;
            DevOverflow &seen = rt.last_over[o];
            if (rt.last_valid && memcmp(&overflow[o], &seen, sizeof(DevOverflow)) != 0) {
                HIPCHK(hipMemcpyAsync(&e->d_over[o], &overflow[o], sizeof(DevOverflow), hipMemcpyHostToDevice, e->stream));
                memcpy(&seen, &overflow[o], sizeof(DevOverflow));
                rt.over_from[o] = rt.submitted;                           // periods submitted from now on count on the new state
                continue;
            }
            if (period < rt.over_from[o]) continue;                       // still a period from before the host's change
            memcpy(&overflow[o], &dev[o], sizeof(DevOverflow));
            memcpy(&seen, &dev[o], sizeof(DevOverflow));
        }
        rt.last_valid = true;
    }
    return rt.h_status[p][0];
}

int bfhip_engine_rt_block(bfhip_engine *e, const void *rawin, void *rawout, bfhip_overflow overflow[]) {
    int r = bfhip_engine_rt_submit(e, rawin);
    if (r != BFHIP_OK) return r;
    return bfhip_engine_rt_wait(e, rawout, overflow);
}

int bfhip_engine_rt_stats(const bfhip_engine *e, unsigned long long *graph_blocks,
                          unsigned long long *direct_blocks, unsigned long long *captures) {
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    if (graph_blocks) *graph_blocks = e->rt.n_graph;
    if (direct_blocks) *direct_blocks = e->rt.n_direct;
    if (captures) *captures = e->rt.n_capture;
    return BFHIP_OK;
}

// ---- self test of the wave FFT's host half: its per-thread twiddle table, no device involved ----
int bfhip_selftest_fail_alloc(int nth) {
    const int left = g_fail_alloc;             // > 0: the allocation armed before was never reached
    g_fail_alloc = nth > 0 ? nth : 0;
    return left;
}

int bfhip_selftest_wave_twiddles(int log2l, int realsize, void *out, int out_bytes) {
    if (!wave_fft_ok(log2l, realsize)) return fail(BFHIP_EINVAL, "selftest_wave_twiddles: length / precision not covered");
    const std::vector<unsigned char> t = make_wave_twiddle_table(log2l, realsize);
    if (out == nullptr) return (int)t.size();
    if ((size_t)out_bytes < t.size()) return fail(BFHIP_EINVAL, "selftest_wave_twiddles: buffer too small");
    memcpy(out, t.data(), t.size());
    return (int)t.size();
}

// ---- self test of the host-side delay machine: no device involved ---------------------------
struct bfhip_selftest_delay { DelayLine dl; };

bfhip_selftest_delay *bfhip_selftest_delay_new(int fragment, int initdelay, int maxdelay, int sample_size) {
    if (fragment < 1 || sample_size < 1 || initdelay < 0) { fail(BFHIP_EINVAL, "selftest_delay_new: bad argument"); return nullptr; }
    bfhip_selftest_delay *d = new bfhip_selftest_delay();
    d->dl.host = true;
    if (d->dl.init(fragment, initdelay, maxdelay, sample_size) != BFHIP_OK) { delete d; fail(BFHIP_ENOMEM, "selftest_delay_new: out of memory"); return nullptr; }
    return d;
}

int bfhip_selftest_delay_update(bfhip_selftest_delay *d, void *buf, int delay) {
    if (!d || !buf) return fail(BFHIP_EINVAL, "selftest_delay_update: bad argument");
    std::vector<ByteOp> ops;
    d->dl.update((uint8_t *)buf, delay, ops);
    for (const ByteOp &o : ops) {
        // the device runs every move with one thread per byte: source and destination of a move
        // must not overlap
        if (o.src && !(o.src + o.n <= o.dst || o.dst + o.n <= o.src)) return fail(BFHIP_ESTATE, "delay machine emitted an overlapping move");
        if (o.src) memcpy(o.dst, o.src, o.n); else memset(o.dst, 0, o.n);
    }
    return (int)ops.size();
}

void bfhip_selftest_delay_free(bfhip_selftest_delay *d) {
    if (!d) return;
    free(d->dl.arena);
    delete d;
}

int bfhip_engine_prewarm(bfhip_engine *e) {
    int r = ensure_ready(e);
    if (r != BFHIP_OK) return r;
    if (e->blocks_done != 0) return fail(BFHIP_ESTATE, "prewarm: blocks have been processed already");
    // N blocks of silence are what the zero-initialised rings hold: count them as processed, and
    // move the block counter past the point where (blockcounter - p - delay) would wrap
    e->blocks_done = (unsigned long long)e->N;
    e->blockcounter = (unsigned int)e->R;
    e->rt.bs_synced = false;
    return BFHIP_OK;
}

int bfhip_engine_set_status_dev(bfhip_engine *e, int *status_dev) {
    if (!e || !e->finalized) return fail(BFHIP_ESTATE, "set_status_dev: engine not finalized");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    if (!e->d_status_own) e->d_status_own = e->d_status;
    e->d_status = status_dev ? status_dev : e->d_status_own;
    e->rt.valid[0] = e->rt.valid[1] = false;
    return BFHIP_OK;
}

int bfhip_engine_set_stream(bfhip_engine *e, void *hip_stream) {
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    if (e->finalized) { int _r = flush_pending(e); if (_r != BFHIP_OK) return _r; }
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    e->stream = (hipStream_t)hip_stream;
    e->ls = e->stream;
    e->own_stream = false;
    return BFHIP_OK;
}

int bfhip_engine_get_overflow(bfhip_engine *e, int ch, bfhip_overflow *of) {
    if (!e || !e->finalized || ch < 0 || ch >= e->n_ch[1] || !of) return fail(BFHIP_EINVAL, "get_overflow: bad argument");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    { int _r = flush_pending(e); if (_r != BFHIP_OK) return _r; }
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    HIPCHK(hipMemcpy(of, e->d_over + ch, sizeof(DevOverflow), hipMemcpyDeviceToHost));
    return BFHIP_OK;
}

int bfhip_engine_reset_overflow(bfhip_engine *e) {
    if (!e || !e->finalized) return fail(BFHIP_ESTATE, "engine not finalized");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    std::vector<DevOverflow> v(e->n_ch[1]);
    for (int c = 0; c < e->n_ch[1]; c++) {
        memset(&v[c], 0, sizeof(DevOverflow));
        v[c].max = overflow_max(e->fmt[1][e->v2p[1][c]]);
    }
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    HIPCHK(hipMemcpy(e->d_over, v.data(), v.size() * sizeof(DevOverflow), hipMemcpyHostToDevice));
    return BFHIP_OK;
}

unsigned int bfhip_engine_blockcounter(const bfhip_engine *e) { return e ? e->blockcounter : 0; }
int bfhip_engine_block_mode(const bfhip_engine *e) {
    if (!e || !e->finalized) return -1;
    if (e->pipe2) return BFHIP_MODE_PINGPONG;
    return e->pipelined ? BFHIP_MODE_PIPELINED : (e->defer_out ? BFHIP_MODE_DEFERRED : BFHIP_MODE_SEQUENTIAL);
}
int bfhip_engine_uses_wave_fft(const bfhip_engine *e) { return e && e->wave ? 1 : 0; }
int bfhip_engine_uses_stream_layout(const bfhip_engine *e) { return e && e->hstream.base ? 1 : 0; }
int bfhip_engine_uses_diag_mac(const bfhip_engine *e) { return e && e->mac_diag ? 1 : 0; }
int bfhip_engine_ring_depth(const bfhip_engine *e) { return e ? e->R : 0; }

int bfhip_engine_enable_timing(bfhip_engine *e, int on) {
    if (!e) return fail(BFHIP_EINVAL, "null engine");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    if (on && e->ev.empty()) {
        e->ev.resize((size_t)MAX_TIMED * EV_PER_BLOCK);
        for (auto &x : e->ev) HIPCHK(hipEventCreate(&x));
        e->ev_mask.assign(MAX_TIMED, 0);
    }
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    e->timing = on != 0;
    e->timing_stride = on > 1 ? on : 1;
    e->ev_used = 0;
    e->timed_for = ~0ull;
    e->timed_now = false;
    return BFHIP_OK;
}

int bfhip_engine_get_timing(bfhip_engine *e, double ms[4]) {
    if (!e || !ms) return fail(BFHIP_EINVAL, "get_timing: bad argument");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    ms[0] = ms[1] = ms[2] = ms[3] = 0;
    int cnt[3] = {0, 0, 0};
    for (int i = 0; i < e->ev_used; i++) {
        for (int k = 0; k < 3; k++) {
            if (!(e->ev_mask[i] & (1 << k))) continue;      // phase not launched through a timed entry point
            float t = 0;
            HIPCHK(hipEventElapsedTime(&t, e->ev[(size_t)i * EV_PER_BLOCK + 2 * k], e->ev[(size_t)i * EV_PER_BLOCK + 2 * k + 1]));
            ms[k] += t;
            cnt[k]++;
        }
    }
    for (int k = 0; k < 3; k++) if (cnt[k] > 0) ms[k] /= cnt[k];
    ms[3] = cnt[1];
    e->ev_used = 0;
    return BFHIP_OK;
}

// The reference's `benchmark: true` table (bfrun.c:2035-2078), device side.
int bfhip_engine_stage_times(bfhip_engine *e, double ms[8]) {
    if (!e || !ms) return fail(BFHIP_EINVAL, "stage_times: bad argument");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    for (int k = 0; k < 8; k++) ms[k] = 0;
    double sum[EV_PAIRS] = {0, 0, 0, 0, 0};
    int n_blocks = 0;
    for (int i = 0; i < e->ev_used; i++) {
        if (!(e->ev_mask[i] & 2)) continue;                 // no MAC timed: not a whole block
        n_blocks++;
        for (int k = 0; k < EV_PAIRS; k++) {
            if (!(e->ev_mask[i] & (1 << k))) continue;
            float t = 0;
            HIPCHK(hipEventElapsedTime(&t, e->ev[(size_t)i * EV_PER_BLOCK + 2 * k], e->ev[(size_t)i * EV_PER_BLOCK + 2 * k + 1]));
            sum[k] += t;
        }
    }
    e->ev_used = 0;
    if (n_blocks == 0) return 0;
    const double post = sum[3] / n_blocks, out = sum[2] / n_blocks;
    ms[1] = sum[0] / n_blocks;                              // time2freq (+ raw2real, fused)
    ms[2] = sum[4] / n_blocks;                              // mixscale1: the per-filter input mixes / cascades
    ms[3] = sum[1] / n_blocks;                              // convolve (+ mixscale1 of plain filters, mixscale2: fused)
    ms[5] = out > post ? out - post : 0.0;                  // freq2time (+ real2raw of undithered 1:1 outputs, fused)
    ms[6] = post;                                           // real2raw: dither, N:1 mix, sub-sample delay passes
    ms[7] = ms[1] + ms[2] + ms[3] + ms[5] + ms[6];
    return n_blocks;
}

int bfhip_engine_algorithmic_bytes(bfhip_engine *e, double bytes[2]) {
    int r = ensure_ready(e);
    if (r != BFHIP_OK) return r;
    if (!bytes) return fail(BFHIP_EINVAL, "algorithmic_bytes: null array");
    bytes[0] = e->alg_bytes_total;
    bytes[1] = e->alg_bytes_mac;
    return BFHIP_OK;
}

int bfhip_engine_read_output_spectrum(bfhip_engine *e, int ch, void *dst) {
    if (!e || !e->finalized || ch < 0 || ch >= e->n_ch[1] || !dst) return fail(BFHIP_EINVAL, "read_output_spectrum: bad argument");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    const size_t row = (size_t)e->L * e->csize();
    std::vector<unsigned char> tmp(row);
    memset(dst, 0, row);
    for (int c = 0; c < (e->zp_is_sum ? 1 : e->n_chunks); c++) {
        const unsigned char *zp = (const unsigned char *)(((e->pipelined || e->defer_out) && e->d_Zp2 && ((e->blocks_done - 1) & 1)) ? e->d_Zp2 : e->d_Zp);
        if (e->pipe2) {
            const int zi = (int)((e->blocks_done - 1) % 3ull);
            zp = (const unsigned char *)(zi == 0 ? e->d_Zp : (zi == 1 ? e->d_Zp2 : e->d_Zp3));
        }
        HIPCHK(hipMemcpy(tmp.data(), zp + ((size_t)c * e->n_out_padded + ch) * row, row, hipMemcpyDeviceToHost));
        const size_t n = (size_t)2 * e->L;
        if (e->rs == 4) for (size_t i = 0; i < n; i++) ((float *)dst)[i] += ((float *)tmp.data())[i];
        else for (size_t i = 0; i < n; i++) ((double *)dst)[i] += ((double *)tmp.data())[i];
    }
    return BFHIP_OK;
}

int bfhip_engine_read_ring_slot(bfhip_engine *e, int ch, int slot, void *dst) {
    if (!e || !e->finalized || ch < 0 || ch >= e->n_ch[0] || slot < 0 || slot >= e->R || !dst) return fail(BFHIP_EINVAL, "read_ring_slot: bad argument");
    { const int ro = check_owner(e); if (ro != BFHIP_OK) return ro; }
    HIPCHK(hipSetDevice(e->device));
    { int _r = sync_all(e); if (_r != BFHIP_OK) return _r; }
    const size_t row = (size_t)e->L * e->csize();
    HIPCHK(hipMemcpy(dst, (unsigned char *)e->d_ring + ((size_t)ch * e->R + slot) * row, row, hipMemcpyDeviceToHost));
    return BFHIP_OK;
}

}  // extern "C"
