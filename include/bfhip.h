/*
 * bfhip.h -- C ABI of the MI355X-native partitioned-FFT convolution engine that replaces
 * BruteFIR's filter path (convolver.h + the per-block body of filter_process()).
 *
 * Plain C: pointers, ints, doubles.  No HIP, torch or C++ types cross this boundary.
 * Citations are file:line in the reference tree (chipfunk/brutefir, v1.0o).
 *
 * Two levels are exported by libbfhip.so:
 *
 *  (1) this file: the FUSED BLOCK API.  One call = everything the reference does between
 *      `timestamp(&t3)` and `t[7] += t4 - t3` of one filter_process() iteration
 *      (bfrun.c:1493-2008): raw -> real, forward FFT, input mix, per-filter partitioned
 *      multiply-accumulate over the ring of past spectra, output mix, inverse FFT,
 *      real -> raw with overflow accounting.  All state (rings, coefficient partitions,
 *      overflow structs) lives in HBM.  The patched filter_process() keeps its pipes,
 *      mutex snapshot and barriers and calls bfhip_engine_block().
 *
 *  (2) bfhip_convolver.h: the 22 link-time symbols of convolver.h with the reference's
 *      host-memory semantics.  The per-block ones run on the device (the unfused fallback of a
 *      host whose modules hook the per-buffer events); the ones the host calls BEFORE it forks
 *      its processes or from module processes (bfconf.c, delay.c, bflogic_eq) are pure host
 *      code: HIP state does not survive fork().
 *
 * Error convention follows the host: functions return 0 / a non-negative index
 * on success and a negative BFHIP_E* code on failure; bfhip_last_error() gives the text.
 * Nothing here calls exit(); the patched caller decides (bf_exit(BF_EXIT_OTHER)).
 * If no HIP device / code object is usable every block entry point fails with BFHIP_ENODEV:
 * there is no CPU fallback for the per-block work.
 */
#ifndef BFHIP_H
#define BFHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BFHIP_OK        0
#define BFHIP_EINVAL   -1   /* bad argument / unsupported configuration */
#define BFHIP_ENODEV   -2   /* no usable HIP device */
#define BFHIP_ENOMEM   -3   /* device allocation failed */
#define BFHIP_EHIP     -4   /* a HIP call or kernel failed */
#define BFHIP_ESTATE   -5   /* call not valid in this state (e.g. before finalize) */

/* per-block status bits (bfhip_engine_block / bfhip_engine_sync return them, >= 0) */
#define BFHIP_ST_NONFINITE 1  /* NaN/Inf reached an output: reference abort()s,
                                 real2raw.h:24-31 and bfrun.c:1903-1911 */
#define BFHIP_ST_SAFETY    2  /* safety_limit exceeded: reference bf_exit()s, real2raw.h:32-41 */

#define BFHIP_IN  0           /* BF_IN,  bfmod.h:30 */
#define BFHIP_OUT 1           /* BF_OUT, bfmod.h:31 */

/* struct bfoverflow, bfmod.h:99-104 -- identical field order and types, so the patched
   host can pass &icomm->overflow[ch] straight through */
typedef struct bfhip_overflow {
    unsigned int n_overflows;
    int32_t intlargest;
    double largest;
    double max;
} bfhip_overflow;

/* struct sample_format (dai.h:21-28, minus the unused `format` tag) followed by the two
   struct buffer_format fields (dai.h:30-34) */
typedef struct bfhip_format {
    int isfloat;
    int swap;
    int bytes;            /* bytes per sample in the raw buffer                   */
    int sbytes;           /* significant bytes (S24_4LE: bytes 4, sbytes 3)       */
    double scale;         /* sf.scale: 1/2^(8*sbytes-1) for ints, 1.0 for floats  */
    int sample_spacing;   /* in samples                                           */
    int byte_offset;      /* of the channel's first sample in the raw buffer      */
} bfhip_format;

typedef struct bfhip_engine bfhip_engine;

/* library / device */
int bfhip_device_count(void);
const char *bfhip_last_error(void);
const char *bfhip_version(void);

/* ---- construction: what bfconf_init() + the set-up part of filter_process() do ------ */

/* convolver_init(wisdom, length, realsize) (fftw_convolver.c:784-851) + the buffer set-up of
   filter_process() (bfrun.c:1227-1304, all zeroed as at :1388).  length = partition size L
   (power of two, 4..1048576: up to 8192 the transforms run in LDS, above as multi-kernel
   sequences over global memory; the reference's stock `filter_length: 65536` is covered),
   n_blocks = N partitions per filter, realsize 4 or 8.
   The device is initialised here: call it in the forked filter process, never in the parent
   (nothing the parent needs -- convolver_init, convolver_coeffs2cbuf, ... -- touches
   the device).  An engine belongs to the process that created it: in a fork()ed child every
   device entry point returns BFHIP_ESTATE (before any HIP call, which would hang in the inherited
   runtime), and bfhip_engine_destroy only frees the child's copy of the handle. */
bfhip_engine *bfhip_engine_create(int device, int length, int n_blocks, int realsize,
                                  int n_in, int n_out);
void bfhip_engine_destroy(bfhip_engine *e);

/* dai_buffer_format[io]->bf[channel] (dai.c:537-576) */
int bfhip_engine_set_format(bfhip_engine *e, int io, int channel, const bfhip_format *bf);
/* N:1 virtual -> physical channel mapping (`mapping:` in an input/output device section,
   bfconf->virt2phys / n_virtperphys).  Any mapping is legal (bench4_config: 0,1,0,1,0,1); the
   members of a shared physical output are mixed in ascending virtual order.  After this call
   bfhip_engine_set_format and bfhip_engine_enable_dither address PHYSICAL channels.  For
   channels that share a physical one, integer delay and mute happen inside the block like in
   filter_process() (bfrun.c:1509-1531, 1938-2003; delay.c) -- set them with the calls below at
   any time (what bfaccess->set_delay / toggle_mute write into icomm).  For 1:1 channels delay and
   mute are dai.c's business on the raw buffers and these settings are ignored. */
int bfhip_engine_map_channels(bfhip_engine *e, int io, int n_phys, const int virt2phys[]);
int bfhip_engine_set_delay(bfhip_engine *e, int io, int virt_channel, int delay_samples);
int bfhip_engine_set_maxdelay(bfhip_engine *e, int io, int virt_channel, int maxdelay);  /* <0: fixed */
int bfhip_engine_set_mute(bfhip_engine *e, int io, int virt_channel, int muted);
/* sub-sample delay: `sdf_length` (half length of the Kaiser-windowed sinc filters; > 0 enables)
   and per-channel `subdelay:` in hundredths of a sample, range (-100, 100); -100
   (BF_UNDEFINED_SUBDELAY) = the channel has no filter.  Which channels have one is fixed at
   finalize; the value may change at run time (bfaccess->set_subdelay).  A filtered channel is
   delayed by sdf_length + subdelay/100 samples (delay.c:416-505); a channel WITHOUT a filter that
   shares a physical channel is delayed by sdf_length whole samples (bfrun.c:1152-1162). */
#define BFHIP_UNDEFINED_SUBDELAY (-100)
int bfhip_engine_enable_subdelay(bfhip_engine *e, int sdf_length, double kaiser_beta);
int bfhip_engine_set_subdelay(bfhip_engine *e, int io, int virt_channel, int subdelay);
/* bfconf->safety_limit (linear, 0 = off), bfconf.c "safety_limit" setting */
int bfhip_engine_set_safety_limit(bfhip_engine *e, double limit);
/* `powersave:` (bfconf.c:1549-1561, bfrun.c:721-771, 1541-1553).  0 = off (default); >= 1.0 =
   `powersave: true`: a 2L input window that is exactly zero is not transformed and inputs that
   have been silent for a whole filter length are skipped by the MAC -- no sample changes;
   10^(dB/20) < 1.0 = `powersave: <dB>`: windows below that level (full scale = 1.0) count as
   silence and are made zero, like the reference does.  Before finalize. */
int bfhip_engine_set_powersave(bfhip_engine *e, double analog_powersave);
/* (physical) outputs to dither + dither_init() parameters (dither.c:75-139, bfconf.c:3170-3230);
   outputs that share a physical channel or carry a sub-sample filter are dithered after their
   mix / filter, like convolver_cbuf2raw on the mixbuf (bfrun.c:1992-1997) */
int bfhip_engine_enable_dither(bfhip_engine *e, const int out_channels[], int n,
                               int sample_rate, int max_size);

/* Optional, before the first add_coeff*: the total size of all coefficient sets to come, in bytes
   of device memory (sum over sets of n_blocks * 2L * realsize -- what bfconf knows as
   n_coeffs x n_blocks x convolver_cbufsize()).  The sets are then carved out of ONE allocation.
   The MAC streams every set of the filter network in every block; how well that runs depends on
   how large the physically contiguous pieces behind them are (address translation): one slab is
   worth ~6 % at the headline configuration over one allocation per set.  Without the call the
   engine allocates 2 GiB slabs as it goes, which gets most of it.  (`double`: the figure easily
   exceeds 2^31.) */
int bfhip_engine_reserve_coeffs(bfhip_engine *e, double total_bytes);
/* load_coeff() for a time-domain coefficient set (bfconf.c:1867-2030): split into
   n_blocks partitions of L taps (n_blocks <= 0: as many as the taps need), each through
   convolver_coeffs2cbuf (fftw_convolver.c:526-573).  taps are `realsize`-wide reals in
   host (or, _dev, device) memory.  Returns the coefficient index (>= 0).
   _dev: the call waits for the device before it reads taps_dev (whatever stream produced them) and
   for its own kernel before it returns -- the buffer is the caller's again afterwards; NaN / Inf
   among device taps is reported by bfhip_engine_finalize. */
int bfhip_engine_add_coeff(bfhip_engine *e, const void *taps, int n_taps, double scale,
                           int n_blocks);
int bfhip_engine_add_coeff_dev(bfhip_engine *e, const void *taps_dev, int n_taps,
                               double scale, int n_blocks);
/* load_coeff() for `format: "processed"` sets and shared-memory coefficient sets
   (bfconf.c:1924-1971): n_blocks cbufs of 2L reals each, contiguous, in the reference's
   internal frequency-domain layout (fftw_convfuns.h:25-43: groups of 4 re / 4 im, Nyquist in
   slot 4, already divided by n_fft).  Checked like convolver_verify_cbuf. */
int bfhip_engine_add_coeff_processed(bfhip_engine *e, const void *cbufs, int n_blocks);
/* the inverse: a loaded set back in that layout (bfaccess->coeffs_data, debug dumps, writing
   "processed" files); cbufs must hold n_blocks * 2L reals.  Returns n_blocks. */
int bfhip_engine_read_coeff_processed(bfhip_engine *e, int coeff, void *cbufs);
/* The same for what an UNMODIFIED bfconf holds: bfconf->coeffs_data[c][0..n_blocks) are separate
   allocations (bfconf.c:1994-2009: one convolver_coeffs2cbuf() result, or one slice of a
   shared-memory segment, per block).  BFHIP_COEFF_WATCH: the set may be rewritten at run time by another
   process (bflogic_eq renders into coeffs_data[c][i] in ITS process through
   bfaccess->convolver_coeffs2cbuf, rendereq.h:87-91); the engine remembers the host address of
   every block and, at the start of each block, re-uploads the ones whose change notice moved --
   see bfhip_coeff_mark_dirty() below.  The pointers must stay valid for the engine's life.
   flags: BFHIP_COEFF_WATCH as above; BFHIP_COEFF_LAZY: nothing goes to the device yet -- the
   host blocks are loaded the first time a filter THIS engine runs refers to the set (an engine that
   runs a shard of the configuration, bfhip_engine_set_filter_active, then holds the sets of its own
   filters only; a run-time switch to a set not seen before loads it at that block). */
#define BFHIP_COEFF_WATCH 1
#define BFHIP_COEFF_LAZY 2
int bfhip_engine_add_coeff_processed_blocks(bfhip_engine *e, void *const cbufs[], int n_blocks,
                                            int flags);
/* 1: the set's partitions are in device memory (always, unless it was registered BFHIP_COEFF_LAZY
   and no filter this engine runs has referred to it yet) */
int bfhip_engine_coeff_is_resident(const bfhip_engine *e, int coeff);
/* re-upload one partition of a loaded set from a cbuf in the reference's layout now (cbuf NULL:
   from the address given to add_coeff_processed_blocks) */
int bfhip_engine_refresh_coeff_processed(bfhip_engine *e, int coeff, int block, const void *cbuf);
/* what every block entry point does first when a watched set exists: returns how many
   partitions were re-uploaded (>= 0) or an error */
int bfhip_engine_poll_coeff_changes(bfhip_engine *e);
/* Cross-process change notices.  convolver_init() (parent, before the fork) creates a small
   MAP_SHARED table; every process forked afterwards shares it.  bfhip_coeff_mark_dirty(cbuf)
   bumps the generation of the cbuf at that ADDRESS (the same in all processes: coefficient
   memory is allocated before the fork); this library's convolver_runtime_coeffs2cbuf() calls it
   on its `dest`, so an unmodified bflogic_eq is covered.  A module that writes coefficient
   memory by other means calls it itself.  Pure host code, no HIP. */
void bfhip_coeff_mark_dirty(const void *cbuf);
unsigned long long bfhip_coeff_dirty_sequence(void);
/* run-time replacement of one partition = convolver_runtime_coeffs2cbuf
   (fftw_convolver.c:575-596) as used by bflogic_eq (rendereq.h:87-91): L reals */
int bfhip_engine_update_coeff_block(bfhip_engine *e, int coeff, int block, const void *taps);

/* struct bffilter (bfmod.h:113-121) + its initial struct bffilter_control
   (bfmod.h:128-133, bfconf->initfctrl).  from_filters must already have been added
   (the order bfconf.c:2933-2964 establishes).  Returns the filter index. */
int bfhip_engine_add_filter(bfhip_engine *e,
                            int n_in_ch, const int in_ch[], const double in_scale[],
                            int n_in_f, const int in_f[], const double in_fscale[],
                            int n_out_ch, const int out_ch[], const double out_scale[],
                            int coeff, int delayblocks, int crossfade);

/* ---- one engine per filter process: a SHARD of the configuration -------------------------
   The reference splits its filters over n_processes forked filter processes (bfconf.c:2227-2318;
   every output is mixed inside one process, connected filters stay together, bfconf.c:2893-2931;
   bfrun.c:2312-2328 forks them).  With this library each of those processes creates its own engine
   (on its own GPU: device = process_index % bfhip_device_count()), describes the WHOLE
   configuration to it -- all channels, all filters -- and marks the filters the other processes
   run inactive.  The engine plans for the whole configuration (output groups, entry order, chunk
   boundaries: every output is summed in exactly the order a single engine would use, so the
   outputs are bit-identical to the one-process run), launches the work of its own filters only,
   transforms every input itself (no exchange between the processes; of the two
   synch_filter_processes barriers of bfrun.c:1563, 1873 the host keeps one per period, which paces
   the processes on their shared wake pipe), and converts and
   writes only the outputs it owns: in the raw output buffer the processes share, and in the
   overflow array, everything else is left untouched -- by the device entry points (the output
   pass skips foreign channels) and by the host ones (bfhip_engine_block, bfhip_engine_rt_wait copy
   this engine's samples only).
   An output is owned by the engine whose active filters feed it; outputs no filter feeds are owned
   unless bfhip_engine_set_output_active says otherwise (the host gives them to the process that
   mixes the other members of their physical channel, else to process 0).
   Virtual outputs that share a physical channel must be owned together.  All before finalize. */
int bfhip_engine_set_filter_active(bfhip_engine *e, int filter, int active);
int bfhip_engine_set_output_active(bfhip_engine *e, int ch, int active);
int bfhip_engine_output_is_active(const bfhip_engine *e, int ch);
/* the host's own number for a filter (struct bffilter.intname): entries are ordered by it, so that
   the summation order does not depend on the order the filters were added in */
int bfhip_engine_set_filter_name(bfhip_engine *e, int filter, int name);

/* build the device plan; no add_* after this */
int bfhip_engine_finalize(bfhip_engine *e);

/* ---- run-time control: the fctrl snapshot of bfrun.c:1460-1484 ----------------------- */
int bfhip_engine_set_coeff(bfhip_engine *e, int filter, int coeff);          /* fctrl.coeff */
int bfhip_engine_set_delayblocks(bfhip_engine *e, int filter, int blocks);   /* .delayblocks */
int bfhip_engine_set_scale(bfhip_engine *e, int filter, int io, int index, double scale);
int bfhip_engine_set_fscale(bfhip_engine *e, int filter, int index, double scale);

/* ---- per block ------------------------------------------------------------------------ */

/* Host buffers (what filter_process() holds: inbuf[curbuf], outbuf[curbuf]).  Copies in,
   runs the block, copies out, waits.  Returns status bits (>= 0) or an error (< 0).
   overflow[] (may be NULL): n_out structs, read-modify-written like bfrun.c:1929-1936. */
int bfhip_engine_block(bfhip_engine *e, const void *rawin, void *rawout,
                       bfhip_overflow overflow[]);

/* Device-resident raw buffers, asynchronous (ordering contract: below).  How the kernels of a
   block are scheduled is decided at finalize (bfhip_engine_block_mode): small crossbars
   (coefficient stream under ~100 us per block) overlap the transforms of neighbouring blocks with
   the MAC on engine-owned side streams, the way the reference overlaps its input, filter and
   output processes; large ones (the headline config) fuse the inverse transforms of a block with
   the forward transforms of the next into one launch on one stream.  BFHIP_OVERLAP=0/1 in the
   environment moves the automatic choice; bfhip_engine_set_overlap decides it. */
int bfhip_engine_block_dev(bfhip_engine *e, const void *rawin_dev, void *rawout_dev);
/* TWO consecutive blocks with ONE pass over the coefficients (exploratory; bench.py --pairs reports it
   as an informative line, never as the headline).  The MAC of a large crossbar is nothing but the
   coefficient stream -- config C reads 8 GiB per block -- and a host that has two periods in hand
   (the blocking-I/O topology keeps two in flight anyway, at one period of extra I/O delay) can
   have both multiplied while each coefficient tile is in registers once.  Needs
   bfhip_engine_enable_pairs(e, 1) BEFORE finalize (one spare ring slot, a second partial-sum
   buffer).  Plans that are not a plain uniform crossbar, and the first N blocks of a run, go through
   two single blocks instead: rawout0 / rawout1 hold the same bits either way
   (tests/test_gpu_pairs.py).  Same ordering contract as bfhip_engine_block_dev, both outputs complete
   in stream order behind the call.  bfhip_engine_pair_launches: how many calls took the paired path. */
int bfhip_engine_enable_pairs(bfhip_engine *e, int on);
int bfhip_engine_block_pair_dev(bfhip_engine *e, const void *rawin0_dev, void *rawout0_dev,
                                const void *rawin1_dev, void *rawout1_dev);
unsigned long long bfhip_engine_pair_launches(const bfhip_engine *e);
/* ORDERING CONTRACT of bfhip_engine_block_dev.  The engine launches on streams of its own (K1 of
   a pipelined block even on a side stream that does NOT follow the stream given to
   bfhip_engine_set_stream), so stream order alone protects nothing:
     - rawin_dev must be COMPLETE on the device when the call is made (the producer has been
       synchronised), and must not be rewritten before bfhip_engine_sync();
     - rawout_dev may be read only after bfhip_engine_sync(), and must stay valid until then or
       until the NEXT block call has returned: for large crossbars on one stream the inverse
       transforms of a block share one launch with the forward transforms of the next block
       ("deferred output"; bfhip_engine_sync flushes what is owed; bfhip_engine_set_overlap(e, 0)
       or BFHIP_DEFER=0 turn it off); small crossbars on the wave FFT owe their output for TWO calls
       (ping-pong schedule, BFHIP_MODE_PINGPONG; BFHIP_PIPE2=0 turns it off).
   A caller that produces the input or consumes the output asynchronously uses the variant
   below instead: in_ready_event (hipEvent_t, may be NULL) is an event the caller recorded behind
   its producer of rawin_dev -- the input transform waits for it, nothing else does, so the
   overlap of neighbouring blocks is kept; out_done_event (hipEvent_t, may be NULL) is recorded by
   the engine behind the last kernel of THIS block: after it rawout_dev is complete and
   rawin_dev may be reused.
   WHEN it is recorded: by the call that launches the block's output pass -- the call itself, or,
   where outputs are owed, the call bfhip_engine_output_lag() blocks later (1: deferred output,
   2: ping-pong), or bfhip_engine_flush / bfhip_engine_sync, whichever comes first.  Until then the
   event still carries whatever was recorded on it before: a caller must not wait on the event
   of block t before call t + lag (or flush, or sync) has returned. */
int bfhip_engine_block_dev_ev(bfhip_engine *e, const void *rawin_dev, void *rawout_dev,
                              void *in_ready_event, void *out_done_event);
/* how many later block calls pass before a block's output pass is launched (0, 1 or 2; fixed at
   finalize, see bfhip_engine_block_mode) */
int bfhip_engine_output_lag(const bfhip_engine *e);
/* launch the output passes still owed now (and record their out_done events); does not wait.
   The schedule picks up again with the next block call. */
int bfhip_engine_flush(bfhip_engine *e);
/* wait for the stream; returns accumulated status bits (and clears them) or an error */
int bfhip_engine_sync(bfhip_engine *e);

/* The three phases of a block, for a host that shards the crossbar over several GPUs and
   puts a collective between them ("multi-GPU"):
     inputs : raw -> real -> FFT -> ring slot, for this engine's input channels
     mac    : Z[o] = sum over this engine's filters (partial sums if inputs are sharded);
              z_dev: n_out * L complex numbers of the working precision, channel-major
     outputs: inverse FFT + real -> raw of channels [first, first+count) from z_dev, whose
              channel 0 is output `first` */
int bfhip_engine_inputs_dev(bfhip_engine *e, const void *rawin_dev);
int bfhip_engine_mac_dev(bfhip_engine *e, void *z_dev);
int bfhip_engine_outputs_dev(bfhip_engine *e, const void *z_dev, int first, int count,
                             void *rawout_dev);
/* outputs of an EARLIER block (spectra in z_dev) and inputs of the CURRENT block in one launch:
   for a host that keeps several blocks in flight (the mix-down of block t travels while block
   t+1 is computed) the two are independent, and one launch instead of two shortens the step */
int bfhip_engine_outputs_inputs_dev(bfhip_engine *e, const void *z_dev, int first, int count,
                                    void *rawout_dev, const void *rawin_dev);
/* advance blockcounter / curbuf (bfrun.c:2031-2034); block/block_dev do it themselves */
int bfhip_engine_advance(bfhip_engine *e);

/* ---- real-time mode: callback I/O -------------------------------------------------------
 * The reference's callback I/O (bfio_jack.c:132-200 -> dai.c:111 process_callback ->
 * bf_callback_ready, bfrun.c:2086-2131) hands filter_process() one period in a shared-memory
 * buffer and blocks until the filtered period is back; what counts there is the round trip of
 * ONE block, not throughput.  In this mode the engine owns a pinned host double buffer, keeps
 * the whole launch sequence of a block (upload, K1, per-filter kernels, K2, K3, dither,
 * download) in a HIP graph that is replayed with one call, carries the block counter in device
 * memory and signals completion through pinned memory.  Control changes (set_coeff, set_scale,
 * ...) keep working: the plan is rebuilt and the next period but one is replayed from a fresh
 * graph; periods that cannot be replayed (the one-block cross-fade, N:1 channels, sub-sample
 * delays) are launched normally.  Results are bit-identical to bfhip_engine_block(). */
#define BFHIP_RT_SPIN     1   /* wait by watching the pinned completion word (lowest latency, burns the core) */
#define BFHIP_RT_NO_GRAPH 2   /* never replay: plain launches from the pinned buffers */
#define BFHIP_RT_COPY_ENGINE 4 /* stage the period with memcpy nodes instead of copy kernels */
#define BFHIP_RT_OVERLAP  8   /* throughput with host buffers: upload of period t+1 and download of
                                 period t-1 ride the copy engines while period t computes (no graph
                                 replay); keep two periods in flight with rt_submit / rt_wait */
int bfhip_engine_rt_begin(bfhip_engine *e, int flags);
int bfhip_engine_rt_end(bfhip_engine *e);
/* the pinned buffers (io 0 = in, 1 = out; index 0/1): period k uses index k & 1.  A host that
   fills / drains them in place passes NULL to rt_submit / rt_wait and saves both memcpys. */
void *bfhip_engine_rt_buffer(bfhip_engine *e, int io, int index);
/* start one period; returns at once.  At most two periods may be in flight. */
int bfhip_engine_rt_submit(bfhip_engine *e, const void *rawin);
/* wait for the oldest period in flight; returns its status bits (>= 0) like bfhip_engine_block.
   overflow[] (n_out structs, may be NULL) receives the device's running overflow state. */
int bfhip_engine_rt_wait(bfhip_engine *e, void *rawout, bfhip_overflow overflow[]);
/* rt_submit + rt_wait */
int bfhip_engine_rt_block(bfhip_engine *e, const void *rawin, void *rawout, bfhip_overflow overflow[]);
/* how many periods were replayed from a graph / launched directly, and how many captures ran */
int bfhip_engine_rt_stats(const bfhip_engine *e, unsigned long long *graph_blocks,
                          unsigned long long *direct_blocks, unsigned long long *captures);


/* The reference skips partitions whose input block does not exist yet (procblocks guard,
   bfrun.c:1745), so the first n_blocks blocks after start cost less than the steady state.  This
   call declares the (zero-initialised) rings to hold n_blocks blocks of silence: every block
   from the first one on does the full work, with the same output (silence contributes
   nothing).  For measurements; only valid before the first block.  The engine's block counter
   starts at its ring depth instead of 0 afterwards. */
int bfhip_engine_prewarm(bfhip_engine *e);

/* Several engines that make up one filter (the segments of include/bfhip_nupc.h) can report into
   one status word in device memory: the NaN/Inf and safety-limit bits (real2raw.h:24-41) are
   OR-ed into *status_dev instead of the engine's own word, and reading / clearing it is the
   caller's business (bfhip_engine_sync then reads that word).  NULL restores the own word. */
int bfhip_engine_set_status_dev(bfhip_engine *e, int *status_dev);

/* before finalize: -1 = decide from the plan (default), 0 = the three kernels of a block on one
   stream, strictly in order, output complete in stream order behind the call; 1 = on three
   engine-owned streams (see bfhip_engine_block_dev) */
int bfhip_engine_set_overlap(bfhip_engine *e, int mode);
/* hipStream_t to run on (default: a stream owned by the engine) */
int bfhip_engine_set_stream(bfhip_engine *e, void *hip_stream);

int bfhip_engine_get_overflow(bfhip_engine *e, int out_channel, bfhip_overflow *of);
int bfhip_engine_reset_overflow(bfhip_engine *e);     /* bf_reset_peak(), bfrun.c */
/* the engine's block counter (bfrun.c:2034).  It wraps by a multiple of the ring depths long
   before 2^32 (so that `counter mod depth` never jumps, also for depths that are not powers of
   two): do not use it to count blocks over long runs */
unsigned int bfhip_engine_blockcounter(const bfhip_engine *e);
/* how bfhip_engine_block_dev schedules the kernels of a block (decided at finalize): */
#define BFHIP_MODE_SEQUENTIAL 0   /* K1, MAC, K3 of a block in order on one stream                 */
#define BFHIP_MODE_PIPELINED  1   /* small MACs: K1 of t+1 and K3 of t-1 on side streams beside MAC t */
#define BFHIP_MODE_DEFERRED   2   /* large MACs: [K3 of t-1 | K1 of t] in one launch, then MAC t     */
#define BFHIP_MODE_PINGPONG   3   /* small MACs, wave FFT: [K3 of t-2 | K1 of t] on a side stream beside
                                     MAC t-1, MAC t behind it; outputs are owed for two calls         */
int bfhip_engine_block_mode(const bfhip_engine *e);
/* 1 if the input / output transforms run on the wave-level FFT (fft_wave.h) */
int bfhip_engine_uses_wave_fft(const bfhip_engine *e);
/* 1 if the MAC reads a stream-ordered copy of the coefficients (uniform crossbar plans: every
   workgroup one sequential slice; costs a second copy of the coefficient memory) */
int bfhip_engine_uses_stream_layout(const bfhip_engine *e);
/* 1: the plan is one-to-one (every output fed by one single-term filter: massive_config, BASELINE
   configs[3]) and the MAC runs as mac_diag_kernel -- a workgroup per (part, output) walking whole
   spectra -- instead of the crossbar kernel */
int bfhip_engine_uses_diag_mac(const bfhip_engine *e);
/* depth of the input spectrum rings (n_blocks, plus one spare slot when the block is pipelined) */
int bfhip_engine_ring_depth(const bfhip_engine *e);

/* ---- measurement ----------------------------------------------------------------------- */

/* HIP-event timing of the three kernels of a block on the engine's stream.  ms[0..2] =
   mean duration of the input-FFT, MAC and output-IFFT launches since the last reset,
   ms[3] = MAC launches averaged.  Reading synchronises the stream.  on = n > 1 times every n-th
   block only (six event records per timed block cost the stream ~20 us).  The phase calls
   (inputs_dev / mac_dev / outputs_dev) are timed the same way; the fused
   bfhip_engine_outputs_inputs_dev launch is reported in the input slot ms[0]. */
int bfhip_engine_enable_timing(bfhip_engine *e, int on);
int bfhip_engine_get_timing(bfhip_engine *e, double ms[4]);
/* The same events in the columns of the reference's `benchmark: true` table (bfrun.c:2035-2078,
   brutefir.html:1254-1261): ms[0] raw2real, [1] time2freq, [2] mixscale1, [3] convolve, [4] mixscale2,
   [5] freq2time, [6] real2raw, [7] total -- mean device milliseconds per block over the whole blocks
   timed since the last call (which it resets); returns how many those were (0: none, ms all zero).
   A stage the GPU path fuses into a neighbour has no time of its own and reads 0:
     raw2real   = 0: sample conversion happens inside the input-transform kernel  -> time2freq
     time2freq  = the input-transform launch (K1)
     mixscale1  = the per-filter kernels in front of the MAC: N-way input mixes, filter-to-filter
                  cascades, cross-fades (0 for plain one-input filters: their scale is a factor in K2)
     convolve   = the crossbar multiply-accumulate launch (K2), input and output scales included
     mixscale2  = 0: the output mix is the accumulation itself                     -> convolve
     freq2time  = the output launch (K3: chunk sum, inverse transform, requantiser of plain outputs)
     real2raw   = what runs as launches of its own behind K3: HP-TPDF dither chains, the N:1
                  time-domain mix, sub-sample delay filters (0 without them)
     total      = the sum of the above (device time; the host's own t[7] stays wall-clock)
   Under the deferred-output / ping-pong schedules of bfhip_engine_block_dev the fused
   [K3 of t-1 | K1 of t] launch is counted under time2freq.  Real-time mode: only plain launches are
   timed (begin with BFHIP_RT_NO_GRAPH); graph-replayed periods are not.  Needs enable_timing. */
int bfhip_engine_stage_times(bfhip_engine *e, double ms[8]);
/* algorithmic bytes of one block per SURVEY 8(d): bytes[0] total, [1] MAC kernel only
   (coefficient partitions + ring slots read + output spectra written) */
int bfhip_engine_algorithmic_bytes(bfhip_engine *e, double bytes[2]);

/* debug / parity taps: copy device state to host memory */
int bfhip_engine_read_output_spectrum(bfhip_engine *e, int out_channel, void *dst_complex);
int bfhip_engine_read_ring_slot(bfhip_engine *e, int in_channel, int slot, void *dst_complex);

/* ---- self test (no device) ------------------------------------------------------------------
 * The integer delay of channels that share a physical channel (delay.c:78-340) is a small machine
 * of buffer moves whose schedule depends on the delay history; the engine runs that machine on the
 * host and has the device execute the moves it emits.  These three calls run the same machine
 * with its buffers in host memory, so that its output can be checked against delay.c without a
 * GPU (tests/test_abi.py).  update returns the number of moves, or a negative error. */
/* The wave-level FFT (csrc/fft_wave.h) reads its twiddles from a host-made table whose second
   part is ordered by thread and register: [0, 2L) exp(-2 pi i m / 2L), then 18 registers x L/16
   threads.  This call returns the table (or, with out == NULL, its size in bytes) so that the
   thread -> butterfly maps it encodes can be checked against a model of the algorithm without a
   GPU (tests/test_wave_fft_model.py). */
int bfhip_selftest_wave_twiddles(int log2l, int realsize, void *out, int out_bytes);
/* tests: the nth device / pinned allocation the engine makes from now on fails with "out of
   memory" (0: none).  tests/test_gpu_alloc_faults.py walks n over a whole engine life with it:
   every allocation failure must come back as an error code, and destroy must still clean up. */
int bfhip_selftest_fail_alloc(int nth);      /* returns what was left of the previous countdown */
typedef struct bfhip_selftest_delay bfhip_selftest_delay;
bfhip_selftest_delay *bfhip_selftest_delay_new(int fragment, int initdelay, int maxdelay, int sample_size);
int bfhip_selftest_delay_update(bfhip_selftest_delay *d, void *buf /* fragment * sample_size bytes */, int delay);
void bfhip_selftest_delay_free(bfhip_selftest_delay *d);

#ifdef __cplusplus
}
#endif
#endif
