#!/usr/bin/env python3
"""(Re)generate patches/bfrun-bfhip.diff: the ONE host change the drop-in needs.

Works on a temporary copy of the reference's bfrun.c (the reference tree is read-only and never
enters this repo): inserts the BF_HAVE_BFHIP blocks below at four anchor lines and writes the
unified diff.  tests/test_bfrun_patch.py applies the committed diff to a fresh temporary copy
and compiles it with `gcc -fsyntax-only -DBF_HAVE_BFHIP` against the reference's headers and
include/bfhip.h.

    python tools/make_bfrun_patch.py [/root/reference]
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"

HELPERS = r'''
#ifdef BF_HAVE_BFHIP
/*
 * MI355X backend (libbfhip.so): the body of one filter_process() iteration -- everything
 * between timestamp(&t3) and timestamp(&t4) -- runs on the GPU through ONE call.  Pipes,
 * the icomm snapshot, block_start/coeff_final events, the filter-process barriers, the output
 * signalling and the benchmark print stay as they are.
 *
 * n_processes > 1: every forked filter process drives its own engine on its own GPU
 * (device = process_index % bfhip_device_count()).  It describes the WHOLE configuration to the
 * engine and marks the filters of the other processes inactive (bfhip.h, "one engine per filter
 * process"): the engine transforms every input itself, runs its own filters, and writes only
 * the outputs its filters feed into the output buffer the processes share -- no data passes
 * between the processes, the two synch_filter_processes() calls of the unfused body are not
 * needed, and the output is bit-identical to the n_processes = 1 run.
 *
 * Not used when a module hooks the per-buffer events (input_timed .. output_timed need the host
 * buffers) or when the virtual members of a shared physical output are mixed in different
 * processes: such configurations keep the unfused path below, which links the same library's
 * convolver_* symbols.  BFHIP_DISABLE=1 in the environment forces the unfused path.
 */
#include "bfhip.h"

static bfhip_engine *bfhip_eng = NULL;
static bool_t bfhip_pipelined = false;
static int bfhip_inflight = 0;
static int bfhip_n_all = 0;                      /* filters of the whole configuration */
static int bfhip_index[BF_MAXFILTERS];           /* intname -> index in the engine */
static int bfhip_local[BF_MAXFILTERS];           /* intname -> index in this process, or -1 */
static struct bffilter *bfhip_filter[BF_MAXFILTERS];    /* intname -> its struct bffilter */

static void
bfhip_die(const char what[])
{
    fprintf(stderr, "bfhip: %s: %s\n", what, bfhip_last_error());
    bf_exit(BF_EXIT_OTHER);
}

/* the process whose filters feed a virtual output (bfconf puts all filters of one output, and of
   the members of one physical output, into one process, bfconf.c:2893-2931), or -1 */
static int
bfhip_output_feeder(int virtch)
{
    int k, i;

    for (k = 0; k < bfconf->n_processes; k++) {
        for (i = 0; i < bfconf->fproc[k].n_unique_channels[OUT]; i++) {
            if (bfconf->fproc[k].unique_channels[OUT][i] == virtch) {
                return k;
            }
        }
    }
    return -1;
}

/* which process mixes (and therefore, on the GPU, converts) a virtual output.  An output no
   filter feeds (bfconf only warns, bfconf.c:2708) goes with the fed members of its physical
   channel, so that one engine mixes the whole group; a physical channel nobody feeds at all
   belongs to process 0 */
static int
bfhip_output_owner(int virtch)
{
    int k, i, physch;

    if ((k = bfhip_output_feeder(virtch)) >= 0) {
        return k;
    }
    physch = bfconf->virt2phys[OUT][virtch];
    for (i = 0; i < bfconf->n_virtperphys[OUT][physch]; i++) {
        if ((k = bfhip_output_feeder(bfconf->phys2virt[OUT][physch][i])) >= 0) {
            return k;
        }
    }
    return 0;
}

static bool_t
bfhip_wanted(void)
{
    int physch, i;

    if (getenv("BFHIP_DISABLE") != NULL) {
        return false;
    }
    if (events.n_input_timed > 0 || events.n_input_freqd > 0 ||
        events.n_pre_convolve > 0 || events.n_post_convolve > 0 ||
        events.n_output_freqd > 0 || events.n_output_timed > 0)
    {
        return false;
    }
    /* the members of a shared physical output are mixed in the time domain by one engine
       (every process computes this from bfconf alone: all of them decide alike) */
    for (physch = 0; physch < bfconf->n_physical_channels[OUT]; physch++) {
        for (i = 1; i < bfconf->n_virtperphys[OUT][physch]; i++) {
            if (bfhip_output_owner(bfconf->phys2virt[OUT][physch][i]) !=
                bfhip_output_owner(bfconf->phys2virt[OUT][physch][0]))
            {
                return false;
            }
        }
    }
    return true;
}

/* Runs in the forked filter process (HIP state does not survive fork()): builds the engine
   from what bfconf_init() left in bfconf and icomm. */
static void
bfhip_setup(int process_index,
            bool_t callback_io)
{
    int n, i, k, c, idx, physch, n_dither, n_devices, flags;
    int dither_ch[BF_MAXCHANNELS];
    int from[BF_MAXFILTERS];
    double scales[BF_MAXCHANNELS], fscales[BF_MAXFILTERS];
    volatile struct bffilter_control *fc;
    struct bffilter *flt;
    struct buffer_format *bf;
    bfhip_format f;

    if ((n_devices = bfhip_device_count()) < 1) {
        bfhip_die("device_count");
    }
    bfhip_eng = bfhip_engine_create(process_index % n_devices,
                                    bfconf->filter_length, bfconf->n_blocks,
                                    bfconf->realsize, bfconf->n_channels[IN],
                                    bfconf->n_channels[OUT]);
    if (bfhip_eng == NULL) {
        bfhip_die("engine_create");
    }
    FOR_IN_AND_OUT {
        if (bfhip_engine_map_channels(bfhip_eng, IO,
                                      bfconf->n_physical_channels[IO],
                                      bfconf->virt2phys[IO]) < 0)
        {
            bfhip_die("map_channels");
        }
        for (physch = 0; physch < bfconf->n_physical_channels[IO]; physch++) {
            bf = &dai_buffer_format[IO]->bf[physch];
            f.isfloat = bf->sf.isfloat;
            f.swap = bf->sf.swap;
            f.bytes = bf->sf.bytes;
            f.sbytes = bf->sf.sbytes;
            f.scale = bf->sf.scale;
            f.sample_spacing = bf->sample_spacing;
            f.byte_offset = bf->byte_offset;
            if (bfhip_engine_set_format(bfhip_eng, IO, physch, &f) < 0) {
                bfhip_die("set_format");
            }
        }
        for (n = 0; n < bfconf->n_channels[IO]; n++) {
            bfhip_engine_set_maxdelay(bfhip_eng, IO, n, bfconf->maxdelay[IO][n]);
            bfhip_engine_set_delay(bfhip_eng, IO, n, icomm->delay[IO][n]);
            bfhip_engine_set_mute(bfhip_eng, IO, n,
                                  bit_isset_volatile(icomm->ismuted[IO], n));
        }
    }
    if (bfconf->use_subdelay[IN] || bfconf->use_subdelay[OUT]) {
        if (bfhip_engine_enable_subdelay(bfhip_eng, bfconf->sdf_length,
                                         bfconf->sdf_beta) < 0)
        {
            bfhip_die("enable_subdelay");
        }
        FOR_IN_AND_OUT {
            if (!bfconf->use_subdelay[IO]) {
                continue;
            }
            for (n = 0; n < bfconf->n_channels[IO]; n++) {
                bfhip_engine_set_subdelay(bfhip_eng, IO, n,
                                          bfconf->subdelay[IO][n]);
            }
        }
    }
    bfhip_engine_set_safety_limit(bfhip_eng, bfconf->safety_limit);
    if (bfconf->powersave) {
        bfhip_engine_set_powersave(bfhip_eng, bfconf->analog_powersave);
    }
    for (physch = n_dither = 0;
         physch < bfconf->n_physical_channels[OUT];
         physch++)
    {
        if (bfconf->dither_state[physch] != NULL) {
            dither_ch[n_dither++] = physch;
        }
    }
    if (n_dither > 0 &&
        bfhip_engine_enable_dither(bfhip_eng, dither_ch, n_dither,
                                   bfconf->sampling_rate,
                                   bfconf->max_dither_table_size) < 0)
    {
        bfhip_die("enable_dither");
    }
    /* Coefficient sets exactly as bfconf_init() prepared them (convolver_coeffs2cbuf in the
       parent, or "processed" / shared-memory data): one cbuf per block.  Sets in shared
       memory may be rewritten by a module process at run time (bflogic_eq): watched.  One of
       several filter processes registers them all but loads a set only when one of ITS
       filters first refers to it (its GPU then holds its share of the coefficients). */
    if (bfconf->n_processes == 1) {
        double total = 0;
        for (c = 0; c < bfconf->n_coeffs; c++) {
            total += (double)bfconf->coeffs[c].n_blocks * (double)convolver_cbufsize();
        }
        if (bfhip_engine_reserve_coeffs(bfhip_eng, total) < 0) {
            bfhip_die("reserve_coeffs");
        }
    }
    for (c = 0; c < bfconf->n_coeffs; c++) {
        flags = bfconf->coeffs[c].is_shared ? BFHIP_COEFF_WATCH : 0;
        if (bfconf->n_processes > 1) {
            flags |= BFHIP_COEFF_LAZY;
        }
        if (bfhip_engine_add_coeff_processed_blocks(bfhip_eng,
                                                    bfconf->coeffs_data[c],
                                                    bfconf->coeffs[c].n_blocks,
                                                    flags) != c)
        {
            bfhip_die("add_coeff_processed_blocks");
        }
    }
    /* every filter of the configuration, process by process (inside a process bfconf has put
       connected filters in evaluation order, bfconf.c:2933-2964); the engine orders its work by
       intname, so the listing order changes no sample */
    for (n = 0; n < BF_MAXFILTERS; n++) {
        bfhip_index[n] = bfhip_local[n] = -1;
        bfhip_filter[n] = NULL;
    }
    bfhip_n_all = 0;
    for (k = 0; k < bfconf->n_processes; k++) {
        for (n = 0; n < bfconf->fproc[k].n_filters; n++) {
            flt = &bfconf->fproc[k].filters[n];
            fc = &icomm->fctrl[flt->intname];
            for (i = 0; i < flt->n_filters[IN]; i++) {
                from[i] = bfhip_index[flt->filters[IN][i]];
                fscales[i] = fc->fscale[i];
            }
            for (i = 0; i < flt->n_channels[IN]; i++) {
                scales[i] = fc->scale[IN][i];
            }
            /* output scales go in a second array: reuse the tail of scales[] */
            for (i = 0; i < flt->n_channels[OUT]; i++) {
                scales[BF_MAXCHANNELS / 2 + i] = fc->scale[OUT][i];
            }
            idx = bfhip_engine_add_filter(bfhip_eng,
                                          flt->n_channels[IN],
                                          flt->channels[IN], scales,
                                          flt->n_filters[IN], from, fscales,
                                          flt->n_channels[OUT],
                                          flt->channels[OUT],
                                          &scales[BF_MAXCHANNELS / 2],
                                          fc->coeff, fc->delayblocks,
                                          flt->crossfade);
            if (idx != bfhip_n_all ||
                bfhip_engine_set_filter_name(bfhip_eng, idx, flt->intname) < 0 ||
                bfhip_engine_set_filter_active(bfhip_eng, idx,
                                               k == process_index) < 0)
            {
                bfhip_die("add_filter");
            }
            bfhip_index[flt->intname] = idx;
            bfhip_filter[flt->intname] = flt;
            if (k == process_index) {
                bfhip_local[flt->intname] = n;
            }
            bfhip_n_all++;
        }
    }
    if (bfconf->n_processes > 1) {
        for (n = 0; n < bfconf->n_channels[OUT]; n++) {
            if (bfhip_engine_set_output_active(bfhip_eng, n,
                                               bfhip_output_owner(n) ==
                                               process_index) < 0)
            {
                bfhip_die("set_output_active");
            }
        }
    }
    if (bfhip_engine_finalize(bfhip_eng) < 0) {
        bfhip_die("finalize");
    }
    /* One period per call keeps the reference's documented I/O delay (brutefir.html:839):
       graph replay, completion watched from the CPU.  BFHIP_TWO_PERIODS=1 (blocking I/O only)
       keeps two periods in flight instead -- upload of t+1 and download of t-1 ride the copy
       engines beside the kernels of t -- at the cost of one period of extra I/O delay.
       `benchmark: true` / `debug: true`: plain launches bracketed by HIP events, so that the
       stage table of bfrun.c:2035-2078 has device times to show. */
    bfhip_pipelined = !callback_io && getenv("BFHIP_TWO_PERIODS") != NULL;
    flags = bfhip_pipelined ? BFHIP_RT_OVERLAP : BFHIP_RT_SPIN;
    if (bfconf->debug || bfconf->benchmark) {
        flags |= BFHIP_RT_NO_GRAPH;
        if (bfhip_engine_enable_timing(bfhip_eng, 1) < 0) {
            bfhip_die("enable_timing");
        }
    }
    if (bfhip_engine_rt_begin(bfhip_eng, flags) < 0) {
        bfhip_die("rt_begin");
    }
    pinfo("MI355X backend active: filter process %d of %d on device %d (%s).\n",
          process_index, bfconf->n_processes, process_index % n_devices,
          bfhip_pipelined ? "two periods in flight" : "one period per call");
}

/* one period: the fctrl state goes to the engine (setters are no-ops when nothing changed),
   then the block itself.  This process's own filters come from the snapshot just taken under
   the mutex; the other processes' filters (only the SHAPE of the plan follows them) are read
   from icomm under the mutex here. */
static void
bfhip_period(struct bffilter_control icomm_fctrl[],
             uint32_t icomm_ismuted[2][BF_MAXCHANNELS/32],
             int icomm_delay[2][BF_MAXCHANNELS],
             int icomm_subdelay[2][BF_MAXCHANNELS],
             void *inbuf,
             void *outbuf)
{
    static struct bffilter_control others[BF_MAXFILTERS];
    struct bffilter_control *fc;
    struct bffilter *flt;
    int n, i, idx, coeff, st;

    if (bfconf->n_processes > 1) {
        /* the other processes' filters: one pass under the mutex */
        icomm_mutex(1);
        for (n = 0; n < BF_MAXFILTERS; n++) {
            if (bfhip_index[n] < 0 || bfhip_local[n] >= 0) {
                continue;
            }
            flt = bfhip_filter[n];
            others[n].coeff = icomm->fctrl[n].coeff;
            others[n].delayblocks = icomm->fctrl[n].delayblocks;
            for (i = 0; i < flt->n_channels[IN]; i++) {
                others[n].scale[IN][i] = icomm->fctrl[n].scale[IN][i];
            }
            for (i = 0; i < flt->n_channels[OUT]; i++) {
                others[n].scale[OUT][i] = icomm->fctrl[n].scale[OUT][i];
            }
            for (i = 0; i < flt->n_filters[IN]; i++) {
                others[n].fscale[i] = icomm->fctrl[n].fscale[i];
            }
        }
        icomm_mutex(0);
    }
    for (n = 0; n < BF_MAXFILTERS; n++) {
        if ((idx = bfhip_index[n]) < 0) {
            continue;
        }
        flt = bfhip_filter[n];
        fc = bfhip_local[n] >= 0 ? &icomm_fctrl[bfhip_local[n]] : &others[n];
        coeff = fc->coeff;
        if (events.n_coeff_final == 1) {
            events.coeff_final[0](n, &coeff);
        }
        bfhip_engine_set_coeff(bfhip_eng, idx, coeff);
        bfhip_engine_set_delayblocks(bfhip_eng, idx, fc->delayblocks);
        for (i = 0; i < flt->n_channels[IN]; i++) {
            bfhip_engine_set_scale(bfhip_eng, idx, BFHIP_IN, i, fc->scale[IN][i]);
        }
        for (i = 0; i < flt->n_channels[OUT]; i++) {
            bfhip_engine_set_scale(bfhip_eng, idx, BFHIP_OUT, i, fc->scale[OUT][i]);
        }
        for (i = 0; i < flt->n_filters[IN]; i++) {
            bfhip_engine_set_fscale(bfhip_eng, idx, i, fc->fscale[i]);
        }
    }
    FOR_IN_AND_OUT {
        for (n = 0; n < bfconf->n_channels[IO]; n++) {
            bfhip_engine_set_delay(bfhip_eng, IO, n, icomm_delay[IO][n]);
            bfhip_engine_set_mute(bfhip_eng, IO, n,
                                  bit_isset(icomm_ismuted[IO], n));
            if (bfconf->use_subdelay[IO]) {
                bfhip_engine_set_subdelay(bfhip_eng, IO, n,
                                          icomm_subdelay[IO][n]);
            }
        }
    }
    /* outbuf and icomm->overflow are shared by the filter processes: the engine writes the
       samples and the overflow entries of the outputs it owns, nothing else
       (struct bfoverflow == bfhip_overflow) */
    if (bfhip_pipelined) {
        st = bfhip_engine_rt_submit(bfhip_eng, inbuf);
        if (st >= 0 && ++bfhip_inflight == 2) {
            /* the period submitted one call ago */
            st = bfhip_engine_rt_wait(bfhip_eng, outbuf,
                                      (bfhip_overflow *)icomm->overflow);
            bfhip_inflight--;
        }
    } else {
        st = bfhip_engine_rt_block(bfhip_eng, inbuf, outbuf,
                                   (bfhip_overflow *)icomm->overflow);
    }
    if (st < 0) {
        bfhip_die("block");
    }
    if ((st & BFHIP_ST_NONFINITE) != 0) {
        fprintf(stderr, "NaN or Inf values in the system! "
                "Invalid input? Aborting.\n");
        bf_exit(BF_EXIT_OTHER);
    }
    if ((st & BFHIP_ST_SAFETY) != 0) {
        fprintf(stderr, "Safety limit exceeded on output! Aborting.\n");
        bf_exit(BF_EXIT_OTHER);
    }
}
#endif /* BF_HAVE_BFHIP */
'''

SETUP_CALL = r'''#ifdef BF_HAVE_BFHIP
    if (bfhip_wanted()) {
        bfhip_setup(process_index, has_cb_input_devs || has_cb_output_devs);
    }
#endif
'''

BLOCK_CALL = r'''#ifdef BF_HAVE_BFHIP
        if (bfhip_eng != NULL) {
            /* No data passes between the processes here (every engine transforms all inputs and
               mixes its own outputs), but ONE of the reference's two barriers (:1563, :1873) stays:
               all filter processes are woken through the same pipe, n_processes tokens per
               period, and only a barrier keeps a fast process from taking a second token of the
               same period -- and with it a period the input process has not written yet. */
            synch_filter_processes(filter_readfd, filter_writefd, process_index);
            bfhip_period(icomm_fctrl, icomm_ismuted, icomm_delay, icomm_subdelay,
                         inbuf[curbuf], outbuf[curbuf]);
            for (n = 0; n < n_filters; n++) {
                if (procblocks[n] < n_blocks) {
                    procblocks[n]++;
                } else {
                    bit_clr(partial_proc, n);
                }
            }
            if ((bfconf->debug || bfconf->benchmark) && (cc + 1) % 10 == 0) {
                /* the stage table below (mean over 10 periods, in clock ticks): device times
                   of the periods since the last print, in the reference's columns */
                double bfhip_ms[8];
                if (bfhip_engine_stage_times(bfhip_eng, bfhip_ms) > 0) {
                    for (i = 0; i < 7; i++) {
                        t[i] += (uint64_t)(bfhip_ms[i] * 10.0 *
                                           bfconf->cpu_mhz * 1000.0);
                    }
                }
            }
            goto bfhip_period_done;
        }
#endif
'''

LABEL = r'''#ifdef BF_HAVE_BFHIP
    bfhip_period_done:
#endif
'''


def insert_before(lines, anchor, text, occurrence=1, after=False):
    hits = [i for i, ln in enumerate(lines) if ln.rstrip("\n") == anchor]
    if len(hits) < occurrence:
        raise SystemExit("anchor not found: %r" % anchor)
    at = hits[occurrence - 1] + (1 if after else 0)
    return lines[:at] + text.splitlines(keepends=True) + lines[at:]


def patched_source(src_lines):
    lines = list(src_lines)
    # 1. helpers: behind the `events` table they refer to, i.e. just before filter_process()'s
    #    nearest preceding function; the init function of the events ends before this helper
    lines = insert_before(lines, "static void", HELPERS.lstrip("\n") + "\n",
                          occurrence=_nth_static_void_before(lines, "filter_process(struct bfaccess *bfaccess,"))
    # 2. engine set-up: in the child, after all buffers exist, before the init handshake
    lines = insert_before(lines, "    if (bfconf->realtime_priority) {", SETUP_CALL,
                          occurrence=_occurrence_after(lines, "    if (bfconf->realtime_priority) {",
                                                       "    memset(ocbuf[0], 0, convbufsize);"))
    # 3. the period itself: right after timestamp(&t3)
    lines = insert_before(lines, "\ttimestamp(&t3);", BLOCK_CALL, after=True)
    # 4. where the unfused body ends
    lines = insert_before(lines, "\ttimestamp(&t4);", LABEL)
    return lines


def _nth_static_void_before(lines, marker):
    m = [i for i, ln in enumerate(lines) if ln.rstrip("\n") == marker][0]
    hits = [i for i, ln in enumerate(lines) if ln.rstrip("\n") == "static void" and i < m]
    return len(hits)                  # the one that opens filter_process() itself


def _occurrence_after(lines, anchor, marker):
    m = [i for i, ln in enumerate(lines) if ln.rstrip("\n") == marker][0]
    hits = [i for i, ln in enumerate(lines) if ln.rstrip("\n") == anchor]
    for k, i in enumerate(hits):
        if i > m:
            return k + 1
    raise SystemExit("no %r after %r" % (anchor, marker))


def main():
    src = open(os.path.join(REF, "bfrun.c")).read().splitlines(keepends=True)
    out = patched_source(src)
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "a"))
        os.makedirs(os.path.join(td, "b"))
        open(os.path.join(td, "a", "bfrun.c"), "w").writelines(src)
        open(os.path.join(td, "b", "bfrun.c"), "w").writelines(out)
        r = subprocess.run(["diff", "-U2", "--label", "a/bfrun.c", "--label", "b/bfrun.c",
                            "a/bfrun.c", "b/bfrun.c"], cwd=td, capture_output=True, text=True)
        if r.returncode != 1:
            raise SystemExit("diff failed: %s" % r.stderr)
    dst = os.path.join(ROOT, "patches", "bfrun-bfhip.diff")
    open(dst, "w").write(r.stdout)
    print("wrote", dst, "(%d lines)" % len(r.stdout.splitlines()))


if __name__ == "__main__":
    main()
