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
 * the icomm snapshot, block_start/coeff_final events, the output signalling and the
 * benchmark print stay as they are.  Used when this is the only filter process and no module
 * hooks the per-buffer events (input_timed .. output_timed need the host buffers; such
 * configurations keep the unfused path below, which links the same library's convolver_*
 * symbols).  BFHIP_DISABLE=1 in the environment forces the unfused path.
 */
#include "bfhip.h"

static bfhip_engine *bfhip_eng = NULL;
static bool_t bfhip_pipelined = false;
static int bfhip_inflight = 0;

static void
bfhip_die(const char what[])
{
    fprintf(stderr, "bfhip: %s: %s\n", what, bfhip_last_error());
    bf_exit(BF_EXIT_OTHER);
}

static bool_t
bfhip_wanted(void)
{
    if (getenv("BFHIP_DISABLE") != NULL || bfconf->n_processes != 1) {
        return false;
    }
    if (events.n_input_timed > 0 || events.n_input_freqd > 0 ||
        events.n_pre_convolve > 0 || events.n_post_convolve > 0 ||
        events.n_output_freqd > 0 || events.n_output_timed > 0)
    {
        return false;
    }
    return true;
}

/* Runs in the forked filter process (HIP state does not survive fork()): builds the engine
   from what bfconf_init() left in bfconf and icomm. */
static void
bfhip_setup(int n_filters,
            struct bffilter filters[],
            bool_t callback_io)
{
    int n, i, j, c, physch, n_dither, flags;
    int dither_ch[BF_MAXCHANNELS];
    int local[BF_MAXFILTERS];
    double scales[BF_MAXCHANNELS], fscales[BF_MAXFILTERS];
    volatile struct bffilter_control *fc;
    struct buffer_format *bf;
    bfhip_format f;

    bfhip_eng = bfhip_engine_create(0, bfconf->filter_length, bfconf->n_blocks,
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
       memory may be rewritten by a module process at run time (bflogic_eq): watched. */
    {
        double total = 0;
        for (c = 0; c < bfconf->n_coeffs; c++) {
            total += (double)bfconf->coeffs[c].n_blocks * (double)convolver_cbufsize();
        }
        if (bfhip_engine_reserve_coeffs(bfhip_eng, total) < 0) {
            bfhip_die("reserve_coeffs");
        }
    }
    for (c = 0; c < bfconf->n_coeffs; c++) {
        if (bfhip_engine_add_coeff_processed_blocks(bfhip_eng,
                                                    bfconf->coeffs_data[c],
                                                    bfconf->coeffs[c].n_blocks,
                                                    bfconf->coeffs[c].is_shared) != c)
        {
            bfhip_die("add_coeff_processed_blocks");
        }
    }
    for (n = 0; n < n_filters; n++) {
        fc = &icomm->fctrl[filters[n].intname];
        for (i = 0; i < filters[n].n_filters[IN]; i++) {
            for (j = 0; j < n_filters; j++) {
                if (filters[n].filters[IN][i] == filters[j].intname) {
                    break;
                }
            }
            local[i] = j;
            fscales[i] = fc->fscale[i];
        }
        for (i = 0; i < filters[n].n_channels[IN]; i++) {
            scales[i] = fc->scale[IN][i];
        }
        /* output scales go in a second array: reuse the tail of scales[] */
        for (i = 0; i < filters[n].n_channels[OUT]; i++) {
            scales[BF_MAXCHANNELS / 2 + i] = fc->scale[OUT][i];
        }
        if (bfhip_engine_add_filter(bfhip_eng,
                                    filters[n].n_channels[IN],
                                    filters[n].channels[IN], scales,
                                    filters[n].n_filters[IN], local, fscales,
                                    filters[n].n_channels[OUT],
                                    filters[n].channels[OUT],
                                    &scales[BF_MAXCHANNELS / 2],
                                    fc->coeff, fc->delayblocks,
                                    filters[n].crossfade) != n)
        {
            bfhip_die("add_filter");
        }
    }
    if (bfhip_engine_finalize(bfhip_eng) < 0) {
        bfhip_die("finalize");
    }
    /* Callback I/O (bfio_jack) waits for every period: lowest round trip, graph replay.
       Blocking I/O keeps two periods in flight -- upload of t+1 and download of t-1 ride the
       copy engines beside the kernels of t -- at the cost of one period of extra I/O delay;
       BFHIP_SYNC=1 keeps the reference's I/O delay instead. */
    bfhip_pipelined = !callback_io && getenv("BFHIP_SYNC") == NULL;
    flags = bfhip_pipelined ? BFHIP_RT_OVERLAP : BFHIP_RT_SPIN;
    if (bfhip_engine_rt_begin(bfhip_eng, flags) < 0) {
        bfhip_die("rt_begin");
    }
    pinfo("MI355X backend active (%s).\n",
          bfhip_pipelined ? "two periods in flight" : "one period per call");
}

/* one period: the fctrl snapshot just taken under the mutex goes to the engine (setters are
   no-ops when nothing changed), then the block itself */
static void
bfhip_period(int n_filters,
             struct bffilter filters[],
             struct bffilter_control icomm_fctrl[],
             uint32_t icomm_ismuted[2][BF_MAXCHANNELS/32],
             int icomm_delay[2][BF_MAXCHANNELS],
             int icomm_subdelay[2][BF_MAXCHANNELS],
             void *inbuf,
             void *outbuf)
{
    int n, i, coeff, st;

    for (n = 0; n < n_filters; n++) {
        coeff = icomm_fctrl[n].coeff;
        if (events.n_coeff_final == 1) {
            events.coeff_final[0](filters[n].intname, &coeff);
        }
        bfhip_engine_set_coeff(bfhip_eng, n, coeff);
        bfhip_engine_set_delayblocks(bfhip_eng, n, icomm_fctrl[n].delayblocks);
        for (i = 0; i < filters[n].n_channels[IN]; i++) {
            bfhip_engine_set_scale(bfhip_eng, n, BFHIP_IN, i,
                                   icomm_fctrl[n].scale[IN][i]);
        }
        for (i = 0; i < filters[n].n_channels[OUT]; i++) {
            bfhip_engine_set_scale(bfhip_eng, n, BFHIP_OUT, i,
                                   icomm_fctrl[n].scale[OUT][i]);
        }
        for (i = 0; i < filters[n].n_filters[IN]; i++) {
            bfhip_engine_set_fscale(bfhip_eng, n, i, icomm_fctrl[n].fscale[i]);
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
    if (bfhip_pipelined) {
        st = bfhip_engine_rt_submit(bfhip_eng, inbuf);
        if (st >= 0 && ++bfhip_inflight == 2) {
            /* the period submitted one call ago; struct bfoverflow == bfhip_overflow */
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
        bfhip_setup(n_filters, filters, has_cb_input_devs || has_cb_output_devs);
    }
#endif
'''

BLOCK_CALL = r'''#ifdef BF_HAVE_BFHIP
        if (bfhip_eng != NULL) {
            bfhip_period(n_filters, filters, icomm_fctrl, icomm_ismuted,
                         icomm_delay, icomm_subdelay,
                         inbuf[curbuf], outbuf[curbuf]);
            for (n = 0; n < n_filters; n++) {
                if (procblocks[n] < n_blocks) {
                    procblocks[n]++;
                } else {
                    bit_clr(partial_proc, n);
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
