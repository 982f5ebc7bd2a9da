/*
 * TEST INFRASTRUCTURE ONLY -- never linked into or called from the product.
 *
 * The reference's OWN filter_process() (bfrun.c:1008-2083: the block loop with its ring-slot
 * arithmetic, delay clamp, cblocks truncation, warm-up guard, coefficient switch with cross-fade,
 * filter-to-filter cascades, input / output mixing), compiled UNCHANGED from the source where it
 * lies under /root/reference (`#include "bfrun.c"` below: the function is static), run as the
 * reference runs it -- in a forked filter process, woken through its pipes, on shared buffers --
 * over the PRODUCT's 22 convolver.h symbols (libbfhip.so: the per-block ops on the GPU, the
 * pre-fork ops on the host).  This is the drop-in itself, unfused: reference host code on top,
 * product below the convolver.h boundary, nothing in between.
 *
 * What it is for: SURVEY 8(a) rows A7, A8, A12 and the flow of A13 had no reference OUTPUT to be
 * pinned by (no FFTW here -> the reference's own convolver cannot be built; no tests or fixtures in
 * the reference).  With this binary the reference's control flow produces outputs: the fused
 * engine (bfhip_engine_block) and the oracle's restatement of filter_process() have to agree with
 * them on random filter networks with run-time control changes (tests/test_gpu_refloop.py).  A
 * wrong ring slot, a missed clamp, a fade on the wrong block is a gross error, not a rounding one.
 * What it does NOT pin: the FFT itself (A2 / A10) -- both sides use this repository's transforms.
 *
 * Linked from the reference's own objects: bfrun.c (included here), dai.c, delay.c, firwindow.c,
 * emalloc.c, shmalloc.c, inout.c -- every undefined symbol of theirs is closed by those files,
 * libc and libbfhip.so; no placeholder, no stub.  What the harness supplies is DATA: the `bfconf`
 * structure bfconf.c would fill from a configuration file (its lexer needs flex, absent here), the
 * dai_buffer_format tables dai_init() would fill from opened devices, and the input / output
 * process on the other end of the pipes.  Spec file format: tests/test_gpu_refloop.py writes it.
 * A spec may carry a channel section (magic ...53): N:1 virtual -> physical maps, integer delays with
 * their maxima, mutes, sub-sample delays, dither flags -- then bfrun.c:1128-1198, 1505-1531 and
 * 1938-2003 (delay, mute and mix of channels that share a physical one) run too, with the reference's
 * own delay.c and dither.c underneath.
 *
 *   ref_filter_process <spec file> <raw output file>
 */
#include <sys/mman.h>
#include <sys/wait.h>

#include "bfrun.c"
#include "dither.h"

struct bfconf *bfconf = NULL;

struct spec_event { int32_t block, kind, filter, index; double value; };

static void
rd(FILE *f, void *p, size_t n)
{
    if (n > 0 && fread(p, 1, n, f) != n) {
        fprintf(stderr, "ref_filter_process: short spec file\n");
        exit(2);
    }
}

static int32_t
rd32(FILE *f)
{
    int32_t v;
    rd(f, &v, 4);
    return v;
}

static void *
shared_zero(size_t bytes)
{
    void *p = mmap(NULL, bytes ? bytes : 1, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) {
        perror("mmap");
        exit(2);
    }
    memset(p, 0, bytes);
    return p;
}

/* the formats a spec may name: code -> (isfloat, bytes, sbytes) as bfconf.c:358-533 tabulates them */
static void
set_format(struct sample_format *sf, int code)
{
    static const int tab[][3] = { {0, 2, 2}, {0, 4, 3}, {0, 4, 4}, {1, 4, 4}, {1, 8, 8}, {0, 3, 3} };   /* S16 S24_4 S32 F32 F64 S24 (LE) */
    memset(sf, 0, sizeof(*sf));
    sf->isfloat = tab[code][0];
    sf->bytes = tab[code][1];
    sf->sbytes = tab[code][2];
    sf->swap = false;
    sf->scale = sf->isfloat ? 1.0 : 1.0 / (double)((uint64_t)1 << ((sf->sbytes << 3) - 1));
}

int
main(int argc, char *argv[])
{
    static struct bfconf conf;
    FILE *f, *fo;
    int L, N, rs, n_ch[2], fmt[2], n_coeffs, n_filters, n_blocks, n_events, n_procs;
    int n, i, j, c, k, curbuf = 0, status = 0;
    int in_pipe[2], out_pipe[2], cb_in[2], cb_out[2], per_in, per_out, f2f[BF_MAXPROCESSES][2], filter_writefd[BF_MAXPROCESSES];
    int *fft_ch[BF_MAXPROCESSES][2], n_fft_ch[BF_MAXPROCESSES][2], cpos[2], *fproc_of;
    char toks[BF_MAXPROCESSES];
    pid_t pids[BF_MAXPROCESSES];
    struct spec_event *ev;
    void *inbuf[2], *outbuf[2], **in_freq, **out_freq;
    struct bfaccess bfaccess;
    uint8_t *rawin;
    size_t in_bytes, out_bytes;
    char tok = 0;
    int ext, n_phys[2], sdf_length = -1;
    int32_t *x_v2p[2] = { NULL, NULL }, *x_delay[2], *x_maxdelay[2], *x_mute[2], *x_subdelay[2], *x_dither = NULL;
    double sdf_beta = 9.0, *x_scale;
    uint8_t **x_taps;
    int *x_ntaps;

    if (argc != 3 || (f = fopen(argv[1], "rb")) == NULL) {
        fprintf(stderr, "usage: ref_filter_process spec out.raw\n");
        return 2;
    }
    ext = rd32(f);
    if (ext != 0x42465251 && ext != 0x42465253) {
        fprintf(stderr, "ref_filter_process: bad magic\n");
        return 2;
    }
    ext = ext == 0x42465253;
    L = rd32(f); N = rd32(f); rs = rd32(f);
    n_ch[IN] = rd32(f); n_ch[OUT] = rd32(f);
    fmt[IN] = rd32(f); fmt[OUT] = rd32(f);
    n_coeffs = rd32(f); n_filters = rd32(f); n_blocks = rd32(f); n_events = rd32(f);
    n_procs = rd32(f);
    if (n_procs < 1 || n_procs > BF_MAXPROCESSES) {
        fprintf(stderr, "ref_filter_process: bad process count\n");
        return 2;
    }

    n_phys[IN] = n_ch[IN];
    n_phys[OUT] = n_ch[OUT];
    if (ext) {
        /* the channel section: per side the physical channel count and, per virtual channel, its
           physical channel, delay, maxdelay, mute, subdelay; then the dither flag of every physical
           output, sdf_length and sdf_beta */
        FOR_IN_AND_OUT {
            n_phys[IO] = rd32(f);
            x_v2p[IO] = emalloc((n_ch[IO] + 1) * 4); rd(f, x_v2p[IO], n_ch[IO] * 4);
            x_delay[IO] = emalloc((n_ch[IO] + 1) * 4); rd(f, x_delay[IO], n_ch[IO] * 4);
            x_maxdelay[IO] = emalloc((n_ch[IO] + 1) * 4); rd(f, x_maxdelay[IO], n_ch[IO] * 4);
            x_mute[IO] = emalloc((n_ch[IO] + 1) * 4); rd(f, x_mute[IO], n_ch[IO] * 4);
            x_subdelay[IO] = emalloc((n_ch[IO] + 1) * 4); rd(f, x_subdelay[IO], n_ch[IO] * 4);
            if (n_phys[IO] < 1 || n_phys[IO] > n_ch[IO]) {
                fprintf(stderr, "ref_filter_process: bad channel section\n");
                return 2;
            }
        }
        x_dither = emalloc((n_phys[OUT] + 1) * 4); rd(f, x_dither, n_phys[OUT] * 4);
        sdf_length = rd32(f);
        rd(f, &sdf_beta, 8);
    }

    /* ---- what bfconf_init() leaves behind (bfconf.c:2786 ff.), for a configuration without modules */
    bfconf = &conf;
    memset(&conf, 0, sizeof(conf));
    conf.cpu_mhz = 1000.0;
    conf.n_cpus = 1;
    conf.sampling_rate = 44100;
    conf.filter_length = L;
    conf.n_blocks = N;
    conf.realsize = rs;
    conf.blocking_io = true;
    conf.quiet = true;
    conf.n_processes = n_procs;
    if (getenv("BFREF_POWERSAVE") != NULL) {
        /* `powersave: true;` (1.0: exact zeros) or `powersave: <dB>;` (its linear value), bfconf.c:1549-1561 */
        conf.powersave = true;
        conf.analog_powersave = atof(getenv("BFREF_POWERSAVE"));
    }
    conf.benchmark = getenv("BFREF_BENCHMARK") != NULL;     /* `benchmark: true;`: the stage table of bfrun.c:2035-2078 on stderr */
    if (!convolver_init(NULL, L, rs)) {
        return 2;
    }
    conf.sdf_length = sdf_length;
    conf.sdf_beta = sdf_beta;
    FOR_IN_AND_OUT {
        conf.n_channels[IO] = n_ch[IO];
        conf.n_physical_channels[IO] = n_phys[IO];
        conf.n_virtperphys[IO] = emalloc(n_phys[IO] * sizeof(int));
        conf.phys2virt[IO] = emalloc(n_phys[IO] * sizeof(int *));
        conf.virt2phys[IO] = emalloc(n_ch[IO] * sizeof(int));
        conf.delay[IO] = emalloc(n_ch[IO] * sizeof(int));
        conf.maxdelay[IO] = emalloc(n_ch[IO] * sizeof(int));
        conf.mute[IO] = emalloc(n_ch[IO] * sizeof(bool_t));
        conf.subdelay[IO] = emalloc(n_ch[IO] * sizeof(int));
        for (n = 0; n < n_phys[IO]; n++) {
            conf.n_virtperphys[IO][n] = 0;
            conf.phys2virt[IO][n] = emalloc((n_ch[IO] + 1) * sizeof(int));
        }
        for (n = 0; n < n_ch[IO]; n++) {
            /* a physical channel's virtual ones in ascending order (bfconf.c:1355-1379) */
            const int p = ext ? x_v2p[IO][n] : n;
            if (p < 0 || p >= n_phys[IO]) {
                fprintf(stderr, "ref_filter_process: bad channel map\n");
                return 2;
            }
            conf.virt2phys[IO][n] = p;
            conf.phys2virt[IO][p][conf.n_virtperphys[IO][p]++] = n;
            conf.delay[IO][n] = ext ? x_delay[IO][n] : 0;
            conf.maxdelay[IO][n] = ext ? x_maxdelay[IO][n] : 0;
            conf.mute[IO][n] = ext ? !!x_mute[IO][n] : false;
            conf.subdelay[IO][n] = ext ? x_subdelay[IO][n] : BF_UNDEFINED_SUBDELAY;
            if (conf.subdelay[IO][n] != BF_UNDEFINED_SUBDELAY) {
                conf.use_subdelay[IO] = true;           /* bfconf.c:1343-1351 */
            }
        }
        dai_buffer_format[IO] = emalloc(sizeof(struct dai_buffer_format));
        memset(dai_buffer_format[IO], 0, sizeof(struct dai_buffer_format));
        dai_buffer_format[IO]->n_samples = L;
        dai_buffer_format[IO]->n_channels = n_phys[IO];
        for (n = 0; n < n_phys[IO]; n++) {
            /* one interleaved device with all channels open (dai.c:537-576) */
            set_format(&dai_buffer_format[IO]->bf[n].sf, fmt[IO]);
            dai_buffer_format[IO]->bf[n].sample_spacing = n_phys[IO];
            dai_buffer_format[IO]->bf[n].byte_offset = n * dai_buffer_format[IO]->bf[n].sf.bytes;
        }
        dai_buffer_format[IO]->n_bytes = L * n_phys[IO] * dai_buffer_format[IO]->bf[0].sf.bytes;
        /* that device as bfconf describes it (bfrun.c:1144-1164 looks a channel's sample size up here) */
        conf.n_subdevs[IO] = 1;
        conf.subdevs[IO] = emalloc(sizeof(struct dai_subdevice));
        memset(conf.subdevs[IO], 0, sizeof(struct dai_subdevice));
        conf.subdevs[IO][0].channels.sf = dai_buffer_format[IO]->bf[0].sf;
        conf.subdevs[IO][0].channels.open_channels = conf.subdevs[IO][0].channels.used_channels = n_phys[IO];
        conf.subdevs[IO][0].channels.channel_selection = emalloc(n_phys[IO] * sizeof(int));
        conf.subdevs[IO][0].channels.channel_name = emalloc(n_phys[IO] * sizeof(int));
        for (n = 0; n < n_phys[IO]; n++) {
            conf.subdevs[IO][0].channels.channel_selection[n] = conf.subdevs[IO][0].channels.channel_name[n] = n;
        }
    }
    if (sdf_length < 0) {
        conf.use_subdelay[IN] = conf.use_subdelay[OUT] = false;         /* bfconf.c:2796-2798 */
    }
    if ((conf.use_subdelay[IN] || conf.use_subdelay[OUT]) &&
        !delay_subsample_init(BF_SAMPLE_SLOTS, conf.sdf_length, conf.sdf_beta, L, rs))      /* bfconf.c:2806-2816 */
    {
        fprintf(stderr, "ref_filter_process: delay_subsample_init failed\n");
        return 2;
    }
    /* dither (bfconf.c:3176-3240): one state per flagged physical output, in channel order */
    conf.dither_state = emalloc(n_phys[OUT] * sizeof(struct dither_state *));
    memset(conf.dither_state, 0, n_phys[OUT] * sizeof(struct dither_state *));
    if (ext) {
        struct dither_state *states[BF_MAXCHANNELS];
        for (n = j = 0; n < n_phys[OUT]; n++) {
            j += !!x_dither[n];
        }
        if (j > 0) {
            if (!dither_init(j, conf.sampling_rate, rs, conf.max_dither_table_size, L, states)) {
                fprintf(stderr, "ref_filter_process: dither_init failed\n");
                return 2;
            }
            for (n = j = 0; n < n_phys[OUT]; n++) {
                if (x_dither[n]) {
                    conf.dither_state[n] = states[j++];
                }
            }
        }
    }

    /* coefficient sets the way load_coeff() prepares them (bfconf.c:1979-2019): one
       convolver_coeffs2cbuf() per block of L taps */
    conf.n_coeffs = n_coeffs;
    conf.coeffs = emalloc(n_coeffs * sizeof(struct bfcoeff));
    conf.coeffs_data = emalloc(n_coeffs * sizeof(void **));
    memset(conf.coeffs, 0, n_coeffs * sizeof(struct bfcoeff));
    x_taps = emalloc((n_coeffs + 1) * sizeof(uint8_t *));
    x_ntaps = emalloc((n_coeffs + 1) * sizeof(int));
    x_scale = emalloc((n_coeffs + 1) * sizeof(double));
    for (c = 0; c < n_coeffs; c++) {
        int n_taps = rd32(f), blocks = rd32(f);
        double scale;
        uint8_t *taps, *zbuf, *dest = NULL;
        rd(f, &scale, 8);
        taps = emalloc((size_t)n_taps * rs + 1);
        rd(f, taps, (size_t)n_taps * rs);
        x_taps[c] = taps; x_ntaps[c] = n_taps; x_scale[c] = scale;
        if (blocks <= 0) {
            blocks = (n_taps + L - 1) / L;
        }
        zbuf = emalloc((size_t)L * rs);
        memset(zbuf, 0, (size_t)L * rs);
        conf.coeffs[c].intname = c;
        conf.coeffs[c].n_blocks = blocks;
        conf.coeffs[c].is_shared = getenv("BFREF_SHARED_COEFFS") != NULL;      /* `shared_mem: true;` */
        if (conf.coeffs[c].is_shared) {
            dest = shared_zero((size_t)2 * blocks * L * rs);                     /* bfconf.c:1983-1989 */
        }
        conf.coeffs_data[c] = emalloc(blocks * sizeof(void *));
        for (n = 0; n < blocks; n++) {
            if (n * L > n_taps) {
                conf.coeffs_data[c][n] = convolver_coeffs2cbuf(zbuf, L, scale, dest);
            } else if ((n + 1) * L > n_taps) {
                conf.coeffs_data[c][n] = convolver_coeffs2cbuf(&taps[(size_t)n * L * rs], n_taps - n * L, scale, dest);
            } else {
                conf.coeffs_data[c][n] = convolver_coeffs2cbuf(&taps[(size_t)n * L * rs], L, scale, dest);
            }
            if (dest != NULL) {
                dest += (size_t)2 * L * rs;
            }
            if (conf.coeffs_data[c][n] == NULL) {
                fprintf(stderr, "ref_filter_process: coefficient set %d rejected\n", c);
                return 2;
            }
        }
    }

    /* filters, already in evaluation order (bfconf.c:2933-2964 sorts them so) */
    conf.n_filters = n_filters;
    conf.filters = emalloc(n_filters * sizeof(struct bffilter));
    conf.initfctrl = emalloc(n_filters * sizeof(struct bffilter_control));
    memset(conf.filters, 0, n_filters * sizeof(struct bffilter));
    memset(conf.initfctrl, 0, n_filters * sizeof(struct bffilter_control));
    fproc_of = emalloc((n_filters + 1) * sizeof(int));
    for (n = 0; n < n_filters; n++) {
        struct bffilter *fl = &conf.filters[n];
        struct bffilter_control *fc = &conf.initfctrl[n];
        fl->intname = n;
        fproc_of[n] = rd32(f);              /* the process bfconf's load balancing gave the filter to (bfconf.c:2227-2318) */
        fl->n_channels[IN] = rd32(f);
        fl->n_filters[IN] = rd32(f);
        fl->n_channels[OUT] = rd32(f);
        fc->coeff = rd32(f);
        fc->delayblocks = rd32(f);
        fl->crossfade = rd32(f);
        fl->channels[IN] = emalloc((fl->n_channels[IN] + 1) * sizeof(int));
        fl->filters[IN] = emalloc((fl->n_filters[IN] + 1) * sizeof(int));
        fl->channels[OUT] = emalloc((fl->n_channels[OUT] + 1) * sizeof(int));
        fl->filters[OUT] = emalloc(sizeof(int));
        rd(f, fl->channels[IN], fl->n_channels[IN] * 4);
        rd(f, fc->scale[IN], fl->n_channels[IN] * 8);
        rd(f, fl->filters[IN], fl->n_filters[IN] * 4);
        rd(f, fc->fscale, fl->n_filters[IN] * 8);
        rd(f, fl->channels[OUT], fl->n_channels[OUT] * 4);
        rd(f, fc->scale[OUT], fl->n_channels[OUT] * 8);
    }
    conf.fproc = emalloc(n_procs * sizeof(struct filter_process));
    for (k = 0; k < n_procs; k++) {
        /* the filters of process k, in evaluation order (bfconf.c:2933-2964), and the channels they
           touch (bfconf.c:2858-2884) */
        conf.fproc[k].filters = emalloc((n_filters + 1) * sizeof(struct bffilter));
        conf.fproc[k].n_filters = 0;
        for (n = 0; n < n_filters; n++) {
            if (fproc_of[n] == k) {
                conf.fproc[k].filters[conf.fproc[k].n_filters++] = conf.filters[n];
            }
        }
        FOR_IN_AND_OUT {
            char used[BF_MAXCHANNELS];
            memset(used, 0, sizeof(used));
            for (n = 0; n < conf.fproc[k].n_filters; n++) {
                for (i = 0; i < conf.fproc[k].filters[n].n_channels[IO]; i++) {
                    used[conf.fproc[k].filters[n].channels[IO][i]] = 1;
                }
            }
            conf.fproc[k].unique_channels[IO] = emalloc((n_ch[IO] + 1) * sizeof(int));
            conf.fproc[k].n_unique_channels[IO] = 0;
            for (n = 0; n < n_ch[IO]; n++) {
                if (used[n]) {
                    conf.fproc[k].unique_channels[IO][conf.fproc[k].n_unique_channels[IO]++] = n;
                }
            }
        }
    }
    /* which channels a process transforms / converts: bfrun.c:2312-2328 */
    cpos[IN] = cpos[OUT] = 0;
    for (k = 0; k < n_procs; k++) {
        FOR_IN_AND_OUT {
            int nc = conf.n_channels[IO] / n_procs;
            fft_ch[k][IO] = emalloc((n_ch[IO] + 1) * sizeof(int));
            j = 0;
            while ((j < nc || k == n_procs - 1) && cpos[IO] < conf.n_physical_channels[IO]) {
                for (i = 0; i < conf.n_virtperphys[IO][cpos[IO]]; i++, j++) {
                    fft_ch[k][IO][j] = conf.phys2virt[IO][cpos[IO]][i];
                }
                cpos[IO]++;
            }
            n_fft_ch[k][IO] = j;
        }
    }
    ev = emalloc((n_events + 1) * sizeof(struct spec_event));
    rd(f, ev, n_events * sizeof(struct spec_event));
    in_bytes = dai_buffer_format[IN]->n_bytes;
    out_bytes = dai_buffer_format[OUT]->n_bytes;
    rawin = emalloc((size_t)n_blocks * in_bytes + 1);
    rd(f, rawin, (size_t)n_blocks * in_bytes);
    fclose(f);

    /* ---- what bfrun() sets up before it forks (bfrun.c:2160-2310) */
    icomm = shared_zero(sizeof(struct intercomm_area));
    for (n = 0; n < n_filters; n++) {
        icomm->fctrl[n] = conf.initfctrl[n];
    }
    icomm->n_pids = 0;
    icomm->exit_status = BF_EXIT_OK;
    FOR_IN_AND_OUT {                    /* bfrun.c:2200-2211 */
        for (n = 0; n < n_ch[IO]; n++) {
            icomm->delay[IO][n] = conf.delay[IO][n];
            icomm->subdelay[IO][n] = conf.subdelay[IO][n];
            if (conf.mute[IO][n]) {
                bit_set_volatile(icomm->ismuted[IO], n);
            }
        }
    }
    for (n = 0; n < n_ch[OUT]; n++) {
        struct sample_format *sf = &dai_buffer_format[OUT]->bf[conf.virt2phys[OUT][n]].sf;
        icomm->overflow[n].max = sf->isfloat ? 1.0 : (double)((uint64_t)1 << ((sf->sbytes << 3) - 1)) - 1;
    }
    if (pipe(mutex_pipe) == -1 || pipe(in_pipe) == -1 || pipe(out_pipe) == -1 || pipe(cb_in) == -1 || pipe(cb_out) == -1 ||
        !writefd(mutex_pipe[1], &tok, 1)) {
        perror("pipe");
        return 2;
    }
    for (k = 0; k < n_procs; k++) {
        if (pipe(f2f[k]) == -1) {
            perror("pipe");
            return 2;
        }
    }
    memset(toks, 0, sizeof(toks));
    init_events();
    memset(&bfaccess, 0, sizeof(bfaccess));
    bfaccess.fctrl = icomm->fctrl;
    bfaccess.overflow = icomm->overflow;
    bfaccess.realsize = rs;
    bfaccess.coeffs_data = conf.coeffs_data;
    bfaccess.control_mutex = icomm_mutex;
    for (k = 0; k < 2; k++) {
        inbuf[k] = shared_zero(in_bytes);
        outbuf[k] = shared_zero(out_bytes);
    }
    in_freq = emalloc(n_ch[IN] * sizeof(void *));
    out_freq = emalloc(n_ch[OUT] * sizeof(void *));
    for (n = 0; n < n_ch[IN]; n++) {
        in_freq[n] = shared_zero(convolver_cbufsize());
    }
    for (n = 0; n < n_ch[OUT]; n++) {
        out_freq[n] = shared_zero(convolver_cbufsize());
    }
    for (k = 0; k < n_procs; k++) {
        switch (pids[k] = fork()) {
        case -1:
            perror("fork");
            return 2;
        case 0:
            /* ---- filter process k (bfrun.c:2329-2375) */
            close(in_pipe[1]);
            close(out_pipe[0]);
            for (i = 0; i < n_procs; i++) {
                if (i == k) {
                    filter_writefd[i] = -1;
                    close(f2f[i][1]);
                } else {
                    close(f2f[i][0]);
                    filter_writefd[i] = f2f[i][1];
                }
            }
            /* BFREF_CALLBACK_IO: the devices are callback ones (JACK: bfio_jack) -- the init handshake stays on
               the blocking pipes (bfrun.c:1124, 1414), the periods are woken and reported on the callback
               pipes (:1431, :2021) */
            if (getenv("BFREF_CALLBACK_IO") != NULL) {
                close(cb_in[1]);
                close(cb_out[0]);
                filter_process(&bfaccess, inbuf, outbuf, in_freq, out_freq,
                               f2f[k][0], filter_writefd,
                               in_pipe[0], cb_in[0], out_pipe[1], cb_out[1],
                               n_fft_ch[k][IN], fft_ch[k][IN], n_fft_ch[k][OUT], fft_ch[k][OUT],
                               conf.fproc[k].n_unique_channels[IN], conf.fproc[k].unique_channels[IN],
                               conf.fproc[k].n_unique_channels[OUT], conf.fproc[k].unique_channels[OUT],
                               conf.fproc[k].n_filters, conf.fproc[k].filters, k,
                               false, false, true, true);
            }
            filter_process(&bfaccess, inbuf, outbuf, in_freq, out_freq,
                           f2f[k][0], filter_writefd,
                           in_pipe[0], -1, out_pipe[1], -1,
                           n_fft_ch[k][IN], fft_ch[k][IN], n_fft_ch[k][OUT], fft_ch[k][OUT],
                           conf.fproc[k].n_unique_channels[IN], conf.fproc[k].unique_channels[IN],
                           conf.fproc[k].n_unique_channels[OUT], conf.fproc[k].unique_channels[OUT],
                           conf.fproc[k].n_filters, conf.fproc[k].filters, k,
                           true, true, false, false);
            _exit(3);       /* never reached */
        default:
            break;
        }
    }
    close(in_pipe[0]);
    close(out_pipe[1]);
    close(cb_in[0]);
    close(cb_out[1]);
    per_in = getenv("BFREF_CALLBACK_IO") != NULL ? cb_in[1] : in_pipe[1];
    per_out = getenv("BFREF_CALLBACK_IO") != NULL ? cb_out[0] : out_pipe[0];

    /* ---- the input and output processes on the other end of the pipes (bfrun.c:2480-2616) */
    if ((fo = fopen(argv[2], "wb")) == NULL) {
        perror(argv[2]);
        return 2;
    }
    if (!writefd(in_pipe[1], toks, n_procs) || !readfd(out_pipe[0], toks, n_procs)) {       /* the init handshake */
        fprintf(stderr, "ref_filter_process: the filter process did not come up\n");
        return 1;
    }
    for (k = 0; k < n_blocks; k++) {
        for (i = 0; i < n_events; i++) {
            if (ev[i].block != k) {
                continue;
            }
            icomm_mutex(1);             /* what bflogic_cli does through bfaccess->control_mutex */
            switch (ev[i].kind) {
            case 0: icomm->fctrl[ev[i].filter].coeff = (int)ev[i].value; break;
            case 1: icomm->fctrl[ev[i].filter].scale[IN][ev[i].index] = ev[i].value; break;
            case 2: icomm->fctrl[ev[i].filter].scale[OUT][ev[i].index] = ev[i].value; break;
            case 3: icomm->fctrl[ev[i].filter].fscale[ev[i].index] = ev[i].value; break;
            case 4: icomm->fctrl[ev[i].filter].delayblocks = (int)ev[i].value; break;
            /* what bfaccess->toggle_mute / set_delay / set_subdelay leave in icomm for channels that
               share a physical one (bfrun.c:458-541; for 1:1 channels they go on to dai.c, which the
               filter process never sees): filter = side, index = virtual channel */
            case 5:
                if (ev[i].value != 0) {
                    bit_set_volatile(icomm->ismuted[ev[i].filter], ev[i].index);
                } else {
                    bit_clr_volatile(icomm->ismuted[ev[i].filter], ev[i].index);
                }
                break;
            case 6:
                if ((int)ev[i].value >= 0 && (int)ev[i].value <= conf.maxdelay[ev[i].filter][ev[i].index]) {      /* set_delay's check, bfrun.c:495 */
                    icomm->delay[ev[i].filter][ev[i].index] = (int)ev[i].value;
                }
                break;
            case 7: icomm->subdelay[ev[i].filter][ev[i].index] = (int)ev[i].value; break;
            default: {
                /* what bflogic_eq does when it has rendered a new curve (render_equaliser, through
                   bfaccess->convolver_coeffs2cbuf = convolver_runtime_coeffs2cbuf, bfrun.c:2305): a
                   partition of a set in shared memory is rewritten in place while the filter process
                   runs; filter = coefficient set, index = partition, value = gain on the set's own taps */
                const int cs = ev[i].filter, b = ev[i].index;
                const int len = x_ntaps[cs] - b * L;
                if (conf.coeffs[cs].is_shared && b < conf.coeffs[cs].n_blocks && len > 0) {
                    uint8_t *tmp = emalloc((size_t)L * rs);
                    memset(tmp, 0, (size_t)L * rs);
                    for (j = 0; j < (len > L ? L : len); j++) {
                        if (rs == 4) {
                            ((float *)tmp)[j] = (float)(((float *)x_taps[cs])[(size_t)b * L + j] * x_scale[cs] * ev[i].value);
                        } else {
                            ((double *)tmp)[j] = ((double *)x_taps[cs])[(size_t)b * L + j] * x_scale[cs] * ev[i].value;
                        }
                    }
                    convolver_runtime_coeffs2cbuf(tmp, conf.coeffs_data[cs][b]);
                    efree(tmp);
                }
                break;
            }
            }
            icomm_mutex(0);
        }
        memcpy(inbuf[curbuf], rawin + (size_t)k * in_bytes, in_bytes);
        /* BFREF_STALL_PROC0: filter process 0 is held while the period's tokens go out and the other
           processes run -- all of them wake on the SAME pipe (bfrun.c:2480-2616), and what keeps a
           fast one from taking process 0's token is a synch_filter_processes() barrier */
        if (getenv("BFREF_STALL_PROC0") != NULL && n_procs > 1) {
            kill(pids[0], SIGSTOP);
            if (!writefd(per_in, toks, n_procs)) {
                return 1;
            }
            usleep((useconds_t)atoi(getenv("BFREF_STALL_PROC0")));      /* the stall, in microseconds */
            kill(pids[0], SIGCONT);
            if (!readfd(per_out, toks, n_procs)) {
                fprintf(stderr, "ref_filter_process: a filter process died in block %d\n", k);
                return 1;
            }
        } else
        if (!writefd(per_in, toks, n_procs) || !readfd(per_out, toks, n_procs)) {
            fprintf(stderr, "ref_filter_process: a filter process died in block %d\n", k);
            return 1;
        }
        fwrite(outbuf[curbuf], 1, out_bytes, fo);
        curbuf = !curbuf;
    }
    fclose(fo);
    for (n = 0; n < n_ch[OUT]; n++) {
        printf("output %d: %u overflows, intlargest %d, largest %.17g\n", n, icomm->overflow[n].n_overflows,
               (int)icomm->overflow[n].intlargest, icomm->overflow[n].largest);
    }
    /* closing the wake pipe ends the filter process the way the reference ends: its read fails and
       it calls bf_exit() */
    close(in_pipe[1]);
    close(cb_in[1]);
    for (k = 0; k < n_procs; k++) {
        waitpid(pids[k], &status, 0);
    }
    return 0;
}
