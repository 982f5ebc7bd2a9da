"""GPU parity for every bookkeeping rule and filter-network feature of the path
(SURVEY A.5 / A.9): HIP engine through the C ABI vs the CPU oracle, same seeded inputs.
Tolerances: f32 1e-5, f64 1e-12 relative RMS on float outputs; integer outputs within 1 LSB;
overflow counters equal."""
import numpy as np
import pytest

import bforacle as bo
import cases

pytestmark = pytest.mark.gpu
TOL = {4: 1e-5, 8: 1e-12}
FLOATFMT = {4: "FLOAT_LE", 8: "FLOAT64_LE"}


def _ir(seed, taps, n_in=1):
    return cases.make_ir(np.random.default_rng(seed), taps, n_in)


def _compare(hip, spec, n_blocks, seed=1, control=None, tol=None):
    ge = cases.build(hip.Engine, spec)
    oe = cases.build(bo.Engine, spec)
    blocks = cases.raw_blocks(seed, n_blocks, spec["L"], spec["n_in"], spec["infmt"])
    worst = 0.0
    for b, blk in enumerate(blocks):
        if control:
            control(b, ge)
            control(b, oe)
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_, b
        gsamp, osamp = cases.samples(g, spec["outfmt"]), cases.samples(o, spec["outfmt"])
        if spec["outfmt"].startswith("FLOAT"):
            if np.abs(osamp).max() == 0:
                assert np.abs(gsamp).max() <= 1e-30, b
            else:
                err = cases.rel_rms(gsamp, osamp)
                worst = max(worst, err)
                assert err <= (tol or TOL[spec["rs"]]), (b, err)
        else:
            assert np.abs(gsamp - osamp).max() <= 1.0, b
    for ch in range(spec["n_out"]):
        g, o = ge.overflow(ch), oe.overflow(ch)
        assert g.n_overflows == o.n_overflows and g.max == o.max, ch
        assert abs(g.intlargest - o.intlargest) <= 1
    return worst


def _spec(L, N, rs, n_in, n_out, filters, coeffs=(), infmt="S24_4LE", outfmt=None, **kw):
    return dict(L=L, N=N, rs=rs, n_in=n_in, n_out=n_out, infmt=infmt,
                outfmt=outfmt or FLOATFMT[rs], coeffs=list(coeffs), filters=filters, **kw)


@pytest.mark.parametrize("rs", [4, 8])
def test_crossbar_both_precisions_nonpow2_partitions(hip, rs):
    """N = 13 (bench4_config: 4096 x 13): `blockcounter % 13` and the warm-up guard"""
    L, N, I, O = 256, 13, 3, 2
    coeffs = [(_ir(10 + k, L * N, I), 1.0, 0) for k in range(I * O)]
    filters = [dict(in_ch=[i], out_ch=[o], coeff=o * I + i) for o in range(O) for i in range(I)]
    _compare(hip, _spec(L, N, rs, I, O, filters, coeffs), 2 * N + 2)


@pytest.mark.parametrize("rs", [4, 8])
def test_delay_dirac_scales_short_sets_multi_output(hip, rs):
    """delayblocks (incl. clamp and cblocks truncation), coeff -1, polarity / attenuation
    scales, `blocks: 2` sets, one filter feeding two outputs, a filter-less output"""
    L, N = 128, 6
    coeffs = [(_ir(1, L * N), 1.0, 0), (_ir(2, L * 2), 0.5, 2), (_ir(3, L * N), 1.0, 0)]
    filters = [
        dict(in_ch=[0], out_ch=[0, 1], out_scale=[1.0, -0.5], coeff=0, delayblocks=2),
        dict(in_ch=[1], out_ch=[1], in_scale=[-1.0], coeff=1),
        dict(in_ch=[1], out_ch=[0], coeff=-1, delayblocks=1),
        dict(in_ch=[2], out_ch=[2], coeff=2, delayblocks=9),        # clamps to N-1
        dict(in_ch=[0], out_ch=[2], coeff=0, in_scale=[0.25]),      # same input+output pair again
        dict(in_ch=[0], out_ch=[2], coeff=2),
    ]
    _compare(hip, _spec(L, N, rs, 3, 4, filters, coeffs), 2 * N + 1)


@pytest.mark.parametrize("rs", [4, 8])
def test_multi_input_mix_filters(hip, rs):
    """bench4-style filters that mix several inputs with their own scales (A3)"""
    L, N = 128, 4
    coeffs = [(_ir(5, L * N, 3), 1.0, 0), (_ir(6, L * N, 3), 1.0, 0)]
    filters = [
        dict(in_ch=[0, 1, 2], in_scale=[1.0, -0.5, 0.25], out_ch=[0], coeff=0),
        dict(in_ch=[2, 0], in_scale=[0.7, 0.3], out_ch=[1, 0], out_scale=[1.0, 0.5], coeff=1, delayblocks=1),
        dict(in_ch=[1], out_ch=[1], coeff=0),
    ]
    _compare(hip, _spec(L, N, rs, 3, 2, filters, coeffs), 2 * N + 1)


@pytest.mark.parametrize("rs", [4, 8])
def test_cascaded_filters_bench1_topology(hip, rs):
    """from_filters / convolve_eval (A8): two input filters feed two output filters, plus a
    third level and a filter mixing a channel input with a filter input"""
    L, N = 128, 4
    coeffs = [(_ir(20 + k, L * N, 2), 1.0, 0) for k in range(5)]
    filters = [
        dict(in_ch=[0], coeff=0),                                            # 0
        dict(in_ch=[1], coeff=1, out_ch=[2]),                                # 1: also to an output
        dict(in_f=[0, 1], in_fscale=[1.0, 0.5], out_ch=[0], coeff=2),        # 2
        dict(in_f=[0, 1], in_fscale=[-1.0, 1.0], out_ch=[1], coeff=3),       # 3
        dict(in_ch=[0], in_scale=[0.5], in_f=[2], out_ch=[2], coeff=4, delayblocks=1),   # 4: level 2
    ]
    tol = 3e-5 if rs == 4 else None      # three FFT round trips deep: error accumulates
    _compare(hip, _spec(L, N, rs, 2, 3, filters, coeffs), 3 * N, tol=tol)


@pytest.mark.parametrize("rs", [4, 8])
def test_coefficient_switch_with_and_without_crossfade(hip, rs):
    """fctrl.coeff changes at run time (bflogic_cli `cfc`): hard switch, cross-faded switch,
    switch to and from the dirac (coeff -1), bench5-style toggling every block"""
    L, N = 128, 4
    coeffs = [(_ir(30, L * N), 1.0, 0), (_ir(31, L * N), 1.0, 0), (_ir(32, L * 2), 1.0, 2)]
    filters = [
        dict(in_ch=[0], out_ch=[0], coeff=0, crossfade=True),
        dict(in_ch=[1], out_ch=[1], coeff=1, crossfade=False),
        dict(in_ch=[0, 1], out_ch=[2], coeff=0, crossfade=True),
    ]
    plan = {3: [(0, 1), (1, 0)], 5: [(0, -1), (2, 2)], 6: [(0, 2)], 7: [(0, 0), (2, -1)],
            8: [(0, 1)], 9: [(0, 0)], 10: [(0, 1)]}

    def control(b, eng):
        for f, c in plan.get(b, []):
            eng.set_coeff(f, c)
    _compare(hip, _spec(L, N, rs, 2, 3, filters, coeffs), 14, control=control)


def test_runtime_coefficient_update_like_bflogic_eq(hip):
    """convolver_runtime_coeffs2cbuf: one partition of a live coefficient set is replaced"""
    L, N = 256, 4
    spec = _spec(L, N, 4, 1, 1, [dict(in_ch=[0], out_ch=[0], coeff=0)], [(_ir(40, L * N), 1.0, 0)])
    ge = cases.build(hip.Engine, spec)
    taps = _ir(40, L * N).copy()
    new = _ir(41, L)
    for b, blk in enumerate(cases.raw_blocks(2, 10, L, 1, "S24_4LE")):
        if b == 4:
            ge.update_coeff_block(0, 2, new)
            taps[2 * L:3 * L] = new
            oe = cases.build(bo.Engine, dict(spec, coeffs=[(taps, 1.0, 0)]))
            # replay history into a fresh oracle that has the updated set from the start:
            # from block 4 + N on the outputs must agree again
            for old in cases.raw_blocks(2, 4, L, 1, "S24_4LE"):
                oe.block(old)
        _, g = ge.block(blk)
        if b >= 4:
            _, o = oe.block(blk)
            if b >= 4 + N:
                assert cases.rel_rms(cases.samples(g, "FLOAT_LE"), cases.samples(o, "FLOAT_LE")) <= 1e-5


ALL_FORMATS = ["S8", "S16_LE", "S16_BE", "S24_LE", "S24_BE", "S24_4LE", "S24_4BE", "S32_LE",
               "S32_BE", "FLOAT_LE", "FLOAT_BE", "FLOAT64_LE", "FLOAT64_BE"]


@pytest.mark.parametrize("fmt", ALL_FORMATS)
def test_every_sample_format_in_and_out(hip, fmt):
    """dirac filter: raw -> real -> FFT -> IFFT -> raw for all 13 formats, interleaved 3 ch"""
    L, N, ch = 64, 2, 3
    nbytes, sbytes, isfloat, le = bo.SAMPLE_FORMATS[fmt]
    rng = np.random.default_rng(50)
    ge = hip.Engine(L, N, 8, ch, ch)
    oe = bo.Engine(L, N, 8, ch, ch)
    for e in (ge, oe):
        e.set_interleaved(0, fmt)
        e.set_interleaved(1, fmt)
        for c in range(ch):
            e.add_filter(in_ch=[c], out_ch=[c], coeff=-1, in_scale=[0.5])
    ge.finalize()
    for _ in range(3):
        if isfloat:
            vals = (rng.standard_normal(L * ch) * 0.5).astype(np.float32 if nbytes == 4 else np.float64)
            raw = vals.view(np.uint8).copy()
            if not le:
                raw = raw.reshape(-1, nbytes)[:, ::-1].copy().ravel()
        else:
            raw = rng.integers(0, 256, L * ch * nbytes, dtype=np.uint8)
        gs, g = ge.block(raw)
        os_, o = oe.block(raw)
        assert gs == os_ == 0
        if isfloat:
            gv = g.reshape(-1, nbytes)[:, ::(1 if le else -1)].copy().view(np.float32 if nbytes == 4 else np.float64)
            ov = o.reshape(-1, nbytes)[:, ::(1 if le else -1)].copy().view(np.float32 if nbytes == 4 else np.float64)
            assert np.abs(gv - ov).max() <= 1e-6
        else:
            # identical up to one LSB of the integer format (f64 path: rounding only)
            dec = bo.Ctx(64, 8)
            gi = dec.raw2real(g, nbytes, 0, 1, 0 if le else 1, L * ch)
            oi = dec.raw2real(o, nbytes, 0, 1, 0 if le else 1, L * ch)
            assert np.abs(gi - oi).max() <= 1.0
    for c in range(ch):
        assert ge.overflow(c).n_overflows == oe.overflow(c).n_overflows


def test_clipping_counters_bit_exact(hip):
    """S16 output driven into clipping: n_overflows / intlargest / largest as the reference's
    requantiser counts them (dither_funs.h:71-114)"""
    L = 256
    spec = _spec(L, 1, 8, 2, 2, [dict(in_ch=[0], out_ch=[0], coeff=-1, out_scale=[3.0]),
                                  dict(in_ch=[1], out_ch=[1], coeff=-1)],
                 infmt="FLOAT64_LE", outfmt="S16_LE")
    ge, oe = cases.build(hip.Engine, spec), cases.build(bo.Engine, spec)
    rng = np.random.default_rng(3)
    for _ in range(4):
        x = rng.standard_normal((L, 2)) * 0.5
        gs, g = ge.block(x)
        os_, o = oe.block(x)
        assert gs == os_ == 0
        assert np.abs(g.view(np.int16).astype(int) - o.view(np.int16).astype(int)).max() <= 1
    for ch in range(2):
        g, o = ge.overflow(ch), oe.overflow(ch)
        assert o.n_overflows > 0 if ch == 0 else True
        assert g.n_overflows == o.n_overflows
        assert g.intlargest == o.intlargest
        assert g.largest == pytest.approx(o.largest, rel=1e-12)
    ge.reset_overflow()
    assert ge.overflow(0).astuple() == (0, 0, 0.0, 32767.0)


def test_nan_and_safety_limit_status(hip):
    L = 64
    spec = _spec(L, 1, 4, 1, 1, [dict(in_ch=[0], out_ch=[0], coeff=-1)], infmt="FLOAT_LE",
                 safety_limit=2.0)
    ge, oe = cases.build(hip.Engine, spec), cases.build(bo.Engine, spec)
    x = np.zeros((L, 1), np.float32)
    assert ge.block(x)[0] == oe.block(x)[0] == 0
    x[5] = 3.0
    assert ge.block(x)[0] == oe.block(x)[0] == hip.ST_SAFETY
    x[5] = np.nan
    assert ge.block(x)[0] & hip.ST_NONFINITE and oe.block(x)[0] == 1
    x[5] = 0.0
    # status is per block: the engine keeps running afterwards (the host decides to exit)
    ge.block(x)


def test_bad_configuration_is_rejected(hip):
    with pytest.raises(hip.BfhipError, match="Invalid length"):
        hip.Engine(100, 4, 4, 1, 1)
    e = hip.Engine(64, 2, 4, 1, 1)
    with pytest.raises(hip.BfhipError, match="NaN or Inf"):
        e.add_coeff(np.array([1.0, np.inf]))
    with pytest.raises(hip.BfhipError):
        e.add_coeff(np.zeros(64 * 3), n_blocks=3)       # more blocks than the engine has
    with pytest.raises(hip.BfhipError):
        e.add_filter(in_ch=[0], out_ch=[0], coeff=7)    # coefficient set not loaded
    with pytest.raises(hip.BfhipError):
        e.add_filter(in_f=[3], out_ch=[0])              # from_filter not defined yet
    with pytest.raises(hip.BfhipError):
        e.block(np.zeros(64, np.float32))               # not finalized


def test_hp_tpdf_dither_bit_exact_f64(hip):
    """HP-TPDF dither to S16 (dither_funs.h:7-69): integer-valued samples through a dirac filter
    in f64 reach the requantiser exact to 1e-12, far from every rounding boundary (those sit at
    odd multiples of 1/510), so the device must reproduce the oracle -- which matches the
    reference's code bit for bit (tests/test_oracle_golden.py) -- exactly: samples, error
    feedback across blocks, table walk incl. a wrap, overflow struct."""
    L, N, ch = 256, 2, 3
    spec = _spec(L, N, 8, ch, ch,
                 [dict(in_ch=[c], out_ch=[c], coeff=-1, out_scale=[1.0 / 32768.0]) for c in range(ch)],
                 infmt="FLOAT64_LE", outfmt="S16_LE")

    def mk(cls):
        e = cls(L, N, 8, ch, ch)
        e.set_interleaved(0, "FLOAT64_LE")
        e.set_interleaved(1, "S16_LE")
        e.enable_dither([0, 2], 300)         # table 2*3000+1 bytes: wraps after ~11 blocks
        for f in spec["filters"]:
            e.add_filter(**f)
        if hasattr(e, "finalize"):
            e.finalize()
        return e
    ge, oe = mk(hip.Engine), mk(bo.Engine)
    rng = np.random.default_rng(77)
    for b in range(30):
        x = np.round(rng.standard_normal((L, ch)) * 9000.0)
        x[5, 0] = 40000.0
        x[9, 2] = -50000.0
        gs, g = ge.block(x)
        os_, o = oe.block(x)
        assert gs == os_ == 0
        assert np.array_equal(g.view(np.int16), o.view(np.int16)), b
    for c in range(ch):
        g, o = ge.overflow(c), oe.overflow(c)
        assert (g.n_overflows, g.intlargest, g.max) == (o.n_overflows, o.intlargest, o.max), c
        assert g.largest == pytest.approx(o.largest, rel=1e-12)     # a float: FFT rounding


def test_hp_tpdf_dither_f32_within_one_lsb(hip):
    L, N, ch = 1024, 4, 2
    coeffs = [(_ir(60 + c, L * N), 1.0, 0) for c in range(ch)]

    def mk(cls):
        e = cls(L, N, 4, ch, ch)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, "S16_LE")
        e.enable_dither([0, 1], 44100)
        for t, s, nb in coeffs:
            e.add_coeff(t, s, nb)
        for c in range(ch):
            e.add_filter(in_ch=[c], out_ch=[c], coeff=c)
        if hasattr(e, "finalize"):
            e.finalize()
        return e
    ge, oe = mk(hip.Engine), mk(bo.Engine)
    for blk in cases.raw_blocks(9, 8, L, ch, "S24_4LE", amplitude=0.3):
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0
        d = np.abs(g.view(np.int16).astype(int) - o.view(np.int16).astype(int))
        assert d.max() <= 2 and (d > 0).mean() < 0.05


@pytest.mark.parametrize("rs", [4, 8])
def test_runtime_scale_and_delay_changes_keep_history_semantics(hip, rs):
    """bflogic_cli changing a filter's input attenuation / delay while it runs: the reference
    scales and delays a block when it ENTERS the filter's ring, so blocks already in the ring
    keep their old scale and slot (bfrun.c:1600,1651-1656).  Output scale changes act at once."""
    L, N = 128, 6
    coeffs = [(_ir(70, L * N), 1.0, 0), (_ir(71, L * N), 1.0, 0)]
    filters = [
        dict(in_ch=[0], out_ch=[0], coeff=0),
        dict(in_ch=[1], out_ch=[1], coeff=1, delayblocks=1),
        dict(in_ch=[0], out_ch=[1], coeff=1, in_scale=[0.5]),
    ]

    def control(b, eng):
        if b == 4:
            eng.set_scale(0, 0, 0, 0.25)        # input attenuation of filter 0
        if b == 6:
            eng.set_delayblocks(1, 3)           # more delay: leaves stale slots behind
            eng.set_scale(2, 1, 0, -2.0)        # output scale of filter 2
        if b == 9:
            eng.set_delayblocks(1, 0)
            eng.set_scale(0, 0, 0, 1.0)
        if b == 10:
            eng.set_scale(2, 0, 0, 1.5)
    _compare(hip, _spec(L, N, rs, 2, 2, filters, coeffs), 20, control=control)


@pytest.mark.parametrize("rs", [4, 8])
def test_processed_coefficient_format_round_trip(hip, rs):
    """`format: "processed"` / shared-memory coefficient sets: cbufs in the reference's internal
    layout go in (here produced by the oracle's coeffs2cbuf, which is the reference's layout),
    filter like the same taps loaded the normal way, and come back out in that layout"""
    L, N = 256, 4
    dt = np.float32 if rs == 4 else np.float64
    taps = _ir(90, L * 3).astype(dt)
    oc = bo.Ctx(L, rs)
    cbufs = np.stack([oc.coeffs2cbuf(taps[b * L:(b + 1) * L], 0.5) for b in range(3)])
    e1 = hip.Engine(L, N, rs, 1, 1)
    e2 = hip.Engine(L, N, rs, 1, 1)
    for e in (e1, e2):
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, FLOATFMT[rs])
    c1 = e1.add_coeff_processed(cbufs)
    c2 = e2.add_coeff(taps, 0.5, 3)
    for e, c in ((e1, c1), (e2, c2)):
        e.add_filter(in_ch=[0], out_ch=[0], coeff=c)
        e.finalize()
    tol = 3e-6 if rs == 4 else 1e-13
    back = e2.read_coeff_processed(c2, 3)
    assert np.abs(back - cbufs).max() <= tol * np.abs(cbufs).max() * 10
    assert np.array_equal(e1.read_coeff_processed(c1, 3), cbufs)        # pure permutation
    for blk in cases.raw_blocks(3, 8, L, 1, "S24_4LE"):
        _, a = e1.block(blk)
        _, b = e2.block(blk)
        assert cases.rel_rms(cases.samples(a, FLOATFMT[rs]), cases.samples(b, FLOATFMT[rs])) <= TOL[rs]
    bad = cbufs.copy()
    bad[1, 7] = np.nan
    with pytest.raises(hip.BfhipError, match="NaN or Inf"):
        hip.Engine(L, N, rs, 1, 1).add_coeff_processed(bad)


@pytest.mark.parametrize("rs", [4, 8])
def test_virtual_channels_delay_mute_and_mix(hip, rs):
    """bench4_config's `mapping`: virtual inputs that share a physical input (each with its own
    integer delay / mute) and virtual outputs mixed into one physical output (bfrun.c:1509-1531,
    1938-2003; delay.c).  Delays change at run time across the short / long buffer regimes."""
    L, N = 128, 4
    coeffs = [(_ir(80 + k, L * N, 2), 1.0, 0) for k in range(4)]
    filters = [dict(in_ch=[0], out_ch=[0], coeff=0), dict(in_ch=[1], out_ch=[1], coeff=1),
               dict(in_ch=[2], out_ch=[2], coeff=2), dict(in_ch=[3], out_ch=[3], coeff=3),
               dict(in_ch=[1, 3], out_ch=[0], coeff=0, in_scale=[0.5, -0.5])]
    ifmt, ofmt = "S16_LE", FLOATFMT[rs]

    def mk(cls):
        e = cls(L, N, rs, 4, 4)
        e.map_channels(0, [0, 0, 0, 1])          # three virtual inputs from physical input 0
        e.map_channels(1, [0, 1, 1, 2])          # virtual outputs 1,2 mix into physical output 1
        e.set_interleaved_phys(0, ifmt, 2)
        e.set_interleaved_phys(1, ofmt, 3)
        for t, s, nb in coeffs:
            e.add_coeff(t, s, nb)
        for v, (d, md) in enumerate([(0, 300), (17, 300), (200, -1), (0, 0)]):
            e.set_delay(0, v, d); e.set_maxdelay(0, v, md)
        for v, (d, md) in enumerate([(0, 0), (5, 400), (130, 400), (0, 0)]):
            e.set_delay(1, v, d); e.set_maxdelay(1, v, md)
        for f in filters:
            e.add_filter(**f)
        if hasattr(e, "finalize"):
            e.finalize()
        return e
    ge, oe = mk(hip.Engine), mk(bo.Engine)
    plan = {2: [(0, 1, 40)], 3: [(1, 1, 0)], 4: [(0, 0, 129), (1, 2, 7)], 6: [(0, 1, 290)],
            8: [(0, 0, 3), (1, 1, 399)], 9: [(0, 1, 17)], 11: [(1, 2, 128)]}
    mutes = {3: [(0, 1, 1)], 5: [(0, 1, 0), (1, 2, 1)], 7: [(1, 2, 0), (1, 1, 1)], 10: [(1, 1, 0)]}
    blocks = cases.raw_blocks(21, 14, L, 2, ifmt, amplitude=0.2)
    for b, blk in enumerate(blocks):
        for eng in (ge, oe):
            for io, v, d in plan.get(b, []):
                eng.set_delay(io, v, d)
            for io, v, m in mutes.get(b, []):
                eng.set_mute(io, v, m)
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0
        gsamp, osamp = cases.samples(g, ofmt), cases.samples(o, ofmt)
        if np.abs(osamp).max() > 0:
            assert cases.rel_rms(gsamp, osamp) <= TOL[rs], b
    for ch in range(4):
        g, o = ge.overflow(ch), oe.overflow(ch)
        assert (g.n_overflows, g.max) == (o.n_overflows, o.max)
        assert g.largest == pytest.approx(o.largest, rel=1e-5)


@pytest.mark.parametrize("rs", [4, 8])
def test_subsample_delay_inputs_and_outputs(hip, rs):
    """`sdf_length` + per-channel `subdelay:` (delay.c:416-505): filtered 1:1 inputs and outputs,
    a filtered member of an N:1 output group, an unfiltered member that gets the integer
    sdf_length compensation, and a run-time change of the sub-sample value (incl. out of
    range = filter bypassed for that block).  The reference evaluates the interpolator with a
    small FFT overlap-save, the device as a direct FIR: same numbers to rounding."""
    L, N, half = 256, 4, 15
    coeffs = [(_ir(95 + k, L * N, 1), 1.0, 0) for k in range(3)]
    filters = [dict(in_ch=[0], out_ch=[0], coeff=0), dict(in_ch=[1], out_ch=[1], coeff=1),
               dict(in_ch=[2], out_ch=[2], coeff=2), dict(in_ch=[0], out_ch=[2], coeff=-1, in_scale=[0.25])]
    ofmt = FLOATFMT[rs]

    def mk(cls):
        e = cls(L, N, rs, 3, 3)
        e.map_channels(1, [0, 1, 1])             # virtual outputs 1, 2 share physical output 1
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved_phys(1, ofmt, 2)
        e.enable_subdelay(half)
        e.set_subdelay(0, 0, 37)                 # input 0 filtered, input 1 and 2 not
        e.set_subdelay(0, 2, 0)
        e.set_subdelay(1, 0, -60)                # 1:1 output with a filter
        e.set_subdelay(1, 1, 12)                 # member of the shared output with a filter
        for v, d in enumerate([0, 3, 140]):      # output 2 (no filter) gets +sdf_length inside
            e.set_delay(1, v, d); e.set_maxdelay(1, v, 300)
        for t, s_, nb in coeffs:
            e.add_coeff(t, s_, nb)
        for f in filters:
            e.add_filter(**f)
        if hasattr(e, "finalize"):
            e.finalize()
        return e
    ge, oe = mk(hip.Engine), mk(bo.Engine)
    changes = {3: [(0, 0, -99)], 5: [(1, 0, 150)], 6: [(1, 0, 5)], 8: [(1, 1, -1), (0, 2, 99)]}
    tol = 2e-5 if rs == 4 else 1e-11             # 31-tap FIR, FFT vs direct summation order
    for b, blk in enumerate(cases.raw_blocks(33, 12, L, 3, "S24_4LE", amplitude=0.2)):
        for eng in (ge, oe):
            for io, v, sd in changes.get(b, []):
                eng.set_subdelay(io, v, sd)
            if b == 4:
                # output 0 is a 1:1 channel that only passes through the shared-output pass for its
                # sub-sample filter: its mute and delay stay dai.c's business (bfrun.c:1926-1936) --
                # the engine once muted it here (found by tests/test_gpu_refloop.py's soak)
                eng.set_mute(1, 0, 1)
                eng.set_delay(1, 0, 77)
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0
        gsamp, osamp = cases.samples(g, ofmt), cases.samples(o, ofmt)
        assert cases.rel_rms(gsamp, osamp) <= tol, (b, cases.rel_rms(gsamp, osamp))


def test_fixed_delays_beside_a_subsample_filter_stay_inside_their_buffers(hip):
    """maxdelay -1 (the reference's default: "cannot be changed") on a channel that shares a physical
    one, beside a sub-sample filter on the same side.  The reference adds sdf_length to delay AND
    maxdelay (bfrun.c:1152-1162, 1185-1197): the limit becomes sdf_length - 1, below the delay, and
    delay.c:357-374 allocates for the limit and fills for the delay -- a heap overrun in the reference
    (tests/test_gpu_refloop.py met it as "corrupted double-linked list"; the engine's mirror of that
    arithmetic was a memory fault on the GPU).  Defined here: the delay is what was asked
    for plus sdf_length and cannot be changed; a delay above a positive limit starts at the limit."""
    L, N, half = 64, 2, 15
    ofmt = "FLOAT_LE"

    def mk(cls):
        e = cls(L, N, 4, 3, 3)
        e.map_channels(0, [0, 0, 1])
        e.map_channels(1, [0, 0, 1])
        e.set_interleaved_phys(0, "FLOAT_LE", 2)
        e.set_interleaved_phys(1, ofmt, 2)
        e.enable_subdelay(half)
        e.set_subdelay(0, 0, 0)                  # virtual input 0 filtered: input 1 gets sdf_length more
        e.set_subdelay(1, 0, 0)                  # the same on the output side
        for io in range(2):
            e.set_maxdelay(io, 1, -1); e.set_delay(io, 1, 700 if io == 0 else 20)
            e.set_maxdelay(io, 0, 40); e.set_delay(io, 0, 90)          # above its limit: starts at 40
        e.add_filter(in_ch=[1], out_ch=[1], coeff=-1)
        e.add_filter(in_ch=[0], out_ch=[0], coeff=-1, in_scale=[0.0])
        e.add_filter(in_ch=[2], out_ch=[2], coeff=-1)
        if hasattr(e, "finalize"):
            e.finalize()
        return e
    ge, oe = mk(hip.Engine), mk(bo.Engine)
    x = np.zeros((16 * L, 2), np.float32)
    x[5, 0] = 1.0
    got = []
    for b in range(16):
        if b == 3:
            for eng in (ge, oe):
                eng.set_delay(0, 1, 10)          # refused: the delay is fixed
        gs, g = ge.block(x[b * L:(b + 1) * L])
        os_, o = oe.block(x[b * L:(b + 1) * L])
        assert gs == os_ == 0
        assert np.abs(cases.samples(g, ofmt) - cases.samples(o, ofmt)).max() <= 2e-5, b        # (an impulse of 1.0)
        got.append(cases.samples(g, ofmt).reshape(L, 2)[:, 0])
    y = np.concatenate(got)
    assert int(np.argmax(np.abs(y))) == 5 + (700 + half) + (20 + half) and abs(y.max() - 1.0) < 1e-5


def test_output_spectrum_reader_with_many_partial_sums(hip, monkeypatch):
    """bfhip_engine_read_output_spectrum (debug / multi-GPU rehearsals): with more than two chunks of
    partial sums the engine adds them up in place before the output pass -- the reader must not add
    the other chunks on top again"""
    L, N, I, O = 1024, 8, 2, 2
    blocks = cases.raw_blocks(4, N + 3, L, I, "S24_4LE")

    def run(target):
        monkeypatch.setenv("BFHIP_MAC_TARGET_WGS", str(target))
        e, _ = cases.crossbar(hip.Engine, L, N, 4, I, O, "S24_4LE", "FLOAT_LE")
        for blk in blocks:
            e.block(blk)
        return np.stack([e.output_spectrum(o) for o in range(O)])
    few, many = run(2), run(64)              # 2 tiles x 1 chunk, 2 tiles x 32 chunks
    scale = float(np.abs(few).max())
    assert scale > 0 and float(np.abs(few - many).max()) <= 1e-5 * scale


def test_prewarm_changes_cost_not_results(hip):
    """bfhip_engine_prewarm: rings declared full of silence; outputs identical to a cold start,
    including for a ring depth that is not a power of two (unsigned wrap of blockcounter - p)"""
    L, N, I, O = 128, 13, 2, 3
    coeffs = [(_ir(90 + k, L * N, I), 1.0, 0) for k in range(I * O)]
    filters = [dict(in_ch=[i], out_ch=[o], coeff=o * I + i, delayblocks=(o + 2 * i) % 5) for o in range(O) for i in range(I)]
    spec = _spec(L, N, 4, I, O, filters, coeffs, outfmt="S32_LE")
    a, b = cases.build(hip.Engine, spec), cases.build(hip.Engine, spec)
    b.prewarm()
    for k, blk in enumerate(cases.raw_blocks(4, 2 * N + 3, L, I, spec["infmt"])):
        sa, ra = a.block(blk)
        sb, rb = b.block(blk)
        assert sa == sb and np.array_equal(ra, rb), k
    with pytest.raises(hip.BfhipError, match="already"):
        b.prewarm()


@pytest.mark.parametrize("L", [128, 16384])
@pytest.mark.parametrize("rs", [4, 8])
@pytest.mark.parametrize("mode", ["exact", "floor"])
def test_powersave(hip, rs, mode, L):
    """`powersave: true` (exact-zero windows) changes no sample; `powersave: <dB>` makes windows
    below the noise floor silence (test_silent, bfrun.c:721-771).  Inputs go silent, come back,
    hover around the floor; one input never carries anything (its MAC entries are skipped)."""
    N, I, O = 4, 3, 2
    coeffs = [(_ir(300 + k, L * N, I), 1.0, 0) for k in range(I * O)]
    filters = [dict(in_ch=[i], out_ch=[o], coeff=o * I + i, delayblocks=i % 2) for o in range(O) for i in range(I)]
    filters.append(dict(in_ch=[0, 1], in_scale=[0.5, 0.5], out_ch=[0], coeff=0))        # a private-ring filter
    spec = _spec(L, N, rs, I, O, filters, coeffs, infmt="S16_LE")
    thr = 1.0 if mode == "exact" else 10 ** (-50 / 20)

    class PS:
        def __init__(self, cls):
            self.cls = cls

        def __call__(self, *a, **k):
            e = self.cls(*a, **k)
            e.set_powersave(thr)
            return e
    ge, oe = cases.build(PS(hip.Engine), spec), cases.build(PS(bo.Engine), spec)
    plain = cases.build(bo.Engine, spec)
    rng = np.random.default_rng(5)
    amps = [3000, 3000, 0, 0, 0, 0, 0, 0, 40, 40, 200, 0, 3000, 0, 0, 0, 0, 0, 0, 3000]
    for b, a in enumerate(amps):
        x = np.zeros((L, I), np.int16)
        x[:, 0] = (rng.standard_normal(L) * a).astype(np.int16)
        x[:, 1] = (rng.standard_normal(L) * (a // 2)).astype(np.int16)          # input 2 stays silent
        gs, g = ge.block(x)
        os_, o = oe.block(x)
        ps_, p = plain.block(x)
        assert gs == os_ == 0
        gsamp, osamp, psamp = (cases.samples(v, spec["outfmt"]) for v in (g, o, p))
        if mode == "exact":
            assert np.array_equal(osamp, psamp), b                  # the oracle: transparent
        scale = max(np.abs(osamp).max(), 1e-3)
        assert np.abs(gsamp - osamp).max() <= (1e-5 if rs == 4 else 1e-12) * scale * 4, b


@pytest.mark.parametrize("rs", [4, 8])
def test_dither_on_shared_and_subsample_filtered_outputs(hip, rs):
    """the mixed samples of an N:1 output group, and the filtered samples of an output with a
    sub-sample delay, go through the HP-TPDF quantiser like any other dithered output
    (bfrun.c:1926-1936, 1992-1997: `bfconf->dither_state[physch]`); float64: bit exact"""
    L, N = 256, 3
    coeffs = [(_ir(400 + k, L * N, 2), 1.0, 0) for k in range(4)]
    filters = [dict(in_ch=[0], out_ch=[0], coeff=0), dict(in_ch=[1], out_ch=[1], coeff=1),
               dict(in_ch=[0], out_ch=[2], coeff=2, out_scale=[0.5]), dict(in_ch=[1], out_ch=[3], coeff=3)]

    def mk(cls):
        e = cls(L, N, rs, 2, 4)
        e.map_channels(1, [0, 1, 0, 2])             # virtual 0 and 2 share physical 0 (not adjacent)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved_phys(1, "S16_LE", 3)
        e.enable_subdelay(15)
        e.set_subdelay(1, 1, 33)                    # physical 1: a 1:1 output with a sub-sample filter
        for v, d in enumerate([0, 0, 70, 0]):
            e.set_delay(1, v, d); e.set_maxdelay(1, v, 200)
        for t, s_, nb in coeffs:
            e.add_coeff(t, s_, nb)
        e.enable_dither([0, 1, 2], 44100, 0)        # all three physical outputs
        for f in filters:
            e.add_filter(**f)
        if hasattr(e, "finalize"):
            e.finalize()
        return e
    ge, oe = mk(hip.Engine), mk(bo.Engine)
    for b, blk in enumerate(cases.raw_blocks(77, 10, L, 2, "S24_4LE", amplitude=0.3)):
        if b == 5:
            for eng in (ge, oe):
                eng.set_mute(1, 2, 1)
                eng.set_delay(1, 2, 10)
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0
        gq = np.frombuffer(g.tobytes(), np.int16).reshape(L, 3).astype(np.int64)
        oq = np.frombuffer(o.tobytes(), np.int16).reshape(L, 3).astype(np.int64)
        d = np.abs(gq - oq).max(axis=0)
        if rs == 8:
            assert d[0] == 0 and d[2] == 0, (b, d)          # mix and plain: exact
            assert d[1] <= 1, (b, d)                        # behind the FIR (FFT vs direct sum): 1 LSB
        else:
            assert d.max() <= 3, (b, d)
    for ch in range(4):
        a, c = ge.overflow(ch), oe.overflow(ch)
        assert a.n_overflows == c.n_overflows and a.max == c.max
