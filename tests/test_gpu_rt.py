"""GPU: real-time (callback I/O) mode, bfhip_engine_rt_* (include/bfhip.h): pinned double
buffer + HIP-graph replay of a block's launch sequence.  The mode changes how a block is
launched, never what it computes: every test drives two identical HIP engines with the same
periods, one through bfhip_engine_block and one through the rt entry points, and wants the raw
output bytes, status bits and overflow structs to be IDENTICAL (parity of bfhip_engine_block
itself against the oracle is tests/test_gpu_features.py)."""
import numpy as np
import pytest

import cases
from test_gpu_features import _ir, _spec

pytestmark = pytest.mark.gpu


def _twin(hip, spec, n_blocks, flags=0, control=None, seed=3, pattern="block", dither=None):
    if dither:
        class _Dithered(hip.Engine):
            def finalize(self):
                self.enable_dither(*dither)
                super().finalize()
        a, b = cases.build(_Dithered, spec), cases.build(_Dithered, spec)
    else:
        a, b = cases.build(hip.Engine, spec), cases.build(hip.Engine, spec)
    b.rt_begin(flags)
    blocks = cases.raw_blocks(seed, n_blocks, spec["L"], spec["n_in"], spec["infmt"], amplitude=0.4)
    outs_a, outs_b = [], []
    for k, blk in enumerate(blocks):
        if control:
            control(k, a)
        outs_a.append(a.block(blk))
    if pattern == "block":
        for k, blk in enumerate(blocks):
            if control:
                control(k, b)
            outs_b.append(b.rt_block(blk))
    else:                       # two periods in flight: submit k+1 before collecting k
        b.rt_submit(blocks[0])
        for k in range(1, n_blocks):
            b.rt_submit(blocks[k])
            outs_b.append(b.rt_wait())
        outs_b.append(b.rt_wait())
    for k in range(n_blocks):
        assert outs_a[k][0] == outs_b[k][0], k
        assert np.array_equal(outs_a[k][1], outs_b[k][1]), k
    for ch in range(spec["n_out"]):
        oa, ob = a.overflow(ch), b.overflow(ch)
        assert oa.astuple() == ob.astuple(), ch
    st = b.rt_stats()
    b.rt_end()
    return st


@pytest.mark.parametrize("rs,flags", [(4, 0), (8, 0), (4, 1), (4, 4)])
def test_crossbar_replayed_from_graph_is_bit_identical(hip, rs, flags):
    L, N, I, O = 256, 13, 3, 4
    coeffs = [(_ir(10 + k, L * N, I), 1.0, 0) for k in range(I * O)]
    filters = [dict(in_ch=[i], out_ch=[o], coeff=o * I + i, delayblocks=(o + i) % 3) for o in range(O) for i in range(I)]
    st = _twin(hip, _spec(L, N, rs, I, O, filters, coeffs, outfmt="S24_4LE"), 3 * N + 2, flags=flags)
    assert st["direct"] == 1 and st["captures"] == 2 and st["graph"] == 3 * N + 1


def test_two_periods_in_flight(hip):
    L, N, I, O = 128, 8, 2, 2
    coeffs = [(_ir(40 + k, L * N, I), 1.0, 0) for k in range(I * O)]
    filters = [dict(in_ch=[i], out_ch=[o], coeff=o * I + i) for o in range(O) for i in range(I)]
    st = _twin(hip, _spec(L, N, 4, I, O, filters, coeffs, outfmt="S32_LE"), 20, pattern="pipelined")
    assert st["graph"] == 19
    # BFHIP_RT_OVERLAP: copies on their own streams beside compute; same bytes out
    for pattern in ("pipelined", "block"):
        st = _twin(hip, _spec(L, N, 4, I, O, filters, coeffs, outfmt="S32_LE"), 20, flags=hip.RT_OVERLAP, pattern=pattern)
        assert st["graph"] == 0 and st["direct"] == 20
    e = cases.build(hip.Engine, _spec(L, N, 4, I, O, filters, coeffs))
    e.rt_begin()
    blk = cases.raw_blocks(1, 1, L, I, "S24_4LE")[0]
    e.rt_submit(blk)
    e.rt_submit(blk)
    with pytest.raises(hip.BfhipError, match="in flight"):
        e.rt_submit(blk)
    e.rt_wait()
    e.rt_wait()
    with pytest.raises(hip.BfhipError, match="nothing in flight"):
        e.rt_wait()


@pytest.mark.parametrize("rs", [4, 8])
def test_cascade_and_mix_levels_replay(hip, rs):
    """per-filter kernels (ring_fill / mac_filter) take the block counter from device memory too"""
    L, N = 128, 4
    coeffs = [(_ir(20 + k, L * N, 2), 1.0, 0) for k in range(5)]
    filters = [
        dict(in_ch=[0], coeff=0),
        dict(in_ch=[1], coeff=1, out_ch=[2]),
        dict(in_f=[0, 1], in_fscale=[1.0, 0.5], out_ch=[0], coeff=2),
        dict(in_f=[0, 1], in_fscale=[-1.0, 1.0], out_ch=[1], coeff=3),
        dict(in_ch=[0, 1], in_scale=[0.5, 0.25], in_f=[2], out_ch=[2], coeff=4, delayblocks=1),
    ]
    st = _twin(hip, _spec(L, N, rs, 2, 3, filters, coeffs), 4 * N)
    assert st["graph"] == 4 * N - 1


def test_control_changes_rebuild_the_graph_and_fades_run_directly(hip):
    L, N = 128, 4
    coeffs = [(_ir(30, L * N), 1.0, 0), (_ir(31, L * N), 1.0, 0), (_ir(32, L * 2), 1.0, 2)]
    filters = [
        dict(in_ch=[0], out_ch=[0], coeff=0, crossfade=True),
        dict(in_ch=[1], out_ch=[1], coeff=1, crossfade=False),
        dict(in_ch=[0, 1], out_ch=[2], coeff=0, crossfade=True),
    ]
    plan = {3: [("coeff", 0, 1), ("coeff", 1, 0)], 5: [("coeff", 0, -1)], 8: [("scale", 1, 0.5)],
            9: [("delay", 1, 2)], 12: [("coeff", 2, 2)]}

    def control(k, eng):
        for what, f, v in plan.get(k, []):
            if what == "coeff":
                eng.set_coeff(f, v)
            elif what == "scale":
                eng.set_scale(f, 0, 0, v)
            else:
                eng.set_delayblocks(f, v)
    st = _twin(hip, _spec(L, N, 4, 2, 3, filters, coeffs), 20, control=control)
    assert st["graph"] >= 8 and st["direct"] >= 5 and st["captures"] >= 4


def test_dithered_output_state_survives_replay(hip):
    L, N = 256, 4
    coeffs = [(_ir(50, L * N), 1.0, 0), (_ir(51, L * N), 1.0, 0)]
    filters = [dict(in_ch=[0], out_ch=[0], coeff=0), dict(in_ch=[1], out_ch=[1], coeff=1)]
    st = _twin(hip, _spec(L, N, 8, 2, 2, filters, coeffs, outfmt="S16_LE"), 24, dither=([0, 1], 100, 0))
    assert st["graph"] == 23


def test_mixing_entry_points_keeps_the_block_counter_in_step(hip):
    L, N, I, O = 128, 5, 2, 2
    coeffs = [(_ir(60 + k, L * N, I), 1.0, 0) for k in range(I * O)]
    filters = [dict(in_ch=[i], out_ch=[o], coeff=o * I + i, delayblocks=i) for o in range(O) for i in range(I)]
    spec = _spec(L, N, 4, I, O, filters, coeffs)
    a, b = cases.build(hip.Engine, spec), cases.build(hip.Engine, spec)
    b.rt_begin()
    blocks = cases.raw_blocks(9, 18, L, I, spec["infmt"])
    for k, blk in enumerate(blocks):
        sa, ra = a.block(blk)
        sb, rb = b.block(blk) if k % 6 in (3, 4) else b.rt_block(blk)
        assert sa == sb and np.array_equal(ra, rb), k


def test_virtual_channels_fall_back_to_plain_launches(hip):
    """N:1 channels upload a job table per block: not replayable, still correct"""
    L, N = 128, 3
    mk = lambda cls: cls(L, N, 4, 3, 2)                 # noqa: E731
    engs = []
    for _ in range(2):
        e = mk(hip.Engine)
        e.map_channels(0, [0, 0, 1])
        e.set_interleaved_phys(0, "S16_LE", 2)
        e.set_interleaved(1, "FLOAT_LE")
        e.set_delay(0, 0, 37)
        e.set_maxdelay(0, 0, 300)
        for v in range(3):
            e.add_filter(in_ch=[v], out_ch=[v % 2], coeff=e.add_coeff(_ir(70 + v, L * N, 3)))
        e.finalize()
        engs.append(e)
    a, b = engs
    b.rt_begin()
    for k, blk in enumerate(cases.raw_blocks(2, 10, L, 2, "S16_LE")):
        if k == 5:
            a.set_delay(0, 0, 150)
            b.set_delay(0, 0, 150)
        sa, ra = a.block(blk)
        sb, rb = b.rt_block(blk)
        assert sa == sb and np.array_equal(ra, rb), k
    assert b.rt_stats()["graph"] == 0


@pytest.mark.parametrize("flags", [0, 2])
def test_peak_reset_by_the_host_reaches_the_device(hip, flags):
    """The reference reads icomm->overflow[ch] before every block and writes it back afterwards
    (bfrun.c:1929-1936): when another process resets the peaks (bf_reset_peak, the CLI's `upk`), the
    next block counts on from the reset values.  In real-time mode the structs live on the device
    between periods: bfhip_engine_rt_wait notices an entry the host changed since the engine last wrote
    it and makes it the device's state.  Same periods through bfhip_engine_block (which uploads the
    host's array with every call) must leave the same array after every period."""
    L, N, I, O = 256, 2, 2, 3
    coeffs = [(_ir(40 + k, L * N, I) * 30.0, 1.0, 0) for k in range(I * O)]      # loud: the 16-bit outputs clip now and then
    filters = [dict(in_ch=[i], out_ch=[o], coeff=o * I + i) for o in range(O) for i in range(I)]
    spec = _spec(L, N, 4, I, O, filters, coeffs, infmt="S24_4LE", outfmt="S16_LE")
    a, b = cases.build(hip.Engine, spec), cases.build(hip.Engine, spec)
    b.rt_begin(flags)
    oa, ob = (hip.Overflow * O)(), (hip.Overflow * O)()
    for ch in range(O):
        oa[ch] = a.overflow(ch)
        ob[ch] = b.overflow(ch)
    counted = False
    for k, blk in enumerate(cases.raw_blocks(9, 14, L, I, "S24_4LE", amplitude=0.9)):
        if k in (5, 9):                                  # "upk": counters and peaks back to zero, max stays
            for arr in (oa, ob):
                for ch in range(O):
                    if k == 9 and ch == 1:
                        continue                         # ... of some channels only
                    arr[ch].n_overflows, arr[ch].intlargest, arr[ch].largest = 0, 0, 0.0
        sa, ra = a.block(blk, overflow=oa)
        sb, rb = b.rt_block(blk, overflow=ob)
        assert sa == sb and np.array_equal(ra, rb), k
        for ch in range(O):
            got = (ob[ch].n_overflows, ob[ch].intlargest, ob[ch].largest, ob[ch].max)
            want = (oa[ch].n_overflows, oa[ch].intlargest, oa[ch].largest, oa[ch].max)
            if k in (5, 9) and not (k == 9 and ch == 1):
                # the period of the reset itself: the reference (and bfhip_engine_block) count this
                # block on top of the reset; the real-time path leaves the host's reset standing for
                # this one period and counts from the next
                assert got[:3] == (0, 0, 0.0), (k, ch, got)
                oa[ch].n_overflows, oa[ch].intlargest, oa[ch].largest = 0, 0, 0.0
                continue
            assert got == want, (k, ch, got, want)
            counted = counted or got[0] > 0
    assert counted
    b.rt_end()
