"""GPU: a LIVE engine handed bad arguments -- null buffers, indices out of range, calls in the wrong
state.  Every one must come back as an error code with a message (never a crash, never a kernel
launched on a null pointer: a faulting kernel can take the whole node down), and the engine must
still compute the right samples afterwards."""
import ctypes as C

import numpy as np
import pytest

import bforacle as bo
import cases

pytestmark = pytest.mark.gpu
L, N, I, O = 256, 2, 2, 2


def test_bad_arguments_are_errors_and_leave_the_engine_usable(hip):
    lib = hip.lib()
    null = C.c_void_p(0)
    e = hip.Engine(L, N, 4, I, O)
    h = e.h
    fmt = hip.make_format("S24_4LE", I, 0)

    def bad(rc, text=None):
        assert rc < 0, rc
        msg = lib.bfhip_last_error().decode()
        assert msg and (text is None or text in msg), msg

    # ---- before finalize
    bad(lib.bfhip_engine_set_format(h, 2, 0, C.byref(fmt)))
    bad(lib.bfhip_engine_set_format(h, 0, I, C.byref(fmt)))
    bad(lib.bfhip_engine_set_format(h, 0, 0, None))
    bad(lib.bfhip_engine_map_channels(h, 0, 0, (C.c_int * I)(0, 0)))
    bad(lib.bfhip_engine_map_channels(h, 0, 1, None))
    bad(lib.bfhip_engine_set_delay(h, 0, -1, 0))
    bad(lib.bfhip_engine_set_delay(h, 1, O, 0))
    bad(lib.bfhip_engine_set_delay(h, 0, 0, -5))
    bad(lib.bfhip_engine_set_subdelay(h, 0, I, 0))
    bad(lib.bfhip_engine_enable_subdelay(h, 0, 9.0), "half filter length")
    bad(lib.bfhip_engine_enable_dither(h, None, 1, 48000, 0))
    bad(lib.bfhip_engine_set_powersave(h, -1.0))
    bad(lib.bfhip_engine_block_dev(h, null, null), "not finalized")
    bad(lib.bfhip_engine_sync(h), "not finalized")
    bad(lib.bfhip_engine_flush(h), "not finalized")
    taps = np.zeros(L * N, np.float32)
    taps[3] = 0.5
    bad(lib.bfhip_engine_add_coeff(h, None, 10, 1.0, 0))
    bad(lib.bfhip_engine_add_coeff(h, taps.ctypes.data_as(C.c_void_p), -1, 1.0, 0))
    bad(lib.bfhip_engine_add_coeff(h, taps.ctypes.data_as(C.c_void_p), L * N, 1.0, N + 1), "blocks")
    bad(lib.bfhip_engine_add_coeff_processed(h, None, 1))
    bad(lib.bfhip_engine_add_coeff_processed_blocks(h, None, 1, 0))
    bad(lib.bfhip_engine_add_coeff_processed_blocks(h, (C.c_void_p * 1)(None), 1, 0), "NULL")
    one, half = (C.c_int * 1)(0), (C.c_double * 1)(1.0)
    bad(lib.bfhip_engine_add_filter(h, 1, None, half, 0, None, None, 1, one, half, -1, 0, 0), "null array")
    bad(lib.bfhip_engine_add_filter(h, 1, one, half, 0, None, None, 1, one, None, -1, 0, 0), "null array")
    bad(lib.bfhip_engine_add_filter(h, 1, (C.c_int * 1)(I), half, 0, None, None, 1, one, half, -1, 0, 0), "input channel")
    bad(lib.bfhip_engine_add_filter(h, 1, one, half, 0, None, None, 1, (C.c_int * 1)(O), half, -1, 0, 0), "output channel")
    bad(lib.bfhip_engine_add_filter(h, 1, one, half, 1, one, half, 1, one, half, -1, 0, 0), "not defined yet")
    bad(lib.bfhip_engine_add_filter(h, 1, one, half, 0, None, None, 1, one, half, 7, 0, 0), "not loaded")
    bad(lib.bfhip_engine_add_filter(h, 1, one, half, 0, None, None, 1, one, half, -2, 0, 0))
    bad(lib.bfhip_engine_add_filter(h, -1, one, half, 0, None, None, 1, one, half, -1, 0, 0))

    # ---- a real network: the bad calls above left nothing behind
    e.set_interleaved(0, "S24_4LE")
    e.set_interleaved(1, "S24_4LE")
    oe = bo.Engine(L, N, 4, I, O)
    oe.set_interleaved(0, "S24_4LE")
    oe.set_interleaved(1, "S24_4LE")
    irs = {}
    for o in range(O):
        for i in range(I):
            hh = cases.make_ir(np.random.default_rng(9 + o * I + i), L * N, I)
            irs[(o, i)] = hh
            e.add_filter(in_ch=[i], out_ch=[o], coeff=e.add_coeff(hh))
            oe.add_filter(in_ch=[i], out_ch=[o], coeff=oe.add_coeff(hh))
    e.finalize()

    # ---- after finalize
    bad(lib.bfhip_engine_set_format(h, 0, 0, C.byref(fmt)), "after finalize")
    bad(lib.bfhip_engine_add_filter(h, 1, one, half, 0, None, None, 1, one, half, -1, 0, 0), "after finalize")
    bad(lib.bfhip_engine_set_overlap(h, 1), "after finalize")
    bad(lib.bfhip_engine_set_coeff(h, I * O, 0))
    bad(lib.bfhip_engine_set_coeff(h, 0, I * O + 3))
    bad(lib.bfhip_engine_set_delayblocks(h, -1, 0))
    bad(lib.bfhip_engine_set_scale(h, 0, 2, 0, 1.0))
    bad(lib.bfhip_engine_set_scale(h, 0, 0, 1, 1.0), "bad index")
    bad(lib.bfhip_engine_set_fscale(h, 0, 0, 1.0), "bad index")
    bad(lib.bfhip_engine_refresh_coeff_processed(h, 0, N, None))
    bad(lib.bfhip_engine_refresh_coeff_processed(h, 0, 0, None), "no host buffer")
    bad(lib.bfhip_engine_update_coeff_block(h, 0, 0, None))
    bad(lib.bfhip_engine_read_coeff_processed(h, 99, taps.ctypes.data_as(C.c_void_p)))
    bad(lib.bfhip_engine_block(h, None, None, None), "null buffer")
    bad(lib.bfhip_engine_block_dev(h, null, null), "null buffer")
    bad(lib.bfhip_engine_block_dev_ev(h, null, null, None, None), "null buffer")
    bad(lib.bfhip_engine_inputs_dev(h, null), "null buffer")
    bad(lib.bfhip_engine_mac_dev(h, null), "null buffer")
    bad(lib.bfhip_engine_outputs_dev(h, null, 0, O, null), "null buffer")
    bad(lib.bfhip_engine_outputs_dev(h, null, 1, O, null), "channel range")
    bad(lib.bfhip_engine_outputs_inputs_dev(h, null, 0, O, null, null), "null buffer")
    bad(lib.bfhip_engine_get_overflow(h, O, C.byref(hip.Overflow())))
    bad(lib.bfhip_engine_get_overflow(h, 0, None))
    bad(lib.bfhip_engine_algorithmic_bytes(h, None), "null array")
    bad(lib.bfhip_engine_get_timing(h, None))
    bad(lib.bfhip_engine_read_ring_slot(h, 0, 99, taps.ctypes.data_as(C.c_void_p)))
    bad(lib.bfhip_engine_read_output_spectrum(h, O, taps.ctypes.data_as(C.c_void_p)))
    bad(lib.bfhip_engine_rt_submit(h, None), "not in real-time mode")
    bad(lib.bfhip_engine_rt_wait(h, None, None), "not in real-time mode")
    bad(lib.bfhip_engine_prewarm(h)) if e.blockcounter else None

    # ---- and the engine still works
    for blk in cases.raw_blocks(3, N + 2, L, I, "S24_4LE"):
        st, out = e.block(blk)
        so, want = oe.block(blk)
        assert st == so == 0
        assert np.abs(np.frombuffer(out.tobytes(), np.int32).astype(np.int64) - want.view(np.int32)).max() <= 1
    bad(lib.bfhip_engine_prewarm(h), "already")


def test_nupc_bad_arguments(hip):
    lib = hip.lib()
    null = C.c_void_p(0)
    assert not lib.bfhip_nupc_create(0, 4, 1, 1, 0, None, None)
    assert not lib.bfhip_nupc_create(0, 4, 1, 1, 2, (C.c_int * 2)(128, 64), (C.c_int * 2)(2, 2))      # descending
    assert b"ascend" in lib.bfhip_nupc_last_error()
    n = hip.Nupc([64, 128], [2, 2], 4, 1, 1)
    assert lib.bfhip_nupc_block_dev(n.h, null, null) < 0                        # not finalized
    assert lib.bfhip_nupc_add_filter(n.h, 1, 0, None, 10, 1.0, 1.0) < 0
    n.add_filter(0, 0, np.ones(10, np.float32))
    n.finalize()
    assert lib.bfhip_nupc_block_dev(n.h, null, null) < 0
    assert b"null buffer" in lib.bfhip_nupc_last_error()
    assert lib.bfhip_nupc_get_overflow(n.h, 1, None) < 0
    x = np.zeros((64, 1), np.float32)
    x[0, 0] = 1.0
    st, out = n.block(x)
    assert st == 0 and np.allclose(np.frombuffer(out.tobytes(), np.float32)[:10], 1.0, atol=1e-6)


def test_a_configuration_that_does_not_fit_the_device_is_an_error_not_a_crash(hip):
    """256 inputs x 60 000 partitions of 8192 taps: the spectrum rings alone would take 1 TB"""
    e = hip.Engine(8192, 60000, 4, 256, 1)
    e.set_interleaved(0, "S24_4LE")
    e.set_interleaved(1, "S24_4LE")
    e.add_filter(in_ch=[0], out_ch=[0], coeff=e.add_coeff(np.ones(16, np.float32)))
    with pytest.raises(hip.BfhipError, match="out of device memory"):
        e.finalize()
    with pytest.raises(hip.BfhipError, match="finalize failed before"):      # not retried on half-built state
        e.finalize()
    with pytest.raises(hip.BfhipError, match="not finalized"):
        e.sync()
    e.close()
    with pytest.raises(hip.BfhipError, match="exceed 4 GiB"):       # 32-bit offsets inside a coefficient set
        e2 = hip.Engine(8192, 70000, 4, 1, 1)
        e2.set_interleaved(0, "S24_4LE")
        e2.set_interleaved(1, "S24_4LE")
        e2.add_filter(in_ch=[0], out_ch=[0], coeff=e2.add_coeff(np.ones(16, np.float32)))
        e2.finalize()
    # the device is fine afterwards
    ge, _ = cases.crossbar(hip.Engine, L, N, 4, I, O)
    oe, _ = cases.crossbar(bo.Engine, L, N, 4, I, O)
    blk = cases.raw_blocks(5, 1, L, I, "S24_4LE")[0]
    assert np.abs(np.frombuffer(ge.block(blk)[1].tobytes(), np.int32).astype(np.int64)
                  - oe.block(blk)[1].view(np.int32)).max() <= 1


@pytest.mark.parametrize("rs", [4, 8])
def test_degenerate_networks(hip, rs):
    """no filter at all; outputs nothing feeds; a filter that feeds nothing; the smallest engine
    there is (L = 4, one partition, one channel) -- against the oracle"""
    def both(spec_fn, L_, N_, I_, O_, n_blocks=4):
        outs = []
        for cls in (hip.Engine, bo.Engine):
            e = cls(L_, N_, rs, I_, O_)
            e.set_interleaved(0, "S16_LE")
            e.set_interleaved(1, "S24_4LE")
            spec_fn(e)
            if hasattr(e, "finalize"):
                e.finalize()
            res = []
            for blk in cases.raw_blocks(21, n_blocks, L_, I_, "S16_LE", amplitude=0.3):
                st, raw = e.block(blk)
                assert st == 0
                res.append(np.frombuffer(raw.tobytes(), np.int32).copy())
            outs.append(np.concatenate(res))
        assert np.abs(outs[0].astype(np.int64) - outs[1]).max() <= 1
        return outs[0]

    y = both(lambda e: None, 64, 2, 2, 3)                                   # no filter at all
    assert not y.any()

    def two_of_three(e):
        c = e.add_coeff(cases.make_ir(np.random.default_rng(1), 128, 2).astype(np.float32 if rs == 4 else np.float64))
        e.add_filter(in_ch=[0], out_ch=[1], coeff=c)                        # outputs 0 and 2 stay silent
        e.add_filter(in_ch=[1], out_ch=[], coeff=c)                         # feeds nothing
    y = both(two_of_three, 64, 2, 2, 3).reshape(-1, 3)
    assert not y[:, 0].any() and not y[:, 2].any() and y[:, 1].any()

    def tiny(e):
        e.add_filter(in_ch=[0], out_ch=[0], coeff=e.add_coeff(np.array([0.5, 0.25, -0.125, 0.0625], np.float32 if rs == 4 else np.float64)))
    y = both(tiny, 4, 1, 1, 1, n_blocks=6)
    assert y.any()


def test_independent_engines_on_concurrent_host_threads(hip):
    """four host threads, an engine each (different lengths, precisions and schedules), stepping at
    the same time: the library's process-wide state (per-kernel LDS limits, twiddle tables, the
    thread-local error text) must not get in each other's way.  ctypes drops the GIL inside every
    call, so the calls really overlap."""
    import threading
    shapes = [(256, 3, 4), (1024, 2, 8), (4096, 2, 4), (64, 5, 8)]
    results = [None] * len(shapes)

    def run(k):
        try:
            L_, N_, rs_ = shapes[k]
            ge, _ = cases.crossbar(hip.Engine, L_, N_, rs_, 2, 2, seed=900 + k)
            oe, _ = cases.crossbar(bo.Engine, L_, N_, rs_, 2, 2, seed=900 + k)
            worst = 0
            for blk in cases.raw_blocks(40 + k, 3 * N_ + 20, L_, 2, "S24_4LE"):
                gs, g = ge.block(blk)
                os_, o = oe.block(blk)
                assert gs == os_ == 0
                worst = max(worst, int(np.abs(np.frombuffer(g.tobytes(), np.int32).astype(np.int64)
                                              - o.view(np.int32)).max()))
            # an error in one thread is that thread's text only
            assert hip.lib().bfhip_engine_set_delay(ge.h, 0, 99, 0) < 0
            assert b"set_delay" in hip.lib().bfhip_last_error()
            results[k] = worst
        except Exception as ex:                   # noqa: BLE001
            results[k] = ex

    threads = [threading.Thread(target=run, args=(k,)) for k in range(len(shapes))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    for k, r in enumerate(results):
        assert isinstance(r, int) and r <= 1, (k, r)


@pytest.mark.parametrize("L_", [256, 4096])
@pytest.mark.parametrize("fmt,rs", [("S24_4LE", 4), ("S32_LE", 4), ("FLOAT_LE", 4), ("FLOAT64_LE", 8), ("S16_LE", 4)])
def test_samples_at_odd_byte_offsets(hip, L_, fmt, rs):
    """frames in which the samples do not sit on their natural alignment (a header byte in front
    of every frame): the word-sized fast paths of K1 / K3 must notice and go byte by byte"""
    nbytes = bo.SAMPLE_FORMATS[fmt][0]
    spacing = 3                                      # samples per frame: 2 channels + room for the skew
    engines = []
    for cls, mod in ((hip.Engine, hip), (bo.Engine, bo)):
        e = cls(L_, 2, rs, 2, 2)
        for io in (0, 1):
            for c in range(2):
                e.set_format(io, c, mod.make_format(fmt, spacing, 1 + c * nbytes))
        e.in_bytes = e.out_bytes = L_ * spacing * nbytes
        for o in range(2):
            for i in range(2):
                h = cases.make_ir(np.random.default_rng(70 + o * 2 + i), 2 * L_, 2)
                e.add_filter(in_ch=[i], out_ch=[o], coeff=e.add_coeff(h.astype(np.float32 if rs == 4 else np.float64)))
        if hasattr(e, "finalize"):
            e.finalize()
        engines.append(e)
    ge, oe = engines
    rng = np.random.default_rng(3)
    dt = {"S24_4LE": np.int32, "S32_LE": np.int32, "FLOAT_LE": np.float32, "FLOAT64_LE": np.float64, "S16_LE": np.int16}[fmt]
    for _ in range(4):
        x = rng.standard_normal((L_, 2)) * 0.1
        if fmt == "S24_4LE":
            vals = (x * 8388608).astype(np.int32)
        elif fmt == "S32_LE":
            vals = (x * 2147483648.0).astype(np.int64).clip(-2**31, 2**31 - 1).astype(np.int32)
        elif fmt == "S16_LE":
            vals = (x * 32768).astype(np.int16)
        else:
            vals = x.astype(dt)
        raw = np.zeros((L_, spacing * nbytes), np.uint8)
        for c in range(2):
            raw[:, 1 + c * nbytes:1 + (c + 1) * nbytes] = vals[:, c:c + 1].copy().view(np.uint8).reshape(L_, nbytes)
        gs, g = ge.block(raw)
        os_, o = oe.block(raw)
        assert gs == os_ == 0
        g = np.frombuffer(g.tobytes(), np.uint8).reshape(L_, spacing * nbytes)
        o = np.frombuffer(o.tobytes(), np.uint8).reshape(L_, spacing * nbytes)
        for c in range(2):
            gv = np.ascontiguousarray(g[:, 1 + c * nbytes:1 + (c + 1) * nbytes]).view(dt).ravel()
            ov = np.ascontiguousarray(o[:, 1 + c * nbytes:1 + (c + 1) * nbytes]).view(dt).ravel()
            if fmt.startswith("FLOAT"):
                assert cases.rel_rms(gv.astype(np.float64), ov.astype(np.float64)) <= (1e-5 if rs == 4 else 1e-12)
            else:
                lsb = 256 if fmt == "S32_LE" and rs == 4 else 1       # float32 carries 24 bits
                assert np.abs(gv.astype(np.int64) - ov.astype(np.int64)).max() <= lsb


@pytest.mark.parametrize("L_", [512, 8192])
def test_device_buffers_that_are_only_sample_aligned(hip, L_):
    """bfhip_engine_block_dev with raw buffers that start 4 bytes into an allocation (a slice of
    somebody's larger buffer): nothing may assume 16-byte alignment of the caller's pointers"""
    import torch
    dev = torch.device("cuda", 0)
    I_, O_ = 3, 2
    ge, _ = cases.crossbar(hip.Engine, L_, 2, 4, I_, O_, "S24_4LE", "S24_4LE", seed=31)
    oe, _ = cases.crossbar(bo.Engine, L_, 2, 4, I_, O_, "S24_4LE", "S24_4LE", seed=31)
    big_in = torch.zeros(L_ * I_ + 8, dtype=torch.int32, device=dev)
    big_out = torch.zeros(L_ * O_ + 8, dtype=torch.int32, device=dev)
    for off in (1, 3):
        big_out.zero_()
        vin = big_in[off:off + L_ * I_]
        vout = big_out[off:off + L_ * O_]
        assert vin.data_ptr() % 16 != 0
        for blk in cases.raw_blocks(50 + off, 4, L_, I_, "S24_4LE"):
            vin.copy_(torch.from_numpy(blk.reshape(-1)).to(dev))
            torch.cuda.synchronize()
            ge.block_dev(vin, vout)
            assert ge.sync() == 0
            _, o = oe.block(blk)
            assert np.abs(vout.cpu().numpy().astype(np.int64) - o.view(np.int32)).max() <= 1
        assert int(big_out[:off].abs().sum()) == 0 and int(big_out[off + L_ * O_:].abs().sum()) == 0   # nothing outside


@pytest.mark.parametrize("I_,O_", [(300, 2), (2, 300), (130, 130)])
def test_very_wide_and_very_tall_crossbars(hip, I_, O_):
    """hundreds of channels on one side (more than BF_MAXCHANNELS = 256 even): output groups of 8
    with hundreds of entries, hundreds of groups with one entry; (130, 130) as one-to-one filters
    plus a few mixes"""
    L_, N_ = 128, 2
    engines = []
    for cls in (hip.Engine, bo.Engine):
        e = cls(L_, N_, 4, I_, O_)
        e.set_interleaved(0, "S16_LE")
        e.set_interleaved(1, "FLOAT_LE")
        rng = np.random.default_rng(11)
        if I_ == O_:
            for c in range(I_):
                h = cases.make_ir(rng, L_ * N_, 2).astype(np.float32)
                e.add_filter(in_ch=[c], out_ch=[c], coeff=e.add_coeff(h))
            for c in range(0, I_, 17):
                h = cases.make_ir(rng, L_ * N_, 4).astype(np.float32)
                e.add_filter(in_ch=[c, (c + 5) % I_], in_scale=[0.5, -0.5], out_ch=[(c + 1) % O_], coeff=e.add_coeff(h))
        else:
            hs = [e.add_coeff(cases.make_ir(rng, L_ * N_, max(I_, 2)).astype(np.float32)) for _ in range(7)]
            k = 0
            for o in range(O_):
                for i in range(I_):
                    e.add_filter(in_ch=[i], out_ch=[o], coeff=hs[k % 7], in_scale=[1.0 + 0.01 * (k % 5)])
                    k += 1
        if hasattr(e, "finalize"):
            e.finalize()
        engines.append(e)
    ge, oe = engines
    for blk in cases.raw_blocks(2, N_ + 3, L_, I_, "S16_LE", amplitude=0.2):
        gs, g = ge.block(blk)
        os_, o = oe.block(blk)
        assert gs == os_ == 0
        assert cases.rel_rms(cases.samples(g, "FLOAT_LE"), cases.samples(o, "FLOAT_LE")) <= 1e-5
