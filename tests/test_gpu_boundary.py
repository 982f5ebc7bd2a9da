"""GPU: the corners of the drop-in boundary that round 1 left without a test.

 * convolver_cbuf2raw(apply_dither = 1) from a C host that owns the dither tables, against the
   goldens the reference's own dither code produced (tests/golden/ref_dither_*.npz);
 * convolver_raw2cbuf with a `postprocess` callback (how delay.c hooks the sub-sample delay in,
   bfrun.c:1503-1508);
 * coefficient sets handed over the way an UNMODIFIED bfconf holds them -- one allocation per
   block (bfconf.c:1994-2009) -- and rewritten at run time from ANOTHER PROCESS the way
   bflogic_eq does it (rendereq.h:87-91), picked up by the engine at the next block;
 * the fork guard: a device op in a child of a process that already used the device fails
   cleanly instead of hanging in a dead runtime."""
import ctypes as C
import mmap
import os
import subprocess
import sys

import numpy as np
import pytest

import bforacle as bo
import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def p(a):
    return a.ctypes.data_as(C.c_void_p)


class SampleFormat(C.Structure):
    _fields_ = [("isfloat", C.c_int), ("swap", C.c_int), ("bytes", C.c_int), ("sbytes", C.c_int),
                ("scale", C.c_double), ("format", C.c_int)]


class BufferFormat(C.Structure):
    _fields_ = [("sf", SampleFormat), ("sample_spacing", C.c_int), ("byte_offset", C.c_int)]


@pytest.fixture()
def cv(hip):
    L = hip.lib()
    L.convolver_coeffs2cbuf.restype = C.c_void_p
    L.convolver_coeffs2cbuf.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    L.convolver_runtime_coeffs2cbuf.argtypes = [C.c_void_p, C.c_void_p]
    L.convolver_raw2cbuf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(BufferFormat), C.c_void_p, C.c_void_p]
    return L


@pytest.mark.parametrize("rs,tag", [(8, "f64"), (4, "f32")])
def test_cbuf2raw_with_dither_from_a_c_host_vs_reference_goldens(hip, tmp_path, rs, tag):
    g = np.load(os.path.join(G, "ref_dither_%s.npz" % tag))
    x = g["x"]
    n_blk, n_ch, L = x.shape
    table = g["table_head"]
    assert len(table) == int(g["table_size"])             # the whole table is in the fixture
    spacing = (len(table) - 1) // n_ch
    exe = str(tmp_path / "dither_host")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "chost", "dither_host.c"), "-o", exe,
                           "-L" + os.path.join(ROOT, "brutefir_amd"), "-lbfhip",
                           "-Wl,-rpath," + os.path.join(ROOT, "brutefir_amd")])
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.array([rs, L, n_ch, n_blk, len(table), spacing], np.int32).tobytes())
        f.write(table.tobytes())
        f.write(np.ascontiguousarray(x).tobytes())
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    out = open(tmp_path / "out.bin", "rb").read()
    n = n_blk * n_ch * L
    raw = np.frombuffer(out, np.int16, n).reshape(n_blk, n_ch, L)
    ptrs = np.frombuffer(out, np.int32, n_blk * n_ch, 2 * n).reshape(n_blk, n_ch)
    of = np.frombuffer(out, np.float64, n_ch * 4, 2 * n + 4 * n_blk * n_ch).reshape(n_ch, 4)
    # the integer bookkeeping (table walk incl. the wrap branch, dither.h:28-38) is exact
    assert np.array_equal(ptrs, g["randtab_ptr"])
    if rs == 8:
        assert np.array_equal(raw, g["raw"])              # bit-exact in the working precision
        assert np.array_equal(of, g["of"])
    else:
        # float32: the error-feedback chain is the same arithmetic without contraction; allow the
        # 1 LSB the engine-level test allows and require the clip accounting to agree
        assert np.abs(raw.astype(np.int32) - g["raw"].astype(np.int32)).max() <= 1
        assert of[:, 0].tolist() == g["of"][:, 0].tolist()


@pytest.mark.parametrize("rs", [4, 8])
def test_raw2cbuf_runs_the_postprocess_callback_on_the_converted_block(cv, rs):
    dt = np.float32 if rs == 4 else np.float64
    L = 256
    assert cv.convolver_init(None, L, rs) == 1
    seen = []
    CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)

    def post(realbuf, n_samples, arg):
        seen.append((n_samples, arg))
        a = np.ctypeslib.as_array(C.cast(realbuf, C.POINTER(C.c_float if rs == 4 else C.c_double)), (n_samples,))
        a *= 2                                            # what apply_subdelay would filter in place

    cb = CB(post)
    rng = np.random.default_rng(5)
    raw = rng.integers(-30000, 30000, (L, 3), dtype=np.int16)      # 3 interleaved S16 channels
    bf = BufferFormat(SampleFormat(0, 0, 2, 2, 1.0 / 32768, 0), 3, 2)   # channel 1
    cbuf, nxt = np.zeros(2 * L, dt), np.zeros(2 * L, dt)
    token = C.c_int(42)
    cv.convolver_raw2cbuf(p(raw), p(cbuf), p(nxt), C.byref(bf), cb, C.addressof(token))
    assert seen == [(L, C.addressof(token))]
    want = raw[:, 1].astype(dt) * 2
    assert np.array_equal(nxt[:L], want)                  # filtered block kept for the next call
    assert np.array_equal(cbuf[L:], want)                 # and copied behind the previous one (:193)
    assert not cbuf[:L].any()


def _render(cv, taps, L, rs, dest_addr):
    dt = np.float32 if rs == 4 else np.float64
    t = np.ascontiguousarray(taps, dt)
    assert cv.convolver_coeffs2cbuf(p(t), len(t), 1.0, C.c_void_p(dest_addr)) == dest_addr


@pytest.mark.parametrize("rs", [4, 8])
def test_coefficient_blocks_in_separate_allocations(hip, cv, rs):
    """bfconf->coeffs_data[c][i] are separate buffers prepared by convolver_coeffs2cbuf in the
    parent (host code, no HIP); the engine takes them as they are"""
    dt = np.float32 if rs == 4 else np.float64
    L, N, I, O = 512, 3, 2, 2
    assert cv.convolver_init(None, L, rs) == 1
    e = hip.Engine(L, N, rs, I, O)
    oe = bo.Engine(L, N, rs, I, O)
    for x in (e, oe):
        x.set_interleaved(0, "S16_LE")
        x.set_interleaved(1, "FLOAT_LE" if rs == 4 else "FLOAT64_LE")
    keep = []
    for o in range(O):
        for i in range(I):
            h = cases.make_ir(np.random.default_rng(9 + o * I + i), L * N - 100, I).astype(dt)
            blocks = []
            for b in range(N):
                buf = np.empty(2 * L, dt)                 # one allocation per block
                _render(cv, h[b * L:(b + 1) * L], L, rs, buf.ctypes.data)
                blocks.append(buf)
            keep.append(blocks)
            c = e.add_coeff_processed_blocks([b.ctypes.data for b in blocks])
            e.add_filter(in_ch=[i], out_ch=[o], coeff=c)
            oe.add_filter(in_ch=[i], out_ch=[o], coeff=oe.add_coeff(h))
    e.finalize()
    for blk in cases.raw_blocks(3, N + 3, L, I, "S16_LE"):
        st, got = e.block(blk)
        _, want = oe.block(blk)
        assert st == 0
        assert cases.rel_rms(np.frombuffer(got.tobytes(), dt), np.frombuffer(want.tobytes(), dt)) <= (1e-5 if rs == 4 else 1e-12)


@pytest.mark.parametrize("rs", [4, 8])
def test_partition_rewritten_by_another_process_is_applied_at_the_next_block(hip, cv, rs):
    """The bflogic_eq flow with nothing modified: the module process renders new taps into the
    shared coefficient memory through bfaccess->convolver_coeffs2cbuf
    (= convolver_runtime_coeffs2cbuf, rendereq.h:87-91); that call is pure host code here and
    leaves a change notice; the filter process's engine re-uploads the partition at the start of
    its next block.  Reference behaviour to match: from that block on ALL history is convolved
    with the new partitions (the filter loop just reads coeffs_data, bfrun.c:1745-1770)."""
    dt = np.float32 if rs == 4 else np.float64
    L, N = 256, 4
    tol = 1e-5 if rs == 4 else 1e-12
    assert cv.convolver_init(None, L, rs) == 1            # parent, "before the fork": creates the notice table
    shm = mmap.mmap(-1, N * 2 * L * rs)                   # MAP_SHARED, like the coefficient segment
    base = np.frombuffer(shm, dt)
    addr = [base[b * 2 * L:].ctypes.data for b in range(N)]
    h0 = cases.make_ir(np.random.default_rng(1), L * N, 1).astype(dt)
    h1 = h0.copy()
    h1[L:3 * L] = cases.make_ir(np.random.default_rng(2), 2 * L, 1).astype(dt)      # partitions 1 and 2 change
    for b in range(N):
        _render(cv, h0[b * L:(b + 1) * L], L, rs, addr[b])
    e = hip.Engine(L, N, rs, 1, 1)
    oe = bo.Engine(L, N, rs, 1, 1)
    for x in (e, oe):
        x.set_interleaved(0, "S16_LE")
        x.set_interleaved(1, "FLOAT_LE" if rs == 4 else "FLOAT64_LE")
    c = e.add_coeff_processed_blocks(addr, watch=True)
    e.add_filter(in_ch=[0], out_ch=[0], coeff=c)
    e.finalize()
    oc0, oc1 = oe.add_coeff(h0), oe.add_coeff(h1)
    oe.add_filter(in_ch=[0], out_ch=[0], coeff=oc0)
    blocks = cases.raw_blocks(8, 2 * N + 4, L, 1, "S16_LE")
    switch_at = N + 2

    def step(k):
        st, got = e.block(blocks[k])
        _, want = oe.block(blocks[k])
        assert st == 0
        return cases.rel_rms(np.frombuffer(got.tobytes(), dt), np.frombuffer(want.tobytes(), dt))

    for k in range(switch_at):
        assert step(k) <= tol, k
    assert e.poll_coeff_changes() == 0
    pid = os.fork()                                       # the module process: never touches HIP
    if pid == 0:
        try:
            for b in (1, 2):
                src = np.ascontiguousarray(h1[b * L:(b + 1) * L])
                cv.convolver_runtime_coeffs2cbuf(p(src), C.c_void_p(addr[b]))
        finally:
            os._exit(0)
    assert os.waitpid(pid, 0)[1] == 0
    oe.set_coeff(0, oc1)
    errs = [step(k) for k in range(switch_at, len(blocks))]
    assert max(errs) <= tol, errs
    assert e.poll_coeff_changes() == 0                    # both notices were consumed by the block call
    # and the device copy now equals what the host memory holds
    got = e.read_coeff_processed(c, N)
    assert np.array_equal(got.ravel(), base)


def test_device_op_in_a_child_of_a_device_process_fails_cleanly(hip):
    """HIP state does not survive fork(): instead of hanging in the inherited runtime the library
    reports fatal code 106.  (Run in a fresh interpreter: the child must be a fork of a process
    whose FIRST device op went through this library.)"""
    code = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, %r)
import brutefir_amd as bf
L = bf.lib()
for f in ("convolver_time2freq",):
    getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
assert L.convolver_init(None, 64, 4) == 1
x = np.ones(128, np.float32); y = np.empty(128, np.float32)
L.convolver_time2freq(x.ctypes.data, y.ctypes.data)          # parent initialises HIP
assert abs(y[0] - 128.0) < 1e-3
r, w = os.pipe()
pid = os.fork()
if pid == 0:
    HANDLER = C.CFUNCTYPE(None, C.c_int, C.c_char_p)
    msgs = []
    h = HANDLER(lambda code, msg: msgs.append((code, msg)))
    L.bfhip_convolver_set_fatal_handler(h)
    L.convolver_time2freq(x.ctypes.data, y.ctypes.data)      # must not touch the dead runtime
    os.write(w, ("%%d" %% L.bfhip_convolver_last_fatal()).encode())
    os._exit(0)
os.close(w)
got = os.read(r, 16).decode()
os.waitpid(pid, 0)
print("child fatal code", got)
assert got == "106", got
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "child fatal code 106" in r.stdout


@pytest.mark.parametrize("L", [2048, 4096])       # 2048: three-stream pipeline; 4096: wave FFT -> ping-pong schedule
def test_block_dev_ev_orders_an_asynchronous_producer_and_consumer(hip, L):
    """The ordering contract of bfhip_engine_block_dev (include/bfhip.h): with the pipelined
    block (overlap forced on) K1 runs on a side stream that does not follow the caller's
    stream.  A producer that fills the SAME input buffer on its own stream right before every
    call, and a consumer that copies the output away on that stream right after, stay correct
    when they hand the engine an `input ready` event and wait on its `output done` event -- no
    host synchronisation anywhere in the loop.  Result must equal the plain synchronous engine
    bit for bit."""
    import torch
    N, I, O = 4, 4, 4
    dev = torch.device("cuda", 0)
    ref_e, _ = cases.crossbar(hip.Engine, L, N, 4, I, O, "S24_4LE", "FLOAT_LE")
    blocks = cases.raw_blocks(31, N + 6, L, I, "S24_4LE")
    want = []
    for blk in blocks:
        st, raw = ref_e.block(blk)
        assert st == 0
        want.append(np.frombuffer(raw.tobytes(), np.float32).reshape(L, O).copy())

    def build(overlap):
        e = hip.Engine(L, N, 4, I, O)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, "FLOAT_LE")
        for o in range(O):
            for i in range(I):
                h = cases.make_ir(np.random.default_rng(4321 + o * I + i), L * N, I)
                e.add_filter(in_ch=[i], out_ch=[o], coeff=e.add_coeff(h))
        e.set_overlap(overlap)
        e.finalize()
        return e

    for overlap in (1, 0):
        e = build(overlap)
        assert e.block_mode == ((3 if L >= 4096 else 1) if overlap else 0)
        owed = e.output_lag                               # the ping-pong schedule writes block k during call k+2
        assert owed == (2 if e.block_mode == 3 else 0)
        side = torch.cuda.Stream()
        pinned = [torch.from_numpy(b).pin_memory() for b in blocks]
        rawin = torch.zeros(L, I, dtype=torch.int32, device=dev)          # ONE buffer, reused
        rawout = torch.zeros(L, O, dtype=torch.float32, device=dev)
        keep = [torch.zeros(L, O, dtype=torch.float32, device=dev) for _ in blocks]
        big = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
        ready = [torch.cuda.Event() for _ in blocks]
        done = [torch.cuda.Event() for _ in blocks]
        for ev in ready + done:
            ev.record(side)                                               # creates the hipEvent_t handles
        torch.cuda.synchronize()
        # one input buffer per block here: with outputs owed for two calls `done[k]` (which also
        # releases the input of block k) fires too late to reuse ONE input buffer without stalling
        rawins = [torch.zeros(L, I, dtype=torch.int32, device=dev) for _ in blocks] if owed else [rawin] * len(blocks)
        rawouts = [torch.zeros(L, O, dtype=torch.float32, device=dev) for _ in range(3)] if owed else [rawout]
        with torch.cuda.stream(side):
            for k in range(len(blocks)):
                if k > 0 and not owed:
                    side.wait_event(done[k - 1])                          # rawin may be rewritten now
                big.fill_(k)                                              # keeps the producer stream busy
                rawins[k].copy_(pinned[k], non_blocking=True)
                ready[k].record(side)
                e.block_dev_ev(rawins[k], rawouts[k % len(rawouts)], ready[k].cuda_event, done[k].cuda_event)
                j = k - owed                                              # the block whose output this call wrote
                if j >= 0:
                    side.wait_event(done[j])
                    keep[j].copy_(rawouts[j % len(rawouts)], non_blocking=True)
            assert e.sync() == 0                                          # flushes what is still owed
            for j in range(len(blocks) - owed, len(blocks)):
                keep[j].copy_(rawouts[j % len(rawouts)], non_blocking=True)
        torch.cuda.synchronize()
        assert e.sync() == 0
        for k in range(len(blocks)):
            assert np.array_equal(keep[k].cpu().numpy(), want[k]), (overlap, k)


def test_deferred_output_gives_the_same_samples(hip, monkeypatch):
    """Large crossbars run [K3 of block t-1 | K1 of block t] as ONE launch (deferred output,
    include/bfhip.h): the output of a block is written during the next call or by sync.  Forced
    on here for a small crossbar; every block lands in a buffer of its own and is read after the
    final sync; also with one shared output buffer and the `output done` event, and across a
    run-time coefficient switch (plan rebuild with an output still owed)."""
    import torch
    L, N, I, O = 4096, 3, 3, 5
    dev = torch.device("cuda", 0)
    ref_e, irs = cases.crossbar(hip.Engine, L, N, 4, I, O, "S24_4LE", "S24_4LE")
    blocks = cases.raw_blocks(12, N + 5, L, I, "S24_4LE")
    extra = ref_e  # noqa: F841

    def build():
        e = hip.Engine(L, N, 4, I, O)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, "S24_4LE")
        cs = {}
        for o in range(O):
            for i in range(I):
                cs[(o, i)] = e.add_coeff(irs[(o, i)])
                e.add_filter(in_ch=[i], out_ch=[o], coeff=cs[(o, i)])
        return e, cs

    # the reference run: synchronous block(), with a coefficient switch on filter 0 at block 4
    monkeypatch.setenv("BFHIP_DEFER", "0")
    plain, cs = build()
    plain.finalize()
    want = []
    for k, blk in enumerate(blocks):
        if k == 4:
            plain.set_coeff(0, cs[(1, 1)])
        st, raw = plain.block(blk)
        assert st == 0
        want.append(raw.view(np.int32).reshape(L, O).copy())
    monkeypatch.setenv("BFHIP_DEFER", "1")
    monkeypatch.setenv("BFHIP_OVERLAP", "0")              # a crossbar this small would be pipelined instead
    e, cs = build()
    e.finalize()
    assert e.block_mode == 2 and e.uses_wave_fft
    srcs = [torch.from_numpy(b).to(dev) for b in blocks]
    outs = [torch.zeros(L, O, dtype=torch.int32, device=dev) for _ in blocks]
    torch.cuda.synchronize()
    for k in range(len(blocks)):
        if k == 4:
            e.set_coeff(0, cs[(1, 1)])
        e.block_dev(srcs[k], outs[k])
    assert e.sync() == 0
    for k in range(len(blocks)):
        assert np.array_equal(outs[k].cpu().numpy(), want[k]), k
    assert [e.overflow(c).astuple() for c in range(O)] == [plain.overflow(c).astuple() for c in range(O)]
    # one shared output buffer: the `output done` event of block k fires when ITS samples are there
    e2, _ = build()
    e2.finalize()
    assert e2.block_mode == 2
    shared = torch.zeros(L, O, dtype=torch.int32, device=dev)
    keep = [torch.zeros(L, O, dtype=torch.int32, device=dev) for _ in blocks]
    side = torch.cuda.Stream()
    done = [torch.cuda.Event() for _ in blocks]
    for ev in done:
        ev.record(side)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for k in range(len(blocks)):
            if k == 4:
                e2.set_coeff(0, cs[(1, 1)])
            e2.block_dev_ev(srcs[k], shared, None, done[k].cuda_event)
            if k > 0:                                     # block k-1's output was written by this call
                side.wait_event(done[k - 1])
                keep[k - 1].copy_(shared, non_blocking=True)
                side.synchronize()                        # shared is free again before the next call
        assert e2.sync() == 0
        keep[-1].copy_(shared)
    torch.cuda.synchronize()
    got = [t.cpu().numpy() for t in keep]
    where = [[j for j in range(len(blocks)) if np.array_equal(got[k], want[j])] for k in range(len(blocks))]
    assert where == [[k] for k in range(len(blocks))], str(where)


@pytest.mark.parametrize("schedule", ["deferred", "pingpong"])
def test_flush_launches_owed_outputs_without_waiting(hip, monkeypatch, schedule):
    """bfhip_engine_flush (include/bfhip.h): a caller that wants the samples of the block it just
    submitted does not wait for later calls -- flush launches the owed output passes and records
    their `output done` events; the consumer waits on the event, never on the host.  After a flush
    the schedule starts over; a phase call on the main stream right behind a flushed ping-pong
    engine (its MAC reuses the partial-sum buffer) stays ordered behind the flushed passes."""
    import torch
    L, N, I, O = 4096, 3, 3, 4
    dev = torch.device("cuda", 0)
    if schedule == "deferred":
        monkeypatch.setenv("BFHIP_OVERLAP", "0")
        monkeypatch.setenv("BFHIP_DEFER", "1")
    else:
        monkeypatch.setenv("BFHIP_OVERLAP", "1")
    e, _ = cases.crossbar(hip.Engine, L, N, 4, I, O, "S24_4LE", "S24_4LE")
    monkeypatch.delenv("BFHIP_OVERLAP")
    monkeypatch.delenv("BFHIP_DEFER", raising=False)
    assert e.output_lag == (1 if schedule == "deferred" else 2) and e.block_mode == (2 if schedule == "deferred" else 3)
    monkeypatch.setenv("BFHIP_OVERLAP", "0")
    monkeypatch.setenv("BFHIP_DEFER", "0")
    plain, _ = cases.crossbar(hip.Engine, L, N, 4, I, O, "S24_4LE", "S24_4LE")
    monkeypatch.delenv("BFHIP_OVERLAP")
    monkeypatch.delenv("BFHIP_DEFER")
    assert plain.output_lag == 0 and plain.block_mode == 0
    blocks = cases.raw_blocks(77, N + 6, L, I, "S24_4LE")
    want = [np.frombuffer(plain.block(b)[1].tobytes(), np.int32).reshape(L, O).copy() for b in blocks]
    srcs = [torch.from_numpy(b).to(dev) for b in blocks]
    outs = [torch.zeros(L, O, dtype=torch.int32, device=dev) for _ in blocks]
    keep = [torch.zeros(L, O, dtype=torch.int32, device=dev) for _ in blocks]
    side = torch.cuda.Stream()
    done = [torch.cuda.Event() for _ in blocks]
    for ev in done:
        ev.record(side)
    torch.cuda.synchronize()
    z = torch.zeros(O, L, 2, dtype=torch.float32, device=dev)
    with torch.cuda.stream(side):
        for k in range(len(blocks)):
            e.block_dev_ev(srcs[k], outs[k], None, done[k].cuda_event)
            if k % 3 == 2:                                # every third block is wanted at once
                e.flush()
                for j in range(k - 2, k + 1):
                    side.wait_event(done[j])
                    keep[j].copy_(outs[j], non_blocking=True)
        e.flush()
        for j in range(len(blocks) - len(blocks) % 3, len(blocks)):
            side.wait_event(done[j])
            keep[j].copy_(outs[j], non_blocking=True)
    torch.cuda.synchronize()
    for k in range(len(blocks)):
        assert np.array_equal(keep[k].cpu().numpy(), want[k]), (schedule, k)
    # a phase call straight behind a flush: same spectra as the plain engine's for the next block
    e.block_dev(srcs[0], outs[0])
    e.block_dev(srcs[1], outs[1])
    e.flush()
    e.inputs_dev(srcs[2])
    e.mac_dev(z)
    e.advance()
    for k in range(3):
        plain.inputs_dev(srcs[k])
        zp = torch.zeros_like(z)
        plain.mac_dev(zp)
        plain.advance()
    assert e.sync() == 0 and plain.sync() == 0
    assert np.array_equal(z.cpu().numpy(), zp.cpu().numpy())


@pytest.mark.parametrize("L", [1024, 2048])
def test_wave_fft_forced_on_at_small_lengths(hip, monkeypatch, L):
    """fft_wave.h is the default from L = 4096 up; BFHIP_FFT_WAVE=1 turns it on down to 1024 so
    that every radix of its first pass (2, 4, 8, 16) is exercised against the oracle"""
    monkeypatch.setenv("BFHIP_FFT_WAVE", "1")
    ge, _ = cases.crossbar(hip.Engine, L, 3, 4, 3, 2, "S24_4LE", "FLOAT_LE")
    monkeypatch.delenv("BFHIP_FFT_WAVE")
    oe, _ = cases.crossbar(bo.Engine, L, 3, 4, 3, 2, "S24_4LE", "FLOAT_LE")
    for blk in cases.raw_blocks(2, 7, L, 3, "S24_4LE"):
        gs, g = ge.block(blk)
        _, o = oe.block(blk)
        assert gs == 0
        assert cases.rel_rms(np.frombuffer(g.tobytes(), np.float32), np.frombuffer(o.tobytes(), np.float32)) <= 1e-5


@pytest.mark.parametrize("rs", [4, 8])
def test_stream_ordered_coefficients_follow_runtime_changes(hip, cv, monkeypatch, rs):
    """Uniform crossbars are multiplied from a stream-ordered COPY of the coefficients (every MAC
    workgroup one sequential slice, kernels.h StreamLayout).  The copy has to follow everything
    that changes coefficients at run time: a filter switching to another set (plan rebuild, only
    the touched entries are laid out again), bfhip_engine_update_coeff_block, and a watched
    partition rewritten by another process.  Forced on for a small crossbar; the oracle is the
    checker, and an engine without the copy must give the same bits."""
    dt = np.float32 if rs == 4 else np.float64
    L, N, I, O = 1024, 4, 8, 8
    tol = 1e-5 if rs == 4 else 1e-12
    assert cv.convolver_init(None, L, rs) == 1
    rng = np.random.default_rng(3)
    irs = {(o, i): cases.make_ir(np.random.default_rng(50 + o * I + i), L * N, I).astype(dt) for o in range(O) for i in range(I)}
    alt = cases.make_ir(rng, L * N, I).astype(dt)
    shm = mmap.mmap(-1, N * 2 * L * rs)
    base = np.frombuffer(shm, dt)
    addr = [base[b * 2 * L:].ctypes.data for b in range(N)]

    def build(mod, stream):
        monkeypatch.setenv("BFHIP_COEFF_STREAM", stream)
        monkeypatch.setenv("BFHIP_MAC_TARGET_WGS", "8")      # 2 bin tiles x 4 chunks of 2 inputs: a uniform grid
        e = mod.Engine(L, N, rs, I, O)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, "FLOAT_LE" if rs == 4 else "FLOAT64_LE")
        cs = {}
        for (o, i), h in irs.items():
            if mod is hip and (o, i) == (0, 0):
                for b in range(N):
                    _render(cv, h[b * L:(b + 1) * L], L, rs, addr[b])
                cs[(o, i)] = e.add_coeff_processed_blocks(addr, watch=True)     # lives in "shared memory"
            else:
                cs[(o, i)] = e.add_coeff(h)
            e.add_filter(in_ch=[i], out_ch=[o], coeff=cs[(o, i)])
        c_alt = e.add_coeff(alt)
        if hasattr(e, "finalize"):
            e.finalize()
        return e, cs, c_alt

    se, scs, salt = build(hip, "2")
    pe, pcs, palt = build(hip, "0")
    assert se.uses_stream_layout and not pe.uses_stream_layout
    oe, ocs, oalt = build(bo, "0")
    new_part = cases.make_ir(rng, L, I).astype(dt)
    h00 = irs[(0, 0)].copy()
    h00[L:2 * L] = new_part
    o_h00 = oe.add_coeff(h00)
    new_blk = cases.make_ir(rng, L, I).astype(dt)
    h35 = irs[(3, 5)].copy()
    h35[2 * L:3 * L] = new_blk
    o_h35 = oe.add_coeff(h35)
    f35 = 3 * I + 5
    blocks = cases.raw_blocks(4, 3 * N + 4, L, I, "S24_4LE")
    for k, blk in enumerate(blocks):
        if k == N + 1:                               # filter (out 2, in 6) switches to another set
            f = 2 * I + 6
            for e, c in ((se, salt), (pe, palt), (oe, oalt)):
                e.set_coeff(f, c)
        if k == N + 3:                               # one partition of set (3, 5) replaced in place
            se.update_coeff_block(scs[(3, 5)], 2, new_blk)
            pe.update_coeff_block(pcs[(3, 5)], 2, new_blk)
            oe.set_coeff(f35, o_h35)
        if k == 2 * N + 2:                           # "another process" rewrites partition 1 of the watched set
            pid = os.fork()
            if pid == 0:
                try:
                    cv.convolver_runtime_coeffs2cbuf(p(np.ascontiguousarray(new_part)), C.c_void_p(addr[1]))
                finally:
                    os._exit(0)
            assert os.waitpid(pid, 0)[1] == 0
            oe.set_coeff(0, o_h00)
        s1, g = se.block(blk)
        s2, q = pe.block(blk)
        _, w = oe.block(blk)
        assert s1 == 0 and s2 == 0
        assert np.array_equal(g, q), k               # same arithmetic, different memory layout
        assert cases.rel_rms(np.frombuffer(g.tobytes(), dt), np.frombuffer(w.tobytes(), dt)) <= tol, k


@pytest.mark.parametrize("rs", [4, 8])
def test_stream_ordered_copy_is_not_left_stale_by_changes_in_one_gap(hip, cv, monkeypatch, rs):
    """An in-place rewrite of a partition does not change the set's address, and the rebuild of a
    dirty plan compares the entries of the stream-ordered copy by address: a rewrite that arrives
    in the SAME inter-block gap as a plan-dirtying change (a scale, another filter's switch), during
    a cross-fade block (which runs without the copy) or in the gap behind one (the plan is dirty
    again after it) must still reach the copy.  ADVICE r2 (medium): it did not."""
    dt = np.float32 if rs == 4 else np.float64
    L, N, I, O = 1024, 4, 8, 8
    tol = 1e-5 if rs == 4 else 1e-12
    assert cv.convolver_init(None, L, rs) == 1
    rng = np.random.default_rng(13)
    irs = {(o, i): cases.make_ir(np.random.default_rng(150 + o * I + i), L * N, I).astype(dt) for o in range(O) for i in range(I)}
    alt = cases.make_ir(rng, L * N, I).astype(dt)
    shm = mmap.mmap(-1, N * 2 * L * rs)
    base = np.frombuffer(shm, dt)
    addr = [base[b * 2 * L:].ctypes.data for b in range(N)]
    fid = lambda o, i: o * I + i

    def build(mod, stream):
        monkeypatch.setenv("BFHIP_COEFF_STREAM", stream)
        monkeypatch.setenv("BFHIP_MAC_TARGET_WGS", "8")
        e = mod.Engine(L, N, rs, I, O)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, "FLOAT_LE" if rs == 4 else "FLOAT64_LE")
        cs = {}
        for (o, i), h in irs.items():
            if mod is hip and (o, i) == (0, 0):
                for b in range(N):
                    _render(cv, h[b * L:(b + 1) * L], L, rs, addr[b])
                cs[(o, i)] = e.add_coeff_processed_blocks(addr, watch=True)
            else:
                cs[(o, i)] = e.add_coeff(h)
            e.add_filter(in_ch=[i], out_ch=[o], coeff=cs[(o, i)], crossfade=(o, i) == (2, 6))
        c_alt = e.add_coeff(alt)
        if hasattr(e, "finalize"):
            e.finalize()
        return e, cs, c_alt

    se, scs, salt = build(hip, "2")
    pe, pcs, palt = build(hip, "0")
    assert se.uses_stream_layout and not pe.uses_stream_layout
    oe, ocs, oalt = build(bo, "0")

    def rewritten(key, part):
        """the oracle's counterpart of an in-place rewrite: the whole set with one partition replaced"""
        blk = cases.make_ir(rng, L, I).astype(dt)
        h = irs[key].copy()
        h[part * L:(part + 1) * L] = blk
        return blk, oe.add_coeff(h)

    b35, o35 = rewritten((3, 5), 2)
    b41, o41 = rewritten((4, 1), 0)
    b57, o57 = rewritten((5, 7), 1)
    b00, o00 = rewritten((0, 0), 3)
    blocks = cases.raw_blocks(4, 3 * N + 2, L, I, "S24_4LE")
    for k, blk in enumerate(blocks):
        if k == N + 1:           # an output scale (plan dirty) and an in-place rewrite in one gap
            for e in (se, pe, oe):
                e.set_scale(fid(1, 2), 1, 0, 0.5)
            se.update_coeff_block(scs[(3, 5)], 2, b35)
            pe.update_coeff_block(pcs[(3, 5)], 2, b35)
            oe.set_coeff(fid(3, 5), o35)
        if k == N + 3:           # this block cross-fades (no stream-ordered copy in use); rewrite in its gap
            for e, c in ((se, salt), (pe, palt), (oe, oalt)):
                e.set_coeff(fid(2, 6), c)
            se.update_coeff_block(scs[(4, 1)], 0, b41)
            pe.update_coeff_block(pcs[(4, 1)], 0, b41)
            oe.set_coeff(fid(4, 1), o41)
        if k == N + 4:           # the gap behind the fade block: the plan is dirty again
            se.update_coeff_block(scs[(5, 7)], 1, b57)
            pe.update_coeff_block(pcs[(5, 7)], 1, b57)
            oe.set_coeff(fid(5, 7), o57)
        if k == 2 * N + 2:       # a watched partition rewritten by another process + a scale in one gap
            pid = os.fork()
            if pid == 0:
                try:
                    cv.convolver_runtime_coeffs2cbuf(p(np.ascontiguousarray(b00)), C.c_void_p(addr[3]))
                finally:
                    os._exit(0)
            assert os.waitpid(pid, 0)[1] == 0
            oe.set_coeff(fid(0, 0), o00)
            for e in (se, pe, oe):
                e.set_scale(fid(6, 3), 1, 0, -0.75)
        s1, g = se.block(blk)
        s2, q = pe.block(blk)
        _, w = oe.block(blk)
        assert s1 == 0 and s2 == 0
        if k != N + 3:
            assert se.uses_stream_layout, k
        assert np.array_equal(g, q), k               # same arithmetic, different memory layout
        assert cases.rel_rms(np.frombuffer(g.tobytes(), dt), np.frombuffer(w.tobytes(), dt)) <= tol, k


def test_coefficient_slabs_reserve_overflow_and_error_paths(hip, monkeypatch):
    """coefficient sets live in slabs (bfhip_engine_reserve_coeffs / 64 MiB + 2 GiB pieces): a
    reservation that turns out too small spills into a further slab, a rejected set (NaN among the
    taps) gives its space back, and the layout knobs change no sample"""
    L, N, I, O = 512, 2, 3, 3
    blocks = cases.raw_blocks(21, N + 3, L, I, "S16_LE")

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = hip.Engine(L, N, 4, I, O)
        e.set_interleaved(0, "S16_LE")
        e.set_interleaved(1, "FLOAT_LE")
        e.reserve_coeffs(4 * N * 2 * L * 4)              # room for 4 of the 9 sets: the rest spills
        bad = np.ones(L * N, np.float32)
        bad[7] = np.nan
        with pytest.raises(hip.BfhipError, match="NaN or Inf"):
            e.add_coeff(bad)
        for o in range(O):
            for i in range(I):
                h = cases.make_ir(np.random.default_rng(70 + o * I + i), L * N, I)
                e.add_filter(in_ch=[i], out_ch=[o], coeff=e.add_coeff(h))
        with pytest.raises(hip.BfhipError, match="NaN or Inf"):
            e.add_coeff(bad)
        e.finalize()
        outs = []
        for blk in blocks:
            st, raw = e.block(blk)
            assert st == 0
            outs.append(raw.copy())
        for k in env:
            monkeypatch.delenv(k)
        return outs

    base = run({})
    for env in ({"BFHIP_COEFF_ARENA": "0"}, {"BFHIP_COEFF_SLAB_MB": "1"}, {"BFHIP_COEFF_PAD_B": "4096"},
                {"BFHIP_COEFF_PAD_B": "r"}):
        got = run(env)
        assert all(np.array_equal(a, b) for a, b in zip(base, got)), env


def test_engine_in_a_forked_child_is_refused_not_hung(hip):
    """include/bfhip.h: an engine lives in the process that created it.  A fork()ed child that
    inherits the handle gets BFHIP_ESTATE from every device entry point -- before any HIP call,
    which would hang in the inherited runtime -- may destroy its copy of the handle, and the
    parent's engine is unaffected.  (Fresh interpreter: os.fork in the pytest process would
    duplicate pytest.)"""
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import brutefir_amd as bf
import cases
e, _ = cases.crossbar(bf.Engine, 256, 2, 4, 2, 2)
blk = cases.raw_blocks(1, 3, 256, 2, "S24_4LE")
st, first = e.block(blk[0])
r, w = os.pipe()
pid = os.fork()
if pid == 0:
    msgs = []
    for call in (lambda: e.block(blk[1]), lambda: e.sync(), lambda: e.add_coeff(np.ones(4, np.float32)),
                 lambda: e.overflow(0), lambda: e.rt_begin(0)):
        try:
            call()
            msgs.append("NO ERROR")
        except bf.BfhipError as ex:
            msgs.append("refused" if "belongs to process" in str(ex) else "OTHER: " + str(ex))
    e.close()                                  # frees the child's copy of the handle, touches no device object
    os.write(w, ";".join(msgs).encode())
    os._exit(0)
os.close(w)
got = os.read(r, 4096).decode()
_, status = os.waitpid(pid, 0)
print("child:", got, "exit", status)
st2, second = e.block(blk[1])                  # the parent goes on as if nothing had happened
ref, _ = cases.crossbar(bf.Engine, 256, 2, 4, 2, 2)
ref.block(blk[0])
st3, want = ref.block(blk[1])
assert st == st2 == st3 == 0 and np.array_equal(second, want)
assert got == ";".join(["refused"] * 5) and status == 0, got
print("PARENT OK")
''' % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "PARENT OK" in r.stdout, r.stdout + r.stderr[-2000:]
