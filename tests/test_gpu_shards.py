"""GPU: one engine per filter process -- a SHARD of the configuration (include/bfhip.h, "one engine per
filter process"; the reference forks n_processes filter processes, bfrun.c:2312-2328, each running
the filters bfconf gave it, bfconf.c:2227-2318, every output mixed inside one process,
bfconf.c:2893-2931).

What must hold, and is checked here bit for bit: the engines of the processes, each writing into the
raw output buffer (and overflow array) they share, leave behind EXACTLY what one engine running the
whole configuration writes -- the property SURVEY B.5(iii) observed on the reference's own
multi-process mode.  The plan of every shard is the whole configuration's plan with the foreign
terms taken out, so every output is summed in the same order."""
import ctypes as C

import numpy as np
import pytest

import bforacle as bo
import cases
import test_gpu_fuzz as fuzz
from test_gpu_boundary import cv  # noqa: F401  (fixture: the host-side convolver_* symbols)

pytestmark = pytest.mark.gpu


def _components(spec):
    """filters that share an output, or are connected, must be run together (bfconf.c:2893-2931)"""
    F = len(spec["filters"])
    parent = list(range(F))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    by_out = {}
    for fi, f in enumerate(spec["filters"]):
        for g in f["in_f"]:
            parent[find(fi)] = find(g)
        for o in f["out_ch"]:
            if o in by_out:
                parent[find(fi)] = find(by_out[o])
            by_out[o] = fi
    return [find(fi) for fi in range(F)]


def _assign(spec, n_proc, rng):
    comp = _components(spec)
    where = {c: int(rng.integers(0, n_proc)) for c in sorted(set(comp))}
    f_owner = [where[c] for c in comp]
    o_owner = [0] * spec["n_out"]                     # outputs nobody feeds: process 0
    for fi, f in enumerate(spec["filters"]):
        for o in f["out_ch"]:
            o_owner[o] = f_owner[fi]
    return f_owner, o_owner


def _build(hip, spec, f_owner=None, o_owner=None, k=None, prepare=None, order=None):
    e = hip.Engine(spec["L"], spec["N"], spec["rs"], spec["n_in"], spec["n_out"])
    e.set_interleaved(0, spec["infmt"])
    e.set_interleaved(1, spec["outfmt"])
    if prepare:
        prepare(e)
    for taps, scale, nb in spec["coeffs"]:
        e.add_coeff(taps, scale, nb)
    for f in spec["filters"]:
        e.add_filter(**f)
    if f_owner is not None:
        for fi, p in enumerate(f_owner):
            e.set_filter_active(fi, p == k)
        for o, p in enumerate(o_owner):
            e.set_output_active(o, p == k)
    e.finalize()
    return e


def _ovf(hip, n):
    return (hip.Overflow * n)()


def _same_overflow(a, b, n):
    for ch in range(n):
        for fld in ("n_overflows", "intlargest", "largest", "max"):
            assert getattr(a[ch], fld) == getattr(b[ch], fld), (ch, fld, getattr(a[ch], fld), getattr(b[ch], fld))


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("BFHIP_SHARD_SEEDS", "40"))))
def test_shards_write_what_the_whole_engine_writes_bit_for_bit(hip, seed):
    spec, n_blocks, events = fuzz._network(seed)
    rng = np.random.default_rng(777 + seed)
    if seed % 2:
        spec["outfmt"] = "S24_4LE" if seed % 4 == 1 else "S16_LE"      # quantiser + overflow counters
    n_proc = 2 + seed % 2
    f_owner, o_owner = _assign(spec, n_proc, rng)
    whole = _build(hip, spec)
    shards = [_build(hip, spec, f_owner, o_owner, k) for k in range(n_proc)]
    for k, s in enumerate(shards):
        assert [s.output_is_active(o) for o in range(spec["n_out"])] == [p == k for p in o_owner]
    blocks = cases.raw_blocks(seed, n_blocks, spec["L"], spec["n_in"], spec["infmt"], amplitude=0.6)
    wo, so = _ovf(hip, spec["n_out"]), _ovf(hip, spec["n_out"])
    for ch in range(spec["n_out"]):
        wo[ch] = whole.overflow(ch)
        so[ch] = whole.overflow(ch)
    for b, blk in enumerate(blocks):
        for e in [whole] + shards:
            fuzz._apply(e, events.get(b, []))           # every process sees the whole fctrl array
        ws, w = whole.block(blk, overflow=wo)
        shared = np.full(w.size, 0xA5, np.uint8)        # what no engine owns stays as it was
        st = 0
        for s in shards:
            st |= s.block(blk, overflow=so, out=shared)[0]
        assert st == ws, (seed, b)
        assert np.array_equal(shared, w), (seed, b, n_proc, f_owner, o_owner)
        _same_overflow(wo, so, spec["n_out"])


def test_filter_order_and_names_do_not_change_a_bit(hip):
    """the entries of an output group are ordered by (ring, delay) with filters going by the host's
    own numbering (bfhip_engine_set_filter_name = struct bffilter.intname): the same configuration
    listed in another order sums every output in the same order"""
    L, N, I, O = 256, 3, 3, 5
    rng = np.random.default_rng(4)
    taps = [cases.make_ir(rng, L * N, I).astype(np.float32) for _ in range(4)]
    flt = []
    for o in range(O):
        for i in range(I):
            flt.append(dict(in_ch=[i], out_ch=[o], coeff=int(rng.integers(0, 4)), delayblocks=int(rng.integers(0, 2)),
                            in_scale=[float(rng.choice([1.0, 0.5]))]))
    flt.append(dict(in_ch=[0, 2], in_scale=[0.5, -0.25], out_ch=[1, 4], coeff=2))        # N-way mix: a private ring
    flt.append(dict(in_ch=[1, 2], in_scale=[1.0, 0.25], out_ch=[4], coeff=1))

    def build(order):
        e = hip.Engine(L, N, 4, I, O)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, "FLOAT_LE")
        for t in taps:
            e.add_coeff(t)
        for pos, fi in enumerate(order):
            assert e.add_filter(**flt[fi]) == pos
            e.set_filter_name(pos, fi)
        e.finalize()
        return e
    a = build(list(range(len(flt))))
    b = build([int(x) for x in rng.permutation(len(flt))])
    for blk in cases.raw_blocks(9, N + 3, L, I, "S24_4LE"):
        assert np.array_equal(a.block(blk)[1], b.block(blk)[1])


def test_shared_physical_outputs_dither_and_subdelay_in_shards(hip):
    """N:1 virtual -> physical outputs (mixed in the time domain), HP-TPDF dither (every channel's walk
    through the random table starts where it starts in the whole configuration) and sub-sample
    delays: each owned by one of two engines"""
    L, N = 256, 2
    rng = np.random.default_rng(11)
    taps = [cases.make_ir(rng, L * N, 2).astype(np.float32) for _ in range(5)]

    def build(k):
        e = hip.Engine(L, N, 4, 2, 5)
        e.set_interleaved(0, "S24_4LE")
        e.map_channels(1, [0, 0, 1, 2, 3])              # virtual 0, 1 share physical 0
        e.set_interleaved_phys(1, "S16_LE", 4)
        e.set_maxdelay(1, 0, 100)
        e.set_maxdelay(1, 1, 100)
        e.set_delay(1, 1, 37)
        e.enable_subdelay(15, 9.0)
        e.set_subdelay(1, 3, 41)
        e.enable_dither([0, 1, 3], 44100, 0)
        for t in taps:
            e.add_coeff(t)
        owner = [0, 0, 1, 1, 0]
        for o in range(5):
            e.add_filter(in_ch=[o % 2], out_ch=[o], coeff=o)
            if k is not None:
                e.set_filter_active(o, owner[o] == k)
        e.finalize()
        return e
    whole, shards = build(None), [build(0), build(1)]
    assert [shards[1].output_is_active(o) for o in range(5)] == [False, False, True, True, False]
    for b, blk in enumerate(cases.raw_blocks(3, 9, L, 2, "S24_4LE", amplitude=0.4)):
        if b == 4:
            for e in [whole] + shards:
                e.set_delay(1, 1, 5)
                e.set_subdelay(1, 3, -20)
        _, w = whole.block(blk)
        shared = np.full(w.size, 0x5A, np.uint8)
        for s in shards:
            s.block(blk, out=shared)
        assert np.array_equal(shared, w), b
    for ch in range(5):
        got = shards[0 if ch in (0, 1, 4) else 1].overflow(ch)
        want = whole.overflow(ch)
        assert (got.n_overflows, got.intlargest, got.largest) == (want.n_overflows, want.intlargest, want.largest)


def test_a_shard_without_any_filtered_or_shared_output_of_its_own(hip):
    """every output with a sub-sample filter and every shared physical output belongs to the OTHER engine:
    this one has no time-sample buffer and nothing to filter behind the inverse transforms (it once
    launched the sub-sample filters of the foreign outputs on that missing buffer: a memory fault on
    the GPU, found by tests/test_gpu_refloop.py's soak through the patched host)"""
    L, N = 256, 2
    rng = np.random.default_rng(5)
    taps = [cases.make_ir(rng, L * N, 2).astype(np.float64) for _ in range(3)]

    def build(k):
        e = hip.Engine(L, N, 8, 2, 6)
        e.map_channels(0, [0, 0])
        e.map_channels(1, [2, 0, 2, 0, 1, 2])           # physical 1 <- virtual 4 alone; 0 and 2 are shared
        e.set_interleaved_phys(0, "S16_LE", 1)
        e.set_interleaved_phys(1, "S24_4LE", 3)
        e.enable_subdelay(7, 9.0)
        e.set_subdelay(1, 0, -51)
        e.set_subdelay(1, 3, -72)
        for v, (d, md) in enumerate([(0, 0), (458, 1200), (866, 900), (165, 300), (15, 40), (0, 0)]):
            e.set_maxdelay(1, v, md); e.set_delay(1, v, d)
        for t in taps:
            e.add_coeff(t)
        e.add_filter(in_ch=[0], out_ch=[2, 5], coeff=1)
        e.add_filter(in_ch=[1], out_ch=[4], coeff=2)
        if k is not None:
            e.set_filter_active(0, k == 0)
            e.set_filter_active(1, k == 1)
            for v in range(6):
                e.set_output_active(v, (k == 1) == (v == 4))      # engine 1: the 1:1 output only
        e.finalize()
        return e
    whole, shards = build(None), [build(0), build(1)]
    for b, blk in enumerate(cases.raw_blocks(7, 8, L, 1, "S16_LE", amplitude=0.3)):
        _, w = whole.block(blk)
        shared = np.full(w.size, 0x5A, np.uint8)
        for s_ in shards:
            st, _ = s_.block(blk, out=shared)
            assert st == 0
        assert np.array_equal(shared, w), b


def test_shard_rules_are_checked_at_finalize(hip):
    L, N = 64, 2
    taps = np.ones(L, np.float32)

    def base(map_out=None):
        e = hip.Engine(L, N, 4, 2, 3)
        e.set_interleaved(0, "S16_LE")
        if map_out:
            e.map_channels(1, map_out)
            e.set_interleaved_phys(1, "S16_LE", max(map_out) + 1)
        else:
            e.set_interleaved(1, "S16_LE")
        e.add_coeff(taps)
        return e
    e = base()                                   # one output mixed from filters of two engines
    e.add_filter(in_ch=[0], out_ch=[0], coeff=0)
    e.add_filter(in_ch=[1], out_ch=[0], coeff=0)
    e.set_filter_active(1, False)
    with pytest.raises(hip.BfhipError, match="two engines"):
        e.finalize()
    e = base()                                   # connected filters in different engines
    e.add_filter(in_ch=[0], out_ch=[], coeff=0)
    e.add_filter(in_f=[0], out_ch=[1], coeff=0)
    e.set_filter_active(0, False)
    with pytest.raises(hip.BfhipError, match="connected"):
        e.finalize()
    e = base()                                   # an owned output declared foreign
    e.add_filter(in_ch=[0], out_ch=[2], coeff=0)
    e.set_output_active(2, False)
    with pytest.raises(hip.BfhipError, match="marked inactive"):
        e.finalize()
    e = base([0, 0, 1])                          # the members of a shared physical output split
    e.add_filter(in_ch=[0], out_ch=[0], coeff=0)
    e.add_filter(in_ch=[1], out_ch=[1], coeff=0)
    e.set_filter_active(1, False)
    with pytest.raises(hip.BfhipError, match="share a physical channel"):
        e.finalize()
    e = base()
    e.add_filter(in_ch=[0], out_ch=[0], coeff=0)
    e.finalize()
    with pytest.raises(hip.BfhipError, match="after finalize"):
        e.set_filter_active(0, False)


@pytest.mark.parametrize("rs", [4, 8])
def test_lazy_coefficient_sets_reach_the_device_when_first_needed(hip, rs):
    """a shard registers every set of the configuration (bfconf->coeffs_data: host memory that lives
    as long as the host) but loads only what its own filters refer to; a run-time switch to a set it
    has never needed loads that set at that block"""
    dt = np.float32 if rs == 4 else np.float64
    L, N, I, O = 512, 3, 4, 8
    rng = np.random.default_rng(21)
    irs = [cases.make_ir(rng, L * N, I).astype(dt) for _ in range(I * O + 1)]
    whole = hip.Engine(L, N, rs, I, O)
    whole.set_interleaved(0, "S24_4LE")
    whole.set_interleaved(1, "S24_4LE")
    for h in irs:
        whole.add_coeff(h)
    for o in range(O):
        for i in range(I):
            whole.add_filter(in_ch=[i], out_ch=[o], coeff=o * I + i)
    whole.finalize()
    host = [np.ascontiguousarray(whole.read_coeff_processed(c, N)) for c in range(len(irs))]     # "bfconf->coeffs_data"
    shards = []
    for k in range(2):
        e = hip.Engine(L, N, rs, I, O)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, "S24_4LE")
        for h in host:
            e.add_coeff_processed_blocks([h[b].ctypes.data for b in range(N)], lazy=True)
        for o in range(O):
            for i in range(I):
                f = e.add_filter(in_ch=[i], out_ch=[o], coeff=o * I + i)
                e.set_filter_active(f, (o < 3) == (k == 0))        # outputs 0..2 | 3..7: the group of eight is split
        e.finalize()
        shards.append(e)
    mine = [[c for c in range(I * O) if (c // I < 3) == (k == 0)] for k in range(2)]
    for k, e in enumerate(shards):
        assert [c for c in range(len(irs)) if e.coeff_is_resident(c)] == mine[k]
    extra = I * O
    for b, blk in enumerate(cases.raw_blocks(5, 2 * N + 3, L, I, "S24_4LE")):
        if b == N + 1:
            for e in [whole] + shards:
                e.set_coeff(5 * I + 2, extra)           # a filter of shard 1 switches to the set nobody has loaded
        _, w = whole.block(blk)
        shared = np.zeros(w.size, np.uint8)
        for s in shards:
            s.block(blk, out=shared)
        assert np.array_equal(shared, w), b
    assert shards[1].coeff_is_resident(extra) and not shards[0].coeff_is_resident(extra)


def test_shards_in_real_time_mode_and_on_device_buffers(hip):
    """the two other ways out of an engine: graph-replayed periods through the pinned double buffer
    (bfhip_engine_rt_*: what the patched host calls) and device buffers (the output pass itself skips
    foreign channels)"""
    import torch
    L, N, I, O = 1024, 3, 4, 6
    spec_f = [dict(in_ch=[i], out_ch=[o], coeff=(o + i) % 3) for o in range(O) for i in range(I)]
    rng = np.random.default_rng(8)
    taps = [cases.make_ir(rng, L * N, I).astype(np.float32) for _ in range(3)]

    def build(k):
        e = hip.Engine(L, N, 4, I, O)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, "S24_4LE")
        for t in taps:
            e.add_coeff(t)
        for f in spec_f:
            n = e.add_filter(**f)
            if k is not None:
                e.set_filter_active(n, (f["out_ch"][0] % 2) == k)      # even | odd outputs: interleaved ownership
        e.finalize()
        return e
    whole, shards = build(None), [build(0), build(1)]
    blocks = cases.raw_blocks(6, N + 6, L, I, "S24_4LE")
    want = [whole.block(blk)[1] for blk in blocks]
    for s in shards:
        s.rt_begin(0)
    for b, blk in enumerate(blocks):
        shared = np.full(want[b].size, 0x11, np.uint8)
        for s in shards:
            s.rt_block(blk, out=shared)
        assert np.array_equal(shared, want[b]), b
    assert shards[0].rt_stats()["graph"] > 0
    for s in shards:
        s.rt_end()
    # device buffers: two fresh engines, one device output buffer
    whole, shards = build(None), [build(0), build(1)]
    dev_out = torch.full((want[0].size,), 0x22, dtype=torch.uint8, device="cuda")
    for b, blk in enumerate(blocks):
        src = torch.from_numpy(np.ascontiguousarray(blk).view(np.uint8).ravel().copy()).cuda()
        dev_out.fill_(0x22)
        torch.cuda.synchronize()
        for s in shards:
            s.block_dev(src, dev_out)
            assert s.sync() == 0
        assert np.array_equal(dev_out.cpu().numpy(), want[b]), b


def test_lazy_and_watched_set_rewritten_before_its_first_use(hip, cv):
    """a set in shared memory that a module process rewrites (bflogic_eq through
    bfaccess->convolver_coeffs2cbuf) while no filter of THIS engine uses it yet: registered lazily
    and watched, it is loaded with its CURRENT content when a filter first switches to it, and
    followed from then on"""
    import mmap
    from test_gpu_boundary import _render, p
    L, N, rs = 512, 3, 4
    assert cv.convolver_init(None, L, rs) == 1
    rng = np.random.default_rng(4)
    first = cases.make_ir(rng, L * N, 1).astype(np.float32)
    newer = cases.make_ir(rng, L * N, 1).astype(np.float32)
    later = cases.make_ir(rng, L, 1).astype(np.float32)
    shm = mmap.mmap(-1, N * 2 * L * rs)
    base = np.frombuffer(shm, np.float32)
    addr = [base[b * 2 * L:].ctypes.data for b in range(N)]
    for b in range(N):
        _render(cv, first[b * L:(b + 1) * L], L, rs, addr[b])

    def build(mod):
        e = mod.Engine(L, N, rs, 1, 2)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, "FLOAT_LE")
        c0 = e.add_coeff(first)
        if mod is hip:
            c1 = e.add_coeff_processed_blocks(addr, watch=True, lazy=True)
        else:
            c1 = e.add_coeff(first)
        e.add_filter(in_ch=[0], out_ch=[0], coeff=c0)
        e.add_filter(in_ch=[0], out_ch=[1], coeff=c0)
        if mod is hip:
            e.set_filter_active(1, False)               # output 1 belongs to another engine
            e.finalize()
        return e, c1
    ge, g1 = build(hip)
    oe, _ = build(bo)
    o_newer = oe.add_coeff(newer)
    mixed = newer.copy()
    mixed[L:2 * L] = later
    o_mixed = oe.add_coeff(mixed)
    assert not ge.coeff_is_resident(g1)
    for k, blk in enumerate(cases.raw_blocks(2, 3 * N + 2, L, 1, "S24_4LE")):
        if k == 2:                                      # the "module" renders a whole new response; nobody uses the set yet
            for b in range(N):
                cv.convolver_runtime_coeffs2cbuf(p(np.ascontiguousarray(newer[b * L:(b + 1) * L])), C.c_void_p(addr[b]))
        if k == 4:
            assert not ge.coeff_is_resident(g1)
            ge.set_coeff(0, g1)
            oe.set_coeff(0, o_newer)
        if k == 2 * N:                                  # ... and one partition again while it is in use
            cv.convolver_runtime_coeffs2cbuf(p(np.ascontiguousarray(later)), C.c_void_p(addr[1]))
            oe.set_coeff(0, o_mixed)
        st, g = ge.block(blk)
        _, o = oe.block(blk)
        assert st == 0
        got = np.frombuffer(g.tobytes(), np.float32).reshape(L, 2)[:, 0]
        want = np.frombuffer(o.tobytes(), np.float32).reshape(L, 2)[:, 0]
        assert cases.rel_rms(got, want) <= 1e-5, k
    assert ge.coeff_is_resident(g1)


def test_block_pairs_in_a_shard_that_owns_whole_output_groups(hip):
    """two blocks per pass over the coefficients in an engine that runs a shard: owning whole groups
    of eight outputs keeps the plan a uniform crossbar, so the paired kernel runs; the shared buffers
    end up bit-identical to one engine's single blocks"""
    import torch
    dev = torch.device("cuda", 0)
    L, N, I, O = 1024, 3, 4, 16
    rng = np.random.default_rng(12)
    taps = [cases.make_ir(rng, L * N, I).astype(np.float32) for _ in range(I * O)]

    def build(k):
        e = hip.Engine(L, N, 4, I, O)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, "S24_4LE")
        if k is not None:
            e.enable_pairs(True)
        for o in range(O):
            for i in range(I):
                f = e.add_filter(in_ch=[i], out_ch=[o], coeff=e.add_coeff(taps[o * I + i]))
                if k is not None:
                    e.set_filter_active(f, (o // 8) == k)
        e.finalize()
        return e
    whole, shards = build(None), [build(0), build(1)]
    blocks = cases.raw_blocks(4, 2 * N + 6, L, I, "S24_4LE")
    srcs = [torch.from_numpy(b).to(dev) for b in blocks]
    outs = [torch.full((L, O), 0x33333333, dtype=torch.int32, device=dev) for _ in blocks]
    for k in range(0, len(blocks), 2):
        for s in shards:
            s.block_pair_dev(srcs[k], outs[k], srcs[k + 1], outs[k + 1])
    for s in shards:
        assert s.sync() == 0 and s.pair_launches > 0
    for k, blk in enumerate(blocks):
        _, w = whole.block(blk)
        assert np.array_equal(outs[k].cpu().numpy().view(np.uint8).ravel(), w), k
