"""Child process of tests/test_gpu_alloc_faults.py: one engine life -- build (N:1 inputs, a
sub-sample delay, dither, powersave, a cross-fading filter, an N-way mix, a cascade), finalize,
blocks through the host call, a run-time scale change (private ring), a coefficient switch, the
real-time mode with graph capture -- repeated with the n-th allocation failing, n = 1, 2, ...
until a whole life passes without reaching the armed allocation.  Every failure has to surface
as BfhipError (an error code), never as a crash; the engine is destroyed each time.  Prints
"n=<k> <where it failed>" lines and a final summary line."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (first: it brings its own HIP runtime, tests/conftest.py)
torch.cuda.init()
import brutefir_amd as bf  # noqa: E402


def free_mib():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2 ** 20

L, N = 256, 3
rs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
big = len(sys.argv) > 2 and sys.argv[2] == "big"      # partition length above the LDS limit
nupc = len(sys.argv) > 2 and sys.argv[2] == "nupc"    # bfhip_nupc instead of a plain engine
shard = len(sys.argv) > 2 and sys.argv[2] == "shard"  # a shard of a configuration: lazy sets, wide I/O, pairs, one-to-one MAC
if shard:
    os.environ["BFHIP_WIDE_IO"] = "1"
if big:
    L = 16384
dt = np.float32 if rs == 4 else np.float64
rng = np.random.default_rng(5)
taps = [(rng.standard_normal(L * N) / 300).astype(dt) for _ in range(3)]
x = (rng.standard_normal((L, 2)) * 2e5).astype(np.int32)


def nupc_life(stage):
    """the non-uniform convolver: three segment engines + its own rings and pinned staging"""
    n = bf.Nupc([64, 128, 256], [2, 2, 3], rs, 1, 2)
    try:
        stage[0] = "setup"
        n.set_interleaved(0, "FLOAT_LE")
        n.set_interleaved(1, "S24_4LE")
        n.add_filter(0, 0, (rng.standard_normal(1000) / 40).astype(dt))
        n.add_filter(0, 1, (rng.standard_normal(700) / 40).astype(dt))
        stage[0] = "finalize"
        n.finalize()
        stage[0] = "blocks"
        xs = (rng.standard_normal((64, 1)) * 0.1).astype(np.float32)
        for _ in range(9):
            n.block(xs)
        stage[0] = "done"
    finally:
        left[0] = bf.lib().bfhip_selftest_fail_alloc(0)
        n.close()


def shard_life(stage):
    """what round 3 added: an engine that runs a shard (lazily registered sets that reach the device
    inside build_plan, owned-output staging), the planar copies of BFHIP_WIDE_IO, block pairs (a
    second partial-sum buffer), and a one-to-one plan (mac_diag_kernel's job table)"""
    I, O = 2, 4
    whole = bf.Engine(L, N, rs, I, O)
    e = d = None
    try:
        stage[0] = "setup"
        for eng in (whole,):
            eng.set_interleaved(0, "S24_4LE")
            eng.set_interleaved(1, "S24_4LE")
        cs = [whole.add_coeff(t) for t in taps]
        whole.add_filter(in_ch=[0], out_ch=[0], coeff=cs[0])
        whole.finalize()
        host = [np.ascontiguousarray(whole.read_coeff_processed(c, N)) for c in cs]
        stage[0] = "shard"
        e = bf.Engine(L, N, rs, I, O)
        e.set_interleaved(0, "S24_4LE")
        e.set_interleaved(1, "S24_4LE")
        e.enable_pairs(True)
        for h in host:
            e.add_coeff_processed_blocks([h[b].ctypes.data for b in range(N)], lazy=True)
        for o in range(O):
            for i in range(I):
                f = e.add_filter(in_ch=[i], out_ch=[o], coeff=(o + i) % 2)
                e.set_filter_active(f, o < 2)
        stage[0] = "finalize"
        e.finalize()
        stage[0] = "blocks"
        xs = (rng.standard_normal((L, I)) * 2e5).astype(np.int32)
        out = np.zeros(L * O * 4, np.uint8)
        for _ in range(N + 1):
            e.block(xs, out=out)
        stage[0] = "switch"
        e.set_coeff(0, 2)                   # a set this engine never needed: loaded inside build_plan
        e.block(xs, out=out)
        stage[0] = "diag"
        d = bf.Engine(L, N, rs, 3, 3)
        d.set_interleaved(0, "S24_4LE")
        d.set_interleaved(1, "FLOAT_LE" if rs == 4 else "FLOAT64_LE")
        for c in range(3):
            d.add_filter(in_ch=[c], out_ch=[c], coeff=d.add_coeff(taps[c]))
        d.finalize()
        assert d.uses_diag_mac
        d.block((rng.standard_normal((L, 3)) * 2e5).astype(np.int32))
        stage[0] = "done"
    finally:
        left[0] = bf.lib().bfhip_selftest_fail_alloc(0)
        for eng in (whole, e, d):
            if eng is not None:
                eng.close()


def life(stage):
    if nupc:
        return nupc_life(stage)
    if shard:
        return shard_life(stage)
    e = bf.Engine(L, N, rs, 3, 3)
    try:
        stage[0] = "setup"
        e.map_channels(0, [0, 1, 1])
        e.set_interleaved_phys(0, "S24_4LE", 2)
        e.set_interleaved(1, "S16_LE")
        e.set_maxdelay(0, 1, 300)
        e.set_maxdelay(0, 2, 300)
        if not big:
            e.enable_subdelay(15, 9.0)
            e.set_subdelay(1, 0, 37)
            e.set_powersave(1.0)
        e.enable_dither([1], 44100, 0)
        stage[0] = "coeffs"
        cs = [e.add_coeff(t) for t in taps]
        e.add_filter(in_ch=[0], out_ch=[0], coeff=cs[0], crossfade=True)
        e.add_filter(in_ch=[1, 2], in_scale=[0.5, 0.25], out_ch=[1], coeff=cs[1])
        e.add_filter(in_ch=[0], out_ch=[], coeff=cs[2])
        e.add_filter(in_f=[2], out_ch=[2], coeff=-1)
        stage[0] = "finalize"
        e.finalize()
        stage[0] = "blocks"
        for _ in range(2):
            e.block(x)
        stage[0] = "control"
        e.set_scale(2, 0, 0, 0.5)          # promotes filter 2 to a private ring
        e.set_coeff(0, cs[1])              # cross-fade
        e.set_delay(0, 1, 123)
        e.block(x)
        e.update_coeff_block(cs[2], 1, taps[0][:L])
        e.block(x)
        stage[0] = "real-time"
        e.rt_begin(0)
        for _ in range(3):
            e.rt_block(x)
        e.rt_end()
        e.rt_begin(bf.RT_OVERLAP)
        e.rt_block(x)
        e.rt_end()
        stage[0] = "done"
    finally:
        left[0] = bf.lib().bfhip_selftest_fail_alloc(0)
        e.close()


left = [0]
life(["warm-up"])              # module loads, LDS attributes, runtime pools


def armed_life(n):
    """one life with the n-th allocation failing: 'absorbed', 'unreached' or the error"""
    stage = ["create"]
    bf.lib().bfhip_selftest_fail_alloc(n)
    try:
        life(stage)
        return ("unreached" if left[0] > 0 else "absorbed"), stage[0]
    except bf.BfhipError as ex:
        if left[0] > 0:
            print("n=%d: error WITHOUT the armed allocation: %s" % (n, ex), flush=True)
            sys.exit(3)
        return str(ex), stage[0]


# The slab-retry path on its own (a coefficient slab that does not fit is retried at half the size:
# the life goes on with a smaller slab).  The first such life moves the runtime's own pools once
# (observed: -24 MiB); a leak on that path would move free memory on EVERY such life.  So: find the
# first absorbed allocation, live it twenty times, and require free memory to be flat from the
# second life on.  Only then is the baseline for the walk below taken -- no re-baseline inside it.
retry_n, retry_drift = 0, 0.0
for cand in range(1, 13):
    if armed_life(cand)[0] == "absorbed":
        retry_n = cand
        break
if retry_n:
    frees = []
    for _ in range(20):
        assert armed_life(retry_n)[0] == "absorbed"
        frees.append(free_mib())
    retry_drift = max(frees[1:]) - min(frees[1:])
    print("RETRY n=%d lives=20 free_first=%.1f free_last=%.1f drift_after_first_mib=%.2f"
          % (retry_n, frees[0], frees[-1], retry_drift), flush=True)
    if retry_drift > 1.0 or frees[1] - frees[-1] > 1.0:
        print("the slab-retry path leaks: free memory per life %s" % ["%.1f" % f for f in frees], flush=True)
        sys.exit(4)

free0 = free_mib()
failed, absorbed, n = 0, 0, 0
while True:
    n += 1
    what, where = armed_life(n)
    if what == "unreached":
        print("n=%d: a whole life makes %d allocations" % (n, n - left[0]), flush=True)
        break
    if what == "absorbed":
        absorbed += 1          # reached, and handled without an error (a slab retried at half the size)
        print("n=%d absorbed   [free %+.1f MiB]" % (n, free_mib() - free0), flush=True)
    else:
        failed += 1
        print("n=%d failed in %s: %s   [free %+.1f MiB]" % (n, where, what[:90], free_mib() - free0), flush=True)
    if n > 600:
        print("too many allocations", flush=True)
        sys.exit(2)
# and a clean life afterwards
life(["clean"])
drift = free0 - free_mib()
print("SUMMARY allocations_walked=%d errors_reported=%d absorbed=%d leaked_mib=%.1f retry_n=%d retry_drift_mib=%.2f"
      % (n - 1, failed, absorbed, drift, retry_n, retry_drift), flush=True)
