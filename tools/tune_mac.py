#!/usr/bin/env python3
"""Interleaved A/B of MAC-kernel launch geometries in ONE process on ONE device (guide rule 24):
builds the config-C engine once, then for every round and every candidate rebuilds the device
plan with BFHIP_MAC_TARGET_WGS=<candidate> and times 40 steady-state blocks with HIP events.

    python tools/tune_mac.py 256 512 1024 2048 [--rounds 3]
    python tools/tune_mac.py --env BFHIP_MAC_UNROLL 1 2 3 4      # same for another plan knob
    python tools/tune_mac.py --grid 2:256 3:256 2:512 3:512      # unroll:target_wgs pairs
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (before the library: see tests/conftest.py)
import bench  # noqa: E402
import brutefir_amd as bf  # noqa: E402


def main():
    argv = sys.argv[1:]

    def opt(name, default=None):
        if name in argv:
            i = argv.index(name)
            v = argv[i + 1]
            del argv[i:i + 2]
            return v
        return default
    rounds = int(opt("--rounds", 3))
    knob = opt("--env", "BFHIP_MAC_TARGET_WGS")
    wl = opt("--workload", "C")
    grid = "--grid" in argv
    if grid:
        argv.remove("--grid")
    cands = (list(argv) if grid else [int(a) for a in argv]) or [512, 1024, 2048]
    I, O, L, N, rs, fmt = bench.WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    eng = bf.Engine(L, N, rs, I, O)
    eng.set_interleaved(bf.IN, fmt)
    eng.set_interleaved(bf.OUT, fmt)
    h = bench.synth_ir_dev(torch, 1, L * N, I, dev).to(torch.float32 if rs == 4 else torch.float64)
    for o in range(O):
        for i in range(I):
            eng.add_filter(in_ch=[i], out_ch=[o], coeff=eng.add_coeff_dev(h, L * N))
    torch.cuda.synchronize()
    eng.finalize()
    raw_in = bench.synth_raw_blocks(torch, 2, L, I, dev)
    raw_out = torch.zeros(L, O, dtype=torch.int32, device=dev)
    if fmt != "S24_4LE":
        raw_in = (torch.randn(2, L, I, device=dev, dtype=torch.float64) * 0.1).contiguous()
        raw_out = torch.zeros(L, O, dtype=torch.float64, device=dev)
    for k in range(N + 2):
        eng.block_dev(raw_in[k % 2], raw_out)
    eng.sync()
    alg = eng.algorithmic_bytes()["mac"]
    res = {c: [] for c in cands}
    for r in range(rounds):
        for c in cands:
            if grid:
                os.environ["BFHIP_MAC_UNROLL"], os.environ["BFHIP_MAC_TARGET_WGS"] = c.split(":")
            else:
                os.environ[knob] = str(c)
            eng.set_delayblocks(0, 1)       # any control change marks the plan dirty ...
            eng.set_delayblocks(0, 0)       # ... and back: same plan, new geometry
            for k in range(4):
                eng.block_dev(raw_in[k % 2], raw_out)
            eng.sync()
            eng.enable_timing(True)
            for k in range(40):
                eng.block_dev(raw_in[k % 2], raw_out)
            t = eng.timing()
            eng.enable_timing(False)
            res[c].append((t["mac_ms"], t["fft_in_ms"] + t["mac_ms"] + t["ifft_out_ms"]))
    for c in cands:
        mac = sorted(m for m, _ in res[c])
        tot = sorted(s for _, s in res[c])
        print((("grid " + c + "  ") if grid else knob + " %5d" % c) + ": mac median %.4f ms (min %.4f) = %.0f GB/s | K1+K2+K3 median %.4f ms"
              % (mac[len(mac) // 2], mac[0], alg / mac[len(mac) // 2] / 1e6, tot[len(tot) // 2]))


if __name__ == "__main__":
    main()
