#!/usr/bin/env python3
"""`powersave: true` on config C: block time with all 64 inputs live vs. only a few.  The reference
skips the convolution of filters whose input buffers are zero (bfrun.c:1694-1770); on the device
the MAC skips the coefficient stream of inputs that have been silent for a whole filter length."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import brutefir_amd as bf  # noqa: E402


def main():
    I, O, L, N, rs, fmt = bench.WORKLOADS["C"]
    dev = torch.device("cuda", 0)
    eng = bf.Engine(L, N, rs, I, O)
    eng.set_interleaved(bf.IN, fmt)
    eng.set_interleaved(bf.OUT, fmt)
    eng.set_powersave(1.0)
    h = bench.synth_ir_dev(torch, 1, L * N, I, dev)
    for o in range(O):
        for i in range(I):
            eng.add_filter(in_ch=[i], out_ch=[o], coeff=eng.add_coeff_dev(h, L * N))
    torch.cuda.synchronize()
    eng.finalize()
    raw = bench.synth_raw_blocks(torch, 2, L, I, dev)
    raw_out = torch.zeros(L, O, dtype=torch.int32, device=dev)
    for live in (64, 32, 8, 1, 0):
        x = raw.clone()
        x[:, :, live:] = 0
        for k in range(N + 4):                       # let the silence reach every ring slot
            eng.block_dev(x[k % 2], raw_out)
        eng.sync()
        eng.enable_timing(True)
        for k in range(40):
            eng.block_dev(x[k % 2], raw_out)
        t = eng.timing()
        eng.enable_timing(False)
        print("live inputs %2d: mac %.4f ms, K1+K2+K3 %.4f ms" % (live, t["mac_ms"], t["fft_in_ms"] + t["mac_ms"] + t["ifft_out_ms"]), flush=True)


if __name__ == "__main__":
    main()
