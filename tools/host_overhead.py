#!/usr/bin/env python3
"""How long does the HOST take to enqueue one block (bfhip_engine_block_dev), compared with
what the GPU takes to execute it?  For small shapes the enqueue side is the limit.

    python tools/host_overhead.py [B|E|D|C8 ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import brutefir_amd as bf  # noqa: E402


def main():
    for wl in (sys.argv[1:] or ["B"]):
        I, O, L, N, rs, fmt = bench.WORKLOADS[wl]
        dev = torch.device("cuda", 0)
        eng = bf.Engine(L, N, rs, I, O)
        eng.set_interleaved(bf.IN, fmt)
        eng.set_interleaved(bf.OUT, fmt)
        h = bench.synth_ir_dev(torch, 1, L * N, I, dev).to(torch.float32 if rs == 4 else torch.float64)
        diag = wl in bench.DIAGONAL
        for o in range(O):
            for i in range(I):
                if diag and i != o:
                    continue
                eng.add_filter(in_ch=[i], out_ch=[o], coeff=eng.add_coeff_dev(h, L * N))
        torch.cuda.synchronize()
        eng.finalize()
        eng.prewarm()
        if fmt == "S24_4LE":
            raw_in = bench.synth_raw_blocks(torch, 2, L, I, dev)
            raw_out = torch.zeros(L, O, dtype=torch.int32, device=dev)
        else:
            raw_in = (torch.randn(2, L, I, device=dev, dtype=torch.float64) * 0.1).contiguous()
            raw_out = torch.zeros(L, O, dtype=torch.float64, device=dev)
        for k in range(50):
            eng.block_dev(raw_in[k & 1], raw_out)
        eng.sync()
        n = 2000
        t0 = time.perf_counter()
        for k in range(n):
            eng.block_dev(raw_in[k & 1], raw_out)
        t1 = time.perf_counter()
        eng.sync()
        t2 = time.perf_counter()
        print("%s: host enqueue %.1f us/block, enqueue+drain %.1f us/block" % (wl, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
