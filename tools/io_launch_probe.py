#!/usr/bin/env python3
"""How long do the transform launches take on their own?  K1 (bfhip_engine_inputs_dev), K3
(bfhip_engine_outputs_dev) and the fused [K3 | K1] launch (bfhip_engine_outputs_inputs_dev) of a
C x C crossbar at L = 8192, timed back to back with events on the engine's stream.

    python tools/io_launch_probe.py            (on the GPU box)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.cuda.init()
import brutefir_amd as bf  # noqa: E402

dev = torch.device("cuda", 0)
L, N = 8192, 2


def probe(rs, C, reps=200):
    e = bf.Engine(L, N, rs, C, C)
    fmt = "S24_4LE" if rs == 4 else "FLOAT64_LE"
    e.set_interleaved(0, fmt)
    e.set_interleaved(1, fmt)
    h = np.zeros(L * N, np.float32 if rs == 4 else np.float64)
    h[0] = 1.0
    c = e.add_coeff(h)
    for i in range(C):
        e.add_filter(in_ch=[i], out_ch=[i], coeff=c)
    e.finalize()
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    tdt = torch.int32 if rs == 4 else torch.float64
    src = torch.zeros(L, C, dtype=tdt, device=dev)
    out = torch.zeros(L, C, dtype=tdt, device=dev)
    z = torch.zeros(C, L, 2, dtype=torch.float32 if rs == 4 else torch.float64, device=dev)
    res = {}
    for name, fn in (("K1", lambda: e.inputs_dev(src)), ("K3", lambda: e.outputs_dev(z, 0, C, out)),
                     ("fused", lambda: e.outputs_inputs_dev(z, 0, C, out, src))):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        res[name] = round(a.elapsed_time(b) / reps * 1e3, 2)
    return res


for rs in (4, 8):
    for C in (1, 2, 8, 64, 256):
        if rs == 8 and C == 256:
            continue
        print(json.dumps({"realsize": rs, "channels": C, "us_per_launch_back_to_back": probe(rs, C)}), flush=True)
