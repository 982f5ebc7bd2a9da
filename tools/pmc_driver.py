#!/usr/bin/env python3
"""Minimal driver for rocprofv3 --pmc passes: runs a few steady-state blocks of a workload
through the C ABI with host buffers only (no torch in the process -- rocprofv3's counter
collection crashed with torch loaded on this pool).  Coefficient VALUES are irrelevant for
traffic, so one seeded IR is reused for every filter (each still gets its own HBM copy).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- \
        python3 tools/pmc_driver.py C 3
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import brutefir_amd as bf  # noqa: E402

WL = {"C": (64, 64, 8192, 32, 4), "B": (8, 8, 8192, 8, 4), "S": (16, 16, 8192, 32, 4),
      "F": (32, 32, 8192, 32, 8),       # F: the float64 crossbar of bench.py --workload F
      "D": (256, 256, 8192, 16, 4)}     # D: 256 one-to-one filters (BASELINE configs[3] on one GPU)
DIAGONAL = {"D"}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C"
    steady = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    I, O, L, N, rs = WL[name]
    fmt = "S24_4LE" if rs == 4 else "FLOAT64_LE"
    e = bf.Engine(L, N, rs, I, O)
    e.set_interleaved(bf.IN, fmt)
    e.set_interleaved(bf.OUT, fmt)
    rng = np.random.default_rng(5)
    h = rng.standard_normal(L * N) * np.exp(-np.arange(L * N) / (L * N / 6.0))
    h = (h / (np.abs(h).sum() * I)).astype(np.float32 if rs == 4 else np.float64)
    n_sets = O if name in DIAGONAL else I * O
    e.reserve_coeffs(float(n_sets) * N * 2 * L * rs)
    for o in range(O):
        for i in range(I):
            if name in DIAGONAL and i != o:
                continue
            e.add_filter(in_ch=[i], out_ch=[o], coeff=e.add_coeff(h))
    e.finalize()
    raw = ((rng.standard_normal((L, I)) * 0.1 * 8388608).astype(np.int32) if rs == 4
           else rng.standard_normal((L, I)) * 0.1)
    for _ in range(N + steady):
        st, _out = e.block(raw)
        assert st == 0
    print("pmc_driver: %s done, %d blocks, alg bytes %s" % (name, N + steady, e.algorithmic_bytes()))


if __name__ == "__main__":
    main()
