"""Build libbfhip.so (the HIP engine + its C ABI) in-tree for gfx950.

    python -m brutefir_amd.build            # build if sources are newer than the .so
    python -m brutefir_amd.build --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbfhip.so")
SOURCES = ["bfhip.hip", "convolver_abi.hip", "nupc.hip"]
DEPS = ["bfhip.hip", "convolver_abi.hip", "nupc.hip", "kernels.h", "fft_lds.h", "bigfft.h",
        os.path.join("..", "..", "include", "bfhip_nupc.h"),
        os.path.join("..", "..", "include", "bfhip.h"),
        os.path.join("..", "..", "include", "bfhip_convolver.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS
               if os.path.exists(os.path.join(CSRC, d)))


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
