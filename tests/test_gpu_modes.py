"""GPU: the engine's optional fast paths are chosen from the shape of the configuration -- wave FFT
from L = 4096, stream-ordered coefficient copy from 64 MiB of a uniform crossbar, deferred output
for long MACs -- so the small networks of the feature and fuzz tests would never reach them.  This
test re-runs those files ONCE in a child interpreter with every fast path forced on wherever it is
legal (BFHIP_FFT_WAVE=1, BFHIP_COEFF_STREAM=2, BFHIP_DEFER=1 with the side-stream pipeline off):
same oracle, same tolerances, different kernels and memory layouts underneath."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("schedule", ["deferred", "pingpong"])
def test_feature_and_fuzz_suites_with_every_fast_path_forced(hip, schedule):
    # ... and BFHIP_WIDE_IO=1: every uniform interleaved side goes through the coalesced frame <-> planar
    # transposes that sides of >= 128 channels get by default (transpose_words_kernel)
    env = dict(os.environ, BFHIP_FFT_WAVE="1", BFHIP_COEFF_STREAM="2", BFHIP_FUZZ_SEEDS="12", BFHIP_WIDE_IO="1")
    if schedule == "deferred":
        env.update(BFHIP_DEFER="1", BFHIP_OVERLAP="0")     # [K3 of t-1 | K1 of t] fused, one stream
    else:
        env.update(BFHIP_OVERLAP="1")                      # [K3 of t-2 | K1 of t] on a side stream beside the MAC
    files = ["test_gpu_engine.py", "test_gpu_features.py", "test_gpu_fuzz.py", "test_gpu_refconfigs.py",
             "test_gpu_numpy.py", "test_gpu_fullsize.py", "test_gpu_rt.py", "test_gpu_shards.py", "test_gpu_diag.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] +
                       [os.path.join(ROOT, "tests", f) for f in files],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    tail = "\n".join(r.stdout.splitlines()[-25:])
    assert r.returncode == 0, tail + r.stderr[-2000:]
    assert " passed" in tail and "failed" not in tail, tail
