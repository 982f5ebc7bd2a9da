"""CPU: the oracle's integer sample delay and Kaiser window against the REFERENCE'S delay.c /
firwindow.c compiled into oracle/_ref (live; these tests skip where _ref is not built, the
committed fixture covers that case)."""
import ctypes as C
import os

import numpy as np
import pytest

import bforacle as bo

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_delay.npz")


def _oracle_run(F, ss, init, maxd, delays, data):
    L = bo.lib()
    L.bfo_delay_new.restype = C.c_void_p
    L.bfo_delay_new.argtypes = [C.c_int] * 4
    L.bfo_delay_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.bfo_delay_free.argtypes = [C.c_void_p]
    d = L.bfo_delay_new(F, init, maxd, ss)
    out = []
    for blk, dl in zip(data, delays):
        b = blk.copy()
        L.bfo_delay_update(d, b.ctypes.data, int(dl))
        out.append(b)
    L.bfo_delay_free(d)
    return np.stack(out)


def _ref_run(R, F, ss, init, maxd, delays, data):
    d = R.ref_delay_allocate(F, init, maxd, ss)
    out = []
    for blk, dl in zip(data, delays):
        b = blk.copy()
        R.ref_delay_update(d, b.ctypes.data, ss, 1, int(dl), None)
        out.append(b)
    return np.stack(out)


CASES = [  # fragment, sample size, initial delay, max delay, delay per block
    (32, 4, 0, 0, [0] * 6),
    (32, 4, 5, -1, [5] * 8),                       # fixed short delay
    (32, 2, 32, -1, [32] * 8),                     # exactly one fragment
    (32, 8, 75, -1, [75] * 10),                    # fixed long delay with a rest
    (32, 4, 64, -1, [64] * 10),                    # whole fragments, no rest
    (32, 4, 3, 100, [3, 3, 10, 10, 10, 2, 2, 40, 40, 40, 70, 70, 70, 70, 33, 33, 33, 0, 0, 96, 96, 96, 96, 5, 5]),
    # (3-byte samples are left out: the reference's shift_samples() never terminates for
    #  sample_size 3 -- delay.c:205-212 advances n by 3 and steps it back by 3 -- so packed
    #  24-bit formats cannot be used on delayed N:1 channels there)
    (16, 2, 0, 64, [0, 0, 16, 16, 17, 17, 17, 64, 64, 64, 64, 64, 64, 1, 1, 200, 1]),   # 200 > max: ignored
]


def _data(case, seed=0):
    F, ss, init, maxd, delays = case
    rng = np.random.default_rng(seed)
    return rng.integers(1, 255, (len(delays), F * ss), dtype=np.uint8)


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_delay_matches_reference_live(idx):
    R = bo.ref_delay()
    if R is None:
        pytest.skip("oracle/_ref not built here")
    case = CASES[idx]
    data = _data(case, idx)
    F, ss, init, maxd, delays = case
    assert np.array_equal(_oracle_run(F, ss, init, maxd, delays, data), _ref_run(R, F, ss, init, maxd, delays, data))


def test_delay_matches_committed_reference_fixture():
    g = np.load(G)
    for idx, case in enumerate(CASES):
        F, ss, init, maxd, delays = case
        got = _oracle_run(F, ss, init, maxd, delays, g["in%d" % idx])
        assert np.array_equal(got, g["out%d" % idx]), idx


def test_steady_state_is_a_pure_delay_line():
    F, ss, D = 32, 4, 75
    n = 12
    x = np.arange(1, n * F + 1, dtype=np.int32)
    blocks = x.reshape(n, F).view(np.uint8).reshape(n, F * ss)
    y = _oracle_run(F, ss, D, -1, [D] * n, blocks).view(np.int32).ravel()
    assert np.array_equal(y[D:], x[:-D]) and not y[:D].any()


@pytest.mark.parametrize("rs", [4, 8])
def test_kaiser_window_matches_reference(rs):
    g = np.load(G)
    dt = np.float32 if rs == 4 else np.float64
    bo.lib().bfo_firwindow_kaiser.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int]
    for k, (ln, off) in enumerate([(63, 0.0), (63, 0.37), (63, -0.25), (64, 0.0), (31, 0.99), (31, -0.01)]):
        t = np.ones(ln, dt)
        bo.lib().bfo_firwindow_kaiser(t.ctypes.data, ln, off, 9.0, rs)
        assert np.array_equal(t, g["kaiser%d_%d" % (rs, k)]), (ln, off)
