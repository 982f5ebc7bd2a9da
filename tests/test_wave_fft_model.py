"""CPU: a numpy model of the wave-level FFT's data flow (brutefir_amd/csrc/fft_wave.h), thread by
thread and register by register, checked against numpy.fft -- and the twiddle table the library
builds for the device kernels checked against the twiddles that model needs.

The model is the algorithm the kernels implement: 16 points per thread, NT = L/16 threads;
  P0  radix R0 = L/512 on elements j + r*L/R0, j = tid + b*NT            -> s[j*R0 + r]
  P1  radix 8, Ns = R0, j = tid + b*NT (ordinary autosort pass through "LDS")
  P2  radix 8, Ns = L/64, j = k + Ns*q with q = lane bits 3..5 of tid, k = the other bits (+ NT/8 * b)
      -- its result in register r of thread (k, q) is P3's input r' = q of thread (k, q' = r):
      a transpose between register index and the lane digit, three butterfly exchange steps
  P3  radix 8, Ns = L/8, in place.
The device does the exchange with DPP row_ror:8 / v_permlane16_swap / v_permlane32_swap; here it is
the same three steps on arrays."""
import ctypes as C

import numpy as np
import pytest


def wave_j(tid, b, L):
    NT, NS2 = L // 16, L // 64
    return (tid & 7) + ((tid >> 6) << 3) + (NT // 8) * b + NS2 * ((tid >> 3) & 7)


def dft(u, sign):
    n = len(u)
    k = np.arange(n)
    return np.exp(sign * 2j * np.pi * np.outer(k, k) / n) @ u


def octet_transpose(regs):
    """regs[tid][r] -> same butterfly steps as fft_wave.h octet_transpose: partner lane ^ 8, 16, 32"""
    NT = regs.shape[0]
    v = regs.copy()
    for m, lane_bit in ((1, 3), (2, 4), (4, 5)):
        new = v.copy()
        for tid in range(NT):
            partner = tid ^ (1 << lane_bit)
            bit = (tid >> lane_bit) & 1
            for i in range(8):
                if i & m:
                    continue
                # lanes with the bit clear receive the partner's v[i] into v[i|m]; the others the
                # partner's v[i|m] into v[i]
                if bit:
                    new[tid, i] = v[partner, i | m]
                else:
                    new[tid, i | m] = v[partner, i]
        v = new
    return v


def model_fft(z, inverse, twiddles=None):
    """returns (X, the twiddles used, as {pass: array[tid, b, 3]} of w, w^2, w^4)"""
    L = len(z)
    NT, R0 = L // 16, L // 512
    B0, T0, T8, NS2 = 16 // R0, L // R0, L // 8, L // 64
    sign = 1.0 if inverse else -1.0
    used = {p: np.zeros((NT, 2, 3), complex) for p in (1, 2, 3)}

    def tw(p, tid, b, turns):
        w = [np.exp(-2j * np.pi * turns * (1 << i)) for i in range(3)]
        used[p][tid, b] = w
        if twiddles is not None:
            w = list(twiddles[p][tid, b])
        if inverse:
            w = [np.conj(x) for x in w]
        w1, w2, w4 = w
        w3 = w1 * w2
        return np.array([1, w1, w2, w3, w4, w4 * w1, w4 * w2, w4 * w3])

    s = np.zeros(L, complex)
    for tid in range(NT):
        for b in range(B0):
            j = tid + b * NT
            s[j * R0:(j + 1) * R0] = dft(z[j + np.arange(R0) * T0], sign)
    s2 = np.zeros(L, complex)
    for tid in range(NT):
        for b in range(2):
            j = tid + b * NT
            k = j % R0
            v = dft(s[j + np.arange(8) * T8] * tw(1, tid, b, k / (R0 * 8)), sign)
            s2[(j - k) * 8 + k + np.arange(8) * R0] = v
    s = s2
    out = np.zeros(L, complex)
    for b in range(2):
        regs = np.zeros((NT, 8), complex)
        for tid in range(NT):
            j = wave_j(tid, b, L)
            k = j % NS2
            regs[tid] = dft(s[j + np.arange(8) * T8] * tw(2, tid, b, k / (NS2 * 8)), sign)
        regs = octet_transpose(regs)
        for tid in range(NT):
            j = wave_j(tid, b, L)
            out[j + np.arange(8) * T8] = dft(regs[tid] * tw(3, tid, b, j / L), sign)
    return out, used


@pytest.mark.parametrize("log2l", [10, 11, 12, 13])
@pytest.mark.parametrize("inverse", [False, True])
def test_model_is_a_natural_order_fft(log2l, inverse):
    L = 1 << log2l
    rng = np.random.default_rng(log2l)
    z = rng.standard_normal(L) + 1j * rng.standard_normal(L)
    X, _ = model_fft(z, inverse)
    want = np.fft.ifft(z) * L if inverse else np.fft.fft(z)
    assert np.abs(X - want).max() <= 1e-11 * np.abs(want).max()
    # every butterfly index of the lane-remapped passes is hit exactly once
    js = sorted(wave_j(t, b, L) for t in range(L // 16) for b in range(2))
    assert js == list(range(L // 8))


@pytest.mark.parametrize("log2l", [10, 11, 12, 13])
@pytest.mark.parametrize("rs", [4, 8])
def test_library_twiddle_table_is_what_the_model_needs(log2l, rs):
    import brutefir_amd as bf
    lib = bf.lib()
    lib.bfhip_selftest_wave_twiddles.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int]
    n = lib.bfhip_selftest_wave_twiddles(log2l, rs, None, 0)
    L, NT = 1 << log2l, (1 << log2l) // 16
    assert n == (2 * L + 18 * NT) * 2 * rs
    buf = np.empty(n // rs, np.float32 if rs == 4 else np.float64)
    assert lib.bfhip_selftest_wave_twiddles(log2l, rs, buf.ctypes.data, n) == n
    t = buf.astype(np.float64).view(np.complex128)
    tol = 1e-7 if rs == 4 else 1e-15
    m = np.arange(2 * L)
    assert np.abs(t[:2 * L] - np.exp(-2j * np.pi * m / (2 * L))).max() <= tol        # untangle part
    regs = t[2 * L:].reshape(18, NT)                                                 # [register][thread]
    table = {p: np.zeros((NT, 2, 3), complex) for p in (1, 2, 3)}
    for p, base in ((1, 0), (2, 6), (3, 12)):
        for b in range(2):
            for i in range(3):
                table[p][:, b, i] = regs[base + b * 3 + i]
    rng = np.random.default_rng(5)
    z = rng.standard_normal(L) + 1j * rng.standard_normal(L)
    X, used = model_fft(z, False, twiddles=table)             # the model runs ON the library's table
    for p in (1, 2, 3):
        assert np.abs(table[p] - used[p]).max() <= tol, p
    assert np.abs(X - np.fft.fft(z)).max() <= (2e-5 if rs == 4 else 1e-11) * np.abs(np.fft.fft(z)).max()
    assert lib.bfhip_selftest_wave_twiddles(9, 4, None, 0) < 0                        # not covered
