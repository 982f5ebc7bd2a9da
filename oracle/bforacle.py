"""ctypes bindings for the CPU oracle (oracle/libbforacle.so) and, when present, for the
reference's own inner loops (oracle/_ref/libbfref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (brutefir_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libbforacle.so")
REF_SO = os.path.join(HERE, "_ref", "libbfref.so")
REF_DELAY_SO = os.path.join(HERE, "_ref", "libbfref_delay.so")

MIX_INPUT, MIX_OUTPUT = 1, 3


class Overflow(C.Structure):
    """struct bfoverflow, bfmod.h:99-104"""
    _fields_ = [("n_overflows", C.c_uint), ("intlargest", C.c_int32),
                ("largest", C.c_double), ("max", C.c_double)]

    def astuple(self):
        return (self.n_overflows, self.intlargest, self.largest, self.max)


class Format(C.Structure):
    """struct sample_format + struct buffer_format, dai.h:21-34"""
    _fields_ = [("isfloat", C.c_int), ("swap", C.c_int), ("bytes", C.c_int),
                ("sbytes", C.c_int), ("scale", C.c_double),
                ("sample_spacing", C.c_int), ("byte_offset", C.c_int)]


# name -> (bytes, sbytes, isfloat, little_endian); bfconf.c:358-480 (the *_NE macro
# formats are left out on purpose: two of them are mis-parsed by the reference)
SAMPLE_FORMATS = {
    "S8": (1, 1, 0, True),
    "S16_LE": (2, 2, 0, True), "S16_BE": (2, 2, 0, False),
    "S24_LE": (3, 3, 0, True), "S24_BE": (3, 3, 0, False),
    "S24_4LE": (4, 3, 0, True), "S24_4BE": (4, 3, 0, False),
    "S32_LE": (4, 4, 0, True), "S32_BE": (4, 4, 0, False),
    "FLOAT_LE": (4, 4, 1, True), "FLOAT_BE": (4, 4, 1, False),
    "FLOAT64_LE": (8, 8, 1, True), "FLOAT64_BE": (8, 8, 1, False),
}


def make_format(name, sample_spacing=1, byte_offset=0):
    nbytes, sbytes, isfloat, le = SAMPLE_FORMATS[name]
    scale = 1.0 if isfloat else 1.0 / float(1 << (8 * sbytes - 1))
    return Format(isfloat, 0 if le else 1, nbytes, sbytes, scale, sample_spacing, byte_offset)


def interleaved_formats(name, n_channels):
    """buffer_format of an interleaved device with n_channels open (dai.c:537-576)."""
    nbytes = SAMPLE_FORMATS[name][0]
    return [make_format(name, n_channels, c * nbytes) for c in range(n_channels)]


def build(force=False):
    """(Re)build the oracle (always possible) and oracle/_ref (only where the reference
    sources exist).  Building the checker is not using it."""
    if force or not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(os.path.join(HERE, f))
                                              for f in ("bf_oracle.c", "bf_oracle_ops.inc",
                                                        "bf_oracle.h")):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.exists("/root/reference/fftw_convfuns.h"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _dtype(realsize):
    return np.float32 if realsize == 4 else np.float64


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(ORACLE_SO)
        vp, ci, cd = C.c_void_p, C.c_int, C.c_double
        L.bfo_ctx_new.restype = vp
        L.bfo_ctx_new.argtypes = [ci, ci]
        L.bfo_ctx_free.argtypes = [vp]
        L.bfo_cbufsize.argtypes = [vp]
        L.bfo_raw2real.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci]
        L.bfo_raw2cbuf.argtypes = [vp, vp, vp, vp, C.POINTER(Format)]
        for f in ("bfo_time2freq", "bfo_freq2time", "bfo_dirac_convolve",
                  "bfo_convolve_inplace", "bfo_runtime_coeffs2cbuf"):
            getattr(L, f).argtypes = [vp, vp, vp]
        for f in ("bfo_convolve", "bfo_convolve_add", "bfo_crossfade_inplace",
                  "bfo_convolve_eval"):
            getattr(L, f).argtypes = [vp, vp, vp, vp]
        L.bfo_dirac_convolve_inplace.argtypes = [vp, vp]
        L.bfo_mixnscale.argtypes = [vp, C.POINTER(vp), vp, C.POINTER(cd), ci, ci]
        L.bfo_coeffs2cbuf.argtypes = [vp, vp, ci, cd, vp]
        L.bfo_verify_cbuf.argtypes = [vp, C.POINTER(vp), ci]
        L.bfo_dither_init.argtypes = [vp, ci, ci, ci, ci]
        L.bfo_dither_table.restype = C.POINTER(C.c_int8)
        L.bfo_dither_table.argtypes = [vp, C.POINTER(ci)]
        L.bfo_dither_randtab_ptr.argtypes = [vp, ci]
        L.bfo_cbuf2raw.argtypes = [vp, vp, vp, C.POINTER(Format), ci, C.POINTER(Overflow), cd]
        L.bfo_engine_new.restype = vp
        L.bfo_engine_new.argtypes = [ci] * 5
        L.bfo_engine_free.argtypes = [vp]
        L.bfo_engine_set_format.argtypes = [vp, ci, ci, C.POINTER(Format)]
        L.bfo_engine_set_safety_limit.argtypes = [vp, cd]
        L.bfo_engine_set_powersave.argtypes = [vp, cd]
        L.bfo_engine_enable_dither.argtypes = [vp, C.POINTER(ci), ci, ci, ci]
        L.bfo_engine_map_channels.argtypes = [vp, ci, ci, C.POINTER(ci)]
        for f in ("bfo_engine_set_delay", "bfo_engine_set_maxdelay", "bfo_engine_set_mute"):
            getattr(L, f).argtypes = [vp, ci, ci, ci]
        L.bfo_engine_enable_subdelay.argtypes = [vp, ci, cd]
        L.bfo_engine_set_subdelay.argtypes = [vp, ci, ci, ci]
        L.bfo_engine_add_coeff.argtypes = [vp, vp, ci, cd, ci]
        ip, dp = C.POINTER(ci), C.POINTER(cd)
        L.bfo_engine_add_filter.argtypes = [vp, ci, ip, dp, ci, ip, dp, ci, ip, dp, ci, ci, ci]
        L.bfo_engine_set_coeff.argtypes = [vp, ci, ci]
        L.bfo_engine_set_delayblocks.argtypes = [vp, ci, ci]
        L.bfo_engine_set_scale.argtypes = [vp, ci, ci, ci, cd]
        L.bfo_engine_set_fscale.argtypes = [vp, ci, ci, cd]
        L.bfo_engine_block.argtypes = [vp, vp, vp]
        L.bfo_engine_get_overflow.argtypes = [vp, ci, C.POINTER(Overflow)]
        L.bfo_engine_blockcounter.restype = C.c_uint
        L.bfo_engine_blockcounter.argtypes = [vp]
        L.bfo_engine_filter_output.restype = vp
        L.bfo_engine_filter_output.argtypes = [vp, ci]
        L.bfo_engine_output_spectrum.restype = vp
        L.bfo_engine_output_spectrum.argtypes = [vp, ci]
        _lib = L
    return _lib


def _iarr(v):
    return (C.c_int * max(len(v), 1))(*v)


def _darr(v):
    return (C.c_double * max(len(v), 1))(*v)


class Ctx:
    """Op-level oracle (one convolver_init worth of state)."""

    def __init__(self, length, realsize=4):
        self.L, self.rs = length, realsize
        self.dt = _dtype(realsize)
        self.h = lib().bfo_ctx_new(length, realsize)
        if not self.h:
            raise ValueError("invalid length/realsize")

    def __del__(self):
        if getattr(self, "h", None):
            lib().bfo_ctx_free(self.h)
            self.h = None

    def _new(self):
        return np.empty(2 * self.L, self.dt)

    def raw2real(self, raw, bytes_, isfloat, spacing, swap, n):
        out = np.empty(n, self.dt)
        lib().bfo_raw2real(self.h, _ptr(out), _ptr(raw), bytes_, isfloat, spacing, swap, n)
        return out

    def time2freq(self, x):
        out = self._new()
        lib().bfo_time2freq(self.h, _ptr(np.ascontiguousarray(x, self.dt)), _ptr(out))
        return out

    def freq2time(self, x):
        out = self._new()
        lib().bfo_freq2time(self.h, _ptr(np.ascontiguousarray(x, self.dt)), _ptr(out))
        return out

    def mixnscale(self, bufs, scales, mode):
        bufs = [np.ascontiguousarray(b, self.dt) for b in bufs]
        arr = (C.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
        out = self._new()
        lib().bfo_mixnscale(self.h, arr, _ptr(out), _darr(list(scales)), len(bufs), mode)
        return out

    def convolve(self, b, h):
        d = self._new()
        lib().bfo_convolve(self.h, _ptr(b), _ptr(h), _ptr(d))
        return d

    def convolve_inplace(self, b, h):
        b = b.copy()
        lib().bfo_convolve_inplace(self.h, _ptr(b), _ptr(h))
        return b

    def convolve_add(self, b, h, d):
        d = d.copy()
        lib().bfo_convolve_add(self.h, _ptr(b), _ptr(h), _ptr(d))
        return d

    def dirac_convolve(self, b):
        d = self._new()
        lib().bfo_dirac_convolve(self.h, _ptr(b), _ptr(d))
        return d

    def crossfade_inplace(self, new, old):
        new, old = new.copy(), old.copy()
        buf = self._new()
        lib().bfo_crossfade_inplace(self.h, _ptr(new), _ptr(old), _ptr(buf))
        return new

    def convolve_eval(self, x, state):
        """state: 3L reals, updated in place"""
        out = self._new()
        lib().bfo_convolve_eval(self.h, _ptr(x), _ptr(state), _ptr(out))
        return out

    def coeffs2cbuf(self, taps, scale=1.0):
        taps = np.ascontiguousarray(taps, self.dt)
        out = self._new()
        ok = lib().bfo_coeffs2cbuf(self.h, _ptr(taps), len(taps), scale, _ptr(out))
        return out if ok else None

    def runtime_coeffs2cbuf(self, taps):
        taps = np.ascontiguousarray(taps, self.dt)
        assert len(taps) == self.L
        out = self._new()
        lib().bfo_runtime_coeffs2cbuf(self.h, _ptr(taps), _ptr(out))
        return out

    def dither_init(self, n_channels, sample_rate, max_size=0):
        return lib().bfo_dither_init(self.h, n_channels, sample_rate, max_size, self.L)

    def dither_table(self):
        n = C.c_int()
        p = lib().bfo_dither_table(self.h, C.byref(n))
        return np.ctypeslib.as_array(p, (n.value,)).copy()

    def cbuf2raw(self, x, fmt, out_bytes, overflow, dither_channel=-1, safety_limit=0.0):
        x = np.ascontiguousarray(x, self.dt)
        raw = np.zeros(out_bytes, np.uint8)
        st = lib().bfo_cbuf2raw(self.h, _ptr(x), _ptr(raw), C.byref(fmt), dither_channel,
                                C.byref(overflow), safety_limit)
        return st, raw


class Engine:
    """Block-level oracle: one filter_process() worth of state (bfrun.c:1420-2083)."""

    def __init__(self, length, n_blocks, realsize, n_in, n_out):
        self.L, self.N, self.rs, self.n_in, self.n_out = length, n_blocks, realsize, n_in, n_out
        self.dt = _dtype(realsize)
        self.h = lib().bfo_engine_new(length, n_blocks, realsize, n_in, n_out)
        if not self.h:
            raise ValueError("invalid engine parameters")
        self.out_bytes = n_out * length * realsize
        self.in_bytes = n_in * length * realsize      # what block() reads: checked there
        self.fmt = [[None] * n_in, [None] * n_out]

    def __del__(self):
        if getattr(self, "h", None):
            lib().bfo_engine_free(self.h)
            self.h = None

    def set_format(self, io, ch, fmt):
        lib().bfo_engine_set_format(self.h, io, ch, C.byref(fmt))
        self.fmt[io][ch] = fmt
        if io == 0:
            # the C side reads byte_offset + ((L-1)*spacing + 1)*bytes of the raw buffer per channel
            # (channels without a format: planar reals, bfo_engine_new)
            ends = []
            for c, f in enumerate(self.fmt[0]):
                ends.append((c + 1) * self.L * self.rs if f is None else
                            f.byte_offset + ((self.L - 1) * f.sample_spacing + 1) * f.bytes)
            self.in_bytes = max(ends)

    def set_interleaved(self, io, name):
        n = self.n_in if io == 0 else self.n_out
        for c, f in enumerate(interleaved_formats(name, n)):
            self.set_format(io, c, f)
        if io == 1:
            self.out_bytes = n * self.L * SAMPLE_FORMATS[name][0]
        else:
            self.in_bytes = n * self.L * SAMPLE_FORMATS[name][0]

    def set_powersave(self, analog_powersave):
        lib().bfo_engine_set_powersave(self.h, analog_powersave)

    def set_safety_limit(self, v):
        lib().bfo_engine_set_safety_limit(self.h, v)

    def map_channels(self, io, virt2phys):
        """N:1 virtual -> physical mapping; afterwards formats address physical channels"""
        n_phys = max(virt2phys) + 1
        if not lib().bfo_engine_map_channels(self.h, io, n_phys, _iarr(list(virt2phys))):
            raise ValueError("bad channel mapping")
        if io == 0:
            self.n_phys_in = n_phys
        else:
            self.n_phys_out = n_phys

    def set_interleaved_phys(self, io, name, n_phys):
        for c, f in enumerate(interleaved_formats(name, n_phys)):
            self.set_format(io, c, f)
        if io == 1:
            self.out_bytes = n_phys * self.L * SAMPLE_FORMATS[name][0]
        else:
            self.in_bytes = n_phys * self.L * SAMPLE_FORMATS[name][0]

    def set_delay(self, io, ch, delay):
        lib().bfo_engine_set_delay(self.h, io, ch, delay)

    def set_maxdelay(self, io, ch, maxdelay):
        lib().bfo_engine_set_maxdelay(self.h, io, ch, maxdelay)

    def set_mute(self, io, ch, muted):
        lib().bfo_engine_set_mute(self.h, io, ch, int(muted))

    def enable_subdelay(self, sdf_length, beta=9.0):
        if not lib().bfo_engine_enable_subdelay(self.h, sdf_length, beta):
            raise ValueError("invalid sdf_length")

    def set_subdelay(self, io, ch, subdelay):
        lib().bfo_engine_set_subdelay(self.h, io, ch, subdelay)

    def enable_dither(self, channels, sample_rate, max_size=0):
        return lib().bfo_engine_enable_dither(self.h, _iarr(channels), len(channels),
                                              sample_rate, max_size)

    def add_coeff(self, taps, scale=1.0, n_blocks=0):
        taps = np.ascontiguousarray(taps, self.dt)
        r = lib().bfo_engine_add_coeff(self.h, _ptr(taps), len(taps), scale, n_blocks)
        if r < 0:
            raise ValueError("coefficient set rejected")
        return r

    def add_filter(self, in_ch=(), in_scale=None, in_f=(), in_fscale=None, out_ch=(),
                   out_scale=None, coeff=-1, delayblocks=0, crossfade=False):
        in_scale = [1.0] * len(in_ch) if in_scale is None else list(in_scale)
        in_fscale = [1.0] * len(in_f) if in_fscale is None else list(in_fscale)
        out_scale = [1.0] * len(out_ch) if out_scale is None else list(out_scale)
        r = lib().bfo_engine_add_filter(
            self.h, len(in_ch), _iarr(list(in_ch)), _darr(in_scale),
            len(in_f), _iarr(list(in_f)), _darr(in_fscale),
            len(out_ch), _iarr(list(out_ch)), _darr(out_scale),
            coeff, delayblocks, int(crossfade))
        if r < 0:
            raise ValueError("filter rejected")
        return r

    def set_coeff(self, f, c):
        lib().bfo_engine_set_coeff(self.h, f, c)

    def set_delayblocks(self, f, d):
        lib().bfo_engine_set_delayblocks(self.h, f, d)

    def set_scale(self, f, io, idx, v):
        lib().bfo_engine_set_scale(self.h, f, io, idx, v)

    def set_fscale(self, f, idx, v):
        lib().bfo_engine_set_fscale(self.h, f, idx, v)

    def block(self, rawin):
        rawin = np.ascontiguousarray(rawin).view(np.uint8).ravel()
        # the C side trusts the formats: a short buffer (int32 frames handed to a FLOAT64 engine, the
        # r02 bench segfault) must stop here, as the product binding stops it
        assert rawin.size >= self.in_bytes, (rawin.size, self.in_bytes)
        out = np.zeros(self.out_bytes, np.uint8)
        st = lib().bfo_engine_block(self.h, _ptr(rawin), _ptr(out))
        return st, out

    def overflow(self, ch):
        of = Overflow()
        lib().bfo_engine_get_overflow(self.h, ch, C.byref(of))
        return of

    def filter_output(self, f):
        p = lib().bfo_engine_filter_output(self.h, f)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float if self.rs == 4
                                                          else C.c_double)),
                                     (2 * self.L,)).copy()

    def output_spectrum(self, ch):
        p = lib().bfo_engine_output_spectrum(self.h, ch)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float if self.rs == 4
                                                          else C.c_double)),
                                     (2 * self.L,)).copy()


# ---------------------------------------------------------------- layout helpers

def hc_to_complex(hc):
    """FFTW halfcomplex (2L reals) -> complex bins 0..L"""
    n = len(hc)
    L = n // 2
    X = np.empty(L + 1, np.complex128)
    X[0] = hc[0]
    X[L] = hc[L]
    X[1:L] = hc[1:L] + 1j * hc[n - 1:L:-1]
    return X


def complex_to_hc(X, dtype=np.float64):
    L = len(X) - 1
    hc = np.empty(2 * L, dtype)
    hc[0:L + 1] = X.real
    hc[2 * L - 1:L:-1] = X[1:L].imag
    return hc


def reordered_to_complex(q):
    """reference ring/coefficient layout -> complex bins 0..L"""
    L = len(q) // 2
    g = q.reshape(L // 4, 2, 4)
    X = np.empty(L + 1, np.complex128)
    X[:L] = (g[:, 0, :] + 1j * g[:, 1, :]).ravel()
    X[L] = q[4]
    X[0] = q[0]
    return X


def complex_to_reordered(X, dtype=np.float64):
    L = len(X) - 1
    q = np.empty((L // 4, 2, 4), dtype)
    q[:, 0, :] = X[:L].real.reshape(-1, 4)
    q[:, 1, :] = X[:L].imag.reshape(-1, 4)
    q = q.ravel()
    q[4] = X[L].real
    return q


# ---------------------------------------------------------------- reference (_ref)

_ref = None


def ref():
    """The reference's own inner loops (oracle/_ref/libbfref.so) or None if not built."""
    global _ref
    if _ref is None:
        if not os.path.exists(REF_SO):
            try:
                build()
            except Exception:
                pass
        if not os.path.exists(REF_SO):
            return None
        L = C.CDLL(REF_SO)
        vp, ci, cd = C.c_void_p, C.c_int, C.c_double
        L.ref_set_length.argtypes = [ci, cd]
        L.ref_mixnscale.argtypes = [ci, C.POINTER(vp), vp, C.POINTER(cd), ci, ci]
        for f in ("ref_convolve", "ref_convolve_add", "ref_convolve_add_simd"):
            getattr(L, f).argtypes = [ci, vp, vp, vp]
        L.ref_convolve_inplace.argtypes = [ci, vp, vp]
        L.ref_dirac_convolve.argtypes = [ci, vp, vp]
        L.ref_dirac_convolve_inplace.argtypes = [ci, vp]
        L.ref_raw2real.argtypes = [ci, vp, vp, ci, ci, ci, ci, ci]
        L.ref_real2raw.argtypes = [ci, vp, vp, ci, ci, ci, ci, ci, ci, C.POINTER(Overflow), ci]
        L.ref_dither_init.argtypes = [ci, ci, ci, ci, ci]
        L.ref_dither_table.argtypes = [C.POINTER(C.POINTER(C.c_int8))]
        L.ref_dither_randtab_ptr.argtypes = [ci]
        _ref = L
    return _ref


_ref_delay = None


def ref_delay():
    """The reference's delay.c / firwindow.c (oracle/_ref/libbfref_delay.so, a library of its own:
    only it carries the link-time placeholder for delay.c's td_* calls) or None if not built."""
    global _ref_delay
    if _ref_delay is None:
        if not os.path.exists(REF_DELAY_SO):
            try:
                build()
            except Exception:
                pass
        if not os.path.exists(REF_DELAY_SO):
            return None
        L = C.CDLL(REF_DELAY_SO)
        L.ref_delay_allocate.restype = C.c_void_p
        L.ref_delay_allocate.argtypes = [C.c_int] * 4
        L.ref_delay_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ref_firwindow_kaiser.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int]
        _ref_delay = L
    return _ref_delay
