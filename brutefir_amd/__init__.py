"""brutefir_amd -- MI355X-native replacement for BruteFIR's filter path.

The product is `libbfhip.so` (HIP kernels + a plain C ABI, `include/bfhip.h`); the C host
(`bfrun`/`bfconf`, bfio and bflogic modules) binds to it directly.  This
package is the thin Python binding used by tests/ and bench.py: it mirrors the C ABI one to
one and adds nothing.  There is no CPU path: if the library or a HIP device is missing every
call raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbfhip.so")

IN, OUT = 0, 1
RT_SPIN, RT_NO_GRAPH, RT_COPY_ENGINE, RT_OVERLAP = 1, 2, 4, 8      # bfhip_engine_rt_begin flags
COEFF_WATCH, COEFF_LAZY = 1, 2                                       # bfhip_engine_add_coeff_processed_blocks flags
ST_NONFINITE, ST_SAFETY = 1, 2


class BfhipError(RuntimeError):
    pass


class Overflow(C.Structure):
    """struct bfoverflow (bfmod.h:99-104)"""
    _fields_ = [("n_overflows", C.c_uint), ("intlargest", C.c_int32),
                ("largest", C.c_double), ("max", C.c_double)]

    def astuple(self):
        return (self.n_overflows, self.intlargest, self.largest, self.max)


class Format(C.Structure):
    """struct sample_format + struct buffer_format (dai.h:21-34)"""
    _fields_ = [("isfloat", C.c_int), ("swap", C.c_int), ("bytes", C.c_int),
                ("sbytes", C.c_int), ("scale", C.c_double),
                ("sample_spacing", C.c_int), ("byte_offset", C.c_int)]


# bfconf.c:358-480 (the _NE macro formats are resolved by the host before they get here)
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
    """buffer_format of an interleaved device with n_channels open (dai.c:537-576)"""
    nbytes = SAMPLE_FORMATS[name][0]
    return [make_format(name, n_channels, c * nbytes) for c in range(n_channels)]


_lib = None


def lib():
    """Load libbfhip.so.  Raises if it has not been built -- never falls back to anything."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BfhipError("%s is missing: build it with `python -m brutefir_amd.build` "
                         "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    ip, dp = C.POINTER(ci), C.POINTER(cd)
    L.bfhip_last_error.restype = C.c_char_p
    L.bfhip_version.restype = C.c_char_p
    L.bfhip_engine_create.restype = vp
    L.bfhip_engine_create.argtypes = [ci] * 6
    L.bfhip_engine_destroy.argtypes = [vp]
    L.bfhip_engine_set_format.argtypes = [vp, ci, ci, C.POINTER(Format)]
    L.bfhip_engine_set_safety_limit.argtypes = [vp, cd]
    L.bfhip_engine_enable_subdelay.argtypes = [vp, ci, cd]
    L.bfhip_engine_set_subdelay.argtypes = [vp, ci, ci, ci]
    L.bfhip_engine_map_channels.argtypes = [vp, ci, ci, ip]
    for f in ("bfhip_engine_set_delay", "bfhip_engine_set_maxdelay", "bfhip_engine_set_mute"):
        getattr(L, f).argtypes = [vp, ci, ci, ci]
    L.bfhip_engine_enable_dither.argtypes = [vp, ip, ci, ci, ci]
    L.bfhip_engine_reserve_coeffs.argtypes = [vp, cd]
    L.bfhip_engine_add_coeff.argtypes = [vp, vp, ci, cd, ci]
    L.bfhip_engine_add_coeff_dev.argtypes = [vp, vp, ci, cd, ci]
    L.bfhip_engine_update_coeff_block.argtypes = [vp, ci, ci, vp]
    L.bfhip_engine_add_coeff_processed.argtypes = [vp, vp, ci]
    L.bfhip_engine_read_coeff_processed.argtypes = [vp, ci, vp]
    L.bfhip_engine_add_coeff_processed_blocks.argtypes = [vp, C.POINTER(vp), ci, ci]
    L.bfhip_engine_refresh_coeff_processed.argtypes = [vp, ci, ci, vp]
    L.bfhip_engine_poll_coeff_changes.argtypes = [vp]
    L.bfhip_coeff_mark_dirty.argtypes = [vp]
    L.bfhip_coeff_mark_dirty.restype = None
    L.bfhip_coeff_dirty_sequence.restype = C.c_ulonglong
    L.bfhip_engine_add_filter.argtypes = [vp, ci, ip, dp, ci, ip, dp, ci, ip, dp, ci, ci, ci]
    L.bfhip_engine_set_filter_active.argtypes = [vp, ci, ci]
    L.bfhip_engine_stage_times.argtypes = [vp, dp]
    L.bfhip_engine_coeff_is_resident.argtypes = [vp, ci]
    L.bfhip_engine_set_output_active.argtypes = [vp, ci, ci]
    L.bfhip_engine_output_is_active.argtypes = [vp, ci]
    L.bfhip_engine_set_filter_name.argtypes = [vp, ci, ci]
    L.bfhip_engine_finalize.argtypes = [vp]
    L.bfhip_engine_set_coeff.argtypes = [vp, ci, ci]
    L.bfhip_engine_set_delayblocks.argtypes = [vp, ci, ci]
    L.bfhip_engine_set_scale.argtypes = [vp, ci, ci, ci, cd]
    L.bfhip_engine_set_fscale.argtypes = [vp, ci, ci, cd]
    L.bfhip_engine_block.argtypes = [vp, vp, vp, C.POINTER(Overflow)]
    L.bfhip_engine_block_dev.argtypes = [vp, vp, vp]
    L.bfhip_engine_block_dev_ev.argtypes = [vp, vp, vp, vp, vp]
    L.bfhip_engine_sync.argtypes = [vp]
    L.bfhip_engine_inputs_dev.argtypes = [vp, vp]
    L.bfhip_engine_mac_dev.argtypes = [vp, vp]
    L.bfhip_engine_outputs_dev.argtypes = [vp, vp, ci, ci, vp]
    L.bfhip_engine_outputs_inputs_dev.argtypes = [vp, vp, ci, ci, vp, vp]
    L.bfhip_engine_advance.argtypes = [vp]
    L.bfhip_engine_set_stream.argtypes = [vp, vp]
    L.bfhip_engine_get_overflow.argtypes = [vp, ci, C.POINTER(Overflow)]
    L.bfhip_engine_reset_overflow.argtypes = [vp]
    L.bfhip_engine_blockcounter.restype = C.c_uint
    L.bfhip_engine_blockcounter.argtypes = [vp]
    L.bfhip_engine_ring_depth.argtypes = [vp]
    L.bfhip_engine_block_mode.argtypes = [vp]
    L.bfhip_selftest_fail_alloc.argtypes = [ci]
    L.bfhip_engine_output_lag.argtypes = [vp]
    L.bfhip_engine_flush.argtypes = [vp]
    L.bfhip_engine_uses_wave_fft.argtypes = [vp]
    L.bfhip_engine_uses_stream_layout.argtypes = [vp]
    L.bfhip_engine_uses_diag_mac.argtypes = [vp]
    L.bfhip_engine_enable_pairs.argtypes = [vp, ci]
    L.bfhip_engine_block_pair_dev.argtypes = [vp, vp, vp, vp, vp]
    L.bfhip_engine_pair_launches.argtypes = [vp]
    L.bfhip_engine_pair_launches.restype = C.c_ulonglong
    L.bfhip_engine_enable_timing.argtypes = [vp, ci]
    L.bfhip_engine_get_timing.argtypes = [vp, dp]
    L.bfhip_engine_algorithmic_bytes.argtypes = [vp, dp]
    L.bfhip_engine_read_output_spectrum.argtypes = [vp, ci, vp]
    L.bfhip_engine_read_ring_slot.argtypes = [vp, ci, ci, vp]
    ull = C.POINTER(C.c_ulonglong)
    L.bfhip_engine_prewarm.argtypes = [vp]
    L.bfhip_engine_set_powersave.argtypes = [vp, cd]
    L.bfhip_engine_rt_begin.argtypes = [vp, ci]
    L.bfhip_engine_rt_end.argtypes = [vp]
    L.bfhip_engine_rt_buffer.restype = vp
    L.bfhip_engine_rt_buffer.argtypes = [vp, ci, ci]
    L.bfhip_engine_rt_submit.argtypes = [vp, vp]
    L.bfhip_engine_rt_wait.argtypes = [vp, vp, C.POINTER(Overflow)]
    L.bfhip_engine_rt_block.argtypes = [vp, vp, vp, C.POINTER(Overflow)]
    L.bfhip_engine_rt_stats.argtypes = [vp, ull, ull, ull]
    # non-uniform partitioned convolver (include/bfhip_nupc.h)
    L.bfhip_nupc_last_error.restype = C.c_char_p
    L.bfhip_nupc_create.restype = vp
    L.bfhip_nupc_create.argtypes = [ci, ci, ci, ci, ci, ip, ip]
    L.bfhip_nupc_destroy.argtypes = [vp]
    L.bfhip_nupc_taps.restype = C.c_long
    L.bfhip_nupc_taps.argtypes = [vp]
    L.bfhip_nupc_latency.argtypes = [vp]
    L.bfhip_nupc_set_format.argtypes = [vp, ci, ci, C.POINTER(Format)]
    L.bfhip_nupc_set_safety_limit.argtypes = [vp, cd]
    L.bfhip_nupc_add_filter.argtypes = [vp, ci, ci, vp, C.c_long, cd, cd]
    L.bfhip_nupc_finalize.argtypes = [vp]
    L.bfhip_nupc_block.argtypes = [vp, vp, vp, C.POINTER(Overflow)]
    L.bfhip_nupc_block_dev.argtypes = [vp, vp, vp]
    L.bfhip_nupc_sync.argtypes = [vp]
    L.bfhip_nupc_get_overflow.argtypes = [vp, ci, C.POINTER(Overflow)]
    L.bfhip_engine_set_overlap.argtypes = [vp, ci]
    _lib = L
    return L


def _check(r):
    if r < 0:
        raise BfhipError("bfhip error %d: %s" % (r, lib().bfhip_last_error().decode()))
    return r


def _iarr(v):
    return (C.c_int * max(len(v), 1))(*v)


def _darr(v):
    return (C.c_double * max(len(v), 1))(*v)


def _ptr(x):
    """host numpy array, torch tensor (host or device) or raw integer address -> void*"""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return x.ctypes.data_as(C.c_void_p)


class Engine:
    """One filter process worth of device state (bfhip_engine)."""

    def __init__(self, length, n_blocks, realsize, n_in, n_out, device=0):
        self.L, self.N, self.rs, self.n_in, self.n_out = length, n_blocks, realsize, n_in, n_out
        self.dt = np.float32 if realsize == 4 else np.float64
        self.cdt = np.complex64 if realsize == 4 else np.complex128
        self.h = lib().bfhip_engine_create(device, length, n_blocks, realsize, n_in, n_out)
        if not self.h:
            raise BfhipError(lib().bfhip_last_error().decode())
        self.out_bytes = n_out * length * realsize
        self.in_bytes = n_in * length * realsize

    def close(self):
        if getattr(self, "h", None):
            lib().bfhip_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    # ---- construction
    def set_format(self, io, ch, fmt):
        _check(lib().bfhip_engine_set_format(self.h, io, ch, C.byref(fmt)))

    def set_interleaved(self, io, name):
        n = self.n_in if io == IN else self.n_out
        for c, f in enumerate(interleaved_formats(name, n)):
            self.set_format(io, c, f)
        nbytes = n * self.L * SAMPLE_FORMATS[name][0]
        if io == IN:
            self.in_bytes = nbytes
        else:
            self.out_bytes = nbytes

    def set_powersave(self, analog_powersave):
        _check(lib().bfhip_engine_set_powersave(self.h, analog_powersave))

    def set_safety_limit(self, v):
        _check(lib().bfhip_engine_set_safety_limit(self.h, v))

    def map_channels(self, io, virt2phys):
        _check(lib().bfhip_engine_map_channels(self.h, io, max(virt2phys) + 1, _iarr(list(virt2phys))))

    def set_interleaved_phys(self, io, name, n_phys):
        for c, f in enumerate(interleaved_formats(name, n_phys)):
            self.set_format(io, c, f)
        nbytes = n_phys * self.L * SAMPLE_FORMATS[name][0]
        if io == IN:
            self.in_bytes = nbytes
        else:
            self.out_bytes = nbytes

    def set_delay(self, io, ch, delay):
        _check(lib().bfhip_engine_set_delay(self.h, io, ch, delay))

    def set_maxdelay(self, io, ch, maxdelay):
        _check(lib().bfhip_engine_set_maxdelay(self.h, io, ch, maxdelay))

    def set_mute(self, io, ch, muted):
        _check(lib().bfhip_engine_set_mute(self.h, io, ch, int(muted)))

    def enable_subdelay(self, sdf_length, beta=9.0):
        _check(lib().bfhip_engine_enable_subdelay(self.h, sdf_length, beta))

    def set_subdelay(self, io, ch, subdelay):
        _check(lib().bfhip_engine_set_subdelay(self.h, io, ch, subdelay))

    def enable_dither(self, channels, sample_rate, max_size=0):
        _check(lib().bfhip_engine_enable_dither(self.h, _iarr(list(channels)), len(channels),
                                                sample_rate, max_size))

    def reserve_coeffs(self, total_bytes):
        _check(lib().bfhip_engine_reserve_coeffs(self.h, float(total_bytes)))

    def add_coeff(self, taps, scale=1.0, n_blocks=0):
        taps = np.ascontiguousarray(taps, self.dt)
        return _check(lib().bfhip_engine_add_coeff(self.h, _ptr(taps), len(taps), scale, n_blocks))

    def add_coeff_dev(self, taps_dev, n_taps, scale=1.0, n_blocks=0):
        return _check(lib().bfhip_engine_add_coeff_dev(self.h, _ptr(taps_dev), n_taps, scale,
                                                       n_blocks))

    def add_coeff_processed(self, cbufs):
        """cbufs: [n_blocks, 2L] reals in the reference's internal layout"""
        cbufs = np.ascontiguousarray(cbufs, self.dt)
        return _check(lib().bfhip_engine_add_coeff_processed(self.h, _ptr(cbufs), cbufs.shape[0]))

    def add_coeff_processed_blocks(self, addresses, watch=False, lazy=False):
        """addresses: host address of each block's cbuf (separate allocations, the reference's
        bfconf->coeffs_data[c][i]); watch: re-upload blocks another process marks dirty"""
        arr = (C.c_void_p * len(addresses))(*addresses)
        return _check(lib().bfhip_engine_add_coeff_processed_blocks(self.h, arr, len(addresses),
                                                                    (COEFF_WATCH if watch else 0) | (COEFF_LAZY if lazy else 0)))

    def refresh_coeff_processed(self, coeff, block, cbuf=None):
        _check(lib().bfhip_engine_refresh_coeff_processed(self.h, coeff, block, _ptr(cbuf)))

    def poll_coeff_changes(self):
        return _check(lib().bfhip_engine_poll_coeff_changes(self.h))

    def read_coeff_processed(self, coeff, n_blocks):
        out = np.empty((n_blocks, 2 * self.L), self.dt)
        got = _check(lib().bfhip_engine_read_coeff_processed(self.h, coeff, _ptr(out)))
        assert got == n_blocks
        return out

    def update_coeff_block(self, coeff, block, taps):
        taps = np.ascontiguousarray(taps, self.dt)
        assert len(taps) == self.L
        _check(lib().bfhip_engine_update_coeff_block(self.h, coeff, block, _ptr(taps)))

    def add_filter(self, in_ch=(), in_scale=None, in_f=(), in_fscale=None, out_ch=(),
                   out_scale=None, coeff=-1, delayblocks=0, crossfade=False):
        in_scale = [1.0] * len(in_ch) if in_scale is None else list(in_scale)
        in_fscale = [1.0] * len(in_f) if in_fscale is None else list(in_fscale)
        out_scale = [1.0] * len(out_ch) if out_scale is None else list(out_scale)
        return _check(lib().bfhip_engine_add_filter(
            self.h, len(in_ch), _iarr(list(in_ch)), _darr(in_scale),
            len(in_f), _iarr(list(in_f)), _darr(in_fscale),
            len(out_ch), _iarr(list(out_ch)), _darr(out_scale),
            coeff, delayblocks, int(crossfade)))

    # ---- a shard of the configuration (one engine per filter process of the host)
    def set_filter_active(self, f, active):
        _check(lib().bfhip_engine_set_filter_active(self.h, f, int(active)))

    def set_output_active(self, ch, active):
        _check(lib().bfhip_engine_set_output_active(self.h, ch, int(active)))

    def output_is_active(self, ch):
        return bool(lib().bfhip_engine_output_is_active(self.h, ch))

    def coeff_is_resident(self, c):
        return bool(lib().bfhip_engine_coeff_is_resident(self.h, c))

    def set_filter_name(self, f, name):
        _check(lib().bfhip_engine_set_filter_name(self.h, f, name))

    def finalize(self):
        _check(lib().bfhip_engine_finalize(self.h))

    # ---- run-time control
    def set_coeff(self, f, c):
        _check(lib().bfhip_engine_set_coeff(self.h, f, c))

    def set_delayblocks(self, f, d):
        _check(lib().bfhip_engine_set_delayblocks(self.h, f, d))

    def set_scale(self, f, io, idx, v):
        _check(lib().bfhip_engine_set_scale(self.h, f, io, idx, v))

    def set_fscale(self, f, idx, v):
        _check(lib().bfhip_engine_set_fscale(self.h, f, idx, v))

    # ---- per block
    def block(self, rawin, overflow=None, out=None):
        """host buffers in / out; returns (status bits, raw output bytes).  out: write into this
        buffer (the one the engines of several filter processes share)"""
        rawin = np.ascontiguousarray(rawin).view(np.uint8).ravel()
        assert rawin.size >= self.in_bytes, (rawin.size, self.in_bytes)
        if out is None:
            out = np.zeros(self.out_bytes, np.uint8)
        assert out.dtype == np.uint8 and out.size >= self.out_bytes and out.flags.c_contiguous
        st = _check(lib().bfhip_engine_block(self.h, _ptr(rawin), _ptr(out), overflow))
        return st, out

    def block_dev(self, rawin_dev, rawout_dev):
        _check(lib().bfhip_engine_block_dev(self.h, _ptr(rawin_dev), _ptr(rawout_dev)))

    def enable_pairs(self, on=True):
        _check(lib().bfhip_engine_enable_pairs(self.h, int(on)))

    def block_pair_dev(self, rawin0_dev, rawout0_dev, rawin1_dev, rawout1_dev):
        """two consecutive blocks, one pass over the coefficients (after enable_pairs before finalize)"""
        _check(lib().bfhip_engine_block_pair_dev(self.h, _ptr(rawin0_dev), _ptr(rawout0_dev), _ptr(rawin1_dev), _ptr(rawout1_dev)))

    @property
    def pair_launches(self):
        return int(lib().bfhip_engine_pair_launches(self.h))

    def block_dev_ev(self, rawin_dev, rawout_dev, in_ready=None, out_done=None):
        """in_ready / out_done: hipEvent_t handles (ints), e.g. torch.cuda.Event().cuda_event"""
        _check(lib().bfhip_engine_block_dev_ev(self.h, _ptr(rawin_dev), _ptr(rawout_dev),
                                               C.c_void_p(in_ready) if in_ready else None,
                                               C.c_void_p(out_done) if out_done else None))

    def set_overlap(self, mode):
        _check(lib().bfhip_engine_set_overlap(self.h, mode))

    def prewarm(self):
        _check(lib().bfhip_engine_prewarm(self.h))

    # real-time mode (callback I/O): pinned double buffer + graph replay
    def rt_begin(self, flags=0):
        _check(lib().bfhip_engine_rt_begin(self.h, flags))

    def rt_end(self):
        _check(lib().bfhip_engine_rt_end(self.h))

    def rt_submit(self, rawin):
        rawin = np.ascontiguousarray(rawin).view(np.uint8).ravel()
        assert rawin.size >= self.in_bytes, (rawin.size, self.in_bytes)
        _check(lib().bfhip_engine_rt_submit(self.h, _ptr(rawin)))

    def rt_wait(self, overflow=None, out=None):
        if out is None:
            out = np.zeros(self.out_bytes, np.uint8)
        assert out.dtype == np.uint8 and out.size >= self.out_bytes and out.flags.c_contiguous
        st = _check(lib().bfhip_engine_rt_wait(self.h, _ptr(out), overflow))
        return st, out

    def rt_block(self, rawin, overflow=None, out=None):
        self.rt_submit(rawin)
        return self.rt_wait(overflow, out)

    def rt_stats(self):
        a, b, c = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
        _check(lib().bfhip_engine_rt_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"graph": a.value, "direct": b.value, "captures": c.value}

    def sync(self):
        return _check(lib().bfhip_engine_sync(self.h))

    def flush(self):
        """launch the output passes still owed (deferred / ping-pong schedule); does not wait"""
        _check(lib().bfhip_engine_flush(self.h))

    @property
    def output_lag(self):
        """block calls that pass before a block's output pass is launched (0, 1 or 2)"""
        return lib().bfhip_engine_output_lag(self.h)

    def inputs_dev(self, rawin_dev):
        _check(lib().bfhip_engine_inputs_dev(self.h, _ptr(rawin_dev)))

    def mac_dev(self, z_dev):
        _check(lib().bfhip_engine_mac_dev(self.h, _ptr(z_dev)))

    def outputs_dev(self, z_dev, first, count, rawout_dev):
        _check(lib().bfhip_engine_outputs_dev(self.h, _ptr(z_dev), first, count, _ptr(rawout_dev)))

    def outputs_inputs_dev(self, z_dev, first, count, rawout_dev, rawin_dev):
        _check(lib().bfhip_engine_outputs_inputs_dev(self.h, _ptr(z_dev), first, count,
                                                     _ptr(rawout_dev), _ptr(rawin_dev)))

    def advance(self):
        _check(lib().bfhip_engine_advance(self.h))

    def set_stream(self, hip_stream):
        _check(lib().bfhip_engine_set_stream(self.h, C.c_void_p(hip_stream)))

    def overflow(self, ch):
        of = Overflow()
        _check(lib().bfhip_engine_get_overflow(self.h, ch, C.byref(of)))
        return of

    def reset_overflow(self):
        _check(lib().bfhip_engine_reset_overflow(self.h))

    @property
    def blockcounter(self):
        return lib().bfhip_engine_blockcounter(self.h)

    @property
    def block_mode(self):
        """0 sequential, 1 pipelined (three streams), 2 deferred output (fused K3|K1 launch),
        3 ping-pong (fused K3|K1 launch on a side stream beside the previous MAC)"""
        return lib().bfhip_engine_block_mode(self.h)

    @property
    def uses_wave_fft(self):
        return bool(lib().bfhip_engine_uses_wave_fft(self.h))

    @property
    def uses_stream_layout(self):
        return bool(lib().bfhip_engine_uses_stream_layout(self.h))

    @property
    def uses_diag_mac(self):
        return bool(lib().bfhip_engine_uses_diag_mac(self.h))

    @property
    def ring_depth(self):
        return lib().bfhip_engine_ring_depth(self.h)

    # ---- measurement / debug
    def enable_timing(self, on=True):
        _check(lib().bfhip_engine_enable_timing(self.h, int(on)))

    def timing(self):
        ms = (C.c_double * 4)()
        _check(lib().bfhip_engine_get_timing(self.h, ms))
        return {"fft_in_ms": ms[0], "mac_ms": ms[1], "ifft_out_ms": ms[2], "launches": int(ms[3])}

    STAGES = ("raw2real", "time2freq", "mixscale1", "convolve", "mixscale2", "freq2time", "real2raw", "total")

    def stage_times(self):
        """the reference's `benchmark: true` columns (bfrun.c:2035-2078): (blocks averaged, {column: ms})"""
        ms = (C.c_double * 8)()
        n = _check(lib().bfhip_engine_stage_times(self.h, ms))
        return n, dict(zip(self.STAGES, ms))

    def algorithmic_bytes(self):
        b = (C.c_double * 2)()
        _check(lib().bfhip_engine_algorithmic_bytes(self.h, b))
        return {"block": b[0], "mac": b[1]}

    def output_spectrum(self, ch):
        """packed spectrum of output ch: element 0 = (DC, Nyquist), element k = bin k"""
        z = np.empty(self.L, self.cdt)
        _check(lib().bfhip_engine_read_output_spectrum(self.h, ch, _ptr(z)))
        return z

    def ring_slot(self, ch, slot):
        z = np.empty(self.L, self.cdt)
        _check(lib().bfhip_engine_read_ring_slot(self.h, ch, slot, _ptr(z)))
        return z


class Nupc:
    """Non-uniform partitioned convolver (bfhip_nupc): low-latency first block."""

    def __init__(self, seg_length, seg_blocks, realsize, n_in, n_out, device=0):
        self.rs, self.n_in, self.n_out = realsize, n_in, n_out
        self.dt = np.float32 if realsize == 4 else np.float64
        self.h = lib().bfhip_nupc_create(device, realsize, n_in, n_out, len(seg_length),
                                         _iarr(list(seg_length)), _iarr(list(seg_blocks)))
        if not self.h:
            raise BfhipError(lib().bfhip_nupc_last_error().decode())
        self.L0 = lib().bfhip_nupc_latency(self.h)
        self.taps = lib().bfhip_nupc_taps(self.h)
        self.out_bytes = n_out * self.L0 * realsize

    def _chk(self, r):
        if r < 0:
            raise BfhipError("bfhip nupc error %d: %s" % (r, lib().bfhip_nupc_last_error().decode()))
        return r

    def close(self):
        if getattr(self, "h", None):
            lib().bfhip_nupc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_interleaved(self, io, name):
        n = self.n_in if io == IN else self.n_out
        for c, f in enumerate(interleaved_formats(name, n)):
            self._chk(lib().bfhip_nupc_set_format(self.h, io, c, C.byref(f)))
        if io == OUT:
            self.out_bytes = n * self.L0 * SAMPLE_FORMATS[name][0]

    def add_filter(self, in_ch, out_ch, taps, in_scale=1.0, out_scale=1.0):
        taps = np.ascontiguousarray(taps, self.dt)
        self._chk(lib().bfhip_nupc_add_filter(self.h, in_ch, out_ch, _ptr(taps), len(taps), in_scale, out_scale))

    def finalize(self):
        self._chk(lib().bfhip_nupc_finalize(self.h))

    def block(self, rawin):
        rawin = np.ascontiguousarray(rawin).view(np.uint8).ravel()
        out = np.zeros(self.out_bytes, np.uint8)
        st = self._chk(lib().bfhip_nupc_block(self.h, _ptr(rawin), _ptr(out), None))
        return st, out

    def block_dev(self, rawin_dev, rawout_dev):
        self._chk(lib().bfhip_nupc_block_dev(self.h, _ptr(rawin_dev), _ptr(rawout_dev)))

    def sync(self):
        return self._chk(lib().bfhip_nupc_sync(self.h))

    def overflow(self, ch):
        of = Overflow()
        self._chk(lib().bfhip_nupc_get_overflow(self.h, ch, C.byref(of)))
        return of


def device_count():
    return lib().bfhip_device_count()
