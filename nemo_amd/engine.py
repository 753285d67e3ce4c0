"""Device-side plumbing: torch owns HBM buffers and streams, every computation is a libnemofilt call.

:class:`TileEngine` holds what is fixed for one tile shape (FFT plan, apodisation / pixel-window tables)
and exposes one thin method per C-ABI entry point.  Nothing here computes on the CPU except the small
1-D tables (apodisation profile, sinc window, Gaussian taps, noise-grid edges) that parametrise kernels.
"""
import ctypes
import os
import threading

import numpy as np

from . import _lib
from ._lib import NF_F32, NF_F64, IN_COMPLEX, IN_REAL64, IN_REAL32, NemoFiltError, check

_torch = None


def torch():
    global _torch
    if _torch is None:
        import torch as _t
        _torch = _t
    return _torch


_device = None       # (torch.device, index): the GPU this PROCESS drives (one process per GPU), fixed by the first caller


def bind_device(index):
    """Make HIP device ``index`` the device of this process (pipelines.initDistributed: LOCAL_RANK).  Every thread that
    later goes through :func:`require_gpu` is switched to it."""
    global _device
    t = torch()
    if not t.cuda.is_available():
        raise NemoFiltError("no HIP device visible: nemo_amd runs on MI355X (gfx950) only and has no CPU fallback")
    _lib.load()
    index = int(index)
    t.cuda.set_device(index)
    _device = (t.device('cuda', index), index)
    return _device[0]


def require_gpu():
    """The HIP device of this process; raises when there is none (no CPU fallback).  The device is chosen once - the
    current device of the first caller, or :func:`bind_device` - and never follows per-thread state afterwards: HIP's
    current device belongs to the host thread and starts at 0 in every new thread, so a worker thread (the detection pool
    of pipelines.py) that arrives here on another device is switched to the process's device instead of dragging the
    library over to device 0.  Called from every tensor hand-over: the check is one raw call (~0.1 us)."""
    global _device
    t = torch()
    if _device is not None:
        if t._C._cuda_getDevice() != _device[1]:
            t.cuda.set_device(_device[1])
        return _device[0]
    if not t.cuda.is_available():
        raise NemoFiltError("no HIP device visible: nemo_amd runs on MI355X (gfx950) only and has no CPU fallback")
    _lib.load()
    idx = t.cuda.current_device()
    _device = (t.device('cuda', idx), idx)
    return _device[0]


def gpu_ready():
    """True once this process is already working on a HIP device through the library (host-side helpers use it to decide
    whether a device kernel is worth calling; it never initialises anything)."""
    from . import _lib as L
    if L._lib is None:
        return False
    t = torch()
    return t.cuda.is_available() and t.cuda.is_initialized()


def xmatch_self(RADeg, decDeg, SNR, radiusDeg):
    """nf_xmatch_self: (count, best, drop) of the self cross match as numpy arrays, or None when a chord lies within 1e-9 of
    the limit (the caller then decides on the host with the exact great-circle formula)."""
    t = torch()
    dev = require_gpu()
    ra, dec = np.radians(np.asarray(RADeg, dtype=np.float64)), np.radians(np.asarray(decDeg, dtype=np.float64))
    u = np.stack([np.cos(dec) * np.cos(ra), np.cos(dec) * np.sin(ra), np.sin(dec)], axis=1)     # catalogs._unitVectors
    n = len(u)
    u_d, s_d = to_dev(u), to_dev(np.asarray(SNR, dtype=np.float64))
    ints = t.empty(2 * n + 1, dtype=t.int32, device=dev)
    drop = t.empty(n, dtype=t.uint8, device=dev)
    lib = _lib.load()
    wbytes = lib.nf_xmatch_workspace_bytes(n)
    work = t.empty(wbytes // 8 + 1, dtype=t.float64, device=dev)
    check(lib.nf_xmatch_self(ptr(u_d), ptr(s_d), n, float(radiusDeg), ptr(ints[:n]), ptr(ints[n:2 * n]), ptr(drop),
                             ptr(ints[2 * n:]), ptr(work), work.numel() * 8, stream_ptr()), "nf_xmatch_self")
    # one transfer: the ints, then the drop bytes widened
    h = t.cat([ints, drop.to(t.int32)]).cpu().numpy()
    if h[2 * n] != 0:
        return None
    return h[:n].astype(np.int64), h[n:2 * n].astype(np.int64), h[2 * n + 1:] != 0


def stream_ptr():
    """The current HIP stream of the calling thread as a void* (raw handle: no Stream object is built)."""
    t = torch()
    dev = require_gpu()
    return ctypes.c_void_p(t._C._cuda_getCurrentRawStream(dev.index))


def ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


_PINNED_COPY_MAX = 4 << 20
_radiiDev = {}                # (length, content hash) -> device copy of a radial table's radii (TileEngine._tableDev)
CHECK_NOISE_TABLES = os.environ.get('NEMO_AMD_CHECK_NOISE_TABLES', '0') == '1'      # debugging: see TileEngine.matched_filter_scales
SCALE_BATCH = None            # scales per group of nf_matched_filter_scales (None: by tile size, Engine.scale_batch)
_AXIS_WINDOW = os.environ.get('NEMO_AMD_AXIS_WINDOW', '1') != '0'    # 0: apply_window through the 2-D transforms always


def to_dev(x, dtype=None):
    """numpy array or torch tensor -> contiguous tensor on the current HIP device."""
    t = torch()
    dev = require_gpu()
    if isinstance(x, t.Tensor):
        out = x.to(dev)
    else:
        a = np.ascontiguousarray(x)
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)
        if 0 < a.nbytes <= _PINNED_COPY_MAX:
            # small tables (profiles, grid edges, evaluation points): through pinned memory and asynchronously, because a
            # pageable copy makes the host wait for everything already queued on the stream
            out = t.from_numpy(a).pin_memory().to(dev, non_blocking=True)
        else:
            out = t.from_numpy(a).to(dev)
    if dtype is not None and out.dtype != dtype:
        out = out.to(dtype)
    return out.contiguous()


def prec_of(tensor):
    t = torch()
    if tensor.dtype in (t.float64, t.complex128):
        return NF_F64
    if tensor.dtype in (t.float32, t.complex64):
        return NF_F32
    raise NemoFiltError("unsupported dtype %s" % tensor.dtype)


def gaussian_kernel1d(sigma, truncate=4.0):
    """Taps exactly as scipy.ndimage.gaussian_filter1d builds them (order 0)."""
    sd = float(sigma)
    radius = int(truncate * sd + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sd * sd) * x ** 2)
    return phi / phi.sum(), radius


_EDGE_SEL = os.environ.get('NEMO_AMD_SPLINE_EDGESEL', '0') != '0'    # measured: no gain on the C2 bench


class _SplinePoints(object):
    """Evaluation points on the device: (y, x) pairs and the indices of the points whose window meets the map border."""
    __slots__ = ('dev', 'n', 'pointsClass', 'halfWin', 'edge', 'nEdge')

    def __init__(self, dev, n, pointsClass, halfWin, edge=None, nEdge=0):
        self.dev, self.n, self.pointsClass, self.halfWin, self.edge, self.nEdge = dev, n, pointsClass, halfWin, edge, nEdge


class DeviceScalar(object):
    """A float64 / float32 scalar that lives on the device until somebody needs it on the host (``value``): keeps the
    host from waiting for the GPU in the middle of a filter build.  ``wide`` = numpy float64 semantics (else float32)."""

    def __init__(self, tensor, wide):
        self.tensor = tensor          # float64 tensor with one element (holds the float32-rounded value if not wide)
        self.wide = wide
        self._host = None

    @property
    def value(self):
        if self._host is None:
            v = float(self.tensor.reshape(-1)[0].item())
            self._host = np.float64(v) if self.wide else np.float32(v)
        return self._host


class TileEngine(object):
    APOD_PIX = 20      # nemo/filters.py:195
    _cache = {}

    @classmethod
    def get(cls, shape):
        dev = require_gpu()
        key = (int(shape[0]), int(shape[1]), dev.index)
        if key not in cls._cache:
            cls._cache[key] = TileEngine(shape)
        return cls._cache[key]

    def __init__(self, shape):
        self.t = torch()
        self.dev = require_gpu()
        self.lib = _lib.load()
        self.H, self.W = int(shape[0]), int(shape[1])
        self.npix = self.H * self.W
        plan = ctypes.c_void_p()
        check(self.lib.nf_plan_create(self.H, self.W, ctypes.byref(plan)), "nf_plan_create")
        self.plan = plan
        # enmap.apod profile (pixell): 0.5*(1-cos(linspace(0, pi, w))) on both ends of each axis
        tabs = []
        for n in (self.H, self.W):
            w = int(min(self.APOD_PIX, n))
            a = np.ones(n, dtype=np.float64)
            if w > 0:
                prof = 0.5 * (1 - np.cos(np.linspace(0, np.pi, w)))
                a[:w] *= prof
                a[-w:] *= prof[::-1]
            tabs.append(to_dev(a))
        self.apodY, self.apodX = tabs
        self._win = {}
        self._axisWin = {}
        self._keptLabels = {}       # (thread, slot) -> (labels, dirty flags) kept between label() calls
        self._chunkFlags = {}       # labels.data_ptr() -> foreground flags per chunk of the call that made them
        self._gauss = {}
        self._geom = {}
        self._geomLast = None
        self._geom2 = {}
        self.statsCap = 2048        # room (labels) of the record table of label_and_stats; grows when a map has more
        self._scaleWork = {}        # raw stream handle -> (workspace bytes, per-lane buffers of nf_matched_filter_scales,
                                    #                       per lane the noiseKey its workspace's noise block tables belong to)

    # ---- tables --------------------------------------------------------------------------------------------
    def window(self, pow):
        if pow not in self._win:
            wy = np.sinc(np.fft.fftfreq(self.H)) ** pow
            wx = np.sinc(np.fft.fftfreq(self.W)) ** pow
            self._win[pow] = (to_dev(wy), to_dev(wx))
        return self._win[pow]

    def gauss_taps(self, sigma):
        key = float(sigma)
        if key not in self._gauss:
            w, rad = gaussian_kernel1d(sigma)
            self._gauss[key] = (to_dev(w), rad)
        return self._gauss[key]

    # ---- FFTs ----------------------------------------------------------------------------------------------
    def fft(self, x, prec=None, apod=False, unitary=True, half=False):
        """enmap.fft of [B,H,W] or [H,W]; real input is promoted to the complex type of ``prec``.

        ``half`` (real input only): a half-stored spectrum comes back - same [.., H, W] shape, but only the columns
        x <= W/2 hold values (include/nemofilt.h, "half-stored spectra").  Hand it only to ifft(half=True),
        cross_power(half=True) and build_filter(half=True)."""
        t = self.t
        x3 = x.reshape(-1, self.H, self.W)
        if x3.is_complex():
            inType = IN_COMPLEX
            prec = prec_of(x3)
        else:
            inType = IN_REAL64 if x3.dtype == t.float64 else IN_REAL32
            if prec is None:
                prec = NF_F64 if x3.dtype == t.float64 else NF_F32
        out = t.empty(x3.shape, dtype=t.complex128 if prec == NF_F64 else t.complex64, device=self.dev)
        if half and inType == IN_COMPLEX:
            raise NemoFiltError("half-stored spectra are spectra of real maps")
        fwd = self.lib.nf_fft2_forward_half if half else self.lib.nf_fft2_forward
        check(fwd(self.plan, prec, ptr(x3), inType, ptr(out), x3.shape[0],
                  ptr(self.apodY) if apod else None, ptr(self.apodX) if apod else None,
                  1 if unitary else 0, stream_ptr()), "nf_fft2_forward")
        return out.reshape(x.shape)

    def ifft(self, X, out='complex', unitary=False, clobber=False, filt=None, sum=False, window=None, half=False, planeScale=None,
             rows=None):
        """enmap.ifft; ``out`` in {'complex', 'f64', 'f32'} (the latter two keep the real part only).

        ``filt`` (float32 [N,H,W] or [H,W]) and ``window`` (power of the pixel window) are multiplied in while the
        transform loads its input (nemo/filters.py:847 and enmap.apply_window); with ``sum`` the N planes of X are summed
        after the product and one map comes out.  ``half``: X holds half-stored spectra (fft(half=True)); real output only.
        ``planeScale`` (half only): one more real factor per plane - the amplitudes of templates whose spectra are given for
        unit amplitude; ``rows`` = (lo, hi) (half only): only the output rows lo .. hi - 1 are computed, the rest of the
        returned map is uninitialised."""
        t = self.t
        X3 = X.reshape(-1, self.H, self.W)
        prec = prec_of(X3)
        if half and out == 'complex':
            raise NemoFiltError("the inverse transform of a half-stored spectrum is real")
        if (planeScale is not None or rows is not None) and not half:
            raise NemoFiltError("planeScale / rows belong to the inverse transform of half-stored spectra")
        if filt is not None or window is not None or half:
            return self._ifft_fused(X3, X.shape, out, unitary, clobber, filt, sum, window, half, planeScale, rows)
        work = X3 if clobber else t.empty_like(X3)
        if out == 'complex':
            res = t.empty_like(X3)
            outType = IN_COMPLEX
        elif out == 'f64':
            res = t.empty(X3.shape, dtype=t.float64, device=self.dev)
            outType = IN_REAL64
        else:
            res = t.empty(X3.shape, dtype=t.float32, device=self.dev)
            outType = IN_REAL32
        check(self.lib.nf_fft2_inverse(self.plan, prec, ptr(X3), ptr(work), ptr(res), outType, X3.shape[0],
                                       1 if unitary else 0, stream_ptr()), "nf_fft2_inverse")
        return res.reshape(X.shape)

    def _ifft_fused(self, X3, shape, out, unitary, clobber, filt, sum, window, half=False, planeScale=None, rows=None):
        t = self.t
        prec = prec_of(X3)
        N = X3.shape[0]
        nSum = N if sum else 1
        batch = 1 if sum else N
        f3 = None
        perBatch = 0
        if filt is not None:
            f3 = filt.reshape(-1, self.H, self.W)
            if half and f3.dtype == t.float32:
                f3 = self.symmetrise_filter(f3)          # the half-spectrum transform takes 0.5 (filt[k] + filt[-k])
            want = t.float64 if half else t.float32
            if f3.dtype != want or not f3.is_contiguous():
                raise NemoFiltError("the fused filter must be a contiguous float32 tensor (float64 if already symmetrised)")
            if f3.shape[0] not in (1, N) or (sum and f3.shape[0] != N):
                raise NemoFiltError("filter cube and Fourier cube have different numbers of planes")
            perBatch = 1 if (not sum and f3.shape[0] == N and N > 1) else 0
        wy = wx = None
        if window is not None:
            wy, wx = self.window(window)
        cdt = X3.dtype
        outShape = (1, self.H, self.W) if sum else tuple(X3.shape)
        work = X3 if clobber else t.empty(outShape, dtype=cdt, device=self.dev)
        if out == 'complex':
            res, outType = t.empty(outShape, dtype=cdt, device=self.dev), IN_COMPLEX
        elif out == 'f64':
            res, outType = t.empty(outShape, dtype=t.float64, device=self.dev), IN_REAL64
        else:
            res, outType = t.empty(outShape, dtype=t.float32, device=self.dev), IN_REAL32
        if half:
            ps = None
            if planeScale is not None:
                if len(planeScale) != N or N > 16:
                    raise NemoFiltError("planeScale: one factor per plane, at most 16 planes")
                ps = (ctypes.c_double * N)(*[float(v) for v in planeScale])
            rowLo, rowHi = (0, 0) if rows is None else (int(rows[0]), int(rows[1]))
            check(self.lib.nf_fft2_inverse_half(self.plan, prec, ptr(X3), ptr(f3), nSum, perBatch, ptr(wy), ptr(wx), ps, ptr(work),
                                                ptr(res), outType, batch, rowLo, rowHi, 1 if unitary else 0, stream_ptr()),
                  "nf_fft2_inverse_half")
        else:
            check(self.lib.nf_fft2_inverse_fused(self.plan, prec, ptr(X3), ptr(f3), nSum, perBatch, ptr(wy), ptr(wx), ptr(work),
                                                 ptr(res), outType, batch, 1 if unitary else 0, stream_ptr()), "nf_fft2_inverse_fused")
        if sum:
            return res.reshape(self.H, self.W)
        return res.reshape(shape)

    # ---- Fourier-domain pointwise --------------------------------------------------------------------------
    def symmetrise_filter(self, filt):
        """0.5 (filt[k] + filt[-k]) as float64, half-stored (nf_symmetrise_filter): what ifft(half=True) multiplies by."""
        t = self.t
        f3 = filt.reshape(-1, self.H, self.W)
        if f3.dtype != t.float32 or not f3.is_contiguous():
            raise NemoFiltError("symmetrise_filter takes a contiguous float32 filter")
        sym = t.empty(f3.shape, dtype=t.float64, device=self.dev)
        check(self.lib.nf_symmetrise_filter(ptr(f3), ptr(sym), f3.shape[0], self.H, self.W, stream_ptr()), "nf_symmetrise_filter")
        return sym.reshape(filt.shape)

    def cross_power(self, F, half=False):
        t = self.t
        N = F.shape[0]
        prec = prec_of(F)
        NP = t.empty((N, N, self.H, self.W), dtype=t.float64 if prec == NF_F64 else t.float32, device=self.dev)
        if half:
            check(self.lib.nf_cross_power_half(prec, ptr(F), ptr(NP), N, self.H, self.W, stream_ptr()), "nf_cross_power_half")
        else:
            check(self.lib.nf_cross_power(prec, ptr(F), ptr(NP), N, self.npix, stream_ptr()), "nf_cross_power")
        return NP

    def gaussian_filter(self, x, sigmaYX):
        """scipy.ndimage.gaussian_filter(x, (sy, sx)), mode reflect, over the last two axes of [..., H, W]."""
        t = self.t
        x3 = x.reshape(-1, self.H, self.W)
        wY, rY = self.gauss_taps(sigmaYX[0])
        wX, rX = self.gauss_taps(sigmaYX[1])
        tmp = t.empty_like(x3)
        out = t.empty_like(x3)
        check(self.lib.nf_gaussian_filter(prec_of(x3), ptr(x3), ptr(tmp), ptr(out), self.H, self.W, x3.shape[0],
                                          ptr(wY), rY, ptr(wX), rX, stream_ptr()), "nf_gaussian_filter")
        return out.reshape(x.shape)

    def build_filter(self, noiseCov, fSignals, w, half=False):
        t = self.t
        N = fSignals.shape[0]
        filt = t.empty((N, self.H, self.W), dtype=t.float32, device=self.dev)
        warr = (ctypes.c_double * N)(*[float(v) for v in w])
        if half:
            check(self.lib.nf_build_filter_half(prec_of(noiseCov), prec_of(fSignals), ptr(noiseCov), ptr(fSignals), warr,
                                                ptr(filt), N, self.H, self.W, stream_ptr()), "nf_build_filter_half")
        else:
            check(self.lib.nf_build_filter(prec_of(noiseCov), prec_of(fSignals), ptr(noiseCov), ptr(fSignals), warr,
                                           ptr(filt), N, self.npix, stream_ptr()), "nf_build_filter")
        return filt

    def build_filter_sym(self, noiseCov, fSignals, w):
        """build_filter(half=True) and symmetrise_filter in one pass: (filt float32 [N,H,W], filtSym float64 [N,H,W])."""
        t = self.t
        N = fSignals.shape[0]
        filt = t.empty((N, self.H, self.W), dtype=t.float32, device=self.dev)
        sym = t.empty((N, self.H, self.W), dtype=t.float64, device=self.dev)
        warr = (ctypes.c_double * N)(*[float(v) for v in w])
        check(self.lib.nf_build_filter_sym_half(prec_of(noiseCov), prec_of(fSignals), ptr(noiseCov), ptr(fSignals), warr, ptr(filt),
                                                ptr(sym), N, self.H, self.W, stream_ptr()), "nf_build_filter_sym_half")
        return filt, sym

    def apply_filter(self, F, filt, sum=True):
        t = self.t
        N = F.shape[0]
        G = t.empty((self.H, self.W) if sum else (N, self.H, self.W), dtype=F.dtype, device=self.dev)
        check(self.lib.nf_apply_filter(prec_of(F), ptr(F), ptr(filt), ptr(G), N, 1 if sum else 0, self.npix,
                                       stream_ptr()), "nf_apply_filter")
        return G

    def sum_planes(self, planes):
        """float32 [N,H,W] -> [H,W], planes added in order (ndarray.sum(axis=0); nemo/filters.py:847)."""
        t = self.t
        p3 = planes.reshape(-1, self.H, self.W)
        if p3.dtype != t.float32 or not p3.is_contiguous():
            raise NemoFiltError("sum_planes takes contiguous float32 planes")
        out = t.empty((self.H, self.W), dtype=t.float32, device=self.dev)
        check(self.lib.nf_sum_planes(ptr(p3), ptr(out), p3.shape[0], self.npix, stream_ptr()), "nf_sum_planes")
        return out

    def window_mul_(self, F, pow):
        wy, wx = self.window(pow)
        F3 = F.reshape(-1, self.H, self.W)
        check(self.lib.nf_window_mul(prec_of(F3), ptr(F3), ptr(wy), ptr(wx), self.H, self.W, F3.shape[0], stream_ptr()),
              "nf_window_mul")
        return F

    def _axis_window(self, pow, prec):
        """Transfer functions of enmap.apply_window along x and y in the order nf_axis_filter reads them (digit-reversed
        positions, divided by the length), or None where a side needs Bluestein's algorithm."""
        key = (float(pow), prec)
        if key not in self._axisWin:
            tabs = []
            for axis, n in ((0, self.W), (1, self.H)):
                rev = (ctypes.c_ushort * n)()
                blue = ctypes.c_int(0)
                check(self.lib.nf_plan_digit_reversal(self.plan, prec, axis, rev, ctypes.byref(blue)), "nf_plan_digit_reversal")
                if blue.value:
                    tabs = None
                    break
                perm = np.empty(n, dtype=np.float64)
                perm[np.frombuffer(rev, dtype=np.uint16).astype(np.int64)] = (np.sinc(np.fft.fftfreq(n)) ** pow) / n
                tabs.append(to_dev(perm))
            self._axisWin[key] = tabs
        return self._axisWin[key]

    def apply_window(self, m, pow, zero_ref=None, zeroRows=None):
        """enmap.apply_window(m, pow): real [H,W] (or [N,H,W]) maps in, real maps of the same dtype out.  The window is a product
        of a function of ky and one of kx: two 1-D circular filters, one pass over the map each (nf_axis_filter); the 2-D
        transform pair with the window fused into the inverse's load where a side's length needs Bluestein's algorithm.
        ``zero_ref``: pixels that are zero there are zero in the result (nemo/filters.py:99-101)."""
        t = self.t
        prec = prec_of(m)
        tabs = self._axis_window(pow, prec) if _AXIS_WINDOW else None
        if tabs is None:
            F = self.fft(m, unitary=True, half=True)
            if zero_ref is not None:
                return self.ifft_half_epi(F, window=pow, unitary=True, out='f64' if m.dtype == t.float64 else 'f32', zero_ref=zero_ref,
                                          clobber=True)
            return self.ifft(F, out='f64' if m.dtype == t.float64 else 'f32', unitary=True, clobber=True, window=pow, half=True)
        m3 = m.reshape(-1, self.H, self.W)
        if not m3.is_contiguous():
            m3 = m3.contiguous()
        if zero_ref is m:
            zero_ref = m3
        if zero_ref is not None and (zero_ref.dtype != m.dtype or not zero_ref.is_contiguous() or m3.shape[0] != 1):
            raise NemoFiltError("apply_window: zero_ref must be one contiguous map of the map's dtype")
        out = t.empty_like(m3)
        ty = 1 if m.dtype == t.float64 else 2
        # columns first: the rows pass (two workgroups per CU, contiguous runs) is the one that can afford the extra read of
        # the zero reference
        check(self.lib.nf_axis_filter(self.plan, prec, 1, ptr(m3), ty, ptr(out), ty, m3.shape[0], ptr(tabs[1]), None, None, stream_ptr()),
              "nf_axis_filter")
        check(self.lib.nf_axis_filter(self.plan, prec, 0, ptr(out), ty, ptr(out), ty, m3.shape[0], ptr(tabs[0]), ptr(zero_ref),
                                      ptr(zeroRows) if zero_ref is not None else None, stream_ptr()), "nf_axis_filter")
        return out.reshape(m.shape)

    # ---- templates -----------------------------------------------------------------------------------------
    def _tableDev(self, table):
        if not hasattr(table, '_dev'):
            # radii, profile and slopes in ONE upload (three views of it): every upload is host time on the enqueueing thread
            r_, p_ = np.asarray(table.r, dtype=np.float64), np.asarray(table.prof, dtype=np.float64)
            n_ = len(r_)
            # (tables on the same radii - the templates of a bank - share ONE device copy of them: nf_paint_radial_planes then
            # searches the interval of a pixel once for all such planes)
            rkey = (n_, hash(r_.tobytes()))
            r_d = _radiiDev.get(rkey)
            if r_d is None:
                if len(_radiiDev) >= 64:
                    _radiiDev.clear()
                r_d = _radiiDev[rkey] = to_dev(r_)
            packed = to_dev(np.concatenate([p_, (p_[1:] - p_[:-1]) / (r_[1:] - r_[:-1])]))
            table._dev = (r_d, packed[:n_], packed[n_:])
        return table._dev

    def _geomDev(self, geom):
        # the device copy is kept for the tables last seen: the same tuple object (every scale of a tile) costs one identity
        # test, anything else is compared by content
        if self._geomLast is not None and self._geomLast[0] is geom:
            return self._geomLast[1]
        g64 = tuple(np.ascontiguousarray(g, dtype=np.float64) for g in geom)
        key = hash(b"".join(g.tobytes() for g in g64))
        if key not in self._geom:
            # one upload for the three tables (views of it)
            packed = to_dev(np.concatenate(g64))
            offs = np.concatenate([[0], np.cumsum([len(g) for g in g64])])
            self._geom = {key: tuple(packed[int(offs[i]):int(offs[i + 1])] for i in range(len(g64)))}
        self._geomLast = (geom, self._geom[key])
        return self._geom[key]

    def paint2(self, geom, table, table2):
        """Two float32 templates [2, H, W] on the same radii in one pass (nf_paint_radial2)."""
        t = self.t
        if len(table.r) != len(table2.r) or table.r0 != table2.r0 or table.dlnr != table2.dlnr:
            raise NemoFiltError("paint2: the two tables must share their radii")
        A, B, C = self._geomDev(geom)
        r, prof, slope = self._tableDev(table)
        r2, prof2, slope2 = self._tableDev(table2)
        out = t.empty((2, self.H, self.W), dtype=t.float32, device=self.dev)
        check(self.lib.nf_paint_radial2(ptr(out[0]), ptr(out[1]), self.H, self.W, ptr(A), ptr(B), ptr(C), ptr(r), ptr(prof),
                                        ptr(slope), ptr(prof2), ptr(slope2), len(table.r), table.r0, table.dlnr, table.amp,
                                        table.sign, table2.amp, table2.sign, stream_ptr()), "nf_paint_radial2")
        return out

    def paint(self, geom, table, out=None):
        """float32 template from a signals.RadialTable and the (A, B, C) haversine tables (into ``out`` if given)."""
        t = self.t
        A, B, C = self._geomDev(geom)
        r, prof, slope = self._tableDev(table)
        if out is None:
            out = t.empty((self.H, self.W), dtype=t.float32, device=self.dev)
        elif out.dtype != t.float32 or tuple(out.shape) != (self.H, self.W) or not out.is_contiguous():
            raise NemoFiltError("paint: out must be a contiguous float32 map")
        check(self.lib.nf_paint_radial(ptr(out), self.H, self.W, ptr(A), ptr(B), ptr(C), ptr(r), ptr(prof), ptr(slope), len(table.r),
                                       table.r0, table.dlnr, table.amp, table.sign, stream_ptr()), "nf_paint_radial")
        return out

    # ---- one call per filter scale (nf_matched_filter_scales) ------------------------------------------------------
    def signal_norm(self, outputUnits, src, y0=0.0):
        """DeviceScalar of the filter normalisation: 'yc' y0 / src[0] in float64 (nemo/filters.py:659), 'uK' 1 / src[0] in
        float32 arithmetic (:684); ``src`` is a float64 device tensor, nobody waits for it."""
        t = self.t
        out = t.empty(1, dtype=t.float64, device=self.dev)
        check(self.lib.nf_signal_norm(0 if outputUnits == 'yc' else 1, ptr(src), float(y0), ptr(out), stream_ptr()), "nf_signal_norm")
        return DeviceScalar(out, wide=outputUnits == 'yc')

    def scale_batch(self):
        """Scales of a tile that walk nf_matched_filter_scales' sequence together (nf_mf_tile.scaleBatch): 8 on 1-2 Mpix tiles -
        one map's kernels leave most of the 256 CUs idle there - falling to 1 where a single map fills the chip and the
        scale-after-scale order keeps a scale's intermediates in the Infinity Cache.  SCALE_BATCH (module attribute) or
        NEMO_AMD_SCALE_BATCH override."""
        v = SCALE_BATCH if SCALE_BATCH is not None else os.environ.get('NEMO_AMD_SCALE_BATCH', None)
        if v is not None and int(v) > 0:
            return min(int(v), 8)
        return 8 if self.npix <= 2.5e6 else (4 if self.npix <= 5e6 else (2 if self.npix <= 9e6 else 1))

    def matched_filter_scales(self, tile, scales, lanes, noiseKey=None, weights=None):
        """nf_matched_filter_scales: ``tile`` an _lib.MfTile, ``scales`` a ctypes array of _lib.MfScale.  The lanes' workspaces
        are kept by the engine (one set per size) and reused call after call: calls are ordered by the stream they are made
        on.  ``noiseKey`` (hashable, or None) stands for the tile's median weights and noise grid: a lane whose workspace was
        last used under an equal key keeps its noise block tables (``noiseTablesLanes`` of the tile struct).  The key is a
        promise of the caller's; with NEMO_AMD_CHECK_NOISE_TABLES=1 (debugging: one device sum and a wait per call) the bits of
        ``weights`` are summed and held beside the key, and kept tables whose weights have changed since raise."""
        t = self.t
        n = len(scales)
        lanes = max(1, min(int(lanes), n, int(self.lib.nf_matched_filter_max_lanes())))
        tile.scaleBatch = self.scale_batch()
        nbytes = int(self.lib.nf_matched_filter_workspace_bytes(ctypes.byref(tile)))
        if nbytes == 0:
            check(-1, "nf_matched_filter_workspace_bytes")
        # the lanes' workspaces belong to the stream the calls are made on (calls on one stream are ordered; two streams - two
        # host threads - must not share them); one size at a time per stream: another configuration of the shape replaces it
        sp = stream_ptr()
        held = self._scaleWork.get(sp.value)
        if held is None or held[0] != nbytes:
            held = self._scaleWork[sp.value] = (nbytes, [], [])
        work, keys = held[1], held[2]
        while len(work) < lanes:
            work.append(t.empty(nbytes, dtype=t.uint8, device=self.dev))
            keys.append(None)
        if CHECK_NOISE_TABLES and noiseKey is not None and weights is not None:
            bits = weights.contiguous().view(t.int64 if weights.element_size() == 8 else t.int32)
            noiseKey = (noiseKey, int(bits.sum(dtype=t.int64).item()))
            for l in range(lanes):
                if keys[l] is not None and keys[l][0] == noiseKey[0] and keys[l] != noiseKey:
                    raise NemoFiltError("kept noise block tables are stale: the tile's weights changed under an unchanged noiseKey")
        tile.noiseTablesLanes = sum(1 << l for l in range(min(lanes, 32)) if noiseKey is not None and keys[l] == noiseKey)
        keys[:lanes] = [None] * lanes             # (an error below leaves the lanes' tables in an unknown state)
        ptrs = (ctypes.c_void_p * lanes)(*[w.data_ptr() for w in work[:lanes]])
        check(self.lib.nf_matched_filter_scales(self.plan, ctypes.byref(tile), scales, n, ptrs, lanes, sp),
              "nf_matched_filter_scales")
        keys[:lanes] = [noiseKey] * lanes

    # ---- real-space tail -----------------------------------------------------------------------------------
    def scale_mask(self, m32, signalNorm, psMask):
        t = self.t
        if isinstance(signalNorm, DeviceScalar):
            wide = signalNorm.wide
            out = t.empty((self.H, self.W), dtype=t.float64 if wide else t.float32, device=self.dev)
            check(self.lib.nf_scale_mask_dev(NF_F64 if wide else NF_F32, ptr(m32), ptr(out), ptr(signalNorm.tensor), ptr(psMask),
                                             self.npix, stream_ptr()), "nf_scale_mask_dev")
            return out
        # numpy >= 2 promotion: float32 array * np.float64 scalar -> float64; * Python float / np.float32 -> float32
        wide = isinstance(signalNorm, np.float64)
        vt = NF_F64 if wide else NF_F32
        out = t.empty((self.H, self.W), dtype=t.float64 if wide else t.float32, device=self.dev)
        check(self.lib.nf_scale_mask(vt, ptr(m32), ptr(out), float(signalNorm), ptr(psMask), self.npix, stream_ptr()),
              "nf_scale_mask")
        return out

    def median_weights(self, weightsList):
        t = self.t
        if len(weightsList) == 1:
            return weightsList[0]
        stack = t.stack(weightsList).contiguous()
        out = t.empty_like(weightsList[0])
        check(self.lib.nf_median_stack(prec_of(stack), ptr(stack), ptr(out), len(weightsList), self.npix, stream_ptr()),
              "nf_median_stack")
        return out

    def noise_grid(self, noiseParams, pixSizeDeg):
        """Grid-mode cells of MapFilter.makeNoiseMap (nemo/filters.py:407-420) as nf_noise_map takes them: (ny, nx, yEdges
        and xEdges on the device, overlap, numBins, quantiles (host doubles), estimator flag), or None when there is no
        cell (the RMS map then stays zero)."""
        if 'noiseGridArcmin' not in noiseParams or noiseParams['noiseGridArcmin'] is None:
            overlap, nx, ny = 0, 1, 1
        else:
            grid = int(round((noiseParams['noiseGridArcmin'] / 60.) / pixSizeDeg))
            overlap = int(grid / 2)
            nx = self.W / grid
            ny = self.H / grid
        yEdges = np.linspace(0, self.H, int(ny + 1), dtype=int).astype(np.int32)
        xEdges = np.linspace(0, self.W, int(nx + 1), dtype=int).astype(np.int32)
        numBins = noiseParams.get('numNoiseBins', 1)
        percentiles = np.arange(0, 100, 100 / numBins)
        nb = len(percentiles)
        quant = (ctypes.c_double * nb)(*[float(np.true_divide(p, 100)) for p in percentiles])
        ncy, ncx = len(yEdges) - 1, len(xEdges) - 1
        if ncy < 1 or ncx < 1:
            return None
        key = ('noiseEdges', yEdges.tobytes(), xEdges.tobytes())
        if key not in self._geom2:
            self._geom2[key] = (to_dev(yEdges), to_dev(xEdges))
        yE, xE = self._geom2[key]
        return ncy, ncx, yE, xE, overlap, nb, quant, 1 if noiseParams.get('RMSEstimator', None) == 'percentile' else 0

    def noise_map(self, filtered, medWeights, noiseParams, pixSizeDeg, work=None, tablesValid=False):
        """MapFilter.makeNoiseMap, grid mode (nemo/filters.py:407-478).  ``work`` / ``tablesValid``: the workspace of an earlier
        call for the same grid and weights (``self._lastNoiseWork[0]``), whose block tables are then reused (nf_noise_map)."""
        t = self.t
        estimator = noiseParams.get('RMSEstimator', None)
        if estimator not in (None, 'percentile'):
            # 'biweight' is astropy.stats.biweight_scale in the reference (nemo/filters.py:382,457): absent offline, unpinned
            raise NemoFiltError("RMSEstimator '%s' is not available on the GPU path (default clipped RMS or 'percentile')" % (estimator))
        if noiseParams.get('noiseGridArcmin', None) == "smart":                  # nemo/filters.py:363-404
            if 'numNoiseBins' not in noiseParams:
                raise Exception("Need to give numNoiseBins in noiseParams when using noiseGridArcmin = 'smart'")
            if estimator is not None:
                raise NemoFiltError("noiseGridArcmin: 'smart' is implemented for the default RMS estimator only")
            numBins = int(noiseParams['numNoiseBins'])
            rms = t.empty((self.H, self.W), dtype=t.float32, device=self.dev)
            if numBins < 2:
                rms.zero_()                      # np.linspace(..., 1) has no bins: the RMS map stays zero
                return rms
            wbytes = self.lib.nf_noise_smart_workspace_bytes()
            work = t.empty(wbytes // 8 + 1, dtype=t.float64, device=self.dev)
            check(self.lib.nf_noise_map_smart(prec_of(filtered), ptr(filtered), ptr(medWeights), prec_of(medWeights), ptr(rms),
                                              self.npix, numBins, ptr(work), work.numel() * 8, stream_ptr()), "nf_noise_map_smart")
            return rms
        grid = self.noise_grid(noiseParams, pixSizeDeg)
        rms = t.empty((self.H, self.W), dtype=t.float32, device=self.dev)
        if grid is None:
            rms.zero_()
            return rms
        ncy, ncx, yE, xE, overlap, nb, quant, est = grid
        wbytes = self.lib.nf_noise_workspace_bytes(self.H, self.W, ncy, ncx, nb)
        if work is None or work.numel() < wbytes // 8 + 1:
            work, tablesValid = t.empty(wbytes // 8 + 1, dtype=t.float64, device=self.dev), False
        check(self.lib.nf_noise_map(prec_of(filtered), ptr(filtered), ptr(medWeights), prec_of(medWeights), ptr(rms),
                                    self.H, self.W, ncy, ncx, ptr(yE), ptr(xE), overlap, nb, quant, est, ptr(work),
                                    work.numel() * 8, 1 if tablesValid else 0, stream_ptr()), "nf_noise_map")
        self._lastNoiseWork = (work, ncy, ncx, nb)       # diagnostics (tools/noisebench.py, the heavy-tail tests)
        return rms

    def edge_check(self, filtered, psMask, size):
        t = self.t
        tmp = t.empty((self.H, self.W), dtype=t.uint8, device=self.dev)
        out = t.empty((self.H, self.W), dtype=t.uint8, device=self.dev)
        check(self.lib.nf_edge_check(prec_of(filtered), ptr(filtered), ptr(psMask), ptr(tmp), ptr(out), self.H, self.W,
                                     int(size), stream_ptr()), "nf_edge_check")
        return out

    def finalize(self, filtered, rms, surveyMask, psMask, edgeCheck, mulSurvey):
        t = self.t
        sn = t.empty_like(filtered)
        rmsOut = rms                                   # masked in place (the caller's noise map is the one made for this call)
        maskOut = t.empty((self.H, self.W), dtype=t.uint8, device=self.dev)
        # largest / smallest S/N per 1024-pixel chunk: lets label_and_stats skip the chunks without foreground unread
        snChunks = t.empty(2 * ((self.npix + 1023) // 1024), dtype=t.int64, device=self.dev)
        check(self.lib.nf_finalize(prec_of(filtered), ptr(filtered), ptr(rms), ptr(surveyMask), ptr(psMask), ptr(edgeCheck),
                                   ptr(sn), ptr(rmsOut), ptr(maskOut), self.H, self.W, self.APOD_PIX, 1 if mulSurvey else 0,
                                   ptr(snChunks), None, stream_ptr()), "nf_finalize")
        self.lastSNChunks = (sn, snChunks)
        return filtered, sn, rmsOut, maskOut

    def finalize_cached(self, filtered, cachedRMS, surveyMask, psMask, edgeCheck, mulSurvey):
        """nf_finalize_cached: the masks of :meth:`finalize` and the S/N map of the driver's cached-noise branch in one pass.
        Returns (filtered map masked in place, float64 S/N map, mask, rows-with-a-zero table for the window undo)."""
        t = self.t
        sn = t.empty((self.H, self.W), dtype=t.float64, device=self.dev)
        maskOut = t.empty((self.H, self.W), dtype=t.uint8, device=self.dev)
        snChunks = t.empty(2 * ((self.npix + 1023) // 1024), dtype=t.int64, device=self.dev)
        zeroRows = t.empty(self.H, dtype=t.int32, device=self.dev)
        check(self.lib.nf_finalize_cached(prec_of(filtered), ptr(filtered), ptr(cachedRMS), ptr(surveyMask), ptr(psMask), ptr(edgeCheck),
                                          ptr(sn), ptr(maskOut), self.H, self.W, self.APOD_PIX, 1 if mulSurvey else 0, ptr(snChunks),
                                          ptr(zeroRows), stream_ptr()), "nf_finalize_cached")
        self.lastSNChunks = (sn, snChunks)
        return filtered, sn, maskOut, zeroRows

    def inject_model(self, base, model, weights=None, out=None):
        """nf_inject_model: base + (weights == 0 ? 0 : model) with numpy's dtype promotion (nemo/maps.py:341-344); ``base`` is
        not modified.  ``out``: a contiguous map of the promoted dtype to write into."""
        t = self.t
        odt = t.float32 if (base.dtype == t.float32 and model.dtype == t.float32) else t.float64
        if out is None:
            out = t.empty((self.H, self.W), dtype=odt, device=self.dev)
        elif out.dtype != odt or not out.is_contiguous():
            raise NemoFiltError("inject_model: out must be a contiguous map of the promoted dtype")
        check(self.lib.nf_inject_model(prec_of(base), ptr(base), prec_of(model), ptr(model), prec_of(weights) if weights is not None else 0,
                                       ptr(weights), ptr(out), self.npix, stream_ptr()), "nf_inject_model")
        return out

    def sn_cached(self, data, rms32):
        """nemo/pipelines.py:244-247: float64 S/N map from a filtered map and a cached float32 RMS map."""
        t = self.t
        sn = t.empty((self.H, self.W), dtype=t.float64, device=self.dev)
        check(self.lib.nf_sn_cached(prec_of(data), ptr(data), ptr(rms32), ptr(sn), self.npix, stream_ptr()), "nf_sn_cached")
        return sn

    def undo_pixel_window(self, data, zeroRows=None):
        """filterMaps' last step (nemo/filters.py:98-101): apply_window(data, -1), zeros of ``data`` kept (restored while the
        inverse transform stores its rows).  ``zeroRows``: the rows-with-a-zero table nf_finalize / nf_finalize_cached made for
        this very map (the zero reference is then looked up in those rows only)."""
        return self.apply_window(data, -1.0, zero_ref=data, zeroRows=zeroRows)

    def ifft_half_epi(self, X, filtSym=None, sum=False, window=None, unitary=False, out='f32', scale=None, mask=None,
                      zero_ref=None, clobber=False):
        """nf_fft2_inverse_half_epi: inverse transform of half-stored spectra into ONE real map with the pointwise steps that
        follow on the path fused into the store: ``scale`` (DeviceScalar or numpy scalar: float32(map) * scale, in float64
        if the scalar is wide, else float32 - then ``out`` follows the scalar) and ``mask`` (uint8), and / or ``zero_ref``."""
        t = self.t
        X3 = X.reshape(-1, self.H, self.W)
        N = X3.shape[0]
        if not sum and N != 1:
            raise NemoFiltError("ifft_half_epi makes one map: one plane, or sum=True")
        f3 = None
        if filtSym is not None:
            f3 = filtSym.reshape(-1, self.H, self.W)
            if f3.dtype == t.float32:
                f3 = self.symmetrise_filter(f3)
            if f3.dtype != t.float64 or not f3.is_contiguous() or f3.shape[0] != N:
                raise NemoFiltError("ifft_half_epi: the filter must be symmetrised float64 planes, one per Fourier plane")
        wy = wx = None
        if window is not None:
            wy, wx = self.window(window)
        sc = None
        if scale is not None:
            if not isinstance(scale, DeviceScalar):
                wide = isinstance(scale, np.float64)
                v = float(scale) if wide else float(np.float32(scale))
                scale = DeviceScalar(to_dev(np.array([v], dtype=np.float64)), wide=wide)
            sc = scale.tensor
            out = 'f64' if scale.wide else 'f32'
        if mask is not None and (mask.dtype != t.uint8 or not mask.is_contiguous()):
            raise NemoFiltError("ifft_half_epi: the mask must be a contiguous uint8 map")
        odt = t.float64 if out == 'f64' else t.float32
        if zero_ref is not None and (zero_ref.dtype != odt or not zero_ref.is_contiguous()):
            raise NemoFiltError("ifft_half_epi: zero_ref must be a contiguous map of the output dtype")
        work = X3 if clobber else t.empty((1, self.H, self.W), dtype=X3.dtype, device=self.dev)
        res = t.empty((self.H, self.W), dtype=odt, device=self.dev)
        check(self.lib.nf_fft2_inverse_half_epi(self.plan, prec_of(X3), ptr(X3), ptr(f3), N, ptr(wy), ptr(wx), ptr(work), ptr(res),
                                                IN_REAL64 if out == 'f64' else IN_REAL32, 1 if unitary else 0, ptr(sc), ptr(mask),
                                                ptr(zero_ref), stream_ptr()), "nf_fft2_inverse_half_epi")
        return res

    def subtract_background(self, data, sigmaYX):
        """maps.subtractBackground (nemo/maps.py:1279-1292)."""
        sm = self.gaussian_filter(data, sigmaYX)
        out = self.t.empty_like(data)
        check(self.lib.nf_subtract(prec_of(data), ptr(data), ptr(sm), ptr(out), data.numel(), stream_ptr()), "nf_subtract")
        return out

    def convolve(self, m, kern2d):
        """scipy.ndimage.convolve(float32 or float64 map, float64 kernel), reflect, output in the map dtype
        (nemo/filters.py:1193)."""
        t = self.t
        kf = to_dev(np.array(np.asarray(kern2d, dtype=np.float64)[::-1, ::-1], order='C', copy=True))
        out = t.empty_like(m)
        fn = self.lib.nf_convolve2d_f64 if m.dtype == t.float64 else self.lib.nf_convolve2d
        check(fn(ptr(m), ptr(out), self.H, self.W, ptr(kf), kern2d.shape[0], kern2d.shape[1], stream_ptr()), "nf_convolve2d")
        return out

    # ---- preprocessing (maps.MapDict.preprocess) -------------------------------------------------------------
    def median_filter(self, x, size):
        """scipy.ndimage.median_filter(x, size), mode reflect (nemo/maps.py:366-368)."""
        out = self.t.empty_like(x)
        check(self.lib.nf_median_filter(prec_of(x), ptr(x), ptr(out), self.H, self.W, int(size), stream_ptr()), "nf_median_filter")
        return out

    def mask_fill_discs(self, data, bck, surveyMask, psMask, recs, boxes, times60=True, persistentLimits=None):
        """Holes around catalogue positions: where the flat-grid distance to the object is below the limit both masks become
        0 and the data take the background (nemo/maps.py:419-424, 459-465).  recs [n, 5] float64 {x0, y0, xPixScale,
        yPixScale, limit}, boxes [n, 4] int32 {minX, maxX, minY, maxY} as makeDegreesDistanceMap forms them.  Independent
        holes by default; ``persistentLimits`` (n limits): the sequential form of maskPointSourcesFromCatalog, whose distance
        map is never reset between catalogue rows."""
        t = self.t
        recs = np.ascontiguousarray(recs, dtype=np.float64).reshape(-1, 5)
        boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 4)
        n = len(recs)
        if n == 0:
            return
        recs_d, boxes_d = to_dev(recs), to_dev(boxes)
        hostBoxes = boxes.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
        if persistentLimits is None:
            check(self.lib.nf_disc_apply(prec_of(data), ptr(data), ptr(bck), ptr(surveyMask), ptr(psMask), self.H, self.W, ptr(recs_d),
                                         ptr(boxes_d), hostBoxes, n, 0.0, 1 if times60 else 0, None, stream_ptr()), "nf_disc_apply")
            return
        # Sequential form.  Row k writes its box into the persistent map, then the WHOLE map is tested against row k's limit
        # (nemo/maps.py:419-424).  Outside the boxes the map holds 1e6, so only boxes need testing, and an earlier box j only
        # when row k's limit exceeds every limit its present values have already been tested against (tested[j]); a later box
        # overlapping j rewrites part of it, values that have met that later row's limit only.
        limits = np.asarray(persistentLimits, dtype=np.float64)
        tested = np.zeros(n)
        sched, starts = [], [0]
        for k in range(n):
            prev = np.flatnonzero(limits[k] > tested[:k])
            tested[prev] = limits[k]
            tested[k] = limits[k]
            b, bk = boxes[:k], boxes[k]
            over = np.flatnonzero((b[:, 0] < bk[1]) & (bk[0] < b[:, 1]) & (b[:, 2] < bk[3]) & (bk[2] < b[:, 3]))
            tested[over] = np.minimum(tested[over], limits[k])
            sched.append(np.concatenate([prev, [k]]))
            starts.append(starts[-1] + len(prev) + 1)
        order = np.concatenate(sched).astype(np.int64)
        schedBoxes = np.ascontiguousarray(boxes[order])
        schedBoxes_d = to_dev(schedBoxes)
        rmap = t.full((self.H, self.W), 1e6, dtype=t.float64, device=self.dev)
        for k in range(n):
            hb = boxes[k].ctypes.data_as(ctypes.POINTER(ctypes.c_int))
            check(self.lib.nf_disc_fill(ptr(rmap), self.H, self.W, ptr(recs_d[k]), ptr(boxes_d[k]), hb, stream_ptr()), "nf_disc_fill")
            a, e = starts[k], starts[k + 1]
            check(self.lib.nf_disc_apply(prec_of(data), ptr(data), ptr(bck), ptr(surveyMask), ptr(psMask), self.H, self.W, ptr(recs_d),
                                         ptr(schedBoxes_d[a:e]), schedBoxes[a:e].ctypes.data_as(ctypes.POINTER(ctypes.c_int)), e - a,
                                         float(limits[k]), 1 if times60 else 0, ptr(rmap), stream_ptr()), "nf_disc_apply")

    def paint_catalog(self, objs, wins, geom, profs, ntab, out=None, geomDev=None, profsDev=None):
        """Sum of the objects' radial templates, each inside its window, accumulated in float32 in catalogue order
        (nf_paint_catalog).  objs [n, 8] float64, wins [n, 8] int32, geom: packed haversine tables, profs [nProf, 3, ntab].
        ``geomDev`` / ``profsDev``: device copies of ``geom`` / ``profs`` the caller keeps (an injection run paints the same
        windows and the same profile pass after pass); the objects and their windows go up in ONE transfer."""
        t = self.t
        objs = np.ascontiguousarray(objs, dtype=np.float64).reshape(-1, 8)
        wins = np.ascontiguousarray(wins, dtype=np.int32).reshape(-1, 8)
        acc = 1
        if out is None:
            out = t.empty((self.H, self.W), dtype=t.float32, device=self.dev)
            acc = 0
        n = len(objs)
        if n == 0:
            if not acc:
                out.zero_()
            return out
        if geomDev is None:
            geom = np.ascontiguousarray(geom, dtype=np.float64)
            geomDev, geomSize = to_dev(geom), geom.size
        else:
            geomSize = geomDev.numel()
        if profsDev is None:
            profs = np.ascontiguousarray(profs, dtype=np.float64)
            profsDev = to_dev(profs)
        nProf = profsDev.numel() // (3 * int(ntab))
        packed = np.empty(n * 12, dtype=np.float64)           # [n, 8] float64 records, then [n, 8] int32 windows
        packed[:n * 8] = objs.reshape(-1)
        packed[n * 8:].view(np.int32)[:] = wins.reshape(-1)
        p_d = to_dev(packed)
        base = p_d.data_ptr()
        check(self.lib.nf_paint_catalog(ptr(out), self.H, self.W, n, ctypes.c_void_p(base), ctypes.c_void_p(base + n * 64),
                                        wins.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ptr(geomDev), geomSize, ptr(profsDev),
                                        nProf, int(ntab), acc, stream_ptr()), "nf_paint_catalog")
        return out

    def resample_bilinear(self, planes, H2, W2, yIdx, xIdx):
        """float32 planes [N, H, W] sampled by linear interpolation at fractional source positions (nf_resample_bilinear)."""
        t = self.t
        p3 = planes.reshape(-1, self.H, self.W)
        if p3.dtype != t.float32 or not p3.is_contiguous():
            p3 = p3.to(t.float32).contiguous()
        out = t.empty((p3.shape[0], int(H2), int(W2)), dtype=t.float32, device=self.dev)
        y_d, x_d = to_dev(np.ascontiguousarray(yIdx, dtype=np.float64)), to_dev(np.ascontiguousarray(xIdx, dtype=np.float64))
        check(self.lib.nf_resample_bilinear(ptr(p3), ptr(out), p3.shape[0], self.H, self.W, int(H2), int(W2), ptr(y_d), ptr(x_d),
                                            stream_ptr()), "nf_resample_bilinear")
        return out

    def add_model(self, data, model, flagMask=None, subtract=False):
        """data -= model, flagMask += (model > 1) (nemo/maps.py:435-437), or data += model (:344); in place."""
        check(self.lib.nf_add_model(prec_of(data), ptr(data), ptr(model), ptr(flagMask), data.numel(), 1 if subtract else 0,
                                    stream_ptr()), "nf_add_model")
        return data

    # ---- detection -----------------------------------------------------------------------------------------
    def label(self, sn, threshold, negate=False, slot=None):
        """scipy.ndimage.label numbering of (map > threshold): (labels tensor, number of objects).  ``slot`` (any hashable):
        the labels live in a buffer the engine keeps for this thread and slot - valid until the next call with the same slot
        - which is cleared and written only where foreground was / is (nf_label_components_kept); without it a new tensor."""
        t = self.t
        nchunks = (self.npix + 1023) // 1024
        work = t.empty(self.npix + 2 * nchunks + 16, dtype=t.int32, device=self.dev)
        n = ctypes.c_int(0)
        if slot is None:
            labels = t.empty((self.H, self.W), dtype=t.int32, device=self.dev)
            check(self.lib.nf_label_components(prec_of(sn), ptr(sn), float(threshold), 1 if negate else 0, ptr(labels), self.H,
                                               self.W, ptr(work), ctypes.byref(n), stream_ptr()), "nf_label_components")
        else:
            key = (threading.get_ident(), slot)
            if key not in self._keptLabels:
                self._keptLabels[key] = (t.zeros((self.H, self.W), dtype=t.int32, device=self.dev),
                                         t.zeros(nchunks, dtype=t.int32, device=self.dev))
            labels, dirty = self._keptLabels[key]
            check(self.lib.nf_label_components_kept(prec_of(sn), ptr(sn), float(threshold), 1 if negate else 0, ptr(labels), self.H,
                                                    self.W, ptr(work), ptr(dirty), ctypes.byref(n), stream_ptr()),
                  "nf_label_components_kept")
        off = int(self.lib.nf_label_chunk_flags_offset(self.H, self.W))
        if len(self._chunkFlags) > 16:
            self._chunkFlags.clear()         # (labels whose statistics were never asked for)
        self._chunkFlags[labels.data_ptr()] = work[off:off + nchunks]        # (keeps `work` alive for object_stats)
        return labels, n.value

    def label_and_stats(self, sn, threshold, negate=False, slot=None, cap=4096, snChunks=None):
        """label() and object_stats() as one enqueue that does not wait for the GPU (nf_label_and_stats): (labels, records)
        with records a float64 device tensor [(cap + 1), 6]; once fetched, records[0, 0] is the number of objects found - if
        it exceeds ``cap`` the records are incomplete and the call must be repeated with room for that many.  ``snChunks``:
        the S/N range table nf_finalize made for this very map (chunks that cannot hold foreground are then not read)."""
        t = self.t
        nchunks = (self.npix + 1023) // 1024
        work = t.empty(self.npix + 2 * nchunks + 16, dtype=t.int32, device=self.dev)
        dirty = None
        if slot is None:
            labels = t.empty((self.H, self.W), dtype=t.int32, device=self.dev)
        else:
            key = (threading.get_ident(), slot)
            if key not in self._keptLabels:
                self._keptLabels[key] = (t.zeros((self.H, self.W), dtype=t.int32, device=self.dev),
                                         t.zeros(nchunks, dtype=t.int32, device=self.dev))
            labels, dirty = self._keptLabels[key]
        stats = t.empty((cap + 1) * 17, dtype=t.float64, device=self.dev)             # NF_OBJECT_STATS_DOUBLES
        check(self.lib.nf_label_and_stats(prec_of(sn), ptr(sn), float(threshold), 1 if negate else 0, ptr(labels), self.H, self.W,
                                          ptr(work), ptr(dirty), int(cap), ptr(stats), ptr(snChunks), stream_ptr()), "nf_label_and_stats")
        return labels, stats[:(cap + 1) * 6].reshape(cap + 1, 6)

    def label_and_stats_batch(self, sns, threshold, negate=False, slot=None, cap=4096, snChunks=None):
        """label_and_stats() of several maps of this shape with every pass one launch for all of them
        (nf_label_and_stats_batch): (list of labels, records [len(sns), cap + 1, 6] - one device tensor, one transfer).  The
        labels of map k live in the buffer kept for (thread, (slot, k))."""
        t = self.t
        M = len(sns)
        nchunks = (self.npix + 1023) // 1024
        wlen = (self.npix + 2 * nchunks + 16 + 3) // 4 * 4
        work = t.empty((M, wlen), dtype=t.int32, device=self.dev)
        stats = t.empty((M, (cap + 1) * 17), dtype=t.float64, device=self.dev)        # NF_OBJECT_STATS_DOUBLES
        labels, dirty = [], []
        for k in range(M):
            if slot is None:
                labels.append(t.empty((self.H, self.W), dtype=t.int32, device=self.dev))
                dirty.append(None)
            else:
                key = (threading.get_ident(), (slot, k))
                if key not in self._keptLabels:
                    self._keptLabels[key] = (t.zeros((self.H, self.W), dtype=t.int32, device=self.dev),
                                             t.zeros(nchunks, dtype=t.int32, device=self.dev))
                lab, dr = self._keptLabels[key]
                labels.append(lab)
                dirty.append(dr)
        arr = ctypes.c_void_p * M
        pv = lambda xs: arr(*[ptr(x) for x in xs])          # noqa: E731
        wp, sp = work.data_ptr(), stats.data_ptr()
        check(self.lib.nf_label_and_stats_batch(prec_of(sns[0]), M, pv(sns), float(threshold), 1 if negate else 0, pv(labels), self.H,
                                                self.W, arr(*[wp + 4 * wlen * k for k in range(M)]), pv(dirty), int(cap),
                                                arr(*[sp + 8 * (cap + 1) * 17 * k for k in range(M)]),
                                                pv(snChunks if snChunks is not None else [None] * M), stream_ptr()),
              "nf_label_and_stats_batch")
        return labels, stats[:, :(cap + 1) * 6].reshape(M, cap + 1, 6)

    def object_stats(self, sn, labels, numObjects, negate=False):
        t = self.t
        stats = t.empty((numObjects + 1) * 17, dtype=t.float64, device=self.dev)      # NF_OBJECT_STATS_DOUBLES
        flags = self._chunkFlags.pop(labels.data_ptr(), None)        # the foreground flags of the call that made these labels
        if flags is not None:
            check(self.lib.nf_object_stats_chunks(prec_of(sn), ptr(sn), 1 if negate else 0, ptr(labels), self.H, self.W, numObjects,
                                                  ptr(stats), ptr(flags), stream_ptr()), "nf_object_stats_chunks")
        else:
            check(self.lib.nf_object_stats(prec_of(sn), ptr(sn), 1 if negate else 0, ptr(labels), self.H, self.W, numObjects,
                                           ptr(stats), stream_ptr()), "nf_object_stats")
        return stats[:(numObjects + 1) * 6].reshape(numObjects + 1, 6)

    def object_shapes(self, sn, labels, numObjects, negate=False):
        """Per label {xMin, yMin, xMax, yMax, S, Sx, Sy, Sxx, Syy, Sxy} (nemo/photometry.py:136-150), tensor [n+1, 10]."""
        t = self.t
        shapes = t.empty((numObjects + 1) * 30, dtype=t.float64, device=self.dev)     # NF_OBJECT_SHAPES_DOUBLES
        check(self.lib.nf_object_shapes(prec_of(sn), ptr(sn), 1 if negate else 0, ptr(labels), self.H, self.W, numObjects,
                                        ptr(shapes), stream_ptr()), "nf_object_shapes")
        return shapes[:(numObjects + 1) * 10].reshape(numObjects + 1, 10)

    def rms_table(self, rms, areaMask, flagMask, maxFlags, intersect, rowAreaDeg2):
        """Distinct non-zero values of a float32 noise map (ascending, as np.unique gives them), pixel counts, and survey
        area in deg^2 per value (nf_rms_table).  rowAreaDeg2: float32 per row, > 2^-27."""
        t = self.t
        ra = np.asarray(rowAreaDeg2, dtype=np.float64)
        if len(ra) != self.H or (ra < 2.0 ** -27).any() or (ra > 1.0).any():
            raise NemoFiltError("rms_table: one pixel area per row, between 2^-27 and 1 deg^2")
        fix = np.round(ra * 2.0 ** 50).astype(np.uint64)
        fix_d = to_dev(fix.view(np.int64))
        size = 1 << 16
        while True:
            keys = t.empty(size, dtype=t.int32, device=self.dev)
            cnt = t.empty(size, dtype=t.int64, device=self.dev)
            area = t.empty(size, dtype=t.int64, device=self.dev)
            over = t.empty(1, dtype=t.int32, device=self.dev)
            check(self.lib.nf_rms_table(ptr(rms), ptr(areaMask), ptr(flagMask), int(maxFlags) if maxFlags is not None else 0,
                                        ptr(intersect), ptr(fix_d), self.H, self.W, ptr(keys), ptr(cnt), ptr(area), size, ptr(over),
                                        stream_ptr()), "nf_rms_table")
            if int(over.item()) == 0:
                break
            size *= 16
            if size > (1 << 28):
                raise NemoFiltError("rms_table: the noise map has too many distinct values")
        used = keys != 0
        k = keys[used].cpu().numpy().view(np.float32)
        c = cnt[used].cpu().numpy()
        a = area[used].cpu().numpy().view(np.uint64).astype(np.float64) / 2.0 ** 50
        order = np.argsort(k, kind='stable')
        return k[order], c[order], a[order]

    def ring_flags(self, ringLabels, ringStats, numRing, minRingPix=30):
        t = self.t
        isRing = t.empty(numRing + 1, dtype=t.uint8, device=self.dev)
        rs = ringStats.contiguous()
        check(self.lib.nf_ring_flags(ptr(ringLabels), ptr(rs), numRing, minRingPix, self.H, self.W, ptr(isRing), stream_ptr()),
              "nf_ring_flags")
        return isRing

    def map_max(self, m, to_host=True):
        """(max value, flat index of its first occurrence) of a device map; with ``to_host=False`` the two stay on the
        device (float64 tensor of 2 elements) and the call does not wait for the GPU."""
        t = self.t
        out = t.empty(4 + 2 * 2048, dtype=t.float64, device=self.dev)      # NF_MAP_MAX_DOUBLES
        check(self.lib.nf_map_max(prec_of(m), ptr(m), m.numel(), ptr(out), stream_ptr()), "nf_map_max")
        if not to_host:
            return out[:2]
        res = out[:2].cpu().numpy()
        return res[0], int(res[1])

    def spline_points(self, yx, halfWin=32):
        """Evaluation points uploaded once (one transfer: the pairs, then the indices of the border points), for several
        spline evaluations at the same positions (different maps)."""
        return self.spline_points_multi([yx], halfWin)[0]

    def spline_points_multi(self, yxList, halfWin=32):
        """spline_points for several point sets in ONE upload (every upload is host time on the enqueueing thread)."""
        sets, total = [], 0
        for yx in yxList:
            yx = np.ascontiguousarray(np.asarray(yx, dtype=np.float64).reshape(-1, 2))
            n = len(yx)
            if n == 0:
                sets.append((yx, 0, None, 1, 0, 0))
                continue
            iy, ix = np.floor(yx[:, 0]), np.floor(yx[:, 1])
            interior = (iy - halfWin - 1 >= 0) & (iy + halfWin + 2 <= self.H - 1) & (ix - halfWin - 1 >= 0) & \
                (ix + halfWin + 2 <= self.W - 1) & (yx[:, 0] >= 0) & (yx[:, 0] <= self.H - 1) & (yx[:, 1] >= 0) & (yx[:, 1] <= self.W - 1)
            pointsClass = 1 if interior.all() else (2 if not interior.any() else 0)
            edgeIdx = np.flatnonzero(~interior).astype(np.int32)
            size = 2 * n + (len(edgeIdx) + 1) // 2
            sets.append((yx, n, edgeIdx, pointsClass, total, size))
            total += size
        if total == 0:
            return [_SplinePoints(None, 0, 1, halfWin) for _ in sets]
        buf = np.empty(total, dtype=np.float64)
        for yx, n, edgeIdx, pointsClass, off, size in sets:
            if n == 0:
                continue
            buf[off:off + 2 * n] = yx.reshape(-1)
            if len(edgeIdx) > 0:
                buf[off + 2 * n:off + size].view(np.int32)[:len(edgeIdx)] = edgeIdx
        dev = to_dev(buf)
        out = []
        for yx, n, edgeIdx, pointsClass, off, size in sets:
            if n == 0:
                out.append(_SplinePoints(None, 0, 1, halfWin))
                continue
            edge = dev[off + 2 * n:off + size].view(self.t.int32) if len(edgeIdx) > 0 else None
            out.append(_SplinePoints(dev[off:off + 2 * n], n, pointsClass, halfWin, edge, len(edgeIdx)))
        return out

    def spline_eval_batch(self, jobs, returnFlat=False, flatOnly=False):
        """[(map tensor, _SplinePoints, negate)] -> list of float64 device tensors, one launch per kernel for up to 32 jobs of a
        dtype (nf_spline_eval_batch).  The results are views of one buffer, job after job (``returnFlat``: that buffer too;
        ``flatOnly``: only that buffer - no view per job is made).  A point set with nEdge < 0 does not know which of its points
        lie near the border: every point is offered to both kernels; its ``dev`` may be a raw device address (int) into a
        tensor the caller keeps alive."""
        t = self.t
        outs = [None] * len(jobs)
        offs = [0] * len(jobs)
        pos = 0
        groups = {}
        for k, (m, pts, neg) in enumerate(jobs):
            offs[k] = pos
            pos += pts.n
            if pts.n > 0:
                groups.setdefault((prec_of(m), pts.halfWin), []).append(k)
        total = pos
        flat = t.empty(max(total, 1), dtype=t.float64, device=self.dev)
        if not flatOnly:
            for k, (m, pts, neg) in enumerate(jobs):
                outs[k] = flat[offs[k]:offs[k] + pts.n]
        VP, CI = ctypes.c_void_p, ctypes.c_int
        base = flat.data_ptr()
        for (prec, halfWin), ks in groups.items():
            for c0 in range(0, len(ks), 32):
                part = ks[c0:c0 + 32]
                nj = len(part)
                maps = (VP * nj)(*[jobs[k][0].data_ptr() for k in part])
                negs = (CI * nj)(*[1 if jobs[k][2] else 0 for k in part])
                yxp = [jobs[k][1].dev if isinstance(jobs[k][1].dev, int) else jobs[k][1].dev.data_ptr() for k in part]
                yxs = (VP * nj)(*yxp)
                ns = (CI * nj)(*[jobs[k][1].n for k in part])
                # (no border points: an empty list, not "unknown" - the pointer must still be non-null to say so)
                edges = (VP * nj)(*[(None if jobs[k][1].nEdge < 0 else (jobs[k][1].edge.data_ptr() if jobs[k][1].edge is not None else yxp[i]))
                                    for i, k in enumerate(part)])
                nes = (CI * nj)(*[max(jobs[k][1].nEdge, 0) for k in part])
                os_ = (VP * nj)(*[base + 8 * offs[k] for k in part])
                check(self.lib.nf_spline_eval_batch(prec, nj, maps, negs, yxs, ns, edges, nes, os_, self.H, self.W, halfWin, stream_ptr()),
                      "nf_spline_eval_batch")
        if flatOnly:
            return None, flat[:total]
        if returnFlat:
            return outs, flat[:total]
        return outs

    def spline_eval(self, m, yx, negate=False, halfWin=32, to_host=True):
        t = self.t
        if isinstance(yx, _SplinePoints):
            if yx.n == 0:
                return np.zeros(0)
            out = self.spline_eval_batch([(m, yx, negate)])[0]
            return out.cpu().numpy() if to_host else out
        yx = np.ascontiguousarray(np.asarray(yx, dtype=np.float64).reshape(-1, 2))
        if len(yx) == 0:
            return np.zeros(0)
        iy, ix = np.floor(yx[:, 0]), np.floor(yx[:, 1])
        interior = (iy - halfWin - 1 >= 0) & (iy + halfWin + 2 <= self.H - 1) & (ix - halfWin - 1 >= 0) & \
            (ix + halfWin + 2 <= self.W - 1) & (yx[:, 0] >= 0) & (yx[:, 0] <= self.H - 1) & (yx[:, 1] >= 0) & (yx[:, 1] <= self.W - 1)
        # which kernel serves which point is known here (the positions come from the host): tell the library, so that it
        # does not launch a kernel in which every workgroup would only find out that it has nothing to do.  (Launching the
        # two kernels on separate point subsets was tried: the extra small copies and scatter launches cost more than the
        # idle workgroups do.)
        pointsClass = 1 if interior.all() else (2 if not interior.any() else 0)
        yx_d = to_dev(yx)
        out = t.empty(len(yx), dtype=t.float64, device=self.dev)
        # mixed: the windowed-solve kernel (85 KB of LDS per workgroup) is launched for the edge points only
        edge = to_dev(np.flatnonzero(~interior).astype(np.int32)) if (pointsClass == 0 and _EDGE_SEL) else None
        check(self.lib.nf_spline_eval(prec_of(m), ptr(m), 1 if negate else 0, self.H, self.W, ptr(yx_d), len(yx), halfWin,
                                      pointsClass, ptr(edge), 0 if edge is None else edge.numel(), ptr(out), stream_ptr()),
              "nf_spline_eval")
        if not to_host:
            return out
        return out.cpu().numpy()
