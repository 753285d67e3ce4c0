"""Restatement of the four ``pixell.enmap`` primitives the hot path calls (TEST INFRASTRUCTURE).

pixell (>= 0.29, reference setup.cfg:32) is a third-party dependency that is absent from
/root/reference and not installed offline, so these follow its published behaviour; call sites in
the reference: nemo/filters.py:525,548,559,615,646,662,679,747,845,847,1150 (apod/fft/ifft) and
nemo/filters.py:100,644, nemo/pipelines.py:249 (apply_window).  PARITY UNPINNED (no golden vectors
exist in the reference for them); the restatement is self-consistent between oracle and GPU path.
"""
import numpy as np


def apod(m, width):
    """``enmap.apod(m, width)`` with the default cosine profile and zero fill: both ends of the last
    two axes are multiplied by ``0.5*(1-cos(linspace(0, pi, w)))`` (first sample 0, last exactly 1);
    axis -2 first, then axis -1, in place on a copy (so the map dtype is kept)."""
    m = np.asarray(m)
    w = np.minimum(np.zeros(2) + width, m.shape[-2:]).astype(np.int32)
    prof = [0.5 * (1 - np.cos(np.linspace(0, np.pi, int(k)))) for k in w]
    res = m.copy()
    if w[0] > 0:
        res[..., :w[0], :] *= prof[0][:, None]
        res[..., -w[0]:, :] *= prof[0][::-1, None]
    if w[1] > 0:
        res[..., :, :w[1]] *= prof[1][None, :]
        res[..., :, -w[1]:] *= prof[1][None, ::-1]
    return res


def _ctype(m):
    return np.result_type(np.asarray(m).dtype, 0j)


def fft(m):
    """``enmap.fft(m)``: full complex 2-D DFT over the last two axes (real input promoted to the matching
    complex type, no rfft), unitary normalisation (divide by sqrt(H*W))."""
    m = np.asarray(m)
    res = np.fft.fft2(m.astype(_ctype(m)), axes=(-2, -1))
    res /= np.prod(m.shape[-2:]) ** 0.5
    return res


def ifft(m, normalize=True):
    """``enmap.ifft(m, normalize=...)``: backward DFT; ``normalize=False`` is the plain backward sum
    (no 1/(H*W) at all), ``True`` divides it by sqrt(H*W)."""
    m = np.asarray(m)
    n = np.prod(m.shape[-2:])
    res = np.fft.ifft2(m.astype(_ctype(m)), axes=(-2, -1), norm='forward')   # unnormalised backward sum
    if normalize:
        res /= n ** 0.5
    return res


def calc_window(shape):
    return [np.sinc(np.fft.fftfreq(n)) for n in shape[-2:]]


def apply_window(m, pow=1.0):
    """``enmap.apply_window(m, pow)`` (order 0): multiply the unitary FFT by
    (sinc(f_y) sinc(f_x))**pow and transform back, keeping the real part."""
    wy, wx = calc_window(np.asarray(m).shape)
    f = fft(m)
    f *= wy[:, None] ** pow * wx[None, :] ** pow
    return ifft(f).real
