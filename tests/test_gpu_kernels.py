"""GPU parity tests, one per C-ABI entry point (include/nemofilt.h), against numpy / scipy - the libraries the
reference itself calls for these primitives.  Tolerances: bit-exact for integer / index
work and for the kernels that reproduce scipy's accumulation order; 1e-12 relative for fp64 FFTs, 2e-5 for fp32."""
import os

import numpy as np
import pytest
from scipy import ndimage, interpolate

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from nemo_amd import engine  # noqa: E402
from nemo_oracle import pixell_restated as enmap, filters as ofilt, photometry as ophot  # noqa: E402

SHAPES = [(96, 120), (211, 256), (64, 1031), (1031, 48), (129, 86), (840, 1440)]


def _eng(shape):
    return engine.TileEngine.get(shape)


def _np(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fft_forward_inverse(shape, dtype):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2,) + shape).astype(dtype)
    eng = _eng(shape)
    tol = 1e-12 if dtype == np.float64 else 3e-5
    F = _np(eng.fft(engine.to_dev(x)))
    ref = enmap.fft(x.astype(np.float64))
    assert F.dtype == (np.complex128 if dtype == np.float64 else np.complex64)
    assert np.abs(F - ref).max() / np.abs(ref).max() < tol
    # apodised forward
    Fa = _np(eng.fft(engine.to_dev(x), apod=True))
    refa = enmap.fft(enmap.apod(x, 20).astype(np.float64))
    assert np.abs(Fa - refa).max() / np.abs(refa).max() < tol
    # inverse (normalize=False), real part as float32 and complex
    G = (rng.standard_normal((2,) + shape) + 1j * rng.standard_normal((2,) + shape)).astype(F.dtype)
    back = _np(eng.ifft(engine.to_dev(G), out='complex'))
    refb = enmap.ifft(G.astype(np.complex128), normalize=False)
    assert np.abs(back - refb).max() / np.abs(refb).max() < tol
    r32 = _np(eng.ifft(engine.to_dev(G), out='f32'))
    assert r32.dtype == np.float32
    assert np.abs(r32 - refb.real).max() / np.abs(refb).max() < max(tol, 2e-7)
    # round trip
    rt = _np(eng.ifft(eng.fft(engine.to_dev(x)), out='complex', unitary=True))
    assert np.abs(rt.real - x).max() < (1e-12 if dtype == np.float64 else 2e-5)


def test_fft_4096_f64():
    shape = (4096, 4096)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(shape)
    F = _np(_eng(shape).fft(engine.to_dev(x)))
    ref = enmap.fft(x)
    assert np.abs(F - ref).max() / np.abs(ref).max() < 1e-12


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("N", [1, 2, 3])
def test_cross_power_gauss_build_apply(dtype, N):
    shape = (129, 86)
    rng = np.random.default_rng(11)
    eng = _eng(shape)
    ctype = np.complex128 if dtype == np.float64 else np.complex64
    F = (rng.standard_normal((N,) + shape) + 1j * rng.standard_normal((N,) + shape)).astype(ctype)
    dF = engine.to_dev(F)
    NP = _np(eng.cross_power(dF))
    ref = np.array([[np.real(F[i] * F[j].conj()) for j in range(N)] for i in range(N)])
    assert NP.dtype == dtype
    # numpy evaluates re*re + im*im without fusing; the kernel may contract one product into an fma
    assert np.abs(NP - ref).max() <= (1e-15 if dtype == np.float64 else 1e-6) * np.abs(ref).max()
    # Gaussian smoothing of exactly the same input: bit-exact against scipy
    sm = _np(eng.gaussian_filter(engine.to_dev(NP), (3, 3)))
    refsm = np.array([[ndimage.gaussian_filter(NP[i, j], (3, 3)) for j in range(N)] for i in range(N)])
    assert np.array_equal(sm, refsm)
    # filter build from a well-conditioned covariance
    cov = refsm.copy()
    for i in range(N):
        cov[i, i] += np.abs(cov).max()
    S = (rng.standard_normal((N,) + shape) + 1j * rng.standard_normal((N,) + shape)).astype(ctype)
    w = rng.uniform(-2, 2, N)
    cov[:, :, 5, 7] = 0       # singular mode stays zero
    filt = _np(eng.build_filter(engine.to_dev(cov), engine.to_dev(S), w))
    rf = np.zeros((N,) + shape, dtype=np.float32)
    for y in range(shape[0]):
        for x in range(shape[1]):
            try:
                rf[:, y, x] = np.dot(np.linalg.inv(cov[:, :, y, x]), w * abs(S[:, y, x]))
            except Exception:
                continue
    assert filt.dtype == np.float32
    assert np.all(filt[:, 5, 7] == 0)
    assert np.abs(filt - rf).max() <= (3e-7 if dtype == np.float64 else 2e-5) * np.abs(rf).max()
    # application: sum_i F_i filt_i and per-frequency products
    G = _np(eng.apply_filter(dF, engine.to_dev(rf), sum=True))
    refG = (F * rf).sum(axis=0)
    assert np.abs(G - refG).max() <= (1e-15 if dtype == np.float64 else 1e-6) * np.abs(refG).max()
    Gp = _np(eng.apply_filter(dF, engine.to_dev(rf), sum=False))
    assert np.abs(Gp - F * rf).max() <= (1e-15 if dtype == np.float64 else 1e-6) * np.abs(refG).max()


@pytest.mark.parametrize("shape", [(96, 120), (64, 1031), (840, 1440)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("N", [1, 3])
def test_fused_inverse(shape, dtype, N):
    """nf_fft2_inverse_fused: filter product, frequency sum and pixel-window factor folded into the inverse transform's
    load - against the separate numpy statements (nemo/filters.py:847, enmap.apply_window)."""
    rng = np.random.default_rng(21)
    eng = _eng(shape)
    ctype = np.complex128 if dtype == np.float64 else np.complex64
    F = (rng.standard_normal((N,) + shape) + 1j * rng.standard_normal((N,) + shape)).astype(ctype)
    filt = rng.standard_normal((N,) + shape).astype(np.float32)
    wy = np.sinc(np.fft.fftfreq(shape[0]))
    wx = np.sinc(np.fft.fftfreq(shape[1]))
    win = wy[:, None] * wx[None, :]
    tol = 1e-12 if dtype == np.float64 else 3e-5
    dF, dfilt = engine.to_dev(F), engine.to_dev(filt)
    # filter + sum (the matched-filter application)
    ref = np.real(enmap.ifft(F * filt, normalize=False)).sum(axis=0)
    got = _np(eng.ifft(dF, out='f64' if dtype == np.float64 else 'f32', filt=dfilt, sum=True))
    assert got.shape == shape
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    # window + filter + sum (the normalisation path), complex output
    ref = enmap.ifft(F * win * filt, normalize=False).sum(axis=0)
    got = _np(eng.ifft(dF, out='complex', filt=dfilt, sum=True, window=1.0))
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    # per-plane products (fRelWeights), unitary scaling
    ref = enmap.ifft(F * filt)
    got = _np(eng.ifft(dF, out='complex', unitary=True, filt=dfilt))
    assert got.shape == F.shape
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    # window only (pixel-window undo); the input must survive when clobber is off
    ref = np.real(enmap.ifft(F * win ** -1.0))
    got = _np(eng.ifft(dF, out='f64' if dtype == np.float64 else 'f32', unitary=True, window=-1.0))
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    assert np.array_equal(_np(dF), F)


@pytest.mark.parametrize("shape", [(96, 120), (129, 86), (64, 1031), (1031, 48), (840, 1440)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("N", [1, 2])
def test_half_stored_spectra(shape, dtype, N):
    """The *_half entry points (include/nemofilt.h): spectra of real maps stored for the columns x <= W/2 only.  Forward
    against enmap.fft of the same maps on those columns; cross power, filter build and the filtered inverse against the
    numpy statements on the full spectra (a filter that is NOT symmetric under k -> -k, like the smoothed noise power)."""
    rng = np.random.default_rng(31)
    eng = _eng(shape)
    H, W = shape
    hw = W // 2 + 1
    x = rng.standard_normal((N,) + shape).astype(dtype)
    tol = 1e-12 if dtype == np.float64 else 3e-5
    dx = engine.to_dev(x)
    for apod in (False, True):
        Fh = eng.fft(dx, apod=apod, half=True)
        ref = enmap.fft((enmap.apod(x, 20) if apod else x).astype(np.float64))
        assert np.abs(_np(Fh)[..., :hw] - ref[..., :hw]).max() <= tol * np.abs(ref).max()
    # the stored half is what the full-spectrum call writes there, bit for bit (same kernels, same order)
    Ffull = _np(eng.fft(dx, apod=True))
    assert np.array_equal(_np(Fh)[..., :hw], Ffull[..., :hw])
    # poison the columns that carry no values: nothing may read them
    Fh[..., hw:] = float('nan')
    NP = _np(eng.cross_power(Fh, half=True))
    refNP = _np(eng.cross_power(engine.to_dev(Ffull)))
    assert np.array_equal(NP, refNP)
    cov = np.abs(refNP) + np.eye(N)[:, :, None, None] * np.abs(refNP).max()
    w = rng.uniform(0.5, 2, N)
    filtH = _np(eng.build_filter(engine.to_dev(cov), Fh, w, half=True))
    filtF = _np(eng.build_filter(engine.to_dev(cov), engine.to_dev(Ffull), w))
    assert np.array_equal(filtH, filtF)
    # the one-pass form (filter and symmetrised filter together): both outputs bit for bit
    f1, sym1 = eng.build_filter_sym(engine.to_dev(cov), Fh, w)
    assert np.array_equal(_np(f1), filtH)
    assert np.array_equal(_np(sym1)[..., :hw], _np(eng.symmetrise_filter(engine.to_dev(filtH)))[..., :hw])
    # filtered inverse: asymmetric float32 filter, window, frequency sum
    filt = rng.standard_normal((N,) + shape).astype(np.float32)
    dfilt = engine.to_dev(filt)
    win = np.sinc(np.fft.fftfreq(H))[:, None] * np.sinc(np.fft.fftfreq(W))[None, :]
    out = 'f64' if dtype == np.float64 else 'f32'
    Fd = Ffull.astype(np.complex128)
    ref = np.real(enmap.ifft(Fd * filt, normalize=False)).sum(axis=0)
    got = _np(eng.ifft(Fh, out=out, filt=dfilt, sum=True, half=True))
    assert got.shape == shape
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    ref = np.real(enmap.ifft(Fd * win * filt, normalize=False))
    got = _np(eng.ifft(Fh, out=out, filt=dfilt, window=1.0, half=True))
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    # plain round trip
    got = _np(eng.ifft(Fh, out=out, unitary=True, half=True))
    assert np.abs(got - enmap.apod(x, 20)).max() <= (1e-12 if dtype == np.float64 else 2e-5)
    with pytest.raises(Exception):
        eng.ifft(Fh, out='complex', half=True)
    # per-plane factors (the amplitude of a template whose spectrum is given for unit amplitude) and a row range: the same
    # transform of the scaled spectra, and - bit for bit - the rows lo .. hi - 1 of the whole-map result
    ps = rng.uniform(-3, 3, N)
    ref = np.real(enmap.ifft(Fd * win * filt * ps[:, None, None], normalize=False))
    got = _np(eng.ifft(Fh, out=out, filt=dfilt, window=1.0, half=True, planeScale=ps))
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    gotSum = _np(eng.ifft(Fh, out=out, filt=dfilt, window=1.0, sum=True, half=True, planeScale=ps))
    assert np.abs(gotSum - ref.sum(axis=0)).max() <= tol * np.abs(ref).max()
    for lo, hi in ((H // 2 - 7, H // 2 + 9), (0, 5), (H - 3, H), (3, 4)):
        part = _np(eng.ifft(Fh, out=out, filt=dfilt, window=1.0, sum=True, half=True, planeScale=ps, rows=(lo, hi)))
        assert np.array_equal(part[lo:hi], gotSum[lo:hi]), (lo, hi)


@pytest.mark.parametrize("W", [80, 64, 4])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("N", [1, 2])
def test_two_pass_columns_of_4096_rows(W, dtype, N):
    """nf_fft2_inverse_half with 4096-long columns and one plane per output map: the column pass runs as two passes of
    64-point transforms (k_cols2p_first / k_cols2p_second, rows handed to the row pass in swapped order).  Against numpy on
    the full spectra: symmetrised filter, window, per-plane factors, the packed column 0 / W/2, a partial last column tile
    (W = 80: 40 columns in tiles of 32)."""
    shape = (4096, W)
    H = 4096
    rng = np.random.default_rng(77)
    eng = _eng(shape)
    x = rng.standard_normal((N,) + shape).astype(dtype)
    tol = 1e-12 if dtype == np.float64 else 3e-5
    out = 'f64' if dtype == np.float64 else 'f32'
    dx = engine.to_dev(x)
    Fh = eng.fft(dx, half=True)
    Fh[..., W // 2 + 1:] = float('nan')
    Fd = enmap.fft(x.astype(np.float64))
    filt = rng.standard_normal((N,) + shape).astype(np.float32)
    dfilt = engine.to_dev(filt)
    win = np.sinc(np.fft.fftfreq(H))[:, None] * np.sinc(np.fft.fftfreq(W))[None, :]
    ps = rng.uniform(-3, 3, N)
    ref = np.real(enmap.ifft(Fd * win * filt * ps[:, None, None], normalize=False))
    got = _np(eng.ifft(Fh, out=out, filt=dfilt, window=1.0, half=True, planeScale=ps))
    assert got.shape == x.shape
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    # no filter, no window: the plain round trip
    got = _np(eng.ifft(Fh, out=out, unitary=True, half=True))
    assert np.abs(got - x).max() <= (1e-12 if dtype == np.float64 else 2e-5)
    # filter alone
    ref = np.real(enmap.ifft(Fd * filt, normalize=False))
    got = _np(eng.ifft(Fh, out=out, filt=dfilt, half=True))
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gaussian_filter_and_background(dtype):
    shape = (211, 256)
    rng = np.random.default_rng(2)
    x = rng.standard_normal(shape).astype(dtype)
    eng = _eng(shape)
    for sig in [(3, 3), (7.3, 6.9), (0.8, 14.0)]:
        got = _np(eng.gaussian_filter(engine.to_dev(x), sig))
        assert np.array_equal(got, ndimage.gaussian_filter(x, sig))
    got = _np(eng.subtract_background(engine.to_dev(x), (7.3, 6.9)))
    assert np.array_equal(got, ofilt.subtract_background(x, (7.3, 6.9)))
    # radius larger than the map (reflection wraps more than once)
    small = (24, 40)
    xs = rng.standard_normal(small).astype(dtype)
    got = _np(_eng(small).gaussian_filter(engine.to_dev(xs), (9.0, 2.0)))
    assert np.array_equal(got, ndimage.gaussian_filter(xs, (9.0, 2.0)))
    # several LDS segments along x, several strips along y, a batch of planes
    wide = (130, 2300)
    xw = rng.standard_normal((2,) + wide).astype(dtype)
    got = _np(_eng(wide).gaussian_filter(engine.to_dev(xw), (3, 3)))
    for k in range(2):
        assert np.array_equal(got[k], ndimage.gaussian_filter(xw[k], (3, 3)))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_apply_window(dtype):
    shape = (129, 86)
    rng = np.random.default_rng(4)
    x = rng.standard_normal(shape).astype(dtype)
    eng = _eng(shape)
    for p in (1.0, -1.0):
        got = _np(eng.apply_window(engine.to_dev(x), p))
        ref = enmap.apply_window(x.astype(np.float64), p)
        assert got.dtype == dtype
        assert np.abs(got - ref).max() < (1e-12 if dtype == np.float64 else 2e-5)
    x[10:20, 10:20] = 0
    got = _np(eng.undo_pixel_window(engine.to_dev(x)))
    assert np.all(got[10:20, 10:20] == 0)
    ref = ofilt.undo_pixel_window(x.astype(np.float64))
    assert np.abs(got - ref).max() < (1e-12 if dtype == np.float64 else 2e-5)


def test_convolve2d():
    shape = (211, 256)
    rng = np.random.default_rng(8)
    x = rng.standard_normal(shape).astype(np.float32)
    eng = _eng(shape)
    for k in (29, 5, 1):
        kern = rng.standard_normal((k, k))
        kern[0, 0] = 0.0
        got = _np(eng.convolve(engine.to_dev(x), kern))
        assert np.array_equal(got, ndimage.convolve(x, kern))
    kern = rng.standard_normal((9, 13))
    got = _np(eng.convolve(engine.to_dev(x), kern))
    assert np.array_equal(got, ndimage.convolve(x, kern))


@pytest.mark.parametrize("vdtype", [np.float64, np.float32])
@pytest.mark.parametrize("numBins", [1, 2, 8])
def test_noise_map(vdtype, numBins):
    shape = (211, 256)
    rng = np.random.default_rng(21)
    m = rng.standard_normal(shape).astype(vdtype)
    m[rng.uniform(size=shape) > 0.999] += 40        # outliers the clipping must reject
    m[:, :9] = 0
    m[100:140, 60:130] = 0                           # a hole bigger than one cell's core
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    w1 = (1.0 + 3.0 * xx / shape[1] * yy / shape[0]).astype(vdtype)
    w2 = (2.0 + np.sin(xx / 17.0)).astype(vdtype)
    eng = _eng(shape)
    pix = 0.5 / 60.0
    for grid, wl in [(20.0, [w1]), (40.0, [w1, w2]), (None, [w1, w2, w2 * 0.5])]:
        params = {'noiseGridArcmin': grid, 'numNoiseBins': numBins}
        if grid is None:
            del params['noiseGridArcmin']
        ref = ofilt.make_noise_map(m, wl, params, pix)
        med = eng.median_weights([engine.to_dev(w) for w in wl])
        assert np.array_equal(_np(med), np.median(np.array(wl), axis=0))
        got = _np(eng.noise_map(engine.to_dev(m), med, params, pix))
        assert got.dtype == np.float32
        nz = ref > 0
        assert np.array_equal(got > 0, nz)
        # float64 sums are accumulated in a different order than numpy's pairwise sum: agree to ~1e-13, i.e. exact
        # after the float32 store except for rare 1-ulp flips
        assert np.abs(got - ref).max() <= 1.2e-7 * ref.max()
        # (float32 maps: numpy's mean / std accumulate in float32 pairwise order, the kernel in float64, so 1-ulp
        # differences of the float32 result are common there)
        assert (got != ref).mean() < (0.02 if vdtype == np.float64 else 0.6)
    # constant weights: every percentile edge coincides (only the last bin is populated)
    params = {'noiseGridArcmin': 20.0, 'numNoiseBins': numBins}
    wc = np.ones(shape, dtype=vdtype)
    ref = ofilt.make_noise_map(m, [wc], params, pix)
    got = _np(eng.noise_map(engine.to_dev(m), engine.to_dev(wc), params, pix))
    assert np.array_equal(got > 0, ref > 0)
    assert np.abs(got - ref).max() <= 1.2e-7 * max(ref.max(), 1e-30)
    # all-zero map
    got = _np(eng.noise_map(engine.to_dev(np.zeros(shape, dtype=vdtype)), engine.to_dev(wc), params, pix))
    assert np.all(got == 0)


@pytest.mark.parametrize("vdtype", [np.float64, np.float32])
def test_noise_map_heavy_tails(vdtype):
    """Blocks under bright extended objects (more than 256 values beyond the core threshold) take the magnitude-class
    route of the block path (k_noise_block_classes): same clipped RMS as the oracle's
    per-window evaluation (nemo/filters.py:407-478), and the route is really taken (workspace header counter)."""
    shape = (420, 530)
    rng = np.random.default_rng(77)
    m = rng.standard_normal(shape)
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    for (cy, cx, amp, sig) in [(100, 120, 60.0, 14.0), (101, 300, -35.0, 20.0), (300, 260, 400.0, 9.0), (330, 60, 25.0, 30.0),
                               (210, 500, 80.0, 12.0)]:
        m += amp * np.exp(-0.5 * ((yy - cy) ** 2 + (xx - cx) ** 2) / sig ** 2)
    m[:, :5] = 0
    m = m.astype(vdtype)
    w = (1.0 + 0.2 * np.sin(xx / 40.0) * np.cos(yy / 55.0)).astype(vdtype)
    eng = _eng(shape)
    pix = 0.5 / 60.0
    for grid in (40.0, 20.0):
        params = {'noiseGridArcmin': grid, 'numNoiseBins': 1}
        ref = ofilt.make_noise_map(m, [w], params, pix)
        got = _np(eng.noise_map(engine.to_dev(m), engine.to_dev(w), params, pix))
        work, ncy, ncx, nb = eng._lastNoiseWork
        hdr = work.view(torch.int32)[ncy * ncx * 4 * 2:ncy * ncx * 4 * 2 + 8].cpu().numpy()
        assert hdr[4] > 0, "no heavy block in this case: the test does not exercise the class route"
        assert np.array_equal(got > 0, ref > 0)
        assert np.abs(got - ref).max() <= 1.2e-7 * ref.max()
        assert (got != ref).mean() < (0.02 if vdtype == np.float64 else 0.6)


@pytest.mark.parametrize("vdtype", [np.float64, np.float32])
def test_noise_map_with_kept_block_tables(vdtype):
    """tablesValid of nf_noise_map: a workspace that was last used with the same grid and weights keeps its block tables; the
    next map (other values, other holes, heavy blocks) gets bit for bit the noise map of a fresh call."""
    shape = (420, 530)
    rng = np.random.default_rng(78)
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    w = (1.0 + 0.2 * np.sin(xx / 40.0) * np.cos(yy / 55.0)).astype(vdtype)
    w[200:260, 100:180] *= 3.0                      # blocks whose weights leave their windows' ranges (painted pixel by pixel)
    maps = []
    for k in range(3):
        m = rng.standard_normal(shape) * (1.0 + k)
        if k == 1:
            m += 300.0 * np.exp(-0.5 * ((yy - 150) ** 2 + (xx - 300) ** 2) / 11.0 ** 2)
            m[90:130, 400:470] = 0                   # a hole the first map did not have: blocks valid in part, blocks empty
            m[rng.uniform(size=shape) > 0.9995] = 0
        if k == 2:
            m[:, :37] = 0
        maps.append(engine.to_dev(m.astype(vdtype)))
    eng = _eng(shape)
    dw = engine.to_dev(w)
    pix = 0.5 / 60.0
    for grid in (40.0, 20.0):
        params = {'noiseGridArcmin': grid, 'numNoiseBins': 1}
        fresh = [_np(eng.noise_map(m, dw, params, pix)) for m in maps]
        work = None
        for k, m in enumerate(maps):
            got = _np(eng.noise_map(m, dw, params, pix, work=work, tablesValid=work is not None))
            work = eng._lastNoiseWork[0]
            assert np.array_equal(got, fresh[k])
        ref = ofilt.make_noise_map(_np(maps[1]), [w], params, pix)
        assert np.array_equal(fresh[1] > 0, ref > 0) and np.abs(fresh[1] - ref).max() <= 1.2e-7 * ref.max()


@pytest.mark.parametrize("vdtype", [np.float64, np.float32])
def test_edge_check_and_finalize(vdtype):
    shape = (129, 86)
    rng = np.random.default_rng(31)
    f = rng.standard_normal(shape).astype(vdtype)
    f[:, :7] = 0
    f[50:60, 30:45] = 0
    ps = np.ones(shape, dtype=np.uint8)
    ps[50:60, 30:45] = 0
    ps[80:84, 10:14] = 0
    f = f * ps
    sm = np.ones(shape, dtype=np.uint8)
    sm[:, -9:] = 0
    rms = np.abs(rng.standard_normal(shape)).astype(np.float32)
    rms[20:30, 20:30] = 0
    eng = _eng(shape)
    for size in (6, 7):
        ref = ndimage.rank_filter(abs(f + (1 - ps)), 0, size=(size, size)) > 0
        got = _np(eng.edge_check(engine.to_dev(f), engine.to_dev(ps), size))
        assert np.array_equal(got > 0, ref)
    for trim, fourier in [(6, True), (0, True), (6, False)]:
        fm, sn, smask = ofilt._snr_and_masks(f.copy(), rms, sm, ps, trim, fourier)
        if fourier:
            fm = fm * smask
        apod = np.equal(enmap.apod(np.ones(shape), 20), 1)
        smask = smask * apod
        sn = sn * smask
        sn[np.isnan(sn)] = 0
        rr = rms * smask
        edge = eng.edge_check(engine.to_dev(f), engine.to_dev(ps), trim) if trim > 0 else None
        d, s, r, mk = eng.finalize(engine.to_dev(f.copy()), engine.to_dev(rms), engine.to_dev(sm), engine.to_dev(ps), edge,
                                   fourier)
        assert np.array_equal(_np(mk) > 0, smask > 0)
        assert np.array_equal(_np(d), fm.astype(vdtype))
        assert np.array_equal(_np(s), sn.astype(vdtype))
        assert np.array_equal(_np(r), rr.astype(np.float32))


def _blob_map(shape, seed, dtype):
    rng = np.random.default_rng(seed)
    m = ndimage.gaussian_filter(rng.standard_normal(shape), 1.5)
    m = m / m.std()
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    # a ring (centre of mass outside itself), a U shape (labels that merge late), a long snake
    r = np.hypot(yy - 60, xx - 70)
    m[(r > 8) & (r < 11)] = 8.0
    m[100:130, 20] = 7.0
    m[100:130, 40] = 7.0
    m[130, 20:41] = 7.0
    for k in range(0, 60, 4):
        m[150 + (k // 4) % 2 * 3:154 + (k // 4) % 2 * 3, 100 + k:104 + k] = 6.0
    # no exact ties: ndimage.maximum_position's choice among equal maxima depends on an unstable sort
    m = m + 1e-3 * rng.standard_normal(shape)
    return m.astype(dtype)


@pytest.mark.parametrize("vdtype", [np.float64, np.float32])
@pytest.mark.parametrize("thr", [4.0, 0.5, 0.0])
def test_label_and_object_stats(vdtype, thr):
    shape = (211, 256)
    m = _blob_map(shape, 5, vdtype)
    eng = _eng(shape)
    d = engine.to_dev(m)
    labels, n = eng.label(d, thr)
    refLab, refN = ndimage.label(m > thr)
    assert n == refN
    assert np.array_equal(_np(labels), refLab)
    st = _np(eng.object_stats(d, labels, n))
    ids = np.arange(1, n + 1)
    cnt = ndimage.sum(m > thr, labels=refLab, index=ids)
    assert np.array_equal(st[1:, 0], cnt)
    com = np.array(ndimage.center_of_mass(m, labels=refLab, index=ids))
    got = np.stack([st[1:, 2] / st[1:, 1], st[1:, 3] / st[1:, 1]], axis=1)
    big = np.abs(st[1:, 1]) > 1e-3         # centre of mass is ill-conditioned when sum v ~ 0 (threshold 0)
    assert np.abs(got - com)[big].max() < 1e-9
    mp = np.array(ndimage.maximum_position(m, labels=refLab, index=ids))
    assert np.array_equal(st[1:, 5].astype(int), mp[:, 0] * shape[1] + mp[:, 1])
    assert np.array_equal(st[1:, 4], ndimage.maximum(m, labels=refLab, index=ids).astype(np.float64))
    # negated map
    labels2, n2 = eng.label(d, 2.0, negate=True)
    refLab2, refN2 = ndimage.label((m * -1) > 2.0)
    assert n2 == refN2 and np.array_equal(_np(labels2), refLab2)


def test_label_edge_cases():
    shape = (64, 1031)
    eng = _eng(shape)
    z = np.zeros(shape)
    labels, n = eng.label(engine.to_dev(z), 1.0)
    assert n == 0 and not _np(labels).any()
    labels, n = eng.label(engine.to_dev(z + 2), 1.0)
    assert n == 1 and np.all(_np(labels) == 1)
    # checkerboard: every foreground pixel is its own 4-connected component
    cb = (np.indices(shape).sum(axis=0) % 2).astype(np.float64) * 3
    labels, n = eng.label(engine.to_dev(cb), 1.0)
    refLab, refN = ndimage.label(cb > 1.0)
    assert n == refN and np.array_equal(_np(labels), refLab)
    # comb: one component whose teeth meet only in the last row
    comb = np.zeros(shape)
    comb[:, ::2] = 3
    comb[-1, :] = 3
    labels, n = eng.label(engine.to_dev(comb), 1.0)
    assert n == 1 and np.array_equal(_np(labels) > 0, comb > 1)


def test_map_max():
    shape = (129, 86)
    rng = np.random.default_rng(9)
    m = rng.standard_normal(shape).astype(np.float32)
    m[40, 40] = m[90, 3] = 11.0
    v, i = _eng(shape).map_max(engine.to_dev(m))
    assert v == 11.0 and i == np.argmax(m)
    v, i = _eng(shape).map_max(engine.to_dev(-np.abs(m).astype(np.float64) - 1))
    assert v == (-np.abs(m).astype(np.float64) - 1).max() and i == np.argmax(-np.abs(m).astype(np.float64) - 1)


@pytest.mark.parametrize("vdtype", [np.float64, np.float32])
def test_spline_eval(vdtype):
    shape = (211, 256)
    rng = np.random.default_rng(13)
    m = ndimage.gaussian_filter(rng.standard_normal(shape), 1.2).astype(vdtype)
    spl = interpolate.RectBivariateSpline(np.arange(shape[0]), np.arange(shape[1]), m, kx=3, ky=3)
    pts = np.stack([rng.uniform(0, shape[0] - 1, 200), rng.uniform(0, shape[1] - 1, 200)], axis=1)
    pts[:6] = [[0, 0], [shape[0] - 1, shape[1] - 1], [0.3, 100.2], [105.5, 0.7], [105.0, 128.0], [209.9, 254.9]]
    got = _eng(shape).spline_eval(engine.to_dev(m), pts)
    ref = np.array([spl(y, x)[0][0] for y, x in pts])
    assert np.abs(got - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()) * (1 if vdtype == np.float64 else 10)


def test_ring_flags_match_oracle():
    shape = (211, 256)
    m = _blob_map(shape, 6, np.float64)
    eng = _eng(shape)
    d = engine.to_dev(m)
    labels, n = eng.label(d, 0.5)
    st = eng.object_stats(d, labels, n)
    isRing = _np(eng.ring_flags(labels, st, n, minRingPix=30))
    ringMask = ophot.ring_mask(m, 0.5)
    lab = _np(labels)
    assert np.array_equal(isRing[lab] > 0, ringMask > 0)
    assert ringMask.sum() > 0


def test_paint_radial_matches_numpy_painter():
    import helpers
    from nemo_amd import signals
    from nemo_amd.wcs import CARWCS
    shape = (129, 86)
    wcs = CARWCS.make(shape, RADeg=10.0, decDeg=-12.0)
    beam = signals.BeamProfile(FWHMArcmin=1.4)
    eng = _eng(shape)
    for kind, params, amp in [('beam', {}, None), ('arnaud', {'z': 0.4, 'M500MSun': 2e14}, None),
                              ('arnaud', {'z': 0.2, 'M500MSun': 8e14}, -3.7e2), ('beam', {}, 612.0)]:
        for (ra, dec) in [wcs.getCentreWCSCoords(), wcs.pix2wcs(3.3, 100.9)]:
            geom = signals.paintGeometry(shape, wcs, ra, dec)
            table = signals.templateTables(kind, beam, params, amplitude=amp)
            got = _np(eng.paint(geom, table))
            ref = helpers.cpu_paint(geom, table)
            assert got.dtype == np.float32
            assert np.array_equal(got, ref)


def test_xmatch_self():
    """nf_xmatch_self against the host implementation of the 'ref-forced' merge (catalogs._pairsWithin + _bestPerGroup):
    groups of jittered positions, exact duplicates, equal SNRs (first partner wins), a NaN SNR, isolated rows."""
    from nemo_amd import catalogs
    rng = np.random.default_rng(77)
    base_ra, base_dec = rng.uniform(10, 44, 300), rng.uniform(-17, 17, 300)
    ra = np.concatenate([base_ra + rng.normal(0, 0.006, 300) for _ in range(7)] + [rng.uniform(10, 44, 500), base_ra[:40]])
    dec = np.concatenate([base_dec + rng.normal(0, 0.006, 300) for _ in range(7)] + [rng.uniform(-17, 17, 500), base_dec[:40]])
    ra[-40:] = ra[:40]
    dec[-40:] = dec[:40]                       # exact duplicates of the first rows
    snr = rng.uniform(4, 40, len(ra))
    snr[300:600] = snr[:300]                   # ties inside groups
    snr[17] = np.nan
    n = len(ra)
    i, j = catalogs._pairsWithin(ra, dec, ra, dec, catalogs.XMATCH_RADIUS_DEG)
    count, best = catalogs._bestPerGroup(i, j, snr, n)
    toRemove = np.zeros(n, dtype=bool)
    drop = (count[i] > 1) & (j != best[i])
    toRemove[j[drop]] = True
    res = engine.xmatch_self(ra, dec, snr, catalogs.XMATCH_RADIUS_DEG)
    assert res is not None
    assert np.array_equal(res[0], count)
    assert np.array_equal(res[1], best)
    assert np.array_equal(res[2], toRemove)
    assert toRemove.sum() > 1000 and (~toRemove).sum() > 500
    # a pair exactly at the limit is handed back to the host
    d = catalogs.XMATCH_RADIUS_DEG
    assert engine.xmatch_self(np.array([10.0, 10.0]), np.array([0.0, d]), np.array([5.0, 6.0]), d) is None


@pytest.mark.parametrize("shape,dtype,pow,batch", [((64, 96), np.float64, -1.0, 1), ((63, 99), np.float64, 1.0, 1), ((120, 90), np.float32, -1.0, 1),
                                                   ((45, 77), np.float32, 1.0, 3), ((256, 512), np.float64, -1.0, 2), ((39, 1024), np.float32, -1.0, 1),
                                                   ((1024, 26), np.float64, 1.0, 1)])
def test_apply_window_as_two_axis_filters(shape, dtype, pow, batch):
    """enmap.apply_window through nf_axis_filter (one pass per axis, spectra never leave LDS) against the oracle's statement of
    it (2-D transforms): odd and even sides, both precisions, several maps at once, zeros restored."""
    from nemo_oracle import pixell_restated
    rng = np.random.default_rng(shape[0] + shape[1])
    m = rng.normal(0, 1, (batch,) + shape).astype(dtype)
    m[:, 3:9, 5:20] = 0
    eng = engine.TileEngine.get(shape)
    assert eng._axis_window(pow, engine.prec_of(engine.to_dev(m))) is not None
    got = eng.apply_window(engine.to_dev(m if batch > 1 else m[0]), pow).cpu().numpy()
    want = np.array([pixell_restated.apply_window(x, pow=pow) for x in m]).astype(dtype)
    tol = 2e-6 if dtype == np.float32 else 1e-13
    assert got.dtype == dtype
    assert np.abs(got.reshape(want.shape) - want).max() <= tol * np.abs(want).max()
    if batch == 1:
        d = engine.to_dev(m[0])
        z = eng.apply_window(d, pow, zero_ref=d).cpu().numpy()
        assert np.all(z[m[0] == 0] == 0)
        assert np.array_equal(z[m[0] != 0], got.reshape(shape)[m[0] != 0])


@pytest.mark.parametrize("vdtype", [np.float64, np.float32])
def test_label_and_stats_with_sn_chunk_ranges(vdtype):
    """nf_finalize's per-chunk S/N range table lets nf_label_and_stats skip chunks unread: same labels and records as without
    it, for both signs of the detection, and the same as the two-call form (label + object_stats)."""
    shape = (300, 517)
    eng = _eng(shape)
    rng = np.random.default_rng(77)
    f = rng.standard_normal(shape).astype(vdtype) * 1.3
    f[40:43, 100:104] += 9.0
    f[200:202, 300:303] -= 8.0
    rms = np.ones(shape, dtype=np.float32)
    ones = np.ones(shape, dtype=np.uint8)
    d, sn, r, mk = eng.finalize(engine.to_dev(f.copy()), engine.to_dev(rms), engine.to_dev(ones), engine.to_dev(ones), None, True)
    snT, chunks = eng.lastSNChunks
    assert snT is sn
    snH = _np(sn)
    # the table holds the chunk extrema of the S/N map that was written
    flat = np.concatenate([snH.reshape(-1), np.full((-snH.size) % 1024, np.nan)]).reshape(-1, 1024)
    keys = _np(chunks).view(np.uint64)
    nch = flat.shape[0]

    def unkey(k):
        b = np.where(k & np.uint64(1 << 63), k & np.uint64((1 << 63) - 1), ~k)
        return b.view(np.float64)
    assert np.array_equal(unkey(keys[:nch]), np.nanmax(flat, axis=1).astype(np.float64))
    assert np.array_equal(unkey(keys[nch:]), np.nanmin(flat, axis=1).astype(np.float64))
    for negate in (False, True):
        for thr in (4.0, 2.5):
            labA, recA = eng.label_and_stats(sn, thr, negate=negate, cap=20000)
            labB, recB = eng.label_and_stats(sn, thr, negate=negate, cap=20000, snChunks=chunks)
            a, b = _np(recA), _np(recB)
            n = int(a[0, 0])
            assert n > 0 and n == int(b[0, 0])
            assert np.array_equal(a[1:n + 1], b[1:n + 1])      # (record 0 is the background: count slot = n, the rest unused)
            assert np.array_equal(_np(labA), _np(labB))
            labC, nC = eng.label(sn, thr, negate=negate)
            stC = _np(eng.object_stats(sn, labC, nC, negate=negate))
            assert nC == n and np.array_equal(_np(labC), _np(labA))
            assert np.array_equal(stC[1:], a[1:n + 1])


@pytest.mark.parametrize("vdtype", [np.float64, np.float32])
def test_label_and_stats_batch_equals_map_by_map(vdtype):
    """nf_label_and_stats_batch (every pass one launch for the maps of a group) leaves the labels and records of single calls:
    11 maps (two launches' worth) of different foreground, one without any, kept label buffers used twice in a row (the second
    time with other maps, so that stale chunks must be cleared), with and without the S/N range tables."""
    shape = (211, 389)
    eng = _eng(shape)
    rng = np.random.default_rng(5)
    ones = np.ones(shape, dtype=np.uint8)
    rms = np.ones(shape, dtype=np.float32)

    def make(k, seed):
        r = np.random.default_rng(seed)
        f = r.standard_normal(shape).astype(vdtype) * (0.2 if k == 3 else 1.0 + 0.1 * k)
        if k != 3:
            for _ in range(3 + k):
                y, x = int(r.integers(5, shape[0] - 8)), int(r.integers(5, shape[1] - 8))
                f[y:y + 1 + k % 3, x:x + 2 + k % 4] += 7.0 + k
        d, sn, _, _ = eng.finalize(engine.to_dev(f.copy()), engine.to_dev(rms), engine.to_dev(ones), engine.to_dev(ones), None, True)
        return sn, eng.lastSNChunks[1]
    for rnd in range(2):
        maps = [make(k, 100 * rnd + k) for k in range(11 if rnd == 0 else 5)]
        sns = [m[0] for m in maps]
        for negate, thr, useChunks in ((False, 4.0, True), (True, 3.0, False), (False, 2.5, False)):
            chunks = [m[1] for m in maps] if useChunks else None
            labs, recs = eng.label_and_stats_batch(sns, thr, negate=negate, slot='batchtest', cap=3000, snChunks=chunks)
            R = _np(recs)
            assert R.shape == (len(sns), 3001, 6)
            for k, sn in enumerate(sns):
                labA, recA = eng.label_and_stats(sn, thr, negate=negate, cap=3000)
                a = _np(recA)
                n = int(a[0, 0])
                assert n == int(R[k, 0, 0]), (rnd, k)
                if k == 3 and rnd == 0 and thr >= 3.0:
                    assert n == 0
                assert np.array_equal(a[1:n + 1], R[k, 1:n + 1])
                assert np.array_equal(_np(labA), _np(labs[k]))


@pytest.mark.gpu
def test_fused_line_filter_equals_staged_kernel_bit_for_bit():
    """nf_axis_filter runs k_line_filter_rows_fused / _cols_fused (first stage from global memory, last forward stage x transfer
    function x first inverse stage in one step, last stage to global memory) where k_line_filter staged everything through LDS:
    the same arithmetic per element, so the same bits - on every shape of tools/linefilt_ab.py (4096^2, the survey tile shapes,
    odd widths, float32 and float64, with and without the zero reference).  The switch is read once per process: two processes
    per axis, digests compared."""
    import subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "linefilt_ab.py")
    for axis in (0, 1):
        outs = []
        for fused in ("0", "1"):
            env = dict(os.environ, NF_LF_FUSED=fused, NF_LF_FUSED_COLS=fused)
            r = subprocess.run([sys.executable, tool, str(axis), "digest"], env=env, capture_output=True, text=True, timeout=110)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(r.stdout.strip().splitlines())
        assert len(outs[0]) >= 8 and outs[0] == outs[1], (axis, outs)


@pytest.mark.gpu
def test_remove_duplicates_on_the_device_equals_the_host_path():
    """catalogs.removeDuplicates (every rank runs it on the gathered catalogue of all ranks after each pass) takes its partner
    counts and best partners from nf_xmatch_self from GPU_XMATCH_MIN_ROWS rows on: same table, same duplicate count and names as
    the host's sweep - groups of coincident rows, equal SNRs, equal right ascensions, lazy and given names."""
    from nemo_amd import catalogs as C
    engine.require_gpu()
    rng = np.random.default_rng(17)
    for trial, n in enumerate((1024, 3000, 5224)):
        ra, dec = rng.uniform(20, 30, n), rng.uniform(-5, 5, n)
        m = n // 6
        src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
        ra[dst] = ra[src] + rng.choice([0, 1e-3, -2e-3, 0.01, 0.03], m)
        dec[dst] = dec[src] + rng.choice([0, 5e-3, -0.01, 0.02], m)
        tab = C.Table()
        tab['RADeg'], tab['decDeg'], tab['SNR'] = ra, dec, np.round(rng.uniform(4, 8, n), 1)
        tab['x'] = rng.uniform(0, 1, n)
        tab['template'] = np.array(['t%d' % (k % 7) for k in range(n)], dtype='U30')
        if trial % 2 == 0:
            tab.setLazyNames('ACT-CL', False)
        else:
            tab['name'] = np.array(['obj%05d' % k for k in range(n)], dtype='U30')
        assert engine.gpu_ready()
        got = C.removeDuplicates(tab.copy())
        keep = C.GPU_XMATCH_MIN_ROWS
        C.GPU_XMATCH_MIN_ROWS = 1 << 30
        try:
            want = C.removeDuplicates(tab.copy())
        finally:
            C.GPU_XMATCH_MIN_ROWS = keep
        assert got[1] == want[1] > 0 and got[2] == want[2]
        assert got[0].keys() == want[0].keys()
        for k in want[0].keys():
            assert np.array_equal(got[0][k], want[0][k]), (n, k)
