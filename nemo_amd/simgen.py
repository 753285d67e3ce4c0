"""Seeded synthetic tiles (CMB + white noise + painted sources), SURVEY.md 8d recipes.

Plays the role nemoModel plays for the reference's tests (tests/lib/NemoTests.py:190-194): a flat-sky Gaussian
CMB realisation with the Planck TT spectrum shipped with the reference (nemo/data/planck_lensedCls.dat, same
table as nemo/filters.py:270) times the beam, white noise following a (varying) inverse-variance map, and
beam-convolved sources painted with the same radial painter the filters use.  numpy only; the painter is
passed in (GPU painter in bench / GPU tests, the oracle's numpy painter in CPU tests), so this module never
imports the oracle.
"""
import os

import numpy as np

from . import signals
from .wcs import CARWCS

_CL_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "planck_lensedCls.dat")


def cmbPower(lmap):
    tab = np.loadtxt(_CL_PATH)
    L, TT = tab[:, 0], tab[:, 1]
    Cl = (TT * 2 * np.pi) / (L * (L + 1))
    return np.interp(lmap, L, Cl, left=0.0, right=0.0)


def makeSimTile(shape, freqsGHz, beams, noiseUK, seed=1234, numSources=50, kind='clusters', pixDeg=0.5 / 60.0,
                dtype=np.float64, painter=None, varyWeights=True, RADeg=10.0, decDeg=0.0, clusterParams=None,
                borderPix=60):
    """Returns (list of dicts with keys mapData, weightsData, obsFreqGHz, units, beamFileName(BeamProfile), label),
    wcs, truth (list of dicts RADeg, decDeg, amplitude)).  ``painter(geom, table) -> float32 [H, W]``."""
    H, W = shape
    wcs = CARWCS.make(shape, pixDeg=pixDeg, RADeg=RADeg, decDeg=decDeg)
    rng = np.random.default_rng(seed)
    pixRad = np.radians(pixDeg)
    ly = np.fft.fftfreq(H, d=pixRad) * 2 * np.pi
    lx = np.fft.fftfreq(W, d=pixRad * np.cos(np.radians(decDeg))) * 2 * np.pi
    lmap = np.sqrt(ly[:, None] ** 2 + lx[None, :] ** 2)
    Cl = cmbPower(lmap)
    white = np.fft.fft2(rng.standard_normal((H, W)))
    pixArea = pixRad * pixRad * np.cos(np.radians(decDeg))

    # source positions / amplitudes (common to all frequencies)
    truth = []
    for k in range(numSources):
        y = rng.uniform(borderPix, H - borderPix)
        x = rng.uniform(borderPix, W - borderPix)
        ra, dec = wcs.pix2wcs(x, y)
        if kind == 'clusters':
            amp = rng.uniform(0.5e-4, 5e-4)       # y0
        else:
            amp = float(np.exp(rng.normal(np.log(600.0), 1.1)))   # uK, as catalogs.generateRandomSourcesCatalog
        truth.append({'RADeg': ra, 'decDeg': dec, 'amplitude': amp, 'x': x, 'y': y})

    if varyWeights:
        yy, xx = np.mgrid[0:H, 0:W]
        mod = 1.0 + 3.0 * (xx / float(W)) * (yy / float(H))
        mod[:, W // 3:W // 2] *= 1.5
    else:
        mod = np.ones((H, W))

    out = []
    for i, (freq, beam, sigma) in enumerate(zip(freqsGHz, beams, noiseUK)):
        Bl = np.interp(lmap, beam.ell, beam.Bell, left=1.0, right=0.0)
        cmb = np.real(np.fft.ifft2(white * np.sqrt(Cl / pixArea) * Bl))
        ivar = mod / (sigma * sigma)
        noise = rng.standard_normal((H, W)) / np.sqrt(ivar)
        m = cmb + noise
        if painter is not None and numSources > 0:
            for src in truth:
                geom = signals.paintGeometry(shape, wcs, src['RADeg'], src['decDeg'])
                if kind == 'clusters':
                    params = clusterParams or {'z': 0.4, 'M500MSun': 2.0e14}
                    amp = signals.convertToDeltaT(src['amplitude'], freq)
                    table = signals.templateTables('arnaud', beam, params, amplitude=amp)
                else:
                    table = signals.templateTables('beam', beam, {}, amplitude=src['amplitude'])
                m = m + np.asarray(painter(geom, table), dtype=np.float64)
        out.append({'mapData': m.astype(dtype), 'weightsData': ivar.astype(dtype), 'obsFreqGHz': freq, 'units': 'uK',
                    'beamFileName': beam, 'label': 'f%03d' % (int(round(freq))), 'wcs': wcs})
    return out, wcs, truth
