"""Shared builders for the parity tests: one seeded synthetic tile -> (oracle inputs, product inputs)."""
import os

import numpy as np

from nemo_amd import signals, simgen, maps as pmaps
from nemo_amd.wcs import calcAngSepDeg
from nemo_oracle import painter as opainter, filters as ofilt, photometry as ophot

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ACT_BEAM = os.path.join(GOLDEN, "data", "profile_AR1_2009_pixwin_130224.txt")


def cpu_paint(geom, table):
    shape = (geom[0].shape[0], geom[2].shape[0])
    return opainter.paint_radial(shape, geom[0], geom[1], geom[2], table.r, table.prof, table.amp, table.sign)


def make_tile(shape, freqs=(150.0,), fwhm=(1.4,), noise=(20.0,), seed=1, numSources=5, kind='clusters', dtype=np.float64,
              varyWeights=True, holes=False):
    beams = [signals.BeamProfile(FWHMArcmin=f) for f in fwhm]
    mapsL, wcs, truth = simgen.makeSimTile(shape, list(freqs), beams, list(noise), seed=seed, numSources=numSources,
                                           kind=kind, painter=cpu_paint, dtype=dtype, varyWeights=varyWeights)
    surveyMask = np.ones(shape, dtype=np.uint8)
    psMask = np.ones(shape, dtype=np.uint8)
    if holes:
        rng = np.random.default_rng(seed + 100)
        for _ in range(3):
            y, x = rng.integers(30, shape[0] - 30), rng.integers(30, shape[1] - 30)
            psMask[y - 4:y + 5, x - 4:x + 5] = 0
        surveyMask[:, :25] = 0
        for m in mapsL:
            m['weightsData'][:, :12] = 0
            m['mapData'][:, :12] = 0
    for m in mapsL:
        m['surveyMask'] = surveyMask
        m['pointSourceMask'] = psMask
    return mapsL, wcs, truth


def oracle_matched(mapsL, wcs, params, kind, faithfulLoop=False):
    """Run the oracle MatchedFilter on a tile; returns (result dict, oracle filter object)."""
    shape = mapsL[0]['mapData'].shape
    RADeg, decDeg = wcs.getCentreWCSCoords()
    geom = signals.paintGeometry(shape, wcs, RADeg, decDeg)

    def templateFn(i, amp):
        return cpu_paint(geom, signals.templateTables(kind, mapsL[i]['beamFileName'], params, amplitude=amp))

    om = []
    for m in mapsL:
        data = m['mapData'].copy()
        data[m['weightsData'] == 0] = 0
        d = {'data': data, 'weights': m['weightsData'], 'obsFreqGHz': m['obsFreqGHz'], 'units': m['units'],
             'surveyMask': m['surveyMask'], 'pointSourceMask': m['pointSourceMask']}
        om.append(d)
    cx, cy = wcs.wcs2pix(RADeg, decDeg)
    xps, yps = pmaps.pixelScalesDeg(wcs)
    mf = ofilt.MatchedFilter(om, params, {'pixSizeDeg': wcs.getPixelSizeDeg(), 'centreYX': (cy, cx),
                                          'xPixScaleDeg': xps, 'yPixScaleDeg': yps}, templateFn,
                             fSZ=signals.fSZ, deltaT0=lambda f: signals.convertToDeltaT(2e-4, f))
    res = mf.buildAndApply(faithfulLoop=faithfulLoop)
    return res, mf


def product_maps(mapsL):
    return [pmaps.MapDict(dict(m)) for m in mapsL]


def oracle_find(SNMap, flagMask, wcs, **kw):
    return ophot.find_objects(SNMap, flagMask, wcs.pix2wcs, angSepDeg=calcAngSepDeg, **kw)
