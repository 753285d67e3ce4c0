"""Source-injection catalogue helpers, numpy/scipy statement (TEST INFRASTRUCTURE - only tests/, smoke() and bench.py's
cpu_baseline leg may import this package).

Follows ``catalogs.generateTestCatalog`` (nemo/catalogs.py:849-947), ``catalogs.crossMatch`` (:950-980),
``catalogs.removeCrossMatched`` (:983-1008) and the loop of ``maps.sourceInjectionTest`` (nemo/maps.py:1856-2153).

Pinned: generate_test_catalog against the reference's own function run on a seeded ``np.random`` stream
(tests/golden/injection.npz, made by tests/golden/make_golden_r3.py).  cross_match / remove_cross_matched stand in for
astropy's ``match_coordinates_sky`` (absent offline): nearest neighbour by great-circle separation, brute force -
PARITY UNPINNED against astropy.  Catalogues are lists of dicts; the WCS is duck-typed (``pix2wcs``)."""
import numpy as np
from scipy import ndimage

from .maps import calc_ang_sep_deg


def generate_test_catalog(tileNames, areaMaskDict, WCSDict, numSourcesPerTile, amplitudeColumnName='fixed_y_c',
                          amplitudeRange=(0.001, 1), amplitudeDistribution='linear', avoidanceRadiusArcmin=20.0, maskDilationPix=0):
    """nemo/catalogs.py:886-947, statement for statement (global np.random stream)."""
    RAs, decs, amps, catTileNames = [], [], [], []
    for tileName in tileNames:
        mapData = areaMaskDict[tileName]
        if mapData.sum() == 0:
            continue
        wcs = WCSDict[tileName]
        for i in range(maskDilationPix):
            mapData = 1 - ndimage.binary_dilation(1 - mapData)
        ys, xs = np.where(mapData != 0)
        ys = ys + np.random.uniform(0, 1, len(ys))
        xs = xs + np.random.uniform(0, 1, len(xs))
        indices = np.random.randint(0, len(ys), len(ys))
        ra, dec = wcs.pix2wcs(xs[indices], ys[indices])
        coords = np.array([ra, dec]).transpose()                      # astWCS returns [[ra, dec], ...]
        tileRAs, tileDecs, keepIndices = [], [], []
        for i in indices:
            rai, deci = coords[i, 0], coords[i, 1]
            rDeg = calc_ang_sep_deg(rai, deci, np.array(tileRAs), np.array(tileDecs))
            keepObj = False
            if len(rDeg) > 0:
                if rDeg.min() * 60 > avoidanceRadiusArcmin:
                    keepObj = True
            else:
                keepObj = True
            if keepObj == True:
                tileRAs.append(rai)
                tileDecs.append(deci)
                keepIndices.append(i)
            if len(keepIndices) == numSourcesPerTile:
                break
        if len(coords) == 0:
            continue
        RAs = RAs + coords[keepIndices, 0].tolist()
        decs = decs + coords[keepIndices, 1].tolist()
        if amplitudeDistribution == 'linear':
            amp = np.random.uniform(amplitudeRange[0], amplitudeRange[1], len(keepIndices))
        elif amplitudeDistribution == 'log':
            amp = np.power(10, np.random.uniform(np.log10(amplitudeRange)[0], np.log10(amplitudeRange)[1], len(keepIndices)))
        else:
            raise Exception("Must be either 'linear' or 'log'.")
        amps = amps + amp.tolist()
        catTileNames = catTileNames + [tileName] * len(amp.tolist())
    return [{'name': k + 1, 'RADeg': RAs[k], 'decDeg': decs[k], amplitudeColumnName: amps[k], 'tileName': catTileNames[k]}
            for k in range(len(amps))]


def _nearest(rows1, rows2):
    ra2 = np.array([r['RADeg'] for r in rows2])
    dec2 = np.array([r['decDeg'] for r in rows2])
    idx, sep = [], []
    for r in rows1:
        d = calc_ang_sep_deg(r['RADeg'], r['decDeg'], ra2, dec2)
        k = int(np.argmin(d))
        idx.append(k)
        sep.append(d[k])
    return np.array(idx, dtype=int), np.array(sep)


def cross_match(refRows, matchRows, radiusArcmin=2.5):
    """nemo/catalogs.py:950-980: (ref rows with a match, their nearest match rows, separations in degrees)."""
    idx, rDeg = _nearest(refRows, matchRows)
    mask = rDeg < radiusArcmin / 60.
    return ([r for r, m in zip(refRows, mask) if m], [matchRows[k] for k, m in zip(idx, mask) if m], rDeg[mask])


def remove_cross_matched(refRows, matchRows, radiusArcmin=2.5):
    """nemo/catalogs.py:983-1008."""
    if len(matchRows) == 0:
        return list(refRows)
    idx, rDeg = _nearest(refRows, matchRows)
    return [r for r, d in zip(refRows, rDeg) if d > radiusArcmin / 60.]


def source_injection_loop(models, numIterations, drawCatalog, refilter, realRows, fluxCol, noiseLevelCol, clusterMode=True,
                          radiusArcmin=5.0):
    """The model x iteration loop of maps.sourceInjectionTest (nemo/maps.py:1946-2128).  ``drawCatalog()`` -> mock rows;
    ``refilter(mockRows, model)`` -> recovered rows (the cached-filter, cached-noise pass of pipelines.py:234-250 followed by
    findObjects / measureFluxes).  Returns the rows of the results table, model after model."""
    out = []
    for model in models:
        for it in range(numIterations):
            mock = drawCatalog()
            if 'theta500Arcmin' in model:
                for r in mock:
                    r['theta500Arcmin'] = model['theta500Arcmin']
            if len(mock) == 0:
                continue
            rec = refilter(mock, model)
            if len(rec) > 0:
                rec = remove_cross_matched(rec, realRows, radiusArcmin=radiusArcmin)
            if len(rec) > 0:
                xMock, xRec, rDeg = cross_match(mock, rec, radiusArcmin=radiusArcmin)
                for m, r, d in zip(xMock, xRec, rDeg):
                    row = {'RADeg': r['RADeg'], 'decDeg': r['decDeg'], 'sourceInjectionModel': model['label'], 'SNR': r['SNR'],
                           'rArcmin': d * 60, 'inFlux': m[fluxCol], 'outFlux': r[fluxCol], 'noiseLevel': r[noiseLevelCol],
                           'tileName': r.get('tileName', 'PRIMARY'), 'iteration': it}
                    if 'theta500Arcmin' in model:
                        row['theta500Arcmin'] = model['theta500Arcmin']
                    out.append(row)
    return out
