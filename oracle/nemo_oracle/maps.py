"""Catalogue-driven map preprocessing, numpy/scipy statement (TEST INFRASTRUCTURE).

Follows ``maps.MapDict.preprocess`` (nemo/maps.py:325-465), ``maps.makeDegreesDistanceMap`` (:2479-2536),
``maps.makeModelImage`` (:1658-1853) and ``catalogs.getCatalogWithinImage`` (nemo/catalogs.py:1039-1086).

Catalogues are lists of dicts here.  The WCS object is duck-typed (``wcs2pix``, ``pix2wcs``, ``getPixelSizeDeg``,
``getXPixelSizeDeg``).  Radial tables and haversine geometry are handed in by the caller (they are host-side numpy in
the product too); the painter is ``painter.paint_radial`` - PARITY UNPINNED against pixell.pointsrcs.sim_objects, which
is absent offline.  Everything else is PINNED by the reference's own code: ``MapDict.preprocess``'s catalogue statements and
``makeModelImage`` were run unmodified under the stub importer (tests/golden/make_golden_r3.py, painter and WCS served from
outside) and this module is held to their outputs bit for bit (tests/golden/preprocess.npz, 12 + 6 cases;
tests/test_cpu_reference_r3.py).
"""
import numpy as np
from scipy import ndimage

from . import painter, pixell_restated


def calc_ang_sep_deg(ra1, dec1, ra2, dec2):
    """astCoords.calcAngSepDeg as used at nemo/maps.py:2509-2510 (great-circle separation, degrees)."""
    ra1, dec1, ra2, dec2 = np.radians(ra1), np.radians(dec1), np.radians(ra2), np.radians(dec2)
    h = np.sin((dec2 - dec1) / 2.0) ** 2 + np.cos(dec1) * np.cos(dec2) * np.sin((ra2 - ra1) / 2.0) ** 2
    return np.degrees(2.0 * np.arcsin(np.sqrt(np.clip(h, 0.0, 1.0))))


def catalog_within_image(rows, shape, wcs):
    """nemo/catalogs.py:1039-1086 (no mask): rows with 0 <= x < W-1 and 0 <= y < H-1."""
    out = []
    for row in rows:
        x, y = wcs.wcs2pix(row['RADeg'], row['decDeg'])
        if x >= 0 and x < shape[1] - 1 and y >= 0 and y < shape[0] - 1:
            out.append(row)
    return out


def degrees_distance_map(degreesMap, wcs, RADeg, decDeg, maxDistDegrees):
    """nemo/maps.py:2506-2536, statement for statement (fills in place, returns the map and the bounds)."""
    x0, y0 = wcs.wcs2pix(RADeg, decDeg)
    ra1, dec1 = wcs.pix2wcs(x0 + 1, y0 + 1)
    xPixScale = calc_ang_sep_deg(RADeg, decDeg, ra1, decDeg)
    yPixScale = calc_ang_sep_deg(RADeg, decDeg, RADeg, dec1)
    xDistPix = int(round(maxDistDegrees / xPixScale))
    yDistPix = int(round(maxDistDegrees / yPixScale))
    Y, X = degreesMap.shape
    minX = max(int(round(x0)) - xDistPix, 0)
    maxX = min(int(round(x0)) + xDistPix, X)
    minY = max(int(round(y0)) - yDistPix, 0)
    maxY = min(int(round(y0)) + yDistPix, Y)
    xDeg = (np.arange(X) - x0) * xPixScale
    yDeg = (np.arange(Y) - y0) * yPixScale
    for i in range(minY, maxY):
        degreesMap[i][minX:maxX] = np.sqrt(yDeg[i] ** 2 + xDeg[minX:maxX] ** 2)
    return degreesMap, [minX, maxX], [minY, maxY]


def object_window(shape, wcs, RADeg, decDeg, maxSizeDeg):
    """Pixel box of the RA/dec clip of signals._paintSignalMap (nemo/signals.py:675-680; astImages.clipUsingRADecCoords is
    third-party: corners to pixels, rounded, clipped to the map)."""
    xa, ya = wcs.wcs2pix(RADeg + maxSizeDeg / np.cos(np.radians(decDeg)), decDeg - maxSizeDeg)
    xb, yb = wcs.wcs2pix(RADeg - maxSizeDeg / np.cos(np.radians(decDeg)), decDeg + maxSizeDeg)
    x0, x1 = sorted([int(round(xa)), int(round(xb))])
    y0, y1 = sorted([int(round(ya)), int(round(yb))])
    x0, y0 = min(max(x0, 0), shape[1]), min(max(y0, 0), shape[0])
    x1, y1 = min(max(x1, x0), shape[1]), min(max(y1, y0), shape[0])
    return x0, x1, y0, y1


def paint_objects(shape, wcs, objects, geometry):
    """The painting loop of makeModelImage (:1790-1840): ``objects`` = [(RADeg, decDeg, table, maxSizeDeg, factors)], each
    painted inside its window and added to the float32 map in order; ``factors`` = (fSZ, TCMB*1e6) converts the clip as
    maps.convertToDeltaT does (float32 * float64 scalars -> float64, nemo/signals.py:684-688) before the float32 store."""
    omap = np.zeros(shape, dtype=np.float32)
    for RADeg, decDeg, table, maxSizeDeg, factors in objects:
        x0, x1, y0, y1 = object_window(shape, wcs, RADeg, decDeg, maxSizeDeg)
        if x1 <= x0 or y1 <= y0:
            continue
        A, B, C = geometry(shape, wcs, RADeg, decDeg)
        clip = painter.paint_radial((y1 - y0, x1 - x0), A[y0:y1], B[y0:y1], C[x0:x1], table.r, table.prof, table.amp, table.sign)
        if factors is not None:
            clip = clip * np.float64(factors[0]) * np.float64(factors[1])
        omap[y0:y1, x0:x1] = omap[y0:y1, x0:x1] + clip
    return omap


def make_model_image(shape, wcs, rows, objectsFor, geometry, validAreaSection=None, minSNR=-99, wholeMapFactors=None,
                     applyPixelWindow=True):
    """nemo/maps.py:1710-1853.  ``objectsFor(rows)`` gives the painter's per-object tuples (1-D profile work);
    ``wholeMapFactors`` is the override branch's conversion of the summed map (:1786-1788)."""
    rows = catalog_within_image(rows, shape, wcs)
    if len(rows) > 0:
        key = 'SNR' if 'SNR' in rows[0] else ('fixed_SNR' if 'fixed_SNR' in rows[0] else None)
        if key is not None:
            rows = [r for r in rows if r[key] > minSNR]
    if validAreaSection is not None and len(rows) > 0:
        x0, x1, y0, y1 = validAreaSection
        keep = []
        for r in rows:
            x, y = wcs.wcs2pix(r['RADeg'], r['decDeg'])
            if x >= x0 and x < x1 and y >= y0 and y < y1:
                keep.append(r)
        rows = keep
    if len(rows) == 0:
        return None
    modelMap = paint_objects(shape, wcs, objectsFor(rows), geometry)
    if wholeMapFactors is not None:
        modelMap = modelMap * np.float64(wholeMapFactors[0]) * np.float64(wholeMapFactors[1])
    if applyPixelWindow:
        modelMap = pixell_restated.apply_window(modelMap, pow=1.0)
    return modelMap


def catalogue_steps(md, data, weights, wcs, surveyMask, psMask, flagMask, modelImage, injectValidArea=None):
    """nemo/maps.py:325-465.  ``md``: the map dictionary's keys; ``modelImage(rows, **kw)`` -> make_model_image bound to the
    map and the caller's table factory.  Returns data, surveyMask, psMask, flagMask."""
    shape = data.shape
    if md.get('injectSources') is not None:
        inj = md['injectSources']
        withFreq = 'GNFWParams' in inj
        modelMap = modelImage(inj['catalog'], cluster=True, obsFreqGHz=md['obsFreqGHz'] if withFreq else None,
                              GNFWParams=inj['GNFWParams'] if withFreq else None, profile=inj['profile'],
                              override=inj['override'], validAreaSection=injectValidArea)
        if modelMap is not None:
            modelMap[weights == 0] = 0
            data = data + modelMap
    holeFilling = any(h in md for h in ['maskPointSourcesFromCatalog', 'maskAndFillFromCatalog', 'extendedMask'])
    if holeFilling:
        pixRad = (10.0 / 60.0) / wcs.getPixelSizeDeg()
        bckData = ndimage.median_filter(data, int(pixRad))
    if md.get('maskPointSourcesFromCatalog') is not None:
        entries = md['maskPointSourcesFromCatalog']
        if type(entries) is not list:
            entries = [entries]
        psMask = np.ones(shape, dtype=np.uint8)
        rDegMap = np.ones(shape, dtype=float) * 1e6
        for info in entries:
            fluxCutJy = 0.0
            rows = info
            if type(info) == dict:
                rows, fluxCutJy = info['path'], info['fluxCutJy']
            if len(rows) > 0 and 'fluxJy' in rows[0]:
                rows = [r for r in rows if r['fluxJy'] > fluxCutJy]
            rows = catalog_within_image(rows, shape, wcs)
            for row in rows:
                if 'rArcmin' in row:
                    maskRadiusArcmin = row['rArcmin']
                elif 'ellipse_A' in row:
                    xPixSizeArcmin = (wcs.getXPixelSizeDeg() / np.cos(np.radians(row['decDeg']))) * 60
                    maskRadiusArcmin = (row['ellipse_A'] / xPixSizeArcmin) / 2
                else:
                    raise Exception("To mask sources in a catalog, need either 'rArcmin' or 'ellipse_A' column to be present.")
                rDegMap, xb, yb = degrees_distance_map(rDegMap, wcs, row['RADeg'], row['decDeg'], maskRadiusArcmin / 60)
                surveyMask[rDegMap < maskRadiusArcmin / 60.0] = 0
                psMask[rDegMap < maskRadiusArcmin / 60.0] = 0
                data[rDegMap < maskRadiusArcmin / 60.0] = bckData[rDegMap < maskRadiusArcmin / 60.0]
    if md.get('subtractModelFromCatalog') is not None:
        tabs = md['subtractModelFromCatalog']
        if type(tabs) is not list:
            tabs = [tabs]
        for rows in tabs:
            rows = catalog_within_image(rows, shape, wcs)
            model = modelImage(rows, cluster=len(rows) > 0 and 'y_c' in rows[0], obsFreqGHz=md['obsFreqGHz'])
            if model is not None:
                data = data - model
                flagMask = flagMask + np.greater(model, 1)
    if md.get('maskAndFillFromCatalog') is not None:
        tabs = md['maskAndFillFromCatalog']
        if type(tabs) is not list:
            tabs = [tabs]
        for rows in tabs:
            rows = catalog_within_image(rows, shape, wcs)
            if len(rows) > 0 and 'ellipse_A' not in rows[0] and 'maskHoleRadiusArcmin' not in md:
                raise Exception("Need to set measureShapes: True or set maskHoleRadiusArcmin to use maskAndFillFromCatalog")
            for row in rows:
                rArcminMap = np.ones(shape, dtype=float) * 1e6
                if 'ellipse_B' in row:
                    xPixSizeArcmin = (wcs.getXPixelSizeDeg() / np.cos(np.radians(row['decDeg']))) * 60
                    maskRadiusArcmin = (row['ellipse_A'] / xPixSizeArcmin) / 2
                if md.get('maskHoleRadiusArcmin') is not None:
                    maskRadiusArcmin = md['maskHoleRadiusArcmin']
                if md.get('maskHoleDilationFactor') is not None:
                    maskRadiusArcmin = maskRadiusArcmin * md['maskHoleDilationFactor']
                rArcminMap, xb, yb = degrees_distance_map(rArcminMap, wcs, row['RADeg'], row['decDeg'], maskRadiusArcmin / 60)
                rArcminMap = rArcminMap * 60
                surveyMask[rArcminMap < maskRadiusArcmin] = 0
                psMask[rArcminMap < maskRadiusArcmin] = 0
                data[rArcminMap < maskRadiusArcmin] = bckData[rArcminMap < maskRadiusArcmin]
    return data, surveyMask, psMask, flagMask
