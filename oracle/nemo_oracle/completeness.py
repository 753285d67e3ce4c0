"""Noise / area tables, numpy statement (TEST INFRASTRUCTURE): completeness.getRMSTab (nemo/completeness.py:1317-1336) and
maps.getPixelAreaArcmin2Map (nemo/maps.py:1385-1406), statement for statement on arrays handed in by the caller."""
import numpy as np

from .maps import calc_ang_sep_deg


def pixel_area_arcmin2_map(shape, wcs):
    """nemo/maps.py:1385-1406."""
    pixAreasDeg2 = []
    RACentre, decCentre = wcs.getCentreWCSCoords()
    x0, y0 = wcs.wcs2pix(RACentre, decCentre)
    x1 = x0 + 1
    for y0 in range(shape[0]):
        y1 = y0 + 1
        ra0, dec0 = wcs.pix2wcs(x0, y0)
        ra1, dec1 = wcs.pix2wcs(x1, y1)
        xPixScale = calc_ang_sep_deg(ra0, dec0, ra1, dec0)
        yPixScale = calc_ang_sep_deg(ra0, dec0, ra0, dec1)
        pixAreasDeg2.append(xPixScale * yPixScale)
    pixAreasDeg2 = np.array(pixAreasDeg2, dtype=np.float32)
    pixAreasArcmin2 = pixAreasDeg2 * (60 ** 2)
    return np.array([pixAreasArcmin2] * shape[1]).transpose()


def get_rms_tab(RMSMap, areaMap, wcs, flagMask=None, maxFlags=None, intersectMask=None):
    """nemo/completeness.py:1317-1342 -> (areaDeg2, y0RMS)."""
    areaMap = np.array(areaMap)
    if maxFlags is not None:
        areaMap[flagMask > maxFlags] = 0
    areaMapSqDeg = (pixel_area_arcmin2_map(areaMap.shape, wcs) * areaMap) / (60.0 ** 2)
    if intersectMask is not None:
        areaMapSqDeg = areaMapSqDeg * intersectMask
        RMSMap = RMSMap * intersectMask
    RMSValues = np.unique(RMSMap[np.nonzero(RMSMap)])
    tileArea = np.zeros(len(RMSValues))
    for i in range(len(RMSValues)):
        tileArea[i] = areaMapSqDeg[np.equal(RMSMap, RMSValues[i])].sum()
    tol = 0.003
    if abs(tileArea.sum() - areaMapSqDeg.sum()) > tol:
        raise Exception("Mismatch between area map and area in RMSTab")
    if np.less(tileArea, 0).sum() > 0:
        raise Exception("Negative area in tile")
    return tileArea, RMSValues
