"""Source finding on the GPU with the reference's call surface (nemo/photometry.py:25-248).

``getObjectPositions``: threshold + 4-connected labelling (scipy.ndimage.label numbering) + per-object
centre of mass / peak position / pixel count, all in libnemofilt kernels.
``findObjects``: ring flagging (second labelling pass at ``ringThresholdSigma``), bicubic-spline S/N at the
sub-pixel positions (windowed not-a-knot spline = RectBivariateSpline(kx=ky=3)), flag lookup, catalogue rows.
Only per-object scalars (a few hundred rows) cross to the host.
"""
import numpy as np

from . import engine, catalogs
from .wcs import calcAngSepDeg


def _galacticLatDeg(RADeg, decDeg):
    """J2000 -> galactic latitude (stands in for astCoords.convertCoords at nemo/photometry.py:118)."""
    ra, dec = np.radians(RADeg), np.radians(decDeg)
    raNGP, decNGP = np.radians(192.85948), np.radians(27.12825)
    return float(np.degrees(np.arcsin(np.sin(dec) * np.sin(decNGP) + np.cos(dec) * np.cos(decNGP) * np.cos(ra - raNGP))))


def _segment(eng, dMap, threshold, findCenterOfMass=True, negate=False):
    """Device part of getObjectPositions; returns (numObjects, labels tensor, stats ndarray [n+1, 6])."""
    labels, numObjects = eng.label(dMap, threshold, negate=negate)
    stats = eng.object_stats(dMap, labels, numObjects, negate=negate)
    return numObjects, labels, stats


def getObjectPositions(mapData, threshold, findCenterOfMass=True):
    """Segmentation map and object positions above ``threshold`` (nemo/photometry.py:213-248).

    Returns (objIDs, objPositions, objNumPix, segmentationMap) - ``objIDs`` includes 0 (background) exactly as
    ``np.unique`` of the reference's segmentation map does when any background exists - or None if no pixel is
    significant.  A device tensor in gives a device segmentation map out.
    """
    if threshold < 0:
        raise Exception("Detection threshold (thresholdSigma in the config file) cannot be negative unless in forced photometry mode.")
    isDev = hasattr(mapData, 'is_cuda')
    d = engine.to_dev(mapData)
    eng = engine.TileEngine.get(d.shape)
    numObjects, labels, stats = _segment(eng, d, threshold, findCenterOfMass)
    if numObjects == 0:
        return None
    st = stats.cpu().numpy()
    counts = st[:, 0]
    hasBackground = int(counts[1:].sum()) < d.numel()
    first = 0 if hasBackground else 1
    objIDs = np.arange(first, numObjects + 1)
    W = d.shape[1]
    positions = []
    for i in objIDs:
        if i == 0:
            # background: centre of mass over label-0 pixels is never used (numPix = 0 < minObjPix); reported as NaN
            positions.append((np.nan, np.nan))
        elif findCenterOfMass:
            positions.append((st[i, 2] / st[i, 1], st[i, 3] / st[i, 1]))
        else:
            idx = int(st[i, 5])
            positions.append((idx // W, idx % W))
    numPix = counts[first:].copy()
    if first == 0:
        numPix[0] = 0.0
    seg = labels if isDev else labels.cpu().numpy()
    return objIDs, positions, numPix, seg


def findObjects(filteredMapDict, threshold=3.0, minObjPix=3, rejectBorder=10, findCenterOfMass=True, flagRings=True,
                ringThresholdSigma=0, ringFlagDistArcmin=12.0, ringFlagMinRingerSNR=20, invertMap=False,
                objIdent='ACT-CL', longNames=False, verbose=True, useInterpolator=True, measureShapes=False,
                DS9RegionsPath=None):
    """Finds objects in the S/N map of ``filteredMapDict`` (nemo/photometry.py:25-210).  Returns a
    :class:`nemo_amd.catalogs.Table` (or [] if nothing is found, as the reference does)."""
    if measureShapes:
        raise NotImplementedError("measureShapes (nemo/photometry.py:133-184) is not part of this build yet")
    if threshold < 0:
        raise Exception("Detection threshold (thresholdSigma in the config file) cannot be negative unless in forced photometry mode.")
    wcs = filteredMapDict['wcs']
    flagMask = filteredMapDict['flagMask']
    dSN = engine.to_dev(filteredMapDict['SNMap'])
    eng = engine.TileEngine.get(dSN.shape)
    H, W = dSN.shape

    numObjects, labels, stats = _segment(eng, dSN, threshold, findCenterOfMass, negate=invertMap)   # :59
    if numObjects == 0:
        return []
    st = stats.cpu().numpy()

    ringLabels = isRing = None
    if flagRings == True:                                                         # :65-75
        numRing, ringLabels, ringStats = _segment(eng, dSN, ringThresholdSigma, True, negate=invertMap)
        isRing = eng.ring_flags(ringLabels, ringStats, numRing, minRingPix=30)

    # objects passing the size cut, in label order (label 0 has numPix 0 and never passes unless minObjPix <= 0)
    ids = np.arange(1, numObjects + 1)
    sel = ids[st[1:, 0] >= minObjPix]
    if len(sel) == 0:
        return []
    if findCenterOfMass:
        ys = st[sel, 2] / st[sel, 1]
        xs = st[sel, 3] / st[sel, 1]
    else:
        idx = st[sel, 5].astype(np.int64)
        ys = (idx // W).astype(np.float64)
        xs = (idx % W).astype(np.float64)

    if useInterpolator == True:                                                   # :81-84,128
        SNRs = eng.spline_eval(dSN, np.stack([ys, xs], axis=1), negate=invertMap)
    else:
        t = engine.torch()
        iy = t.as_tensor(np.round(ys).astype(np.int64), device=dSN.device)
        ix = t.as_tensor(np.round(xs).astype(np.int64), device=dSN.device)
        SNRs = dSN[iy, ix].cpu().numpy().astype(np.float64) * (-1.0 if invertMap else 1.0)

    ringFlags = np.zeros(len(sel), dtype=bool)
    if ringLabels is not None:                                                    # :112-114
        t = engine.torch()
        iy = t.as_tensor(ys.astype(np.int64), device=dSN.device)                  # int() truncation
        ix = t.as_tensor(xs.astype(np.int64), device=dSN.device)
        lab = ringLabels[iy, ix].to(t.int64)
        ringFlags = isRing[lab].cpu().numpy() > 0

    fm = flagMask.cpu().numpy() if hasattr(flagMask, 'is_cuda') else np.asarray(flagMask)
    catalog = []
    for k in range(len(sel)):
        objDict = {}
        objDict['id'] = k + 1                                                     # :108,188
        objDict['x'] = xs[k]
        objDict['y'] = ys[k]
        objDict['ringFlag'] = bool(ringFlags[k])
        objDict['RADeg'], objDict['decDeg'] = wcs.pix2wcs(objDict['x'], objDict['y'])
        if objDict['RADeg'] < 0:
            objDict['RADeg'] = 360.0 + objDict['RADeg']
        objDict['galacticLatDeg'] = _galacticLatDeg(objDict['RADeg'], objDict['decDeg'])
        if longNames == False:
            objDict['name'] = catalogs.makeName(objDict['RADeg'], objDict['decDeg'], prefix=objIdent)
        else:
            objDict['name'] = catalogs.makeLongName(objDict['RADeg'], objDict['decDeg'], prefix=objIdent)
        objDict['numSigPix'] = st[sel[k], 0]
        objDict['template'] = filteredMapDict['label']
        objDict['tileName'] = filteredMapDict['tileName']
        objDict['SNR'] = SNRs[k]
        objDict['flags'] = fm[int(round(objDict['y'])), int(round(objDict['x']))]
        if objDict['SNR'] > threshold:                                            # :186
            catalog.append(objDict)

    if len(catalog) > 0:                                                          # :191-208
        catalog = catalogs.catalogListToTab(catalog)
        highSNR = catalog[catalog['SNR'] > ringFlagMinRingerSNR]
        for row in catalog:
            if row['ringFlag'] == True:
                rDeg = calcAngSepDeg(row['RADeg'], row['decDeg'], highSNR['RADeg'], highSNR['decDeg'])
                if np.logical_and((rDeg * 60) < ringFlagDistArcmin, rDeg > 0).sum() == 0:
                    row['ringFlag'] = False
        catalog['flags'] = catalog['flags'] + catalog['ringFlag']
    return catalog
