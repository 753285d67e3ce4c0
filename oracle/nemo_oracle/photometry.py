"""Oracle for the detection half of the path (TEST INFRASTRUCTURE - see package docstring).

Restates nemo/photometry.py:25-248 (``findObjects``, ``getObjectPositions``) with scipy.ndimage,
the same library the reference calls, so label numbering, centre-of-mass arithmetic and the
whole-map spline are pinned by scipy itself.  WCS work is delegated to a caller-supplied
``pix2wcs(x, y)``; naming / galactic latitude (astLib, absent offline) are left to the caller.
"""
import numpy as np
from scipy import ndimage, interpolate


def get_object_positions(mapData, threshold, findCenterOfMass=True):
    """photometry.getObjectPositions (nemo/photometry.py:213-248)."""
    if threshold < 0:
        raise Exception("Detection threshold (thresholdSigma in the config file) cannot be negative "
                        "unless in forced photometry mode.")
    sigPix = np.array(np.greater(mapData, threshold), dtype=int)
    if (sigPix != 0).sum() == 0:
        return None
    sigPixMask = np.equal(sigPix, 1)
    segMap, numObjects = ndimage.label(sigPix)
    objIDs = np.unique(segMap)
    if findCenterOfMass:
        positions = ndimage.center_of_mass(mapData, labels=segMap, index=objIDs)
    else:
        positions = ndimage.maximum_position(mapData, labels=segMap, index=objIDs)
    numPix = ndimage.sum(sigPixMask, labels=segMap, index=objIDs)
    return objIDs, positions, numPix, segMap


def ring_mask(data, ringThresholdSigma, minRingPix=30):
    """Ring detection (nemo/photometry.py:65-75): components above ``ringThresholdSigma`` with more than
    ``minRingPix`` pixels whose centre of mass does not fall on one of their own pixels."""
    ringIDs, ringPositions, ringNumPix, ringSegMap = get_object_positions(data, ringThresholdSigma,
                                                                         findCenterOfMass=True)
    for i in range(len(ringIDs)):
        if type(ringNumPix) != float and ringNumPix[i] > minRingPix:
            y, x = ringPositions[i]
            if ringSegMap[int(y), int(x)] != ringIDs[i]:
                sel = np.equal(ringSegMap, ringIDs[i])
                ringSegMap[sel] = -1 * ringSegMap[sel]
    return np.array(np.less(ringSegMap, 0), dtype=int)


_NO_SHAPE = {'ellipse_PA': -99, 'ellipse_A': -99, 'ellipse_B': -99, 'ellipse_x0': -99, 'ellipse_y0': -99, 'ellipse_e': -99}


def measure_shape(data, segMap, objID, numSigPix):
    """SExtractor-style moments of one segment (nemo/photometry.py:133-184), statement for statement (``data[mask].sum()`` keeps
    the map's dtype, as it does there)."""
    doubleCheck = False
    out = {}
    if numSigPix > 9:
        mask = np.equal(segMap, objID)
        ys, xs = np.where(segMap == objID)
        yMin, xMin = ys.min(), xs.min()
        xs = xs - xMin
        ys = ys - yMin
        cx2 = (xs * data[mask]).sum() / data[mask].sum()
        cy2 = (ys * data[mask]).sum() / data[mask].sum()
        x2 = (((xs ** 2) * data[mask]).sum() / data[mask].sum()) - cx2 ** 2
        y2 = (((ys ** 2) * data[mask]).sum() / data[mask].sum()) - cy2 ** 2
        xy = (((xs * ys) * data[mask]).sum() / data[mask].sum()) - cx2 * cy2
        theta = np.degrees(np.arctan(2 * (xy / (x2 - y2))) / 2.0)
        if xy > 0 and theta < 0:
            theta = theta + 90
        elif xy < 0 and theta > 0:
            theta = theta - 90
        if theta > 0 and xy > 0:
            doubleCheck = True
        if theta < 0 and xy < 0:
            doubleCheck = True
        if doubleCheck == True:
            A = np.sqrt((x2 + y2) / 2.0 + np.sqrt(((x2 - y2) / 2) ** 2 + xy ** 2))
            B = np.sqrt((x2 + y2) / 2.0 - np.sqrt(((x2 - y2) / 2) ** 2 + xy ** 2))
            segArea = float(np.count_nonzero(np.equal(segMap, objID)))
            curArea = A * B * np.pi
            scaleFactor = np.sqrt(segArea / curArea)
            A = A * scaleFactor
            B = B * scaleFactor
            ecc = np.sqrt(1 - B ** 2 / A ** 2)
            out = {'ellipse_PA': theta, 'ellipse_A': A, 'ellipse_B': B, 'ellipse_x0': cx2 + xMin, 'ellipse_y0': cy2 + yMin,
                   'ellipse_e': ecc}
    if numSigPix <= 9 or doubleCheck == False:
        out = dict(_NO_SHAPE)
    return out


def find_objects(SNMap, flagMask, pix2wcs, threshold=3.0, minObjPix=3, findCenterOfMass=True, flagRings=True,
                 ringThresholdSigma=0, ringFlagDistArcmin=12.0, ringFlagMinRingerSNR=20, invertMap=False,
                 useInterpolator=True, angSepDeg=None, measureShapes=False):
    """photometry.findObjects (nemo/photometry.py:25-210) up to the catalog rows (list of dicts with keys
    id, x, y, RADeg, decDeg, numSigPix, SNR, flags, ringFlag).  Returns [] when nothing is significant."""
    data = SNMap
    if invertMap:
        data = data * -1
    stuff = get_object_positions(data, threshold, findCenterOfMass=findCenterOfMass)
    if stuff is None:
        return []
    objIDs, objPositions, objNumPix, segMap = stuff
    ringMask = ring_mask(data, ringThresholdSigma) if flagRings else None
    if useInterpolator:
        spline = interpolate.RectBivariateSpline(np.arange(data.shape[0]), np.arange(data.shape[1]),
                                                 data, kx=3, ky=3)
    catalog = []
    idNumCount = 1
    for i in range(len(objIDs)):
        if type(objNumPix) != float and objNumPix[i] >= minObjPix:                # :106 (label 0 included)
            obj = {'id': idNumCount, 'x': objPositions[i][1], 'y': objPositions[i][0], 'ringFlag': False}
            if ringMask is not None and ringMask[int(obj['y']), int(obj['x'])] > 0:
                obj['ringFlag'] = True
            obj['RADeg'], obj['decDeg'] = pix2wcs(obj['x'], obj['y'])
            if obj['RADeg'] < 0:
                obj['RADeg'] = 360.0 + obj['RADeg']
            obj['numSigPix'] = objNumPix[i]
            if useInterpolator:
                obj['SNR'] = spline(obj['y'], obj['x'])[0][0]
            else:
                obj['SNR'] = data[int(round(obj['y'])), int(round(obj['x']))]
            obj['flags'] = flagMask[int(round(obj['y'])), int(round(obj['x']))]
            if measureShapes == True:                                             # :133-184
                with np.errstate(divide='ignore', invalid='ignore'):
                    obj.update(measure_shape(data, segMap, objIDs[i], obj['numSigPix']))
            if obj['SNR'] > threshold:                                            # :186
                catalog.append(obj)
            idNumCount = idNumCount + 1                                           # :188
    if len(catalog) > 0 and angSepDeg is not None:                                # :196-208
        ra = np.array([o['RADeg'] for o in catalog])
        dec = np.array([o['decDeg'] for o in catalog])
        snr = np.array([o['SNR'] for o in catalog])
        high = snr > ringFlagMinRingerSNR
        for o in catalog:
            if o['ringFlag']:
                rDeg = angSepDeg(o['RADeg'], o['decDeg'], ra[high], dec[high])
                if np.logical_and((rDeg * 60) < ringFlagDistArcmin, rDeg > 0).sum() == 0:
                    o['ringFlag'] = False
        for o in catalog:
            o['flags'] = o['flags'] + o['ringFlag']
    return catalog


def _whole_map_spline(data):
    return interpolate.RectBivariateSpline(np.arange(data.shape[0]), np.arange(data.shape[1]), data, kx=3, ky=3)


def get_snr_values(catalog, SNMap, wcs2pix, useInterpolator=True, invertMap=False, prefix=''):
    """photometry.getSNRValues (nemo/photometry.py:251-281) on a list of dict rows, in place."""
    if invertMap:
        SNMap = SNMap * -1
    if useInterpolator:
        spline = _whole_map_spline(SNMap)
    for obj in catalog:
        obj[prefix + 'SNR'] = 0.0
    for obj in catalog:
        x, y = wcs2pix(obj['RADeg'], obj['decDeg'])
        if int(x) > 0 and int(x) < SNMap.shape[1] and int(y) > 0 and int(y) < SNMap.shape[0]:
            if useInterpolator:
                obj[prefix + 'SNR'] = spline(y, x)[0][0]
            else:
                obj[prefix + 'SNR'] = SNMap[int(round(y)), int(round(x))]


def measure_fluxes(catalog, data, wcs2pix, mapUnits, convertToDeltaT, deltaTToJyPerSr=None, photSNMap=None, photData=None,
                   useInterpolator=True, ycObsFreqGHz=148.0, beamSolidAngle_nsr=None, obsFreqGHz=None):
    """photometry.measureFluxes (nemo/photometry.py:284-377) on a list of dict rows, in place.  ``mapUnits`` is the
    BUNIT of the filtered map, ``beamSolidAngle_nsr`` / ``obsFreqGHz`` its BEAMNSR / FREQGHZ cards (None: absent);
    unit conversions are supplied by the caller (nemo/maps.py:679, nemo/photometry.py:488)."""
    if photSNMap is not None:
        get_snr_values(catalog, photSNMap, wcs2pix, prefix='fixed_', useInterpolator=useInterpolator)
    reportJyFluxes = beamSolidAngle_nsr is not None
    mapDataList, prefixList = [data], ['']
    if photData is not None:
        mapDataList.append(photData)
        prefixList.append('fixed_')
    splines = [_whole_map_spline(d) if useInterpolator else None for d in mapDataList]
    keysToAdd = ['deltaT_c', 'err_deltaT_c']
    if mapUnits == 'yc':
        keysToAdd = keysToAdd + ['y_c', 'err_y_c']
    elif mapUnits == 'uK':
        keysToAdd = keysToAdd + ['fluxJy', 'err_fluxJy']
    for obj in catalog:
        for prefix in prefixList:
            for k in keysToAdd:
                obj[prefix + k] = 0.0
    with np.errstate(divide='ignore', invalid='ignore'):
        for obj in catalog:
            x, y = wcs2pix(obj['RADeg'], obj['decDeg'])
            for d, prefix, spline in zip(mapDataList, prefixList, splines):
                if useInterpolator:
                    mapValue = spline(y, x)[0][0]
                else:
                    mapValue = d[int(round(y)), int(round(x))]
                if mapUnits == 'yc':
                    obj[prefix + 'y_c'] = mapValue / 1e-4
                    obj[prefix + 'err_y_c'] = obj[prefix + 'y_c'] / obj[prefix + 'SNR']
                    deltaTc = convertToDeltaT(mapValue, obsFrequencyGHz=ycObsFreqGHz)
                    obj[prefix + 'deltaT_c'] = deltaTc
                    obj[prefix + 'err_deltaT_c'] = abs(deltaTc / obj[prefix + 'SNR'])
                elif mapUnits == 'uK':
                    obj[prefix + 'deltaT_c'] = mapValue
                    obj[prefix + 'err_deltaT_c'] = mapValue / obj[prefix + 'SNR']
                    if reportJyFluxes:
                        obj[prefix + 'fluxJy'] = deltaTToJyPerSr(obj[prefix + 'deltaT_c'], obsFreqGHz) * beamSolidAngle_nsr * 1.e-9
                        obj[prefix + 'err_fluxJy'] = deltaTToJyPerSr(obj[prefix + 'err_deltaT_c'], obsFreqGHz) * beamSolidAngle_nsr * 1.e-9
