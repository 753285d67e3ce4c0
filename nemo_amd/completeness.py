"""Noise / area tables of the selection function (the part of nemo/completeness.py the detection path feeds):

* ``getRMSTab`` - nemo/completeness.py:1284-1345: the distinct noise levels of a tile's RMS map and the survey area at each.

The maps come from the run that made them (``config.cachedRMSMapDict`` / ``areaMaskDict`` / ``flagMaskDict``, device
tensors), or from ``selFnDir`` (RMSMap_<photFilter>#<tile>.fits, areaMask#<tile>.fits, flagMask#<tile>.fits) when they
were written there.  The table is made on the device (nf_rms_table: one pass over the maps, integer sums)."""
import os

import numpy as np

from . import engine, catalogs, maps, fitsio


def _tileMap(config, attr, tileName, selFnDir, stem):
    held = getattr(config, attr, None) if config is not None else None
    if held is not None and tileName in held:
        return held[tileName]
    if selFnDir is not None:
        for path in (selFnDir + os.path.sep + tileName + os.path.sep + "%s#%s.fits" % (stem, tileName),
                     selFnDir + os.path.sep + "%s#%s.fits" % (stem, tileName)):
            if os.path.exists(path):
                return fitsio.readFITS(path)[0]
    return None


def _loadTile(tileName, baseDir, baseFileName, extension='fits'):
    """A tile image from <base>#<tile>.fits (in baseDir or in baseDir/<tile>/) or from the extension <tile> of the
    multi-extension file <base>.fits - whichever is found, in the reference's order (nemo/completeness.py:1132-1163).
    Returns (map array, WCS)."""
    from .wcs import CARWCS
    sep = os.path.sep
    perTile = "%s#%s.%s" % (baseFileName, tileName, extension)
    if os.path.exists(baseDir + sep + perTile):
        fileName = baseDir + sep + perTile
    elif os.path.exists(baseDir + sep + tileName + sep + perTile):
        fileName = baseDir + sep + tileName + sep + perTile
    else:
        fileName = baseDir + sep + "%s.%s" % (baseFileName, extension)
    hdus = fitsio.readHDUs(fileName)
    pick = None
    for name, data, header in hdus:
        if name == tileName and data is not None:
            pick = (data, header)
            break
    if pick is None:                       # not found by name: the first extension that holds an image
        for name, data, header in hdus:
            if data is not None and not isinstance(data, dict):
                pick = (data, header)
                break
    if pick is None:
        raise Exception("no image for tile '%s' in %s" % (tileName, fileName))
    return pick[0], CARWCS(pick[1])


def loadAreaMask(tileName, selFnDir):
    """nemo/completeness.py:1031-1047."""
    return _loadTile(tileName, selFnDir, "areaMask", extension='fits')


def loadFlagMask(tileName, selFnDir):
    """nemo/completeness.py:1050-1066."""
    return _loadTile(tileName, selFnDir, "flagMask", extension='fits')


def loadRMSMap(tileName, selFnDir, photFilter):
    """nemo/completeness.py:1069-1088."""
    return _loadTile(tileName, selFnDir, "RMSMap_%s" % (photFilter), extension='fits')


def getRMSTab(tileName, photFilterLabel, selFnDir, footprintLabel=None, maxFlags=None, config=None, intersectMask=None):
    """Table of survey area (``areaDeg2``) against noise level (``y0RMS``) for one tile (nemo/completeness.py:1284-1345).
    ``intersectMask``: the footprint's intersection mask (what makeIntersectionMask returns there), given by the caller."""
    if footprintLabel is not None and intersectMask is None:
        raise NotImplementedError("footprint intersection masks (completeness.makeIntersectionMask, reprojection of survey "
                                  "footprints) are outside the hot path: pass intersectMask")
    RMSMap = _tileMap(config, 'cachedRMSMapDict', tileName, selFnDir, "RMSMap_%s" % (photFilterLabel))
    areaMap = _tileMap(config, 'areaMaskDict', tileName, selFnDir, "areaMask")
    if RMSMap is None or areaMap is None:
        raise Exception("no RMS map / area mask for tile '%s' (run the filters first, or give a selFnDir that holds them)" % (tileName))
    flagMask = None
    if maxFlags is not None:
        flagMask = _tileMap(config, 'flagMaskDict', tileName, selFnDir, "flagMask")
        if flagMask is None:
            raise Exception("maxFlags needs the flag mask of tile '%s'" % (tileName))
    wcs = config.tileCoordsDict[tileName]['wcs'] if config is not None else None
    if wcs is None:
        raise Exception("getRMSTab needs the tile's WCS (config.tileCoordsDict)")
    t = engine.torch()
    dRMS = engine.to_dev(RMSMap)
    if dRMS.dtype != t.float32:
        dRMS = dRMS.to(t.float32)           # (the noise maps of the path are float32: nemo/filters.py:342-480)
    eng = engine.TileEngine.get(tuple(dRMS.shape))
    rowArea = maps.getPixelAreaArcmin2Map(tuple(dRMS.shape), wcs, rowsOnly=True)          # float32 arcmin^2 per row
    rowAreaDeg2 = (rowArea / (60.0 ** 2)).astype(np.float32)                             # :1322 (float32 / python float)
    values, counts, areas = eng.rms_table(dRMS, engine.to_dev(areaMap, dtype=t.uint8),
                                          None if flagMask is None else engine.to_dev(flagMask).to(t.int32), maxFlags,
                                          None if intersectMask is None else engine.to_dev(intersectMask, dtype=t.uint8), rowAreaDeg2)
    RMSTab = catalogs.Table()
    RMSTab['areaDeg2'] = areas
    RMSTab['y0RMS'] = values
    if np.less(RMSTab['areaDeg2'], 0).sum() > 0:
        raise Exception("Negative area in tile '%s' - check your survey mask (and delete/remake tileDir files if necessary)." % (tileName))
    from . import __version__
    RMSTab.meta['NEMOVER'] = __version__
    return RMSTab
