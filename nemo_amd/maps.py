"""Map containers and the two map-level helpers the filters call.

Mirrors, for the hot path only:

* ``MapDict`` (copy / loadTile / preprocess)   - nemo/maps.py:47-480
* ``subtractBackground`` / ``smoothMap``       - nemo/maps.py:1279-1292,1357-1382
* ``convertToDeltaT``                          - nemo/maps.py:679-697

* ``makeModelImage`` / ``makeDegreesDistanceMap`` - nemo/maps.py:1658-1853, 2479-2536

Maps are held in memory (numpy arrays, or torch tensors already resident in HBM): a ``MapDict`` takes
``mapData`` / ``weightsData`` (arrays) or ``mapFileName`` / ``weightsFileName`` (``.npy`` or simple FITS
files readable by :mod:`nemo_amd.fitsio`).  The catalogue-driven preprocessing steps of the reference (nemo/maps.py:326-465:
source injection, the 20 x 20 median-filtered background, holes cut and filled at catalogue positions, model subtraction)
run on the device: as soon as one of them is asked for, the tile is uploaded and stays in HBM for the filters.
Catalogues are :class:`nemo_amd.catalogs.Table` objects (what the pipeline's earlier passes return).
"""
import os

import numpy as np

from . import engine
from .signals import convertToDeltaT  # noqa: F401  (re-exported: reference has it in maps)
from .wcs import calcAngSepDeg

# not on the path: simulated skies (pixell curvedsky), tILe-C beam matching, dilation by mahotas, unseeded random fills
_UNSUPPORTED_KEYS = ['CMBSimSeed', 'applyBeamConvolution', 'smoothKernel', 'extendedMask', 'apodizeUsingSurveyMask']
_DEVICE_KEYS = ['injectSources', 'maskPointSourcesFromCatalog', 'subtractModelFromCatalog', 'maskAndFillFromCatalog']


_FUSED_LOAD = os.environ.get('NEMO_AMD_FUSED_LOAD', '1') != '0'      # 0: MapDict.preprocess loads a tile statement by statement


def _is_tensor(x):
    return hasattr(x, 'is_cuda')


class MapDict(dict):
    """Dictionary describing one frequency map (nemo/maps.py:47-80)."""

    def __init__(self, inputDict, tileCoordsDict=None):
        super(MapDict, self).__init__(inputDict)
        self.tileCoordsDict = tileCoordsDict
        self._maskKeys = ['pointSourceMask', 'surveyMask', 'flagMask', 'extendedMask']
        self.validMapKeys = ['mapFileName', 'weightsFileName'] + self._maskKeys

    def copy(self):
        return MapDict(self, tileCoordsDict=self.tileCoordsDict)

    def _load(self, value):
        if isinstance(value, str):
            if value.endswith('.npy'):
                return np.load(value)
            from . import fitsio
            return fitsio.readFITS(value)[0]
        return value

    def loadTile(self, mapKey, tileName, returnWCS=False):
        """nemo/maps.py:83-175 for in-memory / single-tile inputs (tile clipping via ``tileCoordsDict``)."""
        if mapKey not in self.validMapKeys:
            raise Exception("mapKey must be one of %s - given mapKey = '%s'." % (self.validMapKeys, mapKey))
        alias = {'mapFileName': 'mapData', 'weightsFileName': 'weightsData'}
        if mapKey in alias and self.get(alias[mapKey]) is not None:
            data = self[alias[mapKey]]
        else:
            data = self._load(self.get(mapKey))
        if data is None:
            raise Exception("no map given for '%s'" % (mapKey))
        if self.tileCoordsDict is not None and tileName in self.tileCoordsDict:
            minX, maxX, minY, maxY = self.tileCoordsDict[tileName]['clippedSection']
            data = data[..., minY:maxY, minX:maxX]
        data = data.clone() if _is_tensor(data) else np.array(data)
        if data.ndim == 3:
            data = data[0]
        if mapKey in self._maskKeys and not _is_tensor(data) and data.dtype != np.uint8:
            data = np.array(data, dtype=np.uint8)
        if mapKey == 'surveyMask' and self.tileCoordsDict is not None and tileName in self.tileCoordsDict:
            minX, maxX, minY, maxY = self.tileCoordsDict[tileName]['areaMaskInClipSection']
            data[:minY, :] = 0
            data[maxY:, :] = 0
            data[:, :minX] = 0
            data[:, maxX:] = 0
        if returnWCS:
            wcs = self['wcs']
            if self.tileCoordsDict is not None and tileName in self.tileCoordsDict:
                wcs = self.tileCoordsDict[tileName].get('wcs', wcs)
            return data, wcs
        return data

    def _loadTileStatements(self, tileName, onDevice):
        """The map, its weights and its masks for ``tileName``, statement by statement (nemo/maps.py:232-262)."""
        data, wcs = self.loadTile('mapFileName', tileName, returnWCS=True)
        if onDevice:
            data = engine.to_dev(data)
        if 'calibFactor' in self.keys():
            data = data * self['calibFactor']
        if self.get('weightsFileName') is not None or self.get('weightsData') is not None:
            weights = self.loadTile('weightsFileName', tileName)
            if onDevice:
                weights = engine.to_dev(weights)
        else:
            weights = np.ones(data.shape) if not _is_tensor(data) else engine.torch().ones_like(data)
        data[weights == 0] = 0                                                    # :243
        if self.get('surveyMask') is not None:
            surveyMask = self.loadTile('surveyMask', tileName)
            if onDevice:
                surveyMask = engine.to_dev(surveyMask, dtype=engine.torch().uint8)
        else:
            if _is_tensor(data):
                surveyMask = (weights != 0).to(engine.torch().uint8)
            else:
                surveyMask = np.ones(data.shape, dtype=np.uint8)
                surveyMask[weights == 0] = 0
        if self.get('pointSourceMask') is not None:
            psMask = self.loadTile('pointSourceMask', tileName)
            if onDevice:
                psMask = engine.to_dev(psMask, dtype=engine.torch().uint8)
        else:
            psMask = np.ones(data.shape, dtype=np.uint8) if not _is_tensor(data) else \
                engine.torch().ones(data.shape, dtype=engine.torch().uint8, device=data.device)
        if self.get('flagMask') is not None:
            flagMask = self.loadTile('flagMask', tileName)
            if onDevice:
                flagMask = engine.to_dev(flagMask, dtype=engine.torch().uint8)
            flagMask = flagMask * surveyMask
        else:
            flagMask = np.zeros(tuple(data.shape), dtype=np.uint8) if not _is_tensor(data) else \
                engine.torch().zeros(tuple(data.shape), dtype=engine.torch().uint8, device=data.device)
        return data, weights, wcs, surveyMask, psMask, flagMask

    def _loadTileFused(self, tileName):
        """_loadTileStatements in ONE pass (nf_load_tile) when the map, its weights and its masks are device tensors of the
        plain kinds (2-D float32 / float64 planes, uint8 masks, no calibration factor); None otherwise."""
        if 'calibFactor' in self.keys():
            return None
        t = engine.torch()
        src = self.get('mapData')
        if not (_is_tensor(src) and src.is_cuda and src.ndim == 2 and src.dtype in (t.float32, t.float64) and src.is_contiguous()):
            return None
        w = self.get('weightsData')
        if w is None and self.get('weightsFileName') is not None:
            return None
        if w is not None and not (_is_tensor(w) and w.is_cuda and tuple(w.shape) == tuple(src.shape) and w.is_contiguous() and
                                  w.dtype in (t.float32, t.float64)):
            return None
        masks = []
        for k in ('surveyMask', 'pointSourceMask', 'flagMask'):
            m = self.get(k)
            if m is not None and not (_is_tensor(m) and m.is_cuda and m.dtype == t.uint8 and tuple(m.shape) == tuple(src.shape) and
                                      m.is_contiguous()):
                return None
            masks.append(m)
        srcH, srcW = src.shape
        x0, x1, y0, y1 = 0, srcW, 0, srcH
        area = (-1, -1, -1, -1)
        wcs = self['wcs']
        if self.tileCoordsDict is not None and tileName in self.tileCoordsDict:
            x0, x1, y0, y1 = [int(v) for v in self.tileCoordsDict[tileName]['clippedSection']]
            if x0 < 0 or y0 < 0 or x1 > srcW or y1 > srcH:
                return None              # (a section beyond the map: slicing has its own rules for that)
            area = tuple(int(v) for v in self.tileCoordsDict[tileName]['areaMaskInClipSection'])
            wcs = self.tileCoordsDict[tileName].get('wcs', wcs)
        H, W = y1 - y0, x1 - x0
        if H < 1 or W < 1 or H > 65535 or min(area) < 0 and area != (-1, -1, -1, -1):
            return None
        dev = src.device
        data = t.empty((H, W), dtype=src.dtype, device=dev)
        weights = t.empty((H, W), dtype=w.dtype if w is not None else src.dtype, device=dev)
        surveyMask, psMask, flagMask = (t.empty((H, W), dtype=t.uint8, device=dev) for _ in range(3))
        eng = engine
        # areaMaskInClipSection = (minX, maxX, minY, maxY): rows outside minY..maxY-1 and columns outside minX..maxX-1 are zeroed
        ax0, ax1, ay0, ay1 = area
        eng.check(eng._lib.load().nf_load_tile(eng.prec_of(src), eng.ptr(src), eng.prec_of(weights), eng.ptr(w), eng.ptr(masks[0]),
                                               eng.ptr(masks[1]), eng.ptr(masks[2]), srcH, srcW, x0, y0, H, W, ax0, ax1, ay0, ay1,
                                               eng.ptr(data), eng.ptr(weights), eng.ptr(surveyMask), eng.ptr(psMask), eng.ptr(flagMask),
                                               eng.stream_ptr()), "nf_load_tile")
        return data, weights, wcs, surveyMask, psMask, flagMask

    def preprocess(self, tileName='PRIMARY', diagnosticsDir=None):
        """nemo/maps.py:178-480, the steps that apply to plain (map, weights, masks) inputs."""
        for k in _UNSUPPORTED_KEYS:
            if k in self and self[k] not in (None, False):
                raise NotImplementedError("MapDict key '%s' (nemo/maps.py:252-262,294-323,346-379) is not part of this build" % (k))
        onDevice = any(k in self and self[k] is not None for k in _DEVICE_KEYS)
        # An injection run preprocesses the same tile thousands of times with nothing changed but the catalogue to paint in
        # (maps.sourceInjectionTest): what comes before the catalogue steps is kept in HBM, per tile, in the dictionary the
        # run hung on its maps ('_preprocCache'), and is never written to afterwards.
        cache = self.get('_preprocCache', None) if onDevice else None
        ckey = ('tile', tileName, self.get('label', None), self.get('obsFreqGHz', None))
        if cache is not None and ckey in cache:
            data, weights, wcs, surveyMask, psMask, flagMask = cache[ckey]
            data, surveyMask, psMask, flagMask = self._catalogueSteps(data, weights, wcs, surveyMask, psMask, flagMask, tileName,
                                                                      sharedInputs=True)
            self._setPreprocessed(data, weights, wcs, surveyMask, psMask, flagMask, tileName)
            return
        fused = self._loadTileFused(tileName) if _FUSED_LOAD else None
        if fused is not None:
            data, weights, wcs, surveyMask, psMask, flagMask = fused
        else:
            data, weights, wcs, surveyMask, psMask, flagMask = self._loadTileStatements(tileName, onDevice)
        if self.get('RADecSection') is not None:
            RAMin, RAMax, decMin, decMax = self['RADecSection']
            x0, y0 = wcs.wcs2pix(RAMax, decMin)
            x1, y1 = wcs.wcs2pix(RAMin, decMax)
            xa, xb = sorted([int(round(x0)), int(round(x1))])
            ya, yb = sorted([int(round(y0)), int(round(y1))])
            xa, ya = max(xa, 0), max(ya, 0)
            xb, yb = min(xb, data.shape[1]), min(yb, data.shape[0])
            sl = (slice(ya, yb), slice(xa, xb))
            data, weights, psMask, surveyMask, flagMask = data[sl], weights[sl], psMask[sl], surveyMask[sl], flagMask[sl]
            hdr = dict(wcs.header)
            hdr['NAXIS1'], hdr['NAXIS2'] = xb - xa, yb - ya
            hdr['CRPIX1'] = hdr['CRPIX1'] - xa
            hdr['CRPIX2'] = hdr['CRPIX2'] - ya
            wcs = type(wcs)(hdr)
            if data.shape[0] == 0 or data.shape[1] == 0:
                raise Exception("Clipping using RADecSection returned empty array - check RADecSection in config .yml file is in map")
        if onDevice:
            if cache is not None:
                t = engine.torch()
                cache[ckey] = (data.contiguous(), weights, wcs, surveyMask.to(t.uint8).contiguous(), psMask.to(t.uint8).contiguous(),
                               flagMask.to(t.uint8).contiguous())
                data, weights, wcs, surveyMask, psMask, flagMask = cache[ckey]
            data, surveyMask, psMask, flagMask = self._catalogueSteps(data, weights, wcs, surveyMask, psMask, flagMask, tileName,
                                                                      sharedInputs=cache is not None)
        self._setPreprocessed(data, weights, wcs, surveyMask, psMask, flagMask, tileName)

    def _setPreprocessed(self, data, weights, wcs, surveyMask, psMask, flagMask, tileName):
        self['data'] = data
        self['weights'] = weights
        self['wcs'] = wcs
        self['surveyMask'] = surveyMask
        self['pointSourceMask'] = psMask
        self['flagMask'] = flagMask
        self['tileName'] = tileName
        if tuple(self['data'].shape) != tuple(self['pointSourceMask'].shape):
            raise Exception("Map and point source mask dimensions are not the same (they should also have same WCS)")
        if tuple(self['data'].shape) != tuple(self['surveyMask'].shape):
            raise Exception("Map and survey mask dimensions are not the same (they should also have same WCS)")


def _holeTable(entry):
    if isinstance(entry, str):
        from . import catalogs
        return catalogs.readCatalog(entry)       # FITS table or tab-delimited .csv, as catalogs.writeCatalog writes them
    return entry


def _mapDictCatalogueSteps(self, data, weights, wcs, surveyMask, psMask, flagMask, tileName, sharedInputs=False):
    """nemo/maps.py:325-465 on device tensors: source injection, the median-filtered background, holes at catalogue
    positions (masked in both masks, filled with the background), model subtraction with its flag.  ``sharedInputs``: the
    tensors handed in belong to the tile cache of an injection run and are not written to (steps that work in place get
    copies)."""
    t = engine.torch()
    from . import catalogs
    eng = engine.TileEngine.get(tuple(data.shape))
    shape = tuple(data.shape)
    data = data.contiguous()
    inPlaceSteps = any(self.get(h) is not None for h in ['maskPointSourcesFromCatalog', 'maskAndFillFromCatalog', 'subtractModelFromCatalog'])
    if not sharedInputs or inPlaceSteps:
        surveyMask = surveyMask.to(t.uint8).contiguous().clone()
        psMask = psMask.to(t.uint8).contiguous().clone()
        flagMask = flagMask.to(t.uint8).contiguous().clone()
    dataIsShared = sharedInputs
    if self.get('injectSources') is not None:                                     # :326-344
        inj = self['injectSources']
        GNFWParams, obsFreqGHz = (inj['GNFWParams'], self['obsFreqGHz']) if 'GNFWParams' in inj.keys() else (None, None)
        valid = self.tileCoordsDict[tileName]['areaMaskInClipSection'] if self.tileCoordsDict is not None else None
        # (the object windows and their haversine tables are the same for every frequency map of a tile: kept with the request)
        modelMap = makeModelImage(shape, wcs, inj['catalog'], self['beamFileName'], obsFreqGHz=obsFreqGHz, GNFWParams=GNFWParams,
                                  profile=inj['profile'], validAreaSection=valid, override=inj['override'],
                                  _geomCache=inj.setdefault('_paintCache', {}))
        if modelMap is not None:
            # modelMap[weights == 0] = 0; data = data + modelMap - one pass, into a new map (or into this map's plane of the cube
            # the run keeps for the tile's frequency maps); dtype promotes as numpy's does (float64 model in the override branch)
            data = eng.inject_model(data, modelMap.contiguous(), weights if weights.dtype in (t.float32, t.float64) else weights.to(t.float64),
                                    out=self._injectionPlane(tileName, shape, data.dtype, modelMap.dtype))
            dataIsShared = False
    if dataIsShared and inPlaceSteps:
        data = data.clone()
    # The reference takes the median background whenever one of these keys EXISTS (:360-368) - and startUp puts
    # 'maskPointSourcesFromCatalog': None into every map dictionary (nemo/startUp.py:43-50), so it always does.  The background
    # is only read where holes are cut (:419-424, 459-465): it is made when a catalogue will cut some.
    holeFilling = any(self.get(h) is not None for h in ['maskPointSourcesFromCatalog', 'maskAndFillFromCatalog', 'extendedMask'])
    if holeFilling:
        pixRad = (10.0 / 60.0) / wcs.getPixelSizeDeg()
        bckData = eng.median_filter(data, int(pixRad))
    if self.get('maskPointSourcesFromCatalog') is not None:                       # :383-424
        entries = self['maskPointSourcesFromCatalog']
        if type(entries) is not list:
            entries = self['maskPointSourcesFromCatalog'] = [entries]
        psMask = t.ones_like(psMask)
        recs, boxes, limits = [], [], []
        for info in entries:
            fluxCutJy = 0.0
            if type(info) == dict:
                tab, fluxCutJy = _holeTable(info['path']), info['fluxCutJy']
            else:
                tab = _holeTable(info)
            if 'fluxJy' in tab.keys():
                tab = tab[np.asarray(tab['fluxJy']) > fluxCutJy]
            tab = catalogs.getCatalogWithinImage(tab, shape, wcs)
            for row in tab:
                if 'rArcmin' in tab.keys():
                    maskRadiusArcmin = row['rArcmin']
                elif 'ellipse_A' in tab.keys():
                    xPixSizeArcmin = (wcs.getXPixelSizeDeg() / np.cos(np.radians(row['decDeg']))) * 60
                    maskRadiusArcmin = (row['ellipse_A'] / xPixSizeArcmin) / 2
                else:
                    raise Exception("To mask sources in a catalog, need either 'rArcmin' or 'ellipse_A' column to be present.")
                rec, box = degreesDistanceBox(shape, wcs, row['RADeg'], row['decDeg'], maskRadiusArcmin / 60)
                recs.append(rec + [maskRadiusArcmin / 60.0])
                boxes.append(box)
                limits.append(maskRadiusArcmin / 60.0)
        eng.mask_fill_discs(data, bckData, surveyMask, psMask, recs, boxes, times60=False, persistentLimits=limits)
    if self.get('subtractModelFromCatalog') is not None:                          # :426-437
        tabs = self['subtractModelFromCatalog']
        if type(tabs) is not list:
            tabs = self['subtractModelFromCatalog'] = [tabs]
        for tab in tabs:
            tab = catalogs.getCatalogWithinImage(_holeTable(tab), shape, wcs)
            model = makeModelImage(shape, wcs, tab, self['beamFileName'], obsFreqGHz=self['obsFreqGHz'])
            if model is not None:
                eng.add_model(data, model.contiguous(), flagMask, subtract=True)
    if self.get('maskAndFillFromCatalog') is not None:                            # :439-465
        tabs = self['maskAndFillFromCatalog']
        if type(tabs) is not list:
            tabs = self['maskAndFillFromCatalog'] = [tabs]
        for tab in tabs:
            tab = catalogs.getCatalogWithinImage(_holeTable(tab), shape, wcs)
            if len(tab) > 0 and 'ellipse_A' not in tab.keys() and 'maskHoleRadiusArcmin' not in self.keys():
                raise Exception("Need to set measureShapes: True or set maskHoleRadiusArcmin to use maskAndFillFromCatalog")
            recs, boxes = [], []
            for row in tab:
                if 'ellipse_B' in tab.keys():          # (the reference's test - 'ellipse_A' and 'ellipse_B' in keys - reads so)
                    xPixSizeArcmin = (wcs.getXPixelSizeDeg() / np.cos(np.radians(row['decDeg']))) * 60
                    maskRadiusArcmin = (row['ellipse_A'] / xPixSizeArcmin) / 2
                if self.get('maskHoleRadiusArcmin') is not None:
                    maskRadiusArcmin = self['maskHoleRadiusArcmin']
                if self.get('maskHoleDilationFactor') is not None:
                    maskRadiusArcmin = maskRadiusArcmin * self['maskHoleDilationFactor']
                rec, box = degreesDistanceBox(shape, wcs, row['RADeg'], row['decDeg'], maskRadiusArcmin / 60)
                recs.append(rec + [maskRadiusArcmin])
                boxes.append(box)
            eng.mask_fill_discs(data, bckData, surveyMask, psMask, recs, boxes, times60=True)
    return data, surveyMask, psMask, flagMask


def _mapDictInjectionPlane(self, tileName, shape, dataDtype, modelDtype):
    """Where the injected map of this frequency is written in an injection run: plane '_planeIndex' of a cube [N, H, W] kept per
    tile in '_preprocCache' - the filter then transforms the cube as it is instead of stacking N maps again.  None outside such a
    run (a new map is made)."""
    cache, index = self.get('_preprocCache', None), self.get('_planeIndex', None)
    if cache is None or index is None:
        return None
    t = engine.torch()
    odt = t.float32 if (dataDtype == t.float32 and modelDtype == t.float32) else t.float64
    key = ('cube', tileName, tuple(shape), odt)
    if key not in cache:
        cache[key] = t.empty((int(self['_numPlanes']),) + tuple(shape), dtype=odt, device=engine.require_gpu())
    cube = cache[key]
    self['_cubeView'] = (cube, int(index))
    return cube[int(index)]


MapDict._catalogueSteps = _mapDictCatalogueSteps
MapDict._injectionPlane = _mapDictInjectionPlane


def degreesDistanceBox(shape, wcs, RADeg, decDeg, maxDistDegrees):
    """The scalars of maps.makeDegreesDistanceMap (nemo/maps.py:2506-2529): ([x0, y0, xPixScale, yPixScale], [minX, maxX, minY,
    maxY]) - the map of distances itself is formed on the device (nf_disc_fill / nf_disc_apply)."""
    x0, y0 = wcs.wcs2pix(RADeg, decDeg)
    ra1, dec1 = wcs.pix2wcs(x0 + 1, y0 + 1)
    xPixScale = calcAngSepDeg(RADeg, decDeg, ra1, decDeg)
    yPixScale = calcAngSepDeg(RADeg, decDeg, RADeg, dec1)
    xDistPix = int(round((maxDistDegrees) / xPixScale))
    yDistPix = int(round((maxDistDegrees) / yPixScale))
    Y, X = shape
    minX, maxX = max(int(round(x0)) - xDistPix, 0), min(int(round(x0)) + xDistPix, X)
    minY, maxY = max(int(round(y0)) - yDistPix, 0), min(int(round(y0)) + yDistPix, Y)
    # a negative radius (ellipse_A = -99 rows of a shapes catalogue) gives an empty range there and an empty box here
    minX, minY = min(minX, X), min(minY, Y)
    maxX, maxY = max(maxX, minX), max(maxY, minY)
    return [float(x0), float(y0), float(xPixScale), float(yPixScale)], [minX, maxX, minY, maxY]


def _modelObjects(catalog, beam, obsFreqGHz, GNFWParams, profile, override, TCMBAlpha, maxSizeDegMultiplier, cosmoModel):
    """Per catalogue row what the painter needs (nemo/maps.py:1752-1840): (RADeg, decDeg, radial table, maxSizeDeg, float32
    factors applied to the painted value)."""
    from . import signals
    out = []
    keys = catalog.keys()
    RAs, decs = np.asarray(catalog['RADeg'], dtype=np.float64).tolist(), np.asarray(catalog['decDeg'], dtype=np.float64).tolist()
    if 'y_c' in keys or 'true_y_c' in keys:
        kind = {'A10': 'arnaud', 'B12': 'battaglia'}.get(profile)
        if kind is None:
            raise Exception("Didn't understand profile - should be A10 or B12.")
        gp = 'default' if GNFWParams in (None, 'default') else GNFWParams
        if override is not None:
            # one model for every row (:1752-1758): one table, one size, one spectral factor - only the amplitudes differ
            z, M500 = override['redshift'], override['M500']
            theta500Arcmin = signals.calcTheta500Arcmin(z, M500, cosmoModel)
            maxSizeDeg = 5 * (theta500Arcmin / 60)
            base = signals.clusterModelTable(kind, beam, z, M500, gp, 1.0, maxSizeDeg, baseOnly=True)
            factors = None
            if obsFreqGHz is not None:
                factors = (float(signals.fSZ(obsFreqGHz, TCMBAlpha=TCMBAlpha, z=z)), signals.TCMB * 1e6)
            for RADeg, decDeg, y_c in zip(RAs, decs, np.asarray(catalog['y_c'], dtype=np.float64).tolist()):
                out.append((RADeg, decDeg, signals.scaledTable(base, float(y_c * 1e-4)), maxSizeDeg, factors))
            return out
        for row in catalog:
            if 'true_M500c' in keys:
                M500, z, y0 = row['true_M500c'] * 1e14, row['redshift'], row['true_y_c'] * 1e-4
            else:
                if 'template' not in keys:
                    raise Exception("No M500, z, or template column found in catalog.")
                bits = str(row['template']).split("#")[0].split("_")
                M500 = float(bits[1][1:].replace("p", "."))
                z = float(bits[2][1:].replace("p", "."))
                y0 = row['y_c'] * 1e-4
                if z == 0:
                    raise Exception("z = 0 template? template = %s, tileName = %s" % (row['template'], row['tileName']))
            theta500Arcmin = signals.calcTheta500Arcmin(z, M500, cosmoModel)
            maxSizeDeg = maxSizeDegMultiplier * (theta500Arcmin / 60)
            table = signals.clusterModelTable(kind, beam, z, M500, gp, float(y0), maxSizeDeg)
            factors = None
            if obsFreqGHz is not None:
                factors = (float(signals.fSZ(obsFreqGHz, TCMBAlpha=TCMBAlpha, z=z)), signals.TCMB * 1e6)
            out.append((row['RADeg'], row['decDeg'], table, maxSizeDeg, factors))
    else:
        base = signals.templateTables('beam', beam, {}, amplitude=1.0)       # nemo/signals.py:595-627; one table, many amplitudes
        for RADeg, decDeg, amp in zip(RAs, decs, np.asarray(catalog['deltaT_c'], dtype=np.float64).tolist()):
            out.append((RADeg, decDeg, signals.scaledTable(base, amp), 1.0, None))
    return out


def _modelObjectsUniform(catalog, beam, obsFreqGHz, GNFWParams, profile, override, TCMBAlpha, cosmoModel):
    """_modelObjects for the catalogues whose rows share ONE radial table - an ``override`` model (nemo/maps.py:1752-1758) or point
    sources (:1833-1840) - without a Python statement per row: (RADeg[], decDeg[], base table, maxSizeDeg, factors, amp[]) with
    amp[k] the float32-rounded amplitude signals.scaledTable would give row k; None for any other catalogue."""
    from . import signals
    keys = catalog.keys()
    RAs, decs = np.asarray(catalog['RADeg'], dtype=np.float64), np.asarray(catalog['decDeg'], dtype=np.float64)
    if 'y_c' in keys or 'true_y_c' in keys:
        if override is None:
            return None
        kind = {'A10': 'arnaud', 'B12': 'battaglia'}.get(profile)
        if kind is None:
            raise Exception("Didn't understand profile - should be A10 or B12.")
        gp = 'default' if GNFWParams in (None, 'default') else GNFWParams
        z, M500 = override['redshift'], override['M500']
        theta500Arcmin = signals.calcTheta500Arcmin(z, M500, cosmoModel)
        maxSizeDeg = 5 * (theta500Arcmin / 60)
        base = signals.clusterModelTable(kind, beam, z, M500, gp, 1.0, maxSizeDeg, baseOnly=True)
        factors = None
        if obsFreqGHz is not None:
            factors = (float(signals.fSZ(obsFreqGHz, TCMBAlpha=TCMBAlpha, z=z)), signals.TCMB * 1e6)
        amplitudes = np.asarray(catalog['y_c'], dtype=np.float64) * 1e-4
    else:
        base = signals.templateTables('beam', beam, {}, amplitude=1.0)
        maxSizeDeg, factors = 1.0, None
        amplitudes = np.asarray(catalog['deltaT_c'], dtype=np.float64)
    amps = (base.ampExact * amplitudes).astype(np.float32).astype(np.float64)          # signals.scaledTable, row by row
    return RAs, decs, base, maxSizeDeg, factors, amps


def modelWindow(shape, wcs, RADeg, decDeg, maxSizeDeg):
    """Pixel window of the box of +- maxSizeDeg around an object that signals._paintSignalMap clips before painting into an
    existing map (nemo/signals.py:675-680; the rounding rule of the clip is this build's, as for RADecSection)."""
    xa, ya = wcs.wcs2pix(RADeg + maxSizeDeg / np.cos(np.radians(decDeg)), decDeg - maxSizeDeg)
    xb, yb = wcs.wcs2pix(RADeg - maxSizeDeg / np.cos(np.radians(decDeg)), decDeg + maxSizeDeg)
    x0, x1 = sorted([int(round(xa)), int(round(xb))])
    y0, y1 = sorted([int(round(ya)), int(round(yb))])
    x0, y0 = min(max(x0, 0), shape[1]), min(max(y0, 0), shape[0])
    x1, y1 = min(max(x1, x0), shape[1]), min(max(y1, y0), shape[0])
    return x0, x1, y0, y1


def _modelGeometryMany(shape, wcs, RADeg, decDeg, maxSizeDeg):
    """modelWindow and signals.paintGeometry(window=...) for many objects at once: (windows [n, 4] = x0, x1, y0, y1; the
    objects' tables A[rows], B[rows], C[columns] one after the other; where each object's A and C start).  Every value is
    what the per-object functions give (the same elementwise statements on concatenated arrays)."""
    H, W = shape
    n = len(RADeg)
    cosDec = np.cos(np.radians(decDeg))
    xa, ya = wcs.wcs2pix(RADeg + maxSizeDeg / cosDec, decDeg - maxSizeDeg)
    xb, yb = wcs.wcs2pix(RADeg - maxSizeDeg / cosDec, decDeg + maxSizeDeg)
    xa, ya, xb, yb = (np.rint(np.atleast_1d(v)).astype(np.int64) for v in (xa, ya, xb, yb))      # int(round()): half to even
    x0, x1 = np.minimum(xa, xb), np.maximum(xa, xb)
    y0, y1 = np.minimum(ya, yb), np.maximum(ya, yb)
    x0, y0 = np.minimum(np.maximum(x0, 0), W), np.minimum(np.maximum(y0, 0), H)
    x1, y1 = np.minimum(np.maximum(x1, x0), W), np.minimum(np.maximum(y1, y0), H)
    nrow, ncol = y1 - y0, x1 - x0
    size = 2 * nrow + ncol
    offA = np.concatenate([[0], np.cumsum(size)[:-1]]) if n > 0 else np.zeros(0, dtype=np.int64)
    offC = offA + 2 * nrow
    geom = np.empty(int(size.sum()), dtype=np.float64)
    dec0, ra0 = np.radians(decDeg), np.radians(RADeg)
    # rows: element j of object k is row y0[k] + j
    rowObj = np.repeat(np.arange(n), nrow)
    rowIn = np.arange(len(rowObj)) - np.repeat(np.cumsum(nrow) - nrow, nrow)
    dec = wcs.rowDecRad()[y0[rowObj] + rowIn]
    geom[offA[rowObj] + rowIn] = np.sin((dec - dec0[rowObj]) / 2.0) ** 2
    geom[offA[rowObj] + nrow[rowObj] + rowIn] = np.cos(dec) * np.cos(dec0)[rowObj]
    colObj = np.repeat(np.arange(n), ncol)
    colIn = np.arange(len(colObj)) - np.repeat(np.cumsum(ncol) - ncol, ncol)
    ra = wcs.colRARad()[x0[colObj] + colIn]
    dra = (ra - ra0[colObj] + np.pi) % (2 * np.pi) - np.pi
    geom[offC[colObj] + colIn] = np.sin(dra / 2.0) ** 2
    return np.stack([x0, x1, y0, y1], axis=1), geom, offA, offC


_catalogProfDev = {}      # id(profile array) -> (profile, radii, device [3, ntab] table) for nf_paint_catalog


def makeModelImage(shape, wcs, catalog, beamFileName, obsFreqGHz=None, GNFWParams='default', profile='A10', cosmoModel=None,
                   applyPixelWindow=True, override=None, validAreaSection=None, minSNR=-99, TCMBAlpha=0, reportTimingInfo=False,
                   maxSizeDegMultiplier=5, useInferredSZProperties=False, _geomCache=None):
    """Map of model clusters / point sources at the catalogue's positions (nemo/maps.py:1658-1853), float32, on the device;
    None when no object lies inside the map.  The painter is this build's (signals.RadialTable + nf_paint_catalog): every
    object inside its own +- maxSizeDeg window, summed in float32 in catalogue order."""
    from . import signals, catalogs
    if useInferredSZProperties:
        raise NotImplementedError("useInferredSZProperties (nemoMass -I output) is outside the hot path")
    # (the rows that fall on this tile are the same for every frequency map of the tile: kept with the request)
    given = catalog
    rkey = ('rows', id(given), tuple(shape), id(wcs), minSNR, None if validAreaSection is None else tuple(validAreaSection))
    hit = _geomCache.get(rkey) if _geomCache is not None else None
    if hit is not None and hit[0] is given and hit[1] is wcs:
        catalog = hit[2]
    else:
        catalog = catalogs.getCatalogWithinImage(_holeTable(catalog), shape, wcs)
        SNRKey = 'SNR' if 'SNR' in catalog.keys() else ('fixed_SNR' if 'fixed_SNR' in catalog.keys() else None)
        if SNRKey is not None and len(catalog) > 0:
            catalog = catalog[np.asarray(catalog[SNRKey]) > minSNR]
        if validAreaSection is not None and len(catalog) > 0:
            x0, x1, y0, y1 = validAreaSection
            x, y = wcs.wcs2pix(np.asarray(catalog['RADeg'], dtype=np.float64), np.asarray(catalog['decDeg'], dtype=np.float64))
            x, y = np.atleast_1d(x), np.atleast_1d(y)
            catalog = catalog[(x >= x0) & (x < x1) & (y >= y0) & (y < y1)]
        if _geomCache is not None:
            _geomCache[rkey] = (given, wcs, catalog)
    if len(catalog) == 0:
        return None
    if cosmoModel is None:
        cosmoModel = signals.fiducialCosmoModel
    beam = signals._asBeam(beamFileName)
    perObjectFactors = override is None          # the override branch converts the summed map instead (:1786-1788)
    uniform = _modelObjectsUniform(catalog, beam, obsFreqGHz, GNFWParams, profile, override, TCMBAlpha, cosmoModel)
    if uniform is not None:
        RAs, decs, base, maxSizeDeg, factors, amps = uniform
        objects = None
        nObj = len(RAs)
        sizes = np.full(nObj, float(maxSizeDeg))
    else:
        objects = _modelObjects(catalog, beam, obsFreqGHz, GNFWParams, profile, override, TCMBAlpha, maxSizeDegMultiplier, cosmoModel)
        nObj = len(objects)
        RAs = np.array([o[0] for o in objects], dtype=np.float64)
        decs = np.array([o[1] for o in objects], dtype=np.float64)
        sizes = np.array([o[3] for o in objects], dtype=np.float64)
    # windows and haversine tables of ALL objects in a few numpy statements (modelWindow / signals.paintGeometry element for
    # element); the same catalogue on the same tile - the frequency maps of a tile, pass after pass of a mask catalogue - is
    # served from the cache
    gkey = ('many', tuple(shape), wcs._crval1, wcs._crpix1, wcs._crpix2, wcs._cdelt1, wcs._cdelt2, RAs.tobytes(), decs.tobytes(),
            sizes.tobytes())
    held = _geomCache.get(gkey) if _geomCache is not None else None
    if held is None:
        held = _modelGeometryMany(shape, wcs, RAs, decs, sizes)
        if _geomCache is not None:
            _geomCache[gkey] = held
    winArr, geom, offA, offC = held[:4]
    eng = engine.TileEngine.get(shape)
    if _geomCache is not None and len(held) == 4:
        # (the device copy of the windows' tables stays with them)
        held = _geomCache[gkey] = held + (engine.to_dev(np.ascontiguousarray(geom if len(geom) > 0 else np.zeros(1), dtype=np.float64)),)
    geomDev = held[4] if len(held) > 4 else None
    wins = np.zeros((nObj, 8), dtype=np.int64)
    wins[:, 0:4] = winArr
    wins[:, 4] = offA
    wins[:, 5] = offC
    if uniform is not None:
        # one table for every row: the records in a few array statements
        useF = 1.0 if (factors is not None and perObjectFactors) else 0.0
        f1, f2 = factors if factors is not None else (1.0, 1.0)
        objs = np.zeros((nObj, 8))
        objs[:, 0], objs[:, 1], objs[:, 2], objs[:, 3], objs[:, 5] = amps, base.sign, f1, f2, useF
        profTabs = [base]
    else:
        objs, profTabs, profIndex = [], [], {}
        for k, (RADeg, decDeg, table, maxSizeDeg, factors) in enumerate(objects):
            pid = profIndex.get(id(table.prof))
            if pid is None:
                pid = profIndex[id(table.prof)] = len(profTabs)
                profTabs.append(table)
            useF = 1.0 if (factors is not None and perObjectFactors) else 0.0
            f1, f2 = factors if factors is not None else (1.0, 1.0)
            objs.append([table.amp, table.sign, f1, f2, float(pid), useF, 0.0, 0.0])
            wins[k, 6] = pid
    ntab = len(profTabs[0].r)
    # radii, profile and slopes of a profile as the painter reads them: made once per profile array and kept on the device
    # (scaled tables of one model share their arrays, signals.scaledTable)
    devTabs = []
    for table in profTabs:
        hit = _catalogProfDev.get(id(table.prof))
        if hit is None or hit[0] is not table.prof or hit[1] is not table.r:
            slope = np.zeros(len(table.r))
            slope[:-1] = (table.prof[1:] - table.prof[:-1]) / (table.r[1:] - table.r[:-1])
            if len(_catalogProfDev) >= 256:
                _catalogProfDev.clear()
            hit = _catalogProfDev[id(table.prof)] = (table.prof, table.r, engine.to_dev(np.stack([table.r, table.prof, slope])))
        devTabs.append(hit[2])
    profsDev = devTabs[0] if len(devTabs) == 1 else engine.torch().stack(devTabs)
    modelMap = eng.paint_catalog(objs, wins, None, None, ntab, geomDev=geomDev if geomDev is not None else
                                 engine.to_dev(np.ascontiguousarray(geom if len(geom) > 0 else np.zeros(1), dtype=np.float64)),
                                 profsDev=profsDev.contiguous())
    if override is not None and obsFreqGHz is not None:
        # convertToDeltaT of the whole map (:1786-1788): float32 * float64 scalar -> a float64 map from here on (numpy >= 2)
        fx = float(signals.fSZ(obsFreqGHz, TCMBAlpha=TCMBAlpha, z=override['redshift']))
        modelMap = (modelMap.double() * fx) * (signals.TCMB * 1e6)
    if applyPixelWindow == True:
        modelMap = eng.apply_window(modelMap.contiguous(), 1.0)
    return modelMap


def saveFITS(outputFileName, mapData, wcs, compressionType=None):
    """A map and its WCS header as a FITS image (nemo/maps.py:2436-2476).  ``compressionType`` 'PLIO_1' (masks): uint8, written as
    a tile-compressed image extension behind an empty primary HDU, as astropy's CompImageHDU in an HDUList is; 'RICE_1' (lossy,
    float maps) is not written here: such maps go out as float32 images."""
    from . import fitsio
    data = mapData.cpu().numpy() if _is_tensor(mapData) else np.asarray(mapData)
    header = dict(wcs.header) if wcs is not None else {}
    if compressionType == 'PLIO_1':
        fitsio.writeMEF(outputFileName, [('COMPRESSED_IMAGE', np.array(data, dtype=np.uint8), header)], compressionType='PLIO_1')
        return
    if compressionType is not None:
        data = np.array(data, dtype=np.float32)
    fitsio.writeFITS(outputFileName, data, header)


class TileDict(dict):
    """Tile images by tile name, to be saved as one multi-extension FITS file or stitched into one image
    (nemo/maps.py:508-614).  Values may be device tensors: the stitching is done where the tiles are."""

    def __init__(self, inputDict, tileCoordsDict=None):
        super(TileDict, self).__init__(inputDict)
        self.tileCoordsDict = tileCoordsDict

    def copy(self):
        return TileDict(self, tileCoordsDict=self.tileCoordsDict)

    def _tileHeader(self, tileName):
        if self.tileCoordsDict[tileName]['reprojectToTan'] == True:
            raise NotImplementedError("reprojectToTan (TAN <-> CAR reprojection of tiles) is outside the hot path")
        return self.tileCoordsDict[tileName]['header']

    def saveMEF(self, outFileName, compressionType=None):
        """nemo/maps.py:539-571: one image extension per tile, named after the tile, with the tile's WCS header."""
        from . import fitsio
        hdus = []
        for tileName in self.keys():
            data = self[tileName].cpu().numpy() if _is_tensor(self[tileName]) else np.asarray(self[tileName])
            if compressionType is not None:
                data = np.array(data, dtype=np.uint8 if compressionType == 'PLIO_1' else np.float32)
            hdus.append((tileName, data, self._tileHeader(tileName)))
        fitsio.writeMEF(outFileName, hdus, compressionType='PLIO_1' if compressionType == 'PLIO_1' else None)

    def stitch(self, stitchedWCS):
        """The monolithic float32 image of nemo/maps.py:592-612: tiles added into their clippedSection (overlaps add up:
        the tiles carry zeros outside their own area)."""
        shape = (stitchedWCS.header['NAXIS2'], stitchedWCS.header['NAXIS1'])
        onDevice = any(_is_tensor(v) for v in self.values())
        if onDevice:
            t = engine.torch()
            d = t.zeros(shape, dtype=t.float32, device=engine.require_gpu())
        else:
            d = np.zeros(shape, dtype=np.float32)
        for tileName in self.keys():
            self._tileHeader(tileName)
            minX, maxX, minY, maxY = self.tileCoordsDict[tileName]['clippedSection']
            tile = self[tileName]
            if onDevice:
                tile = engine.to_dev(tile).to(d.dtype)
            try:
                d[minY:maxY, minX:maxX] = d[minY:maxY, minX:maxX] + tile
            except Exception:
                raise Exception("Stitching error on tile %s" % (tileName))
        return d

    def saveStitchedFITS(self, outFileName, stitchedWCS, compressionType=None):
        """nemo/maps.py:574-614."""
        saveFITS(outFileName, self.stitch(stitchedWCS), stitchedWCS, compressionType=compressionType)


def _decBands(wcs, rows, column, targetTileHeight):
    """A mask region's declination range cut into bands of equal height, as many as fit at the wanted height; the height is
    kept to a thousandth of a degree (nemo/maps.py:760-772).  -> [(band index, decLow, decHigh)]."""
    decLow = wcs.pix2wcs(column, rows.start)[1]
    decHigh = wcs.pix2wcs(column, rows.stop - 1)[1]
    count = int((decHigh - decLow) / targetTileHeight)
    if count == 0:
        raise Exception("targetTileHeight is larger than the height of the map - edit your config file accordingly.")
    height = np.round(((decHigh - decLow) / count) * 1000) / 1000
    return [(i, decLow + i * height, decLow + (i + 1) * height) for i in range(count)]


def _bandColumns(wcs, inRegion, column, decLow, decHigh, targetTileWidth, towardsLowerRA, wrapAt180, lastRADeg):
    """One declination band of a region cut into tiles along RA (nemo/maps.py:773-822): the band's extent in x where the
    region has pixels, widened by 1 / cos(dec) of the band's upper edge, split into equal widths kept to a hundredth of a
    degree (rounded up); all tiles of the band at once.  -> (RAMin[], RAMax[]) per tile."""
    RAHere = wcs.pix2wcs(column, 0)[0]
    yLow, yHigh = int(wcs.wcs2pix(RAHere, decLow)[1]), int(wcs.wcs2pix(RAHere, decHigh)[1])
    occupied = np.flatnonzero(inRegion[yLow:yHigh].any(axis=0))
    xFirst, xLast = occupied[0], occupied[-1]
    yMid = int((yHigh + yLow) / 2)
    widthDeg = (xLast - xFirst) * wcs.getXPixelSizeDeg()
    RAStart = wcs.pix2wcs(xFirst, yMid)[0]
    count = int(widthDeg / (targetTileWidth * (1 / np.cos(np.radians(decHigh)))))
    if count == 0:
        return np.zeros(0), np.zeros(0)
    step = np.ceil((widthDeg / count) * 100) / 100
    j = np.arange(count)
    edgeA = RAStart - towardsLowerRA * j * step              # the tile's edge nearer the band's first column
    edgeB = RAStart - towardsLowerRA * (j + 1) * step
    edgeA = np.where(edgeA < 0, edgeA + 360, edgeA)
    edgeB = np.where(edgeB < 0, edgeB + 360, edgeB)
    if wrapAt180:
        # a tile that would straddle RA = 180 in a whole-sky map stops just past it
        edgeB = np.where((edgeB < 180.01) & (edgeA < 180 + step) & (edgeA > 180.01), 180.01, edgeB)
    if towardsLowerRA == -1:
        edgeB = np.where(edgeB >= lastRADeg, lastRADeg, edgeB)
        swap = edgeB > edgeA
        edgeA, edgeB = np.where(swap, edgeB, edgeA), np.where(swap, edgeA, edgeB)
    return edgeB, edgeA


def autotiler(surveyMask, wcs, targetTileWidth, targetTileHeight):
    """Tile definitions ({tileName, RADecSection}) that cover the valid area of a survey mask with tiles of about the wanted
    size in degrees (what nemo/maps.py:700-825 produces; pinned by tests/golden/autotiler.json): every connected region of
    the mask at least 1000 pixels tall is cut into declination bands of equal height, every band into RA tiles whose width
    grows with 1 / cos(dec).  Tiles are named region_band_column.  Host code (once per run, on a mask)."""
    from scipy import ndimage
    towardsLowerRA = -1 if wcs.isFlipped() == 1 else 1
    centreRA = wcs.getCentreWCSCoords()[0]
    skyWidth = wcs.getFullSizeSkyDeg()[0]
    wrapAt180 = (centreRA < 0.1 and skyWidth < 0.1) or skyWidth > 359.9
    lastRADeg = min(wcs.pix2wcs(surveyMask.shape[1], 0)[0], 359.999)
    labels, count = ndimage.label(np.asarray(surveyMask) > 0)
    tiles = []
    for region, (rows, cols) in enumerate(ndimage.find_objects(labels), start=1):
        if (rows.stop - 1) - rows.start < 1000:
            continue                                     # specks and slivers get no tiles
        column = int((cols.start + (cols.stop - 1)) / 2)
        inRegion = labels == region
        for band, decLow, decHigh in _decBands(wcs, rows, column, targetTileHeight):
            RAMin, RAMax = _bandColumns(wcs, inRegion, column, decLow, decHigh, targetTileWidth, towardsLowerRA, wrapAt180, lastRADeg)
            for j in range(len(RAMin)):
                tiles.append({'tileName': '%d_%d_%d' % (region, band, j),
                              'RADecSection': [float(RAMin[j]), float(RAMax[j]), float(decLow), float(decHigh)]})
    return tiles


def getPixelAreaArcmin2Map(shape, wcs, rowsOnly=False):
    """Pixel area in arcmin^2 (nemo/maps.py:1385-1406): measured per row at the map's central column, float32, the same for
    every pixel of a row.  ``rowsOnly``: the per-row values [H] instead of the map [H, W]."""
    RACentre, decCentre = wcs.getCentreWCSCoords()
    x0, _ = wcs.wcs2pix(RACentre, decCentre)
    ys = np.arange(shape[0], dtype=np.float64)
    ra0, dec0 = wcs.pix2wcs(np.full(shape[0], x0), ys)
    ra1, dec1 = wcs.pix2wcs(np.full(shape[0], x0 + 1), ys + 1)
    xPixScale = calcAngSepDeg(ra0, dec0, ra1, dec0)
    yPixScale = calcAngSepDeg(ra0, dec0, ra0, dec1)
    pixAreasArcmin2 = np.array(xPixScale * yPixScale, dtype=np.float32) * (60 ** 2)
    if rowsOnly:
        return pixAreasArcmin2
    return np.array([pixAreasArcmin2] * shape[1]).transpose()


def pixelScalesDeg(wcs, RADeg='centre', decDeg='centre'):
    """x / y pixel scales (degrees) at the given coordinates, as nemo/maps.py:1367-1377 derives them."""
    ra0, dec0 = wcs.getCentreWCSCoords()
    if RADeg != 'centre':
        ra0 = float(RADeg)
    if decDeg != 'centre':
        dec0 = float(decDeg)
    x0, y0 = wcs.wcs2pix(ra0, dec0)
    ra1, dec1 = wcs.pix2wcs(x0 + 1, y0 + 1)
    xPixScale = calcAngSepDeg(ra0, dec0, ra1, dec0)
    yPixScale = calcAngSepDeg(ra0, dec0, ra0, dec1)
    return xPixScale, yPixScale


def smoothMap(data, wcs, RADeg='centre', decDeg='centre', smoothScaleDeg=5.0 / 60.0):
    """nemo/maps.py:1357-1382: Gaussian smoothing at the given scale (device tensor in, device tensor out)."""
    xPixScale, yPixScale = pixelScalesDeg(wcs, RADeg, decDeg)
    eng = engine.TileEngine.get(data.shape)
    d = engine.to_dev(data)
    return eng.gaussian_filter(d, (smoothScaleDeg / yPixScale, smoothScaleDeg / xPixScale))


def subtractBackground(data, wcs, RADeg='centre', decDeg='centre', smoothScaleDeg=30.0 / 60.0):
    """nemo/maps.py:1279-1292."""
    xPixScale, yPixScale = pixelScalesDeg(wcs, RADeg, decDeg)
    eng = engine.TileEngine.get(data.shape)
    d = engine.to_dev(data)
    return eng.subtract_background(d, (smoothScaleDeg / yPixScale, smoothScaleDeg / xPixScale))


def sourceInjectionTest(config, realCatalog=None, areaMasks=None, verbose=True):
    """Source injection and recovery at the reference filter scale (nemo/maps.py:1856-2153): for every injection model and
    iteration a random catalogue is drawn inside the area masks (catalogs.generateTestCatalog), painted into the maps while
    they are preprocessed (``injectSources``, nemo/maps.py:326-344), the maps are filtered again with the reference filter and
    noise map an earlier pass left behind (``useCachedFilters`` / ``useCachedRMSMap``: no filter build, no noise map), and
    what the finder brings back is matched to what went in.  Returns the table of recovered objects: RADeg, decDeg,
    sourceInjectionModel, theta500Arcmin (clusters), SNR, rArcmin, inFlux, outFlux, noiseLevel, tileName.

    ``realCatalog``: the catalogue of the real run - rows near its objects are left out (:2057-2059); read from
    <rootOutDir>/<name>_optimalCatalog.fits as in the reference when not given.  ``areaMasks``: a catalogs.AreaMasks (made from
    the area masks of the last pass, in memory or under selFn/, when not given).  With several ranks every rank draws the
    objects of its own tiles and the rows are exchanged (the reference draws all of them on rank 0 and broadcasts, :2030-2035).
    Configuration key of this build: ``sourceInjectionReferenceStream`` (default False) - draw the catalogues from ``np.random``
    with the reference's own (much longer) sequence of draws, see catalogs.generateTestCatalog."""
    from . import signals, catalogs, pipelines
    realExclusionRadiusArcmin = 5.0
    if verbose:
        print(">>> Position recovery test [rank = %d] ..." % (config.rank))
    numIterations = config.parDict.get('sourceInjectionIterations', 1)
    if 'sourceInjectionModels' in config.parDict.keys():
        clusterMode = True
        sourceInjectionModelList = config.parDict['sourceInjectionModels']
        SNRCol, fluxCol, noiseLevelCol = 'SNR', 'y_c', 'err_y_c'
        for model in sourceInjectionModelList:
            theta500Arcmin = signals.calcTheta500Arcmin(model['redshift'], model['M500'], signals.fiducialCosmoModel)
            model['label'] = '%.2f' % (theta500Arcmin)
            model['theta500Arcmin'] = theta500Arcmin
    else:
        clusterMode = False
        sourceInjectionModelList = [{'label': 'pointSource'}]
        SNRCol, fluxCol, noiseLevelCol = 'SNR', 'deltaT_c', 'err_deltaT_c'
    numSourcesPerTile = config.parDict.get('sourcesPerTile', 300)

    if realCatalog is None:
        if config.rootOutDir is None:
            raise Exception("source injection test: give realCatalog (the catalogue of the real run) for a run without outputDir")
        catFileName = config.rootOutDir + os.path.sep + "%s_optimalCatalog.fits" % (os.path.split(config.rootOutDir)[-1])
        if os.path.exists(catFileName) == False:
            raise Exception("Catalog file '%s' not found - needed to do source injection test." % (catFileName))
        realCatalog = catalogs.readCatalog(catFileName)
    if areaMasks is None:
        areaMasks = catalogs.AreaMasks.fromConfig(config)

    preprocCache = {}       # per tile: the preprocessed maps before the catalogue steps, the cube the injected maps are written to
    for k, mapDict in enumerate(config.unfilteredMapsDictList):
        mapDict['_preprocCache'], mapDict['_planeIndex'], mapDict['_numPlanes'] = preprocCache, k, len(config.unfilteredMapsDictList)
    collected = {}          # model label -> lists per output column
    cols = ('RADeg', 'decDeg', 'SNR', 'rArcmin', 'inFlux', 'outFlux', 'noiseLevel', 'tileName')
    allInputCatalogs = []
    for modelCount, model in enumerate(sourceInjectionModelList):
        if verbose:
            print(">>> Source injection model: %d/%d" % (modelCount + 1, len(sourceInjectionModelList)))
        got = collected[model['label']] = {k: [] for k in cols}
        for i in range(numIterations):
            if verbose:
                print(">>> Source injection and recovery test %d/%d [rank = %d]" % (i + 1, numIterations, config.rank))
            if 'GNFWParams' not in config.parDict.keys():
                config.parDict['GNFWParams'] = 'default'
            for filtDict in config.parDict['mapFilters']:
                filtDict['params']['GNFWParams'] = config.parDict['GNFWParams']
                for key in ['saveFilteredMaps', 'savePlots']:
                    filtDict['params'][key] = False
            # only the reference filter is run (:1973-1979; without a photFilter the last filter listed is what is left)
            for filtDict in config.parDict['mapFilters']:
                if filtDict['label'] == config.parDict['photFilter']:
                    break
            config.parDict['mapFilters'] = [filtDict]

            if filtDict['class'].find("ArnaudModel") != -1:
                amplitudeRange = config.parDict.get('sourceInjectionAmplitudeRange', [0.001, 10])
                if amplitudeRange == 'auto':
                    amplitudeRange = [np.asarray(realCatalog['fixed_y_c']).min() * 0.5, np.asarray(realCatalog['fixed_y_c']).max()]
                distribution = config.parDict.get('sourceInjectionDistribution', 'linear')
                extra = {'GNFWParams': config.parDict['GNFWParams'], 'profile': 'A10'}
            elif filtDict['class'].find("Beam") != -1:
                amplitudeRange = config.parDict.get('sourceInjectionAmplitudeRange', [1, 1000])
                distribution = config.parDict.get('sourceInjectionDistribution', 'log')
                extra = {'profile': None}
            else:
                raise Exception("Don't know how to generate injected source catalogs for filterClass '%s'" % (filtDict['class']))
            mockCatalog = catalogs.generateTestCatalog(config, numSourcesPerTile, amplitudeColumnName=fluxCol,
                                                       amplitudeRange=amplitudeRange, amplitudeDistribution=distribution,
                                                       selFn=areaMasks, maskDilationPix=20, tileNames=areaMasks.tileNames,
                                                       referenceStream=config.parDict.get('sourceInjectionReferenceStream', False))
            if config.MPIEnabled == True and config.size > 1:
                import torch.distributed as dist
                parts = [None] * config.size
                dist.all_gather_object(parts, {k: np.asarray(mockCatalog[k]) for k in mockCatalog.keys()})
                mockCatalog = catalogs.vstack([catalogs.Table(cols=p) for p in parts if len(p['RADeg']) > 0])
                if len(mockCatalog) > 0:
                    mockCatalog['name'] = np.arange(0, len(mockCatalog)) + 1
            if 'theta500Arcmin' in model.keys() and len(mockCatalog) > 0:
                mockCatalog['theta500Arcmin'] = model['theta500Arcmin']
            injectSources = dict(extra, catalog=mockCatalog, override=model)
            allInputCatalogs.append(mockCatalog)
            for mapDict in config.unfilteredMapsDictList:
                mapDict['injectSources'] = injectSources
            if len(mockCatalog) == 0:
                continue
            recCatalog = pipelines.filterMapsAndMakeCatalogs(config, useCachedFilters=True, useCachedRMSMap=True,
                                                             writeAreaMask=False, writeFlagMask=False, verbose=False)
            if len(recCatalog) > 0:
                recCatalog = catalogs.removeCrossMatched(recCatalog, realCatalog, radiusArcmin=realExclusionRadiusArcmin)
            if len(recCatalog) > 0:
                x_mockCatalog, x_recCatalog, rDeg = catalogs.crossMatch(mockCatalog, recCatalog, radiusArcmin=realExclusionRadiusArcmin)
                # a bright point source recovered far from where it went in (:2068-2084; the reference would write debugging
                # region files there through names it never defines)
                if clusterMode == False and np.logical_and(rDeg > 1.5 / 60, np.asarray(x_recCatalog['SNR']) > 10).sum() > 0:
                    msg = "Caught recovered source at large offset"
                    if config.parDict.get('haltOnPositionRecoveryProblem', False) == True:
                        raise Exception(msg)
                    print("... Warning: %s ..." % (msg))
                got['RADeg'] += np.asarray(x_recCatalog['RADeg']).tolist()
                got['decDeg'] += np.asarray(x_recCatalog['decDeg']).tolist()
                got['SNR'] += np.asarray(x_recCatalog[SNRCol]).tolist()
                got['rArcmin'] += (rDeg * 60).tolist()
                got['inFlux'] += np.asarray(x_mockCatalog[fluxCol]).tolist()
                got['outFlux'] += np.asarray(x_recCatalog[fluxCol]).tolist()
                got['noiseLevel'] += np.asarray(x_recCatalog[noiseLevelCol]).tolist()
                got['tileName'] += np.asarray(x_recCatalog['tileName']).astype(str).tolist()

    models, theta500s = [], []
    out = {k: [] for k in cols}
    for model in sourceInjectionModelList:
        got = collected[model['label']]
        if 'theta500Arcmin' in model.keys():
            theta500s += [model['theta500Arcmin']] * len(got['SNR'])
        models += [model['label']] * len(got['SNR'])
        for k in cols:
            out[k] += got[k]
    resultsTable = catalogs.Table()
    resultsTable['RADeg'] = np.array(out['RADeg'], dtype=np.float64)
    resultsTable['decDeg'] = np.array(out['decDeg'], dtype=np.float64)
    resultsTable['sourceInjectionModel'] = np.array(models, dtype='U30') if models else np.zeros(0, dtype='U30')
    if len(theta500s) == len(resultsTable):
        resultsTable['theta500Arcmin'] = np.array(theta500s, dtype=np.float64)
    for k in ('SNR', 'rArcmin', 'inFlux', 'outFlux', 'noiseLevel'):
        resultsTable[SNRCol if k == 'SNR' else k] = np.array(out[k], dtype=np.float64)
    resultsTable['tileName'] = np.array(out['tileName'], dtype='U30') if out['tileName'] else np.zeros(0, dtype='U30')

    if config.rank == 0 and config.selFnDir is not None:                                   # :2142-2148
        nonEmpty = [c for c in allInputCatalogs if len(c) > 0]
        if len(nonEmpty) > 0:
            allInputTab = catalogs.vstack([c.copy() for c in nonEmpty])
            allInputTab['inFlux'] = allInputTab[fluxCol]
            allInputTab.remove_column(fluxCol)
            allInputTab = catalogs.removeCrossMatched(allInputTab, realCatalog, radiusArcmin=realExclusionRadiusArcmin)
            os.makedirs(config.selFnDir, exist_ok=True)
            catalogs.writeCatalog(allInputTab, config.selFnDir + os.path.sep + "sourceInjectionInputCatalog.fits")
    config.restoreConfig()
    return resultsTable
