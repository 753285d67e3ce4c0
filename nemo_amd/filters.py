"""MI355X-native filter classes with the reference's names and call protocol (nemo/filters.py).

* :func:`filterMaps`                                  - nemo/filters.py:51-106
* :class:`MapFilter` (``makeNoiseMap`` etc.)           - nemo/filters.py:109-513
* :class:`MatchedFilter`                              - nemo/filters.py:516-855
* :class:`RealSpaceMatchedFilter`                     - nemo/filters.py:858-1210
* template mixins and the six concrete classes        - nemo/filters.py:1214-1320

All map-sized arithmetic runs in libnemofilt (hand-written gfx950 kernels, see include/nemofilt.h) on
tensors resident in HBM; this module only sequences the calls in the reference's order, keeps the
reference's dtype behaviour (numpy >= 2 promotion rules: the 'yc' normalisation is a float64 scalar, so the
filtered / S/N maps become float64 at nemo/filters.py:853; in 'uK' mode they stay float32), and holds the
small scalar logic (weights, normalisation, file caching).  There is no CPU fallback.

Returned maps are numpy arrays by default (drop-in); set ``nemo_amd.filters.RETURN_DEVICE_ARRAYS = True`` (the
pipeline driver and bench do) to keep them as torch tensors on the GPU for ``photometry.findObjects``.
"""
import os

import numpy as np

import ctypes

from . import engine, maps, signals, fitsio, _lib
from ._lib import NF_F32, NF_F64, NemoFiltError
from .wcs import calcAngSepDeg

RETURN_DEVICE_ARRAYS = False
# One library call per filter scale (nf_matched_filter_scales) where the configuration allows it; 0 = every statement of
# buildAndApply enqueued from here (same kernels, same results - tests/test_gpu_scale_entry.py compares the two)
_SCALE_ENTRY = os.environ.get('NEMO_AMD_SCALE_ENTRY', '1') != '0'
_scaleEntryStats = {'calls': 0, 'scales': 0}      # library calls made / scales they carried (tests, diagnostics)


def _asnumpy(x, dtype=None):
    if hasattr(x, 'is_cuda'):
        x = x.cpu().numpy()
    if dtype is not None:
        x = np.asarray(x).astype(dtype, copy=False)
    return x


# ------------------------------------------------------------------------------------------------------------
def filterMaps(unfilteredMapsDictList, filterParams, tileName, diagnosticsDir='.', selFnDir='.', verbose=True,
               undoPixelWindow=True, useCachedFilter=False, returnFilter=False, tileCache=None):
    """Builds and applies a filter to the unfiltered map(s) (nemo/filters.py:51-106).

    ``tileCache`` (extension; a dict owned by the caller, one per tile) lets the filter scales of one tile share what does
    not depend on the template: the preprocessed maps / weights / masks in HBM, the apodised data FFTs and the smoothed noise
    covariance.  The reference recomputes those identically for every scale (SURVEY.md headline facts)."""
    filteredMapDict, filterObj = filterMapsGroup(unfilteredMapsDictList, [filterParams], tileName, diagnosticsDir=diagnosticsDir,
                                                 selFnDir=selFnDir, verbose=verbose, undoPixelWindow=undoPixelWindow,
                                                 useCachedFilter=useCachedFilter, tileCache=tileCache)[0]
    if returnFilter == True:
        return filteredMapDict, filterObj
    return filteredMapDict


def filterMapsGroup(unfilteredMapsDictList, filterParamsList, tileName, diagnosticsDir='.', selFnDir='.', verbose=True,
                    undoPixelWindow=True, useCachedFilter=False, tileCache=None, lanes=1, keepFilters=None):
    """filterMaps for several filters of one tile: [(filteredMapDict, filterObj), ...] in the order given.  Consecutive
    matched filters that share their maps and their noise / trim settings (the scales of a filter bank) go to the library in
    one call, dealt to ``lanes`` HIP streams that run side by side (nf_matched_filter_scales); everything else runs filter
    by filter exactly as filterMaps does.  ``keepFilters`` (None = all): the labels whose filter cube the caller will still read
    from the returned object (``filterObj.filt``, an in-memory cached filter); the other scales of such a call apply their filter
    without storing the cube (``saveFilter`` is honoured regardless)."""
    if tileCache is None:
        tileCache = {}          # what the filters of this call share (maps in HBM, data spectra, noise covariance)
    objs = []
    for f in filterParamsList:
        if verbose:
            print("... making filtered map %s" % (f['label'] + "#" + tileName))
        filterClass = _FILTER_CLASSES[f['class']]
        objs.append(filterClass(f['label'], unfilteredMapsDictList, f['params'], tileName=tileName,
                                diagnosticsDir=diagnosticsDir, selFnDir=selFnDir, tileCache=tileCache))
        objs[-1]._verbose = verbose
    results = [None] * len(objs)
    k = 0
    while k < len(objs):
        key = objs[k]._scaleKey(useCachedFilter) if isinstance(objs[k], MatchedFilter) else None
        run = [k]
        if key is not None:
            while run[-1] + 1 < len(objs) and isinstance(objs[run[-1] + 1], MatchedFilter) and \
                    objs[run[-1] + 1]._scaleKey(useCachedFilter) == key:
                run.append(run[-1] + 1)
            dicts = MatchedFilter._runScales([objs[i] for i in run], undoPixelWindow, lanes, keepFilters)
        else:
            dicts = None
        if dicts is None:
            run = [k]
            dicts = [objs[k].buildAndApply(useCachedFilter=useCachedFilter)]
            windowDone = False
        else:
            dicts, windowDone = dicts
        for i, filteredMapDict in zip(run, dicts):
            filteredMapDict['wcs'].header['BUNIT'] = filteredMapDict['mapUnits']
            if 'beamSolidAngle_nsr' in filteredMapDict.keys() and filteredMapDict['beamSolidAngle_nsr'] > 0:
                filteredMapDict['wcs'].header['BEAMNSR'] = filteredMapDict['beamSolidAngle_nsr']
                filteredMapDict['wcs'].header['FREQGHZ'] = filteredMapDict['obsFreqGHz']
            filteredMapDict['wcs'].updateFromHeader()
            # Undo the pixel window (FFT . W^-1 . IFFT), restoring exact zeros (nemo/filters.py:98-101)
            if undoPixelWindow == True and not windowDone:
                d = engine.to_dev(filteredMapDict['data'])
                out = objs[i].engine.undo_pixel_window(d)
                filteredMapDict['data'] = out if RETURN_DEVICE_ARRAYS else _asnumpy(out)
            results[i] = (filteredMapDict, objs[i])
        k = run[-1] + 1
    return results


# ------------------------------------------------------------------------------------------------------------
_templateSlots = {}      # MatchedFilter._runScales: (parameter set, maps) -> the template fields of a scale's record


class _DeviceWeights(dict):
    """fRelWeights whose values are still on the device: {obsFreqGHz: float32(plane value at the peak) / float32(peak)}
    (nemo/filters.py:661-667), formed on first access from one small transfer."""

    def __init__(self, freqs, pending):
        super(_DeviceWeights, self).__init__()
        self._freqs, self._pending = list(freqs), pending          # pending: N plane values, then the peak (float64 tensor)

    def _resolve(self):
        if self._pending is not None:
            h = self._pending.cpu().numpy()
            self._pending = None
            total = np.float32(h[-1])
            for f, v in zip(self._freqs, h[:-1]):
                dict.__setitem__(self, f, np.float32(v) / total)
        return self

    def __getitem__(self, k):
        return dict.__getitem__(self._resolve(), k)

    def __iter__(self):
        return dict.__iter__(self._resolve())

    def __len__(self):
        return dict.__len__(self._resolve())

    def __contains__(self, k):
        return dict.__contains__(self._resolve(), k)

    def keys(self):
        return dict.keys(self._resolve())

    def values(self):
        return dict.values(self._resolve())

    def items(self):
        return dict.items(self._resolve())

    def get(self, k, default=None):
        return dict.get(self._resolve(), k, default)

    def __repr__(self):
        return dict.__repr__(self._resolve())

    def __eq__(self, other):
        return dict.__eq__(self._resolve(), other)

    def copy(self):
        return dict(self._resolve())


class MapFilter(object):
    """Base class defining the common interface (nemo/filters.py:109-513)."""

    templateKind = None

    def __init__(self, label, unfilteredMapsDictList, paramsDict, tileName='PRIMARY', writeFilter=False,
                 forceRebuild=False, diagnosticsDir=None, selFnDir=None, tileCache=None):
        self.label = label
        self._tileCache = tileCache
        self.params = paramsDict
        self.diagnosticsDir = diagnosticsDir
        self.selFnDir = selFnDir
        self.tileName = tileName
        if self.diagnosticsDir is not None:
            self.filterFileName = self.diagnosticsDir + os.path.sep + self.tileName + os.path.sep + \
                "filter_%s#%s.fits" % (self.label, self.tileName)
        else:
            self.filterFileName = None

        # Copies, preprocessed per tile (nemo/filters.py:153-161); shared between the scales of a tile via tileCache
        shared = None
        if tileCache is not None and not isinstance(self, RealSpaceMatchedFilter):
            shared = tileCache.setdefault(('tile', tileName, self.params.get('mapToUse', None)), {})
        self._shared = shared
        # the second and later filters of a tile (the scales of a bank) are the first one's state with another label and
        # parameter set: everything below depends on the tile and the maps only
        snap = shared.get(('initSnapshot', type(self))) if shared is not None else None
        if snap is not None:
            self.__dict__.update(snap)
            self.label, self.params, self._tileCache = label, paramsDict, tileCache
            self.diagnosticsDir, self.selFnDir, self.tileName = diagnosticsDir, selFnDir, tileName
            if self.diagnosticsDir is not None:
                self.filterFileName = self.diagnosticsDir + os.path.sep + self.tileName + os.path.sep + \
                    "filter_%s#%s.fits" % (self.label, self.tileName)
            else:
                self.filterFileName = None
            self._shared = shared
            self.signalNorm = 1.0
            self.fRelWeights = {}
            return
        if shared is not None and 'maps' in shared:
            self.unfilteredMapsDictList = shared['maps']
        else:
            self.unfilteredMapsDictList = []
            for mapDict in unfilteredMapsDictList:
                if 'mapToUse' in self.params.keys() and mapDict['label'] != self.params['mapToUse']:
                    continue
                newMapDict = mapDict.copy()
                newMapDict.preprocess(tileName=tileName, diagnosticsDir=diagnosticsDir)
                self.unfilteredMapsDictList.append(newMapDict)
            if shared is not None:
                shared['maps'] = self.unfilteredMapsDictList
        self.wcs = self.unfilteredMapsDictList[0]['wcs']
        self.shape = tuple(self.unfilteredMapsDictList[0]['data'].shape)

        # Combined flag mask (nemo/filters.py:165-168)
        # (held on the device as int32; the numpy form of the reference is made on demand by the flagMask property)
        t = engine.torch()
        if shared is not None and 'dFlagMask' in shared:
            self._dFlagMask = shared['dFlagMask']
        else:
            # (an injection run hands over the same flag masks pass after pass - maps.MapDict's '_preprocCache': their sum is kept)
            runCache = self.unfilteredMapsDictList[0].get('_preprocCache', None)
            fkey = ('flagSum', tileName) + tuple(id(m['flagMask']) for m in self.unfilteredMapsDictList)
            held = runCache.get(fkey) if runCache is not None else None
            if held is not None:
                self._dFlagMask, flagAny = held[0], held[1]
            else:
                self._dFlagMask = None
                for mapDict, i in zip(self.unfilteredMapsDictList, range(len(self.unfilteredMapsDictList))):
                    term = engine.to_dev(mapDict['flagMask']).to(t.int32) * (i + 1)
                    self._dFlagMask = term if self._dFlagMask is None else self._dFlagMask + term
                # "does the tile have any flagged pixel": asked once per tile on the device, read by photometry.findObjects
                # together with its first fetch - a tile without flags needs no flag lookups at the objects
                flagAny = (self._dFlagMask, (self._dFlagMask != 0).any().to(t.float64).reshape(1))
                if runCache is not None and all(hasattr(m['flagMask'], 'is_cuda') for m in self.unfilteredMapsDictList):
                    runCache[fkey] = (self._dFlagMask, flagAny, [m['flagMask'] for m in self.unfilteredMapsDictList])
            if shared is not None:
                shared['dFlagMask'] = self._dFlagMask
                shared['flagAny'] = flagAny
        self._flagMaskHost = None

        # Beam solid angles in nanosteradians (nemo/filters.py:170-189)
        self.beamSolidAnglesDict = {}
        for mapDict in self.unfilteredMapsDictList:
            if 'solidAngle_nsr' in mapDict.keys():
                solidAngle_nsr = mapDict['solidAngle_nsr']
            else:
                solidAngle_nsr = 0.0
                beamFileName = mapDict.get('beamFileName')
                if isinstance(beamFileName, str):
                    with open(beamFileName, "r") as inFile:
                        for line in inFile.readlines():
                            if line.find("solid angle") != -1:
                                bits = line.split("=")
                                solidAngle_nsr = float(bits[1].split()[0])
                                break
            self.beamSolidAnglesDict[mapDict['obsFreqGHz']] = solidAngle_nsr

        self.apodPix = 20
        for mapDict in self.unfilteredMapsDictList:
            if tuple(mapDict['data'].shape) != self.shape:
                raise Exception("Maps at different frequencies have different dimensions!")

        self.makeRadiansMap()
        self.signalNorm = 1.0
        self.fRelWeights = {}

        # device residency: one engine per tile shape, maps / weights / masks uploaded once
        self.engine = engine.TileEngine.get(self.shape)
        if shared is not None and 'dev' in shared:
            self._dData, self._dWeights, self._dSurveyMask, self._dPSMask = shared['dev']
        else:
            self._dData = [engine.to_dev(m['data']) for m in self.unfilteredMapsDictList]
            for d in self._dData:
                if d.dtype not in (t.float32, t.float64):
                    raise NemoFiltError("maps must be float32 or float64")
            self._dWeights = [engine.to_dev(m['weights']) for m in self.unfilteredMapsDictList]
            self._dWeights = [w if w.dtype in (t.float32, t.float64) else w.to(t.float64) for w in self._dWeights]
            self._dSurveyMask = engine.to_dev(self.unfilteredMapsDictList[0]['surveyMask'], dtype=t.uint8)
            self._dPSMask = engine.to_dev(self.unfilteredMapsDictList[0]['pointSourceMask'], dtype=t.uint8)
            if shared is not None:
                shared['dev'] = (self._dData, self._dWeights, self._dSurveyMask, self._dPSMask)
        self._geom = None
        self._dFilt = None
        if shared is not None:
            # (without the references back to the tile cache: a cycle would keep the tile's tensors until the cyclic collector runs)
            shared[('initSnapshot', type(self))] = {k: v for k, v in self.__dict__.items() if k not in ('_shared', '_tileCache')}

    # -- geometry ---------------------------------------------------------------------------------------------
    def makeRadiansMap(self):
        """nemo/filters.py:211-236 (the same for every filter of a tile: kept in the tile cache)."""
        if self._shared is not None and 'radians' in self._shared:
            self.degPerPixX, self.degPerPixY, self._xRad, self._yRad = self._shared['radians']
            self._radiansMap = None
            return
        H, W = self.shape
        x0 = int(W / 2)
        y0 = int(H / 2)
        ra0, dec0 = self.wcs.pix2wcs(x0, y0)
        ra1, dec1 = self.wcs.pix2wcs(x0 + 1, y0 + 1)
        self.degPerPixX = calcAngSepDeg(ra0, dec0, ra1, dec0)
        self.degPerPixY = calcAngSepDeg(ra0, dec0, ra0, dec1)
        xRad = np.arange(int(np.floor(-W / 2)), int(W / 2), dtype=np.float64) * np.radians(self.degPerPixX)
        yRad = np.arange(int(np.floor(-H / 2)), int(H / 2), dtype=np.float64) * np.radians(self.degPerPixY)
        self._xRad, self._yRad = xRad, yRad
        self._radiansMap = None
        if self._shared is not None:
            self._shared['radians'] = (self.degPerPixX, self.degPerPixY, xRad, yRad)

    @property
    def radiansMap(self):
        if self._radiansMap is None:
            self._radiansMap = np.sqrt(self._xRad[None, :] ** 2 + self._yRad[:, None] ** 2)
        return self._radiansMap

    @property
    def signalNorm(self):
        """Filter normalisation as the reference holds it (a numpy / Python float).  During a build it may still be on
        the device (engine.DeviceScalar); reading it here waits for the GPU once."""
        v = self._signalNorm
        return v.value if isinstance(v, engine.DeviceScalar) else v

    @signalNorm.setter
    def signalNorm(self, value):
        self._signalNorm = value

    @property
    def flagMask(self):
        """Combined flag mask as the reference holds it: numpy int array [H, W] (nemo/filters.py:165-168)."""
        if self._flagMaskHost is None:
            self._flagMaskHost = _asnumpy(self._dFlagMask).astype(int)
        return self._flagMaskHost

    @flagMask.setter
    def flagMask(self, value):
        self._flagMaskHost = np.asarray(value)
        self._dFlagMask = engine.to_dev(self._flagMaskHost).to(engine.torch().int32)

    @property
    def filt(self):
        """Fourier-space filter cube, float32 [N, H, W] (numpy view of the device cube)."""
        return None if self._dFilt is None else _asnumpy(self._dFilt)

    @filt.setter
    def filt(self, value):
        self._dFilt = None if value is None else engine.to_dev(np.asarray(value, dtype=np.float32))

    def buildAndApply(self):
        raise Exception("Called a base filter class without a buildAndApply() function implemented.")

    # -- templates --------------------------------------------------------------------------------------------
    def _beamOf(self, beamFileName):
        return signals._asBeam(beamFileName)

    def _paintTemplate(self, beamFileName, amplitude=None):
        """Device float32 template centred on the tile centre (template mixins, nemo/filters.py:1214-1266)."""
        if self.templateKind is None:
            raise Exception("Called a base filter class without a makeSignalTemplateMap() function implemented.")
        self._ensureGeom()
        table = signals.templateTables(self.templateKind, self._beamOf(beamFileName), self.params, amplitude=amplitude)
        return self.engine.paint(self._geom, table)

    def _ensureGeom(self):
        if self._geom is None:
            # the haversine tables of the tile centre are the same for every filter scale of the tile
            shared = self._shared
            if shared is not None and 'paintGeom' in shared:
                self._geom = shared['paintGeom']
            else:
                RADeg, decDeg = self.wcs.getCentreWCSCoords()
                self._geom = signals.paintGeometry(self.shape, self.wcs, RADeg, decDeg)
                if shared is not None:
                    shared['paintGeom'] = self._geom

    def _unitTemplate(self, mapDict, y0):
        """The unit template's radial table for ``mapDict`` and - with ``y0`` - the factor that turns the painted unit
        template into the y0-amplitude template of the normalisation (nemo/filters.py:639-646).  Both templates paint the same
        beam-convolved profile: amp * prof(theta), once with amp = 1 on the profile as it is, once normalised to its central
        value and multiplied by float32(centre * amplitude) (nemo/signals.py:662-664).  Pixel for pixel they differ by the
        constant factor below up to the float32 rounding of the painted values (6e-8), so the scaled template's spectrum is
        the unit template's times that factor and is not computed a second time."""
        if self.templateKind is None:
            raise Exception("Called a base filter class without a makeSignalTemplateMap() function implemented.")
        beam = self._beamOf(mapDict['beamFileName'])
        t1 = signals.templateTables(self.templateKind, beam, self.params, amplitude=None)
        if y0 is None:
            return t1, 1.0
        amp = y0 if mapDict['units'] == 'yc' else maps.convertToDeltaT(y0, mapDict['obsFreqGHz'])
        t2 = signals.templateTables(self.templateKind, beam, self.params, amplitude=amp)
        return t1, (t2.sign * t2.amp * t2.prof[0]) / (t1.sign * t1.amp * t1.prof[0])

    def makeSignalTemplateMap(self, beamFileName, amplitude=None):
        out = self._paintTemplate(beamFileName, amplitude=amplitude)
        return out if RETURN_DEVICE_ARRAYS else _asnumpy(out)

    # -- noise ------------------------------------------------------------------------------------------------
    def makeNoiseMap(self, mapData):
        """Local RMS map of a filtered map (nemo/filters.py:342-480); float32."""
        d = engine.to_dev(mapData)
        rms = self._noiseMapDevice(d)
        return rms if hasattr(mapData, 'is_cuda') else _asnumpy(rms)

    def _medWeights(self):
        if self._shared is not None and 'medWeights' in self._shared:
            return self._shared['medWeights']
        medWeights = self.engine.median_weights(self._dWeights)
        if self._shared is not None:
            self._shared['medWeights'] = medWeights
            self._shared['noiseTablesToken'] = object()      # stands for these weights (nf_noise_map's kept block tables)
        return medWeights

    def _noiseMapDevice(self, dMap):
        return self.engine.noise_map(dMap, self._medWeights(), self.params['noiseParams'], self.wcs.getPixelSizeDeg())

    def makeRealSpaceFilterProfile(self):
        """nemo/filters.py:279-301: 1-D real-space profile of the Fourier filter along +x from the centre."""
        realSpace = self._realSpaceFilter()
        x0 = int(realSpace.shape[2] / 2)
        y0 = int(realSpace.shape[1] / 2)
        normFactor = abs(realSpace[:, y0, x0:]).max()
        prof = realSpace[:, y0, x0:] / normFactor
        arcminRange = np.arange(0, prof.shape[1]) * self.degPerPixX * 60.0
        return prof, arcminRange

    def _realSpaceFilter(self):
        """fftshift(ifft2(filt).real), float32 [N, H, W] (ifft2 of a real cube = conj(fft2)/n).

        The reference calls ``fft.fftshift`` without ``axes`` (nemo/filters.py:288,958), which also rolls the FREQUENCY axis
        by N // 2: with two maps the planes of the real-space cube - and so of ``kern2d`` and of the 1-D profile - come out
        swapped.  Kept as it is (found by the fixtures made from the reference's own class code)."""
        F = self.engine.fft(self._dFilt, prec=NF_F32, unitary=False)
        n = np.float32(self.shape[0] * self.shape[1])
        real = _asnumpy(F.real.contiguous()) / n
        return np.fft.fftshift(real)

    def loadFRelWeights(self):
        """nemo/filters.py:483-495."""
        data, header = fitsio.readFITS(self.filterFileName, headerOnly=True)
        self.fRelWeights = {}
        for i in range(1, 10):
            if 'RW%d_GHZ' % (i) in header.keys():
                self.fRelWeights[header['RW%d_GHZ' % (i)]] = header['RW%d' % (i)]

    # -- shared tail of buildAndApply ---------------------------------------------------------------------------
    def _unitsInfo(self, allowMultiFreqUK=True):
        if self.params['outputUnits'] == 'yc':
            return 'yc', 'yc', 0.0
        elif self.params['outputUnits'] == 'uK':
            if not allowMultiFreqUK and len(self.unfilteredMapsDictList) > 1:
                raise Exception("multi-frequency filtering not currently supported for outputUnits 'uK' (point source finding)")
            combinedObsFreqGHz = float(list(self.beamSolidAnglesDict.keys())[0])
            return 'uK', combinedObsFreqGHz, self.beamSolidAnglesDict[combinedObsFreqGHz]
        raise Exception('need to specify "outputUnits" ("yc" or "uK") in filter params')

    def _finish(self, dFiltered, trimSizePix, fourierVariant):
        """Noise map, S/N, edge trim, masks (nemo/filters.py:716-756 / :1111-1157); all on the device."""
        eng = self.engine
        edge = None
        if trimSizePix > 0:
            edge = eng.edge_check(dFiltered, self._dPSMask, trimSizePix)
        cachedRMS = self._tileCache.get('cachedRMS', None) if self._tileCache is not None else None
        if cachedRMS is not None:
            # The driver replaces the S/N map of this pass with filtered / cached noise map and never reads the pass's own noise
            # map (nemo/pipelines.py:234-250, the re-filters of maps.sourceInjectionTest): that S/N map is made here, together
            # with the masks, and no noise map is computed; 'RMSMap' of the result is the cached one.
            data, sn, mask, zeroRows = eng.finalize_cached(dFiltered, cachedRMS, self._dSurveyMask, self._dPSMask, edge, fourierVariant)
            self._snChunks = eng.lastSNChunks
            self._cachedTail = zeroRows
            return data, sn, cachedRMS, mask, trimSizePix > 0
        rms = self._noiseMapDevice(dFiltered)
        data, sn, rmsOut, mask = eng.finalize(dFiltered, rms, self._dSurveyMask, self._dPSMask, edge, fourierVariant)
        self._snChunks = eng.lastSNChunks
        return data, sn, rmsOut, mask, trimSizePix > 0

    def _packResult(self, data, sn, rms, mask, trimmed, fourierVariant, mapUnits, combinedObsFreqGHz, beamSolidAngle_nsr):
        if RETURN_DEVICE_ARRAYS:
            out = {'data': data, 'SNMap': sn, 'RMSMap': rms, 'surveyMask': mask}
            hint = getattr(self, '_snChunks', None)
            if hint is not None and hint[0] is sn:
                out['_snChunks'] = hint          # (S/N tensor, its per-chunk range table): photometry.findObjects' shortcut
            if self._shared is not None and 'flagAny' in self._shared and self._shared['flagAny'][0] is self._dFlagMask:
                out['_flagAny'] = self._shared['flagAny']
        else:
            # dtypes as the reference produces them under numpy >= 2: the masks become float through
            # edgeCheck (float32 if a rank filter ran in the Fourier variant, else float64)
            mdt = np.float32 if (trimmed and fourierVariant) else np.float64
            out = {'data': _asnumpy(data).astype(np.result_type(_asnumpy(data).dtype, mdt)),
                   'SNMap': _asnumpy(sn).astype(np.result_type(_asnumpy(sn).dtype, mdt)),
                   'RMSMap': _asnumpy(rms).astype(np.result_type(np.float32, mdt)),
                   'surveyMask': _asnumpy(mask).astype(mdt)}
        if getattr(self, '_cachedTail', None) is not None:
            out['_snFromCachedRMS'], out['_zeroRows'] = True, self._cachedTail
        out.update({'wcs': self.wcs, 'obsFreqGHz': combinedObsFreqGHz,
                    'flagMask': self._dFlagMask if RETURN_DEVICE_ARRAYS else self.flagMask, 'mapUnits': mapUnits,
                    'beamSolidAngle_nsr': beamSolidAngle_nsr, 'label': self.label, 'tileName': self.tileName})
        return out


# ------------------------------------------------------------------------------------------------------------
class MatchedFilter(MapFilter):
    """Multi-frequency matched filter in Fourier space (nemo/filters.py:516-855)."""

    def _dataPrec(self):
        t = engine.torch()
        return NF_F64 if self._dData[0].dtype == t.float64 else NF_F32

    def _dataSpectra(self):
        """Half-stored spectra of the apodised maps (nemo/filters.py:523-526), shared by the scales of a tile."""
        t = engine.torch()
        N = len(self.unfilteredMapsDictList)
        shared = self._shared
        if shared is not None and 'fMapsToFilter' in shared:
            return shared['fMapsToFilter']
        views = [m.get('_cubeView', None) for m in self.unfilteredMapsDictList]
        if all(v is not None and v[0] is views[0][0] and v[1] == i for i, v in enumerate(views)) and views[0][0].shape[0] == N:
            dCube = views[0][0]            # an injection run wrote this pass's maps into one cube already (maps.MapDict)
        else:
            dCube = t.stack(self._dData).contiguous() if N > 1 else self._dData[0].reshape(1, *self.shape)
        fMapsToFilter = self.engine.fft(dCube, apod=True, half=True)                      # :523-526  K1+K2
        if shared is not None:
            shared['fMapsToFilter'] = fMapsToFilter
        return fMapsToFilter

    def _noiseCovariance(self, fMapsToFilter):
        """Smoothed noise covariance of the maps (nemo/filters.py:539-580, method 'dataMap'), shared by the scales of a tile."""
        eng = self.engine
        t = engine.torch()
        shared = self._shared
        method = self.params['noiseParams']['method']
        if method != 'dataMap':
            if method == 'model':
                raise NotImplementedError("noise method 'model' (simulated CMB + white noise) is not part of this build yet")
            # 'max(dataMap,CMB)' included: the reference's branch for it (nemo/filters.py:574-577) is unreachable, its
            # loop over the maps raises this for every method but 'dataMap' and 'model' (:539-561; pinned by
            # tests/golden/matched_filter.npz 'maxDataMapCMB_raises')
            raise Exception("'%s' is not a valid filter noise method name - fix the .yml config file" % (method))
        noiseCats = self.params.get('noiseModelCatalog', None)
        covKey = ('noiseCov', method, None if noiseCats is None else id(noiseCats))
        if shared is not None and covKey in shared:
            return shared[covKey]
        fMapsForNoise = fMapsToFilter                                     # identical to :548 without a catalog
        if noiseCats is not None:                                         # :540-547
            assert(type(noiseCats) == list)
            planes = []
            for i, mapDict in enumerate(self.unfilteredMapsDictList):
                d = self._dData[i]
                for noiseModelCatalog in noiseCats:
                    model = maps.makeModelImage(self.shape, self.wcs, noiseModelCatalog, mapDict['beamFileName'],
                                                obsFreqGHz=mapDict['obsFreqGHz'])
                    if model is not None:
                        d = d - model
                planes.append(d)
            fMapsForNoise = eng.fft(t.stack(planes).contiguous(), apod=True, half=True)
            del planes
        NP = eng.cross_power(fMapsForNoise, half=True)                            # :573  K4
        noiseCov = eng.gaussian_filter(NP, (3, 3))                        # :580  K5
        del NP, fMapsForNoise
        if shared is not None:
            shared[covKey] = noiseCov
        return noiseCov

    # -- one library call per scale (nf_matched_filter_scales) -------------------------------------------------------
    def _trimSizePix(self):
        if 'edgeTrimArcmin' in self.params.keys() and self.params['edgeTrimArcmin'] > 0:      # :725-733
            return int(round((self.params['edgeTrimArcmin'] / 60.) / self.wcs.getPixelSizeDeg()))
        return 0

    def _scaleKey(self, useCachedFilter):
        """Hashable description of everything but the template that a library-side run of buildAndApply depends on - filters
        with equal keys can share one nf_matched_filter_scales call - or None if this configuration is enqueued statement by
        statement (cached filters, background subtraction, 'smart' noise grid ...)."""
        p = self.params
        if not _SCALE_ENTRY or self.templateKind is None or self._shared is None or useCachedFilter != False:
            return None
        if self.filterFileName is not None and os.path.exists(self.filterFileName):
            return None
        npar = p.get('noiseParams', {})
        if npar.get('method', None) != 'dataMap' or p.get('outputUnits', None) not in ('yc', 'uK'):
            return None
        if 'bckSub' in p.keys() and 'bckSubScaleArcmin' in p.keys() and p['bckSub'] == True:
            return None
        if npar.get('noiseGridArcmin', None) == 'smart' or npar.get('RMSEstimator', None) not in (None, 'percentile'):
            return None
        if len(self.unfilteredMapsDictList) > _lib.NF_MF_MAXN:
            return None          # (the statement path then fails in nf_build_filter with its own '1 <= N <= 6' message)
        noiseCats = p.get('noiseModelCatalog', None)
        # (the frequency weights depend on the tile's maps, the output units and alpha: once per tile for a bank's scales)
        wkey = ('freqWeightsKey', p['outputUnits'], p.get('alpha', None))
        wts = self._shared.get(wkey)
        if wts is None:
            wts = self._shared[wkey] = tuple(float(v) for v in self._frequencyWeights())
        return (id(self._shared), type(self).templateKind, p['outputUnits'], npar.get('noiseGridArcmin', None), npar.get('numNoiseBins', 1),
                npar.get('RMSEstimator', None), self._trimSizePix(), None if noiseCats is None else id(noiseCats), wts)

    @staticmethod
    def _runScales(objs, undoPixelWindow, lanes, keepFilters=None):
        """buildAndApply of every filter in ``objs`` (equal _scaleKey) through nf_matched_filter_scales.  Returns
        ([filteredMapDict, ...], pixel window undone?) or None if the tile has no noise cell (the caller falls back)."""
        first = objs[0]
        eng = first.engine
        t = engine.torch()
        H, W = first.shape
        N = len(first.unfilteredMapsDictList)
        yc = first.params['outputUnits'] == 'yc'
        y0 = 2e-4
        grid = eng.noise_grid(first.params['noiseParams'], first.wcs.getPixelSizeDeg())
        if grid is None:
            return None
        ncy, ncx, yE, xE, overlap, nb, quant, est = grid
        fMaps = first._dataSpectra()
        noiseCov = first._noiseCovariance(fMaps)
        medWeights = first._medWeights()
        first._ensureGeom()
        A, B, C = eng._geomDev(first._geom)
        tile = _lib.MfTile()
        tile.H, tile.W, tile.N = H, W, N
        tile.prec = engine.prec_of(fMaps)
        tile.covType = engine.prec_of(noiseCov)
        tile.outputUnits = 0 if yc else 1
        tile.fMaps_d, tile.noiseCov_d = fMaps.data_ptr(), noiseCov.data_ptr()
        for i, v in enumerate(first._frequencyWeights()):
            tile.w[i] = float(v)
        tile.A_d, tile.B_d, tile.C_d = A.data_ptr(), B.data_ptr(), C.data_ptr()
        tile.psMask_d = engine.ptr(first._dPSMask)
        tile.surveyMask_d = engine.ptr(first._dSurveyMask)
        tile.medWeights_d, tile.wtype = medWeights.data_ptr(), engine.prec_of(medWeights)
        tile.ny, tile.nx, tile.yEdges_d, tile.xEdges_d = ncy, ncx, yE.data_ptr(), xE.data_ptr()
        tile.overlap, tile.numBins, tile.estimator = overlap, nb, est
        tile.quantiles = ctypes.cast(quant, ctypes.POINTER(ctypes.c_double))
        tile.trimSizePix = first._trimSizePix()
        tile.apodPix = eng.APOD_PIX
        tile.y0 = y0
        tile.halfWin = 32
        keep = [quant]
        if yc:
            shared = first._shared
            if 'centrePoint' not in shared:
                cRADeg, cDecDeg = first.wcs.getCentreWCSCoords()
                cx, cy = first.wcs.wcs2pix(cRADeg, cDecDeg)
                shared['centrePoint'] = (eng.spline_points([[cy, cx]], halfWin=32), first._centreRows(cy))
            pts, (tile.centreRowLo, tile.centreRowHi) = shared['centrePoint']
            tile.centreYX_d, tile.centreClass = pts.dev.data_ptr(), pts.pointsClass
            wy, wx = eng.window(1.0)
            tile.winY_d, tile.winX_d = wy.data_ptr(), wx.data_ptr()
        windowDone = False
        if undoPixelWindow == True:
            tabs = eng._axis_window(-1.0, NF_F64 if yc else NF_F32) if engine._AXIS_WINDOW else None
            if tabs is not None:
                tile.axisWinX_d, tile.axisWinY_d = tabs[0].data_ptr(), tabs[1].data_ptr()
                windowDone = True

        G = len(objs)
        odt = t.float64 if yc else t.float32
        wantCube = [keepFilters is None or obj.label in keepFilters or obj.params.get('saveFilter', False) == True for obj in objs]
        cubes = [t.empty((N, H, W), dtype=t.float32, device=eng.dev) if want else None for want in wantCube]
        data = t.empty((G, H, W), dtype=odt, device=eng.dev)
        sn = t.empty((G, H, W), dtype=odt, device=eng.dev)
        rms = t.empty((G, H, W), dtype=t.float32, device=eng.dev)
        mask = t.empty((G, H, W), dtype=t.uint8, device=eng.dev)
        scal = t.empty((G, N + 2), dtype=t.float64, device=eng.dev)       # per scale: signalNorm, then the N + 1 weight operands
        nch = 2 * ((H * W + 1023) // 1024)
        snChunks = t.empty((G, nch), dtype=t.int64, device=eng.dev)       # per scale: S/N range per 1024-pixel chunk
        scales = (_lib.MfScale * G)()
        es = data.element_size()
        # the template part of a scale's record depends on the filter's parameters and the maps' beams / bands, not on the
        # tile: kept per (parameter set, maps) for the tiles that follow (_templateSlots holds the device tables alive)
        mapsKey = tuple((m['beamFileName'] if isinstance(m['beamFileName'], str) else id(m['beamFileName']), m['units'], m['obsFreqGHz'])
                        for m in first.unfilteredMapsDictList)
        for g, obj in enumerate(objs):
            sc = scales[g]
            p = obj.params
            slotKey = (id(p), type(obj).templateKind, yc, p.get('z'), p.get('M500MSun'), id(p.get('GNFWParams')), mapsKey,
                       tuple(id(m) for m in obj.unfilteredMapsDictList) if obj.unfilteredMapsDictList is not first.unfilteredMapsDictList else None)
            slot = _templateSlots.get(slotKey)
            if slot is None or slot[0] is not p:
                rows = []
                for i, mapDict in enumerate(obj.unfilteredMapsDictList):
                    t1, scale = obj._unitTemplate(mapDict, y0 if yc else None)
                    r, prof, slope = eng._tableDev(t1)
                    rows.append((r.data_ptr(), prof.data_ptr(), slope.data_ptr(), len(t1.r), t1.r0, t1.dlnr, t1.amp, t1.sign, scale, t1))
                if len(_templateSlots) >= 4096:
                    _templateSlots.clear()
                slot = _templateSlots[slotKey] = (p, rows)
            for i, row in enumerate(slot[1]):
                (sc.r_d[i], sc.prof_d[i], sc.slope_d[i], sc.ntab[i], sc.r0[i], sc.dlnr[i], sc.amp[i], sc.sign[i],
                 sc.templateScale[i]) = row[:9]
            sc.filt_d = cubes[g].data_ptr() if cubes[g] is not None else None
            sc.signalNorm_d = scal.data_ptr() + g * (N + 2) * 8
            sc.relWeights_d = scal.data_ptr() + (g * (N + 2) + 1) * 8
            sc.data_d = data.data_ptr() + g * H * W * es
            sc.sn_d = sn.data_ptr() + g * H * W * es
            sc.rms_d = rms.data_ptr() + g * H * W * 4
            sc.mask_d = mask.data_ptr() + g * H * W
            sc.snChunkRange_d = snChunks.data_ptr() + g * nch * 8
        # the noise block tables a lane's workspace holds stay good for this tile's weights and grid (one token per tile cache)
        noiseKey = None
        if first._shared is not None and 'noiseTablesToken' in first._shared:
            noiseKey = (first._shared['noiseTablesToken'], H, W, ncy, ncx, overlap, nb, est)
        eng.matched_filter_scales(tile, scales, lanes, noiseKey=noiseKey, weights=medWeights)
        del keep
        _scaleEntryStats['calls'] += 1
        _scaleEntryStats['scales'] += G

        out = []
        for g, obj in enumerate(objs):
            obj._dFilt = cubes[g]
            obj._signalNorm = engine.DeviceScalar(scal[g, 0:1], wide=yc)
            obj.fRelWeights = {}
            if yc:
                if N == 1:
                    obj.fRelWeights[obj.unfilteredMapsDictList[0]['obsFreqGHz']] = np.float32(1.0)
                else:
                    obj.fRelWeights = _DeviceWeights([m['obsFreqGHz'] for m in obj.unfilteredMapsDictList], scal[g, 1:N + 2])
            mapUnits, combinedObsFreqGHz, beamSolidAngle_nsr = obj._unitsInfo()
            obj._saveFilterIfAsked()
            snG = sn[g]
            obj._snChunks = (snG, snChunks[g])
            out.append(obj._packResult(data[g], snG, rms[g], mask[g], tile.trimSizePix > 0, True, mapUnits, combinedObsFreqGHz,
                                       beamSolidAngle_nsr))
        return out, windowDone

    def _saveFilterIfAsked(self):
        if 'saveFilter' in self.params and self.params['saveFilter'] == True and self.filterFileName is not None:
            header = {'SIGNORM': float(self.signalNorm)}                          # :761-772
            count = 0
            for key in list(self.fRelWeights.keys()):
                count = count + 1
                header['RW%d_GHZ' % (count)] = key
                header['RW%d' % (count)] = float(self.fRelWeights[key])
            os.makedirs(os.path.split(self.filterFileName)[0], exist_ok=True)
            fitsio.writeFITS(self.filterFileName, self.filt, header)

    def buildAndApply(self, useCachedFilter=False):
        eng = self.engine
        t = engine.torch()
        N = len(self.unfilteredMapsDictList)
        fMapsToFilter = self._dataSpectra()

        haveFile = self.filterFileName is not None and os.path.exists(self.filterFileName)
        if useCachedFilter == True and haveFile == False and self._dFilt is None and self._tileCache is not None:
            cached = self._tileCache.get(('cachedFilter', self.label), None)
            if cached is not None and tuple(cached.shape) == tuple(self.shape):
                self._dFilt, self._signalNorm, self.fRelWeights = cached._dFilt, cached._signalNorm, dict(cached.fRelWeights.items())
                cached._filtFor(True)                    # (the symmetrised form is made once, on the kept object)
                self._dFiltSymOf = cached._dFiltSymOf
        if haveFile == False and useCachedFilter == False:
            noiseCov = self._noiseCovariance(fMapsToFilter)

            w = self._frequencyWeights()                                          # :587-608
            y0 = 2e-4
            yc = self.params['outputUnits'] == 'yc'
            # unit templates (:611-618), painted side by side and transformed together; in 'yc' mode the y0-scaled templates of
            # the normalisation (:639-646) are these times a constant (see _unitTemplate): their spectra are not computed again
            templates = t.empty((N,) + tuple(self.shape), dtype=t.float32, device=eng.dev)
            self._ensureGeom()
            templateScale = []
            for i, mapDict in enumerate(self.unfilteredMapsDictList):
                table, scale = self._unitTemplate(mapDict, y0 if yc else None)
                eng.paint(self._geom, table, out=templates[i])
                templateScale.append(scale)
            fSignals = eng.fft(templates, half=True)
            del templates
            self._dFilt, sym = eng.build_filter_sym(noiseCov, fSignals, w)                # :621-627  K6 (+ the symmetrised form)
            self._dFiltSymOf = (self._dFilt, sym)
            del noiseCov

            if self.params['outputUnits'] == 'yc':                                # :632-668
                # The reference pushes the amplitude-scaled template through enmap.apply_window (FFT . W . IFFT, :644)
                # and transforms it again (:646).  FFT(template) . W is Hermitian, so that inverse / forward pair is the
                # identity up to float32 round-off: here the window is multiplied into the spectrum directly, together
                # with the filter, while the inverse transform of :847 loads it.
                cRADeg, cDecDeg = self.wcs.getCentreWCSCoords()
                cx, cy = self.wcs.wcs2pix(cRADeg, cDecDeg)
                bckSub = 'bckSub' in self.params.keys() and 'bckSubScaleArcmin' in self.params.keys() and self.params['bckSub'] == True
                planes = None
                if N == 1 and not bckSub:
                    # one frequency: the filtered template is only read around the tile centre (:654-657) - only those rows are made
                    filteredSignal = self._applyFilterDevice(fSignals, window=1.0, half=True, planeScale=templateScale,
                                                             rows=self._centreRows(cy))
                elif not bckSub:
                    # several frequencies: the per-frequency filtered templates are wanted below anyway (:662-667); the filtered
                    # template is their sum in real space, in plane order, as :847 takes it
                    planes = eng.ifft(fSignals, out='f32', filt=self._filtFor(True), window=1.0, half=True, planeScale=templateScale)
                    filteredSignal = eng.sum_planes(planes)
                else:
                    filteredSignal = self._applyFilterDevice(fSignals, window=1.0, half=True, planeScale=templateScale)
                # signalNorm = y0 / peak stays on the device (float64 division there): the host goes on enqueueing
                peak = eng.spline_eval(filteredSignal, [[cy, cx]], to_host=False)                # :654-657  K9
                self._signalNorm = eng.signal_norm('yc', peak, y0)                # :659
                self.fRelWeights = {}
                if N == 1 and not bckSub:
                    # one frequency, no background subtraction: the per-frequency plane of :662 is the filtered signal
                    # itself, so the relative weight is a float32 value divided by itself
                    self.fRelWeights[self.unfilteredMapsDictList[0]['obsFreqGHz']] = np.float32(1.0)
                else:
                    # the peak and the planes' values there stay on the device; the float32 divisions of :667 are done on
                    # the host when somebody reads the weights (_DeviceWeights) - the enqueueing thread does not wait
                    mm = eng.map_max(filteredSignal, to_host=False)               # :661
                    if planes is None:
                        planes = eng.ifft(fSignals, out='f32', filt=self._filtFor(True), window=1.0, half=True,
                                          planeScale=templateScale)                                                  # :662-667
                    vals = planes.reshape(N, -1).index_select(1, mm[1:2].to(t.int64)).reshape(N)
                    self.fRelWeights = _DeviceWeights([m['obsFreqGHz'] for m in self.unfilteredMapsDictList],
                                                      t.cat([vals.to(t.float64), mm[:1]]))
            elif self.params['outputUnits'] == 'uK':                              # :669-685
                # the reference paints and transforms the unit templates a second time (:672-679): same values as fSignals
                filteredSignal = self._applyFilterDevice(fSignals, half=True)
                mm = eng.map_max(filteredSignal, to_host=False)                   # :684, float32 arithmetic, on the device
                self._signalNorm = eng.signal_norm('uK', mm)
            else:
                raise Exception('need to specify "outputUnits" ("yc" or "uK") in filter params')
            del fSignals
        else:
            if getattr(self, '_verbose', True):
                print("... loading cached filter")
            if self._dFilt is None or haveFile:
                self.loadFilter()
            self.params['saveRMSMap'] = False
            self.params['saveFilter'] = False
            self.params['savePlots'] = False

        # Apply filter (:696), point-source mask (:714)
        bckSub = 'bckSub' in self.params.keys() and 'bckSubScaleArcmin' in self.params.keys() and self.params['bckSub'] == True
        if not bckSub and tuple(fMapsToFilter.shape[-2:]) == self.shape:
            # the float32 cast, the signalNorm product and the mask are applied while the inverse transform stores its rows
            dFiltered = eng.ifft_half_epi(fMapsToFilter.reshape(-1, *self.shape), filtSym=self._filtFor(True), sum=True,
                                          scale=self._signalNorm, mask=self._dPSMask)
        else:
            filtered32 = self._applyFilterDevice(fMapsToFilter, scale=False, half=True)
            dFiltered = eng.scale_mask(filtered32, self._signalNorm, self._dPSMask)
        del fMapsToFilter
        mapUnits, combinedObsFreqGHz, beamSolidAngle_nsr = self._unitsInfo()

        trimSizePix = self._trimSizePix()
        data, sn, rms, mask, trimmed = self._finish(dFiltered, trimSizePix, True)
        self._saveFilterIfAsked()

        return self._packResult(data, sn, rms, mask, trimmed, True, mapUnits, combinedObsFreqGHz, beamSolidAngle_nsr)

    def _frequencyWeights(self):
        """Signal frequency weighting (nemo/filters.py:587-608)."""
        w = []
        for mapDict in self.unfilteredMapsDictList:
            if mapDict['units'] != 'yc':
                if 'specWeight' in mapDict.keys():
                    w.append(mapDict['specWeight'])
                else:
                    if self.params['outputUnits'] == 'yc':
                        w.append(signals.fSZ(mapDict['obsFreqGHz']))
                    elif self.params['outputUnits'] == 'uK':
                        if 'alpha' in self.params.keys() and self.params['alpha'] is not None:
                            w.append(np.power(mapDict['obsFreqGHz'] / self.unfilteredMapsDictList[0]['obsFreqGHz'],
                                              self.params['alpha']))
                        else:
                            w.append(1.0)
                    else:
                        raise Exception('need to specify "outputUnits" ("yc" or "uK") in filter params')
            else:
                w.append(1.0)
        return np.array(w)

    def makeForegroundsPower(self):
        """nemo/filters.py:261-276: CMB power on the 2-D Fourier grid (flat-sky |l| of the tile).  Not used on the path: its
        only caller in the reference is the unreachable 'max(dataMap,CMB)' branch.  |l| convention (enmap.modlmap) unpinned."""
        tab = np.loadtxt(os.path.join(os.path.dirname(__file__), "data", "planck_lensedCls.dat"))
        L, TT = tab[:, 0], tab[:, 1]
        TT = (TT * 2 * np.pi) / (L * (L + 1))
        H, W = self.shape
        ly = np.fft.fftfreq(H, d=np.radians(self.degPerPixY)) * 2 * np.pi
        lx = np.fft.fftfreq(W, d=np.radians(self.degPerPixX)) * 2 * np.pi
        lmap = np.sqrt(ly[:, None] ** 2 + lx[None, :] ** 2)
        fgPower = np.interp(lmap, L, TT, left=0.0, right=0.0) * H * W
        return fgPower

    def loadFilter(self):
        """nemo/filters.py:780-790."""
        data, header = fitsio.readFITS(self.filterFileName)
        self._dFilt = engine.to_dev(data.astype(np.float32))
        self.signalNorm = header['SIGNORM']
        self.loadFRelWeights()

    def reshapeFilter(self, shape):
        """The filter resampled to another map shape of the same pixel size by linear interpolation in multipole space
        (nemo/filters.py:793-817): every output mode (ly', lx') is looked up at its fractional index position in the
        filter's own mode grid.  float32 [N, H', W'] (a device tensor when RETURN_DEVICE_ARRAYS, else numpy).

        The reference's interpolator (scipy.interpolate.interp2d) no longer exists in SciPy >= 1.14, and its call only
        works for square maps (it hands the axes over transposed); restated here as what it computes there - bilinear
        interpolation at (yOut, xOut), positions outside the grid taking the edge value.  PARITY UNPINNED."""
        from scipy import interpolate
        if len(shape) == 2:
            shape = [self._dFilt.shape[0], shape[0], shape[1]]
        assert (len(shape) == 3)
        H, W = self.shape
        H2, W2 = int(shape[1]), int(shape[2])
        # enmap.laxes: multipoles of the modes along each axis, 2 pi fftfreq(n, pixel size in radians)
        ry, rx = np.radians(self.wcs.getYPixelSizeDeg()), np.radians(self.wcs.getXPixelSizeDeg())
        ly, lx = 2 * np.pi * np.fft.fftfreq(H, d=ry), 2 * np.pi * np.fft.fftfreq(W, d=rx)
        lxToX = interpolate.interp1d(lx, np.arange(lx.shape[0]), fill_value='extrapolate')
        lyToY = interpolate.interp1d(ly, np.arange(ly.shape[0]), fill_value='extrapolate')
        xOut = lxToX(2 * np.pi * np.fft.fftfreq(W2, d=rx))
        yOut = lyToY(2 * np.pi * np.fft.fftfreq(H2, d=ry))
        out = self.engine.resample_bilinear(self._dFilt, H2, W2, yOut, xOut)
        return out if RETURN_DEVICE_ARRAYS else _asnumpy(out)

    def _filtFor(self, half):
        """The filter as the inverse transform wants it: float32 planes, or - for half-stored spectra - their symmetrised
        float64 form 0.5 (filt[k] + filt[-k]), made once per filter."""
        if not half:
            return self._dFilt
        cached = getattr(self, '_dFiltSymOf', None)
        if cached is None or cached[0] is not self._dFilt:
            self._dFiltSymOf = (self._dFilt, self.engine.symmetrise_filter(self._dFilt))
        return self._dFiltSymOf[1]

    def _centreRows(self, cy, halfWin=32):
        """The rows of a map that the spline evaluation at row position ``cy`` reads (nf_spline_eval's window, with a margin)."""
        H = self.shape[0]
        iy = int(np.floor(cy))
        return max(0, iy - halfWin - 2), min(H, iy + halfWin + 4)

    def _applyFilterDevice(self, cube, scale=True, window=None, half=False, planeScale=None, rows=None):
        """Device form of applyFilter: complex (Fourier) or real [N,H,W] tensor in, float32 [H,W] map out (``half``: the
        complex cube holds half-stored spectra of real maps, as eng.fft(half=True) returns them); with
        ``scale`` the map is multiplied by signalNorm exactly as nemo/filters.py:853 does (numpy promotion).  The
        product with the filter, the frequency sum (moved in front of the inverse transform, which is linear) and an
        optional pixel-window factor are applied while the inverse transform loads the Fourier planes."""
        eng = self.engine
        t = engine.torch()
        if tuple(cube.shape[-2:]) != self.shape:                                  # :837-840
            # another map shape: the filter is interpolated onto that shape's modes, the transforms run on its own engine
            eng2 = engine.TileEngine.get(tuple(cube.shape[-2:]))
            keep = RETURN_DEVICE_ARRAYS
            globals()['RETURN_DEVICE_ARRAYS'] = True
            try:
                filt = self.reshapeFilter(tuple(cube.shape))
            finally:
                globals()['RETURN_DEVICE_ARRAYS'] = keep
            fMaps = cube if cube.is_complex() else eng2.fft(cube, apod=True)
            out = eng2.ifft(fMaps.reshape(-1, *cube.shape[-2:]), out='f32', unitary=False, filt=filt, sum=True, window=window)
            if 'bckSub' in self.params.keys() and 'bckSubScaleArcmin' in self.params.keys() and self.params['bckSub'] == True:
                out = maps.subtractBackground(out, self.wcs, smoothScaleDeg=self.params['bckSubScaleArcmin'] / 60.)
            if scale:
                if isinstance(self.signalNorm, np.float64):
                    out = out.to(t.float64) * float(self.signalNorm)
                elif float(np.float32(self.signalNorm)) != 1.0:
                    out = out * float(np.float32(self.signalNorm))
            return out
        if cube.is_complex():
            fMaps = cube
        else:
            fMaps = eng.fft(cube, apod=True, half=True)                           # :845
            half = True
        out = eng.ifft(fMaps.reshape(-1, *self.shape), out='f32', unitary=False, filt=self._filtFor(half), sum=True,
                       window=window, half=half, planeScale=planeScale, rows=rows)          # :847  K7 + K3
        if 'bckSub' in self.params.keys() and 'bckSubScaleArcmin' in self.params.keys() and self.params['bckSub'] == True:
            out = maps.subtractBackground(out, self.wcs, smoothScaleDeg=self.params['bckSubScaleArcmin'] / 60.)   # :850-851
        if scale:
            if isinstance(self.signalNorm, np.float64):
                out = out.to(t.float64) * float(self.signalNorm)
            elif float(np.float32(self.signalNorm)) != 1.0:       # x * 1.0 is x (the normalisation is still being measured)
                out = out * float(np.float32(self.signalNorm))
        return out

    def applyFilter(self, mapDataToFilter):
        """nemo/filters.py:820-855.  numpy in -> numpy out; device tensor in -> device tensor out."""
        isDev = hasattr(mapDataToFilter, 'is_cuda')
        out = self._applyFilterDevice(engine.to_dev(mapDataToFilter))
        return out if isDev else _asnumpy(out)


# ------------------------------------------------------------------------------------------------------------
class RealSpaceMatchedFilter(MapFilter):
    """Matched filter truncated to a small real-space kernel applied by direct convolution
    (nemo/filters.py:858-1210)."""

    def loadFilter(self):
        """nemo/filters.py:866-885."""
        kern2d, header = fitsio.readFITS(self.filterFileName)
        self.kern2d = kern2d
        self.signalNorm = header['SIGNORM']
        self.bckSubScaleArcmin = header['BCKSCALE'] if 'BCKSCALE' in header.keys() else 0
        self.applyRACentre = header['APP_RA']
        self.applyDecCentre = header['APP_DEC']
        self.fRelWeights = {}
        for i in range(1, 10):
            if 'RW%d_GHZ' % (i) in header.keys():
                self.fRelWeights[header['RW%d_GHZ' % (i)]] = header['RW%d' % (i)]

    def buildKernel(self, RADecSection, RADeg='centre', decDeg='centre'):
        """nemo/filters.py:888-1068 (plots omitted)."""
        if self.filterFileName is not None and os.path.exists(self.filterFileName) == True:
            return self.loadFilter()

        keysWanted = ['mapFileName', 'weights', 'weightsFileName', 'obsFreqGHz', 'units', 'beamFileName', 'addNoise',
                      'pointSourceRemoval', 'weightsType', 'tileName', 'reprojectToTan',
                      'mapData', 'weightsData', 'wcs', 'label']        # last four: in-memory map carriers of this build
        kernelUnfilteredMapsDictList = []
        for mapDict in self.unfilteredMapsDictList:
            for key in list(mapDict.keys()):
                if key not in keysWanted:
                    del mapDict[key]
            kernelUnfilteredMapsDict = maps.MapDict(mapDict, tileCoordsDict=mapDict.tileCoordsDict)
            kernelUnfilteredMapsDict['RADecSection'] = RADecSection
            kernelUnfilteredMapsDictList.append(kernelUnfilteredMapsDict)
        kernelLabel = "realSpaceKernel_%s" % (self.label)
        matchedFilterDir = None
        if self.diagnosticsDir is not None:
            matchedFilterDir = self.diagnosticsDir + os.path.sep + kernelLabel + "#" + self.tileName
            for d in [matchedFilterDir, matchedFilterDir + os.path.sep + 'diagnostics', matchedFilterDir + os.path.sep + 'selFn']:
                os.makedirs(d, exist_ok=True)
        matchedFilterClass = _FILTER_CLASSES[self.params['noiseParams']['matchedFilterClass']]
        matchedFilter = matchedFilterClass(kernelLabel, kernelUnfilteredMapsDictList, self.params,
                                           tileName=mapDict['tileName'],
                                           diagnosticsDir=None if matchedFilterDir is None else matchedFilterDir + os.path.sep + 'diagnostics',
                                           selFnDir=None if matchedFilterDir is None else matchedFilterDir + os.path.sep + 'selFn')
        matchedFilter.buildAndApply()

        # Truncate the real-space filter to a small kernel (:943-974)
        kernelMaxArcmin = self.params['noiseParams']['kernelMaxArcmin']
        prof, arcminRange = matchedFilter.makeRealSpaceFilterProfile()
        rIndex = np.where(arcminRange > kernelMaxArcmin)[0][0]
        mask = np.less(arcminRange, kernelMaxArcmin)
        if 'symmetrize' in self.params['noiseParams'].keys() and self.params['noiseParams']['symmetrize'] == True:
            from scipy import interpolate
            rRadians = np.radians(arcminRange / 60.)
            profile2d = []
            for i in range(prof.shape[0]):
                r2p = interpolate.interp1d(rRadians[mask], prof[i, mask], bounds_error=False, fill_value=0.0)
                profile2d.append(r2p(matchedFilter.radiansMap))
            profile2d = np.array(profile2d)
        else:
            profile2d = matchedFilter._realSpaceFilter()
        z, y, x = np.where(abs(profile2d) == abs(profile2d).max())
        y = y[0]
        x = x[0]
        yMin, yMax, xMin, xMax = y - rIndex, y + rIndex, x - rIndex, x + rIndex
        if (yMax - yMin) % 2 == 0:
            yMin = yMin + 1
        if (xMax - xMin) % 2 == 0:
            xMin = xMin + 1
        self.kern2d = profile2d[:, yMin:yMax, xMin:xMax]

        if 'bckSubScaleArcmin' in self.params.keys():                             # :977-988
            self.bckSubScaleArcmin = self.params['bckSubScaleArcmin']
        else:
            func = np.min if np.greater(prof[0, 0], 0) == True else np.max
            self.bckSubScaleArcmin = arcminRange[prof[0] == func(prof[0])][0]

        # Normalisation from an ideal signal map (:994-1021)
        signalMaps = []
        for mapDict in self.unfilteredMapsDictList:
            if self.params['outputUnits'] == 'yc':
                y0 = 2e-4
                if mapDict['obsFreqGHz'] is not None:
                    deltaT0 = maps.convertToDeltaT(y0, mapDict['obsFreqGHz'])
                    signalMap = self._paintTemplate(mapDict['beamFileName'], amplitude=deltaT0)
                else:
                    signalMap = self._paintTemplate(mapDict['beamFileName'], amplitude=y0)
            elif self.params['outputUnits'] == 'uK':
                signalMap = self._paintTemplate(mapDict['beamFileName'])
            else:
                raise Exception('need to specify "outputUnits" ("yc" or "uK") in filter params')
            signalMaps.append(signalMap)
        signalMaps = engine.torch().stack(signalMaps)
        filteredSignal = self._applyFilterDevice(signalMaps, calcFRelWeights=True)
        maxValue, maxIndex = self.engine.map_max(filteredSignal)
        if self.params['outputUnits'] == 'yc':
            self.signalNorm = y0 / np.float32(maxValue)
        else:
            self.signalNorm = 1.0 / np.float32(maxValue)

        if self.filterFileName is not None:                                       # :1023-1037
            header = dict(self.wcs.header)
            if self.params['bckSub'] == True:
                header['BCKSCALE'] = float(self.bckSubScaleArcmin)
            header['SIGNORM'] = float(self.signalNorm)
            header['APP_RA'] = float(self.applyRACentre)
            header['APP_DEC'] = float(self.applyDecCentre)
            count = 0
            for key in list(self.fRelWeights.keys()):
                count = count + 1
                header['RW%d_GHZ' % (count)] = key
                header['RW%d' % (count)] = float(self.fRelWeights[key])
            os.makedirs(os.path.split(self.filterFileName)[0], exist_ok=True)
            fitsio.writeFITS(self.filterFileName, np.asarray(self.kern2d), header)

    def buildAndApply(self, useCachedFilter=False):
        """nemo/filters.py:1071-1161."""
        if self.filterFileName is None or os.path.exists(self.filterFileName) == False:
            RAMin, RAMax, decMin, decMax = self.wcs.getImageMinMaxWCSCoords()
            section = self.params['noiseParams']['RADecSection']
            if section == 'tileNoiseRegions':
                RADecSection = [self.wcs.header['NRAMIN'], self.wcs.header['NRAMAX'],
                                self.wcs.header['NDEMIN'], self.wcs.header['NDEMAX']]
            elif section == 'auto':
                cRA, cDec = self.wcs.getCentreWCSCoords()
                halfSizeDeg = 2.0
                RADecSection = [cRA - halfSizeDeg / np.cos(np.radians(cDec)), cRA + halfSizeDeg / np.cos(np.radians(cDec)),
                                cDec - halfSizeDeg, cDec + halfSizeDeg]
            else:
                RADecSection = section
            self.applyDecCentre = (decMax + decMin) / 2.
            self.applyRACentre = (RAMax + RAMin) / 2.
            self.buildKernel(RADecSection, RADeg=self.applyRACentre, decDeg=self.applyDecCentre)
        else:
            self.loadFilter()

        # Apply kernel to the RAW tile (:1102-1106)
        t = engine.torch()
        mapDataToFilter = t.stack([engine.to_dev(m.loadTile('mapFileName', tileName=self.tileName))
                                   for m in self.unfilteredMapsDictList])
        filtered32 = self._applyFilterDevice(mapDataToFilter, scale=False)
        dFiltered = self.engine.scale_mask(filtered32, np.float32(self.signalNorm), self._dPSMask)   # :1205-1208, :1109

        mapUnits, combinedObsFreqGHz, beamSolidAngle_nsr = self._unitsInfo(allowMultiFreqUK=False)
        if 'edgeTrimArcmin' in self.params.keys():                                # :1133-1137
            trimSizePix = int(round((self.params['edgeTrimArcmin'] / 60.) / self.wcs.getPixelSizeDeg()))
        else:
            gridSize = int(round((self.params['noiseParams']['noiseGridArcmin'] / 60.) / self.wcs.getPixelSizeDeg()))
            trimSizePix = int(round(gridSize * 3.0))
        data, sn, rms, mask, trimmed = self._finish(dFiltered, trimSizePix, False)
        return self._packResult(data, sn, rms, mask, trimmed, False, mapUnits, combinedObsFreqGHz, beamSolidAngle_nsr)

    def _applyFilterDevice(self, cube, calcFRelWeights=False, scale=True):
        """Device form of applyFilter (nemo/filters.py:1164-1210): [N,H,W] real tensor -> float32 [H,W]."""
        eng = self.engine
        t = engine.torch()
        N = cube.shape[0]
        planes = []
        for i in range(N):
            if self.params['bckSub'] == True and self.bckSubScaleArcmin > 0:      # :1182-1187  K14
                p = maps.subtractBackground(cube[i], self.wcs, RADeg=self.applyRACentre, decDeg=self.applyDecCentre,
                                            smoothScaleDeg=self.bckSubScaleArcmin / 60.).to(t.float32)
            else:
                p = cube[i]       # float32 zeros + cube (:1189): a float64 cube is convolved, and summed, in float64
            planes.append(eng.convolve(p.contiguous(), np.asarray(self.kern2d[i])))         # :1192-1193  K13
        total = planes[0]
        for p in planes[1:]:
            total = total + p
        if calcFRelWeights == True:                                               # :1196-1203
            self.fRelWeights = {}
            maxValue, maxIndex = eng.map_max(total)
            for plane, mapDict in zip(planes, self.unfilteredMapsDictList):
                ft = np.float64 if plane.dtype == t.float64 else np.float32
                self.fRelWeights[mapDict['obsFreqGHz']] = ft(plane.reshape(-1)[maxIndex].item()) / ft(maxValue)
        if total.dtype != t.float32:
            total = total.to(t.float32)                                           # :1205 .astype(np.float32)
        if scale:
            total = total * float(np.float32(self.signalNorm))
        return total

    def applyFilter(self, mapDataToFilter, calcFRelWeights=False):
        isDev = hasattr(mapDataToFilter, 'is_cuda')
        out = self._applyFilterDevice(engine.to_dev(mapDataToFilter), calcFRelWeights=calcFRelWeights)
        return out if isDev else _asnumpy(out)


# ------------------------------------------------------------------------------------------------------------
class BeamFilter(MapFilter):
    """Point-source (beam) template (nemo/filters.py:1214-1222)."""
    templateKind = 'beam'


class ArnaudModelFilter(MapFilter):
    """Arnaud et al. (2010) GNFW cluster template (nemo/filters.py:1225-1241)."""
    templateKind = 'arnaud'


class BattagliaModelFilter(MapFilter):
    """Battaglia et al. (2012) GNFW cluster template (nemo/filters.py:1244-1266)."""
    templateKind = 'battaglia'


class ArnaudModelMatchedFilter(MatchedFilter, ArnaudModelFilter):
    pass


class BattagliaModelMatchedFilter(MatchedFilter, BattagliaModelFilter):
    pass


class BeamMatchedFilter(MatchedFilter, BeamFilter):
    pass


class ArnaudModelRealSpaceMatchedFilter(RealSpaceMatchedFilter, ArnaudModelFilter):
    pass


class BattagliaModelRealSpaceMatchedFilter(RealSpaceMatchedFilter, BattagliaModelFilter):
    pass


class BeamRealSpaceMatchedFilter(RealSpaceMatchedFilter, BeamFilter):
    pass


# class lookup replacing the reference's eval() of the YAML 'class' string (nemo/filters.py:82,933)
_FILTER_CLASSES = {c.__name__: c for c in [ArnaudModelMatchedFilter, BattagliaModelMatchedFilter, BeamMatchedFilter,
                                           ArnaudModelRealSpaceMatchedFilter, BattagliaModelRealSpaceMatchedFilter,
                                           BeamRealSpaceMatchedFilter]}
