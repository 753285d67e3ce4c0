"""Oracle for the filtering half of the path (TEST INFRASTRUCTURE - see package docstring).

numpy/scipy restatement of nemo/filters.py (MapFilter.makeNoiseMap :342-480,
MatchedFilter.buildAndApply/applyFilter :521-855, RealSpaceMatchedFilter :866-1210,
filterMaps' pixel-window step :98-101) and nemo/maps.py:1279-1292,1357-1382
(subtractBackground / smoothMap).  Array-in / array-out: WCS-derived scalars (pixel size, map
centre, smoothing sigma in pixels) are passed in, templates come from a callable, so that the
arithmetic being restated is exactly the part the HIP kernels replace.

dtype behaviour follows numpy >= 2 promotion rules, which is what the reference runs under
(reference setup.cfg:29): e.g. in 'yc' mode ``signalNorm`` is a numpy float64 scalar, so the
filtered map is promoted to float64 at nemo/filters.py:853; in 'uK' mode it stays float32.
"""
import numpy as np
from scipy import ndimage, interpolate

from . import pixell_restated as enmap

APOD_PIX = 20          # nemo/filters.py:195


# ------------------------------------------------------------------------------------------------------------
def _clipped_rms(values):
    """3-sigma clipped standard deviation, 10 iterations (nemo/filters.py:466-476)."""
    if np.not_equal(values, 0).sum() == 0:
        return 0.
    mean = np.mean(values)
    rms = np.std(values)
    for _ in range(10):
        keep = np.less(abs(values), abs(mean + 3.0 * rms))
        if keep.sum() > 0:
            mean = np.mean(values[keep])
            rms = np.std(values[keep])
    return rms


def make_noise_map(mapData, weightsList, noiseParams, pixSizeDeg):
    """MapFilter.makeNoiseMap (nemo/filters.py:342-480): local RMS of a filtered map in grid cells
    (each widened by half a cell on every side), optionally split into weight-percentile bins per cell;
    or the 'smart' mode (global bins of similar weight).  Returns float32."""
    medWeights = np.median(np.array(weightsList), axis=0)                         # :357-360
    estimator = noiseParams.get('RMSEstimator', None)
    if estimator == 'biweight':
        raise NotImplementedError("biweight needs astropy.stats (absent offline)")

    if noiseParams.get('noiseGridArcmin', None) == "smart":                       # :363-404
        if 'numNoiseBins' not in noiseParams:
            raise Exception("Need to give numNoiseBins in noiseParams when using noiseGridArcmin = 'smart'")
        edges = np.linspace(medWeights.min(), medWeights.max(), noiseParams['numNoiseBins'])
        RMSMap = np.zeros(medWeights.shape, dtype=np.float32)
        nonZero = np.not_equal(mapData, 0)
        for lo, hi in zip(edges[:-1], edges[1:]):
            sel = np.logical_and(medWeights > lo, medWeights < hi)
            vals = mapData[sel]
            good = np.greater_equal(nonZero[sel], 1.0)
            if estimator == 'percentile':
                rms = np.percentile(abs(vals[good]), 68.3)
            elif np.not_equal(vals, 0).sum() != 0:
                mean = np.mean(vals[good])
                rms = np.std(vals[good])
                for _ in range(10):
                    keep = np.logical_and(good, np.less(abs(vals), abs(mean + 3.0 * rms)))
                    if keep.sum() > 0:
                        mean = np.mean(vals[keep])
                        rms = np.std(vals[keep])
            else:
                rms = 0.
            if rms > 0:
                RMSMap[sel] = rms
        return RMSMap

    H, W = mapData.shape
    if noiseParams.get('noiseGridArcmin', None) is None:                          # :409-412
        overlap, nx, ny = 0, 1, 1
    else:                                                                         # :414-417
        grid = int(round((noiseParams['noiseGridArcmin'] / 60.) / pixSizeDeg))
        overlap = int(grid / 2)
        nx = W / grid
        ny = H / grid
    yEdges = np.linspace(0, H, int(ny + 1), dtype=int)
    xEdges = np.linspace(0, W, int(nx + 1), dtype=int)
    nonZero = np.not_equal(mapData, 0)
    RMSMap = np.zeros(mapData.shape, dtype=np.float32)
    numBins = noiseParams.get('numNoiseBins', 1)
    for i in range(len(yEdges) - 1):
        for k in range(len(xEdges) - 1):
            y0 = max(yEdges[i] - overlap, 0)
            y1 = min(yEdges[i + 1] + overlap, H)
            x0 = max(xEdges[k] - overlap, 0)
            x1 = min(xEdges[k + 1] + overlap, W)
            vals = mapData[y0:y1, x0:x1]
            good = np.greater_equal(nonZero[y0:y1, x0:x1], 1.0)
            if good.sum() == 0:
                continue
            wts = medWeights[y0:y1, x0:x1]
            edges = [np.percentile(wts[good], p) for p in np.arange(0, 100, 100 / numBins)]   # :447-450
            edges.append(wts[good].max() + 1e-6)                                             # :451
            for lo, hi in zip(edges[:-1], edges[1:]):
                inBin = np.logical_and(wts >= lo, wts < hi)
                binVals = vals[inBin * good]
                if estimator == 'percentile':
                    rms = np.percentile(abs(binVals), 68.3)
                else:
                    rms = _clipped_rms(binVals)
                if rms > 0:
                    RMSMap[y0:y1, x0:x1][inBin] = rms                                        # :478
    return RMSMap


# ------------------------------------------------------------------------------------------------------------
def smooth_sigma_pix(smoothScaleDeg, xPixScaleDeg, yPixScaleDeg):
    """Gaussian sigma in pixels (y, x) as nemo/maps.py:1378-1379."""
    return smoothScaleDeg / yPixScaleDeg, smoothScaleDeg / xPixScaleDeg


def subtract_background(data, sigmaYX):
    """maps.subtractBackground (nemo/maps.py:1279-1292,1380): data - gaussian_filter(data, (sy, sx))."""
    return data - ndimage.gaussian_filter(data, sigmaYX)


def undo_pixel_window(data):
    """filterMaps' last step (nemo/filters.py:98-101)."""
    zero = np.equal(data, 0)
    out = enmap.apply_window(data, pow=-1.0)
    out[zero] = 0
    return out


def _snr_and_masks(filteredMap, RMSMap, surveyMask, psMask, trimSizePix, edgeAsFloat32):
    """Shared tail of both buildAndApply variants: S/N map, optional rank-filter edge trim, apodisation
    mask, final masking (nemo/filters.py:717-756 and :1112-1157)."""
    validMask = np.greater(RMSMap, 0)
    SNMap = np.zeros(filteredMap.shape, dtype=np.float32) + filteredMap
    SNMap[validMask] = SNMap[validMask] / RMSMap[validMask]
    if trimSizePix > 0:
        edgeCheck = ndimage.rank_filter(abs(filteredMap + (1 - psMask)), 0, size=(trimSizePix, trimSizePix))
        edgeCheck = np.array(np.greater(edgeCheck, 0), dtype=np.float32 if edgeAsFloat32 else float)
    else:
        edgeCheck = np.ones(filteredMap.shape)
    filteredMap = filteredMap * edgeCheck
    surveyMask = edgeCheck * surveyMask * psMask
    return filteredMap, SNMap, surveyMask


class MatchedFilter(object):
    """MatchedFilter (nemo/filters.py:516-855) on plain arrays.

    maps: list (one per frequency) of dicts with keys ``data``, ``weights``, ``obsFreqGHz``, ``units``,
          optionally ``specWeight``; ``surveyMask`` / ``pointSourceMask`` are read from maps[0] (:530-531).
    params: the filter ``params`` dict of the YAML config (noiseParams, outputUnits, alpha, bckSub, ...).
    geom: dict with ``pixSizeDeg`` (wcs.getPixelSizeDeg()), ``centreYX`` (wcs2pix of the centre coords,
          as (cy, cx)) and, when bckSub is used, ``xPixScaleDeg``/``yPixScaleDeg``.
    templateFn(i, amplitude) -> float32 signal template for map i (amplitude None = unit template).
    fSZ(freqGHz), deltaT0(freqGHz): spectral weight and y0 -> deltaT amplitude (nemo/signals.py:352,
          nemo/maps.py:679), supplied by the caller so the oracle has no cosmology/constant code of its own.
    """
    y0 = 2e-4     # :636

    def __init__(self, maps, params, geom, templateFn, fSZ=None, deltaT0=None):
        self.maps = maps
        self.params = params
        self.geom = geom
        self.templateFn = templateFn
        self.fSZ = fSZ
        self.deltaT0 = deltaT0
        self.shape = maps[0]['data'].shape
        self.signalNorm = 1.0
        self.fRelWeights = {}
        self.filt = None
        self.noiseCov = None

    def frequency_weights(self):                                                  # :587-608
        w = []
        for m in self.maps:
            if m['units'] != 'yc':
                if 'specWeight' in m:
                    w.append(m['specWeight'])
                elif self.params['outputUnits'] == 'yc':
                    w.append(self.fSZ(m['obsFreqGHz']))
                elif self.params['outputUnits'] == 'uK':
                    if self.params.get('alpha', None) is not None:
                        w.append(np.power(m['obsFreqGHz'] / self.maps[0]['obsFreqGHz'], self.params['alpha']))
                    else:
                        w.append(1.0)
                else:
                    raise Exception('need to specify "outputUnits" ("yc" or "uK") in filter params')
            else:
                w.append(1.0)
        return np.array(w)

    def build(self, faithfulLoop=False, noiseData=None):
        """Filter construction, nemo/filters.py:535-685.  ``noiseData`` optionally replaces the maps used
        for the noise term (the reference subtracts ``noiseModelCatalog`` models first, :540-547)."""
        n = len(self.maps)
        self.signalNorm = 1.0                                                     # fresh object state (:207)
        method = self.params['noiseParams']['method']
        if method == 'model':
            raise NotImplementedError("oracle restates the 'dataMap' noise method only ('model' needs pixell's rand_map)")
        if method != 'dataMap':
            # including 'max(dataMap,CMB)': its branch at nemo/filters.py:574-577 is dead code, the loop above it raises
            raise Exception("'%s' is not a valid filter noise method name - fix the .yml config file" % (method))   # :561
        fNoise = np.array([enmap.fft(enmap.apod(m['data'] if noiseData is None else noiseData[i], APOD_PIX))
                           for i, m in enumerate(self.maps)])
        noiseCov = []
        for i in range(n):                                                        # :565-584
            row = []
            for j in range(n):
                NP = np.real(fNoise[i] * fNoise[j].conj())
                row.append(ndimage.gaussian_filter(NP, (3, 3)))
            noiseCov.append(row)
        noiseCov = np.array(noiseCov)
        self.noiseCov = noiseCov
        w = self.frequency_weights()
        fSignals = np.array([enmap.fft(self.templateFn(i, None)) for i in range(n)])          # :611-618
        filt = np.zeros([n, self.shape[0], self.shape[1]], dtype=np.float32)                 # :621
        if faithfulLoop:
            for y in range(self.shape[0]):                                        # :622-627 verbatim semantics
                for x in range(self.shape[1]):
                    try:
                        filt[:, y, x] = np.dot(np.linalg.inv(noiseCov[:, :, y, x]), w * abs(fSignals[:, y, x]))
                    except Exception:
                        continue
        else:
            cov = np.moveaxis(noiseCov, (0, 1), (-2, -1))                         # [H, W, n, n]
            rhs = np.moveaxis(w[:, None, None] * abs(fSignals), 0, -1)             # [H, W, n]
            if n == 1:
                ok = cov[..., 0, 0] != 0
            else:
                sign, logdet = np.linalg.slogdet(cov)
                ok = sign != 0
            inv = np.zeros(cov.shape, dtype=cov.dtype)
            inv[ok] = np.linalg.inv(cov[ok])
            sol = np.einsum('...ij,...j->...i', inv.astype(np.float64), rhs)
            sol[~ok] = 0
            filt[:] = np.moveaxis(sol, -1, 0)
        self.filt = filt

        cy, cx = self.geom['centreYX']
        if self.params['outputUnits'] == 'yc':                                    # :632-668
            fSignalMaps = []
            for i, m in enumerate(self.maps):
                amp = self.y0 if m['units'] == 'yc' else self.deltaT0(m['obsFreqGHz'])
                s = self.templateFn(i, amp)
                s = enmap.apply_window(s, pow=1.0)
                fSignalMaps.append(enmap.fft(s))
            fSignalMaps = np.array(fSignalMaps)
            filteredSignal = self.apply(fSignalMaps)
            spline = interpolate.RectBivariateSpline(np.arange(filteredSignal.shape[0]),
                                                     np.arange(filteredSignal.shape[1]),
                                                     filteredSignal, kx=3, ky=3)
            peak = spline(cy, cx)[0][0]
            self.signalNorm = self.y0 / peak
            peakIndex = np.argmax(filteredSignal)
            total = filteredSignal.flatten()[peakIndex]
            cube = np.real(enmap.ifft(fSignalMaps * self.filt, normalize=False))
            self.fRelWeights = {}
            for plane, m in zip(cube, self.maps):
                self.fRelWeights[m['obsFreqGHz']] = plane.flatten()[peakIndex] / total
        elif self.params['outputUnits'] == 'uK':                                  # :669-685
            fSignalMaps = np.array([enmap.fft(self.templateFn(i, None)) for i in range(n)])
            filteredSignal = self.apply(fSignalMaps)
            self.signalNorm = 1.0 / filteredSignal.max()
        else:
            raise Exception('need to specify "outputUnits" ("yc" or "uK") in filter params')

    def apply(self, cube):
        """MatchedFilter.applyFilter (nemo/filters.py:820-855), same-shape case."""
        if 'complex' in cube.dtype.name:
            fMaps = cube
        else:
            fMaps = enmap.fft(enmap.apod(cube, APOD_PIX))
        out = np.real(enmap.ifft(fMaps * self.filt, normalize=False)).sum(axis=0).astype(np.float32)
        if self.params.get('bckSub', False) == True and 'bckSubScaleArcmin' in self.params:
            sig = smooth_sigma_pix(self.params['bckSubScaleArcmin'] / 60., self.geom['xPixScaleDeg'],
                                   self.geom['yPixScaleDeg'])
            out = subtract_background(out, sig)
        return out * self.signalNorm

    def buildAndApply(self, faithfulLoop=False, prebuilt=False):
        """nemo/filters.py:521-777 (no file caching; ``prebuilt`` = the useCachedFilter branch)."""
        fMaps = np.array([enmap.fft(enmap.apod(m['data'], APOD_PIX)) for m in self.maps])     # :523-526
        surveyMask = self.maps[0]['surveyMask']
        psMask = self.maps[0]['pointSourceMask']
        if not prebuilt:
            self.build(faithfulLoop=faithfulLoop)
        filteredMap = self.apply(fMaps)                                           # :696
        filteredMap = filteredMap * psMask                                        # :714
        RMSMap = make_noise_map(filteredMap, [m['weights'] for m in self.maps],
                                self.params['noiseParams'], self.geom['pixSizeDeg'])
        if self.params.get('edgeTrimArcmin', 0) > 0:                              # :725-733
            trim = int(round((self.params['edgeTrimArcmin'] / 60.) / self.geom['pixSizeDeg']))
        else:
            trim = 0
        filteredMap, SNMap, surveyMask = _snr_and_masks(filteredMap, RMSMap, surveyMask, psMask, trim, True)
        filteredMap = filteredMap * surveyMask                                    # :741
        apodMask = np.equal(enmap.apod(np.ones(filteredMap.shape), APOD_PIX), 1)   # :747-748
        surveyMask = surveyMask * apodMask
        SNMap = SNMap * surveyMask                                                # :754-756
        SNMap[np.isnan(SNMap)] = 0.
        RMSMap = RMSMap * surveyMask
        return {'data': filteredMap, 'SNMap': SNMap, 'RMSMap': RMSMap, 'surveyMask': surveyMask}


# ------------------------------------------------------------------------------------------------------------
def real_space_profile(filt, degPerPixX):
    """MapFilter.makeRealSpaceFilterProfile (nemo/filters.py:279-301)."""
    realSpace = np.fft.fftshift(np.fft.ifft2(filt).real)
    x0 = int(realSpace.shape[2] / 2)
    y0 = int(realSpace.shape[1] / 2)
    prof = realSpace[:, y0, x0:] / abs(realSpace[:, y0, x0:]).max()
    return prof, np.arange(0, prof.shape[1]) * degPerPixX * 60.0


def crop_kernel(filt, radiansMap, degPerPixX, kernelMaxArcmin, symmetrize=False):
    """Kernel extraction of RealSpaceMatchedFilter.buildKernel (nemo/filters.py:943-974)."""
    prof, arcminRange = real_space_profile(filt, degPerPixX)
    rIndex = np.where(arcminRange > kernelMaxArcmin)[0][0]
    mask = np.less(arcminRange, kernelMaxArcmin)
    if symmetrize:
        rRadians = np.radians(arcminRange / 60.)
        profile2d = np.array([interpolate.interp1d(rRadians[mask], prof[i, mask], bounds_error=False,
                                                   fill_value=0.0)(radiansMap) for i in range(prof.shape[0])])
    else:
        profile2d = np.fft.fftshift(np.fft.ifft2(filt).real)
    z, y, x = np.where(abs(profile2d) == abs(profile2d).max())
    y, x = y[0], x[0]
    yMin, yMax, xMin, xMax = y - rIndex, y + rIndex, x - rIndex, x + rIndex
    if (yMax - yMin) % 2 == 0:
        yMin = yMin + 1
    if (xMax - xMin) % 2 == 0:
        xMin = xMin + 1
    return profile2d[:, yMin:yMax, xMin:xMax], prof, arcminRange


def real_space_apply(cube, kern2d, signalNorm, bckSubSigmaYX=None, calcFRelWeights=False, obsFreqs=None):
    """RealSpaceMatchedFilter.applyFilter (nemo/filters.py:1164-1210).  ``bckSubSigmaYX`` = Gaussian
    sigma in pixels when ``bckSub`` is on and the scale is > 0, else None."""
    out = np.zeros(cube.shape, dtype=np.float32)
    if bckSubSigmaYX is not None:
        for i in range(cube.shape[0]):
            out[i] = subtract_background(cube[i], bckSubSigmaYX)
    else:
        out = out + cube
    for i in range(out.shape[0]):
        out[i] = ndimage.convolve(out[i], kern2d[i])
    fRelWeights = {}
    if calcFRelWeights:
        total = out.sum(axis=0)
        maxIndex = np.argmax(total)
        for plane, f in zip(out, obsFreqs):
            fRelWeights[f] = plane.flatten()[maxIndex] / total.flatten()[maxIndex]
    res = out.sum(axis=0).astype(np.float32) * signalNorm
    if calcFRelWeights:
        return res, fRelWeights
    return res


def real_space_finish(filteredMap, weightsList, surveyMask, psMask, params, pixSizeDeg):
    """Tail of RealSpaceMatchedFilter.buildAndApply (nemo/filters.py:1108-1161)."""
    filteredMap = filteredMap * psMask
    RMSMap = make_noise_map(filteredMap, weightsList, params['noiseParams'], pixSizeDeg)
    if 'edgeTrimArcmin' in params:
        trim = int(round((params['edgeTrimArcmin'] / 60.) / pixSizeDeg))
    else:
        grid = int(round((params['noiseParams']['noiseGridArcmin'] / 60.) / pixSizeDeg))
        trim = int(round(grid * 3.0))
    filteredMap, SNMap, surveyMask = _snr_and_masks(filteredMap, RMSMap, surveyMask, psMask, trim, False)
    apodMask = np.equal(enmap.apod(np.ones(filteredMap.shape), APOD_PIX), 1)
    surveyMask = surveyMask * apodMask
    SNMap = SNMap * surveyMask
    SNMap[np.isnan(SNMap)] = 0.
    RMSMap = RMSMap * surveyMask
    return {'data': filteredMap, 'SNMap': SNMap, 'RMSMap': RMSMap, 'surveyMask': surveyMask}
