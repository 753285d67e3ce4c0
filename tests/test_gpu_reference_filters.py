"""GPU parity against fixtures produced by the reference's OWN filter classes (tests/golden/make_golden_filters.py:
filters.filterMaps, MatchedFilter.buildAndApply / applyFilter, RealSpaceMatchedFilter.buildKernel / buildAndApply /
applyFilter of simonsobs/nemo run unmodified, with pixell.enmap served by oracle/nemo_oracle/pixell_restated.py).

Tolerances are those of tests/test_gpu_filters.py (same reasons): float64 maps 5e-6 of the map maximum end to end (the
template transforms are complex64 and the filter cube a float32 store in the reference itself), float32 maps 2e-3; staged
with the REFERENCE's filter cube and normalisation in the product object the apply path holds to 1.3e-7 (one float32
rounding).  Masks and zero patterns are exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import helpers  # noqa: E402
from nemo_amd import filters  # noqa: E402

GOLDEN = helpers.GOLDEN


def _rel(a, b):
    return np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max() / max(np.abs(b).max(), 1e-300)


def _check(got, g, p, tol, what):
    for key in ('data', 'SNMap', 'RMSMap'):
        ref = g[p + 'out_' + key]
        assert got[key].dtype == ref.dtype, (what, key, got[key].dtype, ref.dtype)
        assert np.array_equal(got[key] != 0, ref != 0) or key != 'RMSMap', (what, key)
        assert _rel(got[key], ref) < tol, (what, key, _rel(got[key], ref))
    assert np.array_equal(got['surveyMask'] > 0, g[p + 'out_surveyMask'] > 0), what
    assert got['surveyMask'].dtype == g[p + 'out_surveyMask'].dtype, what


@pytest.mark.parametrize("k", range(12))
def test_matched_filter_vs_reference_class_code(k):
    g = np.load(os.path.join(GOLDEN, "matched_filter.npz"))
    mapsL, wcs, meta = helpers.load_reference_case(g, k)
    p = 'c%d_' % k
    f64 = meta['dtype'] == 'float64'
    tol = 5e-6 if f64 else 2e-3
    f = {'label': 'ref%d' % k, 'class': meta['class'], 'params': meta['params']}
    got, fobj = filters.filterMaps(helpers.product_maps(mapsL), dict(f, params=dict(meta['params'])), 'PRIMARY', diagnosticsDir=None,
                                   selFnDir=None, verbose=False, undoPixelWindow=False, returnFilter=True)
    what = "case %d %s" % (k, meta)
    # the product paints its own templates on the GPU: they must be the fixture's (same painter definition)
    N = len(mapsL)
    for i, m in enumerate(mapsL):
        t = fobj.makeSignalTemplateMap(m['beamFileName'])
        assert np.abs(t - g[p + 'tmplUnit'][i]).max() <= 1e-7 * np.abs(g[p + 'tmplUnit'][i]).max(), what
    assert fobj.filt.dtype == np.float32 and fobj.filt.shape == g[p + 'filt'].shape
    # filter cube: the template spectra are complex64 in both paths (different FFT algorithms, ~1e-7 apart); the 3 x 3
    # covariance inverse amplifies that by its condition number, so three frequencies get 2e-5 (measured 5.4e-6)
    filtTol = tol if (N < 3 or not f64) else 2e-5
    assert _rel(fobj.filt, g[p + 'filt']) < filtTol, (what, _rel(fobj.filt, g[p + 'filt']))
    assert abs(float(fobj.signalNorm) / float(g[p + 'signalNorm']) - 1) < tol, what
    assert isinstance(fobj.signalNorm, np.float64) == bool(g[p + 'signalNormIsF64']), what
    if meta['params']['outputUnits'] == 'yc':
        rw = np.array([float(fobj.fRelWeights[m['obsFreqGHz']]) for m in mapsL])
        assert np.abs(rw - g[p + 'fRelWeights']).max() < tol, what
    _check(got, g, p, tol, what)
    # pixel-window undo of filterMaps (nemo/filters.py:98-101)
    got2 = filters.filterMaps(helpers.product_maps(mapsL), dict(f, params=dict(meta['params'])), 'PRIMARY', diagnosticsDir=None,
                              selFnDir=None, verbose=False, undoPixelWindow=True)
    ref2 = g[p + 'out_dataUndone']
    assert got2['data'].dtype == ref2.dtype, what
    assert _rel(got2['data'], ref2) < tol, what
    assert np.array_equal(got2['data'] == 0, ref2 == 0), what
    # staged: the reference's own filter cube and normalisation in the product object (useCachedFilter branch)
    fobj.filt = g[p + 'filt']
    sn = g[p + 'signalNorm']
    fobj.signalNorm = np.float64(sn) if bool(g[p + 'signalNormIsF64']) else np.float32(sn)
    staged = fobj.buildAndApply(useCachedFilter=True)
    if f64:
        _check(staged, g, p, 1.3e-7, what + " staged")


@pytest.mark.parametrize("k", range(5))
def test_realspace_filter_vs_reference_class_code(k, tmp_path):
    g = np.load(os.path.join(GOLDEN, "realspace_build.npz"))
    mapsL, wcs, meta = helpers.load_reference_case(g, k)
    p = 'c%d_' % k
    f64 = meta['dtype'] == 'float64'
    what = "case %d %s" % (k, meta)
    got, fobj = filters.filterMaps(helpers.product_maps(mapsL), {'label': 'rs%d' % k, 'class': meta['class'],
                                                                'params': dict(meta['params'])},
                                   'PRIMARY', diagnosticsDir=str(tmp_path), selFnDir=str(tmp_path), verbose=False,
                                   undoPixelWindow=False, returnFilter=True)
    kref = g[p + 'kern2d']
    assert np.asarray(fobj.kern2d).shape == kref.shape, what
    tol = 5e-5 if f64 else 3e-3
    assert _rel(fobj.kern2d, kref) < (2e-5 if f64 else 3e-3), (what, _rel(fobj.kern2d, kref))
    assert abs(fobj.bckSubScaleArcmin - float(g[p + 'bckSubScaleArcmin'])) < 1e-9, what
    assert abs(float(fobj.signalNorm) / float(g[p + 'signalNorm']) - 1) < (2e-5 if f64 else 3e-3), what
    rw = np.array([float(fobj.fRelWeights[m['obsFreqGHz']]) for m in mapsL])
    assert np.abs(rw - g[p + 'fRelWeights']).max() < tol, what
    _check(got, g, p, tol, what)
    # with the REFERENCE's kernel and normalisation: convolution + tail bit for bit
    fobj.kern2d = kref
    fobj.signalNorm = np.float32(g[p + 'signalNorm'])
    out = fobj.applyFilter(np.array([m['mapData'] for m in mapsL]))
    psMask = g[p + 'psMask']
    refData = g[p + 'out_data']
    inside = g[p + 'out_surveyMask'] > 0
    assert out.dtype == np.float32
    assert np.array_equal((out * psMask)[inside], refData[inside].astype(np.float32)), what


@pytest.mark.parametrize("k", range(4))
def test_measure_fluxes_vs_reference_code(k):
    """Every flux column of photometry.measureFluxes / getSNRValues against the reference's own functions
    (nemo/photometry.py:251-377; fixtures: tests/golden/measure_fluxes.npz).  The device spline is a +-32 pixel window of the
    whole-map FITPACK spline (1e-16 of it): float64 maps 1e-12 relative, float32 maps 1e-6."""
    import json
    from nemo_amd import photometry, catalogs
    from nemo_amd.wcs import CARWCS
    g = np.load(os.path.join(GOLDEN, "measure_fluxes.npz"))
    hdr = json.loads(str(g['header']))
    for key in ('NAXIS', 'NAXIS1', 'NAXIS2'):
        hdr[key] = int(hdr[key])
    p = 'c%d_' % k
    meta = json.loads(str(g[p + 'meta']))
    wcs = CARWCS(hdr)
    wcs.header['BUNIT'] = meta['units']
    if meta['jy']:
        wcs.header['BEAMNSR'] = meta['BEAMNSR']
        wcs.header['FREQGHZ'] = meta['FREQGHZ']
    tab = catalogs.Table(cols={'RADeg': g['RADeg'], 'decDeg': g['decDeg'], 'SNR': g[p + 'SNR']})
    fmd = {'data': g[p + 'data'], 'SNMap': g[p + 'SNMap'], 'wcs': wcs}
    phot = {'data': g[p + 'photData'], 'SNMap': g[p + 'photSNMap']} if meta['usePhot'] else None
    with np.errstate(divide='ignore', invalid='ignore'):
        photometry.measureFluxes(tab, fmd, None, photFilteredMapDict=phot, useInterpolator=meta['useInterpolator'])
    tol = 1e-12 if g[p + 'data'].dtype == np.float64 else 1e-6
    cols = json.loads(str(g[p + 'cols']))
    assert sorted(c for c in tab.keys() if c not in ('RADeg', 'decDeg', 'SNR')) == cols
    for c in cols:
        ref, got = g[p + 'col_' + c], np.asarray(tab[c], dtype=np.float64)
        fin = np.isfinite(ref)
        assert np.array_equal(np.isnan(ref), np.isnan(got)) and np.array_equal(np.isinf(ref), np.isinf(got)), (k, c)
        scale = np.abs(ref[fin]).max()
        assert np.abs(got[fin] - ref[fin]).max() <= tol * scale, (k, c, np.abs(got[fin] - ref[fin]).max() / scale)
    tab2 = catalogs.Table(cols={'RADeg': g['RADeg'], 'decDeg': g['decDeg']})
    photometry.getSNRValues(tab2, g[p + 'SNMap'], wcs, useInterpolator=meta['useInterpolator'], invertMap=True, prefix='inv_')
    ref = g[p + 'col_inv_SNR']
    assert np.array_equal(ref == 0, tab2['inv_SNR'] == 0)
    assert np.abs(tab2['inv_SNR'] - ref).max() <= tol * np.abs(ref).max()


def test_noise_method_max_datamap_cmb_raises_as_the_reference_does():
    g = np.load(os.path.join(GOLDEN, "matched_filter.npz"))
    message = str(g['maxDataMapCMB_raises'])
    mapsL, wcs, meta = helpers.load_reference_case(g, 0)
    params = dict(meta['params'], noiseParams=dict(meta['params']['noiseParams'], method='max(dataMap,CMB)'))
    with pytest.raises(Exception) as e:
        filters.filterMaps(helpers.product_maps(mapsL), {'label': 'x', 'class': meta['class'], 'params': params}, 'PRIMARY',
                           diagnosticsDir=None, selFnDir=None, verbose=False)
    assert str(e.value) == message


@pytest.mark.parametrize("k", range(4))
def test_find_objects_vs_reference_code(k):
    """photometry.findObjects as the reference runs it (fixture: tests/golden/find_objects.npz; measureShapes, ring flags, an
    inverted map, peak positions).  Counts, flags, names: equal.  Positions 1e-9 px (exact integer sums against scipy's float64
    running sums), S/N 1e-9 (spline), ellipses 1e-5 where the S/N map is float32 - the reference divides by a float32 pairwise
    sum there (``data[mask].sum()``), which the build rounds once from its exact sum: most rows come out equal, a last-place
    difference of that divisor moves a second moment by a few 1e-6 - and 1e-9 for float64 maps."""
    from test_cpu_reference_filters import _find_objects_cases
    from nemo_amd import photometry
    kk, meta, wcs, sn, flagMask, cols = list(_find_objects_cases())[k]
    kw = {a: v for a, v in meta.items() if a != 'dtype'}
    fmd = {'SNMap': sn, 'surveyMask': np.ones(sn.shape, dtype=np.uint8), 'wcs': wcs, 'flagMask': flagMask, 'label': 'golden',
           'tileName': 'PRIMARY'}
    tab = photometry.findObjects(fmd, rejectBorder=0, measureShapes=True, objIdent='ACT-S', **kw)
    assert len(tab) == len(cols['SNR'])
    for c in ('numSigPix', 'flags', 'ringFlag'):
        assert np.array_equal(np.asarray(tab[c]).astype(cols[c].dtype), cols[c]), c
    assert list(np.asarray(tab['name'])) == list(cols['name'])
    for c in ('x', 'y'):
        assert np.abs(np.asarray(tab[c]) - cols[c]).max() < 1e-9
    for c in ('RADeg', 'decDeg'):
        assert np.abs(np.asarray(tab[c]) - cols[c]).max() < 1e-11
    assert np.abs(np.asarray(tab['SNR']) / cols['SNR'] - 1).max() < 1e-9
    tol = 1e-5 if sn.dtype == np.float32 else 1e-9
    shaped = cols['ellipse_A'] > 0
    assert shaped.sum() >= 4
    for c in ('ellipse_PA', 'ellipse_A', 'ellipse_B', 'ellipse_x0', 'ellipse_y0', 'ellipse_e'):
        got = np.asarray(tab[c])
        assert np.array_equal(got[~shaped], cols[c][~shaped]), c                 # -99 rows: the same rows
        scale = 90.0 if c == 'ellipse_PA' else np.abs(cols[c][shaped])
        assert (np.abs(got[shaped] - cols[c][shaped]) / scale).max() < tol, c


def test_rms_table_vs_reference_code():
    """completeness.getRMSTab as the reference runs it (fixture: tests/golden/rms_tab.npz): the noise levels bit for bit, the
    areas to 2e-6 - the reference sums float32 pixel areas pairwise in float32, the build adds the same float32 values exactly
    (integer sums) - and the total area likewise."""
    from test_cpu_reference_filters import _rms_tab_cases
    from nemo_amd import completeness
    import types
    for wcs, rms, area, flag, maxFlags, areaDeg2, y0RMS in _rms_tab_cases():
        config = types.SimpleNamespace(cachedRMSMapDict={'PRIMARY': rms}, areaMaskDict={'PRIMARY': area}, flagMaskDict={'PRIMARY': flag},
                                       tileCoordsDict={'PRIMARY': {'wcs': wcs}})
        tab = completeness.getRMSTab('PRIMARY', 'golden', None, maxFlags=maxFlags, config=config)
        assert np.array_equal(np.asarray(tab['y0RMS']), y0RMS)
        got = np.asarray(tab['areaDeg2'])
        assert np.array_equal(got == 0, areaDeg2 == 0)
        nz = areaDeg2 > 0
        assert np.abs(got[nz] / areaDeg2[nz] - 1).max() < 2e-6
        assert abs(got.sum() / areaDeg2.sum() - 1) < 1e-6


def test_rms_tables_through_the_driver():
    """pipelines.makeRMSTables after a run: per tile the table of the photFilter's noise map over the survey area, merged and
    sorted by noise level; its area is the area of the survey mask."""
    from nemo_amd import pipelines, startUp, maps as pmaps
    mapsL, wcs, truth = helpers.make_tile((300, 360), seed=8, numSources=5, kind='clusters', holes=True)
    params = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20.0}, 'outputUnits': 'yc', 'edgeTrimArcmin': 0.0, 'M500MSun': 2e14, 'z': 0.4}
    parDict = {'unfilteredMaps': [dict(m) for m in helpers.product_maps(mapsL)], 'thresholdSigma': 4.0, 'minObjPix': 1, 'photFilter': 'A',
               'mapFilters': [{'label': 'A', 'class': 'ArnaudModelMatchedFilter', 'params': params}], 'useTiling': False, 'outputDir': None}
    config = startUp.NemoConfig(parDict, makeOutputDirs=False, verbose=False)
    pipelines.filterMapsAndMakeCatalogs(config, verbose=False, writeAreaMask=True, writeFlagMask=True)
    tab = pipelines.makeRMSTables(config)
    assert set(tab.keys()) == {'areaDeg2', 'y0RMS', 'tileName'} and len(tab) > 10
    assert np.all(np.diff(np.asarray(tab['y0RMS'])) >= 0) and np.all(np.asarray(tab['tileName']) == 'PRIMARY')
    areaMask = config.areaMaskDict['PRIMARY'].cpu().numpy()
    want = (pmaps.getPixelAreaArcmin2Map(areaMask.shape, wcs) * areaMask / 3600.0).sum(dtype=np.float64)
    rmsMap = config.cachedRMSMapDict['PRIMARY'].cpu().numpy()
    assert ((rmsMap != 0) >= (areaMask != 0)).all()
    assert abs(np.asarray(tab['areaDeg2']).sum() / want - 1) < 1e-6


@pytest.mark.parametrize("k", [k for k in range(12)])
def test_catalogue_membership_from_reference_sn_maps(k):
    """Float32 maps leave the S/N map 2e-3 (relative) from the reference's - wide enough to move objects across the threshold.
    What must hold: every object the reference's S/N map gives (oracle findObjects on the fixture's map) whose S/N clears the
    threshold by more than that margin is in the product's catalogue at the same place, and the product has nothing beyond
    the margin that the reference lacks.  Float64 maps: same test at their own tolerance."""
    from nemo_amd import photometry
    g = np.load(os.path.join(GOLDEN, "matched_filter.npz"))
    mapsL, wcs, meta = helpers.load_reference_case(g, k)
    p = 'c%d_' % k
    f64 = meta['dtype'] == 'float64'
    margin = (5e-6 if f64 else 2e-3) * 3
    thr = 3.5
    f = {'label': 'ref%d' % k, 'class': meta['class'], 'params': dict(meta['params'])}
    keep = filters.RETURN_DEVICE_ARRAYS
    filters.RETURN_DEVICE_ARRAYS = True
    try:
        got = filters.filterMaps(helpers.product_maps(mapsL), f, 'PRIMARY', diagnosticsDir=None, selFnDir=None, verbose=False,
                                 undoPixelWindow=False)
        cat = photometry.findObjects(got, threshold=thr, minObjPix=1, rejectBorder=0, flagRings=False)
    finally:
        filters.RETURN_DEVICE_ARRAYS = keep
    refSN = g[p + 'out_SNMap']
    rows = helpers.oracle_find(refSN, np.zeros(refSN.shape, dtype=int), wcs, threshold=thr, minObjPix=1, flagRings=False)
    peak = max(abs(float(refSN.max())), 1.0)
    gx, gy, gs = np.asarray(cat['x'], dtype=float), np.asarray(cat['y'], dtype=float), np.asarray(cat['SNR'], dtype=float)
    clear = 0
    for r in rows:
        if r['SNR'] < thr + margin * peak:
            continue
        clear += 1
        d = np.hypot(gx - r['x'], gy - r['y']) if len(gx) else np.array([np.inf])
        j = int(np.argmin(d))
        # centroids are S/N-weighted means over the segment: a pixel entering or leaving it at the threshold moves them
        assert d[j] < 0.5, (k, r, d[j])
        assert abs(gs[j] - r['SNR']) < margin * peak, (k, r['SNR'], gs[j])
    rx, ry = np.array([r['x'] for r in rows]), np.array([r['y'] for r in rows])
    for x, y, s in zip(gx, gy, gs):
        if s < thr + margin * peak:
            continue
        assert len(rx) > 0 and np.hypot(rx - x, ry - y).min() < 0.5, (k, x, y, s)
    assert clear >= 1, (k, len(rows))
