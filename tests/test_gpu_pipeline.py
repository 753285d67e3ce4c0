"""GPU tests of the driver level (nemo/pipelines.py:35-398): multi-scale run against the oracle, asynchronous vs
synchronous detection, the cached-filter + cached-RMS re-filter mode used by source-injection runs, forced photometry."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import helpers  # noqa: E402
from nemo_amd import pipelines, startUp, catalogs  # noqa: E402

SHAPE = (300, 360)
SCALES = [('Arnaud_M2e14_z0p4', 2e14, 0.4), ('Arnaud_M4e14_z0p2', 4e14, 0.2), ('Arnaud_M1e14_z0p8', 1e14, 0.8)]


def _filters():
    return [{'label': lab, 'class': 'ArnaudModelMatchedFilter',
             'params': {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20.0}, 'outputUnits': 'yc',
                        'edgeTrimArcmin': 0.0, 'M500MSun': m, 'z': z}} for lab, m, z in SCALES]


def _config(mapsL, **extra):
    parDict = {'unfilteredMaps': [dict(m) for m in mapsL], 'thresholdSigma': 4.0, 'minObjPix': 1, 'findCenterOfMass': True,
               'useInterpolator': True, 'optimalCatalogMethod': 'ref-forced', 'rejectBorder': 0, 'objIdent': 'ACT-CL',
               'longNames': False, 'photFilter': SCALES[0][0], 'mapFilters': _filters(), 'useTiling': False, 'outputDir': None}
    parDict.update(extra)
    return startUp.NemoConfig(parDict, makeOutputDirs=False, verbose=False)


def _same_catalog(a, b, snrTol=0.0, posTol=0.0):
    assert len(a) == len(b)
    if len(a) == 0:
        return
    ia, ib = np.lexsort((a['decDeg'], a['RADeg'])), np.lexsort((b['decDeg'], b['RADeg']))
    for k in ('RADeg', 'decDeg'):
        assert np.abs(np.asarray(a[k])[ia] - np.asarray(b[k])[ib]).max() <= posTol
    assert np.abs(np.asarray(a['SNR'])[ia] / np.asarray(b['SNR'])[ib] - 1).max() <= snrTol


def test_pipeline_async_equals_sync_and_matches_oracle():
    mapsL, wcs, truth = helpers.make_tile(SHAPE, freqs=(150.0, 98.0), fwhm=(1.4, 2.1), noise=(12.0, 16.0), seed=5,
                                          numSources=6, kind='clusters')
    cat = pipelines.filterMapsAndMakeCatalogs(_config(helpers.product_maps(mapsL)), verbose=False)
    catSync = pipelines.filterMapsAndMakeCatalogs(_config(helpers.product_maps(mapsL), asyncDetection=False), verbose=False)
    assert len(cat) >= 3
    # the second stream changes when things run, not what they compute: the centroid sums are exact integer accumulations
    # (order-independent), so positions, S/N values and every other column repeat bit for bit
    _same_catalog(cat, catSync, snrTol=0.0, posTol=0.0)
    for k in cat.keys():
        ia, ib = np.lexsort((cat['decDeg'], cat['RADeg'])), np.lexsort((catSync['decDeg'], catSync['RADeg']))
        assert np.array_equal(np.asarray(cat[k])[ia], np.asarray(catSync[k])[ib]), k
    for k in ('y_c', 'fixed_y_c', 'fixed_SNR', 'deltaT_c'):
        assert k in cat.keys()
    # per scale against the oracle (filter + detection), then the reference's merge rules applied to the oracle's tables
    perScale = {}
    for lab, m, z in SCALES:
        params = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20.0}, 'outputUnits': 'yc', 'edgeTrimArcmin': 0.0,
                  'M500MSun': m, 'z': z}
        ref, omf = helpers.oracle_matched(mapsL, wcs, params, 'arnaud')
        perScale[lab] = helpers.oracle_find(ref['SNMap'], np.zeros(SHAPE, dtype=int), wcs, threshold=4.0, minObjPix=1)
    best = {}
    for lab, rows in perScale.items():
        for r in rows:
            key = None
            for k in best:
                if abs(best[k]['RADeg'] - r['RADeg']) < catalogs.XMATCH_RADIUS_DEG and abs(best[k]['decDeg'] - r['decDeg']) < catalogs.XMATCH_RADIUS_DEG:
                    key = k
            if key is None:
                best[len(best)] = dict(r)
            elif r['SNR'] > best[key]['SNR']:
                best[key] = dict(r)
    refSNR = sorted(b['SNR'] for b in best.values())
    gotSNR = sorted(np.asarray(cat['SNR']).tolist())
    # objects well above the threshold must all be there with the same S/N (float32-class agreement of the filter build);
    # objects within 1e-3 of the threshold may appear on one side only
    strongRef = [s for s in refSNR if s > 4.01]
    strongGot = [s for s in gotSNR if s > 4.01]
    assert len(strongRef) == len(strongGot)
    assert np.abs(np.array(strongRef) / np.array(strongGot) - 1).max() < 2e-5


@pytest.mark.parametrize("threshold,measure", [(4.0, True), (2.5, True), (3.0, False)])
def test_group_detection_written_once_equals_map_by_map(threshold, measure):
    """photometry._findObjectsVector (the detections of a tile's scales with the host side written once for all of them) against
    the map-by-map generators: the optimal catalogue and every per-scale catalogue, column for column, bit for bit."""
    from nemo_amd import photometry
    mapsL, wcs, truth = helpers.make_tile(SHAPE, freqs=(150.0, 98.0), fwhm=(1.4, 2.1), noise=(12.0, 16.0), seed=21,
                                          numSources=8, kind='clusters', holes=True)
    keep = photometry._VECTOR_GROUP
    seen = {}
    orig = pipelines._detectAndMeasure

    def spy(config, jobs, *a, **kw):
        cats = orig(config, jobs, *a, **kw)
        for job, c in zip(jobs, cats):
            seen.setdefault(photometry._VECTOR_GROUP, {})[job[0]] = c
        return cats
    pipelines._detectAndMeasure = spy
    try:
        out = {}
        for mode in (True, False):
            photometry._VECTOR_GROUP = mode
            out[mode] = pipelines.filterMapsAndMakeCatalogs(_config(helpers.product_maps(mapsL), thresholdSigma=threshold,
                                                                    asyncDetection=False), measureFluxes=measure, verbose=False)
    finally:
        photometry._VECTOR_GROUP = keep
        pipelines._detectAndMeasure = orig
    assert len(out[True]) >= 3 and list(out[True].keys()) == list(out[False].keys())
    for k in out[True].keys():
        assert np.asarray(out[True][k]).dtype == np.asarray(out[False][k]).dtype, k
        assert np.array_equal(np.asarray(out[True][k]), np.asarray(out[False][k])), k
    assert set(seen[True].keys()) == set(seen[False].keys()) and len(seen[True]) == len(SCALES)
    for lab in seen[True]:
        a, b = seen[True][lab], seen[False][lab]
        assert len(a) == len(b) and len(a) > 0
        assert list(a.keys()) == list(b.keys()), lab
        for k in a.keys():
            assert np.asarray(a[k]).dtype == np.asarray(b[k]).dtype, (lab, k)
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), (lab, k)


def test_cached_filter_and_rms_refilter_mode():
    """The mode maps.sourceInjectionTest drives (nemo/maps.py:2047-2049, nemo/pipelines.py:234-250): only the reference
    filter, cached filter cube + cached RMS map, pixel window undone after the new S/N map."""
    mapsL, wcs, truth = helpers.make_tile(SHAPE, seed=8, numSources=5, kind='clusters')
    config = _config(helpers.product_maps(mapsL))
    config.parDict['mapFilters'] = [f for f in config.parDict['mapFilters'] if f['label'] == config.parDict['photFilter']]
    first = pipelines.filterMapsAndMakeCatalogs(config, verbose=False)
    assert 'PRIMARY' in config.cachedRMSMapDict and 'PRIMARY' in config.cachedFilters
    again = pipelines.filterMapsAndMakeCatalogs(config, useCachedFilters=True, useCachedRMSMap=True, verbose=False)
    assert len(first) >= 2
    # same maps, same filter, same noise map: the S/N map (hence positions and S/N) is reproduced; fluxes too, because the
    # pixel window is undone after the S/N map in this branch instead of inside filterMaps
    _same_catalog(first, again, snrTol=1e-12, posTol=1e-12)
    assert np.abs(np.asarray(again['y_c']) / np.asarray(first['y_c']) - 1).max() < 1e-12


def test_forced_photometry_in_the_driver(tmp_path):
    mapsL, wcs, truth = helpers.make_tile(SHAPE, seed=8, numSources=5, kind='clusters')
    config = _config(helpers.product_maps(mapsL))
    detected = pipelines.filterMapsAndMakeCatalogs(config, verbose=False)
    assert len(detected) >= 2
    forced = catalogs.Table()
    forced['name'] = np.array(detected['name'])
    forced['RADeg'] = np.asarray(detected['RADeg'])
    forced['decDeg'] = np.asarray(detected['decDeg'])
    config2 = _config(helpers.product_maps(mapsL), forcedPhotometryCatalog=forced)
    cat = pipelines.filterMapsAndMakeCatalogs(config2, verbose=False)
    assert len(cat) == len(detected)
    # forced positions are rounded to the pixel (nemo/photometry.py:414-416): S/N within a few per cent of the detection's
    ia, ib = np.argsort(np.asarray(cat['RADeg'])), np.argsort(np.asarray(detected['RADeg']))
    ratio = np.asarray(cat['SNR'])[ia] / np.asarray(detected['SNR'])[ib]
    assert np.all(ratio > 0.8) and np.all(ratio < 1.05)
    # the same from a catalogue file (nemo/photometry.py:402-403), FITS table and .csv
    for ext in ("fits", "csv"):
        path = str(tmp_path / ("forced." + ext))
        catalogs.writeCatalog(forced, path)
        cat3 = pipelines.filterMapsAndMakeCatalogs(_config(helpers.product_maps(mapsL), forcedPhotometryCatalog=path), verbose=False)
        ic = np.argsort(np.asarray(cat3['RADeg']))
        assert len(cat3) == len(cat) and np.array_equal(np.asarray(cat3['SNR'])[ic], np.asarray(cat['SNR'])[ia])


def test_fitQ_against_the_oracle():
    """signals.fitQ (nemo/signals.py:897-1163) after a two-frequency run: a handful of cluster scales (and the reference
    filter's own, which must give Q = 1) painted, windowed, filtered with the reference filter and read at the centre, against
    the same steps taken with the oracle's filter object."""
    from scipy import interpolate
    from nemo_amd import signals, maps as pmaps
    from nemo_oracle import pixell_restated as enmap
    mapsL, wcs, truth = helpers.make_tile(SHAPE, freqs=(150.0, 98.0), fwhm=(1.4, 2.1), noise=(12.0, 16.0), seed=5, numSources=6,
                                          kind='clusters')
    config = _config(helpers.product_maps(mapsL))
    config.parDict['mapFilters'] = [f for f in config.parDict['mapFilters'] if f['label'] == config.parDict['photFilter']]
    pipelines.filterMapsAndMakeCatalogs(config, verbose=False)
    scales = [(0.3, 6e14), (0.8, 1e14), (1.5, 3e13), (0.1, 4e14)]
    QTab = signals.fitQ(config, scales=scales)['PRIMARY']
    assert QTab.meta['ZDEPQ'] == 0 and len(QTab) == 5
    assert np.all(np.diff(np.asarray(QTab['theta500Arcmin'])) > 0)
    # oracle: the same filter built by the oracle, applied to oracle-painted, oracle-windowed templates
    lab, m, z = SCALES[0]
    params = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20.0}, 'outputUnits': 'yc', 'edgeTrimArcmin': 0.0, 'M500MSun': m, 'z': z}
    ref, omf = helpers.oracle_matched(mapsL, wcs, params, 'arnaud')
    RADeg, decDeg = wcs.getCentreWCSCoords()
    x, y = wcs.wcs2pix(RADeg, decDeg)
    geom = signals.paintGeometry(SHAPE, wcs, RADeg, decDeg)
    want = {}
    for zz, mm in [(z, m)] + scales:
        cube = []
        for mp in mapsL:
            table = signals.clusterModelTable('arnaud', mp['beamFileName'], zz, mm, 'default', signals.convertToDeltaT(2e-4, mp['obsFreqGHz']), 15.0)
            cube.append(enmap.fft(enmap.apply_window(helpers.cpu_paint(geom, table), pow=1.0)))
        filtered = omf.apply(np.array(cube))
        spl = interpolate.RectBivariateSpline(np.arange(SHAPE[0]), np.arange(SHAPE[1]), filtered, kx=3, ky=3)
        want[signals.calcTheta500Arcmin(zz, mm, signals.fiducialCosmoModel)] = spl(y, x)[0][0] / 2e-4
    theta = sorted(want)
    assert np.abs(np.asarray(QTab['theta500Arcmin']) - np.array(theta)).max() < 1e-12
    assert np.abs(np.asarray(QTab['Q']) / np.array([want[k] for k in theta]) - 1).max() < 2e-5
    assert abs(float(np.asarray(QTab['Q'])[np.asarray(QTab['z']) == z][0]) - 1) < 1e-6


def test_driver_writes_the_tile_products(tmp_path):
    """With an output directory the pass ends as the reference's does (nemo/pipelines.py:338-396): area / flag masks and the
    reference filter's RMS map as multi-extension files under selFn/, stitched images of those and of the filtered and S/N
    maps; the catalogue goes out through catalogs.writeCatalog and reads back row for row."""
    from nemo_amd import fitsio
    mapsL, wcs, truth = helpers.make_tile(SHAPE, seed=8, numSources=5, kind='clusters', holes=True)
    config = _config(helpers.product_maps(mapsL), outputDir=str(tmp_path / "run"))
    for f in config.parDict['mapFilters']:
        if f['label'] == config.parDict['photFilter']:
            f['params']['saveFilteredMaps'] = True
    cat = pipelines.filterMapsAndMakeCatalogs(config, verbose=False, writeAreaMask=True, writeFlagMask=True)
    phot = config.parDict['photFilter']
    sel, fm = config.selFnDir, config.filteredMapsDir
    for path in (sel + "/areaMask.fits", sel + "/flagMask.fits", sel + "/RMSMap_%s.fits" % phot, sel + "/stitched_areaMask.fits",
                 sel + "/stitched_RMSMap_%s.fits" % phot, fm + "/stitched_%s_filteredMap.fits" % phot, fm + "/stitched_%s_SNMap.fits" % phot):
        assert os.path.exists(path), path
    hdus = fitsio.readHDUs(sel + "/RMSMap_%s.fits" % phot)
    assert hdus[1][0] == 'PRIMARY' or hdus[1][0] == config.tileNames[0]
    assert np.array_equal(hdus[1][1], config.stitchedRMSMapDict['PRIMARY'].cpu().numpy())
    area = fitsio.readHDUs(sel + "/areaMask.fits")[1][1]
    assert area.dtype == np.uint8 and np.array_equal(area, config.areaMaskDict['PRIMARY'].cpu().numpy())
    sn, header = fitsio.readFITS(fm + "/stitched_%s_SNMap.fits" % phot)
    assert sn.dtype == np.float32 and np.array_equal(sn, config.stitchedSNMapDict['PRIMARY'].cpu().numpy()) and header['CTYPE1'] == 'RA---CAR'
    out = str(tmp_path / "cat.fits")
    catalogs.writeCatalog(cat, out)
    back = catalogs.readCatalog(out)
    assert len(back) == len(cat) and list(np.asarray(back['name'])) == list(np.asarray(cat['name']))
    assert np.array_equal(np.asarray(back['SNR']), np.asarray(cat['SNR']))


def test_use_cached_filtered_maps(tmp_path):
    """useCachedFilteredMaps (nemo/pipelines.py:202-218): filters whose maps lie under filteredMaps/<tile>/ are read, not run;
    detection and photometry on the read maps give the catalogue of the run that made them.  (The reference no longer writes
    these per-tile files itself: the test writes them with the library's own FITS writer.)"""
    from nemo_amd import filters, fitsio
    mapsL, wcs, truth = helpers.make_tile(SHAPE, seed=12, numSources=5, kind='clusters', holes=True)
    config = _config(helpers.product_maps(mapsL), outputDir=str(tmp_path / "run"))
    cat = pipelines.filterMapsAndMakeCatalogs(config, verbose=False, writeAreaMask=True, writeFlagMask=True)
    assert len(cat) >= 3
    tile = config.tileNames[0]
    os.makedirs(config.filteredMapsDir + os.sep + tile, exist_ok=True)
    keep = filters.RETURN_DEVICE_ARRAYS
    filters.RETURN_DEVICE_ARRAYS = False
    try:
        cache = {}
        for f in config.parDict['mapFilters'][:2]:            # two of the three filters on disk, the third is filtered again
            d = filters.filterMaps(config.unfilteredMapsDictList, f, tile, diagnosticsDir=None, selFnDir=None, verbose=False,
                                   undoPixelWindow=config.parDict['undoPixelWindow'], tileCache=cache)
            stem = config.filteredMapsDir + os.sep + tile + os.sep + f['label'] + "#" + tile
            fitsio.writeFITS(stem + "_filteredMap.fits", d['data'], d['wcs'].header)
            fitsio.writeFITS(stem + "_SNMap.fits", d['SNMap'], d['wcs'].header)
    finally:
        filters.RETURN_DEVICE_ARRAYS = keep
    before = dict(filters._scaleEntryStats)
    cat2 = pipelines.filterMapsAndMakeCatalogs(config, verbose=False, useCachedFilteredMaps=True)
    assert filters._scaleEntryStats['scales'] == before['scales'] + 1          # only the third filter ran
    _same_catalog(cat, cat2, snrTol=0.0, posTol=0.0)
    for k in ('y_c', 'fixed_y_c', 'fixed_SNR', 'flags', 'numSigPix'):
        ia, ib = np.lexsort((cat['decDeg'], cat['RADeg'])), np.lexsort((cat2['decDeg'], cat2['RADeg']))
        assert np.array_equal(np.asarray(cat[k])[ia], np.asarray(cat2[k])[ib]), k


def test_passes_leave_no_device_memory_behind():
    """A pass's tile state (maps, spectra, noise power, the scales' outputs) is released by reference counting when the pass
    ends - no reference cycle keeps it for the cyclic collector: with the collector switched off, device memory in use after
    the sixth pass equals that after the third."""
    import gc
    mapsL, wcs, truth = helpers.make_tile(SHAPE, freqs=(150.0, 98.0), fwhm=(1.4, 2.1), noise=(12.0, 16.0), seed=3, numSources=5,
                                          kind='clusters')
    # (detection on the calling thread: with the worker thread the moment its last buffers go is not the caller's to see)
    config = _config(helpers.product_maps(mapsL), asyncDetection=False)
    gc.collect()
    gc.disable()
    try:
        used = []
        for k in range(6):
            cat = pipelines.filterMapsAndMakeCatalogs(config, verbose=False)
            del cat
            torch.cuda.synchronize()
            used.append(torch.cuda.memory_allocated())
        assert used[5] <= used[2] + (1 << 20), used
    finally:
        gc.enable()


def test_cached_refilter_passes_leave_no_device_memory_behind():
    """The re-filter mode of an injection run (cached filter + cached noise map) pass after pass: the kept filter object
    stays the first pass's - a pass's tile state is not chained to the next through config.cachedFilters."""
    import gc
    mapsL, wcs, truth = helpers.make_tile(SHAPE, seed=8, numSources=5, kind='clusters')
    config = _config(helpers.product_maps(mapsL), asyncDetection=False)
    config.parDict['mapFilters'] = [f for f in config.parDict['mapFilters'] if f['label'] == config.parDict['photFilter']]
    pipelines.filterMapsAndMakeCatalogs(config, verbose=False)
    kept = config.cachedFilters['PRIMARY']
    gc.collect()
    gc.disable()
    try:
        used = []
        for k in range(6):
            cat = pipelines.filterMapsAndMakeCatalogs(config, useCachedFilters=True, useCachedRMSMap=True, verbose=False)
            del cat
            torch.cuda.synchronize()
            used.append(torch.cuda.memory_allocated())
        assert config.cachedFilters['PRIMARY'] is kept
        assert used[5] <= used[2] + (1 << 20), used
    finally:
        gc.enable()
