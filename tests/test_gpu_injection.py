"""Config C5's injection / recovery leg: maps.sourceInjectionTest (nemo/maps.py:1856-2153) through the driver - models x
iterations of  draw a catalogue -> paint it into the maps (preprocess) -> re-filter with the cached reference filter and the
cached noise map -> find -> match to what went in  - against the oracle doing the same steps on the CPU from the same cached
state (the filter cube, its normalisation and the noise map of the first pass are handed to the oracle: what is under test is
the loop, not the filter build, which has its own tests).

Bars: same recovered rows (rows whose S/N lies within 1e-3 of the threshold may be on one side only); positions 1e-7 deg,
S/N and fluxes 5e-5 relative (float32 maps: complex64 transforms on both sides), separations 1e-4 arcmin, inFlux exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import helpers  # noqa: E402
from nemo_amd import pipelines, startUp, signals, catalogs, maps as pmaps  # noqa: E402
from nemo_oracle import filters as ofilt, catalogs as ocat, photometry as ophot  # noqa: E402
from test_gpu_preprocess import _oracle_steps  # noqa: E402

SHAPE = (300, 420)
THRESH = 4.0


def _cluster_filters():
    base = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20.0}, 'outputUnits': 'yc', 'edgeTrimArcmin': 0.0}
    return [{'label': 'Arnaud_M4e14_z0p2', 'class': 'ArnaudModelMatchedFilter', 'params': dict(base, M500MSun=4e14, z=0.2)},
            {'label': 'Arnaud_M2e14_z0p4', 'class': 'ArnaudModelMatchedFilter', 'params': dict(base, M500MSun=2e14, z=0.4)}]


def _config(mapsL, mapFilters, photFilter, **extra):
    parDict = {'unfilteredMaps': [dict(m) for m in mapsL], 'thresholdSigma': THRESH, 'minObjPix': 1, 'findCenterOfMass': True,
               'useInterpolator': True, 'optimalCatalogMethod': 'ref-forced', 'rejectBorder': 0, 'objIdent': 'ACT-CL',
               'longNames': False, 'photFilter': photFilter, 'mapFilters': mapFilters, 'useTiling': False, 'outputDir': None,
               'sourceInjectionReferenceStream': True}
    parDict.update(extra)
    return startUp.NemoConfig(parDict, makeOutputDirs=False, verbose=False)


def _oracle_refilter(mapsL, wcs, params, kind, filt, signalNorm, RMSMap, profile, GNFWParams, mapUnits):
    """-> refilter(mockRows, model): one cached-filter, cached-noise pass of the driver with the oracle's statements."""
    RADeg, decDeg = wcs.getCentreWCSCoords()
    geom = signals.paintGeometry(SHAPE, wcs, RADeg, decDeg)
    cx, cy = wcs.wcs2pix(RADeg, decDeg)
    xps, yps = pmaps.pixelScalesDeg(wcs)

    def templateFn(i, amp):
        return helpers.cpu_paint(geom, signals.templateTables(kind, mapsL[i]['beamFileName'], params, amplitude=amp))

    def refilter(mockRows, model):
        tab = catalogs.Table(cols={k: [r[k] for r in mockRows] for k in mockRows[0].keys()})
        inj = {'catalog': tab, 'override': model, 'profile': profile}
        if GNFWParams is not None:
            inj['GNFWParams'] = GNFWParams
        om = []
        for m in mapsL:
            data, sm, pm, fm = _oracle_steps(m, wcs, {'injectSources': inj}, m['beamFileName'])
            om.append({'data': data, 'weights': m['weightsData'], 'obsFreqGHz': m['obsFreqGHz'], 'units': m['units'],
                       'surveyMask': sm, 'pointSourceMask': pm})
        mf = ofilt.MatchedFilter(om, params, {'pixSizeDeg': wcs.getPixelSizeDeg(), 'centreYX': (cy, cx), 'xPixScaleDeg': xps,
                                              'yPixScaleDeg': yps}, templateFn, fSZ=signals.fSZ,
                                 deltaT0=lambda f: signals.convertToDeltaT(2e-4, f))
        mf.filt, mf.signalNorm = filt, signalNorm
        res = mf.buildAndApply(prebuilt=True)
        # nemo/pipelines.py:243-250
        SNMap = np.zeros(res['data'].shape) + res['data']
        valid = RMSMap > 0
        SNMap[valid] = SNMap[valid] / RMSMap[valid]
        data = ofilt.undo_pixel_window(res['data'])
        rows = helpers.oracle_find(SNMap, np.zeros(SHAPE, dtype=int), wcs, threshold=THRESH, minObjPix=1, flagRings=False)
        ophot.measure_fluxes(rows, data, wcs.wcs2pix, mapUnits, signals.convertToDeltaT, photSNMap=SNMap, photData=data)
        return rows

    return refilter


def _compare(got, want, fluxScale):
    """Row by row: ``got`` the product's results table, ``want`` the oracle loop's rows (same order: model, iteration, mock row)."""
    gotRows = got.to_records()
    used = set()
    for w in want:
        near = [k for k, g in enumerate(gotRows) if k not in used and g['sourceInjectionModel'] == w['sourceInjectionModel']
                and abs(g['RADeg'] - w['RADeg']) < 1e-7 and abs(g['decDeg'] - w['decDeg']) < 1e-7
                and g['inFlux'] == w['inFlux']]
        if len(near) == 0:
            assert abs(w['SNR'] - THRESH) < 1e-3, "oracle row missing from the product's table: %s" % (w,)
            continue
        g = gotRows[near[0]]
        used.add(near[0])
        assert abs(g['SNR'] / w['SNR'] - 1) < 5e-5
        assert abs(g['outFlux'] - w['outFlux']) < 5e-5 * fluxScale
        assert abs(g['noiseLevel'] / w['noiseLevel'] - 1) < 1e-4
        assert abs(g['rArcmin'] - w['rArcmin']) < 1e-4
        if 'theta500Arcmin' in w:
            assert g['theta500Arcmin'] == w['theta500Arcmin']
        assert g['tileName'] == 'PRIMARY'
    for k, g in enumerate(gotRows):
        assert k in used or abs(g['SNR'] - THRESH) < 1e-3, "product row the oracle does not have: %s" % (g,)
    return len(used)


def test_cluster_injection_two_models_three_iterations_two_frequencies():
    mapsL, wcs, truth = helpers.make_tile(SHAPE, freqs=(98.0, 150.0), fwhm=(2.1, 1.4), noise=(14.0, 11.0), seed=31, numSources=4,
                                          kind='clusters', dtype=np.float32, holes=True)
    config = _config(helpers.product_maps(mapsL), _cluster_filters(), 'Arnaud_M2e14_z0p4',
                     sourceInjectionModels=[{'redshift': 0.4, 'M500': 2e14}, {'redshift': 0.25, 'M500': 5e14}],
                     sourceInjectionIterations=3, sourcesPerTile=7, sourceInjectionAmplitudeRange=[0.3, 4.0])
    realCatalog = pipelines.filterMapsAndMakeCatalogs(config, verbose=False, writeAreaMask=True, writeFlagMask=True)
    assert len(realCatalog) >= 2
    refFilter = config.cachedFilters['PRIMARY']
    filt, signalNorm = refFilter.filt.copy(), refFilter.signalNorm
    RMSMap = config.cachedRMSMapDict['PRIMARY'].cpu().numpy()
    areaMask = config.areaMaskDict['PRIMARY'].cpu().numpy()
    calls = dict(catalogs_before=0)
    from nemo_amd import filters as pfilters
    builds = pfilters._scaleEntryStats['calls']

    np.random.seed(4711)
    got = pmaps.sourceInjectionTest(config, realCatalog=realCatalog, verbose=False)
    # no filter was built in the loop (the scale entry point builds filters; the cached branch never goes there) and the
    # configuration is back to what it was
    assert pfilters._scaleEntryStats['calls'] == builds
    assert len(config.parDict['mapFilters']) == 2 and 'injectSources' not in config.unfilteredMapsDictList[0]

    params = _cluster_filters()[1]['params']
    refilter = _oracle_refilter(mapsL, wcs, params, 'arnaud', filt, signalNorm, RMSMap, 'A10', 'default', 'yc')
    models = [{'redshift': 0.4, 'M500': 2e14}, {'redshift': 0.25, 'M500': 5e14}]
    for m in models:
        m['theta500Arcmin'] = signals.calcTheta500Arcmin(m['redshift'], m['M500'], signals.fiducialCosmoModel)
        m['label'] = '%.2f' % (m['theta500Arcmin'])

    def draw():
        return ocat.generate_test_catalog(['PRIMARY'], {'PRIMARY': areaMask}, {'PRIMARY': wcs}, 7, amplitudeColumnName='y_c',
                                          amplitudeRange=[0.3, 4.0], amplitudeDistribution='linear', maskDilationPix=20)

    np.random.seed(4711)
    want = ocat.source_injection_loop(models, 3, draw, refilter, realCatalog.to_records(), 'y_c', 'err_y_c')
    assert len(want) >= 12                  # most of the 2 x 3 x (up to 7) injected clusters come back
    assert set(got.keys()) == {'RADeg', 'decDeg', 'sourceInjectionModel', 'theta500Arcmin', 'SNR', 'rArcmin', 'inFlux', 'outFlux',
                               'noiseLevel', 'tileName'}
    assert _compare(got, want, fluxScale=4.0) >= 12
    # recovered amplitudes follow the injected ones for the model that matches the reference filter
    sel = np.asarray(got['sourceInjectionModel']) == models[0]['label']
    ratio = np.asarray(got['outFlux'])[sel] / np.asarray(got['inFlux'])[sel]
    assert 0.5 < np.median(ratio) < 1.5


def test_point_source_injection_float64_single_frequency():
    mapsL, wcs, truth = helpers.make_tile(SHAPE, freqs=(150.0,), fwhm=(1.4,), noise=(25.0,), seed=33, numSources=3, kind='sources',
                                          dtype=np.float64)
    params = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20.0}, 'outputUnits': 'uK', 'edgeTrimArcmin': 0.0}
    filtersList = [{'label': 'Beam', 'class': 'BeamMatchedFilter', 'params': dict(params)}]
    config = _config(helpers.product_maps(mapsL), filtersList, 'Beam', sourceInjectionIterations=2, sourcesPerTile=8,
                     sourceInjectionAmplitudeRange=[300, 3000], objIdent='ACT-S')
    realCatalog = pipelines.filterMapsAndMakeCatalogs(config, verbose=False, writeAreaMask=True)
    refFilter = config.cachedFilters['PRIMARY']
    filt, signalNorm = refFilter.filt.copy(), refFilter.signalNorm
    RMSMap = config.cachedRMSMapDict['PRIMARY'].cpu().numpy()
    areaMask = config.areaMaskDict['PRIMARY'].cpu().numpy()

    np.random.seed(99)
    got = pmaps.sourceInjectionTest(config, realCatalog=realCatalog, verbose=False)
    assert 'theta500Arcmin' not in got.keys() and set(np.asarray(got['sourceInjectionModel']).tolist()) == {'pointSource'}

    refilter = _oracle_refilter(mapsL, wcs, params, 'beam', filt, signalNorm, RMSMap, None, None, 'uK')

    def draw():
        return ocat.generate_test_catalog(['PRIMARY'], {'PRIMARY': areaMask}, {'PRIMARY': wcs}, 8, amplitudeColumnName='deltaT_c',
                                          amplitudeRange=[300, 3000], amplitudeDistribution='log', maskDilationPix=20)

    np.random.seed(99)
    want = ocat.source_injection_loop([{'label': 'pointSource'}], 2, draw, refilter, realCatalog.to_records(), 'deltaT_c',
                                      'err_deltaT_c', clusterMode=False)
    assert len(want) >= 10
    assert _compare(got, want, fluxScale=3000.0) >= 10
    ratio = np.asarray(got['outFlux']) / np.asarray(got['inFlux'])
    assert np.abs(np.median(ratio) - 1) < 0.1
    assert np.median(np.asarray(got['rArcmin'])) < 0.3
