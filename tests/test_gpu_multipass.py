"""A DR6-shaped multi-pass run (config C5 of the scope table: examples/ACT-DR6-clusters/DR6ClusterSearch.yml:133-156) through the
driver: per frequency a bright point-source pass with shapes, a deeper pass on the source-subtracted / hole-filled map, then the
cluster passes on maps with the sources subtracted and the holes filled, the last one with the clusters of the pass before taken
out of the noise term (noiseModelCatalog).  The last pass is rebuilt with the oracle from the catalogues the earlier passes
produced and compared row by row."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import helpers  # noqa: E402
from nemo_amd import pipelines, startUp, signals, catalogs, maps as pmaps  # noqa: E402
from nemo_oracle import filters as ofilt, maps as omaps  # noqa: E402
from test_gpu_preprocess import _oracle_steps  # noqa: E402

SHAPE = (240, 480)
CLUSTER = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20.0}, 'outputUnits': 'yc', 'edgeTrimArcmin': 0.0,
           'M500MSun': 2e14, 'z': 0.4, 'saveFilteredMaps': True}
BEAM = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20.0}, 'outputUnits': 'uK', 'edgeTrimArcmin': 0.0}


def _tile():
    mapsL, wcs, truth = helpers.make_tile(SHAPE, freqs=(98.0, 150.0), fwhm=(2.1, 1.4), noise=(14.0, 11.0), seed=21, numSources=6,
                                          kind='clusters', dtype=np.float32)
    mapsL[0]['label'], mapsL[1]['label'] = 'f090', 'f150'
    rng = np.random.default_rng(22)
    srcX, srcY = rng.uniform(40, SHAPE[1] - 40, 5), rng.uniform(40, SHAPE[0] - 40, 5)
    amps = np.array([25000.0, 9000.0, 4000.0, 2500.0, 600.0])
    for x, y, a in zip(srcX, srcY, amps):
        ra, dec = wcs.pix2wcs(x, y)
        geom = signals.paintGeometry(SHAPE, wcs, ra, dec)
        for m in mapsL:
            m['mapData'] = m['mapData'] + helpers.cpu_paint(geom, signals.templateTables('beam', m['beamFileName'], {}, amplitude=a))
    return mapsL, wcs, (srcX, srcY, amps)


def _config(mapsL):
    strong = {'thresholdSigma': 30.0, 'objIdent': 'ACT-S', 'measureShapes': True, 'ignoreSurveyMask': True}
    deep = {'thresholdSigma': 5.0, 'objIdent': 'ACT-S', 'maskHoleDilationFactor': 3.0}
    opts = {1: dict(strong, mapToUse='f090'),
            2: dict(deep, mapToUse='f090', subtractModelFromSets=[1], addSiphonedFromSets=[1], maskAndFillFromSets=[1]),
            3: dict(strong, mapToUse='f150'),
            4: dict(deep, mapToUse='f150', subtractModelFromSets=[3], addSiphonedFromSets=[3], maskAndFillFromSets=[3]),
            10: {'subtractModelFromSets': [2, 4], 'maskAndFillFromSets': [1, 3], 'maskHoleDilationFactor': 3.0},
            11: {'subtractModelFromSets': [2, 4], 'maskAndFillFromSets': [1, 3], 'maskHoleDilationFactor': 3.0,
                 'noiseModelCatalogFromSets': [10]}}
    parDict = {'unfilteredMaps': [dict(m) for m in mapsL], 'thresholdSigma': 4.0, 'minObjPix': 1, 'findCenterOfMass': True,
               'useInterpolator': True, 'optimalCatalogMethod': 'ref-forced', 'rejectBorder': 0, 'objIdent': 'ACT-CL',
               'longNames': False, 'photFilter': 'Arnaud_M2e14_z0p4', 'filterSetOptions': opts, 'useTiling': False, 'outputDir': None,
               'mapFilters': [{'label': 'BeamPass1', 'class': 'BeamMatchedFilter', 'params': dict(BEAM), 'filterSets': [1, 3]},
                              {'label': 'BeamPass2', 'class': 'BeamMatchedFilter', 'params': dict(BEAM), 'filterSets': [2, 4]},
                              {'label': 'Arnaud_M2e14_z0p4', 'class': 'ArnaudModelMatchedFilter', 'params': dict(CLUSTER),
                               'filterSets': [10, 11]}]}
    return startUp.NemoConfig(parDict, makeOutputDirs=False, verbose=False)


def test_dr6_shaped_multipass_run_last_pass_against_the_oracle():
    mapsL, wcs, (srcX, srcY, amps) = _tile()
    config = _config(helpers.product_maps(mapsL))
    assert config.filterSets == [1, 2, 3, 4, 10, 11]
    final = pipelines.filterMapsAndMakeCatalogs(config, verbose=False)
    cats = {k: config.filterSetOptions[k]['catalog'] for k in config.filterSets}

    # pass 1 / 3: the bright sources, with shapes; pass 2 / 4 reach the faint one and carry the siphoned bright ones along
    for first, second in ((1, 2), (3, 4)):
        assert 3 <= len(cats[first]) <= 12 and 'ellipse_A' in cats[first].keys() and 'deltaT_c' in cats[first].keys()
        if first == 1:          # (at 150 GHz the sharper filter leaves no segment above 30 sigma with more than 9 pixels)
            assert (np.asarray(cats[first]['ellipse_A']) > 0).any()
        assert len(cats[second]) >= len(cats[first])
        sx, sy = wcs.wcs2pix(np.asarray(cats[second]['RADeg']), np.asarray(cats[second]['decDeg']))
        for x, y in zip(srcX, srcY):
            assert np.hypot(sx - x, sy - y).min() < 1.5
    assert len(final) >= 3 and 'y_c' in final.keys()

    # ---- pass 11 with the oracle, from the catalogues of passes 1-4 and 10 ------------------------------------------------
    pre = []
    for m, sub, holes in zip(mapsL, (2, 4), (1, 3)):
        keys = {'subtractModelFromCatalog': [cats[sub]], 'maskAndFillFromCatalog': [cats[holes]], 'maskHoleDilationFactor': 3.0}
        pre.append(_oracle_steps(m, wcs, keys, m['beamFileName']))
    assert (pre[0][1] == 0).sum() > 100           # holes were cut around the bright sources
    noiseRows = cats[10].to_records()
    noiseData, om = [], []
    for m, (data, sm, pm, fm) in zip(mapsL, pre):
        beam = m['beamFileName']

        def objectsFor(rr, beam=beam, freq=m['obsFreqGHz']):
            tab = catalogs.Table(cols={k: [r[k] for r in rr] for k in rr[0].keys()})
            return pmaps._modelObjects(tab, beam, freq, 'default', 'A10', None, 0, 5, signals.fiducialCosmoModel)

        model = omaps.make_model_image(SHAPE, wcs, noiseRows, objectsFor, signals.paintGeometry)
        noiseData.append(data - model if model is not None else data)
        om.append({'data': data, 'weights': m['weightsData'], 'obsFreqGHz': m['obsFreqGHz'], 'units': m['units'],
                   'surveyMask': pre[0][1], 'pointSourceMask': pre[0][2]})
    RADeg, decDeg = wcs.getCentreWCSCoords()
    geom = signals.paintGeometry(SHAPE, wcs, RADeg, decDeg)

    def templateFn(i, amp):
        return helpers.cpu_paint(geom, signals.templateTables('arnaud', mapsL[i]['beamFileName'], CLUSTER, amplitude=amp))

    cx, cy = wcs.wcs2pix(RADeg, decDeg)
    xps, yps = pmaps.pixelScalesDeg(wcs)
    mf = ofilt.MatchedFilter(om, CLUSTER, {'pixSizeDeg': wcs.getPixelSizeDeg(), 'centreYX': (cy, cx), 'xPixScaleDeg': xps,
                                           'yPixScaleDeg': yps}, templateFn, fSZ=signals.fSZ,
                             deltaT0=lambda f: signals.convertToDeltaT(2e-4, f))
    mf.build(noiseData=noiseData)
    res = mf.buildAndApply(prebuilt=True)
    flagMask = pre[0][3].astype(int) * 1 + pre[1][3].astype(int) * 2
    # (1) the S/N map of the last pass against the oracle's
    gotSN = config.stitchedSNMapDict['PRIMARY'].cpu().numpy()
    assert np.abs(gotSN - res['SNMap']).max() < 2e-5 * np.abs(res['SNMap']).max()
    # (2) the catalogue against the oracle's detection on that same S/N map: same places, same S/N, same flags (flagRings is off
    # unless the configuration asks for it, nemo/startUp.py:165-166)
    rows = helpers.oracle_find(gotSN, flagMask, wcs, threshold=4.0, minObjPix=1, flagRings=False)
    refRA, refDec = np.array([r['RADeg'] for r in rows]), np.array([r['decDeg'] for r in rows])
    used = set()
    for i in range(len(final)):
        j = int(np.argmin(np.hypot(refRA - final['RADeg'][i], refDec - final['decDeg'][i])))
        r = rows[j]
        assert abs(final['RADeg'][i] - r['RADeg']) < 1e-9 and abs(final['decDeg'][i] - r['decDeg']) < 1e-9
        assert abs(final['SNR'][i] / r['SNR'] - 1) < 2e-7          # (the kept map is the float32 copy of the one detected on)
        assert int(final['flags'][i]) == int(r['flags']) and bool(final['ringFlag'][i]) == bool(r['ringFlag'])
        used.add(j)
    assert len(used) == len(final) >= 20
    # what the oracle's list has beyond that: the weaker member of a pair closer than the cross-match radius, which
    # catalogs.makeOptimalCatalog merges (nemo/catalogs.py:218-362)
    for j, r in enumerate(rows):
        if j in used:
            continue
        sep = np.hypot(refRA - r['RADeg'], refDec - r['decDeg'])
        near = [k for k in np.flatnonzero(sep < catalogs.XMATCH_RADIUS_DEG) if k != j]
        assert any(rows[k]['SNR'] > r['SNR'] for k in near)
