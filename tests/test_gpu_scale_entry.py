"""nf_matched_filter_scales (one library call per filter scale, scales dealt to HIP streams) against the same statements
enqueued one by one from the host layer (MatchedFilter.buildAndApply + filterMaps' pixel-window undo): every output bit for
bit - filter cube, normalisation, relative weights, filtered map, S/N map, noise map, mask.  The statement-by-statement path
is the one test_gpu_filters.py / test_gpu_reference_filters.py hold against the oracle and the reference's fixtures; with the
default settings those tests now run through the library call as well."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import helpers  # noqa: E402
from nemo_amd import engine, filters  # noqa: E402

CASES = [
    # shape, freqs, fwhm, noise, kind, class, outputUnits, dtype, extra params, undoPixelWindow
    ((160, 200), (150.0,), (1.4,), (20.0,), 'arnaud', 'ArnaudModelMatchedFilter', 'yc', np.float64, {}, True),
    ((211, 256), (150.0, 98.0), (1.4, 2.1), (15.0, 20.0), 'arnaud', 'ArnaudModelMatchedFilter', 'yc', np.float64,
     {'edgeTrimArcmin': 3.0}, True),
    ((180, 1031), (94.1, 145.3, 225.0), (2.1, 1.4, 1.0), (20.0, 15.0, 40.0), 'arnaud', 'ArnaudModelMatchedFilter', 'yc',
     np.float64, {}, True),                      # 1031 is prime: the window undo stays on the 2-D transform path
    ((211, 256), (150.0,), (1.4,), (20.0,), 'beam', 'BeamMatchedFilter', 'uK', np.float64, {'edgeTrimArcmin': 2.0}, True),
    ((211, 256), (150.0, 98.0), (1.4, 2.1), (15.0, 20.0), 'arnaud', 'ArnaudModelMatchedFilter', 'yc', np.float32, {}, False),
    ((160, 200), (150.0,), (1.4,), (20.0,), 'beam', 'BeamMatchedFilter', 'uK', np.float32, {}, True),
    # one noise bin: the block form of the noise map, which takes the scales of a group as its batch dimension too
    ((211, 256), (150.0, 98.0), (1.4, 2.1), (15.0, 20.0), 'arnaud', 'ArnaudModelMatchedFilter', 'yc', np.float32,
     {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20., 'numNoiseBins': 1}}, True),
    ((160, 200), (150.0,), (1.4,), (20.0,), 'arnaud', 'ArnaudModelMatchedFilter', 'yc', np.float64,
     {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 15., 'numNoiseBins': 1}, 'edgeTrimArcmin': 2.0}, True),
    ((200, 240), (150.0, 98.0), (1.4, 2.1), (15.0, 20.0), 'beam', 'BeamMatchedFilter', 'uK', np.float32,
     {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20., 'numNoiseBins': 1}}, True),
]


def _run(mapsL, filtersList, undo, lanes, entry, batch=None):
    keepEntry, keepDev, keepBatch = filters._SCALE_ENTRY, filters.RETURN_DEVICE_ARRAYS, engine.SCALE_BATCH
    filters._SCALE_ENTRY, filters.RETURN_DEVICE_ARRAYS, engine.SCALE_BATCH = entry, True, batch
    try:
        res = filters.filterMapsGroup(helpers.product_maps(mapsL), filtersList, 'PRIMARY', diagnosticsDir=None, selFnDir=None,
                                      verbose=False, undoPixelWindow=undo, tileCache={}, lanes=lanes)
        out = []
        for d, fobj in res:
            out.append({'data': d['data'].cpu().numpy(), 'SNMap': d['SNMap'].cpu().numpy(), 'RMSMap': d['RMSMap'].cpu().numpy(),
                        'surveyMask': d['surveyMask'].cpu().numpy(), 'filt': fobj.filt.copy(), 'signalNorm': fobj.signalNorm,
                        'fRelWeights': dict(fobj.fRelWeights.items()), 'units': d['mapUnits']})
        torch.cuda.synchronize()
        return out
    finally:
        filters._SCALE_ENTRY, filters.RETURN_DEVICE_ARRAYS, engine.SCALE_BATCH = keepEntry, keepDev, keepBatch


@pytest.mark.parametrize("case", CASES, ids=[str(i) for i in range(len(CASES))])
def test_scale_entry_equals_statement_path(case):
    shape, freqs, fwhm, noise, kind, cls, units, dtype, extra, undo = case
    srcKind = 'clusters' if kind == 'arnaud' else 'sources'
    mapsL, wcs, truth = helpers.make_tile(shape, freqs=freqs, fwhm=fwhm, noise=noise, seed=5, numSources=4, kind=srcKind,
                                          dtype=dtype, holes=True)
    base = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20., 'numNoiseBins': 2}, 'outputUnits': units}
    base.update(extra)
    filtersList = []
    for k, (z, m) in enumerate([(0.4, 2e14), (0.2, 4e14), (0.8, 1e14), (0.4, 6e14), (1.2, 2e14)]):
        p = dict(base)
        p.update({'z': z, 'M500MSun': m})
        filtersList.append({'label': 'scale%d' % k, 'class': cls, 'params': p})
    if kind == 'beam':
        filtersList = filtersList[:2]       # the beam template has no scale: two identical filters
    before = dict(filters._scaleEntryStats)
    ref = _run(mapsL, filtersList, undo, 1, False)
    assert filters._scaleEntryStats == before          # the statement-by-statement path
    # (lanes, scales per group): scale after scale; groups of 8 (all five scales in one launch per stage), of 2 and of 3
    for lanes, batch in ((1, 1), (3, 1), (1, 8), (2, 2), (1, 3)):
        before = dict(filters._scaleEntryStats)
        got = _run(mapsL, filtersList, undo, lanes, True, batch)
        # all scales of the bank went out in ONE library call
        assert filters._scaleEntryStats['calls'] == before['calls'] + 1
        assert filters._scaleEntryStats['scales'] == before['scales'] + len(filtersList)
        assert len(got) == len(ref)
        for g, r in zip(got, ref):
            assert g['units'] == r['units']
            assert type(g['signalNorm']) == type(r['signalNorm']) and g['signalNorm'] == r['signalNorm']
            assert g['fRelWeights'].keys() == r['fRelWeights'].keys()
            for f in r['fRelWeights']:
                assert type(g['fRelWeights'][f]) == type(r['fRelWeights'][f]) and g['fRelWeights'][f] == r['fRelWeights'][f]
            for key in ('filt', 'data', 'SNMap', 'RMSMap', 'surveyMask'):
                assert g[key].dtype == r[key].dtype and g[key].shape == r[key].shape, key
                assert np.array_equal(g[key], r[key]), (key, lanes, batch)
    # the scales differ from each other (the comparison above is not between copies of one map)
    if kind == 'arnaud':
        assert not np.array_equal(ref[0]['data'], ref[1]['data'])


def test_scale_entry_mixed_group_and_fallbacks():
    """A group whose members cannot share one call (different noise grids, a background-subtracted filter in between) is
    split into runs; results in the order given, each equal to its own filterMaps call."""
    shape = (200, 240)
    mapsL, wcs, truth = helpers.make_tile(shape, freqs=(150.0, 98.0), fwhm=(1.4, 2.1), noise=(15.0, 20.0), seed=9, numSources=4,
                                          kind='clusters', dtype=np.float64, holes=True)
    n20 = {'method': 'dataMap', 'noiseGridArcmin': 20., 'numNoiseBins': 1}
    n40 = {'method': 'dataMap', 'noiseGridArcmin': 40., 'numNoiseBins': 1}
    specs = [(n20, {}), (n20, {}), (n40, {}), (n20, {'bckSub': True, 'bckSubScaleArcmin': 6.0}), (n20, {})]
    filtersList = []
    for k, (npar, extra) in enumerate(specs):
        p = {'noiseParams': dict(npar), 'outputUnits': 'yc', 'z': 0.3 + 0.1 * k, 'M500MSun': 2e14}
        p.update(extra)
        filtersList.append({'label': 'f%d' % k, 'class': 'ArnaudModelMatchedFilter', 'params': p})
    before = dict(filters._scaleEntryStats)
    got = _run(mapsL, filtersList, True, 2, True)
    # runs: (f0, f1), (f2), f3 statement by statement, (f4)
    assert filters._scaleEntryStats['calls'] == before['calls'] + 3 and filters._scaleEntryStats['scales'] == before['scales'] + 4
    for k, f in enumerate(filtersList):
        one = _run(mapsL, [f], True, 1, False)[0]
        for key in ('filt', 'data', 'SNMap', 'RMSMap', 'surveyMask'):
            assert np.array_equal(got[k][key], one[key]), (k, key)
        assert got[k]['signalNorm'] == one['signalNorm']


def test_scale_entry_without_the_filter_cube():
    """keepFilters: the scales whose filter cube nobody will read apply their filter without storing it - same maps bit for
    bit, ``filt`` is None for them and present for the kept label."""
    shape = (211, 256)
    mapsL, wcs, truth = helpers.make_tile(shape, freqs=(150.0, 98.0), fwhm=(1.4, 2.1), noise=(15.0, 20.0), seed=11, numSources=4,
                                          kind='clusters', dtype=np.float64, holes=True)
    base = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20., 'numNoiseBins': 1}, 'outputUnits': 'yc'}
    filtersList = []
    for k, (z, m) in enumerate([(0.4, 2e14), (0.2, 4e14), (0.8, 1e14)]):
        p = dict(base)
        p.update({'z': z, 'M500MSun': m})
        filtersList.append({'label': 'scale%d' % k, 'class': 'ArnaudModelMatchedFilter', 'params': p})
    ref = _run(mapsL, filtersList, True, 2, True)
    keepDev = filters.RETURN_DEVICE_ARRAYS
    filters.RETURN_DEVICE_ARRAYS = True
    try:
        res = filters.filterMapsGroup(helpers.product_maps(mapsL), filtersList, 'PRIMARY', diagnosticsDir=None, selFnDir=None,
                                      verbose=False, undoPixelWindow=True, tileCache={}, lanes=2, keepFilters={'scale1'})
        for k, (d, fobj) in enumerate(res):
            for key in ('data', 'SNMap', 'RMSMap', 'surveyMask'):
                assert np.array_equal(d[key].cpu().numpy(), ref[k][key]), (k, key)
            assert fobj.signalNorm == ref[k]['signalNorm']
            if k == 1:
                assert np.array_equal(fobj.filt, ref[k]['filt'])
            else:
                assert fobj.filt is None
    finally:
        filters.RETURN_DEVICE_ARRAYS = keepDev


def test_stale_noise_tables_are_caught_in_debug_mode():
    """The noise block tables a lane keeps stand for the tile's weights under a host-side token (tileCache).  With
    engine.CHECK_NOISE_TABLES (NEMO_AMD_CHECK_NOISE_TABLES=1) the weights' bits are summed per call: weights edited in place
    between two calls under the same token raise instead of reusing tables made for other weights."""
    mapsL, wcs, truth = helpers.make_tile((160, 200), freqs=(150.0,), fwhm=(1.4,), noise=(20.0,), seed=5, numSources=3, kind='clusters',
                                          dtype=np.float64, holes=True)
    base = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 15., 'numNoiseBins': 1}, 'outputUnits': 'yc'}
    filtersList = [{'label': 's%d' % k, 'class': 'ArnaudModelMatchedFilter', 'params': dict(base, z=z, M500MSun=m)}
                   for k, (z, m) in enumerate([(0.4, 2e14), (0.2, 4e14)])]
    keep = (filters._SCALE_ENTRY, filters.RETURN_DEVICE_ARRAYS, engine.CHECK_NOISE_TABLES)
    filters._SCALE_ENTRY, filters.RETURN_DEVICE_ARRAYS, engine.CHECK_NOISE_TABLES = True, True, True
    try:
        tileCache = {}
        args = dict(diagnosticsDir=None, selFnDir=None, verbose=False, undoPixelWindow=True, tileCache=tileCache, lanes=1)
        maps = helpers.product_maps(mapsL)
        filters.filterMapsGroup(maps, filtersList[:1], 'PRIMARY', **args)
        filters.filterMapsGroup(maps, filtersList[1:], 'PRIMARY', **args)          # same weights, same token: tables kept, no complaint
        shared = [v for k, v in tileCache.items() if isinstance(k, tuple) and k[0] == 'tile'][0]
        shared['medWeights'][40:60, 50:90] *= 0.5                                     # edited in place under the same token
        with pytest.raises(engine.NemoFiltError, match="stale"):
            filters.filterMapsGroup(maps, filtersList[:1], 'PRIMARY', **args)
        torch.cuda.synchronize()
    finally:
        filters._SCALE_ENTRY, filters.RETURN_DEVICE_ARRAYS, engine.CHECK_NOISE_TABLES = keep


@pytest.mark.parametrize("kind,cls,units,dtype", [('arnaud', 'ArnaudModelMatchedFilter', 'yc', np.float64),
                                                  ('beam', 'BeamMatchedFilter', 'uK', np.float64),
                                                  ('arnaud', 'ArnaudModelMatchedFilter', 'yc', np.float32)])
def test_scale_entry_long_columns_in_streaming_passes(kind, cls, units, dtype):
    """4096-long columns (the headline tile's): the template path of a one-frequency scale runs as three streaming passes
    (k_template2p_*) and the data inverse's column transform as two (k_cols2p_*), both 64 x 64 factorisations of the column
    transform.  The statement-by-statement path transforms the templates with the single-pass column kernels, so the two agree
    to float32 round-off of the template spectra (not bit for bit): filter cube, normalisation, filtered map, S/N map; the noise
    map and the masks follow from those.  160 columns: three column tiles, the last one partial, with the packed column 0 / W/2."""
    shape = (4096, 160)
    srcKind = 'clusters' if kind == 'arnaud' else 'sources'
    mapsL, wcs, truth = helpers.make_tile(shape, freqs=(150.0,), fwhm=(1.4,), noise=(20.0,), seed=9, numSources=6, kind=srcKind,
                                          dtype=dtype, holes=True)
    base = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20., 'numNoiseBins': 1}, 'outputUnits': units}
    filtersList = []
    for k, (z, m) in enumerate([(0.4, 2e14), (0.2, 4e14), (0.8, 1e14)]):
        p = dict(base)
        p.update({'z': z, 'M500MSun': m})
        filtersList.append({'label': 'scale%d' % k, 'class': cls, 'params': p})
    if kind == 'beam':
        filtersList = filtersList[:2]
    ref = _run(mapsL, filtersList, True, 1, False)
    tol = 2e-6 if dtype == np.float64 else 2e-5
    for lanes, batch in ((1, 1), (2, 2), (1, 8)):
        got = _run(mapsL, filtersList, True, lanes, True, batch)
        for g, r in zip(got, ref):
            assert abs(g['signalNorm'] / r['signalNorm'] - 1) < tol, (g['signalNorm'], r['signalNorm'])
            assert np.abs(g['filt'] - r['filt']).max() <= tol * np.abs(r['filt']).max()
            assert np.array_equal(g['surveyMask'], r['surveyMask'])
            for key in ('data', 'SNMap', 'RMSMap'):
                assert np.abs(g[key].astype(np.float64) - r[key]).max() <= 5 * tol * np.abs(r[key]).max(), (key, lanes, batch)
