"""Catalogue-driven preprocessing on the device (maps.MapDict.preprocess nemo/maps.py:325-465,
maps.makeModelImage :1658-1853) against the oracle restatement (oracle/nemo_oracle/maps.py).

Bars: the median background, the holes (masks and filled data) and the painted model before the pixel window are integer /
selection / unfused float work -> bit for bit.  After enmap.apply_window (an FFT pair) the model agrees to 2e-6 of its peak
(float32 model) and what is built from it inherits that tolerance."""
import numpy as np
import pytest
from scipy import ndimage

from nemo_amd import engine, maps as pmaps, signals, catalogs
from nemo_oracle import maps as omaps

import helpers

pytestmark = pytest.mark.gpu


def _dev(x, dtype=None):
    return engine.to_dev(np.ascontiguousarray(x), dtype=dtype)


@pytest.mark.parametrize("shape,size,dtype,quant", [
    ((300, 257), 20, np.float32, None), ((300, 257), 20, np.float64, None), ((67, 130), 5, np.float64, 8), ((64, 64), 3, np.float32, 4),
    ((33, 200), 20, np.float32, 16), ((150, 70), 13, np.float64, None), ((128, 128), 1, np.float32, None), ((90, 90), 48, np.float32, 32),
    ((12, 40), 20, np.float64, None),
])
def test_median_filter_is_scipys(shape, size, dtype, quant):
    rng = np.random.default_rng(shape[0] * 1000 + size)
    x = rng.normal(0, 50, shape)
    if quant is not None:
        x = np.round(x / quant) * quant          # many equal values: rank selection among ties
    x[5:9, 3:30] = 0.0
    x = x.astype(dtype)
    eng = engine.TileEngine.get(shape)
    got = eng.median_filter(_dev(x), size).cpu().numpy()
    want = ndimage.median_filter(x, size)
    assert got.dtype == want.dtype
    assert np.array_equal(got, want)


def _tile(shape=(240, 300), dtype=np.float32, seed=3):
    mapsL, wcs, truth = helpers.make_tile(shape, freqs=(150.0,), fwhm=(1.4,), noise=(25.0,), seed=seed, numSources=4, kind='sources',
                                          dtype=dtype, holes=True)
    return mapsL[0], wcs


def _positions(wcs, shape, n, seed, margin=0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-margin, shape[1] + margin, n)
    y = rng.uniform(-margin, shape[0] + margin, n)
    ra, dec = wcs.pix2wcs(x, y)
    return np.asarray(ra), np.asarray(dec)


def _oracle_steps(m, wcs, keys, beam):
    """oracle catalogue_steps on the same tile (inputs as preprocess has them just before :325)."""
    data = m['mapData'].copy()
    weights = m['weightsData']
    data[weights == 0] = 0
    md = {'obsFreqGHz': m['obsFreqGHz']}
    for k, v in keys.items():
        if isinstance(v, catalogs.Table):
            v = [v.to_records()]          # (a list of rows is the oracle's catalogue: a lone one goes in as the list of one)
        elif isinstance(v, list):
            v = [(dict(e, path=e['path'].to_records()) if isinstance(e, dict) else e.to_records()) for e in v]
        elif isinstance(v, dict) and 'catalog' in v:
            v = dict(v, catalog=v['catalog'].to_records())
        md[k] = v

    def modelImage(rows, cluster=False, obsFreqGHz=None, GNFWParams='default', profile='A10', override=None, validAreaSection=None):
        def objectsFor(rr):
            tab = catalogs.Table(cols={k: [r[k] for r in rr] for k in rr[0].keys()})
            objs = pmaps._modelObjects(tab, beam, obsFreqGHz, GNFWParams, profile, override, 0, 5, signals.fiducialCosmoModel)
            if override is not None:
                objs = [o[:4] + (None,) for o in objs]
            return objs
        whole = None
        if override is not None and obsFreqGHz is not None:
            whole = (signals.fSZ(obsFreqGHz, z=override['redshift']), signals.TCMB * 1e6)
        return omaps.make_model_image(data.shape, wcs, rows, objectsFor, signals.paintGeometry, validAreaSection=validAreaSection,
                                      wholeMapFactors=whole)

    return omaps.catalogue_steps(md, data, weights, wcs, m['surveyMask'].copy(), m['pointSourceMask'].copy(),
                                 np.zeros(data.shape, dtype=np.uint8), modelImage)


def _product_steps(m, keys):
    md = pmaps.MapDict(dict(m, **keys))
    md.preprocess()
    return md


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_mask_and_fill_from_catalog(dtype):
    m, wcs = _tile(dtype=dtype)
    shape = m['mapData'].shape
    ra, dec = _positions(wcs, shape, 40, seed=11, margin=15)          # some rows outside the map: dropped
    tab = catalogs.Table(cols={'RADeg': ra, 'decDeg': dec})
    keys = {'maskAndFillFromCatalog': tab, 'maskHoleRadiusArcmin': 3.0, 'maskHoleDilationFactor': 1.5}
    md = _product_steps(m, keys)
    data, sm, pm, fm = _oracle_steps(m, wcs, keys, m['beamFileName'])
    assert md['data'].cpu().numpy().dtype == dtype
    assert (sm == 0).sum() > (m['surveyMask'] == 0).sum() + 500
    assert np.array_equal(md['surveyMask'].cpu().numpy(), sm)
    assert np.array_equal(md['pointSourceMask'].cpu().numpy(), pm)
    assert np.array_equal(md['data'].cpu().numpy(), data)
    assert np.array_equal(md['flagMask'].cpu().numpy(), fm)


def test_mask_and_fill_hole_size_from_shapes():
    m, wcs = _tile(dtype=np.float32, seed=4)
    shape = m['mapData'].shape
    ra, dec = _positions(wcs, shape, 12, seed=5)
    rng = np.random.default_rng(6)
    tab = catalogs.Table(cols={'RADeg': ra, 'decDeg': dec, 'ellipse_A': rng.uniform(2, 14, 12), 'ellipse_B': rng.uniform(1, 2, 12)})
    keys = {'maskAndFillFromCatalog': [tab]}
    md = _product_steps(m, keys)
    data, sm, pm, fm = _oracle_steps(m, wcs, keys, m['beamFileName'])
    assert (pm == 0).sum() > (m['pointSourceMask'] == 0).sum()
    assert np.array_equal(md['surveyMask'].cpu().numpy(), sm)
    assert np.array_equal(md['pointSourceMask'].cpu().numpy(), pm)
    assert np.array_equal(md['data'].cpu().numpy(), data)
    # neither sizes nor a radius: the reference's exception
    bare = catalogs.Table(cols={'RADeg': ra, 'decDeg': dec})
    with pytest.raises(Exception, match="maskHoleRadiusArcmin"):
        _product_steps(m, {'maskAndFillFromCatalog': bare})


def test_mask_point_sources_from_catalog_keeps_its_distance_map():
    """The reference's distance map lives through the whole loop (nemo/maps.py:391): a later, larger radius also cuts pixels
    around earlier objects.  Radii rise and fall and boxes overlap, so every branch of the re-test schedule is met."""
    m, wcs = _tile(shape=(200, 260), dtype=np.float64, seed=8)
    shape = m['mapData'].shape
    ra, dec = _positions(wcs, shape, 60, seed=21, margin=10)
    rng = np.random.default_rng(22)
    rArcmin = rng.choice([1.0, 2.5, 4.0, 6.0, 9.0], 60)
    flux = rng.uniform(0, 0.05, 60)
    tabA = catalogs.Table(cols={'RADeg': ra[:40], 'decDeg': dec[:40], 'rArcmin': rArcmin[:40], 'fluxJy': flux[:40]})
    tabB = catalogs.Table(cols={'RADeg': ra[40:], 'decDeg': dec[40:], 'ellipse_A': rng.uniform(4, 30, 20)})
    keys = {'maskPointSourcesFromCatalog': [{'path': tabA, 'fluxCutJy': 0.01}, tabB]}
    md = _product_steps(m, keys)
    data, sm, pm, fm = _oracle_steps(m, wcs, keys, m['beamFileName'])
    # the persistent map matters on this input: independent holes would mask fewer pixels
    indep = np.ones(shape, dtype=np.uint8)
    for row in omaps.catalog_within_image([r for r in tabA.to_records() if r['fluxJy'] > 0.01], shape, wcs):
        rmap, _, _ = omaps.degrees_distance_map(np.ones(shape) * 1e6, wcs, row['RADeg'], row['decDeg'], row['rArcmin'] / 60)
        indep[rmap < row['rArcmin'] / 60.0] = 0
    assert (pm == 0).sum() > (indep == 0).sum()
    assert np.array_equal(md['pointSourceMask'].cpu().numpy(), pm)
    assert np.array_equal(md['surveyMask'].cpu().numpy(), sm)
    assert np.array_equal(md['data'].cpu().numpy(), data)


def _cluster_table(wcs, shape, n, seed):
    ra, dec = _positions(wcs, shape, n, seed=seed, margin=20)
    rng = np.random.default_rng(seed + 1)
    templates = np.array(['Arnaud_M2e14_z0p4', 'Arnaud_M4e14_z0p2', 'Arnaud_M1e14_z0p8#extra'])[rng.integers(0, 3, n)]
    return catalogs.Table(cols={'RADeg': ra, 'decDeg': dec, 'y_c': rng.uniform(0.3, 3.0, n), 'template': templates,
                                'SNR': rng.uniform(3, 20, n)})


@pytest.mark.parametrize("kind", ["sources", "clusters_y", "clusters_uK", "override"])
def test_make_model_image(kind):
    shape = (220, 280)
    m, wcs = _tile(shape=shape, dtype=np.float32, seed=12)
    beam = m['beamFileName']
    kw, okw, whole = {}, {}, None
    if kind == "sources":
        ra, dec = _positions(wcs, shape, 50, seed=30, margin=20)
        tab = catalogs.Table(cols={'RADeg': ra, 'decDeg': dec, 'deltaT_c': np.random.default_rng(31).uniform(-300, 3000, 50)})
        args = (None, 'default', 'A10', None)
    else:
        tab = _cluster_table(wcs, shape, 14, seed=40)
        obsFreq = None if kind == "clusters_y" else 148.0
        override = {'redshift': 0.3, 'M500': 3e14} if kind == "override" else None
        args = (obsFreq, 'default', 'A10', override)
        kw = dict(obsFreqGHz=obsFreq, override=override, minSNR=5.0, validAreaSection=[10, 270, 5, 215])
        okw = dict(minSNR=5.0, validAreaSection=[10, 270, 5, 215])
        if override is not None:
            whole = (signals.fSZ(obsFreq, z=0.3), signals.TCMB * 1e6)

    def objectsFor(rr):
        t = catalogs.Table(cols={k: [r[k] for r in rr] for k in rr[0].keys()})
        objs = pmaps._modelObjects(t, beam, args[0], args[1], args[2], args[3], 0, 5, signals.fiducialCosmoModel)
        return [o[:4] + (None,) for o in objs] if args[3] is not None else objs

    rows = tab.to_records()
    for pixwin in (False, True):
        got = pmaps.makeModelImage(shape, wcs, tab, beam, applyPixelWindow=pixwin, **kw)
        want = omaps.make_model_image(shape, wcs, rows, objectsFor, signals.paintGeometry, wholeMapFactors=whole,
                                      applyPixelWindow=pixwin, **okw)
        got = got.cpu().numpy()
        assert got.dtype == want.dtype
        assert np.abs(want).max() > 0
        if not pixwin:
            assert np.array_equal(got, want)
        else:
            assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()


def test_make_model_image_of_nothing_is_none():
    m, wcs = _tile()
    shape = m['mapData'].shape
    ra, dec = wcs.pix2wcs(np.array([-50.0, shape[1] + 40.0]), np.array([10.0, 20.0]))
    tab = catalogs.Table(cols={'RADeg': np.asarray(ra), 'decDeg': np.asarray(dec), 'deltaT_c': [100.0, 200.0]})
    assert pmaps.makeModelImage(shape, wcs, tab, m['beamFileName']) is None


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_subtract_model_and_inject_sources(dtype):
    """All catalogue steps of one preprocess call: clusters injected (override form, as sourceInjectionTest asks), point sources
    subtracted and flagged, then holes cut and filled."""
    m, wcs = _tile(shape=(220, 280), dtype=dtype, seed=14)
    shape = m['mapData'].shape
    ra, dec = _positions(wcs, shape, 25, seed=50)
    srcTab = catalogs.Table(cols={'RADeg': ra, 'decDeg': dec, 'deltaT_c': np.random.default_rng(51).uniform(50, 4000, 25)})
    injTab = _cluster_table(wcs, shape, 8, seed=60)
    hra, hdec = _positions(wcs, shape, 6, seed=70)
    holes = catalogs.Table(cols={'RADeg': hra, 'decDeg': hdec})
    keys = {'injectSources': {'catalog': injTab, 'GNFWParams': 'default', 'profile': 'A10', 'override': {'redshift': 0.4, 'M500': 2e14}},
            'subtractModelFromCatalog': [srcTab], 'maskAndFillFromCatalog': holes, 'maskHoleRadiusArcmin': 2.5}
    md = _product_steps(m, keys)
    data, sm, pm, fm = _oracle_steps(m, wcs, keys, m['beamFileName'])
    got = md['data'].cpu().numpy()
    assert got.dtype == data.dtype == np.float64          # the override branch hands back a float64 model (numpy >= 2)
    scale = 4000.0
    # the median background picks an element: a last-bit difference in the data can pick a neighbouring element, so the filled
    # pixels are compared like the rest, by value
    assert np.abs(got - data).max() <= 4e-6 * scale
    assert np.array_equal(md['surveyMask'].cpu().numpy(), sm)
    assert np.array_equal(md['pointSourceMask'].cpu().numpy(), pm)
    gfm = md['flagMask'].cpu().numpy()
    assert fm.sum() > 100
    assert (gfm != fm).sum() <= 4            # model > 1 uK decided within the FFT's rounding of the threshold


@pytest.mark.parametrize("dtype,wdtype", [(np.float64, np.float64), (np.float32, np.float32), (np.float64, np.float32)])
def test_fused_tile_load_equals_statements(dtype, wdtype):
    """MapDict.preprocess through nf_load_tile (one pass: clip, data[weights == 0] = 0, survey mask zeroed outside the tile's own
    area, flagMask * surveyMask) against the statement-by-statement path, bit for bit: with every mask given, with none, with
    a tile section and without, weights given or absent."""
    torch = engine.torch()
    rng = np.random.default_rng(3)
    Hs, Ws = 150, 260
    m = rng.standard_normal((Hs, Ws)).astype(dtype)
    w = rng.random((Hs, Ws)).astype(wdtype)
    w[rng.random((Hs, Ws)) < 0.1] = 0
    sm = (rng.random((Hs, Ws)) < 0.9).astype(np.uint8)
    ps = (rng.random((Hs, Ws)) < 0.95).astype(np.uint8)
    fl = (rng.random((Hs, Ws)) < 0.2).astype(np.uint8) * rng.integers(1, 200, (Hs, Ws)).astype(np.uint8)
    wcs = None
    tiles = {'T0': {'clippedSection': [17, 201, 9, 131], 'areaMaskInClipSection': [20, 150, 11, 100]}}
    for given in (('w', 'sm', 'ps', 'fl'), (), ('w',), ('sm', 'fl')):
        for tcd, tileName in ((tiles, 'T0'), (None, 'PRIMARY')):
            d = {'mapData': torch.as_tensor(m, device='cuda'), 'wcs': wcs, 'obsFreqGHz': 150.0, 'units': 'uK', 'label': 'x'}
            if 'w' in given:
                d['weightsData'] = torch.as_tensor(w, device='cuda')
            if 'sm' in given:
                d['surveyMask'] = torch.as_tensor(sm, device='cuda')
            if 'ps' in given:
                d['pointSourceMask'] = torch.as_tensor(ps, device='cuda')
            if 'fl' in given:
                d['flagMask'] = torch.as_tensor(fl, device='cuda')
            out = {}
            keep = pmaps._FUSED_LOAD
            try:
                for mode in (True, False):
                    pmaps._FUSED_LOAD = mode
                    md = pmaps.MapDict(dict(d), tileCoordsDict=tcd)
                    md.preprocess(tileName=tileName)
                    out[mode] = {k: md[k] for k in ('data', 'weights', 'surveyMask', 'pointSourceMask', 'flagMask')}
            finally:
                pmaps._FUSED_LOAD = keep
            for k in out[True]:
                a, b = out[True][k], out[False][k]
                assert tuple(a.shape) == tuple(b.shape), (given, tileName, k)
                assert a.dtype == b.dtype, (given, tileName, k, a.dtype, b.dtype)
                assert torch.equal(a, b), (given, tileName, k)
            # the sources are left as they were
            assert np.array_equal(d['mapData'].cpu().numpy(), m)
