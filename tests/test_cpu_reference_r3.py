"""The oracle (and the product's host code) against fixtures made by the reference's OWN code for the rows either side of the
filters (tests/golden/make_golden_r3.py): catalogs.generateTestCatalog on a seeded stream, the catalogue statements of
maps.MapDict.preprocess with maps.makeModelImage, signals.fitQ and the numpy part of signals.makeBattagliaModelProfile.
No GPU: bit-for-bit bars throughout (both sides run numpy / scipy on the same inputs)."""
import json
import os

import numpy as np
from scipy import interpolate

import helpers
from nemo_amd import catalogs, signals, maps as pmaps
from nemo_amd.wcs import CARWCS
from nemo_oracle import catalogs as ocat, maps as omaps, filters as ofilt, pixell_restated as enmap

G = helpers.GOLDEN


def _wcs(g, key):
    hdr = json.loads(str(g[key]))
    for k in ('NAXIS', 'NAXIS1', 'NAXIS2'):
        hdr[k] = int(hdr[k])
    return CARWCS(hdr)


def _table(g, prefix, cols):
    out = {}
    for c in cols:
        v = g[prefix + c]
        out[c] = np.array(json.loads(str(v))) if v.dtype.kind == 'U' and v.ndim == 0 else v
    return catalogs.Table(cols=out)


# ---- generateTestCatalog ------------------------------------------------------------------------------------------------
def test_generate_test_catalog_is_the_references_on_a_seeded_stream():
    g = np.load(os.path.join(G, 'injection.npz'))
    tileNames = json.loads(str(g['tileNames']))
    masks = {t: g['mask_' + t] for t in tileNames}
    wcss = {t: _wcs(g, 'header_' + t) for t in tileNames}
    for k in range(int(g['ncases'])):
        p = 'c%d_' % k
        c = json.loads(str(g[p + 'meta']))
        np.random.seed(c['seed'])
        tab = catalogs.generateTestCatalog(None, c['n'], amplitudeColumnName=c['col'], amplitudeRange=c['rng'], amplitudeDistribution=c['dist'],
                                           selFn=catalogs.AreaMasks(tileNames, masks, wcss), avoidanceRadiusArcmin=c['avoid'],
                                           maskDilationPix=c['dil'])
        assert np.random.uniform() == float(g[p + 'nextUniform'])          # the stream is left where the reference leaves it
        np.random.seed(c['seed'])
        rows = ocat.generate_test_catalog(tileNames, masks, wcss, c['n'], amplitudeColumnName=c['col'], amplitudeRange=c['rng'],
                                          amplitudeDistribution=c['dist'], avoidanceRadiusArcmin=c['avoid'], maskDilationPix=c['dil'])
        want = json.loads(str(g[p + 'tileName']))
        assert len(tab) == len(rows) == len(want) > 0
        assert np.array_equal(np.asarray(tab['RADeg']), g[p + 'RADeg']) and np.array_equal(np.asarray(tab['decDeg']), g[p + 'decDeg'])
        assert np.array_equal(np.asarray(tab[c['col']]), g[p + 'amp']) and np.array_equal(np.asarray(tab['name']), g[p + 'name'])
        assert np.asarray(tab['tileName']).tolist() == want
        assert np.array_equal(np.array([r['RADeg'] for r in rows]), g[p + 'RADeg'])
        assert np.array_equal(np.array([r[c['col']] for r in rows]), g[p + 'amp'])
        assert 'T_EMPTY' not in want


# ---- Battaglia profile -----------------------------------------------------------------------------------------------------
def test_battaglia_profile_is_the_references():
    g = np.load(os.path.join(G, 'battaglia.npz'))
    for k in range(int(g['ncases'])):
        p = 'c%d_' % k
        c = json.loads(str(g[p + 'meta']))
        gp = c['GNFWParams']
        d = signals.makeBattagliaModelProfile(c['z'], c['M500c'], GNFWParams=gp if gp == 'default' else dict(gp))
        assert np.array_equal(d['tckP'][0], g[p + 'tck_t']) and np.array_equal(d['tckP'][1], g[p + 'tck_c']) and d['tckP'][2] == int(g[p + 'tck_k'])
        assert d['theta500Arcmin'] == float(g[p + 'theta500Arcmin']) and np.array_equal(d['rDeg'], g[p + 'rDeg'])


# ---- preprocess / makeModelImage --------------------------------------------------------------------------------------------
def _objects_for(beam, obsFreqGHz, GNFWParams, profile, override, TCMBAlpha=0):
    def objectsFor(rr):
        tab = catalogs.Table(cols={k: [r[k] for r in rr] for k in rr[0].keys()})
        objs = pmaps._modelObjects(tab, beam, obsFreqGHz, GNFWParams, profile, override, TCMBAlpha, 5, signals.fiducialCosmoModel)
        return [o[:4] + (None,) for o in objs] if override is not None else objs
    return objectsFor


def preprocess_case(g, k):
    """Case k of preprocess.npz -> (map dict as helpers.make_tile gives it, wcs, the MapDict keys of the case, meta)."""
    p = 'c%d_' % k
    meta = json.loads(str(g[p + 'meta']))
    q = 'in_%s_' % meta['dtype']
    wcs = _wcs(g, q + 'header')
    m = {'mapData': g[q + 'mapData'], 'weightsData': g[q + 'weightsData'], 'obsFreqGHz': meta['freq'], 'units': 'uK',
         'beamFileName': signals.BeamProfile(FWHMArcmin=meta['fwhm']), 'surveyMask': g[q + 'surveyMaskIn'], 'pointSourceMask': g[q + 'psMaskIn'], 'wcs': wcs}
    tabs = {t: _table(g, p + 'tab_' + t + '_', cols) for t, cols in json.loads(str(g[p + 'tabs'])).items()}
    name = meta['name']
    if name == 'maskAndFill':
        keys = {'maskAndFillFromCatalog': tabs['holes'], 'maskHoleDilationFactor': meta['maskHoleDilationFactor']}
    elif name == 'maskAndFillRadius':
        keys = {'maskAndFillFromCatalog': [tabs['holes']], 'maskHoleRadiusArcmin': meta['maskHoleRadiusArcmin']}
    elif name == 'maskPointSources':
        keys = {'maskPointSourcesFromCatalog': [{'path': tabs['psA'], 'fluxCutJy': 0.01}, tabs['psB']]}
    elif name == 'subtractModel':
        keys = {'subtractModelFromCatalog': [tabs['srcs']]}
    elif name == 'injectOverride':
        keys = {'injectSources': {'catalog': tabs['inj'], 'GNFWParams': 'default', 'profile': 'A10', 'override': {'redshift': 0.4, 'M500': 2e14}},
                'subtractModelFromCatalog': [tabs['srcs']], 'maskAndFillFromCatalog': tabs['holes']}
    elif name == 'injectSourcesOnly':
        keys = {'injectSources': {'catalog': tabs['srcs'], 'profile': None, 'override': None}}
    else:
        raise Exception(name)
    return m, wcs, keys, meta


def oracle_preprocess(m, wcs, keys, meta):
    """oracle catalogue_steps on a fixture case (inputs as preprocess has them just before nemo/maps.py:325)."""
    data = m['mapData'].copy()
    weights = m['weightsData']
    data[weights == 0] = 0
    md = {'obsFreqGHz': m['obsFreqGHz']}
    for k, v in keys.items():
        if isinstance(v, catalogs.Table):
            v = [v.to_records()]
        elif isinstance(v, list):
            v = [(dict(e, path=e['path'].to_records()) if isinstance(e, dict) else e.to_records()) for e in v]
        elif isinstance(v, dict) and 'catalog' in v:
            v = dict(v, catalog=v['catalog'].to_records())
        md[k] = v
    beam = m['beamFileName']

    def modelImage(rows, cluster=False, obsFreqGHz=None, GNFWParams='default', profile='A10', override=None, validAreaSection=None):
        whole = None
        if override is not None and obsFreqGHz is not None:
            whole = (signals.fSZ(obsFreqGHz, z=override['redshift']), signals.TCMB * 1e6)
        return omaps.make_model_image(data.shape, wcs, rows, _objects_for(beam, obsFreqGHz, GNFWParams, profile, override), signals.paintGeometry,
                                      validAreaSection=validAreaSection, wholeMapFactors=whole)

    return omaps.catalogue_steps(md, data, weights, wcs, m['surveyMask'].copy(), m['pointSourceMask'].copy(), np.zeros(data.shape, dtype=np.uint8),
                                 modelImage, injectValidArea=meta['validAreaSection'])


def test_oracle_preprocess_is_the_references():
    g = np.load(os.path.join(G, 'preprocess.npz'))
    for k in range(int(g['ncases'])):
        m, wcs, keys, meta = preprocess_case(g, k)
        data, sm, pm, fm = oracle_preprocess(m, wcs, keys, meta)
        p = 'c%d_' % k
        assert data.dtype == g[p + 'out_data'].dtype, meta
        assert np.array_equal(sm, g[p + 'out_surveyMask']) and np.array_equal(pm, g[p + 'out_pointSourceMask']), meta
        assert np.array_equal(fm, g[p + 'out_flagMask']), meta
        assert np.array_equal(data, g[p + 'out_data']), meta


def model_case(g, j):
    p = 'm%d_' % j
    meta = json.loads(str(g[p + 'meta']))
    tab = _table(g, p + 'tab_', meta['cols'])
    return tab, meta['kw'], meta


def test_oracle_make_model_image_is_the_references():
    g = np.load(os.path.join(G, 'preprocess.npz'))
    wcs = _wcs(g, 'm_header')
    beam = signals.BeamProfile(FWHMArcmin=1.4)
    for j in range(int(g['m_ncases'])):
        tab, kw, meta = model_case(g, j)
        shape = tuple(meta['shape'])
        override, obsFreq = kw.get('override'), kw.get('obsFreqGHz')
        whole = (signals.fSZ(obsFreq, z=override['redshift']), signals.TCMB * 1e6) if (override is not None and obsFreq is not None) else None
        okw = {k: kw[k] for k in ('minSNR', 'validAreaSection') if k in kw}
        for pixwin in (False, True):
            got = omaps.make_model_image(shape, wcs, tab.to_records(), _objects_for(beam, obsFreq, 'default', 'A10', override), signals.paintGeometry,
                                         wholeMapFactors=whole, applyPixelWindow=pixwin, **okw)
            want = g['m%d_' % j + ('model_pixwin' if pixwin else 'model')]
            assert got.dtype == want.dtype and np.abs(want).max() > 0, meta
            assert np.array_equal(got, want), meta


# ---- fitQ ------------------------------------------------------------------------------------------------------------------
def fitq_case(g, k):
    p = 'c%d_' % k
    meta = json.loads(str(g[p + 'meta']))
    wcs = _wcs(g, p + 'header')
    mapsL = []
    for i, (freq, fwhm) in enumerate(zip(meta['freqs'], meta['fwhm'])):
        mapsL.append({'mapData': g[p + 'data'][i], 'weightsData': g[p + 'weights'][i], 'obsFreqGHz': freq, 'units': 'uK',
                      'beamFileName': signals.BeamProfile(FWHMArcmin=fwhm), 'label': 'f%03d' % (int(round(freq))), 'wcs': wcs,
                      'surveyMask': np.ones(tuple(meta['shape']), dtype=np.uint8), 'pointSourceMask': np.ones(tuple(meta['shape']), dtype=np.uint8)})
    return mapsL, wcs, meta


def test_oracle_fitq_loop_is_the_references():
    """signals.fitQ's per-scale loop (nemo/signals.py:1079-1126) restated with the oracle's pieces, the reference's filter cube
    and normalisation handed in."""
    g = np.load(os.path.join(G, 'fit_q.npz'))
    for k in range(int(g['ncases'])):
        mapsL, wcs, meta = fitq_case(g, k)
        p = 'c%d_' % k
        shape = tuple(meta['shape'])
        zRange, MRange = signals._QScaleGrid(meta['params'], meta['ZDEPQ'], signals.fiducialCosmoModel)
        if meta['ZDEPQ'] == 1:
            zRange, MRange = zRange[:40], MRange[:40]          # (265 scales on the CPU: the first 40 pin the loop)
        RADeg, decDeg = wcs.getCentreWCSCoords()
        x, y = wcs.wcs2pix(RADeg, decDeg)
        geom = signals.paintGeometry(shape, wcs, RADeg, decDeg)
        mf = ofilt.MatchedFilter([{'data': m['mapData'], 'units': 'uK', 'obsFreqGHz': m['obsFreqGHz']} for m in mapsL], meta['params'], {}, None)
        mf.filt, mf.signalNorm = g[p + 'filt'], np.float64(g[p + 'signalNorm'])        # (numpy float64: the product is then float64, nemo/filters.py:853)
        Q, theta, zs = [], [], []
        for z, M in zip(zRange, MRange):
            cube = []
            for m in mapsL:
                table = signals.clusterModelTable(meta['kind'], m['beamFileName'], z, M, 'default', signals.convertToDeltaT(2e-4, m['obsFreqGHz']), 15.0)
                cube.append(enmap.fft(enmap.apply_window(helpers.cpu_paint(geom, table), pow=1.0)))
            filtered = mf.apply(np.array(cube))
            spl = interpolate.RectBivariateSpline(np.arange(shape[0]), np.arange(shape[1]), filtered, kx=3, ky=3)
            peak = spl(y, x)[0][0]
            if peak not in Q:
                Q.append(peak)
                theta.append(signals.calcTheta500Arcmin(z, M, signals.fiducialCosmoModel))
                zs.append(z)
        Q = np.array(Q) / 2e-4
        want = dict(zip(zip(g[p + 'theta500Arcmin'].tolist(), g[p + 'z'].tolist()), g[p + 'Q'].tolist()))
        assert abs(Q[0] - 1) < 1e-6
        for q, th, z in zip(Q, theta, zs):
            assert (th, z) in want, (k, th, z)
            assert q == want[(th, z)], (k, th, z, q, want[(th, z)])
