#!/usr/bin/env python3
"""Golden vectors for the rows either side of the filters, produced by running the
reference's OWN code in the build container (nothing at test time imports /root/reference):

  catalogs.generateTestCatalog                     nemo/catalogs.py:849-947     -> injection.npz  (seeded np.random stream)
  maps.MapDict.preprocess (catalogue statements)   nemo/maps.py:325-465         -> preprocess.npz
  maps.makeModelImage, makeDegreesDistanceMap      nemo/maps.py:1658-1853, 2479-2536
  catalogs.getCatalogWithinImage                   nemo/catalogs.py:1039-1076
  signals.makeArnaudModelSignalMap / makeBeamModelSignalMap / makeArnaudModelProfile   nemo/signals.py:450-506, 595-627, 704-774
  signals.fitQ                                     nemo/signals.py:897-1137     -> fit_q.npz
  signals.makeBattagliaModelProfile                nemo/signals.py:509-591      -> battaglia.npz

Stand-ins, as in make_golden_filters.py, for what the reference delegates to ABSENT third-party code (this surface stays
unpinned): pixell.enmap (oracle/nemo_oracle/pixell_restated.py), astLib WCS / astCoords (nemo_amd/wcs.py, wrapped so that
array calls return [[a, b], ...] as astWCS does), astropy Table (a plain column table), pyccl (the flat-LCDM functions of
nemo_amd/signals.py), and - replacing signals._paintSignalMap, whose core is pixell.pointsrcs.sim_objects - the build's own
painter (radial table + oracle painter inside the object's RA/dec box).  Every statement of the functions listed above runs
unmodified.

Usage:  python tests/golden/make_golden_r3.py [injection|preprocess|fitq|battaglia]
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_filters as MGF                                  # noqa: E402  (paths, stub importer, reference filter objects)
from make_golden_filters import MG, psignals, CARWCS, calcAngSepDeg, pixell_restated, opainter, header_json    # noqa: E402
from nemo_oracle import maps as omaps                              # noqa: E402


# ---- stand-ins ----------------------------------------------------------------------------------------------------
class AstWCS(CARWCS):
    """CARWCS with astWCS's return convention: scalar calls give [a, b], array / list calls give [[a, b], ...]."""

    def pix2wcs(self, x, y):
        a, b = CARWCS.pix2wcs(self, x, y)
        if np.ndim(a) == 0:
            return [a, b]
        return np.stack([a, b], axis=1).tolist()

    def wcs2pix(self, RADeg, decDeg):
        a, b = CARWCS.wcs2pix(self, RADeg, decDeg)
        if np.ndim(a) == 0:
            return [a, b]
        return np.stack([a, b], axis=1).tolist()

    def copy(self):
        return AstWCS(dict(self.header))


class Col(np.ndarray):
    """A table column: an array whose .data is the plain array (astropy's Column.data)."""

    @property
    def data(self):
        return np.asarray(self)


class RefTable(object):
    """Stands where astropy.table.Table stands in maps.py / catalogs.py / signals.py."""
    registry = {}
    made = []

    def __init__(self, cols=None):
        self.cols = {} if cols is None else {k: np.asarray(v) for k, v in cols.items()}
        self.meta = {}
        RefTable.made.append(self)

    def read(self, path):
        return RefTable.registry[path]

    def keys(self):
        return list(self.cols.keys())

    def __len__(self):
        return 0 if len(self.cols) == 0 else len(next(iter(self.cols.values())))

    def __getitem__(self, k):
        if isinstance(k, str):
            return self.cols[k].view(Col)
        sel = np.asarray(k)
        return RefTable({c: v[sel] for c, v in self.cols.items()})

    def __iter__(self):
        for i in range(len(self)):
            yield {c: v[i] for c, v in self.cols.items()}

    def add_column(self, col):
        data, name = col
        self.cols[name] = np.asarray(data)

    def sort(self, key):
        order = np.argsort(self.cols[key], kind='stable')
        self.cols = {c: v[order] for c, v in self.cols.items()}


def paint_standin(shape, wcs, tckP, beam=None, RADeg=None, decDeg=None, amplitude=None, maxSizeDeg=10.0, convolveWithBeam=True,
                  vmin=1e-12, omap=None, obsFrequencyGHz=None, TCMBAlpha=0, z=None):
    """signals._paintSignalMap (nemo/signals.py:630-701) with pointsrcs.sim_objects served by the build's painter."""
    from nemo import maps as rmaps
    if RADeg is None and decDeg is None:
        RADeg, decDeg = wcs.getCentreWCSCoords()
    base = psignals.makeRadialTable(tckP, beam=beam if convolveWithBeam else None, amplitude=None if amplitude is None else 1.0,
                                    maxSizeDeg=float(maxSizeDeg), convolveWithBeam=convolveWithBeam)
    RAs, decs = np.atleast_1d(RADeg), np.atleast_1d(decDeg)
    amps = [None] * len(RAs) if amplitude is None else np.atleast_1d(amplitude).tolist()
    if len(amps) == 1 and len(RAs) > 1:
        amps = amps * len(RAs)
    target = omap if omap is not None else np.zeros(shape, dtype=np.float32)
    for ra, dec, amp in zip(RAs.tolist(), decs.tolist(), amps):
        table = base if amp is None else psignals.scaledTable(base, amp)
        if omap is not None or len(RAs) > 1:
            x0, x1, y0, y1 = omaps.object_window(target.shape, wcs, ra, dec, maxSizeDeg)
        else:
            x0, x1, y0, y1 = 0, target.shape[1], 0, target.shape[0]
        if x1 <= x0 or y1 <= y0:
            continue
        A, B, C = psignals.paintGeometry(target.shape, wcs, ra, dec)
        clip = opainter.paint_radial((y1 - y0, x1 - x0), A[y0:y1], B[y0:y1], C[x0:x1], table.r, table.prof, table.amp, 1.0)
        if obsFrequencyGHz is not None:
            clip = rmaps.convertToDeltaT(clip, obsFrequencyGHz=obsFrequencyGHz, TCMBAlpha=TCMBAlpha, z=z)
        target[y0:y1, x0:x1] = target[y0:y1, x0:x1] + clip
    if base.sign < 0:
        target = target * -1
    return target


def install():
    filters, maps, signals = MGF.install_reference()
    from nemo import catalogs, completeness
    ast = types.SimpleNamespace(calcAngSepDeg=calcAngSepDeg)
    catalogs.astCoords = ast
    atpy = types.SimpleNamespace(Table=RefTable, Column=lambda data, name: (data, name))
    catalogs.atpy = atpy
    maps.atpy = atpy
    signals.atpy = atpy
    maps.enmap = types.SimpleNamespace(zeros=np.zeros, apply_window=pixell_restated.apply_window, fft=pixell_restated.fft)
    signals.enmap = pixell_restated
    cosmo = psignals.fiducialCosmoModel
    signals.fiducialCosmoModel = cosmo

    class CCLError(Exception):
        pass
    signals.ccl = types.SimpleNamespace(h_over_h0=lambda c, a: c.h_over_h0(a), angular_diameter_distance=lambda c, a: c.angular_diameter_distance(a),
                                        physical_constants=types.SimpleNamespace(RHO_CRITICAL=psignals.RHO_CRITICAL), CCLError=CCLError)
    signals.calcTheta500Arcmin = psignals.calcTheta500Arcmin          # (pyccl inside; the flat-LCDM stand-in)
    signals.M500cToMdef = lambda M500c, z, massDef, cosmoModel: psignals.M500cToM200c(M500c, z, cosmoModel)
    signals.M200cDef = None
    signals.BeamProfile = lambda beamFileName=None, **kw: psignals._asBeam(beamFileName)
    signals._paintSignalMap = paint_standin
    return filters, maps, signals, catalogs


# ---- generateTestCatalog -------------------------------------------------------------------------------------------
def area_mask(shape, seed):
    rng = np.random.default_rng(seed)
    m = np.ones(shape, dtype=np.uint8)
    m[:, :22] = 0
    m[:9, :] = 0
    for _ in range(3):
        y, x = rng.integers(30, shape[0] - 30), rng.integers(30, shape[1] - 30)
        m[y - 6:y + 7, x - 9:x + 10] = 0
    return m


def run_injection(catalogs):
    out = {}
    shapes = {'T_A': (260, 340), 'T_EMPTY': (70, 80), 'T_B': (180, 420)}
    masks = {k: area_mask(s, 70 + i) for i, (k, s) in enumerate(shapes.items())}
    masks['T_EMPTY'][:] = 0
    wcss = {k: AstWCS(dict(CARWCS.make(s, RADeg=20.0 + 3 * i, decDeg=-5.0 + i).header)) for i, (k, s) in enumerate(shapes.items())}
    selFn = types.SimpleNamespace(tileNames=list(shapes.keys()), areaMaskDict=masks, WCSDict=wcss)
    cases = [dict(seed=11, n=9, col='y_c', rng=[0.3, 4.0], dist='linear', dil=20, avoid=20.0),
             dict(seed=12, n=25, col='deltaT_c', rng=[1, 1000], dist='log', dil=5, avoid=8.0),
             dict(seed=13, n=4, col='fixed_y_c', rng=[0.001, 1], dist='linear', dil=0, avoid=30.0)]
    for k, name in enumerate(shapes):
        out['mask_' + name], out['header_' + name] = masks[name], np.array(header_json(wcss[name]))
    out['tileNames'] = np.array(json.dumps(list(shapes.keys())))
    out['ncases'] = np.int64(len(cases))
    for k, c in enumerate(cases):
        np.random.seed(c['seed'])
        tab = catalogs.generateTestCatalog(None, c['n'], amplitudeColumnName=c['col'], amplitudeRange=c['rng'],
                                           amplitudeDistribution=c['dist'], selFn=selFn, avoidanceRadiusArcmin=c['avoid'],
                                           maskDilationPix=c['dil'])
        after = np.random.uniform()                    # the stream position the reference leaves behind
        p = 'c%d_' % k
        out[p + 'meta'] = np.array(json.dumps(c))
        out[p + 'name'], out[p + 'RADeg'], out[p + 'decDeg'] = np.asarray(tab['name']), np.asarray(tab['RADeg'], dtype=np.float64), np.asarray(tab['decDeg'], dtype=np.float64)
        out[p + 'amp'] = np.asarray(tab[c['col']], dtype=np.float64)
        out[p + 'tileName'] = np.array(json.dumps([str(t) for t in tab['tileName']]))
        out[p + 'nextUniform'] = np.float64(after)
        print("generateTestCatalog case %d: %d rows" % (k, len(tab)))
    np.savez_compressed(os.path.join(HERE, 'injection.npz'), **out)


# ---- MapDict.preprocess / makeModelImage ------------------------------------------------------------------------------
class TileMapDict(object):
    pass


def make_mapdict(maps, m, wcs, surveyMask, psMask, extra):
    """A reference MapDict over in-memory arrays: only loadTile is replaced."""
    md = maps.MapDict.__new__(maps.MapDict)
    dict.__init__(md, {'mapFileName': 'mem', 'weightsFileName': 'mem', 'obsFreqGHz': m['obsFreqGHz'], 'units': m['units'],
                       'beamFileName': m['beamFileName'], 'surveyMask': 'mem', 'pointSourceMask': 'mem'})
    md.update(extra)
    shape = m['mapData'].shape
    md.tileCoordsDict = {'PRIMARY': {'areaMaskInClipSection': [8, shape[1] - 5, 6, shape[0] - 9]}}
    arrays = {'mapFileName': m['mapData'], 'weightsFileName': m['weightsData'], 'surveyMask': surveyMask, 'pointSourceMask': psMask}

    def loadTile(mapKey, tileName, returnWCS=False):
        data = np.array(arrays[mapKey])
        return (data, wcs) if returnWCS else data
    md.loadTile = loadTile
    return md


def positions(wcs, shape, n, seed, margin=0):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-margin, shape[1] + margin, n)
    y = rng.uniform(-margin, shape[0] + margin, n)
    ra, dec = CARWCS.pix2wcs(wcs, x, y)
    return np.asarray(ra), np.asarray(dec)


def cluster_cols(wcs, shape, n, seed):
    ra, dec = positions(wcs, shape, n, seed, margin=15)
    rng = np.random.default_rng(seed + 1)
    templates = np.array(['Arnaud_M2e14_z0p4', 'Arnaud_M4e14_z0p2', 'Arnaud_M1e14_z0p8#extra'])[rng.integers(0, 3, n)]
    return {'RADeg': ra, 'decDeg': dec, 'y_c': rng.uniform(0.3, 3.0, n), 'template': templates, 'SNR': rng.uniform(3, 20, n),
            'tileName': np.array(['PRIMARY'] * n)}


def run_preprocess(maps):
    shape = (140, 180)
    out = {'ncases': np.int64(0)}
    k = 0
    for dtype in (np.float32, np.float64):
        mapsL, wcs0, surveyMask, psMask = MGF.make_inputs(shape, (150.0,), (1.4,), (25.0,), 300 + k, 'beam', dtype, True, numSources=4)
        wcs = AstWCS(dict(wcs0.header))
        m = mapsL[0]
        ra, dec = positions(wcs, shape, 30, 310 + k, margin=12)
        rng = np.random.default_rng(320 + k)
        holes = {'RADeg': ra[:14], 'decDeg': dec[:14], 'ellipse_A': rng.uniform(2, 14, 14), 'ellipse_B': rng.uniform(1, 2, 14)}
        srcs = {'RADeg': ra[14:], 'decDeg': dec[14:], 'deltaT_c': rng.uniform(50, 4000, 16)}
        psA = {'RADeg': ra[:20], 'decDeg': dec[:20], 'rArcmin': rng.choice([1.0, 2.5, 4.0, 6.0], 20), 'fluxJy': rng.uniform(0, 0.05, 20)}
        psB = {'RADeg': ra[20:], 'decDeg': dec[20:], 'ellipse_A': rng.uniform(4, 30, 10)}
        inj = cluster_cols(wcs, shape, 8, 330 + k)
        RefTable.registry = {'psA': RefTable(psA), 'psB': RefTable(psB)}
        cases = [('maskAndFill', {'maskAndFillFromCatalog': RefTable(holes), 'maskHoleDilationFactor': 1.5}, {'holes': holes}),
                 ('maskAndFillRadius', {'maskAndFillFromCatalog': [RefTable({'RADeg': ra[:14], 'decDeg': dec[:14]})], 'maskHoleRadiusArcmin': 3.0},
                  {'holes': {'RADeg': ra[:14], 'decDeg': dec[:14]}}),
                 ('maskPointSources', {'maskPointSourcesFromCatalog': [{'path': 'psA', 'fluxCutJy': 0.01}, 'psB']}, {'psA': psA, 'psB': psB}),
                 ('subtractModel', {'subtractModelFromCatalog': [RefTable(srcs)]}, {'srcs': srcs}),
                 ('injectOverride', {'injectSources': {'catalog': RefTable(inj), 'GNFWParams': 'default', 'profile': 'A10',
                                                      'override': {'redshift': 0.4, 'M500': 2e14}},
                                     'subtractModelFromCatalog': [RefTable(srcs)], 'maskAndFillFromCatalog': RefTable(holes)},
                  {'inj': inj, 'srcs': srcs, 'holes': holes}),
                 ('injectSourcesOnly', {'injectSources': {'catalog': RefTable(srcs), 'profile': None, 'override': None}}, {'srcs': srcs})]
        for name, extra, tabs in cases:
            md = make_mapdict(maps, m, wcs, surveyMask, psMask, extra)
            md.preprocess(tileName='PRIMARY')
            p = 'c%d_' % k
            out[p + 'meta'] = np.array(json.dumps({'name': name, 'dtype': np.dtype(dtype).name, 'fwhm': 1.4, 'freq': 150.0,
                                                   'maskHoleDilationFactor': extra.get('maskHoleDilationFactor'),
                                                   'maskHoleRadiusArcmin': extra.get('maskHoleRadiusArcmin'),
                                                   'validAreaSection': md.tileCoordsDict['PRIMARY']['areaMaskInClipSection']}))
            q = 'in_%s_' % (np.dtype(dtype).name)              # (the six cases of a dtype share their inputs)
            out[q + 'header'] = np.array(header_json(wcs))
            out[q + 'mapData'], out[q + 'weightsData'], out[q + 'surveyMaskIn'], out[q + 'psMaskIn'] = m['mapData'], m['weightsData'], surveyMask, psMask
            for tname, cols in tabs.items():
                for c, v in cols.items():
                    out[p + 'tab_' + tname + '_' + c] = np.asarray(v) if np.asarray(v).dtype.kind != 'U' else np.array(json.dumps([str(s) for s in v]))
            out[p + 'tabs'] = np.array(json.dumps({t: list(c.keys()) for t, c in tabs.items()}))
            for key in ('data', 'surveyMask', 'pointSourceMask', 'flagMask'):
                a = np.asarray(md[key])
                assert a.dtype.kind in 'fiub', (key, a.dtype)
                out[p + 'out_' + key] = a
            print("preprocess case %d %-18s %s: data %s, holes %d px, flagged %d px" % (
                k, name, np.dtype(dtype).name, md['data'].dtype, int((np.asarray(md['surveyMask']) == 0).sum() - (surveyMask == 0).sum()),
                int((np.asarray(md['flagMask']) > 0).sum())))
            k += 1
    out['ncases'] = np.int64(k)
    # makeModelImage on its own: every branch, with and without the pixel window
    shape = (180, 240)
    mapsL, wcs0, surveyMask, psMask = MGF.make_inputs(shape, (150.0,), (1.4,), (25.0,), 400, 'beam', np.float32, False, numSources=2)
    wcs = AstWCS(dict(wcs0.header))
    beam = mapsL[0]['beamFileName']
    ra, dec = positions(wcs, shape, 40, 410, margin=18)
    srcs = {'RADeg': ra, 'decDeg': dec, 'deltaT_c': np.random.default_rng(411).uniform(-300, 3000, 40)}
    cl = cluster_cols(wcs, shape, 12, 420)
    mock = dict(cl, true_M500c=np.random.default_rng(421).uniform(1, 6, 12), redshift=np.random.default_rng(422).uniform(0.15, 1.1, 12),
                true_y_c=np.random.default_rng(423).uniform(0.3, 2.0, 12))
    mcases = [('sources', srcs, {}), ('clusters_y', cl, {'minSNR': 5.0, 'validAreaSection': [10, 230, 5, 175]}),
              ('clusters_uK', cl, {'obsFreqGHz': 148.0, 'minSNR': 5.0}), ('override', cl, {'obsFreqGHz': 148.0, 'override': {'redshift': 0.3, 'M500': 3e14}}),
              ('override_y', cl, {'override': {'redshift': 0.6, 'M500': 1.5e14}}), ('mock', mock, {'obsFreqGHz': 98.0, 'TCMBAlpha': 0.0})]
    out['m_header'] = np.array(header_json(wcs))
    out['m_ncases'] = np.int64(len(mcases))
    for j, (name, cols, kw) in enumerate(mcases):
        p = 'm%d_' % j
        out[p + 'meta'] = np.array(json.dumps({'name': name, 'kw': kw, 'cols': list(cols.keys()), 'shape': list(shape)}))
        for c, v in cols.items():
            out[p + 'tab_' + c] = np.asarray(v) if np.asarray(v).dtype.kind != 'U' else np.array(json.dumps([str(s) for s in v]))
        for pixwin in (False, True):
            model = maps.makeModelImage(shape, wcs, RefTable(cols), beam, applyPixelWindow=pixwin, **kw)
            assert isinstance(model, np.ndarray) and model.dtype.kind == 'f'
            out[p + ('model_pixwin' if pixwin else 'model')] = np.asarray(model)
        print("makeModelImage case %d %-12s dtype %s peak %.4g" % (j, name, model.dtype, np.abs(model).max()))
    assert maps.makeModelImage(shape, wcs, RefTable({'RADeg': np.array([0.0]), 'decDeg': np.array([80.0]), 'deltaT_c': np.array([1.0])}), beam) is None
    np.savez_compressed(os.path.join(HERE, 'preprocess.npz'), **out)


# ---- fitQ ---------------------------------------------------------------------------------------------------------
def run_fitq(filters, signals):
    out = {}
    cases = [dict(shape=(150, 190), freqs=(150.0, 98.0), fwhm=(1.4, 2.1), noise=(12.0, 16.0), seed=500, dtype='float64', cls='ArnaudModelMatchedFilter', kind='arnaud'),
             dict(shape=(128, 160), freqs=(98.0,), fwhm=(2.1,), noise=(20.0,), seed=501, dtype='float32', cls='BattagliaModelMatchedFilter', kind='battaglia')]
    out['ncases'] = np.int64(len(cases))
    for k, c in enumerate(cases):
        params = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20.0}, 'outputUnits': 'yc', 'edgeTrimArcmin': 0.0,
                  'M500MSun': 2e14, 'z': 0.4, 'GNFWParams': 'default'}
        mapsL, wcs0, surveyMask, psMask = MGF.make_inputs(c['shape'], c['freqs'], c['fwhm'], c['noise'], c['seed'], c['kind'], np.dtype(c['dtype']), False)
        wcs = AstWCS(dict(wcs0.header))
        with tempfile.TemporaryDirectory() as tmpDir:
            obj = MGF.reference_filter(filters, c['cls'], c['kind'], mapsL, wcs, surveyMask, psMask, dict(params), tmpDir)
            with np.errstate(divide='ignore', invalid='ignore'):
                res = obj.buildAndApply()
        RefTable.made = []
        config = types.SimpleNamespace(parDict={'photFilter': 'golden', 'mapFilters': [{'label': 'golden', 'class': c['cls'], 'params': dict(params)}],
                                                'unfilteredMaps': [{'obsFreqGHz': m['obsFreqGHz'], 'beamFileName': m['beamFileName']} for m in mapsL],
                                                'GNFWParams': 'default'},
                                       tileNames=['PRIMARY'], allTileNames=['PRIMARY'], cachedFilters={'PRIMARY': obj}, MPIEnabled=False,
                                       rank=1, selFnDir=None, _timeStarted=0.0)
        signals.fitQ(config)
        QTab = [t for t in RefTable.made if 'Q' in t.cols][-1]
        p = 'c%d_' % k
        out[p + 'meta'] = np.array(json.dumps(dict(c, params=params, ZDEPQ=int(QTab.meta['ZDEPQ']))))
        out[p + 'header'] = np.array(header_json(wcs))
        out[p + 'data'] = np.array([m['mapData'] for m in mapsL])
        out[p + 'weights'] = np.array([m['weightsData'] for m in mapsL])
        out[p + 'filt'], out[p + 'signalNorm'] = np.asarray(obj.filt), np.float64(obj.signalNorm)
        # what the reference's own buildAndApply returned for this filter class (the Battaglia case is the only fixture of that class)
        out[p + 'res_data'], out[p + 'res_SNMap'], out[p + 'res_RMSMap'] = np.asarray(res['data']), np.asarray(res['SNMap']), np.asarray(res['RMSMap'])
        out[p + 'fRelWeights'] = np.array([float(obj.fRelWeights[m['obsFreqGHz']]) for m in mapsL])
        for col in ('Q', 'theta500Arcmin', 'z'):
            out[p + col] = np.asarray(QTab.cols[col], dtype=np.float64)
        print("fitQ case %d %s: %d scales, Q in [%.4f, %.4f]" % (k, c['cls'], len(QTab), out[p + 'Q'].min(), out[p + 'Q'].max()))
    np.savez_compressed(os.path.join(HERE, 'fit_q.npz'), **out)


# ---- Battaglia profile -------------------------------------------------------------------------------------------------
def run_battaglia(signals):
    out = {}
    cases = [(0.4, 2e14, 'default'), (0.1, 8e14, 'default'), (1.2, 5e13, 'default'),
             (0.6, 3e14, {'P0': 6.0, 'gamma': 0.3, 'alpha': 1.0, 'beta': 5.0, 'c500': 1.2, 'tol': 1e-7, 'npts': 100})]
    out['ncases'] = np.int64(len(cases))
    for k, (z, M, gp) in enumerate(cases):
        given = gp if gp == 'default' else dict(gp)
        d = signals.makeBattagliaModelProfile(z, M, GNFWParams=given)
        p = 'c%d_' % k
        out[p + 'meta'] = np.array(json.dumps({'z': z, 'M500c': M, 'GNFWParams': gp, 'M200c': float(psignals.M500cToM200c(M, z, psignals.fiducialCosmoModel))}))
        out[p + 'tck_t'], out[p + 'tck_c'], out[p + 'tck_k'] = np.asarray(d['tckP'][0]), np.asarray(d['tckP'][1]), np.int64(d['tckP'][2])
        out[p + 'theta500Arcmin'], out[p + 'rDeg'] = np.float64(d['theta500Arcmin']), np.asarray(d['rDeg'])
        print("Battaglia profile case %d: theta500 %.4f arcmin" % (k, d['theta500Arcmin']))
    np.savez_compressed(os.path.join(HERE, 'battaglia.npz'), **out)


def main():
    what = sys.argv[1:] or ['injection', 'preprocess', 'fitq', 'battaglia']
    filters, maps, signals, catalogs = install()
    if 'injection' in what:
        run_injection(catalogs)
    if 'battaglia' in what:
        run_battaglia(signals)
    if 'preprocess' in what:
        run_preprocess(maps)
    if 'fitq' in what:
        run_fitq(filters, signals)


if __name__ == '__main__':
    main()
