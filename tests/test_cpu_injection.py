"""Host-side pieces of the injection / recovery loop (nemo/catalogs.py:849-1008), no GPU: the random test catalogue on the
reference's np.random stream against the oracle restatement, the one-pass form of the mask dilation, the batched draw, and
the cross-match helpers against brute force."""
import numpy as np
from scipy import ndimage

from nemo_amd import catalogs
from nemo_amd.wcs import CARWCS, calcAngSepDeg
from nemo_oracle import catalogs as ocat


def _mask(shape=(300, 420)):
    m = np.ones(shape, dtype=np.uint8)
    m[:, :25] = 0
    m[100:110, 200:230] = 0
    m[0:5, min(300, shape[1] - 10):] = 0
    m[shape[0] - 50, 50] = 0
    return m


def test_dilated_area_mask_is_twenty_binary_dilations():
    m = _mask()
    for n in (0, 1, 7, 20):
        d = m.copy()
        for i in range(n):
            d = 1 - ndimage.binary_dilation(1 - d)          # nemo/catalogs.py:899-900
        assert np.array_equal(d != 0, catalogs._dilatedAreaMask(m, n))
    assert catalogs._dilatedAreaMask(np.ones((40, 50), dtype=np.uint8), 20).all()      # no hole: nothing shrinks, borders included


def test_generate_test_catalog_on_the_reference_stream_equals_the_oracle():
    m = _mask()
    wcs = CARWCS.make(m.shape, RADeg=15.0, decDeg=-4.0)
    m2 = _mask((200, 260))
    m2[:, 130:] = 0
    wcs2 = CARWCS.make(m2.shape, RADeg=18.5, decDeg=-4.0)
    am = catalogs.AreaMasks(['A', 'EMPTY', 'B'], {'A': m, 'EMPTY': np.zeros((50, 60), dtype=np.uint8), 'B': m2},
                            {'A': wcs, 'EMPTY': wcs, 'B': wcs2})
    for dist, rng_ in (('linear', [0.3, 4.0]), ('log', [1, 1000])):
        np.random.seed(5)
        a = catalogs.generateTestCatalog(None, 7, amplitudeColumnName='y_c', amplitudeRange=rng_, amplitudeDistribution=dist, selFn=am,
                                         maskDilationPix=20)
        np.random.seed(5)
        b = ocat.generate_test_catalog(am.tileNames, am.areaMaskDict, am.WCSDict, 7, amplitudeColumnName='y_c', amplitudeRange=rng_,
                                       amplitudeDistribution=dist, maskDilationPix=20)
        assert len(a) == len(b) > 7
        for r, s in zip(a.to_records(), b):
            assert r['RADeg'] == s['RADeg'] and r['decDeg'] == s['decDeg'] and r['y_c'] == s['y_c']
            assert r['tileName'] == s['tileName'] and r['name'] == s['name']
        assert set(np.asarray(a['tileName']).tolist()) == {'A', 'B'}


def test_batched_draw_has_the_same_properties():
    m = _mask((480, 1200))
    wcs = CARWCS.make(m.shape, RADeg=40.0, decDeg=10.0)
    am = catalogs.AreaMasks(['T'], {'T': m}, {'T': wcs})
    np.random.seed(11)
    c = catalogs.generateTestCatalog(None, 30, amplitudeColumnName='deltaT_c', amplitudeRange=[1, 1000], amplitudeDistribution='log',
                                     selFn=am, maskDilationPix=20, referenceStream=False)
    assert 10 <= len(c) <= 30
    ra, dec = np.asarray(c['RADeg']), np.asarray(c['decDeg'])
    sep = calcAngSepDeg(ra[:, None], dec[:, None], ra[None, :], dec[None, :]) + 99 * np.eye(len(c))
    assert sep.min() * 60 > 20.0                                     # the avoidance radius
    x, y = wcs.wcs2pix(ra, dec)
    valid = catalogs._dilatedAreaMask(m, 20)
    assert valid[np.floor(y).astype(int), np.floor(x).astype(int)].all()    # pixel + offset in [0, 1): inside the dilated mask
    amp = np.asarray(c['deltaT_c'])
    assert amp.min() >= 1 and amp.max() <= 1000


def test_cross_match_helpers_against_brute_force():
    rng = np.random.default_rng(3)
    n = 60
    a = catalogs.Table(cols={'RADeg': rng.uniform(10, 12, n), 'decDeg': rng.uniform(-1, 1, n), 'y_c': rng.uniform(0, 1, n)})
    pick = rng.permutation(n)[:25]
    b = catalogs.Table(cols={'RADeg': np.asarray(a['RADeg'])[pick] + rng.normal(0, 0.01, 25),
                             'decDeg': np.asarray(a['decDeg'])[pick] + rng.normal(0, 0.01, 25), 'SNR': rng.uniform(4, 9, 25)})
    xa, xb, rDeg = catalogs.crossMatch(a, b, radiusArcmin=2.5)
    oa, ob, oDeg = ocat.cross_match(a.to_records(), b.to_records(), radiusArcmin=2.5)
    assert len(xa) == len(oa) >= 20
    assert np.array_equal(np.asarray(xa['y_c']), np.array([r['y_c'] for r in oa]))
    assert np.array_equal(np.asarray(xb['SNR']), np.array([r['SNR'] for r in ob]))
    assert np.abs(rDeg - oDeg).max() < 1e-12
    kept = catalogs.removeCrossMatched(a, b, radiusArcmin=2.5)
    okept = ocat.remove_cross_matched(a.to_records(), b.to_records(), radiusArcmin=2.5)
    assert np.array_equal(np.asarray(kept['y_c']), np.array([r['y_c'] for r in okept]))
    assert len(kept) + len(xa) == n
    assert len(catalogs.removeCrossMatched(a, catalogs.Table(names=('RADeg', 'decDeg')))) == n
