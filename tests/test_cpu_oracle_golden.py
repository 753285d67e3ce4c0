"""CPU suite, part 1: the oracle (oracle/nemo_oracle) is pinned against golden vectors produced by the reference
itself (tests/golden/make_golden.py, run in the build container through a stub importer), and the pure-Python host
logic of the product (names, catalogue merging, GNFW profile) is checked against the same vectors / brute force.
Bit-exact where the arithmetic is the same library call (scipy.ndimage, numpy); tolerance stated otherwise."""
import json
import os

import numpy as np
import pytest

from nemo_oracle import filters as ofilt, photometry as ophot

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_oracle_object_positions_match_reference():
    g = np.load(os.path.join(GOLDEN, "object_positions.npz"))
    for k in range(int(g['ncases'])):
        ids, pos, npx, seg = ophot.get_object_positions(g['map%d' % k], float(g['thr%d' % k]),
                                                        findCenterOfMass=bool(g['com%d' % k]))
        assert np.array_equal(ids, g['ids%d' % k])
        assert np.array_equal(seg, g['seg%d' % k])
        assert np.array_equal(npx, g['npx%d' % k])
        assert np.array_equal(np.asarray(pos, dtype=np.float64)[1:], g['pos%d' % k][1:])     # [0] = background: NaN / unused
    assert ophot.get_object_positions(np.zeros((8, 8)), 1.0) is None
    with pytest.raises(Exception):
        ophot.get_object_positions(np.zeros((8, 8)), -1.0)


def test_oracle_noise_map_matches_reference():
    g = np.load(os.path.join(GOLDEN, "noise_map.npz"))
    for k in range(int(g['ncases'])):
        params = json.loads(str(g['params%d' % k]))
        rms = ofilt.make_noise_map(g['map%d' % k], list(g['weights%d' % k]), params, float(g['pixDeg']))
        assert rms.dtype == np.float32
        assert np.array_equal(rms, g['rms%d' % k]), "case %d (%s)" % (k, params)


def test_oracle_realspace_apply_matches_reference():
    g = np.load(os.path.join(GOLDEN, "realspace_apply.npz"))
    for k in range(int(g['ncases'])):
        out = ofilt.real_space_apply(g['cube%d' % k], g['kern%d' % k], float(g['signalNorm']))
        ref = g['out%d' % k]
        assert out.dtype == ref.dtype
        assert np.array_equal(out, ref)


def test_names_match_reference():
    from nemo_amd import catalogs
    with open(os.path.join(GOLDEN, "names.json")) as f:
        rows = json.load(f)
    assert len(rows) > 400
    for ra, dec, name, longName in rows:
        assert catalogs.makeName(ra, dec) == name
        assert catalogs.makeLongName(ra, dec, prefix='SO-CL') == longName
    # lazily named tables give the same strings
    t = catalogs.Table()
    t.setLazyNames('ACT-CL')
    t['RADeg'] = np.array([r[0] for r in rows])
    t['decDeg'] = np.array([r[1] for r in rows])
    assert list(t['name']) == [r[2] for r in rows]


def test_vector_names_equal_string_route():
    """catalogs.makeNames computes the digits of a name as numbers and sends every object that lies near a place where a digit
    changes through the reference's string route: the names of 100 000 random positions and of positions placed on and beside
    those places (whole hours, tenths of a minute of RA, whole arcminutes and degrees of declination, +-1e-14 .. 1e-5 away)
    equal makeName's character for character."""
    from nemo_amd import catalogs
    rng = np.random.default_rng(1)
    RA, dec = rng.uniform(0, 360, 100000), rng.uniform(-90, 90, 100000)
    advRA, advDec = [], []
    for h in range(24):
        for m10 in (0, 1, 5, 99, 100, 333, 599):
            for e in (0, 1e-13, -1e-13, 1e-9, -1e-9, 3e-8, -3e-8, 1e-6, -1e-6):
                advRA.append(min(max((h + m10 / 600.0) * 15.0 + e, 0.0), 359.9999999))
    for d in range(-89, 90):
        for m in (0, 1, 30, 59):
            for e in (0, 1e-14, -1e-14, 1e-10, -1e-10, 1e-7, -1e-7, 1e-5, -1e-5):
                advDec.append(d + (1 if d >= 0 else -1) * m / 60.0 + e)
    advDec += [0.0, 1e-3, -1e-3, 2e-4, -2e-4, 0.5, -0.5, 9.999999, -9.999999, 10.0, -10.0, 89.99999]
    advDec = [v for v in advDec if v == 0.0 or abs(v) >= 1e-4]         # (below 1e-4 str() has no decimal point: both routes raise)
    k = max(len(advRA), len(advDec))
    RA = np.concatenate([RA, np.resize(np.array(advRA), k)])
    dec = np.concatenate([dec, np.resize(np.array(advDec), k)])
    got = catalogs.makeNames(RA, dec)
    assert got == [catalogs.makeName(r, d) for r, d in zip(RA.tolist(), dec.tolist())]
    with pytest.raises(ValueError):
        catalogs.makeNames(np.full(20, 10.0), np.full(20, 1e-5))
    with pytest.raises(ValueError):
        catalogs.makeName(10.0, 1e-5)


def test_gnfw_profile_matches_reference():
    """Line-of-sight GNFW integral (nemo/gnfw.py:142-195): the product follows the reference's algorithm (quad table at
    100 log-spaced impact parameters, cubic interp1d in log-log, power-law ends) and must reproduce the reference's own
    output to 1e-12 (measured: bit for bit)."""
    from nemo_amd import gnfw
    g = np.load(os.path.join(GOLDEN, "gnfw.npz"))
    x = g['x']
    for key, pars in [('los_arnaud', dict(c=1.177, alpha=1.0510, beta=5.4905, gamma=0.3081)),
                      ('los_ppp', dict(c=1.81, alpha=1.33, beta=4.13, gamma=0.31))]:
        ours = gnfw.tsz_profile_los(x, **pars)
        assert np.abs(ours / g[key] - 1).max() < 1e-12
    assert np.array_equal(gnfw.gnfw(x, 1.177, 1.0510, 5.4905, 0.3081), g['gnfw'])
    # outside the table: constant below x1, power law above x2
    far = gnfw.tsz_profile_los(np.array([1e-9, 1e-8, 1e4, 2e4]))
    assert far[0] == far[1] and far[3] < far[2]


def _brute_forced(tabs):
    from nemo_amd import catalogs as C
    from nemo_amd.wcs import calcAngSepDeg
    allc = C.vstack([t.copy() for t in tabs])
    rem = np.zeros(len(allc))
    for r in range(len(allc)):
        rDeg = calcAngSepDeg(allc['RADeg'][r], allc['decDeg'][r], allc['RADeg'], allc['decDeg'])
        x = np.where(rDeg < C.XMATCH_RADIUS_DEG)[0]
        if len(x) > 1:
            keep = x[np.argmax(allc['SNR'][x])]
            for ix in x:
                if ix != keep:
                    rem[ix] = 1
    m = allc[rem == 0]
    m.sort(['RADeg', 'decDeg'])
    return m


def _brute_first(cd, phot):
    from nemo_amd import catalogs as C
    from nemo_amd.wcs import calcAngSepDeg
    ref, mer = [], []
    for k in cd:
        (ref if k.split('#')[0] == phot else mer).append(cd[k]['catalog'].copy())
    ref = C.vstack(ref) if ref else None
    mer = C.vstack(mer)
    if ref is None:
        ref = mer
    for r in range(len(ref)):
        rDeg = calcAngSepDeg(ref['RADeg'][r], ref['decDeg'][r], mer['RADeg'], mer['decDeg'])
        x = np.where(rDeg < C.XMATCH_RADIUS_DEG)[0]
        if len(x) > 1:
            bi = x[np.argmax(mer['SNR'][x])]
            if mer['SNR'][bi] > ref['SNR'][r]:
                for c in ['SNR', 'template']:
                    ref[c][r] = mer[c][bi]
    ref.sort(['RADeg', 'decDeg'])
    return ref


def _brute_remove_duplicates(tab):
    """nemo/catalogs.py:704-749 with the nearest-other-neighbour search done by brute force."""
    from nemo_amd import catalogs as C
    from nemo_amd.wcs import calcAngSepDeg
    n = len(tab)
    nearest = np.zeros(n)
    for i in range(n):
        r = calcAngSepDeg(tab['RADeg'][i], tab['decDeg'][i], tab['RADeg'], tab['decDeg'])
        r[i] = np.inf
        nearest[i] = r.min()
    mask = nearest < C.XMATCH_RADIUS_DEG
    dup, nodup = tab[mask], tab[~mask]
    keep = np.zeros(len(dup), dtype=bool)
    for i in range(len(dup)):
        r = calcAngSepDeg(dup['RADeg'][i], dup['decDeg'][i], dup['RADeg'], dup['decDeg'])
        m = np.less_equal(r, C.XMATCH_RADIUS_DEG)
        idx = np.where(m)[0]
        keep[idx[np.equal(dup['SNR'][m], dup['SNR'][m].max())][0]] = True
    out = C.vstack([dup[keep], nodup])
    out.sort('RADeg')
    return out, len(dup)


def test_catalogue_merging_equals_the_reference_loops():
    """makeOptimalCatalog / removeDuplicates: the k-d tree formulation gives what the reference's row-by-row loops give
    (nemo/catalogs.py:258-362, 704-749), including ties (identical positions) and the in-place case photFilter=None."""
    from nemo_amd import catalogs as C
    rng = np.random.default_rng(0)

    def mk(n, label):
        t = C.Table()
        t.setLazyNames('ACT-CL')
        t['RADeg'] = rng.uniform(10, 12, n)
        t['decDeg'] = rng.uniform(-1, 1, n)
        t['SNR'] = np.round(rng.uniform(4, 20, n), 1)        # rounded: equal-SNR ties occur
        t['template'] = np.full(n, label, dtype='U30')
        return t
    a, b, c = mk(300, 'a'), mk(400, 'b'), mk(350, 'c')
    b['RADeg'][:20] = a['RADeg'][:20]                       # exact positional coincidences across scales
    b['decDeg'][:20] = a['decDeg'][:20]
    cd = {'a#T': {'catalog': a}, 'b#T': {'catalog': b}, 'c#T': {'catalog': c}}

    def fresh():
        return {k: {'catalog': v['catalog'].copy()} for k, v in cd.items()}
    got = C.makeOptimalCatalog(fresh(), method='ref-forced')
    ref = _brute_forced([a, b, c])
    assert len(got) == len(ref) and np.array_equal(got['RADeg'], ref['RADeg']) and np.array_equal(got['SNR'], ref['SNR'])
    for phot in ['a', None]:
        got = C.makeOptimalCatalog(fresh(), photFilter=phot, method='ref-first')
        ref = _brute_first(fresh(), phot)
        assert np.array_equal(got['SNR'], ref['SNR']) and np.array_equal(got['template'], ref['template'])
    allTab = C.vstack([a.copy(), b.copy(), c.copy()])
    got, nDup, names = C.removeDuplicates(allTab.copy())
    ref, nDupRef = _brute_remove_duplicates(allTab.copy())
    assert nDup == nDupRef and len(names) == nDup
    assert np.array_equal(got['RADeg'], ref['RADeg']) and np.array_equal(got['SNR'], ref['SNR'])
    # cuts
    cut = C.makeOptimalCatalog(fresh(), constraintsList=['SNR > 10'], method='ref-forced')
    assert len(cut) > 0 and cut['SNR'].min() > 10
    empty = C.makeOptimalCatalog({'a#T': {'catalog': []}}, method='ref-first')
    assert len(empty) == 0 and 'SNR' in empty.keys()
    with pytest.raises(Exception):
        C.makeOptimalCatalog(fresh(), method='nope')
