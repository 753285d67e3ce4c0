"""CPU suite: the oracle's matched-filter and real-space-filter restatements are pinned against fixtures produced by the
reference's OWN class code - filters.filterMaps, MatchedFilter.buildAndApply / applyFilter, RealSpaceMatchedFilter.buildKernel /
buildAndApply / applyFilter (nemo/filters.py:51-106, 521-855, 888-1210) run unmodified under a stub importer with
pixell.enmap served by oracle/nemo_oracle/pixell_restated.py (tests/golden/make_golden_filters.py).  The four pixell calls
(fft, ifft, apod, apply_window), the WCS and the template painter are the only surface that stays unpinned."""
import os

import numpy as np
import pytest

import helpers
from nemo_amd import signals
from nemo_oracle import filters as ofilt

GOLDEN = helpers.GOLDEN


def _ulp32(a, b):
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def _oracle_from_fixture(g, k, faithfulLoop):
    mapsL, wcs, meta = helpers.load_reference_case(g, k)
    p = 'c%d_' % k
    params = meta['params']
    unit, scaled = g[p + 'tmplUnit'], g[p + 'tmplScaled']

    def templateFn(i, amp):
        return (unit if amp is None else scaled)[i]
    om = []
    for m in mapsL:
        data = m['mapData'].copy()
        data[m['weightsData'] == 0] = 0
        om.append({'data': data, 'weights': m['weightsData'], 'obsFreqGHz': m['obsFreqGHz'], 'units': m['units'],
                   'surveyMask': m['surveyMask'], 'pointSourceMask': m['pointSourceMask']})
    RADeg, decDeg = wcs.getCentreWCSCoords()
    cx, cy = wcs.wcs2pix(RADeg, decDeg)
    from nemo_amd import maps as pmaps
    xps, yps = pmaps.pixelScalesDeg(wcs)
    mf = ofilt.MatchedFilter(om, params, {'pixSizeDeg': wcs.getPixelSizeDeg(), 'centreYX': (cy, cx), 'xPixScaleDeg': xps,
                                          'yPixScaleDeg': yps}, templateFn, fSZ=signals.fSZ,
                             deltaT0=lambda f: signals.convertToDeltaT(2e-4, f))
    res = mf.buildAndApply(faithfulLoop=faithfulLoop)
    return res, mf, mapsL, meta


@pytest.mark.parametrize("faithfulLoop", [True, False])
def test_oracle_matched_filter_equals_reference_class_code(faithfulLoop):
    g = np.load(os.path.join(GOLDEN, "matched_filter.npz"))
    n = int(g['ncases'])
    assert n == 12
    for k in range(n):
        res, mf, mapsL, meta = _oracle_from_fixture(g, k, faithfulLoop)
        p = 'c%d_' % k
        what = "case %d (%s)" % (k, meta)
        if faithfulLoop:
            # the same numpy / scipy calls in the same order: bit for bit
            assert np.array_equal(mf.filt, g[p + 'filt']), what
            assert mf.signalNorm == g[p + 'signalNorm'], what
            assert isinstance(mf.signalNorm, np.float64) == bool(g[p + 'signalNormIsF64']), what
            if meta['params']['outputUnits'] == 'yc':
                assert np.array_equal(np.array([mf.fRelWeights[m['obsFreqGHz']] for m in mapsL]), g[p + 'fRelWeights']), what
            for key in ('data', 'SNMap', 'RMSMap', 'surveyMask'):
                assert res[key].dtype == g[p + 'out_' + key].dtype, (what, key)
                assert np.array_equal(res[key], g[p + 'out_' + key]), (what, key)
            assert np.array_equal(ofilt.undo_pixel_window(res['data']), g[p + 'out_dataUndone']), what
        else:
            # batched np.linalg.inv instead of one call per mode: the float32 filter may differ in the last place
            assert _ulp32(mf.filt, g[p + 'filt']).max() <= 1, what
            assert abs(mf.signalNorm / g[p + 'signalNorm'] - 1) < 1e-6, what
            scale = np.abs(g[p + 'out_SNMap']).max()
            assert np.abs(res['SNMap'] - g[p + 'out_SNMap']).max() < 2e-6 * scale, what
            assert np.array_equal(res['surveyMask'], g[p + 'out_surveyMask']), what


def test_oracle_realspace_filter_equals_reference_class_code():
    g = np.load(os.path.join(GOLDEN, "realspace_build.npz"))
    n = int(g['ncases'])
    assert n == 5
    for k in range(n):
        mapsL, wcs, meta = helpers.load_reference_case(g, k)
        p = 'c%d_' % k
        params = meta['params']
        res, info = helpers.oracle_realspace(mapsL, wcs, params, meta['kind'])
        what = "case %d (%s)" % (k, meta)
        assert info['kern2d'].shape == g[p + 'kern2d'].shape, what
        # the inner filter is built by the batched inverse (1 ulp of float32, above); its real-space image follows
        kscale = np.abs(g[p + 'kern2d']).max()
        assert np.abs(info['kern2d'] - g[p + 'kern2d']).max() < 1e-6 * kscale, what
        assert abs(info['bckSubScaleArcmin'] - float(g[p + 'bckSubScaleArcmin'])) < 1e-9, what
        assert abs(info['signalNorm'] / float(g[p + 'signalNorm']) - 1) < 2e-6, what
        assert np.allclose([info['fRelWeights'][m['obsFreqGHz']] for m in mapsL], g[p + 'fRelWeights'], rtol=2e-6, atol=1e-7), what
        scale = np.abs(g[p + 'out_SNMap']).max()
        assert np.array_equal(res['surveyMask'], g[p + 'out_surveyMask']), what
        assert np.abs(res['SNMap'] - g[p + 'out_SNMap']).max() < 5e-6 * scale, what
        assert np.abs(res['data'] - g[p + 'out_data']).max() < 5e-6 * np.abs(g[p + 'out_data']).max(), what
        for key in ('data', 'SNMap', 'RMSMap', 'surveyMask'):
            assert res[key].dtype == g[p + 'out_' + key].dtype, (what, key)


def test_oracle_realspace_kernel_from_the_reference_inner_filter_is_bit_exact():
    """Kernel crop (nemo/filters.py:943-974) on the reference's own inner filter cube: same numpy calls, bit for bit."""
    g = np.load(os.path.join(GOLDEN, "realspace_build.npz"))
    for k in range(int(g['ncases'])):
        mapsL, wcs, meta = helpers.load_reference_case(g, k)
        p = 'c%d_' % k
        y0, y1, x0, x1 = meta['sectionSlice']
        sl, subWCS = helpers._section_slice(wcs, mapsL[0]['mapData'].shape, meta['params']['noiseParams']['RADecSection'])
        assert (sl[0].start, sl[0].stop, sl[1].start, sl[1].stop) == (y0, y1, x0, x1)
        subShape = (y1 - y0, x1 - x0)
        degX, degY, radiansMap = helpers.radians_map(subWCS, subShape)
        kern2d, prof, arcminRange = ofilt.crop_kernel(g[p + 'innerFilt'], radiansMap, degX,
                                                      meta['params']['noiseParams']['kernelMaxArcmin'],
                                                      symmetrize=meta['params']['noiseParams'].get('symmetrize', False))
        assert np.array_equal(kern2d, g[p + 'kern2d']), k


def test_oracle_measure_fluxes_equals_reference_code():
    """photometry.getSNRValues / measureFluxes restated in the oracle vs the reference's own functions
    (nemo/photometry.py:251-377): same scipy calls -> bit for bit, including rows on the first pixel row / column (measured
    without a bounds test by measureFluxes, zeroed by getSNRValues) and a zero-S/N row (inf / nan error columns)."""
    import json
    from nemo_amd import photometry as pphot
    from nemo_amd.wcs import CARWCS
    from nemo_oracle import photometry as ophot
    g = np.load(os.path.join(GOLDEN, "measure_fluxes.npz"))
    hdr = json.loads(str(g['header']))
    for key in ('NAXIS', 'NAXIS1', 'NAXIS2'):
        hdr[key] = int(hdr[key])
    wcs = CARWCS(hdr)
    for k in range(int(g['ncases'])):
        p = 'c%d_' % k
        meta = json.loads(str(g[p + 'meta']))
        cat = [{'RADeg': float(r), 'decDeg': float(d), 'SNR': float(s)} for r, d, s in zip(g['RADeg'], g['decDeg'], g[p + 'SNR'])]
        ophot.measure_fluxes(cat, g[p + 'data'], wcs.wcs2pix, meta['units'], signals.convertToDeltaT, pphot.deltaTToJyPerSr,
                             photSNMap=g[p + 'photSNMap'] if meta['usePhot'] else None,
                             photData=g[p + 'photData'] if meta['usePhot'] else None, useInterpolator=meta['useInterpolator'],
                             beamSolidAngle_nsr=meta['BEAMNSR'], obsFreqGHz=meta['FREQGHZ'])
        for c in json.loads(str(g[p + 'cols'])):
            assert np.array_equal(np.array([o[c] for o in cat]), g[p + 'col_' + c], equal_nan=True), (k, c)
        cat2 = [{'RADeg': float(r), 'decDeg': float(d)} for r, d in zip(g['RADeg'], g['decDeg'])]
        ophot.get_snr_values(cat2, g[p + 'SNMap'], wcs.wcs2pix, useInterpolator=meta['useInterpolator'], invertMap=True, prefix='inv_')
        assert np.array_equal(np.array([o['inv_SNR'] for o in cat2]), g[p + 'col_inv_SNR']), k


def test_noise_method_max_datamap_cmb_raises_as_the_reference_does():
    """The reference's 'max(dataMap,CMB)' branch (nemo/filters.py:574-577) is unreachable: the loop at :539-561 raises for
    it.  The fixture records the reference's own exception; the oracle raises the same."""
    g = np.load(os.path.join(GOLDEN, "matched_filter.npz"))
    message = str(g['maxDataMapCMB_raises'])
    assert message == "'max(dataMap,CMB)' is not a valid filter noise method name - fix the .yml config file"
    mapsL, wcs, meta = helpers.load_reference_case(g, 0)
    params = dict(meta['params'], noiseParams=dict(meta['params']['noiseParams'], method='max(dataMap,CMB)'))
    with pytest.raises(Exception) as e:
        helpers.oracle_matched(mapsL, wcs, params, meta['kind'])
    assert str(e.value) == message


def _find_objects_cases():
    import json
    from nemo_amd.wcs import CARWCS
    g = np.load(os.path.join(GOLDEN, 'find_objects.npz'))
    hdr = json.loads(str(g['header']))
    for key in ('NAXIS', 'NAXIS1', 'NAXIS2'):
        hdr[key] = int(hdr[key])
    wcs = CARWCS(hdr)
    for k in range(int(g['ncases'])):
        p = 'c%d_' % k
        meta = json.loads(str(g[p + 'meta']))
        cols = {c: g[p + 'col_' + c] for c in json.loads(str(g[p + 'cols']))}
        yield k, meta, wcs, g[p + 'SNMap'], g[p + 'flagMask'], cols


def test_oracle_find_objects_equals_reference_code():
    """photometry.findObjects (nemo/photometry.py:25-210) run as it is (tests/golden/make_golden_filters.py, measureShapes on,
    ring flags, inverted map) against the oracle's restatement: every numeric column bit for bit."""
    from nemo_amd.wcs import calcAngSepDeg
    from nemo_oracle import photometry as ophot
    seen = 0
    for k, meta, wcs, sn, flagMask, cols in _find_objects_cases():
        kw = {kk: v for kk, v in meta.items() if kk != 'dtype'}
        rows = ophot.find_objects(sn, flagMask, wcs.pix2wcs, angSepDeg=calcAngSepDeg, measureShapes=True, **kw)
        assert len(rows) == len(cols['SNR']) and len(rows) > 0
        for c in ('id', 'x', 'y', 'RADeg', 'decDeg', 'SNR', 'numSigPix', 'flags', 'ringFlag', 'ellipse_PA', 'ellipse_A', 'ellipse_B',
                  'ellipse_x0', 'ellipse_y0', 'ellipse_e'):
            got = np.array([r[c] for r in rows])
            assert np.array_equal(got.astype(cols[c].dtype), cols[c], equal_nan=cols[c].dtype.kind == 'f'), (k, c)
        seen += int((cols['ellipse_A'] > 0).sum())
    assert seen > 40


def _rms_tab_cases():
    import json
    from nemo_amd.wcs import CARWCS
    g = np.load(os.path.join(GOLDEN, 'rms_tab.npz'))
    hdr = json.loads(str(g['header']))
    for key in ('NAXIS', 'NAXIS1', 'NAXIS2'):
        hdr[key] = int(hdr[key])
    wcs = CARWCS(hdr)
    for k in range(int(g['ncases'])):
        p = 'c%d_' % k
        maxFlags = int(g[p + 'maxFlags'])
        yield wcs, g[p + 'RMSMap'], g[p + 'areaMask'], g[p + 'flagMask'], (None if maxFlags < 0 else maxFlags), g[p + 'areaDeg2'], g[p + 'y0RMS']


def test_oracle_rms_table_equals_reference_code():
    """completeness.getRMSTab / maps.getPixelAreaArcmin2Map run as they are (tests/golden/make_golden_filters.py) against the
    oracle's restatement: noise levels and areas bit for bit."""
    from nemo_oracle import completeness as ocomp
    for wcs, rms, area, flag, maxFlags, areaDeg2, y0RMS in _rms_tab_cases():
        gotArea, gotRMS = ocomp.get_rms_tab(rms, area, wcs, flagMask=flag, maxFlags=maxFlags)
        assert np.array_equal(gotRMS, y0RMS) and gotRMS.dtype == y0RMS.dtype
        assert np.array_equal(gotArea, areaDeg2)
        assert len(y0RMS) > 40
