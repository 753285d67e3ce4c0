"""BASELINE config C1 end to end at its own size: 1031 x 1032 pixels (tests/configs/smallTestSurveyMaskHeader.txt:4-5 of the
reference: 0.5' CAR pixels, a prime height), f150, the in-tree ACT beam profile (tests/golden/data/profile_AR1_2009_pixwin_130224.txt
= examples/ACT-DR3-clusters/profiles_ACT/...), CMB + 100 uK/pixel white noise + 100 point sources with lognormal amplitudes,
BeamRealSpaceMatchedFilter (dataMap noise from the central 4 x 4 degrees, 7' kernel, 20' noise grid, bckSub), findObjects at 4 sigma -
against the oracle, with the tolerances of tests/test_gpu_filters.py::test_realspace_filter_end_to_end."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import helpers  # noqa: E402
from nemo_amd import filters, photometry, signals, simgen  # noqa: E402
from test_gpu_filters import _check_maps, _check_catalogs, _rel  # noqa: E402


def test_c1_realspace_point_source_search_at_1031x1032(tmp_path):
    shape = (1031, 1032)
    beam = signals.BeamProfile(beamFileName=helpers.ACT_BEAM)
    mapsL, wcs, truth = simgen.makeSimTile(shape, [149.6], [beam], [100.0], seed=1234, numSources=100, kind='sources',
                                           painter=helpers.cpu_paint, dtype=np.float64, varyWeights=True, RADeg=0.0, decDeg=0.0)
    for m in mapsL:
        m['surveyMask'] = np.ones(shape, dtype=np.uint8)
        m['pointSourceMask'] = np.ones(shape, dtype=np.uint8)
    cRA, cDec = wcs.getCentreWCSCoords()
    params = {'noiseParams': {'method': 'dataMap', 'matchedFilterClass': 'BeamMatchedFilter',
                              'RADecSection': [cRA - 2.0, cRA + 2.0, cDec - 2.0, cDec + 2.0], 'kernelMaxArcmin': 7.0,
                              'symmetrize': False, 'noiseGridArcmin': 20.0, 'numNoiseBins': 1},
              'bckSub': True, 'outputUnits': 'uK'}
    ref, info = helpers.oracle_realspace(mapsL, wcs, params, 'beam')
    f = {'label': 'c1', 'class': 'BeamRealSpaceMatchedFilter', 'params': dict(params)}
    got, fobj = filters.filterMaps(helpers.product_maps(mapsL), f, 'PRIMARY', diagnosticsDir=str(tmp_path), selFnDir=str(tmp_path),
                                   verbose=False, undoPixelWindow=False, returnFilter=True)
    assert np.asarray(fobj.kern2d).shape == info['kern2d'].shape == (1, 29, 29)
    assert _rel(fobj.kern2d, info['kern2d']) < 2e-5
    assert abs(fobj.bckSubScaleArcmin - info['bckSubScaleArcmin']) < 1e-9
    assert abs(float(fobj.signalNorm) / float(info['signalNorm']) - 1) < 2e-5
    _check_maps(got, ref, 5e-5, rmsTol=5e-5)
    # with the oracle's kernel: convolution, noise map, masks bit for bit
    fobj.kern2d = info['kern2d']
    fobj.signalNorm = info['signalNorm']
    from nemo_oracle import filters as ofilt
    raw = np.array([m['mapData'] for m in mapsL])
    assert np.array_equal(fobj.applyFilter(raw), ofilt.real_space_apply(raw, info['kern2d'], info['signalNorm'], bckSubSigmaYX=info['bckSigma']))
    # detection: the same objects at the same pixels
    thr = 4.0
    cat = photometry.findObjects(got, threshold=thr, minObjPix=1, verbose=False)
    refCat = helpers.oracle_find(ref['SNMap'], np.zeros(shape, dtype=int), wcs, threshold=thr, minObjPix=1)
    n = _check_catalogs(cat, refCat, thr, posTol=2e-4, snrTol=1e-4)
    assert n >= 60                                                  # most of the 100 injected sources, and nothing else differs
    # pixel-window undo over the whole 1031 x 1032 tile (prime height: Bluestein columns)
    got2 = filters.filterMaps(helpers.product_maps(mapsL), f, 'PRIMARY', diagnosticsDir=str(tmp_path), selFnDir=str(tmp_path),
                              verbose=False, undoPixelWindow=True)
    refData = ofilt.undo_pixel_window(ref['data'])
    assert _rel(got2['data'], refData) < 5e-5
    assert np.array_equal(got2['data'] == 0, refData == 0)
