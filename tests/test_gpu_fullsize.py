"""Parity at the headline size (C2: 4096 x 4096, float64, one Arnaud scale) through properties that do not need the oracle
to run at that size (SURVEY.md 8c "known-answer checks"): the filtered ideal template returns y0 at the tile centre, the
S/N map of the noise is unit-variance, the filter application is linear, zeros stay zeros, injected clusters are recovered
within their errors.  The same quantities at sizes the oracle finishes in seconds are checked against it in
test_gpu_filters.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import helpers  # noqa: E402
from nemo_amd import engine, filters, photometry, maps  # noqa: E402

SHAPE = (4096, 4096)


@pytest.fixture(scope="module")
def c2():
    mapsL, wcs, truth = helpers.make_tile(SHAPE, freqs=(97.8,), fwhm=(2.1,), noise=(30.0,), seed=17, numSources=5,
                                          kind='clusters', dtype=np.float64)
    params = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 40.0}, 'outputUnits': 'yc', 'edgeTrimArcmin': 0.0,
              'z': 0.4, 'M500MSun': 2e14}
    got, fobj = filters.filterMaps(helpers.product_maps(mapsL), {'label': 'c2', 'class': 'ArnaudModelMatchedFilter',
                                                                 'params': dict(params)},
                                   'PRIMARY', diagnosticsDir=None, selFnDir=None, verbose=False, undoPixelWindow=False,
                                   returnFilter=True)
    return mapsL, wcs, truth, got, fobj


def test_ideal_template_returns_y0(c2):
    """nemo/filters.py:632-659 defines signalNorm so that the windowed template of amplitude y0 comes back as y0 at the tile
    centre (nemo/signals.py:1124 asserts the same to 1e-6)."""
    mapsL, wcs, truth, got, fobj = c2
    eng = fobj.engine
    y0 = 2e-4
    tmpl = fobj.makeSignalTemplateMap(mapsL[0]['beamFileName'], amplitude=maps.convertToDeltaT(y0, 97.8))
    windowed = eng.apply_window(engine.to_dev(tmpl), 1.0)
    out = fobj.applyFilter(windowed.reshape(1, *SHAPE))
    cRA, cDec = wcs.getCentreWCSCoords()
    cx, cy = wcs.wcs2pix(cRA, cDec)
    peak = eng.spline_eval(engine.to_dev(out), [[cy, cx]])[0]
    # applyFilter apodises the map edges first (nemo/filters.py:845); the template has decayed there: agreement is set by the
    # float32 filter cube and transform, 2e-5
    assert abs(peak / y0 - 1.0) < 2e-5


def test_sn_map_is_unit_variance_and_masks_hold(c2):
    mapsL, wcs, truth, got, fobj = c2
    sn, rms, data = np.asarray(got['SNMap']), np.asarray(got['RMSMap']), np.asarray(got['data'])
    assert sn.shape == SHAPE and rms.shape == SHAPE
    inside = np.asarray(got['surveyMask']) > 0
    assert np.all(rms[inside] > 0)
    quiet = inside & (np.abs(sn) < 4)                 # outside the injected clusters
    assert abs(sn[quiet].mean()) < 0.02
    # the RMS of a cell is 3-sigma clipped (2-3 % low for Gaussian noise) and the noise level varies inside a cell: close to
    # one, a few per cent either way
    assert 0.95 < sn[quiet].std() < 1.06
    # S/N is data / RMS where the RMS map is positive (nemo/filters.py:747-750)
    ratio = data[inside] / rms[inside]
    assert np.abs(ratio - sn[inside]).max() < 1e-6 * np.abs(sn[inside]).max()


def test_filter_application_is_linear(c2):
    mapsL, wcs, truth, got, fobj = c2
    rng = np.random.default_rng(5)
    a = engine.to_dev(rng.standard_normal((1,) + SHAPE))
    b = engine.to_dev(rng.standard_normal((1,) + SHAPE) * 3.0)
    fa, fb, fab = fobj.applyFilter(a), fobj.applyFilter(b), fobj.applyFilter(a + b)
    scale = float(fab.abs().max())
    # float32 outputs of float64 transforms: one rounding each
    assert float((fa + fb - fab).abs().max()) < 4e-7 * scale


def test_injected_clusters_are_recovered(c2):
    mapsL, wcs, truth, got, fobj = c2
    cat = photometry.findObjects(got, threshold=5.0, minObjPix=1, verbose=False, flagRings=False)
    assert len(cat) >= len(truth)
    photometry.measureFluxes(cat, got, None, photFilteredMapDict=None)
    for src in truth:
        sep = np.hypot((np.asarray(cat['RADeg']) - src['RADeg']) * np.cos(np.radians(src['decDeg'])),
                       np.asarray(cat['decDeg']) - src['decDeg'])
        k = int(np.argmin(sep))
        assert sep[k] < 1.5 / 60.0
        # the clusters were painted with the filter's own profile (z = 0.4, M500 = 2e14) but without the pixel window the
        # normalisation assumes (a ~2 % effect): recovered amplitude within 5 sigma + 4 % of the injected y0
        y = cat['y_c'][k] * 1e-4
        assert cat['SNR'][k] > 8
        assert abs(y - src['amplitude']) < 5.0 * cat['err_y_c'][k] * 1e-4 + 0.04 * src['amplitude']


_PER_WINDOW_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from nemo_amd import engine
m = np.load(%(mpath)r); w = np.load(%(wpath)r)
eng = engine.TileEngine.get(m.shape)
rms = eng.noise_map(engine.to_dev(m), engine.to_dev(w), {'noiseGridArcmin': 40.0, 'numNoiseBins': 1}, 0.5 / 60.0)
np.save(%(opath)r, rms.cpu().numpy())
print("per-window ok")
'''


def test_noise_block_path_equals_per_window_path(c2, tmp_path):
    """The single-read block evaluation of makeNoiseMap (incl. the magnitude-class route under the bright
    sources) against the library's other, independent evaluation of nemo/filters.py:407-478 - one workgroup per widened cell
    window, the window clipped directly (NF_NOISE_BLOCKS=0, read once per process, hence the child process) - on a
    4096 x 4096 filtered map with heavy tails: float64 sums in another order, so equal as float32 up to rare 1-ulp flips."""
    import os, subprocess, sys
    mapsL, wcs, truth, got, fobj = c2
    rng = np.random.default_rng(23)
    m = np.asarray(got['data'], dtype=np.float64).copy()
    yy, xx = np.mgrid[0:SHAPE[0], 0:SHAPE[1]]
    for k in range(12):        # bright extended objects on top of the filtered noise: blocks with > 256 tail values
        cy, cx = rng.uniform(100, SHAPE[0] - 100), rng.uniform(100, SHAPE[1] - 100)
        sl = (slice(int(cy) - 90, int(cy) + 90), slice(int(cx) - 90, int(cx) + 90))
        m[sl] += 60.0 * m.std() * np.exp(-0.5 * ((yy[sl] - cy) ** 2 + (xx[sl] - cx) ** 2) / rng.uniform(8, 25) ** 2)
    w = (1.0 + 0.3 * np.sin(xx / 300.0) * np.cos(yy / 170.0))
    eng = fobj.engine
    params = {'noiseGridArcmin': 40.0, 'numNoiseBins': 1}
    blockRMS = eng.noise_map(engine.to_dev(m), engine.to_dev(w), params, 0.5 / 60.0).cpu().numpy()
    work, ncy, ncx, nb = eng._lastNoiseWork
    hdr = work.view(torch.int32)[ncy * ncx * 4 * 2:ncy * ncx * 4 * 2 + 8].cpu().numpy()
    assert hdr[4] > 0, "no heavy block: the class route is not exercised"
    mpath, wpath, opath = str(tmp_path / "m.npy"), str(tmp_path / "w.npy"), str(tmp_path / "rms.npy")
    np.save(mpath, m)
    np.save(wpath, w)
    script = tmp_path / "worker.py"
    script.write_text(_PER_WINDOW_WORKER % {'root': os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'mpath': mpath, 'wpath': wpath, 'opath': opath})
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, NF_NOISE_BLOCKS="0"), capture_output=True, text=True,
                       timeout=240)
    assert r.returncode == 0 and "per-window ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
    cellRMS = np.load(opath)
    assert np.array_equal(blockRMS > 0, cellRMS > 0)
    assert np.abs(blockRMS - cellRMS).max() <= 1.2e-7 * cellRMS.max()
    assert (blockRMS != cellRMS).mean() < 0.02
