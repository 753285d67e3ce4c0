#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF (run in the build container only;
/root/reference does not exist on the GPU box and nothing at test time imports it).

The reference cannot be imported as is (pixell, astLib, pyccl, astropy, mahotas, mpi4py ... are absent offline), so the
missing third-party modules are served as inert stubs by a meta-path finder and only reference
functions whose arithmetic is pure numpy / scipy / Python are called:

  photometry.getObjectPositions      nemo/photometry.py:213-248
  filters.MapFilter.makeNoiseMap     nemo/filters.py:342-480   (grid mode, clipped-RMS and percentile estimators; smart mode)
  filters.RealSpaceMatchedFilter.applyFilter (bckSub False)    nemo/filters.py:1164-1210
  gnfw.tsz_profile_los / gnfw.gnfw   nemo/gnfw.py:121-195
  catalogs.makeName / makeLongName   nemo/catalogs.py:436-595

Every output is asserted to be a plain ndarray / float / str before it is saved (a stub leaking through would not be).
Usage:  python tests/golden/make_golden.py        (writes tests/golden/*.npz, names.json)
"""
import importlib.abc
import importlib.machinery
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MISSING = ('astLib', 'pixell', 'pyccl', 'mahotas', 'colorcet', 'astropy', 'reproject', 'mpi4py')


class _Stub(types.ModuleType):
    __path__ = []
    __all__ = []

    def __getattr__(self, n):
        if n.startswith('__'):
            raise AttributeError(n)
        if n == 'value':
            return 1.0
        m = _Stub(self.__name__ + '.' + n)
        setattr(self, n, m)
        return m

    def __call__(self, *a, **k):
        return _Stub('call')

    def __mul__(self, o):
        return 1.0
    __rmul__ = __truediv__ = __rtruediv__ = __add__ = __radd__ = __sub__ = __rsub__ = __pow__ = __mul__


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split('.')[0] in MISSING:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        pass


def _plain(x):
    assert isinstance(x, (np.ndarray, float, int, np.floating, np.integer, str, list, tuple)), type(x)
    return x


class _FakeWCS(object):
    """Only getPixelSizeDeg() is read by makeNoiseMap (nemo/filters.py:414)."""
    def __init__(self, pixDeg):
        self.pixDeg = pixDeg

    def getPixelSizeDeg(self):
        return self.pixDeg


def correlated_map(shape, seed, dtype=np.float64, sigmaPix=1.5):
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    m = ndimage.gaussian_filter(rng.standard_normal(shape), sigmaPix)
    return (m / m.std()).astype(dtype)


def main():
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, '/root/reference')
    import nemo  # noqa: F401
    from nemo import photometry, filters, gnfw, catalogs

    # ---- getObjectPositions -----------------------------------------------------------------------------------
    out = {}
    cases = [((96, 120), 11, 2.0, True), ((96, 120), 11, 2.0, False), ((211, 256), 12, 0.5, True), ((64, 64), 13, 3.0, True)]
    for k, (shape, seed, thr, com) in enumerate(cases):
        m = correlated_map(shape, seed)
        m[: 5, :] = 0.0
        ids, pos, npx, seg = photometry.getObjectPositions(m, thr, findCenterOfMass=com)
        out['map%d' % k] = m
        out['thr%d' % k] = np.float64(thr)
        out['com%d' % k] = np.int64(com)
        out['ids%d' % k] = _plain(np.asarray(ids))
        out['pos%d' % k] = _plain(np.asarray(pos, dtype=np.float64))
        out['npx%d' % k] = _plain(np.asarray(npx, dtype=np.float64))
        out['seg%d' % k] = _plain(np.asarray(seg)).astype(np.int32)
    out['ncases'] = np.int64(len(cases))
    assert photometry.getObjectPositions(np.zeros((8, 8)), 1.0) is None
    np.savez_compressed(os.path.join(HERE, 'object_positions.npz'), **out)

    # ---- makeNoiseMap --------------------------------------------------------------------------------------------
    out = {}
    pixDeg = 0.5 / 60.0
    ncase = 0
    for shape, seed, dtype, nFreq, noiseParams in [
            ((130, 170), 21, np.float64, 1, {'noiseGridArcmin': 20.0}),
            ((130, 170), 22, np.float64, 2, {'noiseGridArcmin': 20.0, 'numNoiseBins': 4}),
            ((211, 256), 23, np.float32, 3, {'noiseGridArcmin': 40.0, 'numNoiseBins': 2}),
            ((120, 150), 24, np.float64, 1, {'noiseGridArcmin': 20.0, 'numNoiseBins': 3, 'RMSEstimator': 'percentile'}),
            ((100, 90), 25, np.float64, 1, {'noiseGridArcmin': None, 'numNoiseBins': 3}),
            ((130, 170), 26, np.float64, 2, {'noiseGridArcmin': 'smart', 'numNoiseBins': 5})]:
        rng = np.random.default_rng(seed)
        yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
        m = (correlated_map(shape, seed) * (1.0 + xx / shape[1])).astype(dtype)
        m[:, :7] = 0
        m[40:48, 60:70] = 0
        weights = []
        for f in range(nFreq):
            w = (1.0 + 3.0 * (xx / shape[1]) * (yy / shape[0]) + 0.3 * f + 0.05 * rng.standard_normal(shape))
            w[:, shape[1] // 2:] *= 1.5
            weights.append(w.astype(dtype))
        obj = object.__new__(filters.MapFilter)
        obj.params = {'noiseParams': dict(noiseParams)}
        obj.wcs = _FakeWCS(pixDeg)
        obj.unfilteredMapsDictList = [{'weights': w} for w in weights]
        rms = obj.makeNoiseMap(m)
        k = ncase
        out['map%d' % k] = m
        out['weights%d' % k] = np.array(weights)
        out['params%d' % k] = np.array(json.dumps(noiseParams))
        out['rms%d' % k] = _plain(np.asarray(rms))
        assert out['rms%d' % k].dtype == np.float32
        ncase += 1
    out['ncases'] = np.int64(ncase)
    out['pixDeg'] = np.float64(pixDeg)
    np.savez_compressed(os.path.join(HERE, 'noise_map.npz'), **out)

    # ---- RealSpaceMatchedFilter.applyFilter, bckSub False -----------------------------------------------------------
    out = {}
    rng = np.random.default_rng(31)
    for k, (shape, ksz, nFreq, dtype) in enumerate([((90, 110), 9, 1, np.float32), ((80, 75), 7, 2, np.float32),
                                                    ((60, 70), 5, 1, np.float64)]):
        cube = rng.standard_normal((nFreq,) + shape).astype(dtype)
        kern = rng.standard_normal((nFreq, ksz, ksz))
        obj = object.__new__(filters.RealSpaceMatchedFilter)
        obj.params = {'bckSub': False}
        obj.kern2d = kern
        obj.signalNorm = 1.7
        obj.unfilteredMapsDictList = [{'obsFreqGHz': 90.0 + 60 * i} for i in range(nFreq)]
        res = obj.applyFilter(cube.copy())
        out['cube%d' % k] = cube
        out['kern%d' % k] = kern
        out['out%d' % k] = _plain(np.asarray(res))
    out['ncases'] = np.int64(3)
    out['signalNorm'] = np.float64(1.7)
    np.savez_compressed(os.path.join(HERE, 'realspace_apply.npz'), **out)

    # ---- gnfw ------------------------------------------------------------------------------------------------------
    x = np.logspace(-6, 2, 300)
    los = gnfw.tsz_profile_los(x, c=1.177, alpha=1.0510, beta=5.4905, gamma=0.3081)
    los2 = gnfw.tsz_profile_los(x, c=1.81, alpha=1.33, beta=4.13, gamma=0.31)
    g = gnfw.gnfw(x, 1.177, 1.0510, 5.4905, 0.3081) if hasattr(gnfw, 'gnfw') else np.zeros(0)
    np.savez_compressed(os.path.join(HERE, 'gnfw.npz'), x=x, los_arnaud=_plain(np.asarray(los)),
                        los_ppp=_plain(np.asarray(los2)), gnfw=_plain(np.asarray(g)))

    # ---- names -----------------------------------------------------------------------------------------------------
    rng = np.random.default_rng(41)
    ras = list(rng.uniform(0, 360, 400)) + [0.0001, 9.999999, 10.0, 149.99999999, 150.0, 359.9999, 3.3, 12.5, 180.0]
    decs = list(rng.uniform(-89, 89, 400)) + [-0.5, 0.5, 9.9999, -9.9999, 10.0, -10.0, 20.7, -20.7, 45.25]
    names = [[float(r), float(d), _plain(catalogs.makeName(float(r), float(d))),
              _plain(catalogs.makeLongName(float(r), float(d), prefix='SO-CL'))] for r, d in zip(ras, decs)]
    with open(os.path.join(HERE, 'names.json'), 'w') as f:
        json.dump(names, f)
    print("golden vectors written to", HERE)


if __name__ == '__main__':
    main()
