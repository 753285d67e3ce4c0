import os, sys
import numpy as np
sys.path[:0] = ['.', 'oracle', 'tests']
import helpers
from nemo_amd import filters, engine
g = np.load('tests/golden/realspace_build.npz')
k = 2
mapsL, wcs, meta = helpers.load_reference_case(g, k)
p = 'c%d_' % k
import tempfile
tmp = tempfile.mkdtemp()
orig = filters.MatchedFilter.buildAndApply
keep = {}
def spy(self, *a, **kw):
    r = orig(self, *a, **kw)
    keep['inner'] = self
    return r
filters.MatchedFilter.buildAndApply = spy
got, fobj = filters.filterMaps(helpers.product_maps(mapsL), {'label': 'rs', 'class': meta['class'], 'params': dict(meta['params'])},
                               'PRIMARY', diagnosticsDir=tmp, selFnDir=tmp, verbose=False, undoPixelWindow=False, returnFilter=True)
inner = keep['inner']
fr = g[p + 'innerFilt']
fp = inner.filt
print('inner filt same', np.abs(fp - fr).max() / np.abs(fr).max(), 'swapped', np.abs(fp[::-1] - fr).max() / np.abs(fr).max())
print('freqs inner', [m['obsFreqGHz'] for m in inner.unfilteredMapsDictList], 'outer', [m['obsFreqGHz'] for m in fobj.unfilteredMapsDictList])
rs = inner._realSpaceFilter()
ref = np.fft.fftshift(np.fft.ifft2(fr).real, axes=(-2, -1))
print('realspace same', np.abs(rs - ref).max() / np.abs(ref).max(), 'swapped', np.abs(rs[::-1] - ref).max() / np.abs(ref).max())
