"""Catalogue rows, naming and merging for the hot path (reference: nemo/catalogs.py).

astropy.table is not available offline, so catalogues are held in :class:`Table`, a small column store with the
handful of operations the path uses (column access, row iteration with in-place updates, boolean selection,
vstack, sort).  Semantics follow:

* ``makeName`` / ``makeLongName``      - nemo/catalogs.py:436-595 (IAU-style, truncated not rounded)
* ``catalogListToTab``                 - nemo/catalogs.py:620-651
* ``selectFromCatalog``                - nemo/catalogs.py:598-617
* ``makeOptimalCatalog``               - nemo/catalogs.py:218-362 ('ref-first' and 'ref-forced')
* ``removeDuplicates``                 - nemo/catalogs.py:704-749
"""
import operator

import numpy as np

from .wcs import calcAngSepDeg

XMATCH_RADIUS_DEG = 1.4 / 60.0          # nemo/catalogs.py:28

COLUMN_NAMES = ['name', 'RADeg', 'decDeg', 'SNR', 'numSigPix', 'template', 'tileName', 'flags', 'ringFlag', 'galacticLatDeg',
                'x', 'y', 'fixed_SNR', 'deltaT_c', 'err_deltaT_c', 'fixed_deltaT_c', 'fixed_err_deltaT_c', 'y_c', 'err_y_c',
                'fixed_y_c', 'fixed_err_y_c', 'fluxJy', 'err_fluxJy', 'fixed_fluxJy', 'fixed_err_fluxJy',
                'ellipse_PA', 'ellipse_A', 'ellipse_B', 'ellipse_x0', 'ellipse_y0', 'ellipse_e']


class _Row(object):
    def __init__(self, tab, i):
        self._tab = tab
        self._i = i

    def __getitem__(self, key):
        return self._tab._cols[key][self._i]

    def __setitem__(self, key, value):
        self._tab._cols[key][self._i] = value

    def keys(self):
        return self._tab.keys()


class Table(object):
    """Minimal stand-in for astropy.table.Table (columns = numpy arrays of equal length)."""

    def __init__(self, cols=None, names=None):
        self._cols = {}
        self.meta = {}
        if cols is not None:
            for k, v in cols.items():
                self._cols[k] = np.array(v)
        elif names is not None:
            for k in names:
                self._cols[k] = np.zeros(0)

    def keys(self):
        return list(self._cols.keys())

    @property
    def colnames(self):
        return self.keys()

    def __len__(self):
        if len(self._cols) == 0:
            return 0
        return len(next(iter(self._cols.values())))

    def __iter__(self):
        for i in range(len(self)):
            yield _Row(self, i)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self._cols[key]
        if isinstance(key, (int, np.integer)):
            return _Row(self, int(key))
        out = Table()
        for k, v in self._cols.items():
            out._cols[k] = v[key]
        out.meta = dict(self.meta)
        return out

    def __setitem__(self, key, value):
        n = len(self)
        arr = np.array(value)
        if arr.ndim == 0 and len(self._cols) > 0:
            arr = np.full(n, value)
        self._cols[key] = arr

    def add_column(self, data, name):
        self[name] = data

    def remove_column(self, name):
        del self._cols[name]

    def copy(self):
        out = Table()
        for k, v in self._cols.items():
            out._cols[k] = v.copy()
        out.meta = dict(self.meta)
        return out

    def sort(self, keys):
        if isinstance(keys, str):
            keys = [keys]
        order = np.lexsort([self._cols[k] for k in reversed(keys)])
        for k in self._cols:
            self._cols[k] = self._cols[k][order]

    def to_records(self):
        return [{k: self._cols[k][i] for k in self._cols} for i in range(len(self))]


def vstack(tables):
    tables = [t for t in tables if len(t) > 0 or len(t.keys()) > 0]
    if len(tables) == 0:
        return Table()
    keys = tables[0].keys()
    out = Table()
    for k in keys:
        out._cols[k] = np.concatenate([t[k] if k in t.keys() else np.full(len(t), -99) for t in tables])
    return out


# ------------------------------------------------------------------------------------------------------------
def _truncStr(value, digitsBefore, digitsAfter):
    """Zero-padded, TRUNCATED decimal string (the reference slices '%.10f' strings; no rounding up)."""
    scaled = int(np.floor(value * 10 ** digitsAfter + 1e-9))
    whole, frac = divmod(scaled, 10 ** digitsAfter)
    s = ("%0" + str(digitsBefore) + "d") % (whole)
    if digitsAfter > 0:
        s = s + "." + ("%0" + str(digitsAfter) + "d") % (frac)
    return s


def makeName(RADeg, decDeg, prefix='ACT-CL'):
    """``prefix JHHMM.m+DDMM`` (nemo/catalogs.py:436-524)."""
    hours = (RADeg / 360.0) * 24
    h = int(hours)
    mins = (float("%.10f" % (hours)) - h) * 60
    sRA = "%02d%s" % (h, _truncStr(mins, 2, 1))
    a = abs(decDeg)
    d = int(a)
    dmins = (a - d) * 60
    sDec = ("+" if decDeg > 0 else "-") + "%02d%02d" % (d, int(dmins))
    return prefix + " J" + sRA + sDec


def makeLongName(RADeg, decDeg, prefix="ACT-CL"):
    """``prefix JHHMMSS.s+DDMMSS`` (nemo/catalogs.py:454-595)."""
    hours = (RADeg / 360.0) * 24
    h = int(hours)
    mins = (hours - h) * 60
    m = int(mins)
    secs = (mins - m) * 60
    a = abs(decDeg)
    d = int(a)
    dm = (a - d) * 60
    dmi = int(dm)
    ds = (dm - dmi) * 60
    return prefix + " J%02d%02d%s%s%02d%02d%02d" % (h, m, _truncStr(secs, 2, 1), "+" if decDeg > 0 else "-", d, dmi, int(ds))


def catalogListToTab(catalogList, keysToWrite=COLUMN_NAMES, minStrLength=30):
    """nemo/catalogs.py:620-651."""
    availKeys = list(catalogList[0].keys())
    tab = Table()
    for key in keysToWrite:
        if key in availKeys:
            arr = [obj[key] if obj[key] is not None else -99 for obj in catalogList]
            col = np.array(arr)
            if col.dtype.kind == 'U':
                col = col.astype('U%d' % (max(minStrLength, col.dtype.itemsize // 4)))
            tab._cols[key] = col
    return tab


_OPS = {'<': operator.lt, '>': operator.gt, '<=': operator.le, '>=': operator.ge, '==': operator.eq, '!=': operator.ne}


def selectFromCatalog(catalog, constraintsList):
    """nemo/catalogs.py:598-617: constraints of the form "key op value" (space separated)."""
    passed = catalog
    for constraintString in constraintsList:
        key, op, value = constraintString.split()
        try:
            v = float(value)
        except ValueError:
            v = value.strip("'\"")
        passed = passed[_OPS[op](passed[key], v)]
    return passed


def makeOptimalCatalog(catalogDict, constraintsList=[], photFilter=None, method='ref-first'):
    """nemo/catalogs.py:218-362."""
    if method == 'ref-first':
        cat = _makeOptimalCatalogRefFirst(catalogDict, constraintsList=constraintsList, photFilter=photFilter)
    elif method == 'ref-forced':
        cat = _makeOptimalCatalogForcedRef(catalogDict, constraintsList=constraintsList)
    else:
        raise Exception("There is no '%s' method - use either 'ref-first' or 'ref-forced'" % (method))
    if cat is None or len(cat) == 0:
        cat = Table(names=('SNR', 'RADeg', 'decDeg', 'name'))
    return cat


def _makeOptimalCatalogRefFirst(catalogDict, constraintsList=[], photFilter=None):
    refScale, merged = [], []
    for key in catalogDict.keys():
        if len(catalogDict[key]['catalog']) > 0:
            if key.split("#")[0] == photFilter:
                refScale.append(catalogDict[key]['catalog'])
            else:
                merged.append(catalogDict[key]['catalog'])
    refScale = vstack(refScale) if len(refScale) > 0 else None
    merged = vstack(merged) if len(merged) > 0 else []
    if refScale is None:
        refScale = merged
    colsToRewrite = ['SNR', 'template', 'y_c', 'err_y_c', 'deltaT_c', 'err_deltaT_c']
    if len(refScale) > 0 and len(merged) > 0:
        for row in refScale:
            rDeg = calcAngSepDeg(row['RADeg'], row['decDeg'], merged['RADeg'], merged['decDeg'])
            xIndices = np.where(rDeg < XMATCH_RADIUS_DEG)[0]
            if len(xIndices) > 1:
                best = xIndices[np.argmax(merged['SNR'][xIndices])]
                if merged['SNR'][best] > row['SNR']:
                    for c in colsToRewrite:
                        if c in refScale.keys():
                            row[c] = merged[c][best]
        refScale.sort(['RADeg', 'decDeg'])
        refScale = selectFromCatalog(refScale, constraintsList)
    return refScale


def _makeOptimalCatalogForcedRef(catalogDict, constraintsList=[]):
    allCatalogs = [catalogDict[k]['catalog'] for k in catalogDict.keys() if len(catalogDict[k]['catalog']) > 0]
    if len(allCatalogs) == 0:
        return Table(names=('SNR', 'RADeg', 'decDeg', 'name'))
    allCatalogs = vstack(allCatalogs)
    toRemove = np.zeros(len(allCatalogs))
    for row in allCatalogs:
        rDeg = calcAngSepDeg(row['RADeg'], row['decDeg'], allCatalogs['RADeg'], allCatalogs['decDeg'])
        xIndices = np.where(rDeg < XMATCH_RADIUS_DEG)[0]
        if len(xIndices) > 1:
            keep = xIndices[np.argmax(allCatalogs['SNR'][xIndices])]
            for index in xIndices:
                if index != keep:
                    toRemove[index] = 1
    merged = allCatalogs[toRemove == 0]
    merged.sort(['RADeg', 'decDeg'])
    return selectFromCatalog(merged, constraintsList)


def removeDuplicates(tab):
    """nemo/catalogs.py:704-749: keep the highest-SNR member of every group closer than XMATCH_RADIUS_DEG."""
    if len(tab) == 1:
        return tab, 1, []
    ra, dec = np.asarray(tab['RADeg'], dtype=float), np.asarray(tab['decDeg'], dtype=float)
    nearest = np.zeros(len(tab))
    for i in range(len(tab)):
        r = calcAngSepDeg(ra[i], dec[i], ra, dec)
        r[i] = np.inf
        nearest[i] = r.min()
    mask = nearest < XMATCH_RADIUS_DEG
    if mask.sum() == 0:
        return tab, 0, []
    dupTab, noDupTab = tab[mask], tab[~mask]
    keepMask = np.zeros(len(dupTab), dtype=bool)
    for i in range(len(dupTab)):
        r = calcAngSepDeg(dupTab['RADeg'][i], dupTab['decDeg'][i], dupTab['RADeg'], dupTab['decDeg'])
        m = np.less_equal(r, XMATCH_RADIUS_DEG)
        indices = np.where(m)[0]
        best = indices[np.equal(dupTab['SNR'][m], dupTab['SNR'][m].max())][0] if m.sum() > 0 else i
        keepMask[best] = True
    keepTab = vstack([dupTab[keepMask], noDupTab])
    keepTab.sort('RADeg')
    return keepTab, len(dupTab), list(dupTab['name']) if 'name' in dupTab.keys() else []
