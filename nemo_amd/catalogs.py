"""Catalogue rows, naming and merging for the hot path (reference: nemo/catalogs.py).

astropy.table is not available offline, so catalogues are held in :class:`Table`, a small column store with the
handful of operations the path uses (column access, row iteration with in-place updates, boolean selection,
vstack, sort).  Semantics follow:

* ``makeName`` / ``makeLongName``      - nemo/catalogs.py:436-595 (IAU-style; digits are cut from decimal strings)
* ``catalogListToTab``                 - nemo/catalogs.py:620-651
* ``selectFromCatalog``                - nemo/catalogs.py:598-617
* ``makeOptimalCatalog``               - nemo/catalogs.py:218-362 ('ref-first' and 'ref-forced')
* ``removeDuplicates``                 - nemo/catalogs.py:704-749
* ``getTableRADecKeys``                - nemo/catalogs.py:1089-1120

The cross-matching loops of the reference are O(rows x rows) Python; here neighbour candidates come from a k-d tree
on unit vectors and the reference's own test (``calcAngSepDeg(...) < XMATCH_RADIUS_DEG``) is then applied to the
candidate pairs, so the outcome is the same and the cost is O(rows log rows).
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
        return self._tab[key][self._i]

    def __setitem__(self, key, value):
        self._tab[key][self._i] = value

    def keys(self):
        return self._tab.keys()


class Table(object):
    """Minimal stand-in for astropy.table.Table (columns = numpy arrays of equal length).

    The ``name`` column may be *lazy* (:meth:`setLazyNames`): names are a pure function of (RADeg, decDeg), so they are
    only formatted when the column is first read - normally once, on the final merged catalogue - instead of for every
    candidate at every filter scale."""

    def __init__(self, cols=None, names=None):
        self._cols = {}
        self._nameSpec = None      # (prefix, longNames) while the name column is not materialised
        self.meta = {}
        if cols is not None:
            for k, v in cols.items():
                self._cols[k] = np.array(v)
        elif names is not None:
            for k in names:
                self._cols[k] = np.zeros(0)

    def setLazyNames(self, prefix, longNames=False):
        self._nameSpec = (prefix, bool(longNames))
        self._cols.pop('name', None)

    def _materialiseNames(self):
        if self._nameSpec is not None and 'RADeg' in self._cols:
            prefix, longNames = self._nameSpec
            if longNames:
                names = [makeLongName(r, d, prefix=prefix) for r, d in zip(self._cols['RADeg'].tolist(), self._cols['decDeg'].tolist())]
            else:
                names = makeNames(self._cols['RADeg'], self._cols['decDeg'], prefix=prefix)
            col = np.array(names, dtype='U30') if len(names) > 0 else np.zeros(0, dtype='U30')
            # keep 'name' as the first column, where catalogListToTab puts it
            self._cols = dict([('name', col)] + [(k, v) for k, v in self._cols.items() if k != 'name'])
        self._nameSpec = None

    def keys(self):
        ks = list(self._cols.keys())
        if self._nameSpec is not None and 'name' not in ks:
            ks = ['name'] + ks
        return ks

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
            if key == 'name' and self._nameSpec is not None:
                self._materialiseNames()
            return self._cols[key]
        if isinstance(key, (int, np.integer)):
            return _Row(self, int(key))
        out = Table()
        for k, v in self._cols.items():
            out._cols[k] = v[key]
        out._nameSpec = self._nameSpec
        out.meta = dict(self.meta)
        return out

    def __setitem__(self, key, value):
        n = len(self)
        arr = np.array(value)
        if arr.ndim == 0 and len(self._cols) > 0:
            arr = np.full(n, value)
        if key == 'name':
            self._nameSpec = None
        self._cols[key] = arr

    def add_column(self, data, name):
        self[name] = data

    def remove_column(self, name):
        del self._cols[name]

    def copy(self):
        out = Table()
        for k, v in self._cols.items():
            out._cols[k] = v.copy()
        out._nameSpec = self._nameSpec
        out.meta = dict(self.meta)
        return out

    def sort(self, keys):
        if isinstance(keys, str):
            keys = [keys]
        order = np.lexsort([self._cols[k] for k in reversed(keys)])
        for k in self._cols:
            self._cols[k] = self._cols[k][order]

    def to_records(self):
        keys = self.keys()
        cols = {k: self[k] for k in keys}
        return [{k: cols[k][i] for k in keys} for i in range(len(self))]


def vstack(tables):
    tables = [t for t in tables if len(t) > 0 or len(t.keys()) > 0]
    if len(tables) == 0:
        return Table()
    specs = set(t._nameSpec for t in tables)
    out = Table()
    if len(specs) == 1 and None not in specs:
        out._nameSpec = tables[0]._nameSpec          # every part still lazy, same naming rule: stay lazy
    else:
        for t in tables:
            t._materialiseNames()
    # columns: the union, in order of first appearance (astropy's outer join masks what a table lacks; here -99, the
    # catalogue's own "no value", or the empty string)
    keys, seen = [], set()
    keySets = [set(t.keys()) for t in tables]
    for t in tables:
        for k in t.keys():
            if k not in seen:
                seen.add(k)
                keys.append(k)
    if all(ks == keySets[0] for ks in keySets) and all(t._nameSpec is None or out._nameSpec is not None for t in tables):
        # the usual case - every table has the same columns (the scales of a run): the column dictionaries directly
        for k in keys:
            if k == 'name' and out._nameSpec is not None:
                continue
            out._cols[k] = np.concatenate([t._cols[k] for t in tables])
        return out
    for k in keys:
        if k == 'name' and out._nameSpec is not None:
            continue
        first = next(t for t, ks in zip(tables, keySets) if k in ks)[k]
        fill = '' if first.dtype.kind in 'US' else -99
        out._cols[k] = np.concatenate([t[k] if k in ks else np.full(len(t), fill) for t, ks in zip(tables, keySets)])
    return out


# ------------------------------------------------------------------------------------------------------------
# Names.  The reference cuts the digits out of decimal strings ("%.10f" for the RA hours / minutes, str() for the
# declination), so values are truncated, never rounded up, and inherit the quirks of those strings; the same string
# route is followed here so that names agree character for character (golden vectors: tests/golden/names.json).
def _afterPoint(text):
    return float(text[text.index("."):])


def _twoDigits(value, text, nKeep):
    """Leading field of a non-negative decimal string: zero-padded when the value is below ten."""
    return "0" + text[:nKeep - 1] if value < 10 else text[:nKeep]


def _raField(RADeg):
    hours = (RADeg / 360) * 24
    sh = "%.10f" % (hours)
    mins = _afterPoint(sh) * 60
    return _twoDigits(hours, sh, 2) + _twoDigits(mins, "%.10f" % (mins), 4)


def _decField(decDeg):
    s = str(decDeg)
    if decDeg > 0:
        sDeg = "+" + _twoDigits(decDeg, s, 2)
    else:
        sDeg = "-0" + s[1] if decDeg > -10 else s[:3]
    mins = _afterPoint(s) * 60
    return sDeg + _twoDigits(mins, str(mins), 2)


def _longRAField(RADeg):
    hours = (RADeg / 360) * 24
    sh = str(hours)
    mins = _afterPoint(sh) * 60
    sm = str(mins)
    secs = _afterPoint(sm) * 60
    return _twoDigits(hours, sh, 2) + _twoDigits(mins, sm, 2) + _twoDigits(secs, str(secs), 4)


def _longDecField(decDeg):
    s = str(decDeg)
    if decDeg > 0:
        sDeg = "+" + _twoDigits(decDeg, s, 2)
    else:
        sDeg = "-0" + s[1] if decDeg > -10 else s[:3]
    mins = _afterPoint(s) * 60
    sm = str(mins)
    secs = _afterPoint(sm) * 60
    return sDeg + _twoDigits(mins, sm, 2) + _twoDigits(secs, str(secs), 4)


def makeName(RADeg, decDeg, prefix='ACT-CL'):
    """``prefix JHHMM.m+DDMM`` (nemo/catalogs.py:436-451,472-524)."""
    return prefix + " J" + _raField(float(RADeg)) + _decField(float(decDeg))


def _makeNamesStrings(RAs, decs, head):
    """makeName for lists of floats: the reference's string route, written out in one loop (no helper calls per object)."""
    out = []
    for ra, dec in zip(RAs, decs):
        hours = (ra / 360) * 24
        sh = "%.10f" % (hours)
        mins = float(sh[sh.index("."):]) * 60
        sm = "%.10f" % (mins)
        raPart = ("0" + sh[:1] if hours < 10 else sh[:2]) + ("0" + sm[:3] if mins < 10 else sm[:4])
        s = str(dec)
        if dec > 0:
            sDeg = "+" + ("0" + s[:1] if dec < 10 else s[:2])
        else:
            sDeg = "-0" + s[1] if dec > -10 else s[:3]
        dm = float(s[s.index("."):]) * 60
        sdm = str(dm)
        out.append(head + raPart + sDeg + ("0" + sdm[:1] if dm < 10 else sdm[:2]))
    return out


def makeNames(RADeg, decDeg, prefix='ACT-CL'):
    """makeName for arrays.  The digits the string route cuts out of its decimal strings are computed as numbers - hours and
    minutes rounded to ten decimals as "%.10f" rounds them, fields truncated as the slices truncate them - and one format
    statement per object puts them together; an object whose value lies within 1e-6 of a place where a digit changes (or
    outside the ordinary ranges: |dec| below 1e-3 deg, minutes below 1e-3, RA outside 0..360) goes through the strings
    themselves, so every name is the string route's, character for character (tests/test_cpu_boundary.py)."""
    RA = np.atleast_1d(np.asarray(RADeg, dtype=np.float64))
    dec = np.atleast_1d(np.asarray(decDeg, dtype=np.float64))
    head = prefix + " J"
    n = len(RA)
    if n < 16:
        return _makeNamesStrings(RA.tolist(), dec.tolist(), head)
    with np.errstate(invalid='ignore', over='ignore'):
        hours = (RA / 360) * 24
        R = np.rint(hours * 1e10)
        ok = np.isfinite(R) & (hours >= 0) & (hours < 24)
        Ri = np.where(ok, R, 0).astype(np.int64)
        HH = Ri // 10 ** 10
        mins = ((Ri % 10 ** 10) / 1e10) * 60                  # float(".dddddddddd") * 60
        t = mins * 10
        ok &= np.abs(t - np.rint(t)) > 1e-5                   # away from the places where the tenth of a minute changes
        Mi = np.rint(mins * 1e10).astype(np.int64)
        MM = Mi // 10 ** 10
        tenth = (Mi // 10 ** 9) % 10
        ad = np.abs(dec)
        ok &= np.isfinite(dec) & (ad > 1e-3) & (ad < 90)
        D = np.floor(np.where(ok, ad, 0.0))
        dm = (np.where(ok, ad, 0.0) - D) * 60
        ok &= (np.abs(dm - np.rint(dm)) > 1e-6) & (dm > 1e-3)
        DM = np.floor(dm).astype(np.int64)
        Di = D.astype(np.int64)
    sign = np.where(dec > 0, '+', '-').tolist()
    out = [head + "%02d%02d.%d%s%02d%02d" % f for f in zip(HH.tolist(), MM.tolist(), tenth.tolist(), sign, Di.tolist(), DM.tolist())]
    bad = np.flatnonzero(~ok)
    if len(bad) > 0:
        for i, name in zip(bad.tolist(), _makeNamesStrings(RA[bad].tolist(), dec[bad].tolist(), head)):
            out[i] = name
    return out


def makeLongName(RADeg, decDeg, prefix="ACT-CL"):
    """``prefix JHHMMSS.s+DDMMSS`` (nemo/catalogs.py:454-469,527-595)."""
    return prefix + " J" + _longRAField(float(RADeg)) + _longDecField(float(decDeg))


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


def getTableRADecKeys(tab):
    """nemo/catalogs.py:1089-1120: names of the RA / dec columns of a table."""
    RAKeysToTry = ['ra', 'RA', 'RADeg']
    decKeysToTry = ['dec', 'DEC', 'decDeg', 'Dec']
    RAKey, decKey = None, None
    for key in RAKeysToTry:
        if key in tab.keys():
            RAKey = key
            break
    for key in decKeysToTry:
        if key in tab.keys():
            decKey = key
            break
    if RAKey is None or decKey is None:
        raise Exception("Couldn't identify RA, dec columns in the supplied table.")
    return RAKey, decKey


_OPS = {'<': operator.lt, '>': operator.gt, '<=': operator.le, '>=': operator.ge, '==': operator.eq, '!=': operator.ne}


def getCatalogWithinImage(tab, shape, wcs, mask=None):
    """Rows whose coordinates fall inside the image - 0 <= x < W - 1 and 0 <= y < H - 1 - and, with ``mask``, on a pixel where
    the mask is 1 (nemo/catalogs.py:1039-1086)."""
    if len(tab) == 0:
        return tab
    x, y = wcs.wcs2pix(np.asarray(tab['RADeg'], dtype=np.float64), np.asarray(tab['decDeg'], dtype=np.float64))
    x, y = np.atleast_1d(x), np.atleast_1d(y)
    with np.errstate(invalid='ignore'):
        sel = ~np.isnan(x) & ~np.isnan(y) & (x >= 0) & (x < shape[1] - 1) & (y >= 0) & (y < shape[0] - 1)
    if mask is not None and sel.any():
        iy, ix = np.round(y[sel]).astype(np.int64), np.round(x[sel]).astype(np.int64)
        if hasattr(mask, 'is_cuda'):
            from . import engine
            t = engine.torch()
            onMask = mask[t.as_tensor(iy, device=mask.device), t.as_tensor(ix, device=mask.device)].cpu().numpy() == 1
        else:
            onMask = np.asarray(mask)[iy, ix] == 1
        idx = np.flatnonzero(sel)
        sel[idx[~onMask]] = False
    return tab[sel]


def writeCatalog(catalog, outFileName, constraintsList=[]):
    """nemo/catalogs.py:675-701: the catalogue (after the cuts) as a FITS table, or tab-delimited text for ``.csv``; the
    build's version goes into the header as NEMOVER.  Nothing is written for an empty catalogue."""
    from . import fitsio, __version__
    if len(catalog) == 0:
        return None
    cut = selectFromCatalog(catalog, constraintsList)
    cut.meta['NEMOVER'] = __version__
    keys = cut.keys()
    if outFileName.split(".")[-1] == 'csv':
        with open(outFileName, "w") as f:
            f.write("\t".join(keys) + "\n")
            cols = [np.asarray(cut[k]) for k in keys]
            for i in range(len(cut)):
                f.write("\t".join(str(c[i]) for c in cols) + "\n")
    else:
        fitsio.writeTable(outFileName, {k: np.asarray(cut[k]) for k in keys}, meta=cut.meta)


def readCatalog(fileName):
    """A catalogue written by :func:`writeCatalog` (FITS table or tab-delimited ``.csv``) as a :class:`Table`."""
    from . import fitsio
    if fileName.split(".")[-1] == 'csv':
        with open(fileName) as f:
            lines = [ln.rstrip("\n") for ln in f if ln.strip() != ""]
        keys = lines[0].split("\t")
        raw = [ln.split("\t") for ln in lines[1:]]
        cols = {}
        for j, k in enumerate(keys):
            vals = [r[j] for r in raw]
            if all(v in ('True', 'False') for v in vals):
                cols[k] = np.array([v == 'True' for v in vals])
                continue
            try:
                cols[k] = np.array([int(v) for v in vals])
            except ValueError:
                try:
                    cols[k] = np.array([float(v) for v in vals])
                except ValueError:
                    cols[k] = np.array(vals)
        return Table(cols=cols)
    data, header = fitsio.readTable(fileName)
    tab = Table(cols=data)
    if 'NEMOVER' in header:
        tab.meta['NEMOVER'] = header['NEMOVER']
    return tab


def catalog2DS9(catalog, outFileName, constraintsList=[], addInfo=[], idKeyToUse='name', RAKeyToUse='RADeg', decKeyToUse='decDeg',
                color="cyan", showNames=True, writeNemoInfo=True, coordSys='fk5', regionShape='point', width=1, radiusArcsec=360):
    """A DS9 region file of the catalogue (nemo/catalogs.py:365-433)."""
    import datetime
    from . import __version__
    cut = selectFromCatalog(catalog, constraintsList)
    with open(outFileName, "w") as outFile:
        comment = "# DS9 region file"
        if writeNemoInfo == True:
            comment = comment + " generated by nemo (version: %s on %s)\n" % (__version__, datetime.datetime.today().date().isoformat())
        else:
            comment = comment + "\n"
        outFile.write(comment)
        outFile.write('global dashlist=8 3 width=%d font="helvetica 10 normal" select=1 highlite=1 dash=0 fixed=0 edit=1 move=1 '
                      'delete=1 include=1 source=1\n' % (width))
        for obj in cut:
            infoString = ""
            if len(addInfo) > 0:
                for d in addInfo:
                    if infoString != "":
                        infoString = infoString + " "
                    infoString = infoString + d['fmt'] % (obj[d['key']])
                infoString = " [" + infoString + "]"
            colorString = obj['color'] if color == 'key' else color
            if showNames == True:
                infoString = str(obj[idKeyToUse]) + infoString
            if regionShape == 'point':
                outFile.write("%s;point(%.6f,%.6f) # point=cross color={%s} text={%s}\n"
                              % (coordSys, obj[RAKeyToUse], obj[decKeyToUse], colorString, infoString))
            elif regionShape == 'circle':
                outFile.write('%s;circle(%.6f,%.6f,%.1f") # color={%s} text={%s}\n'
                              % (coordSys, obj[RAKeyToUse], obj[decKeyToUse], radiusArcsec, colorString, infoString))


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


# ------------------------------------------------------------------------------------------------------------
GPU_XMATCH_MIN_ROWS = 1024
GPU_XMATCH_MAX_ROWS = 65536      # the device test is all pairs (n^2 / 2 chords): beyond this the host's sorted sweep (n log n) takes over


def _unitVectors(RADeg, decDeg):
    ra, dec = np.radians(np.asarray(RADeg, dtype=np.float64)), np.radians(np.asarray(decDeg, dtype=np.float64))
    return np.stack([np.cos(dec) * np.cos(ra), np.cos(dec) * np.sin(ra), np.sin(dec)], axis=1)


def _pairsWithin(RA1, dec1, RA2, dec2, radiusDeg, inclusive=False):
    """All (i, j) with calcAngSepDeg(row i of set 1, row j of set 2) < radiusDeg (<= if inclusive), sorted by (i, j).
    Candidates by a sweep along z = sin(dec) (two unit vectors closer than a chord differ by less than it in every
    coordinate; set 2 sorted by z, one searchsorted window per row of set 1), cut by chord length with a 0.1 % margin, then
    decided by the exact great-circle formula where the chord is not conclusive."""
    if len(RA1) == 0 or len(RA2) == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    lim = 2.0 * np.sin(np.radians(radiusDeg) / 2.0)
    chord = lim * 1.001 + 1e-12
    sameSets = len(RA1) == len(RA2) and np.array_equal(np.asarray(RA1), np.asarray(RA2)) and \
        np.array_equal(np.asarray(dec1), np.asarray(dec2))
    u1 = _unitVectors(RA1, dec1)
    u2 = u1 if sameSets else _unitVectors(RA2, dec2)
    n2 = np.int64(len(u2))
    order = np.argsort(u2[:, 2], kind='stable')
    zs = u2[order, 2]
    lo = np.searchsorted(zs, u1[:, 2] - chord, side='left')
    hi = np.searchsorted(zs, u1[:, 2] + chord, side='right')
    cnt = hi - lo
    i = np.repeat(np.arange(len(u1), dtype=np.int64), cnt)
    k = np.arange(int(cnt.sum()), dtype=np.int64) + np.repeat(lo - (np.cumsum(cnt) - cnt), cnt)
    j = order[k].astype(np.int64)
    d = u1[i] - u2[j]
    chord2 = np.einsum('ij,ij->i', d, d)
    cand = chord2 <= chord * chord
    key = np.sort(i[cand] * n2 + j[cand])
    i, j = key // n2, key % n2
    # Decision by the exact great-circle formula (what the reference evaluates) - but only for the pairs whose chord is
    # within 1e-9 (relative) of the limit; the chord of two unit vectors is good to ~1e-12 of it, so the others are decided.
    d = u1[i] - u2[j]
    chord2 = np.einsum('ij,ij->i', d, d)
    ok = chord2 < (lim * (1.0 - 1e-9)) ** 2
    near = np.flatnonzero(~ok & (chord2 <= (lim * (1.0 + 1e-9)) ** 2))
    if len(near) > 0:
        r = calcAngSepDeg(np.asarray(RA1, dtype=np.float64)[i[near]], np.asarray(dec1, dtype=np.float64)[i[near]],
                          np.asarray(RA2, dtype=np.float64)[j[near]], np.asarray(dec2, dtype=np.float64)[j[near]])
        ok[near] = (r <= radiusDeg) if inclusive else (r < radiusDeg)
    return i[ok], j[ok]


def _bestPerGroup(i, j, SNR, n):
    """For pairs (i, j) sorted by (i, j): per i, the count of partners and the FIRST partner (lowest j) with the highest
    SNR - what ``xIndices[np.argmax(SNR[xIndices])]`` returns.  best = -1 where i has no partner."""
    count = np.bincount(i, minlength=n)
    best = np.full(n, -1, dtype=np.int64)
    if len(i) > 0:
        # pairs are sorted by (i, j): segmented maximum of SNR[j], then the first position attaining it (no sort needed)
        starts = np.flatnonzero(np.concatenate([[True], i[1:] != i[:-1]]))
        sj = np.asarray(SNR, dtype=np.float64)[j]
        segMax = np.maximum.reduceat(sj, starts)
        seg = np.cumsum(np.concatenate([[0], (i[1:] != i[:-1]).astype(np.int64)]))
        pos = np.where(sj == segMax[seg], np.arange(len(i)), len(i))
        firstPos = np.minimum.reduceat(pos, starts)
        ok = firstPos < len(i)                 # a group whose SNRs are all NaN has no maximum position: argmax gives 0
        firstPos = np.where(ok, firstPos, starts)
        best[i[starts]] = j[firstPos]
    return count, best


def makeOptimalCatalog(catalogDict, constraintsList=[], photFilter=None, method='ref-first'):
    """nemo/catalogs.py:218-255."""
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
    """nemo/catalogs.py:258-316."""
    refScale, merged = [], []
    for key in catalogDict.keys():
        if len(catalogDict[key]['catalog']) > 0:
            if key.split("#")[0] == photFilter:
                refScale.append(catalogDict[key]['catalog'])
            else:
                merged.append(catalogDict[key]['catalog'])
    refScale = vstack(refScale) if len(refScale) > 0 else None
    merged = vstack(merged) if len(merged) > 0 else []
    sameTable = refScale is None
    if sameTable:
        refScale = merged
    colsToRewrite = ['SNR', 'template', 'y_c', 'err_y_c', 'deltaT_c', 'err_deltaT_c']
    if len(refScale) > 0 and len(merged) > 0:
        i, j = _pairsWithin(refScale['RADeg'], refScale['decDeg'], merged['RADeg'], merged['decDeg'], XMATCH_RADIUS_DEG)
        cols = [c for c in colsToRewrite if c in refScale.keys()]
        if not sameTable:
            count, best = _bestPerGroup(i, j, merged['SNR'], len(refScale))
            rows = np.where(count > 1)[0]
            rows = rows[merged['SNR'][best[rows]] > refScale['SNR'][rows]]
            for c in cols:
                refScale[c][rows] = merged[c][best[rows]]
        else:
            # the reference updates the table it is searching (refScaleCatalog is mergedCatalog, :297-298): later rows see
            # earlier rewrites, so rows with more than one partner are visited in order
            count = np.bincount(i, minlength=len(refScale))
            start = np.concatenate([[0], np.cumsum(count)])
            SNR = refScale['SNR']
            for row in np.where(count > 1)[0]:
                xIndices = j[start[row]:start[row + 1]]
                b = xIndices[np.argmax(SNR[xIndices])]
                if SNR[b] > SNR[row]:
                    for c in cols:
                        refScale[c][row] = refScale[c][b]
        refScale.sort(['RADeg', 'decDeg'])
        refScale = selectFromCatalog(refScale, constraintsList)
    return refScale


def _makeOptimalCatalogForcedRef(catalogDict, constraintsList=[]):
    """nemo/catalogs.py:319-362."""
    allCatalogs = [catalogDict[k]['catalog'] for k in catalogDict.keys() if len(catalogDict[k]['catalog']) > 0]
    if len(allCatalogs) == 0:
        return Table(names=('SNR', 'RADeg', 'decDeg', 'name'))
    allCatalogs = vstack(allCatalogs)
    n = len(allCatalogs)
    toRemove = None
    if GPU_XMATCH_MIN_ROWS <= n <= GPU_XMATCH_MAX_ROWS:
        # the all-pairs test is a few tens of microseconds on the device the maps were filtered on; identical outcome
        # (tests/test_gpu_kernels.py::test_xmatch_self), and the host path below decides when a chord is too close to call
        from . import engine
        if engine.gpu_ready():
            res = engine.xmatch_self(allCatalogs['RADeg'], allCatalogs['decDeg'], allCatalogs['SNR'], XMATCH_RADIUS_DEG)
            if res is not None:
                toRemove = res[2]
    if toRemove is None:
        i, j = _pairsWithin(allCatalogs['RADeg'], allCatalogs['decDeg'], allCatalogs['RADeg'], allCatalogs['decDeg'],
                            XMATCH_RADIUS_DEG)
        count, best = _bestPerGroup(i, j, allCatalogs['SNR'], n)
        toRemove = np.zeros(n, dtype=bool)
        drop = (count[i] > 1) & (j != best[i])
        toRemove[j[drop]] = True
    merged = allCatalogs[~toRemove]
    merged.sort(['RADeg', 'decDeg'])
    return selectFromCatalog(merged, constraintsList)


def removeDuplicates(tab):
    """nemo/catalogs.py:704-749: keep the highest-SNR member of every group closer than XMATCH_RADIUS_DEG.  Returns (table sorted
    by RADeg, number of rows that had a neighbour, their names).  With several GPUs every rank runs this on the gathered
    catalogue of all ranks after every pass: from GPU_XMATCH_MIN_ROWS rows on the all-pairs test runs on the device
    (engine.xmatch_self: counts and best partners of all rows; the host decides as before when a separation is too close to the
    radius to call), and the kept rows are taken from the table once, by index."""
    n = len(tab)
    if n == 1:
        return tab, 1, []
    ra, dec = np.asarray(tab['RADeg'], dtype=float), np.asarray(tab['decDeg'], dtype=float)
    dupIdx = keepDup = None
    if GPU_XMATCH_MIN_ROWS <= n <= GPU_XMATCH_MAX_ROWS:
        from . import engine
        if engine.gpu_ready():
            res = engine.xmatch_self(ra, dec, tab['SNR'], XMATCH_RADIUS_DEG)
            if res is not None:
                # partners of a row (itself included) closer than the radius, and the first of them with the highest SNR: a row
                # with a partner other than itself is a duplicate (match_coordinates_sky, :722-723), and its partners are
                # duplicates too, so the best partner inside the table of duplicates (:731-741) is the best partner here
                count, best = res[0], res[1]
                dupIdx = np.flatnonzero(count > 1)
                keepDup = np.unique(best[dupIdx])
    if dupIdx is None:
        i, j = _pairsWithin(ra, dec, ra, dec, XMATCH_RADIUS_DEG)
        other = i != j
        mask = np.zeros(n, dtype=bool)             # nearest OTHER row closer than the radius
        mask[i[other]] = True
        dupIdx = np.flatnonzero(mask)
        if len(dupIdx) > 0:
            di, dj = _pairsWithin(ra[dupIdx], dec[dupIdx], ra[dupIdx], dec[dupIdx], XMATCH_RADIUS_DEG, inclusive=True)
            count, best = _bestPerGroup(di, dj, np.asarray(tab['SNR'])[dupIdx], len(dupIdx))
            best = np.where(best < 0, np.arange(len(dupIdx)), best)
            keepDup = dupIdx[np.unique(best)]
    if len(dupIdx) == 0:
        return tab, 0, []
    # vstack([duplicates kept, rows without neighbour]) sorted by RADeg (stable), as one selection of rows
    isDup = np.zeros(n, dtype=bool)
    isDup[dupIdx] = True
    idx = np.concatenate([keepDup, np.flatnonzero(~isDup)])
    idx = idx[np.lexsort([tab['RADeg'][idx]])]
    keepTab = tab[idx]
    dupNames = list(tab[dupIdx]['name']) if 'name' in tab.keys() else []
    return keepTab, len(dupIdx), dupNames


# ------------------------------------------------------------------------------------------------------------
# Source-injection helpers (nemo/catalogs.py:849-1008)
def _dilatedAreaMask(areaMask, maskDilationPix):
    """``for i in range(n): m = 1 - binary_dilation(1 - m)`` (nemo/catalogs.py:899-900) in one pass: n dilations by the
    4-connected cross grow the holes by the city-block ball of radius n, so a pixel survives when its taxicab distance to the
    nearest zero pixel exceeds n (the map border is not a hole: binary_dilation pads with background)."""
    from scipy import ndimage
    m = np.asarray(areaMask) != 0
    if maskDilationPix <= 0 or m.all():
        return m
    return ndimage.distance_transform_cdt(m, metric='taxicab') > maskDilationPix


class AreaMasks(object):
    """What generateTestCatalog reads of a ``completeness.SelFn`` (nemo/catalogs.py:882-897): ``tileNames``, ``areaMaskDict``
    and ``WCSDict``.  Built from the area masks a pass left in memory (``config.areaMaskDict``) or on disk (selFn/areaMask.fits);
    keeps, per tile, the valid pixels after dilation - they are the same in every iteration of an injection run."""

    def __init__(self, tileNames, areaMaskDict, WCSDict):
        self.tileNames = list(tileNames)
        self.areaMaskDict = areaMaskDict
        self.WCSDict = WCSDict
        self._valid = {}

    @classmethod
    def fromConfig(cls, config, tileNames=None):
        from . import completeness
        tileNames = list(config.tileNames if tileNames is None else tileNames)
        masks, wcss = {}, {}
        for tileName in tileNames:
            m = getattr(config, 'areaMaskDict', {}).get(tileName) if hasattr(config, 'areaMaskDict') else None
            if m is not None:
                m = m.cpu().numpy() if hasattr(m, 'is_cuda') else np.asarray(m)
                wcs = config.tileCoordsDict[tileName]['wcs']
            else:
                if config.selFnDir is None:
                    raise Exception("source injection needs the area mask of tile '%s': run the pipeline with writeAreaMask=True "
                                    "first (or give a selFnDir that holds areaMask.fits)" % (tileName))
                m, wcs = completeness.loadAreaMask(tileName, config.selFnDir)
            masks[tileName], wcss[tileName] = m, wcs
        return cls(tileNames, masks, wcss)

    def validPixels(self, tileName, maskDilationPix):
        key = (tileName, int(maskDilationPix))
        if key not in self._valid:
            self._valid[key] = np.where(_dilatedAreaMask(self.areaMaskDict[tileName], maskDilationPix))
        return self._valid[key]


def _drawTilePositions(ys, xs, wcs, numSourcesPerTile, avoidanceRadiusArcmin, referenceStream):
    """Sky positions of up to ``numSourcesPerTile`` objects on the valid pixels (ys, xs) of one tile, each candidate a
    uniformly drawn valid pixel plus a uniform sub-pixel offset, kept when farther than the avoidance radius from every
    object kept before it (nemo/catalogs.py:901-925).

    ``referenceStream``: consume the global ``np.random`` stream exactly as the reference does - two offsets and one index per
    VALID PIXEL of the tile (1.5 million draws x 3 for a DR6 tile, ~30 ms), of which the loop then visits a few hundred -
    so that a seeded stream reproduces the reference's catalogue.  Otherwise candidates are drawn in batches as they are
    needed: the same distribution from far fewer draws (the injection loop's default; its host time would otherwise exceed
    the GPU's re-filter time forty-fold)."""
    n = len(ys)
    tileRAs, tileDecs = [], []
    if referenceStream:
        ysj = ys + np.random.uniform(0, 1, n)
        xsj = xs + np.random.uniform(0, 1, n)
        indices = np.random.randint(0, n, n)

        def batches():
            # the reference converts xs[indices], ys[indices] and then reads entry i of THAT list for i in indices (:905-912)
            for a in range(0, n, 512):
                pick = indices[indices[a:a + 512]]
                yield wcs.pix2wcs(xsj[pick], ysj[pick])
    else:
        def batches():
            tried = 0
            size = max(64, 4 * numSourcesPerTile)
            while tried < n:
                pick = np.random.randint(0, n, size)
                yield wcs.pix2wcs(xs[pick] + np.random.uniform(0, 1, size), ys[pick] + np.random.uniform(0, 1, size))
                tried += size
    # The reference tests one candidate at a time against the objects kept so far (:913-921): "kept unless
    # not calcAngSepDeg(...).min() * 60 > radius".  Same decisions here, with the separations of 64 candidates at a time formed
    # in one call each (candidate first, kept object second, as there): against the objects kept before the chunk, and among
    # the chunk's own candidates.
    for ra, dec in batches():
        ra, dec = np.atleast_1d(ra), np.atleast_1d(dec)
        for c0 in range(0, len(ra), 64):
            cra, cdec = ra[c0:c0 + 64], dec[c0:c0 + 64]
            blocked = np.zeros(len(cra), dtype=bool)
            if len(tileRAs) > 0:
                d1 = calcAngSepDeg(cra[:, None], cdec[:, None], np.array(tileRAs)[None, :], np.array(tileDecs)[None, :])
                blocked = (~(d1 * 60 > avoidanceRadiusArcmin)).any(axis=1)
            close = ~(calcAngSepDeg(cra[:, None], cdec[:, None], cra[None, :], cdec[None, :]) * 60 > avoidanceRadiusArcmin)
            for i in range(len(cra)):
                if blocked[i]:
                    continue
                tileRAs.append(float(cra[i]))
                tileDecs.append(float(cdec[i]))
                if len(tileRAs) == numSourcesPerTile:
                    return tileRAs, tileDecs
                blocked[i + 1:] |= close[i + 1:, i]
    return tileRAs, tileDecs


def generateTestCatalog(config, numSourcesPerTile, amplitudeColumnName='fixed_y_c', amplitudeRange=[0.001, 1],
                        amplitudeDistribution='linear', selFn=None, avoidanceRadiusArcmin=20.0, maskDilationPix=0, tileNames=None,
                        referenceStream=True):
    """Objects at random positions inside the area masks with random amplitudes (nemo/catalogs.py:849-947); at most
    ``numSourcesPerTile`` per tile, none closer than ``avoidanceRadiusArcmin`` to an earlier one of its tile.  With
    ``referenceStream`` (the default here; maps.sourceInjectionTest turns it off unless the configuration asks) the global
    ``np.random`` stream is consumed in the reference's order, so a seeded stream gives the reference's catalogue
    (tests/golden/injection.npz); see :func:`_drawTilePositions`.  ``selFn``: anything with ``tileNames`` / ``areaMaskDict`` /
    ``WCSDict`` (:class:`AreaMasks`)."""
    if selFn is None:
        selFn = AreaMasks.fromConfig(config, tileNames=tileNames)
    if tileNames is None:
        tileNames = selFn.tileNames
    RAs, decs, amps, catTileNames = [], [], [], []
    for tileName in tileNames:
        mapData = selFn.areaMaskDict[tileName]
        if hasattr(selFn, 'validPixels'):
            ys, xs = selFn.validPixels(tileName, maskDilationPix)
            empty = len(ys) == 0
        else:
            empty = np.asarray(mapData).sum() == 0                      # :895
            if not empty:
                ys, xs = np.where(_dilatedAreaMask(mapData, maskDilationPix))
        if empty or len(ys) == 0:         # nothing (left after the dilation; the reference's randint(0, 0) would raise there)
            continue
        tileRAs, tileDecs = _drawTilePositions(ys, xs, selFn.WCSDict[tileName], numSourcesPerTile, avoidanceRadiusArcmin,
                                               referenceStream)
        RAs = RAs + tileRAs
        decs = decs + tileDecs
        if amplitudeDistribution == 'linear':
            amp = np.random.uniform(amplitudeRange[0], amplitudeRange[1], len(tileRAs))
        elif amplitudeDistribution == 'log':
            amp = np.power(10, np.random.uniform(np.log10(amplitudeRange)[0], np.log10(amplitudeRange)[1], len(tileRAs)))
        else:
            raise Exception("Must be either 'linear' or 'log'.")
        amps = amps + amp.tolist()
        catTileNames = catTileNames + [tileName] * len(amp)
    tab = Table()
    tab['name'] = np.arange(0, len(amps)) + 1
    tab['RADeg'] = np.array(RAs, dtype=np.float64)
    tab['decDeg'] = np.array(decs, dtype=np.float64)
    tab[amplitudeColumnName] = np.array(amps, dtype=np.float64)
    tab['tileName'] = np.array(catTileNames, dtype='U30') if len(catTileNames) > 0 else np.zeros(0, dtype='U30')
    return tab


def _nearestNeighbour(RA1, dec1, RA2, dec2):
    """astropy's match_coordinates_sky(cat1, cat2, nthneighbor=1): for every row of set 1 the index of the nearest row of
    set 2 and the separation in degrees.  Nearest by chord length between unit vectors (monotonic in the angle); the
    separation is then the great-circle formula.  astropy is absent offline: PARITY UNPINNED (ties, last-bit separations)."""
    from scipy.spatial import cKDTree
    u1, u2 = _unitVectors(RA1, dec1), _unitVectors(RA2, dec2)
    _, idx = cKDTree(u2).query(u1, k=1)
    idx = np.asarray(idx, dtype=np.int64)
    rDeg = calcAngSepDeg(np.asarray(RA1, dtype=np.float64), np.asarray(dec1, dtype=np.float64),
                         np.asarray(RA2, dtype=np.float64)[idx], np.asarray(dec2, dtype=np.float64)[idx])
    return idx, np.atleast_1d(rDeg)


def crossMatch(refCatalog, matchCatalog, radiusArcmin=2.5):
    """nemo/catalogs.py:950-980: every row of ``refCatalog`` whose nearest row of ``matchCatalog`` lies closer than the radius,
    that nearest row, and the separations in degrees - three outputs in matching row order."""
    RAKey1, decKey1 = getTableRADecKeys(refCatalog)
    RAKey2, decKey2 = getTableRADecKeys(matchCatalog)
    xIndices, rDeg = _nearestNeighbour(refCatalog[RAKey1], refCatalog[decKey1], matchCatalog[RAKey2], matchCatalog[decKey2])
    mask = np.less(rDeg, radiusArcmin / 60.)
    return refCatalog[mask], matchCatalog[xIndices][mask], rDeg[mask]


def removeCrossMatched(refCatalog, matchCatalog, radiusArcmin=2.5):
    """nemo/catalogs.py:983-1008: ``refCatalog`` without the rows that have a row of ``matchCatalog`` within the radius."""
    if len(matchCatalog) == 0:
        return refCatalog                 # (the reference would fail inside astropy on an empty catalogue)
    RAKey1, decKey1 = getTableRADecKeys(refCatalog)
    RAKey2, decKey2 = getTableRADecKeys(matchCatalog)
    xIndices, rDeg = _nearestNeighbour(refCatalog[RAKey1], refCatalog[decKey1], matchCatalog[RAKey2], matchCatalog[decKey2])
    return refCatalog[np.greater(rDeg, radiusArcmin / 60.)]
