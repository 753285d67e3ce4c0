"""Source finding and photometry on the GPU with the reference's call surface (nemo/photometry.py:25-444).

``getObjectPositions``: threshold + 4-connected labelling (scipy.ndimage.label numbering) + per-object
centre of mass / peak position / pixel count, all in libnemofilt kernels.
``findObjects``: ring flagging (second labelling pass at ``ringThresholdSigma``), bicubic-spline S/N at the
sub-pixel positions (windowed not-a-knot spline = RectBivariateSpline(kx=ky=3)), flag lookup, catalogue rows.
``getSNRValues`` / ``measureFluxes`` / ``makeForcedPhotometryCatalog``: spline sampling of the S/N and filtered
maps at catalogue positions.
Only per-object scalars cross to the host, as whole columns (no per-object Python loop).
"""
import os

import numpy as np

from . import engine, catalogs
from .signals import convertToDeltaT
from .wcs import calcAngSepDeg


def galacticLatDeg(RADeg, decDeg):
    """J2000 -> galactic latitude (stands in for astCoords.convertCoords at nemo/photometry.py:118)."""
    ra, dec = np.radians(RADeg), np.radians(decDeg)
    raNGP, decNGP = np.radians(192.85948), np.radians(27.12825)
    return np.degrees(np.arcsin(np.sin(dec) * np.sin(decNGP) + np.cos(dec) * np.cos(decNGP) * np.cos(ra - raNGP)))


_galacticLatDeg = galacticLatDeg


_TIMING = os.environ.get('NEMO_AMD_TIMING') == '1'
_acc = {}


def _mark(key, t0):
    """Diagnostics (NEMO_AMD_TIMING=1): accumulate host wall time per phase of the detection worker."""
    if _TIMING:
        import time
        now = time.perf_counter()
        _acc[key] = _acc.get(key, 0.0) + (now - t0)
        return now
    return t0


def _segment(eng, dMap, threshold, findCenterOfMass=True, negate=False, slot=None):
    """Device part of getObjectPositions; returns (numObjects, labels tensor, stats tensor [n+1, 6]).  ``slot``: the labels
    may live in the engine's kept buffer of that name (they are not handed to the caller)."""
    labels, numObjects = eng.label(dMap, threshold, negate=negate, slot=slot)
    stats = eng.object_stats(dMap, labels, numObjects, negate=negate)
    return numObjects, labels, stats


def getObjectPositions(mapData, threshold, findCenterOfMass=True):
    """Segmentation map and object positions above ``threshold`` (nemo/photometry.py:213-248).

    Returns (objIDs, objPositions, objNumPix, segmentationMap) - ``objIDs`` includes 0 (background) exactly as
    ``np.unique`` of the reference's segmentation map does when any background exists - or None if no pixel is
    significant.  A device tensor in gives a device segmentation map out.
    """
    if threshold < 0:
        raise Exception("Detection threshold (thresholdSigma in the config file) cannot be negative unless in forced photometry mode.")
    isDev = hasattr(mapData, 'is_cuda')
    d = engine.to_dev(mapData)
    eng = engine.TileEngine.get(d.shape)
    numObjects, labels, stats = _segment(eng, d, threshold, findCenterOfMass)
    if numObjects == 0:
        return None
    st = stats.cpu().numpy()
    counts = st[:, 0]
    hasBackground = int(counts[1:].sum()) < d.numel()
    first = 0 if hasBackground else 1
    objIDs = np.arange(first, numObjects + 1)
    W = d.shape[1]
    positions = []
    for i in objIDs:
        if i == 0:
            # background: centre of mass over label-0 pixels is never used (numPix = 0 < minObjPix); reported as NaN
            positions.append((np.nan, np.nan))
        elif findCenterOfMass:
            positions.append((st[i, 2] / st[i, 1], st[i, 3] / st[i, 1]))
        else:
            idx = int(st[i, 5])
            positions.append((idx // W, idx % W))
    numPix = counts[first:].copy()
    if first == 0:
        numPix[0] = 0.0
    seg = labels if isDev else labels.cpu().numpy()
    return objIDs, positions, numPix, seg


def _lookupDev(dMap, iy, ix):
    """dMap[iy, ix] as a float64 device tensor (not fetched)."""
    t = engine.torch()
    return dMap[t.as_tensor(iy, device=dMap.device), t.as_tensor(ix, device=dMap.device)].to(t.float64)


def _numpyDtype(tensor):
    return engine.torch().empty(0, dtype=tensor.dtype).numpy().dtype


def _fetchTogether(queued):
    """One device-to-host transfer for several small 1-D device results: {key: float64 numpy array}."""
    t = engine.torch()
    if len(queued) == 0:
        return {}
    parts = [v.reshape(-1).to(t.float64) for k, v in queued]
    flat = (t.cat(parts) if len(parts) > 1 else parts[0]).cpu().numpy()
    out, pos = {}, 0
    for (k, v), p in zip(queued, parts):
        out[k] = flat[pos:pos + p.numel()]
        pos += p.numel()
    return out


def _lookup(mapLike, iy, ix):
    """mapLike[iy, ix] for a numpy array or a device tensor; returns numpy."""
    if hasattr(mapLike, 'is_cuda'):
        t = engine.torch()
        return mapLike[t.as_tensor(iy, device=mapLike.device), t.as_tensor(ix, device=mapLike.device)].cpu().numpy()
    return np.asarray(mapLike)[iy, ix]


def findObjects(filteredMapDict, threshold=3.0, minObjPix=3, rejectBorder=10, findCenterOfMass=True, flagRings=True,
                ringThresholdSigma=0, ringFlagDistArcmin=12.0, ringFlagMinRingerSNR=20, invertMap=False,
                objIdent='ACT-CL', longNames=False, verbose=True, useInterpolator=True, measureShapes=False,
                DS9RegionsPath=None, _prefetch=None):
    """Finds objects in the S/N map of ``filteredMapDict`` (nemo/photometry.py:25-210).  Returns a
    :class:`nemo_amd.catalogs.Table` (or [] if nothing is found, as the reference does).

    Everything that is read off device maps at the object positions (S/N, ring labels, flags) is enqueued first and
    fetched in ONE transfer.  ``_prefetch`` (used by pipelines._detectAndMeasure): a list of (mapData, negate) whose
    values at the objects' sky positions :func:`measureFluxes` / :func:`getSNRValues` will ask for next - they join the
    same transfer and wait on the returned table, so that those calls do not go back to the device."""
    return findObjectsGroup([filteredMapDict], threshold=threshold, minObjPix=minObjPix, rejectBorder=rejectBorder,
                            findCenterOfMass=findCenterOfMass, flagRings=flagRings, ringThresholdSigma=ringThresholdSigma,
                            ringFlagDistArcmin=ringFlagDistArcmin, ringFlagMinRingerSNR=ringFlagMinRingerSNR, invertMap=invertMap,
                            objIdent=objIdent, longNames=longNames, verbose=verbose, useInterpolator=useInterpolator,
                            measureShapes=measureShapes, DS9RegionsPath=DS9RegionsPath, _prefetch=[_prefetch])[0]


def findObjectsGroup(filteredMapDictList, _prefetch=None, **kwargs):
    """findObjects for several filtered maps at once (the scales of a tile): the detections of all maps are enqueued before
    anything is fetched, and every stage's small results (label records; values read at the object positions) come back in
    ONE transfer for all maps - two host round trips per group instead of two per map.  Same catalogues as findObjects map
    by map.  ``_prefetch``: one list per map (see findObjects)."""
    if _prefetch is None:
        _prefetch = [None] * len(filteredMapDictList)
    if _VECTOR_GROUP and len(filteredMapDictList) > 1:
        res = _findObjectsVector(filteredMapDictList, _prefetch, **kwargs)
        if res is not None:
            return res
    gens = [_findObjectsSteps(d, _prefetch=pf, _slot=k, **kwargs) for k, (d, pf) in enumerate(zip(filteredMapDictList, _prefetch))]
    results = [None] * len(gens)
    waiting = {}
    for i, g in enumerate(gens):
        try:
            waiting[i] = next(g)
        except StopIteration as e:
            results[i] = e.value
    while waiting:
        got = _fetchTogether([((i, k), v) for i, req in waiting.items() for k, v in req])
        nxt = {}
        for i, req in waiting.items():
            try:
                nxt[i] = gens[i].send({k: got[(i, k)] for k, _ in req})
            except StopIteration as e:
                results[i] = e.value
        waiting = nxt
    return results


_VECTOR_GROUP = os.environ.get('NEMO_AMD_VECTOR_DETECT', '1') != '0'      # 0: every map of a group through _findObjectsSteps
_LABEL_BATCH = os.environ.get('NEMO_AMD_LABEL_BATCH', '1') != '0'         # 0: the maps of a group labelled one launch sequence each


def _findObjectsVector(dicts, prefetch, threshold=3.0, minObjPix=3, rejectBorder=10, findCenterOfMass=True, flagRings=True,
                       ringThresholdSigma=0, ringFlagDistArcmin=12.0, ringFlagMinRingerSNR=20, invertMap=False,
                       objIdent='ACT-CL', longNames=False, verbose=True, useInterpolator=True, measureShapes=False,
                       DS9RegionsPath=None):
    """findObjects for the maps of a group that share their tile (shape, WCS, flag mask) - the scales of a tile - with the
    host side written once for all of them: one numpy statement per step over the concatenated object lists instead of one
    per map, one upload of all positions, one spline batch per 32 reads, one transfer of all values.  The same arithmetic per
    object as _findObjectsSteps (same catalogues, row for row); returns None when the group is not of that kind (the
    caller then walks the maps one by one)."""
    if threshold < 0 or flagRings == True or measureShapes == True or useInterpolator != True:
        return None
    first = dicts[0]
    wcs, flagMask = first['wcs'], first['flagMask']
    shape = tuple(first['SNMap'].shape)
    for d in dicts:
        if d['wcs'] is not wcs or d['flagMask'] is not flagMask or tuple(d['SNMap'].shape) != shape or not hasattr(d['SNMap'], 'is_cuda'):
            return None
    import time
    tm = time.perf_counter() if _TIMING else 0.0
    t = engine.torch()
    M = len(dicts)
    H, W = shape
    eng = engine.TileEngine.get(shape)
    cap = eng.statsCap
    dSNs, chunks = [], []
    for d in dicts:
        dSN = d['SNMap']
        hint = d.get('_snChunks', None) if hasattr(d, 'get') else None
        chunks.append(hint[1] if (hint is not None and hint[0] is dSN and hint[0].data_ptr() == dSN.data_ptr()) else None)
        dSNs.append(dSN)
    if len({m.dtype for m in dSNs}) != 1:
        return None
    # labelling and per-label sums of all the maps: every pass one launch (:59)
    if _LABEL_BATCH:
        labels, recs = eng.label_and_stats_batch(dSNs, threshold, negate=invertMap, slot='objects', cap=cap, snChunks=chunks)
    else:
        recs = t.stack([eng.label_and_stats(m, threshold, negate=invertMap, slot='objects', cap=cap, snChunks=c)[1]
                        for m, c in zip(dSNs, chunks)])
    tm = _mark('label+stats enqueue', tm)
    flagAny = first.get('_flagAny', None) if hasattr(first, 'get') else None
    if flagAny is not None and flagAny[0] is not flagMask:
        flagAny = None
    flat = recs.reshape(-1)
    if flagAny is not None:
        flat = t.cat([flat, flagAny[1].reshape(-1).to(t.float64)])
    flat = flat.cpu().numpy()
    tm = _mark('stats fetch', tm)
    recLen = (cap + 1) * 6
    S = flat[:M * recLen].reshape(M, cap + 1, 6)
    noFlags = flagAny is not None and flat[M * recLen] == 0.0
    numObjects = S[:, 0, 0].astype(np.int64)
    if (numObjects > cap).any():
        eng.statsCap = 2 * int(numObjects.max())     # more labels than the record table held: the one-by-one path runs again with room
        return None
    results = [[] for _ in range(M)]
    # objects passing the size cut, map after map, in label order
    top = int(numObjects.max())                       # (records beyond the largest label count of the group hold nothing)
    valid = (np.arange(1, top + 1)[None, :] <= numObjects[:, None]) & (S[:, 1:top + 1, 0] >= minObjPix)
    mi, li = np.nonzero(valid)
    if len(mi) == 0:
        return results
    rows = S[mi, li + 1]
    if findCenterOfMass:
        ys = rows[:, 2] / rows[:, 1]
        xs = rows[:, 3] / rows[:, 1]
    else:
        idx = rows[:, 5].astype(np.int64)
        ys = (idx // W).astype(np.float64)
        xs = (idx % W).astype(np.float64)
    iyr, ixr = np.round(ys).astype(np.int64), np.round(xs).astype(np.int64)
    counts = np.bincount(mi, minlength=M)
    offs = np.concatenate([[0], np.cumsum(counts)])
    tm = _mark('records', tm)
    RADeg, decDeg = wcs.pix2wcs(xs, ys)                                           # :115-117
    RADeg = np.where(RADeg < 0, 360.0 + RADeg, RADeg)
    havePre = any(bool(pf) for pf in prefetch)
    # every position in one upload: the detections' own (S/N read) and - for the prefetch reads - the positions measureFluxes /
    # getSNRValues will compute from the catalogue's coordinates, clamped to the map as FITPACK clamps them
    if havePre:
        x2, y2 = wcs.wcs2pix(np.asarray(RADeg, dtype=np.float64), np.asarray(decDeg, dtype=np.float64))
        x2, y2 = np.atleast_1d(x2), np.atleast_1d(y2)
        inside = (x2.astype(np.int64) > 0) & (x2.astype(np.int64) < W) & (y2.astype(np.int64) > 0) & (y2.astype(np.int64) < H)
        yx = np.concatenate([np.stack([ys, xs], axis=1), np.stack([np.clip(y2, 0.0, H - 1.0), np.clip(x2, 0.0, W - 1.0)], axis=1)])
    else:
        yx = np.stack([ys, xs], axis=1)
    n = len(ys)
    tm = _mark('positions', tm)
    yxDev = engine.to_dev(np.ascontiguousarray(yx).reshape(-1))       # (the jobs below name its parts by address)
    yxBase = yxDev.data_ptr()
    tm = _mark('upload', tm)
    jobs, keys = [], []           # (map, points of the map's own objects, negate)
    for k in range(M):
        a, b = int(offs[k]), int(offs[k + 1])
        if b == a:
            continue
        ptsSN = engine._SplinePoints(yxBase + 16 * a, b - a, 0, 32, None, -1)
        jobs.append((dSNs[k], ptsSN, invertMap))
        keys.append((k, 'SNR'))
        if prefetch[k]:
            pts = engine._SplinePoints(yxBase + 16 * (n + a), b - a, 0, 32, None, -1)
            for j, (mapLike, neg) in enumerate(prefetch[k]):
                jobs.append((engine.to_dev(mapLike), pts, neg))
                keys.append((k, j))
    tm = _mark('jobs', tm)
    outs, flatOut = eng.spline_eval_batch(jobs, flatOnly=True)
    queued = flatOut
    if not noFlags and hasattr(flagMask, 'is_cuda'):
        queued = t.cat([flatOut, _lookupDev(flagMask, iyr, ixr)])
    tm = _mark('enqueue reads', tm)
    vals = queued.cpu().numpy()
    tm = _mark('fetch reads', tm)
    flagsOnDevice = hasattr(flagMask, 'is_cuda')
    if flagsOnDevice and noFlags:
        flagsAll = np.zeros(n, dtype=_numpyDtype(flagMask))
    elif flagsOnDevice:
        flagsAll = vals[len(vals) - n:].astype(_numpyDtype(flagMask))
    else:
        flagsAll = np.asarray(flagMask)[iyr, ixr]
    got, pos = {}, 0
    for key, (m_, pts_, neg_) in zip(keys, jobs):
        got[key] = vals[pos:pos + pts_.n]
        pos += pts_.n
    SNRs = np.concatenate([got[(k, 'SNR')] for k in range(M) if counts[k] > 0])
    keep = SNRs > threshold                                                       # :186
    # the rows that stay, compressed once for all maps; a map's table then takes slices
    keptPerMap = np.bincount(mi[keep], minlength=M)
    offsK = np.concatenate([[0], np.cumsum(keptPerMap)])
    RAk, deck, SNRk = RADeg[keep], decDeg[keep], SNRs[keep]
    numSigPixk = rows[:, 0][keep]
    flagsk = flagsAll[keep]
    galk = galacticLatDeg(RAk, deck)
    xk, yk = xs[keep], ys[keep]
    insidek = inside[keep] if havePre else None
    ringAll = np.zeros(len(RAk), dtype=bool)
    flagsk = flagsk + ringAll                                                      # :208 (flags += ringFlag: no rings here)
    templatek = np.repeat(np.array([d['label'] for d in dicts], dtype='U%d' % max([30] + [len(d['label']) for d in dicts])), keptPerMap)
    tileNamek = np.repeat(np.array([d['tileName'] for d in dicts], dtype='U%d' % max([30] + [len(d['tileName']) for d in dicts])),
                          keptPerMap)
    for k in range(M):
        nk = int(keptPerMap[k])
        if nk == 0:
            continue
        a, b = int(offs[k]), int(offs[k + 1])
        a2, b2 = int(offsK[k]), int(offsK[k + 1])
        d = dicts[k]
        tab = catalogs.Table()
        tab.setLazyNames(objIdent, longNames)                                      # :120-123 (built on first access)
        c = tab._cols
        c['RADeg'] = RAk[a2:b2]
        c['decDeg'] = deck[a2:b2]
        c['SNR'] = SNRk[a2:b2]
        c['numSigPix'] = numSigPixk[a2:b2]
        c['template'] = templatek[a2:b2]
        c['tileName'] = tileNamek[a2:b2]
        c['flags'] = flagsk[a2:b2]
        c['ringFlag'] = ringAll[a2:b2]
        c['galacticLatDeg'] = galk[a2:b2]
        c['x'] = xk[a2:b2]
        c['y'] = yk[a2:b2]
        # column order as findObjects leaves it: flags is set again last there, which keeps its place
        if prefetch[k]:
            kp = keep[a:b]
            tab._prefetched = {}
            for j, (mapLike, neg) in enumerate(prefetch[k]):
                tab._prefetched[(id(mapLike), bool(neg))] = (got[(k, j)][kp], insidek[a2:b2], nk)
        results[k] = tab
    tm = _mark('table', tm)
    return results


def _findObjectsSteps(filteredMapDict, threshold=3.0, minObjPix=3, rejectBorder=10, findCenterOfMass=True, flagRings=True,
                      ringThresholdSigma=0, ringFlagDistArcmin=12.0, ringFlagMinRingerSNR=20, invertMap=False,
                      objIdent='ACT-CL', longNames=False, verbose=True, useInterpolator=True, measureShapes=False,
                      DS9RegionsPath=None, _prefetch=None, _slot=None):
    """findObjects of one map as a generator: it enqueues device work, yields the list of (key, device tensor) it wants on
    the host, is sent {key: float64 numpy array}, and returns the catalogue (findObjectsGroup drives several of these in
    step, one transfer per stage for all of them)."""
    if threshold < 0:
        raise Exception("Detection threshold (thresholdSigma in the config file) cannot be negative unless in forced photometry mode.")
    wcs = filteredMapDict['wcs']
    flagMask = filteredMapDict['flagMask']
    dSN = engine.to_dev(filteredMapDict['SNMap'])
    eng = engine.TileEngine.get(dSN.shape)
    H, W = dSN.shape

    import time
    tm = time.perf_counter() if _TIMING else 0.0
    # labelling and per-label sums go out as one enqueue (:59); the records and the object count come back together
    slot = ('objects', _slot) if (_slot is not None and measureShapes == True) else 'objects'
    hint = filteredMapDict.get('_snChunks', None) if hasattr(filteredMapDict, 'get') else None
    # (the table belongs to one S/N tensor: a map that was replaced since - useCachedRMSMap - has none)
    snChunks = hint[1] if (hint is not None and hint[0] is filteredMapDict['SNMap'] and hint[0].data_ptr() == dSN.data_ptr()) else None
    while True:
        cap = eng.statsCap
        labels, recs = eng.label_and_stats(dSN, threshold, negate=invertMap, slot=slot, cap=cap, snChunks=snChunks)
        tm = _mark('label+stats enqueue', tm)
        flagAny = filteredMapDict.get('_flagAny', None) if hasattr(filteredMapDict, 'get') else None
        if flagAny is not None and flagAny[0] is not flagMask:
            flagAny = None
        got = yield ([('stats', recs)] + ([('flagAny', flagAny[1])] if flagAny is not None else []))
        noFlags = flagAny is not None and got['flagAny'][0] == 0.0
        st = got['stats'].reshape(cap + 1, 6)
        numObjects = int(st[0, 0])
        if numObjects <= cap:
            break
        eng.statsCap = 2 * numObjects          # more labels than the record table held: once more, with room
    tm = _mark('stats fetch', tm)
    if numObjects == 0:
        return []
    st = st[:numObjects + 1].copy()
    st[0, 0] = 0.0

    ringLabels = isRing = None
    if flagRings == True:                                                         # :65-75
        numRing, ringLabels, ringStats = _segment(eng, dSN, ringThresholdSigma, True, negate=invertMap, slot='rings')
        isRing = eng.ring_flags(ringLabels, ringStats, numRing, minRingPix=30)
    tm = _mark('ring label', tm)

    # objects passing the size cut, in label order (label 0 has numPix 0 and never passes unless minObjPix <= 0)
    ids = np.arange(1, numObjects + 1)
    sel = ids[st[1:, 0] >= minObjPix]
    if len(sel) == 0:
        return []
    if findCenterOfMass:
        ys = st[sel, 2] / st[sel, 1]
        xs = st[sel, 3] / st[sel, 1]
    else:
        idx = st[sel, 5].astype(np.int64)
        ys = (idx // W).astype(np.float64)
        xs = (idx % W).astype(np.float64)

    iyr, ixr = np.round(ys).astype(np.int64), np.round(xs).astype(np.int64)      # int(round()): half to even, as np.round
    t = engine.torch()
    queued = []        # device results wanted on the host: (key, 1-D tensor), fetched together below
    splineJobs = []    # (key, map, points, negate): every spline read of this pass goes out as one batch
    if useInterpolator == True:                                                   # :81-84,128
        pass        # (the positions go up below, in one transfer with those of the prefetch reads)
    else:
        queued.append(('SNR', _lookupDev(dSN, iyr, ixr) * (-1.0 if invertMap else 1.0)))
    if ringLabels is not None:                                                    # :112-114
        iy = t.as_tensor(ys.astype(np.int64), device=dSN.device)                  # int() truncation
        ix = t.as_tensor(xs.astype(np.int64), device=dSN.device)
        lab = ringLabels[iy, ix].to(t.int64)
        queued.append(('ring', isRing[lab]))
    flagsOnDevice = hasattr(flagMask, 'is_cuda')
    if flagsOnDevice and not noFlags:
        queued.append(('flags', _lookupDev(flagMask, iyr, ixr)))
    if measureShapes == True:                                                     # :133-150 (the sums; the ellipse below)
        shapes = eng.object_shapes(dSN, labels, numObjects, negate=invertMap)
        queued.append(('shapes', shapes[t.as_tensor(sel, device=dSN.device)]))

    RADeg, decDeg = wcs.pix2wcs(xs, ys)                                           # :115-117
    RADeg = np.where(RADeg < 0, 360.0 + RADeg, RADeg)
    pre = []
    if _prefetch:
        # the positions measureFluxes / getSNRValues will compute from the catalogue's coordinates (_splineAt), every row:
        # clamped to the map as FITPACK clamps them; ``inside`` is getSNRValues' bounds test (:277)
        x2, y2 = wcs.wcs2pix(np.asarray(RADeg, dtype=np.float64), np.asarray(decDeg, dtype=np.float64))
        x2, y2 = np.atleast_1d(x2), np.atleast_1d(y2)
        inside = (x2.astype(np.int64) > 0) & (x2.astype(np.int64) < W) & (y2.astype(np.int64) > 0) & (y2.astype(np.int64) < H)
        yy = np.clip(y2, 0.0, H - 1.0)
        xx = np.clip(x2, 0.0, W - 1.0)
        pts = None
        if useInterpolator:
            ptsSN, pts = eng.spline_points_multi([np.stack([ys, xs], axis=1), np.stack([yy, xx], axis=1)])   # one upload
            splineJobs.append(('SNR', dSN, ptsSN, invertMap))
        for k, (mapLike, neg) in enumerate(_prefetch):
            d = engine.to_dev(mapLike)
            if useInterpolator:
                splineJobs.append(('pre%d' % k, d, pts, neg))
            else:
                iyn, ixn = np.round(y2).astype(np.int64), np.round(x2).astype(np.int64)
                if ((iyn >= H) | (ixn >= W) | (iyn < -H) | (ixn < -W)).any():
                    pre = None           # (measureFluxes raises for such a row, as numpy indexing does: let it find out)
                    break
                queued.append(('pre%d' % k, _lookupDev(d, iyn % H, ixn % W) * (-1.0 if neg else 1.0)))
        if pre is not None:
            pre = [(mapLike, neg) for mapLike, neg in _prefetch]
    elif useInterpolator == True:
        splineJobs.append(('SNR', dSN, eng.spline_points(np.stack([ys, xs], axis=1)), invertMap))
    if splineJobs:
        for (key, _, _, _), out in zip(splineJobs, eng.spline_eval_batch([(m, p, n) for _, m, p, n in splineJobs])):
            queued.append((key, out))
    tm = _mark('positions + enqueue reads', tm)
    got = yield queued
    tm = _mark('fetch reads', tm)
    SNRs = got['SNR']
    ringFlags = got['ring'] > 0 if 'ring' in got else np.zeros(len(sel), dtype=bool)
    keep = SNRs > threshold                                                       # :186
    if keep.sum() == 0:
        return []
    if flagsOnDevice and noFlags:
        flags = np.zeros(len(sel), dtype=_numpyDtype(flagMask))         # (the tile has no flagged pixel at all)
    else:
        flags = got['flags'].astype(_numpyDtype(flagMask)) if flagsOnDevice else np.asarray(flagMask)[iyr, ixr]
    tab = catalogs.Table()
    tab.setLazyNames(objIdent, longNames)                                          # :120-123 (built on first access)
    tab['RADeg'] = RADeg[keep]
    tab['decDeg'] = decDeg[keep]
    tab['SNR'] = SNRs[keep]
    tab['numSigPix'] = st[sel, 0][keep]
    tab['template'] = np.full(int(keep.sum()), filteredMapDict['label'], dtype='U%d' % max(30, len(filteredMapDict['label'])))
    tab['tileName'] = np.full(int(keep.sum()), filteredMapDict['tileName'], dtype='U%d' % max(30, len(filteredMapDict['tileName'])))
    tab['flags'] = np.asarray(flags)[keep]
    tab['ringFlag'] = ringFlags[keep]
    tab['galacticLatDeg'] = galacticLatDeg(tab['RADeg'], tab['decDeg'])
    tab['x'] = xs[keep]
    tab['y'] = ys[keep]
    if measureShapes == True:
        ell = _ellipses(got['shapes'].reshape(-1, 10)[keep], np.asarray(tab['numSigPix']), single=dSN.dtype == t.float32)
        for k in _ELLIPSE_KEYS:
            tab[k] = ell[k]

    # Ring flags only stand near a high S/N object (:196-204); then flags += ringFlag (:208)
    if tab['ringFlag'].any():
        high = tab['SNR'] > ringFlagMinRingerSNR
        hRA, hDec = tab['RADeg'][high], tab['decDeg'][high]
        rf = tab['ringFlag']
        for i in np.where(rf)[0]:
            rDeg = calcAngSepDeg(tab['RADeg'][i], tab['decDeg'][i], hRA, hDec)
            if np.logical_and((rDeg * 60) < ringFlagDistArcmin, rDeg > 0).sum() == 0:
                rf[i] = False
    tab['flags'] = tab['flags'] + tab['ringFlag']
    tm = _mark('table', tm)
    if pre:
        tab._prefetched = {}
        for k, (mapLike, neg) in enumerate(pre):
            if 'pre%d' % k in got:
                tab._prefetched[(id(mapLike), bool(neg))] = (got['pre%d' % k][keep], inside[keep], len(tab))
    return tab


_ELLIPSE_KEYS = ('ellipse_PA', 'ellipse_A', 'ellipse_B', 'ellipse_x0', 'ellipse_y0', 'ellipse_e')


def _ellipses(sh, numSigPix, single=False):
    """SExtractor-style ellipses from the segment moments (nemo/photometry.py:135-184), all objects at once.  ``sh`` [n, 10] =
    {xMin, yMin, xMax, yMax, S, Sx, Sy, Sxx, Syy, Sxy}.  ``single``: the S/N map is float32 - the reference's ``data[mask].sum()``
    is then a float32 number, and so is the divisor here."""
    n = len(sh)
    out = {k: np.full(n, -99.0) for k in _ELLIPSE_KEYS}
    if n == 0:
        return out
    S = sh[:, 4].astype(np.float32).astype(np.float64) if single else sh[:, 4]
    with np.errstate(divide='ignore', invalid='ignore'):
        cx2, cy2 = sh[:, 5] / S, sh[:, 6] / S                                     # :145-146
        x2 = sh[:, 7] / S - cx2 ** 2                                              # :148-150
        y2 = sh[:, 8] / S - cy2 ** 2
        xy = sh[:, 9] / S - cx2 * cy2
        theta = np.degrees(np.arctan(2 * (xy / (x2 - y2))) / 2.0)                 # :152-156
        theta = np.where((xy > 0) & (theta < 0), theta + 90, np.where((xy < 0) & (theta > 0), theta - 90, theta))
        ok = (numSigPix > 9) & (((theta > 0) & (xy > 0)) | ((theta < 0) & (xy < 0)))        # :135,157-160
        root = np.sqrt(((x2 - y2) / 2) ** 2 + xy ** 2)
        A = np.sqrt((x2 + y2) / 2.0 + root)                                       # :162-163
        B = np.sqrt((x2 + y2) / 2.0 - root)
        scaleFactor = np.sqrt(numSigPix / (A * B * np.pi))                        # :166-170 (segment area = its pixel count)
        A, B = A * scaleFactor, B * scaleFactor
        ecc = np.sqrt(1 - B ** 2 / A ** 2)
    for k, v in zip(_ELLIPSE_KEYS, (theta, A, B, cx2 + sh[:, 0], cy2 + sh[:, 1], ecc)):
        out[k][ok] = v[ok]
    return out


def _valuesAt(catalog, mapData, wcs, useInterpolator, negate=False, allRows=False):
    """_splineAt at the catalogue's positions, or what findObjects already fetched for this map (``_prefetch``)."""
    pre = getattr(catalog, '_prefetched', None)
    if pre is not None:
        hit = pre.get((id(mapData), bool(negate)), None)
        if hit is not None and hit[2] == len(catalog):
            vals = hit[0].copy()
            if not allRows:
                vals[~hit[1]] = 0.0
            return vals, hit[1]
    return _splineAt(mapData, wcs, catalog['RADeg'], catalog['decDeg'], useInterpolator, negate=negate, allRows=allRows)


def _splineAt(mapData, wcs, RADeg, decDeg, useInterpolator, negate=False, allRows=False):
    """Map values at sky positions: RectBivariateSpline(y, x) or nearest pixel (nemo/photometry.py:276-281,354-361).
    Returns (values, inside) where ``inside`` follows getSNRValues' int(x) > 0 ... bounds test (:277); rows outside get 0
    unless ``allRows`` (measureFluxes evaluates every row, :353-361: the spline clamps positions to the map, the nearest
    pixel lookup takes Python's negative indices)."""
    d = engine.to_dev(mapData)
    eng = engine.TileEngine.get(d.shape)
    H, W = d.shape
    x, y = wcs.wcs2pix(np.asarray(RADeg, dtype=np.float64), np.asarray(decDeg, dtype=np.float64))
    x, y = np.atleast_1d(x), np.atleast_1d(y)
    inside = (x.astype(np.int64) > 0) & (x.astype(np.int64) < W) & (y.astype(np.int64) > 0) & (y.astype(np.int64) < H)
    vals = np.zeros(len(x), dtype=np.float64)
    use = np.ones(len(x), dtype=bool) if allRows else inside
    if use.sum() == 0:
        return vals, inside
    if useInterpolator:
        # the spline is only defined on [0, H-1] x [0, W-1]; FITPACK clamps outside
        yy = np.clip(y[use], 0.0, H - 1.0)
        xx = np.clip(x[use], 0.0, W - 1.0)
        vals[use] = eng.spline_eval(d, np.stack([yy, xx], axis=1), negate=negate)
    else:
        iy, ix = np.round(y[use]).astype(np.int64), np.round(x[use]).astype(np.int64)
        if allRows:
            if (iy >= H).any() or (ix >= W).any() or (iy < -H).any() or (ix < -W).any():
                raise IndexError("measureFluxes: catalogue position outside the map (useInterpolator False)")
            iy, ix = iy % H, ix % W
        else:
            iy, ix = np.clip(iy, 0, H - 1), np.clip(ix, 0, W - 1)
        vals[use] = _lookup(d, iy, ix).astype(np.float64) * (-1.0 if negate else 1.0)
    return vals, inside


def getSNRValues(catalog, SNMap, wcs, useInterpolator=True, invertMap=False, prefix=''):
    """S/N at catalogue positions, added in place as ``prefix+'SNR'`` (nemo/photometry.py:251-281)."""
    if len(catalog) == 0:
        return
    vals, inside = _valuesAt(catalog, SNMap, wcs, useInterpolator, negate=invertMap)
    col = np.zeros(len(catalog))
    col[inside] = vals[inside]
    catalog[prefix + 'SNR'] = col


def deltaTToJyPerSr(temp, obsFreqGHz):
    """nemo/photometry.py:488-502 (Planck function derivative at T_CMB; CGS constants)."""
    from .signals import TCMB, _h_planck, _k_B, CLight
    kB = _k_B * 1e7          # erg / K
    h = _h_planck * 1e7      # erg s
    c = CLight * 100.0       # cm / s
    nu = obsFreqGHz * 1.e9
    T0 = TCMB
    x = h * nu / (kB * T0)
    cNu = 2 * (kB * T0) ** 3 / (h ** 2 * c ** 2) * x ** 4 / (4 * (np.sinh(x / 2.)) ** 2)
    cNu *= 1e23
    return temp * cNu * 1e-6 / T0


def measureFluxes(catalog, filteredMapDict, diagnosticsDir, photFilteredMapDict=None, useInterpolator=True,
                  ycObsFreqGHz=148.0):
    """Adds flux columns to ``catalog`` in place (nemo/photometry.py:284-377): y_c / deltaT_c (yc maps) or
    deltaT_c / fluxJy (uK maps), plus ``fixed_`` versions from the reference-scale maps."""
    if not hasattr(catalog, 'keys') or len(catalog) == 0:
        return
    mapData = filteredMapDict['data']
    wcs = filteredMapDict['wcs']
    mapUnits = wcs.header['BUNIT']
    if photFilteredMapDict is not None:
        getSNRValues(catalog, photFilteredMapDict['SNMap'], wcs, prefix='fixed_', useInterpolator=useInterpolator)
    reportJyFluxes = 'BEAMNSR' in wcs.header.keys()
    if reportJyFluxes:
        beamSolidAngle_nsr = wcs.header['BEAMNSR']
        obsFreqGHz = wcs.header['FREQGHZ']
    mapDataList, prefixList = [mapData], ['']
    if photFilteredMapDict is not None:
        mapDataList.append(photFilteredMapDict['data'])
        prefixList.append('fixed_')
    keysToAdd = ['deltaT_c', 'err_deltaT_c']
    if mapUnits == 'yc':
        keysToAdd = keysToAdd + ['y_c', 'err_y_c']
    elif mapUnits == 'uK':
        keysToAdd = keysToAdd + ['fluxJy', 'err_fluxJy']
    n = len(catalog)
    for prefix in prefixList:
        for k in keysToAdd:
            catalog[prefix + k] = np.zeros(n)
    for data, prefix in zip(mapDataList, prefixList):
        mapValue, inside = _valuesAt(catalog, data, wcs, useInterpolator, allRows=True)
        with np.errstate(divide='ignore', invalid='ignore'):
            if mapUnits == 'yc':
                catalog[prefix + 'y_c'] = mapValue / 1e-4
                catalog[prefix + 'err_y_c'] = catalog[prefix + 'y_c'] / catalog[prefix + 'SNR']
                deltaTc = convertToDeltaT(mapValue, obsFrequencyGHz=ycObsFreqGHz)
                catalog[prefix + 'deltaT_c'] = deltaTc
                catalog[prefix + 'err_deltaT_c'] = np.abs(deltaTc / catalog[prefix + 'SNR'])
            elif mapUnits == 'uK':
                catalog[prefix + 'deltaT_c'] = mapValue
                catalog[prefix + 'err_deltaT_c'] = mapValue / catalog[prefix + 'SNR']
                if reportJyFluxes:
                    catalog[prefix + 'fluxJy'] = deltaTToJyPerSr(catalog[prefix + 'deltaT_c'], obsFreqGHz) * beamSolidAngle_nsr * 1.e-9
                    catalog[prefix + 'err_fluxJy'] = deltaTToJyPerSr(catalog[prefix + 'err_deltaT_c'], obsFreqGHz) * beamSolidAngle_nsr * 1.e-9


def measureFluxesGroup(catalogList, filteredMapDictList, diagnosticsDir, photFilteredMapDictList, useInterpolator=True,
                       ycObsFreqGHz=148.0):
    """measureFluxes for the catalogues of a group of yc maps (the scales of a tile) with the arithmetic written once over the
    concatenated rows - the same statements per row as measureFluxes, the same columns in the same order; anything else (other
    units, mixed groups, tables without the values findObjectsGroup fetched ahead) goes map by map."""
    live = [k for k, c in enumerate(catalogList) if hasattr(c, 'keys') and len(c) > 0]
    vector = _VECTOR_GROUP and len(live) > 1
    if vector:
        withPhot = photFilteredMapDictList[live[0]] is not None
        for k in live:
            d, ph, c = filteredMapDictList[k], photFilteredMapDictList[k], catalogList[k]
            pre = getattr(c, '_prefetched', None)
            if d['wcs'].header['BUNIT'] != 'yc' or (ph is not None) != withPhot or pre is None or 'SNR' not in c._cols:
                vector = False
                break
            need = [(id(d['data']), False)] + ([(id(ph['SNMap']), False), (id(ph['data']), False)] if withPhot else [])
            if any(pre.get(q) is None or pre[q][2] != len(c) for q in need):
                vector = False
                break
    if not vector:
        for c, d, ph in zip(catalogList, filteredMapDictList, photFilteredMapDictList):
            measureFluxes(c, d, diagnosticsDir, photFilteredMapDict=ph, useInterpolator=useInterpolator, ycObsFreqGHz=ycObsFreqGHz)
        return
    cats = [catalogList[k] for k in live]
    offs = np.concatenate([[0], np.cumsum([len(c) for c in cats])])
    n = int(offs[-1])
    cat = lambda parts: np.concatenate(parts)           # noqa: E731
    out = {}            # column -> values of all rows, in measureFluxes' column order
    if withPhot:
        # getSNRValues (:251-281): 0 where the position falls outside the map
        hits = [c._prefetched[(id(photFilteredMapDictList[k]['SNMap']), False)] for k, c in zip(live, cats)]
        inside = cat([h[1] for h in hits])
        col = np.zeros(n)
        col[inside] = cat([h[0] for h in hits])[inside]
        out['fixed_SNR'] = col
    prefixes = [''] + (['fixed_'] if withPhot else [])
    for prefix in prefixes:
        for key in ('deltaT_c', 'err_deltaT_c', 'y_c', 'err_y_c'):
            out[prefix + key] = None
    SNRs = {'': cat([c._cols['SNR'] for c in cats])}
    if withPhot:
        SNRs['fixed_'] = out['fixed_SNR']
    for prefix in prefixes:
        maps = [(filteredMapDictList[k] if prefix == '' else photFilteredMapDictList[k])['data'] for k in live]
        mapValue = cat([c._prefetched[(id(m), False)][0] for c, m in zip(cats, maps)])
        with np.errstate(divide='ignore', invalid='ignore'):
            y = mapValue / 1e-4                                                    # :357-366
            out[prefix + 'y_c'] = y
            out[prefix + 'err_y_c'] = y / SNRs[prefix]
            deltaTc = convertToDeltaT(mapValue, obsFrequencyGHz=ycObsFreqGHz)
            out[prefix + 'deltaT_c'] = deltaTc
            out[prefix + 'err_deltaT_c'] = np.abs(deltaTc / SNRs[prefix])
    for i, c in enumerate(cats):
        a, b = int(offs[i]), int(offs[i + 1])
        cols = c._cols
        for key, v in out.items():
            cols[key] = v[a:b]


def makeForcedPhotometryCatalog(filteredMapDict, inputCatalog, useInterpolator=True, DS9RegionsPath=None):
    """Catalogue of S/N values at given positions (nemo/photometry.py:380-444).  ``inputCatalog``: a
    :class:`nemo_amd.catalogs.Table` (or dict of columns) with RADeg/decDeg (or ra/dec, RA/DEC) columns."""
    if isinstance(inputCatalog, str):
        # a catalogue file (nemo/photometry.py:402-403 reads it with astropy.table): FITS binary table or tab-delimited .csv,
        # as catalogs.writeCatalog writes them
        inputCatalog = catalogs.readCatalog(inputCatalog)
    tab = inputCatalog if isinstance(inputCatalog, catalogs.Table) else catalogs.Table(cols=dict(inputCatalog))
    RAKey, decKey = catalogs.getTableRADecKeys(tab)
    RA = np.array(tab[RAKey], dtype=np.float64)
    RA = np.where(RA < 0, 360 - np.abs(RA), RA)
    dec = np.array(tab[decKey], dtype=np.float64)
    names = np.array(tab['name']) if 'name' in tab.keys() else np.arange(len(tab)) + 1
    wcs = filteredMapDict['wcs']
    dSN = engine.to_dev(filteredMapDict['SNMap'])
    H, W = dSN.shape
    x, y = wcs.wcs2pix(RA, dec)
    x, y = np.atleast_1d(x), np.atleast_1d(y)
    inImage = (x >= 0) & (x < W - 1) & (y >= 0) & (y < H - 1)                     # catalogs.getCatalogWithinImage
    xi, yi = np.round(x[inImage]).astype(np.int64), np.round(y[inImage]).astype(np.int64)
    area = _lookup(filteredMapDict['surveyMask'], yi, xi) != 0
    xi, yi = xi[area], yi[area]
    if len(xi) == 0:
        return []
    eng = engine.TileEngine.get(dSN.shape)
    if useInterpolator:
        SNR = eng.spline_eval(dSN, np.stack([yi.astype(np.float64), xi.astype(np.float64)], axis=1))
    else:
        SNR = _lookup(dSN, yi, xi).astype(np.float64)
    out = catalogs.Table()
    out['name'] = names[inImage][area]
    out['RADeg'] = RA[inImage][area]
    out['decDeg'] = dec[inImage][area]
    out['SNR'] = SNR
    out['numSigPix'] = np.ones(len(xi))
    out['template'] = np.full(len(xi), filteredMapDict['label'], dtype='U30')
    out['tileName'] = np.full(len(xi), filteredMapDict['tileName'], dtype='U30')
    out['flags'] = _lookup(filteredMapDict['flagMask'], yi, xi)
    out['galacticLatDeg'] = galacticLatDeg(out['RADeg'], out['decDeg'])
    out['x'] = xi
    out['y'] = yi
    return out
