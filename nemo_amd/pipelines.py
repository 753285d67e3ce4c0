"""Tile x filter driver with the reference's call surface (nemo/pipelines.py:35-398).

``filterMapsAndMakeCatalogs(config, ...)`` loops over this rank's tiles and over the filters (photFilter first,
nemo/pipelines.py:161-173), calls :func:`nemo_amd.filters.filterMaps` and :func:`nemo_amd.photometry.findObjects`
exactly as the reference does, merges the per-(filter, tile) catalogues with ``catalogs.makeOptimalCatalog`` and - when
several GPUs share the tiles - all-gathers the merged catalogue so that every rank holds all of it (the reference's
``comm.allgather`` at nemo/pipelines.py:322), then removes the duplicates found in tile overlaps (:335).

What differs from the reference is where things live: per tile, the maps, weights, masks, the apodised data FFTs and
the smoothed noise covariance are kept in HBM and shared by all filter scales of that tile (the reference recomputes
them for each scale - they do not depend on the template), and the filtered / S/N maps stay on the device for source
finding.  Only catalogue rows reach the host.
"""
import concurrent.futures
import contextlib
import os
import threading
import time

import numpy as np

from . import catalogs, filters, photometry, engine, fitsio, maps

# columns exchanged between ranks (fixed-width float64 rows; strings are rebuilt from indices / coordinates)
_GATHER_COLS = ['RADeg', 'decDeg', 'SNR', 'numSigPix', 'flags', 'ringFlag', 'galacticLatDeg', 'x', 'y']


def initDistributed():
    """One process per GPU: bind this process to its GPU and join the job (RCCL via backend 'nccl' when GPUs are present,
    gloo otherwise).  Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as set by ``torch.distributed.run``."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1:
        return 0, 1
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    local = int(os.environ.get('LOCAL_RANK', '0'))
    backend = os.environ.get('NEMO_AMD_DIST_BACKEND', 'nccl')      # 'gloo' to rehearse several ranks on one GPU
    if torch.cuda.is_available() and backend == 'nccl':
        dev = engine.bind_device(local % torch.cuda.device_count())
        dist.init_process_group('nccl', device_id=dev)
    elif torch.cuda.is_available():
        engine.bind_device(local % torch.cuda.device_count())
        dist.init_process_group('gloo')
    else:
        dist.init_process_group('gloo')
    return dist.get_rank(), dist.get_world_size()


_NAME_CHARS = 30      # width of the name column (catalogs.catalogListToTab: minStrLength = 30)


def makeRMSTables(config, selFnDir=None):
    """Noise / area table of every tile for the reference filter (nemo/pipelines.py:401-503, the 'full' footprint): each rank
    tabulates its tiles on its GPU (completeness.getRMSTab), the rows travel to every rank in one all-gather of fixed-width
    rows (the reference gathers pickled tables on rank 0, :464-475), and the merged table - columns areaDeg2, y0RMS, tileName,
    sorted by y0RMS (:496-501) - is returned on every rank.  None without a photFilter (:409-410)."""
    from . import completeness
    if config.parDict['photFilter'] is None:
        return None
    photFilterLabel = config.parDict['photFilter']
    maxFlags = config.parDict.get('selFnOptions', {}).get('maxFlags', None)
    selFnDir = selFnDir if selFnDir is not None else config.selFnDir
    tileIndex = {name: i for i, name in enumerate(config.allTileNames)}
    rows = []
    for tileName in config.tileNames:
        tab = completeness.getRMSTab(tileName, photFilterLabel, selFnDir, maxFlags=maxFlags, config=config)
        part = np.zeros((len(tab), 3), dtype=np.float64)
        part[:, 0], part[:, 1], part[:, 2] = tab['areaDeg2'], tab['y0RMS'], tileIndex[tileName]
        rows.append(part)
    rows = np.concatenate(rows) if rows else np.zeros((0, 3))
    if config.MPIEnabled == True and config.size > 1:
        rows = _gatherRows(rows, config)
    out = catalogs.Table()
    order = np.argsort(rows[:, 1], kind='stable')
    strLen = max([len(name) for name in config.allTileNames] + [1])
    out['areaDeg2'] = rows[order, 0]
    out['y0RMS'] = rows[order, 1].astype(np.float32)
    out['tileName'] = np.array([config.allTileNames[int(i)] for i in rows[order, 2]], dtype='<U%d' % (strLen))
    from . import __version__
    out.meta['NEMOVER'] = __version__
    return out


def _gatherRows(rows, config):
    """All-gather of float64 rows of equal width (counts first, then the rows padded to the largest count); every rank gets the
    rows of all ranks in rank order."""
    import torch
    import torch.distributed as dist
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    n, width = rows.shape
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(config.size)]
    dist.all_gather(counts, torch.tensor([n], dtype=torch.int64, device=dev))
    counts = [int(c) for c in torch.cat(counts).cpu().tolist()]
    nmax = max(counts)
    if nmax == 0:
        return rows
    padded = torch.zeros((nmax, width), dtype=torch.float64, device=dev)
    if n > 0:
        padded[:n] = torch.from_numpy(np.ascontiguousarray(rows)).to(dev)
    parts = [torch.zeros_like(padded) for _ in range(config.size)]
    dist.all_gather(parts, padded)
    return np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)])


def _gatherCatalogs(optimalCatalog, config):
    """All ranks end up with the rows of all ranks: all-gather of the row counts, then of the rows padded to the largest
    count.  Rows are [len(_GATHER_COLS) + 2] float64: the listed columns, the index of the template in the filter list and
    the index of the tile in ``config.allTileNames``.  Extra (photometry) columns travel too when present on every rank,
    and so do the names (each rank formats those of its own rows; a second all-gather of fixed-width bytes)."""
    import torch
    import torch.distributed as dist
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    labels = [f['label'] for f in config.parDict['mapFilters']]
    extra = []
    if len(optimalCatalog) > 0:
        extra = [k for k in optimalCatalog.keys() if k not in _GATHER_COLS + ['name', 'template', 'tileName']
                 and np.asarray(optimalCatalog[k]).dtype.kind in 'fiub']
    # Row counts of all ranks - and, in the same exchange, what each rank offers beside the fixed columns: whether it carries
    # names and a digest of its extra column names.  When every rank has rows and all digests agree (the usual pass) the columns
    # are settled; otherwise the ranks exchange the lists themselves (a rank with no detections has none to offer: the columns
    # kept are those every non-empty rank has).
    n = len(optimalCatalog)
    import hashlib
    digest = int(hashlib.sha1('\0'.join(extra).encode()).hexdigest()[:15], 16)
    metas = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in range(config.size)]
    dist.all_gather(metas, torch.tensor([n, 1 if 'name' in optimalCatalog.keys() else 0, digest], dtype=torch.int64, device=dev))
    metas = torch.stack(metas).cpu().tolist()                             # one transfer, one synchronisation
    counts = [int(m[0]) for m in metas]
    if all(c > 0 for c in counts) and all(m[2] == metas[0][2] for m in metas):
        carryNames = all(m[1] == 1 for m in metas)
    else:
        offers = [None] * config.size
        dist.all_gather_object(offers, (extra, 'name' in optimalCatalog.keys()) if n > 0 else None)
        common, carryNames = None, True
        for o in offers:
            if o is not None:
                common = list(o[0]) if common is None else [k for k in common if k in o[0]]
                carryNames = carryNames and o[1]
        extra = common or []
    cols = _GATHER_COLS + extra
    rows = np.zeros((n, len(cols) + 2), dtype=np.float64)
    if n > 0:
        for j, k in enumerate(cols):
            rows[:, j] = np.asarray(optimalCatalog[k], dtype=np.float64) if k in optimalCatalog.keys() else -99
        labelIndex = {l: i for i, l in enumerate(labels)}
        tileIndex = {l: i for i, l in enumerate(config.allTileNames)}
        rows[:, len(cols)] = [labelIndex.get(t, -1) for t in np.asarray(optimalCatalog['template']).astype(str).tolist()]
        rows[:, len(cols) + 1] = [tileIndex[t] for t in np.asarray(optimalCatalog['tileName']).astype(str).tolist()]
    nmax = max(counts)
    if nmax == 0:
        return catalogs.Table(names=('SNR', 'RADeg', 'decDeg'))
    padded = torch.zeros((nmax, rows.shape[1]), dtype=torch.float64, device=dev)
    if n > 0:
        padded[:n] = torch.from_numpy(rows).to(dev)
    gathered = [torch.empty_like(padded) for _ in range(config.size)]
    dist.all_gather(gathered, padded)
    gatheredHost = torch.stack(gathered).cpu().numpy()                    # [rank][nmax][cols] in one transfer
    allRows = np.concatenate([gatheredHost[r, :c] for r, c in enumerate(counts)], axis=0)
    tab = catalogs.Table()
    if carryNames:
        # every rank formats the names of its own rows only; they travel as fixed-width UTF-32 (numpy 'U30') bytes
        nameBytes = torch.zeros((nmax, _NAME_CHARS * 4), dtype=torch.uint8, device=dev)
        if n > 0:
            own = np.ascontiguousarray(np.asarray(optimalCatalog['name'], dtype='U%d' % _NAME_CHARS))
            nameBytes[:n] = torch.from_numpy(own.view(np.uint8).reshape(n, _NAME_CHARS * 4)).to(dev)
        gatheredNames = [torch.empty_like(nameBytes) for _ in range(config.size)]
        dist.all_gather(gatheredNames, nameBytes)
        namesHost = torch.stack(gatheredNames).cpu().numpy()
        allNames = np.concatenate([namesHost[r, :c] for r, c in enumerate(counts)], axis=0)
        tab['name'] = np.ascontiguousarray(allNames).view('U%d' % _NAME_CHARS).reshape(-1)
    for j, k in enumerate(cols):
        col = allRows[:, j]
        if k == 'ringFlag':
            col = col > 0
        tab[k] = col
    tab['template'] = np.array(labels + [''], dtype='U30')[allRows[:, len(cols)].astype(np.int64)]     # -1 -> ''
    tab['tileName'] = np.array(config.allTileNames, dtype='U30')[allRows[:, len(cols) + 1].astype(np.int64)]
    if not carryNames:
        # names are a function of the coordinates: formatted when first read
        tab.setLazyNames(config.parDict['objIdent'], config.parDict['longNames'])
    return tab


def _loadCachedFilteredMap(config, f, tileName):
    """The filtered map and S/N map of filter ``f`` for ``tileName`` as an earlier run left them under
    filteredMaps/<tile>/<label>#<tile>_filteredMap.fits / _SNMap.fits, with the tile's area mask from selFn/
    (nemo/pipelines.py:202-218), on the device; None if they are not there.  The reference's dictionary has no flag mask
    (its findObjects then fails); here the one in selFn/ is used when present, else no flags."""
    if config.filteredMapsDir is None:
        return None
    from . import completeness
    from .wcs import CARWCS
    label = f['label'] + "#" + tileName
    stem = config.filteredMapsDir + os.path.sep + tileName + os.path.sep + label
    if not (os.path.exists(stem + "_filteredMap.fits") and os.path.exists(stem + "_SNMap.fits")):
        return None
    print("... loading cached filtered map %s ..." % (stem + "_filteredMap.fits"))
    t = engine.torch()
    data, header = fitsio.readFITS(stem + "_filteredMap.fits")
    d = {'data': engine.to_dev(np.ascontiguousarray(data)), 'wcs': CARWCS(header)}
    d['mapUnits'] = header['BUNIT']
    if 'BEAMNSR' in header.keys():
        d['beamSolidAngle_nsr'] = header['BEAMNSR']
        d['obsFreqGHz'] = header['FREQGHZ']
    d['SNMap'] = engine.to_dev(np.ascontiguousarray(fitsio.readFITS(stem + "_SNMap.fits")[0]))
    areaMask, wcs = completeness.loadAreaMask(tileName, config.selFnDir)
    d['surveyMask'] = engine.to_dev(np.ascontiguousarray(areaMask))
    # no flag mask file at all (a run made without writeFlagMask): no flags; a file that is there but unreadable or of another
    # shape is an error, not "no flags"
    sep = os.path.sep
    candidates = [config.selFnDir + sep + "flagMask#%s.fits" % (tileName), config.selFnDir + sep + tileName + sep + "flagMask#%s.fits" % (tileName),
                  config.selFnDir + sep + "flagMask.fits"]
    if any(os.path.exists(c) for c in candidates):
        flagMask = completeness.loadFlagMask(tileName, config.selFnDir)[0]
    else:
        flagMask = np.zeros(data.shape, dtype=np.int32)
    d['flagMask'] = engine.to_dev(np.ascontiguousarray(flagMask)).to(t.int32)
    for key in ('SNMap', 'surveyMask', 'flagMask'):
        if tuple(d[key].shape) != tuple(d['data'].shape):
            raise Exception("cached filtered map %s: '%s' has shape %s, the filtered map %s" % (stem, key, tuple(d[key].shape), tuple(d['data'].shape)))
    d['label'] = f['label']
    d['tileName'] = tileName
    return d


def _loadCachedRMSMap(config, tileName, photFilter):
    """RMS map of the reference filter for ``tileName`` as a float32 device tensor: from HBM if an earlier pass of this
    process left it there, else from selFnDir/<tile>/RMSMap_<photFilter>#<tile>.fits."""
    t = engine.torch()
    if tileName in config.cachedRMSMapDict:
        return engine.to_dev(config.cachedRMSMapDict[tileName], dtype=t.float32)
    if config.selFnDir is not None:
        fileName = config.selFnDir + os.path.sep + tileName + os.path.sep + "RMSMap_%s#%s.fits" % (photFilter, tileName)
        if os.path.exists(fileName):
            data, header = fitsio.readFITS(fileName)
            return engine.to_dev(np.asarray(data, dtype=np.float32))
    raise Exception("useCachedRMSMap: no RMS map of filter '%s' for tile '%s' (run the pipeline once without it)" % (photFilter, tileName))


_workers = {}


def _detectionWorkers(numWorkers):
    """The detection thread pool and its streams, made once per process and worker count.  A new stream per call would
    strand the device memory cached for the old one (the allocator keeps one pool per stream), and costs device
    allocations on every call."""
    if numWorkers not in _workers:
        # every worker starts by binding itself to the GPU of this process (a new host thread starts on HIP device 0)
        _workers[numWorkers] = (concurrent.futures.ThreadPoolExecutor(max_workers=numWorkers, initializer=engine.require_gpu,
                                                                      thread_name_prefix="nemo-amd-detect"), _WorkerStreams())
    return _workers[numWorkers]


class _WorkerStreams(object):
    """One HIP stream per detection worker thread, made when the thread first asks."""

    def __init__(self):
        self._local = threading.local()

    def get(self):
        s = getattr(self._local, 'stream', None)
        if s is None:
            s = self._local.stream = engine.torch().cuda.Stream(device=engine.require_gpu())
        return s


def _scaleGrouping(config, tileName, numFilters):
    """(lanes, filters per library call) for the filter scales of one tile (filters.filterMapsGroup).  Measured on MI355X:
    at 4096^2 two scales side by side on two HIP streams fill the gaps that one scale's small kernels (spline, map maximum,
    noise-grid set-up) and kernel tails leave - 26.5 -> 24.8 ms per 15-scale step; more lanes lose to cache and CU sharing
    (4 lanes: 29.7 ms), and larger groups delay the detection stream.  A 1-2 Mpix tile hands over all its scales at once: the
    cost there is the host's per call and per fetch, not the kernels'.  ``scaleLanes`` / ``scaleGroup`` in the config (or
    NEMO_AMD_SCALE_LANES / _GROUP) override."""
    lanes = int(os.environ.get('NEMO_AMD_SCALE_LANES', config.parDict.get('scaleLanes', 0)))
    group = int(os.environ.get('NEMO_AMD_SCALE_GROUP', config.parDict.get('scaleGroup', 0)))
    if lanes <= 0 or group <= 0:
        npix = None
        try:
            cs = config.tileCoordsDict[tileName]['clippedSection']
            npix = (cs[3] - cs[2]) * (cs[1] - cs[0])
        except Exception:
            pass
        small = npix is not None and npix * max(1, len(config.unfilteredMapsDictList)) < 6e6
        if lanes <= 0:
            lanes = 2
        if group <= 0:
            group = 16 if small else 2
    return max(1, lanes), max(1, min(group, numFilters))


def _detectAndMeasure(config, jobs, stream, invertMap, measureFluxes, photFilter, verbose):
    """photometry.findObjects + measureFluxes for the (filter, tile) jobs of one group - nemo/pipelines.py:267-311; returns
    their catalogues in order.  With ``stream`` the work is enqueued on that HIP stream (called from the detection worker
    thread) after the filtered maps are complete.  The detections of a group go through photometry.findObjectsGroup: all
    maps are enqueued before anything is fetched."""
    t = engine.torch()
    engine.require_gpu()        # (this may be a worker thread: same GPU as the thread that filtered the maps)
    if isinstance(stream, _WorkerStreams):
        stream = stream.get()
    ctx = t.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
    with ctx:
        if stream is not None:
            stream.wait_event(jobs[-1][4])          # (the events of a group are recorded in order on one stream)
        if config.parDict.get('forcedPhotometryCatalog', None) is not None:       # nemo/pipelines.py:267-274
            cats = [photometry.makeForcedPhotometryCatalog(job[2], config.parDict['forcedPhotometryCatalog'],
                                                           useInterpolator=config.parDict['useInterpolator']) for job in jobs]
        else:
            # what measureFluxes / getSNRValues will read at the object positions: fetched with findObjects' own results
            prefetches = []
            for label, f, filteredMapDict, photFilteredMapDict, ready in jobs:
                prefetch = []
                if measureFluxes == True:
                    if photFilteredMapDict is not None:
                        prefetch.append((photFilteredMapDict['SNMap'], False))
                    prefetch.append((filteredMapDict['data'], False))
                    if photFilteredMapDict is not None:
                        prefetch.append((photFilteredMapDict['data'], False))
                elif photFilter is not None:
                    prefetch.append((photFilteredMapDict['SNMap'], invertMap))
                prefetches.append(prefetch)
            cats = photometry.findObjectsGroup([job[2] for job in jobs], _prefetch=prefetches,
                                               threshold=config.parDict['thresholdSigma'],
                                               minObjPix=config.parDict['minObjPix'],
                                               findCenterOfMass=config.parDict['findCenterOfMass'],
                                               flagRings=config.parDict['flagRings'],
                                               ringFlagDistArcmin=config.parDict['ringFlagDistArcmin'],
                                               ringFlagMinRingerSNR=config.parDict['ringFlagMinRingerSNR'],
                                               ringThresholdSigma=config.parDict['ringThresholdSigma'],
                                               rejectBorder=config.parDict['rejectBorder'],
                                               objIdent=config.parDict['objIdent'], longNames=config.parDict['longNames'],
                                               useInterpolator=config.parDict['useInterpolator'],
                                               measureShapes=config.parDict['measureShapes'], invertMap=invertMap,
                                               verbose=verbose)
        if measureFluxes == True:                                                  # nemo/pipelines.py:299-311
            photometry.measureFluxesGroup(cats, [job[2] for job in jobs], config.diagnosticsDir, [job[3] for job in jobs],
                                          useInterpolator=config.parDict['useInterpolator'])
        elif photFilter is not None:
            for catalog, (label, f, filteredMapDict, photFilteredMapDict, ready) in zip(cats, jobs):
                photometry.getSNRValues(catalog, photFilteredMapDict['SNMap'], filteredMapDict['wcs'], prefix='fixed_',
                                        useInterpolator=config.parDict['useInterpolator'], invertMap=invertMap)
        if stream is not None:
            stream.synchronize()      # everything that reads the maps has finished before their memory can be reused
    return cats


def filterMapsAndMakeCatalogs(config, rootOutDir=None, useCachedFilters=False, useCachedRMSMap=False,
                              useCachedFilteredMaps=False, measureFluxes=True, invertMap=False, verbose=True,
                              writeAreaMask=False, writeFlagMask=False):
    """Runs the map filtering and catalogue construction steps (nemo/pipelines.py:35-115).  Returns the optimal
    catalogue (highest S/N detection across filter scales), the same on every rank."""
    if config.filterSets != [] and useCachedFilters == False and useCachedFilteredMaps == False:
        for setNum in config.filterSets:
            if verbose and config.rank == 0:
                print(">>> Filter set: %d" % (setNum))
            config.setFilterSet(setNum)
            if setNum == config.filterSets[-1]:
                writeAreaMask = True
                writeFlagMask = True
            config.filterSetOptions[setNum]['catalog'] = _filterMapsAndMakeCatalogs(config, verbose=verbose,
                                                                                    useCachedFilters=False,
                                                                                    useCachedFilteredMaps=False,
                                                                                    writeAreaMask=writeAreaMask,
                                                                                    writeFlagMask=writeFlagMask)
            if config.filterSetOptions[setNum]['addSiphonedFromSets'] is not None:
                toStack = [config.filterSetOptions[setNum]['catalog']]
                for siphonSetNum in config.filterSetOptions[setNum]['addSiphonedFromSets']:
                    toStack.append(config.filterSetOptions[siphonSetNum]['catalog'])
                config.filterSetOptions[setNum]['catalog'] = catalogs.vstack(toStack)
        catalog = config.filterSetOptions[config.filterSets[-1]]['catalog']
    else:
        catalog = _filterMapsAndMakeCatalogs(config, rootOutDir=rootOutDir, useCachedFilters=useCachedFilters,
                                             useCachedFilteredMaps=useCachedFilteredMaps, useCachedRMSMap=useCachedRMSMap,
                                             measureFluxes=measureFluxes, invertMap=invertMap, verbose=verbose,
                                             writeAreaMask=writeAreaMask, writeFlagMask=writeFlagMask)
    if verbose == True and config.rank == 0:
        print("... after map filtering and making catalogs: time since start = %.3f sec" % (time.time() - config._timeStarted))
    if isinstance(catalog, list) and catalog == []:
        catalog = catalogs.Table(names=('SNR', 'RADeg', 'decDeg'))
    return catalog


def _filterMapsAndMakeCatalogs(config, rootOutDir=None, useCachedFilters=False, useCachedRMSMap=False,
                               useCachedFilteredMaps=False, measureFluxes=True, invertMap=False, verbose=True,
                               writeAreaMask=False, writeFlagMask=False):
    """nemo/pipelines.py:118-398."""
    # nemo/pipelines.py:177-179: with a cached RMS map the pixel window is undone after the new S/N map is made
    undoPixelWindow = config.parDict['undoPixelWindow']
    if useCachedRMSMap == True:
        undoPixelWindow = False
    if rootOutDir is not None:
        diagnosticsDir = rootOutDir + os.path.sep + "diagnostics"
        for d in [rootOutDir, rootOutDir + os.path.sep + "filteredMaps", diagnosticsDir]:
            os.makedirs(d, exist_ok=True)
    diagnosticsDir = config.diagnosticsDir

    photFilter = config.parDict['photFilter']
    filtersList = []
    if photFilter is not None:
        for f in config.parDict['mapFilters']:
            if f['label'] == photFilter:
                filtersList.append(f)
    for f in config.parDict['mapFilters']:
        if photFilter is not None and f['label'] == photFilter:
            continue
        filtersList.append(f)
    if photFilter is not None and len(config.parDict['mapFilters']) > 1:
        assert (filtersList[0]['label'] == photFilter)
    photFilteredMapDict = None

    catalogDict = {}
    config.areaMaskDict, config.flagMaskDict = {}, {}
    config.stitchedFilteredMapDict, config.stitchedSNMapDict, config.stitchedRMSMapDict = {}, {}, {}
    keepDevice = filters.RETURN_DEVICE_ARRAYS
    filters.RETURN_DEVICE_ARRAYS = True
    t = engine.torch()
    engine.require_gpu()
    pool, detStream, pending = None, None, []
    timing = os.environ.get('NEMO_AMD_TIMING') == '1'       # host-side milestones of the step (diagnostics)
    marks = [('start', time.perf_counter())]
    if config.parDict.get('asyncDetection', True):
        # detection workers, each with a HIP stream of its own (scales are independent of each other: with two, the host
        # round trips of one scale's detection overlap the next one's)
        numWorkers = max(1, int(os.environ.get('NEMO_AMD_ASYNC_WORKERS', config.parDict.get('asyncWorkers', 1))))
        pool, detStream = _detectionWorkers(numWorkers)
        si = float(os.environ.get('NEMO_AMD_SWITCH_INTERVAL', '0'))
        if si > 0:
            import sys
            sys.setswitchinterval(si)
    try:
        for tileName in config.tileNames:
            if verbose == True:
                print(">>> [rank = %d] Making filtered maps - tileName = %s " % (config.rank, tileName))
            tileCache = {}      # what all filter scales of this tile share (data FFTs, noise covariance, masks): HBM-resident
            if useCachedFilters == True and tileName in config.cachedFilters:
                # the reference filter object of an earlier pass, still holding its cube in HBM (the reference rereads
                # diagnostics/<tile>/filter_<label>#<tile>.fits; that file is used when present)
                tileCache[('cachedFilter', config.cachedFilters[tileName].label)] = config.cachedFilters[tileName]
            if useCachedRMSMap == True and photFilter is not None and \
                    not any(f['params'].get('saveFilteredMaps', False) == True for f in filtersList):
                # the S/N map of this pass comes from the cached noise map (below) and nothing keeps the pass's own noise map:
                # it is not computed (the reference computes it and drops it; every output is unchanged)
                tileCache['cachedRMS'] = _loadCachedRMSMap(config, tileName, photFilter)
            lanes, groupSize = _scaleGrouping(config, tileName, len(filtersList))
            def filteredScales():
                # the filters of this tile, `groupSize` of them per library call (filters.filterMapsGroup); with
                # useCachedFilteredMaps a filter whose maps an earlier run left on disk is read instead (nemo/pipelines.py:202-218)
                for g0 in range(0, len(filtersList), groupSize):
                    group = filtersList[g0:g0 + groupSize]
                    cached = {}
                    if useCachedFilteredMaps == True:
                        for f in group:
                            loaded = _loadCachedFilteredMap(config, f, tileName)
                            if loaded is not None:
                                cached[f['label']] = loaded
                    toFilter = [f for f in group if f['label'] not in cached]
                    res = filters.filterMapsGroup(config.unfilteredMapsDictList, toFilter, tileName, diagnosticsDir=diagnosticsDir,
                                                  selFnDir=config.selFnDir, verbose=verbose, undoPixelWindow=undoPixelWindow,
                                                  useCachedFilter=useCachedFilters, tileCache=tileCache, lanes=lanes,
                                                  keepFilters={photFilter}) if toFilter else []     # (the one cube kept below)
                    for k, f in enumerate(group):
                        last = k == len(group) - 1
                        if f['label'] in cached:
                            yield (f, cached[f['label']], None, last)
                        else:
                            yield (f,) + res.pop(0) + (last,)

            groupJobs = []
            for f, filteredMapDict, filterObj, lastOfGroup in filteredScales():
                label = f['label'] + "#" + tileName
                catalogDict[label] = {}
                if f['label'] == photFilter and filterObj is not None and not (useCachedFilters == True and tileName in config.cachedFilters):
                    # (a pass that took its filter from the kept object keeps that object: replacing it with this pass's - whose
                    # tile cache refers to the kept one - would chain every pass's tile state to the next, thousands of them in
                    # an injection run)
                    config.cachedFilters[tileName] = filterObj
                if f['label'] == photFilter and useCachedRMSMap == False and 'RMSMap' in filteredMapDict:
                    # kept in HBM for later passes with useCachedRMSMap (the reference reloads selFn/RMSMap_<photFilter>.fits,
                    # nemo/completeness.py:1070-1088); also written there when saveRMSMap is set and there is a selFnDir
                    config.cachedRMSMapDict[tileName] = filteredMapDict['RMSMap']
                    if f['params'].get('saveRMSMap', False) == True and config.selFnDir is not None:
                        rmsDir = config.selFnDir + os.path.sep + tileName
                        os.makedirs(rmsDir, exist_ok=True)
                        fitsio.writeFITS(rmsDir + os.path.sep + "RMSMap_%s#%s.fits" % (photFilter, tileName),
                                         filteredMapDict['RMSMap'].cpu().numpy(), {})
                if useCachedRMSMap == True and photFilter is not None:            # nemo/pipelines.py:234-250
                    eng = engine.TileEngine.get(tuple(filteredMapDict['data'].shape))
                    if not filteredMapDict.get('_snFromCachedRMS', False):          # (else the filter's tail made it already)
                        RMSMap = _loadCachedRMSMap(config, tileName, photFilter)
                        filteredMapDict['SNMap'] = eng.sn_cached(filteredMapDict['data'], RMSMap)
                    filteredMapDict['data'] = eng.undo_pixel_window(filteredMapDict['data'], zeroRows=filteredMapDict.get('_zeroRows'))
                if 'saveFilteredMaps' in f['params'] and f['params']['saveFilteredMaps'] == True:
                    t = engine.torch()
                    config.stitchedFilteredMapDict[tileName] = filteredMapDict['data'].to(t.float32)
                    config.stitchedSNMapDict[tileName] = filteredMapDict['SNMap'].to(t.float32)
                    config.stitchedRMSMapDict[tileName] = filteredMapDict['RMSMap'].to(t.float32)
                if f['label'] == photFilter:
                    photFilteredMapDict = {'SNMap': filteredMapDict['SNMap'], 'data': filteredMapDict['data']}
                # Detection + photometry of this scale run on a second HIP stream from a worker thread, so that their host
                # round trips (object count, catalogue rows) overlap the filtering of the next scale instead of idling the GPU.
                if writeAreaMask == True and tileName not in config.areaMaskDict.keys():
                    config.areaMaskDict[tileName] = filteredMapDict['surveyMask']
                if writeFlagMask == True and tileName not in config.flagMaskDict.keys():
                    config.flagMaskDict[tileName] = filteredMapDict['flagMask']
                # (the worker waits for the group's last event only: the others are not made)
                ready = None
                if lastOfGroup:
                    ready = t.cuda.Event()
                    ready.record(t.cuda.current_stream())
                groupJobs.append((label, f, filteredMapDict, photFilteredMapDict, ready))
                del filteredMapDict, filterObj
                if not lastOfGroup:
                    continue
                # the detections of the group go out together (photometry.findObjectsGroup)
                labels = [job[0] for job in groupJobs]
                if pool is not None:
                    pending.append((labels, pool.submit(_detectAndMeasure, config, groupJobs, detStream, invertMap, measureFluxes,
                                                        photFilter, verbose)))
                else:
                    for lab, cat in zip(labels, _detectAndMeasure(config, groupJobs, None, invertMap, measureFluxes, photFilter,
                                                                  verbose)):
                        catalogDict[lab]['catalog'] = cat
                groupJobs = []
                # bound the maps kept alive for the worker to a few groups (≈0.7 GB per scale at 4096^2); deep enough that the
                # enqueueing thread is not put to sleep while the GPU still has less than a scale of work queued
                depth = int(os.environ.get('NEMO_AMD_ASYNC_DEPTH', config.parDict.get('asyncDepth', 2)))
                if len(pending) > depth:
                    pending[-depth - 1][1].result()
            del tileCache
        marks.append(('filters enqueued', time.perf_counter()))
        for labels, fut in pending:
            for lab, cat in zip(labels, fut.result()):
                catalogDict[lab]['catalog'] = cat
        pending = []
        marks.append(('detection done', time.perf_counter()))
    finally:
        filters.RETURN_DEVICE_ARRAYS = keepDevice
        for labels, fut in pending:       # (only after an exception: let the worker finish with the maps it holds)
            fut.cancel()

    marks.append(('pool closed', time.perf_counter()))
    optimalCatalog = catalogs.makeOptimalCatalog(catalogDict, constraintsList=config.parDict['catalogCuts'],
                                                 photFilter=photFilter, method=config.parDict['optimalCatalogMethod'])
    marks.append(('optimal catalogue', time.perf_counter()))
    if timing:
        print("[timing] " + "  ".join("%s +%.2f ms" % (k, 1e3 * (v - marks[0][1])) for k, v in marks[1:]), flush=True)
        print("[timing] worker: " + "  ".join("%s %.2f ms" % (k, 1e3 * v) for k, v in photometry._acc.items()), flush=True)
        photometry._acc.clear()
    if config.MPIEnabled == True and config.size > 1:
        optimalCatalog = _gatherCatalogs(optimalCatalog, config)
        if len(optimalCatalog) > 0:
            optimalCatalog, numDuplicatesFound, names = catalogs.removeDuplicates(optimalCatalog)
    if config.selFnDir is not None and config.filteredMapsDir is not None and config.parDict.get('writeTileProducts', True):
        _writeTileProducts(config, photFilter, writeAreaMask, writeFlagMask)
    return optimalCatalog


def _writeTileProducts(config, photFilter, writeAreaMask, writeFlagMask):
    """The five per-tile products of a pass - area mask, flag mask, RMS map, filtered map, S/N map of the reference filter -
    brought to rank 0 and written as multi-extension files and / or stitched images (nemo/pipelines.py:338-396; same file
    names).  Tiles travel as host arrays through gather_object (the reference sends pickled dictionaries rank by rank)."""
    fileLabel = ''
    if config.currentFilterSet is not None and 'fileLabel' in config.filterSetOptions[config.currentFilterSet].keys() \
            and config.filterSetOptions[config.currentFilterSet]['fileLabel'] is not None:
        fileLabel = config.filterSetOptions[config.currentFilterSet]['fileLabel'] + '-'
    sep = os.path.sep
    if config.rank == 0:
        for d in (config.selFnDir, config.filteredMapsDir):
            os.makedirs(d, exist_ok=True)
    products = [('area mask', config.areaMaskDict, writeAreaMask, writeAreaMask, config.selFnDir + sep + "areaMask.fits",
                 config.selFnDir + sep + "stitched_areaMask.fits", 'PLIO_1'),
                ('flag mask', config.flagMaskDict, writeFlagMask, writeFlagMask, config.selFnDir + sep + "flagMask.fits",
                 config.selFnDir + sep + "stitched_flagMask.fits", 'PLIO_1'),
                ('RMS map', config.stitchedRMSMapDict, True, True, config.selFnDir + sep + "RMSMap_%s%s.fits" % (fileLabel, photFilter),
                 config.selFnDir + sep + "stitched_RMSMap_%s%s.fits" % (fileLabel, photFilter), None),
                ('filtered map', config.stitchedFilteredMapDict, False, True, None,
                 config.filteredMapsDir + sep + "stitched_%s%s_filteredMap.fits" % (fileLabel, photFilter), None),
                ('S/N map', config.stitchedSNMapDict, False, True, None,
                 config.filteredMapsDir + sep + "stitched_%s%s_SNMap.fits" % (fileLabel, photFilter), None)]
    for label, tiles, writeMEF, writeStitched, MEFPath, stitchedPath, compressionType in products:
        if not (writeMEF or writeStitched):
            continue
        tileDict = maps.TileDict(dict(tiles), tileCoordsDict=config.tileCoordsDict)
        if config.MPIEnabled == True and config.size > 1:
            import torch.distributed as dist
            mine = {k: (v.cpu().numpy() if hasattr(v, 'is_cuda') else np.asarray(v)) for k, v in tileDict.items()}
            gathered = [None] * config.size if config.rank == 0 else None
            dist.gather_object(mine, gathered, dst=0)
            if config.rank == 0:
                for part in gathered:
                    for tileName in part.keys():
                        tileDict[tileName] = part[tileName]
        if config.rank == 0 and len(tileDict) > 0:
            if writeMEF == True and MEFPath is not None:
                tileDict.saveMEF(MEFPath, compressionType=compressionType)
            if writeStitched == True and stitchedPath is not None:
                tileDict.saveStitchedFITS(stitchedPath, config.origWCS, compressionType=compressionType)
