"""Configuration, tiling and the static tile -> GPU assignment (reference: nemo/startUp.py).

* :func:`parseConfigFile`  - nemo/startUp.py:22-228: YAML (or an already-loaded dict) -> ``parDict`` with the reference's
  defaults, the ``allFilters`` merge and the global mask keys copied into each map entry.
* :class:`NemoConfig`      - nemo/startUp.py:231-843 for the attributes the hot path reads (``parDict``,
  ``unfilteredMapsDictList``, ``tileNames`` / ``allTileNames``, ``tileCoordsDict``, output directories, ``rank`` / ``size``,
  ``filterSets``, ``cachedFilters``).

What replaces MPI (nemo/startUp.py:285-304, 409-446): one process per GPU (``torch.distributed``, backend nccl = RCCL on
the GPUs, gloo in CPU tests), tiles assigned statically and greedily by ``numPix`` to ALL ranks (the reference leaves
rank 0 idle; nothing here needs a coordinator).  Tiles are independent, so no collective runs inside the per-tile
compute; the only exchange is the catalogue all-gather in :mod:`nemo_amd.pipelines`.

Maps may be given as files (``mapFileName`` .npy / plain FITS) or in memory (``mapData`` / ``weightsData`` arrays or
device tensors together with a ``wcs``), which is what the synthetic benchmark uses.
"""
import copy
import os
import time

import numpy as np

from . import maps
from .wcs import CARWCS


def _overlay(target, source, levels):
    """``source`` laid over ``target`` in place: values of ``source`` win; a dictionary value is merged into the one already
    there as long as ``levels`` allows, and replaces it whole below that."""
    for key, value in source.items():
        if isinstance(value, dict) and levels > 0:
            if not isinstance(target.get(key), dict):
                target[key] = {}
            _overlay(target[key], value, levels - 1)
        else:
            target[key] = value
    return target


def _applyDefaults(parDict):
    maskKeys = ['pointSourceMask', 'surveyMask', 'maskPointSourcesFromCatalog', 'apodizeUsingSurveyMask',
                'maskSubtractedPointSources', 'RADecSection', 'maskHoleDilationFactor', 'reprojectToTan']
    for mapDict in parDict['unfilteredMaps']:
        for k in maskKeys:
            if k in parDict.keys():
                mapDict[k] = parDict[k]
            elif k not in mapDict:
                mapDict[k] = None
        if 'weightsType' not in mapDict.keys():
            mapDict['weightsType'] = 'invVar'
    # allFilters holds options common to every filter; what a mapFilters entry says itself wins (nemo/startUp.py:57-80:
    # dictionaries are merged key by key for two levels below the entry - params, then e.g. noiseParams - deeper ones whole)
    common = parDict.get('allFilters')
    if common is not None:
        parDict['mapFilters'] = [_overlay(copy.deepcopy(common), entry, 2) for entry in parDict['mapFilters']]
    if 'photFilter' not in parDict.keys():
        parDict['photFilter'] = None
    else:
        for filtDict in parDict['mapFilters']:
            if filtDict['label'] == parDict['photFilter']:
                filtDict['params']['saveRMSMap'] = True
                filtDict['params']['saveFreqWeightMap'] = True
                filtDict['params']['saveFilter'] = True
    if 'tileDefinitions' in parDict.keys() and type(parDict['tileDefinitions']) == list:
        checkList = []
        for tileDef in parDict['tileDefinitions']:
            tileDef['tileName'] = tileDef['tileName'].upper()
            if tileDef['tileName'] in checkList:
                raise Exception("Duplicate tileName '%s' in tileDefinitions - fix in config file" % (tileDef['tileName']))
            checkList.append(tileDef['tileName'])
    if 'tileNameList' in parDict.keys():
        parDict['tileNameList'] = [entry.upper() for entry in parDict['tileNameList']]
    defaults = {'reprojectToTan': False, 'catalogCuts': [], 'measureShapes': False, 'rejectBorder': 0,
                'optimalCatalogMethod': 'ref-forced', 'undoPixelWindow': True, 'useTiling': False, 'GNFWParams': 'default',
                'forcedPhotometryCatalog': None, 'flagRings': False, 'ringFlagDistArcmin': 12.0, 'ringFlagMinRingerSNR': 20.0,
                'ringThresholdSigma': 0.5, 'thresholdSigma': 4.0, 'minObjPix': 1, 'findCenterOfMass': True,
                'useInterpolator': True, 'objIdent': 'ACT-CL', 'longNames': False, 'tileOverlapDeg': 1.0}
    # (the last eight have no default in the reference - its configs always give them; docs/config.rst values)
    for k, v in defaults.items():
        if k not in parDict.keys():
            parDict[k] = v
    for filtDict in parDict['mapFilters']:
        filtDict['params']['GNFWParams'] = parDict['GNFWParams']
    parDict['stitchTiles'] = True
    oldKeyMap = {'makeTileDir': 'useTiling', 'tileDefLabel': None, 'twoPass': None,
                 'clusterInjectionModels': 'sourceInjectionModels'}
    for k in oldKeyMap.keys():
        if k in list(parDict.keys()) and oldKeyMap[k] is None:
            del parDict[k]
        if k in list(parDict.keys()) and type(oldKeyMap[k]) == str:
            parDict[oldKeyMap[k]] = parDict[k]
            del parDict[k]
    return parDict


def parseConfigFile(parDictFileName, verbose=False):
    """Parse a nemo .yml config file (nemo/startUp.py:22-228).  A dict is accepted in place of a path."""
    if isinstance(parDictFileName, dict):
        return _applyDefaults(parDictFileName)
    import yaml
    with open(parDictFileName, "r") as stream:
        parDict = yaml.safe_load(stream)
    parDict = _applyDefaults(parDict)
    parDict['_file_last_modified_ctime'] = os.path.getctime(parDictFileName)
    return parDict


def assignTiles(tileNames, numPix, size):
    """Static tile -> rank lists balanced by pixel count: largest tile first onto the least-loaded rank (the weight
    the reference balances on is the same ``numPix``, nemo/startUp.py:416-426)."""
    order = sorted(range(len(tileNames)), key=lambda i: (-numPix[i], i))
    load = [0] * size
    out = [[] for _ in range(size)]
    for i in order:
        r = min(range(size), key=lambda k: (load[k], k))
        out[r].append(tileNames[i])
        load[r] += numPix[i]
    # keep each rank's tiles in the original order
    pos = {n: i for i, n in enumerate(tileNames)}
    return [sorted(l, key=lambda n: pos[n]) for l in out]


class NemoConfig(object):
    """nemo/startUp.py:231-843, hot-path subset."""

    def __init__(self, config, makeOutputDirs=True, setUpMaps=True, writeTileInfo=False, selFnDir=None, calcSelFn=False,
                 sourceInjectionTest=False, MPIEnabled=False, divideTilesByProcesses=True, verbose=True, strictMPIExceptions=True,
                 rank=None, size=None):
        self.MPIEnabled = MPIEnabled
        self.comm = None
        if rank is None or size is None:
            rank, size = 0, 1
            if MPIEnabled:
                import torch.distributed as dist
                if not dist.is_initialized():
                    raise Exception("MPIEnabled (multi-GPU) needs torch.distributed to be initialised first - launch with "
                                    "python -m torch.distributed.run and call nemo_amd.pipelines.initDistributed()")
                rank, size = dist.get_rank(), dist.get_world_size()
        self.rank, self.size = rank, size
        self.verbose = verbose if self.rank == 0 else False
        self._timeStarted = time.time()
        if type(config) == str:
            self.parDict = parseConfigFile(config, verbose=self.verbose)
            self.configFileName = config
        elif type(config) == dict:
            self.parDict = parseConfigFile(config)
            self.configFileName = ''
        else:
            raise Exception("'config' must be either a path to a .yml file, or a dictionary of parameters.")
        self._origParDict = self._copyParDict(self.parDict)

        self.rootOutDir = None
        if 'outputDir' in self.parDict and self.parDict['outputDir'] is not None:
            self.rootOutDir = os.path.abspath(self.parDict['outputDir'])
        elif self.configFileName != '':
            self.rootOutDir = os.getcwd() + os.path.sep + os.path.split(self.configFileName.replace(".yml", ""))[-1]
        if self.rootOutDir is not None:
            self.filteredMapsDir = self.rootOutDir + os.path.sep + "filteredMaps"
            self.diagnosticsDir = self.rootOutDir + os.path.sep + "diagnostics"
            self.selFnDir = self.rootOutDir + os.path.sep + "selFn"
            if self.rank == 0 and makeOutputDirs:
                for d in [self.rootOutDir, self.diagnosticsDir, self.filteredMapsDir, self.selFnDir]:
                    os.makedirs(d, exist_ok=True)
        else:
            # in-memory run (benchmark, tests): nothing is written
            self.filteredMapsDir = self.diagnosticsDir = self.selFnDir = None
        if selFnDir is not None:
            self.selFnDir = selFnDir

        self._setUpMaps()
        if 'tileNameList' in self.parDict:
            newList = [n for n in self.tileNames if n in self.parDict['tileNameList']]
            if newList == []:
                raise Exception("tileNameList given in nemo config file but no extensions in images match")
            self.tileNames = newList
        self.allTileNames = list(self.tileNames)
        if self.MPIEnabled and divideTilesByProcesses:
            numPix = [self.tileCoordsDict[k]['numPix'] for k in self.tileNames]
            self.tileNames = assignTiles(self.tileNames, numPix, self.size)[self.rank]
        if writeTileInfo == True and self.selFnDir is not None and self.rank == 0:           # nemo/startUp.py:689-693
            import pickle
            os.makedirs(self.selFnDir, exist_ok=True)
            plain = {k: {kk: vv for kk, vv in v.items() if kk != 'wcs'} for k, v in self.tileCoordsDict.items()}
            with open(self.selFnDir + os.path.sep + "tileCoordsDict.pkl", "wb") as pickleFile:
                pickle.Pickler(pickleFile).dump(plain)
        if makeOutputDirs and self.diagnosticsDir is not None:
            for tileName in self.tileNames:
                for d in [self.diagnosticsDir, self.selFnDir]:
                    os.makedirs(d + os.path.sep + tileName, exist_ok=True)
        self._identifyFilterSets()
        self.currentFilterSet = None
        self.cachedFilters = {}
        self.cachedRMSMapDict = {}      # tile -> float32 RMS map of the reference filter (HBM), for useCachedRMSMap passes

    @staticmethod
    def _copyParDict(parDict):
        """deepcopy that shares map arrays / device tensors instead of duplicating them."""
        memo = {}
        for m in parDict.get('unfilteredMaps', []):
            for k in ('mapData', 'weightsData', 'surveyMask', 'pointSourceMask', 'flagMask', 'wcs', 'beamFileName'):
                v = m.get(k)
                if v is not None and not isinstance(v, (str, int, float)):
                    memo[id(v)] = v
        return copy.deepcopy(parDict, memo)

    def _mapWCS(self):
        m = self.parDict['unfilteredMaps'][0]
        if m.get('wcs') is not None:
            wcs = m['wcs']
            shape = tuple(m['mapData'].shape[-2:]) if m.get('mapData') is not None else (wcs.header['NAXIS2'], wcs.header['NAXIS1'])
            return wcs, shape
        name = m['mapFileName']
        if name.endswith('.npy'):
            raise Exception("a .npy map needs a 'wcs' entry (CARWCS) in its unfilteredMaps dictionary")
        from . import fitsio
        data, header = fitsio.readFITS(name, headerOnly=True)
        return CARWCS(header), (header['NAXIS2'], header['NAXIS1'])

    def addAutoTileDefinitions(self):
        """tileDefinitions given as {mask, targetTileWidthDeg, targetTileHeightDeg} become the list maps.autotiler makes
        (nemo/startUp.py:510-563).  ``mask``: an array (with the map's WCS), a FITS / .npy file, or None - then the area where
        the first map is non-zero."""
        td = self.parDict.get('tileDefinitions')
        if type(td) != dict:
            return
        wcs, shape = self._mapWCS()
        mask = td.get('mask')
        if mask is None:
            first = maps.MapDict(self.parDict['unfilteredMaps'][0])
            data = first._load(first.get('mapData', first.get('mapFileName')))
            data = data.cpu().numpy() if hasattr(data, 'is_cuda') else np.asarray(data)
            mask = np.array(data[0] if data.ndim == 3 else data) != 0
        elif isinstance(mask, str):
            if mask.endswith('.npy'):
                mask = np.load(mask)
            else:
                from . import fitsio
                mask, header = fitsio.readFITS(mask)
                wcs = CARWCS(header)
        mask = mask.cpu().numpy() if hasattr(mask, 'is_cuda') else np.asarray(mask)
        self.parDict['tileDefinitions'] = maps.autotiler(np.array(mask != 0, dtype=np.uint8), wcs, td['targetTileWidthDeg'],
                                                         td['targetTileHeightDeg'])
        if self.verbose:
            print("... breaking map into %d tiles" % (len(self.parDict['tileDefinitions'])))

    def getTileCoordsDict(self):
        """nemo/startUp.py:566-661 for CAR maps: pixel sections of every tile (+ overlap on interior sides), the tile WCS
        header, the part of the tile that is kept when stitching, and its pixel count."""
        wcs, shape = self._mapWCS()
        H, W = shape
        out = {}
        if self.parDict['useTiling'] == False:
            out['PRIMARY'] = {'clippedSection': [0, W, 0, H], 'header': dict(wcs.header), 'wcs': wcs,
                              'areaMaskInClipSection': [0, W, 0, H], 'reprojectToTan': False, 'numPix': int(H * W)}
            return out
        self.addAutoTileDefinitions()
        if 'tileDefinitions' not in self.parDict or type(self.parDict['tileDefinitions']) != list:
            raise Exception("useTiling needs tileDefinitions: a list of {tileName, RADecSection}, or {mask, targetTileWidthDeg, "
                            "targetTileHeightDeg} for automatic tiling")
        overlapPix = self.parDict['tileOverlapDeg'] / wcs.getPixelSizeDeg()
        for tileDict in self.parDict['tileDefinitions']:
            ra0, ra1, dec0, dec1 = tileDict['RADecSection']
            xa, ya = wcs.wcs2pix(ra0, dec0)
            xb, yb = wcs.wcs2pix(ra1, dec1)
            x0, x1 = min(xa, xb), max(xa, xb)
            y0, y1 = min(ya, yb), max(ya, yb)
            cx0 = x0 - overlapPix if x0 - overlapPix > 0 else x0
            cx1 = x1 + overlapPix if x1 + overlapPix < W else x1
            cy0 = y0 - overlapPix if y0 - overlapPix > 0 else y0
            cy1 = y1 + overlapPix if y1 + overlapPix < H else y1
            cx0, cx1 = max(int(round(cx0)), 0), min(int(round(cx1)), W)
            cy0, cy1 = max(int(round(cy0)), 0), min(int(round(cy1)), H)
            hdr = dict(wcs.header)
            hdr['NAXIS1'], hdr['NAXIS2'] = cx1 - cx0, cy1 - cy0
            hdr['CRPIX1'] = hdr['CRPIX1'] - cx0
            hdr['CRPIX2'] = hdr['CRPIX2'] - cy0
            ax0, ax1 = int(round(x0)) - cx0, int(round(x1)) - cx0
            ay0, ay1 = int(round(y0)) - cy0, int(round(y1)) - cy0
            out[tileDict['tileName']] = {'clippedSection': [cx0, cx1, cy0, cy1], 'header': hdr, 'wcs': CARWCS(hdr),
                                         'areaMaskInClipSection': [ax0, ax1, ay0, ay1], 'reprojectToTan': False,
                                         'numPix': int(abs(ax1 - ax0) * abs(ay1 - ay0))}
        return out

    def _setUpMaps(self):
        self.tileCoordsDict = self.getTileCoordsDict()
        assert (self.tileCoordsDict != {})
        self.tileNames = list(self.tileCoordsDict.keys())
        self.origWCS, self.origShape = self._mapWCS()
        self.unfilteredMapsDictList = [maps.MapDict(m, tileCoordsDict=self.tileCoordsDict)
                                       for m in self.parDict['unfilteredMaps']]
        self._origUnfilteredMapsDictList = [m.copy() for m in self.unfilteredMapsDictList]

    def restoreConfig(self):
        self.parDict = self._copyParDict(self._origParDict)
        self.unfilteredMapsDictList = [m.copy() for m in self._origUnfilteredMapsDictList]

    def _identifyFilterSets(self):
        """nemo/startUp.py:486-507."""
        self.filterSets = []
        if 'filterSetOptions' in self.parDict.keys():
            self.filterSetOptions = self.parDict['filterSetOptions']
            for filtDict in self.parDict['mapFilters']:
                if 'filterSets' in filtDict.keys():
                    for f in filtDict['filterSets']:
                        if f not in self.filterSets:
                            self.filterSets.append(f)
            self.filterSets.sort()
            self.filterSetLabels = {}
            for setNum in self.filterSetOptions.keys():
                if 'label' in self.filterSetOptions[setNum].keys():
                    raise Exception("Use 'fileLabel' instead of 'label' in 'filterSetOptions' dictionaries")
                self.filterSetLabels[setNum] = self.filterSetOptions[setNum].get('fileLabel', None)

    # what one pass of a multi-pass run may change (nemo/startUp.py:751-843), as tables
    _SET_OPTION_DEFAULTS = {'saveCatalog': False, 'maskHoleDilationFactor': None, 'maskHoleRadiusArcmin': None,
                            'addSiphonedFromSets': None, 'ignoreSurveyMask': False}
    _SET_OVERRIDES = ('thresholdSigma', 'objIdent', 'findCenterOfMass', 'measureShapes')       # run-level keys a set may replace
    _OUTPUT_SWITCHES = ('saveFilteredMaps', 'saveFilter', 'saveRMSMap', 'savePlots', 'saveDS9Regions')
    _CATALOG_ROUTES = (('subtractModelFromSets', 'subtractModelFromCatalog'), ('maskAndFillFromSets', 'maskAndFillFromCatalog'))

    def setFilterSet(self, setNum):
        """Activate pass ``setNum`` of a multi-pass run (``filterSets`` / ``filterSetOptions``; nemo/startUp.py:751-843): start
        from the configuration as read, keep the filters that belong to the pass, and hand the catalogues of earlier passes
        to the filters (noise term) and to the maps (model subtraction, masking and filling)."""
        self.restoreConfig()
        allOptions = self.filterSetOptions
        options = allOptions.get(setNum)
        if options is not None:
            self.currentFilterSet = setNum
            for key, value in self._SET_OPTION_DEFAULTS.items():
                options.setdefault(key, value)
            self.parDict.update({k: options[k] for k in self._SET_OVERRIDES if k in options})
        lastListed = list(allOptions.keys())[-1]
        if setNum != lastListed:
            self.parDict['catalogCuts'] = []                 # cuts apply to the final catalogue only
        finalPass = setNum == self.filterSets[-1]

        active = [f for f in self.parDict['mapFilters'] if setNum in f['filterSets']]
        for f in active:
            params = f['params']
            if options is not None:
                if 'mapToUse' in options:
                    params['mapToUse'] = options['mapToUse']
                if 'noiseModelCatalogFromSets' in options:
                    params['noiseModelCatalog'] = [allOptions[i]['catalog'] for i in options['noiseModelCatalogFromSets']]
            if not finalPass:
                # only the last pass writes maps / filters, or does forced photometry - unless a pass asks for it itself
                params.update({k: False for k in self._OUTPUT_SWITCHES if k in params})
                self.parDict['forcedPhotometryCatalog'] = None
            if options is not None:
                if 'forcedPhotometryCatalog' in options:
                    self.parDict['forcedPhotometryCatalog'] = options['forcedPhotometryCatalog']
                    self.parDict['thresholdSigma'] = -1000
                if 'saveFilteredMaps' in options:
                    params['saveFilteredMaps'] = options['saveFilteredMaps']
        self.parDict['mapFilters'] = active
        if options is None:
            return

        for mapDict in self.unfilteredMapsDictList:
            for fromKey, mapKey in self._CATALOG_ROUTES:
                for index in options.get(fromKey, []):
                    wanted = allOptions[index].get('mapToUse')
                    if wanted is not None and 'label' in mapDict and mapDict['label'] != wanted:
                        continue
                    mapDict[mapKey] = allOptions[index]['catalog']          # (the last listed set wins, as in the reference)
            mapDict['maskHoleDilationFactor'] = options['maskHoleDilationFactor']
            mapDict['maskHoleRadiusArcmin'] = options['maskHoleRadiusArcmin']
            if options['ignoreSurveyMask'] == True:
                mapDict['surveyMask'] = None
