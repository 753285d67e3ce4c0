"""CPU suite, part 2: the C-ABI library loads and exports every symbol include/nemofilt.h declares (no compute calls:
there is no GPU here), the host-side mirror of the reference's config / tiling logic behaves like the reference's,
the product refuses to compute without a GPU instead of falling back, and the multi-GPU catalogue exchange works under
gloo with world_size 2."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "nemofilt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from nemo_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in include/nemofilt.h is not exported" % n
    # the ctypes table binds exactly the declared entry points
    assert sorted(_lib.SIGNATURES.keys()) == names
    loaded = _lib.load()
    assert loaded.nf_version() >= 1
    assert loaded.nf_profile_num_tags() > 10
    assert loaded.nf_profile_tag_name(0).decode() == "fft_rows_f64"
    # argument checking happens before any device work: bad arguments give an error code and a message
    plan = ctypes.c_void_p()
    assert loaded.nf_plan_create(0, 16, ctypes.byref(plan)) < 0
    assert b"nf_plan_create" in loaded.nf_last_error()


def test_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors of nf_mf_tile / nf_mf_scale have the C compiler's size and field offsets."""
    from nemo_amd import _lib
    fields = {'nf_mf_tile': [f[0] for f in _lib.MfTile._fields_], 'nf_mf_scale': [f[0] for f in _lib.MfScale._fields_]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "nemofilt.h"', 'int main(void) {']
    for name, fl in fields.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (name, name))
        for f in fl:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    lines += ['return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, cls in (('nf_mf_tile', _lib.MfTile), ('nf_mf_scale', _lib.MfScale)):
        assert int(out[name]) == ctypes.sizeof(cls)
        for f in fields[name]:
            assert int(out["%s.%s" % (name, f)]) == getattr(cls, f).offset, (name, f)


def test_product_never_imports_the_oracle():
    """Nothing under nemo_amd/ or bench.py's GPU path may import the oracle (it is the checker, not the product)."""
    pkg = os.path.join(ROOT, "nemo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "nemo_oracle" not in src and "import oracle" not in src, f


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nemo_amd import engine, filters, maps
    from nemo_amd._lib import NemoFiltError
    from nemo_amd.wcs import CARWCS
    with pytest.raises(NemoFiltError):
        engine.require_gpu()
    shape = (64, 64)
    md = maps.MapDict({'mapData': np.zeros(shape), 'weightsData': np.ones(shape), 'obsFreqGHz': 150.0, 'units': 'uK',
                       'beamFileName': None, 'label': 'x', 'wcs': CARWCS.make(shape)})
    with pytest.raises(NemoFiltError):
        filters.filterMaps([md], {'label': 'f', 'class': 'BeamMatchedFilter',
                                  'params': {'noiseParams': {'method': 'dataMap'}, 'outputUnits': 'uK'}}, 'PRIMARY',
                           diagnosticsDir=None, selFnDir=None, verbose=False)


def test_config_defaults_and_allfilters_merge(tmp_path):
    """startUp.parseConfigFile: allFilters deep merge (nemo/startUp.py:57-80), photFilter forces saveFilter
    (:88-93), mask keys copied into each map (:43-53), defaults (:124-173)."""
    from nemo_amd import startUp
    yml = tmp_path / "cfg.yml"
    yml.write_text("""
unfilteredMaps:
    - {mapFileName: "a.fits", weightsFileName: "aw.fits", obsFreqGHz: 149.6, units: 'uK', beamFileName: "b.txt"}
surveyMask: "mask.fits"
thresholdSigma: 5.0
photFilter: 'Arnaud_M2e14_z0p4'
allFilters: {class: "ArnaudModelMatchedFilter",
             params: {noiseParams: {method: "dataMap", noiseGridArcmin: 40.}, outputUnits: 'yc', edgeTrimArcmin: 0.0}}
mapFilters:
    - {label: "Arnaud_M2e14_z0p4", params: {M500MSun: 2.0e+14, z: 0.4, noiseParams: {numNoiseBins: 3}}}
    - {label: "Arnaud_M4e14_z0p2", class: "BattagliaModelMatchedFilter", params: {M500MSun: 4.0e+14, z: 0.2}}
""")
    par = startUp.parseConfigFile(str(yml))
    f0, f1 = par['mapFilters']
    assert f0['class'] == 'ArnaudModelMatchedFilter' and f1['class'] == 'BattagliaModelMatchedFilter'
    assert f0['params']['noiseParams'] == {'method': 'dataMap', 'noiseGridArcmin': 40.0, 'numNoiseBins': 3}
    assert f1['params']['noiseParams'] == {'method': 'dataMap', 'noiseGridArcmin': 40.0}
    assert f0['params']['saveFilter'] is True and 'saveFilter' not in f1['params']
    assert f0['params']['outputUnits'] == 'yc' and f0['params']['z'] == 0.4
    assert par['unfilteredMaps'][0]['surveyMask'] == "mask.fits" and par['unfilteredMaps'][0]['pointSourceMask'] is None
    assert par['thresholdSigma'] == 5.0 and par['flagRings'] is False and par['ringThresholdSigma'] == 0.5
    assert par['undoPixelWindow'] is True and par['optimalCatalogMethod'] == 'ref-forced'


def test_tile_assignment_balances_by_numpix():
    from nemo_amd import startUp
    names = ['t%d' % i for i in range(11)]
    numPix = [100, 90, 80, 70, 60, 50, 40, 30, 20, 10, 5]
    parts = startUp.assignTiles(names, numPix, 4)
    assert sorted(sum(parts, [])) == sorted(names)                 # every tile exactly once, all ranks used
    loads = [sum(numPix[names.index(n)] for n in p) for p in parts]
    assert max(loads) - min(loads) <= max(numPix)
    assert all(len(p) > 0 for p in parts)
    assert startUp.assignTiles(names, numPix, 1) == [names]


def test_tile_coords_with_overlap():
    """getTileCoordsDict (nemo/startUp.py:566-661): overlap added on interior sides only; areaMaskInClipSection marks the
    un-overlapped part."""
    from nemo_amd import startUp
    from nemo_amd.wcs import CARWCS
    shape = (600, 1200)
    wcs = CARWCS.make(shape, RADeg=20.0, decDeg=0.0)
    ra0, ra1, dec0, dec1 = wcs.getImageMinMaxWCSCoords()
    mid = 0.5 * (ra0 + ra1)
    ra0, ra1, dec0, dec1 = ra0 - 0.5, ra1 + 0.5, dec0 - 0.5, dec1 + 0.5      # sections reaching past the map are clipped
    par = {'unfilteredMaps': [{'mapData': np.zeros(shape, dtype=np.float32), 'weightsData': np.ones(shape, dtype=np.float32),
                               'wcs': wcs, 'obsFreqGHz': 150.0, 'units': 'uK', 'beamFileName': None, 'label': 'f150'}],
           'mapFilters': [{'label': 'b', 'class': 'BeamMatchedFilter',
                           'params': {'noiseParams': {'method': 'dataMap'}, 'outputUnits': 'uK'}}],
           'useTiling': True, 'tileOverlapDeg': 1.0, 'outputDir': None,
           'tileDefinitions': [{'tileName': 'east', 'RADecSection': [mid, ra1, dec0, dec1]},
                               {'tileName': 'west', 'RADecSection': [ra0, mid, dec0, dec1]}]}
    cfg = startUp.NemoConfig(par, makeOutputDirs=False, verbose=False)
    assert cfg.tileNames == ['EAST', 'WEST']
    e, w = cfg.tileCoordsDict['EAST'], cfg.tileCoordsDict['WEST']
    x0e, x1e, y0e, y1e = e['clippedSection']
    x0w, x1w, y0w, y1w = w['clippedSection']
    assert (y0e, y1e) == (0, 600) and (y0w, y1w) == (0, 600)
    assert x0e == 0 and x1w == 1200
    assert x1e - x0w == 2 * 120                                       # 1 deg = 120 px of overlap on each interior side
    tile = cfg.unfilteredMapsDictList[0].loadTile('mapFileName', 'EAST')
    assert tile.shape == (600, x1e - x0e)


_GLOO_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from nemo_amd import pipelines, catalogs, startUp
rank, size = pipelines.initDistributed()
assert size == 2 and dist.get_backend() == 'gloo'

class Cfg(object):
    pass
cfg = Cfg()
cfg.size, cfg.rank = size, rank
cfg.allTileNames = ['T0', 'T1', 'T2']
cfg.parDict = {'mapFilters': [{'label': 'fa'}, {'label': 'fb'}], 'objIdent': 'ACT-CL', 'longNames': False}
tab = catalogs.Table()
tab.setLazyNames('ACT-CL')
if rank == 0:
    tab['RADeg'] = np.array([10.0, 11.0, 12.0]); tab['decDeg'] = np.array([0.1, 0.2, 0.3]); tab['SNR'] = np.array([5.0, 6.0, 7.0])
    tab['template'] = np.array(['fa', 'fb', 'fa'], dtype='U30'); tab['tileName'] = np.array(['T0', 'T0', 'T1'], dtype='U30')
else:
    tab['RADeg'] = np.array([12.0001, 40.0]); tab['decDeg'] = np.array([0.3, -5.0]); tab['SNR'] = np.array([9.0, 4.5])
    tab['template'] = np.array(['fb', 'fb'], dtype='U30'); tab['tileName'] = np.array(['T2', 'T2'], dtype='U30')
for k in ['numSigPix', 'flags', 'galacticLatDeg', 'x', 'y']:
    tab[k] = np.arange(len(tab), dtype=np.float64) + rank
tab['ringFlag'] = np.zeros(len(tab), dtype=bool)
tab['y_c'] = np.arange(len(tab), dtype=np.float64) * 0.5 + rank
allTab = pipelines._gatherCatalogs(tab, cfg)
assert len(allTab) == 5, len(allTab)
assert list(allTab['template']) == ['fa', 'fb', 'fa', 'fb', 'fb']
assert list(allTab['tileName']) == ['T0', 'T0', 'T1', 'T2', 'T2']
assert 'y_c' in allTab.keys() and allTab['y_c'][3] == 1.0
assert allTab['name'][0] == catalogs.makeName(10.0, 0.1)
assert list(allTab['name']) == [catalogs.makeName(r, d) for r, d in zip(allTab['RADeg'], allTab['decDeg'])]
dedup, nDup, names = catalogs.removeDuplicates(allTab)
assert len(dedup) == 4 and nDup == 2 and 9.0 in list(dedup['SNR']) and 7.0 not in list(dedup['SNR'])
# an empty rank still takes part in the exchange
empty = pipelines._gatherCatalogs(catalogs.Table(names=('SNR', 'RADeg', 'decDeg')) if rank == 1 else tab, cfg)
assert len(empty) == 3
# ranks that offer different extra columns (their digests differ in the counts exchange): only the common ones travel
tab2 = tab.copy()
if rank == 1:
    tab2['fixed_SNR'] = np.ones(len(tab2))
mixed = pipelines._gatherCatalogs(tab2, cfg)
assert len(mixed) == 5 and 'fixed_SNR' not in mixed.keys() and 'y_c' in mixed.keys() and mixed['y_c'][3] == 1.0
parts = startUp.assignTiles(cfg.allTileNames, [3, 2, 1], size)
assert sorted(parts[0] + parts[1]) == cfg.allTileNames
dist.barrier()
dist.destroy_process_group()
print("rank %%d ok" %% rank)
'''


def test_catalogue_allgather_gloo_world_size_2(tmp_path):
    """pipelines._gatherCatalogs (the RCCL all-gather that replaces comm.allgather at nemo/pipelines.py:322) under gloo, 2 ranks."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", OMP_NUM_THREADS="1", CUDA_VISIBLE_DEVICES="",
               HIP_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29613", str(script)], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


def test_worker_threads_are_bound_to_the_process_device(monkeypatch):
    """HIP's current device is per host thread and starts at 0: a detection worker thread must be switched to the GPU of
    the process (LOCAL_RANK), never the other way round.  A fake torch.cuda with per-thread device state
    stands in for the GPU here."""
    import threading
    import types
    from concurrent.futures import ThreadPoolExecutor
    from nemo_amd import engine, _lib

    tls = threading.local()

    def cur():
        return getattr(tls, 'dev', 0)

    def set_device(i):
        tls.dev = int(i)
    fake = types.SimpleNamespace(
        cuda=types.SimpleNamespace(is_available=lambda: True, current_device=cur, set_device=set_device, device_count=lambda: 8),
        _C=types.SimpleNamespace(_cuda_getDevice=cur),
        device=lambda kind, idx: types.SimpleNamespace(type=kind, index=idx))
    monkeypatch.setattr(engine, '_torch', fake)
    monkeypatch.setattr(engine, '_device', None)
    monkeypatch.setattr(_lib, 'load', lambda: None)
    assert engine.bind_device(5).index == 5
    assert cur() == 5
    seen = {}

    def work():
        seen['before'] = cur()
        seen['dev'] = engine.require_gpu().index
        seen['after'] = cur()
    with ThreadPoolExecutor(max_workers=1) as pool:
        pool.submit(work).result()
    assert seen == {'before': 0, 'dev': 5, 'after': 5}
    assert engine.require_gpu().index == 5 and cur() == 5      # the main thread was not dragged to device 0
    # the pool the pipeline uses binds its workers the same way
    from nemo_amd import pipelines
    monkeypatch.setattr(pipelines, '_workers', {})
    pool, streams = pipelines._detectionWorkers(1)
    try:
        assert pool.submit(cur).result() == 5
    finally:
        pool.shutdown()


def _dr6_shaped_config():
    """A configuration with the multi-pass structure of examples/ACT-DR6-clusters/DR6ClusterSearch.yml:133-215 (filter sets
    1-6 per-frequency point-source passes, 10 and 11 the cluster passes), over small in-memory maps."""
    from nemo_amd.wcs import CARWCS
    shape = (40, 50)
    wcs = CARWCS.make(shape)
    maps_ = [{'mapData': np.zeros(shape), 'weightsData': np.ones(shape), 'wcs': wcs, 'obsFreqGHz': f, 'units': 'uK',
              'beamFileName': None, 'label': lab} for lab, f in [('f090', 97.8), ('f150', 149.6), ('f220', 216.5)]]
    opts = {}
    for k, lab in [(1, 'f090'), (3, 'f150'), (5, 'f220')]:
        opts[k] = {'mapToUse': lab, 'thresholdSigma': 50.0, 'objIdent': 'ACT-S', 'measureShapes': True, 'ignoreSurveyMask': True}
        opts[k + 1] = {'mapToUse': lab, 'thresholdSigma': 5.0, 'objIdent': 'ACT-S', 'subtractModelFromSets': [k],
                       'addSiphonedFromSets': [k], 'maskAndFillFromSets': [k], 'maskHoleDilationFactor': 3.0, 'saveCatalog': True,
                       'saveFilteredMaps': False, 'fileLabel': 'ptsrc_%s' % lab}
    opts[10] = {'subtractModelFromSets': [2, 4, 6], 'maskAndFillFromSets': [1, 3, 5], 'maskHoleDilationFactor': 3.0}
    opts[11] = {'subtractModelFromSets': [2, 4, 6], 'maskAndFillFromSets': [1, 3, 5], 'maskHoleDilationFactor': 3.0,
                'noiseModelCatalogFromSets': [10]}
    return {'unfilteredMaps': maps_, 'surveyMask': 'mask.fits', 'thresholdSigma': 4.0, 'objIdent': 'ACT-CL', 'catalogCuts': ['SNR > 4'],
            'photFilter': 'Arnaud_M2e14_z0p4', 'filterSetOptions': opts,
            'allFilters': {'class': 'ArnaudModelMatchedFilter',
                           'params': {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 80.0}, 'saveFilteredMaps': False,
                                      'saveRMSMap': False, 'outputUnits': 'yc', 'edgeTrimArcmin': 0.0},
                           'filterSets': [10, 11]},
            'mapFilters': [{'label': 'BeamPass1', 'class': 'BeamMatchedFilter', 'filterSets': [1, 3, 5],
                            'params': {'outputUnits': 'uK', 'noiseParams': {'method': 'model'}}},
                           {'label': 'BeamPass2', 'class': 'BeamMatchedFilter', 'filterSets': [2, 4, 6],
                            'params': {'outputUnits': 'uK', 'noiseParams': {'method': 'dataMap'}}},
                           {'label': 'Arnaud_M1e14_z0p2', 'params': {'M500MSun': 1e14, 'z': 0.2}},
                           {'label': 'Arnaud_M2e14_z0p4', 'params': {'M500MSun': 2e14, 'z': 0.4, 'saveFilteredMaps': True}}],
            'outputDir': None}


def test_filter_sets_of_a_dr6_shaped_config():
    """startUp.NemoConfig.setFilterSet (nemo/startUp.py:751-843) on the multi-pass layout of the DR6 cluster search."""
    from nemo_amd import startUp
    config = startUp.NemoConfig(_dr6_shaped_config(), makeOutputDirs=False, verbose=False)
    assert config.filterSets == [1, 2, 3, 4, 5, 6, 10, 11]
    # the allFilters merge: entries keep what they say themselves, inherit the rest, two dictionary levels deep
    byLabel = {f['label']: f for f in config.parDict['mapFilters']}
    assert byLabel['BeamPass1']['params']['noiseParams'] == {'method': 'model', 'noiseGridArcmin': 80.0}
    assert byLabel['BeamPass1']['class'] == 'BeamMatchedFilter' and byLabel['BeamPass1']['filterSets'] == [1, 3, 5]
    assert byLabel['Arnaud_M1e14_z0p2']['class'] == 'ArnaudModelMatchedFilter' and byLabel['Arnaud_M1e14_z0p2']['filterSets'] == [10, 11]
    assert byLabel['Arnaud_M2e14_z0p4']['params']['saveFilter'] is True          # photFilter forces its filter to be kept
    for setNum in config.filterSets:
        config.filterSetOptions[setNum]['catalog'] = 'catalogue of set %d' % setNum      # what the driver stores per pass

    config.setFilterSet(1)
    assert [f['label'] for f in config.parDict['mapFilters']] == ['BeamPass1']
    assert config.parDict['thresholdSigma'] == 50.0 and config.parDict['objIdent'] == 'ACT-S' and config.parDict['measureShapes'] is True
    assert config.parDict['catalogCuts'] == [] and config.parDict['forcedPhotometryCatalog'] is None
    assert config.parDict['mapFilters'][0]['params']['mapToUse'] == 'f090'
    assert all(m['surveyMask'] is None for m in config.unfilteredMapsDictList)          # ignoreSurveyMask
    assert config.filterSetOptions[1]['addSiphonedFromSets'] is None and config.filterSetOptions[1]['saveCatalog'] is False

    config.setFilterSet(2)
    assert [f['label'] for f in config.parDict['mapFilters']] == ['BeamPass2']
    assert config.parDict['thresholdSigma'] == 5.0
    byMap = {m['label']: m for m in config.unfilteredMapsDictList}
    assert byMap['f090']['subtractModelFromCatalog'] == 'catalogue of set 1' and byMap['f090']['maskAndFillFromCatalog'] == 'catalogue of set 1'
    assert 'subtractModelFromCatalog' not in byMap['f150'] and byMap['f150'].get('maskAndFillFromCatalog') is None
    assert all(m['maskHoleDilationFactor'] == 3.0 and m['surveyMask'] == 'mask.fits' for m in config.unfilteredMapsDictList)
    assert config.parDict['mapFilters'][0]['params']['saveFilteredMaps'] is False

    config.setFilterSet(10)
    assert [f['label'] for f in config.parDict['mapFilters']] == ['Arnaud_M1e14_z0p2', 'Arnaud_M2e14_z0p4']
    assert config.parDict['thresholdSigma'] == 4.0 and config.parDict['objIdent'] == 'ACT-CL'      # back to the run-level values
    byMap = {m['label']: m for m in config.unfilteredMapsDictList}
    for lab, sub, fill in [('f090', 2, 1), ('f150', 4, 3), ('f220', 6, 5)]:
        assert byMap[lab]['subtractModelFromCatalog'] == 'catalogue of set %d' % sub
        assert byMap[lab]['maskAndFillFromCatalog'] == 'catalogue of set %d' % fill
    assert all(f['params'].get('saveFilter', False) is False and f['params']['saveFilteredMaps'] is False for f in config.parDict['mapFilters'])
    assert config.parDict['catalogCuts'] == []

    config.setFilterSet(11)
    phot = [f for f in config.parDict['mapFilters'] if f['label'] == 'Arnaud_M2e14_z0p4'][0]
    assert phot['params']['saveFilter'] is True and phot['params']['saveFilteredMaps'] is True      # the last pass writes
    assert all(f['params']['noiseModelCatalog'] == ['catalogue of set 10'] for f in config.parDict['mapFilters'])
    assert config.parDict['catalogCuts'] == ['SNR > 4']
