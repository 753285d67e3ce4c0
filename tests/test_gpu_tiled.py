"""The tile-sharded product path end to end on the GPU, with more than one rank and more than one tile: three tiles with
1 degree of overlap cut from one synthetic map (startUp.getTileCoordsDict, nemo/startUp.py:566-661), divided over TWO ranks by
pixel count (MPIEnabled: startUp.assignTiles, replaces nemo/startUp.py:409-446), each rank filtering and searching its tiles
through pipelines.filterMapsAndMakeCatalogs, the per-rank catalogues all-gathered (nemo/pipelines.py:322) and the duplicates of
objects that straddle tile borders removed (nemo/catalogs.py:704-749).  Both ranks drive the one GPU of the box; the exchange
runs under gloo (NEMO_AMD_DIST_BACKEND), the same code path RCCL takes with one GPU per rank.

Checked on rank 0 against the oracle run tile by tile (survey mask zeroed outside each tile's own area, as MapDict.loadTile
does), the reference's merge of the scales ('ref-forced') and the reference's duplicate rule, both by brute force."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
for p in (%(root)r, os.path.join(%(root)r, "oracle"), os.path.join(%(root)r, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.distributed as dist
import helpers
from nemo_amd import pipelines, startUp, signals, engine, catalogs as C

rank, size = pipelines.initDistributed()
assert size == 2 and dist.get_backend() == 'gloo' and torch.cuda.is_available()

SHAPE = (420, 1500)
mapsL, wcs, truth = helpers.make_tile(SHAPE, freqs=(150.0, 98.0), fwhm=(1.4, 2.1), noise=(12.0, 16.0), seed=5, numSources=14,
                                      kind='clusters')
# two bright clusters ON the tile borders (x = 500, 1000): each neighbouring tile detects its own half
for xb, yb in [(500.2, 150.3), (999.6, 300.1)]:
    ra_, dec_ = wcs.pix2wcs(xb, yb)
    geom = signals.paintGeometry(SHAPE, wcs, ra_, dec_)
    for m in mapsL:
        table = signals.templateTables('arnaud', m['beamFileName'], {'z': 0.4, 'M500MSun': 2e14},
                                       amplitude=signals.convertToDeltaT(4e-4, m['obsFreqGHz']))
        m['mapData'] = m['mapData'] + helpers.cpu_paint(geom, table)
ra = [wcs.pix2wcs(x, 0)[0] for x in (0, 500, 1000, 1499)]
dec0, dec1 = wcs.pix2wcs(0, 0)[1], wcs.pix2wcs(0, SHAPE[0] - 1)[1]
tileDefs = [{'tileName': 'T%%d' %% i, 'RADecSection': [ra[i + 1], ra[i], dec0, dec1]} for i in range(3)]
SCALES = [('Arnaud_M2e14_z0p4', 2e14, 0.4), ('Arnaud_M4e14_z0p2', 4e14, 0.2)]
base = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 20.0}, 'outputUnits': 'yc', 'edgeTrimArcmin': 0.0}
filt = [{'label': lab, 'class': 'ArnaudModelMatchedFilter', 'params': dict(base, M500MSun=m, z=z, saveFilteredMaps=(lab == SCALES[0][0]))}
        for lab, m, z in SCALES]
parDict = {'unfilteredMaps': [dict(m) for m in mapsL], 'thresholdSigma': 4.0, 'minObjPix': 1, 'photFilter': SCALES[0][0],
           'mapFilters': filt, 'useTiling': True, 'tileDefinitions': tileDefs, 'tileOverlapDeg': 1.0, 'outputDir': os.environ['NF_TEST_OUT']}
config = startUp.NemoConfig(parDict, makeOutputDirs=False, verbose=False, MPIEnabled=True)
assert config.allTileNames == ['T0', 'T1', 'T2'] and len(config.tileNames) in (1, 2)
mine = list(config.tileNames)
both = [None, None]
dist.all_gather_object(both, mine)
assert sorted(both[0] + both[1]) == ['T0', 'T1', 'T2'] and not set(both[0]) & set(both[1])      # every tile on exactly one rank
cat = pipelines.filterMapsAndMakeCatalogs(config, verbose=False, writeAreaMask=True)
# the tile products of BOTH ranks reach rank 0's files (nemo/pipelines.py:338-396): three extensions, and a stitched S/N map
# whose every tile area holds that tile's map
rmsTab = pipelines.makeRMSTables(config)
assert set(np.unique(rmsTab['tileName'])) == {'T0', 'T1', 'T2'}            # rows of every rank's tiles, on every rank
if rank == 0:
    from nemo_amd import fitsio
    hdus = fitsio.readHDUs(config.selFnDir + "/RMSMap_%%s.fits" %% SCALES[0][0])
    assert sorted(h[0] for h in hdus[1:]) == ['T0', 'T1', 'T2']
    sn, hdr = fitsio.readFITS(config.filteredMapsDir + "/stitched_%%s_SNMap.fits" %% SCALES[0][0])
    assert sn.shape == SHAPE
    for tileName in config.tileNames:
        x0, x1, y0, y1 = config.tileCoordsDict[tileName]['clippedSection']
        ax0, ax1, ay0, ay1 = config.tileCoordsDict[tileName]['areaMaskInClipSection']
        mine = config.stitchedSNMapDict[tileName].cpu().numpy()
        assert np.array_equal(sn[y0 + ay0:y0 + ay1, x0 + ax0:x0 + ax1], mine[ay0:ay1, ax0:ax1])
    for tileName in config.allTileNames:                                  # the other rank's tiles arrived too
        x0, x1, y0, y1 = config.tileCoordsDict[tileName]['clippedSection']
        ax0, ax1, ay0, ay1 = config.tileCoordsDict[tileName]['areaMaskInClipSection']
        assert (sn[y0 + ay0:y0 + ay1, x0 + ax0:x0 + ax1] != 0).mean() > 0.5, tileName
# every rank holds the same merged catalogue
sig = [None, None]
dist.all_gather_object(sig, (len(cat), float(np.sum(cat['RADeg'])), float(np.sum(cat['SNR']))))
assert sig[0] == sig[1], sig
assert set(np.unique(cat['tileName'])) == {'T0', 'T1', 'T2'}

if rank == 0:
    import test_cpu_oracle_golden as G
    perTile = []
    for tileName in config.allTileNames:
        tc = config.tileCoordsDict[tileName]
        x0, x1, y0, y1 = tc['clippedSection']
        ax0, ax1, ay0, ay1 = tc['areaMaskInClipSection']
        sm = np.zeros((y1 - y0, x1 - x0), dtype=np.uint8)
        sm[ay0:ay1, ax0:ax1] = 1                                   # nemo/maps.py:148-153
        sub = []
        for m in mapsL:
            d = dict(m)
            d['mapData'] = np.ascontiguousarray(m['mapData'][y0:y1, x0:x1])
            d['weightsData'] = np.ascontiguousarray(m['weightsData'][y0:y1, x0:x1])
            d['surveyMask'], d['pointSourceMask'] = sm, np.ones(sm.shape, dtype=np.uint8)
            sub.append(d)
        tabs = []
        for lab, mm, z in SCALES:
            ref, omf = helpers.oracle_matched(sub, tc['wcs'], dict(base, M500MSun=mm, z=z), 'arnaud')
            rows = helpers.oracle_find(ref['SNMap'], np.zeros(sm.shape, dtype=int), tc['wcs'], threshold=4.0, minObjPix=1)
            t = C.Table()
            t.setLazyNames('ACT-CL')
            for k in ('RADeg', 'decDeg', 'SNR'):
                t[k] = np.array([r[k] for r in rows], dtype=np.float64)
            t['template'] = np.full(len(rows), lab, dtype='U30')
            t['tileName'] = np.full(len(rows), tileName, dtype='U30')
            tabs.append(t)
        perTile.append(G._brute_forced(tabs))                       # nemo/catalogs.py:319-362
    allTab = C.vstack(perTile)
    ref, nDup = G._brute_remove_duplicates(allTab.copy())           # nemo/catalogs.py:704-749
    assert nDup >= 4, nDup                                          # the border clusters were seen from both sides
    # rows away from the detection threshold must agree one to one
    def solid(t):
        return np.abs(np.asarray(t['SNR']) - 4.0) > 1e-3
    rs, gs = ref[solid(ref)], cat[solid(cat)]
    assert len(rs) == len(gs) and len(rs) > 50, (len(rs), len(gs))
    ir, ig = np.lexsort((rs['decDeg'], rs['RADeg'])), np.lexsort((gs['decDeg'], gs['RADeg']))
    for k in ('RADeg', 'decDeg'):
        assert np.abs(np.asarray(rs[k])[ir] - np.asarray(gs[k])[ig]).max() < 2e-5 / 120.0, k      # 1e-5 px
    # end to end (no staging): the template spectra are complex64 and the filter cube float32 in both paths, computed by
    # different FFTs - the S/N values agree to float32 class, as in tests/test_gpu_filters.py (5e-6 of the map maximum)
    snrDiff = np.abs(np.asarray(gs['SNR'])[ig] / np.asarray(rs['SNR'])[ir] - 1).max()
    assert snrDiff < 2e-5, snrDiff
    assert list(np.asarray(gs['template'])[ig]) == list(np.asarray(rs['template'])[ir])
    assert list(np.asarray(gs['tileName'])[ig]) == list(np.asarray(rs['tileName'])[ir])
    print("tiled ok: %%d tiles on 2 ranks, %%d rows, %%d in duplicate groups, max S/N difference %%.2g" %% (len(config.allTileNames), len(gs), nDup, snrDiff))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_tiled_two_rank_run_matches_the_oracle(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", HSA_ENABLE_IPC_MODE_LEGACY="0",
               NEMO_AMD_DIST_BACKEND="gloo", OMP_NUM_THREADS="4", NF_TEST_OUT=str(tmp_path / "run"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29653", str(script)], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "tiled ok" in r.stdout, r.stdout[-3000:] + r.stderr[-5000:]
