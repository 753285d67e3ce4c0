"""pipelines._gatherCatalogs on the RCCL backend (backend 'nccl'), one process, world size 1: the collectives of the
catalogue exchange (nemo/pipelines.py:322) run on device tensors exactly as they do with more ranks.  The logic for
several ranks is covered under gloo in tests/test_cpu_boundary.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import torch.distributed as dist
from nemo_amd import pipelines, catalogs
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
class Cfg(object):
    pass
cfg = Cfg()
cfg.size, cfg.rank = 1, 0
cfg.allTileNames = ['T0', 'T1']
cfg.parDict = {'mapFilters': [{'label': 'fa'}, {'label': 'fb'}], 'objIdent': 'ACT-CL', 'longNames': False}
tab = catalogs.Table()
tab.setLazyNames('ACT-CL')
tab['RADeg'] = np.array([10.0, 11.0, 12.0]); tab['decDeg'] = np.array([0.1, -0.2, 0.3]); tab['SNR'] = np.array([5.0, 6.0, 7.0])
tab['template'] = np.array(['fa', 'fb', 'fa'], dtype='U30'); tab['tileName'] = np.array(['T0', 'T0', 'T1'], dtype='U30')
for k in ['numSigPix', 'flags', 'galacticLatDeg', 'x', 'y']:
    tab[k] = np.arange(3, dtype=np.float64)
tab['ringFlag'] = np.array([False, True, False])
tab['y_c'] = np.array([1.5, 2.5, 3.5])
out = pipelines._gatherCatalogs(tab, cfg)
assert len(out) == 3
for k in ['RADeg', 'decDeg', 'SNR', 'y_c', 'x', 'y']:
    assert np.array_equal(np.asarray(out[k]), np.asarray(tab[k])), k
assert list(out['ringFlag']) == [False, True, False]
assert list(out['template']) == ['fa', 'fb', 'fa'] and list(out['tileName']) == ['T0', 'T0', 'T1']
assert list(out['name']) == [catalogs.makeName(r, d) for r, d in zip(tab['RADeg'], tab['decDeg'])]
empty = pipelines._gatherCatalogs(catalogs.Table(names=('SNR', 'RADeg', 'decDeg')), cfg)
assert len(empty) == 0
# the multi-rank branch of bench.run_step on a real pipeline catalogue (small tile, 6 scales)
sys.path.insert(0, %(root)r)
import bench
mapsList, wcs, truth = bench.make_workload((1024, 1024), 5, 'f64')
config = bench.make_config(mapsList, 6)
config.rank, config.size = 0, 1
cat = pipelines.filterMapsAndMakeCatalogs(config, verbose=False)
assert len(cat) > 0
allCat = pipelines._gatherCatalogs(cat, config)
assert len(allCat) == len(cat)
for k in ['RADeg', 'decDeg', 'SNR']:
    assert np.array_equal(np.asarray(allCat[k]), np.asarray(cat[k])), k
assert list(allCat['name']) == list(cat['name']) and list(allCat['template']) == list(cat['template'])
assert list(allCat['tileName']) == list(cat['tileName'])
dedup, nDup, names = catalogs.removeDuplicates(allCat)
assert len(dedup) <= len(allCat)
dist.destroy_process_group()
print("gather ok")
'''


@pytest.mark.gpu
def test_catalogue_gather_on_rccl_world_size_1(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "gather ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
