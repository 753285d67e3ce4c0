"""Device memory in use after 3, 11, 21 and 41 steps of a bench workload (python tools/memchk.py c2|c3|inject): a pass must leave nothing behind."""
import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from nemo_amd import pipelines, _lib
pipelines.initDistributed(); torch.cuda.set_device(0); _lib.load()
mode = sys.argv[1]
if mode == 'c3':
    mapsList, wcs, tileDefs = bench.make_workload_tiled((840, 1440), 8, 2, 1234, 'f64')
    config = bench.make_config(mapsList, 16, tileDefs)
else:
    mapsList, wcs, truth = bench.make_workload((4096, 4096), 1234, 'f64')
    config = bench.make_config(mapsList, 15)
config.rank, config.size = 0, 1
step = lambda: bench.run_step(config)      # noqa: E731
if mode == 'inject':
    from nemo_amd import catalogs
    mapsList, wcs, tileDefs = bench.make_workload_tiled((480, 3240), 4, 3, 1234, 'f64')
    config = bench.make_config(mapsList, 15, tileDefs, inject=True)
    config.rank, config.size = 0, 1
    config.parDict['mapFilters'] = [f for f in config.parDict['mapFilters'] if f['label'] == bench.PHOT_FILTER]
    config._origParDict = config._copyParDict(config.parDict)
    realCatalog = pipelines.filterMapsAndMakeCatalogs(config, verbose=False, writeAreaMask=True)
    areaMasks = catalogs.AreaMasks.fromConfig(config)
    step = lambda: bench.run_inject_step(config, realCatalog, areaMasks)      # noqa: E731
for k in range(41):
    step()
    if k in (2, 10, 20, 40):
        torch.cuda.synchronize()
        a = torch.cuda.memory_allocated() / 1e9
        n = gc.collect()
        import psutil
        print(mode, 'step', k, 'reserved GB %.2f allocated GB %.2f, after gc.collect (%d objects) %.2f, host RSS GB %.2f' % (torch.cuda.memory_reserved() / 1e9, a, n, torch.cuda.memory_allocated() / 1e9, psutil.Process().memory_info().rss / 1e9), flush=True)
