"""cProfile of ONE step of bench.py --mode inject after warm-up: where the host spends the wall time.
    python tools/hostprof_inject.py [sync]"""
import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
from nemo_amd import pipelines, catalogs
shape = (480, 3240)
mapsList, wcs, tileDefs = bench.make_workload_tiled(shape, 4, 3, 1234, 'f64')
config = bench.make_config(mapsList, 16, tileDefs, inject=True)
config.parDict['mapFilters'] = [f for f in config.parDict['mapFilters'] if f['label'] == bench.PHOT_FILTER]
if len(sys.argv) > 1 and sys.argv[1] == 'sync':
    config.parDict['asyncDetection'] = False
config._origParDict = config._copyParDict(config.parDict)
real = pipelines.filterMapsAndMakeCatalogs(config, verbose=False, writeAreaMask=True)
am = catalogs.AreaMasks.fromConfig(config)
for _ in range(2):
    bench.run_inject_step(config, real, am); torch.cuda.synchronize()
t0 = time.perf_counter(); bench.run_inject_step(config, real, am); torch.cuda.synchronize(); print("step wall ms", 1e3 * (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable()
bench.run_inject_step(config, real, am); torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(70)
