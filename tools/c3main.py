"""C3-shaped step with the detection threshold raised so that the worker has next to nothing to do: what the enqueueing thread costs alone."""
import os, sys, time
os.environ['NEMO_AMD_TIMING'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from nemo_amd import pipelines, _lib
pipelines.initDistributed(); torch.cuda.set_device(0); _lib.load()
mapsList, wcs, tileDefs = bench.make_workload_tiled((840, 1440), 8, 2, 1234, 'f64')
config = bench.make_config(mapsList, 16, tileDefs)
config.rank, config.size = 0, 1
for thr in (4.0, 400.0, 4.0, 400.0):
    config.parDict['thresholdSigma'] = thr
    for _ in range(3):
        bench.run_step(config)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        bench.run_step(config)
    torch.cuda.synchronize()
    print("threshold %g: %.2f ms per step" % (thr, 1e3 * (time.perf_counter() - t0) / 10), flush=True)
