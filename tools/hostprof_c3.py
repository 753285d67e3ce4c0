"""cProfile of ONE step of bench.py's tiled (C3-shaped) workload after warm-up: where the host spends the wall time.
    python tools/hostprof_c3.py [sync]      (sync: detection on the main thread, so the profile covers it too)"""
import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
shape = (840, 1440)
mapsList, wcs, tileDefs = bench.make_workload_tiled(shape, 8, 2, 1234, 'f64')
config = bench.make_config(mapsList, 16, tileDefs)
if len(sys.argv) > 1 and sys.argv[1] == 'sync':
    config.parDict['asyncDetection'] = False
for _ in range(2):
    bench.run_step(config); torch.cuda.synchronize()
t0 = time.perf_counter(); bench.run_step(config); torch.cuda.synchronize(); print("step wall ms", 1e3 * (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable()
bench.run_step(config); torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(60)
st.sort_stats('cumulative').print_stats(90)
