"""cProfile of ONE step of bench.py's default (C2) workload after warm-up: where the host spends the wall time.
    python tools/hostprof_c2.py [sync]      (sync: detection on the main thread, so the profile covers it too)"""
import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
mapsList, wcs = bench.make_workload((4096, 4096), 1234, 'f64')[:2]
config = bench.make_config(mapsList, 15)
if len(sys.argv) > 1 and sys.argv[1] == 'sync':
    config.parDict['asyncDetection'] = False
for _ in range(3):
    bench.run_step(config); torch.cuda.synchronize()
t0 = time.perf_counter(); bench.run_step(config); torch.cuda.synchronize(); print("step wall ms", 1e3 * (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable()
bench.run_step(config); torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(50)
st.sort_stats('cumulative').print_stats(90)
