"""cProfile of ONE timed step of bench.py's workload (after a warm-up step) - where the host spends the wall time."""
import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
shape = (4096, 4096)
mapsList, wcs, truth = bench.make_workload(shape, 1234, 'f64')
config = bench.make_config(mapsList, 15)
if len(sys.argv) > 1 and sys.argv[1] == 'sync':
    config.parDict['asyncDetection'] = False      # everything on the main thread: the profile then covers the detection work too
bench.run_step(config); torch.cuda.synchronize()
t0 = time.perf_counter(); bench.run_step(config); torch.cuda.synchronize(); print("step wall ms", 1e3 * (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable()
bench.run_step(config); torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(70)
