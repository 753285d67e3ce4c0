import sys, os, gc, time
sys.path.insert(0, os.getcwd())
import bench, torch
mapsList, wcs, truth = bench.make_workload((4096, 4096), 1234, 'f64')
config = bench.make_config(mapsList, 15)
bench.run_step(config); torch.cuda.synchronize()
def cb(phase, info):
    if phase == 'start': cb.t = time.perf_counter()
    else: print("  gc gen%d %.2f ms collected %d" % (info['generation'], 1e3 * (time.perf_counter() - cb.t), info['collected']), flush=True)
gc.callbacks.append(cb)
for k in range(9):
    st0 = torch.cuda.memory_stats()
    t0 = time.perf_counter(); bench.run_step(config); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st1 = torch.cuda.memory_stats()
    print("step %d %.2f ms  device_alloc +%d device_free +%d  reserved %.2f GB  allocs +%d" % (k, 1e3 * dt,
          st1.get('num_device_alloc', 0) - st0.get('num_device_alloc', 0), st1.get('num_device_free', 0) - st0.get('num_device_free', 0),
          st1['reserved_bytes.all.current'] / 1e9, st1['allocation.all.allocated'] - st0['allocation.all.allocated']), flush=True)
print("objects tracked", len(gc.get_objects()))
