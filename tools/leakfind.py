"""Which device tensors accumulate over injection-loop steps (debugging aid): live CUDA tensors by shape / dtype before and after."""
import sys, os, gc, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from nemo_amd import pipelines, _lib, catalogs
pipelines.initDistributed(); torch.cuda.set_device(0); _lib.load()
mapsList, wcs, tileDefs = bench.make_workload_tiled((480, 3240), 4, 3, 1234, 'f64')
config = bench.make_config(mapsList, 15, tileDefs, inject=True)
config.rank, config.size = 0, 1
config.parDict['mapFilters'] = [f for f in config.parDict['mapFilters'] if f['label'] == bench.PHOT_FILTER]
config._origParDict = config._copyParDict(config.parDict)
realCatalog = pipelines.filterMapsAndMakeCatalogs(config, verbose=False, writeAreaMask=True)
areaMasks = catalogs.AreaMasks.fromConfig(config)

def census():
    gc.collect()
    c = collections.Counter()
    seen = set()
    for o in gc.get_objects():
        try:
            if isinstance(o, torch.Tensor) and o.is_cuda:
                st = o.untyped_storage()
                key = st.data_ptr()
                if key in seen:
                    continue
                seen.add(key)
                c[(tuple(o.shape), str(o.dtype), st.nbytes())] += 1
        except Exception:
            pass
    return c

for _ in range(3):
    bench.run_inject_step(config, realCatalog, areaMasks)
a = census()
for _ in range(3):
    bench.run_inject_step(config, realCatalog, areaMasks)
b = census()
print("allocated GB", torch.cuda.memory_allocated() / 1e9)
for k in sorted(b, key=lambda k: -(b[k] - a.get(k, 0)) * k[2]):
    d = b[k] - a.get(k, 0)
    if d > 0:
        print(d, k, '%.1f MB' % (d * k[2] / 1e6))
# who refers to one of the grown tensors
