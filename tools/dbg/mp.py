import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import numpy as np
import test_gpu_multipass as T
from nemo_amd import pipelines
import helpers
mapsL, wcs, src = T._tile()
config = T._config(helpers.product_maps(mapsL))
final = pipelines.filterMapsAndMakeCatalogs(config, verbose=False)
for k in config.filterSets:
    c = config.filterSetOptions[k]['catalog']
    print(k, len(c), c.keys())
    for key in ('SNR', 'numSigPix', 'ellipse_A', 'ellipse_B', 'deltaT_c'):
        if key in c.keys():
            print('   ', key, np.asarray(c[key]))
