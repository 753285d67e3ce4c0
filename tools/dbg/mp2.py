import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle'); sys.path.insert(0, '.')
import numpy as np
import test_gpu_multipass as T
from nemo_amd import pipelines, photometry, engine
from nemo_oracle import photometry as ophot
import helpers
mapsL, wcs, src = T._tile()
config = T._config(helpers.product_maps(mapsL))
final = pipelines.filterMapsAndMakeCatalogs(config, verbose=False)
sn = config.stitchedSNMapDict['PRIMARY'].cpu().numpy()
np.save('gpurun_out/mp_sn.npy', sn)
flag = np.zeros(sn.shape, dtype=np.uint8)
fmd = {'SNMap': sn, 'surveyMask': np.ones(sn.shape, dtype=np.uint8), 'wcs': wcs, 'flagMask': flag, 'label': 'x', 'tileName': 'PRIMARY'}
tab = photometry.findObjects(fmd, threshold=4.0, minObjPix=1, rejectBorder=0)
rows = helpers.oracle_find(sn, flag, wcs, threshold=4.0, minObjPix=1)
print(len(tab), len(rows))
gr = np.asarray(tab['ringFlag']); orr = np.array([r['ringFlag'] for r in rows])
print('product ring', gr.astype(int)); print('oracle  ring', orr.astype(int))
# ring components
ids, pos, npix, seg = ophot.get_object_positions(sn, 0, True)
mask = ophot.ring_mask(sn, 0)
eng = engine.TileEngine.get(sn.shape)
d = engine.to_dev(sn)
n, labels, stats = photometry._segment(eng, d, 0, True)
isRing = eng.ring_flags(labels, stats, n, minRingPix=30).cpu().numpy()
lab = labels.cpu().numpy(); st = stats.cpu().numpy()
print('labels equal', np.array_equal(lab, seg), n, len(ids))
for k, r in enumerate(rows):
    if bool(gr[k]) != bool(orr[k]) or True:
        y, x = int(r['y']), int(r['x'])
        c = seg[y, x]
        if c == 0: continue
        i = list(ids).index(c)
        print(k, 'obj', (r['y'], r['x']), 'comp', c, 'npix', npix[i], 'oracle CoM', pos[i], 'own', seg[int(pos[i][0]), int(pos[i][1])] == c,
              '| product CoM', st[c, 2] / st[c, 1], st[c, 3] / st[c, 1], 'count', st[c, 0], 'isRing', isRing[c], 'oracle mask', mask[y, x])
