#!/usr/bin/env python3
"""CPU baseline leg of bench.py (TEST / MEASUREMENT INFRASTRUCTURE, never imported by the product).

Times the oracle - the numpy/scipy restatement of nemo's MatchedFilter.buildAndApply (nemo/filters.py:521-777), the
pixel-window undo of filters.filterMaps (:98-101) and photometry.findObjects (nemo/photometry.py:25-210) - on one
filter scale of a size x size float64 crop of bench.py's workload (f090, 2.1' beam, CMB + 30 uK white noise + clusters,
dataMap noise, 40' grid, yc).  The per-mode inverse is the vectorised form; the reference's own Python double loop
(nemo/filters.py:622-627) is ~13 us/pixel slower still.  Prints one JSON line {"seconds": ...}.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1536)
    ap.add_argument("--faithful", action="store_true",
                    help="per-mode inverse by the reference's own Python double loop (nemo/filters.py:622-627)")
    args = ap.parse_args()
    import numpy as np
    import helpers
    from nemo_oracle import filters as ofilt
    shape = (args.size, args.size)
    mapsL, wcs, truth = helpers.make_tile(shape, freqs=(97.8,), fwhm=(2.1,), noise=(30.0,), seed=1234, numSources=3,
                                          kind='clusters')
    params = {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 40.0}, 'outputUnits': 'yc', 'z': 0.4,
              'M500MSun': 2.0e14, 'edgeTrimArcmin': 0.0}
    t0 = time.perf_counter()
    ref, omf = helpers.oracle_matched(mapsL, wcs, params, 'arnaud', faithfulLoop=args.faithful)
    ofilt.undo_pixel_window(ref['data'])
    cat = helpers.oracle_find(ref['SNMap'], np.zeros(shape, dtype=int), wcs, threshold=4.0, minObjPix=1)
    dt = time.perf_counter() - t0
    print(json.dumps({"seconds": dt, "objects": len(cat), "size": args.size, "faithful": bool(args.faithful)}))


if __name__ == "__main__":
    main()
