#!/usr/bin/env python3
"""Headline benchmark: filtered Mpix/s of the map-filtering + detection hot path (BASELINE.json `metric`).

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): one 4096 x 4096 f090 tile per GPU, float64 maps,
ArnaudModelMatchedFilter at 15 theta500 scales (examples/ACT-DR5-clusters/DR5ClusterSearch.yml:95-127, photFilter first),
noise method dataMap, 40' noise grid, outputUnits yc, pixel-window undo, then photometry.findObjects at 4 sigma on every
scale.  A "step" is one pass of that whole path over the tile (data FFT and noise covariance once, then the 15 scales);
Mpix counts H*W*N*S per step.  Inputs are synthetic (seeded CMB + white noise + 50 painted GNFW clusters) and resident in
HBM before the timed region starts.  N GPUs = N tiles (weak scaling; tiles are independent, no data-path collective).

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` is for the kernel with the largest share of device time, timed with HIP
events inside the library on the stream it runs on (nf_profile_*); `cpu_baseline` is the oracle (numpy/scipy restatement
of the reference algorithm) timed on the host cores, on a bounded sample, before the GPU is touched.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md)

SCALES = [(m, z) for z in (0.2, 0.4, 0.8, 1.2) for m in (1.0e14, 2.0e14, 4.0e14, 8.0e14)]      # the DR5 bank (16; C2 uses 15)
FREQS = [(97.8, 2.1, 30.0, 'f090'), (149.6, 1.4, 20.0, 'f150'), (225.0, 1.0, 60.0, 'f220')]  # GHz, beam FWHM', noise uK/px
PHOT_FILTER = "Arnaud_M2e14_z0p4"


def filter_list(numScales):
    out = []
    for m, z in SCALES[:numScales]:
        label = "Arnaud_M%de14_z%s" % (int(round(m / 1e14)), ("%.1f" % z).replace(".", "p"))
        out.append({'label': label, 'class': 'ArnaudModelMatchedFilter',
                    'params': {'noiseParams': {'method': 'dataMap', 'noiseGridArcmin': 40.0}, 'outputUnits': 'yc',
                               'edgeTrimArcmin': 0.0, 'M500MSun': m, 'z': z, 'saveFilteredMaps': False,
                               'saveRMSMap': False, 'savePlots': False, 'saveDS9Regions': False}})
    # photFilter first (nemo/pipelines.py:161-173)
    out.sort(key=lambda f: 0 if f['label'] == PHOT_FILTER else 1)
    return out


def make_workload(shape, seed, dtype):
    """Synthetic tile resident in HBM: numpy CMB + noise realisation, clusters painted on the GPU."""
    import numpy as np
    from nemo_amd import signals, simgen, engine, maps as pmaps
    beam = signals.BeamProfile(FWHMArcmin=2.1)
    mapsL, wcs, truth = simgen.makeSimTile(shape, [97.8], [beam], [30.0], seed=seed, numSources=0, painter=None,
                                           dtype=np.float64)
    t = engine.torch()
    eng = engine.TileEngine.get(shape)
    m = mapsL[0]
    data = engine.to_dev(m['mapData'])
    rng = np.random.default_rng(seed + 7)
    bank = [(mm, zz) for mm, zz in SCALES[:15]]
    for k in range(50):
        y = rng.uniform(60, shape[0] - 60)
        x = rng.uniform(60, shape[1] - 60)
        ra, dec = wcs.pix2wcs(x, y)
        y0 = rng.uniform(0.5e-4, 5e-4)
        mm, zz = bank[k % len(bank)]
        geom = signals.paintGeometry(shape, wcs, ra, dec)
        table = signals.templateTables('arnaud', beam, {'z': zz, 'M500MSun': mm},
                                       amplitude=signals.convertToDeltaT(y0, 97.8))
        data += eng.paint(geom, table).to(t.float64)
        truth.append({'RADeg': ra, 'decDeg': dec, 'y0': y0})
    tdt = t.float64 if dtype == 'f64' else t.float32
    md = pmaps.MapDict({'mapData': data.to(tdt), 'weightsData': engine.to_dev(m['weightsData']).to(tdt),
                        'obsFreqGHz': 97.8, 'units': 'uK', 'beamFileName': beam, 'label': 'f090', 'wcs': wcs,
                        'surveyMask': t.ones(shape, dtype=t.uint8, device=data.device),
                        'pointSourceMask': t.ones(shape, dtype=t.uint8, device=data.device)})
    return [md], wcs, truth


OVERLAP_PIX = 120       # tileOverlapDeg 1.0 at 0.5' pixels (examples/ACT-DR5-clusters/DR5ClusterSearch.yml:141-147)


def make_workload_tiled(tileShape, tilesPerGPU, numFreqs, seed, dtype):
    """C3-shaped workload: `tilesPerGPU` tiles of `tileShape` pixels INCLUDING the 1 degree overlap, cut side by
    side from one synthetic strip per frequency (common CMB, white noise, painted clusters), resident in HBM.  Returns the
    MapDicts, the WCS and the tile definitions (the cores; startUp adds the overlap back)."""
    import numpy as np
    from nemo_amd import signals, simgen, engine, maps as pmaps
    H, W = tileShape
    coreH, coreW = H - 2 * OVERLAP_PIX, W - 2 * OVERLAP_PIX
    if coreH < 64 or coreW < 64:
        raise SystemExit("--tile must leave a core beyond the 2 x %d overlap pixels" % OVERLAP_PIX)
    # (one spare pixel all round: the overlap is only added where it stays strictly inside the map, nemo/startUp.py:620-657)
    M = OVERLAP_PIX + 1
    big = (H + 2, tilesPerGPU * coreW + 2 * M)
    freqs = FREQS[:numFreqs]
    beams = [signals.BeamProfile(FWHMArcmin=f[1]) for f in freqs]
    eng = engine.TileEngine.get(big)
    mapsL, wcs, truth = simgen.makeSimTile(big, [f[0] for f in freqs], beams, [f[2] for f in freqs], seed=seed,
                                           numSources=6 * tilesPerGPU, kind='clusters', dtype=np.float64,
                                           painter=lambda geom, table: eng.paint(geom, table).cpu().numpy(), borderPix=OVERLAP_PIX)
    t = engine.torch()
    tdt = t.float64 if dtype == 'f64' else t.float32
    out = []
    for m, f in zip(mapsL, freqs):
        data = engine.to_dev(m['mapData']).to(tdt)
        out.append(pmaps.MapDict({'mapData': data, 'weightsData': engine.to_dev(m['weightsData']).to(tdt), 'obsFreqGHz': f[0],
                                  'units': 'uK', 'beamFileName': m['beamFileName'], 'label': f[3], 'wcs': wcs,
                                  'surveyMask': t.ones(big, dtype=t.uint8, device=data.device),
                                  'pointSourceMask': t.ones(big, dtype=t.uint8, device=data.device)}))
    tileDefs = []
    for k in range(tilesPerGPU):
        x0, x1 = M + k * coreW, M + (k + 1) * coreW
        raA, decA = wcs.pix2wcs(x0, M)
        raB, decB = wcs.pix2wcs(x1, M + coreH)
        tileDefs.append({'tileName': 'T%02d' % k, 'RADecSection': [min(raA, raB), max(raA, raB), min(decA, decB), max(decA, decB)]})
    return out, wcs, tileDefs


INJECTION_MODELS = [{'redshift': 0.8, 'M500': 2.0e+14}, {'redshift': 0.4, 'M500': 2.0e+14}, {'redshift': 0.2, 'M500': 2.0e+14},
                    {'redshift': 0.1, 'M500': 2.0e+14}]          # examples/ACT-DR6-clusters/DR6ClusterSearch.yml:221-229


def make_config(mapsList, numScales, tileDefs=None, inject=False):
    """The DR5 cluster-search configuration (examples/ACT-DR5-clusters/DR5ClusterSearch.yml:26-38,82-129) over in-memory maps;
    ``inject``: with the source-injection keys of the DR6 configuration (one iteration per call of maps.sourceInjectionTest)."""
    from nemo_amd import startUp
    parDict = {'unfilteredMaps': [dict(m) for m in mapsList], 'thresholdSigma': 4.0, 'minObjPix': 1, 'findCenterOfMass': True,
               'useInterpolator': True, 'optimalCatalogMethod': 'ref-forced', 'rejectBorder': 0, 'objIdent': 'ACT-CL',
               'longNames': False, 'photFilter': PHOT_FILTER, 'mapFilters': filter_list(numScales), 'useTiling': False,
               'outputDir': None}
    if inject:
        parDict.update({'sourceInjectionIterations': 1, 'sourcesPerTile': 50, 'sourceInjectionDistribution': 'linear',
                        'sourceInjectionAmplitudeRange': [0.001, 10],
                        'sourceInjectionModels': [dict(m) for m in INJECTION_MODELS]})
    if tileDefs is not None:
        parDict.update({'useTiling': True, 'tileDefinitions': tileDefs, 'tileOverlapDeg': 1.0})
    return startUp.NemoConfig(parDict, makeOutputDirs=False, verbose=False)


def run_step(config):
    """One pass of the hot path over this rank's tile(s): pipelines.filterMapsAndMakeCatalogs = for every filter scale
    filters.filterMaps (+ pixel-window undo), photometry.findObjects, photometry.measureFluxes, then the optimal catalogue."""
    from nemo_amd import pipelines, catalogs
    cat = pipelines.filterMapsAndMakeCatalogs(config, verbose=False)
    if config.size > 1:
        # the one exchange of the path (nemo/pipelines.py:322): every rank ends with the catalogue of all tiles - an RCCL
        # all-gather of fixed-width rows over xGMI, then the duplicate removal of tile overlaps (:335)
        cat = pipelines._gatherCatalogs(cat, config)
        if len(cat) > 0:
            cat, numDup, names = catalogs.removeDuplicates(cat)
    if len(cat) > 0:
        cat['name']          # names are formatted lazily: make them part of the timed work
    return cat


def run_inject_step(config, realCatalog, areaMasks):
    """One iteration of the injection / recovery loop (maps.sourceInjectionTest, nemo/maps.py:1856-2153) over this rank's
    tiles: for each of the four DR6 injection models a catalogue is drawn and painted into the maps, the maps are filtered
    again with the cached reference filter and noise map, objects are found, measured and matched to what went in."""
    from nemo_amd import maps as pmaps
    keep = config.size
    config.size = 1            # (every rank injects into its own tiles: nothing is exchanged inside the loop)
    try:
        return pmaps.sourceInjectionTest(config, realCatalog=realCatalog, areaMasks=areaMasks, verbose=False)
    finally:
        config.size = keep


def _cpu_run(procs, size, faithful):
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--size", str(size)] + (["--faithful"] if faithful else [])
    ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           env=dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1"))
          for _ in range(procs)]
    secs = []
    for p in ps:
        out, err = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("cpu baseline failed: %s" % err[-2000:])
        secs.append(json.loads(out.strip().splitlines()[-1])['seconds'])
    return max(secs)


def cpu_baseline(procs, size):
    """Oracle timed on the host: `procs` processes, each one scale of a size x size crop of the same workload - with the
    vectorised per-mode inverse (the headline figure) and, on a smaller crop, with the reference's own Python loop over
    Fourier modes (nemo/filters.py:622-627), which is how the reference actually runs (mode i)."""
    cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--size", str(size)]
    t0 = time.time()
    ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           env=dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1"))
          for _ in range(procs)]
    secs = []
    for p in ps:
        out, err = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("cpu baseline failed: %s" % err[-2000:])
        secs.append(json.loads(out.strip().splitlines()[-1])['seconds'])
    wall = max(secs)
    fsize = max(256, size // 4)
    fwall = _cpu_run(procs, fsize, True)
    return {'value': round(procs * size * size / 1e6 / wall, 4), 'unit': 'Mpix/s', 'cores': procs,
            'host_cores': os.cpu_count(), 'kind': 'port',
            'sample': "%d single-threaded process(es) (the box's CPU share for one GPU; os.cpu_count() counts the whole machine), each 1 scale (Arnaud M2e14 z0.4) of a %dx%d f64 crop of the same workload: "
                      "oracle MatchedFilter.buildAndApply + pixel-window undo + find_objects (vectorised inverse); "
                      "%.1f s per process" % (procs, size, size, wall),
            'reference_faithful': {'value': round(procs * fsize * fsize / 1e6 / fwall, 4), 'unit': 'Mpix/s', 'cores': procs,
                                   'sample': "same, %dx%d crop, per-mode inverse by the reference's Python loop "
                                             "(nemo/filters.py:622-627); %.1f s per process" % (fsize, fsize, fwall)},
            'launch_wall_s': round(time.time() - t0, 2)}


def pmc_traffic(tag, shape, numScales, dtype):
    """HBM bytes per launch (GB) of the kernel behind `tag` from the committed rocprofv3 PMC passes over this same command
    (profiles/r03_pmc_traffic.json: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE).
    Counters cannot be read from inside a run, so this is the profiled figure for the headline workload only - None for
    any other size / scale count / dtype, or when the file is absent."""
    path = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
    if tuple(shape) != (4096, 4096) or numScales != 15 or dtype != "f64" or not os.path.exists(path):
        return None
    with open(path) as f:
        tab = json.load(f).get("kernels", {})
    return tab[tag]["traffic_GB"] if tag in tab else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=4096, help="tile side in pixels (headline: 4096)")
    ap.add_argument("--scales", type=int, default=15)
    ap.add_argument("--tile", default=None, help="HxW in pixels, overlap included: the tiled (C3-shaped) workload, e.g. 840x1440")
    ap.add_argument("--tiles-per-gpu", type=int, default=1, help="tiles each GPU filters per step (with --tile)")
    ap.add_argument("--freqs", type=int, default=1, choices=[1, 2, 3], help="frequency maps per tile (with --tile)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--mode", default="search", choices=["search", "inject"],
                    help="search: the cluster search (headline).  inject: config C5's injection / recovery loop - with --tile, "
                         "e.g. --mode inject --tile 480x3240 --freqs 3 --tiles-per-gpu 4 (a step = 4 injection models x the tiles)")
    ap.add_argument("--cpu-procs", type=int, default=-1, help="processes for the CPU baseline (0 = skip; -1 = auto)")
    ap.add_argument("--cpu-size", type=int, default=2048)
    ap.add_argument("--no-profile", action="store_true", help="do not record per-entry-point HIP events")
    ap.add_argument("--dump-catalogs", default=None, metavar="DIR",
                    help="after the timed steps, write each rank's own catalogue and the gathered one (rank 0) as .npz into DIR "
                         "(the multi-rank rehearsal test reads them)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # ---- CPU baseline first: nothing has touched the GPU yet, so child processes are plain fork/exec -------------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_procs != 0:
        procs = args.cpu_procs if args.cpu_procs > 0 else max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", "8"))))
        cpu = cpu_baseline(procs, args.cpu_size)

    import torch
    from nemo_amd import _lib, engine, pipelines
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    pipelines.initDistributed()
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % torch.cuda.device_count())
    lib = _lib.load()
    tiled = args.tile is not None
    inject = args.mode == "inject"
    if inject and not tiled:
        raise SystemExit("--mode inject needs --tile HxW (e.g. 480x3240)")
    if tiled:
        shape = tuple(int(v) for v in args.tile.lower().split('x'))
        mapsList, wcs, tileDefs = make_workload_tiled(shape, args.tiles_per_gpu, args.freqs, 1234 + rank, args.dtype)
        config = make_config(mapsList, args.scales, tileDefs, inject=inject)
    else:
        shape = (args.size, args.size)
        mapsList, wcs, truth = make_workload(shape, 1234 + rank, args.dtype)
        config = make_config(mapsList, args.scales)
    # tiles are already one per rank (weak scaling): config.MPIEnabled stays False (no tile division inside the pipeline);
    # run_step() does the catalogue all-gather of the path itself when there are several ranks
    config.rank, config.size = rank, world
    N, S = len(mapsList), len(config.parDict['mapFilters'])
    T = len(config.tileNames)
    step = lambda: run_step(config)      # noqa: E731
    if inject:
        # set-up, untimed: the search pass of the reference filter alone leaves what the loop reuses - the filter cube, the noise
        # map and the area mask of every tile in HBM - and the catalogue of real objects
        from nemo_amd import catalogs
        config.parDict['mapFilters'] = [f for f in config.parDict['mapFilters'] if f['label'] == PHOT_FILTER]
        config._origParDict = config._copyParDict(config.parDict)
        realCatalog = pipelines.filterMapsAndMakeCatalogs(config, verbose=False, writeAreaMask=True)
        areaMasks = catalogs.AreaMasks.fromConfig(config)
        S = len(INJECTION_MODELS)         # re-filters per tile and step
        step = lambda: run_inject_step(config, realCatalog, areaMasks)      # noqa: E731
    if tiled:
        for tn in config.tileNames:
            cs = config.tileCoordsDict[tn]['clippedSection']
            assert (cs[3] - cs[2], cs[1] - cs[0]) == shape, (tn, cs)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import ctypes

    def read_profile():
        tab = {}
        for tag in range(lib.nf_profile_num_tags()):
            n, ms, by = ctypes.c_longlong(0), ctypes.c_double(0), ctypes.c_double(0)
            _lib.check(lib.nf_profile_read(tag, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)), "nf_profile_read")
            if n.value > 0:
                tab[lib.nf_profile_tag_name(tag).decode()] = {
                    'tag': tag, 'calls': n.value, 'ms_total': round(ms.value, 3), 'avg_us': round(1e3 * ms.value / n.value, 2),
                    'alg_GB_per_call': round(by.value / n.value / 1e9, 6),
                    'GBps': round(by.value / 1e9 / (ms.value / 1e3), 1) if ms.value > 0 else None}
        return tab

    # Warm-up, then one untimed SURVEY step in which every C-ABI entry point records a HIP event pair: that gives the
    # per-entry-point table and names the dominant one.  In the timed region only the dominant entry point records events:
    # 1200 event pairs per step cost ~2 ms of queue time, which would be measured as lost throughput.
    survey = {}
    firstStepMs = None
    for k in range(args.warmup):
        if k == 0:
            barrier()
            tf = time.perf_counter()
        cat = step()
        if k == 0:
            # the very first pass: FFT plans, radial tables of the templates, workspaces and the lane streams are made here
            torch.cuda.synchronize()
            firstStepMs = round(1e3 * (time.perf_counter() - tf), 2)
    if not args.no_profile:
        # the SURVEY step: one more untimed step after the warm-up in which every entry point records its event pair.  It runs
        # one scale at a time on one stream, detection included (scaleLanes / scaleGroup 1, asyncDetection off), so that an
        # event pair brackets that entry point's kernels alone; the timed steps run the product's default schedule, in which
        # two scales and the detection stream share the chip.
        barrier()
        lib.nf_profile_select(ctypes.c_ulonglong(0xFFFFFFFFFFFFFFFF))
        lib.nf_profile_enable(1)
        keepSched = {k: config.parDict.get(k, None) for k in ('scaleLanes', 'scaleGroup', 'asyncDetection')}
        config.parDict['scaleLanes'], config.parDict['scaleGroup'] = 1, (1 if not tiled else 16)
        config.parDict['asyncDetection'] = False
        if inject:
            config._origParDict['asyncDetection'] = False
        cat = step()
        for k, v in keepSched.items():
            if v is None:
                config.parDict.pop(k, None)
            else:
                config.parDict[k] = v
        if inject:
            config._origParDict.pop('asyncDetection', None)
        barrier()
        survey = read_profile()
        lib.nf_profile_enable(0)
    barrier()
    dom = max(survey, key=lambda k: survey[k]['ms_total']) if survey else None
    if not args.no_profile:
        if dom is not None:
            lib.nf_profile_select(ctypes.c_ulonglong(1 << survey[dom]['tag']))
        lib.nf_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cat = step()
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if (world == 1 or dist.get_backend() == 'nccl') else 'cpu')
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if args.dump_catalogs is not None and not inject:
        import numpy as np
        os.makedirs(args.dump_catalogs, exist_ok=True)
        own = pipelines.filterMapsAndMakeCatalogs(config, verbose=False)        # this rank's tiles only, nothing exchanged
        both = run_step(config)                                                  # the step as timed: gathered, duplicates removed
        cols = ('RADeg', 'decDeg', 'SNR')
        np.savez(os.path.join(args.dump_catalogs, "rank%d.npz" % rank), **{"own_" + k: np.asarray(own[k], dtype=np.float64) for k in cols},
                 **{"all_" + k: np.asarray(both[k], dtype=np.float64) for k in cols},
                 own_tileName=np.asarray(own['tileName']).astype('U30'), all_tileName=np.asarray(both['tileName']).astype('U30'))
        barrier()

    # ---- device time of the dominant entry point over the timed region, from the library's HIP events -------------
    timed = {}
    if not args.no_profile:
        timed = read_profile()
        lib.nf_profile_enable(0)
        lib.nf_profile_select(ctypes.c_ulonglong(0xFFFFFFFFFFFFFFFF))
    surveySteps = 1
    kernels = {k: {kk: vv for kk, vv in v.items() if kk != 'tag'} for k, v in survey.items()}

    if rank == 0:
        mpix = shape[0] * shape[1] * N * S * T * args.steps * world / 1e6
        out = {"metric": "filtered Mpix/sec (all scales, all freqs)", "value": round(mpix / elapsed, 2), "unit": "Mpix/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * elapsed / args.steps, 2), "first_step_ms": firstStepMs, "higher_is_better": True,
               "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": ("C3-shaped: %d tiles of %dx%d (1 deg overlap included) per GPU, %d frequency map(s), " % (T, shape[0], shape[1], N)
                                       if tiled else "C2: one %dx%d f090 tile per GPU, " % (shape[0], shape[1])) +
                                      "ArnaudModelMatchedFilter x %d theta500 scales (DR5 bank), dataMap noise, 40' grid, yc, "
                                      "pixel-window undo, findObjects 4 sigma" % (S),
                          "tile": list(shape), "frequencies": N, "scales": S, "tiles_per_gpu": T,
                          "objects_in_optimal_catalog": int(len(cat))}}
        if inject:
            out["config"]["workload"] = ("C5 injection / recovery loop: %d tiles of %dx%d (1 deg overlap included) per GPU, %d frequency "
                                         "map(s); a step = maps.sourceInjectionTest over %d injection models x 1 iteration: draw 50 "
                                         "clusters per tile, paint them into the maps, re-filter with the cached reference filter "
                                         "and noise map (no filter build, no noise map), pixel-window undo, findObjects 4 sigma, "
                                         "measureFluxes, cross-match" % (T, shape[0], shape[1], N, S))
            out["config"]["scales"] = 1
            out["config"]["injection_models"] = S
            out["config"].pop("objects_in_optimal_catalog")
            out["config"]["recovered_rows_last_step"] = int(len(cat))
            out["refilters_per_s"] = round(S * T * args.steps * world / elapsed, 2)
        if kernels:
            if dom is None:
                dom = max(kernels, key=lambda k: kernels[k]['ms_total'])
            devMs = sum(k['ms_total'] for k in kernels.values())
            algGB = sum(k['alg_GB_per_call'] * k['calls'] for k in kernels.values())
            d = timed.get(dom, kernels[dom])      # the dominant entry point, timed over the K timed steps
            lanes, group = pipelines._scaleGrouping(config, config.tileNames[0], S)
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": d['GBps'], "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(d['GBps'] / HBM_PEAK_GBS, 4),
                               "traffic": pmc_traffic(dom, shape, S, args.dtype),
                               "traffic_unit": "GB per launch (rocprofv3 PMC, profiles/r03_pmc_traffic.json)",
                               "avg_us": d['avg_us'], "launches_timed": d['calls'], "alg_GB_per_launch": d['alg_GB_per_call'],
                               "share_of_device_time": round(kernels[dom]['ms_total'] / devMs, 3),
                               "lanes": lanes,
                               "solo": {"avg_us": kernels[dom]['avg_us'], "achieved": kernels[dom]['GBps'],
                                        "frac": round(kernels[dom]['GBps'] / HBM_PEAK_GBS, 4)},
                               "note": "achieved / avg_us: this entry point's launches in the timed steps, where %d filter scale(s) run "
                                       "side by side on %d HIP stream(s) - a launch then shares the chip's bandwidth with the other "
                                       "lane's kernel and with the detection stream, so its duration is not its own; 'solo': the same "
                                       "launches in the survey step (one scale at a time, one stream).  The chip-wide figure is "
                                       "pipeline.hbm_frac_wall." % (lanes, lanes)}
            out["pipeline"] = {"device_ms_per_step": round(devMs / surveySteps, 2),
                               "alg_GB_per_step": round(algGB / surveySteps, 3),
                               "hbm_frac_device_time": round(algGB / (devMs / 1e3) / HBM_PEAK_GBS, 4),
                               "hbm_frac_wall": round(algGB / surveySteps / (elapsed / args.steps) / HBM_PEAK_GBS, 4),
                               "note": "per-entry-point table from one untimed survey step after the warm-up, run one scale at a time "
                                       "on one stream, detection included (all entry points record events there; in the timed region only the dominant "
                                       "one does); hbm_frac_wall = algorithmic GB per step / timed wall per step / peak"}
            # SURVEY.md 8(d)'s own accounting of the step (full complex transforms, every stage fused perfectly): per pixel, once
            # per tile N(r + 3c) + Nc + N(N+1)/2 * 5r, per scale N(4 + 3c) + N(N+1)/2 * r + Nc + 4N + 2[N(c + 4) + 2c] + r + 4 + 33 +
            # (8 + 6c) + 12 with r / c the bytes of a real / complex number of the map precision (fp64 N=1: 112 + 345 per scale)
            r_, c_ = (8.0, 16.0) if args.dtype == 'f64' else (4.0, 8.0)
            pairs = N * (N + 1) / 2.0
            once = N * (r_ + 3 * c_) + N * c_ + pairs * 5 * r_
            perScale = (N * (4 + 3 * c_) + pairs * r_ + N * c_ + 4 * N + 2 * (N * (c_ + 4) + 2 * c_) + r_ + 4 + 33 + (8 + 6 * c_) + 12)
            surveyGB = shape[0] * shape[1] * T * (once + S * perScale) / 1e9
            if not inject:
                out["pipeline"]["survey_8d"] = {"alg_GB_per_step": round(surveyGB, 2),
                                                "achieved": round(surveyGB / (elapsed / args.steps), 1), "unit": "GB/s",
                                                "frac": round(surveyGB / (elapsed / args.steps) / HBM_PEAK_GBS, 4),
                                                "note": "SURVEY.md 8(d)'s accounting (full complex transforms, perfect fusion: %.0f B per pixel "
                                                        "once + %.0f B per pixel and scale) over the timed wall time; the half-stored "
                                                        "transforms this build runs move fewer bytes (alg_GB_per_step above)" % (once, perScale)}
            # the quantity the north star sets its target on - the FFT + filter build / apply path - as one fixed block: every
            # transform entry point, the filter build and the two axis filters of the pixel-window undo, from the survey step
            fa = {k: v for k, v in kernels.items() if k.startswith(('fft_', 'axis_filter', 'build_filter', 'apply_filter', 'window_mul',
                                                                    'symmetrise', 'cross_power'))}
            if fa:
                faMs = sum(v['ms_total'] for v in fa.values())
                faGB = sum(v['alg_GB_per_call'] * v['calls'] for v in fa.values())
                out["fft_apply"] = {"entry_points": sorted(fa), "alg_GB_per_step": round(faGB, 3), "device_ms_per_step": round(faMs, 3),
                                    "achieved": round(faGB / (faMs / 1e3), 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                    "frac": round(faGB / (faMs / 1e3) / HBM_PEAK_GBS, 4),
                                    "share_of_device_time": round(faMs / devMs, 3),
                                    "note": "solo durations (survey step, one stream): algorithmic bytes of these entry points / their summed "
                                            "HIP-event time / 8 TB/s"}
            out["kernels"] = kernels
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
