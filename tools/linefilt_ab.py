"""The line filter alone (nf_axis_filter; argv[1] = axis, 0 rows / 1 columns) on a few shapes: a digest of every output and the
time per call - run once with NF_LF_FUSED / NF_LF_FUSED_COLS = 0 and once with 1 (tools/linefilt_ab.sh): the digests must agree,
the times say what the fused stages gain."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nemo_amd import engine
from nemo_amd.engine import ptr, check, stream_ptr, prec_of

axis = int(sys.argv[1]) if len(sys.argv) > 1 else 0
digestOnly = len(sys.argv) > 2 and sys.argv[2] == 'digest'       # (the test's mode: no timing loop)
rng = np.random.default_rng(11)
for (H, W), dt, batch in [((4096, 4096), np.float64, 1), ((840, 1440), np.float64, 8), ((480, 3240), np.float64, 8), ((4096, 4096), np.float32, 1),
                          ((840, 1440), np.float32, 8), ((241, 486), np.float64, 3), ((63, 250), np.float32, 2), ((512, 512), np.float64, 2), ((250, 63), np.float64, 2)]:
    eng = engine.TileEngine.get((H, W))
    x = rng.standard_normal((batch, H, W)).astype(dt)
    x[:, H // 3, :] = 0.0
    x[:, :, W // 5] = 0.0
    m = engine.to_dev(x)
    tabs = eng._axis_window(-1.0, prec_of(m))
    if tabs is None:
        print(H, W, "no axis tables (Bluestein length)")
        continue
    ty = 1 if dt == np.float64 else 2
    out = torch.empty_like(m)
    rows = torch.ones((batch, H), dtype=torch.int32, device=m.device)
    def run(ref):
        check(eng.lib.nf_axis_filter(eng.plan, prec_of(m), axis, ptr(m), ty, ptr(out), ty, batch, ptr(tabs[axis]), ptr(m) if ref else None,
                                     ptr(rows) if (ref and axis == 0) else None, stream_ptr()), "nf_axis_filter")
    digests = []
    for ref in (False, True):
        run(ref)
        torch.cuda.synchronize()
        digests.append(hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12])
    if digestOnly:
        print("%d x %d %s x %d axis %d  %s %s" % (H, W, np.dtype(dt).name, batch, axis, digests[0], digests[1]))
        continue
    timedRef = axis == 0            # (the pipeline's column pass has no zero reference)
    for _ in range(3):
        run(timedRef)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run(timedRef)
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / 20
    gb = batch * H * W * x.itemsize * (3 if timedRef else 2) / 1e9
    print("%5d x %5d %s x %d  %s %s  %7.1f us  %6.0f GB/s" % (H, W, np.dtype(dt).name, batch, digests[0], digests[1], us, gb / us * 1e6))
