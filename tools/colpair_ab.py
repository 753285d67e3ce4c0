"""A/B of the float32 column pass (NF_FFT_COL_PAIR=0/1): checksums of half-stored transforms with and without the fused
filter / window / plane scale / row range, and the time of the pair forward + inverse."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nemo_amd import engine
def sha(x): return hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:10]
for (H, W, B) in [(4096, 4096, 1), (840, 1440, 2), (211, 256, 3), (180, 1031, 1), (1000, 1202, 1)]:
    eng = engine.TileEngine.get((H, W))
    rng = np.random.default_rng(7)
    m = engine.to_dev(rng.standard_normal((B, H, W)).astype(np.float32))
    filt = engine.to_dev(np.abs(rng.standard_normal((B, H, W))).astype(np.float32))
    F = eng.fft(m, unitary=False, half=True)
    a = eng.ifft(F, out='f32', half=True)
    b = eng.ifft(F, out='f32', half=True, filt=filt, sum=B > 1, window=1)
    c = eng.ifft(F, out='f32', half=True, filt=filt, window=1, planeScale=[1.5, 0.25, 3.0][:B])
    d = eng.ifft(F, out='f32', half=True, filt=filt[0], window=1, planeScale=[0.7], rows=(H // 2 - 34, H // 2 + 34)) if B == 1 else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        F2 = eng.fft(m, unitary=False, half=True); a2 = eng.ifft(F2, out='f32', half=True, filt=filt, sum=B > 1, window=1)
    torch.cuda.synchronize(); us = 1e6 / 20 * (time.perf_counter() - t0)
    F3 = F.reshape(-1, H, W)[..., :W // 2 + 1]
    print(H, W, B, "us %.1f" % us, sha(torch.view_as_real(F3.contiguous())), sha(a), sha(b), sha(c),
          sha(d[H // 2 - 34:H // 2 + 34]) if d is not None else '-', flush=True)
