"""nf_xmatch_self_static - the one entry point that allocates nothing (callers under HIP graph capture, serving loops): same
kernels as nf_xmatch_self, with the per-slice partial results in a file-scope device array instead of a caller workspace.
Called through the C ABI and compared, bit for bit, with nf_xmatch_self on two calls in a row, the second with a smaller
input (the array keeps the first call's contents)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from nemo_amd import _lib  # noqa: E402


def _catalogue(n, seed):
    rng = np.random.default_rng(seed)
    ra = rng.uniform(10.0, 14.0, n)
    dec = rng.uniform(-2.0, 2.0, n)
    m = n // 3                      # a third of the rows sit within the match radius of another row
    ra[n - m:] = ra[:m] + rng.normal(0, 0.004, m)
    dec[n - m:] = dec[:m] + rng.normal(0, 0.004, m)
    snr = np.round(rng.uniform(4, 30, n), 1)
    r, d = np.radians(ra), np.radians(dec)
    unit = np.stack([np.cos(d) * np.cos(r), np.cos(d) * np.sin(r), np.sin(d)], axis=1)
    return np.ascontiguousarray(unit), snr


def _run(lib, unit, snr, static):
    dev = torch.device('cuda', torch.cuda.current_device())
    n = len(snr)
    u_d = torch.from_numpy(unit).to(dev)
    s_d = torch.from_numpy(snr).to(dev)
    count = torch.full((n,), -7, dtype=torch.int32, device=dev)
    best = torch.full((n,), -7, dtype=torch.int32, device=dev)
    drop = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    near = torch.full((1,), 5, dtype=torch.int32, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    radius = 1.4 / 60.0
    if static:
        rc = lib.nf_xmatch_self_static(p(u_d), p(s_d), n, radius, p(count), p(best), p(drop), p(near), stream)
    else:
        wbytes = lib.nf_xmatch_workspace_bytes(n)
        work = torch.empty(wbytes // 8 + 1, dtype=torch.float64, device=dev)
        rc = lib.nf_xmatch_self(p(u_d), p(s_d), n, radius, p(count), p(best), p(drop), p(near), p(work), work.numel() * 8, stream)
    assert rc == 0, lib.nf_last_error()
    torch.cuda.synchronize()
    return count.cpu().numpy(), best.cpu().numpy(), drop.cpu().numpy(), int(near.item())


def test_static_scratch_entry_point_equals_the_workspace_one():
    lib = _lib.load()
    nmax = lib.nf_xmatch_static_max_rows()
    assert nmax == 16384
    for n, seed in [(6000, 1), (700, 2), (nmax, 3)]:      # second call smaller than the first; then the largest size allowed
        unit, snr = _catalogue(n, seed)
        ref = _run(lib, unit, snr, static=False)
        got = _run(lib, unit, snr, static=True)
        assert ref[0].min() >= 1 and (ref[0] > 1).sum() > n // 4        # every row is its own partner; many have more
        for a, b in zip(got[:3], ref[:3]):
            assert np.array_equal(a, b)
        assert got[3] == ref[3]
    # sizes beyond the maximum are refused with a clear error, before any device work
    unit, snr = _catalogue(64, 4)
    dev = torch.device('cuda', torch.cuda.current_device())
    u_d, s_d = torch.from_numpy(unit).to(dev), torch.from_numpy(snr).to(dev)
    ints = torch.zeros(4, dtype=torch.int32, device=dev)
    rc = lib.nf_xmatch_self_static(ctypes.c_void_p(u_d.data_ptr()), ctypes.c_void_p(s_d.data_ptr()), nmax + 1, 0.02,
                                   ctypes.c_void_p(ints.data_ptr()), ctypes.c_void_p(ints.data_ptr()), ctypes.c_void_p(ints.data_ptr()),
                                   ctypes.c_void_p(ints.data_ptr()), None)
    assert rc < 0 and b"16384" in lib.nf_last_error()
