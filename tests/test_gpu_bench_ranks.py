"""bench.py --gpus 2 rehearsed on the one GPU of the box (two ranks, gloo): the headline C2 workload and the tiled C3-shaped
one.  Each rank filters its own tile(s); the step ends with the path's one exchange - the all-gather of catalogue rows
(nemo/pipelines.py:322) - and the duplicate removal of tile overlaps (nemo/catalogs.py:704-749).  Asserted: the line rank 0
prints is whole-job (n_gpus 2), and the gathered catalogue on BOTH ranks equals removeDuplicates(rank 0's own rows + rank 1's
own rows), row for row and bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from nemo_amd import catalogs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table(z, prefix):
    return catalogs.Table(cols={'RADeg': z[prefix + 'RADeg'], 'decDeg': z[prefix + 'decDeg'], 'SNR': z[prefix + 'SNR'],
                                'tileName': z[prefix + 'tileName']})


@pytest.mark.gpu
@pytest.mark.parametrize("name,args,port", [
    ("C2", ["--size", "2048", "--scales", "6"], "29671"),
    ("C3", ["--tile", "840x1440", "--freqs", "2", "--tiles-per-gpu", "2", "--scales", "6"], "29672")])
def test_bench_two_ranks_on_one_gpu(tmp_path, name, args, port):
    out = tmp_path / "dump"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0", NEMO_AMD_DIST_BACKEND="gloo",
               OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-procs", "0",
           "--dump-catalogs", str(out)] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['value'] > 0
    z0, z1 = np.load(out / "rank0.npz"), np.load(out / "rank1.npz")
    assert len(z0['own_RADeg']) > 0 and len(z1['own_RADeg']) > 0
    want, numDup, names = catalogs.removeDuplicates(catalogs.vstack([_table(z0, 'own_'), _table(z1, 'own_')]))
    for z in (z0, z1):                       # every rank holds the whole catalogue
        got = _table(z, 'all_')
        assert len(got) == len(want)
        ia, ib = np.lexsort((got['decDeg'], got['RADeg'])), np.lexsort((want['decDeg'], want['RADeg']))
        for k in ('RADeg', 'decDeg', 'SNR', 'tileName'):
            assert np.array_equal(np.asarray(got[k])[ia], np.asarray(want[k])[ib]), (name, k)
    print("%s: %d + %d rows -> %d after duplicate removal (%d in duplicate groups); %.1f ms per step on two ranks sharing one GPU"
          % (name, len(z0['own_RADeg']), len(z1['own_RADeg']), len(want), numDup, line['ms_per_step']))
