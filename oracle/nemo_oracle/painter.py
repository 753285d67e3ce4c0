"""Radial-profile template painter, numpy statement (TEST INFRASTRUCTURE).

Stands where ``pixell.pointsrcs.sim_objects`` stands in the reference (nemo/signals.py:681,691):
paints ``amp * profile(theta)`` for a single object into a float32 map.  sim_objects itself is
third-party compiled code that is absent offline, so the painter semantics are defined by this
build (PARITY UNPINNED against pixell) and shared by the oracle and the GPU kernel
``nf_paint_radial``:

* hav(theta) = A[y] + B[y]*C[x] with A = sin^2(ddec/2), B = cos(dec)cos(dec0), C = sin^2(dra/2)
  (per-row / per-column tables computed once on the host, see nemo_amd.signals.paintGeometry);
* theta = 2*asin(sqrt(h)) evaluated by its series in h to h^5 (only +,*,sqrt: bit-reproducible);
* linear interpolation on the (log-uniform) radial table, prof[0] inside r[0], 0 beyond r[-1];
* float64 arithmetic, one cast to float32 at the end (only float32 is supported by the reference's
  painter: nemo/signals.py:644), then the sign flip of nemo/signals.py:698-699.
"""
import numpy as np

_C1 = 1.0 / 6.0
_C2 = 3.0 / 40.0
_C3 = 15.0 / 336.0
_C4 = 105.0 / 3456.0
_C5 = 945.0 / 42240.0


def theta_from_hav(h):
    s = np.sqrt(h)
    poly = _C5
    poly = poly * h + _C4
    poly = poly * h + _C3
    poly = poly * h + _C2
    poly = poly * h + _C1
    poly = poly * h + 1.0
    return (2.0 * s) * poly


def paint_radial(shape, A, B, C, r, prof, amp, sign):
    h = A[:shape[0], None] + B[:shape[0], None] * C[None, :shape[1]]
    theta = theta_from_hav(h)
    j = np.searchsorted(r, theta, side='right') - 1
    inside = (j >= 0) & (j < len(r) - 1)
    jj = np.clip(j, 0, len(r) - 2)
    slope = (prof[jj + 1] - prof[jj]) / (r[jj + 1] - r[jj])
    val = slope * (theta - r[jj]) + prof[jj]
    val = np.where(inside, val, 0.0)
    val = np.where(theta < r[0], prof[0], val)
    val = np.where(theta == r[-1], prof[-1], val)
    out = (float(np.float32(amp)) * val).astype(np.float32)
    if sign < 0:
        out = out * np.float32(-1.0)
    return out
