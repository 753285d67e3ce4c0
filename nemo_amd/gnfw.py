"""GNFW pressure profile and its line-of-sight projection (1-D, host side).

Role in the path: feeds the cluster signal templates of the filter bank (reference: nemo/gnfw.py:121-195 via
nemo/signals.py:494-496).  The projection follows the reference's algorithm step for step, because the template - and
through it filt, signalNorm, y_c - inherits its values: the integral 2 * int_0^zmax P(sqrt(b^2 + z^2)) dz is taken with
scipy ``quad`` after the substitution z = y^(a+1) / (a+1) (a = 8) at 100 impact parameters log-spaced in [1e-8, 1e4], the
logarithm of the result is interpolated with a cubic ``interp1d`` in ln b, and impact parameters outside the table get the
first value / a power-law continuation.  tests/golden/gnfw.npz (made by the reference's own function) holds to 1e-12.
Parameter names follow Arnaud et al. (2010) as the reference does.
"""
import numpy as np

_default_params = {'P0': 8.403, 'c500': 1.177, 'gamma': 0.3081, 'alpha': 1.0510, 'beta': 5.4905,
                   'tol': 1e-7, 'npts': 100}


def gnfw(x, c, alpha, beta, gamma, P0=1.0):
    """GNFW radial profile P(x), x = r/R500 (reference: nemo/gnfw.py:121-136)."""
    return P0 * ((x * c) ** -gamma * (1 + (x * c) ** alpha) ** ((gamma - beta) / alpha))


def tsz_profile_los_exact(x, c=1.177, alpha=1.0510, beta=5.4905, gamma=0.3081, zmax=1e5, _a=8):
    """Line-of-sight integral by adaptive quadrature, one ``quad`` per impact parameter (nemo/gnfw.py:170-195): the
    substitution z = y^(a+1)/(a+1) flattens the integrand near z = 0."""
    from scipy.integrate import quad
    x = np.asarray(x)
    p = _a + 1
    yTop = (zmax * p) ** (1 / p)
    out = np.empty(x.size)
    for k, b in enumerate(x.reshape(-1)):
        out[k] = quad(lambda y: y ** _a * gnfw(((y ** p / p) ** 2 + b ** 2) ** 0.5, c=c, alpha=alpha, beta=beta, gamma=gamma),
                      0, yTop)[0]
    return (2 * out).reshape(x.shape)


class _LosTable(object):
    """ln(profile) on npoint values of ln b in [ln x1, ln x2], cubic interp1d between them (nemo/gnfw.py:155-159)."""

    def __init__(self, c, alpha, beta, gamma, zmax, npoint, x1, x2, _a):
        from scipy import interpolate
        lnb = np.linspace(np.log(x1), np.log(x2), npoint)
        lnp = np.log(tsz_profile_los_exact(np.exp(lnb), c=c, alpha=alpha, beta=beta, gamma=gamma, zmax=zmax, _a=_a))
        self.inner = interpolate.interp1d(lnb, lnp, "cubic")
        self.x1, self.x2 = x1, x2
        self.first, self.last = lnp[0], lnp[-1]
        self.endSlope = (lnp[-2] - lnp[-1]) / (lnb[-2] - lnb[-1])

    def __call__(self, x):
        x = np.asarray(x, dtype=float)
        below, above = x < self.x1, x > self.x2
        within = (~below) & (~above)
        return np.piecewise(x, [within, below, above],
                            [lambda v: np.exp(self.inner(np.log(v))),
                             lambda v: np.exp(self.first),
                             lambda v: np.exp(self.last + (np.log(v) - np.log(self.x2)) * self.endSlope)])


_losTables = {}


def tsz_profile_los(x, c=1.177, alpha=1.0510, beta=5.4905, gamma=0.3081, zmax=1e5, npoint=100, x1=1e-8, x2=1e4, _a=8):
    """Line-of-sight integral of the GNFW profile at impact parameters ``x`` (units of R500): the tabulated / interpolated
    form of nemo/gnfw.py:142-168 (table cached per parameter set; building it takes about 25 ms)."""
    key = (c, alpha, beta, gamma, zmax, npoint, _a, x1, x2)
    if key not in _losTables:
        _losTables[key] = _LosTable(c, alpha, beta, gamma, zmax, npoint, x1, x2, _a)
    return _losTables[key](x)
