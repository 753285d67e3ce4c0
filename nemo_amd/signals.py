"""Signal templates for the filter bank: beams, GNFW cluster profiles, SZ spectral weights.

Host-side 1-D work only (radial profiles, Hankel transforms, cosmology scalars); the 2-D template
is painted on the GPU by ``nf_paint_radial`` from the log-uniform radial table built here
(:func:`makeRadialTable`).  Mirrors the reference functions on the path:

* ``fSZ``                        - nemo/signals.py:352-377
* ``calcTheta500Arcmin`` etc.    - nemo/signals.py:380-447
* ``BeamProfile``                - nemo/signals.py:75-131
* ``makeArnaudModelProfile``     - nemo/signals.py:450-506
* ``makeBattagliaModelProfile``  - nemo/signals.py:509-591
* ``_paintSignalMap`` (1-D part: beam convolution in harmonic space, amplitude handling,
  sign handling)                - nemo/signals.py:630-701
* ``make{Beam,ArnaudModel,BattagliaModel}SignalMap`` - nemo/signals.py:595-627,704-845

Third-party pieces the reference uses here are absent offline (pyccl, pixell.curvedsky,
pixell.utils.RadialFourierTransform, pixell.pointsrcs.sim_objects: SURVEY.md §8c).  They are
replaced by self-contained equivalents (flat LCDM integrals, an FFTLog Hankel transform, a
table-driven painter).  Those sit upstream of the parity boundary - the same tables feed the
oracle's painter and the GPU painter - and agreement with genuine pixell/pyccl output is unpinned.
"""
import os

import numpy as np
from scipy import interpolate, special, integrate, optimize

from . import gnfw

TCMB = 2.72548
CLight = 299792458.0          # m/s
_h_planck = 6.62607015e-34      # CODATA 2018 (astropy.constants.h)
_k_B = 1.380649e-23             # CODATA 2018 (astropy.constants.k_B)
Mpc_in_cm = 3.0856775814913674e24
MSun_in_g = 1.988409870698051e33

# Default cosmology (nemo/signals.py:52-57)
Om0 = 0.3
Ob0 = 0.05
H0 = 70.0
sigma8 = 0.8
ns = 0.95
RHO_CRITICAL = 2.77536627e11    # h^2 MSun / Mpc^3 (value of ccl.physical_constants.RHO_CRITICAL)


class FlatLCDM(object):
    """Stand-in for the pyccl.Cosmology object passed around as ``cosmoModel`` (flat LCDM, matter + Lambda)."""

    def __init__(self, Om0=Om0, H0=H0):
        self.Om0 = Om0
        self.H0 = H0
        self.h = H0 / 100.0

    def __getitem__(self, key):
        return {'h': self.h, 'Omega_m': self.Om0}[key]

    def h_over_h0(self, a):
        z = 1.0 / a - 1.0
        return np.sqrt(self.Om0 * (1.0 + z) ** 3 + (1.0 - self.Om0))

    def angular_diameter_distance(self, a):
        z = 1.0 / a - 1.0
        dc = integrate.quad(lambda zz: 1.0 / self.h_over_h0(1.0 / (1.0 + zz)), 0.0, z, epsabs=0, epsrel=1e-12)[0]
        return (CLight / 1000.0 / self.H0) * dc / (1.0 + z)


fiducialCosmoModel = FlatLCDM()


# ------------------------------------------------------------------------------------------------------------
def fSZ(obsFrequencyGHz, TCMBAlpha=0.0, z=None):
    """Non-relativistic SZ spectral function (nemo/signals.py:352-377)."""
    x = (_h_planck * obsFrequencyGHz * 1e9) / (_k_B * TCMB)
    if TCMBAlpha != 0 and z is not None:
        assert (z >= 0)
        x = x * np.power(1 + z, TCMBAlpha)
    return x * ((np.exp(x) + 1) / (np.exp(x) - 1)) - 4.0


def convertToDeltaT(mapData, obsFrequencyGHz=148, TCMBAlpha=0.0, z=None):
    """Compton y -> delta T (uK) (nemo/maps.py:679-697)."""
    fx = fSZ(obsFrequencyGHz, TCMBAlpha=TCMBAlpha, z=z)
    return mapData * fx * (TCMB * 1e6)


def calcRDeltaMpc(z, MDelta, cosmoModel, delta=500, wrt='critical'):
    """nemo/signals.py:380-408."""
    if type(MDelta) == str:
        raise Exception("MDelta is a string - use, e.g., 1.0e+14 (not 1e14 or 1e+14)")
    Ez = cosmoModel.h_over_h0(1 / (1 + z))
    if wrt == 'critical':
        wrtDensity = RHO_CRITICAL * (Ez * cosmoModel['h']) ** 2
    elif wrt == 'mean':
        OmegaMz = cosmoModel['Omega_m'] * (1 + z) ** 3 / Ez ** 2
        wrtDensity = OmegaMz * RHO_CRITICAL * (Ez * cosmoModel['h']) ** 2
    else:
        raise Exception("wrt should be either 'critical' or 'mean'")
    return np.power((3 * MDelta) / (4 * np.pi * delta * wrtDensity), 1.0 / 3.0)


def calcR500Mpc(z, M500c, cosmoModel):
    return calcRDeltaMpc(z, M500c, cosmoModel, delta=500, wrt='critical')


_theta500Cache = {}


def calcTheta500Arcmin(z, M500, cosmoModel):
    """(kept per (z, M500, cosmology): maps.makeModelImage asks once per catalogue row, an injection run for the same few
    models thousands of times)"""
    key = (float(z), float(M500), id(cosmoModel))
    if key not in _theta500Cache:
        if len(_theta500Cache) > 4096:
            _theta500Cache.clear()
        _theta500Cache[key] = _calcTheta500Arcmin(z, M500, cosmoModel)
    return _theta500Cache[key]


def _calcTheta500Arcmin(z, M500, cosmoModel):
    """nemo/signals.py:429-447."""
    R500Mpc = calcR500Mpc(z, M500, cosmoModel)
    return np.degrees(np.arctan(R500Mpc / cosmoModel.angular_diameter_distance(1 / (1 + z)))) * 60.0


def _nfw_m(x):
    return np.log(1.0 + x) - x / (1.0 + x)


def M500cToM200c(M500c, z, cosmoModel):
    """NFW mass conversion with the Duffy et al. (2008) c200c(M, z) relation.  Stands in for the CCL
    mass translator behind ``M500cToMdef`` (nemo/signals.py:551); only used by the Battaglia profile."""
    h = cosmoModel['h']

    def c200c(M200c):
        return 5.71 * (M200c * h / 2e12) ** -0.084 * (1 + z) ** -0.47

    def resid(logM200):
        M200 = 10 ** logM200
        c = c200c(M200)
        # x500 = r500/rs solves m(x)/x^3 * (200/500) = m(c)/c^3
        f = lambda x: _nfw_m(x) / x ** 3 * (200.0 / 500.0) - _nfw_m(c) / c ** 3
        x500 = optimize.brentq(f, 1e-3, c)
        return np.log10(M200 * _nfw_m(x500) / _nfw_m(c)) - np.log10(M500c)

    return 10 ** optimize.brentq(resid, np.log10(M500c), np.log10(M500c) + 1.0)


# ------------------------------------------------------------------------------------------------------------
class RadialFourierTransform(object):
    """Flat-sky radial (order-0 Hankel) transform pair on a log grid, by the FFTLog method:

        F(l) = 2 pi int f(r) J0(l r) r dr,     f(r) = 1/(2 pi) int F(l) J0(l r) l dl.

    Plays the role of ``pixell.utils.RadialFourierTransform`` at nemo/signals.py:651-657 (own
    implementation; grid choices are ours).  ``r`` in radians, ``l`` multipole.
    """

    def __init__(self, n=4096, rrange=(1e-10, 1e2), q=0.5):
        self.n = n
        self.q = q
        self.dln = np.log(rrange[1] / rrange[0]) / n
        self.r = rrange[0] * np.exp(np.arange(n) * self.dln)
        # low-ringing style centring: l_j * r_(n-1-j) = 1
        self.l = (1.0 / self.r[-1]) * np.exp(np.arange(n) * self.dln)
        m = np.fft.fftfreq(n) * n
        omega = 2 * np.pi * m / (n * self.dln)
        w = q + 1j * omega
        # Mellin transform of J0: int x^(w-1) J0(x) dx = 2^(w-1) Gamma(w/2) / Gamma(1 - w/2)
        lnM = (w - 1) * np.log(2.0) + special.loggamma(w / 2.0) - special.loggamma(1.0 - w / 2.0)
        self._kern = np.exp(lnM - 1j * omega * (np.log(self.r[0]) + np.log(self.l[0])))
        if n % 2 == 0:
            self._kern[n // 2] = self._kern[n // 2].real

    def real2harm(self, rprof):
        a = rprof * self.r ** (2.0 - self.q)
        c = np.fft.fft(a) / self.n
        h = np.fft.fft(c * self._kern).real
        return 2 * np.pi * h * self.l ** (-self.q)

    def harm2real(self, lprof):
        a = lprof * self.l ** (2.0 - self.q)
        c = np.fft.fft(a) / self.n
        h = np.fft.fft(c * self._kern).real
        return h * self.r ** (-self.q) / (2 * np.pi)


_rht_cache = {}


def _getRHT():
    if 'rht' not in _rht_cache:
        _rht_cache['rht'] = RadialFourierTransform()
    return _rht_cache['rht']


# ------------------------------------------------------------------------------------------------------------
class BeamProfile(object):
    """Beam profile from an ACT-style text file (radial profile, or B_l transform), from arrays, or
    Gaussian (``FWHMArcmin``; convenience for synthetic configs).  nemo/signals.py:75-131.

    Attributes as in the reference: ``profile1d``, ``rDeg``, ``tck``, ``Bell``, ``ell``, ``FWHMArcmin``.
    """

    def __init__(self, beamFileName=None, profile1d=None, rDeg=None, FWHMArcmin=None):
        self.beamFileName = beamFileName
        if FWHMArcmin is not None:
            sigmaDeg = FWHMArcmin / 60.0 / np.sqrt(8 * np.log(2))
            rDeg = np.linspace(0.0, 1.0, 10001)
            profile1d = np.exp(-0.5 * (rDeg / sigmaDeg) ** 2)
        if beamFileName is not None:
            beamData = np.loadtxt(beamFileName).transpose()
            if beamData[0][1] - beamData[0][0] >= 1:
                # beam transform file
                ell = beamData[0]
                Bell = beamData[1]
                if len(np.unique(np.diff(ell))) != 1:
                    raise Exception("If using a beam transform file, need delta ell = 1 between all ell values.")
                Bell = Bell / Bell[0]
                rDeg = np.linspace(0.0, 0.5, 1800)
                rht = _getRHT()
                lB = np.interp(rht.l, ell, Bell, left=1.0, right=0.0)
                prof = np.interp(np.radians(rDeg), rht.r, rht.harm2real(lB))
                prof = prof / prof[0]
                self.profile1d = prof
                self.rDeg = rDeg
                self.Bell = Bell
                self.ell = ell
            else:
                self.profile1d = beamData[1]
                self.rDeg = beamData[0]
                self._makeBell()
        else:
            self.profile1d = profile1d
            self.rDeg = rDeg
            if self.profile1d is not None:
                self._makeBell()
        if self.profile1d is not None and self.rDeg is not None:
            self.tck = interpolate.splrep(self.rDeg, self.profile1d)
        self.FWHMArcmin = self.rDeg[np.argmin(abs(self.profile1d - 0.5))] * 60 * 2

    def _makeBell(self):
        rht = _getRHT()
        prof = np.interp(rht.r, np.radians(self.rDeg), self.profile1d, right=0.0)
        lB = rht.real2harm(prof)
        ell = np.arange(0, 40001, dtype=np.float64)
        Bell = np.interp(ell, rht.l, lB)
        self.Bell = Bell / Bell[0]
        self.ell = ell

    def cacheKey(self):
        key = getattr(self, '_cacheKey', None)
        if key is None:
            if self.beamFileName is not None:
                key = os.path.abspath(self.beamFileName)
            else:
                key = ('arrays', len(self.rDeg), float(np.sum(self.profile1d)), float(self.rDeg[-1]))
            self._cacheKey = key
        return key


_beamCache = {}


def _asBeam(beam):
    if type(beam) == str:
        key = os.path.abspath(beam)
        if key not in _beamCache:
            _beamCache[key] = BeamProfile(beamFileName=beam)
        return _beamCache[key]
    return beam


# ------------------------------------------------------------------------------------------------------------
def makeArnaudModelProfile(z, M500, GNFWParams='default', cosmoModel=None, binning='log'):
    """nemo/signals.py:450-506: spline knots ``tckP`` of the unit-peak projected GNFW profile vs degrees."""
    if cosmoModel is None:
        cosmoModel = fiducialCosmoModel
    if GNFWParams == 'default':
        GNFWParams = gnfw._default_params
    if binning == 'linear':
        bRange = np.linspace(0, 30, 1000)
    elif binning == 'log':
        bRange = np.logspace(np.log10(1e-6), np.log10(100), 300)
    else:
        raise Exception("'binning' must be 'linear' or 'log' (given '%s')." % (binning))
    cylPProfile = gnfw.tsz_profile_los(bRange, c=GNFWParams['c500'], alpha=GNFWParams['alpha'],
                                       beta=GNFWParams['beta'], gamma=GNFWParams['gamma'])
    cylPProfile = cylPProfile / cylPProfile.max()
    theta500Arcmin = calcTheta500Arcmin(z, M500, cosmoModel)
    thetaDegRange = bRange * (theta500Arcmin / 60.)
    tckP = interpolate.splrep(thetaDegRange, cylPProfile)
    return {'tckP': tckP, 'theta500Arcmin': theta500Arcmin, 'rDeg': thetaDegRange}


def makeBattagliaModelProfile(z, M500c, GNFWParams='default', cosmoModel=None):
    """nemo/signals.py:509-591 (Battaglia et al. 2012 mass/redshift-dependent GNFW shape, A10 notation)."""
    if cosmoModel is None:
        cosmoModel = fiducialCosmoModel
    if GNFWParams == 'default':
        GNFWParams = {'P0': 7.49, 'gamma': 0.3, 'alpha': 1.0, 'beta': 4.49, 'c500': 1.408, 'tol': 1e-7, 'npts': 100}
    GNFWParams = dict(GNFWParams)
    P0 = GNFWParams['P0']
    xc = 1 / GNFWParams['c500']
    beta = GNFWParams['beta'] - 0.3
    M200c = M500cToM200c(M500c, z, cosmoModel)
    P0z = P0 * np.power(M200c / 1e14, 0.226) * np.power(1 + z, -0.957)
    xcz = xc * np.power(M200c / 1e14, -0.0833) * np.power(1 + z, 0.853)
    betaz = beta * np.power(M200c / 1e14, 0.0480) * np.power(1 + z, 0.615)
    GNFWParams['P0'] = P0z
    GNFWParams['beta'] = betaz + 0.3
    GNFWParams['c500'] = 1 / xcz
    GNFWParams['gamma'] = 0.3
    GNFWParams['alpha'] = 1.0
    bRange = np.logspace(np.log10(1e-6), np.log10(100), 300)
    cylPProfile = gnfw.tsz_profile_los(bRange, c=GNFWParams['c500'], alpha=GNFWParams['alpha'],
                                       beta=GNFWParams['beta'], gamma=GNFWParams['gamma'])
    cylPProfile = cylPProfile / cylPProfile.max()
    theta500Arcmin = calcTheta500Arcmin(z, M500c, cosmoModel)
    thetaDegRange = bRange * (theta500Arcmin / 60.)
    tckP = interpolate.splrep(thetaDegRange, cylPProfile)
    return {'tckP': tckP, 'theta500Arcmin': theta500Arcmin, 'rDeg': thetaDegRange}


# ------------------------------------------------------------------------------------------------------------
class RadialTable(object):
    """Log-uniform radial table consumed by the painter (oracle: numpy; product: ``nf_paint_radial``).

    value(theta) = amp * interp(theta; r, prof) for r[0] <= theta <= r[-1]; prof[0] below r[0]; 0 beyond r[-1].
    ``r`` in radians, ``r[k] = r0 * exp(k * dlnr)`` (the table itself, not the formula, is authoritative:
    both painters read ``r``).  ``sign`` = -1 if the (beam-convolved) profile is negative at the centre
    (nemo/signals.py:698-699; the painted profile is ``abs(rprof)``, nemo/signals.py:681,691).
    """

    def __init__(self, r, prof, amp, sign):
        self.r = np.ascontiguousarray(r, dtype=np.float64)
        self.prof = np.ascontiguousarray(prof, dtype=np.float64)
        self.amp = float(np.float32(amp))       # amps are cast to float32 (nemo/signals.py:644,670-673)
        self.ampExact = float(amp)
        self.sign = float(sign)
        self.r0 = float(self.r[0])
        self.dlnr = float(np.log(self.r[-1] / self.r[0]) / (len(self.r) - 1))


_TABLE_N = 4096
_profileCache = {}
_tableCache = {}


def makeRadialTable(tckP, beam=None, amplitude=None, maxSizeDeg=10.0, convolveWithBeam=True, cacheKey=None):
    """1-D part of nemo/signals.py:630-701 (``_paintSignalMap``): evaluate the profile spline, optionally
    convolve with the beam in harmonic space (:651-657), apply the amplitude convention (:662-664), and
    return the table for the painter.  ``cacheKey`` (hashable) lets repeated tiles/scales reuse the 1-D work,
    as the reference's own interpolator cache does for the GNFW integral (nemo/gnfw.py:141-157)."""
    key = None
    if cacheKey is not None:
        key = (cacheKey, beam.cacheKey() if (convolveWithBeam and beam is not None) else None,
               maxSizeDeg, convolveWithBeam)
    if key is not None and key in _profileCache:
        r, rprof = _profileCache[key]
    else:
        rDeg = np.logspace(np.log10(1e-6), np.log10(maxSizeDeg), _TABLE_N)
        r = np.radians(rDeg)
        if convolveWithBeam == True:
            if beam is None:
                raise Exception("No beam supplied.")
            rht = _getRHT()
            rp = interpolate.splev(np.degrees(rht.r), tckP, ext=1)
            lbeam = np.interp(rht.l, beam.ell, beam.Bell, left=1.0, right=0.0)
            lprof = rht.real2harm(rp)
            lprof *= lbeam
            rp = rht.harm2real(lprof)
            rprof = np.interp(r, rht.r, rp)
        else:
            rprof = interpolate.splev(rDeg, tckP, ext=1)
        if key is not None:
            _profileCache[key] = (r, rprof)
    # the finished table is kept per (profile, amplitude): the filter bank asks for the same few tables for every tile, and the
    # painter keeps its device copy on the table object - one upload per table and process instead of one per tile and scale
    tkey = None if key is None else (key, None if amplitude is None else float(amplitude))
    if tkey is not None and tkey in _tableCache:
        return _tableCache[tkey]
    amp = 1.0
    if amplitude is not None:
        amp = rprof[0] * amplitude
        rprof = rprof / rprof[0]
    sign = -1.0 if rprof[0] < 0 else 1.0
    table = RadialTable(r, abs(rprof), amp, sign)
    if tkey is not None:
        if len(_tableCache) >= 1024:
            _tableCache.clear()
        _tableCache[tkey] = table
    return table


def paintGeometry(shape, wcs, RADeg, decDeg, window=None):
    """Per-row / per-column haversine tables shared verbatim by the oracle painter and the GPU painter:
    hav(theta) = A[y] + B[y] * C[x].  Painter semantics are ours (pixell.pointsrcs.sim_objects is absent).
    ``window`` = (x0, x1, y0, y1): only the tables' slices A[y0:y1], B[y0:y1], C[x0:x1] (the same values element for element:
    every step is elementwise)."""
    if window is None:
        dec = wcs.rowDecRad()[:shape[0]]
        ra = wcs.colRARad()[:shape[1]]
    else:
        x0, x1, y0, y1 = window
        dec = wcs.rowDecRad()[y0:y1]
        ra = wcs.colRARad()[x0:x1]
    dec0 = np.radians(decDeg)
    ra0 = np.radians(RADeg)
    dra = (ra - ra0 + np.pi) % (2 * np.pi) - np.pi
    A = np.sin((dec - dec0) / 2.0) ** 2
    B = np.cos(dec) * np.cos(dec0)
    C = np.sin(dra / 2.0) ** 2
    return A, B, C


_modelProfileCache = {}


def clusterModelTable(kind, beam, z, M500, GNFWParams, amplitude, maxSizeDeg, baseOnly=False):
    """Radial table of one model cluster of maps.makeModelImage (nemo/maps.py:1821-1831 -> make*ModelSignalMap,
    nemo/signals.py:751-770): the beam-convolved projected GNFW profile out to ``maxSizeDeg``, central amplitude ``amplitude``."""
    beam = _asBeam(beam)
    gkey = 'default' if GNFWParams == 'default' else tuple(sorted(GNFWParams.items()))
    ckey = (kind, float(z), float(M500), gkey)
    if ckey not in _profileCache:
        fn = makeArnaudModelProfile if kind == 'arnaud' else makeBattagliaModelProfile
        _profileCache[ckey] = fn(z, M500, GNFWParams=GNFWParams)
    # one table object per (profile, beam, size, amplitude): tables of equal profile share their arrays through the cache
    key = (ckey, beam.cacheKey(), float(maxSizeDeg))
    base = _modelProfileCache.get(key)
    if base is None:
        base = _modelProfileCache[key] = makeRadialTable(_profileCache[ckey]['tckP'], beam=beam, amplitude=1.0,
                                                         maxSizeDeg=float(maxSizeDeg), convolveWithBeam=True, cacheKey=ckey)
    if baseOnly:
        return base               # the amplitude-1 table: scaledTable(base, amplitude) gives a row's table
    return scaledTable(base, amplitude)


def scaledTable(base, amplitude):
    """A table made with amplitude 1 given another amplitude: amp = rprof[0] * amplitude (nemo/signals.py:662-664), cast to
    float32 once.  The arrays are shared with ``base`` (one upload for a whole catalogue)."""
    tab = RadialTable.__new__(RadialTable)
    tab.r, tab.prof, tab.sign, tab.r0, tab.dlnr = base.r, base.prof, base.sign, base.r0, base.dlnr
    tab.ampExact = base.ampExact * amplitude
    tab.amp = float(np.float32(tab.ampExact))
    return tab


def templateTables(kind, beam, params, amplitude=None):
    """Radial table for one template of the bank.  ``kind`` in {'beam', 'arnaud', 'battaglia'}
    (template mixins, nemo/filters.py:1214-1266)."""
    beam = _asBeam(beam)
    if kind == 'beam':
        # nemo/signals.py:595-627: beam profile itself, maxSizeDeg = 1, no convolution
        return makeRadialTable(beam.tck, beam=None, amplitude=1.0 if amplitude is None else amplitude,
                               maxSizeDeg=1.0, convolveWithBeam=False, cacheKey=('beam', beam.cacheKey()))
    GNFWParams = params.get('GNFWParams', 'default')
    gkey = 'default' if GNFWParams == 'default' else tuple(sorted(GNFWParams.items()))
    ckey = (kind, float(params['z']), float(params['M500MSun']), gkey)
    if ckey not in _profileCache:
        if kind == 'arnaud':
            _profileCache[ckey] = makeArnaudModelProfile(params['z'], params['M500MSun'], GNFWParams=GNFWParams)
        elif kind == 'battaglia':
            _profileCache[ckey] = makeBattagliaModelProfile(params['z'], params['M500MSun'], GNFWParams=GNFWParams)
        else:
            raise Exception("unknown template kind '%s'" % (kind))
    signalDict = _profileCache[ckey]
    return makeRadialTable(signalDict['tckP'], beam=beam, amplitude=amplitude, maxSizeDeg=15.0,
                           convolveWithBeam=True, cacheKey=ckey)


# ------------------------------------------------------------------------------------------------------------
def _QScaleGrid(refParams, zDepQ, cosmoModel):
    """The (z, M500) pairs the filter mismatch function is measured at (nemo/signals.py:933-998): the reference filter's own
    scale first, then masses that give the wanted theta500 values at the listed redshifts."""
    def massFor(theta500Arcmin, z):                                                # :947-950
        Ez = cosmoModel.h_over_h0(1 / (1 + z))
        criticalDensity = RHO_CRITICAL * (Ez * cosmoModel['h']) ** 2
        R500Mpc = np.tan(np.radians(theta500Arcmin / 60.0)) * cosmoModel.angular_diameter_distance(1 / (1 + z))
        return (4 / 3.0) * np.pi * np.power(R500Mpc, 3) * 500 * criticalDensity

    MRange, zRange = [refParams['M500MSun']], [refParams['z']]
    if zDepQ == 0:
        theta500Arcmin_wanted = np.power(10, np.arange(np.log10(0.1), np.log10(50), 0.05055349))
        zRange_wanted = [2.0] * 10 + [1.0] * 10 + [0.6] * 10 + [0.3] * 10 + [0.1] * 10 + [0.07] * 4
        for theta500Arcmin, z in zip(theta500Arcmin_wanted, zRange_wanted):
            MRange.append(massFor(theta500Arcmin, z))
        zRange = zRange + zRange_wanted
    else:
        zGrid = [0.05, 0.1, 0.2, 0.3, 0.4, 0.6, 0.8, 1.0, 1.2, 1.6, 2.0]
        theta500Arcmin_wanted = np.logspace(np.log10(0.1), np.log10(100.0), 24)
        for z in zGrid:
            for theta500Arcmin in theta500Arcmin_wanted:
                MRange.append(massFor(theta500Arcmin, z))
                zRange.append(z)
    return zRange, MRange


def fitQ(config, scales=None):
    """The filter mismatch function Q on a grid of cluster scales, per tile (nemo/signals.py:897-1163): model clusters of
    y0 = 2e-4 at the tile centre, through the pixel window, through the reference filter; Q = interpolated peak / y0.
    Everything a scale needs - painter, window, transforms, filter, spline - runs on the device; the peaks of a tile come
    back in one transfer.  Returns {tileName: Table(Q, theta500Arcmin, z; meta ZDEPQ, TILENAME)} sorted by theta500Arcmin,
    the same on every rank (the reference gathers on rank 0 and writes selFn/QFit.fits, :1140-1160).
    ``scales``: optional [(z, M500MSun)] list replacing the standard grid (the reference filter's own scale is always first)."""
    from . import engine, catalogs, filters
    cosmoModel = fiducialCosmoModel
    photFilterLabel = config.parDict['photFilter']
    ref = None
    for f in config.parDict['mapFilters']:
        if f['label'] == photFilterLabel:
            ref = f
    if ref is None:
        raise Exception("couldn't find filter that matches photFilter")
    if ref['class'].find("Arnaud") != -1:
        kind, zDepQ = 'arnaud', 0
    elif ref['class'].find("Battaglia") != -1:
        kind, zDepQ = 'battaglia', 1
    else:
        raise Exception("Signal model for Q calculation should either be 'Arnaud' or 'Battaglia'")
    if scales is None:
        zRange, MRange = _QScaleGrid(ref['params'], zDepQ, cosmoModel)
    else:
        zRange = [ref['params']['z']] + [s[0] for s in scales]
        MRange = [ref['params']['M500MSun']] + [s[1] for s in scales]
    GNFWParams = config.parDict.get('GNFWParams', 'default')
    t = engine.torch()
    y0 = 2e-04
    QTabDict = {}
    for tileName in config.tileNames:
        if tileName in config.cachedFilters.keys():                                # :1007-1022
            filterObj = config.cachedFilters[tileName]
        else:
            filterClass = getattr(filters, ref['class'])
            filterObj = filterClass(ref['label'], config.unfilteredMapsDictList, ref['params'], tileName=tileName,
                                    diagnosticsDir=config.diagnosticsDir)
            filterObj.loadFilter()
        realSpace = isinstance(filterObj, filters.RealSpaceMatchedFilter)
        eng = filterObj.engine
        shape, wcs = filterObj.shape, filterObj.wcs
        RADeg, decDeg = wcs.getCentreWCSCoords()
        x, y = wcs.wcs2pix(RADeg, decDeg)
        geom = paintGeometry(shape, wcs, RADeg, decDeg)
        beams = [(m['obsFreqGHz'], m['beamFileName']) for m in config.parDict['unfilteredMaps']]       # :1031-1034
        peaks, thetas = [], []
        for z, M500MSun in zip(zRange, MRange):
            planes = []
            for obsFreqGHz, beam in beams:
                amplitude = convertToDeltaT(y0, obsFreqGHz) if obsFreqGHz is not None else y0         # :1085-1088
                table = clusterModelTable(kind, beam, z, M500MSun, GNFWParams if GNFWParams is not None else 'default', amplitude, 15.0)
                planes.append(eng.apply_window(eng.paint(geom, table), 1.0))       # :1094-1099
            cube = t.stack(planes)
            if realSpace:
                filteredSignal = filterObj._applyFilterDevice(cube)
            else:
                filteredSignal = filterObj._applyFilterDevice(eng.fft(cube, unitary=True, half=True), half=True)      # :1108-1112
            peaks.append(eng.spline_eval(filteredSignal, [[y, x]], to_host=False))                  # :1113-1116
            thetas.append(calcTheta500Arcmin(z, M500MSun, cosmoModel))
        peaks = t.cat([p.reshape(1) for p in peaks]).cpu().numpy()
        Q, QTheta500Arcmin, Qz = [], [], []
        for p, th, z in zip(peaks, thetas, zRange):                                # :1119-1122 (a repeated peak value is dropped)
            if p not in Q:
                Q.append(p)
                QTheta500Arcmin.append(th)
                Qz.append(z)
        Q = np.array(Q)
        if abs(1 - Q[0] / y0) > 1e-6:
            raise Exception("Q[0]/y0 outside tolerance")
        Q = Q / y0
        QTab = catalogs.Table()
        QTab['Q'] = Q
        QTab['theta500Arcmin'] = np.array(QTheta500Arcmin)
        QTab['z'] = np.array(Qz)
        QTab.sort('theta500Arcmin')
        from . import __version__
        QTab.meta['NEMOVER'] = __version__
        QTab.meta['ZDEPQ'] = zDepQ
        QTab.meta['TILENAME'] = tileName
        QTabDict[tileName] = QTab
    if config.MPIEnabled == True and config.size > 1:
        from . import pipelines
        tileIndex = {name: i for i, name in enumerate(config.allTileNames)}
        rows = [np.stack([tab['Q'], tab['theta500Arcmin'], tab['z'], np.full(len(tab), tileIndex[name])], axis=1)
                for name, tab in QTabDict.items()]
        rows = pipelines._gatherRows(np.concatenate(rows) if rows else np.zeros((0, 4)), config)
        QTabDict = {}
        for i, name in enumerate(config.allTileNames):
            sel = rows[:, 3] == i
            if sel.any():
                QTab = catalogs.Table()
                QTab['Q'], QTab['theta500Arcmin'], QTab['z'] = rows[sel, 0], rows[sel, 1], rows[sel, 2]
                QTab.meta.update({'ZDEPQ': zDepQ, 'TILENAME': name})
                QTabDict[name] = QTab
    return QTabDict
