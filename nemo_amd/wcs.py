"""Minimal plate-carree (CAR) WCS with the astLib ``astWCS.WCS`` method names the
hot path calls.

The reference drives every pixel<->sky conversion through ``astLib.astWCS.WCS``
(not vendored in /root/reference, not installed here: SURVEY.md §8c).  Only the
methods used on the path are provided, with astLib's calling convention
(0-indexed pixel coordinates, ``(x, y)`` argument order, degrees):

* ``pix2wcs`` / ``wcs2pix``            - nemo/photometry.py:115, nemo/filters.py:224,653
* ``getCentreWCSCoords``               - nemo/filters.py:652,1084,1233
* ``getPixelSizeDeg``                  - nemo/filters.py:414,726,1134
* ``getImageMinMaxWCSCoords``          - nemo/filters.py:1079
* ``header`` / ``copy`` / ``updateFromHeader`` - nemo/filters.py:88-92,1025

Only CAR with CRVAL2 = 0 is supported (what ACT/SO maps and the reference's test
headers use: tests/configs/smallTestSurveyMaskHeader.txt:7-16).  Parity with
genuine astLib output is unpinned (astLib absent) - the convention is shared by
the oracle and the GPU path, so parity between those two is unaffected.
"""
import copy

import numpy as np


class CARWCS(object):
    def __init__(self, header):
        self.header = dict(header)
        self.updateFromHeader()

    # -- construction helpers ------------------------------------------------------------------
    @classmethod
    def make(cls, shape, pixDeg=0.5 / 60.0, RADeg=0.0, decDeg=0.0):
        """CAR WCS for a (H, W) map whose central pixel sits at (RADeg, decDeg)."""
        H, W = int(shape[0]), int(shape[1])
        hdr = {'NAXIS': 2, 'NAXIS1': W, 'NAXIS2': H,
               'CTYPE1': 'RA---CAR', 'CTYPE2': 'DEC--CAR',
               'CUNIT1': 'deg', 'CUNIT2': 'deg',
               'CDELT1': -float(pixDeg), 'CDELT2': float(pixDeg),
               'CRVAL1': float(RADeg), 'CRVAL2': 0.0,
               'CRPIX1': W / 2.0 + 1.0,
               'CRPIX2': H / 2.0 + 1.0 - decDeg / float(pixDeg)}
        return cls(hdr)

    def updateFromHeader(self):
        h = self.header
        if not (str(h.get('CTYPE1', '')).endswith('CAR') and str(h.get('CTYPE2', '')).endswith('CAR')):
            raise Exception("CARWCS only supports RA---CAR / DEC--CAR headers")
        if float(h.get('CRVAL2', 0.0)) != 0.0:
            raise Exception("CARWCS needs CRVAL2 = 0 (standard ACT/SO plate carree)")
        self._crval1 = float(h['CRVAL1'])
        self._crpix1 = float(h['CRPIX1'])
        self._crpix2 = float(h['CRPIX2'])
        self._cdelt1 = float(h['CDELT1'])
        self._cdelt2 = float(h['CDELT2'])

    def copy(self):
        return CARWCS(copy.deepcopy(self.header))

    # -- astLib-style interface ----------------------------------------------------------------
    def pix2wcs(self, x, y):
        ra = self._crval1 + (np.asarray(x, dtype=np.float64) + 1.0 - self._crpix1) * self._cdelt1
        dec = (np.asarray(y, dtype=np.float64) + 1.0 - self._crpix2) * self._cdelt2
        ra = np.where(ra < 0, ra + 360.0, ra)
        ra = np.where(ra >= 360.0, ra - 360.0, ra)
        if np.ndim(ra) == 0:
            return float(ra), float(dec)
        return ra, dec

    def wcs2pix(self, RADeg, decDeg):
        dra = np.asarray(RADeg, dtype=np.float64) - self._crval1
        dra = (dra + 180.0) % 360.0 - 180.0
        x = dra / self._cdelt1 + self._crpix1 - 1.0
        y = np.asarray(decDeg, dtype=np.float64) / self._cdelt2 + self._crpix2 - 1.0
        if np.ndim(x) == 0:
            return float(x), float(y)
        return x, y

    def getCentreWCSCoords(self):
        return self.pix2wcs(self.header['NAXIS1'] / 2.0, self.header['NAXIS2'] / 2.0)

    def getPixelSizeDeg(self):
        return abs(self._cdelt2)

    def isFlipped(self):
        """1 when RA increases with x (CDELT1 > 0), as astWCS.WCS.isFlipped."""
        return 1 if self._cdelt1 > 0 else 0

    def getFullSizeSkyDeg(self):
        """(width, height) of the map on the sky in degrees, along its central row and column."""
        W, H = self.header['NAXIS1'], self.header['NAXIS2']
        ra0, dec0 = self.pix2wcs(0, H / 2.0)
        ra1, dec1 = self.pix2wcs(W - 1, H / 2.0)
        _, decB = self.pix2wcs(W / 2.0, 0)
        _, decT = self.pix2wcs(W / 2.0, H - 1)
        width = abs(self._cdelt1) * (W - 1) * np.cos(np.radians(dec0))
        return float(width), float(abs(decT - decB))

    def getXPixelSizeDeg(self):
        return abs(self._cdelt1)

    def getYPixelSizeDeg(self):
        return abs(self._cdelt2)

    def getImageMinMaxWCSCoords(self):
        W, H = self.header['NAXIS1'], self.header['NAXIS2']
        ra0, dec0 = self.pix2wcs(0.0, 0.0)
        ra1, dec1 = self.pix2wcs(W - 1.0, H - 1.0)
        # unwrap so that a tile straddling RA = 0 gives a contiguous interval
        ras = np.array([ra0, ra1])
        if abs(ra0 - ra1) > 180.0:
            ras = np.where(ras > 180.0, ras - 360.0, ras)
        return float(ras.min()), float(ras.max()), min(dec0, dec1), max(dec0, dec1)

    # -- helpers used by the template painter (tables shared verbatim by oracle and GPU paths) ----
    def rowDecRad(self):
        key = ('rowDec', self.header['NAXIS2'], self._crpix2, self._cdelt2)
        if getattr(self, '_rowDecKey', None) != key:
            H = self.header['NAXIS2']
            self._rowDec = np.radians((np.arange(H, dtype=np.float64) + 1.0 - self._crpix2) * self._cdelt2)
            self._rowDecKey = key
        return self._rowDec

    def colRARad(self):
        key = ('colRA', self.header['NAXIS1'], self._crval1, self._crpix1, self._cdelt1)
        if getattr(self, '_colRAKey', None) != key:
            W = self.header['NAXIS1']
            self._colRA = np.radians(self._crval1 + (np.arange(W, dtype=np.float64) + 1.0 - self._crpix1) * self._cdelt1)
            self._colRAKey = key
        return self._colRA


def calcAngSepDeg(RADeg1, decDeg1, RADeg2, decDeg2):
    """Great-circle separation (degrees); stands in for astCoords.calcAngSepDeg
    (used at nemo/filters.py:226-227, nemo/maps.py:1376-1377, nemo/photometry.py:201)."""
    ra1, dec1 = np.radians(RADeg1), np.radians(decDeg1)
    ra2, dec2 = np.radians(RADeg2), np.radians(decDeg2)
    h = np.sin((dec2 - dec1) / 2.0) ** 2 + np.cos(dec1) * np.cos(dec2) * np.sin((ra2 - ra1) / 2.0) ** 2
    return np.degrees(2.0 * np.arcsin(np.sqrt(np.clip(h, 0.0, 1.0))))
