"""Minimal FITS image reader / writer (primary HDU only, no compression).

The reference stores its filter cache as a FITS primary HDU (float32 [N,H,W] cube with SIGNORM / RWn header cards,
nemo/filters.py:761-790; kernel with BCKSCALE / APP_RA / APP_DEC, nemo/filters.py:866-885,1025-1037) through
astropy.io.fits, which is not available offline.  This module writes and reads that same on-disk layout
(2880-byte blocks, 80-character cards, big-endian data) with numpy only, so caches written here open in
astropy and vice versa for plain images.
"""
import numpy as np

_BITPIX = {np.dtype('uint8'): 8, np.dtype('int16'): 16, np.dtype('int32'): 32, np.dtype('int64'): 64,
           np.dtype('float32'): -32, np.dtype('float64'): -64}
_DTYPE = {8: '>u1', 16: '>i2', 32: '>i4', 64: '>i8', -32: '>f4', -64: '>f8'}
_SKIP = ('SIMPLE', 'BITPIX', 'NAXIS', 'EXTEND', 'END')


def _card(key, value, comment=None):
    key = str(key).upper()[:8]
    if isinstance(value, (bool, np.bool_)):
        v = "%20s" % ('T' if value else 'F')
    elif isinstance(value, (int, np.integer)):
        v = "%20d" % (int(value))
    elif isinstance(value, (float, np.floating)):
        v = "%20s" % (repr(float(value)).upper() if np.isfinite(value) else "'%s'" % (value))
    else:
        s = str(value).replace("'", "''")
        v = "'%-8s'" % (s)
    card = "%-8s= %s" % (key, v)
    if comment:
        card = card + " / " + comment
    return ("%-80s" % (card))[:80]


def writeFITS(fileName, data, header=None):
    data = np.asarray(data)
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    if data.dtype.newbyteorder('=') not in _BITPIX:
        data = data.astype(np.float64)
    bitpix = _BITPIX[data.dtype.newbyteorder('=')]
    cards = [_card('SIMPLE', True), _card('BITPIX', bitpix), _card('NAXIS', data.ndim)]
    for i, n in enumerate(data.shape[::-1]):
        cards.append(_card('NAXIS%d' % (i + 1), n))
    cards.append(_card('EXTEND', True))
    for k, v in (header or {}).items():
        ku = str(k).upper()
        if ku in _SKIP or ku.startswith('NAXIS') or v is None:
            continue
        cards.append(_card(ku, v))
    cards.append("%-80s" % ('END'))
    hdr = "".join(cards)
    hdr = hdr + " " * ((2880 - len(hdr) % 2880) % 2880)
    raw = data.astype(_DTYPE[bitpix]).tobytes()
    raw = raw + b"\0" * ((2880 - len(raw) % 2880) % 2880)
    with open(fileName, "wb") as f:
        f.write(hdr.encode('ascii'))
        f.write(raw)


def _parse_value(s):
    s = s.strip()
    if s.startswith("'"):
        end = s.find("'", 1)
        while end != -1 and end + 1 < len(s) and s[end + 1] == "'":
            end = s.find("'", end + 2)
        return s[1:end].replace("''", "'").rstrip()
    s = s.split('/')[0].strip()
    if s == 'T':
        return True
    if s == 'F':
        return False
    try:
        return int(s)
    except ValueError:
        return float(s.replace('D', 'E'))


def readFITS(fileName, headerOnly=False):
    """Returns (data, header dict) of the primary HDU."""
    header = {}
    with open(fileName, "rb") as f:
        done = False
        while not done:
            block = f.read(2880)
            if len(block) < 2880:
                raise Exception("truncated FITS header in %s" % (fileName))
            for i in range(0, 2880, 80):
                card = block[i:i + 80].decode('ascii', errors='replace')
                key = card[:8].strip()
                if key == 'END':
                    done = True
                    break
                if card[8:10] == '= ':
                    header[key] = _parse_value(card[10:])
        if headerOnly:
            return None, header
        naxis = header.get('NAXIS', 0)
        shape = tuple(header['NAXIS%d' % (i)] for i in range(naxis, 0, -1))
        if naxis == 0:
            return None, header
        dt = np.dtype(_DTYPE[header['BITPIX']])
        n = int(np.prod(shape))
        data = np.frombuffer(f.read(n * dt.itemsize), dtype=dt).reshape(shape)
        data = data.astype(dt.newbyteorder('='))
        if 'BSCALE' in header or 'BZERO' in header:
            data = data * header.get('BSCALE', 1.0) + header.get('BZERO', 0.0)
    return data, header
