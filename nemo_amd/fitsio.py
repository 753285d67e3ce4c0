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


# ------------------------------------------------------------------------------------------------------------
# Multi-extension files and binary tables: what the path's products are stored as (nemo/maps.py:539-571 tile images in one
# MEF; nemo/catalogs.py:675-701 and nemo/pipelines.py:496-503 catalogues and RMS tables as FITS tables).  Tile compression
# (PLIO_1 / RICE_1, astropy CompImageHDU) is not written: masks go out as plain uint8 images, which every reader of the
# compressed form also reads.
def _pad(raw, fill=b"\0"):
    return raw + fill * ((2880 - len(raw) % 2880) % 2880)


def _headerBytes(cards):
    hdr = "".join(cards + ["%-80s" % ('END')])
    return _pad(hdr.encode('ascii'), b" ")


def _userCards(header):
    cards = []
    for k, v in (header or {}).items():
        ku = str(k).upper()
        if ku in _SKIP or ku.startswith('NAXIS') or ku in ('XTENSION', 'PCOUNT', 'GCOUNT', 'EXTNAME', 'TFIELDS') or v is None:
            continue
        cards.append(_card(ku, v))
    return cards


def _imageHDU(name, data, header, primary=False):
    data = np.asarray(data)
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    if data.dtype.newbyteorder('=') not in _BITPIX:
        data = data.astype(np.float64)
    bitpix = _BITPIX[data.dtype.newbyteorder('=')]
    cards = [_card('SIMPLE', True)] if primary else [_card('XTENSION', 'IMAGE')]
    cards += [_card('BITPIX', bitpix), _card('NAXIS', data.ndim)]
    cards += [_card('NAXIS%d' % (i + 1), n) for i, n in enumerate(data.shape[::-1])]
    cards += [_card('EXTEND', True)] if primary else [_card('PCOUNT', 0), _card('GCOUNT', 1)]
    if name is not None and not primary:
        cards.append(_card('EXTNAME', name))
    cards += _userCards(header)
    return _headerBytes(cards) + _pad(data.astype(_DTYPE[bitpix]).tobytes())


def writeMEF(fileName, hdus):
    """``hdus``: [(extension name, array, header dict)] -> an empty primary HDU followed by one IMAGE extension each."""
    with open(fileName, "wb") as f:
        f.write(_headerBytes([_card('SIMPLE', True), _card('BITPIX', 8), _card('NAXIS', 0), _card('EXTEND', True)]))
        for name, data, header in hdus:
            f.write(_imageHDU(name, data, header))


_TFORM = {'f8': ('D', '>f8'), 'f4': ('E', '>f4'), 'i8': ('K', '>i8'), 'i4': ('J', '>i4'), 'i2': ('I', '>i2'), 'u1': ('B', '>u1')}


def writeTable(fileName, cols, meta=None, extName=None):
    """A binary table extension from named columns of equal length (numbers, booleans, strings)."""
    names = list(cols.keys())
    arrays, forms, fields = [], [], []
    for k in names:
        a = np.asarray(cols[k])
        if a.dtype.kind == 'b':
            arrays.append(np.where(a, b'T', b'F').astype('S1'))
            forms.append('L')
            fields.append((k, 'S1'))
        elif a.dtype.kind in 'US':
            b = np.char.encode(a.astype('U'), 'ascii', 'replace') if a.dtype.kind == 'U' else a
            width = max(1, b.dtype.itemsize)
            arrays.append(b.astype('S%d' % width))
            forms.append('%dA' % width)
            fields.append((k, 'S%d' % width))
        else:
            key = a.dtype.newbyteorder('=').str[1:]
            if key not in _TFORM:
                a = a.astype(np.float64) if a.dtype.kind == 'f' else a.astype(np.int64)
                key = a.dtype.str[1:]
            arrays.append(a.astype(_TFORM[key][1]))
            forms.append(_TFORM[key][0])
            fields.append((k, _TFORM[key][1]))
    n = len(arrays[0]) if arrays else 0
    rec = np.zeros(n, dtype=np.dtype([(str(i), f[1]) for i, f in enumerate(fields)]))
    for i, a in enumerate(arrays):
        rec[str(i)] = a
    cards = [_card('XTENSION', 'BINTABLE'), _card('BITPIX', 8), _card('NAXIS', 2), _card('NAXIS1', rec.dtype.itemsize), _card('NAXIS2', n),
             _card('PCOUNT', 0), _card('GCOUNT', 1), _card('TFIELDS', len(names))]
    for i, (k, form) in enumerate(zip(names, forms)):
        cards += [_card('TTYPE%d' % (i + 1), k), _card('TFORM%d' % (i + 1), form)]
    if extName is not None:
        cards.append(_card('EXTNAME', extName))
    cards += _userCards(meta)
    with open(fileName, "wb") as f:
        f.write(_headerBytes([_card('SIMPLE', True), _card('BITPIX', 8), _card('NAXIS', 0), _card('EXTEND', True)]))
        f.write(_headerBytes(cards) + _pad(rec.tobytes()))


def _readHeader(f):
    header = {}
    while True:
        block = f.read(2880)
        if len(block) == 0:
            return None
        if len(block) < 2880:
            raise Exception("truncated FITS header")
        for i in range(0, 2880, 80):
            card = block[i:i + 80].decode('ascii', errors='replace')
            key = card[:8].strip()
            if key == 'END':
                return header
            if card[8:10] == '= ':
                header[key] = _parse_value(card[10:])


def readHDUs(fileName):
    """Every HDU of a file: [(EXTNAME or None, data, header)]; images as arrays, binary tables as {column: array}."""
    out = []
    with open(fileName, "rb") as f:
        while True:
            header = _readHeader(f)
            if header is None:
                break
            naxis = header.get('NAXIS', 0)
            shape = tuple(header['NAXIS%d' % (i)] for i in range(naxis, 0, -1))
            nbytes = (abs(header['BITPIX']) // 8) * int(np.prod(shape)) if naxis > 0 else 0
            raw = f.read(nbytes + (2880 - nbytes % 2880) % 2880)[:nbytes]
            data = None
            if header.get('XTENSION', '').strip() == 'BINTABLE':
                fields = []
                for i in range(1, header['TFIELDS'] + 1):
                    form = str(header['TFORM%d' % i]).strip()
                    code, rep = form[-1], (int(form[:-1]) if form[:-1] else 1)
                    dt = {'D': '>f8', 'E': '>f4', 'K': '>i8', 'J': '>i4', 'I': '>i2', 'B': '>u1', 'L': 'S1', 'A': 'S%d' % rep}[code]
                    fields.append((str(header['TTYPE%d' % i]).strip(), dt, code))
                rec = np.frombuffer(raw, dtype=np.dtype([(str(i), fd[1]) for i, fd in enumerate(fields)]))
                data = {}
                for i, (name, dt, code) in enumerate(fields):
                    col = rec[str(i)]
                    if code == 'L':
                        col = col == b'T'
                    elif code == 'A':
                        col = np.char.rstrip(np.char.decode(col, 'ascii'))
                    else:
                        col = col.astype(np.dtype(dt).newbyteorder('='))
                    data[name] = col
            elif naxis > 0:
                dt = np.dtype(_DTYPE[header['BITPIX']])
                data = np.frombuffer(raw, dtype=dt).reshape(shape).astype(dt.newbyteorder('='))
            name = header.get('EXTNAME')
            out.append((name.strip() if isinstance(name, str) else name, data, header))
    return out


def readTable(fileName, ext=1):
    """({column: array}, header) of a binary table extension (by index, or by EXTNAME)."""
    hdus = readHDUs(fileName)
    if isinstance(ext, str):
        for name, data, header in hdus:
            if name == ext:
                return data, header
        raise KeyError(ext)
    return hdus[ext][1], hdus[ext][2]
