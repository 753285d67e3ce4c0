"""Minimal FITS reader / writer: images, multi-extension files, binary tables, PLIO_1 tile-compressed integer images.

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
        naxis = header.get('NAXIS', 0)
        if naxis == 0 and header.get('EXTEND', False) == True:
            # an empty primary HDU: the image lives in the first extension (a tile-compressed mask, as maps.saveFITS and
            # astropy's CompImageHDU write them)
            f.close()
            for name, data, h in readHDUs(fileName)[1:]:
                if data is not None and not isinstance(data, dict):
                    return (None if headerOnly else data), h
            return None, header
        if headerOnly:
            return None, header
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


# ---- PLIO_1 tile compression (FITS standard 4.0, section 10: tiled image compression; algorithm: the IRAF pixel-list "line
# list" encoding, as cfitsio's pl_p2li / pl_l2pi implement it).  What the reference writes its masks with
# (astropy CompImageHDU(compression_type='PLIO_1'), nemo/maps.py:539-571, 2436-2476).  A line list is a sequence of 16-bit words:
# a 7-word header {0, 7, -100, length low, length high, 0, 0}, then instructions opcode << 12 | 12-bit operand acting on a
# running "high value" (1 at the start of a line):
#   0 ZN: n zeros            1 SH: high = operand + (next word << 12)      2 IH: high += n        3 DH: high -= n
#   4 HN: n pixels of high   5 PN: n - 1 zeros, then one pixel of high     6 IS: high += n, one pixel       7 DS: high -= n, one pixel
# Negative pixel values count as zero; values above 2^24 do not fit.  PARITY UNPINNED: neither astropy nor cfitsio is available
# offline - the writer follows the published format, and the only check is the round trip through the reader below.
def _plioEncodeLine(row):
    row = np.maximum(np.asarray(row, dtype=np.int64), 0)
    n = len(row)
    words = [0, 7, -100, 0, 0, 0, 0]
    if n > 0:
        change = np.flatnonzero(row[1:] != row[:-1]) + 1
        starts = np.concatenate([[0], change])
        lengths = np.diff(np.concatenate([starts, [n]]))
        values = row[starts]
        hi = 1
        nz = 0
        for v, ln in zip(values.tolist(), lengths.tolist()):
            if v == 0:
                nz += ln                      # (zeros wait for the run that follows them, or for the end of the line)
                continue
            single = False
            dv = v - hi
            if dv != 0:
                hi = v
                if abs(dv) > 4095:
                    words += [(v & 4095) + 4096, v // 4096]
                else:
                    words.append((-dv + 12288) if dv < 0 else (dv + 8192))
                    if ln == 1 and nz == 0:
                        words[-1] |= 16384    # IS / DS: change the high value and emit one pixel
                        single = True
            if not single:
                if nz > 0:
                    while nz > 0:
                        words.append(min(4095, nz))
                        nz -= 4095
                    if ln == 1:
                        words[-1] += 20481    # the last zero run becomes PN: its zeros, then this one pixel
                        single = True
                if not single:
                    while ln > 0:
                        words.append(min(4095, ln) + 16384)
                        ln -= 4095
            nz = 0
        while nz > 0:                         # zeros that end the line
            words.append(min(4095, nz))
            nz -= 4095
    words[3], words[4] = len(words) % 32768, len(words) // 32768
    return np.array(words, dtype=np.int16)


def _plioDecodeLine(words, n):
    words = np.asarray(words, dtype=np.int64)
    out = np.zeros(n, dtype=np.int64)
    if len(words) == 0:
        return out
    if words[2] > 0:                         # old-style header: length in word 3, three header words
        end, k = int(words[2]), 3
    else:
        end, k = int(words[4]) * 32768 + int(words[3]), int(words[1])
    hi, x = 1, 0
    while k < end:
        op, data = int(words[k]) >> 12, int(words[k]) & 4095
        k += 1
        if op == 0:
            x += data
        elif op == 1:
            hi = (int(words[k]) << 12) + data
            k += 1
        elif op == 2:
            hi += data
        elif op == 3:
            hi -= data
        elif op == 4:
            out[x:x + data] = hi
            x += data
        elif op == 5:
            x += data
            out[x - 1] = hi
        elif op in (6, 7):
            hi = hi + data if op == 6 else hi - data
            out[x] = hi
            x += 1
    return out


def _compressedImageHDU(name, data, header):
    """An integer image as a PLIO_1 tile-compressed binary-table extension, one tile per image row (the convention's default)."""
    data = np.asarray(data)
    if data.ndim != 2 or data.dtype.kind not in 'ui' or data.dtype.itemsize > 4:
        raise Exception("PLIO_1 compresses 2-D integer images of at most 32 bits")
    if data.size > 0 and int(data.max()) >= (1 << 24):
        raise Exception("PLIO_1 holds pixel values below 2^24")
    H, W = data.shape
    lines = [_plioEncodeLine(data[y]) for y in range(H)]
    lens = np.array([len(l) for l in lines], dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(lens * 2)[:-1]]) if H > 0 else np.zeros(0, dtype=np.int64)
    desc = np.empty((H, 2), dtype='>i4')
    desc[:, 0], desc[:, 1] = lens, offsets
    heap = np.concatenate(lines).astype('>i2').tobytes() if H > 0 else b""
    zbitpix = {1: 8, 2: 16, 4: 32}[data.dtype.itemsize]
    cards = [_card('XTENSION', 'BINTABLE'), _card('BITPIX', 8), _card('NAXIS', 2), _card('NAXIS1', 8), _card('NAXIS2', H),
             _card('PCOUNT', len(heap)), _card('GCOUNT', 1), _card('TFIELDS', 1), _card('TTYPE1', 'COMPRESSED_DATA'),
             _card('TFORM1', '1PI(%d)' % (int(lens.max()) if H > 0 else 0)), _card('ZIMAGE', True), _card('ZTILE1', W), _card('ZTILE2', 1),
             _card('ZCMPTYPE', 'PLIO_1'), _card('ZBITPIX', zbitpix), _card('ZNAXIS', 2), _card('ZNAXIS1', W), _card('ZNAXIS2', H)]
    if name is not None:
        cards.append(_card('EXTNAME', name))
    cards += [c for c in _userCards(header) if not c.startswith(('ZIMAGE', 'ZTILE', 'ZCMPTYPE', 'ZBITPIX', 'ZNAXIS', 'TTYPE', 'TFORM', 'THEAP'))]
    return _headerBytes(cards) + _pad(desc.tobytes() + heap)


def writeMEF(fileName, hdus, compressionType=None):
    """``hdus``: [(extension name, array, header dict)] -> an empty primary HDU followed by one IMAGE extension each - or, with
    ``compressionType`` 'PLIO_1' (integer images: masks), one tile-compressed image extension each."""
    if compressionType not in (None, 'PLIO_1'):
        raise Exception("compressionType '%s' is not written here (PLIO_1 for masks, or none)" % (compressionType))
    with open(fileName, "wb") as f:
        f.write(_headerBytes([_card('SIMPLE', True), _card('BITPIX', 8), _card('NAXIS', 0), _card('EXTEND', True)]))
        for name, data, header in hdus:
            f.write(_compressedImageHDU(name, data, header) if compressionType == 'PLIO_1' else _imageHDU(name, data, header))


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
            total = nbytes + int(header.get('PCOUNT', 0) or 0)          # (a binary table's heap follows its rows)
            block = f.read(total + (2880 - total % 2880) % 2880)
            raw = block[:nbytes]
            data = None
            if header.get('XTENSION', '').strip() == 'BINTABLE' and header.get('ZIMAGE', False) == True:
                if str(header.get('ZCMPTYPE', '')).strip() != 'PLIO_1':
                    raise Exception("tile compression '%s' is not read here (PLIO_1 only)" % (header.get('ZCMPTYPE')))
                H, W = int(header['ZNAXIS2']), int(header['ZNAXIS1'])
                if int(header.get('ZTILE1', W)) != W or int(header.get('ZTILE2', 1)) != 1:
                    raise Exception("PLIO_1 images are read with one tile per row only")
                heap0 = int(header.get('THEAP', nbytes))
                desc = np.frombuffer(raw, dtype='>i4').reshape(H, 2)
                zdt = {8: np.uint8, 16: np.int16, 32: np.int32}[int(header['ZBITPIX'])]
                data = np.zeros((H, W), dtype=zdt)
                for y in range(H):
                    n, off = int(desc[y, 0]), int(desc[y, 1])
                    words = np.frombuffer(block[heap0 + off:heap0 + off + 2 * n], dtype='>i2')
                    data[y] = _plioDecodeLine(words, W)
                header = dict(header, NAXIS=2, NAXIS1=W, NAXIS2=H, BITPIX=int(header['ZBITPIX']))
            elif header.get('XTENSION', '').strip() == 'BINTABLE':
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
