"""
Minimal GeoTIFF reader / writer for the rasters either side of the hot path (no GDAL, no rasterio in this image).

Covers what the reference reads and writes (all of homonim's test rasters and its own output
profile, homonim/fuse.py:124-149): classic or BigTIFF, little / big endian, strips or tiles, planar ``separate`` or
``contig``, uncompressed or DEFLATE (zlib) with predictor none / horizontal differencing, 8 / 16 / 32 / 64-bit integer
and IEEE float samples; north-up geo-referencing from ModelPixelScale + ModelTiepoint or ModelTransformation; nodata
from GDAL_NODATA; EPSG code / citation from the GeoKey directory; the ``<GDALMetadata>`` items (where the reference keeps
its FUSE_* provenance, fuse.py:193-207).  The writer produces tiled, DEFLATE, band-separate files like the reference's
default output profile.  Everything else (other compressions, rotated grids, overviews, palettes) raises.
"""
import re
import struct
import zlib
from typing import Dict, NamedTuple, Optional
from xml.sax.saxutils import escape, unescape

import numpy as np

from homonim_amd.errors import IoError
from homonim_amd.geo import Affine, CRS

_TYPES = {1: 'B', 2: 'c', 3: 'H', 4: 'I', 5: 'II', 6: 'b', 7: 'B', 8: 'h', 9: 'i', 10: 'ii', 11: 'f', 12: 'd', 16: 'Q',
          17: 'q', 18: 'Q'}
_SAMPLE_DTYPES = {(1, 8): 'u1', (1, 16): 'u2', (1, 32): 'u4', (1, 64): 'u8', (2, 8): 'i1', (2, 16): 'i2', (2, 32): 'i4',
                  (2, 64): 'i8', (3, 32): 'f4', (3, 64): 'f8'}

T_WIDTH, T_HEIGHT, T_BITS, T_COMPRESSION, T_PHOTOMETRIC, T_STRIP_OFFSETS, T_SPP, T_ROWS_PER_STRIP = 256, 257, 258, 259, 262, 273, 277, 278
T_STRIP_COUNTS, T_PLANAR, T_PREDICTOR, T_TILE_W, T_TILE_H, T_TILE_OFFSETS, T_TILE_COUNTS, T_EXTRA, T_FORMAT = 279, 284, 317, 322, 323, 324, 325, 338, 339
T_PIXEL_SCALE, T_TIEPOINT, T_TRANSFORMATION, T_GEOKEYS, T_GEODOUBLES, T_GEOASCII, T_GDAL_METADATA, T_GDAL_NODATA = 33550, 33922, 34264, 34735, 34736, 34737, 42112, 42113


class TiffRaster(NamedTuple):
    array: np.ndarray            # (bands, height, width), the file's sample dtype
    transform: Affine
    crs: CRS
    nodata: Optional[float]
    metadata: Dict[str, str]     # dataset-level <GDALMetadata> items


def _read_ifd(buf: bytes):
    if buf[:2] == b'II':
        bo = '<'
    elif buf[:2] == b'MM':
        bo = '>'
    else:
        raise IoError('not a TIFF file')
    magic = struct.unpack(bo + 'H', buf[2:4])[0]
    if magic == 42:
        big, (off,) = False, struct.unpack(bo + 'I', buf[4:8])
        n, pos, esz, cfmt, vsz = struct.unpack(bo + 'H', buf[off:off + 2])[0], off + 2, 12, 'I', 4
    elif magic == 43:
        big, (off,) = True, struct.unpack(bo + 'Q', buf[8:16])
        n, pos, esz, cfmt, vsz = struct.unpack(bo + 'Q', buf[off:off + 8])[0], off + 8, 20, 'Q', 8
    else:
        raise IoError('not a TIFF file')
    tags = {}
    for i in range(n):
        e = buf[pos + i * esz: pos + (i + 1) * esz]
        code, typ = struct.unpack(bo + 'HH', e[:4])
        count = struct.unpack(bo + cfmt, e[4:4 + vsz])[0]
        if typ not in _TYPES:
            continue
        item = _TYPES[typ]
        size = struct.calcsize('=' + item) * count
        if size <= vsz:
            raw = e[4 + vsz:4 + vsz + size]
        else:
            (voff,) = struct.unpack(bo + cfmt, e[4 + vsz:4 + 2 * vsz])
            raw = buf[voff:voff + size]
        if typ == 2:
            tags[code] = raw.split(b'\0')[0].decode('latin-1') if code != T_GEOASCII else raw.decode('latin-1').rstrip('\0')
        elif typ in (5, 10):
            v = struct.unpack(bo + item[0] * (2 * count), raw)
            tags[code] = tuple(v[2 * k] / v[2 * k + 1] if v[2 * k + 1] else 0. for k in range(count))
        else:
            tags[code] = struct.unpack(bo + item * count, raw)
    return bo, tags


def _geo(tags, height):
    if T_TRANSFORMATION in tags:
        m = tags[T_TRANSFORMATION]
        if m[1] != 0 or m[4] != 0:
            raise IoError('rotated / sheared GeoTIFFs are not supported')
        tf = Affine(m[0], 0., m[3], 0., m[5], m[7])
    elif T_PIXEL_SCALE in tags and T_TIEPOINT in tags:
        sx, sy = tags[T_PIXEL_SCALE][:2]
        i, j, _, x, y, _ = tags[T_TIEPOINT][:6]
        tf = Affine(sx, 0., x - i * sx, 0., -sy, y + j * sy)
    else:
        tf = Affine.identity()
    name = None
    if T_GEOKEYS in tags:
        keys = tags[T_GEOKEYS]
        ascii_params, doubles = tags.get(T_GEOASCII, ''), tags.get(T_GEODOUBLES, ())
        entries = {keys[4 + 4 * k]: keys[5 + 4 * k: 8 + 4 * k] for k in range(keys[3])}

        def value(key):
            loc, cnt, off = entries[key]
            if loc == 0:
                return off
            if loc == T_GEODOUBLES:
                return doubles[off] if cnt == 1 else tuple(doubles[off:off + cnt])
            if loc == T_GEOASCII:
                return ascii_params[off:off + cnt].rstrip('|')
            return None

        model_type = value(1024) if 1024 in entries else None
        code_key = 2048 if model_type == 2 else 3072  # geographic / projected CS type
        if code_key in entries and entries[code_key][0] == 0 and value(code_key) not in (0, 32767):
            name = f'EPSG:{value(code_key)}'
        else:
            # user-defined CRS (all of the reference's test rasters): the label is the full key list, citations aside,
            # so two rasters compare equal exactly when their definitions do
            body = '; '.join(f'{k}={value(k)}' for k in sorted(entries) if k not in (1026, 2049, 3073))
            cite = next((value(k) for k in (1026, 3073, 2049) if k in entries and entries[k][0] == T_GEOASCII), '')
            # (a citation that already carries a key list is a label this module wrote: keep it, so it round-trips)
            name = cite if ('[' in cite and cite.endswith(']')) else (f'{cite} [{body}]' if cite else f'[{body}]')
    return tf, CRS(name) if name else CRS()


def _metadata(tags) -> Dict[str, str]:
    xml = tags.get(T_GDAL_METADATA)
    if not xml:
        return {}
    out = {}
    for m in re.finditer(r'<Item name="([^"]*)"([^>]*)>(.*?)</Item>', xml, flags=re.S):
        if 'sample=' not in m.group(2):  # dataset-level items only
            out[unescape(m.group(1), {'&quot;': '"'})] = unescape(m.group(3), {'&quot;': '"'})
    return out


def read_tiff(path) -> TiffRaster:
    """ Read the first image of a GeoTIFF into a (bands, height, width) array. """
    with open(path, 'rb') as f:
        buf = f.read()
    bo, t = _read_ifd(buf)
    w, h = t[T_WIDTH][0], t[T_HEIGHT][0]
    spp = t.get(T_SPP, (1,))[0]
    bits = t.get(T_BITS, (1,))
    fmt = t.get(T_FORMAT, (1,) * spp)
    if len(set(bits)) != 1 or len(set(fmt)) != 1 or (fmt[0], bits[0]) not in _SAMPLE_DTYPES:
        raise IoError(f'unsupported sample layout: bits {bits}, format {fmt}')
    dtype = np.dtype(bo + _SAMPLE_DTYPES[(fmt[0], bits[0])])
    compression = t.get(T_COMPRESSION, (1,))[0]
    if compression not in (1, 8, 32946):
        raise IoError(f'unsupported TIFF compression {compression} (none and DEFLATE are)')
    predictor = t.get(T_PREDICTOR, (1,))[0]
    if predictor not in (1, 2) or (predictor == 2 and dtype.kind == 'f'):
        raise IoError(f'unsupported TIFF predictor {predictor}')
    planar = t.get(T_PLANAR, (1,))[0]
    if T_TILE_OFFSETS in t:
        bw, bh, offs, cnts = t[T_TILE_W][0], t[T_TILE_H][0], t[T_TILE_OFFSETS], t[T_TILE_COUNTS]
    else:
        bw, bh = w, min(t.get(T_ROWS_PER_STRIP, (h,))[0], h)
        offs, cnts = t[T_STRIP_OFFSETS], t[T_STRIP_COUNTS]
    across, down = -(-w // bw), -(-h // bh)
    chunk_spp = 1 if planar == 2 else spp
    tiled = T_TILE_OFFSETS in t
    out = np.empty((spp, h, w), dtype.newbyteorder('='))
    for idx, (off, cnt) in enumerate(zip(offs, cnts)):
        plane, rem = divmod(idx, across * down) if planar == 2 else (0, idx)
        by, bx = divmod(rem, across)
        rows = bh if tiled else min(bh, h - by * bh)
        raw = buf[off:off + cnt]
        if compression != 1:
            raw = zlib.decompress(raw)
        block = np.frombuffer(raw, dtype, count=rows * bw * chunk_spp).reshape(rows, bw, chunk_spp)
        if predictor == 2:
            block = np.cumsum(block, axis=1, dtype=dtype.newbyteorder('='))
        y0, x0 = by * bh, bx * bw
        hh, ww = min(rows, h - y0), min(bw, w - x0)
        if planar == 2:
            out[plane, y0:y0 + hh, x0:x0 + ww] = block[:hh, :ww, 0]
        else:
            out[:, y0:y0 + hh, x0:x0 + ww] = np.moveaxis(block[:hh, :ww, :], 2, 0)
    nodata = None
    if T_GDAL_NODATA in t:
        try:
            nodata = float(t[T_GDAL_NODATA].strip())
        except ValueError:
            nodata = None
    tf, crs = _geo(t, h)
    return TiffRaster(out, tf, crs, nodata, _metadata(t))


# ----------------------------------------------------------------------------------------------------------------------
def write_tiff(path, array: np.ndarray, transform: Affine, crs: Optional[CRS] = None, nodata: Optional[float] = None,
               metadata: Optional[Dict[str, str]] = None, tile: int = 512, compress: bool = True):
    """ Write (bands, height, width) as a classic little-endian GeoTIFF: tiled, DEFLATE, band-separate -- the reference's
    default output profile (homonim/fuse.py:124-149: tiled 512 x 512, compress=deflate, interleave=band). """
    a = np.asarray(array)
    if a.ndim == 2:
        a = a[None]
    key = {('u', 1): (1, 8), ('u', 2): (1, 16), ('u', 4): (1, 32), ('i', 1): (2, 8), ('i', 2): (2, 16), ('i', 4): (2, 32),
           ('f', 4): (3, 32), ('f', 8): (3, 64)}.get((a.dtype.kind, a.dtype.itemsize))
    if key is None:
        raise IoError(f"unsupported dtype '{a.dtype}'")
    if transform.b != 0 or transform.d != 0:
        raise IoError('rotated / sheared grids are not supported')
    fmt, bits = key
    nb, h, w = a.shape
    a = a.astype(a.dtype.newbyteorder('<'), copy=False)
    tile = max(16, (int(tile) + 15) // 16 * 16)
    across, down = -(-w // tile), -(-h // tile)
    chunks = []
    for b in range(nb):
        for by in range(down):
            for bx in range(across):
                blk = np.zeros((tile, tile), a.dtype)
                part = a[b, by * tile:(by + 1) * tile, bx * tile:(bx + 1) * tile]
                blk[:part.shape[0], :part.shape[1]] = part
                raw = blk.tobytes()
                chunks.append(zlib.compress(raw, 6) if compress else raw)

    def ascii_(s):
        return s.encode('latin-1', 'replace') + b'\0'

    entries = []  # (code, type, count, payload bytes)

    def add(code, typ, values):
        if typ == 2:
            payload = ascii_(values)
            entries.append((code, 2, len(payload), payload))
        else:
            payload = struct.pack('<' + _TYPES[typ] * len(values), *values)
            entries.append((code, typ, len(values), payload))

    add(T_WIDTH, 4, [w]), add(T_HEIGHT, 4, [h]), add(T_BITS, 3, [bits] * nb)
    add(T_COMPRESSION, 3, [8 if compress else 1]), add(T_PHOTOMETRIC, 3, [1]), add(T_SPP, 3, [nb]), add(T_PLANAR, 3, [2])
    add(T_TILE_W, 4, [tile]), add(T_TILE_H, 4, [tile])
    add(T_TILE_OFFSETS, 4, [0] * len(chunks)), add(T_TILE_COUNTS, 4, [len(c) for c in chunks])
    if nb > 1:
        add(T_EXTRA, 3, [0] * (nb - 1))
    add(T_FORMAT, 3, [fmt] * nb)
    add(T_PIXEL_SCALE, 12, [float(transform.a), float(-transform.e), 0.])
    add(T_TIEPOINT, 12, [0., 0., 0., float(transform.c), float(transform.f), 0.])
    name = crs.to_string() if crs is not None else ''
    m = re.fullmatch(r'EPSG:(\d+)', name or '')
    if m:
        geographic = int(m.group(1)) in (4326, 4269, 4258)
        add(T_GEOKEYS, 3, [1, 1, 0, 3, 1024, 0, 1, 2 if geographic else 1, 1025, 0, 1, 1,
                           2048 if geographic else 3072, 0, 1, int(m.group(1))])
    elif name:
        add(T_GEOKEYS, 3, [1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 1026, T_GEOASCII, len(name) + 1, 0])
        add(T_GEOASCII, 2, name + '|')
    if metadata:
        items = ''.join(f'  <Item name="{escape(str(k), {chr(34): "&quot;"})}">{escape(str(v))}</Item>\n' for k, v in metadata.items())
        add(T_GDAL_METADATA, 2, f'<GDALMetadata>\n{items}</GDALMetadata>\n')
    if nodata is not None:
        add(T_GDAL_NODATA, 2, 'nan' if (isinstance(nodata, float) and np.isnan(nodata)) else repr(float(nodata)) if a.dtype.kind == 'f' else str(int(nodata)))
    entries.sort(key=lambda e: e[0])

    ifd_off = 8
    ifd_size = 2 + 12 * len(entries) + 4
    extra_off = ifd_off + ifd_size
    extras, placed = [], {}
    for code, typ, count, payload in entries:
        if len(payload) > 4:
            placed[code] = extra_off
            pad = payload + b'\0' * (len(payload) % 2)
            extras.append(pad)
            extra_off += len(pad)
    data_off = extra_off
    offsets, pos = [], data_off
    for c in chunks:
        offsets.append(pos)
        pos += len(c) + (len(c) % 2)
    if pos >= 2 ** 32:
        raise IoError('raster too large for a classic TIFF (BigTIFF writing is not built)')
    off_payload = struct.pack('<' + 'I' * len(offsets), *offsets)
    with open(path, 'wb') as f:
        f.write(b'II' + struct.pack('<HI', 42, ifd_off))
        f.write(struct.pack('<H', len(entries)))
        for code, typ, count, payload in entries:
            if code == T_TILE_OFFSETS:
                payload = off_payload
            f.write(struct.pack('<HHI', code, typ, count))
            f.write(payload.ljust(4, b'\0') if len(payload) <= 4 else struct.pack('<I', placed[code]))
        f.write(struct.pack('<I', 0))
        for (code, typ, count, payload), _ in zip([e for e in entries if len(e[3]) > 4], extras):
            if code == T_TILE_OFFSETS:
                payload = off_payload
            f.write(payload + b'\0' * (len(payload) % 2))
        for c in chunks:
            f.write(c + b'\0' * (len(c) % 2))
