"""8-bit grayscale PNG codec for the pickled hand-off between pipeline steps (compressed_frames / clean_binary are lists of
PNG byte arrays, FCN_lecturenet_binarizer.py:56,62; helper.py:31; cc_stability_estimator.py:678).

File-format edge, not hot path: zlib does the work.  Any valid PNG is acceptable to the consumers (cv2.imdecode); pixels
are what is compared, never PNG bytes.
"""
import struct
import zlib

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def encode_gray8(img, level=1):
    """uint8 [H,W] -> numpy uint8 array holding a PNG file (filter 0 on every row)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    raw = np.zeros((h, w + 1), np.uint8)
    raw[:, 1:] = img
    data = _SIG + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + \
        _chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + _chunk(b"IEND", b"")
    return np.frombuffer(data, np.uint8)


def _paeth_row(raw, up):
    out = np.zeros(len(raw), np.int32)
    a = c = 0
    for i in range(len(raw)):
        b = int(up[i])
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
        a = (int(raw[i]) + pr) & 255
        out[i] = a
        c = b
    return out.astype(np.uint8)


def decode_gray8(data):
    """PNG bytes (8-bit grayscale, non-interlaced) -> uint8 [H,W].  Other PNG flavours go through cv2 / PIL when present."""
    buf = bytes(np.asarray(data, np.uint8).tobytes()) if not isinstance(data, (bytes, bytearray)) else bytes(data)
    assert buf[:8] == _SIG, "not a PNG"
    pos, idat, ihdr = 8, [], None
    while pos < len(buf):
        n, tag = struct.unpack(">I4s", buf[pos:pos + 8])
        body = buf[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = ihdr
    if depth != 8 or ctype != 0 or interlace != 0:
        return _decode_fallback(buf)
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, w + 1)
    filt, body = raw[:, 0], raw[:, 1:]
    if not filt.any():
        return body.copy()
    out = np.zeros((h, w), np.uint8)
    prev = np.zeros(w, np.uint8)
    for y in range(h):
        f = filt[y]
        if f == 0:
            row = body[y]
        elif f == 1:
            row = (np.cumsum(body[y], dtype=np.uint32) & 255).astype(np.uint8)
        elif f == 2:
            row = body[y] + prev
        elif f == 3:
            row = np.zeros(w, np.uint8)
            a = 0
            for i in range(w):
                a = (int(body[y, i]) + ((a + int(prev[i])) >> 1)) & 255
                row[i] = a
        elif f == 4:
            row = _paeth_row(body[y], prev)
        else:
            raise ValueError("bad PNG filter %d" % f)
        out[y] = row
        prev = out[y]
    return out


def _decode_fallback(buf):
    try:
        import cv2
        return cv2.imdecode(np.frombuffer(buf, np.uint8), cv2.IMREAD_GRAYSCALE)
    except ImportError:
        import io
        from PIL import Image
        return np.array(Image.open(io.BytesIO(buf)).convert("L"))
