"""Container-only stand-in for the few cv2 entry points the reference hot path touches.

Used ONLY by tests/golden/make_golden.py to import the reference (which lives in /root/reference,
never in this repo) so that golden vectors can be produced.  PNG goes through PIL (lossless, so
decoded pixels are identical to what OpenCV would give).  Not product code, never shipped on the
hot path.
"""
import io
import numpy as np
from PIL import Image

__version__ = "4.0.0-shim"
IMREAD_GRAYSCALE = 0
IMREAD_COLOR = 1
COLOR_RGB2BGR = 4
COLOR_BGR2RGB = 4
INTER_NEAREST = 0
INTER_CUBIC = 2
BORDER_CONSTANT = 0


def imencode(ext, img):
    assert ext == ".png"
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(img)).save(buf, format="PNG")
    return True, np.frombuffer(buf.getvalue(), dtype=np.uint8)


def imdecode(raw, flag):
    im = Image.open(io.BytesIO(np.asarray(raw, dtype=np.uint8).tobytes()))
    if flag == IMREAD_GRAYSCALE:
        return np.array(im.convert("L"))
    return np.array(im.convert("RGB"))[:, :, ::-1].copy()


def cvtColor(img, code):
    return np.ascontiguousarray(img[:, :, ::-1])


def resize(img, size, interpolation=INTER_NEAREST):
    w, h = size
    H, W = img.shape[:2]
    if interpolation != INTER_NEAREST:
        raise NotImplementedError("shim only implements INTER_NEAREST")
    ys = np.minimum((np.arange(h) * (H / h)).astype(np.int64), H - 1)
    xs = np.minimum((np.arange(w) * (W / w)).astype(np.int64), W - 1)
    return img[ys][:, xs]


def imwrite(path, img):
    Image.fromarray(img if img.ndim == 2 else img[:, :, ::-1]).save(path)
    return True


def imread(path, flag=IMREAD_COLOR):
    try:
        im = Image.open(path)
    except (FileNotFoundError, OSError):
        return None
    if flag == IMREAD_GRAYSCALE:
        return np.array(im.convert("L"))
    return np.array(im.convert("RGB"))[:, :, ::-1].copy()
