"""Minimal PNG file IO for the dataset loader and the frame writer.

The reference reads its images with ``imageio.imread`` (datasets/blender.py:78) and writes frames with
``imageio.imwrite`` (scripts/render_nerf.py:314).  Here Pillow decodes when it is importable; otherwise a
small decoder for the files the Blender scenes contain (8-bit grey / RGB / RGBA, non-interlaced) is
used.  Writing needs nothing but zlib.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path: str, image: np.ndarray, compress_level: int = 6) -> None:
    """uint8 [H, W], [H, W, 1], [H, W, 3] or [H, W, 4] -> 8-bit PNG (filter type 0 on every row)."""
    a = np.asarray(image)
    if a.dtype != np.uint8:
        raise TypeError(f"write_png takes uint8 data, got {a.dtype}")
    if a.ndim == 2:
        a = a[..., None]
    if a.ndim != 3 or a.shape[2] not in (1, 3, 4):
        raise ValueError(f"write_png takes [H,W], [H,W,1|3|4], got {a.shape}")
    h, w, c = a.shape
    colour = {1: 0, 3: 2, 4: 6}[c]
    rows = np.zeros((h, 1 + w * c), dtype=np.uint8)
    rows[:, 1:] = a.reshape(h, w * c)
    body = (_SIG + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, colour, 0, 0, 0))
            + _chunk(b"IDAT", zlib.compress(rows.tobytes(), compress_level)) + _chunk(b"IEND", b""))
    with open(path, "wb") as f:
        f.write(body)


def _decode(raw: bytes) -> np.ndarray:
    if raw[:8] != _SIG:
        raise ValueError("not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(raw):
        (n,), tag = struct.unpack(">I", raw[pos:pos + 4]), raw[pos + 4:pos + 8]
        data = raw[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif tag == b"IDAT":
            idat.append(data)
        elif tag == b"IEND":
            break
    if hdr is None:
        raise ValueError("PNG without IHDR")
    w, h, depth, colour, _, _, interlace = hdr
    if depth != 8 or colour not in (0, 2, 6) or interlace != 0:
        raise ValueError(f"unsupported PNG (bit depth {depth}, colour type {colour}, interlace {interlace}); "
                         "install Pillow for other variants")
    c = _CHANNELS[colour]
    data = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * c)
    out = np.zeros((h, w, c), dtype=np.uint8)
    prior = np.zeros((w, c), dtype=np.int32)
    for y in range(h):
        ft = int(data[y, 0])
        line = data[y, 1:].reshape(w, c).astype(np.int32)
        if ft == 0:
            rec = line
        elif ft == 1:                               # Sub: running sum along the row
            rec = np.cumsum(line, axis=0) & 255
        elif ft == 2:                               # Up
            rec = (line + prior) & 255
        elif ft in (3, 4):                          # Average / Paeth: left neighbour is sequential
            rec = np.zeros_like(line)
            left = np.zeros(c, dtype=np.int32)
            upleft = np.zeros(c, dtype=np.int32)
            for x in range(w):
                up = prior[x]
                if ft == 3:
                    pred = (left + up) >> 1
                else:
                    p = left + up - upleft
                    pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - upleft)
                    pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
                left = (line[x] + pred) & 255
                rec[x] = left
                upleft = up
        else:
            raise ValueError(f"bad PNG filter type {ft}")
        out[y] = rec
        prior = rec
    return out


def read_png(path: str) -> np.ndarray:
    """8-bit PNG -> uint8 [H, W, C] (C = 1, 3 or 4; palette images are expanded by Pillow)."""
    try:
        from PIL import Image
    except ImportError:
        Image = None
    if Image is not None:
        with Image.open(path) as im:
            if im.mode not in ("L", "RGB", "RGBA"):
                im = im.convert("RGBA" if "A" in im.getbands() or "transparency" in im.info else "RGB")
            a = np.asarray(im, dtype=np.uint8)
        return a[..., None] if a.ndim == 2 else a
    with open(path, "rb") as f:
        return _decode(f.read())
