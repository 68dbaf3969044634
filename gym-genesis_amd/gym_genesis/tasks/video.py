"""Writing a recorded episode to a file (``cam.stop_recording(save_to_filename=..., fps=...)``:
/root/reference/gym_genesis/env.py:71-79; Genesis hands the frames to an external H.264 encoder, which this image does not have).

``.mp4`` / ``.mov`` / ``.m4v``: an ISO base media file with ONE video track of Motion-JPEG samples (sample entry ``mp4v`` with an
``esds`` whose object type is 0x6C = ISO/IEC 10918-1, the way ffmpeg muxes ``-c:v mjpeg`` into mp4): every frame is an independent
JPEG (Pillow), every sample a sync sample, one chunk.  ``.gif``: Pillow's animated GIF.  Host-side only, off the ``env.step()`` path.
"""
import io
import os
import struct
from typing import Sequence

import numpy as np


def _box(kind: bytes, *payload: bytes) -> bytes:
    body = b"".join(payload)
    return struct.pack(">I4s", 8 + len(body), kind) + body


def _full(kind: bytes, version: int, flags: int, *payload: bytes) -> bytes:
    return _box(kind, struct.pack(">I", version << 24 | flags), *payload)


def _descr(tag: int, payload: bytes) -> bytes:
    assert len(payload) < 128
    return bytes([tag, len(payload)]) + payload


_MATRIX = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


def write_mjpeg_mp4(path: str, frames: Sequence[np.ndarray], fps: float = 60.0, quality: int = 90) -> int:
    """frames: (H, W, 3) uint8 arrays of one size.  Returns the number of bytes written."""
    from PIL import Image

    if not len(frames):
        raise ValueError("no frames were recorded")
    h, w = frames[0].shape[:2]
    jpegs = []
    for f in frames:
        f = np.ascontiguousarray(f, dtype=np.uint8)
        if f.shape != (h, w, 3):
            raise ValueError(f"frame of shape {f.shape} in a recording of {(h, w, 3)}")
        buf = io.BytesIO()
        Image.fromarray(f).save(buf, format="JPEG", quality=int(quality))
        jpegs.append(buf.getvalue())
    n = len(jpegs)
    delta = 1000
    timescale = max(1, int(round(float(fps) * delta)))
    duration = n * delta
    ftyp = _box(b"ftyp", b"isom", struct.pack(">I", 512), b"isomiso2mp41")
    mdat = _box(b"mdat", *jpegs)
    first = len(ftyp) + 8  # file offset of the first sample (the only chunk)
    if first + len(mdat) >= 1 << 32:
        raise ValueError("recording larger than 4 GiB")
    esds = _full(b"esds", 0, 0, _descr(3, struct.pack(">HB", 1, 0) +
                                      _descr(4, struct.pack(">BB", 0x6C, 0x11) + b"\0\0\0" + struct.pack(">II", 0, 0)) + _descr(6, b"\x02")))
    entry = _box(b"mp4v", b"\0" * 6, struct.pack(">H", 1), b"\0" * 16, struct.pack(">HHIIIH", w, h, 0x480000, 0x480000, 0, 1),
                 b"\0" * 32, struct.pack(">Hh", 24, -1), esds)
    stbl = _box(b"stbl",
                _full(b"stsd", 0, 0, struct.pack(">I", 1), entry),
                _full(b"stts", 0, 0, struct.pack(">III", 1, n, delta)),
                _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, n, 1)),
                _full(b"stsz", 0, 0, struct.pack(">II", 0, n), struct.pack(f">{n}I", *[len(j) for j in jpegs])),
                _full(b"stco", 0, 0, struct.pack(">II", 1, first)))
    minf = _box(b"minf", _full(b"vmhd", 0, 1, b"\0" * 8),
                _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1), _full(b"url ", 0, 1))), stbl)
    mdia = _box(b"mdia", _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, timescale, duration, 0x55C4, 0)),
                _full(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"vide"), b"\0" * 12, b"VideoHandler\0"), minf)
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, duration), b"\0" * 8, struct.pack(">hhhH", 0, 0, 0, 0), _MATRIX,
                 struct.pack(">II", w << 16, h << 16))
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIIIIH", 0, 0, timescale, duration, 0x10000, 0x100), b"\0" * 10, _MATRIX, b"\0" * 24,
                 struct.pack(">I", 2))
    moov = _box(b"moov", mvhd, _box(b"trak", tkhd, mdia))
    with open(path, "wb") as f:
        f.write(ftyp)
        f.write(mdat)
        f.write(moov)
    return len(ftyp) + len(mdat) + len(moov)


def write_video(path: str, frames: Sequence[np.ndarray], fps: float = 60.0) -> int:
    ext = os.path.splitext(path)[1].lower()
    if ext == ".gif":
        from PIL import Image

        if not len(frames):
            raise ValueError("no frames were recorded")
        ims = [Image.fromarray(np.ascontiguousarray(f, dtype=np.uint8)) for f in frames]
        ims[0].save(path, save_all=True, append_images=ims[1:], duration=max(1, int(round(1000.0 / float(fps)))), loop=0)
        return os.path.getsize(path)
    if ext in (".mp4", ".mov", ".m4v", ""):
        return write_mjpeg_mp4(path, frames, fps)
    raise ValueError(f"cannot write {ext!r} files: use .mp4 (Motion-JPEG) or .gif")


def read_mjpeg_mp4(path: str):
    """The inverse, for the tests: (frames as (H, W, 3) uint8 arrays, fps, (width, height) of the track header)."""
    from PIL import Image

    data = open(path, "rb").read()

    def boxes(lo, hi):
        while lo + 8 <= hi:
            size, kind = struct.unpack(">I4s", data[lo:lo + 8])
            yield kind, lo + 8, lo + size
            lo += size

    def find(lo, hi, *kinds):
        for k in kinds:
            for kind, a, b in boxes(lo, hi):
                if kind == k:
                    lo, hi = a, b
                    break
            else:
                raise KeyError(k)
        return lo, hi

    a, b = find(0, len(data), b"moov", b"trak", b"mdia")
    m0, _ = find(a, b, b"mdhd")
    timescale, _ = struct.unpack(">II", data[m0 + 12:m0 + 20])
    t0, _ = find(*find(0, len(data), b"moov", b"trak"), b"tkhd")
    wh = struct.unpack(">II", data[t0 + 76:t0 + 84])
    s0, s1 = find(a, b, b"minf", b"stbl")
    z0, _ = find(s0, s1, b"stsz")
    n = struct.unpack(">I", data[z0 + 8:z0 + 12])[0]
    sizes = struct.unpack(f">{n}I", data[z0 + 12:z0 + 12 + 4 * n])
    c0, _ = find(s0, s1, b"stco")
    off = struct.unpack(">I", data[c0 + 8:c0 + 12])[0]
    d0, _ = find(s0, s1, b"stts")
    delta = struct.unpack(">I", data[d0 + 12:d0 + 16])[0]
    frames = []
    for sz in sizes:
        frames.append(np.asarray(Image.open(io.BytesIO(data[off:off + sz])).convert("RGB")))
        off += sz
    return frames, timescale / delta, (wh[0] >> 16, wh[1] >> 16)
