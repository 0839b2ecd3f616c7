"""Motion-JPEG AVI reader / writer in pure Python (PIL for the JPEG frames): the video container the CLIs can read and
write when OpenCV is not installed.

The reference decodes its input with cv2.VideoCapture (/root/reference/demo_video.py:78-110) and exports the annotated
frames with cv2.VideoWriter as MP4V (demo_video.py:25-43).  Neither codec can be produced without OpenCV / FFmpeg; an
AVI whose video stream is MJPG (every frame an independent baseline JPEG) needs nothing but a RIFF walker, and plays in
any player.  Layout written: RIFF 'AVI ' { LIST hdrl { avih, LIST strl { strh(vids/MJPG), strf(BITMAPINFOHEADER) } },
LIST movi { 00dc ... }, idx1 }.
"""
import io
import struct

import numpy as np


def _chunk(fourcc, payload):
    pad = b"\x00" if len(payload) & 1 else b""
    return fourcc + struct.pack("<I", len(payload)) + payload + pad


def _list(kind, payload):
    return b"LIST" + struct.pack("<I", len(payload) + 4) + kind + payload


def write_mjpeg_avi(path, frames, fps, quality=92):
    """frames: iterable of (H,W,3) uint8 RGB arrays of equal size.  Returns the number of frames written."""
    from PIL import Image
    jpegs = []
    size = None
    for fr in frames:
        a = np.asarray(fr, dtype=np.uint8)
        if size is None:
            size = (a.shape[1], a.shape[0])
        elif (a.shape[1], a.shape[0]) != size:
            raise ValueError("write_mjpeg_avi: frames must have equal size")
        buf = io.BytesIO()
        Image.fromarray(a).save(buf, format="JPEG", quality=quality)
        jpegs.append(buf.getvalue())
    if not jpegs:
        raise ValueError("write_mjpeg_avi: no frames")
    w, h = size
    n = len(jpegs)
    biggest = max(len(j) for j in jpegs)
    scale = 1000
    rate = int(round(float(fps) * scale))
    avih = struct.pack("<14I", int(round(1e6 / float(fps))), int(biggest * float(fps)), 0, 0x10, n, 0, 1, biggest, w, h, 0, 0, 0, 0)
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII4h", 0, 0, 0, 0, scale, rate, 0, n, biggest, 0xFFFFFFFF, 0, 0, 0, w, h)
    strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, 24, b"MJPG", w * h * 3, 0, 0, 0, 0)
    hdrl = _list(b"hdrl", _chunk(b"avih", avih) + _list(b"strl", _chunk(b"strh", strh) + _chunk(b"strf", strf)))
    movi_payload = b""
    index = b""
    off = 4                                    # offsets count from the 'movi' fourcc
    parts = []
    for j in jpegs:
        c = _chunk(b"00dc", j)
        index += b"00dc" + struct.pack("<III", 0x10, off, len(j))
        off += len(c)
        parts.append(c)
    movi_payload = b"".join(parts)
    body = b"AVI " + hdrl + _list(b"movi", movi_payload) + _chunk(b"idx1", index)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    return n


def _walk(buf, start, end):
    """Yield (fourcc, payload_start, payload_size) of the chunks in buf[start:end]."""
    p = start
    while p + 8 <= end:
        cc = bytes(buf[p:p + 4])
        (sz,) = struct.unpack_from("<I", buf, p + 4)
        yield cc, p + 8, sz
        p += 8 + sz + (sz & 1)


def read_mjpeg_avi(path):
    """-> (fps, random-access sequence of (H,W,3) uint8 RGB frames, frame count).  Raises ValueError for anything that
    is not an AVI with an MJPG video stream."""
    data = np.memmap(path, dtype=np.uint8, mode="r")
    if len(data) < 12 or bytes(data[0:4]) != b"RIFF" or bytes(data[8:12]) != b"AVI ":
        raise ValueError("%s: not a RIFF AVI file" % path)
    fps = None
    handler = None
    spans = []
    for cc, ps, sz in _walk(data, 12, len(data)):
        if cc != b"LIST":
            continue
        kind = bytes(data[ps:ps + 4])
        if kind == b"hdrl":
            for c2, p2, s2 in _walk(data, ps + 4, ps + sz):
                if c2 == b"avih":
                    (usec,) = struct.unpack_from("<I", data, p2)
                    if usec:
                        fps = 1e6 / usec
                elif c2 == b"LIST" and bytes(data[p2:p2 + 4]) == b"strl":
                    for c3, p3, s3 in _walk(data, p2 + 4, p2 + s2):
                        if c3 == b"strh" and bytes(data[p3:p3 + 4]) == b"vids":
                            handler = bytes(data[p3 + 4:p3 + 8])
                            sc, rt = struct.unpack_from("<II", data, p3 + 20)
                            if sc and rt:
                                fps = rt / sc
                        elif c3 == b"strf" and s3 >= 20 and handler is not None:
                            handler = bytes(data[p3 + 16:p3 + 20]) or handler
        elif kind == b"movi":
            for c2, p2, s2 in _walk(data, ps + 4, ps + sz):
                if c2[2:4] in (b"dc", b"db") and s2 > 0:
                    spans.append((p2, s2))
    if handler is None or handler.upper() not in (b"MJPG", b"JPEG"):
        raise ValueError("%s: the video stream is %r, only Motion-JPEG (MJPG) AVI can be decoded without OpenCV" % (path, handler))

    return float(fps or 25.0), MjpegFrames(data, spans), len(spans)


class MjpegFrames:
    """Random-access sequence of the decoded frames (every MJPG frame is a key frame): a rank of a multi-GPU run
    decodes only the frames of its own batches (video.FrameSource)."""

    def __init__(self, data, spans):
        self._data, self._spans = data, spans

    def __len__(self):
        return len(self._spans)

    def __getitem__(self, i):
        from PIL import Image
        ps, sz = self._spans[i]
        return np.asarray(Image.open(io.BytesIO(bytes(self._data[ps:ps + sz]))).convert("RGB"))

    def __iter__(self):
        return (self[i] for i in range(len(self)))
