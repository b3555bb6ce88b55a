"""Frame tail of the eval loop (NeRFs/HeadNeRF/test/eval_aud_exp_nerf.py:480-496).

The reference does, per frame and on the host:  ``rgb = to8b(rgb.cpu().numpy())`` then
``vid_out.write(rgb)`` into an MJPG ``cv2.VideoWriter`` at 25 fps (plus a JPEG every 10th
frame), after eight ``isnan/isinf().any()`` host syncs per chunk inside the render
(train/audio_exp_nerf.py:367-369).

Here the conversion runs on the device (``ops.to8b``, bit-identical to the numpy formula), the
NaN/Inf scan is one device flag per frame read together with the pixels, and the device-to-host
copy of frame i runs on a side stream into one of two pinned buffers while frame i+1 renders.
The container is the reference's: an MJPG AVI (one JPEG per ``00dc`` chunk, ``fccHandler`` 'MJPG',
25 fps) plus the every-10th-frame ``.jpg`` (:492-493).  cv2 is not part of this image, so the JPEGs
are encoded with PIL on a writer thread; an uncompressed AVI (``DIB `` / BI_RGB) stays available as
``codec="raw"``.  ``swap_rb`` keeps the reference's channel handling selectable: the
reference hands its RGB frame to cv2 unswapped (``cvtColor`` is commented out, :490), i.e. its
files have red and blue exchanged; ``swap_rb=False`` reproduces those bytes.
"""
import queue
import struct
import threading
from typing import List, Optional

import numpy as np
import torch

from . import ops
from ._lib import IdealNerfError


class _AviWriter:
    """Minimal RIFF/AVI writer: one video stream, one 'movi' list, an idx1 index.  Subclasses say how a
    frame becomes a chunk payload."""
    chunk_id = b"00db"
    handler = b"DIB "
    compression = 0          # BITMAPINFOHEADER.biCompression (BI_RGB) or a fourcc

    def __init__(self, path: str, width: int, height: int, fps: float = 25.0):
        self.path, self.w, self.h, self.fps = path, int(width), int(height), float(fps)
        self.row = (self.w * 3 + 3) & ~3  # DIB rows are padded to 4 bytes
        self.frame_bytes = self.row * self.h
        self.n = 0
        self.index = []           # (offset from the 'movi' fourcc, payload bytes) per frame
        self.f = open(path, "wb")
        self.f.write(b"\0" * self._header_len())  # patched in release()
        self.movi_start = self.f.tell()
        self.f.write(b"LIST" + struct.pack("<I", 0) + b"movi")

    @staticmethod
    def _header_len() -> int:
        return 12 + (8 + 4 + (8 + 56) + (8 + 4 + (8 + 56) + (8 + 40)))

    def _check(self, frame_bgr: np.ndarray) -> None:
        if frame_bgr.shape != (self.h, self.w, 3) or frame_bgr.dtype != np.uint8:
            raise ValueError(f"expected uint8 [{self.h},{self.w},3], got {frame_bgr.dtype} {frame_bgr.shape}")

    def _payload(self, frame_bgr: np.ndarray) -> bytes:
        raise NotImplementedError

    def write(self, frame_bgr: np.ndarray) -> None:
        """frame [H, W, 3] uint8 in the order cv2.VideoWriter.write takes it (BGR)."""
        self._check(frame_bgr)
        data = self._payload(frame_bgr)
        self.index.append((self.f.tell() - self.movi_start - 8, len(data)))
        self.f.write(self.chunk_id + struct.pack("<I", len(data)) + data)
        if len(data) & 1:
            self.f.write(b"\0")   # RIFF chunks are word-aligned
        self.n += 1

    def release(self) -> None:
        if self.f is None:
            return
        movi_end = self.f.tell()
        idx = b"".join(self.chunk_id + struct.pack("<III", 0x10, off, size) for off, size in self.index)
        self.f.write(b"idx1" + struct.pack("<I", len(idx)) + idx)
        end = self.f.tell()
        usec = int(round(1e6 / self.fps))
        biggest = max([size for _, size in self.index], default=0)
        avih = struct.pack("<IIIIIIIIIIIIII", usec, int(biggest * self.fps), 0, 0x10, self.n, 0, 1,
                           biggest, self.w, self.h, 0, 0, 0, 0)
        strh = b"vids" + self.handler + struct.pack("<IHHIIIIIIIIhhhh", 0, 0, 0, 0, 1000, int(round(self.fps * 1000)), 0,
                                                   self.n, biggest, 0xFFFFFFFF, 0, 0, 0, self.w, self.h)
        comp = self.compression if isinstance(self.compression, int) else struct.unpack("<I", self.compression)[0]
        strf = struct.pack("<IiiHHIIiiII", 40, self.w, self.h, 1, 24, comp, self.frame_bytes, 0, 0, 0, 0)
        strl = b"LIST" + struct.pack("<I", 4 + 8 + len(strh) + 8 + len(strf)) + b"strl" + \
            b"strh" + struct.pack("<I", len(strh)) + strh + b"strf" + struct.pack("<I", len(strf)) + strf
        hdrl = b"LIST" + struct.pack("<I", 4 + 8 + len(avih) + len(strl)) + b"hdrl" + \
            b"avih" + struct.pack("<I", len(avih)) + avih + strl
        head = b"RIFF" + struct.pack("<I", end - 8) + b"AVI " + hdrl
        assert len(head) == self._header_len(), (len(head), self._header_len())
        self.f.seek(self.movi_start + 4)
        self.f.write(struct.pack("<I", movi_end - self.movi_start - 8))
        self.f.seek(0)
        self.f.write(head)
        self.f.close()
        self.f = None


class RawAviWriter(_AviWriter):
    """24-bit uncompressed frames ('DIB ' / BI_RGB, bottom-up rows): lossless, for tests and debugging."""

    def _payload(self, frame_bgr: np.ndarray) -> bytes:
        rows = frame_bgr[::-1]  # bottom-up
        if self.row != self.w * 3:
            padded = np.zeros((self.h, self.row), dtype=np.uint8)
            padded[:, : self.w * 3] = rows.reshape(self.h, -1)
            rows = padded
        return np.ascontiguousarray(rows).tobytes()


def encode_jpeg(frame_bgr: np.ndarray, quality: int = 95) -> bytes:
    """One baseline JPEG of a BGR frame (what cv2.imwrite / cv2's MJPG writer produce from the same array)."""
    import io

    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(frame_bgr[..., ::-1])).save(buf, format="JPEG", quality=int(quality))
    return buf.getvalue()


class MjpgAviWriter(_AviWriter):
    """The reference's container: ``cv2.VideoWriter(path, fourcc('M','J','P','G'), 25, (W, H))``
    (test/eval_aud_exp_nerf.py:482-483).  One JPEG per '00dc' chunk, fccHandler and biCompression 'MJPG'.
    cv2 is not in this image; the JPEGs are encoded by PIL (libjpeg)."""
    chunk_id = b"00dc"
    handler = b"MJPG"
    compression = b"MJPG"

    def __init__(self, path: str, width: int, height: int, fps: float = 25.0, quality: int = 95):
        super().__init__(path, width, height, fps)
        self.quality = int(quality)

    def _payload(self, frame_bgr: np.ndarray) -> bytes:
        return encode_jpeg(frame_bgr, self.quality)


def read_avi_chunks(path: str):
    """(header fields, [payload bytes per frame]) of a file written by the writers above, located
    through its idx1 index the way a player seeks (used by the tests)."""
    data = open(path, "rb").read()
    assert data[:4] == b"RIFF" and data[8:12] == b"AVI "
    movi = data.index(b"movi") - 8
    assert data[movi:movi + 4] == b"LIST"
    movi_size = struct.unpack("<I", data[movi + 4:movi + 8])[0]
    idx = movi + 8 + movi_size
    assert data[idx:idx + 4] == b"idx1"
    n = struct.unpack("<I", data[idx + 4:idx + 8])[0] // 16
    frames = []
    for i in range(n):
        cid, _flags, off, size = struct.unpack("<4sIII", data[idx + 8 + 16 * i: idx + 24 + 16 * i])
        at = movi + 8 + off
        assert data[at:at + 4] == cid and struct.unpack("<I", data[at + 4:at + 8])[0] == size
        frames.append(data[at + 8:at + 8 + size])
    strh = data.index(b"strh") + 8
    strf = data.index(b"strf") + 8
    usec, _, _, _, total = struct.unpack("<IIIII", data[32:52])
    info = dict(handler=data[strh + 4:strh + 8], compression=data[strf + 16:strf + 20],
                width=struct.unpack("<i", data[strf + 4:strf + 8])[0], height=struct.unpack("<i", data[strf + 8:strf + 12])[0],
                fps=1e6 / usec, frames=total, chunk_id=cid if n else None)
    return info, frames


class FrameSink:
    """``sink.submit(rgb)`` right after a frame is rendered; ``sink.release()`` at the end.

    submit() enqueues to8b on the render stream, then the D2H copy on a side stream that waits for
    it; the host only blocks on frame i-1's copy (already finished in steady state), writes it, and
    returns -- so the copy and the file write of one frame overlap the render of the next.
    ``nonfinite_frames`` lists the indices of frames that held a NaN/Inf (the reference prints a
    line per offending key; nothing else depends on it).
    """

    def __init__(self, path: Optional[str], width: int, height: int, fps: float = 25.0, swap_rb: bool = False,
                 device="cuda", keep_frames: bool = False, codec: str = "MJPG", jpeg_quality: int = 95,
                 still_every: int = 0, still_path: Optional[str] = None):
        """codec "MJPG" (the reference's fourcc) or "raw" (lossless DIB).  ``still_every=10`` with
        ``still_path="dir/name_{i}.jpg"`` also writes the every-10th-frame JPEG of
        eval_aud_exp_nerf.py:492-493.  Encoding and file writes run on a worker thread, so submit() only
        waits for a frame's copy, never for its JPEG."""
        if torch.device(device).type != "cuda":
            raise IdealNerfError("FrameSink copies from the GPU; the HIP path has no CPU fallback")
        if codec not in ("MJPG", "raw"):
            raise ValueError("codec must be 'MJPG' or 'raw'")
        if still_every and not still_path:
            raise ValueError("still_every needs still_path, e.g. 'out/frame_{i}.jpg'")
        self.h, self.w, self.swap_rb = int(height), int(width), bool(swap_rb)
        self.writer = None
        if path:
            self.writer = (MjpgAviWriter(path, width, height, fps, jpeg_quality) if codec == "MJPG"
                           else RawAviWriter(path, width, height, fps))
        self.jpeg_quality, self.still_every, self.still_path = int(jpeg_quality), int(still_every), still_path
        self.stills: List[str] = []
        self._queue: "queue.Queue" = queue.Queue(maxsize=4)
        self._error: Optional[BaseException] = None
        self._worker = None
        if self.writer is not None or self.still_every:
            self._worker = threading.Thread(target=self._drain, name="idn-frame-writer", daemon=True)
            self._worker.start()
        self.copy_stream = torch.cuda.Stream(device=device)
        self.host = [torch.empty((self.h, self.w, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.host_flag = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.dev_flag = [torch.zeros(1, dtype=torch.int32, device=device) for _ in range(2)]
        self.done = [None, None]
        self.pending: List[int] = []
        self.count = 0
        self.nonfinite_frames: List[int] = []
        self.frames: Optional[List[np.ndarray]] = [] if keep_frames else None

    def _retire(self, slot: int, index: int) -> None:
        self.done[slot].synchronize()
        if int(self.host_flag[slot][0]) != 0:
            self.nonfinite_frames.append(index)
        frame = self.host[slot].numpy().copy()   # the pinned buffer is reused two submits later
        if self.frames is not None:
            self.frames.append(frame)
        if self._worker is not None:
            if self._error is not None:
                raise self._error
            self._queue.put((index, frame))

    def _drain(self) -> None:
        while True:
            item = self._queue.get()
            if item is None:
                return
            index, frame = item
            try:
                if self._error is None:
                    if self.still_every and index % self.still_every == 0:
                        name = self.still_path.format(i=index)
                        with open(name, "wb") as f:
                            f.write(encode_jpeg(frame, self.jpeg_quality))
                        self.stills.append(name)
                    if self.writer is not None:
                        self.writer.write(frame)
            except BaseException as e:  # surfaced by the next submit() / release()
                self._error = e

    def submit(self, rgb: torch.Tensor) -> None:
        slot = self.count & 1
        if self.count >= 2:
            self._retire(slot, self.pending.pop(0))  # the frame that used this slot two submits ago
        self.dev_flag[slot].zero_()
        u8 = ops.to8b(rgb.reshape(self.h, self.w, 3), self.swap_rb, self.dev_flag[slot])
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(ready)
            self.host[slot].copy_(u8, non_blocking=True)
            self.host_flag[slot].copy_(self.dev_flag[slot], non_blocking=True)
            u8.record_stream(self.copy_stream)
            self.done[slot] = torch.cuda.Event()
            self.done[slot].record()
        self.pending.append(self.count)
        self.count += 1

    def release(self) -> None:
        for index in list(self.pending):
            self._retire(index & 1, index)
        self.pending.clear()
        if self._worker is not None:
            self._queue.put(None)
            self._worker.join()
            self._worker = None
        if self.writer is not None:
            self.writer.release()
        if self._error is not None:
            raise self._error
