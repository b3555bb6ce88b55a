"""Frame tail of the eval loop (NeRFs/HeadNeRF/test/eval_aud_exp_nerf.py:480-496).

The reference does, per frame and on the host:  ``rgb = to8b(rgb.cpu().numpy())`` then
``vid_out.write(rgb)`` into an MJPG ``cv2.VideoWriter`` at 25 fps (plus a JPEG every 10th
frame), after eight ``isnan/isinf().any()`` host syncs per chunk inside the render
(train/audio_exp_nerf.py:367-369).

Here the conversion runs on the device (``ops.to8b``, bit-identical to the numpy formula), the
NaN/Inf scan is one device flag per frame read together with the pixels, and the device-to-host
copy of frame i runs on a side stream into one of two pinned buffers while frame i+1 renders.
cv2 is not part of this image, so the container written is an uncompressed AVI (``DIB `` /
BI_RGB, bottom-up BGR rows as the format requires) instead of MJPG: same frames, same fps,
readable by ffmpeg/VLC/cv2.  ``swap_rb`` keeps the reference's channel handling selectable: the
reference hands its RGB frame to cv2 unswapped (``cvtColor`` is commented out, :490), i.e. its
files have red and blue exchanged; ``swap_rb=False`` reproduces those bytes.
"""
import struct
from typing import List, Optional

import numpy as np
import torch

from . import ops
from ._lib import IdealNerfError


class RawAviWriter:
    """Minimal RIFF/AVI writer for 24-bit uncompressed frames (one 'movi' list + idx1)."""

    def __init__(self, path: str, width: int, height: int, fps: float = 25.0):
        self.path, self.w, self.h, self.fps = path, int(width), int(height), float(fps)
        self.row = (self.w * 3 + 3) & ~3  # DIB rows are padded to 4 bytes
        self.frame_bytes = self.row * self.h
        self.n = 0
        self.f = open(path, "wb")
        self.f.write(b"\0" * self._header_len())  # patched in close()
        self.movi_start = self.f.tell()
        self.f.write(b"LIST" + struct.pack("<I", 0) + b"movi")

    @staticmethod
    def _header_len() -> int:
        return 12 + (8 + 4 + (8 + 56) + (8 + 4 + (8 + 56) + (8 + 40)))

    def write(self, frame_bgr: np.ndarray) -> None:
        """frame [H, W, 3] uint8 in the order it should sit in the file (cv2 convention: BGR)."""
        if frame_bgr.shape != (self.h, self.w, 3) or frame_bgr.dtype != np.uint8:
            raise ValueError(f"expected uint8 [{self.h},{self.w},3], got {frame_bgr.dtype} {frame_bgr.shape}")
        rows = frame_bgr[::-1]  # bottom-up
        if self.row != self.w * 3:
            padded = np.zeros((self.h, self.row), dtype=np.uint8)
            padded[:, : self.w * 3] = rows.reshape(self.h, -1)
            rows = padded
        self.f.write(b"00db" + struct.pack("<I", self.frame_bytes))
        self.f.write(np.ascontiguousarray(rows).tobytes())
        self.n += 1

    def release(self) -> None:
        if self.f is None:
            return
        movi_end = self.f.tell()
        idx = b"".join(b"00db" + struct.pack("<III", 0x10, 4 + i * (8 + self.frame_bytes), self.frame_bytes)
                       for i in range(self.n))
        self.f.write(b"idx1" + struct.pack("<I", len(idx)) + idx)
        end = self.f.tell()
        usec = int(round(1e6 / self.fps))
        avih = struct.pack("<IIIIIIIIIIIIII", usec, int(self.frame_bytes * self.fps), 0, 0x10, self.n, 0, 1,
                           self.frame_bytes, self.w, self.h, 0, 0, 0, 0)
        strh = b"vids" + b"DIB " + struct.pack("<IHHIIIIIIIIhhhh", 0, 0, 0, 0, 1000, int(round(self.fps * 1000)), 0,
                                                  self.n, self.frame_bytes, 0xFFFFFFFF, 0, 0, 0, self.w, self.h)
        strf = struct.pack("<IiiHHIIiiII", 40, self.w, self.h, 1, 24, 0, self.frame_bytes, 0, 0, 0, 0)
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


class FrameSink:
    """``sink.submit(rgb)`` right after a frame is rendered; ``sink.release()`` at the end.

    submit() enqueues to8b on the render stream, then the D2H copy on a side stream that waits for
    it; the host only blocks on frame i-1's copy (already finished in steady state), writes it, and
    returns -- so the copy and the file write of one frame overlap the render of the next.
    ``nonfinite_frames`` lists the indices of frames that held a NaN/Inf (the reference prints a
    line per offending key; nothing else depends on it).
    """

    def __init__(self, path: Optional[str], width: int, height: int, fps: float = 25.0, swap_rb: bool = False,
                 device="cuda", keep_frames: bool = False):
        if torch.device(device).type != "cuda":
            raise IdealNerfError("FrameSink copies from the GPU; the HIP path has no CPU fallback")
        self.h, self.w, self.swap_rb = int(height), int(width), bool(swap_rb)
        self.writer = RawAviWriter(path, width, height, fps) if path else None
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
        frame = self.host[slot].numpy()
        if self.writer is not None:
            self.writer.write(frame)
        if self.frames is not None:
            self.frames.append(frame.copy())

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
        if self.writer is not None:
            self.writer.release()
