"""The AVI writers of the frame tail on the host (no GPU): the reference's MJPG container
(NeRFs/HeadNeRF/test/eval_aud_exp_nerf.py:482-483,495) and the lossless one."""
import io

import numpy as np
import pytest


def _frames(n, h, w):
    yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    out = []
    for i in range(n):
        f = np.stack([0.5 + 0.5 * np.sin(6 * xx + 0.3 * i), yy, 0.5 + 0.5 * np.cos(5 * yy * xx + 0.2 * i)], -1)
        out.append((255 * np.clip(f, 0, 1)).astype(np.uint8))
    return out


@pytest.mark.parametrize("w", [80, 81])     # 81: odd JPEG sizes and padded DIB rows
def test_mjpg_avi_round_trip(tmp_path, w):
    from PIL import Image
    from idealnerf_amd.frame_io import MjpgAviWriter, read_avi_chunks
    h, n = 64, 5
    frames = _frames(n, h, w)
    wr = MjpgAviWriter(str(tmp_path / "c.avi"), w, h, fps=25.0, quality=95)
    for f in frames:
        wr.write(f)
    wr.release()
    info, chunks = read_avi_chunks(str(tmp_path / "c.avi"))
    assert info["handler"] == b"MJPG" and info["compression"] == b"MJPG" and info["chunk_id"] == b"00dc"
    assert (info["width"], info["height"], info["frames"]) == (w, h, n) and abs(info["fps"] - 25.0) < 1e-6
    assert len(chunks) == n
    for f, c in zip(frames, chunks):
        dec = np.asarray(Image.open(io.BytesIO(c)).convert("RGB"))[..., ::-1]
        assert dec.shape == f.shape
        assert np.abs(dec.astype(np.int32) - f.astype(np.int32)).mean() < 2.0
    with pytest.raises(ValueError):
        MjpgAviWriter(str(tmp_path / "d.avi"), w, h).write(np.zeros((h, w + 1, 3), np.uint8))


def test_raw_avi_round_trip_is_lossless(tmp_path):
    from idealnerf_amd.frame_io import RawAviWriter, read_avi_chunks
    h, w, n = 10, 7, 3     # 21-byte rows: padded to 24
    frames = _frames(n, h, w)
    wr = RawAviWriter(str(tmp_path / "r.avi"), w, h, fps=30.0)
    for f in frames:
        wr.write(f)
    wr.release()
    info, chunks = read_avi_chunks(str(tmp_path / "r.avi"))
    assert info["handler"] == b"DIB " and info["compression"] == b"\0\0\0\0" and info["frames"] == n
    for f, c in zip(frames, chunks):
        rows = np.frombuffer(c, np.uint8).reshape(h, 24)[:, :21].reshape(h, w, 3)[::-1]
        np.testing.assert_array_equal(rows, f)
