"""Reader for the reference's on-disk dataset and its region-weighted ray sampler
(``GetData``, NeRFs/HeadNeRF/train/audio_exp_nerf.py:45-195; format written by
data_util/process_data.py:250-288):

    <dir>/transforms_exp_{train,val}.json   focal_len, cx, cy, frames[{img_id, aud_id,
                                            transform_matrix[4][4], face_rect[4], exp[...]}]
    <dir>/<aud_file>.npy  [F, 16, 29]       DeepSpeech windows
    <dir>/bc.jpg                            background
    <dir>/<gt_dirs>/<id>.jpg, ori_imgs/<id>.lms [68,2], parsing/<id>.png

``__getitem__`` returns the reference's 8-tuple.  Pixel selection reproduces upstream's numpy
RNG call sequence (mouth, torso, face rect, outside rect -- each ``np.random.choice`` without
replacement), including its row/column convention (pixel rows are compared against the
landmark / rect *x* bounds, :150-155); rays come from the device-side pinhole kernel with the
dataset's principal point and are gathered on the device.
"""
import json
import os

import numpy as np
import torch

from . import ops


def _imread(path):
    try:
        from PIL import Image
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("reading dataset images needs Pillow") from e
    return np.asarray(Image.open(path))


def select_pixels(H, W, face_rect, landmark, parse_img, n_rand, mouth_rays, torso_rays, sample_rate):
    """-> int64 [n_rand, 2] (row, col) in the reference's order: face rect, outside rect, mouth,
    torso (audio_exp_nerf.py:143-187).  Draws from the global numpy RNG exactly like upstream."""
    mouth = landmark[48:]
    max_x, min_x = np.max(mouth[:, 0]) + 20, np.min(mouth[:, 0]) - 20
    max_y, min_y = np.max(mouth[:, 1]) + 20, np.min(mouth[:, 1]) - 20
    rows, cols = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    coords = np.stack([rows, cols], -1).reshape(-1, 2)
    mouth_w = (coords[:, 0] >= min_x) & (coords[:, 0] <= max_x) & (coords[:, 1] >= min_y) & (coords[:, 1] <= max_y)
    rect_w = ((coords[:, 0] >= face_rect[0]) & (coords[:, 0] <= face_rect[0] + face_rect[2]) &
              (coords[:, 1] >= face_rect[1]) & (coords[:, 1] <= face_rect[1] + face_rect[3]))
    torso = (parse_img[:, :, 0] == 255) & (parse_img[:, :, 1] == 0) & (parse_img[:, :, 2] == 0)
    c_mouth, c_rect, c_norect = coords[mouth_w], coords[rect_w & ~mouth_w], coords[~rect_w]
    c_torso = np.stack([rows, cols], -1)[torso].reshape(-1, 2)
    sample_num = n_rand - mouth_rays - torso_rays
    rect_num = int(sample_num * sample_rate)
    norect_num = sample_num - rect_num
    pick = lambda c, k: c[np.random.choice(c.shape[0], size=[k], replace=False)].astype(np.int64)
    s_mouth = pick(c_mouth, mouth_rays)
    s_torso = pick(c_torso, torso_rays)
    s_rect = pick(c_rect, rect_num)
    s_norect = pick(c_norect, norect_num)
    return np.concatenate([s_rect, s_norect, s_mouth, s_torso], 0)


def sample_rays(pose, face_rect, target, bc_img, landmark, parse_img, H, W, focal, cx, cy, n_rand, mouth_rays,
                torso_rays, sample_rate, device):
    """audio_exp_nerf.py:134-195 -> (batch_rays [2, n, 3], target_s [n, 3], bc_rgb [n, 3]) on `device`."""
    sel = torch.from_numpy(select_pixels(H, W, face_rect, landmark, parse_img, n_rand, mouth_rays, torso_rays, sample_rate))
    flat = (sel[:, 0] * W + sel[:, 1]).to(device)
    rec = ops.frame_rays(torch.as_tensor(pose, dtype=torch.float32), H, W, focal, 0.0, 1.0, cx=cx, cy=cy, device=device)
    batch_rays = torch.stack([rec[flat, 0:3], rec[flat, 3:6]], 0)
    target_s = target.reshape(-1, 3)[flat]
    bc_rgb = bc_img.reshape(-1, 3)[flat]
    return batch_rays, target_s, bc_rgb


class GetData(torch.utils.data.Dataset):
    """mode in {train, val, test}; ``args`` needs gt_dirs, testskip, N_rand, sample_rate, mouth_rays, torso_rays."""

    def __init__(self, data_dir, aud_file, mode, args, skip=1, device="cuda"):
        self.data_dir, self.aud_file, self.mode, self.args, self.device = data_dir, aud_file, mode, args, device
        with open(os.path.join(data_dir, f"transforms_exp_{mode}.json")) as fp:
            self.meta = json.load(fp)
        self.aud_features = np.load(os.path.join(data_dir, aud_file))
        self.background_img = torch.tensor(_imread(os.path.join(data_dir, "bc.jpg")) / 255.0).to(device)
        self.focal, self.cx, self.cy = float(self.meta["focal_len"]), float(self.meta["cx"]), float(self.meta["cy"])
        self.H, self.W = int(self.cy * 2), int(self.cx * 2)
        self.skip = 1 if mode == "train" else args.testskip
        self.all_imgs, self.all_parse_imgs, self.all_landmarks = [], [], []
        self.all_poses, self.all_face_rects, self.all_exprs, auds = [], [], [], []
        for frame in self.meta["frames"][::skip]:
            fid = str(frame["img_id"])
            self.all_imgs.append(os.path.join(data_dir, args.gt_dirs, fid + ".jpg"))
            self.all_landmarks.append(os.path.join(data_dir, "ori_imgs", fid + ".lms"))
            self.all_parse_imgs.append(os.path.join(data_dir, "parsing", fid + ".png"))
            self.all_poses.append(np.array(frame["transform_matrix"]))
            auds.append(self.aud_features[min(frame["aud_id"], self.aud_features.shape[0] - 1)])
            self.all_face_rects.append(np.array(frame["face_rect"], dtype=np.int32))
            self.all_exprs.append(frame["exp"])
        self.data_size = len(self.all_imgs)
        self.auds = torch.tensor(np.asarray(auds), dtype=torch.float)

    def __len__(self):
        return self.data_size

    def __getitem__(self, index):
        if index is None:
            index = np.random.choice(self.data_size)
        raw = _imread(self.all_imgs[index])[..., ::-1].copy()  # upstream reads with cv2: BGR
        raw_img = torch.tensor(raw)
        self.H, self.W = raw_img.shape[0], raw_img.shape[1]
        target = raw_img.to(self.device).float() / 255.0
        parse = _imread(self.all_parse_imgs[index])
        pose = self.all_poses[index][:3, :4]
        landmark = np.loadtxt(self.all_landmarks[index])
        a = self.args
        batch_rays, target_s, bc_rgb = sample_rays(pose, self.all_face_rects[index], target, self.background_img,
                                                   landmark, parse, self.H, self.W, self.focal, self.cx, self.cy,
                                                   a.N_rand, a.mouth_rays, a.torso_rays, a.sample_rate, self.device)
        bc_rgb = bc_rgb if self.mode == "train" else self.background_img
        exp = torch.tensor(self.all_exprs[index], dtype=torch.float32)
        return batch_rays, target_s, bc_rgb, self.auds, raw_img, pose, exp, index
