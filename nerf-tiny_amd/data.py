"""Data side of the hot path.

* ``create_npy`` / ``convert_npy`` / ``NeRFDataset``: the reference's pre-processors and per-pixel dataset
  (``loader.py:12-133``) restated so that a checkout that has the Blender / LLFF files behaves the same
  (same ``.npy`` side files, same ``__getitem__`` tuple).  CPU code, I/O only.
* ``DeviceRays``: the GPU-resident ray sampler that replaces ``NeRFDataset.__getitem__`` + ``DataLoader`` in the
  training loop: all pixels ``[N*H*W, 3]`` and pose rows ``[N, 17]`` live in HBM, a batch is a device-side random
  permutation slice gathered by one kernel (``nerf_hip_gather_rays``) -- no per-pixel Python, no collate, no H2D.
* ``synthetic_scene``: procedural images + poses for demos/tests (the datasets are not available offline).
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from . import _abi

NEAR_FACTOR = 2.0  # loader.py:9
FAR_FACTOR = 6.0  # loader.py:10


def create_npy(root_dir: str, mode: str) -> None:
    """Blender ``transforms_<mode>.json`` -> ``<mode>.npy`` of [N,17] rows (loader.py:12-36): the 3x4 camera-to-world
    matrix with a fifth column (height, width, focal), flattened, then near, far."""
    from PIL import Image

    with open(root_dir + "transforms_" + mode + ".json") as f:
        meta = json.load(f)
    frames = meta["frames"]
    with Image.open(root_dir + frames[0]["file_path"][2:] + ".png") as im:
        width, height = im.size
    focal = 0.5 * width / np.tan(0.5 * meta["camera_angle_x"])
    out = np.zeros((len(frames), 17))
    hwf = np.array([[height], [width], [focal]], dtype=np.float64)
    for i, fr in enumerate(frames):
        m = np.array(fr["transform_matrix"])[:3, :4]
        out[i, :15] = np.concatenate((m, hwf), axis=1).reshape(-1)
        out[i, 15:] = (NEAR_FACTOR, FAR_FACTOR)
    np.save(root_dir + mode + ".npy", out)


def convert_npy(root_dir: str) -> None:
    """LLFF ``poses_bounds.npy`` -> ``new.npy`` (loader.py:38-53): columns reordered (y, -x, z) and the 3x3 block
    transposed, translation and hwf kept."""
    src = np.load(root_dir + "poses_bounds.npy")
    dst = np.zeros_like(src)
    for i, row in enumerate(src):
        pose = row[:-2].reshape(3, 5)
        c2w, hwf = pose[:, :4], pose[:, 4]
        rot = np.stack((c2w[:, 1], -c2w[:, 0], c2w[:, 2]), axis=0).reshape(3, 3).transpose()
        dst[i, :15] = np.concatenate((rot, c2w[:, 3:4], hwf.reshape(3, 1)), axis=1).reshape(-1)
        dst[i, 15:] = row[-2:]
    np.save(root_dir + "new.npy", dst)


def data_preprocess(root_dir: str, type: str, mode: str) -> None:
    """loader.py:55-59."""
    if type == "llff":
        convert_npy(root_dir)
    else:
        create_npy(root_dir, mode)


class NeRFDataset(torch.utils.data.Dataset):
    """Per-pixel dataset with the reference's constructor, attributes and ``__getitem__`` tuple (loader.py:61-133):
    ``(row, column, pix_val[3], poses_bound[17], pic)``.  ``low_res`` is stored and not applied, like the reference."""

    def __init__(self, root_dir, low_res=8, transform=None, type="sync", mode="train"):
        from PIL import Image

        self.root_dir, self.low_res, self.transform, self.type = root_dir, low_res, transform, type
        trans_path = root_dir + ("new.npy" if type == "llff" else mode + ".npy")
        if not os.path.isfile(trans_path):
            data_preprocess(root_dir, type, mode)
        self.poses_bounds = np.load(trans_path)
        img_dir = root_dir + ("images/" if type == "llff" else mode + "/")
        # every file of the directory, ordered by the integer after the last '_' (loader.py:107-112)
        self.file_list = sorted((os.path.join(img_dir, f) for f in os.listdir(img_dir)), key=lambda n: int(n.split("_")[-1][:-4]))
        self.pic_num = len(self.file_list)
        self.height = int(self.poses_bounds[0][4])
        self.width = int(self.poses_bounds[0][9])
        self.focal = self.poses_bounds[0][14]
        self.pic_size = self.height * self.width
        self.num_pix = self.pic_size * self.pic_num
        imgs = torch.zeros(self.pic_num, self.height, self.width, 3)
        for i, path in enumerate(self.file_list):
            with Image.open(path) as im:
                im.load()
                if type == "sync":  # RGBA on white (loader.py:67-71)
                    bg = Image.new("RGB", im.size, (255, 255, 255))
                    bg.paste(im, mask=im.split()[3])
                    im = bg
                imgs[i] = torch.tensor(np.array(im) / 255.0)
        self.all_pix = imgs.flatten(0, 2)

    def __len__(self):
        return self.num_pix

    def __getitem__(self, idx):
        pic, rem = divmod(idx, self.pic_size)
        row, column = divmod(rem, self.width)
        return row, column, self.all_pix[idx][0:3], self.poses_bounds[pic], pic


class ArrayDataset(torch.utils.data.Dataset):
    """Same tuple as NeRFDataset from in-memory arrays (synthetic scenes, tests)."""

    def __init__(self, images: torch.Tensor, poses_bounds: np.ndarray):
        self.pic_num, self.height, self.width, _ = images.shape
        self.poses_bounds = np.asarray(poses_bounds, dtype=np.float64)
        self.focal = self.poses_bounds[0][14]
        self.pic_size = self.height * self.width
        self.num_pix = self.pic_size * self.pic_num
        self.all_pix = images.to(torch.float32).flatten(0, 2)

    def __len__(self):
        return self.num_pix

    __getitem__ = NeRFDataset.__getitem__


def synthetic_scene(n_pic=8, H=64, W=64, seed=0):
    """Procedural stand-in for the lego set: cameras on a ring looking at the origin, smooth colour fields as
    'photographs'.  Returns an ArrayDataset."""
    rng = np.random.default_rng(seed)
    focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
    rows = []
    imgs = torch.zeros(n_pic, H, W, 3)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    for i in range(n_pic):
        th = 2 * np.pi * i / n_pic
        pos = np.array([4.0 * np.cos(th), 4.0 * np.sin(th), 1.0 + 0.3 * rng.standard_normal()])
        back = pos / np.linalg.norm(pos)  # camera looks along -z (local z = back)
        right = np.cross([0.0, 0.0, 1.0], back)
        right /= np.linalg.norm(right)
        up = np.cross(back, right)
        c2w = np.stack((right, up, back, pos), axis=1)
        rows.append(np.concatenate((np.concatenate((c2w, [[H], [W], [focal]]), axis=1).reshape(-1), [NEAR_FACTOR, FAR_FACTOR])))
        r2 = (xx - 0.3 * np.cos(th)) ** 2 + (yy - 0.2 * np.sin(th)) ** 2
        imgs[i, ..., 0] = torch.exp(-3.0 * r2)
        imgs[i, ..., 1] = 0.5 + 0.5 * torch.sin(3.0 * xx + th)
        imgs[i, ..., 2] = 0.5 + 0.5 * torch.cos(2.0 * yy - th)
    return ArrayDataset(imgs, np.stack(rows))


def analytic_sphere_scene(n_pic=24, H=64, W=64, seed=5, device="cuda:0"):
    """A multi-view-CONSISTENT synthetic scene: a unit sphere at the origin (position-coloured, Lambert-shaded) in front of
    a white background, seen by the ring of cameras of ``synthetic_scene``.  The rays come from ``nerf_hip_rays``, i.e. the
    reference's camera convention incl. quirk Q2.  Returns an ArrayDataset."""
    from . import ops

    base = synthetic_scene(n_pic=n_pic, H=H, W=W, seed=seed)
    poses = base.poses_bounds
    K_inv = torch.tensor([[1.0, 0.0, -0.5 * W], [0.0, -1.0, 0.5 * H], [0.0, 0.0, -base.focal]]).float().t()
    imgs = torch.ones(n_pic, H, W, 3)
    rr, cc = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    row, col = rr.reshape(-1).to(device), cc.reshape(-1).to(device)
    light = torch.tensor([0.5, 0.3, 0.8]).double()
    light = light / light.norm()
    for i in range(n_pic):
        pb = torch.from_numpy(np.tile(poses[i], (H * W, 1))).float().to(device)
        _, d_wrd, _ = ops.rays(row, col, pb, K_inv, 2)
        d = d_wrd.cpu().double()
        o = torch.from_numpy(poses[i, :15].reshape(3, 5)[:, 3].copy()).double()
        b = (d * o).sum(1)
        disc = b * b - ((o * o).sum() - 1.0)
        t = -b - torch.sqrt(disc.clamp_min(0))
        p = o + t[:, None] * d
        colour = (0.5 + 0.5 * p) * (p * light).sum(1).clamp_min(0.15)[:, None]
        imgs[i] = torch.where((disc > 0)[:, None], colour, torch.ones_like(colour)).float().reshape(H, W, 3)
    return ArrayDataset(imgs, poses)


class DeviceRays:
    """GPU-resident ray sampler (row f1).  ``for row, col, pix_val, poses_bound, pic in rays.epoch(B)`` yields what the
    reference's ``DataLoader(shuffle=True, drop_last=True)`` yields (nerf.py:424, 458) -- but as device tensors produced
    by one gather kernel, with the shuffle done by ``torch.randperm`` on the device."""

    def __init__(self, dataset, device, seed: int | None = None):
        self.device = torch.device(device)
        self.height, self.width, self.focal = dataset.height, dataset.width, dataset.focal
        self.pic_num, self.num_pix = dataset.pic_num, dataset.num_pix
        self.pixels = dataset.all_pix.to(self.device, torch.float32).contiguous()
        self.poses = torch.as_tensor(np.asarray(dataset.poses_bounds)).to(torch.float32).to(self.device).contiguous()  # cast like nerf.py:338
        self.gen = torch.Generator(device=self.device)
        if seed is not None:
            self.gen.manual_seed(seed)

    def gather(self, index: torch.Tensor):
        """index [B] i64 (device) -> (row, col, pix_val, poses_bound, pic)"""
        B = index.shape[0]
        dev = self.device
        row = torch.empty(B, dtype=torch.int64, device=dev)
        col = torch.empty(B, dtype=torch.int64, device=dev)
        pic = torch.empty(B, dtype=torch.int64, device=dev)
        pix = torch.empty(B, 3, dtype=torch.float32, device=dev)
        pb = torch.empty(B, 17, dtype=torch.float32, device=dev)
        index = index.to(dev, torch.int64).contiguous()
        _abi.check(_abi.lib().nerf_hip_gather_rays(index.data_ptr(), self.pixels.data_ptr(), self.poses.data_ptr(), B, self.height,
                                                   self.width, row.data_ptr(), col.data_ptr(), pic.data_ptr(), pix.data_ptr(),
                                                   pb.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        return row, col, pix, pb, pic

    def epoch_order(self, shuffle: bool = True) -> torch.Tensor:
        """The pixel order of one epoch (device tensor): ``DataLoader(shuffle=True)``'s permutation, drawn on the device from this
        sampler's generator -- ranks that were built with the same seed draw the same order."""
        if shuffle:
            return torch.randperm(self.num_pix, device=self.device, generator=self.gen)
        return torch.arange(self.num_pix, device=self.device)

    def epoch(self, batch_ray: int, shuffle: bool = True, first_batch: int = 0):
        """`first_batch`: skip that many batches of the epoch's order (a resumed run continues the interrupted epoch)."""
        order = self.epoch_order(shuffle)
        for s in range(first_batch * batch_ray, self.num_pix - batch_ray + 1, batch_ray):  # drop_last=True
            yield self.gather(order[s:s + batch_ray])

    def epoch_sharded(self, batch_ray: int, rank: int, world: int, shuffle: bool = True, first_batch: int = 0):
        """One epoch of a data-parallel trainer: every rank draws the SAME order (same seed) and gathers only its contiguous slice
        [lo, hi) of every global `batch_ray` batch (parallel.shard_bounds).  Yields (row, col, pix_val, poses_bound, pic, ray0) where
        ray0 = (near, far) of the GLOBAL batch's ray 0 as host floats -- the one cross-ray term of the path (quirk Q6, nerf.py:233); the
        pictures of all first rays come to the host in ONE copy per epoch, so the loop itself has no host sync."""
        from .parallel import shard_bounds

        order = self.epoch_order(shuffle)
        lo, hi = shard_bounds(batch_ray, rank, world)
        starts = range(0, self.num_pix - batch_ray + 1, batch_ray)  # drop_last=True
        if len(starts) == 0:
            return
        first_pic = (order[0:starts[-1] + 1:batch_ray] // (self.height * self.width)).cpu()
        nf = self.poses[:, 15:17].cpu()[first_pic]  # [n_batches, 2] fp32, exactly the values the kernels see (nerf.py:338 cast)
        for b, s in enumerate(starts):
            if b < first_batch:  # (a resumed run continues the interrupted epoch)
                continue
            yield (*self.gather(order[s + lo:s + hi]), (float(nf[b, 0]), float(nf[b, 1])))

    def __len__(self):
        return self.num_pix
