"""Pins the data side to the reference's own loader and writes tests/golden/data_golden.npz.

Run ONLY in the build container (needs /root/reference):

    cd /tmp && python /root/repo/tests/golden/make_data_golden.py

A tiny Blender tree (RGBA PNGs + transforms_<mode>.json) and a tiny LLFF tree (poses_bounds.npy + RGB PNGs) are
written to a scratch directory from seeded arrays, the REFERENCE's loader.py (imported unmodified) is run over them, and
the fixture stores the input arrays plus what the reference produced: the `.npy` side files (`train.npy`, `new.npy`),
dataset attributes, the flattened pixel table and a handful of `__getitem__` tuples.  Data only, no reference code.
tests/test_data_cpu.py rebuilds the trees from the stored arrays and compares nerf-tiny_amd/data.py byte for byte;
tests/test_gpu_train.py compares the gather kernel (DeviceRays) with the stored tuples.
"""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import loader  # the reference (loader.py)  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from data_trees import write_blender_tree, write_llff_tree  # noqa: E402  (shared with the tests: builds the trees from arrays)


def main():
    rng = np.random.default_rng(20240)
    out = {}
    tmp = tempfile.mkdtemp(prefix="nerf_data_golden_")
    try:
        # ---- Blender ("sync"): 3 pictures of 6 x 8, RGBA, file names r_0..r_2, plus the depth/normal style extras the real
        # set does NOT have in train/ (so every file of the directory is a picture, loader.py:107-109)
        H, W, n = 6, 8, 3
        rgba = rng.integers(0, 256, size=(n, H, W, 4), dtype=np.uint8)
        rgba[0, :2, :, 3] = 0      # fully transparent rows -> white
        rgba[1, :, :3, 3] = 255    # opaque columns
        mats = np.zeros((n, 4, 4))
        mats[:, 3, 3] = 1.0
        mats[:, :3, :4] = rng.standard_normal((n, 3, 4))
        angle = 0.6911112070083618
        broot = os.path.join(tmp, "blender") + "/"
        write_blender_tree(broot, "train", rgba, mats, angle)
        ds = loader.NeRFDataset(root_dir=broot, low_res=1, transform=None, type="sync", mode="train")  # runs create_npy
        out.update(b_rgba=rgba, b_mats=mats, b_angle=np.float64(angle), b_train_npy=np.load(broot + "train.npy"),
                   b_all_pix=ds.all_pix.numpy(), b_attrs=np.array([ds.pic_num, ds.height, ds.width, ds.pic_size, ds.num_pix], dtype=np.int64),
                   b_focal=np.float64(ds.focal))
        idx = np.array([0, 1, W - 1, W, H * W - 1, H * W, H * W + 4 * W + 5, n * H * W - 1, 77, 100], dtype=np.int64)
        items = [ds[int(i)] for i in idx]
        out.update(b_idx=idx, b_item_row=np.array([it[0] for it in items], dtype=np.int64), b_item_col=np.array([it[1] for it in items], dtype=np.int64),
                   b_item_pix=np.stack([it[2].numpy() for it in items]), b_item_pose=np.stack([it[3] for it in items]),
                   b_item_pic=np.array([it[4] for it in items], dtype=np.int64))

        # ---- LLFF: 4 pictures of 5 x 7, RGB, poses_bounds.npy rows = 3x5 [R|t|hwf] + near, far
        Hl, Wl, nl = 5, 7, 4
        rgb = rng.integers(0, 256, size=(nl, Hl, Wl, 3), dtype=np.uint8)
        pb = rng.standard_normal((nl, 17))
        pb[:, 4], pb[:, 9], pb[:, 14] = Hl, Wl, 6.5          # hwf column of the 3x5 pose
        pb[:, 15] = np.abs(pb[:, 15]) + 0.5
        pb[:, 16] = pb[:, 15] + np.abs(pb[:, 16]) + 1.0
        lroot = os.path.join(tmp, "llff") + "/"
        write_llff_tree(lroot, rgb, pb)
        _stdout = sys.stdout
        sys.stdout = open(os.devnull, "w")  # convert_npy prints a row (loader.py:42)
        try:
            dl = loader.NeRFDataset(root_dir=lroot, low_res=1, transform=None, type="llff", mode="train")  # runs convert_npy
        finally:
            sys.stdout.close()
            sys.stdout = _stdout
        out.update(l_rgb=rgb, l_poses_bounds=pb, l_new_npy=np.load(lroot + "new.npy"), l_all_pix=dl.all_pix.numpy(),
                   l_attrs=np.array([dl.pic_num, dl.height, dl.width, dl.pic_size, dl.num_pix], dtype=np.int64), l_focal=np.float64(dl.focal))
        lidx = np.array([0, Wl, Hl * Wl - 1, Hl * Wl, 2 * Hl * Wl + 3 * Wl + 2, nl * Hl * Wl - 1], dtype=np.int64)
        litems = [dl[int(i)] for i in lidx]
        out.update(l_idx=lidx, l_item_row=np.array([it[0] for it in litems], dtype=np.int64), l_item_col=np.array([it[1] for it in litems], dtype=np.int64),
                   l_item_pix=np.stack([it[2].numpy() for it in litems]), l_item_pose=np.stack([it[3] for it in litems]),
                   l_item_pic=np.array([it[4] for it in litems], dtype=np.int64))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    path = os.path.join(HERE, "data_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)} bytes): {sorted(out)}")


if __name__ == "__main__":
    main()
