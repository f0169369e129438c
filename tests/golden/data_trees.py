"""Writes tiny Blender / LLFF dataset trees from arrays (used by make_data_golden.py in the build container and by the
tests on any box: PNG is lossless, so the pixel arrays of the fixture reproduce the files the reference's loader read)."""
import json
import os

import numpy as np


def write_blender_tree(root, mode, rgba, mats, angle):
    """root/transforms_<mode>.json + root/<mode>/r_<i>.png (RGBA), the layout loader.py:12-36 / :95-110 reads."""
    from PIL import Image

    os.makedirs(os.path.join(root, mode), exist_ok=True)
    frames = []
    for i in range(rgba.shape[0]):
        Image.fromarray(np.ascontiguousarray(rgba[i]), "RGBA").save(os.path.join(root, mode, f"r_{i}.png"))
        frames.append({"file_path": f"./{mode}/r_{i}", "rotation": 0.012566370614359171, "transform_matrix": np.asarray(mats[i]).tolist()})
    with open(os.path.join(root, f"transforms_{mode}.json"), "w") as f:
        json.dump({"camera_angle_x": float(angle), "frames": frames}, f)


def write_llff_tree(root, rgb, poses_bounds):
    """root/poses_bounds.npy + root/images/image_<i>.png (RGB), the layout loader.py:38-53 / :95-110 reads."""
    from PIL import Image

    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    np.save(os.path.join(root, "poses_bounds.npy"), np.asarray(poses_bounds))
    for i in range(rgb.shape[0]):
        Image.fromarray(np.ascontiguousarray(rgb[i]), "RGB").save(os.path.join(root, "images", f"image_{i:03d}.png"))
