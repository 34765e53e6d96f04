#!/usr/bin/env python3
"""Convert one of the reference's HDF5 captures (captured_data.py:94-108, 136-149) to the .npz that
drt_amd.captured_data reads where h5py is not installed.   usage: h5_to_npz.py hand.h5 [hand.npz]"""
import sys

import h5py
import numpy as np

src = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else src.rsplit(".", 1)[0] + ".npz"
with h5py.File(src, "r") as f:
    keys = [k for k in ("cam_proj", "cam_k", "screen_position", "mask", "ray_origin", "ray_dir") if k in f]
    np.savez(dst, **{k: f[k][...] for k in keys})
print(f"{dst}: {', '.join(keys)}")
