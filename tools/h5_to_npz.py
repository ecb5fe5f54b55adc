#!/usr/bin/env python3
"""Convert a view-feature HDF5 file (`"{scan}_{viewpoint}"` -> [36, feat + prob], precompute_img_features_vit.py:148-159) into a
directory of `<key>.npy` files that vln_hamt_amd.data.r2r_data.ViewFeatureStore memory-maps -- for machines without h5py.
Run where h5py is installed:   python tools/h5_to_npz.py features.hdf5 out_dir [--fp16]"""
import os
import sys

import numpy as np

if __name__ == "__main__":
    import h5py
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    with h5py.File(src, "r") as f:
        for k in f:
            a = f[k][...]
            np.save(os.path.join(dst, k + ".npy"), a.astype(np.float16 if "--fp16" in sys.argv else np.float32))
    print("wrote", dst)
