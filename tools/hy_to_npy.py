#!/usr/bin/env python3
"""Convert the reference's region-feature store (`size,rcnn_arch,224.hy`: HDF5 dataset 'att' [n_img,36,2048] float32,
datasets.py:405-412) into the `.npy` file feed.FeatureStore memory-maps.  Needs h5py (not part of the build image: run it
wherever the dataset was extracted).  For a contiguous (un-chunked) dataset it also prints the byte offset of the array
inside the .hy file: FeatureStore(path_to_hy, names, shape=..., offset=...) then reads the original file in place.

    python tools/hy_to_npy.py /data/VQA/preprocess/size,rcnn_arch,224.hy /data/VQA/preprocess/size,rcnn_arch,224.npy"""
import sys

import numpy as np


def main():
    try:
        import h5py
    except ImportError:
        sys.exit("hy_to_npy.py needs h5py (pip install h5py) -- it is a one-off conversion on the machine that holds the dataset")
    src, dst = sys.argv[1:3]
    with h5py.File(src, "r") as hy:
        att = hy["att"]
        offset = att.id.get_offset()          # None for chunked / compressed layouts
        print("att: shape %s dtype %s, contiguous at byte offset %s" % (att.shape, att.dtype, offset))
        out = np.lib.format.open_memmap(dst, mode="w+", dtype=np.float32, shape=att.shape)
        step = 1024
        for lo in range(0, att.shape[0], step):
            out[lo:lo + step] = att[lo:lo + step]
        out.flush()
    print("wrote", dst)


if __name__ == "__main__":
    main()
