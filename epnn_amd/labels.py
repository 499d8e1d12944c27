"""Label tooling: HORTON MBIS multipole text output -> per-atom charge arrays.

Mirror of the reference's data/horton_txt2npy.py:1-20: every `*-mtp.txt` under `path` holds a 4-line header and then one
line per atom whose 5th space-separated token is the MBIS charge; the charges go to `<name>-mtp.npy` next to it (the
label files gen_padded_init_state looks for are `<xyz name>.npy`, charge_gn.py:302-310 -- the reference's datasets were
renamed by hand afterwards).  Host-side file conversion; nothing here touches the GPU.

One difference, a fix: the reference opens `os.path.join(path, filename)` while walking sub-directories too
(horton_txt2npy.py:9), which fails for files below the top level; this version opens the file where os.walk found it.
"""
from __future__ import annotations

import os

import numpy as np


def read_mtp_charges(txt_path):
    """Charges of one `-mtp.txt` file: token 4 (split on single spaces, like the reference) of every line after the 4th."""
    with open(txt_path, "r") as f:
        lines = f.readlines()
    return np.array([float(line.split(" ")[4]) for line in lines[4:]])


def horton_txt2npy(path="SSI_outputs_h"):
    """Convert every `*-mtp.txt` under `path`; returns the list of written .npy files."""
    written = []
    for root, _dirs, files in os.walk(path):
        for filename in sorted(files):
            if filename.endswith("-mtp.txt"):
                charges = read_mtp_charges(os.path.join(root, filename))
                np_name = os.path.join(root, filename[:-4] + ".npy")
                np.save(np_name, charges, allow_pickle=True)
                written.append(np_name)
    return written


if __name__ == "__main__":
    import sys
    for name in horton_txt2npy(sys.argv[1] if len(sys.argv) > 1 else "SSI_outputs_h"):
        print(name)
