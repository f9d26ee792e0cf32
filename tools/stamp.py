#!/usr/bin/env python3
"""Source stamp of the HIP library: sha256 over csrc/*.hip, *.h, the Makefile and include/ct_hip.h (sorted by path).
Every summary under profiles/ carries the stamp of the build it was measured on; bench.py quotes a profile-derived number
only when that stamp equals the stamp of the tree it runs from (a git commit id cannot serve: committing the profile would
change it, and .git does not travel to the GPU box).
usage: tools/stamp.py  -> prints the stamp"""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files():
    c = os.path.join(ROOT, "color-transfer_amd", "csrc")
    f = glob.glob(os.path.join(c, "*.hip")) + glob.glob(os.path.join(c, "*.h")) + [os.path.join(c, "Makefile"), os.path.join(ROOT, "include", "ct_hip.h")]
    return sorted(f)


def source_stamp():
    h = hashlib.sha256()
    for p in source_files():
        h.update(os.path.relpath(p, ROOT).encode() + b"\0")
        h.update(open(p, "rb").read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def read_stamped(path):
    """json of a profiles/ summary if it was measured on this tree's sources, else None"""
    import json
    if not os.path.exists(path):
        return None
    try:
        j = json.load(open(path))
    except ValueError:
        return None
    return j if isinstance(j, dict) and j.get("source_stamp") == source_stamp() else None


if __name__ == "__main__":
    print(source_stamp())
