#!/usr/bin/env python3
"""Writes a copy of an OBJ whose faces are in random order (every face keeps its material: `usemtl` is re-emitted per face).

    python tools/shuffle_obj.py models/Tor.obj /tmp/shuffled/Tor.obj [--seed 4]

The hierarchy the table builder lays over the triangles must not depend on the file order (results never do; speed must
not either): tests/test_cull_tables_host.py and tools/run_configs.py use this.
"""
import argparse
import os
import random
import shutil


def shuffle(src, dst, seed=4):
    lines = open(src).read().split("\n")
    head, faces, mtl, lib = [], [], None, None
    for l in lines:
        if l.startswith("usemtl"):
            mtl = l
        elif l.startswith("f "):
            faces.append((mtl, l))
        else:
            head.append(l)
            if l.startswith("mtllib"):
                lib = l.split()[1]
    order = list(range(len(faces)))
    random.Random(seed).shuffle(order)
    os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
    open(dst, "w").write("\n".join(head + [x for k in order for x in faces[k] if x]) + "\n")
    if lib and os.path.abspath(os.path.dirname(src)) != os.path.abspath(os.path.dirname(dst)):
        shutil.copy(os.path.join(os.path.dirname(src), lib), os.path.join(os.path.dirname(os.path.abspath(dst)), lib))
    return order


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--seed", type=int, default=4)
    a = ap.parse_args()
    print(len(shuffle(a.src, a.dst, a.seed)), "faces shuffled")
