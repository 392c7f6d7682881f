#!/usr/bin/env python3
"""An OPEN scene for the skybox legs: models/Tor.obj without its back wall, and a generated sky bitmap.

    python tools/make_open_scene.py --out-dir /tmp/open        ->  TorOpen.obj, Tor.mtl, sky.bmp

The back wall is the last four `f` lines of Tor.obj (tests/test_gpu_configs.py builds its open scenes the same way): with them
gone every primary ray that passes the torus leaves through the back, so -- unlike in the closed room, where 98 % of the paths
run all -MRR segments -- most paths end on their first or second segment (scene.cpp:125-155: a miss ends the path, with a skybox
it also contributes).  The sky is a smooth gradient with a sun disc, written as a 24-bit BMP (bitmap_image.hpp's format).
"""
import argparse
import os
import shutil
import struct


def write_sky_bmp(path, w=256, h=128):
    rows = []
    for y in range(h):                 # top-down in memory, written bottom-up
        row = bytearray()
        for x in range(w):
            t = y / (h - 1)
            b, g, r = int(255 - 90 * t), int(200 - 120 * t), int(120 + 60 * t)
            dx, dy = (x - 0.7 * w) / w, (y - 0.25 * h) / h
            if dx * dx + dy * dy < 0.002:
                b, g, r = 235, 250, 255
            row += bytes((b, g, r))
        row += b"\0" * ((4 - (3 * w) % 4) % 4)
        rows.append(bytes(row))
    size_image = len(rows[0]) * h
    with open(path, "wb") as f:
        f.write(struct.pack("<HIHHI", 19778, 54 + size_image, 0, 0, 54))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, size_image, 0, 0, 0, 0))
        for row in reversed(rows):
            f.write(row)


def generate(models_dir, out_dir, name="TorOpen.obj", sky="sky.bmp", source="Tor.obj", source_dir=None):
    os.makedirs(out_dir, exist_ok=True)
    lines = open(os.path.join(source_dir or models_dir, source)).read().split("\n")
    faces = [i for i, l in enumerate(lines) if l.startswith("f ")]
    for i in faces[-4:]:
        lines[i] = ""
    open(os.path.join(out_dir, name), "w").write("\n".join(lines))
    if os.path.abspath(models_dir) != os.path.abspath(out_dir):
        shutil.copy(os.path.join(models_dir, "Tor.mtl"), os.path.join(out_dir, "Tor.mtl"))
    write_sky_bmp(os.path.join(out_dir, sky))
    return len(faces) - 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out-dir", required=True)
    ap.add_argument("--models", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models"))
    a = ap.parse_args()
    print(f"{a.out_dir}/TorOpen.obj: {generate(a.models, a.out_dir)} triangles, sky.bmp 256x128")


if __name__ == "__main__":
    main()
