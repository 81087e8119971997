"""Cut golden fixtures from the reference's own committed renders (run once, in the build container).

Reads /root/reference/renders/out_iso{0..255}.png (C60.vol, 512x512 RGB8, iso = file index; the
output of the reference driver src/main.zig:96-163) and writes

  tests/golden/c60_renders_512.npz   gray[256,512,512] uint8 (zlib-compressed; all three channels of
                                     every reference image are equal, which this script asserts)
  tests/golden/c60_renders_512.json  per-iso SHA-256 of the decoded RGB24 buffer + hit-pixel counts

and copies the two volumes the reference ships as test inputs:

  tests/data/C60.vol        (verbatim)
  tests/data/Skull.vol.gz   (verbatim copy of "Skull(1).vol.gz": 28-byte header + 68x256x256 u8)

Only data is copied -- no reference source text.  The GPU box has no /root/reference; tests read
these fixtures instead.
"""
import hashlib
import json
import os
import shutil

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(os.path.dirname(HERE), "data")


def main():
    os.makedirs(DATA, exist_ok=True)
    gray = np.zeros((256, 512, 512), dtype=np.uint8)
    meta = {"source": "renders/out_iso{N}.png of the reference snapshot", "width": 512, "height": 512,
            "volume": "C60.vol", "camera_pos": [-30.0, -300.0, -30.0], "viewport_width": 2.0, "isos": []}
    for iso in range(256):
        im = Image.open(os.path.join(REF, "renders", "out_iso%d.png" % iso))
        assert im.size == (512, 512), im.size
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
        assert (rgb[..., 0] == rgb[..., 1]).all() and (rgb[..., 1] == rgb[..., 2]).all()
        gray[iso] = rgb[..., 0]
        meta["isos"].append({"iso": iso, "sha256_rgb24": hashlib.sha256(rgb.tobytes()).hexdigest(),
                             "nonzero_pixels": int((rgb[..., 0] != 0).sum())})
    np.savez_compressed(os.path.join(HERE, "c60_renders_512.npz"), gray=gray)
    with open(os.path.join(HERE, "c60_renders_512.json"), "w") as f:
        json.dump(meta, f, indent=1)
    shutil.copyfile(os.path.join(REF, "C60.vol"), os.path.join(DATA, "C60.vol"))
    shutil.copyfile(os.path.join(REF, "Skull(1).vol.gz"), os.path.join(DATA, "Skull.vol.gz"))
    print("wrote", os.path.getsize(os.path.join(HERE, "c60_renders_512.npz")), "bytes of golden renders")


if __name__ == "__main__":
    main()
