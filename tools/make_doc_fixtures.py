#!/usr/bin/env python3
"""Regenerate tests/golden/doc_* from the reference's doc/ artefacts.

Run in the build container only (needs /root/reference and PIL):
    python tools/make_doc_fixtures.py

Outputs (data only, no reference source text):
  tests/golden/doc_image_gray.npy      uint8 563x1000 luma of doc/image.jpg
                                       (PIL 'L' = ITU-R 601, close to OpenCV BGR2GRAY
                                       which autorectify.cpp:329-330 uses)
  tests/golden/doc_warp_lines.csv      copy of doc/image.jpg_warp_lines.csv (848 rows)
  tests/golden/doc_warp_tform.csv      copy of doc/image.jpg_warp_tform.csv (6 rows)
"""
import os, shutil, sys
import numpy as np
from PIL import Image

REF = "/root/reference/doc"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

def main():
    im = Image.open(os.path.join(REF, "image.jpg")).convert("L")
    a = np.asarray(im, dtype=np.uint8)
    assert a.shape == (563, 1000), a.shape
    np.save(os.path.join(OUT, "doc_image_gray.npy"), a)
    shutil.copyfile(os.path.join(REF, "image.jpg_warp_lines.csv"), os.path.join(OUT, "doc_warp_lines.csv"))
    shutil.copyfile(os.path.join(REF, "image.jpg_warp_tform.csv"), os.path.join(OUT, "doc_warp_tform.csv"))
    print("wrote fixtures to", os.path.normpath(OUT))

if __name__ == "__main__":
    sys.exit(main())
