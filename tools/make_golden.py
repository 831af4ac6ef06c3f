#!/usr/bin/env python3
"""Generates tests/golden/synth_*.npz: seeded synthetic frames with the CPU oracle's outputs.

These are "self-golden" vectors (today's-constant detector output and group ids
are unpinned by the reference, so the oracle is the definition).  They let the GPU box check
the HIP path against committed numbers as well as against the oracle built there.
    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from librectify_amd import synth  # noqa: E402

CASES = [("synth_96x64_s11", 96, 64, 11, 10), ("synth_257x131_s12", 257, 131, 12, 14), ("synth_320x240_s13", 320, 240, 13, 24)]


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, W, H, seed, bars in CASES:
        img = synth.frame(W, H, seed, bars=bars)
        f = O.filter_stage(img)
        seeds = O.find_seeds(f["mag"], f["bin"])
        det = O.find_line_segments(img)
        groups, _ = O.find_line_segment_groups(img, float(max(W, H)) / 100.0, seed=0)
        T = O.compute_rectification_transform(groups, W, H)
        extra = dict(dx=f["dx"], dy=f["dy"]) if W * H <= 8192 else {}  # keep the big fixtures small
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            image=img,
            dmask=f["dmask"],
            **extra,
            seed_idx=(seeds["rows"].astype(np.int64) * W + seeds["cols"]).astype(np.int32),
            seed_bin=seeds["bins"],
            label=det["label"],
            raw_lines=det["lines"],
            grouped_lines=groups,
            transform=O.transform_to_array(T),
        )
        print(name, "seeds", len(seeds["rows"]), "raw", len(det["lines"]), "grouped", len(groups))


if __name__ == "__main__":
    main()
