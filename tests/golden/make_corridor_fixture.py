#!/usr/bin/env python3
"""Converts the reference's only map fixture, data/corridor.yaml (config C1 of
BASELINE.json: test_planner_2d), into tests/golden/corridor_map.npz.

Follows the reference's MapReader (test/read_map.hpp:13-45): keys start, goal,
origin, dim, resolution, data; cells are mapped data > 0 -> 100, else 0.  The
YAML lives only in the build container (/root/reference), so the converted map
is committed (bit-packed, a few KiB) with this script as its provenance.
"""
import os

import numpy as np
import yaml

SRC = "/root/reference/data/corridor.yaml"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "corridor_map.npz")

doc = yaml.safe_load(open(SRC))
cfg = {}
for item in doc:
    cfg.update(item)
data = np.asarray(cfg["data"], dtype=np.int64)
dim = [int(x) for x in cfg["dim"]]
assert data.size == dim[0] * dim[1]
occ = data > 0
np.savez_compressed(DST, start=np.asarray(cfg["start"], float), goal=np.asarray(cfg["goal"], float),
                    origin=np.asarray(cfg["origin"], float), dim=np.asarray(dim, np.int64),
                    resolution=float(cfg["resolution"]), occupied_bits=np.packbits(occ), n_cells=occ.size,
                    raw_counts=np.asarray([int((data == -1).sum()), int((data == 0).sum()), int((data == 100).sum())]))
print("wrote", DST, os.path.getsize(DST), "bytes; occupied", int(occ.sum()), "of", occ.size)
