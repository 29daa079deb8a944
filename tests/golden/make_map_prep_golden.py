"""Generates tests/golden/map_prep_golden.npz from the REFERENCE's own
MapPlanner<Dim>::updatePotentialMap / setSearchRegion (compiled from
/root/reference by `make -C oracle ref`).  Run in the build container:

    python tests/golden/make_map_prep_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
from test_map_prep import CASES  # noqa: E402

out = {}
for name, dim, grid, md, org, res, centre, pots, path, regs in CASES:
    for k, (radius, rng_, pw) in enumerate(pots):
        out["%s/pot%d" % (name, k)] = O.update_potential_map(grid, md, org, res, centre, radius, rng_, pw, ref=True)
    for k, (sr, dense) in enumerate(regs):
        out["%s/reg%d" % (name, k)] = np.packbits(O.search_region(md, org, res, path, sr, dense, ref=True))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "map_prep_golden.npz"), **out)
print("wrote %d arrays" % len(out))
