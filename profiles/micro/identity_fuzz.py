"""Randomised check of the node-identity pass (mplx_post_packed_device on synthetic hashes) against numpy:

    python profiles/micro/identity_fuzz.py [first_seed] [n_cases]

Per case: 0.3 - 4 M pairs; the number of distinct keys anywhere between 1 and all; multiplicities uniform or heavy-tailed;
now and then the two unstorable keys (~0 and the hash whose mixed key is ~0); the claimed or the exact partition, automatic or
forced digit bits, tiny tables (MPLX_POST_FILL) now and then.  canon[] and the first-occurrence flag, entry for entry."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import motion_primitive_library_amd as m  # noqa: E402

M = (1 << 64) - 1


def unmix(v):
    u = lambda x: x ^ (x >> 33)
    v = u(v); v = (v * pow(0xc4ceb9fe1a85ec53, -1, 1 << 64)) & M
    v = u(v); v = (v * pow(0xff51afd7ed558ccd, -1, 1 << 64)) & M
    return u(v)


def want_canon(h):
    idx = np.arange(h.size, dtype=np.int64)
    order = np.lexsort((idx, h))
    hs, ids = h[order], idx[order]
    first = np.concatenate([[True], hs[1:] != hs[:-1]])
    cs = np.maximum.accumulate(np.where(first, np.arange(ids.size), 0))
    out = np.empty(h.size, np.int64)
    out[order] = ids[cs]
    return out


first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
wl = m.workloads.make("C2", scale=0.125, n_nodes=8)
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
F = 4 * wl.dim + 2
os.environ["MPLX_POST_PARTITION_MIN"] = "0"
bad, forms = 0, {}
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([300_000, 700_001, 1_500_000, 2_999_999, 4_000_000]))
    k = int(rng.choice([1, 7, 1000, n // 100, n // 15, n // 2, n]))
    keys = rng.integers(1, 1 << 63, size=k, dtype=np.int64).astype(np.uint64)
    if rng.random() < 0.5:
        pick = rng.integers(0, k, size=n)
    else:
        pick = np.minimum((rng.pareto(1.1, size=n) * k / 50).astype(np.int64), k - 1)
    h = keys[pick]
    if rng.random() < 0.3:
        sp = rng.choice(n, size=200, replace=False)
        h[sp[:100]] = np.uint64(M)
        h[sp[100:]] = np.uint64(unmix(M))
    knobs = {"MPLX_POST_CLAIMED": str(int(rng.random() < 0.7))}
    if rng.random() < 0.3:
        knobs["MPLX_POST_BITS"] = "%d,%d" % (int(rng.integers(1, 7)), int(rng.integers(1, 9)))
    if rng.random() < 0.2:
        knobs["MPLX_POST_FILL"] = str(int(rng.choice([40, 300, 900])))
    for var in ("MPLX_POST_CLAIMED", "MPLX_POST_BITS", "MPLX_POST_FILL"):
        os.environ.pop(var, None)
    os.environ.update(knobs)
    offs, hd, st = m.env.DeviceArray(env, 16), m.env.DeviceArray(env, n * 8), m.env.DeviceArray(env, F * n * 8)
    offs.upload(np.array([0, n], dtype=np.int64))
    hd.upload(h)
    ps = m._abi.PackedLists()
    ps.count, ps.offs, ps.action, ps.cost, ps.hash, ps.state = None, offs.ptr, None, None, hd.ptr, st.ptr
    ps.state_stride, ps.capacity = n, n
    got = env.post_packed(ps, 1, np.zeros(F))
    form = env.last_identity_form()
    forms[form] = forms.get(form, 0) + 1
    want = want_canon(h)
    ok = np.array_equal(got["canon"].astype(np.int64), want) and np.array_equal((got["flags"] & 4) != 0, want == np.arange(n))
    bad += 0 if ok else 1
    print("case %d %s: n %d, %d keys (%d distinct), %s, form %s" % (seed, "ok" if ok else "FAILED", n, k, np.unique(h).size, knobs, form))
    for b in (offs, hd, st):
        b.free()
env.close()
print("identity fuzz: %d of %d cases failed; forms %s" % (bad, count, forms))
sys.exit(1 if bad else 0)
