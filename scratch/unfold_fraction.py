"""How often a component of (64 i-atoms, 64 j-atoms) needs no minimum-image fold when the atoms
of a species are ordered along a Hilbert curve of 2^bits cells per axis: numpy estimate."""
import sys
import numpy as np

def hilbert_rank(q, bits):
    X = [q[:, 0].copy(), q[:, 1].copy(), q[:, 2].copy()]
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(3):
            hit = (X[i] & Q) != 0
            X[0] = np.where(hit, X[0] ^ P, X[0])
            t = np.where(hit, 0, (X[0] ^ X[i]) & P)
            X[0] ^= t
            X[i] ^= t
        Q >>= 1
    for i in range(1, 3):
        X[i] ^= X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for i in range(3):
        X[i] ^= t
    rank = np.zeros_like(X[0])
    for j in range(bits - 1, -1, -1):
        for i in range(3):
            rank = (rank << 1) | ((X[i] >> j) & 1)
    return rank

def groups(pos, L, bits, mode):
    if mode == "hilbert":
        q = np.minimum((pos / L * (1 << bits)).astype(np.int64), (1 << bits) - 1)
        order = np.argsort(hilbert_rank(q, bits), kind="stable")
    else:  # slabs: equal counts in x, then y, then z
        n = len(pos)
        nx = ny = int(mode.split("x")[0]); nz = int(mode.split("x")[2])
        order = np.argsort(pos[:, 0], kind="stable")
        out = []
        for sx in np.array_split(order, nx):
            sx = sx[np.argsort(pos[sx, 1], kind="stable")]
            for sy in np.array_split(sx, ny):
                out.append(sy[np.argsort(pos[sy, 2], kind="stable")])
        order = np.concatenate(out)
    p = pos[order]
    n = len(p) // 64 * 64
    g = p[:n].reshape(-1, 64, 3)
    return g.min(axis=1), g.max(axis=1)

def unmixed_fraction(lo_a, hi_a, lo_b, hi_b, L):
    h = L / 2
    d1 = hi_b[None] - lo_a[:, None]
    d2 = hi_a[:, None] - lo_b[None]
    ok = (d1 <= h) & (d2 <= h)
    return ok.mean(), np.bincount(ok.sum(axis=2).ravel(), minlength=4) / ok[..., 0].size

rng = np.random.default_rng(0)
n, L = int(sys.argv[1]), float(sys.argv[2])
a, b = rng.uniform(0, L, (n, 3)), rng.uniform(0, L, (n, 3))
for mode, bits in (("hilbert", 3), ("hilbert", 4), ("hilbert", 5), ("4x4x5", 0), ("5x5x3", 0)):
    la, ha = groups(a, L, bits, mode); lb, hb = groups(b, L, bits, mode)
    f, by_count = unmixed_fraction(la, ha, lb, hb, L)
    print(mode, bits, "mean extent %.1f" % (ha - la).mean(), "unfolded fraction %.3f" % f, "by number of unfolded components", np.round(by_count, 3))

# brute-force check of the box criterion on the first species pair, Hilbert bits 4
q = np.minimum((a / L * 16).astype(np.int64), 15)
pa = a[np.argsort(hilbert_rank(q, 4), kind="stable")][: n // 64 * 64].reshape(-1, 64, 3)
q = np.minimum((b / L * 16).astype(np.int64), 15)
pb = b[np.argsort(hilbert_rank(q, 4), kind="stable")][: n // 64 * 64].reshape(-1, 64, 3)
ext = pa.max(axis=1) - pa.min(axis=1)
print("extent percentiles", np.percentile(ext, [5, 25, 50, 75, 95]).round(1))
cnt = tot = 0
for i in range(0, len(pa), 7):
    for j in range(0, len(pb), 5):
        r = np.abs(pb[j][None, :, :] - pa[i][:, None, :])
        cnt += (r.max(axis=(0, 1)) <= L / 2).sum(); tot += 3
print("brute force unfolded fraction %.3f" % (cnt / tot))

# components whose differences all lie beyond half a box (one image shift for the whole group pair)
la, ha = groups(a, L, 3, "hilbert"); lb, hb = groups(b, L, 3, "hilbert")
h = L / 2
up = (lb[None] - ha[:, None] > h)      # every x_j - x_i above h
down = (la[:, None] - hb[None] > h)    # every x_i - x_j above h
inside = (hb[None] - la[:, None] <= h) & (ha[:, None] - lb[None] <= h)
print("inside %.3f shifted %.3f" % (inside.mean(), (up | down).mean()))
