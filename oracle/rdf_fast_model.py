"""Host-side model of the default RDF pair kernel's arithmetic.  TEST INFRASTRUCTURE ONLY.

The default kernel (``rdf_fast_kernel`` of lammps_analysis_amd/csrc/mdx_rdf.hip) does not
replay the reference's float32 operations (mdsuite/calculators/
radial_distribution_function.py:647-689 ``get_dij``, mdsuite/utils/linalg.py:84-99
``apply_minimum_image``, ``tf.histogram_fixed_width`` at :642-644) for every pair; it
evaluates an approximate bin position and re-evaluates only the pairs whose position lies
within an error bound of a bin edge.  Its exactness therefore rests on that bound.  This
module restates both sides in NumPy float32 -- the kernel's operations (every one of them
is an IEEE single-rounding operation except ``v_sqrt_f32``, which is modelled as the
correctly rounded root and its two float32 neighbours) and the literal rule -- so that
tests can hold the library's own constants (``mdx_rdf_fast_plan``) to the bound on pairs
built to break it.  Nothing under lammps_analysis_amd/ imports this file.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
F64 = np.float64


def fma32(a, b, c):
    """float32 fused multiply-add with a single rounding.

    The product of two float32 is exact in float64; its sum with ``c`` is rounded to
    odd in float64 (TwoSum gives the error term), which makes the final rounding to
    float32 equal to rounding the exact value once."""
    a = np.asarray(a, F32).astype(F64)
    b = np.asarray(b, F32).astype(F64)
    c = np.asarray(c, F32).astype(F64)
    p = a * b
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)
    s = np.array(s, F64, ndmin=1)
    err = np.broadcast_to(err, s.shape)
    even = (s.view(np.int64) & 1) == 0
    need = (err != 0.0) & even & np.isfinite(s)
    toward = np.where(err > 0.0, np.inf, -np.inf)
    s = np.where(need, np.nextafter(s, toward), s)
    return s.astype(F32)


def fold_half(r, h):
    """``h - ||r| - h|`` as two float32 subtractions (the kernel's ``fold_half``)."""
    t = (np.abs(r) - h).astype(F32)
    return (h - np.abs(t)).astype(F32)


def fast_bits(xi, xj, plan, sqrt_ulps=0, fold=7):
    """Bits of the kernel's position float for pairs ``xi``, ``xj`` ([n, 3] float32).

    ``sqrt_ulps`` moves the square root by that many float32 steps from the correctly
    rounded value (the hardware instruction is accurate to one).  ``fold`` (an integer or
    one per pair): bit c set = component c goes through the minimum-image fold, clear = the
    loop variant that takes the difference as it is, which the kernel picks only where
    ``|x_j - x_i| <= L_c / 2`` is known for the whole group of pairs (``foldable_mask``)."""
    xi = np.asarray(xi, F32)
    xj = np.asarray(xj, F32)
    h = np.asarray(plan["half_box"], F32)
    r = (xj - xi).astype(F32)
    fold = np.broadcast_to(np.asarray(fold, np.int64), (len(r),))
    m = [np.where((fold >> c) & 1 == 1, fold_half(r[:, c], h[c]), r[:, c]).astype(F32)
         for c in range(3)]
    s = fma32(m[2], m[2], fma32(m[1], m[1], (m[0] * m[0]).astype(F32)))
    root = np.sqrt(s).astype(F32)
    for _ in range(abs(int(sqrt_ulps))):
        root = np.nextafter(root, F32(np.inf) if sqrt_ulps > 0 else F32(0.0)).astype(F32)
    out = fma32(root, F32(plan["scale"]), F32(plan["magic"]))
    return out.view(np.uint32)


def foldable_mask(xi, xj, plan):
    """Per pair, the components the kernel could not leave unfolded: bit c set when
    ``|fl(x_j - x_i)|`` exceeds ``L_c / 2`` (the box criterion of the kernel guarantees the
    opposite for every pair of a group before it drops a fold)."""
    xi = np.asarray(xi, F32)
    xj = np.asarray(xj, F32)
    h = np.asarray(plan["half_box"], F32)
    r = (xj - xi).astype(F32)
    need = np.abs(r) > h[None, :]
    return (need[:, 0] * 1 + need[:, 1] * 2 + need[:, 2] * 4).astype(np.int64)


def literal_bins(xi, xj, box, cutoff, nbins):
    """(kept, bin, position) of the reference's operation sequence for the same pairs."""
    xi = np.asarray(xi, F32)
    xj = np.asarray(xj, F32)
    box = np.asarray(box, F32)
    r = (xj - xi).astype(F32)
    r = (r - (np.rint((r / box).astype(F32)) * box).astype(F32)).astype(F32)
    s = (((r[:, 0] * r[:, 0]).astype(F32) + (r[:, 1] * r[:, 1]).astype(F32)).astype(F32)
         + (r[:, 2] * r[:, 2]).astype(F32)).astype(F32)
    d = np.sqrt(s).astype(F32)
    cutoff = F32(cutoff)
    step = F64(F32(cutoff - F32(0.0))) / F64(nbins)
    pos = d.astype(F64) / step
    bins = np.minimum(pos, F64(nbins - 1)).astype(np.int64)
    return d < cutoff, bins, pos


def kernel_decision(bits, plan):
    """What the hot loop does with a pair: (dropped, resolved_literally, counted_bin)."""
    bits = np.asarray(bits, np.uint32)
    frac_bits = int(plan["frac_bits"])
    dropped = ~(bits < np.uint32(plan["keep_bits"]))  # NaN bit patterns compare above
    frac = bits & np.uint32((1 << frac_bits) - 1)
    near = (~dropped) & (frac < np.uint32(plan["edge_units"]))
    counted = ((bits & np.uint32(0x7FFFFF)) >> np.uint32(frac_bits)).astype(np.int64)
    return dropped, near, counted


def stored_position(bits, plan):
    """The fixed-point position the mantissa holds (bin position + shift), float64."""
    frac_bits = int(plan["frac_bits"])
    return (np.asarray(bits, np.uint32) & np.uint32(0x7FFFFF)).astype(F64) / F64(1 << frac_bits)


def check_pairs(xi, xj, box, cutoff, nbins, plan, fold=7):
    """Holds the kernel's decisions to the literal rule for every pair and every modelled
    square-root result; returns counts for the report.  Raises AssertionError.  ``fold``:
    the loop variant per pair (see ``fast_bits``)."""
    kept, bins, pos = literal_bins(xi, xj, box, cutoff, nbins)
    grid = 1.0 / (1 << int(plan["frac_bits"]))
    width = float(plan["edge_units"]) * grid  # 2 * shift
    report = {"pairs": int(len(kept)), "resolved": 0, "worst_low": 0.0, "worst_high": 0.0}
    for ulps in (-1, 0, 1):
        bits = fast_bits(xi, xj, plan, sqrt_ulps=ulps, fold=fold)
        dropped, near, counted = kernel_decision(bits, plan)
        bad_drop = dropped & kept
        assert not bad_drop.any(), (
            f"{int(bad_drop.sum())} pair(s) dropped by the kernel but inside the cutoff "
            f"(sqrt {ulps:+d} ulp), e.g. literal position {pos[bad_drop][0]!r}")
        direct = ~dropped & ~near
        bad_keep = direct & ~kept
        assert not bad_keep.any(), (
            f"{int(bad_keep.sum())} pair(s) counted unchecked but outside the cutoff "
            f"(sqrt {ulps:+d} ulp), e.g. literal position {pos[bad_keep][0]!r}")
        bad_bin = direct & (counted != bins)
        assert not bad_bin.any(), (
            f"{int(bad_bin.sum())} pair(s) counted unchecked in the wrong bin "
            f"(sqrt {ulps:+d} ulp), e.g. literal position {pos[bad_bin][0]!r} "
            f"counted in {int(counted[bad_bin][0])}")
        # the statement behind it: stored position in [pos_lit, pos_lit + 2 shift]
        inside = ~dropped & (pos <= nbins + 2.0)
        diff = stored_position(bits, plan)[inside] - pos[inside]
        if diff.size:
            report["worst_low"] = min(report["worst_low"], float(diff.min()))
            report["worst_high"] = max(report["worst_high"], float(diff.max()))
            assert diff.min() >= 0.0, f"stored position {diff.min()} below the literal one"
            assert diff.max() <= width, (
                f"stored position {diff.max()} above the literal one, allowed {width}")
        report["resolved"] += int(near.sum())
    return report


# --------------------------------------------------------------- adversarial pairs


def _nudge(x, ulps):
    """float32 array moved by an integer number of float32 steps per element."""
    x = np.asarray(x, F32)
    bits = x.view(np.int32).astype(np.int64)
    sign = np.where(bits < 0, -1, 1)
    mag = np.abs(x).view(np.int32).astype(np.int64) + np.asarray(ulps, np.int64) * 1
    mag = np.clip(mag, 0, 0x7F7FFFFF)
    return (mag.astype(np.int32).view(F32) * sign.astype(F32)).astype(F32)


def pairs_near_fold_points(box, n, rng):
    """Coordinate differences within a few float32 steps of 0, L/2 and L per component."""
    box = np.asarray(box, F32)
    xi = np.empty((n, 3), F32)
    xj = np.empty((n, 3), F32)
    for c in range(3):
        anchors = np.array([0.0, box[c] / 2, box[c]], F32)
        # x_i at a few places of the box, x_j = x_i +- anchor (+- ulps), kept inside [0, L]
        base = rng.choice(np.array([0.0, box[c] / 4, box[c] / 2, box[c]], F32), n)
        target = rng.choice(anchors, n)
        sign = rng.choice(np.array([-1.0, 1.0], F32), n)
        other = (base + sign * target).astype(F32)
        flip = (other < 0) | (other > box[c])
        other = np.where(flip, (base - sign * target).astype(F32), other).astype(F32)
        other = _nudge(other, rng.integers(-4, 5, n))
        other = np.clip(other, F32(0.0), box[c]).astype(F32)
        xi[:, c] = base
        xj[:, c] = other
    return xi, xj


def pairs_near_bin_edges(box, cutoff, nbins, edges, per_edge, rng, through_wrap=True):
    """Pairs whose distance lies within a few float32 steps of the given bin edges
    (edge k = k * cutoff / nbins; k = nbins is the cutoff itself)."""
    box = np.asarray(box, F32)
    step = F64(F32(cutoff)) / F64(nbins)
    edges = np.repeat(np.asarray(edges, np.int64), per_edge)
    n = len(edges)
    target = edges.astype(F64) * step
    # random direction with components that the box admits (|m_c| <= L_c / 2)
    direction = rng.normal(size=(n, 3))
    kind = rng.integers(0, 4, n)
    direction[kind == 0, 1:] = 0.0                    # along one axis
    direction[kind == 1, 2] = 0.0                     # in a plane
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    m = np.abs(direction) * target[:, None]
    for _ in range(8):  # shrink what sticks out of the half box, grow the rest
        over = m > (box / 2).astype(F64)
        if not over.any():
            break
        m = np.minimum(m, (box / 2).astype(F64) * 0.999)
        rest = target ** 2 - np.sum(np.where(over, m, 0.0) ** 2, axis=1)
        free = np.sum(np.where(over, 0.0, m) ** 2, axis=1)
        scale = np.sqrt(np.maximum(rest, 0.0) / np.maximum(free, 1e-300))
        m = np.where(over, m, m * scale[:, None])
    m32 = m.astype(F32)
    # last component re-solved in float64 and nudged by a few steps: both sides of the edge
    c_last = np.argmax(m32, axis=1)
    others = np.sum(np.where(np.arange(3)[None, :] == c_last[:, None], 0.0,
                             m32.astype(F64) ** 2), axis=1)
    solved = np.sqrt(np.maximum(target ** 2 - others, 0.0)).astype(F32)
    solved = _nudge(solved, rng.integers(-3, 4, n))
    m32[np.arange(n), c_last] = solved
    m32 = np.minimum(m32, (box / 2)[None, :]).astype(F32)  # a minimum image fits the half box
    xi = np.zeros((n, 3), F32)
    xj = m32.copy()
    if through_wrap:
        # half of the components reach the same separation through the periodic image
        wrap = rng.random((n, 3)) < 0.5
        xj = np.where(wrap, (box[None, :] - m32).astype(F32), m32).astype(F32)
    # random origin that keeps both atoms inside [0, L]
    lo = np.minimum(xi, xj)
    hi = np.maximum(xi, xj)
    room = (box[None, :] - (hi - lo)).astype(F32)
    shift = (rng.random((n, 3)).astype(F32) * np.maximum(room, 0)).astype(F32)
    moved_i = (xi - lo + shift).astype(F32)
    moved_j = (xj - lo + shift).astype(F32)
    ok = (moved_i >= 0).all(1) & (moved_j >= 0).all(1) & (moved_i <= box).all(1) & \
         (moved_j <= box).all(1)
    swap = rng.random(n) < 0.5
    out_i = np.where(swap[:, None], moved_j, moved_i)[ok]
    out_j = np.where(swap[:, None], moved_i, moved_j)[ok]
    # the unshifted pairs as well (exact small coordinates), as far as they are pairs of a
    # compact frame: every coordinate difference at most one box length
    fits = ((xj >= 0) & (xj <= box[None, :])).all(1)
    return (np.concatenate([out_i, xi[fits]]).astype(F32),
            np.concatenate([out_j, xj[fits]]).astype(F32))


def pairs_of_spread_frames(box, n, rng):
    """Pairs of a frame that spreads over up to 1.5 box lengths (atoms that left the box
    since they were last wrapped): coordinates in [-L/4, 5L/4], half of the pairs with
    differences within a few float32 steps of 0, L/2, L, 5L/4 and 3L/2 per component (at
    3L/2 the literal rint reaches 2; a frame the kernel takes stays below it), half uniform."""
    box = np.asarray(box, F32)
    xi = np.empty((n, 3), F32)
    xj = np.empty((n, 3), F32)
    for c in range(3):
        length = box[c]
        anchors = np.array([0.0, length / 2, length, 1.25 * length, 1.5 * length], F32)
        base = rng.choice(np.array([-length / 4, 0.0, length / 4], F32), n)
        other = (base + rng.choice(anchors, n)).astype(F32)
        other = _nudge(other, rng.integers(-4, 5, n))
        top = np.nextafter(F32(1.25) * length, F32(0.0))   # differences stay below 3L/2
        other = np.minimum(other, top).astype(F32)
        uniform = rng.random(n) < 0.5
        base = np.where(uniform, (rng.random(n).astype(F32) * F32(1.5) - F32(0.25)) * length,
                        base).astype(F32)
        other = np.where(uniform, (rng.random(n).astype(F32) * F32(1.5) - F32(0.25)) * length,
                         other).astype(F32)
        other = np.minimum(other, top).astype(F32)
        base = np.minimum(base, top).astype(F32)
        swap = rng.random(n) < 0.5
        xi[:, c] = np.where(swap, other, base)
        xj[:, c] = np.where(swap, base, other)
    return xi, xj


def uniform_pairs(box, n, rng):
    box = np.asarray(box, F32)
    xi = (rng.random((n, 3)).astype(F32) * box).astype(F32)
    xj = (rng.random((n, 3)).astype(F32) * box).astype(F32)
    return xi, xj


def plan_dict(plan) -> dict:
    """ctypes ``RdfFastPlan`` -> plain dict."""
    return {"usable": int(plan.usable), "frac_bits": int(plan.frac_bits),
            "edge_units": int(plan.edge_units), "keep_bits": int(plan.keep_bits),
            "magic": float(plan.magic), "scale": float(plan.scale), "s_cut": float(plan.s_cut),
            "half_box": [float(v) for v in plan.half_box], "step": float(plan.step)}
