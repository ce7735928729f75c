"""fbk_bsi_range_sum's per-plane schedule (host arithmetic, exported as fbk_bsi_range_sum_plan) against the oracle:
the schedule is executed here on Python sets exactly as k_bsi_range_sum_slot executes it on bit planes — matched /
remaining, the constant high part of a column at the plane where it is matched, the low planes added as they come —
and the totals must equal fragment.sum over fragment.rangeOp (oracle/pybsi.py), for every operation, predicates around
every special form of the reference (0, +-1, saturated, beyond the bit depth), several depths."""
import ctypes as C

import numpy as np
import pytest

import datagen as D
from featurebase_amd import lib as L

U64 = (1 << 64) - 1


def plan(op, depth, pred):
    lib = L.load()
    act = (C.c_uint8 * 64)()
    vhi = (C.c_uint64 * 64)()
    sp, to = C.c_uint32(), C.c_uint32()
    rc = lib.fbk_bsi_range_sum_plan(op, depth, C.c_int64(pred), act, vhi, C.byref(sp), C.byref(to))
    assert rc in (0, 1), rc
    return None if rc else (list(act), list(vhi), bool(sp.value), bool(to.value))


def run_plan(values, depth, pl):
    """values: column -> signed value (sign-magnitude planes, fragment.go:62-65).  -> (sum mod 2^64 as int64, count)"""
    act, vhi, scan_positive, take_other = pl
    pos = {c for c, v in values.items() if v >= 0}
    neg = set(values) - pos
    X = set(pos if scan_positive else neg)
    O = set(neg if scan_positive else pos) if take_other else set()
    M = set()
    sum_m = sum_o = 0
    for i in range(depth - 1, -1, -1):
        plane = {c for c, v in values.items() if (abs(v) >> i) & 1}
        sum_m += len(M & plane) << i
        sum_o += len(O & plane) << i
        a = act[i]
        if a == 1:
            X &= plane
        elif a == 2:
            X -= plane
        elif a in (3, 4):
            new = ((X & plane) if a == 3 else (X - plane)) - M
            sum_m += len(new) * vhi[i]
            M |= new
    total = (sum_m - sum_o) if scan_positive else (sum_o - sum_m)
    total &= U64
    return total - (1 << 64) if total >> 63 else total, len(M) + len(O)


def oracle_range_sum(B, frag, op, depth, pred):
    rng = B.bsi_range(frag, op, depth, pred)
    return B.bsi_sum(frag, rng, True)


@pytest.mark.parametrize("depth", [1, 5, 12, 63, 64])
def test_schedule_equals_range_then_sum(oracle, depth):
    from oracle import pybsi as B

    B._lib()
    rng = D.rng_for(8100, depth)
    lim = (1 << depth) - 1
    ncol = 300
    cols = rng.choice(1 << 20, size=ncol, replace=False)
    mags = [int(rng.integers(0, lim + 1)) if depth < 63 else int(rng.integers(0, 1 << 62)) * 2 + int(rng.integers(0, 2)) for _ in range(ncol)]
    mags[:6] = [0, 0, lim, lim, 1, min(lim, 2)]  # zeros (of both signs below), the all-ones magnitude
    if depth == 64:
        mags = [min(m, (1 << 63) - 1) for m in mags]  # int64 values
        lim = (1 << 63) - 1
    signs = [1 if rng.random() < 0.5 else -1 for _ in range(ncol)]
    values = {int(c): m * s for c, m, s in zip(cols, mags, signs)}
    frag = B.bsi_fragment_from_values(values, depth)
    some = [int(abs(v)) for v in list(values.values())[:12]]
    preds = {0, 1, -1, 2, -2, lim, -lim, lim - 1, -(lim - 1), lim + 1 if lim < (1 << 63) - 1 else lim, -(lim + 1), (1 << 63) - 1, -(1 << 63)}
    for m in some:
        preds |= {m, -m, m + 1, -(m + 1), max(m - 1, 0)}
    fused = two_pass = 0
    for name, op in B.OPS.items():
        for pred in sorted(p for p in preds if -(1 << 63) <= p < (1 << 63)):
            pl = plan(L.BSI_OPS[name], depth, pred)
            if pl is None:
                two_pass += 1
                continue
            fused += 1
            exp = oracle_range_sum(B, frag, op, depth, pred)
            assert run_plan(values, depth, pl) == (int(exp[0]), int(exp[1])), (name, pred, depth)
    assert fused > (20 if depth > 1 else 0) and two_pass > 0


def test_special_forms_take_the_two_pass_path():
    for name, pred in (("EQ", 5), ("NEQ", 5), ("GT", 0), ("GTE", 0), ("LT", 0), ("LTE", 0), ("GT", -1), ("LT", 1)):
        assert plan(L.BSI_OPS[name], 16, pred) is None, (name, pred)
    assert plan(L.BSI_OPS["LT"], 8, 255) is None and plan(L.BSI_OPS["LT"], 8, 300) is None  # saturated / beyond the depth
    assert plan(L.BSI_OPS["GT"], 8, 300) is None
    assert plan(L.BSI_OPS["GT"], 8, 7) is not None and plan(L.BSI_OPS["LTE"], 8, -7) is not None


@pytest.mark.parametrize("depth", [1, 2, 3, 4, 5, 6])
def test_schedule_exhaustive_small_depths(oracle, depth):
    """Every magnitude of the depth with both signs (one column each), every operation, every predicate from below
    -(2^depth) to above 2^depth: the schedule (where there is one) gives the oracle's Range-then-Sum totals."""
    from oracle import pybsi as B

    B._lib()
    top = 1 << depth
    values, col = {}, 0
    for m in range(top):
        for sgn in (1, -1):
            values[col * 37 + (col % 5) * 70000] = m * sgn  # spread over several containers
            col += 1
    frag = B.bsi_fragment_from_values(values, depth)
    fused = 0
    for name, op in B.OPS.items():
        for pred in range(-top - 3, top + 4):
            pl = plan(L.BSI_OPS[name], depth, pred)
            if pl is None:
                continue
            fused += 1
            exp = oracle_range_sum(B, frag, op, depth, pred)
            assert run_plan(values, depth, pl) == (int(exp[0]), int(exp[1])), (name, pred, depth)
    assert fused >= 4 * (top - 2)  # LT / LTE / GT / GTE, most predicates inside the depth


# ---- lo <= v <= hi ----------------------------------------------------------------------------------------------
def between_plan(depth, lo, hi):
    lib = L.load()
    act = (C.c_uint8 * 128)()
    vhi = (C.c_uint64 * 128)()
    split = (C.c_uint8 * 64)()
    vfin = (C.c_uint64 * 2)()
    flags = (C.c_uint32 * 5)()
    rc = lib.fbk_bsi_between_sum_plan(depth, C.c_int64(lo), C.c_int64(hi), act, vhi, split, vfin, flags)
    assert rc in (0, 1), rc
    if rc:
        return None
    return dict(act=[list(act[:64]), list(act[64:])], vhi=[list(vhi[:64]), list(vhi[64:])], split=list(split), vfin=list(vfin),
                class_pos=[bool(flags[0]), bool(flags[1])], init_b=bool(flags[2]), whole=[bool(flags[3]), bool(flags[4])])


def run_between_plan(values, depth, pl):
    """The two scan lanes of k_bsi_between_sum_half on Python sets."""
    pos = {c for c, v in values.items() if v >= 0}
    neg = set(values) - pos
    cls = [pos if pl["class_pos"][0] else neg, (pos if pl["class_pos"][1] else neg) if pl["init_b"] else set()]
    X = [set() if pl["whole"][l] else set(cls[l]) for l in (0, 1)]
    M = [set(cls[l]) if pl["whole"][l] else set() for l in (0, 1)]
    s = [0, 0]
    n = [len(M[0]), len(M[1])]
    for i in range(depth - 1, -1, -1):
        plane = {c for c, v in values.items() if (abs(v) >> i) & 1}
        for l in (0, 1):
            s[l] += len(M[l] & plane) << i
        if pl["split"][i]:
            X[1] |= X[0] & plane
            X[0] -= plane
        for l in (0, 1):
            a = pl["act"][l][i]
            if a == 1:
                X[l] &= plane
            elif a == 2:
                X[l] -= plane
            elif a in (3, 4):
                new = (X[l] & plane) if a == 3 else (X[l] - plane)
                s[l] += len(new) * pl["vhi"][l][i]
                n[l] += len(new)
                M[l] |= new
                X[l] -= new
    for l in (0, 1):
        s[l] += len(X[l]) * pl["vfin"][l]
        n[l] += len(X[l])
    total = sum(s[l] if pl["class_pos"][l] else -s[l] for l in (0, 1)) & U64
    return total - (1 << 64) if total >> 63 else total, n[0] + n[1]


@pytest.mark.parametrize("depth", [1, 2, 3, 4, 5])
def test_between_schedule_exhaustive_small_depths(oracle, depth):
    """Every magnitude with both signs, every pair of bounds from below -(2^depth) to above 2^depth."""
    from oracle import pybsi as B

    B._lib()
    top = 1 << depth
    values, col = {}, 0
    for m in range(top):
        for sgn in (1, -1):
            values[col * 37 + (col % 5) * 70000] = m * sgn
            col += 1
    frag = B.bsi_fragment_from_values(values, depth)
    fused = 0
    for lo in range(-top - 2, top + 3):
        for hi in range(-top - 2, top + 3):
            pl = between_plan(depth, lo, hi)
            if pl is None:
                continue
            fused += 1
            exp = B.bsi_sum(frag, B.bsi_range_between(frag, depth, lo, hi), True)
            assert run_between_plan(values, depth, pl) == (int(exp[0]), int(exp[1])), (lo, hi, depth)
    assert fused > top  # most pairs lo < hi


@pytest.mark.parametrize("depth", [12, 33, 63, 64])
def test_between_schedule_random_bounds(oracle, depth):
    from oracle import pybsi as B

    B._lib()
    rng = D.rng_for(8200, depth)
    lim = (1 << min(depth, 63)) - 1
    ncol = 400
    cols = rng.choice(1 << 20, size=ncol, replace=False)
    mags = [int(rng.integers(0, lim + 1)) if lim < (1 << 62) else int(rng.integers(0, 1 << 62)) * 2 + int(rng.integers(0, 2)) for _ in range(ncol)]
    mags[:5] = [0, 0, lim, lim, 1]
    values = {int(c): m * (1 if rng.random() < 0.5 else -1) for c, m in zip(cols, mags)}
    frag = B.bsi_fragment_from_values(values, depth)
    stored = sorted(values.values())
    edges = [0, 1, -1, lim, -lim, (1 << 63) - 1, -(1 << 63)] + [stored[int(j)] for j in rng.integers(0, ncol, 10)]
    edges += [e + 1 for e in edges if e < (1 << 63) - 1] + [e - 1 for e in edges if e > -(1 << 63)]
    fused = 0
    for _ in range(400):
        lo, hi = sorted(int(edges[int(j)]) for j in rng.integers(0, len(edges), 2))
        pl = between_plan(depth, lo, hi)
        if pl is None:
            continue
        fused += 1
        exp = B.bsi_sum(frag, B.bsi_range_between(frag, depth, lo, hi), True)
        assert run_between_plan(values, depth, pl) == (int(exp[0]), int(exp[1])), (lo, hi, depth)
    assert fused > 100
