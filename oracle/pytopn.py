"""CPU restatement of fragment.top (fragment.go:1317-1437) — TEST INFRASTRUCTURE ONLY (nothing under
featurebase_amd/ may import this; see tests/test_abi.py).

`fragment_top` follows the reference line by line, including Go's container/heap (heap.Push = append
+ up, heap.Pop = swap + down + remove last; pairHeap.Less compares counts only, cache.go:439-445) and
the rank-cache input order (count descending, cache.go:371).  Two things the reference leaves
unspecified are fixed here so that the function is deterministic: rows of equal count enter in
ascending id order (Go sorts a map's entries with an unstable sort), and nothing else.  It is pinned
to the reference's own known answers (tests/golden/topn_vectors.json, extracted mechanically).

`execute_topn` composes it per SHARD into executeTopN's two passes (executor.go:2779-2864) and is pinned to the
expectations of TestExecutor_Execute_TopN, _TopN_fill, _TopN_fill_small and _TopN_Src
(tests/golden/executor_topn_vectors.json, extracted mechanically): this is what fbk_topn / fbk_group_topn return under
option topn_semantics = 1 (reference, the default).

`top_exact` is the specification of option topn_semantics = 0 (include/fbk.h fbk_topn): the same
threshold rules applied to EVERY row of every shard, no candidate pass, ordered count descending / id ascending."""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple


# ---- container/heap over []Pair with Less = Count < Count -------------------------------------
def _up(h: List[Tuple[int, int]], j: int) -> None:
    while True:
        i = (j - 1) // 2  # parent
        if i == j or j <= 0 or not h[j][1] < h[i][1]:
            break
        h[i], h[j] = h[j], h[i]
        j = i


def _down(h: List[Tuple[int, int]], i0: int, n: int) -> bool:
    i = i0
    while True:
        j1 = 2 * i + 1
        if j1 >= n or j1 < 0:
            break
        j = j1
        j2 = j1 + 1
        if j2 < n and h[j2][1] < h[j1][1]:
            j = j2
        if not h[j][1] < h[i][1]:
            break
        h[i], h[j] = h[j], h[i]
        i = j
    return i > i0


def heap_push(h, x):
    h.append(x)
    _up(h, len(h) - 1)


def heap_pop(h):
    n = len(h) - 1
    h[0], h[n] = h[n], h[0]
    _down(h, 0, n)
    return h.pop()


def fragment_top(rows: Dict[int, Iterable[int]], n: int = 0, src: Optional[Iterable[int]] = None, row_ids: Optional[Sequence[int]] = None,
                 min_threshold: int = 0, tanimoto_threshold: int = 0) -> List[Tuple[int, int]]:
    """fragment.top (fragment.go:1317-1437).  rows: row id -> columns; returns [(id, count)]."""
    rows = {k: set(v) for k, v in rows.items()}
    srcset = set(src) if src is not None else None
    # topBitmapPairs (:1439-1479): every cached row, or the requested ones that are not empty
    ids = sorted(rows) if not row_ids else [r for r in row_ids]
    pairs = [(r, len(rows.get(r, ()))) for r in ids if len(rows.get(r, ())) > 0]
    pairs.sort(key=lambda p: (-p[1], p[0]))
    if row_ids:
        n = 0  # :1326-1328
    tanimoto = 0
    min_t = max_t = 0.0
    src_count = 0
    if tanimoto_threshold > 0 and srcset is not None:  # :1334-1339
        tanimoto = tanimoto_threshold
        src_count = len(srcset)
        min_t = float(src_count * tanimoto) / 100
        max_t = float(src_count * 100) / float(tanimoto)
    results: List[Tuple[int, int]] = []
    for row_id, cnt in pairs:
        if cnt == 0:
            continue
        if tanimoto > 0:
            if float(cnt) <= min_t or float(cnt) >= max_t:
                continue
        elif cnt < min_threshold:
            continue
        if n == 0 or len(results) < n:  # :1365
            count = cnt
            if srcset is not None:
                count = len(srcset & rows[row_id])
            if count == 0:
                continue
            if tanimoto > 0:
                t = math.ceil(float(count * 100) / float(cnt + src_count - count))
                if t <= float(tanimoto):
                    continue
            elif count < min_threshold:
                continue
            heap_push(results, (row_id, count))
            if n > 0 and len(results) == n and srcset is None:
                break
            continue
        threshold = results[0][1]  # :1406
        if threshold < min_threshold or cnt < threshold:
            break
        count = len(srcset & rows[row_id])
        if count < threshold:
            continue
        heap_push(results, (row_id, count))
    out = [None] * len(results)  # :1428-1435: popped smallest first, filled from the back
    x, i = len(results), 1
    while results:
        out[x - i] = heap_pop(results)
        i += 1
    return out


def row_passes(cnt: int, count: int, src_count: int, has_src: bool, min_threshold: int, tanimoto_threshold: int) -> bool:
    """The qualification rule of fragment.top for ONE row of ONE shard, in integer arithmetic (equal to
    the reference's float64 comparisons for every count a shard can hold):
    cnt = row cardinality, count = |row ∩ src| (= cnt without a source row)."""
    if cnt == 0 or count == 0:
        return False
    if tanimoto_threshold > 0 and has_src:
        if cnt * 100 <= src_count * tanimoto_threshold or cnt * tanimoto_threshold >= src_count * 100:
            return False
        den = cnt + src_count - count
        return (count * 100 + den - 1) // den > tanimoto_threshold
    return cnt >= min_threshold and count >= min_threshold


def top_exact(shards: Sequence[Dict[int, Iterable[int]]], ids: Sequence[int], n: int = 0, srcs: Optional[Sequence[Iterable[int]]] = None,
              min_threshold: int = 0, tanimoto_threshold: int = 0) -> List[Tuple[int, int]]:
    """What fbk_topn returns: per shard the rows that pass row_passes contribute their count, counts
    are summed over the shards (Pairs.Add, cache.go:463), order = count descending, id ascending."""
    tot = {r: 0 for r in ids}
    for s, rows in enumerate(shards):
        srcset = set(srcs[s]) if srcs is not None else None
        for r in ids:
            cols = set(rows.get(r, ()))
            cnt = len(cols)
            count = len(cols & srcset) if srcset is not None else cnt
            if row_passes(cnt, count, len(srcset) if srcset is not None else 0, srcset is not None, min_threshold, tanimoto_threshold):
                tot[r] += count
    out = sorted([(r, c) for r, c in tot.items() if c], key=lambda p: (-p[1], p[0]))
    return out[:n] if n else out


def pairs_add(p: Dict[int, int], other: Sequence[Tuple[int, int]]) -> Dict[int, int]:
    """Pairs.Add (cache.go:463-483): counts of equal ids added up (the map form; the slice order it returns is Go's map
    order, i.e. unspecified — every caller sorts afterwards)."""
    for r, c in other:
        p[r] = p.get(r, 0) + c
    return p


def pairs_sorted(p: Dict[int, int]) -> List[Tuple[int, int]]:
    """sort.Sort(Pairs) (Less = Count >, cache.go:434): count descending; Go's sort is not stable and the input order is a
    map's, so ties are unspecified in the reference — fixed here (and in fbk_topn) as row id ascending."""
    return sorted(p.items(), key=lambda q: (-q[1], q[0]))


DEFAULT_MIN_THRESHOLD = 1  # executor.go:40-42


def execute_topn_shards(shards: Sequence[Dict[int, Iterable[int]]], n: int, srcs: Optional[Sequence[Optional[Iterable[int]]]], row_ids: Optional[Sequence[int]],
                        min_threshold: int, tanimoto_threshold: int) -> List[Tuple[int, int]]:
    """executeTopNShards (executor.go:2829-2864): fragment.top of EVERY shard with the call's own n (executeTopNShard
    :2869-2944 — MinThreshold 0 becomes 1, :2918-2920), merged with Pairs.Add in the reduce — nothing is trimmed here, by
    shard beyond fragment.top's own rule or by node — then sorted."""
    if min_threshold == 0:
        min_threshold = DEFAULT_MIN_THRESHOLD
    merged: Dict[int, int] = {}
    for s, rows in enumerate(shards):
        if not rows:  # no fragment for this shard (:2911-2913): an empty PairsField
            continue
        src = srcs[s] if srcs is not None else None
        pairs_add(merged, fragment_top(rows, n, src, row_ids, min_threshold, tanimoto_threshold))
    return pairs_sorted(merged)


def execute_topn(shards: Sequence[Dict[int, Iterable[int]]], n: int = 0, srcs: Optional[Sequence[Optional[Iterable[int]]]] = None,
                 ids_arg: Optional[Sequence[int]] = None, min_threshold: int = 0, tanimoto_threshold: int = 0) -> List[Tuple[int, int]]:
    """executeTopN (executor.go:2779-2827) on the coordinating node.  shards[s] = row id -> columns of shard s (how the
    shards are dealt to nodes does not matter: a remote node runs executeTopNShards over ITS shards with opt.Remote and
    returns the merged, untrimmed list (:2800-2806), and Pairs.Add is associative).
      pass 1 (:2795)       executeTopNShards with the call as written: per SHARD, fragment.top(N = n);
      early out (:2802)    no pairs, or the call named ids: the pass-1 list is the answer (NOT trimmed to n);
      ids (:2812-2816)     the keys of the pass-1 pairs, sorted;
      pass 2 (:2818)       the same call with those ids: fragment.top sets N = 0 when ids are given (fragment.go:1324-1327),
                           so every shard reports every id that passes the thresholds there;
      trim (:2823-2825)    the first n of the sorted merge."""
    pairs = execute_topn_shards(shards, n, srcs, ids_arg, min_threshold, tanimoto_threshold)
    if not pairs or ids_arg:
        return pairs
    ids = sorted(r for r, _ in pairs)
    trimmed = execute_topn_shards(shards, n, srcs, ids, min_threshold, tanimoto_threshold)
    if n != 0 and n < len(trimmed):
        trimmed = trimmed[:n]
    return trimmed


def topn_candidates(shards: Sequence[Dict[int, Iterable[int]]], n: int, srcs: Optional[Sequence[Optional[Iterable[int]]]] = None, min_threshold: int = 0,
                    tanimoto_threshold: int = 0) -> List[int]:
    """The sorted candidate ids of execute_topn's pass 1 (what fbk_topn_partials reports per member, as flags)."""
    return sorted(r for r, _ in execute_topn_shards(shards, n, srcs, None, min_threshold, tanimoto_threshold))
