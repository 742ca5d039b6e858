"""numpy model of the splitter (sample) ordering of dc3_ssort.hip.hpp — the bookkeeping of the three passes, with the
freedom the GPU passes have (arbitrary order inside every reserved region) made explicit by a random generator:

  sample      one record per cell of n / S records, at a fixed pseudo-random offset (ss_sample_index)
  splitters   fine[j] = sorted sample value (j + 1) * over; coarse[b] = fine[(b + 1) * F2 - 1]
  pass 1      tiles of `tile` records; group g owns the tiles [g * cpx, (g + 1) * cpx); every (bucket, group) pair has its
              own region, sized by the counting kernel; a tile appends its records of a bucket to the region in ANY order
  pass 2      the tiles are those of the bucket list (never across a bucket); a bucket's records are counted, and
              reserve, in the plane of the group of the bucket's FIRST tile
  pass 3      every sub-bucket is ordered by (key, pos)

The result must be the records in ascending (key, pos) order whatever the random generator does, and every region must
be filled exactly.  Values are (key, pos) pairs with distinct pos; Python ints for keys (any width)."""
import numpy as np

GROUPS = 8


def sample_index(i, n, S):
    lo, hi = i * n // S, (i + 1) * n // S
    h = (i * 0x9E3779B1) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x85EBCA77) & 0xFFFFFFFF
    h ^= h >> 13
    return min(lo + h % (hi - lo), n - 1)


def count_le(splitters, v):
    """how many splitters are <= v (splitters ascending): the digit of v"""
    lo, hi = 0, len(splitters)
    while lo < hi:
        mid = (lo + hi) // 2
        if splitters[mid] <= v:
            lo = mid + 1
        else:
            hi = mid
    return lo


def splitter_order(recs, nb1, F2, over, tile, rng):
    """recs: list of (key, pos) in position order.  Returns the ordered list and a dict of facts the tests look at."""
    n = len(recs)
    n2 = nb1 * F2
    S = n2 * over
    assert S * 4 <= n
    sample = sorted(recs[sample_index(i, n, S)] for i in range(S))
    fine = [sample[(j + 1) * over] for j in range(n2 - 1)]
    coarse = [fine[(b + 1) * F2 - 1] for b in range(nb1 - 1)]
    # ---- pass 1 ----
    ntiles = -(-n // tile)
    cpx = -(-ntiles // GROUPS)
    dig1 = [count_le(coarse, r) for r in recs]
    cnt = np.zeros((nb1, GROUPS), dtype=np.int64)
    for t in range(ntiles):
        g = t // cpx
        for i in range(t * tile, min(n, (t + 1) * tile)):
            cnt[dig1[i], g] += 1
    start1 = np.concatenate([[0], np.cumsum(cnt.reshape(-1))])            # region (b, g) = [start1[b*8+g], ...)
    cur = start1[:-1].copy()
    b1 = [None] * n
    tiles = list(range(ntiles)); rng.shuffle(tiles)                        # tiles run in any order
    for t in tiles:
        g = t // cpx
        idx = list(range(t * tile, min(n, (t + 1) * tile))); rng.shuffle(idx)
        for i in idx:
            k = dig1[i] * GROUPS + g
            b1[cur[k]] = recs[i]; cur[k] += 1
    assert np.array_equal(cur, start1[1:]), "pass 1: a region was not filled exactly"
    # ---- pass 2 ----
    bstart = start1[::GROUPS]                                              # bucket b = [bstart[b], bstart[b + 1])
    tpre = np.concatenate([[0], np.cumsum([-(-(int(bstart[b + 1]) - int(bstart[b])) // tile) for b in range(nb1)])])
    T2 = int(tpre[-1]); cpx2 = max(1, -(-T2 // GROUPS))
    cnt2 = np.zeros((n2, GROUPS), dtype=np.int64)
    dig2 = [0] * n
    straddling = 0
    for b in range(nb1):
        g = int(tpre[b]) // cpx2                                           # the group of the bucket's FIRST tile
        if tpre[b + 1] > tpre[b] and (int(tpre[b + 1]) - 1) // cpx2 != g:
            straddling += 1
        spl = fine[b * F2:b * F2 + F2 - 1]
        for i in range(int(bstart[b]), int(bstart[b + 1])):
            dig2[i] = count_le(spl, b1[i])
            cnt2[b * F2 + dig2[i], g] += 1
    start2 = np.concatenate([[0], np.cumsum(cnt2.reshape(-1))])
    cur2 = start2[:-1].copy()
    a2 = [None] * n
    order = list(range(T2)); rng.shuffle(order)
    for t in order:
        b = int(np.searchsorted(tpre, t, side="right")) - 1
        g = int(tpre[b]) // cpx2
        lo = int(bstart[b]) + (t - int(tpre[b])) * tile
        idx = list(range(lo, min(lo + tile, int(bstart[b + 1])))); rng.shuffle(idx)
        for i in idx:
            k = (b * F2 + dig2[i]) * GROUPS + g
            a2[cur2[k]] = b1[i]; cur2[k] += 1
    assert np.array_equal(cur2, start2[1:]), "pass 2: a region was not filled exactly"
    # ---- pass 3 ----
    out = []
    sub_start = start2[::GROUPS]
    sizes = []
    for s in range(n2):
        part = a2[int(sub_start[s]):int(sub_start[s + 1])]
        sizes.append(len(part))
        out.extend(sorted(part))
    return out, {"largest_sub_bucket": max(sizes), "mean": n / n2, "straddling_buckets": straddling, "T2": T2, "cpx2": cpx2}
