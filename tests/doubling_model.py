"""numpy model of the prefix-doubling finish (stringsearch_amd/csrc/dc3_doubling.hip.hpp, doubling_finish in dc3hip.hip):
the whole-text order by W-symbol windows, the tied positions refined by doubling with the SAME bookkeeping as the kernels
(ranks of untied positions = their index in the window order, groups of tied positions in a position-sorted map, records
sorted by (group, rank of p + d), new slot = gid + (j - g0), new group = gid + (r0 - g0), singletons dropped).
Also the names-from-longer-windows argument used when the order is handed to level 1 instead: see test_doubling_model.py.
TEST INFRASTRUCTURE ONLY."""
import numpy as np


def window_order(t, W):
    """positions sorted by their W-symbol window (sentinel -1 past the end), neq[i] = window(i) != window(i-1)."""
    n = len(t)
    pad = np.concatenate([t.astype(np.int64), -np.ones(W, dtype=np.int64)])
    win = np.stack([pad[k:k + n] for k in range(W)], axis=1)
    order = np.lexsort(tuple(win[:, k] for k in range(W - 1, -1, -1))).astype(np.int64)
    sw = win[order]
    neq = np.ones(n, dtype=bool)
    neq[1:] = (sw[1:] != sw[:-1]).any(axis=1)
    return order, neq


def doubling_finish(t, W, max_rounds=64):
    """returns (sa, tied, rounds)"""
    n = len(t)
    order, neq = window_order(t, W)
    sa = order.copy()                                         # k_emit_sorted
    tied = ~neq | np.concatenate([~neq[1:], [False]])         # k_dbl_count / k_dbl_collect
    slot = np.nonzero(tied)[0]
    if len(slot) == 0:
        return sa, 0, 0
    pos = order[slot]
    start = neq[slot]
    last_start = np.maximum.accumulate(np.where(start, np.arange(len(slot)), 0))
    gid = slot[last_start]                                    # k_dbl_gid
    mp = np.argsort(pos, kind="stable")                       # map sorted by position
    map_pos = pos[mp]
    map_val = gid[mp].copy()
    mapidx = np.empty(len(slot), dtype=np.int64); mapidx[mp] = np.arange(len(slot))
    rank_of = np.empty(n, dtype=np.int64); rank_of[order] = np.arange(n)   # (the kernels find this by binary search)
    a_gid, a_pos, a_map = gid.copy(), pos.copy(), mapidx.copy()
    d, rounds = W, 0
    while len(a_pos) and rounds < max_rounds:
        q = a_pos + d                                         # k_dbl_key
        key2 = np.zeros(len(q), dtype=np.int64)
        inside = q < n
        qi = q[inside]
        k = np.searchsorted(map_pos, qi)
        is_tied = (k < len(map_pos)) & (map_pos[np.minimum(k, len(map_pos) - 1)] == qi)
        val = np.where(is_tied, map_val[np.minimum(k, len(map_pos) - 1)], rank_of[qi])
        key2[inside] = val + 1
        o = np.lexsort((key2, a_gid))                         # the two radix sorts
        a_gid, a_pos, a_map, key2 = a_gid[o], a_pos[o], a_map[o], key2[o]
        j = np.arange(len(a_pos))                             # k_dbl_regroup
        fg = np.ones(len(j), dtype=bool); fg[1:] = a_gid[1:] != a_gid[:-1]
        fr = fg.copy(); fr[1:] |= key2[1:] != key2[:-1]
        g0 = np.maximum.accumulate(np.where(fg, j, 0))
        r0 = np.maximum.accumulate(np.where(fr, j, 0))
        new_slot = a_gid + (j - g0)
        new_gid = a_gid + (r0 - g0)
        sa[new_slot] = a_pos
        map_val[a_map] = new_gid
        nxt = np.concatenate([fr[1:], [True]])
        keep = ~(fr & nxt)
        a_gid, a_pos, a_map = new_gid[keep], a_pos[keep], a_map[keep]
        d *= 2; rounds += 1
    assert len(a_pos) == 0, "doubling did not converge"
    return sa, int(len(slot)), rounds
