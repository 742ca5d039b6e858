"""numpy model of the wide mode's deepening by rank look-ups (stringsearch_amd/csrc/dc3_wide.hip.hpp: k_wide_eq,
k_wide_isa_scatter, k_wide_ties_isa; host: wide_deepen in dc3_global_host.hpp), with the SAME bookkeeping as the kernels:
the order by D-symbol windows (ties in any order), eq[i] = "window of entry i equals that of entry i - 1", isa[p] = 1 + index
of the first entry of p's group (isa[n] = 0: the empty suffix), every group re-ordered by the look-ups isa[p + jD], j = 0..W,
new flags from the same compare, depth x (W + 1) per round.  The rank split of the real thing does not change the
arithmetic (a group never straddles two ranks) and is modelled by shuffling the ties of the starting order.
TEST INFRASTRUCTURE ONLY."""
import numpy as np


def order_by_windows(t, D, rng):
    """entries sorted by their D-symbol window (a suffix that ends sorts before its extensions), ties in random order"""
    n = len(t)
    pad = np.concatenate([t.astype(np.int64) + 1, np.zeros(D, dtype=np.int64)])          # 0 = past the end
    win = np.stack([pad[k:k + n] for k in range(D)], axis=1)
    tie_break = rng.permutation(n)
    order = np.lexsort((tie_break,) + tuple(win[:, k] for k in range(D - 1, -1, -1))).astype(np.int64)
    sw = win[order]
    eq = np.zeros(n, dtype=bool)
    eq[1:] = (sw[1:] == sw[:-1]).all(axis=1)
    return order, eq


def isa_of(sa, eq):
    """k_wide_isa_scatter"""
    n = len(sa)
    start = np.maximum.accumulate(np.where(~eq, np.arange(n), 0))
    isa = np.zeros(n + 1, dtype=np.int64)
    isa[sa] = start + 1
    return isa


def keys(isa, n, p, D, W):
    """the W + 1 look-ups of wide_cmp_isa for positions p, as rows"""
    j = np.arange(W + 1, dtype=np.int64)
    q = np.minimum(p[:, None] + j[None, :] * D, n)
    return isa[q]


def deepen(t, D, W, rng, max_rounds=64):
    """returns (sa, rounds)"""
    n = len(t)
    sa, eq = order_by_windows(t, D, rng)
    rounds = 0
    while eq.any():
        assert rounds < max_rounds, "suffixes still tied"
        isa = isa_of(sa, eq)
        grp = np.maximum.accumulate(np.where(~eq, np.arange(n), 0))       # index of the group's first entry
        k = keys(isa, n, sa, D, W)
        # k_wide_ties_isa: inside a group, order by the look-ups (lexicographic); across groups nothing moves
        perm = np.lexsort(tuple(k[:, c] for c in range(W, -1, -1)) + (grp,))
        sa, k, grp = sa[perm], k[perm], grp[perm]
        assert (grp == np.sort(grp)).all()
        neweq = np.zeros(n, dtype=bool)
        neweq[1:] = (grp[1:] == grp[:-1]) & (k[1:] == k[:-1]).all(axis=1)
        eq = neweq
        D *= W + 1
        rounds += 1
    return sa, rounds
