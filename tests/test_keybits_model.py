"""CPU model of KeyBits (dc3_order.hip.hpp, round 6): the sort image of a power-of-two alphabet read off a bit-packed copy of
the text.  What partition pass 1 relies on is checked here on small random texts, in numpy:
  * k_pack_bits' layout: lg bits per symbol, digit = code - 1, most significant bit first, groups of 8 symbols = lg bytes;
  * the image of position p = the stream's bits [p lg, p lg + nbits) through ONE 8-byte big-endian load at byte (p lg) >> 3
    shifted by (p lg) & 7 — also for the 4 consecutive positions a thread takes (images4), whose bits come from one load;
  * the image is a monotone (not strict) map of the suffix order: suffix(p) < suffix(q) implies image(p) <= image(q), also
    where a window runs into the zero bits behind the text (the smallest symbol and "no symbol" both read as digit 0: ties
    that the window compare settles, never inversions)."""
import numpy as np
import pytest


def pack_bits(digits, lg):
    """k_pack_bits: groups of 8 symbols -> lg bytes, first symbol in the top bits; 10 zero groups behind, then whatever the arena holds"""
    n = len(digits)
    groups = (n + 7) // 8 + 10             # order_all_positions' formula: 10 groups of zero digits behind the text
    d = np.zeros(groups * 8, dtype=np.uint64)
    d[:n] = digits
    # what lies behind the stream in the arena is NOT zero (a context that has built before): all ones here — an image that
    # reads it breaks the monotone map for the last suffixes of the text (the bug the round-6 soak found with + 2 groups)
    out = np.full(groups * lg + 64, 255, dtype=np.uint8)
    for g in range(groups):
        acc = 0
        for k in range(8):
            acc = (acc << lg) | int(d[8 * g + k])
        for i in range(lg):
            out[g * lg + i] = (acc >> (8 * (lg - 1 - i))) & 255
    return out


def image(bits, p, lg, nbits):
    """KeyBits::image_hi: one unaligned 8-byte load, byte swap, two shifts"""
    b = p * lg
    v = int.from_bytes(bits[b >> 3:(b >> 3) + 8].tobytes(), "big")
    return ((v << (b & 7)) & (2**64 - 1)) >> (64 - nbits)


def images4(bits, p0, lg, nbits):
    b = p0 * lg
    V = (int.from_bytes(bits[b >> 3:(b >> 3) + 8].tobytes(), "big") << (b & 7)) & (2**64 - 1)
    return [((V << (j * lg)) & (2**64 - 1)) >> (64 - nbits) for j in range(4)]


def want_image(digits, p, lg, nbits):
    """the first nbits bits of the digit string that starts at p, zero digits behind the end"""
    need = (nbits + lg - 1) // lg
    acc = 0
    for k in range(need):
        acc = (acc << lg) | (int(digits[p + k]) if p + k < len(digits) else 0)
    return acc >> (need * lg - nbits)


@pytest.mark.parametrize("lg", [1, 2, 3, 4])
def test_keybits_image_is_the_leading_bits_and_monotone(lg):
    rng = np.random.default_rng(100 + lg)
    sigma = 1 << lg
    for trial in range(6):
        n = int(rng.integers(40, 400))
        digits = rng.integers(0, sigma, size=n)
        if trial % 2:
            digits[-int(rng.integers(1, 20)):] = 0          # a run of the smallest symbol at the very end
        bits = pack_bits(digits, lg)
        for nbits in (34, 42, 44, 64 - 7 - 3 * lg):
            assert nbits + 7 + 3 * lg <= 64                  # the condition order_all_positions tests
            imgs = [image(bits, p, lg, nbits) for p in range(n)]
            assert imgs == [want_image(digits, p, lg, nbits) for p in range(n)]
            for p0 in range(0, n - 3, 4):                    # (p0 % 4 == 0, as the partition pass calls it)
                assert images4(bits, p0, lg, nbits) == imgs[p0:p0 + 4]
            # monotone along the true suffix order (codes = digit + 1, the end of the text below every symbol)
            text = bytes((digits + 1).astype(np.uint8))
            order = sorted(range(n), key=lambda p: text[p:])
            seq = [imgs[p] for p in order]
            assert all(a <= b for a, b in zip(seq, seq[1:])), (lg, nbits, trial)
