"""The sweep kernel finds the LDS slot of tile T as  li - nl * umulhi(li, magic)  with magic = floor(2^32 / nl) + 1
(gmrm_amd/csrc/sweep.hip: carve_for, lds_slot) instead of li % nl -- exact only while li < 2^32 / nl.  li is at most the number
of tiles of a launch (16 order positions each), and gmrm_sweep_launch refuses launches of 2^24 - 2 markers or more (the packed
exchange's 24-bit tags), so li < 2^20.  ADVICE r3 (low): nothing checked the bound of the round-3 ring's modulo; this pins the
arithmetic for every slot count the carve can produce, far beyond the bound the launch enforces."""
import numpy as np


def test_modulo_by_multiplication_is_exact_for_every_tile_of_a_legal_launch():
    li = np.arange(0, 1 << 22, dtype=np.uint64)                      # four times the largest tile index of a legal launch
    for nl in range(1, 31):                                          # LDS tile slots: 1 .. NLMAX (30 at R = 1)
        magic = np.uint64((1 << 32) // nl + 1)
        got = li - np.uint64(nl) * ((li * magic) >> np.uint64(32))
        assert np.array_equal(got, li % np.uint64(nl)), nl
        assert (1 << 22) < (1 << 32) // nl                           # the range checked lies inside the exact range of the identity


def test_the_identity_does_fail_beyond_its_range():
    """Negative control: the bound is real (so the launch limit matters)."""
    nl = 15
    magic = (1 << 32) // nl + 1
    bad = [x for x in range((1 << 32) // nl * 2, (1 << 32) // nl * 2 + 4 * nl) if x - nl * ((x * magic) >> 32) != x % nl]
    assert bad
