"""The two ways the search kernel numbers a frame's work units (csrc/rmd_frame.hpp, seed_search_compact_kernel), restated on the host and checked for
what the kernel relies on: every unit of every shard is searched by exactly one workgroup, whatever the sixteen counts are.

  * heavy frames: the shards' lists read as one list (unit_entry): workgroup b starts with unit b, the rest is dealt by sixteen counters -- counter c
    deals the units n_wg + c + 16 k to the workgroups of class c (b % 16 == c), which draw until a draw lies beyond the last unit;
  * light frames (no shard holds more than n_wg / 16 units): entry i of shard s belongs to workgroup 16 i + s, nothing is dealt.

CPU test of the scheme's arithmetic (the kernel itself is covered bit for bit by the -m gpu parity tests)."""
import random

UNIT_SHARDS = 16


def unit_entry(counts, g):
    """(shard, index) of unit g of the concatenated lists -- the kernel's loop, literally"""
    acc, sh, sh_first = 0, 0, 0
    for q in range(UNIT_SHARDS - 1):
        acc += counts[q]
        if g >= acc:
            sh, sh_first = q + 1, acc
    return sh, g - sh_first


def searched_units(counts, n_wg):
    """every (shard, index) a launch searches, with multiplicity, under the kernel's rules"""
    n_units = sum(counts)
    light = max(counts) <= n_wg // UNIT_SHARDS
    taken = []
    if light:
        for b in range(n_wg):
            s, i = b % UNIT_SHARDS, b // UNIT_SHARDS
            if i < counts[s]:
                taken.append((s, i))
        return taken, light
    handout = n_units > n_wg
    dealt = [0] * UNIT_SHARDS  # the sixteen counters
    for b in range(n_wg):  # (the order in which workgroups draw does not matter: a counter deals every number once)
        if b >= n_units:
            continue
        taken.append(unit_entry(counts, b))
        while handout:
            c = b % UNIT_SHARDS
            u = n_wg + c + UNIT_SHARDS * dealt[c]
            dealt[c] += 1
            if u >= n_units:
                break
            taken.append(unit_entry(counts, u))
    # a workgroup that draws past the end stops; the numbers below n_units of every class must all have been dealt by then.  Classes whose
    # workgroups all stopped early cannot strand units: a class stops only at a number >= n_units, and its numbers are dealt in order.
    return taken, light


def _check(counts, n_wg):
    taken, light = searched_units(counts, n_wg)
    want = [(s, i) for s in range(UNIT_SHARDS) for i in range(counts[s])]
    assert sorted(taken) == want, (counts, n_wg, light)
    return light


def test_every_unit_is_searched_exactly_once():
    rng = random.Random(5)
    seen = {True: 0, False: 0}
    for _ in range(400):
        n_wg = rng.choice([1024, 1024, 1024, 1000, 96, 17, 16, 1216])
        scale = rng.choice([0, 1, 3, 20, 64, 65, 80, 400])
        counts = [rng.randint(0, scale) for _ in range(UNIT_SHARDS)]
        if rng.random() < 0.2:
            counts[rng.randrange(UNIT_SHARDS)] = n_wg // UNIT_SHARDS + rng.choice([0, 1])  # on the light / heavy border
        seen[_check(counts, n_wg)] += 1
    assert seen[True] > 50 and seen[False] > 50


def test_borders():
    assert _check([64] * 16, 1024) is True        # the fullest light frame: every workgroup has a unit
    assert _check([65] + [0] * 15, 1024) is False  # one shard over the border: the other numbering, 65 units
    assert _check([0] * 16, 1024) is True          # nothing to search
    assert _check([3] * 16, 40) is False           # fewer workgroups than units: units are dealt
    assert _check([2] * 16, 40) is True            # n_wg not a multiple of 16: workgroups 32..39 have index 2 >= every count
