"""CPU: the oracle reproduces the committed golden vectors (reference outputs, tests/golden/)
and the known answers of the reference's own unit tests."""
import json
import os

import dp_cases as D
import golden_cases as G


def test_burset_known_answers(O):
    ka = json.load(open(os.path.join(G.GOLD, "burset_known_answers.json")))
    assert len(ka["getBursetFrequency"]) >= 60
    for r in ka["getBursetFrequency"]:
        assert O.oracle().orc_burset_frequency(r["donor"].encode(), r["acceptor"].encode()) == r["freq"], r
    for r in ka["Check_Burset_patterns"]:
        # Check_Burset_patterns (src/refine-intron.c:346-360): donor = g[dl+1..dl+2], acceptor = g[ar-2..ar-1]
        g, dl, ar = r["genomic"].encode(), r["donor_left"], r["acceptor_right"]
        d = g[dl + 1:dl + 3] if 0 <= dl + 1 <= len(g) else b""
        a = g[ar - 2:ar] if 0 <= ar - 2 <= len(g) else b""
        assert O.oracle().orc_burset_frequency(d, a) == r["freq"], r


def test_golden_dp_calls(O):
    cases = G.load()
    assert len(cases) > 1000
    kinds = {c.kind for c, _ in cases}
    assert kinds == set(range(7))
    bad = [(c, e) for c, e in cases if not D.check_case(c, c.expected(O), O, expected=e)]
    assert not bad, "%d golden mismatches, first %r expected %r" % (len(bad), *bad[0])


def test_c3_sample_calls(O):
    """The oracle against every DP call (and answer) of the reference on a C3-shaped sample."""
    cases = G.load_c3_sample()
    assert len(cases) > 10000
    bad = [(c, e) for c, e in cases if not D.check_case(c, c.expected(O), O, expected=e)]
    assert not bad, "%d golden mismatches, first %r expected %r" % (len(bad), *bad[0])
