"""CPU: the pairing oracle (suffix array + LCP restatement of build_vertex_set) against the
reference's own suffix tree code (oracle/_ref) and against the committed golden pairings."""
import gzip
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import pairing_lib as PL

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")


@needs_ref
@pytest.mark.parametrize("seed", [5, 6, 7])
def test_repeats_match_reference(seed):
    gen, ests = PL.repeat_workload(seed)
    oi, ri = PL.OracleIndex(gen), PL.RefIndex(gen)
    for L, rate in ((15, 0.2), (16, 0.2), (22, 0.2), (15, 0.5), (15, 0.0), (15, 1.0)):
        for e in ests:
            if not e:
                continue
            a, b = oi.pairings(e, L, rate), ri.pairings(e, L, rate)
            assert np.array_equal(a, b), (seed, L, rate, len(e), a[:5], b[:5])
    oi.close()


@needs_ref
def test_ambn_fixture_matches_reference():
    gen = PL.read_fasta(os.path.join(GOLD, "ambn", "genomic.txt"))[0]
    ests = PL.read_fasta(os.path.join(GOLD, "ambn", "ests.txt"))
    oi, ri = PL.OracleIndex(gen), PL.RefIndex(gen)
    tot = 0
    for e in ests + [PL.revcomp(x) for x in ests]:
        a, b = oi.pairings(e), ri.pairings(e)
        assert np.array_equal(a, b)
        tot += len(a)
    assert tot > 300
    oi.close()


def test_golden_pairings():
    """Reference outputs committed by tools/make_golden.py (runs anywhere)."""
    with gzip.open(os.path.join(GOLD, "pairings.json.gz"), "rt") as f:
        gold = json.load(f)
    assert len(gold["sets"]) >= 2
    n = 0
    for s in gold["sets"]:
        gen = s["genomic"].encode()
        oi = PL.OracleIndex(gen)
        for c in s["cases"]:
            got = oi.pairings(c["est"].encode(), c["L"], c["rate"])
            assert got.tolist() == c["pairings"], (c["L"], c["rate"], c["est"][:40])
            n += 1
        oi.close()
    assert n > 300
