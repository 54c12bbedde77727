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


@needs_ref
def test_region_start_copies_match_reference():
    """The occurrence t == 0 reported at an upper level of the reference's tree comes out once per
    symbol slice walked (src/max-emb-graph.c:168-216): the oracle's closed form against the
    reference's own build_vertex_set on fresh random cases (the committed golden set holds others)."""
    n_dup = n = 0
    for gen, ests in PL.region_start_cases(101, n_cases=40):
        oi, ri = PL.OracleIndex(gen), PL.RefIndex(gen)
        for e in ests:
            for L, rate in ((15, 0.2), (18, 0.1)):
                a, b = oi.pairings(e, L, rate).tolist(), ri.pairings(e, L, rate).tolist()
                assert a == b, (gen[:70], e[:70], L, rate)
                n_dup += len(set(map(tuple, b))) != len(b)
                n += 1
        oi.close()
    assert n_dup >= 20 and n >= 200


def test_golden_set_holds_repeated_pairings():
    with gzip.open(os.path.join(GOLD, "pairings.json.gz"), "rt") as f:
        gold = json.load(f)
    rep = sum(1 for s in gold["sets"] for c in s["cases"] if len(set(map(tuple, c["pairings"]))) != len(c["pairings"]))
    assert rep >= 30
