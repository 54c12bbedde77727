"""GPU: suffix array and pairings (MEG vertex sets) from the HIP index vs the CPU oracle and the
golden pairings captured from the reference."""
import gzip
import json
import os
import random

import numpy as np
import pytest

import pairing_lib as PL

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gpu_pairings(ctx, gen, ests, L=15, rate=0.2):
    import pintron_amd.capi as capi
    idx = capi.Index(ctx, gen)
    plan = capi.PairingPlan(ctx, idx, ests)
    try:
        plan.run(L, rate)
        tri, first = plan.fetch()
        return [tri[int(first[i]):int(first[i + 1])] for i in range(len(ests))]
    finally:
        plan.close()
        idx.close()


def test_suffix_array_matches_oracle(gpu_ctx):
    import pintron_amd.capi as capi
    rng = random.Random(4)
    for gen in (b"", b"A", b"ACGT", b"AAAAAAAAAA", b"ACACACACACAC", PL.repeat_workload(8)[0],
                bytes(rng.choice(b"ACGTN") for _ in range(50_000)),
                PL.read_fasta(os.path.join(GOLD, "ambn", "genomic.txt"))[0]):
        idx = capi.Index(gpu_ctx, gen)
        sa = idx.suffix_array() if gen else np.zeros(0, dtype=np.uint32)
        idx.close()
        oi = PL.OracleIndex(gen)
        assert np.array_equal(sa, oi.sa()), len(gen)
        oi.close()


def test_golden_pairings(gpu_ctx):
    with gzip.open(os.path.join(GOLD, "pairings.json.gz"), "rt") as f:
        gold = json.load(f)
    n = 0
    for s in gold["sets"]:
        gen = s["genomic"].encode()
        by_param = {}
        for c in s["cases"]:
            by_param.setdefault((c["L"], c["rate"]), []).append(c)
        for (L, rate), cases in by_param.items():
            got = gpu_pairings(gpu_ctx, gen, [c["est"].encode() for c in cases], L, rate)
            for c, g in zip(cases, got):
                assert g.tolist() == c["pairings"], (L, rate, c["est"][:50])
                n += 1
    assert n > 300


@pytest.mark.parametrize("seed", [11, 12])
def test_repeats_vs_oracle(gpu_ctx, seed):
    gen, ests = PL.repeat_workload(seed, gen_len=60000)
    ests = ests + [PL.revcomp(e) for e in ests]
    oi = PL.OracleIndex(gen)
    for L, rate in ((15, 0.2), (18, 0.3)):
        got = gpu_pairings(gpu_ctx, gen, ests, L, rate)
        for e, g in zip(ests, got):
            assert np.array_equal(g.reshape(-1, 3), oi.pairings(e, L, rate)), (L, rate, len(e))
    oi.close()


def test_c3_shape_vs_oracle(gpu_ctx):
    """C3-shaped batch (200 kb genomic, 3 % errors): 3 000 ESTs, both strands."""
    from pintron_amd import synth
    w = synth.make("C3", n_est=1500)
    ests = []
    for s in w.est_seqs:
        ests += [s, PL.revcomp(s)]
    got = gpu_pairings(gpu_ctx, w.genomic, ests)
    oi = PL.OracleIndex(w.genomic)
    tot = 0
    for e, g in zip(ests, got):
        exp = oi.pairings(e)
        assert np.array_equal(g.reshape(-1, 3), exp), len(e)
        tot += len(exp)
    assert tot > 10000
    oi.close()
