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


def _first_attempt_megs(tmp_path, genomic_fasta, ests_fasta):
    """Host MEG code (checked against the reference's megs.txt in test_host_meg.py) + pairing oracle:
    first-attempt graph of every prepared sequence."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.run(["make", "-s", "-C", os.path.join(here, "hostcheck"), "meg_check"], check=True)
    (tmp_path / "genomic.txt").write_text(genomic_fasta)
    (tmp_path / "ests.txt").write_text(ests_fasta)
    subprocess.run([os.path.join(here, "hostcheck", "meg_check")], cwd=tmp_path, check=True,
                   env=dict(os.environ, MEG_CHECK_FIRST_ATTEMPT="1"), stderr=subprocess.DEVNULL)
    out = []
    for blk in (tmp_path / "megs-first.txt").read_text().split("@@end\n")[:-1]:
        head, rest = blk.split("\n", 1)
        cx, rest = rest.split("\n", 1)
        meg, edges = rest.split("@@edges\n")
        out.append(dict(seq=head[len("@@seq "):].encode(), complex=int(cx.split()[1]), meg=meg.encode(), edges=edges.encode()))
    return out, (tmp_path / "genomic-prepared.txt").read_bytes()


@pytest.mark.parametrize("source", ["c3", "ambn", "repeats"])
def test_meg_stage_vs_host_meg_code(gpu_ctx, tmp_path, source):
    """pgpu_pairing_plan_run_meg: for every prepared sequence (both strands) the finished graph --
    vertices in position-list order, adjacency in list order, the too_complex verdict and the two
    texts est-fact prints -- against the host MEG code over the pairing oracle."""
    import pintron_amd.capi as capi
    from pintron_amd import synth
    if source == "c3":
        w = synth.make("C3", n_est=400, seed=21)
        gfa, efa = w.genomic_fasta(), w.ests_fasta()
    elif source == "repeats":
        gfa, efa = synth.make_region_start_repeats()
    else:
        gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ambn")
        gfa, efa = open(os.path.join(gold, "genomic.txt")).read(), open(os.path.join(gold, "ests.txt")).read()
    exp, genomic = _first_attempt_megs(tmp_path, gfa, efa)
    assert len(exp) >= 40
    idx = capi.Index(gpu_ctx, genomic)
    plan = capi.PairingPlan(gpu_ctx, idx, [e["seq"] for e in exp])
    plan.run(15, 0.2)
    plan.run_meg()
    recs = [capi.parse_meg_record(r) for r in plan.fetch_meg()]
    plan.close()
    idx.close()
    n_unavailable = 0
    for e, r in zip(exp, recs):
        if r["flags"] & 2:
            n_unavailable += 1
            continue
        assert (r["flags"] & 1) == e["complex"]
        assert r["meg_text"] == e["meg"] and r["edges_text"] == e["edges"]
        # the structured part says what the text says
        lines = e["meg"].decode().split("#adj#\n")
        verts = [tuple(int(x) for x in ln.strip("()").split(",")) for ln in lines[0].splitlines()]
        assert [tuple(v) for v in r["vertices"]] == verts
        edges = [tuple(int(x) for x in ln.split("-")) for ln in lines[1].splitlines()]
        assert [(k, t) for k, a in enumerate(r["adj"]) for t in a] == edges
    assert n_unavailable * 20 <= len(exp)


def test_index_file_roundtrip(gpu_ctx, tmp_path):
    """pgpu_index_save / pgpu_index_load: the index read back from disk gives the same suffix array
    and the same pairings; a file made for another sequence is refused."""
    import pintron_amd.capi as capi
    from pintron_amd import synth
    w = synth.make("C2", n_est=60, seed=8)
    ests = list(w.est_seqs) + [PL.revcomp(e) for e in w.est_seqs]
    built = capi.Index(gpu_ctx, w.genomic)
    path = str(tmp_path / "gene.idx")
    built.save(path)
    loaded = capi.Index(gpu_ctx, w.genomic, load_from=path)
    assert np.array_equal(built.suffix_array(), loaded.suffix_array())
    out = []
    for ix in (built, loaded):
        plan = capi.PairingPlan(gpu_ctx, ix, ests)
        plan.run(15, 0.2)
        plan.run_meg()
        out.append((plan.fetch()[0].tobytes(), plan.fetch_meg()))
        plan.close()
    assert out[0] == out[1] and len(out[0][0]) > 1000
    other = bytes(w.genomic[:-1]) + (b"A" if w.genomic[-1:] != b"A" else b"C")
    with pytest.raises(capi.PgpuError):
        capi.Index(gpu_ctx, other, load_from=path)
    with pytest.raises(capi.PgpuError):
        capi.Index(gpu_ctx, w.genomic, load_from=str(tmp_path / "missing.idx"))
    built.close()
    loaded.close()


def test_long_exact_repeat_in_the_genomic(gpu_ctx):
    """A genomic sequence that contains a 30 kb exact duplicate: neighbouring suffixes share tens of
    thousands of characters, which is where a character-by-character LCP scan goes quadratic.  The
    index is built from the rank arrays of the doubling rounds (log n steps per pair) and the pairings
    -- whose thresholds come from LCP-interval borders -- must still be the oracle's."""
    import time
    import pintron_amd.capi as capi
    rng = random.Random(5)
    unit = bytes(rng.choice(b"ACGT") for _ in range(30000))
    flank = [bytes(rng.choice(b"ACGT") for _ in range(7000)) for _ in range(3)]
    gen = flank[0] + unit + flank[1] + unit + flank[2]
    ests = [gen[a:a + 400] for a in (6800, 7100, 20000, 36900, 43950, 44100, 60000, 73800)]
    ests += [PL.revcomp(e) for e in ests]
    t0 = time.time()
    got = gpu_pairings(gpu_ctx, gen, ests)
    assert time.time() - t0 < 20
    oi = PL.OracleIndex(gen)
    for e, g in zip(ests, got):
        assert np.array_equal(g.reshape(-1, 3), oi.pairings(e)), len(e)
    oi.close()


def test_resident_plans_share_their_scratch(gpu_ctx):
    """pgpu_pairing_plan_create_resident: plans that own only their patterns and write every run into buffers shared
    by the context's resident plans (the prefetch chunks of the batched host).  Chunks of DIFFERENT sizes, used in
    turn and out of order, give exactly the pairings and MEG records of ordinary plans -- whatever the other plan
    left in the shared buffers -- and results that another plan's run has overwritten are refused, not returned."""
    import pintron_amd.capi as capi
    from pintron_amd import synth
    w = synth.make("C3", n_est=900, seed=11)
    ests = []
    for s in w.est_seqs:
        ests += [s, PL.revcomp(s)]
    chunks = [ests[:200], ests[200:1400], ests[1400:1500], ests[1500:]]          # the second makes the pool grow
    idx = capi.Index(gpu_ctx, w.genomic)
    want = []
    for ch in chunks:                                                            # ordinary plans: the expectation
        p = capi.PairingPlan(gpu_ctx, idx, ch)
        p.run()
        tri, first = p.fetch()
        p.run_meg()
        want.append((tri.copy(), first.copy(), p.fetch_meg()))
        p.close()
    plans = [capi.PairingPlan(gpu_ctx, idx, ch, resident=True) for ch in chunks]
    try:
        for order in ([0, 1, 2, 3], [3, 1, 0, 2, 1]):
            for k in order:
                plans[k].run()
                tri, first = plans[k].fetch()
                assert np.array_equal(tri, want[k][0]) and np.array_equal(first, want[k][1]), k
                plans[k].run_meg()
                for i, (a, b) in enumerate(zip(plans[k].fetch_meg(), want[k][2])):
                    if a != b:
                        at = next((o for o in range(min(len(a), len(b))) if a[o] != b[o]), min(len(a), len(b)))
                        raise AssertionError("chunk %d record %d differs at byte %d of %d / %d: %r vs %r"
                                             % (k, i, at, len(a), len(b), a[max(0, at - 8):at + 8], b[max(0, at - 8):at + 8]))
        plans[0].run()
        plans[1].run()                                                           # overwrites what plan 0 left
        for stale in (plans[0].fetch, plans[0].run_meg, plans[0].fetch_meg):
            with pytest.raises(capi.PgpuError) as e:
                stale()
            assert "overwritten" in str(e.value)
        tri, first = plans[1].fetch()                                            # the owner is still served
        assert np.array_equal(tri, want[1][0])
        # an ordinary plan beside them keeps buffers of its own
        q = capi.PairingPlan(gpu_ctx, idx, chunks[2])
        q.run()
        plans[3].run()
        tri, first = q.fetch()
        assert np.array_equal(tri, want[2][0])
        q.close()
    finally:
        for p in plans:
            p.close()
        idx.close()
