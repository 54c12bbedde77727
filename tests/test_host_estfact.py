"""CPU: the complete est-fact host program (pintron_amd/host/) against the reference's outputs.

Two check builds (tests/hostcheck/Makefile), both with the CPU oracle where the product has the GPU:
  estfact_check        one EST after the other, oracle as the backend
  estfact_sched_check  the product's own main + fibre scheduler + GPU backend code, linked against
                       a CPU stand-in of the C-ABI (fake_pgpu.c), 1 and 4 threads
Goldens: tests/golden/ambn/expected-* (reference est-fact on regressionTest/test-AMBN); on a
seeded synthetic C2-shaped sample the compiled reference (oracle/_ref) is run side by side."""
import filecmp
import os
import shutil
import subprocess

import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "ambn")
FILES = ["raw-multifasta-out.txt", "processed-ests.txt", "megs.txt", "processed-megs.txt", "meg-edges.txt"]


@pytest.fixture(scope="module")
def bins():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "hostcheck"), "all"], check=True)
    return {n: os.path.join(HERE, "hostcheck", n) for n in ("estfact_check", "estfact_sched_check")}


def run(exe, cwd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    subprocess.run([exe], cwd=cwd, env=e, check=True, stderr=subprocess.DEVNULL)


@pytest.mark.parametrize("which,env", [("estfact_check", {}),
                                       ("estfact_sched_check", {"PINTRON_THREADS": "1"}),
                                       ("estfact_sched_check", {"PINTRON_THREADS": "4", "PINTRON_FIBERS": "7"}),
                                       ("estfact_sched_check", {"PINTRON_LANES": "1", "PINTRON_THREADS": "2"}),
                                       ("estfact_sched_check", {"PINTRON_LANES": "4", "PINTRON_SERVICES": "2", "PINTRON_FIBERS": "8"}),
                                       ("estfact_sched_check", {"PINTRON_ESTFACT_MODE": "direct"}),
                                       # the two short cuts of the host code switched off: one-path graphs through the
                                       # list structure, every question asked in its turn
                                       ("estfact_sched_check", {"PINTRON_CHAIN": "0", "PINTRON_THREADS": "2"}),
                                       ("estfact_sched_check", {"PINTRON_AHEAD": "0", "PINTRON_THREADS": "2"}),
                                       # MEG stage of the C-ABI: graphs beyond the device limits come back
                                       # "unavailable" and are built on the host; a library without the stage;
                                       # the stage switched off
                                       ("estfact_sched_check", {"PINTRON_FAKE_MEG_LIMIT": "6", "PINTRON_THREADS": "2"}),
                                       ("estfact_sched_check", {"PINTRON_FAKE_NO_MEG": "1", "PINTRON_THREADS": "2"}),
                                       ("estfact_sched_check", {"PINTRON_GPU_MEG": "0", "PINTRON_THREADS": "2"})])
def test_ambn_outputs_match_reference(bins, tmp_path, which, env):
    for f in ("genomic.txt", "ests.txt"):
        shutil.copy(os.path.join(GOLD, f), tmp_path)
    run(bins[which], tmp_path, env)
    for f in FILES:
        assert filecmp.cmp(os.path.join(tmp_path, f), os.path.join(GOLD, "expected-" + f), shallow=False), f


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_synthetic_sample_matches_compiled_reference(bins, tmp_path):
    from pintron_amd import synth
    w = synth.make("C2", n_est=300)
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    for d in (ref_dir, my_dir):
        synth.write_files(w, str(d))
    subprocess.run([os.path.join(O.REF_DIR, "est-fact-core")], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
    run(bins["estfact_sched_check"], my_dir, {"PINTRON_THREADS": "3"})
    for f in FILES:
        assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), f
    assert os.path.getsize(my_dir / "raw-multifasta-out.txt") > 10000


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", [12, 13, 16])   # seeds whose inputs reach every branch (occurrences found through the other-byte index too)
def test_small_exons_and_non_acgt_bytes_match_compiled_reference(bins, tmp_path, seed):
    """search_for_new_small_exons / remove_false_small_exons (src/factorization-refinement.c:641-1125) on genes that
    have small exons, with Ns in ESTs and genomic sequence and lower-case stretches: the 6-mer index, the index of
    the other bytes and the classification tables against the reference's strstr() / matrices."""
    from pintron_amd import synth
    g, e = synth.make_small_exons(seed)
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    for d in (ref_dir, my_dir):
        d.mkdir()
        (d / "genomic.txt").write_text(g)
        (d / "ests.txt").write_text(e)
    subprocess.run([os.path.join(O.REF_DIR, "est-fact-core")], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
    run(bins["estfact_sched_check"], my_dir, {"PINTRON_THREADS": "3"})
    for f in FILES:
        assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), f
    assert os.path.getsize(my_dir / "raw-multifasta-out.txt") > 10000


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_long_transcripts_match_compiled_reference(bins, tmp_path):
    """Exons of several kb (full-length mRNAs): the host logic has no size limits of its own."""
    from pintron_amd import synth
    g, e = synth.make_long_transcripts()
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    for d in (ref_dir, my_dir):
        d.mkdir()
        (d / "genomic.txt").write_text(g)
        (d / "ests.txt").write_text(e)
    subprocess.run([os.path.join(O.REF_DIR, "est-fact-core")], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
    run(bins["estfact_sched_check"], my_dir, {"PINTRON_THREADS": "2"})
    for f in FILES:
        assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), f


@pytest.mark.parametrize("name,sub", [("dist-docs/example", "example"), ("test-issue-13", "issue13")])
def test_reference_regression_inputs_md5(bins, tmp_path, name, sub):
    """The reference's own example and regression inputs (real ESTs; tests/golden/<sub>/*.gz): the
    host logic reproduces the checksums of the reference's outputs (tests/golden/reference_md5.json,
    SURVEY 8c)."""
    import gzip
    import hashlib
    import json
    gdir = os.path.join(HERE, "golden")
    gold = json.load(open(os.path.join(gdir, "reference_md5.json")))[name]
    for f in ("genomic.txt", "ests.txt"):
        (tmp_path / f).write_bytes(gzip.open(os.path.join(gdir, sub, f + ".gz")).read())
    run(bins["estfact_sched_check"], tmp_path, {"PINTRON_THREADS": "4"})
    for f, md5 in gold.items():
        assert hashlib.md5(open(tmp_path / f, "rb").read()).hexdigest() == md5, (name, f)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_edge_case_inputs_match_compiled_reference(bins, tmp_path):
    from pintron_amd import synth
    g, e = synth.make_edge_cases()
    for tag, ests in (("edge", e), ("empty", "")):
        ref_dir, my_dir = tmp_path / (tag + "_ref"), tmp_path / (tag + "_mine")
        for d in (ref_dir, my_dir):
            d.mkdir()
            (d / "genomic.txt").write_text(g)
            (d / "ests.txt").write_text(ests)
        subprocess.run([os.path.join(O.REF_DIR, "est-fact-core")], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
        run(bins["estfact_sched_check"], my_dir, {"PINTRON_THREADS": "2"})
        for f in FILES:
            assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), (tag, f)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_region_start_pairings_match_compiled_reference(bins, tmp_path):
    """Pairings with t == 0 (transcripts that begin on the first base of the region) and repeats of
    the region's first bases: the case DESIGN.md section 4b singles out."""
    from pintron_amd import synth
    g, e = synth.make_region_start_repeats()
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    for d in (ref_dir, my_dir):
        d.mkdir()
        (d / "genomic.txt").write_text(g)
        (d / "ests.txt").write_text(e)
    subprocess.run([os.path.join(O.REF_DIR, "est-fact-core")], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
    run(bins["estfact_sched_check"], my_dir, {"PINTRON_THREADS": "2"})
    for f in FILES:
        assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), f
    assert (my_dir / "raw-multifasta-out.txt").read_text().count(">/gb=T0") >= 30


def test_packed_records_file_and_c_reader(bins, tmp_path):
    """PINTRON_RECORDS_FILE: est-fact also leaves the packed factorization records; a C consumer of
    include/pintron_records.h rebuilds raw-multifasta-out.txt from them byte for byte,
    and reports a truncated blob."""
    for f in ("genomic.txt", "ests.txt"):
        shutil.copy(os.path.join(GOLD, f), tmp_path)
    run(bins["estfact_sched_check"], tmp_path, {"PINTRON_THREADS": "2", "PINTRON_RECORDS_FILE": "records.bin"})
    conv = os.path.join(HERE, "hostcheck", "records_to_text")
    out = subprocess.run([conv, "records.bin", "processed-ests.txt", "genomic.txt"], cwd=tmp_path, capture_output=True, check=True)
    assert out.stdout == (tmp_path / "raw-multifasta-out.txt").read_bytes()
    assert out.stdout == open(os.path.join(GOLD, "expected-raw-multifasta-out.txt"), "rb").read()
    blob = (tmp_path / "records.bin").read_bytes()
    from pintron_amd.estfact import parse_factorization_records
    recs = parse_factorization_records(blob)
    assert len(recs) > 0 and all(facts for _, facts in recs)
    (tmp_path / "cut.bin").write_bytes(blob[:-7])
    bad = subprocess.run([conv, "cut.bin", "processed-ests.txt", "genomic.txt"], cwd=tmp_path, capture_output=True)
    assert bad.returncode == 1


def test_cli_options_and_config_dump(bins, tmp_path):
    for f in ("genomic.txt", "ests.txt"):
        shutil.copy(os.path.join(GOLD, f), tmp_path)
    subprocess.run([bins["estfact_check"], "-l", "18", "--min-intron-length=60", "--retain-externals", "false"],
                   cwd=tmp_path, check=True, stderr=subprocess.DEVNULL)
    dump = open(os.path.join(tmp_path, "config-dump.ini")).read()
    assert 'min-factor-length="18"' in dump and 'min-intron-length="60"' in dump and 'retain-externals="false"' in dump
    bad = subprocess.run([bins["estfact_check"], "--min-string-depth-rate", "1.5"], cwd=tmp_path, stderr=subprocess.DEVNULL)
    assert bad.returncode != 0


def test_work_budget_replaces_the_timeout(bins, tmp_path):
    """The reference's wall-clock limit on the embedding enumeration (max_single_factorization_time,
    src/est-factorizations.c:626-629,706-711; retry with a longer factor, src/compute-est-fact.c:
    277-283) is a deterministic work budget here.  A budget of a few units forces the retry branch on
    ordinary ESTs: the run ends, and the sequential program, 1 and 4 scheduler threads agree byte for
    byte.  With the default budget (900 s x 10^6 units) the largest enumeration of the run stays six
    orders of magnitude below it and the output is the reference's."""
    outs = []
    for which, env in (("estfact_check", {}), ("estfact_sched_check", {"PINTRON_THREADS": "1"}),
                       ("estfact_sched_check", {"PINTRON_THREADS": "4"})):
        d = tmp_path / ("b_%d" % len(outs))
        d.mkdir()
        for f in ("genomic.txt", "ests.txt"):
            shutil.copy(os.path.join(GOLD, f), d)
        run(bins[which], d, dict(env, PINTRON_WORK_BUDGET="30"))
        outs.append([(d / f).read_bytes() for f in FILES])
    assert outs[0] == outs[1] == outs[2]
    assert outs[0][0] != open(os.path.join(GOLD, "expected-raw-multifasta-out.txt"), "rb").read()   # the retries happened
    d = tmp_path / "default"
    d.mkdir()
    for f in ("genomic.txt", "ests.txt"):
        shutil.copy(os.path.join(GOLD, f), d)
    e = dict(os.environ, ESTFACT_CHECK_DIMS="1")
    r = subprocess.run([bins["estfact_check"]], cwd=d, env=e, check=True, stderr=subprocess.PIPE, text=True)
    hw = int([ln for ln in r.stderr.splitlines() if ln.startswith("work: high water")][0].split()[3])
    assert 0 < hw < 1000
    assert filecmp.cmp(d / "raw-multifasta-out.txt", os.path.join(GOLD, "expected-raw-multifasta-out.txt"), shallow=False)


def test_c_program_shards_one_gene_over_ranks(bins, tmp_path):
    """`est-fact --gpus=N` (pintron_amd/host/ef_multi.c): the program starts the other ranks itself,
    each factorizes its range of the ESTs, rank 0 gathers the factorization records and the processed
    ESTs through pgpu_gather (ONE payload per rank, after an all-gather of status + sizes) and writes
    what a single process writes; the MEG side files are written in place by every rank.  Here over the
    CPU stand-in of the C-ABI, whose exchanges go through files; on the GPU box the same flow runs
    over RCCL."""
    from pintron_amd import synth
    w = synth.make("C2", n_est=240, seed=9)
    one, many, envd = tmp_path / "one", tmp_path / "many", tmp_path / "env"
    for d in (one, many, envd):
        synth.write_files(w, str(d))
    e = dict(os.environ, TMPDIR=str(tmp_path), PINTRON_THREADS="2")
    subprocess.run([bins["estfact_sched_check"]], cwd=one, env=e, check=True, stderr=subprocess.DEVNULL)
    subprocess.run([bins["estfact_sched_check"], "--gpus=3"], cwd=many, env=e, check=True, stderr=subprocess.DEVNULL)
    subprocess.run([bins["estfact_sched_check"]], cwd=envd, env=dict(e, PINTRON_GPUS="2"), check=True, stderr=subprocess.DEVNULL)
    for f in FILES:
        assert filecmp.cmp(one / f, many / f, shallow=False), f
        assert filecmp.cmp(one / f, envd / f, shallow=False), f
    assert os.path.getsize(one / "raw-multifasta-out.txt") > 10000
    # PINTRON_SHARD_DIAGNOSTICS=0: the two files the pipeline reads, no MEG side files
    nod = tmp_path / "nodiag"
    synth.write_files(w, str(nod))
    subprocess.run([bins["estfact_sched_check"], "--gpus=2"], cwd=nod, env=dict(e, PINTRON_SHARD_DIAGNOSTICS="0"), check=True,
                   stderr=subprocess.DEVNULL)
    for f in FILES[:2]:
        assert filecmp.cmp(one / f, nod / f, shallow=False), f
    assert not os.path.exists(nod / "megs.txt")
    assert not [p for p in os.listdir(tmp_path) if p.startswith(".pintron-comm-id")], "rendezvous files left behind"


def test_c_program_gathers_records_not_text(bins, tmp_path):
    """What travels in `est-fact --gpus=N` is [packed factorization records | processed ESTs] per rank; rank 0
    prints raw-multifasta-out.txt from them (pintron_amd/host/ef_records.c, the format of
    src/io-multifasta.c:187-229).  The gathered bytes are reported under PINTRON_VERBOSE: far fewer than the text;
    and the printing obeys --retain-externals=false like the single process (src/io-multifasta.c:204,217-222)."""
    import re
    from pintron_amd import synth
    w = synth.make("C2", n_est=300, seed=21)
    for opts, tag in (([], "dflt"), (["--retain-externals=false"], "noext")):
        one, many = tmp_path / (tag + "1"), tmp_path / (tag + "3")
        for d in (one, many):
            synth.write_files(w, str(d))
        e = dict(os.environ, TMPDIR=str(tmp_path), PINTRON_THREADS="2", PINTRON_VERBOSE="1")
        subprocess.run([bins["estfact_sched_check"]] + opts, cwd=one, env=e, check=True, stderr=subprocess.DEVNULL)
        r = subprocess.run([bins["estfact_sched_check"], "--gpus=3"] + opts, cwd=many, env=e, check=True, stderr=subprocess.PIPE, text=True)
        for f in FILES:
            assert filecmp.cmp(one / f, many / f, shallow=False), (tag, f)
        got = [int(m.group(1)) for m in re.finditer(r"rank 0 receives (\d+)", r.stderr)]
        assert len(got) == 3 and len(set(got)) == 1
        raw, pests = os.path.getsize(one / "raw-multifasta-out.txt"), os.path.getsize(one / "processed-ests.txt")
        assert raw > 10000
        assert got[0] - pests < raw // 6, "the records are a fraction of the text"
        assert got[0] < raw + pests


def test_c_program_ranks_read_only_their_part_of_the_file(bins, tmp_path):
    """A rank of a sharded run reads and parses only its byte range of ests.txt, cut at record starts by a rule
    every rank applies to the file on its own (ef_read_multifasta_part): whatever the layout -- text before the
    first record, wrapped and empty lines, the '#\\#' terminator, CRLF, records of very different lengths, more
    ranks than records -- the parts are disjoint, ordered and complete, i.e. the gathered files are those of one
    process."""
    import random
    from pintron_amd import synth
    w = synth.make("C2", n_est=60, seed=21)
    rng = random.Random(5)
    recs = []
    for k, (h, sq) in enumerate(zip(w.est_headers, w.est_seqs)):
        sq = sq.decode() if isinstance(sq, bytes) else sq
        h = h.decode() if isinstance(h, bytes) else h
        if k % 7 == 3:
            sq = sq[:rng.randrange(30, 60)]                           # a short record between long ones
        width = rng.choice([60, 70, 200, 10 ** 6])
        lines = [sq[i:i + width] for i in range(0, len(sq), width)]
        if k % 5 == 1:
            lines.insert(1, "")                                       # an empty line inside a record
        eol = "\r\n" if k % 4 == 2 else "\n"
        recs.append(">" + h.lstrip(">") + eol + eol.join(lines) + eol)
    layouts = {
        "plain": "".join(recs),
        "junk_first": "this line precedes the first record\n\n" + "".join(recs),
        "few": "".join(recs[:3]),                                     # fewer records than ranks
        "no_final_newline": "".join(recs).rstrip("\r\n"),
    }
    e = dict(os.environ, TMPDIR=str(tmp_path), PINTRON_THREADS="2")
    for tag, text in layouts.items():
        one = tmp_path / (tag + "_one")
        one.mkdir()
        (one / "genomic.txt").write_text(w.genomic_fasta())
        (one / "ests.txt").write_text(text)
        subprocess.run([bins["estfact_sched_check"]], cwd=one, env=e, check=True, stderr=subprocess.DEVNULL)
        for world in (2, 5):
            many = tmp_path / ("%s_%d" % (tag, world))
            many.mkdir()
            (many / "genomic.txt").write_text(w.genomic_fasta())
            (many / "ests.txt").write_text(text)
            subprocess.run([bins["estfact_sched_check"], "--gpus=%d" % world], cwd=many, env=e, check=True, stderr=subprocess.DEVNULL,
                           timeout=120)
            for f in FILES:
                assert filecmp.cmp(one / f, many / f, shallow=False), (tag, world, f)
    assert os.path.getsize(tmp_path / "plain_one" / "raw-multifasta-out.txt") > 5000


@pytest.mark.parametrize("fault", ["1:open", "2:step", "1:abort", "0:step", "0:open"])
def test_c_program_a_failing_rank_ends_all_ranks(bins, tmp_path, fault):
    """A rank that cannot open its session, whose step fails, or that dies outright (abort) must not
    leave its peers blocked in an exchange: every rank ends, est-fact exits non-zero, within seconds
    (health markers before the communicator, status word in the first exchange, the parent's
    watchdog; pintron_amd/host/ef_multi.c)."""
    import time
    from pintron_amd import synth
    synth.write_files(synth.make("C2", n_est=90, seed=9), str(tmp_path))
    e = dict(os.environ, TMPDIR=str(tmp_path), PINTRON_THREADS="2", PINTRON_FAULT_INJECT=fault)
    t0 = time.time()
    r = subprocess.run([bins["estfact_sched_check"], "--gpus=3"], cwd=tmp_path, env=e, stderr=subprocess.PIPE, text=True, timeout=60)
    took = time.time() - t0
    assert r.returncode != 0 and took < 10, (r.returncode, took, r.stderr[-1500:])
    time.sleep(0.3)
    left = subprocess.run(["pgrep", "-c", "-f", str(tmp_path)], capture_output=True, text=True).stdout.strip()
    assert left in ("", "0"), "ranks left running: " + left


def test_c_program_runs_many_genes(bins, tmp_path):
    """`est-fact --genes=FILE --gpus=N`: gene g on rank g mod N, every gene's files in its own
    directory (BASELINE.json configs[3] in miniature)."""
    from pintron_amd import synth
    dirs = []
    for g in range(3):
        w = synth.make("C4", n_est=60, seed=synth.CONFIGS["C4"]["seed"] + g, gen_len=40_000)
        for tag in ("solo", "multi"):
            synth.write_files(w, str(tmp_path / ("%s%d" % (tag, g))))
        dirs.append(str(tmp_path / ("multi%d" % g)))
        subprocess.run([bins["estfact_sched_check"]], cwd=tmp_path / ("solo%d" % g), check=True, stderr=subprocess.DEVNULL,
                       env=dict(os.environ, PINTRON_THREADS="2"))
    (tmp_path / "genes.txt").write_text("# three genes\n" + "\n".join(dirs) + "\n")
    subprocess.run([bins["estfact_sched_check"], "--genes=" + str(tmp_path / "genes.txt"), "--gpus=2"], cwd=tmp_path, check=True,
                   stderr=subprocess.DEVNULL, env=dict(os.environ, PINTRON_THREADS="2", TMPDIR=str(tmp_path)))
    for g in range(3):
        for f in FILES:
            assert filecmp.cmp(tmp_path / ("solo%d" % g) / f, tmp_path / ("multi%d" % g) / f, shallow=False), (g, f)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_split_parse_of_ests_file(bins, tmp_path):
    """ests.txt is cut at record starts and parsed on several threads (files of 8 MB and more; forced
    here): the records and their order are those of the sequential reader, on the edge-case inputs
    (wrapped lines, odd headers, the '#\\#' terminator) and on a C2 sample."""
    from pintron_amd import synth
    g, e = synth.make_edge_cases()
    w = synth.make("C2", n_est=200, seed=4)
    for tag, gfa, efa in (("edge", g, e), ("c2", w.genomic_fasta(), w.ests_fasta())):
        ref_dir, my_dir = tmp_path / (tag + "_ref"), tmp_path / (tag + "_mine")
        for d in (ref_dir, my_dir):
            d.mkdir()
            (d / "genomic.txt").write_text(gfa)
            (d / "ests.txt").write_text(efa)
        subprocess.run([os.path.join(O.REF_DIR, "est-fact-core")], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
        run(bins["estfact_sched_check"], my_dir, {"PINTRON_THREADS": "2", "PINTRON_PARSE_SPLIT": "1"})
        for f in FILES:
            assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), (tag, f)


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
def test_region_start_copies_match_compiled_reference(bins, tmp_path):
    """ESTs for which the reference holds the t == 0 pairing more than once (synth.make_region_start_copies):
    the repeated vertex through MEG construction, simplification, factorization and the writers,
    sequential program and scheduler (device-style MEG records included), against the reference."""
    from pintron_amd import synth
    g, e = synth.make_region_start_copies()
    ref_dir = tmp_path / "ref"
    ref_dir.mkdir()
    (ref_dir / "genomic.txt").write_text(g)
    (ref_dir / "ests.txt").write_text(e)
    subprocess.run([os.path.join(O.REF_DIR, "est-fact-core")], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
    megs = (ref_dir / "megs.txt").read_text()
    blocks = [b.split("#adj#")[0].splitlines()[2:] for b in megs.split("***********")[1:]]
    assert sum(1 for b in blocks if len(set(b)) != len(b)) >= 4, "the generator no longer produces repeated vertices"
    for which, env in (("estfact_check", {}), ("estfact_sched_check", {"PINTRON_THREADS": "2"}),
                       ("estfact_sched_check", {"PINTRON_THREADS": "2", "PINTRON_GPU_MEG": "0"})):
        my_dir = tmp_path / ("mine_%s_%d" % (which, len(env)))
        my_dir.mkdir()
        (my_dir / "genomic.txt").write_text(g)
        (my_dir / "ests.txt").write_text(e)
        run(bins[which], my_dir, env)
        for f in FILES:
            assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), (which, f)


def test_closing_stderr_lines_have_the_reference_format(bins, tmp_path):
    """The reference ends with its five timers, "End" and the resource usage on stderr
    (src/main-est-fact.c:321-335, src/util.c:184-208; include/log.h gives the line format), and the
    pipeline driver appends that to its log.  Same lines, same format, here."""
    import re
    for f in ("genomic.txt", "ests.txt"):
        shutil.copy(os.path.join(GOLD, f), tmp_path)
    r = subprocess.run([bins["estfact_sched_check"]], cwd=tmp_path, env=dict(os.environ, PINTRON_THREADS="2"),
                       stderr=subprocess.PIPE, text=True, check=True)
    tail = r.stderr.splitlines()[-9:]
    names = ["Suffix Tree", "Algorithm", "Compositions", "IO", "Total"]
    for k, nm in enumerate(names):
        assert re.fullmatch(r"\* INFO \(main            @ src/main-est-fact.c:%d \) @Timer %-22s\. Time elapsed: +\d+ microsec  " % (321 + k, nm),
                            tail[k]), tail[k]
        assert len(tail[k].split("Time elapsed: ")[1]) == len("%15d microsec  " % 0)
    assert tail[5] == "* INFO (main            @ src/main-est-fact.c:335 ) End  "
    assert re.fullmatch(r"\* INFO \(resource_usage\.\.@          src/util.c:187 \) User time:   +\d+s +\d+microsec\.  ", tail[6]), tail[6]
    assert re.fullmatch(r"\* INFO \(resource_usage\.\.@          src/util.c:188 \) System time: +\d+s +\d+microsec\.  ", tail[7]), tail[7]
    assert re.fullmatch(r"\* INFO \(resource_usage\.\.@          src/util.c:199 \) Mem\. used: +\d+KB  ", tail[8]), tail[8]


def test_dp_trace_and_replay(bins, tmp_path):
    """PINTRON_DP_TRACE records every answered DP request of a run; tools/replay_dp_trace.py puts them through
    the oracle.  Here the answers come from the oracle itself (CPU stand-in of the C-ABI), so none may differ --
    and a record corrupted on purpose must be reported.  (On the GPU box the same pair splits a wrong output into
    "a kernel answered wrongly" and "the host logic went wrong": tools/stress_parity.py.)"""
    import struct
    import sys
    for f in ("genomic.txt", "ests.txt"):
        shutil.copy(os.path.join(GOLD, f), tmp_path)
    trace = tmp_path / "dp.trace"
    run(bins["estfact_sched_check"], tmp_path, {"PINTRON_THREADS": "3", "PINTRON_DP_TRACE": str(trace)})
    tool = os.path.join(os.path.dirname(HERE), "tools", "replay_dp_trace.py")
    ok = subprocess.run([sys.executable, tool, str(trace)], capture_output=True, text=True)
    assert ok.returncode == 0 and "; 0 differ from the oracle" in ok.stdout, ok.stdout[-500:]
    n = int(ok.stdout.split()[0])
    assert n > 1500
    data = bytearray(trace.read_bytes())
    v0 = struct.unpack_from("<i", data, 36)[0]
    struct.pack_into("<i", data, 36, v0 + 1)                 # first value of the first record
    bad = tmp_path / "bad.trace"
    bad.write_bytes(bytes(data))
    r = subprocess.run([sys.executable, tool, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "; 1 differ from the oracle" in r.stdout, r.stdout[-500:]
