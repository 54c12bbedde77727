"""GPU: the product est-fact binary (pintron_amd/bin/est-fact: C host + libpintron_gpu.so, fibre
scheduler, no CPU fallback) against the reference's outputs."""
import filecmp
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden", "ambn")
EXE = os.path.join(ROOT, "pintron_amd", "bin", "est-fact")
FILES = ["raw-multifasta-out.txt", "processed-ests.txt", "megs.txt", "processed-megs.txt", "meg-edges.txt"]


@pytest.fixture(scope="module")
def exe():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(EXE)
    return EXE


@pytest.mark.parametrize("env", [{}, {"PINTRON_ESTFACT_MODE": "direct"}, {"PINTRON_THREADS": "2", "PINTRON_FIBERS": "5"},
                                 {"PINTRON_LANES": "1"}, {"PINTRON_LANES": "4", "PINTRON_SERVICES": "2", "PINTRON_FIBERS": "8"},
                                 {"PINTRON_NO_PREFETCH": "1", "PINTRON_THREADS": "3"},
                                 {"PINTRON_GPU_MEG": "0"},
                                 {"PGPU_MERGED": "1"}, {"PGPU_MERGED": "0"}, {"PGPU_MERGED": "0", "PGPU_FANOUT": "0"},
                                 {"PGPU_WAIT": "0"}, {"PINTRON_NO_FIBER_POOL": "1", "PINTRON_NO_FIBER_PREFETCH": "1"},
                                 # round 4's short cuts switched off one by one: questions in their turn, one-path graphs
                                 # through the lists, alignments always over the whole matrix, an N prefix by the matrix kernel
                                 {"PINTRON_AHEAD": "0"}, {"PINTRON_CHAIN": "0"}, {"PGPU_ALIGN_BAND": "0"}, {"PGPU_LCF_SA_N": "0"},
                                 {"PINTRON_KEEP": "0", "PINTRON_PRE_RAMP": "1:2:3"},
                                 # ... and the end-exon alignments that answer the trimmed exon's check too
                                 {"PINTRON_ENDPOINT_CHECKS": "0"}, {"PINTRON_ENDPOINT_CHECKS": "0", "PGPU_MERGED": "1"}])
def test_ambn_golden(exe, tmp_path, env):
    for f in ("genomic.txt", "ests.txt"):
        shutil.copy(os.path.join(GOLD, f), tmp_path)
    e = dict(os.environ)
    e.update(env)
    subprocess.run([exe], cwd=tmp_path, env=e, check=True)
    for f in FILES:
        assert filecmp.cmp(os.path.join(tmp_path, f), os.path.join(GOLD, "expected-" + f), shallow=False), f


@pytest.mark.parametrize("env", [{}, {"PGPU_MERGED": "1"}, {"PGPU_MERGED": "0"},
                                 {"PINTRON_AHEAD": "0", "PINTRON_CHAIN": "0", "PGPU_ALIGN_BAND": "0", "PGPU_LCF_SA_N": "0"},
                                 {"PINTRON_ENDPOINT_CHECKS": "0"}])
def test_c3_sample_vs_compiled_reference(exe, tmp_path, env):
    """2 000 C3-shaped ESTs (200 kb genomic, 3 % errors): byte-identical to the reference binary
    (oracle/_ref/est-fact-core travels with the repository snapshot), in every launch mode of the library
    (one batch launch + LCF; wave-per-job launch + sweeps; a launch per family)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "est-fact-core")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/est-fact-core not present")
    from pintron_amd import synth
    w = synth.make("C3", n_est=2000)
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    for d in (ref_dir, my_dir):
        synth.write_files(w, str(d))
    subprocess.run([ref], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
    subprocess.run([exe], cwd=my_dir, check=True, env=dict(os.environ, **env))
    for f in FILES:
        assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), f


@pytest.mark.parametrize("cfg,n_est", [("C2", 1000), ("C5", 3000)])
def test_other_configs_vs_compiled_reference(exe, tmp_path, cfg, n_est):
    """BASELINE.json's other shapes as parity cases: C2 (50 kb x 1 000 ESTs ~500 bp) in full, C5
    (1 Mb genomic, 150 bp reads) on a 3 000-read sample."""
    ref = os.path.join(ROOT, "oracle", "_ref", "est-fact-core")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/est-fact-core not present")
    from pintron_amd import synth
    w = synth.make(cfg, n_est=n_est)
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    for d in (ref_dir, my_dir):
        synth.write_files(w, str(d))
    subprocess.run([ref], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
    subprocess.run([exe], cwd=my_dir, check=True)
    for f in FILES:
        assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), f


@pytest.mark.parametrize("name,sub", [("test-issue-13", "issue13"), ("dist-docs/example", "example")])
def test_real_ests_of_the_reference_regression_sets(exe, tmp_path, name, sub):
    """Real ESTs: the inputs of the reference's regressionTest/test-issue-13 and dist-docs/example
    (tests/golden/<sub>/*.gz, data files of the reference's own tests).  The checksums of the two
    main outputs are the reference's (tests/golden/reference_md5.json); when the compiled reference
    travelled with the snapshot, all five files are compared with its run as well."""
    import gzip
    import hashlib
    import json
    gold = json.load(open(os.path.join(HERE, "golden", "reference_md5.json")))[name]
    my_dir, ref_dir = tmp_path / "mine", tmp_path / "ref"
    for d in (my_dir, ref_dir):
        d.mkdir()
        for f in ("genomic.txt", "ests.txt"):
            (d / f).write_bytes(gzip.open(os.path.join(HERE, "golden", sub, f + ".gz")).read())
    subprocess.run([exe], cwd=my_dir, check=True)
    for f, md5 in gold.items():
        assert hashlib.md5((my_dir / f).read_bytes()).hexdigest() == md5, (name, f)
    ref = os.path.join(ROOT, "oracle", "_ref", "est-fact-core")
    if os.path.exists(ref):
        subprocess.run([ref], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
        for f in FILES:
            assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), (name, f)


def test_edge_case_inputs_vs_compiled_reference(exe, tmp_path):
    """N tails, negative-strand header, too short / all-N / lower-case / polyA-only ESTs, duplicated
    and odd headers, /fixed_strand, wrapped lines (pintron_amd/synth.py: make_edge_cases); and an
    empty ests.txt."""
    ref = os.path.join(ROOT, "oracle", "_ref", "est-fact-core")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/est-fact-core not present")
    from pintron_amd import synth
    g, e = synth.make_edge_cases()
    for tag, ests in (("edge", e), ("empty", "")):
        ref_dir, my_dir = tmp_path / (tag + "_ref"), tmp_path / (tag + "_mine")
        for d in (ref_dir, my_dir):
            d.mkdir()
            (d / "genomic.txt").write_text(g)
            (d / "ests.txt").write_text(ests)
        subprocess.run([ref], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
        subprocess.run([exe], cwd=my_dir, check=True)
        for f in FILES:
            assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), (tag, f)
    assert (tmp_path / "edge_mine" / "raw-multifasta-out.txt").read_text().count(">") == 10


def test_region_start_pairings_vs_compiled_reference(exe, tmp_path):
    """Transcripts that begin on the first base of the genomic region (t == 0 pairings) and repeats
    of the region's first bases through the GPU pairing kernels."""
    ref = os.path.join(ROOT, "oracle", "_ref", "est-fact-core")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/est-fact-core not present")
    from pintron_amd import synth
    g, e = synth.make_region_start_repeats()
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    for d in (ref_dir, my_dir):
        d.mkdir()
        (d / "genomic.txt").write_text(g)
        (d / "ests.txt").write_text(e)
    subprocess.run([ref], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
    subprocess.run([exe], cwd=my_dir, check=True)
    for f in FILES:
        assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), f
    assert (my_dir / "raw-multifasta-out.txt").read_text().count(">/gb=T0") >= 30


def test_region_start_copies_vs_compiled_reference(exe, tmp_path):
    """ESTs whose t == 0 pairing the reference lists more than once (synth.make_region_start_copies):
    the GPU pairing kernels must produce the copies and the GPU MEG stage must carry the repeated
    vertex exactly as the reference's lists do."""
    ref = os.path.join(ROOT, "oracle", "_ref", "est-fact-core")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/est-fact-core not present")
    from pintron_amd import synth
    g, e = synth.make_region_start_copies()
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    for d in (ref_dir, my_dir):
        d.mkdir()
        (d / "genomic.txt").write_text(g)
        (d / "ests.txt").write_text(e)
    subprocess.run([ref], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
    subprocess.run([exe], cwd=my_dir, check=True)
    for f in FILES:
        assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), f


def test_long_transcripts_vs_compiled_reference(exe, tmp_path):
    """Full-length transcripts with a 5.6 kb exon: alignments, K-band distances and affix searches
    with more than 4096 rows (strips) inside the whole program."""
    ref = os.path.join(ROOT, "oracle", "_ref", "est-fact-core")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/est-fact-core not present")
    from pintron_amd import synth
    g, e = synth.make_long_transcripts()
    ref_dir, my_dir = tmp_path / "ref", tmp_path / "mine"
    for d in (ref_dir, my_dir):
        d.mkdir()
        (d / "genomic.txt").write_text(g)
        (d / "ests.txt").write_text(e)
    subprocess.run([ref], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
    subprocess.run([exe], cwd=my_dir, check=True)
    for f in FILES:
        assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), f
    assert os.path.getsize(my_dir / "raw-multifasta-out.txt") > 50000


def test_session_steps_are_idempotent(exe, tmp_path):
    """The bench's step: two passes over the resident batch give the same text, equal to the files
    the binary writes."""
    from pintron_amd import estfact, synth
    w = synth.make("C3", n_est=600, seed=11)
    synth.write_files(w, str(tmp_path))
    L = estfact.load_host_lib()
    s = estfact.Session(L, str(tmp_path))
    s.step()
    first = [s.output(k) for k in (0, 1, 2, 3, 5)]
    st = s.step()
    second = [s.output(k) for k in (0, 1, 2, 3, 5)]
    s.close()
    assert first == second and st.units == s_units(w)
    subprocess.run([exe], cwd=tmp_path, check=True)
    for text, f in zip(first, FILES):
        assert text == open(tmp_path / f, "rb").read(), f


RCCL_WORKER = r"""
import os, sys
import torch, torch.distributed as dist          # torch (and its bundled HIP runtime) first, as in bench.py
sys.path.insert(0, %(root)r)
from pintron_amd import estfact, synth
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29655")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
w = synth.make("C3", n_est=300, seed=12)
synth.write_files(w, %(tmp)r)
s = estfact.Session(estfact.load_host_lib(), %(tmp)r)
s.step()
parts = estfact.gather_tensor(s.output_tensor(0), dist, 0, 1, "cuda")
assert len(parts) == 1 and bytes(parts[0].cpu().numpy().tobytes()) == s.records() and len(s.records()) > 10000
assert estfact.gather_bytes(b"", dist, 0, 1, "cuda") == [b""]
s.close()
dist.destroy_process_group()
print("rccl gather ok")
"""


def test_record_gather_over_rccl_single_rank(exe, tmp_path):
    """The exchange bench.py performs for N > 1 (all_gather of sizes + gather of the padded records)
    on a 1-rank RCCL group: API usage and payload integrity on the real backend.  Own process:
    torch brings its own HIP runtime, which has to be the first one loaded (as in bench.py)."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER % dict(root=ROOT, tmp=str(tmp_path)))
    out = subprocess.run([os.sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "rccl gather ok" in out.stdout, out.stderr[-2000:]


def s_units(w):
    return len(w.est_seqs)


def test_bench_flow_with_two_ranks_on_one_gpu(exe, tmp_path):
    """bench.py's N > 1 flow (barrier, per-step gather of the packed records to rank 0, max over
    ranks) with two ranks sharing this box's GPU; the exchange runs over gloo because two RCCL
    ranks cannot sit on one device (PINTRON_DIST_BACKEND, as in pintron_amd.multi)."""
    import json
    import sys
    env = dict(os.environ, PINTRON_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "1", "--warmup", "1", "--ests", "6000", "--no-cpu"]
    out = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "est-shard x2" and d["scaling"] == "weak"
    assert d["config"]["ests_per_gpu"] == 6000 and d["value"] > 0


def test_bench_launches_its_own_ranks(exe, tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two rank processes
    itself (children, never an exec) and relays rank 0's line.  Two ranks share this box's one GPU,
    so the exchange runs over gloo; on a node with >= 2 GPUs the same command uses RCCL."""
    import json
    import sys
    import torch
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env["PINTRON_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--ests", "3000", "--no-cpu"]
    out = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ests_per_gpu"] == 3000 and d["value"] > 0
    # a launcher that disagrees with --gpus is an error, not a silent 1-rank run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu"],
                         cwd=tmp_path, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in bad.stderr


def test_c4_two_gene_sample_vs_compiled_reference(exe, tmp_path):
    """BASELINE.json configs[3] (8 genes x 200 kb, ESTs sharded gene-wise over the GPUs) on a
    2-gene x 1 500-EST sample: every gene's files against the reference object code, and
    `bench.py --workload C4` over the same genes (one rank, both genes resident)."""
    import json
    import sys
    ref = os.path.join(ROOT, "oracle", "_ref", "est-fact-core")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/est-fact-core not present")
    from pintron_amd import synth
    for g in range(2):
        w = synth.make("C4", n_est=1500, seed=synth.CONFIGS["C4"]["seed"] + g)
        ref_dir, my_dir = tmp_path / ("ref%d" % g), tmp_path / ("mine%d" % g)
        for d in (ref_dir, my_dir):
            synth.write_files(w, str(d))
        subprocess.run([ref], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
        subprocess.run([exe], cwd=my_dir, check=True)
        for f in FILES:
            assert filecmp.cmp(my_dir / f, ref_dir / f, shallow=False), (g, f)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "C4", "--ests", "1500",
                          "--steps", "1", "--warmup", "0"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["scaling"] == "strong" and d["config"]["ests_per_gpu"] == 8 * 1500 and d["n_gpus"] == 1


GATHER_WORKER = r"""
import ctypes as C, sys
sys.path.insert(0, %(root)r)
import pintron_amd.capi as capi
ctx = capi.Context(0)
L = ctx.L
ident = (C.c_char * 128)()
ctx.check(L.pgpu_comm_unique_id(ctx.h, ident))
comm = C.c_void_p()
ctx.check(L.pgpu_comm_init(ctx.h, 0, 1, ident, C.byref(comm)))
payload = bytes(range(256)) * 4000 + b"tail"
recv = C.create_string_buffer(len(payload))
counts = (C.c_uint64 * 1)()
for data in (payload, b"", payload[:17]):
    ctx.check(L.pgpu_gather(ctx.h, comm, data, len(data), recv, len(payload), counts))
    assert counts[0] == len(data) and recv.raw[:len(data)] == data
assert L.pgpu_gather(ctx.h, comm, payload, len(payload), recv, 10, counts) == -28      # PGPU_ENOSPC
ctx.check(L.pgpu_comm_destroy(ctx.h, comm))
ctx.close()
print("gather ok")
"""


def test_gather_entry_point_on_rccl(exe, tmp_path):
    """pgpu_comm_* / pgpu_gather on the real backend with a communicator of one rank (this box has one
    GPU): RCCL is loaded on demand, the sizes round and the payload round come back intact.  Own
    process, without torch: the library loads the system's RCCL, and a process that already carries
    torch's bundled HIP runtime and RCCL is not the place to map a second pair (the C program that
    uses this entry point never has torch in it)."""
    script = tmp_path / "gather_worker.py"
    script.write_text(GATHER_WORKER % dict(root=ROOT))
    out = subprocess.run([os.sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "gather ok" in out.stdout, out.stderr[-2000:]


def test_more_ranks_than_gpus_is_refused_before_anything_starts(exe, tmp_path):
    """`est-fact --gpus=8` on a box with fewer GPUs: refused by the parent before a rank is started or the GPU runtime
    is touched (pintron_amd/host/ef_multi.c: visible_gpu_count), with a message that says why; nothing is written."""
    import time
    import torch
    have = torch.cuda.device_count()
    if have >= 8:
        pytest.skip("this box has 8 GPUs")
    from pintron_amd import synth
    synth.write_files(synth.make("C2", n_est=50), str(tmp_path))
    t0 = time.time()
    r = subprocess.run([exe, "--gpus=8"], cwd=tmp_path, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode != 0 and time.time() - t0 < 5
    assert "--gpus=8" in r.stderr and "visible" in r.stderr
    assert not os.path.exists(tmp_path / "raw-multifasta-out.txt")


def test_c_program_many_genes_and_shards(exe, tmp_path):
    """`est-fact --genes=FILE` (gene g on rank g mod N) on the GPU, and -- on a node with at least two
    GPUs -- `est-fact --gpus=2`: the C program shards one gene over two ranks and rank 0 writes the
    files of a single process (RCCL gather, pintron_amd/host/ef_multi.c)."""
    import torch
    from pintron_amd import synth
    dirs = []
    for g in range(2):
        w = synth.make("C4", n_est=800, seed=synth.CONFIGS["C4"]["seed"] + g)
        for tag in ("solo", "multi"):
            synth.write_files(w, str(tmp_path / ("%s%d" % (tag, g))))
        dirs.append(str(tmp_path / ("multi%d" % g)))
        subprocess.run([exe], cwd=tmp_path / ("solo%d" % g), check=True)
    (tmp_path / "genes.txt").write_text("\n".join(dirs) + "\n")
    n = min(torch.cuda.device_count(), 2)
    subprocess.run([exe, "--genes=" + str(tmp_path / "genes.txt"), "--gpus=%d" % n], cwd=tmp_path, check=True)
    for g in range(2):
        for f in FILES:
            assert filecmp.cmp(tmp_path / ("solo%d" % g) / f, tmp_path / ("multi%d" % g) / f, shallow=False), (g, f)
    if n >= 2:
        shard = tmp_path / "shard"
        synth.write_files(synth.make("C4", n_est=800, seed=synth.CONFIGS["C4"]["seed"]), str(shard))
        subprocess.run([exe, "--gpus=2"], cwd=shard, check=True)
        for f in FILES:
            assert filecmp.cmp(shard / f, tmp_path / "solo0" / f, shallow=False), f
