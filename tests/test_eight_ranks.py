"""8-rank readiness without multi-GPU hardware (CPU; the check build of the host code over the CPU stand-in of the
C-ABI, exchanges over files / gloo):
  * `est-fact --gpus=8` on one gene, also with fewer ESTs than ranks (some ranks get nothing);
  * bench.py's C4 flow -- 8 genes over 3, 5 and 8 ranks and 2 genes over 3 (a rank without a gene), ONE gather of
    the packed records per step whatever the number of genes on a rank;
  * the worker-count rule under a CPU quota shared by the ranks of a node."""
import filecmp
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HC = os.path.join(HERE, "hostcheck")
FILES = ["raw-multifasta-out.txt", "processed-ests.txt", "megs.txt", "processed-megs.txt", "meg-edges.txt"]


@pytest.fixture(scope="module")
def check_bin():
    subprocess.run(["make", "-s", "-C", HC, "all"], check=True)
    return os.path.join(HC, "estfact_sched_check")


@pytest.mark.parametrize("n_est", [5, 37])
def test_one_gene_over_eight_ranks(check_bin, tmp_path, n_est):
    """ranks = 8 > ESTs = 5: three ranks factorize nothing and still take part in every exchange"""
    from pintron_amd import synth
    w = synth.make("C2", n_est=n_est, seed=31)
    one, many = tmp_path / "one", tmp_path / "many"
    for d in (one, many):
        synth.write_files(w, str(d))
    e = dict(os.environ, TMPDIR=str(tmp_path), PINTRON_THREADS="1", PINTRON_VERBOSE="1")
    subprocess.run([check_bin], cwd=one, env=e, check=True, stderr=subprocess.DEVNULL)
    r = subprocess.run([check_bin, "--gpus=8"], cwd=many, env=e, check=True, stderr=subprocess.PIPE, text=True, timeout=300)
    for f in FILES:
        assert filecmp.cmp(one / f, many / f, shallow=False), f
    ranks = [ln for ln in r.stderr.splitlines() if ln.startswith("* rank ") and " ESTs (" in ln]
    assert len(ranks) == 8
    assert sum(int(ln.split(": ")[1].split()[0]) for ln in ranks) == n_est


def run_bench(world, genes, tmp_path, port):
    subprocess.run(["make", "-s", "-C", HC, "libestfact_check.so"], check=True)
    env = dict(os.environ, PINTRON_THREADS="1", PINTRON_FIBERS="16", MASTER_ADDR="127.0.0.1", PINTRON_DIST_BACKEND="gloo",
               PINTRON_ESTFACT_LIB=os.path.join(HC, "libestfact_check.so"), PYTHONPATH=ROOT, OMP_NUM_THREADS="1", TMPDIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--workload", "C4", "--genes", str(genes), "--ests", "12", "--steps", "1", "--warmup", "0",
           "--no-cpu", "--no-oneshot"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(line) == 1, "rank 0 prints ONE line"
    return json.loads(line[0])


@pytest.mark.parametrize("world,genes,port", [(3, 8, 29731), (5, 8, 29732), (8, 8, 29733), (3, 2, 29734)])
def test_bench_c4_genes_over_ranks(tmp_path, world, genes, port):
    d = run_bench(world, genes, tmp_path, port)
    assert d["n_gpus"] == world and d["scaling"] == "strong"
    assert d["config"]["parallelism"] == "est-shard x%d" % world
    g = d["gathered"]
    assert len(g["bytes_per_rank"]) == world and g["equals_aligned"]
    # every gene was factorized exactly once: the ESTs of all ranks add up, and a rank without a gene sent nothing
    assert d["input_ests_per_s"] > 0 and d["value"] > 0
    if genes < world:
        assert g["bytes_per_rank"].count(0) >= world - genes
    else:
        assert all(b > 0 for b in g["bytes_per_rank"])


def test_worker_count_follows_the_quota_shared_by_the_ranks(check_bin, tmp_path):
    """The product's worker-count rule (ef_sched.c: host_core_share): cores by affinity and cgroup quota, divided by
    the ranks of the node (LOCAL_WORLD_SIZE), at most 16; an eighth more workers than that (default_workers: a worker
    is off the CPU a fifth of its time) and three service threads per eight cores, at most six.  The check program
    prints what it took (PINTRON_VERBOSE); here the quota is this container's own."""
    from pintron_amd import synth
    synth.write_files(synth.make("C2", n_est=40, seed=3), str(tmp_path))
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    cores = bench.usable_cores(cap=10 ** 6)
    for ranks in (1, 2, 8):
        e = dict(os.environ, PINTRON_VERBOSE="1", LOCAL_WORLD_SIZE=str(ranks))
        e.pop("PINTRON_THREADS", None)
        r = subprocess.run([check_bin], cwd=tmp_path, env=e, check=True, stderr=subprocess.PIPE, text=True)
        line = [ln for ln in r.stderr.splitlines() if ln.startswith("est-fact:")][0]
        threads = int(line.split(" threads")[0].split()[-1])
        want = max(1, min(16, cores // ranks if ranks > 1 else cores))
        assert threads == min(want + want // 8, 40), (ranks, line)      # (never more workers than ESTs)
        svc = [ln for ln in r.stderr.splitlines() if ln.startswith("* service ")]
        assert len(svc) == max(1, min(6, want * 3 // 8)), (ranks, r.stderr[-800:])
