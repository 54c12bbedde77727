"""The N > 1 path on CPU: two gloo ranks run the multi-GPU launcher (python -m pintron_amd.multi ->
pintron_amd/estfact.py: run_sharded) over the check build of the host library (tests/hostcheck: scheduler + host C over a CPU stand-in
of the C-ABI) and rank 0 must end up with exactly the files a single rank writes."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HC = os.path.join(ROOT, "tests", "hostcheck")
AMBN = os.path.join(ROOT, "tests", "golden", "ambn")

def test_partition_is_contiguous_and_balanced():
    sys.path.insert(0, ROOT)
    from pintron_amd.estfact import partition
    w = [600] * 1000 + [100] * 1000
    for world in (1, 2, 3, 8):
        parts = partition(w, world)
        assert parts[0][0] == 0 and parts[-1][1] == len(w)
        assert all(parts[r][1] == parts[r + 1][0] for r in range(world - 1))
        loads = [sum(w[a:b]) for a, b in parts]
        assert max(loads) - min(loads) <= 2 * max(w)
    assert partition([], 2) == [(0, 0), (0, 0)]
    assert partition([5], 4) == [(0, 1), (1, 1), (1, 1), (1, 1)]


def test_two_ranks_equal_one_rank(tmp_path):
    subprocess.run(["make", "-s", "-C", HC, "libestfact_check.so"], check=True)
    lib = os.path.join(HC, "libestfact_check.so")
    outs = {}
    for world in (1, 2):
        data = tmp_path / ("data%d" % world)
        data.mkdir()
        shutil.copy(os.path.join(AMBN, "genomic.txt"), data / "genomic.txt")
        shutil.copy(os.path.join(AMBN, "ests.txt"), data / "ests.txt")
        env = dict(os.environ, PINTRON_THREADS="2", PINTRON_FIBERS="16", MASTER_ADDR="127.0.0.1",
                   PINTRON_DIST_BACKEND="gloo", PINTRON_ESTFACT_LIB=lib, PYTHONPATH=ROOT)
        subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29611 + world),
                        "-m", "pintron_amd.multi", str(data)],
                       check=True, env=env, timeout=600)
        outs[world] = {f: (data / f).read_bytes() for f in
                       ("raw-multifasta-out.txt", "processed-ests.txt", "megs.txt", "processed-megs.txt", "meg-edges.txt")}
    for f, text in outs[1].items():
        assert text, f
        assert outs[2][f] == text, f
    # and both equal the reference's own output for this input (golden fixture)
    exp = open(os.path.join(AMBN, "expected-raw-multifasta-out.txt"), "rb").read()
    assert outs[2]["raw-multifasta-out.txt"] == exp


def test_packed_records_carry_the_whole_output(tmp_path):
    """the packed factorization records (16 B per exon + 4 B per factorization) plus the
    sequences reproduce raw-multifasta-out.txt byte for byte (check build of the host library)."""
    sys.path.insert(0, ROOT)
    from pintron_amd import estfact
    subprocess.run(["make", "-s", "-C", HC, "libestfact_check.so"], check=True)
    for f in ("genomic.txt", "ests.txt"):
        shutil.copy(os.path.join(AMBN, f), tmp_path)
    env_threads = os.environ.get("PINTRON_THREADS")
    os.environ["PINTRON_THREADS"] = "2"
    try:
        s = estfact.Session(estfact.load_host_lib(os.path.join(HC, "libestfact_check.so")), str(tmp_path))
        s.step()
        raw, pests, packed = s.output(0), s.output(1), s.output(estfact.RECORDS)
        s.close()
    finally:
        if env_threads is None:
            del os.environ["PINTRON_THREADS"]
        else:
            os.environ["PINTRON_THREADS"] = env_threads
    recs = estfact.parse_factorization_records(packed)
    assert len(recs) == pests.count(b">") == 25
    gen = b"".join(open(os.path.join(AMBN, "genomic.txt"), "rb").read().split(b"\n")[1:])
    assert estfact.format_raw_multifasta(recs, pests, gen) == raw
    assert raw == open(os.path.join(AMBN, "expected-raw-multifasta-out.txt"), "rb").read()
    assert len(packed) < len(raw) // 8
