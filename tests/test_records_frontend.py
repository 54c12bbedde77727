"""SURVEY.md section 8(f).1 -- min-factorization input coupling.  est-fact's packed factorization records
(include/pintron_records.h) carry everything the reference's stage 2 parses out of
raw-multifasta-out.txt (src/io-factorizations.c:104-238): the reference's own min-factorization,
given a front end that reads the records (oracle/ref_minfact_records_main.c, the binding shown in
INTEGRATION.md section 6), must print the same out-agree.txt as the unmodified program fed the text.

CPU: records written by the check build (host C + oracle); GPU: by the product binary."""
import os
import shutil
import subprocess

import pytest

import regression_lib as RL

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.path.join(ROOT, "oracle", "_ref")
TEXT_FRONT = os.path.join(REF, "min-factorization-ref")
REC_FRONT = os.path.join(REF, "min-factorization-records")
needs_ref = pytest.mark.skipif(not (os.path.exists(TEXT_FRONT) and os.path.exists(REC_FRONT)),
                               reason="oracle/_ref stage binaries not present")


def both_front_ends(exe, workdir, env=None):
    e = dict(os.environ, PINTRON_RECORDS_FILE="records.bin")
    e.update(env or {})
    subprocess.run([exe], cwd=workdir, env=e, check=True, stderr=subprocess.DEVNULL)
    with open(os.path.join(workdir, "raw-multifasta-out.txt"), "rb") as fin:
        text = subprocess.run([TEXT_FRONT], cwd=workdir, stdin=fin, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    rec = subprocess.run([REC_FRONT, "records.bin", "processed-ests.txt"], cwd=workdir, stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, check=True).stdout
    return text, rec


def stage(case, workdir):
    if case in RL.manifest()["cases"]:
        RL.stage_inputs(case, str(workdir))
    else:
        from pintron_amd import synth
        name, n = case.split(":")
        synth.write_files(synth.make(name, n_est=int(n)), str(workdir))


@needs_ref
@pytest.mark.parametrize("case", ["test-AMBN", "test-CPB2", "test-issue-31", "C2:300"])
def test_records_front_end_cpu(tmp_path, case):
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "hostcheck"), "estfact_sched_check"], check=True)
    stage(case, tmp_path)
    text, rec = both_front_ends(os.path.join(HERE, "hostcheck", "estfact_sched_check"), tmp_path, {"PINTRON_THREADS": "2"})
    assert text == rec and len(text) > 1000


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("case", ["test-AMBN", "test-issue-13", "test_gtf8", "C2:1000", "C3:3000"])
def test_records_front_end_gpu(tmp_path, case):
    import __graft_entry__ as g
    g.build()
    stage(case, tmp_path)
    text, rec = both_front_ends(os.path.join(ROOT, "pintron_amd", "bin", "est-fact"), tmp_path)
    assert text == rec and len(text) > 1000
