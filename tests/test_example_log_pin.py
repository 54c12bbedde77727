"""A reference-held pin of an INTERMEDIATE result: the log of a complete reference run on
dist-docs/example (dist-docs/example/sample-output/pintron-pipeline-log.txt) reports the size of the
finished MEG of every processed sequence -- the outcome of pairings, edge construction,
simplification, transitive reduction and compaction.  tools/pin_example_log.py extracted the 713 sizes
(tests/golden/example_log_megs.json); matched by FASTA header, every one of them must be the size of
a MEG we print for that header, except the two entries the reference's present sources do not
reproduce either (`drift`: that 2012 version logged an empty MEG twice where today's prints one).

CPU: host C over the oracle (the MEG code the GPU stage is checked against);  GPU: the product, whose
MEGs come out of the device's MEG stage."""
import gzip
import json
import os
import subprocess

import pytest

import example_log_lib as EL

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def run_and_compare(exe, tmp_path):
    gold = json.load(open(os.path.join(HERE, "golden", "example_log_megs.json")))
    for f in ("genomic.txt", "ests.txt"):
        (tmp_path / f).write_bytes(gzip.open(os.path.join(HERE, "golden", "example", f + ".gz")).read())
    subprocess.run([exe], cwd=tmp_path, check=True, stderr=subprocess.DEVNULL)
    assert gold["megs"] == 713 and gold["headers"] == 623
    assert EL.not_reproduced(gold["sizes"], EL.meg_sizes(str(tmp_path / "megs.txt"))) == gold["drift"]
    assert len(gold["drift"]) == 2


def test_example_log_meg_sizes_cpu(tmp_path):
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "hostcheck"), "estfact_check"], check=True)
    run_and_compare(os.path.join(HERE, "hostcheck", "estfact_check"), tmp_path)


@pytest.mark.gpu
def test_example_log_meg_sizes_gpu(tmp_path):
    import __graft_entry__ as g
    g.build()
    run_and_compare(os.path.join(ROOT, "pintron_amd", "bin", "est-fact"), tmp_path)
