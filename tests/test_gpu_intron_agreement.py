"""SURVEY.md section 8(f).2 -- the alignment core of intron-agreement on the same kernels.  The reference's
stage 3 re-aligns the exon borders around every intron with compute_alignment, edit_distance and
compute_gap_alignment (src/agree-introns.c:629,695,762,837; src/main-intron-agreement.c:857,864).
oracle/_ref/intron-agreement-gpu is the reference's own intron-agreement with those three entry points
answered by libpintron_gpu.so (oracle/ref_agree_gpu_shim.c): its two output files
must equal those of the unmodified intron-agreement, on est-fact output produced by the HIP path.  Both
programs run on zero-filled malloc blocks (regression_lib.STAGE_ENV says why)."""
import os
import shutil
import subprocess

import pytest

import regression_lib as RL

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
AGREE_GPU = os.path.join(ROOT, "oracle", "_ref", "intron-agreement-gpu")


@pytest.mark.parametrize("case", ["test-AMBN", "test-CPB2", "test-issue-31", "test-mattia1", "test_gtf6"])
def test_intron_agreement_with_device_alignments(tmp_path, case):
    if not (RL.have_stages() and os.path.exists(AGREE_GPU)):
        pytest.skip("oracle/_ref stage binaries not present")
    import __graft_entry__ as g
    g.build()
    work = tmp_path / "work"
    RL.stage_inputs(case, str(work))
    subprocess.run([os.path.join(ROOT, "pintron_amd", "bin", "est-fact")], cwd=work, check=True, stderr=subprocess.DEVNULL)
    with open(work / "raw-multifasta-out.txt", "rb") as fin, open(work / "out-agree.txt", "wb") as fout:
        subprocess.run([RL.MINFACT], cwd=work, stdin=fin, stdout=fout, stderr=subprocess.DEVNULL, check=True)
    outs = {}
    for tag, exe in (("ref", RL.AGREE), ("gpu", AGREE_GPU)):
        d = tmp_path / tag
        shutil.copytree(work, d)
        r = subprocess.run([exe], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, errors="replace",
                           env=dict(RL.STAGE_ENV, PINTRON_VERBOSE="1"))
        assert r.returncode == 0, r.stderr[-1500:]
        outs[tag] = [(d / f).read_bytes() for f in ("predicted-introns.txt", "out-after-intron-agree.txt")]
        if tag == "gpu":
            line = [ln for ln in r.stderr.splitlines() if ln.startswith("intron-agreement-gpu:")]
            assert line, "the device was not used"
            counts = [int(x) for x in line[-1].replace(",", "").split() if x.isdigit()]
            assert sum(counts) > 0
    assert outs["ref"] == outs["gpu"]
    assert len(outs["ref"][0]) > 100
