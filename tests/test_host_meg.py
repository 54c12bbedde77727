"""CPU: the host-side sequence preparation + MEG construction (pintron_amd/host/) reproduces the
reference's megs.txt records byte for byte on the test-AMBN fixture (pairings from the CPU
oracle here; from the HIP index in the GPU tests)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "ambn")


def meg_blocks(path):
    d = {}
    for b in open(path).read().split("\n\n***********\n\n")[1:]:
        lines = b.split("\n")
        d.setdefault((lines[0], lines[1]), []).append(b.rstrip("\n"))
    return d


def test_ambn_megs_match_reference(tmp_path):
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "hostcheck"), "meg_check"], check=True)
    for f in ("genomic.txt", "ests.txt"):
        shutil.copy(os.path.join(GOLD, f), tmp_path)
    subprocess.run([os.path.join(HERE, "hostcheck", "meg_check")], cwd=tmp_path, check=True)
    ref = meg_blocks(os.path.join(GOLD, "expected-megs.txt"))
    got = meg_blocks(os.path.join(tmp_path, "megs-check.txt"))
    assert len(ref) >= 25
    for k, v in ref.items():
        assert k in got, k[0]
        assert got[k][-1] == v[-1], k[0]
