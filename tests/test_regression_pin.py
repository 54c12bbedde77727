"""The reference-held regressionTest goldens (referenceOutput/full.json of 18 cases) as the pin of
the whole est-fact path -- see tests/regression_lib.py for what is compared and why it is a
function of est-fact's output alone.

CPU (-m "not gpu"): the product's host C with the CPU oracle as backend (tests/hostcheck/
estfact_check) -> reference stages 2/3 -> table == full.json's table, except the records listed in
drift.json (where the reference's present sources differ from their own older golden); this is
what pins oracle/dp_oracle.c and oracle/pairing_oracle.c on real data.
GPU (-m gpu): the same with the product binary (HIP path) in place of the check build."""
import os
import subprocess

import pytest

import regression_lib as RL

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CASES = sorted(RL.manifest()["cases"])


def check_case(case, exe, tmp_path, env=None):
    RL.stage_inputs(case, str(tmp_path))
    e = dict(os.environ)
    e.update(env or {})
    run = subprocess.run([exe], cwd=tmp_path, env=e, stderr=subprocess.PIPE, text=True, errors="replace")
    assert run.returncode == 0, "est-fact failed (%d): %s" % (run.returncode, run.stderr[-1500:])
    raw = (tmp_path / "raw-multifasta-out.txt").read_bytes()
    assert raw == RL.expected_raw(case), "raw-multifasta-out.txt differs from the reference object code's"
    if not RL.have_stages():
        pytest.skip("oracle/_ref stage binaries not present: compared est-fact's own output only")
    RL.run_stages(str(tmp_path))
    diff = RL.diff_tables(RL.introns_table(str(tmp_path)), RL.reference_introns(case))
    with open(os.path.join(RL.GOLD, case, "drift.json")) as f:
        import json
        drift = json.load(f)
    assert diff == drift, "fields differ from referenceOutput/full.json beyond the recorded drift"


def test_manifest_is_consistent():
    m = RL.manifest()
    assert len(m["cases"]) == 18
    assert m["summary"]["compared_fields"] == sum(c["compared_fields"] for c in m["cases"].values())
    # the drift must stay a small minority of what is compared, else the pin means nothing
    assert m["summary"]["drift_records"] * 200 < m["summary"]["compared_fields"]
    for c in m["cases"].values():
        assert c["files_identical_core_vs_host"]


@pytest.fixture(scope="module")
def check_bin():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "hostcheck"), "estfact_check"], check=True)
    return os.path.join(HERE, "hostcheck", "estfact_check")


@pytest.mark.parametrize("case", CASES)
def test_host_code_with_oracle_reproduces_reference_goldens(check_bin, tmp_path, case):
    check_case(case, check_bin, tmp_path)


@pytest.fixture(scope="module")
def product_bin():
    import __graft_entry__ as g
    g.build()
    return os.path.join(ROOT, "pintron_amd", "bin", "est-fact")


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_path_reproduces_reference_goldens(product_bin, tmp_path, case):
    check_case(case, product_bin, tmp_path)
