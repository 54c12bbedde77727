"""est-fact under NON-default options against the compiled reference (oracle/_ref/est-fact-core with the same
effective options, tests/option_sets.py): five files byte-identical.  CPU: the product's host code over the CPU
oracle (tests/hostcheck/estfact_sched_check); `-m gpu`: the product binary."""
import filecmp
import gzip
import os
import shutil
import subprocess

import pytest

import option_sets as OS
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FILES = ["raw-multifasta-out.txt", "processed-ests.txt", "megs.txt", "processed-megs.txt", "meg-edges.txt"]
REF = os.path.join(ROOT, "oracle", "_ref", "est-fact-core")


def write_inputs(which, d):
    os.makedirs(d, exist_ok=True)
    if which == "ambn":
        for f in ("genomic.txt", "ests.txt"):
            shutil.copy(os.path.join(HERE, "golden", "ambn", f), d)
    elif which == "issue13":
        for f in ("genomic.txt", "ests.txt"):
            open(os.path.join(d, f), "wb").write(gzip.open(os.path.join(HERE, "golden", "issue13", f + ".gz")).read())
    else:
        from pintron_amd import synth
        synth.write_files(synth.make("C2", n_est=250), d)


def run_pair(exe, tmp_path, which, opt, env=None):
    _id, argv, ini, ini_name, eff = opt
    ref_dir, my_dir = str(tmp_path / "ref"), str(tmp_path / "mine")
    write_inputs(which, ref_dir)
    write_inputs(which, my_dir)
    open(os.path.join(ref_dir, "ref-options.ini"), "w").write(OS.ref_options_text(eff))
    if ini is not None:
        open(os.path.join(my_dir, ini_name), "w").write(ini)
    subprocess.run([REF], cwd=ref_dir, check=True, stderr=subprocess.DEVNULL)
    e = dict(os.environ)
    e.update(env or {})
    subprocess.run([exe] + argv, cwd=my_dir, check=True, env=e, stderr=subprocess.DEVNULL)
    for f in FILES:
        assert filecmp.cmp(os.path.join(my_dir, f), os.path.join(ref_dir, f), shallow=False), (opt[0], which, f)
    # the product's own record of what it ran with agrees with the table
    dump = open(os.path.join(my_dir, "config-dump.ini")).read()
    for k, v in eff.items():
        assert (('%s="%s"' % (k, v)) in dump) if v is not None else (("\n" + k + "\n") in dump), (opt[0], k)
    return os.path.getsize(os.path.join(my_dir, "raw-multifasta-out.txt"))


def test_every_option_is_moved_off_its_default_by_some_set():
    names = set(OS.DEFAULTS) | set(OS.FLAGS) | {"config-file"}
    assert len(names) == 24
    assert OS.covered_options() == names


@pytest.fixture(scope="module")
def check_bin():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "hostcheck"), "all"], check=True)
    return os.path.join(HERE, "hostcheck", "estfact_sched_check")


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("opt", OS.SETS, ids=[s[0] for s in OS.SETS])
@pytest.mark.parametrize("which", ["ambn", "c2"])
def test_options_host_logic_over_the_oracle(check_bin, tmp_path, which, opt):
    run_pair(check_bin, tmp_path, which, opt, {"PINTRON_THREADS": "2"})


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("opt", [s for s in OS.SETS if s[0] in ("no-externals", "windows", "cmeg", "filters", "ini")],
                         ids=lambda s: s[0])
def test_options_host_logic_over_the_oracle_real_ests(check_bin, tmp_path, opt):
    assert run_pair(check_bin, tmp_path, "issue13", opt, {"PINTRON_THREADS": "4"}) > 10000


@pytest.mark.gpu
@pytest.mark.parametrize("opt", OS.SETS, ids=[s[0] for s in OS.SETS])
@pytest.mark.parametrize("which", ["ambn", "c2", "issue13"])
def test_options_product_binary(tmp_path, which, opt):
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/est-fact-core not present")
    run_pair(os.path.join(ROOT, "pintron_amd", "bin", "est-fact"), tmp_path, which, opt)
