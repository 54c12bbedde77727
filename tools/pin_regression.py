#!/usr/bin/env python3
"""Pins the whole est-fact path against the reference-held regressionTest goldens (build container
only: needs /root/reference and oracle/_ref).  TEST INFRASTRUCTURE.

For every regressionTest/<case> with ests.txt and referenceOutput/full.json:
  1. run oracle/_ref/est-fact-core (reference object code under oracle/ref_core_driver.c) and
     tests/hostcheck/estfact_check (the product's host C with the CPU oracle as its backend);
     their raw-multifasta-out.txt / processed-ests.txt must be byte-identical;
  2. run the reference's unmodified min-factorization and intron-agreement on the result and
     read the predicted introns + supporting-EST factor boundaries (tests/regression_lib.py);
  3. compare with the case's full.json; every difference is recorded as "drift" (the reference's
     present sources against its own older golden), nothing is hidden;
  4. write tests/golden/regression/<case>/: the reference's input data files (xz), the table
     extracted from full.json, the drift list, est-fact's expected raw-multifasta-out.txt (xz).
"""
import hashlib
import json
import lzma
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import regression_lib as RL  # noqa: E402

REF = os.environ.get("PINTRON_REFERENCE", "/root/reference")
RT = os.path.join(REF, "regressionTest")
CORE = os.path.join(ROOT, "oracle", "_ref", "est-fact-core")
MINE = os.path.join(ROOT, "tests", "hostcheck", "estfact_check")


def xz_write(path, data):
    with lzma.open(path, "wb", preset=9 | lzma.PRESET_EXTREME) as f:
        f.write(data)


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostcheck"), "estfact_check"], check=True)
    os.makedirs(RL.GOLD, exist_ok=True)
    by_md5 = {}
    man = {"source": "regressionTest/*/{genomic.txt,ests.txt,referenceOutput/full.json} of the reference",
           "generator": "tools/pin_regression.py", "cases": {}}
    only = set(sys.argv[1:])
    cases = sorted(c for c in os.listdir(RT)
                   if os.path.exists(os.path.join(RT, c, "ests.txt"))
                   and os.path.exists(os.path.join(RT, c, "referenceOutput", "full.json")))
    ok = True
    for case in cases:
        if only and case not in only:
            continue
        out = os.path.join(RL.GOLD, case)
        os.makedirs(out, exist_ok=True)
        entry = {}
        for name in ("genomic", "ests"):
            data = open(os.path.join(RT, case, name + ".txt"), "rb").read()
            h = hashlib.md5(data).hexdigest()
            if h not in by_md5:                       # identical inputs are stored once
                by_md5[h] = "%s/%s.txt.xz" % (case, name)
                xz_write(os.path.join(RL.GOLD, by_md5[h]), data)
            entry[name] = by_md5[h]
        golden = RL.table_from_full_json(os.path.join(RT, case, "referenceOutput", "full.json"))
        with open(os.path.join(out, "reference_introns.json"), "w") as f:
            json.dump(golden, f, indent=0, sort_keys=True)
        tables, raws, secs = {}, {}, {}
        for tag, exe in (("core", CORE), ("mine", MINE)):
            w = tempfile.mkdtemp(prefix="pin_%s_%s_" % (case, tag))
            for name in ("genomic", "ests"):
                shutil.copy(os.path.join(RT, case, name + ".txt"), w)
            t = time.time()
            subprocess.run([exe], cwd=w, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            secs[tag] = round(time.time() - t, 2)
            raws[tag] = {f: open(os.path.join(w, f), "rb").read()
                         for f in ("raw-multifasta-out.txt", "processed-ests.txt", "megs.txt",
                                   "processed-megs.txt", "meg-edges.txt")}
            RL.run_stages(w)
            tables[tag] = RL.introns_table(w)
            shutil.rmtree(w)
        same_files = all(raws["core"][f] == raws["mine"][f] for f in raws["core"])
        drift = RL.diff_tables(tables["core"], golden)
        mine_vs_core = RL.diff_tables(tables["mine"], tables["core"])
        with open(os.path.join(out, "drift.json"), "w") as f:
            json.dump(drift, f, indent=0)
        xz_write(os.path.join(out, "expected-raw-multifasta-out.txt.xz"), raws["core"]["raw-multifasta-out.txt"])
        n_rec = RL.count_records(golden)
        entry.update(introns=len(golden), supporting_records=n_rec, compared_fields=4 * n_rec + 2 * len(golden),
                     drift_records=len(drift), files_identical_core_vs_host=same_files,
                     raw_md5=hashlib.md5(raws["core"]["raw-multifasta-out.txt"]).hexdigest(),
                     processed_ests_md5=hashlib.md5(raws["core"]["processed-ests.txt"]).hexdigest(),
                     seconds=secs)
        man["cases"][case] = entry
        good = same_files and not mine_vs_core
        ok = ok and good
        print("%-42s introns %3d records %5d drift %3d  core==host files %s  %s" %
              (case, len(golden), n_rec, len(drift), same_files, "ok" if good else "MISMATCH"), flush=True)
    if not only:
        tot = sum(c["compared_fields"] for c in man["cases"].values())
        dr = sum(c["drift_records"] for c in man["cases"].values())
        man["summary"] = {"cases": len(man["cases"]), "compared_fields": tot, "drift_records": dr}
        with open(os.path.join(RL.GOLD, "manifest.json"), "w") as f:
            json.dump(man, f, indent=1, sort_keys=True)
        print("total: %d cases, %d compared fields, %d drift records" % (len(man["cases"]), tot, dr))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
