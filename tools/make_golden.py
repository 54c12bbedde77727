#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the reference.

Runs only in the build container (needs /root/reference and oracle/_ref, see oracle/Makefile):
  1. burset_known_answers.json  -- the (donor, acceptor) -> frequency known answers asserted by
     the reference's own unit tests (test/refine-intron_test.c:148-920), extracted as DATA.
  2. dp_calls.jsonl.gz          -- a seeded sample of the DP calls the unmodified reference
     est-fact makes on regressionTest/test-AMBN and test-issue-13 (inputs + outputs), captured at
     the dynamic-linker level by oracle/dp_capture_shim.c.
  3. ambn/                      -- the reference's regression fixture regressionTest/test-AMBN
     (genomic.txt, ests.txt: data files of the reference's tests) and the outputs the compiled
     reference est-fact produces on it (raw-multifasta-out.txt, processed-ests.txt), plus md5s
     for the larger fixtures (SURVEY.md section 8c lists the same checksums).
"""
import gzip
import hashlib
import json
import os
import random
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PINTRON_REF", "/root/reference")
REFBIN = os.path.join(ROOT, "oracle", "_ref", "est-fact-core")
SHIM = os.path.join(ROOT, "oracle", "_ref", "libdpcapture.so")
GOLD = os.path.join(ROOT, "tests", "golden")


def burset_known_answers():
    src = open(os.path.join(REF, "test", "refine-intron_test.c")).read()
    out = []
    pat = re.compile(r'asprintf\(&gsd, "([^"]*)"\);\s*char \*gsa;\s*asprintf\(&gsa, "([^"]*)"\);\s*'
                     r'cr_expect\(getBursetFrequency\(gsd,gsa\)==(\d+)\)')
    for d, a, f in pat.findall(src):
        out.append(dict(donor=d, acceptor=a, freq=int(f)))
    pat2 = re.compile(r'char \*gs="([^"]*)";\s*int nl=(-?\d+);\s*int nr=(-?\d+);\s*'
                      r'cr_expect\(Check_Burset_patterns\(gs,nl,nr\)==(\d+)\)')
    chk = [dict(genomic=g, donor_left=int(l), acceptor_right=int(r), freq=int(f))
           for g, l, r, f in pat2.findall(src)]
    json.dump(dict(getBursetFrequency=out, Check_Burset_patterns=chk),
              open(os.path.join(GOLD, "burset_known_answers.json"), "w"), indent=0)
    print("burset known answers:", len(out), "+", len(chk))


def run_ref(fixture_dir, capture=None):
    tmp = tempfile.mkdtemp(prefix="pintron_gold_")
    for f in ("genomic.txt", "ests.txt"):
        shutil.copy(os.path.join(fixture_dir, f), tmp)
    env = dict(os.environ)
    if capture:
        env["PINTRON_DP_CAPTURE"] = capture
        env["LD_PRELOAD"] = SHIM
    subprocess.run([REFBIN], cwd=tmp, env=env, stderr=subprocess.DEVNULL, check=True)
    return tmp


def md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def main():
    os.makedirs(GOLD, exist_ok=True)
    burset_known_answers()
    rng = random.Random(20260101)
    sample = []
    sums = {}
    for name, per_kind in (("test-AMBN", 150), ("test-issue-13", 400)):
        fx = os.path.join(REF, "regressionTest", name)
        cap = tempfile.mktemp(suffix=".jsonl")
        tmp = run_ref(fx, cap)
        sums[name] = {f: md5(os.path.join(tmp, f)) for f in ("raw-multifasta-out.txt", "processed-ests.txt")}
        by_kind = {}
        for line in open(cap):
            r = json.loads(line)
            if len(r["a"]) + len(r["b"]) > 3000:
                continue
            by_kind.setdefault(r["k"], []).append(line)
        for k, lines in sorted(by_kind.items()):
            # keep the interesting ones: prefer calls whose operands differ
            rng.shuffle(lines)
            sample.extend(lines[:per_kind])
        if name == "test-AMBN":
            dst = os.path.join(GOLD, "ambn")
            os.makedirs(dst, exist_ok=True)
            for f in ("genomic.txt", "ests.txt"):
                shutil.copy(os.path.join(fx, f), dst)
            for f in ("raw-multifasta-out.txt", "processed-ests.txt", "megs.txt", "processed-megs.txt", "meg-edges.txt"):
                shutil.copy(os.path.join(tmp, f), os.path.join(dst, "expected-" + f))
        os.unlink(cap)
        shutil.rmtree(tmp)
    ex = os.path.join(REF, "dist-docs", "example")
    if os.path.exists(os.path.join(ex, "genomic.txt")):
        tmp = run_ref(ex)
        sums["dist-docs/example"] = {f: md5(os.path.join(tmp, f)) for f in ("raw-multifasta-out.txt", "processed-ests.txt")}
        shutil.rmtree(tmp)
    # LCF and AFFIX are `static` in the reference (not interposable): evaluate the compiled
    # reference routines (oracle/ref_static_access.c) on seeded inputs instead
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_cases as D
    import ref_lib as R
    for c in D.random_cases(random.Random(77), n_per_kind=200, max_len=500):
        if c.kind == D.LCF:
            r = R.lcf(c.a, c.b)
            sample.append(json.dumps(dict(k="LCF", a=c.a.decode(), b=c.b.decode(), **r)) + "\n")
        elif c.kind == D.AFFIX:
            r = R.longest_affix(c.a, c.b)
            sample.append(json.dumps(dict(k="AFFIX", a=c.a.decode(), b=c.b.decode(), **r)) + "\n")
    # pairings (MEG vertex sets) from the reference's own suffix tree + build_vertex_set
    import pairing_lib as PL
    sets = []
    gen = PL.read_fasta(os.path.join(REF, "regressionTest", "test-AMBN", "genomic.txt"))[0]
    ests = PL.read_fasta(os.path.join(REF, "regressionTest", "test-AMBN", "ests.txt"))
    ri = PL.RefIndex(gen)
    sets.append(dict(genomic=gen.decode(), cases=[
        dict(est=e.decode(), L=15, rate=0.2, pairings=ri.pairings(e).tolist())
        for e in ests + [PL.revcomp(x) for x in ests]]))
    gen, ests = PL.repeat_workload(5)
    ri = PL.RefIndex(gen)
    cases = []
    for L, rate in ((15, 0.2), (16, 0.2), (22, 0.2), (15, 0.5), (15, 1.0)):
        for e in ests:
            if e:
                cases.append(dict(est=e.decode(), L=L, rate=rate, pairings=ri.pairings(e, L, rate).tolist()))
    sets.append(dict(genomic=gen.decode(), cases=cases))
    # the occurrence t == 0 at upper tree levels (copies of one pairing): tests/pairing_lib.py
    n_dup = 0
    for gen, ests in PL.region_start_cases(11):
        ri = PL.RefIndex(gen)
        cases = []
        for e in ests:
            for L, rate in ((15, 0.2), (18, 0.1)):
                pr = ri.pairings(e, L, rate).tolist()
                n_dup += len(set(map(tuple, pr))) != len(pr)
                cases.append(dict(est=e.decode(), L=L, rate=rate, pairings=pr))
        sets.append(dict(genomic=gen.decode(), cases=cases))
    print("region-start cases with a repeated pairing:", n_dup)
    with gzip.open(os.path.join(GOLD, "pairings.json.gz"), "wt") as f:
        json.dump(dict(sets=sets), f)
    print("pairing cases:", sum(len(s["cases"]) for s in sets))
    with gzip.open(os.path.join(GOLD, "dp_calls.jsonl.gz"), "wt") as f:
        f.writelines(sample)
    json.dump(sums, open(os.path.join(GOLD, "reference_md5.json"), "w"), indent=1, sort_keys=True)
    print("dp calls sampled:", len(sample))
    print(json.dumps(sums, indent=1))


if __name__ == "__main__":
    sys.exit(main())
