#!/usr/bin/env python3
"""Builds tests/golden/c3_sample_jobs.jsonl.gz: the DP calls the unmodified reference est-fact makes
on a seeded C3-shaped sample (200 kb genomic, 3 % errors, ~600 bp ESTs), with their outputs.

Container only (needs /root/reference -> oracle/_ref).  Calls to the exported DP routines are
captured by oracle/dp_capture_shim.c; find_longest_common_factor_dp is `static` and cannot be
interposed, so its calls in search_small_exon_at_prefix (src/factorization-refinement.c:500-536)
are reconstructed from the reference's own raw-multifasta-out.txt: one call
LCF(GEN[0..GEN_start), EST[EST_start-eplen..EST_start)) per output factorization whose first exon
satisfies the routine's entry condition (:517-519); its expected output comes from the compiled
static routine (oracle/ref_static_access.c).  Used as golden vectors by tests/test_oracle_golden.py
and tests/test_gpu_dp_parity.py (tests/golden_cases.py: load_c3_sample).
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pintron_amd import synth  # noqa: E402

N_EST = 400
LB_SMALL, UB_SMALL = 6, 23     # _LB/_UB_SMALL_EXON_LENGTH_ (src/factorization-refinement.c:56,58)


def main():
    import ref_lib as R
    w = synth.make("C3", n_est=N_EST)
    tmp = tempfile.mkdtemp(prefix="pintron_c3fx_")
    synth.write_files(w, tmp)
    cap = os.path.join(tmp, "cap.jsonl")
    env = dict(os.environ, PINTRON_DP_CAPTURE=cap,
               LD_PRELOAD=os.path.join(ROOT, "oracle", "_ref", "libdpcapture.so"))
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "est-fact-core")], cwd=tmp, env=env,
                   stderr=subprocess.DEVNULL, check=True)
    lines = open(cap).read().splitlines()
    # LCF reconstruction from the reference output
    gen = w.genomic
    proc = {}
    cur = None
    for ln in open(os.path.join(tmp, "processed-ests.txt")):
        ln = ln.rstrip("\n")
        if ln.startswith(">"):
            cur = ln
        else:
            proc[cur] = ln.encode()
    n_lcf = 0
    hdr, first = None, None
    recs = []
    for ln in open(os.path.join(tmp, "raw-multifasta-out.txt")):
        ln = ln.rstrip("\n")
        if ln.startswith(">"):
            hdr, first = ln, None
        elif ln.startswith("#"):
            continue
        elif first is None:
            f = ln.split()
            first = tuple(int(x) - 1 for x in f[:4])
            es, ee, gs, ge = first
            e1len = ee + 1 - es
            if e1len + es >= LB_SMALL + UB_SMALL:
                eplen = min(es, gs, 2 * UB_SMALL)
                est = proc[hdr]
                a, b = gen[:gs], est[es - eplen:es]
                r = R.lcf(a, b) if eplen and gs else dict(occ1=0, occ2=0, len=0)
                recs.append(json.dumps(dict(k="LCF", a_gen_len=gs, b=b.decode("latin1"), **r)))
                n_lcf += 1
    out = os.path.join(ROOT, "tests", "golden", "c3_sample_jobs.jsonl.gz")
    with gzip.open(out, "wt") as f:
        f.write(json.dumps(dict(k="META", n_est=N_EST, config="C3", seed=synth.CONFIGS["C3"]["seed"],
                                aligned=len(proc), dp_calls=len(lines), lcf_calls=n_lcf)) + "\n")
        for ln in lines:
            f.write(ln + "\n")
        for ln in recs:
            f.write(ln + "\n")
    print("wrote", out, os.path.getsize(out), "bytes;", len(lines), "captured calls +", n_lcf, "LCF")


if __name__ == "__main__":
    main()
