#!/usr/bin/env python3
"""Build container: the reference's raw-multifasta-out.txt for the EXACT batches bench.py times at
N = 1 (C3: 200 kb x 100 000 ESTs, seed 3; C2 in full; one gene of C4; one GPU's eighth of C5), as md5 ->
tests/golden/bench_md5.json.

The reference est-fact is one process per gene; ESTs are independent given the genomic sequence, so
the batch is cut into chunks of whole ESTs, oracle/_ref/est-fact-core (reference object code) runs on
every chunk (as many at a time as there are cores) and the texts are joined in input order -- the
file a single reference process would write, in 1/8 of the time.  bench.py compares the md5 of the
text its LAST TIMED step produced with this checksum."""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pintron_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "est-fact-core")


def reference_text(w, chunks, workers):
    base = tempfile.mkdtemp(prefix="bench_md5_")
    n = len(w.est_seqs)
    per = (n + chunks - 1) // chunks
    dirs = []
    for c in range(chunks):
        d = os.path.join(base, "c%03d" % c)
        os.makedirs(d)
        open(os.path.join(d, "genomic.txt"), "w").write(w.genomic_fasta())
        open(os.path.join(d, "ests.txt"), "w").write(
            "".join("%s\n%s\n" % (h, s.decode()) for h, s in zip(w.est_headers[c * per:(c + 1) * per], w.est_seqs[c * per:(c + 1) * per])))
        dirs.append(d)
    running, todo = [], list(dirs)
    while todo or running:
        while todo and len(running) < workers:
            running.append(subprocess.Popen([REF], cwd=todo.pop(0), stderr=subprocess.DEVNULL))
        for p in list(running):
            if p.poll() is not None:
                if p.returncode != 0:
                    raise SystemExit("reference failed")
                running.remove(p)
        time.sleep(0.05)
    out = {f: b"".join(open(os.path.join(d, f), "rb").read() for d in dirs)
           for f in ("raw-multifasta-out.txt", "processed-ests.txt")}
    shutil.rmtree(base)
    return out


# the per-GPU batches bench.py times (bench.py: PER_GPU): C2 in full, C3, one gene of C4, one GPU's eighth of C5
DEFAULT_N = {"C2": 1_000, "C3": 100_000, "C4": 62_500, "C5": 250_000}


def main():
    """make_bench_md5.py [WORKLOAD [N [SEED]]]  (a bare number = C3 with that many ESTs, as before)"""
    args = sys.argv[1:]
    name = "C3"
    if args and args[0] in synth.CONFIGS:
        name = args.pop(0)
    n = int(args[0]) if args else DEFAULT_N[name]
    seed = int(args[1]) if len(args) > 1 else synth.CONFIGS[name]["seed"]
    t0 = time.time()
    w = synth.make(name, n_est=n, seed=seed)
    texts = reference_text(w, chunks=min(64, max(1, n // 200)), workers=max(1, (os.cpu_count() or 2) - 1))
    path = os.path.join(ROOT, "tests", "golden", "bench_md5.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    key = "%s:%d:seed%d" % (name, n, seed)
    data[key] = {f: hashlib.md5(t).hexdigest() for f, t in texts.items()}
    data[key]["aligned"] = texts["processed-ests.txt"].count(b">")
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    print("%s x %d (seed %d): %s  [%.0f s wall]" % (name, n, seed, data[key], time.time() - t0))


if __name__ == "__main__":
    main()
