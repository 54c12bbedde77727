#!/usr/bin/env python3
"""Build container: a second reference-held pin, of an INTERMEDIATE result.  dist-docs/example/
sample-output/pintron-pipeline-log.txt is the log of a complete reference run (2012) on
dist-docs/example; est-fact's part of it records, per processed sequence, the size of the finished MEG
("The MEG has V vertices and E edges", is_too_complex, src/meg-simplification.c:105) -- i.e. the
outcome of pairings + edges + simplification + transitive reduction + compaction.  That version tried
the strands in another order than the present sources, so sequences are matched by their FASTA header:
every (V, E) the log holds for a header must be the size of one of the MEGs we print for that header
in megs.txt.  Extracts the data to tests/golden/example_log_megs.json."""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PINTRON_REFERENCE", "/root/reference")


def main():
    log = open(os.path.join(REF, "dist-docs/example/sample-output/pintron-pipeline-log.txt"), errors="replace").read().split("\n")
    start = next(i for i, l in enumerate(log) if "Creating the suffix tree" in l)
    sizes = collections.defaultdict(list)
    cur = None
    for l in log[start:]:
        m = re.match(r"\* INFO \(main\s*\) EST: (.*?)\s+\(src/main-est-fact", l)
        if m:
            cur = m.group(1)
            continue
        if "MIN-FACTORIZATION" in l:
            break
        m = re.search(r"\(is_too_complex\s*\) The MEG has\s+(\d+) vertices and\s+(\d+) edges", l)
        if m and cur is not None:
            sizes[cur].append([int(m.group(1)), int(m.group(2))])
    # what the reference's PRESENT sources (object code, est-fact-core) do not reproduce of that 2012 log
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import example_log_lib as EL
    import shutil
    import subprocess
    import tempfile
    w = tempfile.mkdtemp()
    for f in ("genomic.txt", "ests.txt"):
        shutil.copy(os.path.join(REF, "dist-docs", "example", f), w)
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "est-fact-core")], cwd=w, stderr=subprocess.DEVNULL, check=True)
    drift = EL.not_reproduced(sizes, EL.meg_sizes(os.path.join(w, "megs.txt")))
    shutil.rmtree(w)
    out = {"drift": drift, "source": "dist-docs/example/sample-output/pintron-pipeline-log.txt (reference-held log of a full run)",
           "what": "per FASTA header: (vertices, edges) of every finished MEG the log reports",
           "headers": len(sizes), "megs": sum(len(v) for v in sizes.values()), "sizes": sizes}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "example_log_megs.json"), "w"), indent=0, sort_keys=True)
    print("headers", len(sizes), "MEG sizes", out["megs"], "not reproduced by the present reference sources:", len(drift))


if __name__ == "__main__":
    sys.exit(main())
