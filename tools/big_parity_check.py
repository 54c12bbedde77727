#!/usr/bin/env python3
"""GPU box: whole-program parity at a larger scale than the test suite -- C3 x 24 000, C2 x 12 000 and
C5 x 60 000 reads (fresh seeds) through est-fact on the GPU and through the compiled reference (16
processes over EST chunks); all five compared files must be identical."""
import os, sys, subprocess, time, filecmp, shutil
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from pintron_amd import synth
base = "/tmp/bigcheck"; shutil.rmtree(base, ignore_errors=True)
ok = True
for cfg, n, seed in (("C3", 24000, 11), ("C2", 12000, 12), ("C5", 60000, 13)):
    w = synth.make(cfg, n_est=n, seed=seed)
    # split the reference run over 16 processes (chunks of ESTs), our run is one process
    mine = os.path.join(base, cfg, "mine"); synth.write_files(w, mine)
    t0 = time.time(); subprocess.run([os.path.join(ROOT, "pintron_amd/bin/est-fact")], cwd=mine, check=True, stderr=subprocess.DEVNULL); t_m = time.time() - t0
    chunks = 16; per = (n + chunks - 1) // chunks; procs = []
    for c in range(chunks):
        d = os.path.join(base, cfg, "ref%02d" % c); os.makedirs(d)
        open(d + "/genomic.txt", "w").write(w.genomic_fasta())
        open(d + "/ests.txt", "w").write("".join("%s\n%s\n" % (h, s.decode()) for h, s in zip(w.est_headers[c*per:(c+1)*per], w.est_seqs[c*per:(c+1)*per])))
        procs.append(subprocess.Popen([os.path.join(ROOT, "oracle/_ref/est-fact-core")], cwd=d, stderr=subprocess.DEVNULL))
    t0 = time.time(); rcs = [p.wait() for p in procs]; t_r = time.time() - t0
    for f in ("raw-multifasta-out.txt", "processed-ests.txt", "megs.txt", "processed-megs.txt", "meg-edges.txt"):
        ref = b"".join(open(os.path.join(base, cfg, "ref%02d" % c, f), "rb").read() for c in range(chunks))
        got = open(os.path.join(mine, f), "rb").read()
        same = ref == got
        ok &= same
        print(cfg, n, f, "identical" if same else "DIFFERENT", len(got))
    print(cfg, "ours %.1f s, reference (16 processes) %.1f s" % (t_m, t_r), rcs.count(0), "ref procs ok")
print("ALL IDENTICAL" if ok else "MISMATCH")
