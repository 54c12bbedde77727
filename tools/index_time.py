import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
from pintron_amd import capi, synth
for wl in ("C3", "C5"):
    w = synth.make(wl, n_est=10)
    with capi.Context(0) as ctx:
        for rep in range(3):
            t0 = time.perf_counter()
            idx = capi.Index(ctx, w.genomic)
            print(wl, "index build %.1f ms" % (1e3 * (time.perf_counter() - t0)), file=sys.stderr)
            idx.close()
