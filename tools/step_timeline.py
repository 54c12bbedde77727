"""GPU box: PINTRON_VERBOSE=2 output of warm C3 steps (when each range of pairings + MEGs was ready, the service
threads' phases) -- where the start of a step goes."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PINTRON_VERBOSE"] = "2"
from pintron_amd import synth
from pintron_amd.estfact import Session, load_host_lib
L = load_host_lib()
d = tempfile.mkdtemp()
synth.write_files(synth.make(os.environ.get("WORKLOAD", "C3")), d)
s = Session(L, d)
for k in range(4):
    sys.stderr.write("=== step %d\n" % k); sys.stderr.flush()
    t0 = time.perf_counter(); st = s.step(); t1 = time.perf_counter()
    sys.stderr.write("=== step %d took %.1f ms (prefetch %.3f workers %.3f)\n" % (k, 1e3 * (t1 - t0), st.prefetch_s, st.workers_s)); sys.stderr.flush()
s.close()
