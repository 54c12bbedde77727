"""GPU box: where a fresh batch's time goes inside a warm process (PINTRON_VERBOSE output of two sessions
opened one after the other on the same C3 batch)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pintron_amd import synth
from pintron_amd.estfact import Session, load_host_lib
os.environ["PINTRON_VERBOSE"] = "1"
L = load_host_lib()
d = tempfile.mkdtemp()
synth.write_files(synth.make("C3"), d)
for rep in range(3):
    t0 = time.perf_counter()
    s = Session(L, d); t1 = time.perf_counter()
    st = s.step(); t2 = time.perf_counter()
    s.close(); t3 = time.perf_counter()
    print("fresh %d: open %.3f step %.3f close %.3f (prefetch %.3f workers %.3f host/thread %.3f)" %
          (rep, t1 - t0, t2 - t1, t3 - t2, st.prefetch_s, st.workers_s, st.host_s / st.threads), file=sys.stderr)
