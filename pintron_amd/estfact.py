"""Python side of the est-fact host library (pintron_amd/lib/libestfact.so): sessions, and the
EST-sharded multi-GPU driver.

ESTs are independent given the genomic sequence (src/main-est-fact.c:249-291 creates and destroys
all per-EST state inside its loop), so N ranks each run est-fact on a contiguous range of the input
ESTs with the genomic sequence and its index replicated.  The only exchange is the text of the
output files, gathered to rank 0 in rank order -- which is input order -- after the ranks are done.
There is no collective on the data path.

Nothing here computes: the library does, and it has no CPU fallback.
"""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUTPUT_FILES = ("raw-multifasta-out.txt", "processed-ests.txt", "megs.txt", "processed-megs.txt",
                "processed-megs-info.txt", "meg-edges.txt")


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_double), ("launches", C.c_size_t),
                ("jobs", C.c_size_t), ("cells", C.c_ulonglong), ("algo_bytes", C.c_ulonglong)]


class SchedStats(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("threads", "units", "aligned", "dp_batches", "dp_jobs",
                                          "pairing_batches", "pairing_requests")] + \
               [(n, C.c_double) for n in ("load_s", "index_s", "prefetch_s", "workers_s", "host_s",
                                          "pairing_s", "dp_s")] + \
               [("n_kernels", C.c_int), ("kernels", KernelStat * 64), ("dp_busy_union_ms", C.c_double),
                ("suspensions_per_unit", C.c_double)]


def load_host_lib(path=None):
    """libestfact.so = pintron_amd/host/*.c (the est-fact program) linked against libpintron_gpu.so.
    PINTRON_ESTFACT_LIB overrides the path (the CPU test-suite points it at the check build)."""
    path = path or os.environ.get("PINTRON_ESTFACT_LIB") or os.path.join(ROOT, "pintron_amd", "lib", "libestfact.so")
    if not os.path.exists(path):
        raise RuntimeError("%s missing: run __graft_entry__.build() (there is no CPU fallback)" % path)
    L = C.CDLL(path)
    L.ef_session_open.restype = C.c_void_p
    L.ef_session_open.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    L.ef_session_step.argtypes = [C.c_void_p, C.POINTER(SchedStats)]
    L.ef_session_output.restype = C.c_void_p
    L.ef_session_output.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
    L.ef_session_n_ests.restype = C.c_size_t
    L.ef_session_n_ests.argtypes = [C.c_void_p]
    L.ef_session_write_outputs.argtypes = [C.c_void_p]
    L.ef_session_close.argtypes = [C.c_void_p]
    return L


class Session:
    """Inputs of `directory` (genomic.txt, ests.txt, optional config.ini) loaded, genomic index and
    prepared sequences resident in HBM; step() = one pass of the whole est-fact hot path."""

    def __init__(self, L, directory):
        self.L, self.dir = L, directory
        cwd = os.getcwd()
        os.chdir(directory)
        try:
            argv = (C.c_char_p * 2)(b"est-fact", None)
            self.h = L.ef_session_open(1, argv)
        finally:
            os.chdir(cwd)
        if not self.h:
            raise RuntimeError("est-fact session could not start (no MI355X / libpintron_gpu.so?)")

    def step(self) -> SchedStats:
        st = SchedStats()
        if self.L.ef_session_step(self.h, C.byref(st)) != 0:
            raise RuntimeError("est-fact step failed")
        return st

    def n_ests(self) -> int:
        return int(self.L.ef_session_n_ests(self.h))

    def output(self, which: int) -> bytes:
        n = C.c_size_t()
        p = self.L.ef_session_output(self.h, which, C.byref(n))
        data = C.string_at(p, n.value)
        C.CDLL(None).free(C.c_void_p(p))
        return data

    def records(self) -> bytes:
        return self.output(0)

    def output_tensor(self, which: int):
        """The same text as a torch uint8 tensor (one copy out of the C buffer, none through bytes)."""
        import torch
        n = C.c_size_t()
        p = self.L.ef_session_output(self.h, which, C.byref(n))
        try:
            if n.value == 0:
                return torch.empty(0, dtype=torch.uint8)
            view = (C.c_ubyte * n.value).from_address(p)
            return torch.frombuffer(view, dtype=torch.uint8).clone()
        finally:
            C.CDLL(None).free(C.c_void_p(p))

    def close(self):
        if self.h:
            self.L.ef_session_close(self.h)
            self.h = None


# ---- packed factorization records -----------------------------------------
RECORDS = 6     # Session.output(RECORDS): see ef_write_factorization_records (pintron_amd/host/ef_estfact.c)


def parse_factorization_records(data: bytes):
    """[(est_index, [(polya, polyad, [(est_start, est_end, gen_start, gen_end), ...]), ...]), ...]
    with the coordinates exactly as raw-multifasta-out.txt prints them (1-based, inclusive)."""
    import struct
    out, pos = [], 0
    while pos < len(data):
        est_index, n_fact = struct.unpack_from("<II", data, pos)
        pos += 8
        facts = []
        for _ in range(n_fact):
            polya, polyad, n_exons = struct.unpack_from("<BBH", data, pos)
            pos += 4
            exons = [struct.unpack_from("<4i", data, pos + 16 * k) for k in range(n_exons)]
            pos += 16 * n_exons
            facts.append((polya, polyad, exons))
        out.append((est_index, facts))
    return out


def format_raw_multifasta(records, processed_ests: bytes, genomic_seq: bytes) -> bytes:
    """raw-multifasta-out.txt rebuilt from the packed records, processed-ests.txt (header and
    strand-corrected sequence of every aligned EST, in the same order) and the genomic sequence as
    it stands in genomic.txt -- i.e. the text carries nothing that the records do not."""
    ests = []
    lines = processed_ests.split(b"\n")
    for k in range(0, len(lines) - 1, 2):
        ests.append((lines[k][1:], lines[k + 1]))
    out = []
    for (est_index, facts), (header, seq) in zip(records, ests):
        for polya, polyad, exons in facts:
            out.append(b">" + header + b"\n#polya=%d\n#polyad=%d\n" % (polya, polyad))
            for es, ee, gs, ge in exons:
                out.append(b"%d %d %d %d %s %s\n" % (es, ee, gs, ge, seq[es - 1:ee], genomic_seq[gs - 1:ge]))
    return b"".join(out)


# ---- sharding -----------------------------------------------------------------------------------
def read_multifasta_records(path):
    """ests.txt as a list of byte records (header line + sequence lines), in file order."""
    recs, cur = [], []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b">") and cur:
                recs.append(b"".join(cur))
                cur = []
            if line.strip() or cur:
                cur.append(line)
    if cur:
        recs.append(b"".join(cur))
    return recs


def partition(weights, world):
    """Contiguous ranges [lo, hi) per rank, balanced by the sum of weights (sequence lengths).
    An input EST and its reverse-complement retry are one record, so they stay on one rank."""
    total, n = sum(weights), len(weights)
    bounds, acc, i = [0], 0, 0
    for r in range(1, world):
        while i < n and acc * world < total * r:
            acc += weights[i]
            i += 1
        bounds.append(i)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def gather_tensor(t, dist, rank, world, device):
    """Variable-length gather of a uint8 tensor to rank 0: all_gather of the byte counts, then one
    gather of the padded payloads (RCCL on GPUs, gloo on CPU).  Returns the list of per-rank
    tensors (on `device`, trimmed) on rank 0, None elsewhere."""
    import torch
    n = torch.tensor([t.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    pad = torch.empty(max(max(sizes), 1), dtype=torch.uint8, device=device)
    pad[: t.numel()] = t.to(device)
    outl = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, outl, dst=0)
    if rank != 0:
        return None
    return [o[: sizes[r]] for r, o in enumerate(outl)]


def gather_bytes(data: bytes, dist, rank, world, device):
    """gather_tensor for a bytes payload; returns the list of payloads (bytes) on rank 0."""
    import torch
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8) if data else torch.empty(0, dtype=torch.uint8)
    parts = gather_tensor(t, dist, rank, world, device)
    if parts is None:
        return None
    return [bytes(o.cpu().numpy().tobytes()) for o in parts]


def run_sharded(directory, workdir, dist, rank, world, device, L=None, files=(0, 1)):
    """est-fact on `directory` split over `world` ranks.  Every rank writes its EST range (+ the
    genomic sequence and config.ini) under `workdir`, runs one session step, and rank 0 receives
    the output text of `files` (indices into OUTPUT_FILES) in input order and writes it into
    `directory`.  Returns per-rank statistics."""
    import shutil
    L = L or load_host_lib()
    recs = read_multifasta_records(os.path.join(directory, "ests.txt"))
    lo, hi = partition([len(r) for r in recs], world)[rank]
    os.makedirs(workdir, exist_ok=True)
    shutil.copy(os.path.join(directory, "genomic.txt"), os.path.join(workdir, "genomic.txt"))
    if os.path.exists(os.path.join(directory, "config.ini")):
        shutil.copy(os.path.join(directory, "config.ini"), os.path.join(workdir, "config.ini"))
    with open(os.path.join(workdir, "ests.txt"), "wb") as f:
        f.write(b"".join(recs[lo:hi]))
    texts = {k: b"" for k in files}
    stats = {"ests": 0, "aligned": 0, "dp_jobs": 0}
    error = None
    if hi > lo:
        try:
            sess = Session(L, workdir)
            try:
                st = sess.step()
                for k in files:
                    texts[k] = sess.output(k)
                stats = {"ests": hi - lo, "aligned": int(st.aligned), "dp_jobs": int(st.dp_jobs)}
            finally:
                sess.close()
        except Exception as e:      # noqa: BLE001 -- reported below, on every rank
            error = e
    # a rank that failed must not leave the others waiting in the gathers: agree first
    import torch
    ok = torch.tensor([0 if error else 1], dtype=torch.int32, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        raise RuntimeError("est-fact failed on rank %d: %s" % (rank, error) if error
                           else "est-fact failed on another rank")
    for k in files:
        parts = gather_bytes(texts[k], dist, rank, world, device)
        if rank == 0:
            with open(os.path.join(directory, OUTPUT_FILES[k]), "wb") as f:
                f.write(b"".join(parts))
    return stats
