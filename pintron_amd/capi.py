"""ctypes binding of include/pintron_gpu.h (the C-ABI of libpintron_gpu.so)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpintron_gpu.so")

PGPU_OK, PGPU_EDEVICE, PGPU_ENOMEM, PGPU_EINVAL, PGPU_ENOSPC, PGPU_ERANGE, PGPU_ENOSYS = \
    0, -5, -12, -22, -28, -34, -38
ALIGN, GAP, ED, KBAND, LCF, BORDERS, AFFIX = range(7)
KIND_NAMES = ["ALIGN", "GAP", "ED", "KBAND", "LCF", "BORDERS", "AFFIX"]
JOB_A_GENOMIC, JOB_B_GENOMIC = 1, 2


class DpJob(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("flags", C.c_uint32), ("a_off", C.c_uint64),
                ("b_off", C.c_uint64), ("a_len", C.c_uint32), ("b_len", C.c_uint32),
                ("p0", C.c_uint32), ("p1", C.c_uint32), ("p2", C.c_uint32), ("tail", C.c_uint32)]


class DpResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("v", C.c_int32 * 6), ("pad", C.c_int32),
                ("str", C.c_uint64 * 2)]


class GroupInfo(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("kind", C.c_int32), ("pad", C.c_int32),
                ("jobs", C.c_uint64), ("cells", C.c_uint64), ("algo_bytes", C.c_uint64),
                ("ms", C.c_double), ("t0_ms", C.c_double)]


class PairingParams(C.Structure):
    _fields_ = [("min_factor_len", C.c_uint32), ("reserved", C.c_uint32),
                ("min_string_depth_rate", C.c_double)]


class Pairing(C.Structure):
    _fields_ = [("p", C.c_int32), ("t", C.c_int32), ("l", C.c_int32)]


assert C.sizeof(DpJob) == 48 and C.sizeof(DpResult) == 48

# every symbol include/pintron_gpu.h declares
EXPORTS = [
    "pgpu_init", "pgpu_destroy", "pgpu_last_error", "pgpu_abi_version", "pgpu_set_timing", "pgpu_device_numa_node",
    "pgpu_index_build", "pgpu_index_destroy", "pgpu_index_suffix_array", "pgpu_index_save", "pgpu_index_load", "pgpu_pairings",
    "pgpu_pairing_plan_create", "pgpu_pairing_plan_create_resident", "pgpu_pairing_plan_run", "pgpu_pairing_plan_count",
    "pgpu_pairing_plan_positions", "pgpu_pairing_plan_kernel_ms", "pgpu_pairing_plan_fetch",
    "pgpu_pairing_plan_destroy",
    "pgpu_pairing_plan_run_meg", "pgpu_pairing_plan_meg_bytes", "pgpu_pairing_plan_fetch_meg",
    "pgpu_pairing_plan_meg_ms", "pgpu_host_alloc", "pgpu_host_free",
    "pgpu_comm_unique_id", "pgpu_comm_init", "pgpu_gather", "pgpu_allgather", "pgpu_comm_destroy",
    "pgpu_range_push", "pgpu_range_pop", "pgpu_build_info",
    "pgpu_dp_plan_create", "pgpu_dp_plan_create_parts", "pgpu_dp_plan_launch", "pgpu_dp_plan_sync",
    "pgpu_dp_plan_string_bytes", "pgpu_dp_plan_fetch", "pgpu_dp_plan_destroy",
    "pgpu_dp_plan_results_to_device",
    "pgpu_dp_plan_cells", "pgpu_dp_plan_algo_bytes", "pgpu_dp_plan_kernel_ms",
    "pgpu_dp_plan_launches", "pgpu_dp_plan_n_groups", "pgpu_dp_plan_group_info", "pgpu_dp_batch",
]

_lib = None


class PgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pgpu error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load libpintron_gpu.so (no fallback: a missing library is an error)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (hipcc --offload-arch=gfx950)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, u64, sz = C.c_void_p, C.c_uint64, C.c_size_t
        L.pgpu_init.argtypes = [C.c_int, C.POINTER(vp)]
        L.pgpu_destroy.argtypes = [vp]
        L.pgpu_set_timing.argtypes = [vp, C.c_int]
        L.pgpu_last_error.argtypes = [vp]
        L.pgpu_last_error.restype = C.c_char_p
        L.pgpu_build_info.restype = C.c_char_p
        L.pgpu_range_push.argtypes = [C.c_char_p]
        L.pgpu_range_push.restype = None
        L.pgpu_range_pop.restype = None
        L.pgpu_allgather.argtypes = [vp, vp, vp, u64, vp]
        L.pgpu_index_build.argtypes = [vp, C.c_char_p, sz, C.POINTER(vp)]
        L.pgpu_index_destroy.argtypes = [vp, vp]
        L.pgpu_index_save.argtypes = [vp, vp, C.c_char_p, C.c_char_p]
        L.pgpu_index_load.argtypes = [vp, C.c_char_p, C.c_char_p, sz, C.POINTER(vp)]
        L.pgpu_index_suffix_array.argtypes = [vp, vp, C.POINTER(C.c_uint32), sz]
        L.pgpu_pairing_plan_run_meg.argtypes = [vp, vp, vp]
        L.pgpu_pairing_plan_meg_bytes.argtypes = [vp]
        L.pgpu_pairing_plan_meg_bytes.restype = u64
        L.pgpu_pairing_plan_meg_ms.argtypes = [vp]
        L.pgpu_pairing_plan_meg_ms.restype = C.c_double
        L.pgpu_pairing_plan_fetch_meg.argtypes = [vp, vp, vp, sz, C.POINTER(u64)]
        L.pgpu_host_alloc.argtypes = [vp, sz, C.POINTER(vp)]
        L.pgpu_host_free.argtypes = [vp, vp]
        L.pgpu_comm_unique_id.argtypes = [vp, vp]
        L.pgpu_comm_init.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
        L.pgpu_gather.argtypes = [vp, vp, C.c_char_p, u64, vp, u64, C.POINTER(u64)]
        L.pgpu_comm_destroy.argtypes = [vp, vp]
        L.pgpu_pairings.argtypes = [vp, vp, C.c_char_p, C.POINTER(u64), sz,
                                    C.POINTER(PairingParams), C.POINTER(Pairing), sz,
                                    C.POINTER(u64), C.POINTER(sz)]
        L.pgpu_pairing_plan_create.argtypes = [vp, vp, C.c_char_p, C.POINTER(u64), sz, C.POINTER(vp)]
        L.pgpu_pairing_plan_create_resident.argtypes = [vp, vp, C.c_char_p, C.POINTER(u64), sz, C.POINTER(vp)]
        L.pgpu_pairing_plan_run.argtypes = [vp, vp, C.POINTER(PairingParams)]
        L.pgpu_pairing_plan_count.argtypes = [vp]
        L.pgpu_pairing_plan_count.restype = u64
        L.pgpu_pairing_plan_positions.argtypes = [vp]
        L.pgpu_pairing_plan_positions.restype = u64
        L.pgpu_pairing_plan_kernel_ms.argtypes = [vp, C.c_int]
        L.pgpu_pairing_plan_kernel_ms.restype = C.c_double
        L.pgpu_pairing_plan_fetch.argtypes = [vp, vp, C.POINTER(Pairing), sz, C.POINTER(u64)]
        L.pgpu_pairing_plan_destroy.argtypes = [vp, vp]
        L.pgpu_dp_plan_create_parts.argtypes = [vp, vp, vp, sz, C.POINTER(vp)]
        L.pgpu_dp_plan_create.argtypes = [vp, vp, C.POINTER(DpJob), sz, C.c_char_p, sz,
                                          C.POINTER(vp)]
        L.pgpu_dp_plan_launch.argtypes = [vp, vp]
        L.pgpu_dp_plan_sync.argtypes = [vp, vp]
        L.pgpu_dp_plan_string_bytes.argtypes = [vp]
        L.pgpu_dp_plan_string_bytes.restype = sz
        L.pgpu_dp_plan_fetch.argtypes = [vp, vp, C.POINTER(DpResult), C.c_char_p, sz]
        L.pgpu_dp_plan_destroy.argtypes = [vp, vp]
        L.pgpu_dp_plan_results_to_device.argtypes = [vp, vp, vp, sz]
        for nm in ("pgpu_dp_plan_cells", "pgpu_dp_plan_algo_bytes", "pgpu_dp_plan_launches"):
            getattr(L, nm).argtypes = [vp, C.c_int]
            getattr(L, nm).restype = u64
        L.pgpu_dp_plan_kernel_ms.argtypes = [vp, C.c_int]
        L.pgpu_dp_plan_kernel_ms.restype = C.c_double
        L.pgpu_dp_plan_n_groups.argtypes = [vp]
        L.pgpu_dp_plan_group_info.argtypes = [vp, C.c_int, C.POINTER(GroupInfo)]
        L.pgpu_dp_batch.argtypes = [vp, vp, C.POINTER(DpJob), sz, C.c_char_p, sz,
                                    C.POINTER(DpResult), C.c_char_p, sz, C.POINTER(sz)]
        _lib = L
    return _lib


class Context:
    """One pgpu_ctx (one HIP stream on one device)."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        rc = self.L.pgpu_init(device, C.byref(self.h))
        if rc != PGPU_OK:
            raise PgpuError(rc, "pgpu_init(%d) failed (no usable gfx950 device?)" % device)
        self.L.pgpu_set_timing(self.h, 1)      # tests and bench read the per-kernel timings

    def check(self, rc):
        if rc != PGPU_OK:
            raise PgpuError(rc, self.L.pgpu_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.pgpu_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Index:
    def __init__(self, ctx: Context, genomic: bytes, load_from=None):
        """Built on the device, or (load_from=<path>) read from a file written by save()."""
        self.ctx = ctx
        self.h = C.c_void_p()
        self.n = len(genomic)
        self.genomic = genomic
        if load_from is None:
            ctx.check(ctx.L.pgpu_index_build(ctx.h, genomic, len(genomic), C.byref(self.h)))
        else:
            rc = ctx.L.pgpu_index_load(ctx.h, os.fsencode(load_from), genomic, len(genomic), C.byref(self.h))
            if rc != PGPU_OK:
                raise PgpuError(rc, "index file %s does not fit this sequence" % load_from)

    def save(self, path):
        self.ctx.check(self.ctx.L.pgpu_index_save(self.ctx.h, self.h, self.genomic, os.fsencode(path)))

    def suffix_array(self):
        import numpy as np
        sa = np.empty(self.n, dtype=np.uint32)
        self.ctx.check(self.ctx.L.pgpu_index_suffix_array(
            self.ctx.h, self.h, sa.ctypes.data_as(C.POINTER(C.c_uint32)), self.n))
        return sa

    def close(self):
        if self.h:
            self.ctx.L.pgpu_index_destroy(self.ctx.h, self.h)
            self.h = C.c_void_p()


class MegParams(C.Structure):       # pgpu_meg_params; defaults = src/options.ggo:94-370
    _fields_ = [("min_factor_len", C.c_uint32), ("min_intron_length", C.c_int32), ("max_intron_length", C.c_int32),
                ("max_pairings_in_MEG", C.c_uint32), ("max_prefix_discarded_rate", C.c_double),
                ("max_suffix_discarded_rate", C.c_double), ("max_freq_shortest_pairing", C.c_double),
                ("trans_red", C.c_uint32), ("short_edge_comp", C.c_uint32)]


def parse_meg_record(rec: bytes):
    """One record of pgpu_pairing_plan_fetch_meg -> dict(flags, vertices [(p,t,l)], adj [[targets]],
    meg_text, edges_text); vertices/adj are empty for an unavailable record."""
    import struct
    nv, ne, flags, _ = struct.unpack_from("<4I", rec, 0)
    if flags & 2:
        return dict(flags=flags, vertices=[], adj=[], meg_text=b"", edges_text=b"")
    verts = [struct.unpack_from("<3i", rec, 16 + 12 * k) for k in range(nv)]
    first = struct.unpack_from("<%dH" % (nv + 1), rec, 16 + 12 * nv)
    base = 16 + 12 * nv + 2 * (nv + 1)
    adj = [list(rec[base + first[k]:base + first[k + 1]]) for k in range(nv)]
    graph = (base + ne + 3) & ~3
    ml, el = struct.unpack_from("<2I", rec, graph)
    return dict(flags=flags, vertices=verts, adj=adj, meg_text=rec[graph + 8:graph + 8 + ml],
                edges_text=rec[graph + 8 + ml:graph + 8 + ml + el])


class PairingPlan:
    """Patterns resident in HBM; run() = all pairing kernels; fetch() -> (triples[n,3], first[n_pat+1]);
    run_meg()/fetch_meg() = the MEG stage behind the pairings."""
    STAGES = ["locate", "chain", "count+scan", "fill", "cross+scan", "emit"]

    def __init__(self, ctx: Context, index: Index, patterns, resident=False):
        import numpy as np
        self.ctx, self.n_pat = ctx, len(patterns)
        self._blob = b"".join(patterns)
        self._off = np.zeros(len(patterns) + 1, dtype=np.uint64)
        np.cumsum([len(p) for p in patterns], out=self._off[1:])
        self.h = C.c_void_p()
        # resident: the buffers a run writes are shared with the context's other resident plans (results valid
        # until another of them runs)
        create = ctx.L.pgpu_pairing_plan_create_resident if resident else ctx.L.pgpu_pairing_plan_create
        ctx.check(create(ctx.h, index.h, self._blob, self._off.ctypes.data_as(C.POINTER(C.c_uint64)),
                         self.n_pat, C.byref(self.h)))

    def run(self, min_factor_len=15, rate=0.2):
        prm = PairingParams(min_factor_len, 0, rate)
        self.ctx.check(self.ctx.L.pgpu_pairing_plan_run(self.ctx.h, self.h, C.byref(prm)))
        return self.ctx.L.pgpu_pairing_plan_count(self.h)

    def stage_ms(self):
        return {s: self.ctx.L.pgpu_pairing_plan_kernel_ms(self.h, k) for k, s in enumerate(self.STAGES)}

    def fetch(self):
        import numpy as np
        n = self.ctx.L.pgpu_pairing_plan_count(self.h)
        out = np.zeros((max(n, 1), 3), dtype=np.int32)
        first = np.zeros(self.n_pat + 1, dtype=np.uint64)
        self.ctx.check(self.ctx.L.pgpu_pairing_plan_fetch(
            self.ctx.h, self.h, out.ctypes.data_as(C.POINTER(Pairing)), n,
            first.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out[:n], first

    def run_meg(self, min_factor_len=15, min_intron_length=40, max_intron_length=0, max_pairings_in_MEG=80,
                max_prefix_discarded_rate=0.6, max_suffix_discarded_rate=0.6, max_freq_shortest_pairing=0.4,
                trans_red=True, short_edge_comp=True):
        prm = MegParams(min_factor_len, min_intron_length, max_intron_length, max_pairings_in_MEG,
                        max_prefix_discarded_rate, max_suffix_discarded_rate, max_freq_shortest_pairing,
                        int(trans_red), int(short_edge_comp))
        self.ctx.check(self.ctx.L.pgpu_pairing_plan_run_meg(self.ctx.h, self.h, C.byref(prm)))
        return self.ctx.L.pgpu_pairing_plan_meg_bytes(self.h)

    def fetch_meg(self):
        """list of raw records (bytes), one per pattern"""
        import numpy as np
        n = self.ctx.L.pgpu_pairing_plan_meg_bytes(self.h)
        out = np.zeros(max(n, 1), dtype=np.uint8)
        first = np.zeros(self.n_pat + 1, dtype=np.uint64)
        self.ctx.check(self.ctx.L.pgpu_pairing_plan_fetch_meg(
            self.ctx.h, self.h, out.ctypes.data_as(C.c_void_p), n, first.ctypes.data_as(C.POINTER(C.c_uint64))))
        raw = out.tobytes()
        return [raw[int(first[i]):int(first[i + 1])] for i in range(self.n_pat)]

    def close(self):
        if self.h:
            self.ctx.L.pgpu_pairing_plan_destroy(self.ctx.h, self.h)
            self.h = C.c_void_p()


class DpPart(C.Structure):
    _fields_ = [("jobs", C.c_void_p), ("n_jobs", C.c_size_t), ("arena", C.c_void_p), ("arena_len", C.c_size_t)]


class JobList:
    """Accumulates DP jobs and their operand bytes (the arena)."""

    def __init__(self):
        self.jobs = []
        self.chunks = []
        self.size = 0

    def _put(self, b: bytes) -> int:
        off = self.size
        self.chunks.append(b)
        self.size += len(b)
        return off

    def add(self, kind, a: bytes, b: bytes, p0=0, p1=0, p2=0, b_tail: bytes = b"",
            a_gen_off=None, b_gen_off=None, tail=None):
        """a/b are operand bytes; with *_gen_off the operand is the slice [off, off+len) of the
        resident genomic instead (a/b then only supply the length)."""
        flags = 0
        if a_gen_off is None:
            a_off = self._put(a)
        else:
            a_off, flags = a_gen_off, flags | JOB_A_GENOMIC
        if b_gen_off is None:
            b_off = self._put(b + b_tail)
        else:
            b_off, flags = b_gen_off, flags | JOB_B_GENOMIC
        self.jobs.append(DpJob(kind, flags, a_off, b_off, len(a), len(b), p0, p1, p2,
                               len(b_tail) if tail is None else tail))
        return len(self.jobs) - 1

    def arrays(self):
        arr = (DpJob * max(len(self.jobs), 1))(*self.jobs)
        return arr, b"".join(self.chunks)


class Plan:
    def __init__(self, ctx: Context, joblist: JobList, index: Index = None):
        self.ctx = ctx
        self.n = len(joblist.jobs)
        self._jobs, self._arena = joblist.arrays()
        self.h = C.c_void_p()
        ctx.check(ctx.L.pgpu_dp_plan_create(ctx.h, index.h if index else None, self._jobs, self.n,
                                            self._arena, len(self._arena), C.byref(self.h)))

    @classmethod
    def from_parts(cls, ctx: Context, joblists, index: Index = None):
        """One plan over several JobLists, each with its own arena (pgpu_dp_plan_create_parts)."""
        self = cls.__new__(cls)
        self.ctx = ctx
        self.n = sum(len(jl.jobs) for jl in joblists)
        self._keep = []
        parts = (DpPart * max(len(joblists), 1))()
        for k, jl in enumerate(joblists):
            jobs, arena = jl.arrays()
            buf = C.create_string_buffer(arena, len(arena) + 1)
            self._keep += [jobs, buf]
            parts[k] = DpPart(C.cast(jobs, C.c_void_p), len(jl.jobs), C.cast(buf, C.c_void_p), len(arena))
        self.h = C.c_void_p()
        ctx.check(ctx.L.pgpu_dp_plan_create_parts(ctx.h, index.h if index else None, parts, len(joblists), C.byref(self.h)))
        return self

    def launch(self):
        self.ctx.check(self.ctx.L.pgpu_dp_plan_launch(self.ctx.h, self.h))

    def sync(self):
        self.ctx.check(self.ctx.L.pgpu_dp_plan_sync(self.ctx.h, self.h))

    def fetch(self):
        L = self.ctx.L
        nbytes = L.pgpu_dp_plan_string_bytes(self.h)
        res = (DpResult * max(self.n, 1))()
        sbuf = C.create_string_buffer(max(nbytes, 1))
        self.ctx.check(L.pgpu_dp_plan_fetch(self.ctx.h, self.h, res, sbuf, nbytes))
        return res, sbuf.raw

    def results_to_device(self, device_ptr: int, cap: int):
        self.ctx.check(self.ctx.L.pgpu_dp_plan_results_to_device(self.ctx.h, self.h, device_ptr, cap))

    def groups(self):
        L = self.ctx.L
        out = []
        for i in range(L.pgpu_dp_plan_n_groups(self.h)):
            g = GroupInfo()
            L.pgpu_dp_plan_group_info(self.h, i, C.byref(g))
            out.append(dict(name=g.name.decode(), kind=g.kind, jobs=g.jobs, cells=g.cells,
                            algo_bytes=g.algo_bytes, ms=g.ms, t0_ms=g.t0_ms))
        return out

    def close(self):
        if self.h:
            self.ctx.L.pgpu_dp_plan_destroy(self.ctx.h, self.h)
            self.h = C.c_void_p()


def cstr(buf: bytes, off: int) -> bytes:
    end = buf.index(b"\0", off)
    return buf[off:end]


def decode(kind, r: DpResult, strings: bytes):
    """DpResult -> dict with the field names the oracle wrappers (tests/oracle_lib.py) use."""
    v = r.v
    if r.status != PGPU_OK:
        return dict(status=r.status)
    if kind == ALIGN:
        return dict(status=0, score=v[0], dim=v[1], ea=cstr(strings, r.str[0]),
                    ga=cstr(strings, r.str[1]))
    if kind == GAP:
        return dict(status=0, dim=v[0], factor_cut=v[1], intron_start=v[2], intron_end=v[3],
                    intron_start_on_align=v[4], intron_end_on_align=v[5],
                    ea=cstr(strings, r.str[0]), ga=cstr(strings, r.str[1]))
    if kind == ED:
        return dict(status=0, score=v[0])
    if kind == KBAND:
        return dict(status=0, ok=v[0], edit=v[1], dust=v[2])       # dust: the exon check's two flags (jobs with tail = 1)
    if kind == LCF:
        return dict(status=0, len=v[0], occ1=v[1], occ2=v[2])
    if kind == BORDERS:
        return dict(status=0, ok=v[0], off_p=v[1], off_t1=v[2], off_t2=v[3], ed=v[4])
    if kind == AFFIX:
        return dict(status=0, valid=v[0], ecut=v[1], gcut=v[2])
    raise ValueError(kind)


def run_jobs(ctx: Context, joblist: JobList, index: Index = None):
    """create + launch + sync + fetch + destroy; returns a list of decoded dicts."""
    plan = Plan(ctx, joblist, index)
    try:
        plan.launch()
        plan.sync()
        res, strings = plan.fetch()
        return [decode(joblist.jobs[i].kind, res[i], strings) for i in range(plan.n)]
    finally:
        plan.close()
