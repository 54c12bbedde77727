"""ctypes bindings for the CPU oracle (oracle/liboracle_dp.so) and, when present, the compiled
reference (oracle/_ref/*.so).  TEST INFRASTRUCTURE: imported only from tests/, smoke() and
bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


class GapResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "factor_cut", "intron_start", "intron_end",
                                         "intron_start_on_align", "intron_end_on_align",
                                         "start_matrix", "score")]


class BordersResult(C.Structure):
    _fields_ = [("offset_p", C.c_uint32), ("offset_t1", C.c_uint32), ("offset_t2", C.c_uint32),
                ("edit_distance", C.c_uint32), ("ok", C.c_int32)]


_lib = None


def oracle():
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle_dp.so")
        if not os.path.exists(path):
            build_oracle()
        _lib = C.CDLL(path)
        L = _lib
        L.orc_align.restype = C.c_uint32
        L.orc_align.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p,
                                C.c_char_p, C.POINTER(C.c_int32)]
        L.orc_edit_distance.restype = C.c_uint32
        L.orc_edit_distance.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_kband.restype = C.c_int
        L.orc_kband.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_uint32,
                                C.POINTER(C.c_uint32)]
        L.orc_gap_align.restype = None
        L.orc_gap_align.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p,
                                    C.c_char_p, C.POINTER(GapResult)]
        L.orc_lcf.restype = None
        L.orc_lcf.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t] + \
            [C.POINTER(C.c_uint32)] * 3
        L.orc_burset_frequency.restype = C.c_int
        L.orc_burset_frequency.argtypes = [C.c_char_p, C.c_char_p]
        L.orc_burset_adaptor.restype = C.c_int
        L.orc_burset_adaptor.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t]
        L.orc_refine_borders.restype = None
        L.orc_refine_borders.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                         C.c_char_p, C.c_size_t, C.c_uint32,
                                         C.POINTER(BordersResult)]
        L.orc_longest_affix.restype = C.c_int
        L.orc_longest_affix.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        for nm in ("orc_best_suffix_cut", "orc_best_prefix_cut"):
            fn = getattr(L, nm)
            fn.restype = C.c_uint32
            fn.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                           C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    return _lib


# ---- python-level wrappers (bytes in, plain tuples/dicts out) ------------------------------

def align(a: bytes, b: bytes):
    L = oracle()
    ea = C.create_string_buffer(len(a) + len(b) + 1)
    ga = C.create_string_buffer(len(a) + len(b) + 1)
    dim = C.c_int32()
    score = L.orc_align(a, len(a), b, len(b), ea, ga, C.byref(dim))
    return dict(score=score, dim=dim.value, ea=ea.value, ga=ga.value)


def edit_distance(a: bytes, b: bytes) -> int:
    return oracle().orc_edit_distance(a, len(a), b, len(b))


def dust_flags(gen: bytes, est: bytes, threshold: float) -> int:
    L = oracle()
    L.orc_dust_flags.restype = C.c_uint32
    L.orc_dust_flags.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_double]
    return int(L.orc_dust_flags(gen, len(gen), est, len(est), threshold))


def dust_score(s: bytes) -> float:
    L = oracle()
    L.orc_dust_score.restype = C.c_double
    L.orc_dust_score.argtypes = [C.c_char_p, C.c_size_t]
    return float(L.orc_dust_score(s, len(s)))


def kband(a: bytes, b: bytes, ub: int):
    e = C.c_uint32()
    ok = oracle().orc_kband(a, len(a), b, len(b), ub, C.byref(e))
    return dict(ok=ok, edit=e.value)


def gap_align(a: bytes, b: bytes):
    L = oracle()
    ea = C.create_string_buffer(len(a) + len(b) + 1)
    ga = C.create_string_buffer(len(a) + len(b) + 1)
    r = GapResult()
    L.orc_gap_align(a, len(a), b, len(b), ea, ga, C.byref(r))
    d = {n: getattr(r, n) for n, _ in GapResult._fields_}
    d.update(ea=ea.value, ga=ga.value)
    return d


def lcf(a: bytes, b: bytes):
    o1, o2, ln = C.c_uint32(), C.c_uint32(), C.c_uint32()
    oracle().orc_lcf(a, len(a), b, len(b), C.byref(o1), C.byref(o2), C.byref(ln))
    return dict(occ1=o1.value, occ2=o2.value, len=ln.value)


def refine_borders(p: bytes, t: bytes, min_cut: int, max_cut: int, max_errs: int,
                   t_tail: bytes = b""):
    """t_tail: up to two bytes that follow t in the caller's buffer (see dp_capture_shim.c)."""
    r = BordersResult()
    buf = t + t_tail + b"\0\0"
    oracle().orc_refine_borders(p, len(p), min_cut, max_cut, buf, len(t), max_errs, C.byref(r))
    return dict(off_p=r.offset_p, off_t1=r.offset_t1, off_t2=r.offset_t2, ed=r.edit_distance,
                ok=r.ok)


def longest_affix(est: bytes, gen: bytes):
    e, g = C.c_uint32(0), C.c_uint32(0)
    v = oracle().orc_longest_affix(est, len(est), gen, len(gen), C.byref(e), C.byref(g))
    return dict(valid=v, ecut=e.value, gcut=g.value)


# ---- compiled reference (only where oracle/_ref has been built) ----------------------------

def have_ref() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "libpintron_ref_core.so"))


_ref = None
_ref_static = None


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(os.path.join(REF_DIR, "libpintron_ref_core.so"), mode=C.RTLD_GLOBAL)
    return _ref


def ref_static():
    global _ref_static
    if _ref_static is None:
        ref()
        _ref_static = C.CDLL(os.path.join(REF_DIR, "libref_harness.so"))
    return _ref_static
