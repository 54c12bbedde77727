"""Direct ctypes calls into the compiled reference (oracle/_ref/libpintron_ref_core.so), used to pin
the oracle.  TEST INFRASTRUCTURE; only usable where oracle/_ref has been built."""
import ctypes as C

import oracle_lib as O


class _Alignment(C.Structure):      # include/types.h:221-227
    _fields_ = [("EST_alignment", C.c_char_p), ("GEN_alignment", C.c_char_p),
                ("alignment_dim", C.c_int), ("score", C.c_int)]


class _GapAlignment(C.Structure):   # include/types.h:229-256
    _fields_ = [("EST_gap_alignment", C.c_char_p), ("GEN_gap_alignment", C.c_char_p),
                ("gap_alignment_dim", C.c_int), ("factor_cut", C.c_int),
                ("intron_start", C.c_int), ("intron_end", C.c_int),
                ("intron_start_on_align", C.c_int), ("intron_end_on_align", C.c_int)]


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _R():
    R = O.ref()
    if not getattr(R, "_typed", False):
        R.compute_alignment.restype = C.c_void_p
        R.compute_alignment.argtypes = [C.c_char_p, C.c_char_p, C.c_bool]
        R.compute_gap_alignment.restype = C.c_void_p
        R.compute_gap_alignment.argtypes = [C.c_char_p, C.c_char_p, C.c_bool, C.c_int, C.c_int, C.c_int]
        R.list_head.restype = C.c_void_p
        R.list_head.argtypes = [C.c_void_p]
        R.edit_distance.restype = C.POINTER(C.c_uint)
        R.edit_distance.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        R.compute_edit_distance.restype = C.c_size_t
        R.compute_edit_distance.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        R.K_band_edit_distance.restype = C.c_bool
        R.K_band_edit_distance.argtypes = [C.c_char_p, C.c_char_p, C.c_uint, C.POINTER(C.c_uint)]
        R.general_refine_borders.restype = C.c_bool
        R.general_refine_borders.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t,
                                             C.c_char_p, C.c_size_t, C.c_uint] + \
            [C.POINTER(C.c_size_t)] * 3 + [C.POINTER(C.c_uint)]
        R.getBursetFrequency.restype = C.c_int
        R.getBursetFrequency.argtypes = [C.c_char_p, C.c_char_p]
        for nm in ("compute_best_suffix_cut", "compute_best_prefix_cut"):
            f = getattr(R, nm)
            f.restype = C.c_size_t
            f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t,
                          C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        R._typed = True
    return R


def align(a, b):
    R = _R()
    lst = R.compute_alignment(a, b, True)
    al = C.cast(R.list_head(lst), C.POINTER(_Alignment)).contents
    return dict(score=al.score, dim=al.alignment_dim, ea=al.EST_alignment, ga=al.GEN_alignment)


def gap_align(a, b):
    R = _R()
    lst = R.compute_gap_alignment(a, b, True, 0, 0, 0)
    g = C.cast(R.list_head(lst), C.POINTER(_GapAlignment)).contents
    return dict(dim=g.gap_alignment_dim, factor_cut=g.factor_cut, intron_start=g.intron_start,
                intron_end=g.intron_end, intron_start_on_align=g.intron_start_on_align,
                intron_end_on_align=g.intron_end_on_align, ea=g.EST_gap_alignment,
                ga=g.GEN_gap_alignment)


def edit_distance_last(a, b):
    R = _R()
    M = R.edit_distance(a, len(a), b, len(b))
    v = M[(len(a) + 1) * (len(b) + 1) - 1]
    _libc.free(M)
    return v


def compute_edit_distance(a, b):
    return _R().compute_edit_distance(a, len(a), b, len(b))


def kband(a, b, ub):
    e = C.c_uint(0)
    ok = _R().K_band_edit_distance(a, b, ub, C.byref(e))
    return dict(ok=int(ok), edit=e.value)


def refine_borders(p, t, lo, hi, max_errs, t_tail=b""):
    op, o1, o2, ed = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_uint()
    buf = t + t_tail + b"\0\0"
    ok = _R().general_refine_borders(p, len(p), lo, hi, buf, len(t), max_errs, C.byref(op),
                                     C.byref(o1), C.byref(o2), C.byref(ed))
    return dict(ok=int(ok), off_p=op.value, off_t1=o1.value, off_t2=o2.value, ed=ed.value)


def burset(d, a):
    return _R().getBursetFrequency(C.create_string_buffer(d), C.create_string_buffer(a))


def suffix_cut(a, b, prefix=False):
    c1, c2 = C.c_size_t(), C.c_size_t()
    f = _R().compute_best_prefix_cut if prefix else _R().compute_best_suffix_cut
    ed = f(a, len(a), b, len(b), C.byref(c1), C.byref(c2))
    return ed, c1.value, c2.value


def lcf(a, b):
    S = O.ref_static()
    S.ref_static_lcf.restype = None
    S.ref_static_lcf.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t] + [C.POINTER(C.c_size_t)] * 3
    o1, o2, ln = C.c_size_t(), C.c_size_t(), C.c_size_t()
    S.ref_static_lcf(a, len(a), b, len(b), C.byref(o1), C.byref(o2), C.byref(ln))
    return dict(occ1=o1.value, occ2=o2.value, len=ln.value)


def longest_affix(a, b):
    S = O.ref_static()
    S.ref_static_longest_affix.restype = C.c_int
    S.ref_static_longest_affix.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t] + [C.POINTER(C.c_size_t)] * 2
    e, g = C.c_size_t(0), C.c_size_t(0)
    v = S.ref_static_longest_affix(a, len(a), b, len(b), C.byref(e), C.byref(g))
    return dict(valid=v, ecut=e.value if v else 0, gcut=g.value if v else 0)
