"""Pairing (MEG vertex set) helpers for the tests: oracle + compiled-reference bindings and
seeded inputs with planted exact repeats (the case where the reference's suffix-link descent
changes the threshold, SURVEY.md section 7 hard part 1)."""
import ctypes as C
import os
import random

import numpy as np

import oracle_lib as O

COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def revcomp(s: bytes) -> bytes:
    return s.translate(COMP)[::-1]


def read_fasta(path):
    seqs, cur = [], []
    for ln in open(path):
        ln = ln.strip()
        if ln.startswith(">"):
            if cur:
                seqs.append("".join(cur))
                cur = []
        else:
            cur.append(ln)
    if cur:
        seqs.append("".join(cur))
    return [s.encode() for s in seqs]


class OracleIndex:
    def __init__(self, genomic: bytes):
        L = O.oracle()
        L.orc_index_create.restype = C.c_void_p
        L.orc_index_create.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_index_destroy.argtypes = [C.c_void_p]
        L.orc_index_sa.restype = C.POINTER(C.c_uint32)
        L.orc_index_sa.argtypes = [C.c_void_p]
        L.orc_pairings.restype = C.c_long
        L.orc_pairings.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_double,
                                   C.c_void_p, C.c_long, C.c_void_p]
        self.L, self.n = L, len(genomic)
        self.h = L.orc_index_create(genomic, len(genomic))
        self.buf = np.zeros(3 * 400000, dtype=np.int32)

    def sa(self):
        return np.ctypeslib.as_array(self.L.orc_index_sa(self.h), shape=(self.n,)).copy()

    def pairings(self, est: bytes, min_factor_len=15, rate=0.2):
        n = self.L.orc_pairings(self.h, est, len(est), min_factor_len, rate, self.buf.ctypes.data,
                                len(self.buf) // 3, None)
        assert n <= len(self.buf) // 3
        return self.buf[:3 * n].reshape(-1, 3).copy()

    def close(self):
        self.L.orc_index_destroy(self.h)


class RefIndex:
    """The reference's own suffix tree + build_vertex_set (oracle/ref_pairing_harness.c)."""

    def __init__(self, genomic: bytes):
        R = O.ref_static()   # oracle/ref_pairing_harness.c, linked against the core library
        R.ref_index_create.restype = C.c_void_p
        R.ref_index_create.argtypes = [C.c_char_p]
        R.ref_build_pairings.restype = C.c_long
        R.ref_build_pairings.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.c_double, C.c_void_p, C.c_long]
        self.R = R
        self.h = self._quiet(lambda: R.ref_index_create(genomic))
        self.buf = np.zeros(3 * 400000, dtype=np.int32)

    @staticmethod
    def _quiet(fn):
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(2)
        os.dup2(devnull, 2)
        try:
            return fn()
        finally:
            os.dup2(saved, 2)
            os.close(devnull)
            os.close(saved)

    def pairings(self, est: bytes, min_factor_len=15, rate=0.2):
        n = self._quiet(lambda: self.R.ref_build_pairings(self.h, est, min_factor_len, rate,
                                                          self.buf.ctypes.data, len(self.buf) // 3))
        return self.buf[:3 * n].reshape(-1, 3).copy()


def repeat_workload(seed, gen_len=30000):
    """Genomic with planted exact repeats (40..500 bp, some sharing the left context), a
    dinucleotide and a homopolymer stretch; ESTs spanning the repeats with a few substitutions,
    plus ESTs starting at genomic position 0."""
    rng = random.Random(seed)

    def rs(n):
        return [rng.choice("ACGT") for _ in range(n)]

    g = rs(gen_len)
    reps = []
    for _ in range(12):
        ln = rng.choice([40, 80, 100, 150, 300, 500])
        src = rng.randint(100, len(g) - ln - 100)
        seg = g[src:src + ln]
        for _ in range(rng.randint(1, 3)):
            dst = rng.randint(100, len(g) - ln - 100)
            g[dst:dst + ln] = seg
            if rng.random() < 0.5:
                g[dst - 1] = g[src - 1]
        reps.append((src, ln))
    p = rng.randint(1000, gen_len - 2000)
    g[p:p + 120] = list("AC" * 60)
    p = rng.randint(1000, gen_len - 2000)
    g[p:p + 90] = list("A" * 90)
    gen = "".join(g).encode()
    ests = []
    for src, ln in reps:
        for _ in range(4):
            a = max(0, src - rng.randint(0, 200))
            b = min(len(gen), src + ln + rng.randint(0, 200))
            e = bytearray(gen[a:b])
            for _ in range(rng.randint(0, 4)):
                e[rng.randrange(len(e))] = rng.choice(b"ACGT")
            ests.append(bytes(e))
    for _ in range(20):
        a = rng.randint(0, len(gen) - 700)
        ests.append(gen[a:a + rng.randint(100, 700)])
    ests += [gen[:300], gen[5:200], b"ACAC" * 40, b"A" * 100, b"ACGTNNNNACGT" * 10, b"A", b""]
    return gen, ests


def region_start_cases(seed, n_cases=60):
    """Small genomic sequences whose first bases are repeated elsewhere behind another character, and
    ESTs that match the repeat for longer than they match the region start: the occurrence t == 0 is
    then reported at an upper level of the reference's suffix tree, once per symbol slice it walks
    (oracle/pairing_oracle.c) -- about half of the cases hold the pairing (i, 0, l) twice.  Alphabets
    of 2..5 symbols, the preceding EST character present, absent (i == 0) or foreign ('N').
    Returns [(genomic, [est, ...]), ...]."""
    rng = random.Random(seed)

    def rs(n, al=b"ACGT"):
        return bytes(rng.choice(al) for _ in range(n))

    out = []
    for _ in range(n_cases):
        al = rng.choice([b"ACGT", b"ACGT", b"ACGT", b"ACGTN", b"AC", b"ACG"])
        head, x = rs(60, al), rs(120, al)
        rep = rng.choice([20, 40, 55])
        gen = head + rs(rng.choice([5, 300]), al) + rs(1, al) + head[:rep] + x + rs(300, al)
        if rng.random() < 0.3:
            gen += head[:35] + rs(50, al)
        ests = []
        for lead in (0, 1, 30):
            est = rs(lead, b"ACGTN") + head[:rep] + x[:rng.choice([60, 100])]
            if rng.random() < 0.3 and len(est) > 60:
                est = est[:50] + rs(1, al) + est[51:]
            ests.append(est)
        out.append((gen, ests))
    return out
