"""Seeded synthetic est-fact workloads: one genomic sequence with a planted
gene model and ESTs sampled from its transcripts.  Used by bench.py, the tests and the tools; it is
input generation only (numpy), not part of the accelerated path.

Config shapes (BASELINE.json `configs`):
  C2: 50 kb x 1 000 ESTs ~500 bp, 1 % errors      C3: 200 kb x 100 000 ESTs ~600 bp, 3 % errors
  C4: 8 genes x (200 kb x 62 500 ESTs ~600 bp, 3 % errors), gene g = seed 40 + g
  C5: 1 Mb x 2 000 000 ESTs of exactly 150 bp, 1 % errors
"""
from dataclasses import dataclass, field

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b


def revcomp(a: np.ndarray) -> np.ndarray:
    return _COMP[a[::-1]]


@dataclass
class Workload:
    name: str
    genomic: bytes
    exons: list                 # [(start, end)) on the genomic, ascending
    est_seqs: list              # list[bytes] as they would appear in ests.txt
    est_headers: list           # list[str]
    truth: list = field(default_factory=list)   # per EST: dict(tstart, tend, rc, exon_blocks)

    def genomic_fasta(self) -> str:
        return ">chrS:1:%d:+1\n%s\n" % (len(self.genomic), self.genomic.decode())

    def ests_fasta(self) -> str:
        return "".join("%s\n%s\n" % (h, s.decode()) for h, s in zip(self.est_headers, self.est_seqs))


CONFIGS = {
    "C2": dict(gen_len=50_000, n_est=1_000, est_len=500, est_sd=100, err=0.01, seed=2),
    "C3": dict(gen_len=200_000, n_est=100_000, est_len=600, est_sd=100, err=0.03, seed=3),
    # C4 = 8 genes of the C3 shape (gene g: seed 40 + g), 62 500 ESTs each = 500 000 in all
    "C4": dict(gen_len=200_000, n_est=62_500, est_len=600, est_sd=100, err=0.03, seed=40, genes=8),
    "C5": dict(gen_len=1_000_000, n_est=2_000_000, est_len=150, est_sd=0, err=0.01, seed=5),
}


def make(name="C3", n_est=None, seed=None, gen_len=None) -> Workload:
    cfg = dict(CONFIGS[name])
    if n_est is not None:
        cfg["n_est"] = n_est
    if seed is not None:
        cfg["seed"] = seed
    if gen_len is not None:
        cfg["gen_len"] = gen_len
    rng = np.random.default_rng(cfg["seed"])
    L = cfg["gen_len"]
    gen = _ACGT[rng.integers(0, 4, L)]

    # gene model: 8-12 exons of 80-300 bp, introns 500 bp .. 20 kb (scaled to fit), GT..AG ends
    n_ex = int(rng.integers(8, 13))
    ex_len = rng.integers(80, 301, n_ex)
    room = L - int(ex_len.sum()) - 2_000
    raw = rng.integers(500, 20_001, n_ex - 1).astype(np.float64)
    if raw.sum() > room:
        raw = np.maximum(500, raw * (room / raw.sum()) * 0.98)
    introns = raw.astype(np.int64)
    pos = int(rng.integers(500, max(501, L - int(ex_len.sum()) - int(introns.sum()) - 500)))
    exons = []
    for k in range(n_ex):
        exons.append((pos, pos + int(ex_len[k])))
        pos += int(ex_len[k])
        if k < n_ex - 1:
            donor = b"GC" if rng.random() < 0.02 else b"GT"
            gen[pos:pos + 2] = np.frombuffer(donor, dtype=np.uint8)
            pos += int(introns[k])
            gen[pos - 2:pos] = np.frombuffer(b"AG", dtype=np.uint8)

    # transcripts: the full one plus a few exon-skipping isoforms
    isoforms = [list(range(n_ex))]
    for _ in range(3):
        skip = int(rng.integers(1, n_ex - 1))
        isoforms.append([k for k in range(n_ex) if k != skip])
    tx = []
    for iso in isoforms:
        seq = np.concatenate([gen[exons[k][0]:exons[k][1]] for k in iso])
        bounds = np.cumsum([0] + [exons[k][1] - exons[k][0] for k in iso])
        tx.append((seq, iso, bounds))

    n = cfg["n_est"]
    which = rng.integers(0, len(tx), n)
    if cfg["est_sd"]:
        lens = np.clip(rng.normal(cfg["est_len"], cfg["est_sd"], n).astype(np.int64), 100, None)
    else:
        lens = np.full(n, cfg["est_len"], dtype=np.int64)
    is_rc = rng.random(n) < 0.5
    has_polya = rng.random(n) < 0.10
    polya_len = rng.integers(20, 41, n)

    est_seqs, headers, truth = [], [], []
    err = cfg["err"]
    for i in range(n):
        seq, iso, bounds = tx[int(which[i])]
        ln = int(min(lens[i], len(seq)))
        st = int(rng.integers(0, len(seq) - ln + 1))
        s = seq[st:st + ln].copy()
        # errors: 2/3 substitutions, 1/3 indels (half insertions, half deletions)
        r = rng.random(ln)
        sub = r < err * 2 / 3
        dele = (r >= err * 2 / 3) & (r < err * 5 / 6)
        ins = (r >= err * 5 / 6) & (r < err)
        s[sub] = _ACGT[(np.searchsorted(_ACGT, s[sub]) + rng.integers(1, 4, int(sub.sum()))) % 4]
        if rng.random() < 0.3:                      # <= 0.1 % N overall
            nn = rng.random(ln) < 0.003
            s[nn] = ord("N")
        rep = np.ones(ln, dtype=np.int64)
        rep[dele] = 0
        rep[ins] = 2
        out = np.repeat(s, rep)
        if ins.any():
            ends = np.cumsum(rep)[ins] - 1          # second copy of each duplicated base
            out[ends] = _ACGT[rng.integers(0, 4, len(ends))]
        if has_polya[i]:
            out = np.concatenate([out, np.full(int(polya_len[i]), ord("A"), dtype=np.uint8)])
        rc = bool(is_rc[i])
        if rc:
            out = revcomp(out)
        est_seqs.append(out.tobytes())
        headers.append(">/gb=SYN%07d /clone_end=%s" % (i, "5'" if rc else "3'"))
        # ground truth exon blocks (genomic coordinates) of the error-free EST
        blocks = []
        for bi, k in enumerate(iso):
            lo, hi = max(st, int(bounds[bi])), min(st + ln, int(bounds[bi + 1]))
            if lo < hi:
                g0 = exons[k][0] + (lo - int(bounds[bi]))
                blocks.append((lo - st, hi - st, g0, g0 + (hi - lo)))
        truth.append(dict(rc=rc, blocks=blocks, polya=bool(has_polya[i])))
    return Workload(name, gen.tobytes(), exons, est_seqs, headers, truth)


def write_files(w: Workload, directory: str) -> None:
    import os
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "genomic.txt"), "w") as f:
        f.write(w.genomic_fasta())
    with open(os.path.join(directory, "ests.txt"), "w") as f:
        f.write(w.ests_fasta())


def make_long_transcripts(seed=77):
    """A gene with a 5.6 kb last exon and full-length transcripts of it (RefSeq-like input): DP
    matrices with more than 4096 rows.  Returns (genomic_fasta_text, ests_fasta_text)."""
    import random
    rng = random.Random(seed)

    def rs(n):
        return "".join(rng.choice("ACGT") for _ in range(n))

    def mut(s, rate):
        out = []
        for c in s:
            x = rng.random()
            if x < rate * 0.6:
                out.append(rng.choice("ACGT"))
            elif x < rate * 0.8:
                pass
            elif x < rate:
                out.append(c)
                out.append(rng.choice("ACGT"))
            else:
                out.append(c)
        return "".join(out)

    exons = [rs(350), rs(420), rs(5600)]
    introns = ["GT" + rs(1500) + "AG", "GT" + rs(2500) + "AG"]
    gen = rs(3000) + exons[0] + introns[0] + exons[1] + introns[1] + exons[2] + rs(3000)
    tr = "".join(exons)
    ests = [mut(tr, 0.01), mut(tr[200:], 0.02), mut(tr[:3000], 0.01), mut(tr[500:6300], 0.03) + "A" * 30]
    for _ in range(20):
        a = rng.randint(0, len(tr) - 700)
        ests.append(mut(tr[a:a + rng.randint(400, 700)], 0.02))
    g = ">chrL:1:%d:+1\n%s\n" % (len(gen), gen)
    e = "".join(">/gb=LNG%04d /clone_end=3'\n%s\n" % (i, x) for i, x in enumerate(ests))
    return g, e


def make_small_exons(seed=11, n_est=240):
    """Genes with exons of 8 - 22 bases between long ones, Ns in the ESTs AND in the genomic sequence (some of them
    inside a small exon and at the same place in the ESTs), soft-masked (lower-case) stretches in the introns:
    what search_for_new_small_exons / remove_false_small_exons (src/factorization-refinement.c:641-1125) look at,
    including the patterns with a byte that is no upper-case ACGT.  Returns (genomic_fasta_text, ests_fasta_text)."""
    import random
    rng = random.Random(seed)

    def rs(n):
        return "".join(rng.choice("ACGT") for _ in range(n))

    def mut(s, rate, n_rate):
        out = []
        for c in s:
            x = rng.random()
            if c == "N":
                out.append(c)
            elif x < n_rate:
                out.append("N")
            elif x < n_rate + rate * 0.6:
                out.append(rng.choice("ACGT"))
            elif x < n_rate + rate * 0.8:
                pass
            elif x < n_rate + rate:
                out.append(c)
                out.append(rng.choice("ACGT"))
            else:
                out.append(c)
        return "".join(out)

    def intron(n):
        body = rs(n)
        if rng.random() < 0.5:                       # a soft-masked repeat and a few Ns inside
            a = rng.randint(0, n // 2)
            body = body[:a] + body[a:a + n // 4].lower() + body[a + n // 4:]
        if rng.random() < 0.5:
            a = rng.randint(0, n - 10)
            body = body[:a] + "NNN" + body[a + 3:]
        return ("GC" if rng.random() < 0.1 else "GT") + body + "AG"

    gen = rs(1500)
    transcripts = []
    for _gene in range(4):
        exons = []
        for k in range(5):
            if k in (1, 3):
                x = rs(rng.randint(8, 22))
                if rng.random() < 0.5:                # an N of the sequence inside the small exon
                    a = rng.randint(2, len(x) - 3)
                    x = x[:a] + "N" + x[a + 1:]
            else:
                x = rs(rng.randint(120, 260))
            exons.append(x)
        tr = ""
        for k, x in enumerate(exons):
            gen += x
            tr += x
            if k + 1 < len(exons):
                gen += intron(rng.randint(300, 2500))
        gen += rs(800)
        transcripts.append(tr)
    ests = []
    for i in range(n_est):
        tr = transcripts[i % len(transcripts)]
        a = rng.randint(0, max(0, len(tr) - 450))
        piece = tr[a:a + rng.randint(300, 600)]
        x = mut(piece, rng.choice((0.0, 0.01, 0.03, 0.05)), rng.choice((0.0, 0.002, 0.01)))
        if rng.random() < 0.4:
            x = str(_rc_text(x))
        ests.append(x)
    g = ">chrS:1:%d:+1\n%s\n" % (len(gen), gen)
    e = "".join(">/gb=SMX%04d /clone_end=3'\n%s\n" % (i, x) for i, x in enumerate(ests))
    return g, e


def _rc_text(s):
    return s[::-1].translate(str.maketrans("ACGTacgtN", "TGCAtgcaN"))


def make_region_start_repeats(seed=5):
    """Transcripts that begin with the very first bases of the genomic region (pairings with t == 0,
    the occurrence without a preceding character: DESIGN.md section 4b), and repeats of those first
    bases elsewhere in the region with other contexts, so that the t == 0 occurrence is reported at
    an upper level of the reference's suffix tree while a longer match lies elsewhere.
    Returns (genomic_fasta_text, ests_fasta_text)."""
    import random
    w = make("C2", n_est=10)
    G = bytearray(w.genomic)
    rng = random.Random(seed)
    spots = ((3000, 60), (9000, 45), (12000, 80), (15000, 30))
    for pos, n in spots:
        G[pos:pos + n] = G[0:n]
    G = bytes(G)
    w.genomic = G

    def rc(s):
        return s[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))

    def mut(s, r):
        b = bytearray(s)
        for i in range(len(b)):
            if rng.random() < r:
                b[i] = rng.choice(b"ACGT")
        return bytes(b)

    e1s, e1e = w.exons[1]
    extra = [G[0:350], G[0:180] + G[e1s:e1e][:150], rc(G[0:260] + G[e1s:e1e][:90]), G[0:25] + G[e1s:e1e],
             G[0:80] + G[3060:3300]]
    for pos, n in spots:
        extra += [G[pos:pos + 300], G[pos - 100:pos + 250], G[pos:pos + n] + G[20000:20200], rc(G[pos:pos + 280])]
    extra += [mut(x, 0.02) for x in extra]
    seqs, heads = list(w.est_seqs), list(w.est_headers)
    for k, x in enumerate(extra):
        seqs.append(x)
        heads.append(">/gb=T0%05d /clone_end=3'" % k)
    w.est_seqs, w.est_headers = seqs, heads
    return w.genomic_fasta(), w.ests_fasta()


def make_region_start_copies(seed=7):
    """ESTs for which the reference lists the pairing with t == 0 more than once (the occurrence without
    a preceding character sits in every symbol slice of the reference's suffix tree and is reported per
    slice at an upper tree level, src/max-emb-graph.c:168-216): exon-like blocks elsewhere in the
    region begin with a copy of the region's first 40-70 bases behind a varying character, and
    transcripts start (after 0, 1 or 30 unrelated bases) on such a copy.  The repeated vertex goes
    through MEG construction, simplification and the writers.  Returns (genomic, ests) FASTA texts."""
    import random
    w = make("C2", n_est=6, seed=seed)
    G = bytearray(w.genomic)
    rng = random.Random(seed)
    spots = []
    for k, pos in enumerate((2500, 6000, 11000, 16000, 21000, 26000, 31000, 36000)):
        n = (40, 55, 70, 45, 60, 50, 65, 42)[k]
        G[pos - 1] = b"ACGT"[k % 4]
        G[pos:pos + n] = G[0:n]
        spots.append((pos, n))
    G = bytes(G)
    w.genomic = G

    def rc(s):
        return s[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))

    def junk(n):
        return bytes(rng.choice(b"ACGT") for _ in range(n))

    e1s, e1e = w.exons[1]
    seqs, heads = list(w.est_seqs), list(w.est_headers)
    k = 0
    for pos, n in spots:
        for lead in (0, 1, 30):
            for tail in (G[pos + n:pos + n + 120], G[pos + n:pos + n + 90] + G[e1s:e1e][:120]):
                x = junk(lead) + G[pos:pos + n] + tail
                for y in (x, rc(x)):
                    seqs.append(y)
                    heads.append(">/gb=TC%05d /clone_end=3'" % k)
                    k += 1
    w.est_seqs, w.est_headers = seqs, heads
    return w.genomic_fasta(), w.ests_fasta()


def make_edge_cases(seed=5):
    """Inputs around the corners of the input handling: N tails and internal N runs in the genomic
    sequence, negative-strand header, ESTs that are too short / all N / lower case / polyA only /
    unrelated, duplicated headers, /fixed_strand, RefSeq and token-less headers, exon skipping, an
    EST read off the genomic flank, wrapped lines.  Returns (genomic_fasta_text, ests_fasta_text)."""
    import random
    rng = random.Random(seed)

    def rs(n):
        return "".join(rng.choice("ACGT") for _ in range(n))

    def rc(s):
        return s[::-1].translate(str.maketrans("ACGTacgtNn", "TGCAtgcaNn"))

    ex = [rs(200), rs(150), rs(300)]
    gen = ("NNNNNNNNNN" + rs(500) + ex[0] + "GT" + rs(700) + "AG" + ex[1] + "GT" + rs(50) + "N" * 20 + rs(600) +
           "AG" + ex[2] + rs(400) + "NNNNN")
    tr = "".join(ex)
    ests = [
        (">/gb=E0001 /clone_end=3'", tr[:10]),
        (">/gb=E0002 /clone_end=3'", "N" * 120),
        (">/gb=E0003 /clone_end=5'", tr.lower()),
        (">/gb=E0004 /clone_end=3'", tr),
        (">/gb=E0004 /clone_end=3'", tr[50:500]),
        (">/gb=E0005 /clone_end=3'", "A" * 80),
        (">/gb=E0006 /clone_end=3' /fixed_strand=1", rc(tr)),
        (">/gb=E0007 /clone_end=3' /fixed_strand=0", rc(tr)),
        (">NM_000001.1 some refseq", tr + "A" * 25),
        (">/gb=E0008", rc(tr[20:600])),
        (">weird header without tokens", tr[100:640]),
        (">/gb=E0009 /clone_end=5'", "T" * 30 + rc(tr)[:500]),
        (">/gb=E0010 /clone_end=3'", tr[:300] + "N" * 5 + tr[305:]),
        (">/gb=E0011 /clone_end=3'", tr[:180] + tr[360:]),
        (">/gb=E0012 /clone_end=3'", rs(400)),
        (">/gb=E0013 /clone_end=3'", gen[10:400]),
        (">/gb=E0014 /clone_end=3'", ex[0][:14]),
        (">/gb=E0015 /clone_end=3'", ex[0][:15] + ex[1][:15]),
    ]
    g = ">chrE:1000:%d:-1\n" % (999 + len(gen)) + "\n".join(gen[i:i + 70] for i in range(0, len(gen), 70)) + "\n"
    e = "".join(h + "\n" + "\n".join(x[i:i + 60] for i in range(0, len(x), 60)) + "\n" for h, x in ests)
    return g, e
