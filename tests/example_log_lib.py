"""TEST INFRASTRUCTURE: sizes of the MEGs in a megs.txt, and their comparison with the sizes a
reference-held pipeline log reports (tools/pin_example_log.py)."""
import collections


def meg_sizes(megs_path):
    """FASTA header -> [(vertices, edges), ...] of every MEG block of megs.txt (src/io-meg.c:146-190)"""
    out = collections.defaultdict(list)
    for b in open(megs_path).read().split("\n\n***********\n\n")[1:]:
        lines = b.split("\n")
        v, e = "\n".join(lines[2:]).split("#adj#\n")
        out[lines[0][1:]].append((v.count("("), e.count("-")))
    return out


def not_reproduced(logged, ours):
    """logged sizes (header -> [[V, E], ...]) that are not among our MEGs of the same header (multiset)"""
    miss = []
    for h in sorted(logged):
        have = collections.Counter(ours.get(h, []))
        for s in logged[h]:
            if have[tuple(s)] > 0:
                have[tuple(s)] -= 1
            else:
                miss.append([h, list(s)])
    return miss
