#!/usr/bin/env python3
"""Replays a PINTRON_DP_TRACE file (pintron_amd/host/ef_gpu_backend.c: every DP request est-fact asked
with the answer it got) through the CPU oracle and lists the answers that differ.

A parity hunt splits here: a differing answer = a kernel (or the library around it) gave a wrong result
for that job; no differing answer while the program's output is wrong = the host logic or the scheduler
(an answer delivered to the wrong fibre shows up as a differing answer too: its operands are the asker's).

TEST INFRASTRUCTURE (uses oracle/): python tools/replay_dp_trace.py TRACE [--max-report N] [--json OUT]"""
import argparse
import json
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O          # noqa: E402
from dp_cases import KIND_NAMES, ALIGN, GAP, ED, KBAND, LCF, BORDERS, AFFIX   # noqa: E402

MAGIC = 0x31545044


def records(path):
    with open(path, "rb") as f:
        data = f.read()
    pos, n = 0, len(data)
    while pos + 68 <= n:
        magic, unit, kind, la, lb, p0, p1, p2, tail = struct.unpack_from("<9I", data, pos)
        if magic != MAGIC:
            raise SystemExit("bad record at byte %d" % pos)
        v = struct.unpack_from("<6i", data, pos + 36)
        n0, n1 = struct.unpack_from("<2I", data, pos + 60)
        pos += 68
        t2 = min(tail, 2)
        if pos + la + lb + t2 + n0 + n1 > n:
            break                                   # a truncated last record (the program was killed)
        a = data[pos:pos + la]; pos += la
        b = data[pos:pos + lb]; pos += lb
        bt = data[pos:pos + t2]; pos += t2
        s0 = data[pos:pos + n0]; pos += n0
        s1 = data[pos:pos + n1]; pos += n1
        yield dict(unit=unit, kind=kind, a=a, b=b, b_tail=bt, p=(p0, p1, p2), tail=tail, v=v, s0=s0, s1=s1)


def expected(r):
    """what the host reads of the answer (ef_decode_result + the callers): (values, s0, s1)"""
    k, a, b = r["kind"], r["a"], r["b"]
    if k == ALIGN:
        e = O.align(a, b)
        return {0: e["score"], 1: e["dim"]}, e["ea"], e["ga"]
    if k == GAP:
        e = O.gap_align(a, b)
        return {0: e["dim"], 1: e["factor_cut"], 2: e["intron_start"], 3: e["intron_end"],
                4: e["intron_start_on_align"], 5: e["intron_end_on_align"]}, e["ea"], e["ga"]
    if k == ED:
        return {0: O.edit_distance(a, b)}, b"", b""
    if k == KBAND:
        e = O.kband(a, b, r["p"][0])
        return {0: e["ok"], 1: e["edit"]}, b"", b""
    if k == LCF:
        e = O.lcf(a, b)
        return {0: e["len"], 1: e["occ1"], 2: e["occ2"]}, b"", b""
    if k == BORDERS:
        e = O.refine_borders(a, b, r["p"][0], r["p"][1], r["p"][2], r["b_tail"])
        return {0: e["ok"], 1: e["off_p"], 2: e["off_t1"], 3: e["off_t2"], 4: e["ed"]}, b"", b""
    if k == AFFIX:
        e = O.longest_affix(a, b)
        return ({0: 1, 1: e["ecut"], 2: e["gcut"]} if e["valid"] else {0: 0}), b"", b""
    raise SystemExit("unknown kind %d" % k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--max-report", type=int, default=20)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    n = 0
    per_kind = [0] * 7
    bad = []
    for r in records(args.trace):
        n += 1
        per_kind[r["kind"]] += 1
        vals, s0, s1 = expected(r)
        diff = {i: (r["v"][i], x) for i, x in vals.items() if r["v"][i] != x}
        if r["kind"] in (ALIGN, GAP) and (r["s0"] != s0 or r["s1"] != s1):
            diff["rows"] = (len(r["s0"]), len(s0))
        if diff:
            bad.append(dict(record=n - 1, unit=r["unit"], kind=KIND_NAMES[r["kind"]], la=len(r["a"]), lb=len(r["b"]),
                            p=r["p"], tail=r["tail"], got_vs_expected={str(k): v for k, v in diff.items()}))
    print("%d answered DP requests replayed (%s); %d differ from the oracle" %
          (n, ", ".join("%s %d" % (KIND_NAMES[k], c) for k, c in enumerate(per_kind) if c), len(bad)))
    for b in bad[:args.max_report]:
        print("  ", json.dumps(b))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(dict(requests=n, per_kind=dict(zip(KIND_NAMES, per_kind)), differ=bad[:1000]), f, indent=1)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
