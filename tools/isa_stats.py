"""Static shape of one kernel in the engine's ISA listing (newmap_amd/csrc: `make asm` writes nm_engine.s; with
HIPFLAGS_EXTRA=-gline-tables-only the listing carries .loc lines and --lines attributes instructions to source lines):
instruction counts by class, the loops (backward branches) with their sizes, and which of them hold no memory operation --
a loop of pure arithmetic inside a divergent state machine runs, in a wave, as long as its longest lane.

    python tools/isa_stats.py newmap_amd/csrc/nm_engine.s k_sweepILb1ELb0ELb0E [--lines]
"""
import collections
import re
import sys


def kernel_body(lines, pattern):
    i0 = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pattern) + r"\w*:", l))
    i1 = next(i for i in range(i0, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    return lines[i0].rstrip(":"), lines[i0:i1]


def main():
    path, pattern = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    name, body = kernel_body(lines, pattern)
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    labels, rows, where, cur = {}, [], [], None
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(rows)
            continue
        s = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if l.startswith("\t") and not s.startswith((".", ";")):
            rows.append(s)
            where.append(cur)
    kinds = collections.Counter("VALU" if r.startswith("v_") else "SALU / control" if r.startswith("s_") else "LDS" if r.startswith("ds_")
                                else "memory" if r.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other" for r in rows)
    print(name)
    print(f"  {len(rows)} instructions: {dict(kinds)}")
    loops = set()
    for i, s in enumerate(rows):
        m = re.match(r"(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", s)
        if m and m.group(2) in labels and labels[m.group(2)] <= i:
            loops.add((labels[m.group(2)], i))
    for a, b in sorted(loops):
        seg = rows[a:b + 1]
        mem = sum(1 for x in seg if x.startswith(("global_", "buffer_", "flat_", "scratch_", "ds_")))
        print(f"  loop {a:5d} .. {b:5d}: {b - a + 1:5d} instructions" + ("" if mem else "   (no memory operation)"))
    if "--lines" in sys.argv:
        cnt = collections.Counter(where)
        for (f, ln), n in sorted(((k or ("?", 0), v) for k, v in cnt.items()), key=lambda kv: -kv[1])[:40]:
            print(f"  {f}:{ln}  {n}")


if __name__ == "__main__":
    main()
