#!/usr/bin/env python3
"""Scratch (spill) operations inside the LDS-DMA loops of a device assembly file -- the rule
tests/test_abi_and_host.py::test_no_register_spills_inside_the_counted_vmcnt_pipelines enforces, as a report.
usage: tools/spill_check.py lib/filterinterp_multi.s"""
import re
import sys


def report(path):
    blocks, cur, func = [], None, ""
    for line in open(path):
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", line)
        if m:
            hdr = re.search(r"Header[:=]\s*(BB\d+_\d+)", m.group(2))
            cur = {"label": m.group(1)[2:], "loop": hdr.group(1) if hdr else None, "ins": [],
                   "is_header": "Loop Header" in m.group(2), "func": func}
            blocks.append(cur)
        elif re.match(r"^_Z\w+:", line):
            cur = None
            func = line.split(":")[0]
        elif cur is not None and re.match(r"^\s+[a-z]", line):
            cur["ins"].append(line.strip())
    for b in blocks:
        if b["is_header"]:
            b["loop"] = b["label"]
    for b in blocks:
        # what follows the branch back to the loop's header in the same text block is the loop's exit path (an unlabelled
        # fall-through block), not the loop
        if b["loop"]:
            for k, i in enumerate(b["ins"]):
                if re.match(r"s_c?branch\w*\s+\.L" + re.escape(b["loop"]) + r"\b", i):
                    b["ins"] = b["ins"][:k + 1]
                    break
    is_dma = lambda i: i.startswith("buffer_load") and " lds" in i      # noqa: E731
    dma_loops = {b["loop"] for b in blocks if b["loop"] and any(is_dma(i) for i in b["ins"])}
    bad = 0
    for b in blocks:
        if b["loop"] in dma_loops:
            sp = [i for i in b["ins"] if i.startswith("scratch_")]
            n = len(b["ins"])
            kinds = {}
            for i in b["ins"]:
                k = i.split()[0]
                kinds[k] = kinds.get(k, 0) + 1
            top = sorted(kinds.items(), key=lambda kv: -kv[1])[:8]
            print("%s %s: %d instructions, %d scratch; %s" % (b["func"][:40], b["label"], n, len(sp), top))
            bad += len(sp)
    return bad


if __name__ == "__main__":
    sys.exit(1 if sum(report(p) for p in sys.argv[1:]) else 0)
