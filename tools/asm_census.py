#!/usr/bin/env python3
"""tools/asm_census.py <tag> [hipcc flags...]: what the register allocator did to the chain kernels, per kernel and per basic block.

Compiles digiham_amd/csrc/engine.hip for gfx950 with -DDH_ASM_MARKERS (device only, assembly), writes every chain kernel to its OWN file
(/tmp/census_<tag>_<proto>.s) and reports, per kernel:
  * registers, scratch size, LDS;
  * the basic blocks of the RUN LOOP of the slicer half (the depth-1 loop that holds the `DH_PHASE 0` marker): how many there are, how many
    instructions, and which of them touch scratch -- with the block's size, its loop depth and whether it lies on the LAYOUT HOT PATH;
  * the layout hot path = the walk from the loop header that follows unconditional branches and otherwise falls through (the compiler
    lays the likely successor out as the fall-through, and the DH_LIKELY / DH_UNLIKELY weights decide what is likely): its vector / scalar /
    LDS / memory / scratch instruction counts per phase (between the DH_PHASE markers met on the walk).  The walk stops when it is back at
    the header.  A scratch access ON this path is a spill in the hot loop; one in a block off the path sits behind a rare branch.
The same for the decoder half's sections (DH_DMARK markers) is left to the counters (tools/pmc_insts.sh).
Run before and after touching a rare path: one more live value there can move spills into the hot loop."""
import os, re, subprocess, sys
from collections import Counter, OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = OrderedDict((("dmr", "_ZN12_GLOBAL__N_17k_chainILi80ELb0ELi1ELi10ELi0EE"), ("ysf", "_ZN12_GLOBAL__N_17k_chainILi80ELb0ELi2ELi10ELi0EE"),
                       ("nxdn", "_ZN12_GLOBAL__N_17k_chainILi160ELb0ELi3ELi20ELi0EE")))


def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "vector"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "scalar"
    return None


def parse(lines):
    """-> ordered list of blocks: dict(label, depth, header, insts [(op, text)], marks [(index, phase)])"""
    blocks, cur = [], dict(label="entry", depth=0, header=None, insts=[], marks=[])
    for line in lines:
        t = line.strip()
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", t)
        if m:
            blocks.append(cur)
            c = m.group(2)
            depth, header = 0, None
            mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", c)
            if mm: header, depth = "." + "L" + mm.group(1), int(mm.group(2))
            mm = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", c)
            if mm: header, depth = m.group(1), int(mm.group(1))
            cur = dict(label=m.group(1), depth=depth, header=header, insts=[], marks=[], parents=[])
            continue
        mm = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", t)
        if mm and not cur["insts"]:
            cur.setdefault("parents", []).append(".L" + mm.group(1))
            continue
        mm = re.search(r"(?:This (?:Inner )?Loop Header|in Loop): ?(?:Header=(BB\d+_\d+) )?Depth=(\d+)", t)
        if t.startswith(";") and mm and not cur["insts"]:
            if mm.group(1): cur["header"] = ".L" + mm.group(1)
            elif "Loop Header" in t: cur["header"] = cur["label"]
            cur["depth"] = int(mm.group(2))
            continue
        mm = re.search(r"; DH_PHASE (\w+)", t)
        if mm:
            cur["marks"].append((len(cur["insts"]), mm.group(1)))
            continue
        if not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        if kind(op):
            cur["insts"].append((op, t))
    blocks.append(cur)
    return blocks


def census(path, out):
    lines = open(path).read().splitlines()
    blocks = parse(lines)
    by_label = {b["label"]: i for i, b in enumerate(blocks)}
    # the run loop: the depth-1 loop whose blocks hold the DH_PHASE 0 marker
    run = None
    for b in blocks:
        if any(ph == "0" for _, ph in b["marks"]):
            run = b["header"] if b["depth"] == 1 else (b.get("parents") or [None])[0]
    if run is None:
        out.append("  (no DH_PHASE 0 marker inside a loop: cannot place the run loop)")
        return
    def in_run(b):
        return (b["depth"] == 1 and b["header"] == run) or run in (b.get("parents") or [])
    rb = [b for b in blocks if in_run(b)]
    tot = Counter()
    for b in rb:
        for op, _ in b["insts"]: tot[kind(op)] += 1
    out.append("  run loop %s: %d basic blocks, %d instructions in the text (%s)" % (run, len(rb), sum(tot.values()), ", ".join("%s %d" % kv for kv in sorted(tot.items()))))
    # layout hot path
    path_blocks, per_phase, phase, seen = [], OrderedDict(), "loop top", set()
    i = by_label.get(run)
    steps = 0
    while i is not None and i < len(blocks) and steps < 2000:
        b = blocks[i]; steps += 1
        if b["label"] in seen and b["label"] == run: break
        if b["label"] in seen: break
        seen.add(b["label"]); path_blocks.append(b["label"])
        marks = dict(b["marks"])
        nxt = i + 1
        ended = False
        for j, (op, text) in enumerate(b["insts"]):
            if j in marks: phase = "after " + marks[j]
            per_phase.setdefault(phase, Counter())[kind(op)] += 1
            if op == "s_branch":
                tgt = text.split()[1]
                nxt = by_label.get(tgt); ended = True
                break
            if op in ("s_endpgm",):
                nxt = None; ended = True; break
        if len(b["insts"]) in marks: phase = "after " + marks[len(b["insts"])]
        if nxt is not None and nxt < len(blocks) and blocks[nxt]["label"] == run: break
        if nxt is not None and nxt < len(blocks) and not in_run(blocks[nxt]) and not ended:
            break                                                  # fell out of the loop
        i = nxt
    hot = Counter()
    for c in per_phase.values(): hot.update(c)
    out.append("  layout hot path: %d blocks, %d instructions (%s)" % (len(path_blocks), sum(hot.values()), ", ".join("%s %d" % kv for kv in sorted(hot.items()))))
    for ph, c in per_phase.items():
        out.append("    %-12s %s" % (ph, ", ".join("%s %d" % kv for kv in sorted(c.items()))))
    out.append("  scratch accesses ON the layout hot path: %d" % hot.get("scratch", 0))
    hotset = set(path_blocks)
    rows = []
    for b in rb:
        n = sum(1 for op, _ in b["insts"] if kind(op) == "scratch")
        if n:
            loads = sum(1 for op, _ in b["insts"] if op.startswith("scratch_load"))
            rows.append((b["label"], len(b["insts"]), n, loads, b["depth"], "HOT PATH" if b["label"] in hotset else "off the hot path"))
    out.append("  blocks of the run loop with scratch accesses (%d of %d blocks; label, instructions, scratch accesses, of which loads, loop depth):" % (len(rows), len(rb)))
    for r in rows:
        out.append("    %-12s %5d instr  %3d scratch (%d loads)  depth %d  %s" % r)
    off = sum(r[2] for r in rows if r[5] != "HOT PATH")
    out.append("  scratch accesses in run-loop blocks off the hot path: %d (behind branches the layout treats as unlikely: the reference-order FIR, exact symbol / ring evaluations, edge windows, the ordered timing chain)" % off)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "cur"
    flags = sys.argv[2:]
    asm = "/tmp/census_%s.s" % tag
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
           "-Wno-parentheses-equality", "-DDH_ASM_MARKERS", "--cuda-device-only", "-S"] + flags + [os.path.join(ROOT, "digiham_amd/csrc/engine.hip"), "-o", asm]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    text = open(asm).read().splitlines()
    out = ["# tools/asm_census.py %s %s" % (tag, " ".join(flags)), "# hipcc -O3 --offload-arch=gfx950 -DDH_ASM_MARKERS, digiham_amd/csrc/engine.hip"]
    for proto, sym in KERNELS.items():
        start = next((i for i, l in enumerate(text) if l.startswith(sym) and re.match(r"^\S+:\s*(;.*)?$", l)), None)
        if start is None:
            out.append("== %s: kernel %s not found" % (proto, sym)); continue
        end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
        kpath = "/tmp/census_%s_%s.s" % (tag, proto)
        open(kpath, "w").write("\n".join(text[start:end + 1]) + "\n")
        meta = {}
        for l in text[end:end + 400]:
            m = re.match(r"^; (NumVgprs|NumSgprs|ScratchSize|LDSByteSize|Occupancy|TotalNumVgprs): (\S+)", l)
            if m and m.group(1) not in meta: meta[m.group(1)] = m.group(2)
        out.append("== k_chain<%s> (%s; own file %s): %s" % (proto, sym, kpath, ", ".join("%s %s" % kv for kv in meta.items())))
        census(kpath, out)
    print("\n".join(out))


if __name__ == "__main__":
    main()
