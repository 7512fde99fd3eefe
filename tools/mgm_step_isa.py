#!/usr/bin/env python3
"""tools/mgm_step_isa.py -- instruction census of the band kernel's unrolled sweep step, from the compiler's own listing.

    python tools/mgm_step_isa.py [--kernel 'k_mgm_bands<16,4,false,3,4>'] [--asm census.s] [--listing] > profiles/r06/mgm_step_isa.txt

Compiles s2p_amd/csrc/census_kernels.hip for gfx950 with the shipped flags to assembly (`--offload-device-only -S`: the same code
generation as the library's object, no GPU needed), cuts out one instantiation of k_mgm_bands, finds the sweep loop (the depth-2
loop whose body holds the PF = 16 unrolled steps, each with its depth-3 poll loops) and counts instructions per category:

  * per STEP, the straight-line block every lane executes (the two ds_read_b128 of the predecessors' messages ... the ds_write_b128
    of this pixel's message): this is the dependent chain DESIGN_KERNELS.md 1 calls "the step";
  * per step, the scalar control around it at loop depth 2 (progress words, poll tests that fall through when data is there);
  * the depth-3 poll loops (executed only when a wave has to wait).

The categories are the hardware's issue classes: VALU (split into packed 16-bit, DPP, v_perm / v_alignbit, other), SALU, LDS (ds_*),
VMEM (buffer_* / global_*), s_waitcnt, s_nop, branches.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value"]     # s2p_amd/build.py FLAGS minus the link ones


def mangled(kernel):
    """k_mgm_bands<16,4,false,3,4> -> the Itanium name fragment the listing uses."""
    m = re.match(r"(\w+)<(.*)>", kernel)
    name, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
    enc = "".join("Lb%d" % (a == "true") if a in ("true", "false") else "Li%sE" % a for a in args)
    enc = re.sub(r"Lb(\d)", r"Lb\1E", enc)
    return "_ZN3s2p%d%sI%sEEv" % (len(name), name, enc)


def classify(op):
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_nop") or op.startswith("s_sleep"):
        return "s_nop/s_sleep"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc") or op.startswith("s_endpgm"):
        return "branch"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "VMEM"
    if op.startswith("v_"):
        return "VALU"
    return "other"


def valu_kind(line, op):
    if "row_sh" in line or "quad_perm" in line or "row_mirror" in line or "row_half_mirror" in line or "wave_sh" in line or "row_bcast" in line or op.startswith("v_permlane"):
        return "VALU dpp / lane exchange"
    if op.startswith("v_pk_"):
        return "VALU packed 16-bit"
    if op.startswith("v_perm_b32") or op.startswith("v_alignbit"):
        return "VALU v_perm / v_alignbit"
    if "sdwa" in op or "sdwa" in line:
        return "VALU sdwa"
    if op.startswith("v_readfirstlane") or op.startswith("v_readlane") or op.startswith("v_writelane"):
        return "VALU lane <-> scalar"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"):
        return "VALU compare / select"
    return "VALU other 32-bit"


INSN = re.compile(r"^\t([a-z_0-9]+)\b(.*)$")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="k_mgm_bands<16,4,false,3,4>")
    ap.add_argument("--asm", default=None, help="an existing listing of census_kernels.hip (skips the compile)")
    ap.add_argument("--listing", action="store_true", help="append the listing of one step's straight-line block")
    a = ap.parse_args()

    asm = a.asm
    if asm is None:
        tmp = tempfile.mkdtemp(prefix="mgm_isa_")
        asm = os.path.join(tmp, "census.s")
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["--offload-device-only", "-S", os.path.join(ROOT, "s2p_amd", "csrc", "census_kernels.hip"), "-o", asm]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    frag = mangled(a.kernel)
    start = next(i for i, l in enumerate(lines) if l.startswith(frag) and l.rstrip().split(":")[0].startswith(frag) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
    fn = lines[start:end]
    meta = {}
    for l in lines[end:end + 400]:
        m = re.match(r"; (NumVgprs|NumAgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte): (\d+)", l.strip())
        if m and m.group(1) not in meta:
            meta[m.group(1)] = int(m.group(2))
        if l.startswith("_ZN") and meta:
            break

    # loop structure from LLVM's block comments: every block label / %bb comment says which loop header it is in and at what depth
    hdr_of = {}                     # line index -> (header block, depth) in force from that line on
    cur = (None, 0)
    block_at = []
    depth_hdr = re.compile(r"(?:in Loop: Header=(BB\d+_\d+) Depth=(\d+))|(?:This (?:Inner )?Loop Header: Depth=(\d+))")
    label = re.compile(r"^\.L(BB\d+_\d+):")
    pend_label = None
    loops = collections.OrderedDict()     # header -> dict(depth, first, last, parent)
    for i, l in enumerate(fn):
        lm = label.match(l)
        if lm:
            pend_label = lm.group(1)
        m = depth_hdr.search(l)
        if m and (l.startswith(".LBB") or l.startswith("; %bb") or l.lstrip().startswith("; =>") or l.lstrip().startswith(";")):
            if m.group(1):
                cur = (m.group(1), int(m.group(2)))
            else:
                cur = (pend_label, int(m.group(3)))
                loops.setdefault(pend_label, {"depth": int(m.group(3)), "first": i, "last": i})
        elif (lm or l.startswith("; %bb")) and "Loop" not in l and "Depth" not in l:
            # a block outside every loop, or a continuation line of a multi-line loop comment: look ahead two lines for the header note
            nxt = " ".join(fn[i + 1:i + 4])
            if not depth_hdr.search(nxt) or INSN.match(fn[i + 1] if i + 1 < len(fn) else ""):
                cur = (None, 0)
        block_at.append(cur)
        if cur[0] in loops:
            loops[cur[0]]["last"] = i
    # children: depth-3 loops whose lines fall inside a depth-2 loop's line span
    d2 = [(h, v) for h, v in loops.items() if v["depth"] == 2]
    d3 = [(h, v) for h, v in loops.items() if v["depth"] == 3]

    def span_children(v):
        return [h for h, c in d3 if v["first"] <= c["first"] <= v["last"]]
    # the depth-2 loop's extent must include its depth-3 children: extend `last` to the last line attributed to it or to a child
    for h, v in d2:
        for i in range(v["first"], len(fn)):
            bh = block_at[i][0]
            if bh == h or (bh in loops and loops[bh]["depth"] == 3 and v["first"] <= loops[bh]["first"]):
                if bh == h:
                    v["last"] = max(v["last"], i)
    sweep_h, sweep = max(d2, key=lambda hv: len(span_children(hv[1])))
    kids = span_children(sweep)
    first, last = sweep["first"], sweep["last"]

    cat2, cat3, val2 = collections.Counter(), collections.Counter(), collections.Counter()
    steps = []                   # straight-line compute blocks: (start line, end line)
    cur_start = None
    for i in range(first, last + 1):
        m = INSN.match(fn[i])
        if not m:
            continue
        op = m.group(1)
        h, d = block_at[i]
        c = classify(op)
        if d >= 3 and h in kids:
            cat3[c] += 1
            continue
        if h != sweep_h:
            continue
        cat2[c] += 1
        if c == "VALU":
            val2[valu_kind(fn[i], op)] += 1
        if op == "ds_read_b128" and cur_start is None:
            cur_start = i
        if op == "ds_write_b128" and cur_start is not None:
            steps.append((cur_start, i))
            cur_start = None
    nsteps = len(steps)
    blk, blk_valu = collections.Counter(), collections.Counter()
    straight = True
    for s, e in steps:
        for i in range(s, e + 1):
            if label.match(fn[i]):
                straight = False
            m = INSN.match(fn[i])
            if not m:
                continue
            op = m.group(1)
            c = classify(op)
            blk[c] += 1
            if c == "VALU":
                blk_valu[valu_kind(fn[i], op)] += 1

    out = []
    P = out.append
    P("instruction census of %s (gfx950; hipcc %s)" % (a.kernel, " ".join(FLAGS)))
    P("listing: `hipcc ... --offload-device-only -S s2p_amd/csrc/census_kernels.hip` (tools/mgm_step_isa.py); function %s..., %d lines" % (frag[:48], len(fn)))
    if meta:
        P("kernel resources: " + ", ".join("%s %d" % kv for kv in meta.items()))
    P("sweep loop: header .L%s, lines %d-%d of the function, %d unrolled steps (one ds_write_b128 of the message each), %d depth-3 poll loops (%.1f per step)"
      % (sweep_h, first, last, nsteps, len(kids), len(kids) / max(nsteps, 1)))
    P("")
    P("A. the step's straight-line block (first ds_read_b128 of the predecessors' messages ... ds_write_b128 of the message; %s)"
      % ("no label inside: one basic block, every lane executes all of it" if straight else "contains labels"))
    P("   per step, averaged over the %d unrolled steps:" % nsteps)
    tot = sum(blk.values())
    for c in ("VALU", "SALU", "LDS", "VMEM", "s_waitcnt", "s_nop/s_sleep", "branch", "other"):
        if blk[c]:
            P("     %-16s %7.2f" % (c, blk[c] / nsteps))
    P("     %-16s %7.2f" % ("total", tot / nsteps))
    P("   VALU by kind:")
    for k, v in sorted(blk_valu.items(), key=lambda kv: -kv[1]):
        P("     %-28s %7.2f" % (k, v / nsteps))
    P("")
    P("B. everything at loop depth 2 in the sweep loop (A + the scalar control between steps: progress words, the tests in front of the")
    P("   poll loops, the band hand-off store, loop bookkeeping), per step:")
    tot2 = sum(cat2.values())
    for c in ("VALU", "SALU", "LDS", "VMEM", "s_waitcnt", "s_nop/s_sleep", "branch", "other"):
        if cat2[c]:
            P("     %-16s %7.2f" % (c, cat2[c] / nsteps))
    P("     %-16s %7.2f   (an upper bound of what a step issues when it never waits: some of these blocks are skipped)" % ("total", tot2 / nsteps))
    P("   VALU by kind:")
    for k, v in sorted(val2.items(), key=lambda kv: -kv[1]):
        P("     %-28s %7.2f" % (k, v / nsteps))
    P("")
    P("C. the depth-3 poll loops (run only while a wave waits for a progress word): %d instructions in %d loops" % (sum(cat3.values()), len(kids)))
    P("     " + ", ".join("%s %d" % kv for kv in cat3.most_common()))
    P("")
    P("VALU per step against the model of DESIGN_KERNELS.md 1 (K = 4 registers of two candidates per lane, three predecessors):")
    P("     block A: %.1f VALU;  depth 2 total: %.1f VALU" % (blk["VALU"] / nsteps, cat2["VALU"] / nsteps))
    if a.listing and steps:
        s, e = steps[min(2, nsteps - 1)]
        P("")
        P("D. listing of one step's straight-line block (step %d of the unrolled loop):" % min(2, nsteps - 1))
        for i in range(s, e + 1):
            if fn[i].strip():
                P("   " + fn[i].replace("\t", "    "))
    print("\n".join(out))


if __name__ == "__main__":
    main()
