#!/usr/bin/env python3
"""VGPR ledger per phase of the tile-fused kernels ("a VGPR ledger per phase that shows the registers do not
exist").  Compiles lerf_fused.hip with -DLERF_STAMPS to device assembly: the s_memtime instructions of the stamped build are
the phase boundaries of tools/stamps.py.  For every phase: instructions, the VGPRs it references (an upper bound of what is
live inside it; values that only pass through are not counted), the highest register index the allocator reached there, and
what the phase issues (exp, packed f32, LDS gathers).  The stage-3 range is additionally cut into windows of 250 lines: its
task shapes (block tasks, dword-column tasks, edge variants, the float64 tie queue) are separate code paths.

    python tools/vgpr_ledger.py [S ...]          (default: 2 4; prints to stdout -> profiles/r05_vgpr_ledger.txt)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = ["prologue (ids, geometry search issue)", "feat tile load + geometry search", "binning + slot set-up",
          "stage 2: 18 x (piece store, prefetch, lookups)", "finalisation + stage 3 (all task shapes) + tie queue + stores"]


def regs(line):
    out = set()
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def summarise(chunk):
    r = set()
    n = 0
    for l in chunk:
        if re.match(r"\s+[a-z]", l) and not l.strip().startswith((";", ".")):
            n += 1
            r |= regs(l)
    cnt = lambda pat: sum(1 for l in chunk if re.match(r"\s+" + pat, l))
    return n, len(r), (max(r) if r else -1), cnt("v_exp_f32"), cnt("v_pk_"), cnt("ds_read_b32"), cnt("v_cvt_")


def main():
    supports = [int(a) for a in sys.argv[1:]] or [2, 4]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "fused.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fno-slp-vectorize", "--offload-arch=gfx950", "-w",
                               "-I" + os.path.join(ROOT, "include"), "-DLERF_STAMPS", "-S", "--cuda-device-only", "-o", asm,
                               os.path.join(ROOT, "lerf-pytorch_amd", "csrc", "lerf_fused.hip")])
        text = open(asm).read()
    for S in supports:
        sym = "_ZN4lerf5fused15sr_fused_kernelILi%dELi0ELb0ELb1ELb0EEE" % S
        m = re.search(r"^%s[^\n]*\n" % re.escape(sym), text, re.M)
        body = text[m.end():]
        body = body[:body.index("s_endpgm")].splitlines()
        total = re.search(r"\.name:\s+%s\S*\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)" % re.escape(sym), text)
        marks = [i for i, l in enumerate(body) if re.match(r"\s+s_memtime", l)]
        # phase boundaries: stamp 0, stamp 6 (tile loaded), stamp 7 (slots ready), stamp 10 (stage 2 done), stamp 12 (end)
        cut = [0, marks[0], marks[1], marks[2], None, len(body)]
        # stamp 10 = the first s_memtime behind the last lookups stamp: the marks between marks[2] and the stage-3 range are the
        # per-phase copy / lookup stamps inside the (non-unrolled) phase loop
        inner = [k for k in marks[3:] if k < marks[2] + 8000]
        cut[4] = inner[-1]
        print("sr_fused_kernel<S = %d, gauss, FROM_FEAT>: %d instructions, %s VGPRs allocated (launch bound 1024 threads = 128)" % (
            S, summarise(body)[0], total.group(1) if total else "?"))
        print("  %-66s %7s %9s %8s %5s %5s %8s %5s" % ("phase", "instr", "VGPRs ref", "max idx", "exp", "v_pk", "ds_b32", "cvt"))
        for k in range(5):
            n, d, mx, ex, pk, ds, cv = summarise(body[cut[k]:cut[k + 1]])
            print("  %-66s %7d %9d %8d %5d %5d %8d %5d" % (PHASES[k], n, d, mx, ex, pk, ds, cv))
        print("  stage-3 range in windows of 250 lines (code paths of one phase, not a time line):")
        lo = cut[4]
        rows = []
        for i in range(lo, len(body), 250):
            n, d, mx, ex, pk, ds, cv = summarise(body[i:i + 250])
            rows.append((i - lo, n, d, mx, ex, pk, ds, cv))
        for r in rows:
            if r[4] or r[5]:
                print("    +%-6d %5d instr  %3d VGPRs referenced  max index v%-3d  exp %3d  v_pk %3d  ds_b32 %3d" % r[:7])
        blk = [r for r in rows if r[5] >= 40]                      # windows of the packed-f32 block tasks
        oth = [r for r in rows if r[4] and r[5] < 40]
        print("  -> block tasks (windows with >= 40 packed operations): at most %d VGPRs referenced, highest index v%d;" % (
            max(r[2] for r in blk), max(r[3] for r in blk)))
        print("     other task shapes: at most %d referenced, highest index v%d.  The allocation (%s) is stage 2's: prefetch 36 + sums up to 2 x 14 + slot"
              % (max(r[2] for r in oth), max(r[3] for r in oth), total.group(1) if total else "?"))
        print("     temporaries; stage 3 runs after it with those registers dead -- a larger task shape has %d registers to grow into.\n"
              % (128 - max(r[2] for r in blk) - 12))

if __name__ == "__main__":
    main()
