#!/usr/bin/env python3
"""Re-encodes selected VOP1 / VOP2 instructions of a gfx950 assembly file in their 64-bit VOP3 form (`_e32` -> `_e64`).

Why (measured, profiles/r03_issue_rates.txt, tools/ubench/gen_issue_order.py / gen_e64.py): on gfx950 a 32-bit-encoded
simple VALU instruction (v_add_u32, v_lshrrev_b32, v_or_b32, v_mul_f32, v_add_f32, v_fmac_f32 ...) issues in ~2.2 cycles only
inside a run of such instructions; next to a VOP3 / transcendental / 24-bit-multiply instruction it costs a full 4-cycle
pass (A^15 M: 4.0 cycles per v_add_u32), while the SAME operation in its 64-bit encoding costs ~2.8 there.  hipcc always
shrinks to the 32-bit form (SIShrinkInstructions, no switch), so the build re-encodes the hot kernels' assembly instead:
    hipcc -S --cuda-device-only  ->  e64.py  ->  clang -x assembler  ->  lld  ->  clang-offload-bundler  ->  host object

usage: e64.py in.s out.s [--ops simple|all|<comma list>] [--kernels REGEX] [--stats]
Instructions that cannot take the VOP3 form on gfx9 (a 32-bit literal operand, implicit VCC, SDWA / DPP) are left alone.
"""
import re
import sys

SIMPLE = {
    "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
    "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_mov_b32",
}
COMPLEX = {
    "v_lshlrev_b32", "v_max_f32", "v_min_f32", "v_max_u32", "v_min_u32", "v_max_i32", "v_min_i32", "v_mul_u32_u24", "v_mul_i32_i24",
    "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte2", "v_cvt_f32_ubyte3", "v_cvt_f32_u32", "v_cvt_f32_i32",
    "v_cvt_u32_f32", "v_cvt_i32_f32", "v_rndne_f32", "v_exp_f32", "v_rcp_f32", "v_cvt_f64_f32", "v_cvt_f32_f64",
}
INLINE_F = {"0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0", "0.15915494"}


def is_literal(tok):
    t = tok.strip()
    if not t:
        return False
    if re.fullmatch(r"-?\|?[vs]\d+\|?|[vs]\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0|scc|src_\w+|-?\|?v\d+\|?", t):
        return False
    if t in INLINE_F:
        return False
    try:
        v = int(t, 0)
        return not (-16 <= v <= 64)
    except ValueError:
        pass
    try:
        float(t)
        return True            # a float that is not one of the inline constants
    except ValueError:
        return True            # symbols, expressions: literal


def convert(lines, ops, kre, stats):
    out = []
    active = kre is None
    fn = None
    for ln in lines:
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", ln)
        if m and not ln.startswith(".L"):
            fn = m.group(1)
            active = kre is None or re.search(kre, fn) is not None
        m = re.match(r"^(\s+)(v_\w+?)_e32(\s+)([^;\n]*)(.*)$", ln)
        if m and active:
            ind, op, sp, args, rest = m.groups()
            toks = [a for a in args.split(",")]
            if op in ops and not any(x in args for x in ("sdwa", "dpp", "vcc")) and not any(is_literal(t) for t in toks[1:]):
                stats[op] = stats.get(op, 0) + 1
                out.append("%s%s_e64%s%s%s\n" % (ind, op, sp, args, rest))
                continue
        out.append(ln if ln.endswith("\n") else ln + "\n")
    return out


def main():
    a = sys.argv[1:]
    src, dst = a[0], a[1]
    ops = set(SIMPLE)
    kre = None
    show = False
    i = 2
    while i < len(a):
        if a[i] == "--ops":
            v = a[i + 1]
            ops = set(SIMPLE) if v == "simple" else (SIMPLE | COMPLEX if v == "all" else set(v.split(",")))
            i += 2
        elif a[i] == "--kernels":
            kre = a[i + 1]
            i += 2
        elif a[i] == "--stats":
            show = True
            i += 1
        else:
            raise SystemExit("unknown option " + a[i])
    stats = {}
    with open(src) as f:
        lines = f.readlines()
    out = convert(lines, ops, kre, stats)
    with open(dst, "w") as f:
        f.writelines(out)
    if show:
        print("e64.py: %d instructions re-encoded: %s" % (sum(stats.values()), ", ".join("%s %d" % kv for kv in sorted(stats.items()))), file=sys.stderr)


if __name__ == "__main__":
    main()
