#!/usr/bin/env python3
"""Would factoring the steering-Gaussian weight cut the transcendentals of stage 3 at S = 4, x2?  

A weight is w = exp2(-(tx^2 + ty^2 + m tx ty)) with tx = k1 dx, ty = k2 dy and (k1, k2, m) the TAP's hyper-parameters
(lerf_stage3.h gauss_form_u8; reference resize_right2d_numpy.py:150-160).  One thread forms all weights of a block of
outputs, so what it can share are the evaluations of ONE tap towards the outputs of ITS block.  Three ways to form them:

  direct      one v_exp_f32 per (tap, output)                                         [what the kernel does: 3 packed FMA-class
                                                                                       operations per 2 pairs + the exp]
  separable   exp2(-tx^2) exp2(-ty^2) exp2(-m tx ty): one exp per DISTINCT dx, per distinct dy and per distinct signed dx dy
              the tap sees inside the block, then 2 multiplies per (tap, output)
  recurrence  along an output row, E(j+1) = E(j) R(j), R(j+1) = R(j) Q: 2 exps per (tap, output row) run + 1 per tap (Q),
              2 multiplies per further output of the run; (rounding grows with the run: the tie guard would have to widen)

Issue cost per wave instruction on gfx950 (profiles/r03_issue_rates.txt): v_exp_f32 8.1 cycles, v_mul/v_fma 4.0 (a packed
one serves two values).  The table prints, per block shape, taps, (tap, output) pairs, exps and cycles per pair."""
import math, itertools
S, SC = 4, 2
def taps_of(o):
    c = (o + 0.5) / SC - 0.5
    t0 = math.floor(c) - S // 2 + 1
    return [(t, c - t) for t in range(t0, t0 + S)]
EXP, OP = 8.1, 4.0
print("block (rows x cols) | taps | pairs | accumulators/ch | direct exps  cyc/pair | separable exps  cyc/pair | recurrence exps  cyc/pair")
for bh, bw in [(2, 2), (2, 4), (4, 4), (2, 8), (4, 8), (8, 8)]:
    rows = range(1, 1 + bh); cols = range(1, 1 + bw)           # odd start: the pairs (2i+1, 2i+2) share their taps
    rt = {y: taps_of(y) for y in rows}; ct = {x: taps_of(x) for x in cols}
    taps = {}
    for y, x in itertools.product(rows, cols):
        for (ty, dy), (tx, dx) in itertools.product(rt[y], ct[x]):
            taps.setdefault((ty, tx), []).append((y, x, dy, dx))
    pairs = sum(len(v) for v in taps.values())
    # direct: per pair 1 exp + 1.5 packed FMA-class operations (tx*p0 + (tx*tx + ty2) as 2 pk_fma per 2 pairs, + the per-column
    # terms amortised) + 1 packed accumulate pair (num, den) per 2 pairs
    direct_cyc = EXP + OP * (2 / 2 + 2 / 2)
    sep_exps = sum(len({d[3] for d in v}) + len({d[2] for d in v}) + len({round(d[2] * d[3], 6) for d in v}) for v in taps.values())
    sep_cyc = sep_exps / pairs * (EXP + OP) + OP * (2 / 2 + 2 / 2)          # each exp also needs its argument (1 op); 2 muls + accumulate, packed
    rec_exps = 0; rec_mul = 0
    for v in taps.values():
        rec_exps += 1
        for y in {d[0] for d in v}:
            L = len([d for d in v if d[0] == y])
            rec_exps += min(L, 2); rec_mul += 2 * max(L - 2, 0)
    rec_cyc = (rec_exps * (EXP + 3 * OP / 1) + rec_mul * OP) / pairs + OP * (2 / 2)   # an exp's argument is the full form (3 ops); unpacked chain
    print("%9d x %-8d | %4d | %5d | %15d | %11d %9.1f | %14d %9.1f | %15d %9.1f" %
          (bh, bw, len(taps), pairs, 2 * bh * bw, pairs, direct_cyc, sep_exps, sep_cyc, rec_exps, rec_cyc))
print("""
registers: the headline kernel runs at 125-128 VGPRs with 2 x 2 blocks (8 accumulators per channel, channels in turn);
a 4 x 4 block needs 32, an 8 x 8 block 128 accumulators per channel.  At x2 a tap reaches at most 8 x 8 outputs
(S * scale per axis), so the runs of the recurrence are at most 8 long and 1-4 long inside a 4 x 4 block.""")
