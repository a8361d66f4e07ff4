#!/usr/bin/env python3
"""Summarise a tools/profile.sh run (gpurun_out/prof_<tag>) into profiles/: kernel stats csv, PMC summary txt,
hbm_traffic.json (read by bench.py for roofline.traffic).   usage: summarize_profile.py <tag> <round-label>"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, label = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
ks = glob.glob(src + "/stats/*/*kernel_stats.csv")[0]
shutil.copy(ks, os.path.join(dst, label + "_kernel_stats.csv"))
out = ["rocprofv3 PMC summary (%s), s1_kernel + sr_fused_kernel<2,0,false,true>, command: bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-input" % label,
       "(8 frames 1920x1080->3840x2160 per launch, uniform-noise input; one --pmc pass per counter group; mean over the launches of a run)"]
vals = {}
for d in sorted(glob.glob(src + "/pmc_*/")):
    f = glob.glob(d + "*/*counter_collection.csv")
    if not f: continue
    # one bench step = s1_kernel + sr_fused_kernel<.., FROM_FEAT>: counters are summed over the pair
    agg = collections.defaultdict(list)
    first = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "s1_kernel" in r["Kernel_Name"]: first[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "sr_fused" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        a = first.get(k, [])
        if len(a) == len(v):
            v = [x + y for x, y in zip(v, a)]
        vals[k] = sum(v) / len(v)
        out.append("%-24s mean %.6g   min %.6g   max %.6g   (per step: both launches)" % (k, vals[k], min(v), max(v)))
avg = 0.0
for r in csv.DictReader(open(ks)):
    if "sr_fused" in r["Name"] or "s1_kernel" in r["Name"]:
        avg += float(r["AverageNs"]); out.append("kernel-trace: %s: %s calls, average %.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
out.append("kernel-trace: one step (stage-1 launch + stages-2/3 launch) = %.1f us" % (avg / 1e3))
fetch, write = vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
alg = 250585941
out += ["", "HBM-side traffic per launch: FETCH_SIZE %.1f MB raw (x2 by the gfx950 rule for 16-B/lane streaming reads = %.1f MB), WRITE_SIZE %.1f MB" % (fetch / 1e6, 2 * fetch / 1e6, write / 1e6),
        "algorithmic bytes per launch: %.3f MB -> traffic/algorithmic = %.2f (fetch doubled) / %.2f (raw)" % (alg / 1e6, (2 * fetch + write) / alg, (fetch + write) / alg),
        "L2 hit rate TCC_HIT/(HIT+MISS) = %.3f" % (vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]))]
cyc = vals["GRBM_GUI_ACTIVE"] / 8
out.append("GRBM_GUI_ACTIVE/8 = %.4g cycles per launch -> effective clock %.2f GHz over %.3f ms" % (cyc, cyc / avg, avg / 1e6))
cu = cyc * 256
out.append("LDS array busy = SQ_LDS_IDX_ACTIVE / (256 CU x cycles) = %.2f ; bank-conflict share of LDS cycles = %.2f" % (vals["SQ_LDS_IDX_ACTIVE"] / cu, vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"]))
out.append("VALU issue = SQ_INSTS_VALU / (256 CU x cycles) = %.2f wave-instr per CU-cycle (4 SIMDs; 2-cycle and 4-cycle instruction classes, see r01_valu_instruction_rates.txt)" % (vals["SQ_INSTS_VALU"] / cu))
# per-kernel view of the same counters
per = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(src + "/pmc_*/")):
    f = glob.glob(d + "*/*counter_collection.csv")
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        kn = "s1_kernel" if "s1_kernel" in r["Kernel_Name"] else ("sr_fused_kernel" if "sr_fused" in r["Kernel_Name"] else None)
        if kn: per[kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
out.append("")
out.append("per kernel (means per launch):")
for kn, c in per.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    if "GRBM_GUI_ACTIVE" not in m: continue
    cyc_k = m["GRBM_GUI_ACTIVE"] / 8 * 256
    out.append("  %-16s %.3g cycles; VALU issue %.2f per CU-cycle; LDS array busy %.2f (bank conflicts %.2f of it); L2 hit %.3f; FETCH %.1f MB raw, WRITE %.1f MB"
               % (kn, m["GRBM_GUI_ACTIVE"] / 8, m["SQ_INSTS_VALU"] / cyc_k, m["SQ_LDS_IDX_ACTIVE"] / cyc_k,
                  m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]),
                  m["FETCH_SIZE"] * 1024 / 1e6, m["WRITE_SIZE"] * 1024 / 1e6))
open(os.path.join(dst, label + "_pmc_summary.txt"), "w").write("\n".join(out) + "\n")
json.dump({"frames": 8, "input": "noise", "bytes_per_launch": int(2 * fetch + write),
           "valu_instr_per_cu_cycle": round(vals["SQ_INSTS_VALU"] / cu, 3), "lds_array_busy": round(vals["SQ_LDS_IDX_ACTIVE"] / cu, 3),
           "lds_bank_conflict_share": round(vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"], 3),
           "l2_hit_rate": round(vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 4),
           "kernel_trace_avg_us": round(avg / 1e3, 1), "fetch_size_bytes_raw": int(fetch), "write_size_bytes": int(write),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) x 1024; FETCH doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B streaming request; "
                   "the 1-B/lane input-tile reads are uncalibrated, so this is an upper bound)", "source": "profiles/%s_pmc_summary.txt" % label},
          open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
print("\n".join(out))
