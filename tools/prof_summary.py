"""Summarise rocprofv3 CSV output of tools/prof.sh: per-kernel time (kernel-trace) and
per-kernel mean PMC counters.  FETCH_SIZE is doubled (gfx950 reports 64 B per 128 B
request for wide coalesced reads, MI355X_MICROARCH.md §HBM); unit KiB -> bytes."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "")
    return name[:70]


def find(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


f = find("trace", "*kernel_stats.csv")
if f:
    print("== kernel-trace stats (all dispatches of the run: 1 warmup + 3 steps + golden check)")
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print(f"{short(r['Name']):72s} calls {int(r['Calls']):5d} total_us {float(r['TotalDurationNs'])/1e3:10.1f} "
              f"avg_us {float(r['AverageNs'])/1e3:9.2f} pct {float(r['Percentage']):6.2f}")

pm = defaultdict(lambda: defaultdict(list))
for sub in ["pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_tcc"]:
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        pm[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
if pm:
    print("\n== PMC means per dispatch")
    names = sorted(pm, key=lambda k: -sum(pm[k].get("SQ_BUSY_CYCLES", [0])))
    for k in names[:14]:
        c = {n: sum(v) / len(v) for n, v in pm[k].items()}
        line = f"{k:60s}"
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
            wc = c["SQ_WAVE_CYCLES"]
            line += (f" waves {c.get('SQ_WAVES',0):9.0f} wait_any {c.get('SQ_WAIT_ANY',0)/wc:5.2f} "
                     f"wait_inst {c.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} active {c.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} "
                     f"mfma_busy_cyc {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3g} gui {c.get('GRBM_GUI_ACTIVE',0):.3g} "
                     f"sq_busy {c.get('SQ_BUSY_CYCLES',0):.3g}")
        print(line)
        l2 = ""
        if "FETCH_SIZE" in c:
            l2 += f"   FETCH {c['FETCH_SIZE']*1024*2/1e6:9.2f} MB (x2-corrected)"
        if "WRITE_SIZE" in c:
            l2 += f"  WRITE {c['WRITE_SIZE']*1024/1e6:9.2f} MB"
        if "TCC_HIT_sum" in c:
            h, m = c["TCC_HIT_sum"], c.get("TCC_MISS_sum", 0)
            l2 += f"  L2 hit {h/(h+m+1e-9):5.3f}"
        if "SQ_INSTS_VALU" in c:
            l2 += (f"  valu_insts {c['SQ_INSTS_VALU']:.3g} mfma_i8 {c.get('SQ_INSTS_VALU_MFMA_I8',0):.3g} "
                   f"lds_insts {c.get('SQ_INSTS_LDS',0):.3g} bank_conf {c.get('SQ_LDS_BANK_CONFLICT',0):.3g} "
                   f"act_lds {c.get('SQ_ACTIVE_INST_LDS',0):.3g}")
        if l2:
            print(l2)

# ---- HBM traffic per QuantLinear GEMM launch -> pmc_traffic.json (feeds bench.py roofline.traffic)
import json
gemm = {k: v for k, v in pm.items() if k.startswith("gemm_") or k.startswith("mlp384")}
if gemm and all("FETCH_SIZE" in v and "WRITE_SIZE" in v for v in gemm.values()):
    per, tot_b, tot_n = {}, 0.0, 0
    for k, v in gemm.items():
        n = len(v["FETCH_SIZE"])
        fb = sum(v["FETCH_SIZE"]) / n * 1024 * 2
        wb = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) * 1024
        per[k] = {"launches": n, "fetch_MB_x2": round(fb / 1e6, 2), "write_MB": round(wb / 1e6, 2)}
        tot_b += (fb + wb) * n
        tot_n += n
    js = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; FETCH doubled per MI355X_MICROARCH.md HBM section)",
          "command": "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --streams 1 --graph 0",
          "per_kernel": per, "avg_bytes_per_launch": tot_b / tot_n}
    json.dump(js, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print("\npmc_traffic.json: avg HBM bytes per GEMM-class launch (gemm_*, mlp384_kernel, mlp384rs_kernel) %.1f MB" % (tot_b / tot_n / 1e6))
