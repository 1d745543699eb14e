"""markdown table of DESIGN.md section 5 from a bench line (+ the PMC passes of the same sources): python tools/design_table.py bench.log pmc_traffic.json mfma_util.txt"""
import json
import re
import sys

line = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][0])
pmc = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else {}
busy = {}
if len(sys.argv) > 3:
    for ln in open(sys.argv[3]):
        m = re.match(r"\s*\d+\s+[\d.]+\s+[\d.]+\s+([\d.]+)\s+[\d.]+\s+[\d.]+%\s+(\S+(?: \S+)*)", ln)
        if m:
            busy[re.sub(r"\s", "", m.group(2).split("(")[0])] = float(m.group(1))
r = line["roofline"]
print(f"# {line['value']} pairs/s, {line['ms_per_step']} ms/step; code {line.get('code')}")
print("| kernel (rocprofv3 name) | share of GEMM time | TFLOP/s (events) | `frac_mfma` | `frac_hbm` | bound | MFMA pipe busy (PMC) | HBM-side bytes / launch (PMC, FETCH doubled) |")
print("|---|---|---|---|---|---|---|---|")
for k in r["by_kernel"]:
    key = k["kernel"].split(" ")[0]
    t = pmc.get("by_kernel", {}).get(key, {}).get("bytes_per_launch")
    b = busy.get(re.sub(r"\s", "", key))
    print(f"| `{key}` {k['kernel'][len(key):].strip()} | {100 * k['share_of_gemm_time']:.1f} % | {k['achieved']:.0f} | {k['frac_mfma']:.3f} | {k['frac_hbm']:.3f} | {k['bound']} | "
          f"{'%.0f %%' % b if b else '-'} | {'%.2f GB' % (t / 1e9) if t else '-'} (algorithmic {k['algorithmic_gb_per_launch_avg']:.2f}) |")
for nm, key in (("NT family", "gemm_nt_family"), ("TN family", "gemm_tn_family"), ("all GEMM launches", "all_gemm_launches")):
    a = r[key]
    print(f"| {nm} ({a['launches']} launches) | | {a['achieved']:.0f} | {a['frac_mfma']:.3f} | {a['frac_hbm']:.3f} | | | |")
for k in ("dense_text_tower", "reference_work", "accum8_gbs32768", "torch_eager_baseline", "cpu_baseline"):
    if k in line:
        print("#", k, {kk: vv for kk, vv in line[k].items() if kk not in ("what", "sample", "reference_in_build_container")})
