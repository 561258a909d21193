"""Copy the judged summaries of scripts/round_measure.sh from gpurun_out/<round>/ (scratch) into
profiles/ (tracked): kernel stats of the rocprofv3 --kernel-trace --stats run, the PMC per-kernel
traffic table, the bench JSON lines and the CPU sweep logs.  usage: collect_profiles.py r01"""
import glob, json, os, shutil, subprocess, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
src, dst = os.path.join("gpurun_out", rnd), "profiles"
os.makedirs(dst, exist_ok=True)
newest = lambda pattern: sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True)  # gpurun_out keeps earlier runs too
ks = newest(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if ks: shutil.copy(ks[0], os.path.join(dst, f"{rnd}_vls128_kernel_stats.csv"))
f = newest(os.path.join(src, "pmc_fetch", "*", "*counter_collection.csv"))
w = newest(os.path.join(src, "pmc_write", "*", "*counter_collection.csv"))
if f and w:
    out = subprocess.run([sys.executable, "scripts/pmc_summary.py", f[0], w[0], os.path.join(dst, f"{rnd}_vls128_pmc_traffic.json")],
                         capture_output=True, text=True, check=True).stdout
    open(os.path.join(dst, f"{rnd}_vls128_pmc_traffic.txt"), "w").write(out)
lines = []
for m in ("vls128", "hdl64", "vlp16", "vls128_noevents", "vls128_causal", "vls128_resident", "vls128_hostmaps"):
    p = os.path.join(src, f"bench_{m}.json")
    if os.path.exists(p):
        for l in open(p):
            if l.startswith("{"): lines.append(l.strip())
    p = os.path.join(src, f"cpu_sweep_{m}.log")
    if os.path.exists(p): shutil.copy(p, os.path.join(dst, f"{rnd}_cpu_sweep_{m}.log"))
if os.path.exists(os.path.join(src, "batch_sweep.jsonl")): shutil.copy(os.path.join(src, "batch_sweep.jsonl"), os.path.join(dst, f"{rnd}_batch_sweep.jsonl"))
open(os.path.join(dst, f"{rnd}_bench_lines.jsonl"), "w").write("\n".join(lines) + "\n")
for l in lines:
    d = json.loads(l)
    r = d.get("roofline", {})
    print(d["config"]["workload"].split(",")[0], round(d["value"], 1), d["unit"], "| roofline", r.get("kernel", "-"), r.get("frac", "-"))
