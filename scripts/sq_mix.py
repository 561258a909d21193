"""Instruction mix per wavefront of the hot kernels, from one rocprofv3 --pmc pass
(SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY) of bench.py.
usage: python scripts/sq_mix.py <counter_collection.csv> > profiles/rNN_vls128_sq_mix.txt
SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md): x 4 for cycles."""
import csv, collections, re, sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::|void |lsa::", "", r["Kernel_Name"]).split("(")[0][:34]
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        calls[n] += 1
print("%-34s %5s %8s %7s %7s %6s %7s %10s %8s %7s" % ("kernel", "calls", "waves", "valu/w", "salu/w", "lds/w", "vmrd/w", "cycles/w", "valu act", "waiting"))
for n, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:16]:
    w = c["SQ_WAVES"] or 1
    print("%-34s %5d %8.0f %7.0f %7.0f %6.0f %7.0f %10.0f %7.1f%% %6.1f%%" % (
        n, calls[n], w / calls[n], c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_LDS"] / w, c["SQ_INSTS_VMEM_RD"] / w,
        4 * c["SQ_WAVE_CYCLES"] / w, 100 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
