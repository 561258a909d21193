#!/bin/bash
# Diagnostic: per-kernel times of the bench (a) as is, (b) without look-ahead, (c) with another process keeping the GPU
# busy (does the clock state of a mostly idle GPU explain in-pipeline kernel times twice those of back-to-back launches?)
out=gpurun_out/r2c
mkdir -p $out
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs "$@"; }
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "fps", round(d["value"], 1), {k: round(v["us_per_launch"], 1) for k, v in d["kernels"].items() if k in ("match_search", "match_model", "lm_solve", "map_add", "label_nms")})
PY
}
run > $out/p_default.json 2>/dev/null && show $out/p_default.json
(for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | tr '\n' ' '; echo; sleep 0.5; done) > $out/clocks_during.txt &
run > $out/p_default2.json 2>/dev/null && show $out/p_default2.json
wait
run --causal > $out/p_causal.json 2>/dev/null && show $out/p_causal.json
python - <<'PY' &
import time, torch
a = torch.randn(2048, 2048, device="cuda")
t0 = time.time()
while time.time() - t0 < 40:
    for _ in range(50):
        b = a @ a
    torch.cuda.synchronize()
PY
heater=$!
sleep 8
run > $out/p_heated.json 2>/dev/null && show $out/p_heated.json
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | tr '\n' ' '; echo
kill $heater 2>/dev/null
wait
cat $out/clocks_during.txt
