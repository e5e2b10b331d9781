#!/bin/bash
# Experiment (round 2): sclk / power while the step loop runs on the fp16 and the bf16 kernels — is the fp16 build's 3 % a clock effect?
for dt in fp16 bf16 fp16 bf16; do
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/smi_$dt.txt &
  SMI=$!
  VMV_DTYPE=$dt python bench.py --no-cpu-baseline --no-sample --no-frame-parallel --no-op-profile --no-alt-dtype --steps 80 --warmup 10 2>/dev/null | head -1 > gpurun_out/bench_pw_$dt.json
  kill $SMI
  python - <<PY
import json,re,statistics
d=json.loads(open("gpurun_out/bench_pw_$dt.json").readline()); print("$dt", d["value"], d["ms_per_step"])
L=[l for l in open("gpurun_out/smi_$dt.txt") if "sclk" in l]
clk=[int(m.group(1)) for l in L for m in [re.search(r"\((\d+)Mhz\)", l)] if m]
pw=[float(m.group(1)) for l in L for m in [re.search(r"Power[^:]*:\s*([\d.]+)", l)] if m]
k=max(1,len(clk)//3)
print("  samples", len(L), "sclk median (last 2/3)", statistics.median(clk[k:]) if clk[k:] else None, "power median", statistics.median(pw[k:]) if pw[k:] else None)
PY
done
