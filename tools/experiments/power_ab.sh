#!/bin/bash
# Experiment: sample sclk / power while bench.py runs under GEMM policy 2 and 3 (is the chip power-limited in steady state?)
for pol in 2 4 3 2 4; do
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/smi_pol$pol.txt &
  SMI=$!
  VMV_GEMM_POLICY=$pol python bench.py --no-cpu-baseline --no-sample --no-frame-parallel --no-op-profile --steps 60 --warmup 10 > gpurun_out/bench_pw$pol.json 2>/dev/null
  kill $SMI
  python - <<PY
import json,re
d=json.load(open("gpurun_out/bench_pw$pol.json")); print("policy $pol", d["value"], d["ms_per_step"])
L=[l for l in open("gpurun_out/smi_pol$pol.txt") if "sclk" in l]
import statistics
clk=[int(m.group(1)) for l in L for m in [re.search(r"\((\d+)Mhz\)", l)] if m]
pw=[float(m.group(1)) for l in L for m in [re.search(r"Power[^:]*:\s*([\d.]+)", l)] if m]
print("  samples", len(L), "sclk tail", clk[-12:], "power tail", pw[-12:])
print("  raw:", L[-1][:300])
PY
done
