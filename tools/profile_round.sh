#!/bin/bash
# Round profile bundle (run on the GPU box through gpurun): bench lines, per-op table, rocprofv3 kernel stats and the
# separate PMC passes; summaries land in gpurun_out/ and are copied to profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
RN=${RN:-r6}      # round tag of the output files; VMV_COMMIT (git hash of the submitted tree) is recorded in the traffic JSON
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
P="--no-cpu-baseline --no-sample --no-op-profile --simulate-rank 0"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_kt -- $B $P --steps 5 --warmup 1 > $O/prof_kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $O/prof_fetch -- $B $P --steps 2 --warmup 1 > $O/prof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $O/prof_write -- $B $P --steps 2 --warmup 1 > $O/prof_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -f csv -d $O/prof_mfma -- $B $P --steps 2 --warmup 1 > $O/prof_mfma.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -f csv -d $O/prof_lds -- $B $P --steps 2 --warmup 1 > $O/prof_lds.log 2>&1
cd $R
python tools/prof_summary.py $O/prof_kt $O/${RN}_kernel_stats.txt
python tools/prof_summary.py $O/prof_fetch $O/${RN}_pmc_fetch.txt
python tools/prof_summary.py $O/prof_write $O/${RN}_pmc_write.txt
python tools/prof_summary.py $O/prof_mfma $O/${RN}_pmc_mfma.txt
python tools/prof_summary.py $O/prof_lds $O/${RN}_pmc_lds.txt
# HBM traffic per GEMM launch, attributed per (kernel, op kind, shape): the plan replayed op by op under the two counters (round 6;
# tools/traffic_by_op.py — the family figure bench.py reports comes from the same passes)
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE -f csv -d $O/tb_fetch -- python $R/tools/traffic_by_op.py run $O/${RN}_traffic_ops.json > $O/tb_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -f csv -d $O/tb_write -- python $R/tools/traffic_by_op.py run $O/${RN}_traffic_ops.json > $O/tb_write.log 2>&1
cd $R
python tools/traffic_by_op.py table $O/${RN}_traffic_ops.json $O/tb_fetch $O/tb_write $O/${RN}_gemm_traffic_by_kernel.tsv $O/${RN}_gemm_traffic.json
rm -rf $O/tb_fetch $O/tb_write
# the bench lines come last: roofline.traffic is read from profiles/${RN}_gemm_traffic.json, i.e. from THIS run's PMC passes
cp $O/${RN}_gemm_traffic.json $R/profiles/${RN}_gemm_traffic.json
cd /tmp
$B > $O/${RN}_bench_40x64.json 2> $O/bench_full.err
$B --latent 32x32 --no-cpu-baseline --no-sample --simulate-rank 0 --dump-ops $O/${RN}_ops_32x32.tsv > $O/${RN}_bench_32x32.json 2>> $O/bench_full.err
$B --no-cpu-baseline --no-sample --simulate-rank 8 --dump-ops $O/${RN}_ops_40x64.tsv --dump-ops-sim $O/${RN}_ops_sim_rank0of8.tsv > $O/${RN}_bench_sim8.json 2>> $O/bench_full.err
for w in 2 4; do $B --no-cpu-baseline --no-sample --no-op-profile --simulate-rank $w > $O/${RN}_bench_sim$w.json 2>> $O/bench_full.err; done
python tools/kernel_resources.py > $O/${RN}_kernel_resources.txt 2>/dev/null || true
cd $R
rm -rf $O/prof_kt $O/prof_fetch $O/prof_write $O/prof_mfma $O/prof_lds
tail -c 600 $O/${RN}_bench_40x64.json
