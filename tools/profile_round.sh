#!/bin/bash
# Round profile set (run on the GPU box from the repo root): per-kernel stats of the eager step, one step's timeline
# of the default (hipGraph, tuned GEMMs) run, HBM traffic and SQ counters of the mixture tile kernel, the bench line,
# and (last: it rebuilds the library with -DPFPN_ENABLE_STAMPS) the in-kernel phase stamps.
# Outputs under gpurun_out/prof/ ; tools/profile_collect.py turns them into the files committed under profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
TAG=${1:-r05}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-tune --no-graph > $O/stats.log 2>&1
echo "stats done" > $O/progress.log
rocprofv3 --kernel-trace --output-format csv -d $O/timeline -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/timeline.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/probe_trace -- python3 tools/mix_probe.py > $O/probe_trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/sac_trace -- python3 tools/sac_trace.py > $O/sac_trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/c5_trace -- python3 tools/config5_trace.py > $O/c5_trace.log 2>&1
echo "traces done" >> $O/progress.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/gemm_mfma -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/gemm_mfma.log 2>&1
echo "gemm mfma pmc done" >> $O/progress.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/mix_probe.py > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/mix_probe.py > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $O/sq1 -- python3 tools/mix_probe.py > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --output-format csv -d $O/sq2 -- python3 tools/mix_probe.py > $O/sq2.log 2>&1
echo "mix pmc done" >> $O/progress.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/sfetch -- python3 tools/stress_probe.py > $O/sfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/swrite -- python3 tools/stress_probe.py > $O/swrite.log 2>&1
echo "stress pmc done" >> $O/progress.log
python3 bench.py > $O/bench.log 2>&1
echo "bench done" >> $O/progress.log
rm -rf pfpn_amd/build && PFPN_BUILD_STAMPS=1 python3 -m pfpn_amd.build > $O/stamp_build.log 2>&1 && python3 tools/stamp_probe.py 2>&1 | grep -v amdgpu.ids > $O/stamps.txt
python3 tools/profile_collect.py $O gpurun_out/prof_summary $TAG > $O/collect.log 2>&1
python3 tools/gemm_mfma_pmc.py $O/gemm_mfma gpurun_out/prof_summary/${TAG}_gemm_mfma.json >> $O/collect.log 2>&1
python3 tools/step_timeline.py $(ls $O/sac_trace/*/*kernel_trace.csv | head -1) k_act_bwd_partial_value > gpurun_out/prof_summary/${TAG}_sac_step_timeline.txt 2>> $O/collect.log
python3 tools/step_timeline.py $(ls $O/c5_trace/*/*kernel_trace.csv | head -1) k_clip_adam > gpurun_out/prof_summary/${TAG}_config5_step_timeline.txt 2>> $O/collect.log
for d in stats timeline probe_trace sac_trace c5_trace gemm_mfma fetch write sq1 sq2 sfetch swrite; do rm -rf $O/$d; done   # raw traces: large; the summaries are what is kept
tail -3 $O/collect.log

# leave the DEFAULT library behind: later bench / test runs must not silently measure a diagnostic or flagged build
rm -rf pfpn_amd/build && python3 -m pfpn_amd.build > /dev/null 2>&1
