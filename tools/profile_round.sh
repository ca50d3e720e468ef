#!/bin/bash
# Round profile set (run on the GPU box from the repo root): per-kernel stats of the eager step, one step's timeline
# of the default (hipGraph, tuned GEMMs) run, HBM traffic and SQ counters of the packed mixture kernel.
# Outputs under gpurun_out/prof/ ; tools/profile_collect.py turns them into the files committed under profiles/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-tune --no-graph > $O/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/timeline -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline > $O/timeline.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/mix_probe.py > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/mix_probe.py > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $O/sq1 -- python3 tools/mix_probe.py > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --output-format csv -d $O/sq2 -- python3 tools/mix_probe.py > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/sfetch -- python3 tools/stress_probe.py > $O/sfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/swrite -- python3 tools/stress_probe.py > $O/swrite.log 2>&1
python3 bench.py > $O/bench.log 2>&1
python3 tools/profile_collect.py $O gpurun_out/prof_summary r02 > $O/collect.log 2>&1
tail -3 $O/collect.log
