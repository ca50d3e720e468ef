#!/bin/bash
# Same positional interface as the reference's benchmark.sh (env, setting, particles, seed index, extra flags):
#   ./benchmark.sh DeepMimicWalk-v0 deepmimic.deepmimic_dppo_pfpn 35 0 --train
# NGPUS=8 runs the reference's 8 synchronous workers as 8 ranks (one per MI355X) under torchrun.
s=$4
env=$1
setting=$2
par=$3
suf=${@:5}
SEEDS=( 34114 33406 28949 12831 39907 )
seed=${SEEDS[s]}
NGPUS=${NGPUS:-1}
if [ "$NGPUS" -gt 1 ]; then
    cmd="python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NGPUS} --master-addr 127.0.0.1 --master-port ${PORT:-29511} -m pfpn_amd.main --env ${env} --seed ${seed} --setting settings.${setting} --particles ${par} ${suf}"
else
    cmd="python -m pfpn_amd.main --env ${env} --seed ${seed} --setting settings.${setting} --particles ${par} ${suf}"
fi
echo $cmd
eval $cmd
