#!/bin/bash
# Mixture tile kernel on the GPU box: parity tests of the mixture paths, probe timings (optionally of a second build
# with the flags in $2), then the stamped build's phase timeline.  Outputs under gpurun_out/$1/.
O=gpurun_out/${1:-ab}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_mixture.py tests/test_gpu_dppo_step.py tests/test_gpu_graph_golden.py -x -q > $O/t.log 2>&1
echo "pytest rc=$?" >> $O/t.log
tail -3 $O/t.log
python tools/mix_probe.py 2>&1 | grep -v amdgpu.ids > $O/probe.log
if [ -n "$2" ]; then
  rm -rf pfpn_amd/build && PFPN_EXTRA_FLAGS="$2" python -m pfpn_amd.build > $O/build.log 2>&1 && python tools/mix_probe.py 2>&1 | grep -v amdgpu.ids > $O/probe_flags.log
fi
cat $O/probe*.log
rm -rf pfpn_amd/build && PFPN_BUILD_STAMPS=1 python -m pfpn_amd.build > $O/build.log 2>&1 && python tools/stamp_probe.py 2>&1 | grep -v amdgpu.ids > $O/stamps.log

# leave the DEFAULT library behind: later bench / test runs must not silently measure a diagnostic or flagged build
rm -rf pfpn_amd/build && python3 -m pfpn_amd.build > /dev/null 2>&1
