#!/bin/bash
# same-box comparison of mixture-kernel variants: the previous commit's source (tools/ab_prev_mixture_kernels.hip.txt,
# if present) and the current source under each flag set given as arguments 2.. ; probe timings only, 3 repeats each
O=gpurun_out/${1:-var}
shift
mkdir -p $O
: > $O/probe.log
cp pfpn_amd/csrc/mixture_kernels.hip /tmp/cur.hip
# the tracked source is overwritten below: put it back however the script ends
trap 'cp /tmp/cur.hip pfpn_amd/csrc/mixture_kernels.hip' EXIT
if [ -f tools/ab_prev_mixture_kernels.hip.txt ]; then   # (untracked: git show HEAD:pfpn_amd/csrc/mixture_kernels.hip > that file)
  cp tools/ab_prev_mixture_kernels.hip.txt pfpn_amd/csrc/mixture_kernels.hip
  rm -rf pfpn_amd/build && python -m pfpn_amd.build > $O/build.log 2>&1
  for i in 1 2 3; do echo "prev:" >> $O/probe.log; python tools/mix_probe.py 2>&1 | grep -v amdgpu.ids >> $O/probe.log; done
  cp /tmp/cur.hip pfpn_amd/csrc/mixture_kernels.hip
fi
for f in "$@"; do
  rm -rf pfpn_amd/build && PFPN_EXTRA_FLAGS="$f" python -m pfpn_amd.build > $O/build.log 2>&1
  for i in 1 2 3; do echo "cur [$f]:" >> $O/probe.log; python tools/mix_probe.py 2>&1 | grep -v amdgpu.ids >> $O/probe.log; done
done
cat $O/probe.log

# leave the DEFAULT library behind: later bench / test runs must not silently measure a diagnostic or flagged build
rm -rf pfpn_amd/build && python3 -m pfpn_amd.build > /dev/null 2>&1
