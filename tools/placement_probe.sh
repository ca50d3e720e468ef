# Does the headline step depend on where its buffers land?  bench.py with a dummy allocation of varying size in front.
# GPU box: bash tools/placement_probe.sh
cd ${GRAFT_REPO_ROOT:-.}
for rnd in 1 2; do
for sh in 0 2097152 4194304 6291456 8388608 12582912 16777216 35651584 69206016 1048576 524288; do
  timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --alloc-shift $sh 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('round $rnd shift %9d: %.1f steps/s  (%.1f us, windows %.1f-%.1f)' % ($sh, d['value'], d['ms_per_step']*1e3, d['ms_per_step_min']*1e3, d['ms_per_step_max']*1e3))"
done; done
