# A/B of the backward wide (N >= 64) mixture kernel's cache policy and chunk count on the stress shape, alternating
# between the settings in one session (box drift over a session is larger than the effect).
# GPU box: bash tools/wide_sweep.sh      (PFPN_WIDE_STREAM / PFPN_WIDE_C are read by the library at load)
cd ${GRAFT_REPO_ROOT:-.}
for round in 1 2 3; do
  for S in 0 1; do
    for C in 0 14 16 18 24; do
      echo "round $round stream=$S C=$C $(PFPN_WIDE_STREAM=$S PFPN_WIDE_C=$C timeout -k 10 100 python tools/stress_probe.py 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("fwd %.1f us %.3f  bwd %.1f us %.3f  copy %.2f TB/s" % (d["fwd"]["us"], d["fwd"]["frac_of_8TBps"], d["bwd"]["us"], d["bwd"]["frac_of_8TBps"], d["torch_copy"]["GBps"]/1e3))')"
    done
  done
done
