# same-box A/B of diagnostic builds of csrc/rowdense_kernels.hip (wrong results; timing only):
#   RD_DIAG_NO_STAGE: no global loads / LDS stores in the k-loop   RD_DIAG_NO_MFMA: staging only
#   RD_DIAG_NO_LOOP: prologue + epilogue only                     RD_DIAG_NO_EPI: no reduction / epilogue
set -e
cd $GRAFT_REPO_ROOT
for flags in "" "-DRD_DIAG_NO_LOOP" "-DRD_DIAG_NO_EPI" "-DRD_DIAG_NO_LOOP -DRD_DIAG_NO_EPI" ${RD_DIAG_MORE}; do
  rm -rf pfpn_amd/build/rowdense_kernels.o
  PFPN_EXTRA_FLAGS="$flags" python -m pfpn_amd.build > /dev/null 2>&1
  echo "=== flags: $flags"
  PROBE_CFGS="${PROBE_CFGS:-0 1 2}" timeout -k 10 200 python tools/rowdense_gemm_probe.py ${RD_DIAG_ROWS:-32 512} 2>&1 | grep -v amdgpu.ids
done
rm -rf pfpn_amd/build/rowdense_kernels.o; python -m pfpn_amd.build > /dev/null 2>&1
