cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "512 1" "1024 0"; do
  set -- $spec
  for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM"; do
    rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pm -- python3 $R/tools/rowdense_pmc.py $1 $2 > /tmp/pm.log 2>&1
    f=$(find /tmp/pm -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$1 $2" <<'PY'
import csv, sys
from collections import defaultdict
acc=defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_dense_jobs" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v[10:])/max(1,len(v[10:]))) for k,v in acc.items()})
PY
  done
done
