cd /tmp && export TMPDIR=/tmp
P=$GRAFT_REPO_ROOT/tools/proto
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/k4b_pmc_$tag -o p --output-format csv -- $P/k4b_proto 100000000 full > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('/root/repo/gpurun_out/k4b_pmc_*/**/p_counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:30]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
    for k,v in acc.items():
        if 'c5_single' in k: print(k, {a:b for a,b in v.items()})
PY
