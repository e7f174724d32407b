export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc5; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "^\s*(Name|Counter_Name)\s*:\s*[A-Za-z0-9_]+" | awk '{print $NF}' | sort -u > $OUT/avail.txt
wc -l $OUT/avail.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_GDS" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d $OUT/$tag -o c5 --output-format csv -- python $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>$OUT/$tag.err
done
python3 - $OUT <<'PY'
import csv, glob, sys
acc = {}
for p in glob.glob(sys.argv[1] + "/**/c5_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Kernel_Name"].startswith("gdv_k_"):
            a = acc.setdefault(r["Counter_Name"], [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print(f"{k:42s} per launch {v / n:.4g}")
PY
