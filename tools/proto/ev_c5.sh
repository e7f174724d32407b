export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02; mkdir -p $OUT
cd $R
python bench.py --workload c5 --no-cpu-baseline > $OUT/bench_c5.json 2>$OUT/bench_c5.err
python bench.py --workload c5 --steps 3 --warmup 1 > $OUT/bench_c5_cpu.json 2>/dev/null
cd /tmp
rm -rf $OUT/prof_c5 $OUT/pmc_fetch_c5 $OUT/pmc_write_c5 $OUT/sq_c5_*
rocprofv3 --kernel-trace --stats -d $OUT/prof_c5 -o c5 --output-format csv -- python $R/bench.py --workload c5 --no-cpu-baseline > $OUT/prof_c5_bench.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch_c5 -o c5 --output-format csv -- python $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write_c5 -o c5 --output-format csv -- python $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $OUT/sq_c5_$tag -o c5 --output-format csv -- python $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
GDV_NO_OPTFLAT=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch_c5_noopt -o c5 --output-format csv -- python $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
find $OUT -name "*kernel_trace.csv" -size +1000k -delete
cut -c1-300 $OUT/bench_c5.json
