#!/bin/bash
# End-of-round evidence, regenerated in ONE go from the tree as it is: full GPU suite, smoke, the
# default bench line, rocprofv3 kernel stats of the same command, PMC passes (FETCH_SIZE /
# WRITE_SIZE, separate, kernel-trace only) on the FINAL kernels of C2..C5, the other workloads'
# bench lines + kernel stats, Make latency, small-batch latency, registry-tail timing, the
# in-process multi-device runs.  Raw output: gpurun_out/<round>/ ; condensed into profiles/ by
# tools/summarize_round.py <round>.      ROUND=r03 bash tools/gpu_evidence.sh
export TMPDIR=/tmp
ROUND=${ROUND:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$ROUND; rm -rf $OUT; mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q --timeout 1800 > $OUT/pytest_gpu_full.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu_full.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cut -c1-400 $OUT/bench_c2.json
for w in c1 c3 c4 c5; do python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2>$OUT/bench_$w.err; done
for w in c3 c4 c5; do python bench.py --workload $w --steps 3 --warmup 1 > $OUT/bench_${w}_cpu.json 2>/dev/null; done
cd /tmp
for w in c2 c3 c4 c5; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o $w --output-format csv -- python $R/bench.py --workload $w --no-cpu-baseline > $OUT/prof_${w}_bench.json 2> /dev/null
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch_$w -o $w --output-format csv -- python $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write_$w -o $w --output-format csv -- python $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $OUT/sq_c5_$tag -o c5 --output-format csv -- python $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python tools/make_latency.py > $OUT/make_latency.txt 2>&1
[ -x tools/hbm_ceiling ] && timeout 120 tools/hbm_ceiling > $OUT/hbm_ceiling.txt 2>&1
[ -x tools/small_batch_bench ] && timeout 120 tools/small_batch_bench > $OUT/small_batches.txt 2>&1
python tools/micro_benchmarks.py > $OUT/micro_benchmarks.txt 2>&1
PYTHONPATH=$R timeout 60 python tools/flat_only_timing.py > $OUT/flat_only_plans.txt 2>&1
PYTHONPATH=$R timeout 120 python tools/registry_tail_timing.py > $OUT/registry_tail_timing.txt 2>&1
PYTHONPATH=$R timeout 120 python tools/filter_project_chain.py > $OUT/filter_project_chain.txt 2>&1
# in-process multi-device: N host threads over N device contexts (virtual on a one-GPU box)
for n in 1 2 8; do echo "--inproc --gpus $n: $(timeout 300 python bench.py --inproc --gpus $n --steps 10 --warmup 2 2>&1 | tail -1 | cut -c1-420)"; done > $OUT/inproc_bench.txt
find $OUT -name "*kernel_trace.csv" -size +1000k -delete   # raw traces are large; stats / counters stay
find $OUT -name "*.csv" -size +8000k -delete
du -sh $OUT; ls $OUT | head -60
