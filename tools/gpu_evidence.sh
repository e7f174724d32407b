#!/bin/bash
# End-of-round evidence, regenerated in ONE go from the tree as it is: full GPU suite, smoke, the
# default bench line, rocprofv3 kernel stats of the same command, PMC passes (FETCH_SIZE /
# WRITE_SIZE, separate, kernel-trace only) on the FINAL kernels of C2..C5, the other workloads'
# bench lines + kernel stats, Make latency, small-batch latency, registry-tail timing, the
# in-process multi-device runs; round 4: the fused filter-project, C5 on non-ASCII columns, C4 run
# three times plain and three times under rocprofv3 on this one box (the 9 % gap of round 3).
# Raw output: gpurun_out/<round>/ ; condensed into profiles/ by tools/summarize_round.py <round>.
#      ROUND=r05 bash tools/gpu_evidence.sh
# Round 5: + the filter -> string-projection chain (selection-mode wave kernels), the fused filter-project's shapes side by
# side; the C4 plain / rocprof repeats are gone (the "two box states" are diagnosed: profiles/r05_box_states.txt) and the
# counter passes run on ONE placement (--placements 1: counters are per launch, the placement search only adds launches).
export TMPDIR=/tmp
ROUND=${ROUND:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
# WORKLOADS="c5" + NO_EXTRAS=1: re-take one workload's bench / rocprofv3 / PMC files after a change that renamed only its
# kernels (the files merge into the round's directory; the other workloads' evidence stays valid — kernel identity).
WL=${WORKLOADS:-"c2 c3 c4 c5 k2f"}
OUT=$R/gpurun_out/$ROUND; [ -z "$WORKLOADS" ] && rm -rf $OUT; mkdir -p $OUT
cd $R
if [ -z "$SKIP_TESTS" ]; then
  ls gandiva_amd/_kcache 2>/dev/null | sort > $OUT/kcache_before.txt
  python -m pytest tests -m gpu -q --timeout 1800 --durations=15 > $OUT/pytest_gpu_full.log 2>&1
  grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu_full.log | tail -5
  # Round 6: the code objects this run compiled travel back (gpurun_out/<round>/kcache, <= 64 MiB): copied into
  # gandiva_amd/_kcache they spare the next run of the suite its hipRTC compilations (keyed by kernel name + library hash:
  # a stale file is never loaded)
  mkdir -p $OUT/kcache
  ls gandiva_amd/_kcache | sort | comm -13 $OUT/kcache_before.txt - | head -3000 | while read f; do cp gandiva_amd/_kcache/$f $OUT/kcache/; done
  du -sh $OUT/kcache | tee $OUT/kcache_size.txt
  # ... and the whole suite once more with every plan that has a tier-0 program forced onto the interpreter kernel
  if [ -z "$SKIP_TIER0" ]; then
    GDV_FORCE_TIER0=1 python -m pytest tests -m gpu -q --timeout 1800 -p no:cacheprovider > $OUT/pytest_gpu_tier0.log 2>&1
    { echo "# GDV_FORCE_TIER0=1 python -m pytest tests -m gpu: every Projector / Filter plan inside the tier-0 core is INTERPRETED (gdv_tier0.hip), always;"; echo "# plans outside it run their specialised kernels as before.  Same tests, same oracle."; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu_tier0.log | tail -5; } > $OUT/pytest_gpu_tier0.txt
    cat $OUT/pytest_gpu_tier0.txt
  fi
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
# the driver's own command: C2 headline + every other BASELINE config and K2F under "workloads"
case " $WL " in *" c2 "*) ( time python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err ) 2> $OUT/bench_c2_time.txt; cut -c1-400 $OUT/bench_c2.json;; esac
[ -z "$WORKLOADS" ] && BENCH_WL="$WL c1" || BENCH_WL="$WL"
for w in c1 c3 c4 c5 k2f; do case " $BENCH_WL " in *" $w "*) python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2>$OUT/bench_$w.err;; esac; done
cd /tmp
for w in $WL; do
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o $w --output-format csv -- python $R/bench.py --workload $w --no-extras --no-cpu-baseline --no-verify > $OUT/prof_${w}_bench.json 2> /dev/null
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch_$w -o $w --output-format csv -- python $R/bench.py --workload $w --steps 3 --warmup 1 --pool-candidates 0 --no-extras --no-cpu-baseline --no-verify > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write_$w -o $w --output-format csv -- python $R/bench.py --workload $w --steps 3 --warmup 1 --pool-candidates 0 --no-extras --no-cpu-baseline --no-verify > /dev/null 2>&1
done
# a workload whose kernels were renamed since the last round of counters: its first bench line above could not quote the (then
# stale) profiles/pmc_<w>.json.  Condense the counters just taken into profiles/ on THIS box and take the line again.
cd $R; python tools/summarize_round.py $ROUND > /dev/null 2>&1
for w in $WL; do
  if python -c "import json,sys; sys.exit(0 if json.load(open('$OUT/bench_$w.json'))['roofline'].get('traffic') is None else 1)" 2>/dev/null; then
    if [ $w = c2 ]; then python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
    else python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2>$OUT/bench_$w.err; fi
  fi
done
cd /tmp
case " $WL " in *" c5 "*) ;; *) SKIP_SQ=1;; esac
[ -z "$SKIP_SQ" ] && for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $OUT/sq_c5_$tag -o c5 --output-format csv -- python $R/bench.py --workload c5 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-verify > /dev/null 2>&1
done
cd $R
if [ -z "$NO_EXTRAS" ]; then
python tools/make_latency.py > $OUT/make_latency.txt 2>&1
[ -x tools/hbm_ceiling ] && timeout 120 tools/hbm_ceiling > $OUT/hbm_ceiling.txt 2>&1
[ -x tools/small_batch_bench ] && timeout 120 tools/small_batch_bench > $OUT/small_batches.txt 2>&1
python tools/micro_benchmarks.py > $OUT/micro_benchmarks.txt 2>&1
PYTHONPATH=$R timeout 60 python tools/flat_only_timing.py > $OUT/flat_only_plans.txt 2>&1
PYTHONPATH=$R timeout 120 python tools/registry_tail_timing.py > $OUT/registry_tail_timing.txt 2>&1
PYTHONPATH=$R timeout 200 python tools/filter_project_chain.py 2>&1 | grep -v amdgpu.ids > $OUT/filter_project_chain.txt
PYTHONPATH=$R timeout 200 python tools/fused_fp_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/filter_project.txt
FP_VARIANTS=0,1,3,7 PYTHONPATH=$R timeout 300 python tools/fused_fp_sweep.py 1000000000 250 60 2>&1 | grep -v amdgpu.ids > $OUT/filter_project_shapes.txt
PYTHONPATH=$R timeout 200 python tools/filter_string_chain.py 2>&1 | grep -v amdgpu.ids > $OUT/filter_string_chain.txt
GDV_NO_SEL_WAVE=1 PYTHONPATH=$R timeout 200 python tools/filter_string_chain.py 2>&1 | grep -v amdgpu.ids | sed 's/^/[scanner shape, rounds 2-4] /' >> $OUT/filter_string_chain.txt
# the fused kernel's HBM traffic against the chain's (two --pmc passes) and its rocprofv3 kernel stats
timeout 300 bash tools/fused_fp_traffic.sh $OUT/fp_traffic > /dev/null 2>&1; cp $OUT/fp_traffic/fp_traffic.txt $OUT/filter_project_traffic.txt 2>/dev/null
( cd /tmp; rocprofv3 --kernel-trace --stats -d $OUT/prof_fp -o fp --output-format csv -- python $R/tools/fused_fp_timing.py > /dev/null 2>&1 )
PYTHONPATH=$R timeout 300 python tools/c5_nonascii.py 2>&1 | grep -v amdgpu.ids > $OUT/c5_nonascii.txt
( cd /tmp; rocprofv3 --kernel-trace --stats -d $OUT/prof_c5na -o na --output-format csv -- python $R/tools/c5_nonascii_profile.py > /dev/null 2>&1 )
grep '^"gdv_k_' $(find $OUT/prof_c5na -name "*kernel_stats.csv" | head -1) | awk -F, '{printf "rocprofv3, 1 %% non-ASCII rows: %s average %.4f ms over %s calls\n", $1, $4/1e6, $2}' >> $OUT/c5_nonascii.txt
# in-process multi-device: N host threads over N device contexts (virtual on a one-GPU box)
for n in 1 2 8; do echo "--inproc --gpus $n: $(timeout 300 python bench.py --inproc --gpus $n --steps 10 --warmup 2 2>&1 | tail -1 | cut -c1-420)"; done > $OUT/inproc_bench.txt
# the N-rank launch path on this one-GPU box (gloo: the ranks share cuda:0 — control flow, not scaling)
{ echo "GDV_BENCH_BACKEND=gloo bench.py --gpus 2 --rows 16777216 (weak): $(GDV_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --rows 16777216 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | cut -c1-700)"
  for w in c2 c3 c4; do echo "GDV_BENCH_BACKEND=gloo bench.py --gpus 2 --workload $w --scaling strong --rows 33554432: $(GDV_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --workload $w --scaling strong --rows 33554432 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | cut -c1-700)"; done; } > $OUT/bench_two_ranks.txt
fi
if [ -n "$NO_EXTRAS" ]; then case " $WL " in *" c5 "*)
  PYTHONPATH=$R timeout 300 python tools/c5_nonascii.py 2>&1 | grep -v amdgpu.ids > $OUT/c5_nonascii.txt
  ( cd /tmp; rocprofv3 --kernel-trace --stats -d $OUT/prof_c5na -o na --output-format csv -- python $R/tools/c5_nonascii_profile.py > /dev/null 2>&1 )
  grep '^"gdv_k_' $(find $OUT/prof_c5na -name "*kernel_stats.csv" | head -1) | awk -F, '{printf "rocprofv3, 1 %% non-ASCII rows: %s average %.4f ms over %s calls\n", $1, $4/1e6, $2}' >> $OUT/c5_nonascii.txt
  PYTHONPATH=$R timeout 200 python tools/filter_string_chain.py 2>&1 | grep -v amdgpu.ids > $OUT/filter_string_chain.txt
  GDV_NO_SEL_WAVE=1 PYTHONPATH=$R timeout 200 python tools/filter_string_chain.py 2>&1 | grep -v amdgpu.ids | sed 's/^/[scanner shape, rounds 2-4] /' >> $OUT/filter_string_chain.txt
  PYTHONPATH=$R timeout 120 python tools/registry_tail_timing.py > $OUT/registry_tail_timing.txt 2>&1;; esac; fi
find $OUT -name "*kernel_trace.csv" -size +1000k -delete   # raw traces are large; stats / counters stay
find $OUT -name "*.csv" -size +8000k -delete
du -sh $OUT; ls $OUT | head -70
