#!/bin/bash
# One parameterised evidence script for the GPU box (replaces the per-round job lists):
#   gpurun --timeout 3000 -- 'EVIDENCE_COMMIT=<sha> bash tools/evidence.sh <tag> <step> [<step> ...]'
# Everything lands in gpurun_out/<tag>/; copy what is to be judged into profiles/<tag>_*.
# steps:
#   smoke        __graft_entry__.smoke()
#   bench        the driver's command (--gpus 1 --steps 20 --warmup 5) and the default line
#   bench-lite   the default line without the config5 / hbm_bound records (fast)
#   bench-extra  LFR K=28 and ca-AstroPh K=200 lines
#   bench2       bare `python bench.py --gpus 2 --test-one-gpu` (self-spawned ranks on GPU 0, tests' transport, async)
#   prof         rocprofv3 --kernel-trace --stats of the default command (ca-AstroPh K=20) and of LFR K=28
#   prof-large   the same for ca-AstroPh K=200 and the config-5 workload (n=1e6, k=512)
#   pmc          rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the three roofline workloads -> traffic.json + table
#   pmc-small    the same for astroph-k20 only
#   native       tests/test_gpu_native_ranks.py
#   pytest       the whole -m gpu suite          (PYTEST_ARGS adds arguments, e.g. PYTEST_ARGS="-k config5"; PYTEST_PATHS
#                replaces `tests` by a list of files)
#   kscan        tools/k_scan.sh
#   shardcost    per-rank cost-model inputs (tools/shard_cost.py) with per-kernel durations -> shard_cost_model.json
#   pmc-sq       one SQ-counter pass: share of wave cycles parked / issue-stalled / issuing per kernel (config 5, K = 200)
#   pmc-shard    rocprofv3 kernel stats + --pmc FETCH_SIZE / WRITE_SIZE passes of rank 0 of 8 in both sharded layouts -> per-kernel roofline table
#   cli          bench.py's cli_end_to_end record alone
#   forced       bench.py --force-sharded (real librccl, world of one) on ca-AstroPh K=20 and K=200: sharded driver vs plain engine
#   cli5         tools/cli_config5.py (the binary at config-5 size): 4 sweeps with -no-stop, then the default flags
#   benchN       bare `python bench.py --gpus 4|8 --test-one-gpu` (tests' transport)
#   bench8       the same at 8 with ALL seven side records (flow rehearsal of the first real node: order, wall time)
#   cliff        per-kernel times and sweep time on ca-AstroPh at K = 20 / 22 / 24 (the register cliff of the small-K kernels)
#   ab-mid12     A/B of tools/build_variant.sh mid12 -DLPL_MID_KC=12 (K = 21..24 phi in the 4-wave block shape)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
prof_one() {  # name, bench args...
  local name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o k -- python $R/bench.py "$@" > $O/prof_$name.log 2>&1)
  local f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${name}_kernel_stats.csv && head -6 $f | cut -c1-160
  rm -rf $O/prof_$name
}
for step in "$@"; do
  echo "=== $step"
  case $step in
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke.txt ;;
    bench)
      python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_astroph_k20_steps20.json 2> $O/bench.err; tail -c 300 $O/bench_astroph_k20_steps20.json; echo
      python bench.py > $O/bench_astroph_k20.json 2>> $O/bench.err; tail -c 300 $O/bench_astroph_k20.json; echo ;;
    bench-lite) python bench.py --no-hbm-bound --no-config5 > $O/bench_astroph_k20_lite.json 2>> $O/bench.err; tail -c 300 $O/bench_astroph_k20_lite.json; echo ;;
    bench-extra)
      python bench.py --workload lfr-k28 --no-hbm-bound --no-config5 --no-cpu-baseline > $O/bench_lfr_k28.json 2>> $O/bench.err
      python bench.py --workload astroph-k200 --no-hbm-bound --no-config5 --no-cpu-baseline > $O/bench_astroph_k200.json 2>> $O/bench.err
      tail -c 200 $O/bench_lfr_k28.json; echo ;;
    bench2)
      python -c "import __graft_entry__ as g; g.build_test_transport()"
      SVILS_RCCL_LIBRARY=$R/tests/fakerccl/libfakerccl.so FAKERCCL_ASYNC=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 2 --test-one-gpu \
        --extra-list config4_astroph_k200 > $O/bench_gpus2_one_gpu.json 2> $O/bench2.err; echo "rc=$?"; tail -c 400 $O/bench_gpus2_one_gpu.json; echo; tail -5 $O/bench2.err ;;
    prof)
      prof_one astroph_k20 --no-cpu-baseline --no-hbm-bound --no-config5
      prof_one lfr_k28 --workload lfr-k28 --no-cpu-baseline --no-hbm-bound --no-config5 ;;
    prof-large)   # rocprofv3 kernel stats of BASELINE config 4's shape and of config 5 at full size (one GPU)
      prof_one astroph_k200 --workload astroph-k200 --no-cpu-baseline --no-hbm-bound --no-config5 --no-cli
      prof_one mmsb_n1m_k512 --workload mmsb:1000000:512:24 --steps 10 --warmup 2 --reps 0 --no-cpu-baseline --no-hbm-bound --no-config5 --no-cli ;;
    pmc|pmc-small)
      WLS="astroph-k20 synthetic:200000:512:24 mmsb:1000000:512:24"; [ $step = pmc-small ] && WLS="astroph-k20"
      ARGS=""
      for wl in $WLS; do
        w=$(echo $wl | tr ':' '_')
        (cd /tmp && export TMPDIR=/tmp
         timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf_$w -o p -- python $R/tools/kernel_times.py $wl 6 > $O/pmcf_$w.log 2>&1
         timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw_$w -o p -- python $R/tools/kernel_times.py $wl 6 > $O/pmcw_$w.log 2>&1)
        find $O/pmcf_$w $O/pmcw_$w -type f ! -name "*counter_collection.csv" -delete
        ARGS="$ARGS $wl $O/pmcf_$w $O/pmcw_$w"
      done
      python tools/pmc_traffic.py $O/traffic.json $O/hbm_traffic_pmc.txt $ARGS
      rm -rf $O/pmcf_* $O/pmcw_* ;;
    native) timeout 2400 python -m pytest tests/test_gpu_native_ranks.py tests/test_gpu_fakerccl_async.py -q -m gpu --timeout 900 $PYTEST_ARGS > $O/pytest_native.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_native.txt; tail -25 $O/pytest_native.txt ;;
    pytest) timeout 3000 python -m pytest ${PYTEST_PATHS:-tests} -q -m gpu --timeout 900 --durations=15 $PYTEST_ARGS > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -25 $O/pytest_gpu.txt ;;
    kscan) bash tools/k_scan.sh 2>&1 | tee $O/k_scan_astroph.txt ;;
    shardcost)   # the cost model's inputs on one GPU -> shard_cost_model.json (what bench.py prints as `model` beside every N > 1 record)
      (cd /tmp; export TMPDIR=/tmp
       for wl in astroph-k20 astroph-k200 synthetic:200000:512:24 mmsb:1000000:512:24; do
         t=$(echo $wl | tr ':' '_')
         python $R/tools/shard_cost.py $wl 2,4,8 --json $O/shard_cost_model.json 2>/dev/null | tee -a $O/shard_cost_model_inputs.txt
       done
       for wl in astroph-k200 mmsb:1000000:512:24; do
         t=$(echo $wl | tr ':' '_')
         rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$t -o p -- python $R/tools/shard_cost.py $wl 8 > /dev/null 2>&1
         f=$(find $O/prof_$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/shard_rank0of8_kernel_stats_$t.csv
         rm -rf $O/prof_$t
       done) ;;
    pmc-sq)   # where the wave cycles go (parked on memory / issue-stalled / issuing, VALU share): config 5 and config 4's shape on one GPU
      ARGS=""
      for wl in mmsb:1000000:512:24 astroph-k200; do
        w=$(echo $wl | tr ':' '_')
        (cd /tmp && export TMPDIR=/tmp
         timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES \
           --kernel-trace --output-format csv -d $O/sq_$w -o p -- python $R/tools/kernel_times.py $wl 6 > $O/pmcsq_$w.log 2>&1)
        ARGS="$ARGS $wl $O/sq_$w"
      done
      python tools/pmc_sq.py $O/wave_cycles_sq_pmc.txt $ARGS
      rm -rf $O/sq_* ;;
    pmc-shard)   # roofline records of the kernels a SHARDED run executes: rank 0 of 8 of config 5 in both layouts, config 4 K-sharded
      ARGS=""
      for spec in "mmsb:1000000:512:24 kshard 8 0" "mmsb:1000000:512:24 nodeblock 8 0" "astroph-k200 kshard 8 0" "astroph-k200 nodeblock 8 0"; do
        set -- $spec; w=$(echo $1 | tr ':' '_')_$2_r$4of$3
        (cd /tmp && export TMPDIR=/tmp
         timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$w -o p -- python $R/tools/shard_rank.py $spec 6 > $O/shard_rank_$w.log 2>&1
         timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf_$w -o p -- python $R/tools/shard_rank.py $spec 6 > /dev/null 2>&1
         timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw_$w -o p -- python $R/tools/shard_rank.py $spec 6 > /dev/null 2>&1)
        f=$(find $O/st_$w -name '*kernel_stats.csv' | head -1); cp $f $O/shard_${w}_kernel_stats.csv
        find $O/pf_$w $O/pw_$w -type f ! -name "*counter_collection.csv" -delete
        ARGS="$ARGS $w $O/shard_${w}_kernel_stats.csv $O/pf_$w $O/pw_$w"
        tail -1 $O/shard_rank_$w.log
      done
      set -- dummy
      python tools/pmc_sharded.py $O/traffic_sharded.json $O/sharded_roofline_pmc.txt $ARGS
      rm -rf $O/st_* $O/pf_* $O/pw_* ;;
    forced)   # the N > 1 driver of bench.py on the REAL librccl with a world of one (eager window, then hipGraph replay with the collectives captured)
      for wl in astroph-k20 astroph-k200; do
        timeout 600 python bench.py --force-sharded --workload $wl --no-cpu-baseline --no-extra > $O/bench_force_sharded_world1_$wl.json 2>> $O/bench_forced.err
        python - $O/bench_force_sharded_world1_$wl.json <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]   # (librccl prints a banner after the line)
e = d.get("eager_window") or {}
print(d["config"]["workload"][:40], "| sharded ms/sweep %.4f" % d["ms_per_step"], "| plain engine, same box %.4f" % d["n1_same_box"]["ms_per_step"],
      "| ratio %.3f" % (d["ms_per_step"] / d["n1_same_box"]["ms_per_step"]), "| eager %.4f" % e.get("ms_per_step", float("nan")),
      "| replay == eager:", e.get("replayed_end_state_identical"), "| rccl", d["rccl"]["rccl_version"], d["rccl"]["library"])
PY
      done 2>&1 | tee $O/force_sharded_world1.txt ;;
    cli5)     # the drop-in binary at config-5 size: four sweeps with -no-stop, then the default flags to the stop rule
      (export TMPDIR=/tmp; timeout 900 python tools/cli_config5.py 4 2>&1 | tee $O/cli_config5.txt
       timeout 900 python tools/cli_config5.py stop 2>&1 | tee $O/cli_config5_to_stop.txt) ;;
    benchN)   # bare `python bench.py --gpus 4 / 8` in one-GPU test mode (tests' transport; a code-path check, never a measurement)
      python -c "import __graft_entry__ as g; g.build_test_transport()" >/dev/null 2>&1
      for N in 4 8; do
        SVILS_RCCL_LIBRARY=$R/tests/fakerccl/libfakerccl.so FAKERCCL_ASYNC=1 timeout 700 python bench.py --gpus $N --steps 10 --warmup 2 --test-one-gpu \
          --extra-list ksharded_config4_astroph_k200 --no-cpu-baseline > $O/bench_gpus${N}_one_gpu_test_mode.json 2> $O/benchN_err$N.txt
        echo "N=$N rc=$?"; tail -2 $O/benchN_err$N.txt; tail -c 600 $O/bench_gpus${N}_one_gpu_test_mode.json; echo
      done ;;
    bench8)   # bare `python bench.py --gpus 8 --test-one-gpu` with ALL SEVEN side records (tests' transport; a flow rehearsal,
              # never a measurement): the line, the order the records came in, the wall time
      python -c "import __graft_entry__ as g; g.build_test_transport()" >/dev/null 2>&1
      t0=$(date +%s)
      SVILS_RCCL_LIBRARY=$R/tests/fakerccl/libfakerccl.so FAKERCCL_ASYNC=1 timeout 2400 python bench.py --gpus 8 --steps 10 --warmup 2 --test-one-gpu \
        --no-cpu-baseline > $O/bench_gpus8_all_records_one_gpu_test_mode.json 2> $O/bench8_err.txt
      rc=$?; t1=$(date +%s)
      python - $O/bench_gpus8_all_records_one_gpu_test_mode.json $rc $((t1 - t0)) <<'PY' | tee $O/bench8_summary.txt
import json, sys
lines = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")]
print("bare bench.py --gpus 8 --test-one-gpu, all side records: rc=%s wall=%s s, JSON lines: %d" % (sys.argv[2], sys.argv[3], len(lines)))
if lines:
    d = json.loads(lines[-1])
    print("main: value=%s ms_per_step=%s model_ms_per_step=%s error=%s" % (d.get("value"), d.get("ms_per_step"), d.get("model_ms_per_step"), d.get("error")))
    for k, v in (d.get("sharded_extra") or {}).items():
        print("  %-34s %s" % (k, ("ms_per_step %.3f | model %.3f | n1_same_box %.3f" % (v.get("ms_per_step", float("nan")), v.get("model_ms_per_step") or float("nan"), v.get("n1_same_box_ms_per_step") or float("nan"))) if isinstance(v, dict) and "error" not in v else v))
PY
      tail -3 $O/bench8_err.txt ;;
    cli) python bench.py --cli-only > $O/bench_cli_end_to_end.json 2> $O/bench_cli.err; tail -c 1500 $O/bench_cli_end_to_end.json; echo ;;
    cliff) for k in 20 22 24; do python tools/kernel_times.py astroph-k$k 100 2>/dev/null | tail -1; python bench.py --workload astroph-k$k --steps 100 --warmup 5 --reps 20 --no-hbm-bound --no-config5 --no-cpu-baseline --no-cli 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('astroph-k$k graph-replayed sweep %.2f us' % (d['ms_per_step']*1e3))"; done | tee $O/k20_k24_cliff_kernel_times.txt ;;
    ab-mid12) WLS="astroph-k22 astroph-k24 astroph-k20 lfr-k28" bash tools/ab_libs.sh $O/ab_mid_kc12.txt libsvils.so libsvils_mid12.so ;;
    *) echo "unknown step $step" ;;
  esac
done
du -sh $O
