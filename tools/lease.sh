#!/bin/bash
# One GPU lease, parameterised (replaces the per-call run scripts of earlier rounds).  Runs ON the GPU box:
#   gpurun --timeout N -- 'bash tools/lease.sh <verb> <args...> [-- <verb> <args...>]...'
# verbs
#   test <pytest args...>            pytest -m gpu on the given files / -k filters (tail of the report)
#   desc <tag> [variant...]          kernel statistics of the describe leg (tools/prof_describe.py under rocprofv3) for the shipped
#                                    library ("base") and / or variants built by tools/variant.sh; summary -> gpurun_out/<tag>/
#   vtest <variant> <pytest args>    the given tests against a variant library (MODS_LIB)
#   bench <tag> [variant...]         bench.py --no-cpu-baseline per library, the JSON line -> gpurun_out/<tag>/
#   run <tag> <command...>           any command, stdout + stderr -> gpurun_out/<tag>/run.log (tail shown)
# Everything that is judged is copied from gpurun_out/ into profiles/ by hand afterwards.
ulimit -c 0
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/mods-light-zmq_amd/_variants
export TMPDIR=/tmp
libof() { if [ "$1" = base ]; then echo ""; else echo "$V/libmodsgpu_$1.so"; fi; }

summarise() {   # <kernel_stats.csv> <label>
  python3 - "$1" "$2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("== %s: total %.3f ms per batch of 16 images" % (sys.argv[2], tot / 4e6))
for r in rows[:12]:
    name = r["Name"].replace("mods::", "").replace("void ", "")
    print("   %-44s calls %4s  %8.3f ms/batch %5.1f%%" % (name[:44], r["Calls"], int(r["TotalDurationNs"]) / 4e6, float(r["Percentage"])))
PY
}

one() {
  local verb=$1; shift
  case $verb in
    test)
      (cd $R && timeout 1500 python -m pytest "$@" -q -m gpu -x 2>&1 | tail -4) ;;
    vtest)
      local v=$1; shift
      echo "== tests against variant $v"
      (cd $R && MODS_LIB=$(libof $v) timeout 1500 python -m pytest "$@" -q -m gpu -x 2>&1 | tail -4) ;;
    desc)
      local tag=$1; shift; local out=$R/gpurun_out/$tag; mkdir -p $out
      [ $# -eq 0 ] && set -- base
      for v in "$@"; do
        rm -rf $out/dstats
        (cd /tmp && MODS_LIB=$(libof $v) timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/dstats -- python $R/tools/prof_describe.py > $out/run_$v.log 2>&1)
        local f=$(find $out/dstats -name "*kernel_stats.csv" | head -1)
        if [ -n "$f" ]; then cp $f $out/describe_leg_kernel_stats_$v.csv; summarise $out/describe_leg_kernel_stats_$v.csv $v; else echo "== $v: no statistics"; tail -5 $out/run_$v.log; fi
        rm -rf $out/dstats
      done ;;
    bench)
      local tag=$1; shift; local out=$R/gpurun_out/$tag; mkdir -p $out
      [ $# -eq 0 ] && set -- base
      for v in "$@"; do
        (cd $R && MODS_LIB=$(libof $v) timeout 600 python bench.py --no-cpu-baseline > $out/bench_$v.json 2> $out/bench_$v.err)
        python3 -c "
import json,sys
d=json.loads(open('$out/bench_$v.json').read().strip().splitlines()[-1])
print('== bench $v: %.1f %s, %.3f ms/step' % (d['value'], d['unit'], d['ms_per_step']))" 2>/dev/null || { echo "== bench $v failed"; tail -3 $out/bench_$v.err; } ;
      done ;;
    run)
      local tag=$1; shift; local out=$R/gpurun_out/$tag; mkdir -p $out
      (cd $R && timeout 1500 "$@" > $out/run.log 2>&1; echo "exit $?" >> $out/run.log); tail -25 $out/run.log ;;
    *) echo "lease.sh: unknown verb $verb"; return 2 ;;
  esac
}

args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then [ ${#args[@]} -gt 0 ] && one "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && one "${args[@]}"
exit 0
