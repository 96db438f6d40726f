cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r03b
V=$R/mods-light-zmq_amd/_variants/libmodsgpu_sstride.so
(for rep in 1 2; do for L in "" $V; do echo "== lib '$L'"; rm -rf /tmp/ds; MODS_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ds -- python $R/tools/prof_describe.py > /dev/null 2>&1; f=$(find /tmp/ds -name "*kernel_stats.csv" | head -1); python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('extract_small', 'big_fused', 'sift_wave2', 'orient_kernel')):
        print(r['Name'][:40], r['Calls'], round(float(r['AverageNs']) / 1e3, 1), 'us')
PY
done; done) > $R/gpurun_out/r03b/sstride.log 2>&1
