# rocprofv3 kernel stats of one quick_time.py configuration (development aid): usage  kstats_one.sh B BF16 TRAIN NAME
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/kstats
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BF16=$2 TRAIN=$3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $4 -- python3 $ROOT/scripts/quick_time.py $1 > $OUT/$4.log 2>&1
python3 - "$OUT/$4_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5:
        print(f'{r["Name"][:64]:64s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:8.1f} min {float(r["MinNs"])/1e3:8.1f}')
PY
rm -f $OUT/*_kernel_trace.csv $OUT/*agent_info.csv
