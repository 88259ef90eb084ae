#!/bin/bash
# SQ instruction-mix counters per kernel (eager replay, one frame mix): usage bash tools/pmc_sq.sh <tag>
set -u
TAG=${1:-sq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --mode eager"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT"; do
	i=$((i+1))
	timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT" -o s$i -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_s$i.json" 2> "$OUT/s$i.err"
done
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, sys, collections, glob
out = sys.argv[1]
agg = collections.OrderedDict()
for path in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        a = agg.setdefault(k, collections.OrderedDict())
        c = a.setdefault(r["Counter_Name"], [0, 0.0])
        c[0] += 1; c[1] += float(r["Counter_Value"])
names = []
for a in agg.values():
    for n in a:
        if n not in names: names.append(n)
with open(out + "/sq_summary.csv", "w") as f:
    f.write("kernel,dispatches," + ",".join(names) + "\n")
    for k, a in agg.items():
        n = max(v[0] for v in a.values())
        f.write('"%s",%d,' % (k, n) + ",".join("%.0f" % (a[c][1] / a[c][0]) if c in a else "" for c in names) + "\n")
PY
find "$OUT" -name '*_counter_collection.csv' -delete; find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*agent_info.csv' -delete
ls "$OUT"; tail -3 "$OUT/s1.err"
