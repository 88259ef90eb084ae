cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/pmc_icache -o pmc --output-format csv -- python3 $R/tools/enc_run.py --width 832 --height 480 --frames 2 > $R/gpurun_out/pmc_icache.log 2>&1
python3 - <<PY
import csv, collections
acc=collections.defaultdict(float)
for r in csv.DictReader(open("$R/gpurun_out/pmc_icache/pmc_counter_collection.csv")):
    if 'k_encode_ctus' in r['Kernel_Name']: acc[r['Counter_Name']]+=float(r['Counter_Value'])
print({k:int(v) for k,v in acc.items()})
PY
