set -x
REPO=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/pk; cd /tmp
run() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $REPO/gpurun_out/pk/$name -- env PBSED_WGRAD_X3=4 ONLY=128x128 python $REPO/tools/gpu_conv_bench.py > /dev/null 2>&1; }
run fetch FETCH_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
cd $REPO
python - <<'PY'
import csv,glob,collections
for name in ('fetch','tcc','sq','sq2'):
    files=glob.glob(f'gpurun_out/pk/{name}/**/*counter_collection.csv', recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:60]
            if 'conv_' in k:
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(name, k, {c:(sum(x)/len(x)) for c,x in v.items()}, 'n', len(next(iter(v.values()))))
PY
