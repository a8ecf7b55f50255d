#!/bin/bash
# PMC counters of the kernels one command launches (GPU box; counters only, one pass per group):
#   bash tools/prof_kernel.sh <tag> <kernel name substring> <command ...>
# e.g. bash tools/prof_kernel.sh wx conv_winox3 env NOERR=1 PRECS=winox3 ONLY=128x128 python tools/wino_bench.py
REPO=$(pwd); export TMPDIR=/tmp; tag=$1; pat=$2; shift 2
mkdir -p gpurun_out/pk_$tag; cd /tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $REPO/gpurun_out/pk_$tag/$name -- "${CMD[@]}" > /dev/null 2>&1; }
CMD=("$@")
for i in "${!CMD[@]}"; do case "${CMD[$i]}" in tools/*|bench.py) CMD[$i]="$REPO/${CMD[$i]}";; esac; done
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
run mfma SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
cd $REPO
python - "$tag" "$pat" <<'PY'
import csv, glob, collections, sys, json
tag, pat = sys.argv[1:3]
res = collections.defaultdict(dict)
for name in ('fetch', 'write', 'tcc', 'sq', 'sq2', 'mfma'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'gpurun_out/pk_{tag}/{name}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']:
                agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        for c, x in v.items():
            res[k][c] = sum(x) / len(x)
            res[k]['launches_' + name] = len(x)
for k, v in res.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f'   {c:34s} {x:16.1f}')
json.dump(res, open(f'gpurun_out/pk_{tag}/summary.json', 'w'), indent=1)
PY
