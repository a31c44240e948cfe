#!/bin/bash
# PMC passes (separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes)
REPO=$PWD
python tools/kernel_probe.py all 10
cd /tmp
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rm -rf $REPO/gpurun_out/pmc_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $pass -d $REPO/gpurun_out/pmc_$tag -o p --output-format csv -- python $REPO/tools/kernel_probe.py all 2 > $REPO/gpurun_out/pmc_$tag.log 2>&1
  echo "pmc $tag exit $?"
done
cd $REPO
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][-60:]
            if 'llmrec' not in r['Kernel_Name']: continue
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in agg.items():
            print(d, k, {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
PY
find gpurun_out -name "*.csv" -size +5M -delete
