#!/bin/bash
# round 4, evidence set: the default bench line, the driver's command (wall time), ML, cfg 4 / cfg 5 whole on one GPU, the N = 2 plumbing check
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r04q; mkdir -p $OUT
t0=$(date +%s); timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "driver-command bench exit $? wall $(( $(date +%s) - t0 )) s"
timeout 1200 python bench.py > $OUT/bench_nf_final.json 2> $OUT/bench_nf_final.err; echo "bench (defaults) exit $?"
timeout 900 python bench.py --workload ml --no-cpu-baseline --no-row-sharded > $OUT/bench_ml.json 2> $OUT/bench_ml.err; echo "bench ml exit $?"
timeout 900 python bench.py --workload cfg4 --synth-scaling strong --steps 10 --warmup 2 > $OUT/bench_cfg4_full_1gpu.json 2> $OUT/cfg4.err; echo "cfg4 exit $?"
timeout 900 python bench.py --workload cfg4 --synth-scaling strong --steps 10 --warmup 2 --synth-restricted-forward --no-kernel-roofline > $OUT/bench_cfg4_full_1gpu_restricted.json 2> $OUT/cfg4r.err; echo "cfg4 restricted exit $?"
timeout 900 python bench.py --workload cfg4 --synth-scaling strong --steps 10 --warmup 2 --synth-exchange rs_ag --no-kernel-roofline > $OUT/bench_cfg4_full_1gpu_rs_ag.json 2> $OUT/cfg4rs.err; echo "cfg4 rs_ag exit $?"
timeout 1200 python bench.py --workload cfg5 --synth-scaling strong > $OUT/bench_cfg5_full_1gpu.json 2> $OUT/cfg5.err; echo "cfg5 exit $?"
timeout 1200 python bench.py --workload cfg5 --synth-scaling strong --synth-restricted-forward --no-kernel-roofline --no-parity > $OUT/bench_cfg5_full_1gpu_restricted.json 2> $OUT/cfg5r.err; echo "cfg5 restricted exit $?"
LLMREC_BENCH_SINGLE_DEVICE=1 LLMREC_DIST_BACKEND=gloo timeout 1200 python bench.py --gpus 2 --steps 4 --warmup 1 > $OUT/bench_n2_one_gpu_gloo_smoke.json 2> $OUT/n2.err; echo "n2 exit $?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04q/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        msg={k:d.get(k) for k in ('value','ms_per_step','n_gpus','n_ranks_seen')}
        for k in ('exact_f32','reference_order','pre_propagated_order'):
            if k in d: msg[k]=d[k].get('ms_per_step', d[k].get('error'))
        if 'parity' in d and d['parity']: msg['parity']=d['parity'].get('ok')
        if 'eval' in d: msg['eval_ms']=d['eval']['ms']
        if 'eval_sample' in d: msg['eval_sample']=(d['eval_sample']['ms'], d['eval_sample']['frac_mfma_f32'])
        if 'roofline' in d: msg['roof']=(d['roofline']['frac'], d['roofline'].get('traffic'))
        if 'ingest' in d: msg['hbm_gb']=d['ingest'].get('hbm_peak_gb')
        if 'single_gpu_reference' in d: msg['ref1']=d['single_gpu_reference'].get('ms_per_step')
        if 'row_restricted_forward' in d: msg['restricted']=d['row_restricted_forward'].get('ms_per_step')
        print(f.split('/')[-1], msg)
    except Exception as e: print(f, 'no line', repr(e))
PY
tail -3 $OUT/n2.err | cut -c1-300
