# (GPU box, from the repo root) the default bench line under HIP runtime switches that change how kernels of a hipGraph are dispatched:
# kernel arguments in device memory or not, AQL packet capture for graph launches or not.  Same box, alternating, two passes.
O=gpurun_out/abflags; mkdir -p $O
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-batch64 --no-c3-grid > $O/$name.json 2> $O/$name.err; }
for i in 1 2; do
  run default_$i TSM_NOOP=1
  run devkernarg1_$i HIP_FORCE_DEV_KERNARG=1
  run devkernarg0_$i HIP_FORCE_DEV_KERNARG=0
  run pktcap1_$i DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
  run pktcap0_$i DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/abflags/*.json')):
    try:
        d=json.load(open(f)); print(f, round(d['value']/1e6,2), round(d['ms_per_step'],4), round(d['collect_ms'],4), round(d['ppo_update_ms'],4), d['roofline'].get('grad_step_us'))
    except Exception as e: print(f, 'failed', e)
P
