mkdir -p gpurun_out/r3_stress
for frac in 0.05 0.1 0.15 0.2 0.3 1; do
  BARBELL_AMD_ADAPT_FRAC=$frac python bench.py --steps 3 --no-other-configs --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('frac $frac value',round(d['value']/1e6,2), d['kernel_ms_per_step']['k_flank_scan']);
for k,v in d['stress'].items(): print('   ',k, round(v['reads_per_s']/1e6,2), round(v['flagged_fraction'],4), v['scan'], round(v['scan_stage_ms'],2), round(v['barcode_stage_ms'],2))"
done
