HERRO_FC_G=3 timeout 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "forward_vs_twin or tiling_edges or full_width or sibling" 2>&1 | tail -2
q="--no-cpu-baseline --self-check 2 --e2e-jobs 0 --strong-windows 0 --repeats 1 --long-run-steps 0"
for g in 4 0 4 0; do HERRO_FC_G=$g timeout 100 python bench.py $q --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print('G=$g driver', round(d['value']), {k:round(v['avg_us']) for k,v in d['kernels'].items() if k in ('fc_gemm','conv_fused','layers_fused')}, d['self_check']['ok'])"; done
for g in 4 0; do HERRO_FC_G=$g timeout 100 python bench.py $q 2>/dev/null | python -c "
import sys,json
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); print('G=$g default', round(d['value']), {k:round(v['avg_us']) for k,v in d['kernels'].items() if k in ('fc_gemm','conv_fused','layers_fused')}, d['self_check']['ok'])"; done
