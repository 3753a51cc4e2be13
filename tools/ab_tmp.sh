for i in 1 2; do for e in 1 0; do UCOD_DBA_EXACT_F32=$( [ $e = 1 ] && echo 1 ) python bench.py --steps 40 --warmup 8 --no-cpu-baseline --lora-steps -1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('exact=$e', d['value'], d['ms_per_step'], d['kernels']['dba_project_f32']['avg_us'])"; done; done
