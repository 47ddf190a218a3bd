mkdir -p gpurun_out/ab
B="python bench.py --workload cfg3 --dtype bf16 --steps 30 --warmup 8 --no-cpu-baseline --no-extras"
for i in 1 2; do
$B > gpurun_out/ab/a$i.json 2>/dev/null && \
SCN_PYRAMID_ONE_STREAM=1 $B > gpurun_out/ab/b$i.json 2>/dev/null && \
SCN_WD_NO_T3=1 $B > gpurun_out/ab/c$i.json 2>/dev/null && \
SCN_EXEC=0 $B > gpurun_out/ab/d$i.json 2>/dev/null || exit 1
done
python bench.py --dtype bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/ab/e1.json 2>/dev/null && \
SCN_EXEC=0 python bench.py --dtype bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/ab/f1.json 2>/dev/null && \
python bench.py --dtype bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/ab/e2.json 2>/dev/null && \
SCN_EXEC=0 python bench.py --dtype bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/ab/f2.json 2>/dev/null
for f in gpurun_out/ab/*.json; do python -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', round(d['ms_per_step'],3))"; done
