"""Developer tool (GPU box): A/B bench.py variants through environment switches in one gpurun call."""
import subprocess, sys, json, os
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
runs = [({"SCN_BENCH_NO_TIMER": "1"}, 20), ({}, 10), ({}, 20), ({}, 20), ({}, 40), ({}, 100)]
for env, steps in runs:
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", str(steps), "--warmup", "5"], env=e,
                       capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(env, steps, round(d["ms_per_step"], 3), d.get("roofline", {}).get("achieved"), flush=True)
    except Exception:
        print(env, steps, "failed", r.stderr[-300:])
