"""Developer tool (GPU box): A/B of an environment switch through bench.py on ONE box, alternating.
usage: ab_env.py NAME=VALUE [bench args...]   -- `this` runs without the variable, `other` with it."""
import subprocess, sys, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
name, val = sys.argv[1].split("=", 1)
args = sys.argv[2:]
for rnd in range(3):
    for tag, env in (("this", {}), ("other", {name: val})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-extras"] + args, env=e, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(tag, round(d["ms_per_step"], 3), round(d.get("roofline", {}).get("avg_launch_us", 0), 2), flush=True)
        except Exception:
            print(tag, "failed", r.stderr[-400:], flush=True)
