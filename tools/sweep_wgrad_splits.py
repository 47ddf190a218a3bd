"""Developer tool (GPU box): SCN_WGRAD_SPLITS sweep (each value needs a fresh process: the override is read per call)."""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__))
for sp in sys.argv[1:]:
    env = dict(os.environ); env["SCN_WGRAD_SPLITS"] = sp
    r = subprocess.run([sys.executable, os.path.join(here, "ablate_wgrad_direct.py")], env=env, capture_output=True, text=True)
    for line in r.stdout.splitlines():
        if "subm" in line:
            print("splits", sp, line, flush=True)
