"""Developer tool (GPU box): bench.py with several builds of the library, alternating, on ONE box.  usage: ab_libs.py a.so b.so ... [-- bench args]"""
import subprocess, sys, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); os.chdir(ROOT)
argv = sys.argv[1:]
libs = argv[:argv.index("--")] if "--" in argv else argv
args = argv[argv.index("--") + 1:] if "--" in argv else []
for rnd in range(3):
    for lib in ["default"] + libs:
        e = dict(os.environ)
        if lib != "default": e["SCN_MI355X_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-extras"] + args, env=e, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(os.path.basename(lib), round(d["ms_per_step"], 3), round(d.get("roofline", {}).get("avg_launch_us", 0), 2), flush=True)
        except Exception:
            print(lib, "failed", r.stderr[-300:], flush=True)
