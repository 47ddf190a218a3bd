"""Developer tool (GPU box): A/B of two builds of the library on ONE box, alternating, through bench.py.
usage: ab_lib.py <other.so> [bench args...]   -- `other` may lack entry points newer than it (they are dropped)."""
import subprocess, sys, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
other, args = sys.argv[1], sys.argv[2:]
code = ("import sys, ctypes, os; sys.path.insert(0, %r); import sparse_rcnn_amd._lib as L\n"
        "l = ctypes.CDLL(L.LIB_PATH)\n"
        "[L._SIGS.pop(k) for k in list(L._SIGS) if not hasattr(l, k)]\n"
        "import runpy; sys.argv = ['bench.py'] + %r; runpy.run_path('bench.py', run_name='__main__')\n") % (ROOT, args + ["--no-cpu-baseline", "--no-extras"])
for rnd in range(3):
    for name, lib in (("this", None), ("other", other)):
        e = dict(os.environ)
        if lib: e["SCN_MI355X_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            print(name, round(d["ms_per_step"], 3), round(d.get("roofline", {}).get("avg_launch_us", 0), 2), flush=True)
        except Exception:
            print(name, "failed", r.stderr[-400:], flush=True)
