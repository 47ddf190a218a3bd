"""Developer tool (CPU): per-kernel statistics and (optionally) the launch timeline of one window from a rocprofv3
rocpd database (`rocprofv3 --kernel-trace` without --output-format csv writes <name>_results.db)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'info_kernel_symbol' in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
namecol = 'display_name' if 'display_name' in cols else ('kernel_name' if 'kernel_name' in cols else cols[-1])
names = dict(cur.execute(f"select id, {namecol} from {ks}"))
rows = list(cur.execute(f"select kernel_id, start, end, grid_size_x, workgroup_size_x from {kd} order by start"))
short = lambda s: (s.split('(')[0].replace('void ', ''))[:70]
st = collections.defaultdict(list)
for k, s, e, g, w in rows: st[short(names[k])].append((e - s) / 1e3)
tot = sum(sum(v) for v in st.values())
print(f"{len(rows)} dispatches, {tot/1e3:.3f} ms of kernel time")
for n, v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:70s} {len(v):6d} {sum(v)/1e3:9.3f} ms  avg {sum(v)/len(v):8.1f} us  min {min(v):7.1f} max {max(v):7.1f}")
if len(sys.argv) > 3:          # timeline of dispatches [a, b)
    a, b = int(sys.argv[2]), int(sys.argv[3]); t0 = rows[a][1]
    prev = t0
    for k, s, e, g, w in rows[a:b]:
        print(f"{(s-t0)/1e3:9.1f} us  +gap {(s-prev)/1e3:6.1f}  dur {(e-s)/1e3:7.1f}  grid {g//max(w,1):6d}x{w:4d}  {short(names[k])}")
        prev = e
