"""Developer tool: busy / idle time of the compute stream per benchmark step from a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace -d gpurun_out/trace -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline
    python tools/stream_gaps.py gpurun_out/trace
Kernels are grouped by queue/stream id; the compute stream is the one that runs k_conv_ts.  A step is delimited by the
multi-tensor SGD kernel that ends it."""
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0])))
key = 'Stream_Id' if 'Stream_Id' in rows[0] else 'Queue_Id'
by = collections.defaultdict(list)
for r in rows:
    by[r[key]].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
main = max(by, key=lambda q: sum(('k_conv_ts<' in k[2]) or ('k_conv_tb' in k[2]) for k in by[q]))
ks = sorted(by[main])
print("streams/queues:", {q: len(v) for q, v in by.items()}, "compute:", main)
# steps: split at the SGD update (multi_tensor_apply) kernels
ends = [i for i, k in enumerate(ks) if 'multi_tensor_apply' in k[2]]
steps = []
prev = None
for e in ends:
    if prev is not None and e - prev > 50:
        steps.append(ks[prev + 1:e + 1])
    prev = e
steps = steps[len(steps) // 3:]                       # drop the warm-up part
tot = busy = 0.0; gaps = []; n = 0
for st in steps:
    t0, t1 = st[0][0], st[-1][1]
    b = 0; cur_end = st[0][0]
    for s, e, _ in st:
        if s > cur_end: gaps.append((s - cur_end) / 1e3)
        b += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
    tot += (t1 - t0) / 1e3; busy += b / 1e3; n += len(st)
ns = len(steps)
print(f"{ns} steps: {n / ns:.0f} kernels/step on the compute stream, span {tot / ns / 1e3:.3f} ms, busy {busy / ns / 1e3:.3f} ms, "
      f"idle {(tot - busy) / ns / 1e3:.3f} ms")
import statistics
g = sorted(gaps)
print(f"gaps between consecutive kernels: n/step {len(g) / ns:.0f}, median {statistics.median(g):.2f} us, mean {sum(g) / len(g):.2f} us, "
      f"p90 {g[int(len(g) * 0.9)]:.2f} us, max {g[-1]:.1f} us; sum of gaps > 20 us per step: {sum(x for x in g if x > 20) / ns / 1e3:.3f} ms")
# where the large gaps are: (kernel before -> kernel after), summed over the steps
where = collections.Counter(); cnt = collections.Counter()
for st in steps:
    cur_end = st[0][0]; prev = st[0][2]
    for s, e, name in st:
        if s - cur_end > 5000:
            k = (prev[:40], name[:40]); where[k] += (s - cur_end) / 1e3; cnt[k] += 1
        if e > cur_end: cur_end = e; prev = name
for k, v in where.most_common(14):
    print(f"  {v / ns:7.1f} us/step in {cnt[k] / ns:4.1f} gaps/step   {k[0]}  ->  {k[1]}")
# gap at the step boundary
b = [(steps[i + 1][0][0] - steps[i][-1][1]) / 1e3 for i in range(len(steps) - 1)]
print("step boundary gap (SGD kernel end -> first kernel of the next step):", [round(x) for x in b], "us")
