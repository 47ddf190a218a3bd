"""Developer tool (CPU): derived figures from the SQ counter passes of tools/collect_sq.sh.
  MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
  wave residency        = 4 x SQ_WAVE_CYCLES (quad-cycles) / (waves in flight x kernel cycles)
  stall split           = SQ_WAIT_ANY (parked on s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue stall: pipe busy, MFMA
                          dependency), SQ_ACTIVE_INST_ANY (issuing) as fractions of SQ_WAVE_CYCLES
usage: sq_report.py <dir with the pass directories> [kernel name filters...]"""
import csv, glob, sys, collections
root, filt = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*counter_collection.csv") + glob.glob(root + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if filt and not any(s in k for s in filt):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_waves"].append(float(r["Grid_Size"]) / 64.0)
print(f"{'kernel':52s} {'n':>4s} {'cycles':>8s} {'MFMA busy':>9s} {'resid.':>7s} {'parked':>7s} {'issue-st':>8s} {'issuing':>7s} "
      f"{'VALU/MFMA':>9s} {'LDS/MFMA':>8s} {'VMEM/MFMA':>9s} {'LDS confl':>9s}")
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if "GRBM_GUI_ACTIVE" not in m or not m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        continue
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    n_mfma = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0                      # 16x16x4 / 32x32x2 fp32: 32 / 64 cycles each
    waves = min(m["_waves"], 256 * 16.0)
    print(f"{k[:52]:52s} {len(d['GRBM_GUI_ACTIVE']):4d} {cyc:8.0f} {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):9.3f} "
          f"{4 * wc / (waves * cyc):7.3f} {m.get('SQ_WAIT_ANY', 0) / max(wc, 1):7.3f} {m.get('SQ_WAIT_INST_ANY', 0) / max(wc, 1):8.3f} "
          f"{m.get('SQ_ACTIVE_INST_ANY', 0) / max(wc, 1):7.3f} {(m.get('SQ_INSTS_VALU', 0)) / n_mfma:9.2f} "
          f"{m.get('SQ_INSTS_LDS', 0) / n_mfma:8.2f} {m.get('SQ_INSTS_VMEM_RD', 0) / n_mfma:9.3f} "
          f"{m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):9.3f}")
