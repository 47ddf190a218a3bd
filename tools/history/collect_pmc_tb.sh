# Developer tool (GPU box): PMC passes (HBM-side bytes, L2 hit rate, SQ stall split, instruction mix) of the bf16 tile kernels over bench.py; summary per workload in gpurun_out/pmc_tb_<workload>/summary.txt
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for wl in cfg5 cfg2; do
O=$R/gpurun_out/pmc_tb_$wl; mkdir -p $O
ARGS="--workload $wl --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $set | tr ' ' '+' | cut -c1-60)
  rocprofv3 --pmc $set --output-format csv -d $O/$tag -o pmc -- python3 $R/bench.py $ARGS > $O/$tag.log 2>&1; echo "$wl pmc [$set] rc=$?"
done
done
cd $R
python - <<'PY'
import csv,glob,collections,os
for wl in ['cfg5','cfg2']:
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'gpurun_out/pmc_tb_{wl}/*/*counter_collection.csv')+glob.glob(f'gpurun_out/pmc_tb_{wl}/*/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0].replace('void ','')
            if 'k_conv_tb' not in k and 'k_wgrad_tb' not in k: continue
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    with open(f'gpurun_out/pmc_tb_{wl}/summary.txt','w') as out:
        for k,d in agg.items():
            out.write(k+'\n')
            for c,v in sorted(d.items()):
                out.write('   %-40s n=%4d mean=%.4g min=%.4g max=%.4g\n'%(c,len(v),sum(v)/len(v),min(v),max(v)))
    os.system(f'cat gpurun_out/pmc_tb_{wl}/summary.txt')
PY
find gpurun_out/pmc_tb_* -name "*counter_collection.csv" -size +5M -delete
