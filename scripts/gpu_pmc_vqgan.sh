R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcv
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcv -o p1 -- python $R/scripts/bench_vqgan.py ${PMCV_FRAMES:-32} 2>&1 | tail -3) > $R/gpurun_out/pmcv.log
cd $R
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/pmcv/p1_counter_collection.csv')))
agg=collections.OrderedDict()
for r in rows:
    k=r['Kernel_Name'].split('(')[0]
    if 'lwm::' not in k: continue
    key=(k.replace('lwm::',''), r['Grid_Size'], r['Dispatch_Id'])
    agg.setdefault(key,{})[r['Counter_Name']]=float(r['Counter_Value'])
seen=set()
for (k,g,d),v in agg.items():
    if (k,g) in seen: continue
    seen.add((k,g))
    gui=v.get('GRBM_GUI_ACTIVE',0)/8
    print(f"{k:22s} grid {int(g)//256:6d} cyc/xcd {gui:10.0f} mfma_util {v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/max(gui,1):.3f} lds_conf/idx {v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_LDS_IDX_ACTIVE',1),1):.3f} wait_any {v.get('SQ_WAIT_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1):.2f} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1):.2f} active {v.get('SQ_ACTIVE_INST_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1):.2f}")
PY
