# PMC passes for the roofline kernel (separate passes, --kernel-trace only; MI355X_MICROARCH.md HBM section) + the
# FETCH_SIZE calibration kernels.  Run on the GPU box:  bash tools/pmc_conv_r2.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_pmc
rm -rf $O; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 $R/tools/micro/fetch_calib.hip -o /tmp/fetch_calib 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib -- /tmp/fetch_calib > $O/calib.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/roofline_conv.py > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/roofline_conv.py > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- python3 $R/tools/roofline_conv.py > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/sq2 -- python3 $R/tools/roofline_conv.py > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/roofline_conv.py > $O/stats.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, json, os
O = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out', 'r2_pmc')
def counters(d, kern):
    out = {}
    for f in glob.glob(os.path.join(O, d, '*', '*counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if kern in r['Kernel_Name']:
                out.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    return out
res = {}
for k in ('stream_kernel', 'segments_kernel'):
    c = counters('calib', k)
    res['calib_' + k] = {n: [round(x, 1) for x in v] for n, v in c.items()}
for d in ('fetch', 'write', 'sq1', 'sq2'):
    c = counters(d, 'conv3x3_halo_kernel')
    res[d] = {n: {'launches': len(v), 'avg': sum(v) / len(v), 'min': min(v), 'max': max(v)} for n, v in c.items()}
for f in glob.glob(os.path.join(O, 'stats', '*', '*kernel_stats.csv')):
    res['stats'] = [r for r in csv.DictReader(open(f)) if 'conv3x3' in r['Name']]
json.dump(res, open(os.path.join(O, 'summary.json'), 'w'), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
rm -rf $O/*/*/*.db
