"""Summarise tools/pmc_sweep.sh passes for mc_sweep_shared_kernel."""
import csv, glob, collections, json, sys
d = sys.argv[1]
out = {}
for p in sorted(glob.glob(f'{d}/pass*/*/*_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if 'mc_sweep' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
dur = []
for p in sorted(glob.glob(f'{d}/pass1/*/*_kernel_trace.csv')):
    for r in csv.DictReader(open(p)):
        if 'mc_sweep' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
            out['VGPR_Count'] = r.get('VGPR_Count'); out['accum_VGPR'] = r.get('Accum_VGPR_Count'); out['LDS'] = r.get('LDS_Block_Size')
out['kernel_ms_under_pmc'] = sum(dur) / max(len(dur), 1)
w = out.get('SQ_WAVES', 1); steps = 252
out['valu_insts_per_wave_step'] = out.get('SQ_INSTS_VALU', 0) / w / steps
out['mfma_insts_per_wave_step'] = out.get('SQ_INSTS_MFMA', 0) / w / steps
out['lds_insts_per_wave_step'] = out.get('SQ_INSTS_LDS', 0) / w / steps
if 'GRBM_GUI_ACTIVE' in out:
    cyc = out['GRBM_GUI_ACTIVE'] / 8
    out['kernel_cycles'] = cyc
    out['simd_cycles_per_wave_step'] = cyc * 1024 / (w * steps)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in out:
        out['mfma_busy_frac'] = out['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024)
print(json.dumps(out, indent=1))
