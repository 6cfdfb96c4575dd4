"""Summarise rocprofv3 --pmc passes (tools/pmc_passes.sh) for the path kernel into a small JSON."""
import csv, glob, collections, json, sys
d = sys.argv[1]
out = {}
for p in sorted(glob.glob(f'{d}/pass*/runc/*_counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if 'mc_paths' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
dur = []
for p in sorted(glob.glob(f'{d}/pass1/runc/*_kernel_trace.csv')):
    for r in csv.DictReader(open(p)):
        if 'mc_paths' in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
            out['VGPR_Count'] = r.get('VGPR_Count'); out['SGPR_Count'] = r.get('SGPR_Count'); out['LDS'] = r.get('LDS_Block_Size')
out['kernel_ms_under_pmc'] = sum(dur) / max(len(dur), 1)
w = out.get('SQ_WAVES', 1)
steps = 252
out['valu_insts_per_wave_step'] = out.get('SQ_INSTS_VALU', 0) / w / steps
out['salu_insts_per_wave_step'] = out.get('SQ_INSTS_SALU', 0) / w / steps
out['smem_insts_per_wave_step'] = out.get('SQ_INSTS_SMEM', 0) / w / steps
if 'GRBM_GUI_ACTIVE' in out:
    cyc = out['GRBM_GUI_ACTIVE'] / 8          # summed over 8 XCDs
    out['kernel_cycles'] = cyc
    out['cycles_per_valu_inst_per_simd'] = cyc * 1024 / out['SQ_INSTS_VALU']
    out['sq_cycles_per_wave_step'] = cyc * 1024 / (w * steps)
if 'WRITE_SIZE' in out:
    out['mc_paths_kernel_hbm_bytes_per_launch'] = (out['WRITE_SIZE'] + 2 * out.get('FETCH_SIZE', 0)) * 1024   # KiB units; gfx950 FETCH_SIZE x2 (MI355X_MICROARCH.md HBM)
print(json.dumps(out, indent=1))
