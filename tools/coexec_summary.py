"""Summarise a rocprofv3 --pmc run of tools/microbench/mfma_coexec: per configuration the kernel time of MFMA alone, fillers
alone and both (profiler timestamps), and the SQ_VALU_MFMA_* counters.   python tools/coexec_summary.py <counter_collection.csv>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    e = d.setdefault(r['Dispatch_Id'], {'k': r['Kernel_Name'], 'wg': int(r['Workgroup_Size']), 't': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3})
    e[r['Counter_Name']] = float(r['Counter_Value'])
FILL = {0: 'none', 1: 'v_fma_f32', 2: 'v_mad_u64_u32', 3: 'v_bitop3_b32', 4: 'ds_read_b128', 5: 'v_pk_fma_f32', 6: 'mad64+bitop3'}
MF = {0: '-', 1: 'v_mfma_f32_32x32x2_f32', 2: 'v_mfma_f32_16x16x4_f32', 3: 'v_mfma_f32_32x32x16_bf16 (control)'}
seen = {}
for k, v in d.items():
    m = re.search(r'coexec<(\d+), *(\d+), *(\d+)>', v['k'])
    if m:
        seen[(tuple(map(int, m.groups())), v['wg'] // 256)] = v          # the last (warm) dispatch of each configuration
print("rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -- tools/microbench/mfma_coexec")
print("one workgroup of 4*w waves per CU (w waves co-resident on every SIMD), 1024 iterations x 8 MFMA, NF fillers behind every MFMA; kernel time in us (profiler timestamps)")
print("hidden = 1 - (t_both - max(t_mfma, t_fill)) / min(t_mfma, t_fill): 1 = the cheaper stream disappears behind the other, 0 = the times add")
print(f"{'mfma':34s} {'filler':14s} NF  w | t_mfma  t_fill  t_both     sum     max hidden | MFMA_BUSY_CYCLES MFMA_COEXEC_CYCLES")
for (cfg, w), v in seen.items():
    mf, fill, nf = cfg
    if mf == 0 or fill == 0:
        continue
    a = seen[((mf, 0, 0), w)]['t']; b = seen[((0, fill, nf), w)]['t']; c = v['t']
    print(f"{MF[mf]:34s} {FILL[fill]:14s} {nf:2d} {w:2d} | {a:6.1f} {b:7.1f} {c:7.1f} {a + b:7.1f} {max(a, b):7.1f} {1 - (c - max(a, b)) / min(a, b):6.2f} | "
          f"{v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):16.0f} {v.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0):14.0f}")
