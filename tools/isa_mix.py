"""Instruction mix of the innermost (T-step) loop of one kernel in an AMDGPU assembly listing.
   python tools/isa_mix.py file.s [substring of the mangled kernel name]     (hipcc -S --cuda-device-only ... -o file.s)"""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else "mc_paths_kernelILi4ELi1ELi1ELb0ELb0E"
k0 = next(i for i, l in enumerate(lines) if l.startswith("_ZN3mcp") and pat in l and l.rstrip().endswith(("PathArgsE", ":")) or (l.startswith("_ZN3mcp") and pat in l and ":" in l))
k1 = next(i for i in range(k0, len(lines)) if "s_endpgm" in lines[i])
ends = [i for i in range(k0, k1) if re.search(r"s_cbranch_scc0\s+\.LBB", lines[i])]
end = ends[-1]
lbl = lines[end].split()[1]
start = next(i for i in range(k0, end) if lines[i].startswith(lbl + ":"))
ops = [l.split()[0] for l in lines[start:end + 1] if l.strip() and not l.strip().startswith(";") and not l.startswith(".")]
c = collections.Counter(ops)
valu = sum(n for o, n in c.items() if o.startswith("v_"))
print(f"{lines[k0][:70]}  loop {lbl}: VALU {valu}  LDS {sum(n for o, n in c.items() if o.startswith('ds_'))}  "
      f"SMEM {sum(n for o, n in c.items() if o.startswith('s_load'))}  s_nop {c.get('s_nop', 0)}")
print("  " + "; ".join(f"{n} {o}" for o, n in c.most_common(18)))
meta = [l.strip() for l in lines[k1:k1 + 60] if "NumVgprs" in l or "; Occupancy" in l or "ScratchSize" in l]
print("  " + " ".join(meta[:4]))
