"""Sanity check of hand-waited loads in compiler output: between an inline-asm `global_load_dwordx4 v[a:b], ...` and the next
`s_waitcnt vmcnt(..)` no other instruction may mention one of its destination registers (the compiler does not know the load is
still in flight, so a register copy it decided to insert there would read stale data).  Linear scan of one kernel's text.
    python tools/check_asm_loads.py file.s kernel_name_substring"""
import re, sys
txt = open(sys.argv[1]).read().splitlines()
name = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and name in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(txt)) if ".Lfunc_end" in txt[i])
pending = {}     # reg -> line of the load
bad = 0
def regs(tok):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out
nload = 0
for i in range(start, end):
    l = txt[i].split(";")[0].strip()
    if not l or l.startswith(".") or l.endswith(":"):
        continue
    if l.startswith("global_load_dwordx4"):
        ops = l.split(None, 1)[1].split(",")
        dst = regs(ops[0]); addr = regs(ops[1])
        hit = (addr & set(pending))
        if hit:
            print(f"line {i+1}: address uses pending regs {sorted(hit)}: {l}"); bad += 1
        for r in dst:
            pending[r] = i + 1
        nload += 1
        continue
    if l.startswith("s_waitcnt") and "vmcnt" in l:
        n = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
        # in-order return: everything but the newest n loads has landed; loads here are 4-register groups
        keep = sorted(set(pending.values()))[-n:] if n else []
        pending = {r: ln for r, ln in pending.items() if ln in keep}
        continue
    hit = regs(l) & set(pending)
    if hit:
        print(f"line {i+1}: touches in-flight regs {sorted(hit)[:8]}: {l}"); bad += 1
print(f"{nload} asm loads checked, {bad} suspicious instructions")
sys.exit(1 if bad else 0)
