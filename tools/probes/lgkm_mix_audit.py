"""Audit of the compiled kernels for the pattern behind profiles/r06_reproducibility.txt (6): an LDS READ and a scalar (SMEM) load in flight at the
same time, both retired by one `s_waitcnt lgkmcnt`.  Per kernel of the ISA (hipcc -S): a linear scan, then every backward branch once more from
its target label with the state the branch is taken in (what a loop carries into its next pass).  A site is flagged when a ds_read is issued
while an s_load has not been waited for with lgkmcnt(0), or the other way round.
    python tools/probes/lgkm_mix_audit.py /tmp/isa/all/*.s"""
import re, sys


def scan(lines, lo, hi, lds, smem, sites, states=None):
    for i in range(lo, hi):
        ln, s = lines[i]
        op = s.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m and int(m.group(1)) == 0:
                lds = smem = 0
        elif op.startswith("ds_read") or op.startswith("ds_load"):
            if smem:
                sites.add((ln, s))
            lds += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            if lds:
                sites.add((ln, s))
            smem += 1
        if states is not None:
            states[i] = (lds, smem)
    return lds, smem


for path in sys.argv[1:]:
    kernels, cur = {}, None
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None or not s or s[0] == ";":
            continue
        if s.startswith(".LBB") and s.split()[0].endswith(":"):
            cur.append((ln, "label " + s.split(":")[0]))
            continue
        if s[0] == ".":
            continue
        cur.append((ln, s))
        if s.split()[0] == "s_endpgm":
            cur = None
    name, any_ = path.split("/")[-1], False
    for k, lines in kernels.items():
        sites, states = set(), {}
        labels = {s.split()[1]: i for i, (ln, s) in enumerate(lines) if s.startswith("label ")}
        body = [(ln, s if not s.startswith("label ") else "s_nop 0") for ln, s in lines]
        scan(body, 0, len(body), 0, 0, sites, states)
        for i, (ln, s) in enumerate(lines):
            if s.split()[0] in ("s_branch", "s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz", "s_cbranch_execnz"):
                j = labels.get(s.split()[-1])
                if j is not None and j < i:
                    scan(body, j, i, *states[i], sites)
        if sites:
            any_ = True
            first = min(sites)
            print(f"{name}: {k[:88]}: {len(sites)} sites, first at line {first[0]}: {first[1]}")
    if not any_:
        print(f"{name}: none")
