"""Audit of the compiled kernels for the pattern behind profiles/r06_reproducibility.txt (6): an LDS READ and a scalar (SMEM) load in flight at the
same time, both retired by one `s_waitcnt lgkmcnt`.  Linear scan of the ISA (hipcc -S) per kernel: a site is flagged when a ds_read is issued
while an s_load has not been waited for with lgkmcnt(0), or the other way round.  (Control flow is ignored: a back edge may add sites.)
    python tools/probes/lgkm_mix_audit.py /tmp/isa/all/*.s"""
import re, sys
for path in sys.argv[1:]:
    kern, lds, smem, sites = None, 0, 0, {}
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            kern, lds, smem = m.group(1), 0, 0
            continue
        if kern is None or not s or s[0] in ".;":
            continue
        op = s.split()[0]
        if op == "s_endpgm":
            kern = None
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m and int(m.group(1)) == 0:
                lds = smem = 0
            continue
        if op.startswith("ds_read") or op.startswith("ds_load"):
            if smem:
                sites.setdefault(kern, []).append((ln, s))
            lds += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            if lds:
                sites.setdefault(kern, []).append((ln, s))
            smem += 1
    name = path.split("/")[-1]
    if not sites:
        print(f"{name}: none")
    for k, v in sites.items():
        print(f"{name}: {k[:90]}: {len(v)} sites, e.g. line {v[0][0]}: {v[0][1]}")
