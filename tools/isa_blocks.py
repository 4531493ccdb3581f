"""Basic blocks of one kernel in a hipcc -S listing with their instruction mix: python tools/isa_blocks.py <listing.s> <kernel substring>
(which loop of a kernel the VALU instructions are in: multiply by the trip counts you know)"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
cur = None; blocks = []; name = None
for ln in open(path):
    m = re.match(r'^(_Z\w+):', ln)
    if m:
        cur = m.group(1) if key in m.group(1) else None
        if cur: blocks.append(["entry", collections.Counter(), []])
        continue
    if not cur: continue
    if ln.startswith('.Lfunc_end'): cur = None; continue
    m = re.match(r'^(\.LBB\w+):', ln)
    if m: blocks.append([m.group(1), collections.Counter(), []]); continue
    m = re.match(r'^\s+((v_|s_|ds_|global_|buffer_|flat_|scratch_)\w+)', ln)
    if m:
        op = m.group(1)
        c = blocks[-1][1]
        c['valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem'] += 1
        if 'cbranch' in op or op == 's_branch': blocks[-1][2].append(ln.split()[-1])
for b, c, t in blocks:
    print("%-12s valu %4d salu %4d lds %3d vmem %3d  -> %s" % (b, c['valu'], c['salu'], c['lds'], c['vmem'], " ".join(t)))
