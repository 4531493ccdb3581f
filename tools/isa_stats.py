"""Static instruction mix per kernel from a hipcc -S listing:  python tools/isa_stats.py /tmp/x.s"""
import re, collections, sys
cur = None; stats = {}
for ln in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', ln)
    if m: cur = m.group(1); stats[cur] = collections.Counter(); continue
    if cur and re.match(r'^\s+(v_|s_|ds_|global_|buffer_|flat_|scratch_)', ln): stats[cur][ln.split()[0]] += 1
    if ln.startswith('.Lfunc_end'): cur = None
for k, c in stats.items():
    if not c: continue
    g = lambda pre: sum(n for o, n in c.items() if o.startswith(pre))
    ub = sum(n for o, n in c.items() if 'ubyte' in o or 'sbyte' in o)
    f64 = sum(n for o, n in c.items() if 'f64' in o)
    print("%-52s valu %5d (f64 %4d) salu %5d lds %4d global %4d (byte %3d) scratch %3d branches %4d" % (
        re.sub(r'^_ZN5orbfe\d+', '', k)[:52], g('v_'), f64, g('s_'), g('ds_'), g('global_') + g('flat_'), ub, g('scratch_'), sum(n for o, n in c.items() if 'cbranch' in o)))
