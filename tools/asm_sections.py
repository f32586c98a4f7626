"""Static instruction counts of k_queue between the TRC_MARK section markers (development aid).
    hipcc ... -DTRC_MARKERS -S --cuda-device-only -o /tmp/kq.s kernels.hip ; python3 tools/asm_sections.py /tmp/kq.s [kernel]"""
import re, sys
path = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_queue"
lines = open(path).read().split('\n')
start = [i for i, l in enumerate(lines) if kern in l and l.split(':')[0].startswith('_ZN') and ':' in l and not l.startswith('\t')][0]
end = [i for i, l in enumerate(lines[start:]) if l.strip().startswith('.Lfunc_end')][0] + start
sec = 'prologue'; cnt = {}; order = []
for l in lines[start:end]:
    s = l.strip()
    m = re.match(r'; MARK (\S+)', s)
    if m: sec = m.group(1)
    if sec not in cnt: cnt[sec] = {'v': 0, 's': 0, 'ds': 0, 'mem': 0}; order.append(sec)
    if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'): continue
    op = s.split()[0]
    if op.startswith('v_'): cnt[sec]['v'] += 1
    elif op.startswith('s_'): cnt[sec]['s'] += 1
    elif op.startswith('ds_'): cnt[sec]['ds'] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cnt[sec]['mem'] += 1
for k in order: print(f"{k:16s}", cnt[k])
print("total VALU", sum(c['v'] for c in cnt.values()))
