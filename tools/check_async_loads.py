"""Static check of token_gemm.hip's untracked (inline-asm) global loads: between the loads of a tile and the counted
s_waitcnt that precedes their first use, no instruction may read or write the destination registers (a register copy or
spill placed there by the compiler would move data that has not landed). Usage: python tools/check_async_loads.py file.s"""
import collections, re, sys
src = open(sys.argv[1]).read()
bad = 0
for m in re.finditer(r'^(_ZN\S*rowgemm_kernel\S*):[^\n]*\n(.*?)s_endpgm', src, re.S | re.M):
    name, body = m.group(1), m.group(2).split('\n')
    heads = [i for i, l in enumerate(body) if 'Loop Header' in l]
    if not heads:
        continue
    i, loaded, first = heads[0], set(), None
    while i < len(body):
        l = body[i]
        mm = re.search(r'global_load_dword(?:x4)?\s+v(?:\[(\d+):(\d+)\]|(\d+)),', l)
        if mm:
            first = i if first is None else first
            loaded.update(range(int(mm.group(1)), int(mm.group(2)) + 1) if mm.group(1) else [int(mm.group(3))])
        if first is not None and 'global_load_lds' in l:
            break
        i += 1
    if not loaded:
        continue
    viol, waits, j = [], 0, i
    while j < len(body):
        l = body[j].split(';')[0]
        if 's_waitcnt vmcnt' in l:
            waits += 1
            if waits >= 2:
                break
        regs = set()
        for a, b in re.findall(r'v\[(\d+):(\d+)\]', l):
            regs.update(range(int(a), int(b) + 1))
        regs.update(int(a) for a in re.findall(r'\bv(\d+)\b', l))
        if regs & loaded:
            viol.append(l.strip())
        j += 1
    short = re.sub(r'.*rowgemm_kernelI(.*?)EEv.*', r'\1', name)
    print(f"{short}: {len(loaded)} async destination registers, {len(viol)} touched before their wait")
    for v in viol[:6]:
        print("    ", v)
    bad += len(viol)

# tokgrad_regs_kernel: EVERY stage travels through untracked loads, three stages deep. The memory counter retires in issue
# order: walk the instruction stream with a queue of outstanding vector-memory operations; an s_waitcnt vmcnt(n) retires all
# but the youngest n; no other instruction may touch the destination registers of a load that is still in the queue, and no
# load may be issued whose data is never read (it would land in a register the compiler has handed to something else).
def regs_of(text):
    r = set()
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', text):
        r.update(range(int(a), int(b) + 1))
    r.update(int(a) for a in re.findall(r'\bv(\d+)\b', text))
    return r
for m in re.finditer(r'^(_ZN\S*tokgrad_regs_kernel\S*):[^\n]*\n(.*?)s_endpgm', src, re.S | re.M):
    queue, viol, nloads, dests = [], [], 0, collections.Counter() if False else {}
    for l in m.group(2).split('\n'):
        l = l.split(';')[0].strip()
        if not l or l.endswith(':') or l.startswith('.'):
            continue
        mm = re.match(r'global_load_dword(?:x4)?\s+v(?:\[(\d+):(\d+)\]|(\d+)),(.*)', l)
        if mm:
            dest = set(range(int(mm.group(1)), int(mm.group(2)) + 1)) if mm.group(1) else {int(mm.group(3))}
            pend = set().union(*queue) if queue else set()
            if (regs_of(mm.group(4)) | dest) & pend:
                viol.append(l)
            queue.append(dest)
            nloads += 1
            continue
        if l.startswith(('global_atomic', 'global_store', 'scratch_')):
            queue.append(set())
        w = re.match(r's_waitcnt.*vmcnt\((\d+)\)', l)
        if w:
            while len(queue) > int(w.group(1)):
                queue.pop(0)
            continue
        pend = set().union(*queue) if queue else set()
        if regs_of(l) & pend:
            viol.append(l)
    print(f"tokgrad_regs_kernel: {nloads} async destination registers' loads, {len(viol)} touched before their wait")
    for v in viol[:6]:
        print("    ", v)
    bad += len(viol)
sys.exit(1 if bad else 0)
