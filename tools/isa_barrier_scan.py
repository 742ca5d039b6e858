"""Static check of the compiled kernels (no GPU): every s_barrier must be reached with no LDS write / atomic of the same
wave still pending, i.e. on EVERY control-flow path into the barrier an `s_waitcnt ... lgkmcnt(0)` must lie between the last
ds_write / ds_add / ... and the barrier.  hipcc normally guarantees that (the workgroup-scope release fence of
__syncthreads()); in round 3 it did not for a barrier inside a rolled loop whose pending ds_add_u32 came around the loop's
back-edge (the first k_ss_hist2: the LDS histogram was read while other SIMDs' adds were still queued -> sub-bucket counts
off by a few records, one build in eight).  Usage:  isa_barrier_scan.py file.s   (hipcc -S --offload-device-only)"""
import re, sys
from collections import defaultdict

# (ds_bpermute / ds_permute / ds_swizzle move data between lanes and do not touch LDS memory: not counted)
LDS_WRITE = re.compile(r'^\s*ds_(write|add|sub|inc|dec|min|max|and|or|xor|cmpst|wrxchg|append|consume)')
LDS_ANY = re.compile(r'^\s*ds_')
WAIT0 = re.compile(r's_waitcnt\b.*lgkmcnt\(0\)')
BR = re.compile(r'^\s*(s_branch|s_cbranch_\w+)\s+(\.LBB\w+)')


def functions(path):
    cur, name = None, None
    for l in open(path):
        l = l.rstrip('\n')
        if l.startswith('_Z') and '; @' in l:
            name, cur = l.split(':')[0], []
        elif cur is not None:
            cur.append(l)
            if 's_endpgm' in l:
                yield name, cur
                cur = None


def scan(name, body):
    # basic blocks
    blocks, labels = [[]], {}
    for l in body:
        t = l.strip()
        if not t or t.startswith(';') or (t.startswith('.') and not t.startswith('.LBB')):
            continue
        m = re.match(r'^(\.LBB\w+):', t)
        if m:
            if blocks[-1]:
                blocks.append([])
            labels[m.group(1)] = len(blocks) - 1
            continue
        blocks[-1].append(t)
        if BR.match(t) or 's_endpgm' in t or 's_setpc' in t:
            blocks.append([])
    preds = defaultdict(set)
    for i, b in enumerate(blocks):
        last = b[-1] if b else ''
        m = BR.match(last)
        if m and m.group(2) in labels:
            preds[labels[m.group(2)]].add(i)
        if not (last.startswith('s_branch') or 's_endpgm' in last or 's_setpc' in last) and i + 1 < len(blocks):
            preds[i + 1].add(i)
    bad = []
    for bi, b in enumerate(blocks):
        for k, t in enumerate(b):
            if not t.startswith('s_barrier'):
                continue
            # backward search
            seen, stack, flagged = set(), [(bi, k - 1)], False
            while stack and not flagged:
                blk, idx = stack.pop()
                j = idx
                stop = False
                while j >= 0:
                    u = blocks[blk][j]
                    if WAIT0.search(u) or u.startswith('s_barrier') and False:
                        stop = True; break
                    if LDS_WRITE.match(u):
                        flagged = True; break
                    j -= 1
                if flagged or stop:
                    continue
                for p in preds[blk]:
                    if p not in seen:
                        seen.add(p); stack.append((p, len(blocks[p]) - 1))
            if flagged:
                bad.append((bi, k))
    return bad


if __name__ == '__main__':
    total = 0
    for name, body in functions(sys.argv[1]):
        bad = scan(name, body)
        if bad:
            total += 1
            print(f"{len(bad)} barrier(s) reachable with a pending LDS write: {name[:110]}")
    print(f"kernels flagged: {total}")
    sys.exit(1 if total else 0)                 # (the library's Makefile runs this on every build: a flagged kernel fails it)
