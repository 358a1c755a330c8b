"""ADVICE r2 (medium), roi_backward.hip: the grad_out loads of the backward walk are issued from inline asm and waited
for by hand, so the compiler does not know they are in flight.  This reads the kernel ISA (hipcc --save-temps) and
reports every instruction that, in program order, touches a VGPR between its asm `global_load_dword` and the asm
`s_waitcnt vmcnt(N)` that covers it (loads return in order).  Expected: 0 violations, and one `vmcnt(0)` drain per walk.
    cd /tmp/x && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c --save-temps .../roi_backward.hip
    python tools/check_asm_loads.py /tmp/x/roi_backward-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
start = {}
for i, l in enumerate(lines):
    m = re.match(r'^(_Z\w+):', l)
    if m:
        start[m.group(1)] = i


def regs_in(s):
    out = set(int(m.group(1)) for m in re.finditer(r'\bv(\d+)\b', s))
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', s):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


names = sorted(start, key=start.get)
total_bad = 0
for ki, k in enumerate(names):
    a, b = start[k], (start[names[ki + 1]] if ki + 1 < len(names) else len(lines))
    inasm, pending, bad, nload, nwait, drains = False, {}, 0, 0, 0, 0
    for i, l in enumerate(lines[a:b]):
        s = l.strip()
        if s.startswith(';;#ASMSTART'):
            inasm = True
            continue
        if s.startswith(';;#ASMEND'):
            inasm = False
            continue
        if not s or s[0] in ';.' or s.endswith(':'):
            continue
        code = s.split(';')[0]
        if inasm and code.startswith('global_load_dword'):
            pending[int(re.match(r'global_load_dword v(\d+)', code).group(1))] = i
            nload += 1
            continue
        if inasm and code.startswith('s_waitcnt vmcnt'):
            n = int(re.search(r'vmcnt\((\d+)\)', code).group(1))
            nwait += 1
            drains += n == 0
            order = sorted(pending.items(), key=lambda kv: kv[1])
            for r, _ in order[:max(0, len(order) - n)]:
                pending.pop(r)
            continue
        used = regs_in(code) & set(pending)
        if used:
            bad += 1
            print(k[:60], 'line', a + i, 'touches in-flight', sorted(used), ':', code)
    if nload:
        print('%s: %d asm loads, %d asm waits (%d drains), %d violations' % (k[:64], nload, nwait, drains, bad))
        total_bad += bad
sys.exit(1 if total_bad else 0)
