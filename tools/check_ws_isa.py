"""Static check of the third-edition GEMM's producer loop in the compiled ISA (hipcc -S): inside the steady-state loop there must be no
copy or spill of a register that a hand-placed load is still writing, and the prologue must load into the registers the loop reloads
(errors); scratch RELOADS of other values there are reported (each drains the pipeline once: a performance matter, not a correctness one).
usage: check_ws_isa.py file.s"""
import re, sys
s = open(sys.argv[1]).read()
bad = 0
for m in re.finditer(r'^(_ZN12_GLOBAL__N_114gemm_ws_kernel\w+):', s, re.M):
    name = m.group(1)
    i = m.end(); j = s.index('.Lfunc_end', i)
    body = s[i:j].splitlines()
    loads = [n for n, l in enumerate(body) if 'global_load_dwordx4' in l and 'ASMSTART' in body[n - 1]]
    waits = [n for n, l in enumerate(body) if re.search(r's_waitcnt vmcnt\((20|22|23)\)', l) and 'ASMSTART' in body[n - 1]]
    if not waits:
        print(name, 'no hand-placed waits found'); bad += 1; continue
    hdr = max(n for n, l in enumerate(body) if 'Loop Header' in l and n < waits[0])
    end = waits[-1] + 80
    regs = set()
    for n in loads:
        mm = re.search(r'v\[(\d+):(\d+)\]', body[n]); regs.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
    scratch = [n for n in range(hdr, end) if 'scratch_' in body[n]]
    spills = []
    for n in scratch:
        mm = re.search(r'scratch_store_dword\S* off, v(\d+)', body[n])
        if mm and int(mm.group(1)) in regs: spills.append(n)
    copies = []
    for n in range(hdr, end):
        l = body[n].strip()
        mm = re.match(r'v_mov_b64\S* v\[\d+:\d+\], v\[(\d+):\d+\]', l) or re.match(r'v_mov_b32_e32 v\d+, v(\d+)', l)
        if mm and int(mm.group(1)) in regs: copies.append((n, l))
    # the prologue = the two register sets (24 loads) issued right in front of the loop; kernels with a fused epilogue also hold hand-placed
    # loads of the CONSUMER waves (elsewhere in the text: not part of this check)
    pro = [re.search(r'v\[\d+:\d+\]', body[n]).group(0) for n in [n for n in loads if n < waits[0]][-24:]]
    loop = [re.search(r'v\[\d+:\d+\]', body[n]).group(0) for n in loads if hdr < n < end]
    same = pro[:len(loop)] == loop[:len(pro)] if len(pro) == len(loop) else sorted(set(pro)) == sorted(set(loop))
    ok = not spills and not copies and same
    bad += not ok
    print(name[-22:], 'loop lines', hdr, end, '| asm loads in loop', len(loop), '| scratch reloads in loop', len(scratch) - len(spills), '| spills of load registers', len(spills), '| copies of load registers', len(copies),
          '| prologue registers == loop registers', same, '->', 'ok' if ok else 'CHECK')
    for c in copies[:4]: print('   ', c)
    for n in scratch[:6]: print('   ', n, body[n].strip())

# ---- consumer waves of the act-4 (dact) epilogue: 8 tiles x 4 hand-placed Y loads ("=v" outputs the compiler does not count) and one
# hand-written `s_waitcnt vmcnt(N)` per tile, N = 12 12 12 12 12 8 4 0 (PF = 3 tiles ahead).  The count is right only if NOTHING the
# compiler emits sits between the first Y load and the last wait that is itself a vector-memory operation (a scratch spill store or
# reload, a global / buffer / flat access: a YOUNGER operation shifts the count and a tile's own loads may still be in flight when its
# registers are read), and if no instruction reads (copies, spills) a Y register between its load and its wait.
Y_WAITS = [12, 12, 12, 12, 12, 8, 4, 0]
for m in re.finditer(r'^(_ZN12_GLOBAL__N_114gemm_ws_kernel\w+Li2ELi4EE\w+):', s, re.M):
    name = m.group(1)
    i = m.end(); j = s.index('.Lfunc_end', i)
    body = s[i:j].splitlines()
    asm = lambda n: 'ASMSTART' in body[n - 1]
    waits = [(n, int(re.search(r'vmcnt\((\d+)\)', body[n]).group(1))) for n in range(1, len(body)) if re.search(r's_waitcnt vmcnt\(\d+\)\s*$', body[n]) and asm(n)]
    start = next((k for k in range(len(waits) - 7) if [v for _, v in waits[k:k + 8]] == Y_WAITS), None)
    if start is None:
        print(name[-22:], 'dact epilogue: the 8 hand-written waits', Y_WAITS, 'were not found -> CHECK'); bad += 1; continue
    wl = [n for n, _ in waits[start:start + 8]]
    loads = [n for n in range(1, wl[-1]) if 'global_load_dwordx4' in body[n] and asm(n)]
    loads = [n for n in loads if n < wl[-1]][-32:]                      # the 32 Y loads are the last hand-placed loads in front of the last wait
    ok = len(loads) == 32 and loads[15] < wl[0] < loads[16]            # 16 loads (prologue 12 + tile 3) ahead of the first wait
    foreign, reads = [], []
    if ok:
        for n in range(loads[0], wl[-1]):
            l = body[n].strip()
            if re.match(r'(scratch_|buffer_|global_|flat_)', l) and not asm(n):
                foreign.append((n, l))
        for k, n in enumerate(loads):
            mm = re.search(r'v\[(\d+):(\d+)\]', body[n]); lo, hi = int(mm.group(1)), int(mm.group(2))
            for q in range(n + 1, wl[k // 4]):
                l = body[q].split(';')[0]
                for r in re.finditer(r'\bv(\d+)\b|v\[(\d+):(\d+)\]', l):
                    a, b = (int(r.group(1)),) * 2 if r.group(1) else (int(r.group(2)), int(r.group(3)))
                    if a <= hi and b >= lo:
                        reads.append((q, body[q].strip(), f'load at {n}: v[{lo}:{hi}]'))
    ok = ok and not foreign and not reads
    bad += not ok
    print(name[-22:], 'dact epilogue lines', loads[0] if loads else None, wl[-1], '| Y loads', len(loads), '| compiler-emitted vector-memory ops among them', len(foreign),
          '| uses of a Y register before its wait', len(reads), '->', 'ok' if ok else 'CHECK')
    for c in (foreign + reads)[:6]: print('   ', c)
sys.exit(1 if bad else 0)
