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
sys.exit(1 if bad else 0)
