"""ISA audit (round 6): 16-byte (and 12-byte) buffer stores whose scalar-offset field is a REGISTER, followed within two instructions by a VALU write to
one of the store's data registers.  LLVM's hazard recognizer (GCNHazardRecognizer::createsVALUHazard) exempts stores with an SGPR soffset from the
"VALU overwrites store data" wait states; on MI355X that exemption does not hold (csrc/k_convgemm.h: the last lanes of each 16 stored the NEW contents
of a data dword).  Usage: python tools/isa_store_hazard_audit.py file.s [...]   (device assembly from hipcc -S --cuda-device-only)"""
import re, sys
pat = re.compile(r'\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)')
wr = re.compile(r'\s*(v_\S+)\s+(v\[(\d+):(\d+)\]|v(\d+))')
bad = 0
for path in sys.argv[1:]:
    lines = open(path).read().split('\n')
    func = '?'
    for i, ln in enumerate(lines):
        if ln.startswith('_Z') and ln.rstrip().endswith(':') or (ln.startswith('_Z') and ':' in ln):
            func = ln.split(':')[0]
        m = pat.match(ln)
        if not m:
            continue
        lo, hi, soff = int(m.group(2)), int(m.group(3)), m.group(5)
        if not soff.startswith('s') or soff in ('s_nop',):
            continue
        n = 0
        j = i + 1
        while j < len(lines) and n < 2:
            t = lines[j].strip()
            j += 1
            if not t or t.startswith((';', '.')) or t.endswith(':'):
                continue
            if t.startswith('s_nop'):
                n += 1 + int(t.split()[1])
                continue
            n += 1
            w = wr.match(lines[j - 1])
            if w and not t.startswith(('v_cmp', 'v_mfma')):
                a, b = (int(w.group(3)), int(w.group(4))) if w.group(3) else (int(w.group(5)), int(w.group(5)))
                if a <= hi and b >= lo:
                    bad += 1
                    print(f'{path}:{i + 1} {func}\n    {ln.strip()}\n    {t}')
print('hazardous sites:', bad)
