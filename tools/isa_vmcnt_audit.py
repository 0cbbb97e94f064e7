"""Static audit of the kernels' gfx950 ISA for the stall pattern the flat pooling backward (csrc/k_pool3.h) taught in round 4: a wave's vector-memory
counter (vmcnt) is IN ORDER, so any wait for a load also waits for every global store issued before it.  Two shapes make that expensive:
  * a spill reload (scratch_load) in a region that also issues global stores -- the reload's wait drains the store queue each time;
  * s_waitcnt vmcnt(0) (e.g. the fence of a __syncthreads()) inside a loop that issues global stores -- every trip waits for its own stores.
Usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I news_recommendation_amd/csrc [-mllvm -amdgpu-mfma-vgpr-form=1] -S --cuda-device-only -o X.s FILE.hip
       python tools/isa_vmcnt_audit.py X.s [more.s]
Per kernel: registers, spills, and per LOOP (assembler loop headers, innermost first): global / buffer stores, loads, scratch reloads, vmcnt(0) waits,
MFMAs.  Loops without stores are not listed.  A heuristic reading aid, not a proof: counted waits (vmcnt(n > 0)) are what a fix looks like."""
import re, sys

ST = re.compile(r'^\s*(global_store|buffer_store|flat_store)')
LD = re.compile(r'^\s*(global_load|buffer_load|flat_load)(?!.*lds)')
DMA = re.compile(r'^\s*(global_load_lds|buffer_load.*\blds\b)')
SCL = re.compile(r'^\s*scratch_load')
SCS = re.compile(r'^\s*scratch_store')
W0 = re.compile(r'^\s*s_waitcnt\s+.*vmcnt\(0\)')
WN = re.compile(r'^\s*s_waitcnt\s+.*vmcnt\(([1-9]\d*)\)')
MF = re.compile(r'^\s*v_mfma')
HDR = re.compile(r'^(\.LBB\d+_\d+):.*Loop Header: Depth=(\d+)')
LBL = re.compile(r'^(\.LBB\d+_\d+):')
BR = re.compile(r'^\s*s_c?branch\S*\s+(\.LBB\d+_\d+)')


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r'^(_Z\w+):\s*; @', line)
        if m:
            if name: yield name, body
            name, body = m.group(1), []
        elif name is not None:
            body.append(line.rstrip('\n'))
            if line.strip().startswith('s_endpgm'):
                yield name, body
                name, body = None, []


def demangle(n):
    m = re.match(r'_ZN2nr(\d+)', n)
    if m:
        k = int(m.group(1)); st = m.end()
        return n[st:st + k] + ('<…>' if 'I' in n[st + k:st + k + 2] else '')
    return n


def audit(name, body):
    labels = {}
    for i, l in enumerate(body):
        m = LBL.match(l)
        if m: labels[m.group(1)] = i
    loops = []                                   # (start, end, depth): from the header label to the end of the last block the assembler annotates as
    hdr_at = -1                                  # belonging to that loop ("in Loop: Header=BBx_y" / "Parent Loop BBx_y")
    for i, l in enumerate(body):
        m = HDR.match(l)
        if not m and hdr_at == i - 1 and 'Loop Header: Depth=' in l:      # (inner headers carry the annotation on the lines after the label)
            m2 = re.search(r'Loop Header: Depth=(\d+)', l)
            lab = LBL.match(body[hdr_at]).group(1)
            loops.append([hdr_at, hdr_at, int(m2.group(1)), lab[2:]])
            continue
        if LBL.match(l):
            hdr_at = i if not m else -1
        if m:
            loops.append([i, i, int(m.group(2)), m.group(1)[2:]])
    for lp in loops:
        key = lp[3]
        last = lp[0]
        for j in range(lp[0], len(body)):
            if ('Header=' + key + ' ') in body[j] + ' ' or ('Parent Loop ' + key + ' ') in body[j] + ' ':
                last = j
        end = last
        for j in range(last + 1, len(body)):     # to the end of that block
            if LBL.match(body[j]): break
            end = j
        lp[1] = end
    loops = [(a, b, d) for a, b, d, _ in loops]
    rows = []
    for (a, b, d) in loops:
        seg = body[a:b + 1]
        c = lambda rx: sum(1 for l in seg if rx.match(l))
        st, ld, dma, scl, w0, wn, mf = c(ST), c(LD), c(DMA), c(SCL), c(W0), c(WN), c(MF)
        if st and (scl or w0):
            rows.append((d, b - a + 1, st, ld, dma, scl, w0, wn, mf))
    tot = lambda rx: sum(1 for l in body if rx.match(l))
    return tot(ST), tot(SCL), tot(SCS), tot(W0), rows


if __name__ == '__main__':
    for path in sys.argv[1:]:
        for name, body in kernels(path):
            st, scl, scs, w0, rows = audit(name, body)
            if not rows and not scl:
                continue
            print(f"{demangle(name):34s} stores {st:4d}  spill stores {scs:3d} reloads {scl:3d}  vmcnt(0) {w0:3d}")
            for (d, n, s_, ld, dma, sl, w, wn, mf) in rows:
                print(f"    loop depth {d} ({n:5d} lines): stores {s_:3d}  loads {ld:3d}  lds-dma {dma:3d}  scratch reloads {sl:3d}  vmcnt(0) {w:3d}  counted waits {wn:3d}  mfma {mf:4d}")
