#!/usr/bin/env python3
"""gemm_asm256.h keeps its accumulators in a[0:255] across the C++ between the K loop's asm text and the v_accvgpr_read
statements behind it, without the compiler knowing that they are live.  This script compiles the translation units that
instantiate gemm_tn256a_kernel to device assembly and fails if, in any instantiation,
  * the compiler writes an AGPR (v_accvgpr_write / v_accvgpr_mov / an MFMA destination) BEFORE the asm statement that reads
    that accumulator out (it may use an AGPR as spill space once its accumulator has been consumed),
  * there is a backward branch between the loop and the last read (program order would not be execution order), or
  * the kernel uses scratch.
usage: python scripts/check_asm256.py [--hipcc PATH] [--flags "the build's CXXFLAGS"] [vq_core.hip vq_encoder.hip ...]
`make DIAG=1` (the only builds that contain the kernel since round 4) runs it with its own $(HIPCC) and $(CXXFLAGS), so what is
checked is the compile that ships; run by hand it defaults to the Makefile's defaults plus -DVQ_DIAG."""
import os, re, subprocess, sys, tempfile

here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, '..', 'video-quierer_amd', 'csrc')
argv = sys.argv[1:]
hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
flags = None
while argv and argv[0] in ('--hipcc', '--flags'):
    if argv[0] == '--hipcc':
        hipcc = argv[1]
    else:
        flags = argv[1].split()
    argv = argv[2:]
tus = [a for a in argv if not a.startswith('-')] or ['vq_core.hip', 'vq_encoder.hip', 'vq_diag.hip']
if flags is None:
    objdir = os.path.join(csrc, '..', 'lib', 'obj_diag')
    flags = f'-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-cuda-compat -ffp-contract=on -DVQ_DIAG -I{objdir}'.split()
flags = [f for f in flags if f != '-c'] + [a for a in argv if a.startswith('-')] + ['--cuda-device-only', '-S']


def check(name, text):
    lines = text.split('\n')
    in_asm, loop_end, reads, writes, labels, problems = False, None, {}, [], {}, []
    for i, l in enumerate(lines):
        if ';;#ASMSTART' in l:
            in_asm = True
            continue
        if ';;#ASMEND' in l:
            in_asm = False
            continue
        if in_asm:
            if 's_nop 15' in l:
                loop_end = i                      # the K loop's text ends in two of these
            m = re.search(r'v_accvgpr_read_b32 v\d+, a(\d+)', l)
            if m and loop_end is not None:
                reads[int(m.group(1))] = i
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
        if loop_end is None:
            continue
        m = re.search(r'\b(v_accvgpr_write_b32|v_accvgpr_mov_b32) a(\d+)', l)
        if m:
            writes.append((i, int(m.group(2)), l.strip()))
        m = re.search(r'\bv_mfma\S* a\[(\d+):(\d+)\]', l)
        if m:
            writes += [(i, r, l.strip()) for r in range(int(m.group(1)), int(m.group(2)) + 1)]
        m = re.search(r'\bs_c?branch\S*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] > loop_end and (not reads or i < max(reads.values())):
            problems.append(f'backward branch at line {i} between the loop and the last accumulator read')
    if loop_end is None or len(reads) != 256:
        problems.append(f'found {len(reads)} accumulator reads behind the loop (expected 256)')
    for i, r, l in writes:
        if r in reads and i < reads[r]:
            problems.append(f'line {i}: `{l}` overwrites a{r} before it is read out (line {reads[r]})')
    sc = re.search(r'; ScratchSize: (\d+)', text)
    if sc and int(sc.group(1)):
        problems.append(f'scratch: {sc.group(1)} bytes')
    return problems, len(writes)


bad, seen = 0, 0
for tu in tus:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'tu.s')
        subprocess.run([hipcc] + flags + [os.path.join(csrc, tu), '-o', out], check=True)
        txt = open(out).read()
    for f in re.split(r'\n\t\.globl\t', txt):
        name = f.split('\n', 1)[0]
        if 'gemm_tn256a' not in name:
            continue
        seen += 1
        problems, nw = check(name, f)
        if problems:
            bad += 1
            print(f'{tu}: {name}')
            for p in problems[:8]:
                print('    ' + p)
        elif nw:
            print(f'{tu}: {name}: {nw} AGPR writes by the compiler, all behind the reads of the registers they reuse')
print(f'{seen} instantiations of gemm_tn256a_kernel checked, {bad} bad')
sys.exit(1 if bad or not seen else 0)
