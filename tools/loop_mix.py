#!/usr/bin/env python3
"""Static instruction mix of the loops of the shipped kernels whose name matches a regular expression:
    python3 tools/loop_mix.py 'attn_mfma_lds_kernelIDF16_'
For every backward branch it prints the body's counts by class (MFMA, v_exp, packed f32, other VALU, DS, VMEM, scalar) and
the most frequent VALU opcodes.  STATIC counts: code behind a not-taken branch inside the loop (e.g. the tail mask of the
attention kernels) is included, so read the listing (tools/isa_lint.py's disassembly) before concluding anything about
the taken path.  Uses tools/isa_lint.py to unbundle and disassemble audio_diffusion_amd/libadk.so."""
import sys, re, collections, tempfile
sys.path.insert(0, '/root/repo/tools')
import isa_lint as L
tmp = tempfile.mkdtemp()
pat = re.compile(sys.argv[1])
for elf in L.code_objects_of('/root/repo/audio_diffusion_amd/libadk.so', tmp):
    ks = L.parse_objdump(L.disassemble(elf))
    for name, ins in ks.items():
        if not pat.search(name): continue
        addr = {i.addr: n for n, i in enumerate(ins)}
        # backward branches -> loops; take each, report the instruction mix of the body
        loops = []
        for n, i in enumerate(ins):
            if i.op.startswith('s_cbranch') or i.op == 's_branch':
                if i.target is not None and i.target in addr and addr[i.target] <= n:
                    loops.append((addr[i.target], n))
        print(name[:150], len(ins), 'instructions; loops:', [(a, b, b - a + 1) for a, b in loops])
        for a, b in loops:
            if b - a < 100: continue
            c = collections.Counter()
            for i in ins[a:b + 1]:
                op = i.op
                k = ('mfma' if 'mfma' in op else 'exp' if op.startswith('v_exp') else 'pk_f32' if op.startswith('v_pk_') and 'f32' in op
                     else 'valu' if op.startswith('v_') else 'ds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_')) else 'salu/other')
                c[k] += 1
            print('   loop', a, b, dict(c))
            cv = collections.Counter(i.op for i in ins[a:b+1] if i.op.startswith('v_') and 'mfma' not in i.op)
            print('   VALU ops:', cv.most_common(25))
