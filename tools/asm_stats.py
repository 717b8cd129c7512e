"""CPU-side: per-kernel register / LDS / scratch / instruction-class counts from a gfx950 .s file
(hipcc --save-temps).  Usage: python tools/asm_stats.py file.s substring [substring ...]"""
import re
import sys
import collections


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    txt = open(path).read().split('\n')
    # function bodies: from "<name>:" label to s_endpgm ... .amdhsa_kernel block
    i = 0
    while i < len(txt):
        m = re.match(r'^(_Z\w+):\s*(;.*)?$', txt[i])
        if m and any(p in m.group(1) for p in pats):
            name = m.group(1)
            cnt = collections.Counter()
            j = i + 1
            while j < len(txt) and not txt[j].startswith('.Lfunc_end'):
                t = txt[j].strip()
                if t and not t.startswith((';', '.', '_')) and not t.endswith(':'):
                    op = t.split()[0]
                    if op.startswith('v_mfma'): cnt['mfma'] += 1
                    elif op.startswith('v_pk_'): cnt['v_pk'] += 1
                    elif op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane')): cnt['v_lane'] += 1
                    elif op.startswith('v_'): cnt['valu'] += 1
                    elif op.startswith('s_waitcnt'): cnt['waitcnt'] += 1
                    elif op.startswith('s_cbranch') or op.startswith('s_branch'): cnt['branch'] += 1
                    elif op.startswith('s_'): cnt['salu'] += 1
                    elif op.startswith('ds_'): cnt['lds'] += 1
                    elif op.startswith(('global_load', 'buffer_load', 'flat_load')): cnt['vmem_ld'] += 1
                    elif op.startswith(('global_store', 'buffer_store', 'flat_store')): cnt['vmem_st'] += 1
                    elif op.startswith(('scratch_',)): cnt['scratch'] += 1
                    else: cnt['other'] += 1
                j += 1
            meta = {}
            for k in range(j, min(j + 80, len(txt))):
                mm = re.search(r'\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)', txt[k])
                if mm: meta[mm.group(1)] = int(mm.group(2))
                mm = re.search(r'\.amdhsa_group_segment_fixed_size (\d+)', txt[k])
                if mm: meta['lds'] = int(mm.group(1))
            for k in range(max(0, i - 0), j):
                pass
            print(name[:70])
            print('   ', meta, dict(cnt))
            i = j
        i += 1


main()
