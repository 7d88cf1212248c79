"""Basic-block instruction mix of one kernel in a `hipcc --cuda-device-only -S` listing (ISA review without a GPU).
usage: python tools/isa_blocks.py <file.s> <substring of the mangled kernel name> [min block size]"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    want = sys.argv[2]
    minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 15
    for m in re.finditer(r'\n(_Z[A-Za-z0-9_]+):', s):
        nm = m.group(1)
        if want not in nm:
            continue
        i = m.start()
        j = s.index('.Lfunc_end', i)
        blocks, cur, label = [], [], 'entry'
        for l in s[i:j].split('\n'):
            if re.match(r'^\.LBB\d+_\d+:', l):
                blocks.append((label, cur))
                label, cur = l.split(':')[0], []
            elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
                cur.append(l.strip())
        blocks.append((label, cur))
        print(nm[:100], 'total instr', sum(len(c) for _, c in blocks))
        for lab, c in blocks:
            if len(c) < minsz:
                continue
            cnt = lambda *p: sum(1 for x in c if x.startswith(p))
            print('  %-10s n=%4d valu=%4d (pk %3d, exp %2d, mfma %3d) ds=%3d salu=%3d vmem=%3d waitcnt=%2d' % (
                lab, len(c), cnt('v_'), cnt('v_pk_'), cnt('v_exp_f32'), cnt('v_mfma'), cnt('ds_'), cnt('s_') - cnt('s_waitcnt'),
                cnt('global_', 'buffer_', 'flat_', 'scratch_'), cnt('s_waitcnt')))


if __name__ == "__main__":
    main()
