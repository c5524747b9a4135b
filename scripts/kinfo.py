"""Instruction statistics of one kernel in a -save-temps .s file:
    python scripts/kinfo.py <file.s> <kernel name substring> [--blocks]
Register counts, spills, MFMA / LDS / VALU counts -- the audit cdna_hip_programming.md section 5.7 item 4 asks for;
--blocks lists every basic block that holds an MFMA (where the compiler's own copies and waits sit)."""
import re
import sys
from collections import Counter

KEYS = ['v_mfma_f32_32x32x16_bf16', 'v_accvgpr_read_b32', 'v_accvgpr_write_b32', 'v_mov_b32_e32', 'scratch_load_dword',
        'scratch_store_dword', 's_waitcnt', 's_nop', 'ds_read_b128', 'ds_read_b64_tr_b16', 'v_exp_f32_e32', 'v_fma_f32',
        'v_mul_f32_e32', 'v_cvt_pk_bf16_f32', 'global_load_lds_dwordx4', 's_barrier', 'v_add_u32_e32', 'v_xor_b32_e32',
        'v_cndmask_b32_e32', 'v_readfirstlane_b32']


def main():
    s = open(sys.argv[1]).read()
    name = sys.argv[2]
    start = re.search(r'^(_ZN3lwm\d+%s[0-9A-Za-z_]*):' % name, s, re.M)
    end = s.find('.Lfunc_end', start.end())
    body = s[start.end():end]
    lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith((';', '.s', '.p', '.a'))]
    ins = [l for l in lines if not l.endswith(':') and not l.startswith('.L')]
    c = Counter(l.split()[0] for l in ins)
    print(start.group(1), 'instructions', len(ins))
    for k in KEYS:
        print('  ', k, c.get(k, 0))
    mm = re.search(r'; NumVgprs: (\d+)\n; NumAgprs: (\d+)\n; TotalNumVgprs: (\d+)\n; ScratchSize: (\d+)', s[end:end + 4000])
    print('   NumVgprs, NumAgprs, Total, Scratch =', mm.groups() if mm else None)
    # The compiler does not know that an asm statement is an MFMA: any non-MFMA instruction that touches the D registers
    # of an MFMA with fewer than two later MFMAs or 16 wait states in between reads (or overwrites) a tuple the matrix
    # pipe has not written yet.  Report every such instruction with its block.
    def regs(tok_text):
        out = set()
        for kind, lo, hi in re.findall(r'\b([va])\[(\d+):(\d+)\]', tok_text):
            out.update((kind, r) for r in range(int(lo), int(hi) + 1))
        for kind, r in re.findall(r'\b([va])(\d+)\b', tok_text):
            out.add((kind, int(r)))
        return out
    cur, pending, flagged = 'entry', [], []      # pending: [D register set, mfmas since, nop states since]
    for t in (l.strip() for l in body.split('\n')):
        if not t or t.startswith(';'):
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            cur = m.group(1)
            continue
        if t.startswith('.'):
            continue
        t = t.split(';')[0]
        op = t.split()[0]
        if op.startswith('v_mfma'):
            for e in pending:
                e[1] += 1
            d = regs(t.split(None, 1)[1].split(',')[0])
            pending.append([d, 0, 0])
        elif op == 's_nop':
            for e in pending:
                e[2] += int(t.split()[1]) + 1
        else:
            r = regs(t.split(None, 1)[1]) if len(t.split(None, 1)) > 1 else set()
            for e in pending:
                if r & e[0]:
                    flagged.append((cur, t.strip()))
                    break
        pending = [e for e in pending if e[1] < 2 and e[2] < 16]
    print('   instructions touching an MFMA result too early:', len(flagged))
    for b_, t in flagged[:40]:
        print('      ', b_, t)
    # ... and the other direction: a VALU instruction (a compiler copy, a late zero) that writes a register fewer than two
    # wait states before an MFMA reads it as A, B or C
    cur, recent, flagged2 = 'entry', [], []      # recent: (instruction text, destination registers, states since)
    for t in (l.strip() for l in body.split('\n')):
        if not t or t.startswith(';'):
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            cur = m.group(1)
            continue
        if t.startswith('.'):
            continue
        t = t.split(';')[0].strip()
        op = t.split()[0]
        if op.startswith('v_mfma'):
            ops_ = t.split(None, 1)[1].split(',')
            src = regs(','.join(ops_[1:]))
            for txt, dst, age in recent:
                if age < 2 and dst & src:
                    flagged2.append((cur, txt + '   ->   ' + t))
        states = (int(t.split()[1]) + 1) if op == 's_nop' else 1
        recent = [(a_, d_, g_ + states) for a_, d_, g_ in recent if g_ + states < 2]
        if op.startswith('v_') and not op.startswith('v_mfma') and len(t.split(None, 1)) > 1:
            recent.append((t, regs(t.split(None, 1)[1].split(',')[0]), 0))
    print('   VALU writes fewer than 2 wait states before an MFMA reads them:', len(flagged2))
    for b_, t in flagged2[:40]:
        print('      ', b_, t)
    if '--blocks' in sys.argv:
        blocks, cur, curname = [], [], 'entry'
        for l in body.split('\n'):
            t = l.strip()
            m = re.match(r'^(\.LBB\d+_\d+):', t)
            if m:
                blocks.append((curname, cur))
                cur, curname = [], m.group(1)
            elif t and not t.startswith((';', '.')):
                cur.append(t)
        blocks.append((curname, cur))
        for nm, b in blocks:
            cc = Counter(x.split()[0] for x in b)
            if cc.get('v_mfma_f32_32x32x16_bf16', 0):
                print('%-12s n=%4d mfma=%3d accrd=%3d accwr=%3d mov=%3d wait=%3d nop=%3d dsr=%3d tr=%3d dma=%2d valu(fma/exp/mul/cvt)=%d/%d/%d/%d add=%d' % (
                    nm, len(b), cc['v_mfma_f32_32x32x16_bf16'], cc['v_accvgpr_read_b32'], cc['v_accvgpr_write_b32'],
                    cc['v_mov_b32_e32'], cc['s_waitcnt'], cc['s_nop'], cc['ds_read_b128'], cc['ds_read_b64_tr_b16'],
                    cc['global_load_lds_dwordx4'], cc['v_fma_f32'], cc['v_exp_f32'], cc['v_mul_f32'], cc['v_cvt_pk_bf16_f32'],
                    cc['v_add_u32_e32']))


main()
