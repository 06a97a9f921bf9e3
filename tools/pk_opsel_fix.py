#!/usr/bin/env python3
"""gfx950 erratum workaround on the DEVICE ASSEMBLY of a translation unit (used by epro-pnp_amd/build.py, and as an audit).

On the MI355X a packed fp32 VALU instruction -- v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 -- whose LOW lane takes the low half
of its first vector-register source and the high half of its second one (op_sel:[0,1] / op_sel:[0,1,x]; with a scalar-register
first source: `s[..], v, v op_sel:[0,0,1]`) returns wrong results now and then while a v_mfma_f32_16x16x32_bf16 executes on the same
SIMD (the wave's own or a neighbour's); never with one wave per SIMD, never behind the fp32 MFMA, never for the other op_sel
forms -- (hi, lo), (hi, hi), op_sel_hi of any kind (tools/ubench/pk_after_mfma.hip, profiles/r05_pk_opsel_erratum.txt).  The
compiler emits the form whenever the SLP vectoriser broadcasts a scalar that sits in an odd VGPR as the second operand of a packed
multiply / add / fma.

mul, add and the two factors of an fma commute: swapping src0 and src1 together with their op_sel / op_sel_hi / neg_lo / neg_hi bits
gives the same arithmetic with the halves taken as (hi, lo) -- a form that is clean.  This script does that swap.  Where the
pair is a factor and the addend (a swap does not help) it refuses: such a shape has to go at the source (amis_backward_mfma.hip
keeps the camera matrix in vector registers for that reason).

    pk_opsel_fix.py in.s out.s      rewrite; prints how many instructions were swapped and how many it could not fix (left as they are)
    pk_opsel_fix.py --audit in.s    list kernels that contain v_mfma_f32_16x16x32_bf16 AND the unsafe form (exit 1 if any)
The build runs both on every translation unit: kernels without a matrix instruction are rewritten too (they may share a SIMD with
one that has, from another stream), only a kernel WITH a bf16 MFMA fails the build.
"""
import re
import sys

PK = re.compile(r'^(\s*)(v_pk_(?:mul|add|fma)_f32)\s+(.*?)\s*$')
MOD = re.compile(r'\b(op_sel_hi|op_sel|neg_lo|neg_hi):\[([^\]]*)\]')


def split_operands(text):
    return [t.strip() for t in re.split(r',\s*(?![^\[]*\])', text)]


def parse(code):
    m = PK.match(code)
    if not m:
        return None
    rest = m.group(3)
    ops = split_operands(re.split(r'\s+(?=op_sel|neg_)', rest)[0])
    mods = [(k, [int(x) for x in v.split(',')]) for k, v in MOD.findall(rest)]
    return m.group(1), m.group(2), ops, mods


def unsafe(line):
    """(is packed fp32 arithmetic, its low lane reads (lo, hi) of its first two vector-register sources)"""
    p = parse(line.split(';')[0])
    if p is None:
        return False, False
    _, _, ops, mods = p
    srcs = ops[1:]
    sel = dict(mods).get('op_sel') or [0] * len(srcs)
    vg = [i for i, o in enumerate(srcs) if re.match(r'v\[\d+:\d+\]$', o)]
    return True, len(vg) >= 2 and sel[vg[0]] == 0 and sel[vg[1]] == 1


def fix_line(line):
    code, _, comment = line.partition(';')
    indent, op, ops, mods = parse(code)
    srcs = ops[1:]
    vg = [i for i, o in enumerate(srcs) if re.match(r'v\[\d+:\d+\]$', o)]
    if vg[:2] != [0, 1]:
        raise SystemExit(f'pk_opsel_fix: the (lo, hi) pair is not the two commuting sources, a swap does not help: {code.strip()}')
    ops[1], ops[2] = ops[2], ops[1]
    out = []
    for k, v in mods:
        v[0], v[1] = v[1], v[0]
        out.append(f'{k}:[{",".join(str(x) for x in v)}]')
    # (op_sel_hi defaults to all ones: nothing to add when it was absent -- a swap of two ones)
    return f'{indent}{op} {", ".join(ops)} {" ".join(out)}' + (f' ;{comment}' if comment else '')


def main():
    if sys.argv[1] == '--audit':
        kern, has_mfma, bad, report = None, False, 0, []
        for line in open(sys.argv[2]):
            m = re.match(r'^(_Z\w+):', line)
            if m:
                if kern and has_mfma and bad:
                    report.append((kern, bad))
                kern, has_mfma, bad = m.group(1), False, 0
                continue
            if 'v_mfma_f32_16x16x32_bf16' in line:
                has_mfma = True
            bad += unsafe(line)[1]
        if kern and has_mfma and bad:
            report.append((kern, bad))
        for k, n in report:
            print(f'{n:3d} unsafe packed fp32 instruction(s) next to v_mfma_f32_16x16x32_bf16 in {k}')
        sys.exit(1 if report else 0)
    src, dst = sys.argv[1], sys.argv[2]
    out, n, left = [], 0, 0
    for line in open(src).read().split('\n'):
        if unsafe(line)[1]:
            try:
                line = fix_line(line)
                n += 1
            except SystemExit:
                left += 1
        out.append(line)
    open(dst, 'w').write('\n'.join(out))
    print(f'pk_opsel_fix: {n} instruction(s) swapped, {left} left (no swap helps) in {src}')


if __name__ == '__main__':
    main()
