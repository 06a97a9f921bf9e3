"""Rules about the kernel SOURCES that no numerical test would catch until the wrong day.

The matrix-core results of the AMIS kernels are consumed by ordinary VALU instructions, and it is the compiler's hazard recogniser
that keeps the wait states between the two (8 behind v_mfma_f32_16x16x32_bf16, 10 behind v_mfma_f32_16x16x4_f32 on gfx950).  It cannot
see through an inline asm: an asm that takes an MFMA operand or result in-out becomes, for the compiler, the register's definition, the
wait states disappear, and the kernel reads stale registers whenever the wave is not held up by something else -- wrong for a tenth of the
objects, different on every run, only at full occupancy (profiles/r04_bwd_bf16_projection.txt: what round 3 spent ten commits on).
So: no inline asm in the kernel sources except the one opaque-constant helper."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'epro-pnp_amd', 'csrc')


def _code_lines(path):
    """source lines with // comments and /* */ comments removed"""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', lambda m: '\n' * m.group(0).count('\n'), text, flags=re.S)
    for no, line in enumerate(text.splitlines(), 1):
        yield no, line.split('//', 1)[0]


def test_no_inline_asm_near_the_matrix_cores():
    found = []
    for name in sorted(os.listdir(CSRC)):
        if not name.endswith(('.h', '.hip', '.cpp')):
            continue
        for no, line in _code_lines(os.path.join(CSRC, name)):
            if re.search(r'\basm\b|__asm__', line):
                found.append((name, line.strip()))
    # to_vgpr: an empty asm on a wave-uniform CONSTANT (keeps it out of an SGPR operand slot); it never touches an MFMA operand or result
    assert found == [('pnp_math.h', 'asm volatile("" : "+v"(x));')], found


def test_mfma_only_through_the_wrappers():
    """every matrix instruction goes through wave_ops.h (mfma_16x16x4, mfma_16x16x32_bf16), where the rule above is written down"""
    users = []
    for name in sorted(os.listdir(CSRC)):
        if not name.endswith(('.h', '.hip', '.cpp')):
            continue
        for no, line in _code_lines(os.path.join(CSRC, name)):
            if '__builtin_amdgcn_mfma' in line:
                users.append(name)
    assert set(users) == {'wave_ops.h'}, users
