#!/usr/bin/env python3
"""Builds variants of the packed kernel's two-row cell (sw128_pk_kernel) that differ only in the ORDER of the
eleven instructions and in where an s_nop follows one, as separate libraries -- how profiles/r02_pk_cell_order_search.txt was
made.  Here (no GPU needed):

    python tools/experiments/pk_cell_variants.py shipped=01001010100 none=00000000000 all=11111111111 swap=01001010000:0123456a789

then on the GPU box, per variant:   SWMI_LIB=$PWD/tools/experiments/pk_variants/libswmi_<name>.so python tools/quick_pk_timing.py

A variant is  name=<11 digits>[:<11 hex digits>]  -- digit k = what follows the k-th instruction issued (0 nothing, 1
`s_nop 0`, 2 `s_nop 1`); the optional second field permutes the instructions (index into M0 T1 S0 P2 A0 M1 T2 S1 P3 A1 B).
Every consumer of a packed result must stay at least one instruction behind its producer (quick_pk_timing.py compares every
variant's scores with the int32 kernel's, so a violated wait state shows up as mismatches, not as a silent error)."""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "smith-waterman-simd_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "experiments", "pk_variants")
INS = ["v_pk_maximum3_f16 %[x0], %[hq0], %[up], %[t]", "v_add_u32 %[t], %[hu0], %[sc]", "v_pk_sub_u16 %[hu0], %[x0], %[gq] clamp",
       "v_perm_b32 %[s2], %[cy], %[cx], %[sel2]", "v_add_u32 %[hq0], %[q], %[hu0]", "v_pk_maximum3_f16 %[x1], %[hq1], %[hq0], %[t]",
       "v_add_u32 %[t], %[hu1], %[s2]", "v_pk_sub_u16 %[hu1], %[x1], %[gq] clamp", "v_perm_b32 %[sc], %[cy], %[cx], %[sel3]",
       "v_add_u32 %[hq1], %[q], %[hu1]", "v_pk_maximum3_f16 %[best], %[best], %[x0], %[x1]"]


def block(pattern, order):
    lines = []
    for k, i in enumerate(order):
        lines.append('"%s\\n\\t"' % INS[i])
        if int(pattern[k]):
            lines.append('"s_nop %d\\n\\t"' % (int(pattern[k]) - 1))
    return "        asm volatile(" + "\n                     ".join(lines) + "\n"


def main():
    src = open(os.path.join(CSRC, "sw_kernels.hip")).read()
    start = src.index("    if constexpr (BIAS && !LAST) {\n        asm volatile(") + len("    if constexpr (BIAS && !LAST) {\n")
    end = src.index('                     : [hq0] "+v"(hq0), [hq1] "+v"(hq1), [hu0] "+v"(hu0), [hu1] "+v"(hu1), [best] "+v"(best), [t] "+v"(t),\n'
                    '                       [sc] "+v"(sc)', start)
    os.makedirs(OUT, exist_ok=True)
    objs = [os.path.join(CSRC, "..", "lib", o) for o in ("sg_kernels.o", "swmi_api.o", "swmi_multi.o")]
    for spec in sys.argv[1:]:
        name, _, rest = spec.partition("=")
        pattern, _, perm = rest.partition(":")
        order = [int(c, 16) for c in perm] if perm else list(range(11))
        assert len(pattern) == 11 and sorted(order) == list(range(11)), spec
        tmp = os.path.join(CSRC, "_variant.hip")
        open(tmp, "w").write(src[:start] + block(pattern, order) + src[end:])
        try:
            obj = os.path.join(OUT, name + ".o")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                                   "-c", "-o", obj, tmp])
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                                   os.path.join(OUT, "libswmi_%s.so" % name), obj] + objs + ["-ldl", "-lpthread"])
            os.remove(obj)
        finally:
            os.remove(tmp)
        print("built", name, pattern, "".join("%x" % i for i in order))


if __name__ == "__main__":
    main()
