#!/usr/bin/env python3
"""Generates pk_sweeps_gen.inc: the anti-diagonal sweep of sw128_pk_kernel (sw_kernels.hip) for the cell bodies whose
additions run as 64-bit instructions on ALIGNED REGISTER PAIRS, one asm statement per instruction, in a fixed order of issue.

Why generated: gfx950 issues every VALU instruction of the packed cell at ~4.3 cycles per wavefront whatever its class
(profiles/r03_microbench_cell_v3.txt), so the lever is the instruction COUNT, and the two additions of the cell halve
theirs when rows r and r+1 are added by ONE v_lshl_add_u64 -- which needs (row r, row r+1) in an even-aligned register
pair.  hipcc's inline asm cannot name the halves of a 64-bit operand, so the column lives in explicit register variables
(`register uint32_t h0 asm("v32")`), the asm text names the physical registers, and the operand lists tell the compiler
what is read and written.  Every statement is `asm volatile`: they are issued in the order written.

Sections (selected with PK_SECTION = 100 * variant + rows per lane before including the file inside the kernel body):
  variant 0 "q0"    every score + gap >= 0.  Per two rows: 2 v_perm (lookup), 1 v_lshl_add_u64 (both diagonal terms),
                    2 v_pk_maximum3_f16, 2 v_pk_sub_u16 clamp, 1 v_pk_maximum3_f16 (running best) = 8 instructions
                    (round 2: 9).  Rows 0 and R - 1, whose diagonal terms have no neighbour row, share a paired add too.
  variant 2 "vert"  some score + gap < 0 but every score + 2 gap >= 0.  Row r of a lane works in its own domain
                    D_r = gap * (r + 1): the value handed down the column then needs no subtraction -- `up` is the row above's
                    max3 result as it is, the dependency chain is max3 -> max3 -- and the diagonal term is
                    L_{r-1} + (score + 2 gap), never negative, with L_r = H_r + D_r.  Off the chain: H_r = x_r -sat (gap + D_r)
                    (exact floor at 0: what the diagonal and the running best need), L = H + D as a paired 64-bit add.
                    Lanes hand each other H itself (exact), so row 0 adds D_0 to what it takes as `up`: that addition and row 0's
                    diagonal term share one paired add.
                    Per two rows: 2 v_perm, 2 max3, 2 sat-sub, 2 paired adds, 1 max3 (best) = 9 instructions (round 2's
                    bias form: 11).  Pad columns look up 0, i.e. act as score -2 gap: harmless (kernel header).
The code expects, in the including scope: `col` (per-column table offsets), `tables_at(t, cx, cy)`, `rsel[R]`, `group_mask`,
`gap` (int), `L` (lanes per pair of alignments) and defines `pk_best` (packed running best) for the epilogue.

Run: python3 gen_pk_sweeps.py > pk_sweeps_gen.inc   (the Makefile does; tests/test_generated_sources.py checks the committed
file is what this script prints)."""
import sys

H0 = 32          # v32.. : the column (h for q0, L for vert), R registers, even-aligned pairs
HU = (16, 18)    # v[16:17], v[18:19]: H of the two rows of a block (vert), alternating by block
SC = (20, 22)    # v[20:21], v[22:23]: looked-up scores of rows (2k+1, 2k+2), alternating by block
TT = 24          # v[24:25]: diagonal terms of rows (2k+1, 2k+2)
D0 = 40          # s40.. : D_r = gap * (r + 1) in both halves, r = 0 .. R (vert)


def emit(lines, text, outs=(), ins=(), sregs=()):
    """one asm volatile statement; outs / ins = C variable names (pinned ones carry their register through the declaration)"""
    o = ", ".join('"=v"(%s)' % v for v in outs)
    i = ", ".join(['"v"(%s)' % v for v in ins] + ['"s"(%s)' % v for v in sregs])
    lines.append('    asm volatile("%s" : %s : %s);' % (text, o, i))


def step(variant, R, cx, cy, up, diag, out):
    """code of one anti-diagonal step: column tables cx / cy, `up` / `diag` from the lane before, hand-over value -> out"""
    c = []
    h = lambda r: "pk_h%d" % r
    # q0: row R - 2 alternates between two registers from step to step (each sits below one of the two hand-over registers,
    # see below): hr = the name a step reads (the previous step's value), hw = the name it writes
    q0_first = variant == 0 and diag == "pk_u1"
    hr = lambda r: ("pk_hpA" if q0_first else "pk_hpB") if variant == 0 and r == R - 2 else h(r)
    hw = lambda r: ("pk_hpB" if q0_first else "pk_hpA") if variant == 0 and r == R - 2 else h(r)
    sc = lambda k, half: "pk_sc%d" % (2 * (k % 2) + half)
    sc_reg = lambda k: SC[k % 2]
    # score lookups: row 0 into an ordinary register, row i >= 1 into its half of SC_{(i-1)//2}
    def P(i):
        if i >= R or (variant == 0 and i == R - 1):        # (q0 looks the last row up at the start of the step)
            c.append('    asm volatile("s_nop 0");')
            return
        dst = "pk_s0" if i == 0 else sc((i - 1) // 2, (i - 1) % 2)
        emit(c, "v_perm_b32 %0, %1, %2, %3", [dst], [cy, cx, "rsel[%d]" % i])
    t0 = "pk_t0"
    if variant == 2:
        # Row 0's two additions -- the diagonal term  diag + score  and, because the lane before hands over H itself, `up` + D_0
        # (row 0's max3 runs in domain D_0) -- are ONE v_lshl_add_u64: the two hand-over values live in the aligned pair
        # v[30:31] (pk_u0, pk_u1; which of them is this step's `up` alternates), the addend pair is (D_0, score) or
        # (score, D_0) accordingly, the sums land in v[28:29].
        first = up == "pk_u0"                              # v30 is `up`: addends (D_0, score) = v[26:27]; else (score, D_0) = v[14:15]
        s0 = "pk_sA" if first else "pk_sB"
        emit(c, "v_perm_b32 %0, %1, %2, %3", [s0], [cy, cx, "rsel[0]"])
        P(1)
        if first:
            emit(c, "v_lshl_add_u64 v[28:29], v[30:31], 0, v[26:27]", ["pk_ta", "pk_tb"], ["pk_u0", "pk_u1", "pk_dA", "pk_sA"])
            up, t0 = "pk_ta", "pk_tb"
        else:
            emit(c, "v_lshl_add_u64 v[28:29], v[30:31], 0, v[14:15]", ["pk_ta", "pk_tb"], ["pk_u0", "pk_u1", "pk_sB", "pk_dB"])
            up, t0 = "pk_tb", "pk_ta"
    else:
        # The two diagonal terms that have no neighbour row to pair with -- row 0's (hand-over value + score) and the last
        # row's (row R - 2 of the previous column + score) -- are ONE v_lshl_add_u64: row R - 2 and the hand-over value sit in
        # an aligned pair, v[H0+R-2 : H0+R-1] in one step and v[H0+R+2 : H0+R+3] in the next (both alternate), the two looked-up
        # scores in v[26:27], the sums land in v[28:29].
        pair = H0 + R - 2 if q0_first else H0 + R + 2
        emit(c, "v_perm_b32 %0, %1, %2, %3", ["pk_s0"], [cy, cx, "rsel[0]"])
        emit(c, "v_perm_b32 %0, %1, %2, %3", ["pk_sL"], [cy, cx, "rsel[%d]" % (R - 1)])
        P(1)
        emit(c, "v_lshl_add_u64 v[28:29], v[%d:%d], 0, v[26:27]" % (pair, pair + 1), ["pk_ttL", "pk_t0"], [hr(R - 2), diag, "pk_sL", "pk_s0"])
    P(2)
    nblk = R // 2
    for k in range(nblk):
        a, b = 2 * k, 2 * k + 1
        t_a = t0 if k == 0 else "pk_tt1"
        last = b + 1 >= R
        if variant == 0:
            emit(c, "v_pk_maximum3_f16 %0, %1, %2, %3", ["pk_xa"], [hr(a), up if k == 0 else h(a - 1), t_a])
            if not last:
                emit(c, "v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, v[%d:%d]" % (TT, TT + 1, H0 + a, H0 + b, sc_reg(k), sc_reg(k) + 1),
                     ["pk_tt0", "pk_tt1"], [h(a), h(b), sc(k, 0), sc(k, 1)])
            else:
                c.append('    asm volatile("s_nop 0");')     # (the last row's diagonal term came with row 0's)
            emit(c, "v_pk_sub_u16 %0, %1, %2 clamp", [hw(a)], ["pk_xa"], ["pk_g2"])
            P(b + 2)
            emit(c, "v_pk_maximum3_f16 %0, %1, %2, %3", ["pk_xb"], [h(b), hw(a), "pk_ttL" if last else "pk_tt0"])
            P(b + 3)
            emit(c, "v_pk_sub_u16 %0, %1, %2 clamp", [h(b)], ["pk_xb"], ["pk_g2"])
            emit(c, "v_pk_maximum3_f16 %0, %1, %2, %3", ["pk_best"], ["pk_best", "pk_xa", "pk_xb"])
        else:
            hu = lambda kk, half: "pk_hu%d" % (2 * (kk % 2) + half)
            hu_reg = lambda kk: HU[kk % 2]
            emit(c, "v_pk_maximum3_f16 %0, %1, %2, %3", ["pk_xa"], [h(a), up if k == 0 else "pk_xb", t_a])
            if k:
                emit(c, "v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, s[%d:%d]" % (H0 + a - 2, H0 + a - 1, hu_reg(k - 1), hu_reg(k - 1) + 1, D0 + a - 2, D0 + a - 1),
                     [h(a - 2), h(a - 1)], [hu(k - 1, 0), hu(k - 1, 1)], ["pk_d%d" % (a - 2), "pk_d%d" % (a - 1)])
            else:
                c.append('    asm volatile("s_nop 0");')
            emit(c, "v_pk_sub_u16 %0, %1, %2 clamp", [hu(k, 0)], ["pk_xa"], ["pk_d%d" % (a + 1)])
            if not last:
                emit(c, "v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, v[%d:%d]" % (TT, TT + 1, H0 + a, H0 + b, sc_reg(k), sc_reg(k) + 1),
                     ["pk_tt0", "pk_tt1"], [h(a), h(b), sc(k, 0), sc(k, 1)])
            else:
                emit(c, "v_add_u32 %0, %1, %2", ["pk_tt0"], [h(a), sc(k, 0)])
            if k:
                emit(c, "v_pk_maximum3_f16 %0, %1, %2, %3", ["pk_best"], ["pk_best", hu(k - 1, 0), hu(k - 1, 1)])
            P(b + 2)
            emit(c, "v_pk_maximum3_f16 %0, %1, %2, %3", ["pk_xb"], [h(b), "pk_xa", "pk_tt0"])
            P(b + 3)
            emit(c, "v_pk_sub_u16 %0, %1, %2 clamp", [hu(k, 1)], ["pk_xb"], ["pk_d%d" % (b + 1)])
    if variant == 0:
        c.append('    asm volatile("s_nop 1");            // the DPP below reads a register the asm above wrote: 2 wait states, by hand')
        c.append("    %s = (uint32_t)from_prev_lane<L>((int)%s, group_mask);" % (out, h(R - 1)))
    else:
        k = nblk - 1
        c.append('    asm volatile("s_nop 0");')
        emit(c, "v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, s[%d:%d]" % (H0 + R - 2, H0 + R - 1, HU[k % 2], HU[k % 2] + 1, D0 + R - 2, D0 + R - 1),
             [h(R - 2), h(R - 1)], ["pk_hu%d" % (2 * (k % 2)), "pk_hu%d" % (2 * (k % 2) + 1)], ["pk_d%d" % (R - 2), "pk_d%d" % (R - 1)])
        emit(c, "v_pk_maximum3_f16 %0, %1, %2, %3", ["pk_best"], ["pk_best", "pk_hu%d" % (2 * (k % 2)), "pk_hu%d" % (2 * (k % 2) + 1)])
        c.append("    %s = (uint32_t)from_prev_lane<L>((int)pk_hu%d, group_mask);   // H of the lane's last row, exact" % (out, 2 * (k % 2) + 1))
    return c


def section(variant, R):
    o = []
    o.append("#if PK_SECTION == %d" % (100 * variant + R))
    o.append("{")
    o.append("    // ---- generated by gen_pk_sweeps.py: %s cell, %d rows per lane ----" % ("q0" if variant == 0 else "vertical-offset", R))
    for r in range(R):
        init = "0" if variant == 0 else "pk_dval(%d)" % r
        if variant == 0 and r == R - 2:
            o.append('    register uint32_t pk_hpA asm("v%d") = 0, pk_hpB asm("v%d") = 0;   // row %d, alternating by step' % (H0 + R - 2, H0 + R + 2, r))
        elif variant == 0 and r == R - 1:
            o.append('    register uint32_t pk_h%d asm("v%d") = 0;' % (r, H0 + R))
        else:
            o.append('    register uint32_t pk_h%d asm("v%d") = %s;' % (r, H0 + r, init))
    for q in range(4):
        o.append('    register uint32_t pk_sc%d asm("v%d") = 0;' % (q, SC[0] + q))
    o.append('    register uint32_t pk_tt0 asm("v%d") = 0, pk_tt1 asm("v%d") = 0;' % (TT, TT + 1))
    if variant == 2:
        for q in range(4):
            o.append('    register uint32_t pk_hu%d asm("v%d") = 0;' % (q, HU[0] + q))
        for r in range(R + 1):
            o.append('    register uint32_t pk_d%d asm("s%d") = pk_dval(%d);' % (r, D0 + r, r))
    else:
        o.append("    const uint32_t pk_g2 = (uint32_t)gap | ((uint32_t)gap << 16);")
    if variant == 2:
        o.append("    uint32_t pk_xa = 0, pk_xb = 0;")
        o.append('    register uint32_t pk_u0 asm("v30") = 0, pk_u1 asm("v31") = 0;   // H(last row of the lane before): this step\'s column / the previous one, alternating')
        o.append('    register uint32_t pk_ta asm("v28") = 0, pk_tb asm("v29") = 0;   // row 0: `up` + D_0 and the diagonal term, in the order of (pk_u0, pk_u1)')
        o.append('    register uint32_t pk_dA asm("v26") = pk_dval(0), pk_sA asm("v27") = 0;   // addends (D_0, score of row 0)')
        o.append('    register uint32_t pk_sB asm("v14") = 0, pk_dB asm("v15") = pk_dval(0);   // addends (score of row 0, D_0)')
        o.append('    asm volatile("" : "+v"(pk_dA), "+v"(pk_dB));      // opaque: or hipcc re-creates the two constants from the scalar D_0 every step')
    else:
        o.append("    uint32_t pk_xa = 0, pk_xb = 0;")
        o.append('    register uint32_t pk_u1 asm("v%d") = 0, pk_u0 asm("v%d") = 0;   // H(last row of the lane before): this step\'s column / the previous one, alternating; each above one copy of row %d' % (H0 + R - 1, H0 + R + 3, R - 2))
        o.append('    register uint32_t pk_sL asm("v26") = 0, pk_s0 asm("v27") = 0;   // scores of the last row and of row 0')
        o.append('    register uint32_t pk_ttL asm("v28") = 0, pk_t0 asm("v29") = 0;  // their diagonal terms')
    o.append("    uint32_t pk_x0, pk_y0;")
    o.append("    tables_at(0, pk_x0, pk_y0);")
    o.append("    for (int t2 = 0; t2 < T2; ++t2) {")
    o.append("        uint32_t pk_x1, pk_y1, pk_x2, pk_y2;")
    o.append("        tables_at(2 * t2 + 1, pk_x1, pk_y1);")
    o += ["    " + l for l in step(variant, R, "pk_x0", "pk_y0", "pk_u0", "pk_u1", "pk_u1")]
    o.append("        tables_at(2 * t2 + 2, pk_x2, pk_y2);")
    o += ["    " + l for l in step(variant, R, "pk_x1", "pk_y1", "pk_u1", "pk_u0", "pk_u0")]
    o.append("        pk_x0 = pk_x2; pk_y0 = pk_y2;")
    o.append("    }")
    o.append("    {   // step 128 + L - 2, the last lane's last column")
    o.append("        uint32_t pk_last;")
    o += ["    " + l for l in step(variant, R, "pk_x0", "pk_y0", "pk_u0", "pk_u1", "pk_last")]
    o.append("        (void)pk_last;")
    o.append("    }")
    o.append("}")
    o.append("#endif")
    return o


def main():
    out = ["// GENERATED by gen_pk_sweeps.py -- do not edit.  Included inside sw128_pk_kernel (sw_kernels.hip) with PK_SECTION set."]
    for variant in (0, 2):
        for R in (32, 16, 8):
            out += section(variant, R)
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
