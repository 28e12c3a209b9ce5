#!/usr/bin/env python3
"""isa_census.py -- instruction census of a kernel AS SHIPPED in libswmi.so, and the VALU issue bound that follows.

    python tools/isa_census.py [--lib smith-waterman-simd_amd/lib/libswmi.so] [--kernel REGEX] [--json out.json]

What it does (no GPU needed; llvm-objdump from the ROCm image):
  1. pulls the gfx950 code objects out of the shared library's offload bundle (llvm-objdump --offloading, on a copy
     in a temporary directory) and disassembles them;
  2. for every kernel whose demangled-ish symbol matches REGEX: finds the loops (backward branches), takes the one with
     the largest body as the MAIN loop and counts wave-instructions per class inside it and outside it;
  3. prices the VALU instructions with the per-instruction issue costs measured on MI355X by tools/microbench/valu_rate*.hip
     (profiles/r01_microbench_valu_rate.txt, ..._more.txt, 8 wavefronts per SIMD): full rate 2 cycles per wave64
     instruction, half rate 4, quarter rate 8 -- the `ideal` cost -- and with the measured figures themselves;
  4. stamps the result with a hash of the kernel's instruction text, so that numbers taken with a profiler (PMC traffic,
     effective clock) can be tied to the exact code they were taken on.

bench.py imports census_for() to turn a measured kernel time into `roofline.frac` = VALU issue cycles needed /
SIMD cycles elapsed, which is a utilisation (<= 1) by construction.
"""
import argparse
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "smith-waterman-simd_amd", "lib", "libswmi.so")
OBJDUMP_CANDIDATES = ["/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/llvm/bin/llvm-objdump", "llvm-objdump"]

# ---- per-instruction issue cost, cycles per wave64 instruction on one SIMD (8 wavefronts per SIMD) ----------------
# ideal class / measured clk per instruction (profiles/r01_microbench_valu_rate.txt section "waves/SIMD = 8" and
# profiles/r01_microbench_valu_rate_more.txt section "waves/SIMD = 8")
FULL = {  # 2-cycle class
    "v_add_u32": 2.30, "v_sub_u32": 2.25, "v_subrev_u32": 2.30, "v_and_b32": 2.26, "v_or_b32": 2.27, "v_xor_b32": 2.26,
    "v_lshrrev_b32": 2.25, "v_ashrrev_i32": 2.27, "v_mov_b32": 2.21, "v_max_i16": 2.25, "v_max_u16": 2.30,
    "v_min_i16": 2.25, "v_min_u16": 2.30, "v_add_u16": 2.30, "v_sub_u16": 2.26, "v_subrev_u16": 2.26, "v_fma_f32": 2.29,
    "v_fmac_f32": 2.28, "v_add_f32": 2.33, "v_sub_f32": 2.33, "v_mul_f32": 2.26, "v_max_f16": 2.30, "v_add_f16": 2.24,
    "v_bitop3_b32": 2.37, "v_not_b32": 2.25, "v_accvgpr_write_b32": 2.21, "v_accvgpr_read_b32": 2.21, "v_accvgpr_mov_b32": 2.21,
}
HALF = {  # 4-cycle class
    "v_max_i32": 4.09, "v_max_u32": 4.15, "v_max3_u32": 4.15, "v_min3_u32": 4.15, "v_min_i32": 4.13, "v_min_u32": 4.15, "v_max3_i32": 4.13, "v_min3_i32": 4.17,
    "v_med3_i32": 4.25, "v_dot4_i32_i8": 4.17, "v_dot4c_i32_i8": 4.12, "v_perm_b32": 4.16, "v_bfe_i32": 4.16,
    "v_bfe_u32": 4.16, "v_add3_u32": 4.17, "v_lshl_add_u32": 4.17, "v_lshl_or_b32": 4.16, "v_and_or_b32": 4.18,
    "v_or3_b32": 4.17, "v_xad_u32": 4.17, "v_alignbit_b32": 4.17, "v_alignbyte_b32": 4.17, "v_bfi_b32": 4.17,
    "v_cndmask_b32": 4.17, "v_lshlrev_b32": 4.06, "v_mul_u32_u24": 4.08, "v_mad_i32_i24": 4.16, "v_mad_u32_u24": 4.16,
    "v_add_co_u32": 4.17, "v_addc_co_u32": 4.17, "v_sub_co_u32": 4.17, "v_subb_co_u32": 4.17, "v_sub_i32": 4.16,
    "v_add_i32": 4.19, "v_max_f32": 4.15, "v_min_f32": 4.15, "v_max3_f32": 4.17, "v_sad_u8": 4.17,
    "v_pk_add_u16": 4.16, "v_pk_max_i16": 4.16, "v_pk_max_u16": 4.16, "v_pk_sub_u16": 4.16, "v_pk_mad_i16": 4.16,
    "v_pk_add_f16": 4.08, "v_mad_legacy_u16": 4.17,
    # profiles/r02_microbench_valu_rate5.txt (8 wavefronts per SIMD)
    "v_lshlrev_b64": 4.27, "v_lshrrev_b64": 4.28, "v_ashrrev_i64": 4.28, "v_lshl_add_u64": 4.22, "v_mov_b64": 4.26,
    "v_readfirstlane_b32": 4.25, "v_readlane_b32": 4.32, "v_bcnt_u32_b32": 4.28, "v_ffbh_u32": 4.21, "v_ffbl_b32": 4.21,
    "v_bfm_b32": 4.21, "v_dot2_i32_i16": 4.27, "v_dot8_i32_i4": 4.18, "v_pk_maximum3_f16": 4.23, "v_pk_max_f16": 4.28,
}
QUARTER = {  # 8-cycle class
    "v_max3_i16": 8.43, "v_max3_u16": 8.44, "v_med3_i16": 8.46, "v_mad_i16": 8.18, "v_add_i16": 8.15, "v_sub_i16": 8.15,
}
# Anything else is priced at 4 and reported as `unmeasured`, so that a reader can see how much of a total rests on the
# default.  v_cmp_* (every compare, VCC or SGPR-pair destination) measured 4.05 - 4.35 and is matched by prefix below.
DEFAULT_COST = 4.0


def _objdump():
    for c in OBJDUMP_CANDIDATES:
        p = shutil.which(c) if not os.path.isabs(c) else (c if os.path.exists(c) else None)
        if p:
            return p
    raise RuntimeError("llvm-objdump not found")


def disassemble(lib_path=DEFAULT_LIB):
    """{kernel symbol: [(address, mnemonic, operand text)]} for every function in the library's gfx950 code objects."""
    objdump = _objdump()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([objdump, "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        for name in sorted(os.listdir(tmp)):
            if "amdgcn" not in name:
                continue
            text = subprocess.run([objdump, "-d", "--no-show-raw-insn", os.path.join(tmp, name)], stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL, text=True, check=True).stdout
            cur = None
            for line in text.splitlines():
                m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
                if m:
                    cur = m.group(2)
                    out[cur] = []
                    continue
                if cur is None or not line.startswith("\t"):
                    continue
                body, _, comment = line.partition("//")
                parts = body.strip().split(None, 1)
                if not parts:
                    continue
                am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
                addr = int(am.group(1), 16) if am else -1
                tm = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", comment)
                out[cur].append({"addr": addr, "op": parts[0], "args": parts[1] if len(parts) > 1 else "",
                                 "target_off": int(tm.group(1), 16) if tm else None})
    return out


KERNEL_NAMES = ("sw128_lut_kernel", "sw128_pk_kernel", "sw128_kernel", "sw_banded_affine_pk_kernel", "sw_banded_affine_kernel", "sw_banded_affine_tile_kernel",
                "sg_forward_split_kernel", "sg_forward_lane_kernel", "sg_walk_lane_kernel", "sg_expand_kernel",
                "sg_pack_streams_kernel", "pk_max3_selftest_kernel", "generate_kernel", "unpack_kernel")


def readable(symbol):
    """Itanium-mangled kernel name -> 'sw128_kernel<4,1,0,0>' style (template arguments of kind int / bool only)."""
    for name in KERNEL_NAMES:
        m = re.search(r"\d+%s(I((?:L[a-z]\d+E)+)E)?" % name, symbol)
        if m:
            if not m.group(2):
                return name
            args = re.findall(r"L[a-z](\d+)E", m.group(2))
            if name == "sw128_pk_kernel" and len(args) == 3 and args[2] == "4":
                args = args[:2]             # <MODE, VARIANT, L>: the L = 4 instantiation keeps its two-argument name
            return "%s<%s>" % (name, ",".join(args))
    return symbol


def cost_of(op):
    """(class name, ideal cycles, measured cycles) of one instruction mnemonic as objdump prints it."""
    base = re.sub(r"_(e32|e64)$", "", op)
    if not base.startswith("v_"):
        if base.startswith("s_"):
            return "salu", 0.0, 0.0
        if base.startswith("ds_"):
            return "lds", 0.0, 0.0
        if base.startswith(("global_", "buffer_", "flat_", "scratch_")):
            return "vmem", 0.0, 0.0
        return "other", 0.0, 0.0
    if base.endswith("_dpp") or base.endswith("_sdwa"):     # every DPP / SDWA form issues at half rate
        return "valu_half", 4.0, 4.16
    if base in FULL:
        return "valu_full", 2.0, FULL[base]
    if base in HALF:
        return "valu_half", 4.0, HALF[base]
    if base in QUARTER:
        return "valu_quarter", 8.0, QUARTER[base]
    if base.startswith("v_cmp_") or base.startswith("v_cmpx_"):
        return "valu_half", 4.0, 4.3
    return "valu_unmeasured", DEFAULT_COST, DEFAULT_COST


def census(instrs, marker_op=None, exclude_op=None, pick="innermost"):
    """Split a kernel at its main loop and price both parts.  Main loop = the backward branch with the largest body, or --
    when marker_op names an instruction the hot loop issues (v_dot4_i32_i8 for the scorers) -- the innermost loop that
    holds the most of them (a staging loop can be longer than a tight DP loop, and an enclosing loop holds more of everything).
    pick="most": the SMALLEST loop among those that hold the most markers (the semi-global sweeps: two sibling loops of eight
    unrolled rounds each, with out-of-line blocks that branch back into them -- "innermost" would take a fragment);
    exclude_op: loops that hold this instruction are not candidates (the sweeps' calm loop is the one WITHOUT the X-drop test)."""
    start = instrs[0]["addr"]
    loops = []
    for k, ins in enumerate(instrs):
        if ins["op"].startswith(("s_cbranch", "s_branch")) and ins["target_off"] is not None:
            tgt = start + ins["target_off"]
            if tgt <= ins["addr"]:
                first = next(i for i, x in enumerate(instrs) if x["addr"] >= tgt)
                loops.append((first, k))
    main = max(loops, key=lambda ab: ab[1] - ab[0]) if loops else None
    if loops and marker_op:
        def marks(ab):
            return sum(1 for x in instrs[ab[0]:ab[1] + 1] if re.sub(r"_(e32|e64)$", "", x["op"]) == marker_op)
        # innermost: a loop that holds the marker but no other loop that holds it (the packed scorer's sweep sits inside a
        # loop over the wavefront's sets of alignments, whose prologue uses the marker instruction too)
        def count(ab, op):
            return sum(1 for x in instrs[ab[0]:ab[1] + 1] if re.sub(r"_(e32|e64)$", "", x["op"]) == op)
        holders = [ab for ab in loops if marks(ab) > 0 and not (exclude_op and count(ab, exclude_op) > 0)]
        inner = [ab for ab in holders if not any(o != ab and ab[0] <= o[0] and o[1] <= ab[1] for o in holders)]
        if pick == "most" or exclude_op:
            if not holders:
                raise RuntimeError("no loop holds %s%s" % (marker_op, " without " + exclude_op if exclude_op else ""))
            best = max(marks(ab) for ab in holders)
            main = min((ab for ab in holders if marks(ab) == best), key=lambda ab: ab[1] - ab[0])
        elif inner:
            best = max(marks(ab) for ab in inner)
            main = min((ab for ab in inner if marks(ab) == best), key=lambda ab: ab[1] - ab[0])

    def tally(seq):
        t = {"instructions": len(seq), "valu": 0, "by_class": {}, "by_op": {}, "issue_cycles_ideal": 0.0,
             "issue_cycles_measured_rates": 0.0, "unmeasured_valu": 0}
        for ins in seq:
            cls, ideal, meas = cost_of(ins["op"])
            t["by_class"][cls] = t["by_class"].get(cls, 0) + 1
            if cls.startswith("valu"):
                base = re.sub(r"_(e32|e64)$", "", ins["op"])
                t["valu"] += 1
                t["by_op"][base] = t["by_op"].get(base, 0) + 1
                t["issue_cycles_ideal"] += ideal
                t["issue_cycles_measured_rates"] += meas
                if cls == "valu_unmeasured":
                    t["unmeasured_valu"] += 1
        t["issue_cycles_measured_rates"] = round(t["issue_cycles_measured_rates"], 2)
        return t

    conditional = []
    if main:
        a, b = main
        # Blocks inside the main loop that a forward SCALAR-condition branch jumps over run only on some iterations (a flush
        # every 16th round ...): they are tallied apart, and the caller says how often they run.  (Blocks under an EXEC mask,
        # s_cbranch_execz, are per-lane predication and count as always executed.)
        skip = set()
        for k in range(a, b + 1):
            ins = instrs[k]
            # (scc = a scalar compare such as `(round & 15) == 15`; vcc branches come from wavefront votes like __any(alive),
            # which guard the loop's exit, not an occasional block)
            if ins["op"] in ("s_cbranch_scc0", "s_cbranch_scc1") and ins["target_off"] is not None:
                tgt = start + ins["target_off"]
                if ins["addr"] < tgt <= instrs[b]["addr"]:
                    span = [i for i in range(k + 1, b + 1) if instrs[i]["addr"] < tgt]
                    if len(span) >= 8:
                        skip.update(span)
        inside = [instrs[i] for i in range(a, b + 1) if i not in skip]
        conditional = [instrs[i] for i in sorted(skip)]
        outside = instrs[:a] + instrs[b + 1:]
    else:
        inside, outside = [], instrs
    text = "\n".join("%s %s" % (i["op"], re.sub(r"\s+", " ", i["args"])) for i in instrs)
    return {"main_loop": tally(inside), "main_loop_conditional": tally(conditional), "outside_main_loop": tally(outside),
            "other_loops": max(0, len(loops) - 1),
            "code_sha256": hashlib.sha256(text.encode()).hexdigest()[:16], "code_bytes": max(i["addr"] for i in instrs) - start + 4}


_cache = {}


def census_for(kernel_regex, lib_path=DEFAULT_LIB, marker_op=None, exclude_op=None, pick="innermost"):
    """{readable kernel name: census} for the kernels of lib_path whose readable name matches kernel_regex."""
    key = (os.path.abspath(lib_path), os.path.getmtime(lib_path))
    if key not in _cache:
        _cache[key] = disassemble(lib_path)
    out = {}
    for sym, instrs in _cache[key].items():
        name = readable(sym)
        if instrs and re.search(kernel_regex, name):
            c = census(instrs, marker_op, exclude_op, pick)
            c["symbol"] = sym
            out[name] = c
    return out


def issue_cycles_per_wave(c, main_loop_trips, rates="ideal", conditional_share=1.0):
    """VALU issue cycles one wavefront needs: everything outside the main loop once + the loop body x trips, the loop's
    conditional blocks (see census) weighted by the share of trips they run on."""
    key = "issue_cycles_ideal" if rates == "ideal" else "issue_cycles_measured_rates"
    return c["outside_main_loop"][key] + main_loop_trips * (c["main_loop"][key] + conditional_share * c["main_loop_conditional"][key])


def valu_instructions_per_wave(c, main_loop_trips, conditional_share=1.0):
    return c["outside_main_loop"]["valu"] + main_loop_trips * (c["main_loop"]["valu"] + conditional_share * c["main_loop_conditional"]["valu"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=DEFAULT_LIB)
    ap.add_argument("--kernel", default=r"^sw128_kernel<4,1,0,0>$", help="regex on the readable kernel name")
    ap.add_argument("--json", default=None)
    ap.add_argument("--marker", default=None, help="instruction that identifies the hot loop (e.g. v_dot4_i32_i8)")
    ap.add_argument("--exclude", default=None, help="loops that hold this instruction are not candidates for the main loop")
    ap.add_argument("--pick", default="innermost", choices=["innermost", "most"], help="how the marker picks the loop (see census())")
    ap.add_argument("--list", action="store_true")
    args = ap.parse_args()
    if args.list:
        for sym in disassemble(args.lib):
            print(readable(sym))
        return 0
    res = census_for(args.kernel, args.lib, args.marker, args.exclude, args.pick)
    if not res:
        sys.stderr.write("no kernel matches %r\n" % args.kernel)
        return 1
    for name, c in sorted(res.items()):
        ml, ol = c["main_loop"], c["outside_main_loop"]
        print("%s  [code %s, %d bytes]" % (name, c["code_sha256"], c["code_bytes"]))
        print("  main loop : %4d instr, %4d VALU, %7.1f issue cycles per iteration (ideal 2/4/8), %7.1f at measured rates; %s"
              % (ml["instructions"], ml["valu"], ml["issue_cycles_ideal"], ml["issue_cycles_measured_rates"], ml["by_class"]))
        print("              VALU by op: %s" % dict(sorted(ml["by_op"].items(), key=lambda kv: -kv[1])))
        cl = c["main_loop_conditional"]
        if cl["instructions"]:
            print("  + blocks under a scalar condition (not every iteration): %4d instr, %4d VALU, %7.1f issue cycles (ideal)"
                  % (cl["instructions"], cl["valu"], cl["issue_cycles_ideal"]))
        print("  elsewhere : %4d instr, %4d VALU, %7.1f issue cycles (ideal), unmeasured VALU %d; other loops %d"
              % (ol["instructions"], ol["valu"], ol["issue_cycles_ideal"], ol["unmeasured_valu"], c["other_loops"]))
    if args.json:
        with open(args.json, "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
