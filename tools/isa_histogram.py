#!/usr/bin/env python3
"""Instruction histogram of a step kernel, from the disassembly of the built library (static: what the code object
holds, not what a run executes).

  python tools/isa_histogram.py [library.so | file.hsaco] [substring of the demangled kernel name] [--loops] [--json out.json]

For the kernel whose demangled name contains the substring (default: the C4 bench kernel -- fp64, monopod, contact,
per-env parameters, default solver settings, free_hip layout, not the counting variant) prints the code size and the
instruction mix by class: fp64 arithmetic (FMA / MUL / ADD / min-max), the moves that are pure overhead for a wave that
owns its SIMD (AGPR moves `v_accvgpr_*`, SGPR-spill lane moves `v_readlane / v_writelane`, `v_mov`, `v_cndmask`),
compares, conversions, scalar ALU, scalar loads, LDS, global memory, waits (`s_waitcnt`), branches.
--loops additionally lists the backward branches (loops) with the mix of each loop body: the sweep loops and the
exact-finish loop of the contact solver are the bodies that matter.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_meta import LLVM, code_objects, demangle   # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_KERNEL = "step_kernel<double, os2r::StModel<double, 0>, true, true, true, os2r::StLayout"

CLASSES = [
    ("fp64 fma", r"v_fma_f64|v_fmac_f64"),
    ("fp64 mul", r"v_mul_f64"),
    ("fp64 add", r"v_add_f64"),
    ("fp64 min/max", r"v_(min|max)_f64"),
    ("fp64 other (rcp, rsq, ldexp, ...)", r"v_\w+_f64"),
    ("agpr move", r"v_accvgpr_(read|write|mov)"),
    ("sgpr-spill lane move", r"v_(readlane|writelane|readfirstlane)"),
    ("v_mov", r"v_mov_|v_pk_mov"),
    ("v_cndmask", r"v_cndmask"),
    ("v_cmp", r"v_cmp"),
    ("other valu", r"v_"),
    ("s_waitcnt", r"s_waitcnt"),
    ("s_load (scalar memory)", r"s_load|s_buffer_load"),
    ("branch", r"s_cbranch|s_branch"),
    ("s_nop / s_sleep", r"s_nop|s_sleep"),
    ("other salu", r"s_"),
    ("lds", r"ds_"),
    ("global / flat / scratch memory", r"global_|flat_|scratch_|buffer_"),
]


def classify(mn):
    for name, pat in CLASSES:
        if re.match(pat, mn):
            return name
    return "other"


def disassemble(path, needle, counting=False):
    """-> (demangled name, [(address, mnemonic, operands)]) of the first kernel whose name contains `needle`"""
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", f.name], capture_output=True, text=True).stdout
            funcs = [ln.split() for ln in syms.splitlines() if " FUNC " in ln]
            names = [x[7] for x in funcs]
            for x, dn in zip(funcs, demangle(names)):
                # the last template arguments: COUNT, SOLVER, ROLLOUT (the fused K-step variants are listed only when asked for by name)
                tail = re.search(r", (true|false), (\d+), (true|false)>\(os2r::StepArgs<", dn)
                is_counting = tail is not None and tail.group(1) == "true"
                is_rollout = tail is not None and tail.group(3) == "true"
                if needle in dn and is_counting == counting and (not is_rollout or "true>(" in needle):
                    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f"--disassemble-symbols={x[7]}", f.name],
                                         capture_output=True, text=True).stdout
                    insts = []
                    for ln in dis.splitlines():
                        m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
                        if m:
                            insts.append((int(m.group(3), 16), m.group(1), m.group(2)))
                    return dn, int(x[2]), insts
    raise SystemExit(f"no kernel containing {needle!r} in {path}")


def mix(insts):
    c = collections.Counter(classify(mn) for _, mn, _ in insts)
    return c


def report(title, insts, out):
    c = mix(insts)
    n = len(insts)
    valu = sum(v for k, v in c.items() if k.startswith("fp64") or k in ("agpr move", "sgpr-spill lane move", "v_mov", "v_cndmask", "v_cmp", "other valu"))
    arith = sum(v for k, v in c.items() if k in ("fp64 fma", "fp64 mul", "fp64 add", "fp64 min/max"))
    print(f"{title}: {n} instructions, {valu} VALU of which {arith} fp64 FMA/MUL/ADD/min-max ({100.0 * arith / max(valu, 1):.1f} %), "
          f"non-arithmetic VALU {valu - arith} ({100.0 * (valu - arith) / max(valu, 1):.1f} %)")
    for name, _ in CLASSES + [("other", "")]:
        if c.get(name):
            print(f"    {name:36s} {c[name]:7d}  {100.0 * c[name] / n:5.1f} %")
    out[title] = {"instructions": n, "valu": valu, "fp64_arithmetic": arith, "classes": dict(c)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path", nargs="?", default=os.path.join(ROOT, "gym-os2r_amd", "libos2r.so"))
    ap.add_argument("kernel", nargs="?", default=DEFAULT_KERNEL)
    ap.add_argument("--loops", action="store_true")
    ap.add_argument("--min-loop", type=int, default=40, help="smallest loop body (instructions) worth listing")
    ap.add_argument("--marks", action="store_true",
                    help="for an object built with -DOS2R_ISA_MARKS: instructions between consecutive `s_nop 9..14` markers, i.e. per "
                         "pass of the exact solve (pass 1, factorisation + proximal solves, pass 2, step length, apply), once per "
                         "instantiated form of the solver")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    name, size, insts = disassemble(a.path, a.kernel)
    out = {"kernel": name.strip(), "code_bytes": size}
    print(f"{name.strip()}\n  code size {size} bytes ({size / 1024:.1f} KiB)")
    report("whole kernel", insts, out)
    if a.loops:
        addr_index = {ad: i for i, (ad, _, _) in enumerate(insts)}
        loops = []
        for i, (ad, mn, ops) in enumerate(insts):
            if mn.startswith("s_cbranch") or mn == "s_branch":
                tgt = None
                m = re.search(r"<[^>]*\+0x([0-9A-Fa-f]+)>", ops)
                if m:                                  # symbolic form: <kernel+0x1234>
                    tgt = insts[0][0] + int(m.group(1), 16)
                elif re.fullmatch(r"\d+", ops.strip()):  # raw simm16: dwords relative to the next instruction
                    off = int(ops)
                    tgt = ad + 4 + 4 * (off - 65536 if off > 32767 else off)
                if tgt is not None and tgt in addr_index and addr_index[tgt] <= i and i - addr_index[tgt] >= a.min_loop:
                    loops.append((addr_index[tgt], i))
        # innermost first, drop exact duplicates
        seen = set()
        for lo, hi in sorted(loops, key=lambda x: x[1] - x[0]):
            if (lo, hi) in seen:
                continue
            seen.add((lo, hi))
            report(f"loop body at +0x{insts[lo][0] - insts[0][0]:x} .. +0x{insts[hi][0] - insts[0][0]:x}", insts[lo:hi + 1], out)
    if a.marks:
        names = {9: "pass 1 (S, h)", 10: "factorisation + proximal solves", 11: "pass 2 (impulses, cut test)", 12: "step length (cut steps only)",
                 13: "apply (cut and full-step variants)"}
        marks = [(i, int(o)) for i, (_, m, o) in enumerate(insts) if m == "s_nop" and o.strip().isdigit() and 9 <= int(o) <= 14]
        for k in range(len(marks) - 1):
            (i, m0), (j, m1) = marks[k], marks[k + 1]
            if m1 == m0 + 1:
                report(f"exact solve, {names.get(m0, m0)} [instantiation {1 + sum(1 for _, mm in marks[:k + 1] if mm == 9)}]", insts[i + 1:j], out)
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
