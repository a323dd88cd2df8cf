#!/usr/bin/env python3
"""Price the hot loop of a kernel in libkaptive_amd.so with the measured gfx950 VALU issue costs.

    python tools/isa_cost.py [kernel-substring] [--marker v_addc_co_u32] [--top 1]

Compiles the source to gfx950 assembly, splits the kernel into basic blocks, takes the block(s) holding most `marker`
instructions (the fill kernel's 8-step body is the one full of v_addc_co_u32 bit pushes) and sums, per opcode, count x
issue cost: 2.35 cycles per wave64 for the opcodes profiles/valu_rate*_r2.txt measured at ~1.04 ns, 4 for the rest.
"""
import argparse
import collections
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
# Measured (profiles/valu_rate2_r2.txt, valu_dep_r2.txt): these issue every ~1.04 ns per SIMD, everything else measured
# every ~1.77 ns; with the 2.26 GHz the slow class implies (4 cycles) that is 2.35 cycles for the fast class.
FAST = {
    "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
    "v_mov_b32", "v_not_b32", "v_add_f32", "v_mul_f32", "v_fma_f32", "v_max_i16", "v_max_u16", "v_min_i16", "v_add_u16",
    "v_sub_u16", "v_lshlrev_b16", "v_lshrrev_b16", "v_mul_lo_u16",
}
FAST_CYCLES, SLOW_CYCLES = 2.35, 4.0


def base(op: str) -> str:
    return re.sub(r"_(e32|e64)$", "", op)  # _dpp / _sdwa forms keep their suffix: they measured at 4 cycles


def cost(op: str) -> float:
    return FAST_CYCLES if op in FAST else SLOW_CYCLES


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("kernel", nargs="?", default="kp_sw_kernel")
    ap.add_argument("--marker", default="v_pk_max_u16,v_pk_maximum3_f16", help="comma-separated opcodes that identify the hot blocks")
    ap.add_argument("--top", type=int, default=1)
    ap.add_argument("--source", default="kp_sw.hip")
    ap.add_argument("--flags", default="", help="extra compile flags")
    args = ap.parse_args()
    src = ROOT / "kaptive_amd" / "csrc" / args.source
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / "dev.s"
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                            f"-I{ROOT / 'include'}", *args.flags.split(), str(src), "-o", str(out)], capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr, file=sys.stderr)
            return 1
        text = out.read_text()
    blocks, cur, inside = [], [], False
    for line in text.splitlines():
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not m.group(1).startswith(".L"):  # a function label
            if inside and cur:
                blocks.append(cur)
            inside, cur = args.kernel in m.group(1), []
            continue
        if not inside:
            continue
        if re.match(r"^\.L\w+:", line):  # a basic-block label
            if cur:
                blocks.append(cur)
            cur = []
            continue
        ins = line.split(";")[0].strip().split()
        if not ins or ins[0].startswith("."):
            continue
        op = ins[0]
        cur.append(op)
        if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm":
            blocks.append(cur)
            cur = []
    # labels inside a kernel show up as "<L..>:" lines: treat them as block starts too
    blocks = [b for b in blocks if b]
    markers = set(args.marker.split(","))
    ranked = sorted(blocks, key=lambda b: -sum(1 for o in b if base(o) in markers))[: args.top]
    for b in ranked:
        cnt = collections.Counter(base(o) for o in b)
        valu = {o: n for o, n in cnt.items() if o.startswith("v_")}
        cyc = sum(n * cost(o) for o, n in valu.items())
        print(f"block of {len(b)} instructions: {sum(valu.values())} VALU, {sum(n for o, n in cnt.items() if o.startswith('s_'))} scalar, "
              f"{sum(n for o, n in cnt.items() if o.startswith('ds_'))} LDS, {sum(n for o, n in cnt.items() if o.startswith(('global_', 'buffer_', 'flat_')))} VMEM")
        for o, n in sorted(valu.items(), key=lambda kv: -kv[1] * cost(kv[0])):
            print(f"  {o:24s} {n:5d} x {cost(o):4.2f} = {n * cost(o):8.1f}")
        print(f"  VALU issue cycles for the block: {cyc:.0f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
