#!/usr/bin/env python3
"""Instruction-class histogram of a gfx950 kernel's hot loop, from the code object the library ships.

    python3 tools/isa_hist.py [za_amd/_obj/msm.o | lib.so | file.co | file.dis] [--kernel REGEX] [--blocks] [--json]

The device code object is taken out of the object's .hip_fatbin section (llvm-objcopy + clang-offload-bundler), disassembled
with llvm-objdump -d, and the kernel's control-flow graph is rebuilt from the branch targets.  The "hot loop" is the
strongly connected component that holds the most v_mad_u64_u32; its HOT PATH is one trip from the loop header with every
two-way branch resolved by one rule (see hot_path): EXEC branches not taken (guarded regions run, out-of-line rare code
does not), uniform SCC / VCC branches stay in the loop; --taken overrides single blocks.  --blocks lists the loop's blocks
with the path marked, so the assumption can be checked against SQ_INSTS_VALU from the PMC passes
(profiles/r0N_pmc_valu_instructions.json): static hot-path VALU count ~ measured VALU instructions per wave-level addition.

Classes (VALU unless noted): mad64 = v_mad_u64_u32 (the multiplier), mul = v_mul_lo/hi, sh64 = 64-bit shifts, add64 =
v_lshl_add_u64 (a 64-bit add: where clang joins two partial column sums), alignbit, sh32, logic = and/or/xor/bfe/bfi/and_or..., add = add/sub/addc/add3 (carry chains), cndmask, mov
(incl. accvgpr moves, dpp, readlane), cmp, other_valu; mem = global/flat/scratch/ds/buffer; wait = s_waitcnt/s_nop;
salu = every other scalar instruction.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def disassemble(path):
    if path.endswith(".dis"):
        return open(path).read()
    tmp = tempfile.mkdtemp(prefix="isa_hist_")
    co = path
    if not path.endswith(".co"):
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path, os.path.join(tmp, "x.o")], check=True)
        co = os.path.join(tmp, "dev.co")
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=" + TARGET, "--input=" + fat,
                        "--output=" + co], check=True)
    return subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout


def classify(op):
    if op.startswith("v_"):
        if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
            return "mad64"
        if op.startswith(("v_mul_lo", "v_mul_hi", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_i32", "v_mad_u32", "v_mad_i32")):
            return "mul"
        if op.startswith("v_lshl_add_u64"):
            return "add64"
        if op.startswith(("v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64")):
            return "sh64"
        if op.startswith("v_alignbit"):
            return "alignbit"
        if op.startswith(("v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_lshl_add_u32", "v_lshl_or_b32", "v_add_lshl", "v_lshlrev_b16", "v_lshrrev_b16")):
            return "sh32"
        if op.startswith(("v_and", "v_or", "v_xor", "v_bfe", "v_bfi", "v_not", "v_xnor", "v_xad")):
            return "logic"
        if op.startswith(("v_add", "v_sub", "v_subrev", "v_addc", "v_subb", "v_subbrev")):
            return "add"
        if op.startswith("v_cndmask"):
            return "cndmask"
        if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_swap", "v_permlane", "v_perm_b32")):
            return "mov"
        if op.startswith("v_cmp"):
            return "cmp"
        return "other_valu"
    if op.startswith(("global_", "flat_", "scratch_", "ds_", "buffer_", "s_load", "s_buffer_load", "s_store")):
        return "mem"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep")):
        return "wait"
    return "salu"


VALU = ("mad64", "mul", "sh64", "add64", "alignbit", "sh32", "logic", "add", "cndmask", "mov", "cmp", "other_valu")
# issue cost in cycles per wave64 instruction per SIMD (tools/ubench_valu.hip): the multiplier, 64-bit
# shifts / adds, carry-producing adds, alignbit and mul_lo are "quarter rate" (4), plain VOP2 logic / shifts / moves 2
SLOW = {"mad64": 4, "mul": 4, "sh64": 4, "add64": 4, "alignbit": 4, "add": 4, "sh32": 2, "logic": 2, "cndmask": 2, "mov": 2, "cmp": 2, "other_valu": 4}


def parse_function(dis, kernel_re):
    """-> (name, [(offset, op, target_offset|None)])"""
    rx = re.compile(kernel_re)
    lines = dis.splitlines()
    out, name, start = [], None, None
    for i, ln in enumerate(lines):
        m = re.match(r"^([0-9a-f]{16}) <(.+)>:$", ln)
        if m:
            if name is not None:
                break
            if rx.search(m.group(2)):
                name, start = m.group(2), int(m.group(1), 16)
            continue
        if name is None:
            continue
        m = re.match(r"^\s+(\S+)(.*?)//\s*([0-9A-Fa-f]+):", ln)
        if not m:
            continue
        op, rest, addr = m.group(1), m.group(2), int(m.group(3), 16)
        tgt = None
        if op.startswith(("s_cbranch", "s_branch")):
            t = re.search(r"<.*\+0x([0-9a-f]+)>\s*$", ln)
            tgt = start + int(t.group(1), 16) if t else (start if re.search(r"<[^+]+>\s*$", ln) else None)
        out.append((addr, op, tgt))
    if name is None:
        raise SystemExit("no kernel matches %r" % kernel_re)
    return name, out


def build_blocks(ins):
    addrs = [a for a, _, _ in ins]
    leaders = {addrs[0]}
    for k, (a, op, tgt) in enumerate(ins):
        if tgt is not None:
            leaders.add(tgt)
            if k + 1 < len(ins):
                leaders.add(addrs[k + 1])
        if op in ("s_endpgm",) and k + 1 < len(ins):
            leaders.add(addrs[k + 1])
    blocks, cur = [], None
    for a, op, tgt in ins:
        if a in leaders:
            cur = {"start": a, "ins": [], "succ": []}
            blocks.append(cur)
        cur["ins"].append((a, op, tgt))
    by_start = {b["start"]: i for i, b in enumerate(blocks)}
    for i, b in enumerate(blocks):
        a, op, tgt = b["ins"][-1]
        b["hist"] = {}
        for _, o, _ in b["ins"]:
            c = classify(o)
            b["hist"][c] = b["hist"].get(c, 0) + 1
        b["term"] = op if tgt is not None else None
        if op == "s_endpgm":
            continue
        if tgt is not None and tgt in by_start:
            b["succ"].append(by_start[tgt])      # [0] = taken
        if op != "s_branch" and i + 1 < len(blocks):
            b["succ"].append(i + 1)              # [-1] = fall through
    return blocks


def sccs(blocks):
    n = len(blocks)
    index, low, on, st, comp, out = [None] * n, [0] * n, [False] * n, [], [None] * n, []
    counter = [0]
    sys.setrecursionlimit(100000)

    def dfs(v):
        index[v] = low[v] = counter[0]
        counter[0] += 1
        st.append(v)
        on[v] = True
        for w in blocks[v]["succ"]:
            if index[w] is None:
                dfs(w)
                low[v] = min(low[v], low[w])
            elif on[w]:
                low[v] = min(low[v], index[w])
        if low[v] == index[v]:
            c = []
            while True:
                w = st.pop()
                on[w] = False
                comp[w] = len(out)
                c.append(w)
                if w == v:
                    break
            out.append(sorted(c))
    for v in range(n):
        if index[v] is None:
            dfs(v)
    return out


def add_hist(dst, src):
    for k, v in src.items():
        dst[k] = dst.get(k, 0) + v


def hot_path(blocks, loop, taken_at=()):
    """One trip around the loop from its header, resolving every two-way branch by ONE stated rule: a branch on EXEC
    (s_cbranch_execz = "skip the region when no lane wants it", s_cbranch_execnz = "some lane needs the out-of-line rare
    code") is NOT taken -- the region guarded by execz runs, the rare code behind execnz does not; a branch on SCC / VCC (a
    uniform loop test) goes to the side that stays in the loop.  taken_at: start offsets of blocks whose branch IS taken
    (for a region that the common trip skips, e.g. the first-point-of-an-item block of k_accumulate)."""
    inloop = set(loop)
    header = min(loop)   # first block of the loop in address order: the header for these kernels
    path, v, seen = [], header, set()
    while v not in seen:
        seen.add(v)
        path.append(v)
        b = blocks[v]
        succ = [w for w in b["succ"] if w in inloop]
        if not succ:
            break
        if b["term"] and b["term"].startswith("s_cbranch") and len(b["succ"]) == 2:
            nxt = b["succ"][0] if b["start"] in taken_at else b["succ"][-1]
            if nxt not in inloop:
                nxt = succ[0]
        else:
            nxt = succ[0]
        if nxt == header:
            break
        v = nxt
    cold = {i for i in loop if i not in path and blocks[i]["hist"].get("mad64", 0) >= 100}
    return header, path, cold


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("path", nargs="?", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "za_amd", "_obj", "msm.o"))
    ap.add_argument("--kernel", default=r"k_accumulateINS_2FpINS_8FqParamsEEENS_3FqLELi1E")
    ap.add_argument("--blocks", action="store_true", help="list the loop's basic blocks")
    ap.add_argument("--taken", action="append", default=[], help="offset (+0x...) of a block whose branch IS taken on the hot path (repeatable)")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    dis = disassemble(a.path)
    name, ins = parse_function(dis, a.kernel)
    blocks = build_blocks(ins)
    comps = [c for c in sccs(blocks) if len(c) > 1 or c[0] in blocks[c[0]]["succ"]]
    if not comps:
        raise SystemExit("no loop in " + name)
    loop = max(comps, key=lambda c: sum(blocks[i]["hist"].get("mad64", 0) for i in c))
    base = ins[0][0]
    header, path, cold = hot_path(blocks, loop, {base + int(t.lstrip("+"), 16) for t in a.taken})
    whole, inloop, hot = {}, {}, {}
    for b in blocks:
        add_hist(whole, b["hist"])
    for i in loop:
        add_hist(inloop, blocks[i]["hist"])
    for i in path:
        add_hist(hot, blocks[i]["hist"])

    def summary(h):
        valu = sum(h.get(k, 0) for k in VALU)
        cyc = sum(h.get(k, 0) * SLOW[k] for k in VALU)
        return {"total": sum(h.values()), "valu": valu, "valu_non_mad": valu - h.get("mad64", 0), "issue_cycles_model": cyc, **{k: h.get(k, 0) for k in VALU + ("mem", "wait", "salu")}}
    res = {"kernel": name, "function": summary(whole), "loop_all_blocks": summary(inloop), "loop_hot_path": summary(hot),
           "loop_blocks": len(loop), "hot_path_blocks": len(path), "cold_blocks": sorted(blocks[i]["start"] for i in cold)}
    if a.json:
        print(json.dumps(res, indent=1))
        return
    print("kernel:", name)
    cols = ("total", "valu", "valu_non_mad") + VALU + ("mem", "wait", "salu", "issue_cycles_model")
    print("%-18s" % "" + " ".join("%9s" % c[:9] for c in cols))
    for k in ("function", "loop_all_blocks", "loop_hot_path"):
        print("%-18s" % k + " ".join("%9d" % res[k][c] for c in cols))
    print("loop: %d blocks, hot path %d blocks (cold regions at %s)" % (len(loop), len(path), ", ".join("+0x%x" % (s - ins[0][0]) for s in res["cold_blocks"]) or "none"))
    if a.blocks:
        for i in loop:
            b = blocks[i]
            print("  block +0x%-6x %5d ins  mad64 %4d  valu %4d  %s succ %s%s" % (
                b["start"] - ins[0][0], len(b["ins"]), b["hist"].get("mad64", 0), sum(b["hist"].get(k, 0) for k in VALU),
                (b["term"] or "-").ljust(18), ["+0x%x" % (blocks[s]["start"] - ins[0][0]) for s in b["succ"]],
                "  HOT" if i in path else ("  cold" if i in cold else "")))


if __name__ == "__main__":
    main()
