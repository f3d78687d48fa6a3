#!/usr/bin/env python3
"""Code objects of the built library: unbundle, disassemble, check.

  tools/codeobj.py list   [lib]      kernels of every gfx950 code object: code bytes, VGPRs / AGPRs / SGPRs, scratch, LDS
  tools/codeobj.py check  [lib]      the build contracts below; exit status 1 when one is broken
  tools/codeobj.py disasm [lib] KERNEL-SUBSTRING      the disassembly of the matching kernels (stdout)

Contracts (`check`, also run by tests/test_host.py on the CPU):
 1. Long branches.  Above 128 KB of code some branches need the long form (s_getpc_b64 / s_add_u32 / s_addc_u32 /
    s_setpc_b64 through a scalar register pair).  LLVM's size estimate does not see inline asm, so the pair may be
    scavenged after register allocation -- and a scavenged pair may be LIVE (DESIGN: sc_pair_kernel faulted from 2^15 rows
    on).  A kernel passes when the pair of every long branch is used by NOTHING else in the kernel (the pair that
    `-mllvm -amdgpu-long-branch-factor` reserves up front has that property).
 2. Hand-allocated registers.  The lane / pair kernels keep their decoding tree in v[104 .. 255] behind inline asm and
    confine the compiler to v0 .. v103 (amdgpu_num_vgpr); the pair kernel also owns all AGPRs.  The note records of the
    code object must say: no scratch (private segment 0, no VGPR spills), and for the pair kernels no compiler AGPRs.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "polar-codes_amd", "lib", "libpolar_amd.so")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """[(index, bytes)] of the gfx950 code objects inside the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, "fatbin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fb])
        blob = open(fb, "rb").read()
    out = []
    pos = 0
    while True:
        at = blob.find(MAGIC, pos)
        if at < 0:
            break
        n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        end = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append((len(out), blob[at + off:at + off + size]))
            end = max(end, at + off + size)
        pos = max(end, at + len(MAGIC))
    return out


def kernel_notes(co_path):
    """{kernel symbol: {field: value}} from the AMDGPU metadata note."""
    import yaml
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co_path], capture_output=True, text=True).stdout
    out = {}
    for doc in re.findall(r"^\s*---\n(.*?)^\s*\.\.\.\s*$", txt, re.S | re.M):
        meta = yaml.safe_load(doc) or {}
        for k in meta.get("amdhsa.kernels", []):
            out[k[".symbol"]] = {key.lstrip("."): val for key, val in k.items() if key != ".args"}
    return out


def disassemble(co_path):
    """{function symbol: [instruction text, ...]}"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co_path], capture_output=True, text=True).stdout
    funcs = {}
    cur = None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        s = line.split("//")[0].strip()
        if s and not s.startswith("Disassembly"):
            cur.append(s)
    return funcs


def function_sizes(co_path):
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", co_path], capture_output=True, text=True).stdout
    sizes = {}
    for line in txt.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC":
            sizes[f[7]] = int(f[2])
    return sizes


def sregs_of(ins):
    """set of scalar register numbers an instruction text mentions"""
    regs = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", ins):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", ins):
        regs.add(int(m.group(1)))
    return regs


def long_branch_report(instrs):
    """Long-branch sequences of one function: [(index, pair, clean)], clean = the pair is used nowhere else."""
    seqs = []
    member = set()
    for i, ins in enumerate(instrs):
        m = re.match(r"s_getpc_b64 s\[(\d+):(\d+)\]", ins)
        if not m:
            continue
        pair = (int(m.group(1)), int(m.group(2)))
        # a long branch ends in s_setpc_b64 on the same pair within the next few instructions (an address computation
        # for a global -- s_getpc + s_add with a relocation -- does not)
        for j in range(i + 1, min(i + 6, len(instrs))):
            if re.match(r"s_setpc_b64 s\[%d:%d\]" % pair, instrs[j]):
                seqs.append((i, j, pair))
                member.update(range(i, j + 1))
                break
    rep = []
    if not seqs:
        return rep
    # scalar registers used outside the long-branch sequences themselves (one pass over the function)
    used = set()
    wanted = set()
    for _, _, pair in seqs:
        wanted.update(pair)
    for k, ins in enumerate(instrs):
        if k in member or " s" not in ins and "\ts" not in ins and ",s" not in ins:
            continue
        r = sregs_of(ins) & wanted
        if r:
            used |= r
            if used == wanted:
                break
    for i, j, pair in seqs:
        rep.append((i, pair, not (used & set(pair))))
    return rep


HAND_ALLOCATED = ("scl_lane", "sc_pair_kernel")  # kernels that own v[104..255] behind inline asm


def analyse(lib):
    rows = []
    problems = []
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as td:
        def load(item):  # (llvm-readelf / llvm-objdump per code object: subprocesses, so threads run them side by side)
            idx, data = item
            p = os.path.join(td, "co%d.o" % idx)
            open(p, "wb").write(data)
            return idx, kernel_notes(p), disassemble(p), function_sizes(p)
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
            loaded = list(pool.map(load, code_objects(lib)))
        for idx, notes, funcs, sizes in loaded:
            for sym, d in notes.items():
                name = d.get("name", sym).strip("'\"")
                fn = sym[:-3] if sym.endswith(".kd") else sym
                ins = funcs.get(fn, [])
                lb = long_branch_report(ins)
                row = dict(co=idx, kernel=fn, code_bytes=sizes.get(fn, 0), vgpr=int(d.get("vgpr_count", 0)), agpr=int(d.get("agpr_count", 0)),
                           sgpr=int(d.get("sgpr_count", 0)), scratch=int(d.get("private_segment_fixed_size", 0)),
                           vgpr_spill=int(d.get("vgpr_spill_count", 0)), sgpr_spill=int(d.get("sgpr_spill_count", 0)),
                           lds=int(d.get("group_segment_fixed_size", 0)), long_branches=len(lb),
                           long_branch_pairs=sorted({pair for _, pair, _ in lb}), unsafe=[pair for _, pair, clean in lb if not clean])
                rows.append(row)
                if row["unsafe"]:
                    problems.append("%s: %d long branch(es) through a register pair that other instructions use: %s" %
                                    (fn, len(row["unsafe"]), sorted(set(row["unsafe"]))))
                if any(h in fn for h in HAND_ALLOCATED):
                    # (the BIG lane instantiations spill two of the COMPILER's registers to scratch: its own business, 12 bytes;
                    #  anything larger means the register budget of the compiled half has changed -- look at it)
                    if row["scratch"] > 64:
                        problems.append("%s: hand-allocated kernel with %d B of scratch (%d VGPR spills)" % (fn, row["scratch"], row["vgpr_spill"]))
                    if "sc_pair_kernel" in fn and row["agpr"] != 256:
                        problems.append("%s: the pair kernel owns all 256 AGPRs, the descriptor says %d" % (fn, row["agpr"]))
                    # the compiler's own registers must stay below the hand-allocated range: every VGPR operand >= 104 in the
                    # disassembly comes from an inline-asm block, and those are written with fixed numbers on purpose; what
                    # the descriptor guarantees is that the kernel was given the whole file
                    # (vgpr_count of the note = VGPRs + AGPRs on gfx950's unified file)
                    if row["vgpr"] - row["agpr"] < 256:
                        problems.append("%s: hand-allocated kernel with fewer than 256 VGPRs (%d incl. %d AGPRs)" % (fn, row["vgpr"], row["agpr"]))
    return rows, problems


def main(argv):
    cmd = argv[1] if len(argv) > 1 else "list"
    lib = DEFAULT_LIB
    rest = argv[2:]
    if rest and os.path.exists(rest[0]):
        lib = rest[0]
        rest = rest[1:]
    if cmd == "disasm":
        pat = rest[0]
        with tempfile.TemporaryDirectory() as td:
            for idx, data in code_objects(lib):
                p = os.path.join(td, "co%d.o" % idx)
                open(p, "wb").write(data)
                for fn, ins in disassemble(p).items():
                    if pat in fn:
                        print("== %s (code object %d, %d instructions)" % (fn, idx, len(ins)))
                        print("\n".join(ins))
        return 0
    rows, problems = analyse(lib)
    if cmd == "list":
        for r in sorted(rows, key=lambda r: -r["code_bytes"]):
            print("%7d B  v%-3d a%-3d s%-3d scratch %-4d sgpr-spill %-3d long-br %-2d %s  %s" %
                  (r["code_bytes"], r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["sgpr_spill"], r["long_branches"],
                   "UNSAFE" if r["unsafe"] else ("pair %s" % r["long_branch_pairs"] if r["long_branches"] else ""), r["kernel"][:110]))
    for p in problems:
        print("PROBLEM:", p)
    return 1 if (problems and cmd == "check") else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
