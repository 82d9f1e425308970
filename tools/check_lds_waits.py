#!/usr/bin/env python3
"""Build-time check of the hand-placed LDS waits in the convolution kernels (CPU only; needs the built objects).

    python tools/check_lds_waits.py [objects ...]      (default: deepvideocodec_amd/lib/obj/conv_mfma_f32_*.hip.o)

The fp32 K loop reads its operands with `asm volatile("ds_read_b32 ...")` and places the covering `s_waitcnt`
itself: the compiler believes the destination register is valid at once.  Nothing in the language stops it from
copying, spilling or re-materialising such a register between the read and the wait -- it just has not done so with the
compiler this was written against.  This script makes that an assertion on the machine code: it disassembles the gfx950
code objects and walks every kernel (all paths, loops included) with a model of the LGKM counter:

  * every `ds_read*` puts its destination registers on a pending list; LDS operations of a wave complete in order,
    so `s_waitcnt lgkmcnt(k)` retires all but the newest k pending LDS operations (with a scalar load pending too,
    only lgkmcnt(0) retires anything: scalar loads return out of order);
  * any other instruction that reads or writes a pending register is a violation;
  * scratch (private memory) instructions inside an fp32 convolution kernel are a violation as well (a spill could move
    an operand register through memory at the wrong time).

Exit status 0 and one summary line per object when everything holds.  The check covers compiler-placed reads too
(they trivially hold).  Validated with hipcc of ROCm 7.2.0 (the version is printed)."""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
FUNC = re.compile(r"^([0-9a-f]+) <(\S+)>:$")
INSN = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")
TARGET = re.compile(r"<(\S+?)\+0x([0-9a-f]+)>")
LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def vregs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def extract_device_code(obj, tmp):
    local = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, local)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, cwd=tmp)
    cos = [p for p in glob.glob(local + ".*") if "gfx950" in p]
    if not cos:
        raise SystemExit(f"{obj}: no gfx950 code object inside")
    return cos[0]


def parse(disasm):
    """-> {function name: [(address, mnemonic, operand text, branch target address or None)]}"""
    funcs, cur, start = {}, None, {}
    for line in disasm.splitlines():
        m = FUNC.match(line)
        if m:
            cur = m.group(2)
            start[cur] = int(m.group(1), 16)
            funcs[cur] = []
            continue
        m = INSN.match(line)
        if not m or cur is None:
            continue
        mnem, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        tgt = None
        if mnem.startswith("s_cbranch") or mnem == "s_branch":
            t = TARGET.search(line)
            if t and t.group(1) in start:
                tgt = start[t.group(1)] + int(t.group(2), 16)
        funcs[cur].append((addr, mnem, ops, tgt))
    return funcs


def check_function(name, insns):
    """Worklist over (instruction index, pending list); pending = tuple of (kind, frozenset of dest vregs)."""
    index = {a: i for i, (a, _, _, _) in enumerate(insns)}
    seen, work, problems, n_reads, reported = set(), [(0, ())], [], 0, set()
    is_fp32 = "conv_mfma_f32_kernel" in name and re.search(r"ELb[01]ELi0E", name) is not None
    while work:
        i, pending = work.pop()
        while i < len(insns):
            key = (i, pending)
            if key in seen:
                break
            seen.add(key)
            if len(seen) > 4_000_000:
                raise SystemExit(f"{name}: state space too large")
            addr, mnem, ops, tgt = insns[i]
            if mnem == "s_waitcnt":
                m = LGKM.search(ops)
                if m and pending:
                    k = int(m.group(1))
                    if any(kind == "smem" for kind, _ in pending):
                        if k == 0:
                            pending = ()
                    else:
                        pending = pending[len(pending) - k:] if k else ()
            elif mnem == "s_endpgm":
                break
            else:
                is_read = mnem.startswith("ds_read") or mnem.startswith("ds_load")
                # (a later LDS read into the same register is harmless: LDS data returns in order)
                touched = vregs(ops.split(",", 1)[1] if is_read and "," in ops else ops)
                for kind, dests in pending:
                    if dests & touched:
                        msg = f"{name[:60]}… @{addr:#x}: {mnem} {ops} touches v{sorted(dests & touched)} of a pending LDS read"
                        if msg not in reported:
                            reported.add(msg)
                            problems.append(msg)
                if mnem.startswith("scratch_") and is_fp32:
                    problems.append(f"{name[:60]}… @{addr:#x}: scratch access in an fp32 convolution kernel")
                if is_read:
                    dest = ops.split(",")[0]
                    pending = pending + (("ds", frozenset(vregs(dest))),)
                    n_reads += 1
                elif mnem.startswith("ds_"):
                    pending = pending + (("ds", frozenset()),)
                elif mnem.startswith("s_load") or mnem.startswith("s_buffer_load") or mnem in ("s_memtime", "s_memrealtime"):
                    pending = pending + (("smem", frozenset()),)
                if len(pending) > 64:   # the counter saturates far below this; keep the newest (older ones cannot be told apart)
                    pending = pending[-64:]
            if tgt is not None and tgt in index:
                work.append((index[tgt], pending))
                if mnem == "s_branch":
                    break
            i += 1
    return n_reads, problems


def main():
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                          "deepvideocodec_amd", "lib", "obj", "conv_mfma_f32_*.hip.o")))
    if not objs:
        raise SystemExit("no objects: build first (make -C deepvideocodec_amd/csrc)")
    ver = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.splitlines()
    print("compiler:", next((l for l in ver if "HIP version" in l), ver[0] if ver else "?"))
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            co = extract_device_code(obj, tmp)
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
            reads = kernels = 0
            for name, insns in parse(dis).items():
                if not insns:
                    continue
                n, problems = check_function(name, insns)
                kernels += 1
                reads += n
                for p in problems[:20]:
                    print("VIOLATION", p)
                bad += len(problems)
            print(f"{os.path.basename(obj)}: {kernels} kernels, {reads} LDS reads followed to their waits, "
                  f"{'ok' if bad == 0 else str(bad) + ' violations so far'}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
