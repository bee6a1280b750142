#!/usr/bin/env python3
"""Scans the gfx950 machine code of a built library for a select on a stale condition code.

Seen with ROCm 7.2's compiler: `x = (a < b) ? p : q` on uniform 64-bit values was compiled to a
vector compare into VCC followed by `s_cselect_b32`, which reads SCC; the instruction that should
have copied "VCC != 0" into SCC was dropped, so the select saw the carry of whatever scalar add
came before (a partial tile was taken for a full one).  Nothing in the source is wrong when this
happens, so the build is checked instead: every s_cselect / s_cbranch_scc whose nearest earlier
SCC writer on the straight-line path is an add, subtract or shift (carry or "result != 0", never
meant as a condition in this code base) is reported.

    tools/check_scc.py path/to/libmrs.so        # extract code objects, disassemble, scan
    tools/check_scc.py file.s ...               # scan hipcc -S --cuda-device-only output
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"

SCC_LOGIC = re.compile(r"^\s*s_(cmp|cmpk|bitcmp|and|or|xor|andn2|orn2|nand|nor|xnor|not|bfe|min|max|abs|wqm|quadmask|"
                       r"bcnt|ff0|ff1|flbit|absdiff)")
SCC_ARITH = re.compile(r"^\s*s_(add_u32|add_i32|addc_u32|sub_u32|sub_i32|subb_u32|lshl_b32|lshl_b64|lshr_b32|lshr_b64|"
                       r"ashr_i32|ashr_i64|lshl[1-4]_add_u32)\b")
SCC_USE = re.compile(r"^\s*s_(cselect_b32|cselect_b64|cbranch_scc[01])\b")
# the carry of an add turned into a boolean (64-bit division and overflow checks do this): intended
CARRY_BOOL = re.compile(r"^\s*s_cselect_b(32|64)\s+\S+,\s*(-1|1),\s*0\s*$")
FUNC = re.compile(r"^(?:[0-9a-f]+ <(_Z\w+)>:|(_Z\w+):)")
LABEL = re.compile(r"^(?:[0-9a-f]+ <L\d+>:|\.LBB\S*:)")


def scan(lines, where):
    """Yields (location, function, use, writer) for every suspicious use in one listing."""
    func, last = "?", None
    for ln, line in enumerate(lines, 1):
        m = FUNC.match(line)
        if m:
            func, last = m.group(1) or m.group(2), None
            continue
        if LABEL.match(line):
            last = None  # a join: the condition may come from another path
            continue
        text = line.split("//")[0]
        if not text.strip() or text.lstrip().startswith(";"):
            continue
        if SCC_USE.match(text) and last is not None and not CARRY_BOOL.match(text.rstrip()):
            yield f"{where}:{ln}", func, text.strip(), last.strip()
        if SCC_LOGIC.match(text):
            last = None
        elif SCC_ARITH.match(text):
            last = text


def code_objects(lib_path):
    """The gfx950 code objects inside the .hip_fatbin section of a shared library."""
    with tempfile.TemporaryDirectory() as tmp:
        raw = os.path.join(tmp, "fatbin.bin")
        subprocess.run([f"{LLVM_BIN}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib_path, raw],
                       check=True)
        data = open(raw, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            return
        count = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(count):
            o, size, tsize = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tsize].decode()
            off += tsize
            if "gfx950" in triple and size:
                yield data[i + o:i + o + size]
        pos = i + 24


def scan_library(lib_path):
    found, objects = [], 0
    for k, blob in enumerate(code_objects(lib_path)):
        objects += 1
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(blob)
            f.flush()
            dis = subprocess.run([f"{LLVM_BIN}/llvm-objdump", "-d", "--symbolize-operands", f.name], check=True,
                                 capture_output=True, text=True).stdout
        found.extend(scan(dis.splitlines(), f"{os.path.basename(lib_path)}[code object {k}]"))
    return found, objects


def main(paths):
    found = []
    for path in paths:
        if path.endswith(".s"):
            found.extend(scan(open(path, errors="replace"), path))
        else:
            hits, objects = scan_library(path)
            if objects == 0:
                print(f"{path}: no gfx950 code object found")
                return 2
            found.extend(hits)
    for where, func, use, writer in found[:40]:
        print(f"{where}: {func}: {use}   <- SCC last written by: {writer}")
    print(f"{len(found)} suspicious use(s) of the scalar condition code")
    return 1 if found else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
