#!/usr/bin/env python3
"""Timing experiments on the EMITTED code: a copy of libtamago_hip.so in which, inside one kernel, every `s_waitcnt` loses its
lgkmcnt and / or vmcnt part, every `s_barrier` / `s_nop N` becomes `s_nop 0` (same instruction sizes, nothing moves).  Results of the
patched kernel are WRONG (it reads registers and LDS before the data is there); its run time says what the waits cost, i.e. what a
schedule that hid them completely would gain.  tools/experiments/w1_time.py / wb_time.py take the patched library by name.

    python tools/experiments/patch_waits.py <kernel name substring> <modes: lgkm,vm,barrier,nop> <out.so>"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "..", "tamago_amd", "libtamago_hip.so")


def elf_size(buf, off):
    (e_shoff,) = struct.unpack_from("<Q", buf, off + 0x28)
    e_shentsize, e_shnum = struct.unpack_from("<HH", buf, off + 0x3A)
    return e_shoff + e_shentsize * e_shnum


def text_map(buf, off):
    """(sh_addr, sh_offset, sh_size) of .text of the ELF at `off`"""
    (e_shoff,) = struct.unpack_from("<Q", buf, off + 0x28)
    e_shentsize, e_shnum, e_shstrndx = struct.unpack_from("<HHH", buf, off + 0x3A)

    def sh(i):
        return struct.unpack_from("<IIQQQQIIQQ", buf, off + e_shoff + i * e_shentsize)
    strtab = sh(e_shstrndx)
    for i in range(e_shnum):
        s = sh(i)
        name = bytes(buf[off + strtab[4] + s[0]: off + strtab[4] + s[0] + 32]).split(b"\0", 1)[0]
        if name == b".text":
            return s[3], s[4], s[5]
    return None


def main():
    flt, modes, out = sys.argv[1], set(sys.argv[2].split(",")), sys.argv[3]
    buf = bytearray(open(LIB, "rb").read())
    n_patched = {}
    for m in re.finditer(b"\x7fELF", bytes(buf)):
        off = m.start()
        if off == 0 or buf[off + 0x12] != 0xE0:            # e_machine: EM_AMDGPU = 224
            continue
        size = elf_size(buf, off)
        tm = text_map(buf, off)
        if not tm:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(buf[off:off + size])
            path = f.name
        txt = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", path], text=True)
        os.unlink(path)
        cur = None
        # MFMA-dense regions only (the tower, the stem): the prologue's waits guard kernel arguments and addresses
        lines = txt.splitlines()
        is_mfma = [("v_mfma" in ln.split("//")[0]) for ln in lines]
        pref = [0]
        for b in is_mfma:
            pref.append(pref[-1] + (1 if b else 0))
        def dense(i):
            lo, hi = max(0, i - 50), min(len(lines), i + 50)
            return pref[i] - pref[lo] >= 5 and pref[hi] - pref[i] >= 5      # (MFMAs on both sides: not the code behind the tower's last row)
        for li, line in enumerate(lines):
            sm = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if sm:
                cur = sm.group(1)
                continue
            if cur is None or flt not in cur or "//" not in line:
                continue
            body, tail = line.split("//", 1)
            am = re.match(r"\s*([0-9A-Fa-f]+):\s*([0-9A-Fa-f]{8})", tail)
            if not am or not body.split():
                continue
            addr, word = int(am.group(1), 16), int(am.group(2), 16)
            op = body.split()[0]
            new = None
            if op.startswith("global_atomic_or") and "noflag" in modes:
                fo = off + tm[1] + (addr - tm[0])
                struct.pack_into("<II", buf, fo, 0xBF800000, 0xBF800000)
                n_patched[(cur, op)] = n_patched.get((cur, op), 0) + 1
                continue
            if op in ("s_waitcnt", "s_barrier", "s_nop") and not dense(li):
                continue
            if op == "s_waitcnt":
                w = word
                if "lgkm" in modes:
                    w |= 0x0F00
                if "vm" in modes:
                    w |= 0xC00F
                if w != word:
                    new = w
            elif op == "s_barrier" and "barrier" in modes:
                new = 0xBF800000
            elif op == "s_nop" and "nop" in modes and (word & 0xFFFF):
                new = 0xBF800000
            if new is not None:
                fo = off + tm[1] + (addr - tm[0])
                assert struct.unpack_from("<I", buf, fo)[0] == word, (hex(addr), hex(word))
                struct.pack_into("<I", buf, fo, new)
                n_patched[(cur, op)] = n_patched.get((cur, op), 0) + 1
    for k, v in sorted(n_patched.items()):
        print(f"{v:5d} x {k[1]:10s} in {k[0][:80]}")
    open(out, "wb").write(buf)
    os.chmod(out, 0o755)


if __name__ == "__main__":
    main()
