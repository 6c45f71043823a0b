"""Per-kernel register / scratch / LDS figures of a hipcc object's gfx950 code (from the code object's metadata notes).

    python tools/kernel_resources.py build/obj/net_forward_w1d.hip.o [name filter]

A non-zero `.private_segment_fixed_size` or `spill` count in a kernel written against the whole register file means hipcc
spilled: every such register costs a scratch round trip per use (DESIGN.md 4.1 "no register spills").
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_check import LLVM, TARGET  # noqa: E402


def resources(obj_path: str):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj_path])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               f"--targets={TARGET}", f"--output={co}"])
        text = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
    out, cur = [], None
    for line in text.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key == "agpr_count":
            cur = {}
            out.append(cur)
        if cur is not None and key in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                       "private_segment_fixed_size", "group_segment_fixed_size", "name"):
            cur[key] = val
    return out


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in resources(sys.argv[1]):
        if flt in k.get("name", ""):
            print({a: b for a, b in k.items() if a != "name"}, k.get("name", "")[:90])
