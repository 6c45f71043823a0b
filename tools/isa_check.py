"""Build-time check of the EMITTED gfx950 code of the forward kernels that hand-place their waits.

The weight fragments of `dualnet_fwd_w1d_kernel` / `dualnet_fwd_w1dband_kernel` are requested by inline-asm
`global_load_dwordx4` into AGPRs and guarded by hand-written `s_waitcnt vmcnt(N)`.  An MFMA is not a memory operation, so
nothing but a `sched_barrier` keeps hipcc from hoisting it above the wait that guards its operands (that
happened once and showed up as one self-play game in 44 that was off by one move).  This checker disassembles the device
code (`llvm-objdump -d`) and replays every kernel's instruction stream:

* a load marks its destination registers "in flight"; every later vector-memory instruction ages them by one
  (gfx9: loads return in order, `vmcnt` counts them);
* `s_waitcnt vmcnt(N)` retires every load that has at least N newer ones behind it;
* a `v_mfma` (or any VALU instruction) that READS a register still in flight is a violation.

The replay is a forward data-flow analysis over the kernel's control-flow graph (basic blocks from the branch targets
objdump resolves; at a join a register keeps the younger of its possible loads), iterated to its fixed point: a fragment
requested in one layer instantiation and used in the next crosses the loop's back edge, cold blocks that hipcc lays out
behind the loop are followed where they are entered.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

_REG = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+)\b)")
_VMEM = re.compile(r"^(global|buffer|scratch|flat)_(load|store|atomic)")


def disassemble(obj_path: str) -> str:
    """Device-side disassembly of a hipcc object (fat binary -> gfx950 code object -> text)."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        co = os.path.join(tmp, "dev.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj_path])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               f"--targets={TARGET}", f"--output={co}"])
        return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], text=True)


def kernels(text: str):
    """-> {mangled name: [(address, mnemonic, operand string, branch target address or None), ...]}"""
    out, cur, base = {}, None, 0
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <([^>]+)>:", line)
        if m:
            cur = out.setdefault(m.group(2), [])
            base = int(m.group(1), 16)
            continue
        if cur is None or "//" not in line:
            continue
        body, tail = line.split("//", 1)
        body = body.strip()
        if not body:
            continue
        am = re.match(r"\s*([0-9A-Fa-f]+):", tail)
        if not am:
            continue
        parts = body.split(None, 1)
        target = None
        if parts[0].startswith("s_cbranch") or parts[0] == "s_branch":
            tm = re.search(r"<[^>+]+(?:\+0x([0-9a-f]+))?>", tail)     # objdump resolves the target as <symbol+0xOFF>
            if tm:
                target = base + (int(tm.group(1), 16) if tm.group(1) else 0)
        cur.append((int(am.group(1), 16), parts[0], parts[1] if len(parts) > 1 else "", target))
    return out


def _regs(tok: str):
    """register names in one operand token: 'a[0:3]' -> ['a0', ..], 'v5' -> ['v5']"""
    res = []
    for m in _REG.finditer(tok):
        if m.group(4) is not None:
            res.append(f"{m.group(1)}{m.group(4)}")
        else:
            res.extend(f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1))
    return res


def _split_ops(ops: str):
    return [t.strip() for t in ops.split(",")] if ops else []


_CAP = 64                    # vmcnt is a 6-bit counter on gfx9: a load with 64 newer ones behind it has been waited for or never will be


def check_kernel(insts, max_report=8):
    """Forward data-flow over the kernel's control-flow graph.  State: register -> number of vector-memory instructions
    issued AFTER the load that is still to write it (joins keep the smaller number: the load that a given vmcnt(N) is
    less likely to have retired).  -> (violations, stats); a violation = (address, instruction text, register)."""
    n = len(insts)
    index = {a: i for i, (a, _, _, _) in enumerate(insts)}
    leaders = {0}
    for i, (_, mn, _, tgt) in enumerate(insts):
        if tgt is not None or mn == "s_endpgm":
            if i + 1 < n:
                leaders.add(i + 1)
            if tgt is not None and tgt in index:
                leaders.add(index[tgt])
    order = sorted(leaders)
    block_end = {b: (order[k + 1] if k + 1 < len(order) else n) for k, b in enumerate(order)}

    def successors(b):
        last = block_end[b] - 1
        _, mn, _, tgt = insts[last]
        succ = []
        if mn == "s_endpgm":
            return succ
        if tgt is not None and tgt in index:
            succ.append(index[tgt])
        if mn != "s_branch" and last + 1 < n:
            succ.append(last + 1)
        return succ

    stats = {"vmem": 0, "mfma": 0, "waits": 0, "agpr_loads": 0, "mfma_on_agpr": 0, "blocks": len(order)}
    for _, mn, ops, _ in insts:
        toks = _split_ops(ops)
        if _VMEM.match(mn):
            stats["vmem"] += 1
            if "_load" in mn and toks and toks[0].startswith("a"):
                stats["agpr_loads"] += 1
        elif mn == "s_waitcnt" and "vmcnt" in ops:
            stats["waits"] += 1
        elif mn.startswith("v_mfma"):
            stats["mfma"] += 1
            if any(t.startswith("a") for t in toks[1:]):
                stats["mfma_on_agpr"] += 1

    violations = {}

    def transfer(b, state, report):
        state = dict(state)
        for i in range(b, block_end[b]):
            addr, mn, ops, _ = insts[i]
            toks = _split_ops(ops)
            if _VMEM.match(mn):
                for r in state:
                    if state[r] < _CAP:
                        state[r] += 1
                if "_load" in mn and toks:
                    for r in _regs(toks[0]):
                        state[r] = 0
                continue
            if mn == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ops)
                if m:
                    keep = int(m.group(1))
                    state = {r: c for r, c in state.items() if c < keep}
                continue
            if mn.startswith("s_") or mn.startswith("ds_") or not toks:
                # (LDS stores of loaded data are compiler-tracked; scalar code does not read vector registers)
                continue
            if report:
                for t in toks[1:]:
                    for r in _regs(t):
                        if r in state:
                            violations.setdefault(addr, (addr, f"{mn} {ops}", r))
            for r in _regs(toks[0]):                       # a redefinition ends the dependence on the load
                state.pop(r, None)
        return state

    in_state = {0: {}}
    work = [0]
    while work:
        b = work.pop()
        out = transfer(b, in_state[b], False)
        for s in successors(b):
            cur = in_state.get(s)
            if cur is None:
                in_state[s] = dict(out)
                work.append(s)
            else:
                changed = False
                for r, c in out.items():
                    if r not in cur or c < cur[r]:
                        cur[r] = c
                        changed = True
                if changed:
                    work.append(s)
    for b, st in in_state.items():
        transfer(b, st, True)
    viol = [violations[a] for a in sorted(violations)]
    return (viol[:max_report] if max_report else viol), stats


def check_object(obj_path: str, name_filter: str):
    """Check every kernel of an object whose (mangled) name contains `name_filter`."""
    result = {}
    for name, insts in kernels(disassemble(obj_path)).items():
        if name_filter in name:
            result[name] = check_kernel(insts)
    return result


if __name__ == "__main__":
    obj, flt = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
    bad = 0
    for name, (viol, stats) in check_object(obj, flt).items():
        print(name, stats, "violations:", len(viol))
        for v in viol:
            print("   0x%x  %s   reads %s while its load may still be in flight" % v)
        bad += len(viol)
    sys.exit(1 if bad else 0)
