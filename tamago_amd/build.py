"""In-tree build of libtamago_hip.so (hipcc, gfx950 only).

``python -m tamago_amd.build`` or ``tamago_amd.build.build()``.  Objects are cached under
``build/`` by source mtime; the shared library lands next to this file so that it ships
with the repository snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtamago_hip.so")
OBJ_DIR = os.path.join(REPO, "build", "obj")

ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


# per-source flags.  net_forward_w1d.hip / net_forward_w1dband.hip: MFMA results in VGPRs (the accumulation half of the register file holds the
# layer's weight fragments), and no SLP vectorisation (v_pk_add_f32 / v_pk_fma_f32 beside an MFMA stream cost ~12 cycles
# each against ~3.4 for two plain instructions: profiles/r04_microbench_wino_issue_model.txt)
_W1D_FLAGS = ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
EXTRA_FLAGS = {"net_forward_w1d.hip": _W1D_FLAGS, "net_forward_w1dband.hip": _W1D_FLAGS}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found - the HIP extension cannot be built")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)
                  if f.endswith(".hip") or f.endswith(".cpp"))


FORWARD_SOURCES = tuple(f for f in ("net_forward.hip", "net_forward_split.hip", "net_forward_band.hip", "net_forward_w1d.hip",
                                     "net_forward_w1dband.hip", "w1d_common.h", "split_common.h", "net_device.h", "common.h")
                        if os.path.exists(os.path.join(CSRC, f)))


def source_digest(names=None) -> str:
    """sha256 (16 hex digits) over the code lines (// comments and blank lines dropped) of kernel sources: profiles/ summaries carry it, bench.py only quotes a PMC
    summary that was measured on the kernels it is running.  `names`: basenames under csrc/ (default: all
    sources + headers + the ABI header); FORWARD_SOURCES = what the DualNet forward kernels are built from."""
    import hashlib
    h = hashlib.sha256()
    if names is None:
        files = sources() + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
        files.append(os.path.join(REPO, "include", "tamago_hip.h"))
    else:
        files = [os.path.join(CSRC, f) for f in names]
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "r") as f:
            for line in f:
                code = line.split("//", 1)[0].strip()     # comment edits do not invalidate a measurement
                if code:
                    h.update(code.encode() + b"\n")
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(REPO, "include", "tamago_hip.h"))
    newest_header = max(os.path.getmtime(h) for h in headers)
    objs = []
    jobs = []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or \
            os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header)
        if stale:
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-x", "hip", "-c", src, "-o", obj])
    rebuilt = bool(jobs)
    if jobs:
        # the translation units are independent: compile them side by side (search.hip alone takes over a minute)
        from concurrent.futures import ThreadPoolExecutor, as_completed

        def run(cmd):
            name = os.path.basename(cmd[-3])
            if verbose:
                print("[tamago_amd.build]", " ".join(cmd), flush=True)
            # (output collected per source and printed under its name: the jobs run side by side)
            res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if res.stdout.strip():
                print("".join(f"[{name}] {line}\n" for line in res.stdout.splitlines()), end="", flush=True)
            if res.returncode != 0:
                raise subprocess.CalledProcessError(res.returncode, cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2), 8)) as pool:
            futures = [pool.submit(run, cmd) for cmd in jobs]
            try:
                for fut in as_completed(futures):
                    fut.result()
            except BaseException:
                for fut in futures:                  # the first failure ends the build: jobs not yet started are dropped
                    fut.cancel()
                raise
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print("[tamago_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
