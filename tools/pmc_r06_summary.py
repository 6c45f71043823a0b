#!/usr/bin/env python3
"""Arithmetic over the rocprofv3 passes of tools/pmc_r06.sh -> JSON summaries stamped with the csrc digest
(tamago_amd.build.source_digest) so that bench.py only quotes counters measured on the kernels it runs.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KB; on
gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled; TCC_* requests are
128-byte lines."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def counters(out, passes, match):
    """{kernel short name: {counter: per-launch average}}, {kernel: (launches, avg ns)} over the given passes."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in passes:
        path = f"{out}/{d}/p_counter_collection.csv"
        if not os.path.exists(path):
            cands = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)
            if not cands:
                continue
            path = cands[0]
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"]
                if any(m in name for m in match):
                    agg[short(name)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        tr = path.replace("counter_collection", "kernel_trace")
        if os.path.exists(tr):
            with open(tr) as f:
                for r in csv.DictReader(f):
                    if any(m in r["Kernel_Name"] for m in match):
                        dur[(d, short(r["Kernel_Name"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    per = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
    times = {}
    for (d, k), v in dur.items():
        times.setdefault(k, []).extend(v)
    return per, {k: (len(v) // max(1, len(passes)), sum(v) / len(v)) for k, v in times.items()}


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0].strip()


def forward_summary(out, prefix, match, positions, size, digest, min_ns=1e5):
    """summaries of the forward kernels matching `match` in the passes <prefix>_{a..g}"""
    per, times = counters(out, [f"{prefix}_{x}" for x in "abgcdef"], match)
    res = {}
    for kname, c in per.items():
        if "GRBM_GUI_ACTIVE" not in c or times[kname][1] < min_ns:      # (the guarded fp32 fallback launch exits in microseconds)
            continue
        launches, dur_ns = times[kname]
        simd_cycles = 256 * 4 * c["GRBM_GUI_ACTIVE"] / 8          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
        p = size * size
        d = {"shader_clock_GHz": c["GRBM_GUI_ACTIVE"] / 8 / dur_ns,
             "mfma_busy_fraction_of_simd_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
             "hbm_read_bytes": 2 * c.get("FETCH_SIZE", 0) * 1024, "hbm_write_bytes": c.get("WRITE_SIZE", 0) * 1024,
             "hbm_bytes_per_position": hbm / positions, "hbm_GBps": hbm / dur_ns,
             "algorithmic_io_bytes_per_position": 6 * p * 4 + (p + 1) * 4 + 3 * 4,
             "l2_request_bytes": c.get("TCC_REQ_sum", 0) * 128,
             "l2_request_bytes_per_position": c.get("TCC_REQ_sum", 0) * 128 / positions,
             "l2_read_TBps": c.get("TCC_READ_sum", 0) * 128 / dur_ns / 1e3,
             "l2_hit_rate": c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_REQ_sum", 0)),
             "tcc_busy_fraction": c.get("TCC_BUSY_avr", 0) / max(1.0, c["GRBM_GUI_ACTIVE"] / 8)}
        if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_ANY" in c:
            d["wave_cycles_waiting_waitcnt_or_barrier"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
            d["wave_cycles_issue_stalled"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
            d["lds_bank_conflict_share_of_lds_active"] = c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"])
        if "SQ_INSTS_VALU" in c:
            d["instructions_per_launch"] = {k: c[k] for k in c if k.startswith("SQ_INSTS_")}
            d["valu_per_mfma"] = c["SQ_INSTS_VALU"] / max(1.0, c.get("SQ_INSTS_MFMA", 0))
        res[kname] = {"kernel": kname, "csrc_digest": digest, "positions_per_launch": positions, "launches_averaged": launches,
                      "avg_duration_ms_profiled": dur_ns / 1e6, "counters_per_launch": c, "derived": d}
    return res


def tag_of(kname):
    for t in ("w1dband", "w1d", "band", "split", "wino", "heads19"):
        if t in kname:
            return t
    return "direct"


def main(out):
    from tamago_amd.build import FORWARD_SOURCES, source_digest
    digest = source_digest(FORWARD_SOURCES)          # forward summaries: the forward kernels' sources only
    digest_all = source_digest()
    for prefix, match, positions, size in (("fwd", ["dualnet_fwd"], 65536, 9), ("wn", ["dualnet_fwd_wino8"], 65536, 9),
                                           ("f19", ["dualnet_fwd", "dualnet_heads19"], 4096, 19), ("b19", ["dualnet_fwd", "dualnet_heads19"], 64, 19)):
        for kname, summary in forward_summary(out, prefix, match, positions, size, digest, 1e5 if size == 9 else 2e4).items():
            path = f"{out}/r06_pmc_forward_{tag_of(kname)}_{size}x{size}_b{positions}.json"
            with open(path, "w") as f:
                json.dump(summary, f, indent=1)
            print(path, json.dumps(summary["derived"], indent=1))
    # ---- 2. tree kernels + 3. featurise -------------------------------------------------------------
    leaves = 3 * 2048 * 1001                              # bench.py --steps 2 --warmup 1 --trees 2048: leaf evaluations
    tree = {"csrc_digest": digest_all, "run": "bench.py --steps 2 --warmup 1 --trees 2048 (3 move searches x 2048 trees x 1001 leaf evaluations)",
            "leaf_evals_in_run": leaves, "kernels": {},
            "algorithmic_bytes_per_leaf_SURVEY_8d": {
                "select (PUCB walk 24 B x children x ~3.5 levels + node init 38 B x A + planes 1944 B)": 3.5 * 82 * 24 + 38 * 82 + 1944,
                "backup (policy A x 4 B read + n x 8 B written + 64 B RMW x ~3.5 path edges)": 82 * 4 + 82 * 8 + 64 * 3.5,
                "featurise (P B in + 6 P fp32 out)": 2025}}
    per, times = counters(out, ["tree_c", "tree_d", "tree_e"], ["select_", "backup_kernel", "root_kernel", "play_kernel"])
    for kname, c in per.items():
        launches, dur_ns = times[kname]
        hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
        tree["kernels"][kname] = {
            "launches_in_run": launches, "avg_duration_us": dur_ns / 1e3,
            "hbm_bytes_per_launch": hbm, "hbm_GBps": hbm / dur_ns,
            "hbm_bytes_per_leaf_eval": hbm * launches / leaves,
            "l2_request_bytes_per_launch": c.get("TCC_REQ_sum", 0) * 128,
            "l2_request_GBps": c.get("TCC_REQ_sum", 0) * 128 / dur_ns,
            "l2_request_bytes_per_leaf_eval": c.get("TCC_REQ_sum", 0) * 128 * launches / leaves,
            "l2_hit_rate": c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_REQ_sum", 0)),
            "fraction_of_8TBps_HBM_peak": hbm / dur_ns / 8000.0}
    per, times = counters(out, ["feat_c", "feat_d"], ["featurize_kernel"])
    for kname, c in per.items():
        launches, dur_ns = times[kname]
        hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
        b = (1 << 21) if "<9>" in kname else (1 << 19)
        p = 81 if "<9>" in kname else 361
        tree["kernels"][kname] = {
            "launches_in_run": launches, "avg_duration_us": dur_ns / 1e3, "positions_per_launch": b,
            "hbm_bytes_per_launch": hbm, "hbm_GBps": hbm / dur_ns, "hbm_bytes_per_position": hbm / b,
            "algorithmic_bytes_per_position": p + 9 + 24 * p, "algorithmic_GBps": b * (p + 9 + 24 * p) / dur_ns,
            "fraction_of_8TBps_HBM_peak": hbm / dur_ns / 8000.0}
    path = f"{out}/r06_pmc_tree_and_featurize_kernels.json"
    with open(path, "w") as f:
        json.dump(tree, f, indent=1)
    print(path)
    for k, v in tree["kernels"].items():
        print(f"  {k:40s} {v['avg_duration_us']:10.1f} us  HBM {v['hbm_GBps']:8.1f} GB/s ({100 * v['fraction_of_8TBps_HBM_peak']:.1f} % of 8 TB/s)")
    # ---- 4. kernel stats of the bench ---------------------------------------------------------------------
    for cand in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
        dst = f"{out}/r06_bench_trees2048_kernel_stats.csv"
        with open(cand) as f, open(dst, "w") as g:
            g.write(f.read())
        print(dst)
        break


if __name__ == "__main__":
    main(sys.argv[1])
