#!/usr/bin/env python3
"""Arithmetic over the rocprofv3 passes of tools/pmc_r04.sh -> JSON summaries stamped with the csrc digest
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


def main(out):
    from tamago_amd.build import FORWARD_SOURCES, source_digest
    digest = source_digest(FORWARD_SOURCES)          # forward summaries: the forward kernels' sources only
    digest_all = source_digest()
    # ---- 1. forward kernel ----------------------------------------------------------------------
    per, times = counters(out, ["fwd_a", "fwd_b", "fwd_c", "fwd_d", "fwd_e", "fwd_f", "fwd_g"], ["dualnet_fwd"])
    per2, times2 = counters(out, ["w2_a", "w2_b", "w2_c", "w2_d", "w2_e", "w2_f"], ["dualnet_fwd_split_kernel<9"])
    per.update(per2)
    times.update(times2)
    per4, times4 = counters(out, ["ws_a", "ws_b", "ws_c", "ws_d", "ws_e", "ws_f", "ws_g"], ["dualnet_fwd_wsplit"])
    per.update(per4)
    times.update(times4)
    per3, times3 = counters(out, ["wn_a", "wn_b", "wn_c", "wn_d", "wn_e", "wn_f"], ["dualnet_fwd_wino8"])
    per.update(per3)
    times.update(times3)
    per19, times19 = counters(out, ["f19_a", "f19_c", "f19_d", "f19_e"], ["dualnet_fwd_split_kernel<19"])
    for kname, c in per19.items():
        if "GRBM_GUI_ACTIVE" not in c:
            continue
        launches, dur_ns = times19[kname]
        positions = 4096
        hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
        summary = {"kernel": kname, "csrc_digest": digest, "positions_per_launch": positions, "avg_duration_ms_profiled": dur_ns / 1e6,
                   "counters_per_launch": c,
                   "derived": {"shader_clock_GHz": c["GRBM_GUI_ACTIVE"] / 8 / dur_ns,
                               "mfma_busy_fraction_of_simd_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (256 * 4 * c["GRBM_GUI_ACTIVE"] / 8),
                               "hbm_bytes_per_position": hbm / positions, "algorithmic_io_bytes_per_position": 6 * 361 * 4 + 362 * 4 + 12,
                               "l2_request_bytes_per_position": c.get("TCC_REQ_sum", 0) * 128 / positions,
                               "l2_hit_rate": c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_REQ_sum", 0))}}
        path = f"{out}/r04_pmc_forward_split_19x19_b4096.json"
        with open(path, "w") as f:
            json.dump(summary, f, indent=1)
        print(path, json.dumps(summary["derived"], indent=1))
    # the banded 19x19 kernel (a board over four workgroups) on one tree's mini-batch of 64 positions
    perb, timesb = counters(out, ["b19_a", "b19_c", "b19_d", "b19_e"], ["dualnet_fwd_band_kernel"])
    for kname, c in perb.items():
        if "GRBM_GUI_ACTIVE" not in c:
            continue
        launches, dur_ns = timesb[kname]
        positions = 64
        hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
        summary = {"kernel": kname, "csrc_digest": digest, "positions_per_launch": positions, "workgroups": 256,
                   "avg_duration_us_profiled": dur_ns / 1e3, "counters_per_launch": c,
                   "derived": {"shader_clock_GHz": c["GRBM_GUI_ACTIVE"] / 8 / dur_ns,
                               "mfma_busy_fraction_of_simd_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (256 * 4 * c["GRBM_GUI_ACTIVE"] / 8),
                               "hbm_bytes_per_position": hbm / positions, "algorithmic_io_bytes_per_position": 6 * 361 * 4 + 362 * 4 + 12,
                               "l2_request_bytes_per_position": c.get("TCC_REQ_sum", 0) * 128 / positions,
                               "l2_hit_rate": c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_REQ_sum", 0))}}
        path = f"{out}/r04_pmc_forward_band_19x19_b64.json"
        with open(path, "w") as f:
            json.dump(summary, f, indent=1)
        print(path, json.dumps(summary["derived"], indent=1))
    for kname, c in per.items():
        if "GRBM_GUI_ACTIVE" not in c or times[kname][1] < 1e5:      # (the guarded fp32 fallback launch exits in microseconds)
            continue
        positions = 65536
        launches, dur_ns = times[kname]
        simd_cycles = 256 * 4 * c["GRBM_GUI_ACTIVE"] / 8          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
        summary = {
            "kernel": kname, "csrc_digest": digest, "positions_per_launch": positions,
            "launches_averaged": launches, "avg_duration_ms_profiled": dur_ns / 1e6,
            "counters_per_launch": c,
            "derived": {
                "shader_clock_GHz": c["GRBM_GUI_ACTIVE"] / 8 / dur_ns,
                "mfma_busy_fraction_of_simd_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
                "wave_cycles_waiting_waitcnt_or_barrier": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                "wave_cycles_issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                "lds_bank_conflict_share_of_lds_active": c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]),
                "hbm_read_bytes": 2 * c.get("FETCH_SIZE", 0) * 1024, "hbm_write_bytes": c.get("WRITE_SIZE", 0) * 1024,
                "hbm_bytes_per_position": hbm / positions,
                "hbm_GBps": hbm / dur_ns,
                "algorithmic_io_bytes_per_position": 6 * 81 * 4 + 82 * 4 + 3 * 4,
                "l2_request_bytes": c.get("TCC_REQ_sum", 0) * 128,
                "l2_request_bytes_per_position": c.get("TCC_REQ_sum", 0) * 128 / positions,
                "l2_read_TBps": c.get("TCC_READ_sum", 0) * 128 / dur_ns / 1e3,
                "l2_hit_rate": c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_REQ_sum", 0)),
                "tcc_busy_fraction": c.get("TCC_BUSY_avr", 0) / max(1.0, c["GRBM_GUI_ACTIVE"] / 8),
            },
        }
        tag = "w1d" if "w1d" in kname else "wsplit" if "wsplit" in kname else ("w2" if "w2" in kname else ("split" if "split" in kname else ("wino" if "wino" in kname else "direct")))
        if "SQ_INSTS_VALU" in c:      # instruction mix per wave: the kernel is issue-bound
            waves = 4 * 256 * (positions // 3 // 256 + 1) * 0 + 1
            summary["derived"]["instructions_per_launch"] = {k: c[k] for k in c if k.startswith("SQ_INSTS_")}
            summary["derived"]["valu_per_mfma"] = c["SQ_INSTS_VALU"] / max(1.0, c.get("SQ_INSTS_MFMA", 0))
        path = f"{out}/r04_pmc_forward_{tag}_9x9_b65536.json"
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
    path = f"{out}/r04_pmc_tree_and_featurize_kernels.json"
    with open(path, "w") as f:
        json.dump(tree, f, indent=1)
    print(path)
    for k, v in tree["kernels"].items():
        print(f"  {k:40s} {v['avg_duration_us']:10.1f} us  HBM {v['hbm_GBps']:8.1f} GB/s ({100 * v['fraction_of_8TBps_HBM_peak']:.1f} % of 8 TB/s)")
    # ---- 4. kernel stats of the bench ---------------------------------------------------------------------
    for cand in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
        dst = f"{out}/r04_bench_trees2048_kernel_stats.csv"
        with open(cand) as f, open(dst, "w") as g:
            g.write(f.read())
        print(dst)
        break


if __name__ == "__main__":
    main(sys.argv[1])
