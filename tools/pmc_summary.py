#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_fwd.sh for the fused forward kernel
into profiles/<name>.json (per launch averages, derived MFMA utilisation and HBM traffic).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are
in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled."""
import collections
import csv
import json
import sys


def main(pmc_dir, out_path, positions):
    agg = collections.defaultdict(list)
    dur = []
    kname = None
    for grp in "abcd":
        with open(f"{pmc_dir}/{grp}/p_counter_collection.csv") as f:
            for r in csv.DictReader(f):
                if "dualnet_fwd" in r["Kernel_Name"]:
                    kname = r["Kernel_Name"]
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(f"{pmc_dir}/{grp}/p_kernel_trace.csv") as f:
            for r in csv.DictReader(f):
                if "dualnet_fwd" in r["Kernel_Name"]:
                    dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    c = {k: sum(v) / len(v) for k, v in agg.items()}
    dur_ns = sum(dur) / len(dur)
    clock_ghz = c["GRBM_GUI_ACTIVE"] / 8 / dur_ns          # summed over the 8 XCDs
    simd_cycles = 256 * 4 * c["GRBM_GUI_ACTIVE"] / 8
    out = {
        "kernel": kname,
        "positions_per_launch": positions,
        "launches_averaged": len(dur) // 4,
        "avg_duration_ms_profiled": dur_ns / 1e6,
        "counters_per_launch": c,
        "derived": {
            "shader_clock_GHz": clock_ghz,
            "mfma_busy_fraction_of_simd_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
            "wave_cycles_waiting_waitcnt_or_barrier": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
            "wave_cycles_issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
            "lds_bank_conflict_share_of_lds_active": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
            "hbm_read_bytes": 2 * c["FETCH_SIZE"] * 1024,
            "hbm_write_bytes": c["WRITE_SIZE"] * 1024,
            "hbm_bytes_per_position": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / positions,
            "algorithmic_io_bytes_per_position": 6 * 81 * 4 + 82 * 4 + 3 * 4,
        },
    }
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["derived"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
