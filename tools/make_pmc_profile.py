"""profiles/<tag>_pmc.json from the three passes of tools/gpu_pmc.sh:
python tools/make_pmc_profile.py gpurun_out/<dir> profiles/<tag>_pmc.json"""
import glob
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
csvs = sorted(glob.glob(src + "/*/*/*_counter_collection.csv"))
d = json.loads(subprocess.check_output([sys.executable, __file__.replace("make_pmc_profile", "pmc_summary")] + csvs))
out = {"source": "tools/gpu_pmc.sh: three rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES "
                 "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE), each over `python3 "
                 "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-f16` (23 steps), summarised by "
                 "tools/pmc_summary.py; per-launch averages",
       "correction": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE and WRITE_SIZE are in KiB and gfx950 tallies a "
                     "16-byte-per-lane read stream at half its bytes (MI355X_MICROARCH.md, HBM).  The factor 2 is right for the kernels "
                     "that read 16 B per lane (convs, wgrad, norm passes); for 4-byte-per-lane readers (pack_batched, adam, l2, sumsq) "
                     "read hbm_bytes_raw = (FETCH_SIZE + WRITE_SIZE) * 1024 instead.",
       "wave_cycle_split": "shares of SQ_WAVE_CYCLES (fourth pass): SQ_WAIT_ANY (parked at s_waitcnt / s_barrier), SQ_WAIT_INST_ANY "
                           "(has an instruction, cannot issue it: pipe or operand not ready), SQ_ACTIVE_INST_ANY (issuing)",
       "mfma_util": "SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)", "lds_conflict_share": "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
       "source_sha": kernel_source_sha(), "kernels": {}}
for k, v in d.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    f, w = v["FETCH_SIZE"]["avg"], v["WRITE_SIZE"]["avg"]
    e = {"launches": v["FETCH_SIZE"]["launches"], "fetch_size_kb_avg": round(f, 1), "write_size_kb_avg": round(w, 1),
         "hbm_bytes_per_launch": int((2 * f + w) * 1024), "hbm_bytes_raw": int((f + w) * 1024)}
    if "SQ_BUSY_CU_CYCLES" in v and v["SQ_BUSY_CU_CYCLES"]["avg"] > 0:
        e["mfma_util"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] / (4 * v["SQ_BUSY_CU_CYCLES"]["avg"]), 4)
        mops = v["SQ_INSTS_VALU_MFMA_MOPS_BF16"]["avg"] + v.get("SQ_INSTS_VALU_MFMA_MOPS_F16", {"avg": 0.0})["avg"]
        e["mfma_gflop_issued_per_launch"] = round(mops * 512 / 1e9, 2)
        if v["SQ_LDS_IDX_ACTIVE"]["avg"] > 0:
            e["lds_conflict_share"] = round(v["SQ_LDS_BANK_CONFLICT"]["avg"] / v["SQ_LDS_IDX_ACTIVE"]["avg"], 4)
    if "SQ_WAVE_CYCLES" in v and "SQ_WAIT_ANY" in v and v["SQ_WAVE_CYCLES"]["avg"] > 0:
        wc = v["SQ_WAVE_CYCLES"]["avg"]
        e["wave_cycle_split"] = {"parked_at_wait_or_barrier": round(v["SQ_WAIT_ANY"]["avg"] / wc, 4),
                                 "issue_stalled": round(v["SQ_WAIT_INST_ANY"]["avg"] / wc, 4),
                                 "issue_stalled_on_lds": round(v["SQ_WAIT_INST_LDS"]["avg"] / wc, 4),
                                 "issuing": round(v["SQ_ACTIVE_INST_ANY"]["avg"] / wc, 4),
                                 "issuing_valu_incl_mfma": round(v["SQ_ACTIVE_INST_VALU"]["avg"] / wc, 4),
                                 "issuing_lds": round(v["SQ_ACTIVE_INST_LDS"]["avg"] / wc, 4),
                                 "issuing_vmem": round(v["SQ_ACTIVE_INST_VMEM"]["avg"] / wc, 4)}
    for c in ("SQ_LDS_CMD_FIFO_FULL", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_ADDR_CONFLICT", "SQ_INST_CYCLES_VMEM_RD",
              "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_BUSY_CYCLES"):
        if c in v:
            e.setdefault("raw", {})[c] = round(v[c]["avg"], 1)
    if k.startswith("at::") or k.startswith("__amd"):
        continue
    out["kernels"][k] = e
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(dst, len(out["kernels"]), "kernels")
