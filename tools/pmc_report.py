"""Join rocprofv3 --pmc CSVs (tools/pmc_run.sh) with the plan's op order and print a per-op table."""
import csv
import json
import os
import sys
from collections import defaultdict

# usage: tools/pmc_report.py <tag> [ops.json from SBBSEG_BENCH_OPS] [summary.json to write] [precision] [patches per launch]
tag = sys.argv[1] if len(sys.argv) > 1 else "pmc1"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
ops_json = sys.argv[2] if len(sys.argv) > 2 else None
summary_out = sys.argv[3] if len(sys.argv) > 3 else None
precision = sys.argv[4] if len(sys.argv) > 4 else "f16"
patches_per_launch = float(sys.argv[5]) if len(sys.argv) > 5 else 140.0
if ops_json:                                                  # (round 5: bench.py records the launch size of its profiling pass)
    _ppl = [o.get("patches_per_launch") for o in json.load(open(ops_json)) if o.get("patches_per_launch")]
    if _ppl:
        patches_per_launch = float(max(_ppl))


def csrc_sha():
    """hash of the kernel / host sources the profiled library was built from (bench.py drops `traffic` when it differs)"""
    import hashlib
    h = hashlib.sha1()
    base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sbb_textline_detection_amd", "csrc")
    for name in ("kernels.hip", "block_x3.hip", "stem_pool_x3.hip", "dec_halo_x3.hip", "dec_halo_f16.hip", "expand_reduce_x3.hip", "conv3_expand_reduce.hip", "region.hip", "region.h",
                 "api.hip", "internal.h"):
        with open(os.path.join(base, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def load(name):
    rows = defaultdict(dict)
    meta = {}
    with open(os.path.join(root, f"{tag}_{name}", "pmc_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            d = int(r["Dispatch_Id"])
            rows[d][r["Counter_Name"]] = float(r["Counter_Value"])
            meta[d] = (r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]), int(r["VGPR_Count"]),
                       int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return rows, meta


sq, meta = load("sq")
fe, meta_f = load("fetch")
wr, meta_w = load("write")
# plan kernels only (conv / maxpool / head), in dispatch order; one chunk = 61 ops
def plan_ids(meta):
    return [d for d in sorted(meta) if any(s in meta[d][0] for s in ("conv_igemm", "maxpool", "head_kernel", "dec_tail", "stem_conv", "stem_pool", "dec_halo", "expand_reduce", "conv3x3_c64_direct", "bottleneck_fused", "block_x3"))]
ids, idf, idw = plan_ids(meta), plan_ids(meta_f), plan_ids(meta_w)
names = None
if ops_json:
    # (an op that launches nothing -- the max-pool the split mode's stem computes itself -- has no dispatch to join with)
    names = [o["name"] for o in json.load(open(ops_json)) if o.get("ms_per_launch", 1.0) >= 0.02]
n_ops = len(names) if names else 61
# take the LAST full chunk (steady state)
ids, idf, idw = ids[-n_ops:], idf[-n_ops:], idw[-n_ops:]
summary = {}
print(f"{'op':46s} {'us':>7s} {'grid':>6s} {'MHz':>5s} {'mfma%':>6s} {'wait%':>6s} {'instwait%':>9s} {'active%':>8s} {'ldsconf%':>8s} {'fetchMB':>8s} {'writeMB':>8s} {'HBM GB/s':>8s} {'L2hit%':>6s}")
for k, (d, df, dw) in enumerate(zip(ids, idf, idw)):
    c = sq[d]; nm = names[k] if names else meta[d][0][:40]
    us = meta[d][4] / 1e3
    wc = c.get("SQ_WAVE_CYCLES", 1) or 1
    busy = c.get("SQ_BUSY_CYCLES", 1) or 1
    # SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles (16 per 16x16x32 MFMA).  In this rocprofv3 build the SQ
    # totals cover one XCD's worth of SIMDs (1/8 of the chip): checked against the issued-FLOP rate of the decoder
    # launches (1 000 TFLOP/s of 2 500 = 40 %; raw ratio 5.1 %).  Normalise by GPU-active cycles x 32 CUs x 4 SIMDs.
    gui = fe[df].get("GRBM_GUI_ACTIVE", 0) or 1
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 32 * 4) * 100
    fetch = fe[df].get("FETCH_SIZE", 0) * 1024 * 2 / 1e6        # gfx950: FETCH_SIZE reads half (MI355X_MICROARCH.md)
    write = wr[dw].get("WRITE_SIZE", 0) * 1024 / 1e6
    hit = wr[dw].get("TCC_HIT_sum", 0); miss = wr[dw].get("TCC_MISS_sum", 0)
    us_f = meta_f[df][4] / 1e3
    mhz = gui / 8 / us_f                  # GRBM_GUI_ACTIVE = busy shader-clock cycles summed over the 8 XCDs: the launch's average clock
                                          # (short launches read high: the counter also covers the dispatch's lead-in / drain)
    gbps = (fetch + write) / us * 1e3     # MB / us = TB/s -> GB/s
    print(f"{nm:46s} {us:7.1f} {meta[d][1]//meta[d][2]:6d} {mhz:5.0f} {mfma:6.1f} {c.get('SQ_WAIT_ANY',0)/wc*100:6.1f} {c.get('SQ_WAIT_INST_ANY',0)/wc*100:9.1f} "
          f"{c.get('SQ_ACTIVE_INST_ANY',0)/wc*100:8.1f} {c.get('SQ_LDS_BANK_CONFLICT',0)/busy*100:8.2f} {fetch:8.1f} {write:8.1f} {gbps:8.0f} {hit/(hit+miss+1e-9)*100:6.1f}")
    summary[nm] = {"us": round(us, 1), "shader_clock_mhz": round(mhz), "mfma_busy_pct": round(mfma, 1), "fetch_bytes": round(fetch * 1e6), "write_bytes": round(write * 1e6),
                   "hbm_gbps": round(gbps), "l2_hit_pct": round(hit / (hit + miss + 1e-9) * 100, 1), "lds_conflict_pct": round(c.get('SQ_LDS_BANK_CONFLICT', 0) / busy * 100, 2)}
if summary_out:
    json.dump({"source": "rocprofv3 --pmc passes over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-mode` (tools/pmc_run.sh: SQ_*, "
                         "FETCH_SIZE+GRBM_GUI_ACTIVE, WRITE_SIZE+TCC_HIT/MISS in separate runs); last launch of every op = the bench's roofline pass "
                         "(one whole chunk per launch on one lane); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts half of wide coalesced reads); "
                         "duplicate op names keep the last launch; mfma_busy_pct = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 32 CUs x 4 SIMDs) "
                         "(the SQ totals of this rocprofv3 build cover one XCD); shader_clock_mhz = GRBM_GUI_ACTIVE / 8 XCDs / launch duration",
               "precision": precision, "patches_per_launch": patches_per_launch, "csrc_sha": csrc_sha(), "ops": summary}, open(summary_out, "w"), indent=1)
