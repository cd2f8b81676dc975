#!/usr/bin/env python3
"""bench.py -- segmented patches/sec (448x448x3) on MI355X for the do_prediction hot path.

A *step* is one pass of the fused hot path over one synthetic page resident in HBM:
u8 page -> LUT normalise + tiling -> ResNet-50-U-Net forward (HIP, MFMA) -> softmax/argmax ->
margin-crop stitch -> u8 label map in HBM.  Workload (BASELINE.json configs[1]): one 3500x2500 page,
textline model (2 classes), margin 0.1 -> 70 tiles of 448x448 per step and per GPU.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank segments its own page
(weak scaling) and the per-rank label maps are exchanged with one RCCL all-gather per step
(the "stitch" exchange north_star names).  value = tiles of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.  Extra objects: `roofline` (dominant conv kernel, HIP-event timed
per launch on the library's stream) and `cpu_baseline` (oracle port on the host cores, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0      # dense bf16/f16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PAGE_H, PAGE_W = 3500, 2500    # BASELINE.json configs[1]
MODEL_HW, CLASSES = 448, 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default=os.environ.get("SBBSEG_PRECISION", "f16"), choices=["f16", "bf16", "f16x3"])
    ap.add_argument("--max-batch", type=int, default=int(os.environ.get("SBBSEG_MAX_BATCH", "0")),
                    help="tiles per chunk (0 = one page per chunk: 70 for the 3500x2500 page, 108 for the 4000x3000 pages of batch64)")
    ap.add_argument("--conv-variant", type=int, default=int(os.environ.get("SBBSEG_CONV_VARIANT", "0")),
                    help="0 auto, 1 force 4-wave/2-stage conv tiles, 2 force 8-wave/3-stage (A/B only)")
    ap.add_argument("--workload", default="page", choices=["page", "pipeline3", "batch64"],
                    help="page = BASELINE configs[1] (default, the metric's config); pipeline3 = configs[2] (border whole-image "
                         "+ layout + textline on one page); batch64 = configs[3] (64 pages of 4000x3000 sharded over the ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-patches", type=int, default=8)
    args = ap.parse_args()
    if args.max_batch <= 0:
        args.max_batch = 108 if args.workload == "batch64" else 70

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from sbb_textline_detection_amd.model import SegModel
    from sbb_textline_detection_amd.synthetic import synthetic_page
    from tools.synth_model import calibrated_model

    cfg, weights = calibrated_model(CLASSES, MODEL_HW, MODEL_HW, seed=0)
    model = SegModel(cfg, weights, device=local_rank, max_batch=args.max_batch, precision=args.precision)
    ctx = model.ctx
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_conv_variant(args.conv_variant)

    page = synthetic_page(PAGE_H, PAGE_W, seed=rank)
    d_page = torch.from_numpy(page).cuda()
    d_labels = torch.empty((PAGE_H, PAGE_W), dtype=torch.uint8, device="cuda")
    d_all = torch.empty((world, PAGE_H, PAGE_W), dtype=torch.uint8, device="cuda") if world > 1 else None
    from sbb_textline_detection_amd import _capi
    tiles_per_page = _capi.tile_grid(PAGE_H, PAGE_W, MODEL_HW, MODEL_HW)[0].shape[0]

    scaling = "weak"
    workload_desc = (f"one {PAGE_H}x{PAGE_W} page per GPU per step, textline model (ResNet-50-U-Net, {CLASSES} classes, "
                     f"seeded synthetic weights), margin 0.1 -> {tiles_per_page} tiles of 448x448")
    tiles_per_step = tiles_per_page * world

    def step():
        ctx.segment_page_dev(d_page.data_ptr(), PAGE_H, PAGE_W, d_labels.data_ptr())
        if world > 1:
            dist.all_gather_into_tensor(d_all.view(-1), d_labels.view(-1))

    if args.workload == "pipeline3":
        # configs[2]: the three stage models on one page (model load excluded, models stay resident)
        cfg_b, w_b = calibrated_model(2, MODEL_HW, MODEL_HW, seed=11)
        cfg_l, w_l = calibrated_model(4, MODEL_HW, MODEL_HW, seed=12)
        m_border = SegModel(cfg_b, w_b, device=local_rank, max_batch=1, precision=args.precision)
        m_layout = SegModel(cfg_l, w_l, device=local_rank, max_batch=args.max_batch, precision=args.precision)
        for m in (m_border, m_layout):
            m.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        d_thr = torch.zeros(1, dtype=torch.int32, device="cuda")
        d_tiles2 = torch.empty((tiles_per_page, MODEL_HW, MODEL_HW), dtype=torch.uint8, device="cuda")
        d_lab2 = torch.empty((PAGE_H, PAGE_W), dtype=torch.uint8, device="cuda")
        tiles_per_step = (1 + 2 * tiles_per_page) * world
        workload_desc = (f"three-model pipeline on one {PAGE_H}x{PAGE_W} page per GPU: border (whole image, 1 forward) + layout "
                         f"(device Otsu + binarising gather, 4 classes, {tiles_per_page} tiles) + textline ({tiles_per_page} tiles); models resident")

        def step():  # noqa: F811
            m_border.segment_whole(page, PAGE_H, PAGE_W)                       # host page in / host mask out (1 forward)
            # layout stage = otsu_copy + do_prediction (main.py:443-447): histogram, threshold and the
            # binarising gather all run on the device inside the timed step
            m_layout.ctx.otsu_dev(d_page.data_ptr(), PAGE_H, PAGE_W, d_thr.data_ptr())
            m_layout.ctx.segment_tile_range_bin_dev(d_page.data_ptr(), PAGE_H, PAGE_W, 0, tiles_per_page, d_thr.data_ptr(),
                                                    d_tiles2.data_ptr())
            m_layout.ctx.stitch_dev(d_tiles2.data_ptr(), PAGE_H, PAGE_W, d_lab2.data_ptr())
            ctx.segment_page_dev(d_page.data_ptr(), PAGE_H, PAGE_W, d_labels.data_ptr())
    elif args.workload == "batch64":
        # configs[3]: 64 pages of 4000x3000, whole pages per rank, one all-gather of the masks per step
        BH, BW, NPAGES = 4000, 3000, 64
        from sbb_textline_detection_amd.distributed import shard_block
        first, count, block = shard_block(NPAGES, rank, world)
        pages = [torch.from_numpy(synthetic_page(BH, BW, seed=100 + first + k)).cuda() for k in range(count)]
        d_mine = torch.empty((block, BH, BW), dtype=torch.uint8, device="cuda")
        d_everything = torch.empty((world * block, BH, BW), dtype=torch.uint8, device="cuda") if world > 1 else None
        tpp = _capi.tile_grid(BH, BW, MODEL_HW, MODEL_HW)[0].shape[0]
        tiles_per_step = tpp * NPAGES
        scaling = "strong"
        workload_desc = (f"{NPAGES} pages of {BH}x{BW} ({tpp} tiles each) sharded as whole pages over the ranks, textline model, "
                         f"one RCCL all-gather of the u8 masks per step")

        def step():  # noqa: F811
            for k in range(count):
                ctx.segment_page_dev(pages[k].data_ptr(), BH, BW, d_mine[k].data_ptr())
            if world > 1:
                dist.all_gather_into_tensor(d_everything.view(-1), d_mine.view(-1))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_tiles = tiles_per_step * args.steps
    value = total_tiles / dt

    # ---- roofline of the dominant kernel: per-launch HIP events on the library's stream ----------
    roofline = None
    per_op = []
    if rank == 0:
        ctx.profile_enable(True)
        ctx.profile_reset()
        for _ in range(max(2, min(args.steps, 5))):
            ctx.segment_page_dev(d_page.data_ptr(), PAGE_H, PAGE_W, d_labels.data_ptr())
        torch.cuda.synchronize()
        prof = ctx.profile()
        ctx.profile_enable(False)
        convs = [o for o in prof if "conv" in o["name"] and o["launches"] > 0]       # incl. stem_conv*, direct_conv*, tail_conv*
        tot_ms = sum(o["total_ms"] for o in prof)
        conv_ms = sum(o["total_ms"] for o in convs)
        conv_flops = sum(o["flops"] * o["patches"] for o in convs)
        # dominant kernel launch = the conv launch with the largest average duration (the four output-
        # parity classes of a decoder conv run as one grouped launch)
        dom = max(convs, key=lambda o: o["total_ms"] / o["launches"])
        ach = dom["flops"] * dom["patches"] / (dom["total_ms"] * 1e-3) / 1e12
        k3 = [o for o in convs if any(t in o["name"] for t in ("conv3x3", "conv2x2"))]          # the 3x3 conv stages
        k3_ms = sum(o["total_ms"] for o in k3)
        k3_flops = sum(o["flops"] * o["patches"] for o in k3)
        roofline = {
            "bound": "mfma", "kernel": "conv_igemm_mfma:" + dom["name"],
            "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None,
            "avg_launch_ms": round(dom["total_ms"] / dom["launches"], 4),
            "patches_per_launch": dom["patches"] / dom["launches"],
            "flops_per_launch": dom["flops"] * dom["patches"] / dom["launches"],
            "launch_mode": "per-launch HIP events in a profiling pass right after the timed region: one 70-tile launch per op on "
                           "one lane (exclusive GPU).  The timed region runs the same kernels as two concurrent 35-tile halves "
                           "(lanes=2), where per-launch durations overlap and are not separable",
            "flops_note": "algorithmic FLOPs (reference formulation: 2*MACs of the 3x3 conv over the upsampled+"
                          "concatenated input); the parity-split kernels issue 13/18 of them as MFMA work",
            "conv3x3_stages": {"achieved": round(k3_flops / (k3_ms * 1e-3) / 1e12, 2),
                               "frac": round(k3_flops / (k3_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                               "share_of_gpu_time": round(k3_ms / tot_ms, 4)},
            "all_convs": {"achieved": round(conv_flops / (conv_ms * 1e-3) / 1e12, 2),
                          "frac": round(conv_flops / (conv_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                          "share_of_gpu_time": round(conv_ms / tot_ms, 4)},
        }
        # HBM-side traffic of that launch from the committed rocprofv3 PMC passes of this same command
        # (tools/pmc_run.sh; counters need their own runs, see profiles/): bytes per launch
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01u_pmc_summary.json")))["ops"].get(dom["name"])
            if pmc and abs(dom["patches"] / dom["launches"] - 70) < 1e-6:
                roofline["traffic"] = round(pmc["fetch_bytes"] + pmc["write_bytes"])
                if "mfma_busy_pct" in pmc:
                    roofline["mfma_pipe_busy_pct_pmc"] = pmc["mfma_busy_pct"]   # issued work (13/18 of the algorithmic FLOPs)
                roofline["traffic_source"] = "profiles/r01u_pmc_summary.json (FETCH_SIZE x2 + WRITE_SIZE per launch, L2 hit %.0f %%)" % pmc["l2_hit_pct"]
        except Exception:
            pass
        for o in prof:
            if o["launches"]:
                per_op.append({"name": o["name"], "ms_per_launch": round(o["total_ms"] / o["launches"], 4),
                               "tflops": round(o["flops"] * o["patches"] / (o["total_ms"] * 1e-3) / 1e12, 1) if o["total_ms"] else 0})

    # ---- CPU baseline (oracle port) on a bounded sample; also the live label-map check -----------
    cpu_baseline, label_match = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import keras_forward as kf
        xy = _capi.tile_grid(PAGE_H, PAGE_W, MODEL_HW, MODEL_HW)[0]
        pick = np.linspace(0, len(xy) - 1, args.cpu_patches).astype(int)
        patches = np.stack([page[y0:y0 + MODEL_HW, x0:x0 + MODEL_HW] for (x0, y0) in xy[pick]])
        x = (patches / 255.0).astype(np.float32)
        kf.forward_config(cfg, weights, x[:1])                        # warm (page-in, thread pool)
        t1 = time.perf_counter()
        ref = kf.forward_config(cfg, weights, x)
        cpu_dt = time.perf_counter() - t1
        got = model.predict(x)
        srt = np.sort(ref, axis=-1)
        margin = srt[..., -1] - srt[..., -2]
        mism = ref.argmax(-1) != got.argmax(-1)
        label_match = {"vs": "oracle (fp32 CPU port)", "patches": int(len(pick)),
                       "max_abs_softmax_diff": round(float(np.abs(ref - got).max()), 5),
                       "label_mismatch_frac": round(float(mism.mean()), 6),
                       "max_oracle_margin_among_mismatches": round(float(margin[mism].max()) if mism.any() else 0.0, 5)}
        cpu_baseline = {"value": round(len(pick) / cpu_dt, 3), "unit": "patches/s", "cores": kf.num_threads(),
                        "kind": "port",
                        "sample": f"{len(pick)} of the page's {len(xy)} 448x448 tiles through oracle/keras_forward "
                                  f"(fp32 C conv + numpy, OpenMP); host has {os.cpu_count()} logical CPUs"}

    if rank == 0:
        out = {
            "metric": "segmented patches/sec (448x448x3) per GPU + per-pixel label-map match vs ref",
            "value": round(value, 2), "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": workload_desc, "workload_id": args.workload,
                       "tiles_per_step": tiles_per_step, "max_batch": args.max_batch,
                       "lanes": int(os.environ.get("SBBSEG_LANES", "2")),
                       "exchange": "all_gather of u8 label maps over RCCL" if world > 1 else "none (1 GPU)",
                       "flops_per_patch": 2 * model.plan.macs_per_patch()},
            "patches_per_s_per_gpu": round(value / world, 2),
            "achieved_tflops_end_to_end": round(value / world * 2 * model.plan.macs_per_patch() / 1e12, 1),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "label_match": label_match,
        }
        print(json.dumps(out))
        if os.environ.get("SBBSEG_BENCH_OPS"):
            with open(os.environ["SBBSEG_BENCH_OPS"], "w") as f:
                json.dump(per_op, f, indent=1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    model.release()


if __name__ == "__main__":
    main()
