#!/usr/bin/env python3
"""Throughput bench for the MI355X text-detection hot path.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic frames that are already resident in HBM:
fused preprocess (BGR uint8 720p -> 640x640 fp16) -> DBNet-ResNet18 (fp16 MFMA) -> post-process to boxes
[-> crop + CRNN + CTC decode when --workload full].  Workload at N=1 = BASELINE.json configs[1]
("Batch=32 720p frames, DBNet detector only, 1xMI355X fp16"); frames and "margin" weights as SURVEY.md 8d.
For N>1 every rank runs its own 32 frames (weak scaling, frames shard with no data-path collective); the
only exchange is the gather of detection records to every rank (RCCL all_gather of a small padded block).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = the
implicit-GEMM convolution, timed with HIP events on its launch stream across the timed region) and
`cpu_baseline` (the CPU oracle timed on this host on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "video-text-detection-system_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

MFMA_PEAK_TFLOPS = 2500.0  # dense fp16, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--backbone", default="resnet18")
    ap.add_argument("--workload", default="full", choices=["full", "detector"],
                    help="full = BASELINE configs[2] (detect + recognize); detector = configs[1]")
    ap.add_argument("--recognizer", default="crnn", choices=["crnn", "trocr"],
                    help="crnn = BASELINE configs[2]; trocr = the Transformer recogniser of configs[4] (trocr-base-printed architecture, seeded weights)")
    ap.add_argument("--mixed", action="store_true",
                    help="BASELINE configs[4] batch: half of the frames 720p, half 1080p (two equally sized sub-batches per step)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP-event timing")
    ap.add_argument("--upload", action="store_true",
                    help="PCIe-inclusive variant (not the headline value): every step uploads its batch from pinned host memory "
                         "on an upload stream")
    ap.add_argument("--layers-out", default=None, help="write the per-launch table as JSON to this path")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / aggregation only (gloo, no GPU work): what the CPU test of the N>1 entry runs")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a torchrun wrapper: this parent starts the N ranks itself as fresh child processes
    (one per GPU) and never touches the GPU (it does not even import torch), relays rank 0's JSON line and exits non-zero
    if any rank does.  Under `python -m torch.distributed.run ... bench.py --gpus N` WORLD_SIZE is already set and this is
    not taken."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # a rank that dies leaves the others waiting in a collective: stop them (exact PIDs) instead of hanging with them
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.05)
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=20))
        except subprocess.TimeoutExpired:
            p.kill()
            codes.append(p.wait())
    reader.join(timeout=5)
    for line in b"".join(chunks).decode().splitlines():   # the JSON line to stdout, library chatter (gloo banners) to stderr
        print(line, file=sys.stdout if line.lstrip().startswith("{") else sys.stderr)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def dry_run(args, rank, world):
    """The N>1 control flow without the GPU: rendezvous, barrier, MAX-over-ranks of the elapsed time, one JSON line."""
    if os.environ.get("VTD_BENCH_FAIL_RANK") == str(rank):   # test hook: this rank dies before the rendezvous
        raise SystemExit(3)
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ranks = [None] * world
        dist.all_gather_object(ranks, (rank, int(os.environ.get("LOCAL_RANK", "0"))))
    else:
        ranks = [(0, 0)]
    if rank == 0:
        print(json.dumps({"metric": "dry-run", "value": 0.0, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "dry_run": True, "ranks": ranks, "elapsed_max_s": float(t.item())}))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.dry_run:
        return dry_run(args, rank, world)
    import numpy as np
    import torch
    from vtd_amd._fixtures import synth, weights
    from vtd_amd.engine import DeviceFrames, detector_profile

    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if os.environ.get("VTD_BENCH_DET_PRIORITY") == "1":   # experiment: the caller's (detector) stream above the side streams
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
    dist = None
    if world > 1:
        import torch.distributed as dist
        # backend "nccl" is RCCL on ROCm; VTD_DIST_BACKEND=gloo lets two ranks rehearse the N>1 path on one GPU
        dist.init_process_group(os.environ.get("VTD_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    B, H, W = args.batch, args.height, args.width
    if args.mixed:   # configs[4]: alternating 720p / 1080p frames, processed as one 720p and one 1080p sub-batch per step
        if args.upload or args.workload != "full" or B % 2:
            raise SystemExit("--mixed needs --workload full, an even --batch and no --upload")
        sizes = [(720, 1280), (1080, 1920)]
        sub = [np.stack([synth.text_frame(2000 + rank * B + 2 * i + k, *sizes[k])[0] for i in range(B // 2)]) for k in range(2)]
        sub_frames = [DeviceFrames(f) for f in sub]
        frames = sub[0]
    else:
        frames = np.stack([synth.text_frame(100 + rank * B + i, H, W)[0] for i in range(B)])
    dev_frames = DeviceFrames(frames)
    if args.upload:
        host_frames = torch.from_numpy(frames).pin_memory()
        upload_stream = torch.cuda.Stream()
    sd = weights.margin_detector_state_dict(args.backbone, 0)
    from vtd_amd import nets as mynets
    from vtd_amd import shard
    from vtd_amd.pipeline import VideoTextPipeline
    os.environ["VTD_MAX_BATCH"] = str(B)
    if args.recognizer == "trocr":
        os.environ.setdefault("VTD_TROCR_MAX_CROPS", "512")
        os.environ.setdefault("VTD_TROCR_SEEDED", "0")   # explicit opt-in: the architecture on synthetic weights (nothing is fetchable)
    pipe = VideoTextPipeline(use_transformer_ocr=args.recognizer == "trocr", backbone=args.backbone, batch_size=B)
    pipe.detector.max_detections = MAX_DET = 64
    pipe.detector.model.load_state_dict(sd)
    # VTD_BENCH_CRNN=default: torch-default-init recogniser weights (round-1 bench; every crop decodes to two characters)
    rec_sd = (mynets.seeded_state_dict(lambda: mynets.CRNN(97), seed=11) if os.environ.get("VTD_BENCH_CRNN") == "default"
              else weights.margin_crnn_state_dict(11))
    if args.recognizer == "crnn":
        pipe.recognizer.model.load_state_dict(rec_sd)
    eng = pipe.detector.model.engine()
    lib = eng.lib
    last = {}

    inflight = {"det": None, "rec": None}

    turn = {"k": 0}

    def next_batch():
        if args.mixed:
            turn["k"] ^= 1
            return sub_frames[turn["k"] ^ 1]
        return DeviceFrames(host_frames, stream=upload_stream) if args.upload else dev_frames

    def step_detector():
        # the product's detector half, two batches in flight: enqueue batch i (preprocess -> DBNet on the caller's stream,
        # post-process + record copy on the side stream), then turn the records of batch i-1 into the result dicts
        t = pipe.detector.submit_batch(next_batch(), 0.5)
        if inflight["det"] is not None:
            last["detections"] = pipe.detector.finish_batch(inflight["det"])
        inflight["det"] = t
        return t["keep"][1][:B], t["keep"][2][:B]

    def drain_detector():
        if inflight["det"] is not None:
            last["detections"] = pipe.detector.finish_batch(inflight["det"])
            inflight["det"] = None

    # full workload: the product's batched pass, software-pipelined three deep exactly as a video loop would run it --
    # detector(i) is enqueued, then the host collects the boxes of batch i-1 and enqueues its recogniser, then builds
    # the result dicts of batch i-2.  Every step retires one whole batch (result dicts included).
    def step_full():
        job = pipe.submit_detection(next_batch())
        keep = job["det"]["keep"]
        if inflight["rec"] is not None:
            last["results"] = pipe.collect(inflight["rec"])
            inflight["rec"] = None
        if inflight["det"] is not None:
            inflight["rec"] = pipe.submit_recognition(inflight["det"])
        inflight["det"] = job
        return keep[1][:B], keep[2][:B]

    def drain_full():
        while inflight["det"] is not None or inflight["rec"] is not None:
            if inflight["rec"] is not None:
                last["results"] = pipe.collect(inflight["rec"])
                inflight["rec"] = None
            if inflight["det"] is not None:
                inflight["rec"] = pipe.submit_recognition(inflight["det"])
                inflight["det"] = None

    def step():
        if args.mixed:
            step_full()
        rec, cnt = step_full() if args.workload == "full" else step_detector()
        if world > 1:
            # the one exchange step of the path: detection records of every rank to every rank.  Enqueued on the stream
            # that produced them (the detector's post-process side stream), so it is ordered behind the records without
            # stalling the caller's stream, where the next batch's DBNet is already queued.
            side = pipe.detector._post_stream()
            with torch.cuda.stream(side if side is not None else torch.cuda.current_stream()):
                last["gathered"] = shard.gather_detections(rec, cnt)
        return rec, cnt

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    drain = drain_full if args.workload == "full" else drain_detector
    # one priming step whatever --warmup says: the first call at a batch size runs the per-layer autotune and the workspace
    # allocations (hundreds of ms), which are set-up cost, not a step of the path
    step()
    drain()
    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    # Untimed survey pass: every launch of the detector graph bracketed with HIP events -> per-launch table and the identity
    # of the dominant launch.  The timed region then brackets only that launch (2 events per step).
    survey, dom = [], -1
    if not args.no_profile:
        lib.vtd_detector_set_profiling(eng.handle, 1)
        for _ in range(3):  # one batch at a time, drained: every launch has the GPU to itself (no other stream running)
            step()
            drain()
            torch.cuda.synchronize()
        barrier()
        survey = detector_profile(eng)
        dom = max(range(len(survey)), key=lambda i: survey[i][1] if survey[i][3] > 0 else -1.0)
        lib.vtd_detector_set_profiling(eng.handle, 2 + dom)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec, cnt = step()
    drain()  # K batches submitted -> K batches retired (result dicts built) inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- sanity: the timed path produced detections (not part of the timed region)
    counts = cnt.cpu().numpy()
    n_det = int(counts.sum())

    roofline = None
    layer_rows = []
    if not args.no_profile:
        timed = detector_profile(eng)  # only slot `dom` carries events from the timed region
        lib.vtd_detector_set_profiling(eng.handle, 0)
        convs = []
        for name, ms, calls, macs in survey:
            layer_rows.append({"launch": name, "ms_total": ms, "calls": calls, "gmac_total": macs / 1e9,
                               "tflops": (2 * macs / (ms * 1e-3) / 1e12) if ms > 0 and macs > 0 else None})
            if macs > 0 and calls > 0:
                convs.append((name, ms / calls, macs / calls))
        name, ms, calls, macs = timed[dom]
        achieved = 2 * macs / (ms * 1e-3) / 1e12
        iso_ms, iso_calls, iso_macs = survey[dom][1], survey[dom][2], survey[dom][3]
        conv_us = sum(r[1] for r in convs) * 1e3  # per step, survey pass
        # algorithmic (reference-graph) FLOPs of the whole detector per step, SURVEY 8d: 69.8 GFLOP / frame for R18
        algo_flops_step = 2.0 * eng.macs_per_frame * B
        traffic = traffic_detail = None
        try:  # HBM bytes per launch of that kernel from the separate rocprofv3 --pmc passes (profiles/, FETCH_SIZE x2 corrected)
            import glob
            pmc_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_per_launch.json")))[-1]
            pmc = json.load(open(pmc_path))

            def pick(substr):
                hits = [v for k, v in pmc.items() if substr in k and v["launches"] >= 3]
                return max(hits, key=lambda h: h["launches"]) if hits else None

            # launch-slot description (vtd_api.cpp) -> device kernel symbol of exactly that variant
            symbol = next((sym for key, sym in (("head_entry_pair", "head_entry_pair_kernel("), ("head_entry_half", "head_entry_half_kernel<false>("),
                                                ("head_entry_halo256", "head_entry_halo256_kernel<false>("),
                                                ("head_entry_halo ", "head_entry_halo_kernel<"),
                                                ("classed", "true>(")) if key in name), None)
            parts = [v for v in [pick(symbol) if symbol else None] if v]
            if parts:
                rd = sum(v["hbm_read_MB_corrected_x2"] for v in parts)
                wr = sum(v["hbm_write_MB"] for v in parts)
                traffic = int((rd + wr) * 1e6)
                traffic_detail = {"hbm_read_MB": round(rd, 2), "hbm_write_MB": round(wr, 2),
                                  "kernel_symbol": symbol,
                                  "source": "profiles/" + os.path.basename(pmc_path) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                                            "median over launches, FETCH_SIZE x2 per the gfx950 note)"}
        except Exception:
            traffic = traffic_detail = None
        roofline = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch (read + written)", "traffic_detail": traffic_detail,
                    "launches": calls, "avg_launch_us": round(ms / calls * 1e3, 2),
                    "executed_gflop_per_launch": round(2 * macs / calls / 1e9, 3),
                    "alone_on_gpu": {"avg_launch_us": round(iso_ms / iso_calls * 1e3, 2),
                                     "tflops": round(2 * iso_macs / (iso_ms * 1e-3) / 1e12, 2),
                                     "frac": round(2 * iso_macs / (iso_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                     "what": "same launch in the untimed survey pass, one batch at a time: in the timed region the "
                                             "recogniser and post-process streams share the CUs with it, which stretches its wall time"},
                    "note": "achieved = FLOPs this launch EXECUTES / its HIP-event time on the launch stream; the composed conv replaces "
                            "lateral(C2)+top-down add+P2 smooth+head conv of the reference graph with ~3x fewer FLOPs, so the "
                            "reference-graph rate over all matrix launches is given as net_algorithmic_tflops; peak = nominal dense fp16 "
                            "(a bare MFMA loop sustains ~2.3 PFLOP/s on this part, tools/mfma_peak.hip)",
                    "all_mfma_launches_tflops_executed": round(sum(2 * r[2] for r in convs) / (conv_us * 1e-6) / 1e12, 2),
                    "net_algorithmic_tflops": round(algo_flops_step / (conv_us * 1e-6) / 1e12, 2),
                    "detector_mfma_launches_us_per_step": round(conv_us, 1)}
        if args.layers_out and rank == 0:
            with open(args.layers_out, "w") as f:
                json.dump(layer_rows, f, indent=1)

    cpu_baseline = None
    if rank == 0 and args.gpus == 1 and args.cpu_seconds > 0:
        from oracle import pipeline as opipe
        # the 1-GPU box grants a 16-CPU share of a much larger host: more threads than that only thrash
        ncores = min(len(os.sched_getaffinity(0)), int(os.environ.get("VTD_CPU_CORES", "16")))
        torch.set_num_threads(ncores)
        def cpu_frame(fr):
            if args.workload == "full":  # reference-shaped: one detect per frame, one recognize per crop (pipeliine.py:96-133)
                return opipe.process_frame_batch([fr], [(0, 0.0)], sd, args.backbone, rec_sd, 0.5)
            return opipe.detect(fr, sd, args.backbone, 0.5)
        cpu_frame(frames[0])  # warm-up (oneDNN primitive creation)
        done, t_cpu = 0, time.perf_counter()
        while done < B and time.perf_counter() - t_cpu < args.cpu_seconds:
            cpu_frame(frames[done])
            done += 1
        dt = time.perf_counter() - t_cpu
        what = ("preprocess + DBNet-%s fp32 torch CPU + C post-process%s" %
                (args.backbone, " + per-crop cv-resize + CRNN fp32 + CTC decode" if args.workload == "full" else ""))
        cpu_baseline = {"value": round(done / dt, 3), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                        "sample": f"{done} of the {B} {H}p frames, one at a time ({what})"}

    if rank == 0:
        total_frames = world * B * args.steps
        out = {
            "metric": "frames/sec (detect+recognize) @720p, 1/2/4/8 MI355X; box IoU vs CPU ref",
            "value": round(total_frames / elapsed, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic" + (" (uploaded from pinned host memory every step: PCIe-inclusive variant)" if args.upload else ""),
            "config": {"workload": (f"B={B} {H}p frames, DBNet-{args.backbone} detector only (preprocess+net+post-process), fp16"
                                    if args.workload == "detector" else
                                    f"B={B} {'mixed 720p/1080p' if args.mixed else str(H) + 'p'} frames, full pipeline: DBNet-{args.backbone} + crop + "
                                    + ("TrOCR (ViT-base-384 encoder + 12-layer decoder, greedy max_length=50)" if args.recognizer == "trocr"
                                       else "CRNN + CTC decode") + " -> result dicts, fp16"),
                       "global_batch": world * B, "frame": [H, W], "backbone": args.backbone,
                       "parallelism": f"frames sharded over {world} rank(s), detections all-gathered",
                       "detections_last_step_rank0": n_det,
                       "crops_recognized_last_step_rank0": (sum(len(r["detections"]) for r in last["results"]) if "results" in last else 0)},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
