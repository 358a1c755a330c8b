#!/usr/bin/env python
"""Benchmark of the Faster R-CNN proposal + RoI-pool hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic RPN-head outputs and feature maps
already resident in HBM:  flatten -> per-image top-k (12000) -> decode / clip / small-box filter ->
NMS (0.7, keep 2000) -> gather -> RoI list -> RoIAlign 7x7 (C=512).  Workload = BASELINE.json
configs[1] as made concrete in SURVEY.md 8(d) "config 2": batch 16 synthetic 600x1000 images per
GPU, one stride-16 map 37x62, A = 20,646 anchors/image, reference (bug-compatible) decode.
For N > 1 every rank runs its own 16 images (weak scaling, no collective inside the timed region;
one RCCL all_gather of the padded proposals afterwards, timed separately).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel of the step as measured with
HIP events on the launch stream inside the timed region; `cpu_baseline` times the CPU oracle
(oracle/, a C restatement of the reference's algorithm) on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
IMG_H, IMG_W, FH, FW, NA, C = 600, 1000, 37, 62, 9, 512
A = FH * FW * NA
PRE_NMS, POST_NMS, NMS_THR, MIN_SIZE = 12000, 2000, 0.7, 1e-3
SETTLE_STEPS = 40
TOPK_SLICES_PIPELINED = int(os.environ.get("BENCH_TOPK_SLICES", "1"))
# (experimental, see DESIGN.md section 5: +2 % as the run's main schedule, unstable as the second schedule of a process)
PIPE_ONE_CALL = int(os.environ.get('BENCH_PIPE_ONE_CALL', '0'))
PIPE_ONE_CALL_SLICES = int(os.environ.get('BENCH_PIPE_SLICES', '0'))


_ANCHOR_WH = None


def anchor_wh():
    """fp32 (w, h) of the A anchors in flat order, from the reference's formula
    (RPN/anchor_generator.py:59-61,86-93): host arithmetic, used only to synthesise inputs."""
    global _ANCHOR_WH
    if _ANCHOR_WH is None:
        sizes = torch.tensor([128., 128., 128., 256., 256., 256., 512., 512., 512.])
        ratios = torch.tensor([0.5, 1.0, 2.0] * 3)
        w, h = sizes * ratios, sizes
        base = torch.stack([-w, -h, w, h], 1) / 2
        sy, sx = torch.tensor(IMG_H / FH, dtype=torch.float32), torch.tensor(IMG_W / FW, dtype=torch.float32)
        gy, gx = torch.meshgrid(torch.arange(FH, dtype=torch.float32) * sy, torch.arange(FW, dtype=torch.float32) * sx,
                                indexing="ij")
        coord = torch.stack([gx.flatten(), gy.flatten(), gx.flatten(), gy.flatten()], 1)
        anc = (coord[:, None, :] + base[None]).reshape(-1, 4)
        _ANCHOR_WH = (anc[:, 2] - anc[:, 0], anc[:, 3] - anc[:, 1])
    return _ANCHOR_WH


def synth_image(index, workload="realistic"):
    """Synthetic RPN-head outputs + feature map for one image (seed 1234 + index).

    Logits are tie-free.  Deltas:
      "gaussian"  -- SURVEY.md 8(d) literally: N(0, .1) for dx, dy and N(0, .2) for dw, dh.  Under the
                     reference's decode (quirk Q1: the anchor origin is not added) every proposal lands
                     on the image origin and NMS keeps only ~270 of 12,000.
      "realistic" -- the deltas a trained head would have to emit UNDER THE SAME reference decode for
                     proposals spread over the image: target centre uniform in the image, target size
                     16 * 2^U[0,5]; dx = (cx - w_a/2) / w_a, dw = log(w_t / w_a).  NMS keeps the full
                     2,000 proposals per image, which is the "12000 -> 2000, RoIPool 7x7" case
                     BASELINE.json describes (206 MB of RoI features per image)."""
    g = torch.Generator().manual_seed(1234 + index)
    cls = (((torch.randperm(A, generator=g).float() / A) - 0.5) * 8).reshape(FH, FW, NA).permute(2, 0, 1).contiguous()
    if workload == "gaussian":
        reg = torch.randn(4 * NA, FH, FW, generator=g)
        reg[0::4] *= 0.1
        reg[1::4] *= 0.1
        reg[2::4] *= 0.2
        reg[3::4] *= 0.2
    else:
        wa, ha = anchor_wh()
        cx = torch.rand(A, generator=g) * IMG_W
        cy = torch.rand(A, generator=g) * IMG_H
        wt = 16 * 2 ** (torch.rand(A, generator=g) * 5)
        ht = 16 * 2 ** (torch.rand(A, generator=g) * 5)
        d = torch.stack([(cx - wa / 2) / wa, (cy - ha / 2) / ha, torch.log(wt / wa), torch.log(ht / ha)], 1)
        reg = d.reshape(FH, FW, 4 * NA).permute(2, 0, 1).contiguous()  # (A,4) flat order -> (36,h,w)
    feat = torch.randn(C, FH, FW, generator=g)
    return cls, reg, feat


def make_batch(first_index, batch, workload="realistic"):
    items = [synth_image(first_index + i, workload) for i in range(batch)]
    return (torch.stack([i[0] for i in items]), torch.stack([i[1] for i in items]),
            torch.stack([i[2] for i in items]))


class Path:
    """The hot path on one GPU, stage by stage so that each stage can be bracketed by events."""

    def __init__(self, dev, textbook=False):
        from objectdetection_amd import ops
        from objectdetection_amd.RPN.anchor_generator import AnchorGenerator
        from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
        self.ops = ops
        # NMS launches bounded by what the finished steps needed (ops.NmsRounds: same keep lists; the first steps of a
        # workload, before any has finished, enqueue every round)
        self.nms_rounds = ops.NmsRounds()
        self.path_ws = None
        self.pipe_ws = None
        self.dev = dev
        self.ag = AnchorGenerator()
        self.pool = ROIPooling(['1'], [7, 7], 2)
        self.textbook = textbook
        # two HIP streams: the proposal stages (a few single-workgroup-per-image kernels that leave most
        # of the chip idle) of batch i+1 run while RoIAlign (the whole chip, store-bound) of batch i drains
        from objectdetection_amd.pipeline import ProposalRoIPipeline
        self.pipe = ProposalRoIPipeline(dev)
        # BENCH_FLATTEN=1: the three-launch form with flattened copies (rpn_flatten + topk_desc + decode)
        self.in_place = os.environ.get("BENCH_FLATTEN", "0") != "1"

    def anchors(self, batch, feat):
        return torch.cat([a[0].reshape(-1, 4) for a in self.ag((batch, IMG_H, IMG_W), [feat])], 0)

    def step(self, anchors, cls, reg, feat, ev=None, only_roi=False):
        """ev: list that receives (stage, event) marks.  only_roi: bracket RoIAlign alone -- an event between
        two launches costs the stream ~5 us, so the timed region carries just the two the roofline needs and
        the other stages are timed in a few extra steps outside it."""
        ops = self.ops

        def mark(name):
            if ev is not None and (not only_roi or name in ("gather", "roi_align")):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((name, e))

        mark("start")
        if self.in_place and (ev is None or only_roi):
            # the proposal half as the RPN module runs it: ONE library call (ops.rpn_proposals_nchw); the staged form
            # below, same kernels and results, is what the instrumented steps time stage by stage
            props, n_keep, self.path_ws = ops.rpn_proposals_nchw([cls], [reg], anchors, PRE_NMS, IMG_H, IMG_W, MIN_SIZE,
                                                                 NMS_THR, POST_NMS, rounds=self.nms_rounds,
                                                                 workspace=self.path_ws)
            mark("gather")
            out, total = self.pool.forward_padded({'1': feat}, props, n_keep, [(IMG_H, IMG_W)])
            mark("roi_align")
            return props, n_keep, out, total
        if self.in_place:  # RPN/rpn.py:56-57 never materialised: the head's NCHW outputs are read where they lie
            top_val, top_idx = ops.topk_desc_nchw([cls], PRE_NMS)
            mark("topk")
            boxes, _, _, count = ops.decode_clip_filter_nchw(anchors, [reg], top_idx, top_val, IMG_H, IMG_W, MIN_SIZE)
            mark("decode_clip_filter")
        else:
            scores, deltas = ops.rpn_flatten([cls], [reg])
            mark("flatten")
            top_val, top_idx = ops.topk_desc(scores, PRE_NMS)
            mark("topk")
            boxes, _, _, count = ops.decode_clip_filter(anchors, deltas, top_idx, top_val, IMG_H, IMG_W, MIN_SIZE)
            mark("decode_clip_filter")
        keep, n_keep = ops.nms_sorted(boxes, count, NMS_THR, POST_NMS, rounds=self.nms_rounds)
        mark("nms")
        props = ops.gather_boxes(boxes, keep, n_keep)
        mark("gather")
        out, total = self.pool.forward_padded({'1': feat}, props, n_keep, [(IMG_H, IMG_W)])
        mark("roi_align")
        return props, n_keep, out, total

    def _pipe_rounds(self):
        mode = os.environ.get("BENCH_PIPE_ROUNDS", "shared")
        if mode == "none":
            return None
        if mode == "own":
            if not hasattr(self, "_own_rounds"):
                self._own_rounds = self.ops.NmsRounds()
            return self._own_rounds
        return self.nms_rounds

    def step_pipelined(self, anchors, cls, reg, feat, ev=None, only_roi=False):
        """Same work as step(); proposal stages and RoIAlign on the two streams of ProposalRoIPipeline,
        so that consecutive steps overlap (software pipeline of depth 2 over batches)."""
        ops, pipe = self.ops, self.pipe

        def mark(name, stream):
            if ev is not None and (not only_roi or name in ("roi_queue_wait", "roi_align")):
                e = torch.cuda.Event(enable_timing=True)
                e.record(stream)
                ev.append((name, e))

        def proposals():
            mark("start", pipe.s_prop)
            if self.in_place and (ev is None or only_roi) and PIPE_ONE_CALL:
                # the proposal half as ONE library call on the proposal stream (its workspace is reused by the next
                # batch's call on the same stream: calls on one stream run in order)
                props, n_keep, self.pipe_ws = ops.rpn_proposals_nchw([cls], [reg], anchors, PRE_NMS, IMG_H, IMG_W,
                                                                     MIN_SIZE, NMS_THR, POST_NMS,
                                                                     max_slices=PIPE_ONE_CALL_SLICES,
                                                                     rounds=self._pipe_rounds(), workspace=self.pipe_ws)
                mark("gather", pipe.s_prop)
                return props, n_keep
            scores, deltas = ops.rpn_flatten([cls], [reg])
            mark("flatten", pipe.s_prop)
            # one workgroup per image: beside RoIAlign every slice workgroup would wait for a CU of its own
            top_val, top_idx = ops.topk_desc(scores, PRE_NMS, max_slices=TOPK_SLICES_PIPELINED)
            mark("topk", pipe.s_prop)
            boxes, _, _, count = ops.decode_clip_filter(anchors, deltas, top_idx, top_val, IMG_H, IMG_W, MIN_SIZE)
            mark("decode_clip_filter", pipe.s_prop)
            keep, n_keep = ops.nms_sorted(boxes, count, NMS_THR, POST_NMS, rounds=self.nms_rounds)
            mark("nms", pipe.s_prop)
            props = ops.gather_boxes(boxes, keep, n_keep)
            mark("gather", pipe.s_prop)
            return props, n_keep

        def roi(props, n_keep):
            mark("roi_queue_wait", pipe.s_roi)
            res = self.pool.forward_padded({'1': feat}, props, n_keep, [(IMG_H, IMG_W)])
            mark("roi_align", pipe.s_roi)
            return res

        props, n_keep, (out, total) = pipe.submit(proposals, roi)
        return props, n_keep, out, total


def host_cpu():
    """What the CPU baseline ran on: model name, logical CPUs of the machine, and the threads this process may
    use -- on a shared GPU box the job owns a SHARE of the machine (OMP_NUM_THREADS / the affinity mask say how
    much), and oversubscribing it would time the neighbours, not the oracle."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 1
    env = os.environ.get("OMP_NUM_THREADS")
    share = int(env) if env and env.isdigit() and int(env) > 0 else affinity
    return model, os.cpu_count() or 1, affinity, min(share, affinity)


def cpu_baseline(batch_sample, workload, seconds=10.0):
    """Times the CPU oracle on `batch_sample` images of the same workload (full path): passes of
    `batch_sample` images each, at least 5 and about `seconds` of CPU work; value = median pass."""
    from oracle import oracle as orc
    orc.build()
    model, machine_cpus, affinity, threads = host_cpu()
    orc.set_threads(threads)
    params = dict(num_pre_nms=PRE_NMS, num_post_nms=POST_NMS, nms_threshold=NMS_THR, proposal_min_size=MIN_SIZE)
    imgs = [synth_image(i, workload) for i in range(batch_sample)]

    def one(i):
        cls, reg, feat = imgs[i]
        props = orc.rpn_proposals((1, IMG_H, IMG_W), [cls[None].numpy()], [reg[None].numpy()], params)
        out, _ = orc.multiscale_roi_align([feat[None].numpy()], props, [(IMG_H, IMG_W)], (7, 7), 2)
        return len(props[0]), out

    for i in range(min(2, batch_sample)):  # warm-up 2
        one(i)
    passes, rois = [], 0
    t_all = time.perf_counter()
    while len(passes) < 5 or (time.perf_counter() - t_all < seconds and len(passes) < 10000):
        t0 = time.perf_counter()
        for i in range(batch_sample):
            rois += one(i)[0]
        passes.append(time.perf_counter() - t0)
    total = time.perf_counter() - t_all
    med = float(np.median(passes))
    # the metric's second half on the same host: the oracle's NMS alone (one thread: greedy NMS is a serial chain),
    # ms per image on the micro-benchmark's box sets -- median of 3 images each
    nms_ms = {}
    rng = np.random.default_rng(0)
    for n in (PRE_NMS, A):
        for kind in ("realistic", "clustered"):
            ts, keeps = [], []
            for _ in range(3):
                boxes = nms_box_set(kind, rng, n)
                score = -np.arange(n, dtype=np.float32)
                t0 = time.perf_counter()
                keep = orc.nms(boxes, score, NMS_THR)[:POST_NMS]
                ts.append((time.perf_counter() - t0) * 1e3)
                keeps.append(len(keep))
            nms_ms["n%d_%s" % (n, kind)] = {"ms_per_img": float(np.median(ts)), "keeps_mean": float(np.mean(keeps))}
    return dict(value=batch_sample / med, unit="images/s", cores=threads, kind="port", nms_ms_per_img=nms_ms,
                cpu_model=model, machine_logical_cpus=machine_cpus, affinity_cpus=affinity, threads_used=threads,
                threads_note="threads = this job's CPU share (OMP_NUM_THREADS / affinity mask), not the machine's count",
                passes=len(passes), pass_s_median=med, pass_s_min=float(min(passes)), pass_s_max=float(max(passes)),
                sample="median of %d passes over %d images of the same workload (full path incl. RoIAlign C=%d), "
                       "oracle/frcnn_oracle.c with OpenMP (RoIAlign, IoU rows) on %d threads, %.2f s in all, %d RoIs"
                       % (len(passes), batch_sample, C, threads, total, rois))


def run_workload(path, dev, dist, rank, world, B, workload, steps, warmup, pipelined=True):
    """Warm up, then time exactly `steps` steps between barrier+synchronize pairs; max over ranks."""
    step = path.step_pipelined if pipelined else path.step
    cls, reg, feat = (t.to(dev) for t in make_batch(rank * B, B, workload))
    anchors = path.anchors(B, feat)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed, before the W warm-up steps the contract asks for: the first few dozen steps after start-up run
    # 3-4 % slower (the 3.2 GB output blocks are new to the allocator and the page tables, clocks ramp), and
    # how many of them a run sees should not depend on the W the caller picked.
    for _ in range(SETTLE_STEPS):
        step(anchors, cls, reg, feat)
    for _ in range(warmup):
        step(anchors, cls, reg, feat)
    barrier()
    events = []
    t0 = time.perf_counter()
    for _ in range(steps):
        ev = []
        props, n_keep, out, total = step(anchors, cls, reg, feat, ev, only_roi=True)
        events.append(ev)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    def stage_means(all_events, stat=np.mean):
        acc = {}
        for ev in all_events:
            for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:]):
                acc.setdefault(n1, []).append(e0.elapsed_time(e1))
        return {k: float(stat(v)) for k, v in acc.items()}

    # RoIAlign: HIP events on its launch stream inside the timed region
    roi_ms = stage_means(events)["roi_align"]
    # the MEDIAN interval between the ends of consecutive RoIAlign launches, from the same events (GPU clock): the step
    # time with a one-off hiccup of the runtime left out (two-stream runs now and then lose 36-46 ms once, somewhere in
    # their first seventy steps: DESIGN.md section 5) -- reported beside the wall-clock figure, never instead of it
    ends = [e for ev in events for (name, e) in ev if name == "roi_align"]
    steady_ms = float(np.median([a.elapsed_time(b) for a, b in zip(ends[:-1], ends[1:])])) if len(ends) > 1 else None
    # every stage: a few fully instrumented steps after the timed region
    # (first a few of them unrecorded: the staged calls' first allocations leave the GPU waiting for the host, and its
    # clocks take some 20 ms to come back after an idle stretch -- RoIAlign 0.83-0.89 ms in the steps right behind one)
    for _ in range(SETTLE_STEPS):
        step(anchors, cls, reg, feat, [])
    extra = []
    for _ in range(max(3, steps // 5)):
        ev = []
        step(anchors, cls, reg, feat, ev)
        extra.append(ev)
    barrier()
    if os.environ.get("BENCH_DEBUG"):
        for ev in extra:
            print("BENCH_DEBUG instrumented step:", [(n1, round(e0.elapsed_time(e1), 4)) for (n0, e0), (n1, e1) in zip(ev[:-1], ev[1:])],
                  file=sys.stderr, flush=True)
    stage_ms = stage_means(extra, np.median)  # (median of the instrumented steps: one late launch does not colour a stage)
    stage_ms["roi_align"] = roi_ms
    queue_wait = stage_ms.pop("roi_queue_wait", None)  # time RoIAlign of step i waited behind that of step i-1
    counts = n_keep.tolist()
    n_rois = int(total.item())
    ms_per_step = elapsed / steps * 1e3
    # algorithmic HBM bytes of the RoIAlign launch: output write + one read of the feature maps + RoIs
    roi_bytes = n_rois * C * 49 * 4 + B * C * FH * FW * 4 + n_rois * 16
    # ... and of the rest of the path: scores read, top-k out, gathers, boxes, keeps
    rest_bytes = B * (A * 4 + PRE_NMS * 8 + PRE_NMS * 32 + PRE_NMS * 20 + PRE_NMS * 20 + POST_NMS * 24)
    ra_ms = stage_ms["roi_align"]
    traffic = None  # HBM bytes per launch from a separate rocprofv3 --pmc pass, valid for this exact launch shape
    try:
        with open(os.path.join(ROOT, "pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("rois") == n_rois and B == 16:
            traffic = pmc["traffic_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    if os.environ.get("BENCH_DEBUG"):
        st = torch.cuda.memory_stats()
        print("BENCH_DEBUG run", workload, "B", B, "steps", steps, "pipelined", pipelined, "ms/step %.3f" % ms_per_step,
              "segments +%s -%s" % (st.get("segment.all.allocated"), st.get("segment.all.freed")),
              "reserved %.1f GB" % (st.get("reserved_bytes.all.current", 0) / 1e9),
              {k: list(v) for k, v in path.nms_rounds._seen.items()},
              "step intervals (GPU clock, ms):", [round(a.elapsed_time(b), 2) for a, b in zip(ends[:-1], ends[1:])],
              file=sys.stderr, flush=True)
    res = dict(workload=workload, value=world * B * steps / elapsed, ms_per_step=ms_per_step,
               ms_per_step_median_gpu_clock=steady_ms, stage_ms=stage_ms,
               roi_queue_wait_ms=queue_wait,
               proposals_per_image_mean=float(np.mean(counts)), rois_per_step=n_rois,
               path_algorithmic_bytes_per_step=roi_bytes + rest_bytes,
               path_hbm_frac=(roi_bytes + rest_bytes) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
               roofline=dict(bound="hbm", kernel="roi_align_plane_kernel", achieved=roi_bytes / (ra_ms * 1e-3) / 1e9,
                             peak=HBM_PEAK_GBPS, unit="GB/s", frac=roi_bytes / (ra_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             traffic=traffic, traffic_source=("pmc_traffic.json (rocprofv3 --pmc WRITE_SIZE + 2 x FETCH_SIZE, "
                             "separate passes; " + str(pmc.get("source", "profiles/")) + ")") if traffic else None,
                             algorithmic_bytes_per_launch=roi_bytes, launch_ms=ra_ms,
                             dominant_stage=max(stage_ms, key=stage_ms.get)))
    return res, props, n_keep


def launch_ranks(n, argv):
    """`bench.py --gpus N` started by hand (no WORLD_SIZE in the environment): start N fresh child
    processes, one rank per GPU, BEFORE this process makes any GPU call (it never does: it only waits),
    relay rank 0's JSON line and exit non-zero if any rank fails.  Under torch.distributed.run the ranks
    already exist and this is not used."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's pipe is drained by a thread; every child is polled: as soon as one exits non-zero the others -- which may
    # sit in a collective waiting for it (RCCL gives up after ten minutes, gloo possibly never) -- are killed
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "1500"))
    failed = False
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        failed = any(c not in (None, 0) for c in codes)
        if failed or time.monotonic() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            codes = [c if c is not None else "killed" for c in codes]
            for p in procs:
                p.wait()
            failed = True
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    if failed or any(codes):
        sys.exit("bench.py: ranks exited with %r" % (codes,))


def dry_run(args, world, rank):
    """CPU rehearsal of the multi-rank protocol (no kernel runs, nothing is measured): process group over
    gloo, the barrier + max-over-ranks timing bracket around K empty steps, the final proposal gather.
    Exists so that the launcher and the rank plumbing can be tested where there is no GPU."""
    import torch.distributed as dist
    from objectdetection_amd.sharding import gather_proposals
    if world > 1:
        dist.init_process_group("gloo")
    if os.environ.get("BENCH_TEST_FAIL_RANK") == str(rank):  # test hook: this rank dies before the first collective
        sys.exit(3)
    B = args.batch
    props = torch.full((B, 8, 4), float(rank))
    n_keep = torch.full((B,), rank + 1, dtype=torch.int32)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    gather_ms = None
    n_ranks = 1
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        g0 = time.perf_counter()
        all_props, all_counts = gather_proposals(props, n_keep, dist)
        gather_ms = (time.perf_counter() - g0) * 1e3
        n_ranks = dist.get_world_size()
        assert all_props.shape[0] == n_ranks * B
        assert all_counts.tolist() == [r + 1 for r in range(n_ranks) for _ in range(B)]
    if rank == 0:
        print(json.dumps({"metric": "images/sec RPN proposal+RoIPool", "value": None, "unit": "images/s",
                          "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "gather_ms": gather_ms, "max_rank_elapsed_s": float(t.item()),
                          "data": "none (CPU rehearsal of the rank protocol over gloo; no kernel ran)"}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--workload", choices=["realistic", "gaussian"], default="realistic")
    ap.add_argument("--cpu-sample", type=int, default=4, help="distinct images timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work in the cpu_baseline sample, seconds")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workload and NMS microbenchmarks")
    ap.add_argument("--pipelined", action="store_true",
                    help="two batches in flight on two HIP streams (ProposalRoIPipeline) instead of one stream")
    ap.add_argument("--serial", action="store_true", help="one stream (the default; kept for old command lines)")
    ap.add_argument("--both-schedules", action="store_true",
                    help="also time the schedule that is not the headline (an extra in the line; off by default)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the rank protocol over gloo (launcher, barrier, gather); measures nothing")
    args = ap.parse_args()
    args.serial = not args.pipelined

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus, sys.argv[1:])
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; the launcher's world size is used" % (args.gpus, world),
              file=sys.stderr)
    if args.dry_run:
        dry_run(args, world, rank)
        return
    # BENCH_REHEARSE_ON_ONE_GPU=1: every rank on GPU 0 with gloo in place of RCCL -- the rank plumbing (image shards,
    # barrier + max-over-ranks bracket, proposal gather) around the real kernels where only one GPU is at hand; the
    # ranks share the card, so the line it prints is a rehearsal, not a measurement.
    rehearse = os.environ.get("BENCH_REHEARSE_ON_ONE_GPU") == "1"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearse:
            dist.init_process_group("gloo")
            local_rank = 0
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()  # n_gpus of the line = what RCCL reports
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from objectdetection_amd import _lib
    _lib.load()
    path = Path(dev)
    B = args.batch
    main_res, props, n_keep = run_workload(path, dev, dist, rank, world, B, args.workload, args.steps, args.warmup,
                                           pipelined=not args.serial)

    # final gather of per-image proposals over RCCL (outside the timed region)
    gather_ms = None
    if dist is not None:
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        from objectdetection_amd.sharding import gather_proposals
        if rehearse:  # gloo: the collective runs on host copies
            all_props, all_counts = gather_proposals(props.cpu(), n_keep.cpu(), dist)
        else:
            all_props, all_counts = gather_proposals(props, n_keep, dist)
        assert all_props.shape[0] == world * B
        assert all_counts.shape[0] == world * B and int(all_counts.min()) >= 0
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3

    extras = {}
    if not args.no_extras and world == 1:
        other = "gaussian" if args.workload == "realistic" else "realistic"
        sec, _, _ = run_workload(path, dev, None, 0, 1, B, other, max(5, args.steps // 5), 2, pipelined=not args.serial)
        sec.pop("roofline")
        extras["secondary_workload"] = sec
        # The other scheduling of the same steps (two batches in flight when the headline is one stream, and back) is
        # no longer part of the default line (round 3): on the driver's box it was 5 % SLOWER than one stream, and as
        # the second schedule of a process it showed a one-off 36-46 ms idle interval whose cause was not found
        # (DESIGN.md section 5).  --both-schedules brings it back for whoever wants to look.
        if args.both_schedules:
            oth, _, _ = run_workload(path, dev, None, 0, 1, B, args.workload, max(5, args.steps // 2), 2,
                                     pipelined=args.serial)
            key = "two_batches_in_flight" if args.serial else "serial"
            extras[key] = {k: oth[k] for k in ("value", "ms_per_step", "ms_per_step_median_gpu_clock", "stage_ms")}
            if oth["ms_per_step_median_gpu_clock"]:
                extras[key]["value_median_gpu_clock"] = B / oth["ms_per_step_median_gpu_clock"] * 1e3
            extras[key]["roofline_frac"] = oth["roofline"]["frac"]
        # BASELINE configs[3] (batch 256 over 8 GPUs): one rank's share, 32 images per step, on this GPU
        if B != 32:
            r32, _, _ = run_workload(path, dev, None, 0, 1, 32, args.workload, max(5, args.steps // 5), 2, pipelined=False)
            extras["configs3_rank_share_batch32"] = {k: r32[k] for k in ("value", "ms_per_step", "stage_ms")}
            extras["configs3_rank_share_batch32"]["roofline_frac"] = r32["roofline"]["frac"]
            del r32
            torch.cuda.empty_cache()
        extras.update(nms_microbench(path, dev))
        extras.update(next_rows_microbench(dev))
        extras.update(targets_microbench(path, dev))
        extras.update(roi_pool_microbench(path, dev))
        extras.update(config5_microbench(dev))
        extras["next_rows"].update(config5_backward_microbench(dev))
        extras.update(config5_microbench(dev, B=8, reps=3, levels=5))   # SURVEY's extra stress point, A = 805,707
        c64 = config5_microbench(dev, B=64, reps=3)   # BASELINE configs[4] names batch 64
        extras["config5_A200907_k30000_C256_batch64"] = c64["config5_A200907_k30000_C256"]
        extras.update(hbm_probe(dev))

    if rank == 0:
        desc = {"realistic": "proposal-realistic deltas (NMS keeps 2000/img)",
                "gaussian": "SURVEY 8d Gaussian deltas (reference decode collapses them: ~270 kept/img)"}
        line = {
            "metric": "images/sec RPN proposal+RoIPool",
            "value": main_res["value"], "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
            "ms_per_step_median_gpu_clock": main_res["ms_per_step_median_gpu_clock"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks share GPU 0 over gloo -- not a measurement)",
            "config": {"workload": "BASELINE configs[1] / SURVEY 8d config 2: batch %d x 600x1000 per GPU, 37x62 "
                                   "stride-16 map, A=%d anchors/img, top-%d -> NMS %.1f -> <=%d, RoIAlign 7x7 C=%d, "
                                   "reference (bug-compatible) decode, %s"
                                   % (B, A, PRE_NMS, NMS_THR, POST_NMS, C, desc[args.workload]),
                       "batch_per_gpu": B, "parallelism": "image-sharded x%d" % world,
                       "pipelining": "none (one stream)" if args.serial else
                       "2 HIP streams: proposal stages of batch i+1 overlap RoIAlign of batch i",
                       "proposals_per_image_mean": main_res["proposals_per_image_mean"],
                       "rois_per_step": main_res["rois_per_step"]},
            "stage_ms": main_res["stage_ms"],
            "roi_queue_wait_ms": main_res["roi_queue_wait_ms"],
            "path_algorithmic_bytes_per_step": main_res["path_algorithmic_bytes_per_step"],
            "path_hbm_frac": main_res["path_hbm_frac"],
            "roofline": main_res["roofline"],
            "gather_ms": gather_ms,
        }
        line.update(extras)
        if "hbm_measured_GBps" in extras:  # the same launch against what a plain fill achieves on this device
            line["roofline"]["frac_of_measured_fill"] = (line["roofline"]["achieved"] /
                                                         extras["hbm_measured_GBps"]["write_only_fill"])
        if args.cpu_sample > 0 and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.workload, args.cpu_seconds)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def settled_time(fn, reps, settle_ms=25.0):
    """Mean HIP-event time of `reps` back-to-back calls of fn, behind enough unrecorded calls to keep the GPU busy for
    ~settle_ms: after an idle stretch (input generation on the host) its clocks need some 20 ms to come back."""
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    one = max(e0.elapsed_time(e1), 1e-3)
    for _ in range(min(2000, int(settle_ms / one) + 1)):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def hbm_probe(dev):
    """Achievable HBM rates on this device with plain streaming kernels (SURVEY 8d asks for them next to the
    8 TB/s spec): a 2 GB write-only fill and a 1 GB -> 1 GB copy."""
    n = 512 * 1024 * 1024
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n // 2, dtype=torch.float32, device=dev)

    def timed(fn, reps=5):
        return settled_time(fn, reps)

    fill = n * 4 / timed(lambda: a.fill_(1.0)) / 1e6
    copy = n * 4 / timed(lambda: b.copy_(a[:n // 2])) / 1e6   # bytes read + written
    del a, b
    return {"hbm_measured_GBps": {"write_only_fill": fill, "copy_read_plus_write": copy}}


def next_rows_microbench(dev):
    """SURVEY 8f rows 2-4 (the callers either side of the path): time and algorithmic bytes of each launch."""
    from objectdetection_amd import ops
    from objectdetection_amd.generalized_transform import GeneralizedTransform
    g = torch.Generator().manual_seed(99)
    B = 16

    def timed(fn, reps=20):
        return settled_time(fn, reps)

    res = {}
    # row 3: batch assembly, 16 images 3x600x1000 -> max side 1000 (scale 1) -> (16,3,608,1024).  Three figures:
    # the launch alone (event bracket, preallocated batch), the whole GeneralizedTransform.forward into a reused
    # batch buffer (wall clock, synchronised), and the same allocating a fresh 120 MB batch per call -- the last
    # one depends on the state of torch's caching allocator (0.06 ms when it has a block, milliseconds when it
    # goes to hipMalloc), which is why a caller that cares passes out=.
    images = [torch.rand(3, IMG_H, IMG_W, generator=g).to(dev) for _ in range(B)]
    gt = GeneralizedTransform(600, 1000)
    batch = torch.empty((B, 3, 608, 1024), dtype=torch.float32, device=dev)
    sizes = [(IMG_H, IMG_W)] * B
    ms = timed(lambda: ops.resize_pad_batch(images, sizes, batch))

    def wall(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    wall_reuse = wall(lambda: gt(list(images), [None] * B, out=batch))
    wall_alloc = wall(lambda: gt(list(images), [None] * B))
    nbytes = B * 3 * (IMG_H * IMG_W + 608 * 1024) * 4
    res["batch_assembly"] = {"ms": ms, "algorithmic_bytes": nbytes, "GBps": nbytes / ms / 1e6,
                             "hbm_frac": nbytes / ms / 1e6 / HBM_PEAK_GBPS, "launches": (B + 31) // 32,
                             "what": "resize_pad_batch launch alone, preallocated batch, HIP events",
                             "forward_wall_ms_reused_out": wall_reuse, "forward_wall_ms_fresh_batch": wall_alloc}
    del batch
    # row 2: RPN loss tail, 256 sampled anchors per image out of A; forward + gradients in one launch (+2 memsets)
    scores = torch.randn(B, A, generator=g).to(dev)
    regs = torch.randn(B, A, 4, generator=g).to(dev)
    labels = (torch.rand(B, A, generator=g) < 0.5).float().to(dev)
    reg_labels = torch.randn(B, A, 4, generator=g).to(dev)
    index = torch.cat([torch.randperm(A, generator=g)[:256] + b * A for b in range(B)]).to(dev)
    ms = timed(lambda: ops.rpn_sampled_loss(scores, regs, labels, reg_labels, index))
    res["rpn_sampled_loss_fwd_bwd"] = {"ms": ms, "samples": int(index.numel()),
                                       "algorithmic_bytes": int(index.numel()) * (8 + 40 + 20) + B * A * 20}
    # row 2: RoI loss tail, 128 sampled proposals per image, 21 classes
    N, Kc = 128 * B, 21
    logits = torch.randn(N, Kc, generator=g).to(dev)
    pred = torch.randn(N, Kc * 4, generator=g).to(dev)
    cls_label = torch.randint(0, Kc, (N,), generator=g).float().to(dev)
    reg_label = torch.randn(N, 4, generator=g).to(dev)
    ms = timed(lambda: ops.roi_loss(logits, pred, cls_label, reg_label))
    res["roi_loss_fwd_bwd"] = {"ms": ms, "rows": N, "algorithmic_bytes": N * (Kc * 20 * 2 + 20)}
    # row 1: ROIHead eval post-process, 16 x 2,000 proposals x 21 classes (filter -> sort -> NMS), device part
    Pn, Kc2 = 2000, 21
    pr = torch.rand(B * Pn, 4, generator=g) * 400
    pr[:, 2:] += pr[:, :2] + 8
    pr = pr.to(dev)
    rg = (torch.randn(B * Pn, Kc2 * 4, generator=g) * 0.2).to(dev)
    sc = torch.softmax(torch.randn(B * Pn, Kc2, generator=g) * 3, -1).to(dev)
    row_off = torch.arange(0, (B + 1) * Pn, Pn, dtype=torch.int32, device=dev)
    sizes = torch.tensor([[float(IMG_H), float(IMG_W)]] * B, device=dev)

    def post():
        bx, s2, lb, cnt = ops.roi_detect_filter(pr, rg, sc, row_off, sizes, Pn, 0.5, 1e-3)
        vmax = 4096  # fixed candidate budget for the timing (the module reads the true maximum from the device)
        _, order = ops.topk_desc(s2[:, :vmax].contiguous(), vmax, want_values=False)
        sb = ops.gather_boxes(bx, order, cnt)
        return ops.nms_sorted(sb, cnt, 0.5, 100)

    ms = timed(post)
    res["roi_head_eval_postprocess"] = {"ms": ms, "candidates": B * Pn * Kc2,
                                        "algorithmic_bytes": B * Pn * Kc2 * 4 + B * Pn * 16}
    # RoIAlign backward at the headline shape (16 x 2,000 RoIs, C = 512): what autograd runs through the pooling when
    # the RoI losses train the backbone (roi_head.py:127-130).  Reads grad_out once, writes the gradient planes once.
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    feat = torch.randn((B, C, FH, FW), generator=g).to(dev).requires_grad_(True)
    rng = np.random.default_rng(17)
    cx, cy = rng.uniform(0, IMG_W, (B, Pn)), rng.uniform(0, IMG_H, (B, Pn))
    bw, bh = 16 * 2 ** rng.uniform(0, 5, (B, Pn)), 16 * 2 ** rng.uniform(0, 5, (B, Pn))
    props = torch.as_tensor(np.stack([(cx - bw / 2).clip(0, IMG_W), (cy - bh / 2).clip(0, IMG_H),
                                      (cx + bw / 2).clip(0, IMG_W), (cy + bh / 2).clip(0, IMG_H)], 2).astype(np.float32)).to(dev)
    cfg = ((7, 7), 2, (1.0 / 16,), 4, 4)
    gout = torch.randn((B * Pn, C, 7, 7), generator=g, dtype=torch.float32).to(dev)
    ms = timed(lambda: ops.roi_align_levels_backward(cfg, props, None, gout, [tuple(feat.shape)]), reps=5)
    nbytes = B * Pn * C * 49 * 4 + B * C * FH * FW * 4
    res["roi_align_backward_B16_K2000_C512"] = {"ms": ms, "algorithmic_bytes": nbytes, "GBps": nbytes / ms / 1e6,
                                                "hbm_frac": nbytes / ms / 1e6 / HBM_PEAK_GBPS,
                                                "kernel": "roi_align_plane_backward_kernel (MFMA form: Wy^T (G Wx) per RoI into LDS-resident planes, no atomics)"}
    del gout, feat
    # row 4: decode backward on the batch's 16 x 12,000 candidates
    n = B * PRE_NMS
    box = torch.rand(n, 4, generator=g).to(dev) * 500
    box[:, 2:] += box[:, :2] + 1
    dl = (torch.randn(n, 4, generator=g) * 0.2).to(dev)
    up = torch.randn(n, 4, generator=g).to(dev)
    ms = timed(lambda: ops.box_regression_backward(box, dl, up))
    res["decode_backward"] = {"ms": ms, "rows": n, "algorithmic_bytes": n * 80, "GBps": n * 80 / ms / 1e6}
    return {"next_rows": res}


def targets_microbench(path, dev):
    """training targets for a batch of 16.  AnchorTarget = tagging_box + box_deviation of
    A = 20,646 anchors against M = 100 boxes per image (thresholds 0.3 / 0.7); ProposalTarget = the same on
    2,000 proposals per image with class labels and weights [10,10,5,5] (thresholds 0.5 / 0.5).  One launch
    each; bytes are SURVEY 8d's algorithmic figures, pairs are N*M IoU evaluations (17 flop each)."""
    ops = path.ops
    g = torch.Generator().manual_seed(3)
    B, M, P = 16, 100, POST_NMS
    w, h = torch.rand(B, M, generator=g) * 240 + 16, torch.rand(B, M, generator=g) * 240 + 16
    x1, y1 = torch.rand(B, M, generator=g) * (IMG_W - w), torch.rand(B, M, generator=g) * (IMG_H - h)
    gt = torch.stack([x1, y1, x1 + w, y1 + h], 2).to(dev)
    labels = torch.randint(1, 21, (B, M), generator=g).float().to(dev)
    feat = torch.empty((B, 1, FH, FW), device=dev)
    anchors = path.anchors(B, feat)
    rng = np.random.default_rng(5)
    cx, cy = rng.uniform(0, IMG_W, (B, P)), rng.uniform(0, IMG_H, (B, P))
    pw, ph = 16 * 2 ** rng.uniform(0, 5, (B, P)), 16 * 2 ** rng.uniform(0, 5, (B, P))
    props = np.stack([(cx - pw / 2).clip(0, IMG_W), (cy - ph / 2).clip(0, IMG_H), (cx + pw / 2).clip(0, IMG_W),
                      (cy + ph / 2).clip(0, IMG_H)], 2).astype(np.float32)
    props = torch.as_tensor(props).to(dev)

    def timed(fn, reps=20):
        return settled_time(fn, reps)

    res = {}
    cases = {"anchor_target_A20646_M100": (lambda: ops.iou_label_encode(anchors, gt, None, 0.3, 0.7, None, None,
                                                                          want=("label", "argmax", "dev")),
                                           A, A * 16 + M * 16 + A * 24),
             "proposal_target_N2000_M100": (lambda: ops.iou_label_encode(props, gt, None, 0.5, 0.5, labels,
                                                                           [10., 10., 5., 5.],
                                                                           want=("label", "argmax", "dev")),
                                            P, P * 16 + M * 16 + P * 24)}
    for name, (fn, n, bytes_img) in cases.items():
        ms = timed(fn)
        pairs = B * n * M
        res[name] = {"ms_per_batch16": ms, "us_per_img": ms * 1e3 / B, "algorithmic_bytes_per_img": bytes_img,
                     "achieved_GBps": B * bytes_img / (ms * 1e-3) / 1e9,
                     "hbm_frac": B * bytes_img / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "pairs_per_s": pairs / (ms * 1e-3),
                     "fp32_valu_frac": pairs * 17 / (ms * 1e-3) / 157e12}
    res.update(training_halves_wall(path, dev, gt, labels, props))
    return {"training_targets_config3": res}


def training_halves_wall(path, dev, gt, labels, props):
    """Config 3 end to end, wall clock with a synchronisation either side (median of 9 after 3 untimed calls):
    `rpn_forward_train_ms` = RPN.forward in training mode on recorded head outputs (RPN/rpn.py:46-142: proposals,
    anchor targets for 16 x 20,646 anchors x 100 boxes, the 256-per-image sampling on the host RNG in the
    reference's order, both losses); `rpn_training_half_ms` = that minus the same call in eval mode;
    `roi_head_targets_ms` = ROIHead's target half alone (ROIHead/roi_head.py:99-124: targets for 16 x 2,000 proposals,
    128-per-image sampling, the three gathers).  What a user of BASELINE configs[2] waits for, not a kernel time."""
    import time
    from objectdetection_amd.RPN import RPN, AnchorGenerator
    from objectdetection_amd.ROIHead import ROIHead
    B = gt.shape[0]
    cls, reg, feat = (t.to(dev) for t in make_batch(0, B))

    class Head(torch.nn.Module):
        def forward(self, feats):
            return [cls], [reg]

    targets = [{"box": gt[b].tolist(), "class": labels[b].tolist()} for b in range(B)]
    rpn = RPN(dict(anchor_generator=AnchorGenerator(), rpn_head=Head(), num_pre_nms=PRE_NMS, num_post_nms=POST_NMS,
                   nms_threshold=NMS_THR, proposal_min_size=MIN_SIZE, pos_threshold=0.7, neg_threshold=0.3,
                   num_sample=256, pos_fraction=0.5))
    head = ROIHead(dict(pos_threshold=0.5, neg_threshold=0.5, num_sample=128, pos_fraction=0.25,
                        regression_weights=[10., 10., 5., 5.], score_thresh=0.05, proposal_min_size=1e-3,
                        nms_threshold=0.5, num_post_nms=100, roi_pooling=None, mlp_head=None, predictor=None))
    list_props = [props[b] for b in range(B)]

    def roi_targets():
        p, n, lab, rg = head._proposal_targets_padded(list_props, targets, dev)
        valid = torch.arange(p.shape[1], device=dev)[None, :] < n[:, None]
        flat, sizes = path.ops.sample_rows(torch.gt(lab, 0) & valid, torch.eq(lab, 0) & valid, 128, 0.25)
        return lab.reshape(-1)[flat], rg.reshape(-1, 4)[flat], p.reshape(-1, 4)[flat]

    def wall(fn):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(9):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts))

    info = (B, IMG_H, IMG_W)
    torch.manual_seed(0)
    t_train = wall(lambda: rpn.train()(info, [feat], targets))
    with torch.no_grad():
        t_eval = wall(lambda: rpn.eval()(info, [feat], None))
    return {"rpn_forward_train_ms": t_train, "rpn_forward_eval_ms": t_eval, "rpn_training_half_ms": t_train - t_eval,
            "roi_head_targets_ms": wall(roi_targets), "batch": B, "gt_per_image": int(gt.shape[1]),
            "host_reads_per_batch_training_half": 1}


def config5_inputs(B, dev, seed=7, levels=4):
    """800x1344, levels 13x21 .. 100x168 (A = 200,907), C = 256, tie-free logits,
    proposal-realistic deltas under the reference decode.  levels=5 adds P2 (200x336): A = 805,707, SURVEY's
    extra stress point."""
    from objectdetection_amd.RPN.anchor_generator import AnchorGenerator
    Himg, Wimg, C5 = 800, 1344, 256
    fsz = [(13, 21), (25, 42), (50, 84), (100, 168), (200, 336)][:levels]
    g = torch.Generator().manual_seed(seed)
    A5 = sum(h * w for h, w in fsz) * NA
    perm = (torch.stack([torch.randperm(A5, generator=g) for _ in range(B)]).float() / A5 - 0.5) * 8
    cls, reg, feats, off = [], [], [], 0
    for h, w in fsz:
        n = h * w * NA
        cls.append(perm[:, off:off + n].reshape(B, h, w, NA).permute(0, 3, 1, 2).contiguous().to(dev))
        off += n
        feats.append(torch.randn(B, C5, h, w, generator=g).to(dev))
    anchors = torch.cat([a[0].reshape(-1, 4) for a in AnchorGenerator()((B, Himg, Wimg), feats)], 0)
    wa, ha = (anchors[:, 2] - anchors[:, 0]).cpu(), (anchors[:, 3] - anchors[:, 1]).cpu()
    off = 0
    for h, w in fsz:
        n = h * w * NA
        cx, cy = torch.rand(B, n, generator=g) * Wimg, torch.rand(B, n, generator=g) * Himg
        # object size 16 .. 724 px, aspect ratio 1:3 .. 3:1
        size = 16 * 2 ** (torch.rand(B, n, generator=g) * 5.5)
        ratio = 2 ** ((torch.rand(B, n, generator=g) * 2 - 1) * 1.585)
        wt, ht = size * ratio.sqrt(), size / ratio.sqrt()
        w_, h_ = wa[off:off + n], ha[off:off + n]
        d = torch.stack([(cx - w_ / 2) / w_, (cy - h_ / 2) / h_, torch.log(wt / w_), torch.log(ht / h_)], 2)
        reg.append(d.reshape(B, h, w, 4 * NA).permute(0, 3, 1, 2).contiguous().to(dev))
        off += n
    return dict(cls=cls, reg=reg, feats=feats, anchors=anchors, Himg=Himg, Wimg=Wimg, A=A5, C=C5, fsz=fsz)


def config5_microbench(dev, B=16, reps=5, levels=4):
    """The whole path at SURVEY 8d config 5 (k = 30,000, P = 2,000, four-level MultiScaleRoIAlign), one
    stream; a parity-test configuration reported beside the headline, not the metric.  levels=5: with P2."""
    from objectdetection_amd import ops
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    x = config5_inputs(B, dev, levels=levels)
    pool = ROIPooling([str(i + 1) for i in range(levels)], [7, 7], 2)
    fd = {str(i + 1): f for i, f in enumerate(x["feats"][::-1])}  # fine to coarse
    rounds = ops.NmsRounds()  # (as the RPN module does: 30 rounds of which the first few work)

    def step(ev):
        def mark(n):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.append((n, e))
        mark("start")
        if os.environ.get("BENCH_FLATTEN", "0") != "1":
            tv, ti = ops.topk_desc_nchw(x["cls"], 30000)
            mark("topk")
            boxes, _, _, count = ops.decode_clip_filter_nchw(x["anchors"], x["reg"], ti, tv, x["Himg"], x["Wimg"], 1e-3)
            mark("decode_clip_filter")
        else:
            scores, deltas = ops.rpn_flatten(x["cls"], x["reg"])
            mark("flatten")
            tv, ti = ops.topk_desc(scores, 30000)
            mark("topk")
            boxes, _, _, count = ops.decode_clip_filter(x["anchors"], deltas, ti, tv, x["Himg"], x["Wimg"], 1e-3)
            mark("decode_clip_filter")
        keep, nk = ops.nms_sorted(boxes, count, 0.7, 2000, rounds=rounds)
        mark("nms")
        props = ops.gather_boxes(boxes, keep, nk)
        mark("gather")
        out, total = pool.forward_padded(fd, props, nk, [(x["Himg"], x["Wimg"])])
        mark("roi_align")
        return nk, total

    # settling steps, then the recorded ones back to back (no synchronisation in between: after an idle stretch the
    # clocks need ~20 ms to come back, see run_workload); a stage = the median over the recorded steps
    for _ in range(max(4, 24 // max(1, B // 16))):
        step([])
    evs = []
    for _ in range(reps):
        ev = []
        nk, total = step(ev)
        evs.append(ev)
    torch.cuda.synchronize()
    acc = {}
    for ev in evs:
        for (_, e0), (b, e1) in zip(ev[:-1], ev[1:]):
            acc.setdefault(b, []).append(e0.elapsed_time(e1))
    stage = {k: float(np.median(v)) for k, v in acc.items()}
    ms = sum(stage.values())
    out_bytes = int(total) * x["C"] * 49 * 4
    map_bytes = sum(f.numel() for f in x["feats"]) * 4
    res = {
        "batch": B, "ms_per_step": ms, "images_per_s": B / (ms * 1e-3), "stage_ms": stage,
        "rois_per_step": int(total), "keeps_per_img_mean": float(nk.float().mean()),
        "roi_align_write_GBps": out_bytes / (stage["roi_align"] * 1e-3) / 1e9,
        "roi_align_hbm_frac": out_bytes / (stage["roi_align"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        # reads + writes: every RoI's (C, 7, 7) block written once, every level's maps read once
        "roi_align_roofline": {"bound": "hbm", "algorithmic_bytes": out_bytes + map_bytes,
                               "achieved": (out_bytes + map_bytes) / (stage["roi_align"] * 1e-3) / 1e9,
                               "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": (out_bytes + map_bytes) / (stage["roi_align"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                               "kernel": "roi_align_tile_kernel (NCHW maps: LDS tiles, all levels in one launch)"}}
    if levels == 4 and B <= 16:
        # The pooling alone, by kernel family (settled, HIP events): LDS tiles of the NCHW maps (the default for NCHW),
        # lanes = channels (csrc/roi_cl.hip) on the same NCHW maps (copied to channel-innermost first) and on maps that
        # already are channels-last (read in place: the default for such maps).
        boxes, keep, nk2 = _config5_keeps(ops, x, rounds)
        props = ops.gather_boxes(boxes, keep, nk2)
        pool2 = ops.MultiScaleRoIAlign([str(i + 1) for i in range(levels)], (7, 7), 2)
        fl, scales, k_min, k_max = pool2._levels(fd, [(x["Himg"], x["Wimg"])])
        cfg = ((7, 7), 2, tuple(float(s_) for s_ in scales), int(k_min), int(k_max))
        fl_cl = [f.contiguous(memory_format=torch.channels_last) for f in fl]
        paths = {}
        for name, maps, pth in (("tiles_nchw_maps", fl, "tiles"), ("lanes_channels_nchw_maps", fl, "cl"),
                                ("lanes_channels_channels_last_maps", fl_cl, "cl")):
            t = settled_time(lambda: ops._roi_align_levels(cfg, props, nk2, maps, path=pth), reps=10)
            paths[name] = {"ms": t, "hbm_frac_reads_and_writes": (out_bytes + map_bytes) / t / 1e6 / HBM_PEAK_GBPS}
        res["roi_align_by_kernel_family"] = paths
        del fl_cl
    return {"config5_A%d_k30000_C256" % x["A"]: res}


def _config5_keeps(ops, x, rounds):
    tv, ti = ops.topk_desc_nchw(x["cls"], 30000)
    boxes, _, _, count = ops.decode_clip_filter_nchw(x["anchors"], x["reg"], ti, tv, x["Himg"], x["Wimg"], 1e-3)
    keep, nk = ops.nms_sorted(boxes, count, 0.7, 2000, rounds=rounds)
    return boxes, keep, nk


def config5_backward_microbench(dev, B=16, reps=3):
    """RoIAlign backward at config 5 (four FPN levels; the finest one's planes do not fit LDS and are walked window by
    window): d(sum(out * G)) / d(feature maps) through the autograd wiring of MultiScaleRoIAlign."""
    from objectdetection_amd import ops
    from objectdetection_amd.ROIHead.roi_pooling import ROIPooling
    x = config5_inputs(B, dev)
    tv, ti = ops.topk_desc_nchw(x["cls"], 30000)
    boxes, _, _, count = ops.decode_clip_filter_nchw(x["anchors"], x["reg"], ti, tv, x["Himg"], x["Wimg"], 1e-3)
    keep, nk = ops.nms_sorted(boxes, count, 0.7, 2000)
    props = ops.gather_boxes(boxes, keep, nk)
    feats = [f.requires_grad_(True) for f in x["feats"][::-1]]
    pool = ROIPooling(['1', '2', '3', '4'], [7, 7], 2)
    out, total = pool.forward_padded({str(i + 1): f for i, f in enumerate(feats)}, props, nk, [(x["Himg"], x["Wimg"])])
    G = torch.randn_like(out)

    def run():
        for f in feats:
            f.grad = None
        out.backward(G, retain_graph=True)
    ms = settled_time(run, reps)
    nbytes = int(total) * x["C"] * 49 * 4 + sum(f.numel() for f in feats) * 4
    return {"roi_align_backward_config5_B16_C256": {"ms": ms, "algorithmic_bytes": nbytes, "GBps": nbytes / ms / 1e6,
                                                    "hbm_frac": nbytes / ms / 1e6 / HBM_PEAK_GBPS}}


def roi_pool_microbench(path, dev):
    """K7, torchvision roi_pool (max) semantics -- not called by the reference (north_star: "RoI pooling to 7x7 ...
    argmax"): the headline shape, 16 x 2,000 RoIs, C = 512, 7x7, through the batched plane-scheme kernel: write
    C*49*8 B per RoI (values + int32 argmax) + one read of the feature maps.  The single-image roi-list kernel
    (2,000 RoIs) is reported beside it."""
    ops = path.ops
    rng = np.random.default_rng(11)
    B, K = 16, POST_NMS

    def boxes(n):
        cx, cy = rng.uniform(0, IMG_W, n), rng.uniform(0, IMG_H, n)
        w, h = 16 * 2 ** rng.uniform(0, 5, n), 16 * 2 ** rng.uniform(0, 5, n)
        return np.stack([(cx - w / 2).clip(0, IMG_W), (cy - h / 2).clip(0, IMG_H), (cx + w / 2).clip(0, IMG_W),
                         (cy + h / 2).clip(0, IMG_H)], 1).astype(np.float32)

    def timed(fn, reps=5):
        return settled_time(fn, reps)

    feat = torch.randn((B, C, FH, FW), generator=torch.Generator().manual_seed(5)).to(dev)
    props = torch.as_tensor(np.stack([boxes(K) for _ in range(B)])).to(dev)
    ms = timed(lambda: ops.roi_pool_batched(feat, props, None, (7, 7), 1.0 / 16))
    nbytes = B * K * C * 49 * 8 + B * C * FH * FW * 4
    res = {"roi_pool_max_B16_K2000_C512": {"ms": ms, "achieved_GBps": nbytes / (ms * 1e-3) / 1e9,
                                           "hbm_frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                           "kernel": "roi_pool_plane_kernel (frcnn_roi_pool_batched_f32)"}}
    rois = torch.cat([torch.zeros((K, 1), device=dev), props[0]], 1)
    ms = timed(lambda: ops.roi_pool(feat[:1], rois, (7, 7), 1.0 / 16))
    nbytes = K * C * 49 * 8 + C * FH * FW * 4
    res["roi_pool_max_K2000_C512"] = {"ms": ms, "achieved_GBps": nbytes / (ms * 1e-3) / 1e9,
                                      "hbm_frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                      "kernel": "roi_pool_kernel (frcnn_roi_pool_f32, roi-list form, one image)"}
    return res


def nms_box_set(kind, rng, n):
    """The NMS micro-benchmark's box sets (n boxes in score order, 600 x 1000 image).
    realistic: centres uniform, sides 16 * 2^U[0,5] -- ~95 % survive, 2,000 keeps by candidate ~2,100.
    lattice:   disjoint 4 x 4 boxes -- nothing is suppressed.
    clustered: what a trained RPN emits -- 88 % of the boxes are jittered copies of 100 objects per image (centre
               sigma 0.10 of the side, log-size sigma 0.12), the rest background as in `realistic`; at IoU 0.7 the
               2,000th keep is candidate ~7,800-8,000 of 12,000 (measured with the oracle), ~2,550 keeps in all."""
    if kind == "lattice":
        side = int(np.ceil(np.sqrt(n)))
        ys, xs = np.divmod(np.arange(n), side)
        return np.stack([xs * 5.0, ys * 5.0, xs * 5.0 + 4, ys * 5.0 + 4], 1).astype(np.float32)
    cx, cy = rng.uniform(0, IMG_W, n), rng.uniform(0, IMG_H, n)
    w, h = 16 * 2 ** rng.uniform(0, 5, n), 16 * 2 ** rng.uniform(0, 5, n)
    if kind == "clustered":
        K = 100
        ow, oh = 32 * 2 ** rng.uniform(0, 3.5, K), 32 * 2 ** rng.uniform(0, 3.5, K)
        ox, oy = rng.uniform(0, IMG_W, K), rng.uniform(0, IMG_H, K)
        k = rng.integers(0, K, n)
        obj = rng.uniform(0, 1, n) <= 0.88
        cx = np.where(obj, ox[k] + rng.normal(0, 0.10, n) * ow[k], cx)
        cy = np.where(obj, oy[k] + rng.normal(0, 0.10, n) * oh[k], cy)
        w = np.where(obj, ow[k] * np.exp(rng.normal(0, 0.12, n)), w)
        h = np.where(obj, oh[k] * np.exp(rng.normal(0, 0.12, n)), h)
    else:
        assert kind == "realistic"
    b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    b[:, 0::2] = b[:, 0::2].clip(0, IMG_W)
    b[:, 1::2] = b[:, 1::2].clip(0, IMG_H)
    return b.astype(np.float32)


def nms_microbench(path, dev):
    """'NMS ms/img at 20k anchors' (SURVEY.md 8d): NMS alone on pre-sorted boxes, batch 16, keep <= 2,000, three box
    sets -- realistic (spread over the image), bug-compatible (what the reference's decode, quirk Q1, makes of
    config 1's Gaussian deltas: everything lands on the origin) and lattice (disjoint: nothing is suppressed) --
    at n = 12,000 (the pre-NMS count), 20,646 (every anchor of the 37 x 62 map) and 30,000 (config 5's pre-NMS
    count; the bug-compatible set exists only up to the map's 20,646 anchors)."""
    ops = path.ops
    rng = np.random.default_rng(0)
    B = 16

    cls, reg, feat = (t.to(dev) for t in make_batch(0, B, "gaussian"))
    anchors = path.anchors(B, feat)
    scores, deltas = ops.rpn_flatten([cls], [reg])

    def bug_compatible(n):
        top_val, top_idx = ops.topk_desc(scores, n)
        boxes, _, _, count = ops.decode_clip_filter(anchors, deltas, top_idx, top_val, IMG_H, IMG_W, MIN_SIZE)
        return boxes, count

    res = {}
    for n in (PRE_NMS, A, 30000):
        sets = {kind: (torch.as_tensor(np.stack([nms_box_set(kind, rng, n) for _ in range(B)])).to(dev), None)
                for kind in ("realistic", "clustered", "lattice")}
        if n <= A:
            sets["bug_compatible"] = bug_compatible(n)
        for name, (boxes, count) in sets.items():
            def timed(fn, reps=10):
                for _ in range(150):   # ~20 ms of launches: the clocks are back before anything is timed
                    keep, nk = fn()
                torch.cuda.synchronize()   # (the bounded form learns from calls that have finished)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    keep, nk = fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / reps / B, nk
            every, nk = timed(lambda: ops.nms_sorted(boxes, count, NMS_THR, POST_NMS))
            rr = ops.NmsRounds()
            bounded, nk2 = timed(lambda: ops.nms_sorted(boxes, count, NMS_THR, POST_NMS, rounds=rr))
            assert torch.equal(nk, nk2)
            keep, _ = ops.nms_sorted(boxes, count, NMS_THR, POST_NMS)
            full = nk >= POST_NMS   # reach = the candidate that became the last keep (all of them when the list is short)
            reach = torch.where(full, keep[:, POST_NMS - 1].long() + 1,
                                count.long() if count is not None else torch.full_like(nk, boxes.shape[1]).long())
            entry = {"ms_per_img": bounded, "ms_per_img_every_round": every, "keeps_mean": float(nk.float().mean()),
                     "reach_candidates_mean": float(reach.float().mean()),
                     "blocks_of_64_enqueued": rr.bound((dev.index, B, boxes.shape[1], POST_NMS, float(NMS_THR)), B)[0],
                     "blocks_of_64_in_all": (boxes.shape[1] + 63) // 64}
            if name == "clustered":
                # a distribution shift: the bound learned on the easy set (2,000 keeps by candidate ~2,100), then the
                # first clustered batch -- its tail works the missing ~5,800 candidates off alone -- and the next ones
                easy = sets["realistic"][0]
                shift = ops.NmsRounds()
                for _ in range(150):
                    ops.nms_sorted(easy, None, NMS_THR, POST_NMS, rounds=shift)
                torch.cuda.synchronize()
                firsts = []
                for _ in range(3):
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                    ev[0].record()
                    k1, n1 = ops.nms_sorted(boxes, count, NMS_THR, POST_NMS, rounds=shift)
                    ev[1].record()
                    torch.cuda.synchronize()
                    firsts.append(ev[0].elapsed_time(ev[1]) / B)
                    assert torch.equal(n1, nk)
                entry["ms_per_img_first_batches_after_the_easy_set"] = firsts
            if n == PRE_NMS and name == "realistic":
                # the price of a wrong bound: one round enqueued, the tail launch finishes the batch alone
                class OneRound(ops.NmsRounds):
                    def bound(self, key, B_):
                        return 1, self._slot(key, B_)
                wr = OneRound()
                entry["ms_per_img_bound_1_tail_works"], nk3 = timed(
                    lambda: ops.nms_sorted(boxes, count, NMS_THR, POST_NMS, rounds=wr), reps=3)
                assert torch.equal(nk, nk3)
            res.setdefault("n%d" % n, {})[name] = entry
    out = {"nms_ms_per_img_batch16": res}
    out["nms_ms_per_img_n12000_batch16"] = res["n%d" % PRE_NMS]  # (the key round 1's line carried)
    return out


if __name__ == "__main__":
    main()
