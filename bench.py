#!/usr/bin/env python
"""Benchmark of the S4Former training step (DeiT-B / SETR-PUP, 512x512, synthetic data, random-init weights).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

A step = zero_grad, forward_train (EMA update, supervised branch, teacher pseudo-labels, pseudo-label CE),
backward, gradient all-reduce (N > 1), fused SGD.  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line (metric / roofline / cpu_baseline as specified by the build contract).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC for RCCL's intra-node transport (must be set before HIP initialises)
os.environ.setdefault('TORCH_NCCL_HIGH_PRIORITY', '0')   # a high-priority stream opens a FIFTH hardware queue: +8 ms per step
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}      # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md

_WD_FILE = None
TINY = dict(embed=256, layers=4, heads=4, channels=128, out_indices=(0, 1, 2, 3))   # the tests' scaled-down model
WORKLOADS = {
    # name: (n_sup, n_unsup, img, classes, model flags, description)
    'sup': (8, 0, 512, 21, dict(unsup_weight=0), 'cfg2: SETR DeiT-B PUP supervised-only bs=8 512x512 (EMA on, unsup_weight=0)'),
    'semi': (8, 8, 512, 21, dict(unsup_weight=1.0, plain_mt_pseudo_loss=True),
             'cfg3/4: S4Former mean-teacher semi 8+8 512x512 th=0.95, pseudo-label CE enabled'),
    'semi768': (4, 4, 768, 19, dict(unsup_weight=1.0, plain_mt_pseudo_loss=True),
                'cfg5: S4Former Cityscapes 768x768 4+4, pseudo-label CE enabled'),
    # not BASELINE's metric (reported in DESIGN.md only): the paper's full method, configs/setr/..._MT_w_ours.py:236-256
    'ours': (8, 8, 512, 21, dict(unsup_weight=1.0, attn_mask_seperate_head=True, attn_mask_weight=5, adaptive_attn_mask=True,
                                 use_PatchShuffle_w_Cutmix=True, PatchMix_N=8, negative_class_ranking=True,
                                 negative_class_ranking_mode='unsup_only'),
             'S4Former full method 8+8 512x512: PASA (masked + plain student pass) + CutMix / PatchShuffle + NCR'),
    # not a benchmark: the N > 1 control flow of this very script on a model that steps in milliseconds (tests/test_zz_dist_gpu.py)
    'tiny': (2, 2, 64, 21, dict(unsup_weight=1.0, plain_mt_pseudo_loss=True, **TINY), 'test-only: tiny SETR-PUP 2+2 64x64'),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='semi', choices=sorted(WORKLOADS))
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--no-parity-mode', action='store_true',
                    help='skip the secondary measurements behind the timed windows: the fp32 parity mode, the precise-teacher mode, the `ours` workload')
    ap.add_argument('--windows', type=int, default=3, help='timed windows of --steps steps each; the median window is reported')
    ap.add_argument('--timeline', default='', help='write the HIP-event timeline of the profiled step (no tracer attached) to this JSON file')
    ap.add_argument('--mask-ratio', type=float, default=0.5,
                    help='target fraction of confident teacher pixels: the randomly initialised teacher conv_seg is '
                         'rescaled (bisection, untimed) so that the pseudo-label path is not degenerate (SURVEY §7)')
    return ap.parse_args()


def usable_cores():
    """host cores this process may really use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline():
    """the oracle (CPU restatement of the reference step) timed on this box's host cores: cfg1 = DeiT-B PUP
    supervised-only semantics (EMA on), bs 2, 512x512, SGD momentum; 1 warm-up + median of 3 timed steps."""
    from oracle import model as OM
    from s4former_amd.presets import setr_pup_model, synthetic_batch
    cores = usable_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    cfg = setr_pup_model(unsup_weight=0)
    orc = OM.oracle_from_cfg(cfg)
    orc.train()
    opt = OM.build_optimizer(orc, 0.001)
    imgs, gt, metas = synthetic_batch(1999, 2, 0)
    tags = [m['tag'] for m in metas]
    times = []
    for it in range(4):
        t0 = time.time()
        OM.set_poly_lr(opt, it)
        opt.zero_grad()
        loss, _ = orc.parse_losses(orc.forward_train(imgs, tags, gt))
        loss.backward()
        opt.step()
        times.append(time.time() - t0)
    med = sorted(times[1:])[1]
    return dict(value=round(2.0 / med, 4), unit='images/s', cores=torch.get_num_threads(), kind='port',
                sample='cfg1: DeiT-B PUP sup-only (EMA on) bs=2 512x512 fp32, oracle (torch CPU restatement of the '
                       'reference step), 1 warm-up + median of 3 timed steps', seconds_per_step=round(med, 2),
                seconds_per_step_all=[round(t, 2) for t in times], torch=torch.__version__)


def calibrate_teacher(model, batch, n_sup, n_unsup, target):
    """rescale decode_head_ema.conv_seg (weights are random: max-softmax ~ 1/C, nothing would pass th=0.95) so that
    about `target` of the teacher pixels are confident. Deterministic (same seeds on every rank), untimed."""
    imgs, gt, metas = batch
    timg = imgs[n_sup + n_unsup:]
    w = model.decode_head_ema.conv_seg.weight
    base = w.detach().clone()
    lo, hi, gain = 1.0, 1e5, 1.0
    for _ in range(18):
        gain = (lo * hi) ** 0.5
        with torch.no_grad():
            w.copy_(base * gain)
            model.teacher_store.mark_dirty()
            model.ensure_engine(imgs.device)
            model.set_eval(True)
            info = model.extract_teacher_info_ema(timg, metas[n_sup + n_unsup:])
            model.set_train(True)
        r = float(info['conf_count']) / info['hard_seg_label'].numel()
        if r < target:
            lo = gain
        else:
            hi = gain
        if abs(r - target) < 0.03:
            break
    return gain, base


def main():
    args = parse()
    import s4former_amd as S
    from s4former_amd import _lib
    from s4former_amd.dist import init_distributed, setup_data_parallel
    from s4former_amd.functional import join_side_streams
    from s4former_amd.presets import MAX_ITERS, OPTIMIZER, setr_pup_model, step_gflop, synthetic_batch

    wd = os.environ.get('S4F_BENCH_WATCHDOG', '300' if int(os.environ.get('WORLD_SIZE', '1')) > 1 else '')
    if wd:
        # a stalled rank dumps every thread's stack to a per-rank FILE (a captured pipe loses it) and EXITS non-zero, so
        # that torchrun tears the peers down instead of waiting for them
        import faulthandler
        ddir = os.environ.get('S4F_WATCHDOG_DIR', os.path.join(ROOT, 'gpurun_out'))
        os.makedirs(ddir, exist_ok=True)
        global _WD_FILE
        _WD_FILE = open(os.path.join(ddir, f"watchdog_rank{os.environ.get('RANK', '0')}.txt"), 'w')
        faulthandler.dump_traceback_later(int(wd), repeat=False, file=_WD_FILE, exit=True)
    rank, local, world = init_distributed()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)'
    assert torch.cuda.is_available(), 'bench.py needs a GPU (there is no CPU fallback on the product path)'
    local_dev = local % torch.cuda.device_count()      # (gloo smoke runs may put several ranks on one GPU)
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    S.set_compute_dtype(args.dtype)

    n_sup, n_unsup, img, ncls, flags, desc = WORKLOADS[args.workload]
    bkw = dict(block=8, border=2) if args.workload == 'tiny' else {}
    torch.manual_seed(1999)                       # identical random-init weights on every rank
    model = S.build_segmentor(setr_pup_model(img=img, num_classes=ncls, **flags))
    model.init_weights()
    model.train()
    model.to(dev)
    model.log_vars_as_tensors = True              # no host sync inside the step
    opt = S.build_optimizer(model, dict(OPTIMIZER))
    opt.fused_zero_grad = os.environ.get('S4F_FUSED_ZERO_GRAD', '1') != '0'     # the SGD kernels zero the gradients they consume
    sched = S.PolyLR(opt, MAX_ITERS)

    if os.environ.get('S4F_PRETOUCH'):              # experiment: first-use order of the streams = who shares a hardware queue
        from s4former_amd.functional import pretouch_streams
        pretouch_streams(dev, [x for x in os.environ['S4F_PRETOUCH'].split(',') if x], opt)
    batches = [synthetic_batch(1999 + 17 * rank + i, n_sup, n_unsup, img=img, num_classes=ncls, device=dev, **bkw) for i in range(2)]
    reducer = setup_data_parallel(model, opt, dev)   # replicas made identical, per-range all-reduce + eager SGD hooked in

    seg_gain = 1.0
    if n_unsup:
        # every rank calibrates on the SAME batch (rank 0's): the teacher replicas stay bit-identical
        calib = batches[0] if rank == 0 else synthetic_batch(1999, n_sup, n_unsup, img=img, num_classes=ncls, device=dev, **bkw)
        seg_gain, seg_base = calibrate_teacher(model, calib, n_sup, n_unsup, args.mask_ratio)
        del calib
        if world > 1:
            # split-K fp32 atomics make the teacher logits differ in their last bits from run to run, so two ranks can leave the
            # bisection with different gains: rank 0's is the one every replica uses (the teachers stay bit-identical)
            g = torch.tensor([seg_gain], device=dev, dtype=torch.float64)
            dist.broadcast(g, 0)
            seg_gain = float(g)
            with torch.no_grad():
                model.decode_head_ema.conv_seg.weight.copy_(seg_base * seg_gain)
            model.teacher_store.mark_dirty()
            model.ensure_engine(dev)
        del seg_base

    def step(it):
        imgs, gt, metas = batches[it % 2]
        sched.step(it)
        opt.zero_grad()
        out = model.train_step(dict(img=imgs, img_metas=metas, gt_semantic_seg=gt), opt, iter=it)
        out['loss'].backward()
        join_side_streams()
        reducer.reduce_(model.student_store.grad)
        reducer.wait()
        opt.step(grad_scale=reducer.grad_scale())
        return out

    # Hardware queues: the step is fastest with exactly FOUR (GPU_MAX_HW_QUEUES, the runtime's default; measured ms per
    # step at 2 / 3 / 4 / 5 queues: 35.7 / 34.1 / 32.65 / 40.5).  A fifth queue is a cliff - and a high-priority HIP stream
    # opens one beyond the pool of four, which is what made "one more stream" cost 8 ms under a high-priority chain.
    # The critical chain (forward, input gradients, optimizer) stays on the default stream (tools/exp/prio_ab.sh, ms per
    # step on one box: default stream 32.6-32.7, a high-priority stream (S4F_MAIN_PRIORITY=1) 32.6-32.7, a created stream of
    # normal priority (=0) 33.4-33.5).  The high-priority variant is NOT used: any fifth stream of normal priority (the
    # gradient reducer's communication stream at N > 1, the eager-SGD stream) then starves: 40.5 ms.
    mode = os.environ.get('S4F_MAIN_PRIORITY', 'default')
    main = None if mode == 'default' else torch.cuda.Stream(device=dev, priority=-1 if mode == '1' else 0)
    if main is not None:
        main.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(main)
    it = 0
    for _ in range(args.warmup):
        out = step(it)
        it += 1
    schedule = None
    if world > 1:
        # round 6: the N > 1 schedule knobs (head lockstep, gradient bucket size) rest on one-rank evidence - they are timed on THIS
        # job's ranks here, untimed for the benchmark, and the fastest setting is kept (dist.autotune_schedule; every setting is
        # parity-tested at world 2).  The first three steps of the run (the warm-up's) have checked their flush order across the ranks.
        from s4former_amd.dist import autotune_schedule
        state_it = [it]

        def _tune_step():
            step(state_it[0])
            state_it[0] += 1
        schedule = autotune_schedule(model, reducer, _tune_step, steps=int(os.environ.get('S4F_AUTOTUNE_STEPS', '10')))
        it = state_it[0]
        if rank == 0 and schedule is not None:
            print('bench.py: N > 1 schedule by measurement:', json.dumps(schedule), file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    launches0 = getattr(reducer, 'launches', 0)
    # The timed region: EXACTLY K steps between a barrier + device synchronise on both sides, as the contract says - taken
    # `--windows` times back to back (default 3: three windows of 20 steps are < 2 s); `value` is the MEDIAN window, the spread
    # across the windows is reported beside it (a single 0.6 s window moved by +-1 % between runs of one binary).
    win_dt, win_host = [], []
    for _ in range(max(1, args.windows)):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step(it)
            it += 1
        win_host.append(time.perf_counter() - t0)  # time the host needed to enqueue the K steps (no sync inside)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        wdt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(wdt, op=dist.ReduceOp.MAX)
        win_dt.append(float(wdt))
    order = sorted(range(len(win_dt)), key=lambda i: win_dt[i])
    mid = order[len(order) // 2]
    dt, host_dt = win_dt[mid], win_host[mid]
    grad_collectives = (getattr(reducer, 'launches', 0) - launches0) / max(1, args.steps * len(win_dt))
    losses = {k: float(v) for k, v in out['log_vars'].items()}
    mask_ratio = float(model.last_mask_ratio) if model.last_mask_ratio is not None else None

    ips = world * (n_sup + n_unsup) * args.steps / dt
    gflop_step = step_gflop(n_sup, n_unsup, img=img, num_classes=ncls, pseudo_loss=True,
                            student_passes=2 if flags.get('attn_mask_seperate_head') else 1)
    step_tflops = gflop_step * args.steps / dt / 1e3          # per GPU

    if os.environ.get('S4F_LAYOUT_REPORT'):
        from s4former_amd.functional import check_stream_layout
        print('[layout]', os.environ.get('S4F_PRETOUCH', '-'), f'{dt / args.steps * 1e3:.3f} ms/step',
              check_stream_layout(dev, extra=[('opt', getattr(opt, '_stream', None))]), file=sys.stderr, flush=True)
    # ---- per-collective latency of the gradient exchange (N > 1): two more steps with an event pair around every all-reduce
    coll = None
    if world > 1:
        reducer.timing = True
        for _ in range(2):
            step(it)
            it += 1
        reducer.timing = False
        coll = reducer.latency_summary()
    # ---- host cost of one step, measured from an IDLE device (nothing queued: no back-pressure from a full HIP queue in it);
    # host_enqueue_ms_per_step above is the average inside the timed region, where submits can block on the queue
    torch.cuda.synchronize()
    th = time.perf_counter()
    step(it)                                       # (every rank: the step contains collectives)
    it += 1
    host_idle_ms = 1e3 * (time.perf_counter() - th)
    torch.cuda.synchronize()

    # ---- live per-kernel durations (HIP events on the launch stream) for the dominant kernel
    roofline = None
    attention = None
    kprof = None
    if not args.no_kernel_profile and rank != 0:
        for _ in range(2):
            step(it)                               # the profiled extra steps contain collectives: every rank takes part
            it += 1
    if rank == 0 and not args.no_kernel_profile:
        with _lib.CallProfiler():                  # first use of the per-kernel launch path: its one-time allocations and any
            step(it)                               # variant tuning stay out of the recorded step (a 60 ms hole in the timeline)
            it += 1
        torch.cuda.synchronize()
        with _lib.CallProfiler() as prof:
            step(it)
            it += 1
        summ = prof.summary()
        if args.timeline:
            with open(args.timeline, 'w') as f:
                json.dump([(n, list(t) if t is not None else None, st, round(t0_, 4), round(d_, 4)) for n, t, st, t0_, d_ in prof.timeline()], f)
        fam = {}
        for (name, tag), d in summ.items():
            is_gemm = name in ('s4f_gemm', 's4f_gemm_grouped') and tag is not None
            key = name if not is_gemm else f's4f_gemm[a{tag[0]},b{tag[1]}]'
            f = fam.setdefault(key, dict(calls=0, ms=0.0, gflop=0.0))
            f['calls'] += d['calls']
            f['ms'] += d['ms']
            if is_gemm:
                f['gflop'] += d['calls'] * 2.0 * tag[2] * tag[3] * tag[4] / 1e9
        shapes = {}
        for (name, tag), d in summ.items():
            if name == 's4f_gemm':
                shapes[f'a{tag[0]}b{tag[1]} M={tag[2]} N={tag[3]} K={tag[4]}'] = dict(
                    calls=d['calls'], ms=round(d['ms'], 3), tflops=round(d['calls'] * 2.0 * tag[2] * tag[3] * tag[4] / d['ms'] / 1e9, 1))
        # attention (the kernels furthest below the MFMA roofline): algorithmic flop = 4 B h N^2 64 forward (QK^T, PV), twice that
        # backward (four products: dP, dV, dQ, dK - the recomputed S is NOT counted), from the same event pairs
        att = dict(fwd_ms=0.0, fwd_gflop=0.0, bwd_ms=0.0, bwd_gflop=0.0, fwd_calls=0, bwd_calls=0)
        for (name, tag), d in summ.items():
            if tag is not None and tag[0] == 'attn':
                unit = 4.0 * tag[1] * tag[3] * tag[2] * tag[2] * 64 / 1e9
                k = 'fwd' if 'fwd' in name else 'bwd'
                att[k + '_ms'] += d['ms']
                att[k + '_gflop'] += d['calls'] * unit * (1 if k == 'fwd' else 2)
                att[k + '_calls'] += d['calls']
        total_ms = sum(f['ms'] for f in fam.values())
        kprof = {k: dict(calls=v['calls'], ms=round(v['ms'], 3), tflops=round(v['gflop'] / v['ms'], 1) if v['gflop'] else None)
                 for k, v in sorted(fam.items(), key=lambda kv: -kv[1]['ms'])}
        gemm_ms = sum(v['ms'] for k, v in fam.items() if k.startswith('s4f_gemm'))
        gemm_gflop = sum(v['gflop'] for k, v in fam.items() if k.startswith('s4f_gemm'))
        gemm_calls = sum(v['calls'] for k, v in fam.items() if k.startswith('s4f_gemm'))
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        traffic = None
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_gemm_hbm_traffic.json')))     # the latest round's passes
        tpath = tfiles[-1] if tfiles else ''
        traffic_note = None
        if args.workload == 'semi' and args.dtype == 'bf16' and tpath:
            # HBM bytes per GEMM launch from the rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this same command
            # (separate --pmc runs, FETCH_SIZE doubled for gfx950); bench.py cannot run the profiler on itself.  The file names
            # the kernel sources it was measured on: a profile that predates a kernel change is NOT reported.
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            from pmc_traffic import kernel_src_sha
            tj = json.load(open(tpath))
            if tj.get('kernel_src_sha') == kernel_src_sha():
                traffic = round(tj['hbm_bytes_per_launch'])
            else:
                traffic_note = (f'{os.path.basename(tpath)} was measured on other GEMM kernel sources (sha {tj.get("kernel_src_sha")} '
                                f'vs {kernel_src_sha()}): re-run tools/profile_round.sh; traffic not reported')
                print('bench.py: STALE TRAFFIC PROFILE - ' + traffic_note, file=sys.stderr, flush=True)
        # ---- what the event pairs can and cannot say.  The launches sit on 4 - 5 streams that overlap in time: the SUM of the GEMM
        # durations can exceed the step (it did in round 5: 28.67 ms of GEMM events in a 28.26 ms step).  Reported beside it, from the
        # same profiled step: the busy time of every stream (union of its launch intervals: each <= the step by construction) and the
        # wall time during which at least one GEMM-family launch was running.
        def union_ms(iv):
            tot, end = 0.0, None
            for a0, a1 in sorted(iv):
                if end is None or a0 > end:
                    tot, end = tot + (a1 - a0), a1
                elif a1 > end:
                    tot, end = tot + (a1 - end), a1
            return tot
        tl = prof.timeline()
        by_stream, gemm_iv = {}, []
        for name_, tag_, st_, t0_, d_ in tl:
            g_ = name_ in ('s4f_gemm', 's4f_gemm_grouped')
            q = by_stream.setdefault(st_, dict(all=[], gemm=[]))
            q['all'].append((t0_, t0_ + d_))
            if g_:
                q['gemm'].append((t0_, t0_ + d_))
                gemm_iv.append((t0_, t0_ + d_))
        span = (max(t0_ + d_ for _, _, _, t0_, d_ in tl) - min(t0_ for _, _, _, t0_, _ in tl)) if tl else 0.0
        queues = sorted((dict(busy_ms=round(union_ms(q['all']), 3), gemm_busy_ms=round(union_ms(q['gemm']), 3), launches=len(q['all']))
                         for q in by_stream.values()), key=lambda d: -d['busy_ms'])
        # ---- the same GEMM family from the round's rocprofv3 files (tools/profile_round.sh -> tools/gemm_time_profile.py), when they
        # were measured on THESE kernel sources: under the tracer on the default streams, and alone on the chip (one stream)
        traced = serial = None
        gfiles = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_gemm_kernel_time.json')))
        if args.workload == 'semi' and args.dtype == 'bf16' and gfiles:
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            from pmc_traffic import kernel_src_sha as _sha
            gj = json.load(open(gfiles[-1]))
            if gj.get('kernel_src_sha') == _sha():
                src_ = 'profiles/' + os.path.basename(gfiles[-1])
                traced = dict(achieved=round(gemm_gflop / gj['traced']['gemm_ms_per_step'], 1), frac=round(gemm_gflop / gj['traced']['gemm_ms_per_step'] / peak, 4),
                              gemm_kernel_ms_per_step=round(gj['traced']['gemm_ms_per_step'], 3), source=src_ + ' (rocprofv3 --kernel-trace --stats of this command)')
                serial = dict(achieved=round(gemm_gflop / gj['serial']['gemm_ms_per_step'], 1), frac=round(gemm_gflop / gj['serial']['gemm_ms_per_step'] / peak, 4),
                              gemm_kernel_ms_per_step=round(gj['serial']['gemm_ms_per_step'], 3), step_ms_on_one_stream=gj['serial']['step_ms'],
                              source=src_ + ' (the same step on ONE stream: every kernel alone on the chip)')
        roofline = dict(bound='mfma', kernel='s4f_gemm family: g2::gemm2_kernel / g5::gemm5_kernel / g6::gemm6_kernel (dense + implicit-GEMM conv, all launches of a step)',
                        achieved=round(gemm_gflop / gemm_ms, 1), peak=peak, unit='TFLOP/s',
                        frac=round(gemm_gflop / gemm_ms / peak, 4), traffic=traffic,
                        gemm_event_ms_sum=round(gemm_ms, 3), gemm_wall_ms=round(union_ms(gemm_iv), 3), profiled_step_span_ms=round(span, 3),
                        queues=queues, traced=traced, serial=serial,
                        duration_note='achieved / frac = GEMM GFLOP / SUM of the per-launch HIP-event durations; the launches of different streams '
                                      'overlap, so that sum (gemm_event_ms_sum) is a contention-inflated total and may exceed ms_per_step; queues[] = busy '
                                      'time per stream of the same profiled step (each <= its span), gemm_wall_ms = time with at least one GEMM running; '
                                      'traced / serial = the same family from the rocprofv3 files of this command',

                        traffic_source=('profiles/' + os.path.basename(tpath)) if traffic is not None else traffic_note,
                        algorithmic_gflop_per_launch=round(gemm_gflop / gemm_calls, 2),
                        launches_per_step=gemm_calls, avg_launch_ms=round(gemm_ms / gemm_calls, 4),
                        share_of_step_kernel_time=round(gemm_ms / total_ms, 3),
                        # which launch path the per-GEMM durations were taken on: one extra step AFTER the timed windows with
                        # an event pair around every C-ABI launch - the encoder layers therefore on their per-kernel path
                        # (LayerFn), not behind s4f_encoder_layer_fwd / _bwd as in the timed steps (same kernels, same streams;
                        # tests/test_step_gpu.py::test_launch_paths_of_the_encoder_layer_agree holds the two paths together)
                        launch_path='per-kernel C-ABI launches under _lib.CallProfiler (one profiled step after the timed windows); '
                                    'the timed steps issue each encoder layer through s4f_encoder_layer_fwd / _bwd',
                        step=dict(achieved=round(step_tflops, 1), frac=round(step_tflops / peak, 4),
                                  gflop_per_step_per_gpu=round(gflop_step, 1)))
        if att['fwd_ms'] > 0 and att['bwd_ms'] > 0:
            attention = dict(fwd_tflops=round(att['fwd_gflop'] / att['fwd_ms'], 1), bwd_tflops_algorithmic=round(att['bwd_gflop'] / att['bwd_ms'], 1),
                             frac=round((att['fwd_gflop'] + att['bwd_gflop']) / (att['fwd_ms'] + att['bwd_ms']) / peak, 4),
                             fwd_frac=round(att['fwd_gflop'] / att['fwd_ms'] / peak, 4), bwd_frac=round(att['bwd_gflop'] / att['bwd_ms'] / peak, 4),
                             fwd_calls=att['fwd_calls'], bwd_calls=att['bwd_calls'], ms_per_step=round(att['fwd_ms'] + att['bwd_ms'], 3),
                             note='flash-style kernels at head dim 64 (attn_fwd2, one-sweep backward with its pre / post passes): '
                                  'forward 4 B h N^2 64 flop, backward 8 B h N^2 64 (four products; the recomputed scores are not counted), '
                                  'durations from the same HIP-event pairs as roofline')

    # ---- secondary measurements on the SAME box, after the timed windows (rank 0, one GPU): each builds its own model / optimizer,
    # runs `warm` untimed + `nsteps` timed steps of the same step function, and is wrapped so that whatever goes wrong in it (out of
    # memory on a smaller card, a failing launch) is recorded in the line and never costs the headline number measured above.
    def fresh_run(workload, dtype_name, precise, warm, nsteps):
        from s4former_amd import runtime as RT
        n_sup_, n_unsup_, img_, ncls_, flags_, _ = WORKLOADS[workload]
        S.set_compute_dtype(dtype_name)
        RT.set_teacher_precise(precise)
        try:
            torch.manual_seed(1999)
            m_ = S.build_segmentor(setr_pup_model(img=img_, num_classes=ncls_, **flags_))
            m_.init_weights()
            m_.train()
            m_.to(dev)
            m_.log_vars_as_tensors = True
            o_ = S.build_optimizer(m_, dict(OPTIMIZER))
            o_.fused_zero_grad = os.environ.get('S4F_FUSED_ZERO_GRAD', '1') != '0'
            sc_ = S.PolyLR(o_, MAX_ITERS)
            r_ = setup_data_parallel(m_, o_, dev)
            if n_unsup_:
                with torch.no_grad():
                    m_.decode_head_ema.conv_seg.weight.mul_(seg_gain)      # the gain the bf16 teacher was calibrated to
            bt_ = batches if (n_sup_, n_unsup_, img_, ncls_) == (n_sup, n_unsup, img, ncls) else \
                [synthetic_batch(1999 + i, n_sup_, n_unsup_, img=img_, num_classes=ncls_, device=dev) for i in range(2)]

            def one(i):
                imgs_, gt_, metas_ = bt_[i % 2]
                sc_.step(i)
                o_.zero_grad()
                out_ = m_.train_step(dict(img=imgs_, img_metas=metas_, gt_semantic_seg=gt_), o_, iter=i)
                out_['loss'].backward()
                join_side_streams()
                r_.reduce_(m_.student_store.grad)
                r_.wait()
                o_.step(grad_scale=r_.grad_scale())
                return out_
            for i in range(warm):
                one(i)
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for i in range(nsteps):
                out_ = one(warm + i)
            torch.cuda.synchronize()
            sec = (time.perf_counter() - t0_) / nsteps
            gf = step_gflop(n_sup_, n_unsup_, img=img_, num_classes=ncls_, pseudo_loss=True,
                            student_passes=2 if flags_.get('attn_mask_seperate_head') else 1)
            res = dict(ms_per_step=round(1e3 * sec, 2), images_per_s=round((n_sup_ + n_unsup_) / sec, 2), steps=nsteps, warmup=warm,
                       tflops=round(gf / sec / 1e3, 1), gflop_per_step=round(gf, 1),
                       mask_ratio=float(m_.last_mask_ratio) if m_.last_mask_ratio is not None else None,
                       losses={k: round(float(v), 6) for k, v in out_['log_vars'].items()})
            del m_, o_, sc_, r_, out_
            torch.cuda.empty_cache()
            return res
        finally:
            RT.set_teacher_precise(False)
            S.set_compute_dtype(args.dtype)

    def guarded(fn, *a):
        try:
            return fn(*a)
        except Exception as e:      # noqa: BLE001
            try:
                torch.cuda.synchronize()
            except Exception:       # noqa: BLE001
                pass
            return dict(error=f'{type(e).__name__}: {e}'[:300])

    parity_mode = None
    secondary = None
    fused_zero_flag = bool(getattr(opt, 'fused_zero_grad', False))
    stream_layout = getattr(reducer, 'stream_layout', None) if world > 1 else None
    if rank == 0 and world == 1 and args.dtype == 'bf16' and not args.no_parity_mode:
        del out
        model = opt = sched = reducer = None
        torch.cuda.empty_cache()
        # (a) the numeric mode that meets north_star's parity clause on every count (fp32 MFMA chain: losses 1e-4, pseudo-label masks
        # equal outside the reference's tie set), timed on the SAME workload: what that clause costs
        parity_mode = guarded(fresh_run, args.workload, 'fp32', False, 2, 3)
        if 'error' not in parity_mode:
            parity_mode.update(dtype='fp32', frac_of_fp32_mfma_peak=round(parity_mode['tflops'] / MFMA_PEAK_TFLOPS['fp32'], 4),
                               note='same workload in the fp32 parity mode (v_mfma_f32_16x16x4_f32 chains, fp32 everything): the mode whose losses '
                                    'meet the goldens of the reference to 1e-4 and whose pseudo-label masks differ from it only inside its tie set '
                                    '(tests/test_fullsize_gpu.py); value / ms_per_step above are the bf16 perf mode')
        # (b) round 6, the PRECISE TEACHER (S4F_TEACHER_PRECISE=1): student bf16, teacher pass on the fp32 parity kernels - the
        # pseudo-label masks and PASA confidences are then the parity mode's (reference: encoder_decoder.py:888-901 is fp32)
        if n_unsup:
            tp = guarded(fresh_run, args.workload, 'bf16', True, 3, 5)
            if 'error' not in tp:
                tp.update(dtype='bf16 student + fp32 teacher', switch='S4F_TEACHER_PRECISE=1',
                          parity='pseudo-labels equal to the reference outside its tie set (8 of 2,097,152 on cfg3\'s own batch, all inside), named '
                                 'losses of the first iteration within 1e-4 (6.8e-5 / 9.1e-5): tests/test_fullsize_gpu.py::'
                                 'test_precise_teacher_gives_the_reference_pseudo_labels_in_the_bf16_mode, profiles/r06_parity_report.json')
            parity_mode['teacher_precise'] = tp
        # (c) the one reference config that reaches compute_pseudo_loss as written (configs/setr/..._MT_w_ours.py:236-256: PASA with its
        # masked AND plain student pass, CutMix / PatchShuffle, NCR): `--workload ours`, pinned at DeiT-B size by tests/golden/full_ours.npz
        if args.workload == 'semi':
            secondary = guarded(fresh_run, 'ours', 'bf16', False, 8, 10)      # (8 untimed steps: GEMM signatures of the 24-image student batch are tuned on first use)
            if 'error' not in secondary:
                secondary.update(workload='ours: ' + WORKLOADS['ours'][5], dtype='bf16',
                                 frac_of_bf16_mfma_peak=round(secondary['tflops'] / MFMA_PEAK_TFLOPS['bf16'], 4),
                                 pinned_by='tests/golden/full_ours.npz (the reference\'s own code, two iterations at DeiT-B size): '
                                           'tests/test_fullsize_gpu.py::test_fullsize_step_vs_reference_golden[full_ours-fp32|bf16]; measured margins '
                                           'in profiles/r06_parity_report.json (deit_b/full_ours/*)')

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline()
        except Exception as e:          # noqa: BLE001
            cpu = dict(error=f'{type(e).__name__}: {e}'[:300])

    if rank == 0:
        line = dict(metric='train images/sec DeiT-B 512x512 S4Former step', value=round(ips, 3), unit='images/s',
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 3),
                    ms_per_step_windows=[round(1e3 * w / args.steps, 3) for w in win_dt],
                    higher_is_better=True, scaling='weak', vs_baseline=None, dtype=args.dtype, data='synthetic',
                    config=dict(workload=f'{args.workload}: {desc}', images_per_step_per_gpu=n_sup + n_unsup,
                                crop=f'{img}x{img}', classes=ncls, parallelism=f'dp{world}', weights='random-init DeiT-B',
                                teacher_conv_seg_gain=round(seg_gain, 2),
                                pinned_by=('the pseudo-label CE on the plain mean-teacher branch is an extension flag (plain_mt_pseudo_loss; the '
                                           'reference step yields no unsupervised loss there, SURVEY Q1): this flow is pinned against '
                                           'the oracle on the tiny model (tests/test_step_gpu.py::test_plain_mt_pseudo_loss_vs_oracle); the '
                                           'reference goldens full_pasa / full_semi4 / full_semi8_fwd cover a superset of its kernels at DeiT-B size')
                                if flags.get('plain_mt_pseudo_loss') else 'reference goldens (tests/golden/full_*.npz)',
                                dist_backend=dist.get_backend() if world > 1 else None,
                                ranks_seen=dist.get_world_size() if world > 1 else 1,
                                grad_collectives_per_step=round(grad_collectives, 2) if grad_collectives else None,
                                grad_collective_latency=coll, schedule=schedule,
                                flush_order_checked_steps=(int(os.environ.get('S4F_CHECK_FLUSH', '3') or 0) if world > 1 else None),
                                fused_zero_grad=fused_zero_flag,
                                stream_layout=stream_layout),
                    roofline=roofline, attention=attention, parity_mode=parity_mode, secondary=secondary, cpu_baseline=cpu, losses=losses, mask_ratio=mask_ratio,
                    host_enqueue_ms_per_step=round(1e3 * host_dt / args.steps, 3),
                    host_enqueue_idle_queue_ms=round(host_idle_ms, 3))
        if kprof is not None:
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            with open(os.path.join(ROOT, 'gpurun_out', f'bench_kernels_{args.workload}_{args.dtype}.json'), 'w') as f:
                json.dump(dict(families=kprof, gemm_shapes=dict(sorted(shapes.items(), key=lambda kv: -kv[1]['ms']))), f, indent=1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
