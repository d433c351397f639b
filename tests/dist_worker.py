"""Worker of tests/test_zz_dist_gpu.py: two training steps of the tiny S4Former model through the PRODUCT's data-parallel
path (dist.setup_data_parallel: replica broadcast, per-range gradient all-reduce during backward, eager SGD behind it,
SyncBN statistics exchange, batched log scalars), one process per rank.

  world 1 (plain `python tests/dist_worker.py ...`):   the whole global batch in one process = the expected result
  world N (torch.distributed.run):                     rank r takes images r*n/N .. (r+1)*n/N of every tag group

Reference semantics checked by the caller (mmseg/apis/train.py:129-138, segmentors/base.py:257-272):
N-rank losses (mean over ranks) == 1-rank losses of the concatenated batch, SyncBN == BN over the concatenated batch
including the running statistics, replicas bit-identical after the step.

Every rank writes <out>/rank<r>.npz; a stalled rank dumps its stacks to <out>/watchdog_rank<r>.txt and exits non-zero."""
import argparse
import faulthandler
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--dtype', default='fp32')
    ap.add_argument('--flags', default='pasa', choices=['pasa', 'plain'])
    ap.add_argument('--n-sup', type=int, default=4)        # GLOBAL batch (split over the ranks)
    ap.add_argument('--n-unsup', type=int, default=4)
    ap.add_argument('--iters', type=int, default=2)
    ap.add_argument('--watchdog', type=int, default=90)
    ap.add_argument('--autotune', action='store_true',
                    help='after the recorded steps: dist.autotune_schedule (the N > 1 schedule picked by measurement), then the replica check again')
    ap.add_argument('--rccl-one-rank', action='store_true',
                    help='a process group of ONE rank over backend nccl (= RCCL) with every collective of the data path forced on')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    rank = int(os.environ.get('RANK', '0'))
    wd = open(os.path.join(args.out, f'watchdog_rank{rank}.txt'), 'w')
    faulthandler.dump_traceback_later(args.watchdog, repeat=False, file=wd, exit=True)

    import numpy as np
    import torch
    import torch.distributed as dist
    import s4former_amd as S
    from s4former_amd.dist import init_distributed, setup_data_parallel
    from s4former_amd.functional import join_side_streams
    from tests import common as C

    rank, local, world = init_distributed(timeout_s=60)
    issued = dict(grad=0, bn=0)
    if args.rccl_one_rank:
        # RCCL needs one GPU per rank, so on a one-GPU box its API path (ProcessGroupNCCL's enqueue on its own stream, async work
        # handles waited inside backward, in-place all-reduce of arena views / the SyncBN buffers / the packed log scalars) is
        # exercised with a group of one: the all-reduce is the identity, the result must equal the plain run's.
        import datetime
        import s4former_amd.dist as D
        import s4former_amd.functional as F_
        assert world == 1
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29640')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(0)
        try:
            dist.init_process_group('nccl', rank=0, world_size=1, timeout=datetime.timedelta(seconds=60))
            probe = torch.ones(8, device='cuda')
            dist.all_reduce(probe)
            torch.cuda.synchronize()
        except Exception as e:                      # noqa: BLE001 - no usable RCCL on this box: the caller skips
            print(f'RCCL-UNAVAILABLE {type(e).__name__}: {e}', flush=True)
            sys.exit(77)
        D.collectives_active = lambda: True
        issue0 = D.GradReducer.issue

        def _issue(t):
            issued['grad'] += 1
            return issue0(t)
        D.GradReducer.issue = staticmethod(_issue)

        def _reduce(self):
            if self.buf is not None:
                issued['bn'] += 1
                self.issue(self.buf)
            self.buf = None
        F_._Exchange.reduce = _reduce
    dev = torch.device('cuda', local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    S.set_compute_dtype(args.dtype)
    flags = dict(unsup_weight=1.0, attn_mask_seperate_head=True, attn_mask_weight=5, adaptive_attn_mask=True) \
        if args.flags == 'pasa' else dict(unsup_weight=1.0, plain_mt_pseudo_loss=True)
    model = S.build_segmentor(C.tiny_model_cfg(**flags))
    model.train()
    # every rank starts from DIFFERENT values: the broadcast of setup_data_parallel must make them rank 0's
    C.load_filled(model, 1999 + 7 * rank, 60.0)
    model.to(dev)
    opt = S.build_optimizer(model, dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0,
                                        paramwise_cfg=dict(custom_keys={'head': dict(lr_mult=10.)})))
    sched = S.PolyLR(opt, 80001)
    reducer = setup_data_parallel(model, opt, dev)

    ns, nu = args.n_sup, args.n_unsup
    assert ns % world == 0 and nu % world == 0
    a, b = ns // world, nu // world
    rec = {}
    for it in range(args.iters):
        imgs, gt, metas = C.make_batch(4242 + it, ns, nu)
        idx = list(range(rank * a, (rank + 1) * a)) + [ns + i for i in range(rank * b, (rank + 1) * b)] + \
            [ns + nu + i for i in range(rank * b, (rank + 1) * b)]
        imgs, gt, metas = imgs[idx].to(dev), gt[idx].to(dev), [metas[i] for i in idx]
        sched.step(it)
        opt.zero_grad()
        out = model.train_step(dict(img=imgs, img_metas=metas, gt_semantic_seg=gt), opt, iter=it)
        out['loss'].backward()
        join_side_streams()
        reducer.reduce_(model.student_store.grad)
        reducer.wait()
        torch.cuda.synchronize()
        # gradients as the optimiser sees them (sum over ranks, scaled by 1/world) - unless already consumed by eager SGD
        rec[f'it{it}_loss_keys'] = np.array(list(out['log_vars'].keys()))
        rec[f'it{it}_loss_vals'] = np.array([float(v) for v in out['log_vars'].values()], dtype=np.float64)
        rec[f'it{it}_local_loss'] = np.float64(float(out['loss']))
        rec[f'it{it}_mask_ratio'] = np.float64(float(model.last_mask_ratio)) if model.last_mask_ratio is not None else np.float64(-1)
        opt.step(grad_scale=reducer.grad_scale())
    torch.cuda.synchronize()
    sd = model.state_dict()
    keys = [k for k, v in sd.items() if v.dtype == torch.float32]
    rec['state_keys'] = np.array(keys)
    rec['state_abs_sum'] = np.array([float(sd[k].double().abs().sum()) for k in keys])
    rec['state_sum'] = np.array([float(sd[k].double().sum()) for k in keys])
    rec['student_sha'] = np.array(hashlib.sha256(model.student_store.flat.cpu().numpy().tobytes()).hexdigest())
    rec['teacher_sha'] = np.array(hashlib.sha256(model.teacher_store.flat.cpu().numpy().tobytes()).hexdigest())
    rec['mom_sha'] = np.array(hashlib.sha256(model.student_store.mom.cpu().numpy().tobytes()).hexdigest())
    rec['nbt'] = np.array([int(v) for k, v in sd.items() if k.endswith('num_batches_tracked')])
    if args.autotune:
        # round 6: the warm-up auto-selection of the N > 1 schedule (head lockstep, bucket size) on this very model, every rank the
        # same number of steps; afterwards the replicas must still be bit-identical and agree on what was chosen
        from s4former_amd.dist import autotune_schedule
        state = dict(it=args.iters)

        def run_step():
            it = state['it']
            state['it'] += 1
            imgs, gt, metas = C.make_batch(4242 + it % 2, ns, nu)
            idx = list(range(rank * a, (rank + 1) * a)) + [ns + i for i in range(rank * b, (rank + 1) * b)] + \
                [ns + nu + i for i in range(rank * b, (rank + 1) * b)]
            imgs, gt, metas = imgs[idx].to(dev), gt[idx].to(dev), [metas[i] for i in idx]
            sched.step(it)
            opt.zero_grad()
            out = model.train_step(dict(img=imgs, img_metas=metas, gt_semantic_seg=gt), opt, iter=it)
            out['loss'].backward()
            join_side_streams()
            reducer.reduce_(model.student_store.grad)
            reducer.wait()
            opt.step(grad_scale=reducer.grad_scale())
            state['loss'] = float(out['loss'])
        sch = autotune_schedule(model, reducer, run_step, steps=2)
        torch.cuda.synchronize()
        rec['schedule'] = np.array(json.dumps(sch))
        rec['tuned_steps'] = np.int64(state['it'] - args.iters)
        rec['tuned_last_loss'] = np.float64(state.get('loss', float('nan')))
        rec['tuned_student_sha'] = np.array(hashlib.sha256(model.student_store.flat.cpu().numpy().tobytes()).hexdigest())
        rec['tuned_teacher_sha'] = np.array(hashlib.sha256(model.teacher_store.flat.cpu().numpy().tobytes()).hexdigest())
        rec['tuned_mom_sha'] = np.array(hashlib.sha256(model.student_store.mom.cpu().numpy().tobytes()).hexdigest())
        rec['check_steps_left'] = np.int64(reducer.check_steps)
    rec['issued'] = np.array([issued['grad'], issued['bn']])
    rec['stream_layout'] = np.array(json.dumps(getattr(reducer, 'stream_layout', None)))
    rec['meta'] = np.array(json.dumps(dict(world=world, rank=rank, backend=dist.get_backend() if dist.is_initialized() else None,
                                           env={k: v for k, v in os.environ.items() if k.startswith('S4F_')})))
    np.savez(os.path.join(args.out, f'rank{rank}.npz'), **rec)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()
    wd.close()


if __name__ == '__main__':
    main()
