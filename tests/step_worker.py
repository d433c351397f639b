"""Worker of tests/test_switch_matrix_gpu.py: two iterations of the tiny S4Former step (one golden fixture, one numeric mode) in a
FRESH process, so that the package's import-time switches (S4F_* read at module level) take the values of the caller's environment.
Writes losses, per-parameter gradient norms and the final state sums to <out>.npz.

  python tests/step_worker.py --name mt_ours --dtype bf16 --out /tmp/x.npz"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--name', default='mt_ours')
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--out', required=True)
    args = ap.parse_args()
    import numpy as np
    from tests import test_step_gpu as T
    z, meta = T.load_gold(args.name)
    model, opt, sched = T.build_product(meta, args.dtype)
    rec = T.run_product(model, opt, sched, meta)
    sd = model.state_dict()
    out = {}
    for it in range(2):
        keys = sorted(k for k in rec[it]['log'])
        out[f'it{it}_loss_keys'] = np.array(keys)
        out[f'it{it}_loss_vals'] = np.array([float(rec[it]['log'][k]) for k in keys], dtype=np.float64)
        gk = sorted(rec[it]['gn'])
        out[f'it{it}_gn_keys'] = np.array(gk)
        out[f'it{it}_gn_vals'] = np.array([rec[it]['gn'][k] for k in gk], dtype=np.float64)
    fk = [str(k) for k in z['final_sha_keys']]
    out['final_keys'] = np.array(fk)
    out['final_abs_sum'] = np.array([float(sd[k].double().abs().sum()) for k in fk])
    out['env'] = np.array(str({k: v for k, v in os.environ.items() if k.startswith('S4F_')}))
    np.savez(args.out, **out)


if __name__ == '__main__':
    main()
