"""Build recipe of libs4f_hip.so (hipcc, gfx950 only).  `python -m s4former_amd.build` or `build_library()`.

The shared library is built in-tree (s4former_amd/libs4f_hip.so) so that it travels to the GPU box with
the repo snapshot; objects are cached under s4former_amd/csrc/_obj keyed by a hash of source + flags.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
OBJ_DIR = os.path.join(CSRC, '_obj')
LIB_PATH = os.environ.get('S4F_LIB_OUT') or os.path.join(PKG_DIR, 'libs4f_hip.so')      # S4F_LIB_OUT: experiment builds beside the shipped one
SOURCES = ['gemm.hip', 'gemm2.hip', 'gemm5.hip', 'gemm6.hip', 'attention.hip', 'attn_bwd.hip', 'elementwise.hip', 'head.hip', 'eval.hip', 'pipeline.hip', 'layer.hip']
HEADERS = ['common.h', os.path.join('..', '..', 'include', 's4f.h')]
BASE_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-ffp-contract=off',
              '-Wno-unused-result', '-Wno-unused-value']
FLAGS = BASE_FLAGS + ['-mllvm', '-amdgpu-mfma-vgpr-form=1']
FILE_FLAGS = {}      # per-file overrides
# attn_bwd.hip: no SLP packing of the per-score fp32 work (v_pk_mul_f32 + the moves that pair its operands cost more issue
# cycles beside MFMAs than two scalar multiplies: MI355X_MICROARCH, 'price of one filler beside MFMAs')
FILE_FLAGS['attn_bwd.hip'] = FLAGS + ['-fno-slp-vectorize']
if os.environ.get('S4F_FB_STAMPS'):   # diagnostic: s_memtime stamps around the pipeline pieces (tools/exp/fb_stamps.py)
    FILE_FLAGS['attn_bwd.hip'] = FILE_FLAGS['attn_bwd.hip'] + ['-DFB_STAMPS']
if os.environ.get('S4F_FB_ABL'):      # ablation builds of the one-sweep attention backward (tools/exp/fb_ablate.sh)
    FILE_FLAGS['attn_bwd.hip'] = FILE_FLAGS['attn_bwd.hip'] + ['-DFB_ABL=' + os.environ['S4F_FB_ABL']]
if os.environ.get('S4F_G5P_AUTO'):    # A/B build: S4F_G5P_AUTO=0 keeps tile_hint 10 on the one-tile kernel
    FILE_FLAGS['gemm5.hip'] = FILE_FLAGS.get('gemm5.hip', FLAGS) + ['-DG5P_AUTO=' + os.environ['S4F_G5P_AUTO']]
if os.environ.get('S4F_G5_ST_AUX'):  # A/B build: cache policy bits of the ping-pong GEMM's bf16 output stores (16 = sc1)
    FILE_FLAGS['gemm5.hip'] = FILE_FLAGS.get('gemm5.hip', FLAGS) + ['-DG5_ST_AUX=' + os.environ['S4F_G5_ST_AUX']]
if os.environ.get('S4F_G5_PROBES'):
    FILE_FLAGS['gemm5.hip'] = FILE_FLAGS.get('gemm5.hip', FLAGS) + ['-DG5_PROBES']
if os.environ.get('S4F_EXTRA_DEFS'):  # experiment builds: extra -D switches for every source (space separated, e.g. "FB_PRE_ORDER=1")
    for _f in ('gemm.hip', 'gemm2.hip', 'gemm5.hip', 'gemm6.hip', 'attention.hip', 'attn_bwd.hip', 'elementwise.hip', 'head.hip',
               'eval.hip', 'pipeline.hip', 'layer.hip'):
        FILE_FLAGS[_f] = FILE_FLAGS.get(_f, FLAGS) + ['-D' + d for d in os.environ['S4F_EXTRA_DEFS'].split()]
_DIAG = [v for v in ('S4F_FB_STAMPS', 'S4F_FB_ABL', 'S4F_G5_PROBES', 'S4F_G5P_AUTO', 'S4F_G5_ST_AUX', 'S4F_EXTRA_DEFS') if os.environ.get(v)]
if _DIAG and not os.environ.get('S4F_LIB_OUT'):
    # stamp / ablation builds compute wrong results by construction: they never overwrite the shipped library
    raise RuntimeError(f'{", ".join(_DIAG)} select a diagnostic build: set S4F_LIB_OUT=<path of the experiment library> as well')
FILE_DEPS = {'gemm5.hip': ['gemm2.hip'], 'gemm6.hip': ['gemm2.hip']}


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _digest(paths, extra):
    h = hashlib.sha256()
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(extra).encode())
    return h.hexdigest()[:16]


def build_library(verbose=False, force=False):
    """Compile every HIP source for gfx950 and link libs4f_hip.so. Returns the library path."""
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        flags = FILE_FLAGS.get(src, FLAGS)
        deps = [os.path.join(CSRC, x) for x in FILE_DEPS.get(src, [])]
        tag = _digest([sp] + deps + hdrs, flags)
        obj = os.path.join(OBJ_DIR, f'{os.path.splitext(src)[0]}.{tag}.o')
        objs.append(obj)
        if force or not os.path.exists(obj):
            jobs.append((sp, obj, flags))

    def compile_one(job):
        sp, obj, flags = job
        cmd = [hipcc] + flags + ['-c', sp, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {sp}:\n{r.stdout}\n{r.stderr}')
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    link_tag = _digest(objs, ['link'])
    stamp = os.path.join(OBJ_DIR, 'link.stamp')
    old = open(stamp).read().strip() if os.path.exists(stamp) else ''
    if force or jobs or old != link_tag or not os.path.exists(LIB_PATH):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        with open(stamp, 'w') as f:
            f.write(link_tag)
    # drop stale objects
    keep = set(os.path.basename(o) for o in objs) | {'link.stamp'}
    for fn in os.listdir(OBJ_DIR):
        if fn not in keep:
            try:
                os.remove(os.path.join(OBJ_DIR, fn))
            except OSError:
                pass
    return LIB_PATH


if __name__ == '__main__':
    print(build_library(verbose=True, force='--force' in sys.argv))
