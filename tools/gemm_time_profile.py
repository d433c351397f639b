"""GEMM-family kernel time per step out of a round's rocprofv3 files, for bench.py's `roofline.traced` / `roofline.serial`:

  python tools/gemm_time_profile.py <kernel_stats.csv of `bench.py` under --kernel-trace --stats> <serial_step_report.txt> <out.json>

`traced` = the default multi-queue schedule under the tracer (durations of one queue's kernels include their neighbours on the other
queues), `serial` = the same step on ONE stream (S4F_SIDE_STREAM=0 S4F_HEAD_STREAMS=0 S4F_EAGER_SGD=0: every kernel alone on the chip).
The file names the kernel sources it was measured on (pmc_traffic.kernel_src_sha): bench.py refuses it after a kernel change."""
import csv
import json
import re
import sys

from pmc_traffic import kernel_src_sha

LAYERS = 12      # DeiT-B: one fb::main_kernel (attention backward sweep) per encoder layer and step


def is_gemm(name):
    return 'gemm' in name and 'GemmArgs' in name or 'gemm_kernel' in name or 'GroupArgs' in name


def traced(path):
    rows = list(csv.DictReader(open(path)))
    sweeps = sum(int(r['Calls']) for r in rows if 'fb::main_kernel' in r['Name'])
    steps = sweeps / LAYERS
    gemm_ns = sum(float(r['TotalDurationNs']) for r in rows if is_gemm(r['Name']))
    all_ns = sum(float(r['TotalDurationNs']) for r in rows)
    calls = sum(int(r['Calls']) for r in rows if is_gemm(r['Name']))
    return dict(steps=steps, gemm_ms_per_step=gemm_ns / steps / 1e6, all_kernels_ms_per_step=all_ns / steps / 1e6, gemm_launches_per_step=calls / steps)


def serial(path):
    step_ms, gemm_ms, calls = None, 0.0, 0
    for ln in open(path):
        m = re.match(r'step ([\d.]+) ms', ln)
        if m:
            step_ms = float(m.group(1))
        m = re.match(r'\s+(.*?)\s+(\d+)\s+([\d.]+) ms\s+avg', ln)
        if m and is_gemm(m.group(1)):
            gemm_ms += float(m.group(3))
            calls += int(m.group(2))
    return dict(step_ms=step_ms, gemm_ms_per_step=gemm_ms, gemm_launches_per_step=calls)


if __name__ == '__main__':
    out = dict(traced=traced(sys.argv[1]), serial=serial(sys.argv[2]), kernel_src_sha=kernel_src_sha(),
               note='GEMM-family kernel time per step (gemm_kernel / gemm2 / gemm5 / gemm5p / gemm6, grouped): `traced` from rocprofv3 '
                    '--kernel-trace --stats of bench.py (default streams), `serial` from the one-stream run of the same step')
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps(out))
