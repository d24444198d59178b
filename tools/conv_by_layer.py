#!/usr/bin/env python
"""Joins the per-launch conv list of bench.py (--trace-layers: layer shape, useful FLOPs, pipe, plan, in launch order) with
the rocprofv3 kernel trace of THE SAME RUN, so that every conv layer's TFLOP/s and the per-pipe roofline fractions can be
recomputed from profiles/ alone.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 6 --warmup 2 --no-graph --seqs 1 \
        --no-em --no-cpu-baseline --load-plans PLANS --trace-layers OUT/layers.json
    python tools/conv_by_layer.py OUT/layers.json OUT/*/*_kernel_trace.csv profiles/r02_conv_by_layer.csv

A traced launch = [operand-split kernels] + ONE implicit-GEMM kernel + [split-K reduce kernel].  bench.py's traced frames are
the last conv launches of the run, so the last len(launches) implicit-GEMM dispatches of the trace are matched in order."""
import csv
import json
import sys

PEAK = {'bf16': 2500.0 / 6, 'bf16x3': 2500.0 / 3, 'f16x3': 2500.0 / 3, 'fp32': 157.3}


def peak_of(pipe, kname):
    """matching's readout GEMM is launched by the library with its own plan: its pipe follows from the kernel that ran"""
    if pipe != 'readout':
        return PEAK[pipe]
    if 'bf3s' in kname:
        return PEAK['bf16x3'] if kname[kname.index('<') + 1:].rstrip('>').split()[5] == '2' else PEAK['bf16']   # (template parameter NPL)
    return PEAK['bf16'] if 'bf3_kernel' in kname else PEAK['fp32']


def main(layers_json, trace_csv, out_csv):
    import re
    launches = json.load(open(layers_json))['launches']
    rows = sorted(csv.DictReader(open(trace_csv)), key=lambda r: int(r['Start_Timestamp']))
    dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    is_conv = lambda r: 'conv_igemm' in r['Kernel_Name'] or 'conv_t256_kernel' in r['Kernel_Name']
    is_red = lambda r: 'conv_splitk_epilogue' in r['Kernel_Name']
    is_split = lambda r: 'split_bf16x3' in r['Kernel_Name'] or 'split_f16x2' in r['Kernel_Name']
    # The traced frames are the LAST conv launches of the run: walk both lists backwards.  One traced launch = [operand-split
    # kernels] + one implicit-GEMM kernel (TWO for a tail-split plan whose last round of tiles was split over K: the whole
    # rounds, then the tail) + [split-K reduce kernel].
    agg = {}
    i = len(rows) - 1
    for la in reversed(launches):
        while i >= 0 and not (is_conv(rows[i]) or is_red(rows[i])):
            i -= 1
        assert i >= 0, 'kernel trace holds fewer conv launches than the layer list'
        t_main, t_split, t_red = 0.0, 0.0, 0.0
        if is_red(rows[i]):
            t_red = dur(rows[i])
            i -= 1
            while not is_conv(rows[i]):
                i -= 1
            ts = (la['plan'] >> 24) & 15
            if ts > 1 and i >= 1 and is_conv(rows[i - 1]) and rows[i - 1]['Kernel_Name'] == rows[i]['Kernel_Name'] and la['pipe'] != 'readout':
                t_main += dur(rows[i])          # the tail launch of a tail-split layer; the whole rounds precede it
                i -= 1
        mi = i
        t_main += dur(rows[mi])
        i -= 1
        j = i
        while j >= 0 and not (is_conv(rows[j]) or is_red(rows[j])) and mi - j <= 6:   # operand splits issued for this launch
            if is_split(rows[j]):
                t_split += dur(rows[j])
            j -= 1
        kname = re.search(r'(conv_(?:igemm|t256)\w*(<[^>]*>)?)', rows[mi]['Kernel_Name']).group(1).replace(', ', ' ')
        a = agg.setdefault((la['layer'], la['pipe'], '%#x' % la['plan'], kname), [0, 0.0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += la['flops']
        a[2] += t_main
        a[3] += t_split
        a[4] += t_red
        a[5] += la['event_us'] or 0.0
    frames = json.load(open(layers_json))['frames']
    with open(out_csv, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['layer (BxHxW k s cin->ncols)', 'pipe', 'plan', 'kernel', 'launches_per_frame', 'gflop_per_launch',
                    'main_kernel_us', 'operand_split_us', 'splitk_reduce_us', 'total_us', 'hip_event_us', 'tflops_useful',
                    'pipe_peak_tflops', 'frac_of_pipe_peak', 'ms_per_frame'])
        tot = {}
        for (layer, pipe, plan, kname), (n, fl, tm, ts, tr_, te) in sorted(agg.items(), key=lambda kv: -(kv[1][2] + kv[1][3] + kv[1][4])):
            t = tm + ts + tr_
            tf = fl / (t * 1e-6) / 1e12
            w.writerow([layer, pipe, plan, kname, '%.1f' % (n / frames), '%.3f' % (fl / n / 1e9), '%.1f' % (tm / n),
                        '%.1f' % (ts / n), '%.1f' % (tr_ / n), '%.1f' % (t / n), '%.1f' % (te / n), '%.1f' % tf,
                        '%.1f' % peak_of(pipe, kname), '%.4f' % (tf / peak_of(pipe, kname)), '%.3f' % (t / frames / 1e3)])
            d = tot.setdefault(pipe, [0.0, 0.0, peak_of(pipe, kname)])
            d[0] += fl
            d[1] += t
        for pipe, (fl, t, pk) in sorted(tot.items()):
            tf = fl / (t * 1e-6) / 1e12
            w.writerow(['TOTAL ' + pipe + ' pipe', pipe, '', '', '', '', '', '', '', '', '', '%.1f' % tf, '%.1f' % pk,
                        '%.4f' % (tf / pk), '%.3f' % (t / frames / 1e3)])
            print('%s pipe: %.1f useful TFLOP/s = %.3f of %.1f (%.3f ms per frame, rocprofv3 kernel durations)'
                  % (pipe, tf, tf / pk, pk, t / frames / 1e3))


if __name__ == '__main__':
    main(*sys.argv[1:4])
