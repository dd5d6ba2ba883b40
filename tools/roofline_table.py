"""per-kernel roofline table (markdown) from the committed round profiles:
  kernel statistics (rocprofv3 --kernel-trace --stats), the FETCH_SIZE / WRITE_SIZE PMC passes, the MFMA shape report
usage: python tools/roofline_table.py <kernel_stats.csv> <pmc_fetch_write.json> <mfma_shapes.json> [steps | auto] [no-algo] > table.md"""
import csv
import json
import sys

HBM_PEAK_GBS, MFMA_PEAK_TF = 8000.0, 157.3
stats = list(csv.DictReader(open(sys.argv[1])))
pmc = json.load(open(sys.argv[2]))
shapes = json.load(open(sys.argv[3]))['mfma_conv_launches']
steps_arg = sys.argv[4] if len(sys.argv) > 4 else 'auto'
if steps_arg == 'auto':   # every step the traced process ran = calls of the once-per-step loss kernel
    steps = next(int(r['Calls']) for r in stats if r['Name'].split('(')[0].strip().endswith(('dice_partial_kernel', 'focal_partial_kernel')))
else:
    steps = int(steps_arg)

# algorithmic FLOPs per second of the 3x3x3 forward/dgrad instantiations from the live shape report
variant_flops = {}
for r in shapes:
    N, D, H, W, Ci, Co, v = r['N_D_H_W_Cin_Cout_variant']
    if v == 600:     # Winograd F(2x2,3x3): TFLOP/s below are ALGORITHMIC (the matrix cores execute 4/9 of them)
        name = 'conv3d_k3_wino2d_kernel'
    elif v == 601:   # the same on 4^3 cells
        name = 'conv3d_k3_wino2d_c4_kernel'
    elif v == 500:   # Winograd F(2,3): algorithmic as well (2/3 executed)
        name = 'conv3d_k3_wino_kernel'
    elif v >= 400:
        name = 'conv3d_k3_mfma2w8_bf16_kernel<{}, {}, true>'.format((v - 400) // 10, v % 10)
    elif v >= 300:
        name = 'conv3d_k3_mfma2w8_kernel<{}, {}>'.format((v - 300) // 10, v % 10)
    elif v >= 200:
        name = 'conv3d_k3_mfma2_bf16_kernel<{}, {}, true>'.format((v - 200) // 10, v % 10)
    else:
        name = 'conv3d_k3_mfma2_kernel<{}, {}>'.format((v - 100) // 10, v % 10) if v >= 100 else 'conv3d_k3_mfma_kernel<{}>'.format(v)
    e = variant_flops.setdefault(name, [0.0, 0.0])
    e[0] += 2.0 * N * D * H * W * 27 * Ci * Co * r['launches_per_step']
    e[1] += r['avg_ms'] * r['launches_per_step']

# ALGORITHMIC bytes per launch (input + output tensors once, fp32) of the HBM-bound kernels that run once per step with ONE shape
# in the headline configuration (vnet(1,2), batch 4, 96^3; SURVEY.md 8a layer table x 4 patches): the roof these kernels are priced
# against is 8 TB/s on THESE bytes -- counter traffic above them is waste, not credit.  Multi-shape kernels (GroupNorm passes,
# the stride-2 forward at four levels) have no single figure and keep '-'.
V = 4 * 96 ** 3 * 4 / 1e6   # MB of one fp32 channel of the batch at full resolution
ALGO_MB = {
    'conv3d_k3_thin_in_persistent16_kernel<1': (1 + 16) * V,        # stem forward: 1 -> 16 channels
    'conv3d_k3_thin_out_f32mfma_kernel<32, 2': (32 + 2) * V,         # head forward: 32 -> 2
    'conv3d_k3_thin_in_persistent_kernel<2': (2 + 32) * V,           # head data-gradient: 2 -> 32
    'k3_thin_wgrad_kernel<2>': (32 + 2) * V,                         # head weight gradient: x (32) and dy (2)
    'k3_thin_wgrad_kernel<1>': (1 + 16) * V,                         # stem weight gradient: x (1) and dy (16)
    'convT3d_k2s2_mfma_kernel<0, false, false, true>': (64 / 8 + 16) * V,        # up_32.up_conv: 64 ch at 48^3 -> 16 ch at 96^3
    'convT3d_k2s2_mfma_kernel<0, false, true, true>': (32 / 8 + 16 + 16) * V,    # down_32 data-gradient + skip addend -> 16 ch at 96^3
    'convT3d_k2s2_direct_kernel<0, true, true>': (32 / 8 + 16 + 16) * V,         # the same launch on the direct scatter kernel (round 4)
    'conv3d_k2s2_direct_kernel<0, false, 2, 1>': (16 + 64 / 8) * V,              # up_32.up_conv data-gradient: 16 ch at 96^3 -> 64 ch at 48^3
    'adam_step_devstep_kernel': 28 * 14563296 / 1e6,                 # 28 B per parameter
}
if len(sys.argv) > 5 and sys.argv[5] == 'no-algo':   # other configurations (bf16 mode, other networks): the figures above do not apply
    ALGO_MB = {}
print('| kernel | launches/step | ms/step | avg us | traffic at the L2 boundary, GB/launch (2 x FETCH + WRITE) | GB/s | % of 8 TB/s | algorithmic MB/launch | % of 8 TB/s on algorithmic bytes | TFLOP/s | % of 157.3 |')
print('|---|---|---|---|---|---|---|---|---|---|---|')
for r in stats[:40]:
    name = r['Name'].split('(')[0].strip()
    key = name
    short = name[5:] if name.startswith('void ') else name
    calls = float(r['Calls']) / steps
    ms = float(r['TotalDurationNs']) / 1e6 / steps
    avg_us = float(r['AverageNs']) / 1e3
    e = pmc.get(key) or pmc.get('void ' + short) or {}
    gb = None
    if 'FETCH_SIZE_KB_per_launch' in e and 'WRITE_SIZE_KB_per_launch' in e:
        gb = (2.0 * e['FETCH_SIZE_KB_per_launch'] + e['WRITE_SIZE_KB_per_launch']) * 1024 / 1e9
    gbs = gb / (avg_us * 1e-6) if gb else None
    tf = None
    # template instantiations of one kernel (conv3d_k3_wino2d_kernel<BIAS, ADD> since round 3) share the live shape report's
    # entry of the base name: their TFLOP/s figure is the average over all instantiations
    vkey = short if short in variant_flops else short.split('<')[0]
    if vkey in variant_flops and variant_flops[vkey][1] > 0:
        tf = variant_flops[vkey][0] / (variant_flops[vkey][1] * 1e-3) / 1e12
    amb = next((v for k, v in ALGO_MB.items() if short.startswith(k)), None)
    if amb is not None and abs(calls - 1.0) > 0.2:
        amb = None   # not the once-per-step launch the figure was derived for
    print('| `{}` | {:.1f} | {:.3f} | {:.1f} | {} | {} | {} | {} | {} | {} | {} |'.format(
        short[:60], calls, ms, avg_us, '-' if gb is None else '{:.3f}'.format(gb), '-' if gbs is None else '{:.0f}'.format(gbs),
        '-' if gbs is None else '{:.0f}'.format(100 * gbs / HBM_PEAK_GBS),
        '-' if amb is None else '{:.0f}'.format(amb), '-' if amb is None else '{:.0f}'.format(100 * amb / 1e3 / (avg_us * 1e-6) / HBM_PEAK_GBS),
        '-' if tf is None else '{:.1f}'.format(tf), '-' if tf is None else '{:.0f}'.format(100 * tf / MFMA_PEAK_TF)))
