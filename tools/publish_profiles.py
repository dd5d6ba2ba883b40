"""copy the summaries of one tools/collect_profiles.sh collection (gpurun_out/prof_<tag>/) into profiles/ under the round's names:
kernel statistics, MFMA shape report, PMC fetch / write table (the file bench.py reads), SQ + LDS counter summaries, roofline
tables with the algorithmic-bytes column, single-stream inference statistics, the three bench lines.
usage: python tools/publish_profiles.py <collection dir> <prefix, e.g. r04_b> <round prefix of the PMC table, e.g. r04> [parity_report.txt]"""
import json, os, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
P = os.path.join(REPO, 'profiles')
T = os.path.join(REPO, 'tools')


def run(args, out):
    with open(out, 'w') as f:
        subprocess.check_call([sys.executable] + args, stdout=f)


def line(log, out):
    with open(log) as f:
        last = [l for l in f.read().splitlines() if l.startswith('{')][-1]
    with open(out, 'w') as f:
        json.dump(json.loads(last), f, indent=1)


shutil.copy(os.path.join(src, 'fp32', 'k_kernel_stats.csv'), os.path.join(P, tag + '_bench_kernel_stats.csv'))
shutil.copy(os.path.join(src, 'fp32_shapes.json'), os.path.join(P, tag + '_mfma_conv_shapes.json'))
shutil.copy(os.path.join(src, 'infer', 'k_kernel_stats.csv'), os.path.join(P, tag + '_infer_fp32_single_stream_kernel_stats.csv'))
pmc = os.path.join(P, rnd + '_pmc_fetch_write_per_kernel.json')
subprocess.check_call([sys.executable, os.path.join(T, 'pmc_traffic.py'), os.path.join(src, 'fp32_fetch', 'p_counter_collection.csv'),
                       os.path.join(src, 'fp32_write', 'p_counter_collection.csv'), pmc])
with open(os.path.join(P, tag + '_pmc_sq_summary.txt'), 'w') as f:
    f.write('== SQ wave-cycle counters (tools/pmc_summary.py; fractions of SQ_WAVE_CYCLES, mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles)\n')
    f.flush()
    subprocess.check_call([sys.executable, os.path.join(T, 'pmc_summary.py'), os.path.join(src, 'fp32_sq', 'p_counter_collection.csv')], stdout=f)
    f.write('== LDS / instruction counters (lds_conflict/active = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; lds_pipe_busy = '
            'SQ_LDS_IDX_ACTIVE / 256 CUs / elapsed cycles)\n')
    f.flush()
    subprocess.check_call([sys.executable, os.path.join(T, 'pmc_summary.py'), os.path.join(src, 'fp32_lds', 'p_counter_collection.csv')], stdout=f)
run([os.path.join(T, 'roofline_table.py'), os.path.join(P, tag + '_bench_kernel_stats.csv'), pmc, os.path.join(P, tag + '_mfma_conv_shapes.json')],
    os.path.join(P, tag + '_roofline_table.md'))
line(os.path.join(src, 'fp32_bench_line.log'), os.path.join(P, tag + '_bench_line.json'))
line(os.path.join(src, 'fp32_bench_line_no_overlap.log'), os.path.join(P, tag + '_bench_line_no_overlap.json'))
if os.path.isdir(os.path.join(src, 'bf16')):
    shutil.copy(os.path.join(src, 'bf16', 'k_kernel_stats.csv'), os.path.join(P, tag + '_bf16_kernel_stats.csv'))
    shutil.copy(os.path.join(src, 'bf16_shapes.json'), os.path.join(P, tag + '_bf16_mfma_conv_shapes.json'))
    pmc16 = os.path.join(P, rnd + '_bf16_pmc_fetch_write_per_kernel.json')
    subprocess.check_call([sys.executable, os.path.join(T, 'pmc_traffic.py'), os.path.join(src, 'bf16_fetch', 'p_counter_collection.csv'),
                           os.path.join(src, 'bf16_write', 'p_counter_collection.csv'), pmc16])
    run([os.path.join(T, 'roofline_table.py'), os.path.join(P, tag + '_bf16_kernel_stats.csv'), pmc16, os.path.join(P, tag + '_bf16_mfma_conv_shapes.json'),
         'auto', 'no-algo'], os.path.join(P, tag + '_bf16_roofline_table.md'))
    line(os.path.join(src, 'bf16_bench_line.log'), os.path.join(P, tag + '_bf16_bench_line.json'))
if len(sys.argv) > 4:
    shutil.copy(sys.argv[4], os.path.join(P, tag + '_parity_report.txt'))
print('published', tag)
