"""summarise a rocprofv3 --pmc counter_collection.csv per kernel: the SQ wave-cycle breakdown + MFMA busy fraction of the SQ pass,
and, for the LDS / instruction pass (which carries no SQ_WAVE_CYCLES), bank-conflict cycles per LDS-array cycle, how busy the LDS
pipe was (LDS-array cycles / CU / elapsed cycle) and the instruction mix per MFMA
usage: python tools/pmc_summary.py <counter_collection.csv> [name filter]"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else 'conv'   # every convolution kernel (MFMA, Winograd, thin, stride-2)
agg = collections.OrderedDict()
for r in rows:
    k = (r['Kernel_Name'].split('(')[0][:64], r['Grid_Size'])
    agg.setdefault(k, collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if flt not in k[0]:
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get('SQ_WAVE_CYCLES', 0.0)
    line = '{} grid={} n={}'.format(k[0], k[1], len(next(iter(d.values()))))
    if 'GRBM_GUI_ACTIVE' in m and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        simd_cycles = m['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0        # GUI_ACTIVE is summed over 8 XCDs; 1024 SIMDs
        line += '  mfma_busy={:.3f} gui_cycles/xcd={:.3g}'.format(m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles, m['GRBM_GUI_ACTIVE'] / 8.0)
    if wc:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC'):
            if c in m:
                line += ' {}={:.3f}'.format(c.replace('SQ_', ''), m[c] / wc)
    if 'SQ_LDS_IDX_ACTIVE' in m:   # the LDS / instruction pass
        idx = m['SQ_LDS_IDX_ACTIVE']
        if idx:
            line += ' lds_conflict/active={:.3f}'.format(m.get('SQ_LDS_BANK_CONFLICT', 0.0) / idx)
        # elapsed cycles of the launch: GRBM_GUI_ACTIVE (summed over the 8 XCDs) where collected, else SQ_BUSY_CYCLES (summed
        # over the XCDs' shader engines: 32 on MI355X; busy = any wave resident, ~ the launch for a persistent grid)
        cyc = m['GRBM_GUI_ACTIVE'] / 8.0 if 'GRBM_GUI_ACTIVE' in m else (m['SQ_BUSY_CYCLES'] / 32.0 if 'SQ_BUSY_CYCLES' in m else 0.0)
        if cyc:
            line += ' lds_pipe_busy={:.3f} (conflict share {:.3f}) cycles={:.3g}'.format(idx / 256.0 / cyc, m.get('SQ_LDS_BANK_CONFLICT', 0.0) / 256.0 / cyc, cyc)
        if m.get('SQ_INSTS_MFMA'):
            line += ' insts_mfma={:.4g} lds/mfma={:.2f} valu/mfma={:.2f}'.format(m['SQ_INSTS_MFMA'], m.get('SQ_INSTS_LDS', 0.0) / m['SQ_INSTS_MFMA'], (m.get('SQ_INSTS_VALU', 0.0) - m['SQ_INSTS_MFMA']) / m['SQ_INSTS_MFMA'])
    print(line)
