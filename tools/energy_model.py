"""
The energy model of FLOOR.md section 2, reproducible: per-ingredient energies from tools/power_phases.sh's table
(package watts and rate per instruction class / stream, measured beside rocm-smi) + the headline's hot-path instruction mix
(tools/dis.sh) + its PMC summary (wave count, VALU instructions, HBM bytes) -> energy per launch and the
launch time at the package power cap.

    python tools/energy_model.py profiles/r03_power_phases.txt profiles/r03_pmc_headline_summary.json [cap_watts]
"""
import json
import re
import sys

phases, pmc = sys.argv[1], json.load(open(sys.argv[2]))
cap = float(sys.argv[3]) if len(sys.argv) > 3 else 1400.0
rows = {}
for line in open(phases):
    m = re.match(r'^(.*?)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+(GB/s|Ginstr/s|-)', line)
    if m:
        rows[m.group(1).strip()] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))
base_w = next(v[0] for k, v in rows.items() if 's_nop' in k)                 # every CU clocked, nothing issued
nj = {}                                                                      # energy per wave-instruction (nJ)
for k, (w, clk, rate) in rows.items():
    if k.startswith('valu ') and 's_nop' not in k and rate > 0:
        nj[k[5:].split(' (')[0]] = (w - base_w) / rate                      # W / (G instr/s) = nJ per instruction
w_s, _, gbps = rows['stream 2-read 1-write nt (GB/s)']
pj_per_byte = (w_s - base_w) / gbps * 1e3                                    # W / (GB/s) = nJ per byte -> pJ
lds_read_nj = nj['ds_read_b128']                                             # per 1 KB wave-instruction
lds_write_nj = nj['ds_write_b128 + ds_read_b128'] - lds_read_nj

# hot path of the dense certificate-only 5x5 gain-offset kernel, wave-instructions per wave-row
mix = {
    'v_add_f64': 24 + 36 + 4, 'v_fma_f64': 16 + 8, 'v_mul_f64': 12, 'v_rcp_f64': 4,
    'v_cvt_f64_f32': 24 + 12, 'v_cvt_f32_f64': 16 + 8, 'v_mov_b32_dpp wave_shr': 36,
    'v_pk_fma_f32': 4 + 32, 'v_add_f32': 9 + 20, 'v_cmp_lt_f32': 10, 'v_mov_b32': 6,
}
valu_per_row = sum(mix.values())
e_row = sum(nj[k] * n for k, n in mix.items())
lds_row = 2 * lds_write_nj + 3 * lds_read_nj                                 # ring: 2 KB written, 3 KB read per wave-row
valu_total = pmc['counters']['SQ_INSTS_VALU']
wave_rows = valu_total / valu_per_row                                        # incl. halo lanes and priming rows
e_valu = wave_rows * e_row * 1e-9
e_lds = wave_rows * lds_row * 1e-9
e_mem = pmc['hbm_traffic_bytes'] * pj_per_byte * 1e-12
t_ms = (e_valu + e_lds + e_mem) / (cap - base_w) * 1e3
print(f'clocked base {base_w:.0f} W; HBM stream {pj_per_byte:.0f} pJ/B; LDS {lds_read_nj:.1f} nJ per KB read, {lds_write_nj:.1f} per KB written')
print('nJ per wave-instruction: ' + ', '.join(f'{k} {v:.2f}' for k, v in sorted(nj.items()) if not k.startswith('ds_')))
print(f'hot path: {valu_per_row} VALU instructions = {e_row:.0f} nJ per wave-row (+ {lds_row:.0f} nJ of LDS); '
      f'{valu_total / 1e9:.3f} G VALU instructions per launch = {wave_rows / 1e6:.2f} M wave-rows')
print(f'per launch: VALU {e_valu:.2f} J + LDS {e_lds:.2f} J + memory {e_mem:.2f} J ({pmc["hbm_traffic_bytes"] / 1e9:.2f} GB) '
      f'+ base {base_w / 1e3:.3f} J per ms')
print(f'at {cap:.0f} W: {t_ms:.2f} ms per launch  (shares at that time: base {base_w * t_ms / 1e3 / (cap * t_ms / 1e3):.0%}, '
      f'memory {e_mem / (cap * t_ms / 1e3):.0%}, VALU {e_valu / (cap * t_ms / 1e3):.0%}, LDS {e_lds / (cap * t_ms / 1e3):.0%})')
for target, name in ((0.60, '0.60'), (0.70, '0.70')):
    t = pmc['algorithmic_bytes'] / (target * 8e12) * 1e3
    budget = (cap - base_w) * t / 1e3 - e_mem - e_lds
    print(f'{name} of 8 TB/s = {t:.2f} ms leaves {budget:.2f} J for the arithmetic: {budget / e_valu - 1:+.0%} against {e_valu:.2f} J')
