#!/usr/bin/env python3
"""profiles/r04_pmc_summary.txt and profiles/traffic.json from the outputs of tools/final_prof.sh r04 (gpurun_out/final_r04/), after the
bench line and the kernel trace have been copied into profiles/.  Run here (no GPU)."""
import csv, json
new = open('gpurun_out/final_r04/pmc_summary.txt').read().rstrip('\n').split('\n')
c = {l.split()[0]: float(l.split()[1]) for l in new}
F = 9009528
rows = list(csv.reader(open('profiles/r04_bench_kernel_stats.csv')))
kt = float(rows[1][3]) * 1e-9
d = json.load(open('profiles/r04_bench_full.json'))
clk = c['GRBM_GUI_ACTIVE'] / 8 / kt / 1e9
valu = c['SQ_INSTS_VALU'] / F; lds = c['SQ_LDS_IDX_ACTIVE'] / F; conf = c['SQ_LDS_BANK_CONFLICT'] / F
hbm = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / F
slots = 197.0 - (123.2 - valu)  # round 4's price list: the packed phase 1 kept its 197 issue slots; what left phase 2 since were plain instructions
hdr = """Round 4: counters of frontend_kernel<13, DCTC, MODE 0, plain, MD> on the bench workload (10 000 S-MFCC utterances, 9 009 528 frames per
launch), rocprofv3 --pmc in four separate passes (tools/final_prof.sh r04: python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extra),
means over 7 dispatches.  The kernel's phase 1 is round 3's statement for statement with the two passes of a step in the halves of
packed registers (v_pk_add / mul / fma_f32, CTU_PK): 196.6 -> 123.2 vector instructions per frame; phase 2's slot walk with its
records one slot ahead and chunk reads addressed base + immediate: 123.2 -> %.1f (the LDS side unchanged).
""" % valu
per = """
Per frame:
  HBM traffic (2 x FETCH_SIZE + WRITE_SIZE, KiB; the guide's gfx950 correction for wide reads)   %.0f B   (algorithmic 372 B: %.2fx)
  VALU wave-instructions                                                                       %.1f   (of them packed: 72; issue slots at the in-situ prices: %.0f)
  LDS-array cycles (SQ_LDS_IDX_ACTIVE), of which bank conflicts                                  %.1f / %.1f
  SALU / SMEM / VMEM read / VMEM write instructions                                             %.1f / %.2f / %.2f / %.2f
  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                                                             %.3f   (0.322 in round 3: a packed instruction holds the issue longer)
  SQ_WAIT_ANY / SQ_WAVE_CYCLES                                                                  %.3f
  SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES                                                             %.3f
  effective clock (GRBM_GUI_ACTIVE / 8 / kernel time %.2f ms)                                   %.2f GHz
Kernel time: %.4f ms average of %s launches under rocprofv3 --kernel-trace --stats (r04_bench_kernel_stats.csv); %.4f ms by the
library's HIP events in the unprofiled bench run of the same call (r04_bench_full.json: roofline.kernel_ms); %.3ge9 frames/s on this box
(boxes of the pool differ by +-5 %%: the interleaved A/Bs of profiles/r04_frontend_cost_model.txt have the packed build 1.6 %% ahead of round 3's
and the phase-2 walk another 1.0 %%).
""" % (hbm, hbm / 372, valu, slots, lds, conf, c['SQ_INSTS_SALU'] / F, c['SQ_INSTS_SMEM'] / F, c['SQ_INSTS_VMEM_RD'] / F, c['SQ_INSTS_VMEM_WR'] / F,
       c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], c['SQ_WAIT_INST_LDS'] / c['SQ_WAVE_CYCLES'], kt * 1e3, clk, kt * 1e3, rows[1][1],
       d['roofline']['kernel_ms'], d['value'] / 1e9)
open('profiles/r04_pmc_summary.txt', 'w').write(hdr + "\n" + "\n".join(new) + "\n" + per)
t = json.load(open('profiles/traffic.json'))
t.update(valu_instr_per_frame=round(valu, 1), lds_cycles_per_frame=round(lds, 1), hbm_bytes_per_frame=int(round(hbm)), valu_issue_slots_per_frame=round(slots, 1), clock_ghz=round(clk, 2))
json.dump(t, open('profiles/traffic.json', 'w'), indent=1)
print("clock %.3f GHz, VALU %.1f, LDS %.1f cycles, HBM %.0f B per frame" % (clk, valu, lds, hbm))
for k, v in d['configs'].items():
    print(k, round(v.get('ms_per_pass'), 3), '%.3g' % v.get('frames_per_s'), round(v.get('hbm_frac'), 4), round(v.get('front_kernel_ms'), 3))
