import json, re
d = json.loads(open('/root/repo/gpurun_out/prof_r06/r06_bench_n1.json').read().strip().splitlines()[-1])
r = d['roofline']; ex = d['extra']; fr = ex['frame_nvi_288x512']; fk = fr['kernel_ms_per_frame_rank0_one_stream']
ks = {}
for line in open('/root/repo/gpurun_out/prof_r06/r06_bench_kernel_stats.txt'):
  m = re.match(r'(k_\w+)\S*\s.*?(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)', line)
  if m: ks[m.group(1)] = float(m.group(4))
k = d['kernels_avg_ms']
pw = ex['power_under_step_loop']
txt = f'''## 5. Measurements (MI355X, round 6)

All numbers: one MI355X per run through `gpurun` (a fresh box per call; boxes differ by ±4 % — the round's sessions saw the same library between 2.58 and 2.84 ms per step), the library and `bench.py`
of the final commit of the round; every change of the round was judged by same-box A/B in alternating processes (`profiles/r06_ab_variants.txt`).  Summaries under `profiles/r06_*`: `r06_bench_n1.json` = the
bench line, `r06_bench_kernel_stats.txt` / `r06_frame_kernel_stats.txt` = rocprofv3 `--kernel-trace --stats`, `r06_bench_pmc.txt` / `r06_frame_pmc.txt` = separate `--pmc` passes, `r06_traffic.json`,
`r06_frame_nvi_288x512.txt`, `r06_ab_variants.txt`, `r06_stream_determinism.txt`, `r06_parity_margins.json`.

**Bench** (`bench.py`, BASELINE configs[1]: 4096 rays × 64 samples × 8 views, static branch, inputs resident in HBM): **{d['value'] / 1e6:.3f} M rays/s on one GPU, {d['ms_per_step']:.3f} ms per step** in the profiled session
({sum(pw['package_power_w']) / 3:.0f} W of the 1400 W cap at {sum(pw['shader_clock_mhz']) / 3000:.2f} GHz under the step loop); CPU oracle on the same box {d['cpu_baseline']['value']:.0f} rays/s (a port, {d['cpu_baseline']['cores']} threads); colours vs the
oracle {d['check_vs_oracle']['max_abs_rgb_err']:.2e} ({d['check_vs_oracle']['psnr_db']:.0f} dB).  **The exact 6-term engine** (`libdynibar_hip_x6.so`, timed by default since round 6): {d['x6_engine']['value'] / 1e6:.3f} M rays/s, {d['x6_engine']['ms_per_step']:.3f} ms per step =
**{r['x6_over_shipped_time']:.2f} ×** the shipped step, `k_static_views` {d['x6_engine']['k_static_views_ms']:.2f} ms = {d['x6_engine']['roofline_frac_of_its_own_peak']:.2f} of ITS ceiling (2500 / 6), the same {d['x6_engine']['max_abs_rgb_err_vs_oracle']:.2e} against the oracle: fp32-class products cost 70 % more time and buy nothing
the 1e-4 contract can see.  (The box decides ±4 % of this line: the same kernels on the faster boxes of the round — collection of commit `50704b9`: 1.547 M rays/s, 2.648 ms per step at 2.01 GHz,
`k_static_views` 1739 µs = 0.50, blend 322 µs = 0.50, frame 630 ms; the A/B session of call 31 (`r06_ab_variants.txt`): 2.574 ms per step, `k_static_views` 1665 µs, frame 634 ms.  The two source changes
since — the `torch.cross` axis and the `expm1` pooling weights — cost nothing in same-box A/B, calls 23 and 31.)

| kernel | avg µs per launch, HIP events (rocprofv3 `r06_bench_kernel_stats.txt`) | roofline | round 5 (same measure) |
|---|---|---|---|
| `k_static_views<8>` | {k['k_static_views'] * 1e3:.0f} ({ks.get('k_static_views', 0):.0f}) | MFMA: {r['achieved']:.0f} TFLOP/s algorithmic = **{r['frac']:.2f}** of 833 (`roofline.frac`; {718.33 / ks.get('k_static_views', 1) * 1e3 / 833.3:.2f} from the rocprofv3 average); matrix pipe {r['mfma_busy']:.2f} busy at {r['effective_clock_ghz_under_profiler']:.2f} GHz under the profiler; {r['valu_per_mfma']:.2f} other VALU per MFMA; traffic {r['traffic'] / 1e9:.2f} GB per launch | 1695–1860 (1860); 0.46–0.51 |
| `k_net_points<false,0>` | {k['k_static_points'] * 1e3:.0f} ({ks.get('k_net_points', 0):.0f}) | MFMA: **{r['points_frac']:.2f}** of 833 (`roofline.points_frac`); SEs {d['roofline']['state']['se_busy_fraction']['k_net_points']:.2f} busy (no slow state) | 451 (471); 0.28 |
| `k_static_blend_ws<8>` | {k['k_static_blend'] * 1e3:.0f} ({ks.get('k_static_blend_ws', 0):.0f}) | HBM: 1.29 GB compulsory = **{r['blend_frac']:.2f}** of 8 TB/s (`roofline.blend_frac`); counter traffic {r['blend_traffic_over_compulsory']:.3f} × compulsory; zero scratch | 343–358; 0.45–0.47; 36 B / lane of scratch |
| `k_project_gather_tile<16>` | {k['k_project_gather'] * 1e3:.1f} ({ks.get('k_project_gather_tile', 0):.1f}) | HBM: 360.3 MB = **{r['k1_v8_frac']:.2f}** of 8 TB/s; 11 views {r['k1_v11_us']:.0f} µs = **{r['k1_v11_frac']:.2f}**; 7 views {r['k1_v7_us']:.0f} µs = **{r['k1_v7_frac']:.2f}**; inside the frame {r['k1_inframe_ms']:.1f} ms for {fr['gather_algorithmic_bytes_rank0'] / 1e9:.1f} GB of §8(d) bytes = **{r['k1_inframe_frac']:.2f}** | 80.4 (81.7); 0.56 / 0.52 / 0.46 / 0.46 |
| `k_static_ref_feat` | {k['k_static_ref_feat'] * 1e3:.0f} ({ks.get('k_static_ref_feat', 0):.0f}) | one embedding per ray instead of 35; whole register file (§6) | 21 |

**Full frame** (BASELINE configs[2] on one GPU: 288 × 512 rays, 64 + 64 samples, 7 dynamic + 11 static views, chunk 8192, two chunk streams): **{fr['ms_per_frame']:.0f} ms** in the profiled session = {r['frame_frac']:.2f} of the split-MFMA
ceiling on SURVEY's 1.64 GFLOP per ray (round 5: 654–679 ms; the sessions of this round: 626–668 depending on the box; one stream +2.0 %, three streams ±0).  Per kernel on one stream, ms per frame (round 5 in brackets):
static views {fk['k_static_views']:.1f} [281.4], dynamic views {fk['k_dynamic_views']:.1f} [128.0], motion {fk['k_motion_mlp']:.1f} [63.1], blend {fk['k_static_blend']:.1f} [63.2], dynamic points {fk['k_dynamic_points']:.1f} [60.5], static points {fk['k_static_points']:.1f} [51.3], gather {fk['k_project_gather']:.1f} [22.9],
the ragged plan {fk.get('k_ragged_plan', 0):.1f}, `k_static_ref_feat` {fk['k_static_ref_feat']:.1f}.  Against regular dense rows on the same box (alternating processes): static views −7.4 %, blend −4.5 %, frame −3.3 %.
Other legs of the line: 11 static views {ex['views_11']['value'] / 1e6:.2f} M rays/s, {ex['views_11']['ms_per_step']:.2f} ms per step (`k_static_views` **{ex['views_11']['k_static_views_vs_8_views']:.2f} ×** its 8-view time for 1.375 × the rows; round 5: 1.45–1.50);
configs[3] kid-running monocular frame **{ex['mono_frame_kid']['ms_per_frame']:.0f} ms** [282]; configs[4] stress chunk **{ex['stress_chunk']['ms_per_chunk']:.1f} ms** [112.5]; eval-loop body **{ex['eval_loop']['ms_per_view']:.0f} ms per view** of GPU-path time on synthetic data [666–693];
training {ex['train_static_step']['ms_per_step']:.1f} ms (bootstrap step) and {ex['train_full_iteration']['rays_3072']['ms_per_step']:.1f} ms (full iteration at 3072 rays, `k_train_gemm` {ex['train_full_iteration']['rays_3072']['roofline_train_gemm']['frac']:.2f} of 8 TB/s, {ex['train_full_iteration']['peak_mem_gb']:.1f} GB peak) [41.7 / 95–100: unchanged code].

'''
s = open('/root/repo/DESIGN.md').read()
i0 = s.index('## 5. Measurements (MI355X, round')
i1 = s.index('## 6. Multi-GPU')
s = s[:i0] + txt + s[i1:]
open('/root/repo/DESIGN.md', 'w').write(s)
print(txt[:1500])
