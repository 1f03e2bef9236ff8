"""Generate the measurement tables of DESIGN.md section 6 from the files committed under profiles/, so that the text cannot drift
from the evidence (a quoted 40.2 / 47.6 us had become 41.66 / 45.87 in the committed summary).

    python scripts/design_tables.py            # print the markdown block
    python scripts/design_tables.py --write    # replace the block between the GENERATED markers in DESIGN.md

tests/test_host_logic.py::test_design_tables_are_generated_from_the_committed_profiles fails when DESIGN.md and this output differ.
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")
BEGIN = "<!-- BEGIN GENERATED: round-5 tables (scripts/design_tables.py) -->"
END = "<!-- END GENERATED -->"
PEAK = 8000.0  # GB/s, MI355X_MICROARCH.md


def _json_line(name):
    path = os.path.join(PROFILES, name)
    if not os.path.exists(path):
        return None
    text = open(path).read().strip()
    try:
        return json.loads(text)
    except ValueError:
        for line in reversed(text.splitlines()):
            line = line.strip()
            if line.startswith("{"):
                try:
                    return json.loads(line)
                except ValueError:
                    continue
    return None


def _jsonl(name):
    path = os.path.join(PROFILES, name)
    if not os.path.exists(path):
        return []
    out = []
    for line in open(path):
        line = line.strip()
        if line.startswith("{"):
            out.append(json.loads(line))
    return out


def _kernel_summary(name):
    """{kernel short name: (calls, avg us)} from a scripts/summarize_prof.py `_kernel_summary.txt`."""
    path = os.path.join(PROFILES, name)
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"(\S+?)[(<].*calls=\s*(\d+)\s+avg=\s*([0-9.]+)us", line)
        if m:
            key = m.group(1)
            tmpl = re.match(r"\S+?<([^>]*)>", line)
            if tmpl and key.startswith(("gm_fwd", "gm_bwd", "bn_eval_combine", "bn_finalize")):
                key = f"{key}<{tmpl.group(1)}>"
            out.setdefault(key, (int(m.group(2)), float(m.group(3))))
    return out


def _pmc(name, kernel):
    path = os.path.join(PROFILES, name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["kernel"].startswith(kernel):
                return float(row["avg_value"])
    return None


def fmt(v, nd=1):
    return "n/a" if v is None else f"{v:.{nd}f}"


N_LIST = {"resnet18": 11_689_512, "resnet50": 25_557_032, "bert_base": 86_073_402}


def _by_grid(name):
    """[(kernel short name with template arguments, workgroups, calls, avg us, median us)] from a `_kernel_by_grid.csv`."""
    path = os.path.join(PROFILES, name)
    out = []
    if not os.path.exists(path):
        return out
    with open(path) as f:
        for row in csv.DictReader(f):
            m = re.match(r"([A-Za-z_0-9]+(?:<[^>]*>)?)", row["kernel"])
            out.append((m.group(1) if m else row["kernel"][:40], int(row["workgroups"]), int(row["calls"]), float(row["avg_us"]), float(row["median_us"])))
    return out


def _frac(nbytes, us):
    return nbytes / us / 1e3 / PEAK


def _pmc_ratio(fetch_file, write_file, kernel, algorithmic, dispatches_per_call=1):
    f, w = _pmc(fetch_file, kernel), _pmc(write_file, kernel)
    if f is None or w is None:
        return None
    traffic = (2.0 * f + w) * 1024.0 * dispatches_per_call  # FETCH_SIZE counts half of wide coalesced reads on gfx950; values in KiB
    return traffic, traffic / algorithmic


def build():
    L = [BEGIN, "", "*(generated from the files named next to each figure; do not edit by hand)*", ""]
    b = _json_line("r5_bench_driver_style.json")
    L += ["**The driver-style line** (`python bench.py --gpus 1 --steps 20 --warmup 5`, `r5_bench_driver_style.json`):", "",
          "| quantity | value |", "|---|---|"]
    if b:
        r, k = b["roofline"], b["kernels"]
        L.append(f"| attack iterations / s, 1 GPU, one trial, {b['launch_mode']} | **{b['value']:.1f}** ({b['ms_per_step']:.3f} ms / iteration; eager launches {fmt(b.get('eager_ms_per_step'), 2)} ms) |")
        g = b.get("gpu_torch_baseline") or {}
        c = b.get("cpu_baseline") or {}
        if g.get("value"):
            L.append(f"| the same attack with PyTorch-ROCm ops on the same GPU (`gpu_torch_baseline`: oracle/restate.py on cuda) | {g['value']:.1f} it/s -> the HIP path is **{b['value'] / g['value']:.1f}x** |")
        if c:
            anchor = c.get("anchor") or {}
            L.append(f"| CPU baseline in the same run ({c['kind']}; {c['cores']} of {c['host_cpu_count']} logical cores) | {c['value']:.2f} it/s" +
                     (f"; port / unmodified reference on equal threads = {anchor.get('port_over_reference')} (`{anchor.get('file')}`)" if anchor else "") + " |")
        L.append(f"| kernel A forward, dispatch start/stop events, ResNet-18 list ({r['algorithmic_bytes'] / 1e6:.1f} MB, Infinity-Cache resident) | {r['avg_launch_us']:.2f} us -> {r['achieved']:.0f} GB/s = **{r['frac']:.3f}** of 8 TB/s; stage with finalize {r['stage_us']:.2f} us = {r['stage_frac']:.3f}; device span inside graph replay {fmt(r.get('timed_region_span_us'), 2)} us |")
        ce = r.get("ceiling") or {}
        if ce.get("us"):
            L.append(f"| `roofline.ceiling`: bare read of the same bytes with kernel A's grid and loads, behind a writer, same events, same run | {ce['us']:.2f} us = {ce['GBs']:.0f} GB/s ({ce['behind_writer']['frac_of_hbm_peak']:.3f} of 8 TB/s; warm {ce['warm']['median_us']:.2f} us) -> kernel A forward at **{r.get('frac_of_ceiling', 0):.2f} of the measured ceiling** |")
        L.append(f"| kernel A backward ({k['bwd']['algorithmic_bytes'] / 1e6:.1f} MB) | {k['bwd']['avg_us']:.2f} us -> {k['bwd']['achieved_GBs']:.0f} GB/s = **{k['bwd']['frac_of_hbm_peak']:.3f}** |")
        eager = _kernel_summary("r5_bench_eager_kernel_summary.txt")
        replay = _kernel_summary("r5_bench_kernel_summary.txt")
        fk, bk = "gm_fwd_kernel<0, false>", "gm_bwd_kernel<0, false, false>"
        if fk in eager and fk in replay:
            L.append(f"| the same two kernels by rocprofv3, by launch mode: `bench.py --no-graph` (eager launches -- the mode the event-timed iterations above run in, a replayed graph cannot carry event pairs; `r5_bench_eager_kernel_summary.txt`) / inside graph replay (`r5_bench_kernel_summary.txt`) | forward {eager[fk][1]:.2f} / {replay[fk][1]:.2f} us = {_frac(r['algorithmic_bytes'], eager[fk][1]):.3f} / **{_frac(r['algorithmic_bytes'], replay[fk][1]):.3f}**; backward {eager[bk][1]:.2f} / {replay[bk][1]:.2f} us = {_frac(k['bwd']['algorithmic_bytes'], eager[bk][1]):.3f} / **{_frac(k['bwd']['algorithmic_bytes'], replay[bk][1]):.3f}** -- events and rocprofv3 agree within {abs(eager[fk][1] / r['avg_launch_us'] - 1) * 100:.0f} % in the same launch mode; inside the replayed graph (the timed region) the kernels are {(1 - replay[fk][1] / eager[fk][1]) * 100:.0f} % faster |")
        if r.get("traffic") and k["bwd"].get("traffic"):
            how = "collected by the bench itself in this run" if str(r.get("traffic_source", "")).startswith("collected in this run") else "committed passes at HEAD"
            L.append(f"| PMC traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc passes; {how}), forward / backward | {r['traffic'] / 1e6:.2f} MB vs {r['algorithmic_bytes'] / 1e6:.2f} MB (**{r['traffic'] / r['algorithmic_bytes']:.3f}**) / {k['bwd']['traffic'] / 1e6:.2f} vs {k['bwd']['algorithmic_bytes'] / 1e6:.2f} MB (**{k['bwd']['traffic'] / k['bwd']['algorithmic_bytes']:.3f}**) |")
        h = r.get("hbm_resident") or {}
        for kind in ("cosine-similarity", "tag-euclidean"):
            if kind in h:
                e = h[kind]
                sp = e.get("fwd_us_min_median_max") or {}
                L.append(f"| **HBM-resident** list (BERT-base, {2 * h['elements'] * 4 / 1e6:.1f} MB per forward launch), {kind}, same run | fwd {e['fwd_us']:.1f} us = **{e['frac']:.3f}**" +
                         (f" (min / median / max of 30: {sp.get('min')} / {sp.get('median')} / {sp.get('max')} us)" if sp else "") +
                         f", stage {e['stage_frac']:.3f}; bwd {e['bwd_us']:.1f} us = **{e['bwd_frac']:.3f}** |")
        pa = b.get("parity") or {}
        if pa.get("iterate"):
            L.append(f"| `parity`: the timed configuration at the reference's iterate k = {pa['iterate']} of its 24 000-iteration CPU run | loss {pa['loss_hip']:.6f} vs {pa['loss_reference']:.6f} (rel {pa['loss_rel_err']:.1e}, tolerance {pa['loss_tolerance']:.1e} = 10x the reference's own kink sensitivity there); sign(d total/dx) equal on **{pa['sign_agreement']:.4f}** of the pixels (reference vs itself 16 ulp away: {pa['reference_twin_agreement']:.4f}), |g|-weighted {pa['weighted_sign_agreement']:.5f} ({pa['reference_twin_weighted_agreement']:.4f}) |")
    for name, label in (("r5_bench_n1.json", "default flags (200 steps)"), ("r5_bench_n1_4trials_in_flight.json", "four restarts in flight on one GPU")):
        x = _json_line(name)
        if x:
            L.append(f"| {label} (`{name}`) | **{x['value']:.1f}** it/s ({x['ms_per_step']:.3f} ms per step) |")
    b8 = _json_line("r5_bench_8ranks_one_gpu.json")
    if b8:
        L.append(f"| rehearsal of the 8-rank launch on ONE GPU (`torch.distributed.run --nproc-per-node 8`, {b8.get('collective_backend')}, oversubscribed; `r5_bench_8ranks_one_gpu.json`) | {b8['value']:.1f} it/s aggregate, every rank `{b8['launch_mode']}`, per-rank ms/step {b8.get('per_rank_ms_per_step')}, skew {b8.get('rank_skew')}; staged start: {b8.get('staged_start')} |")
    L.append("")

    # --- kernel A in the loop at the three BASELINE sizes
    rows = []
    for label, fname, lst, fwd_key, bwd_key, pmc_tag in (
            ("ResNet-18 (configs[1], bench command)", "r5_bench_kernel_summary.txt", "resnet18", "gm_fwd_kernel<0, false>", "gm_bwd_kernel<0, false, false>", "resnet18"),
            ("ResNet-50 B = 8 see-through (configs[2])", "r5_config3_resnet50_seethrough_kernel_summary.txt", "resnet50", "gm_fwd_kernel<4, true>", "gm_bwd_kernel<4, false, false>", None),
            ("BERT-base TAG (configs[4])", "r5_config5_bert_tag_kernel_summary.txt", "bert_base", "gm_fwd_kernel<6, true>", "gm_bwd_kernel<6, true, true>", "bert")):
        ks = _kernel_summary(fname)
        if not ks or fwd_key not in ks:
            continue
        n = N_LIST[lst]
        f_us, b_us = ks[fwd_key][1], ks[bwd_key][1]
        fin = ks.get("gm_finalize_kernel", (0, 0.0))[1]
        cell = f"| {label} | {2 * n * 4 / 1e6:.1f} | `{fwd_key}` {f_us:.2f} us = **{_frac(2 * n * 4, f_us):.2f}** (stage + {fin:.2f} us = {_frac(2 * n * 4, f_us + fin):.2f}) | `{bwd_key}` {b_us:.2f} us = **{_frac(3 * n * 4, b_us):.2f}** |"
        if pmc_tag:
            pf = _pmc_ratio(f"r5_pmc_fetch_{pmc_tag}_pmc_summary.csv", f"r5_pmc_write_{pmc_tag}_pmc_summary.csv", "gm_fwd_kernel", 2 * n * 4)
            pb = _pmc_ratio(f"r5_pmc_fetch_{pmc_tag}_pmc_summary.csv", f"r5_pmc_write_{pmc_tag}_pmc_summary.csv", "gm_bwd_kernel", 3 * n * 4)
            cell += f" {fmt(pf[1], 3) if pf else 'n/a'} / {fmt(pb[1], 3) if pb else 'n/a'} |"
        else:
            cell += " (round 3: 1.00 / 1.00) |"
        rows.append(cell + f" `{fname}` |")
    if rows:
        L += ["**Kernel A inside the attack loop** (rocprofv3 kernel trace of the whole run, average dispatch duration; fraction of the 8 TB/s peak on ALGORITHMIC bytes 2 N 4 / 3 N 4; PMC = (2 x FETCH_SIZE + WRITE_SIZE) / algorithmic bytes from separate `--pmc` passes over `scripts/pmc_target.py` with the AUTO cache policy, i.e. the non-temporal instantiations at BERT size):", "",
              "| list | forward MB | forward | backward | PMC traffic ratio fwd / bwd | file |", "|---|---|---|---|---|---|"] + rows + [""]

    # --- read ceiling
    probe = _jsonl("r5_read_ceiling_probe.jsonl")
    rows = []
    for lst in ("resnet18", "resnet50", "bert_base"):
        ka = next((p for p in probe if p.get("list") == lst and p.get("kernel", "").startswith("kernel A forward (cosine), model list")), None)
        one = next((p for p in probe if p.get("list") == lst and "one tensor" in p.get("kernel", "")), None)
        dg = next((p for p in probe if p.get("list") == lst and p.get("kernel", "").startswith("diag_read, dispatch")), None)
        if ka and dg:
            best_w = min(v for v in (dg.get("behind_writer_us"), dg.get("nt_behind_writer_us")) if v)
            best = min(v for v in (dg.get("warm_us"), dg.get("nt_warm_us")) if v)
            rows.append(f"| {lst} ({ka['bytes'] / 1e6:.1f} MB) | {dg['warm_us']} / {dg['nt_warm_us']} | {dg['behind_writer_us']} / {dg['nt_behind_writer_us']} | {ka['warm_us']} | {ka['behind_writer_us']} ({one['behind_writer_us'] if one else 'n/a'}) | **{best_w / ka['behind_writer_us']:.2f}** ({best / ka['warm_us']:.2f}) |")
    if rows:
        L += ["**The ceiling kernel A's forward is priced against** (`r5_read_ceiling_probe.jsonl`; `diag_read` = kernel A's persistent grid of 512 workgroups and eight staged 16-byte loads per lane over two buffers, two multiply-adds per element, outside the library; all columns: dispatch start/stop events, median of 30, us):", "",
              "| list | bare read warm: plain / non-temporal loads | bare read behind a writer: plain / nt | kernel A forward warm | kernel A forward behind a writer (same bytes as ONE tensor: full chunks only) | best bare read / kernel A, behind a writer (warm) |", "|---|---|---|---|---|---|"] + rows + [""]

    rows = []
    for lst in ("resnet18", "resnet50", "bert_base"):
        rw = next((p for p in probe if p.get("list") == lst and p.get("kernel", "").startswith("diag_rw")), None)
        kb = next((p for p in probe if p.get("list") == lst and p.get("kernel", "").startswith("kernel A backward")), None)
        if rw and kb:
            best = min(v for k, v in rw.items() if k.endswith("_us") and v)
            rows.append(f"| {lst} ({kb['bytes'] / 1e6:.1f} MB) | {rw['plain_warm_us']} / {rw['plain_behind_writer_us']} | {rw['nt_loads_warm_us']} / {rw['nt_loads_behind_writer_us']} | "
                        f"{rw['nt_loads_and_stores_warm_us']} / {rw['nt_loads_and_stores_behind_writer_us']} | {kb['us']} ({kb['frac_of_8TBps']:.2f} of 8 TB/s) | **{best / kb['us']:.2f}** |")
    if rows:
        L += ["The same for the read-two-write-one shape of kernel A's backward and the multi-tensor kernels (`diag_rw`: one workgroup per chunk, four staged 16-byte loads per lane and buffer, 16-byte stores; warm / behind a writer, us):", "",
              "| list | plain | non-temporal loads | non-temporal loads and stores | kernel A backward right after its forward | best bare launch / kernel A backward |", "|---|---|---|---|---|---|"] + rows + [""]

    # --- multi-tensor kernels
    rows = []
    sa = _by_grid("r5_mt_kernel_probe_kernel_by_grid.csv")
    wg = {2893: "resnet18", 6334: "resnet50", 21110: "bert_base"}
    ops = {"mt_kernel<0": ("a + alpha b", 3), "mt_kernel<2": ("alpha a", 2)}
    for kname, wgs, calls, avg, med in sa:
        for pref, (label, words) in ops.items():
            if kname.startswith(pref) and wgs in wg:
                n = N_LIST[wg[wgs]]
                rows.append(f"| {label} | {wg[wgs]} | one launch, {wgs} workgroups | {avg:.2f} | **{_frac(words * n * 4, avg):.2f}** | stand-alone, back to back (`r5_mt_kernel_probe_kernel_by_grid.csv`) |")
    for fname, lst in (("r5_fedavg_resnet18_kernel_by_grid.csv", "resnet18"), ("r5_fedavg_resnet50_kernel_by_grid.csv", "resnet50")):
        g = _by_grid(fname)
        n = N_LIST[lst]
        for pref, (label, words) in ops.items():
            hit = [x for x in g if x[0].startswith(pref)]
            if hit:
                us = sum(x[3] for x in hit)
                rows.append(f"| {label} | {lst} | {len(hit)} launch(es) | {us:.2f} | **{_frac(words * n * 4, us):.2f}** | inside a FedAvg attack iteration (`{fname}`) |")
        minus = [x for x in g if x[0].startswith("mt_kernel<1")]
        if minus:
            us = sum(x[3] for x in minus)
            rows.append(f"| (a + alpha b) - c | {lst} | {len(minus)} launch(es) (three pointer lists: 112 tensors per launch) | {us:.2f} | **{_frac(4 * n * 4, us):.2f}** | inside a FedAvg attack iteration (`{fname}`) |")
        bw = [x for x in g if x[0].startswith("gm_bwd_kernel")]
        if bw:
            rows.append(f"| kernel A backward, for comparison (3 N 4) | {lst} | 1 | {bw[0][3]:.2f} | {_frac(3 * n * 4, bw[0][3]):.2f} | same trace |")
    if rows:
        L += ["**Multi-tensor kernels (SURVEY section 8 f2 / f3)** -- rocprofv3 dispatch durations, fraction of 8 TB/s on algorithmic bytes (3 N 4 / 2 N 4 / 4 N 4):", "",
              "| form | list | launches per call | us per call | of peak | where |", "|---|---|---|---|---|---|"] + rows + [""]
        pr = {lst: _pmc_ratio(f"r5_pmc_fetch_mt_{t}_pmc_summary.csv", f"r5_pmc_write_mt_{t}_pmc_summary.csv", "mt_kernel<0", 3 * N_LIST[lst] * 4, 1)
              for lst, t in (("resnet50", "resnet50"), ("bert_base", "bert"))}
        if all(pr.values()):
            L += [f"PMC traffic of `a + alpha b` per call (one dispatch per call; non-temporal stores at BERT size; `r5_pmc_{{fetch,write}}_mt_{{resnet50,bert}}_pmc_summary.csv`): "
                  f"ResNet-50 {pr['resnet50'][0] / 1e6:.1f} MB = {pr['resnet50'][1]:.3f} of the algorithmic bytes, BERT-base {pr['bert_base'][0] / 1e6:.1f} MB = {pr['bert_base'][1]:.3f}.", ""]

    # --- kernels B and C
    sp = _by_grid("r5_step_prior_probe_kernel_by_grid.csv")
    c3 = _kernel_summary("r5_config3_resnet50_seethrough_kernel_summary.txt")
    rows = []
    for kname, wgs, calls, avg, med in sp:
        if kname.startswith("candidate_step") or kname.startswith("tv_norm"):
            B = 8 if wgs in (1176, 392, 1568, 2048) else 1
            nbytes = (8 if "true, false" in kname else 8) * B * 150528 * 4 if kname.startswith("candidate_step") else 2 * B * 150528 * 4
            rows.append(f"| `{kname}` | B = {B} ({wgs} workgroups) | {avg:.2f} | {nbytes / 1e6:.1f} MB -> {nbytes / avg / 1e3:.0f} GB/s |")
    if rows:
        L += ["**Kernels B and C alone** (rocprofv3, `r5_step_prior_probe_kernel_by_grid.csv`; 3 x 224 x 224 images; kernel B: 8 P 4 bytes -- hard sign with the best copy taken, or plain Adam with the noise operand; kernel C: 2 P 4):", "",
              "| kernel | size | avg us | bytes -> rate |", "|---|---|---|---|"] + rows + [""]
    if c3:
        kb = next((v for k, v in c3.items() if k.startswith("candidate_step")), None)
        kc = next((v for k, v in c3.items() if k.startswith("tv_norm")), None)
        if kb and kc:
            L += [f"Inside the see-through loop at B = 8 (`r5_config3_resnet50_seethrough_kernel_summary.txt`): kernel B {kb[1]:.2f} us (round 4: 11.4-11.9), kernel C {kc[1]:.2f} us (round 4: 8.2-8.5).", ""]

    # --- in-loop durations, bench command
    ks = _kernel_summary("r5_bench_kernel_summary.txt")
    if ks:
        L += ["**In-loop durations of our kernels**, rocprofv3 kernel trace of the bench command at HEAD (`r5_bench_kernel_summary.txt`):", "",
              "| kernel | calls | avg us |", "|---|---|---|"]
        for name, (calls, avg) in sorted(ks.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:14]:
            L.append(f"| `{name}` | {calls} | {avg:.2f} |")
        L.append("")
    c = _json_line("r5_1trial_gap_census.json")
    if c and c.get("per_queue"):
        q = next(iter(c["per_queue"].values()))
        L += [f"Gap census of the replayed iteration under rocprofv3 (`r5_1trial_gap_census.json`): {q['dispatches_per_iter']:.0f} dispatches, {q['kernel_us_per_iter']:.0f} us of dispatch durations + {q['gap_us_per_iter']:.0f} us of gaps = {q['wall_us_per_iter']:.0f} us per iteration under the profiler.", ""]

    # --- the end-of-run statistics: same-GPU control and the larger HIP sample
    ctl, big, ref32 = _json_line("r5_control_same_gpu_torch_8starts.json"), _json_line("r5_hip_64starts_1000its.json"), _json_line("r5_reference_cpu_1000its_more_starts.json")
    more = _json_line("r5_torch_on_gpu_24_more_starts.json")
    if ctl and big:
        import math

        marks = (373, 380, 624, 630, 874, 880, 999)
        files = "`r5_control_same_gpu_torch_8starts.json`, `r5_hip_64starts_1000its.json`" + (", `r5_reference_cpu_1000its_more_starts.json`" if ref32 else "") + (", `r5_torch_on_gpu_24_more_starts.json`" if more else "")
        L += [f"**End-of-run statistics of configs[1] at 1 000 iterations** (mean +- sd over starts <= 16 ulp apart; {files}).  Last column: difference of the two GPU implementations' means and of HIP's from the larger CPU sample, in standard errors:", "",
              "| quantity | unmodified reference, CPU (n = 8) |" + (f" unmodified reference, CPU (n = {ref32['n']}) |" if ref32 else "") +
              " oracle/restate.py with PyTorch-ROCm ops on the GPU (n = 8) |" + (" ... (n = 32) |" if more else "") + " HIP, the same 8 starts | HIP, 64 starts | HIP - torch on GPU / HIP - CPU reference |",
              "|---|---|" + ("---|" if ref32 else "") + "---|" + ("---|" if more else "") + "---|---|---|"]
        for key in ("loss@373", "loss@624", "loss@999", "opt_value", "psnr"):
            q = ctl["quantities"][key]
            hb = big["hip"][key]
            cells = [f"{q['reference_cpu_mean']:.6f} +- {q['reference_cpu_sd']:.6f}"]
            ref_stat = (q["reference_cpu_mean"], q["reference_cpu_sd"], 8)
            if ref32:
                e = ref32["quantities"][key]
                cells.append(f"{e['mean']:.6f} +- {e['sd']:.6f}")
                ref_stat = (e["mean"], e["sd"], ref32["n"])
            cells.append(f"{q['torch_on_gpu_mean']:.6f} +- {q['torch_on_gpu_sd']:.6f}")
            torch_stat = (q["torch_on_gpu_mean"], q["torch_on_gpu_sd"], 8)
            if more:
                first = [r["history"][marks.index(int(key[5:]))] if key.startswith("loss@") else r[key] for r in ctl["runs"]["torch_on_gpu"]]
                vals = first + list(more["torch_on_gpu"][key]["values"])
                mean = sum(vals) / len(vals)
                sd = math.sqrt(sum((v - mean) ** 2 for v in vals) / (len(vals) - 1))
                cells.append(f"{mean:.6f} +- {sd:.6f}")
                torch_stat = (mean, sd, len(vals))
            cells += [f"{q['hip_mean']:.6f} +- {q['hip_sd']:.6f}", f"{hb['mean']:.6f} +- {hb['sd']:.6f}"]

            def z(a, b):
                return (a[0] - b[0]) / math.sqrt(a[1] ** 2 / a[2] + b[1] ** 2 / b[2])

            hip_stat = (hb["mean"], hb["sd"], hb["n"])
            cells.append(f"{z(hip_stat, torch_stat):+.1f} / {z(hip_stat, ref_stat):+.1f}")
            L.append(f"| {key} | " + " | ".join(cells) + " |")
        L.append("")
    L.append(END)
    return "\n".join(L)


def main():
    block = build()
    if "--write" in sys.argv:
        path = os.path.join(ROOT, "DESIGN.md")
        text = open(path).read()
        if BEGIN in text and END in text:
            text = text[: text.index(BEGIN)] + block + text[text.index(END) + len(END):]
        else:
            raise SystemExit("DESIGN.md has no GENERATED markers")
        open(path, "w").write(text)
    else:
        print(block)


if __name__ == "__main__":
    main()
