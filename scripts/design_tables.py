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
BEGIN = "<!-- BEGIN GENERATED: round-4 tables (scripts/design_tables.py) -->"
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


def build():
    L = [BEGIN, "", "*(generated from the files named in the right-hand column; do not edit by hand)*", ""]
    b = _json_line("r4_bench_n1.json")
    b4 = _json_line("r4_bench_n1_4trials_in_flight.json")
    b2 = _json_line("r4_bench_2ranks_one_gpu.json")
    L += ["| quantity | value | file |", "|---|---|---|"]
    if b:
        r, k = b["roofline"], b["kernels"]
        L.append(f"| attack iterations / s, 1 GPU, one trial, {b['launch_mode']} | **{b['value']:.1f}** ({b['ms_per_step']:.3f} ms / iteration; eager launches: {fmt(b.get('eager_ms_per_step'), 2)} ms) | `r4_bench_n1.json` |")
        L.append(f"| kernel A forward, HIP ext-launch events, ResNet-18 list ({r['algorithmic_bytes'] / 1e6:.1f} MB, Infinity-Cache resident) | {r['avg_launch_us']:.2f} us -> {r['achieved']:.0f} GB/s = **{r['frac']:.3f}** of 8 TB/s; stage with finalize {r['stage_us']:.2f} us = {r['stage_frac']:.3f}; device span inside graph replay {fmt(r.get('timed_region_span_us'), 2)} us | same |")
        L.append(f"| kernel A backward, same method ({k['bwd']['algorithmic_bytes'] / 1e6:.1f} MB) | {k['bwd']['avg_us']:.2f} us -> {k['bwd']['achieved_GBs']:.0f} GB/s = **{k['bwd']['frac_of_hbm_peak']:.3f}** | same |")
        L.append(f"| PMC traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE), forward / backward | {r['traffic'] / 1e6:.2f} MB vs {r['algorithmic_bytes'] / 1e6:.2f} MB ({r['traffic'] / r['algorithmic_bytes']:.3f}) / {k['bwd']['traffic'] / 1e6:.2f} MB vs {k['bwd']['algorithmic_bytes'] / 1e6:.2f} MB ({k['bwd']['traffic'] / k['bwd']['algorithmic_bytes']:.3f}) | `{', '.join(sorted(set(re.findall(r'r[0-9]_pmc_[a-z]+_pmc_summary.csv', r.get('traffic_source', '')))))}` |")
        h = r.get("hbm_resident") or {}
        for kind in ("cosine-similarity", "tag-euclidean"):
            if kind in h:
                e = h[kind]
                L.append(f"| **HBM-resident** list (BERT-base, {h['elements'] / 1e6:.2f} M elements, {2 * h['elements'] * 4 / 1e6:.1f} MB per forward launch), {kind}, in the SAME run | fwd {e['fwd_us']:.1f} us = **{e['frac']:.3f}** (stage {e['stage_frac']:.3f}" + (f"; not behind a writer: {e['fwd_not_behind_a_writer_us']:.1f} us = {e['frac_not_behind_a_writer']:.3f}" if "frac_not_behind_a_writer" in e else "") + f"); bwd {e['bwd_us']:.1f} us = **{e['bwd_frac']:.3f}** | same (`roofline.hbm_resident`) |")
        c = b.get("cpu_baseline")
        if c:
            anchor = c.get("anchor") or {}
            L.append(f"| CPU baseline in the same run ({c['kind']}; {c['cores']} of {c['host_cpu_count']} logical cores) | {c['value']:.2f} it/s" +
                     (f"; port / unmodified reference on equal threads = {anchor.get('port_over_reference')} (`{anchor.get('file')}`)" if anchor else "") + " | same |")
    if b4:
        L.append(f"| four restarts in flight on one GPU (streams on four different hardware pipes) | **{b4['value']:.1f}** it/s ({b4['ms_per_step']:.2f} ms per round of four) | `r4_bench_n1_4trials_in_flight.json` |")
    if b2:
        L.append(f"| two ranks sharing the one GPU (`bench.py --gpus 2`, {b2.get('collective_backend')}, oversubscribed: functional check) | {b2['value']:.1f} it/s; per-rank ms/step {b2.get('per_rank_ms_per_step')}, skew {b2.get('rank_skew')} | `r4_bench_2ranks_one_gpu.json` |")
    sweep = _json_line("r4_cpu_thread_sweep.json")
    if sweep:
        L.append("| CPU thread sweep on the GPU box's host (port, it/s by threads) | " + ", ".join(f"{t}: {v}" for t, v in sweep["iterations_per_s"].items()) + f" -> {sweep['best_threads']} threads | `r4_cpu_thread_sweep.json` |")
    L.append("")
    # --- in-loop kernel durations
    ks = _kernel_summary("r4_bench_kernel_summary.txt")
    if ks:
        L += ["In-loop durations of our kernels, rocprofv3 kernel trace of the bench command (`r4_bench_kernel_summary.txt`; rocprofv3 adds ~2-3 us to every dispatch it times, see the node-cost probe below):", "",
              "| kernel | calls | avg us |", "|---|---|---|"]
        for name, (calls, avg) in sorted(ks.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:12]:
            L.append(f"| `{name}` | {calls} | {avg:.2f} |")
        L.append("")
    # --- gap census
    rows = []
    for tag, label in (("r4_1trial_gap_census.json", "round-3 form (BatchNorm, ReLU, residual add separate)"),
                       ("r4_1trial_fused_gap_census.json", "BatchNorm + residual + ReLU in kernel E"),
                       ("r4_1trial_head_gap_census.json", "... and the second gradient of each BatchNorm input folded into the launch (HEAD, default)"),
                       ("r4_1trial_gemm0_gap_census.json", "MIOPEN_DEBUG_CONV_GEMM=0")):
        c = _json_line(tag)
        if c and c.get("per_queue"):
            q = next(iter(c["per_queue"].values()))
            rows.append(f"| {label} | {q['dispatches_per_iter']:.0f} | {q['kernel_us_per_iter']:.0f} | {q['gap_us_per_iter']:.0f} | {q['wall_us_per_iter']:.0f} | `{tag}` |")
    if rows:
        L += ["Gap census of the replayed iteration under rocprofv3 (per iteration, one trial):", "",
              "| variant | dispatches | sum of dispatch durations, us | sum of gaps, us | wall, us | file |", "|---|---|---|---|---|---|"] + rows + [""]
    probe = _jsonl("r4_node_cost_probe.jsonl")
    if probe:
        L += ["Cost of ONE node of a replayed hipGraph without a profiler (`r4_node_cost_probe.jsonl`, 600-node chains):", "",
              "| chain | us per node, graph replay | us per node, eager |", "|---|---|---|"]
        for p in probe:
            L.append(f"| {p['chain']} | {p['graph_us_per_node']:.2f} | {p['eager_us_per_node']:.2f} |")
        L.append("")
    pipes = _jsonl("r4_inflight_pipes_probe.jsonl")
    if pipes:
        L += ["Trials in flight by WHICH streams carry them (`r4_inflight_pipes_probe.jsonl`; streams numbered in creation order, fresh process each):", "",
              "| busy streams | trial-iterations / s | ms per round |", "|---|---|---|"]
        for p in pipes:
            if "use" in p and "trial_iterations_per_s" in p:
                L.append(f"| {', '.join(str(u) for u in p['use'])} | {p['trial_iterations_per_s']:.1f} | {p['ms_per_round']:.2f} |")
        L.append("")
    # --- kernel D backward A/B
    t1, t0 = _kernel_summary("r4_config3_fused_tap_1_kernel_summary.txt"), _kernel_summary("r4_config3_fused_tap_0_kernel_summary.txt")
    if t1 and t0:
        L += ["DeepInversion backward inside kernel E's backward launch vs round 3's launch per layer (ResNet-50, B = 8, same box; `r4_config3_fused_tap_{1,0}_kernel_summary.txt`):", "",
              "| kernel | fused: calls, avg us | separate: calls, avg us |", "|---|---|---|"]
        for name in ("bn_bwd_acc_kernel", "bn_eval_bwd_kernel", "bn_eval_bwd_bwd_kernel", "bn_eval_fwd_kernel", "bn_finalize_kernel<1024>"):
            a, c = t1.get(name), t0.get(name)
            L.append(f"| `{name}` | {'absent' if a is None else f'{a[0]}, {a[1]:.2f}'} | {'absent' if c is None else f'{c[0]}, {c[1]:.2f}'} |")
        f_plain, f_tap = _pmc("r4_pmc_fetch_bneval_plain_pmc_summary.csv", "bn_eval_bwd_kernel"), _pmc("r4_pmc_fetch_bneval_tap_pmc_summary.csv", "bn_eval_bwd_kernel")
        w_plain, w_tap = _pmc("r4_pmc_write_bneval_plain_pmc_summary.csv", "bn_eval_bwd_kernel"), _pmc("r4_pmc_write_bneval_tap_pmc_summary.csv", "bn_eval_bwd_kernel")
        if None not in (f_plain, f_tap, w_plain, w_tap):
            L += ["", f"PMC per `bn_eval_bwd_kernel` launch (average over the 53 layers, KB): FETCH_SIZE {f_plain:.0f} without / {f_tap:.0f} with the term riding along; WRITE_SIZE {w_plain:.0f} / {w_tap:.0f} (`r4_pmc_{{fetch,write}}_bneval_{{plain,tap}}_pmc_summary.csv`)."]
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
