"""Gap census of the replayed attack iteration from a rocprofv3 kernel trace.

    python scripts/gap_census.py <rocprof_dir or kernel_trace.csv> <out_prefix> [--iters 40] [--skip-tail 0] [--label text]

Reads `*kernel_trace.csv` (rocprofv3 --kernel-trace --output-format csv), finds the hardware queues that carry attack
iterations (those with `candidate_step_kernel` dispatches: one per iteration per trial), cuts each queue's timeline at the
end of every candidate step, and reports for the last `--iters` complete iterations of every such queue

  * per iteration: dispatches, sum of kernel durations (End - Start of the dispatch), sum of gaps (Start of the next
    dispatch on the queue - latest End so far), wall time of the iteration;
  * the same split by kernel FAMILY -- as the dispatch itself (duration), as the PREDECESSOR of a gap (what the queue
    waited behind) and as the SUCCESSOR of a gap (what was slow to start);
  * a histogram of the gaps;
  * across queues: how many queues / streams carried iterations, and the fraction of wall time in which k queues were
    busy at once (for the trials-in-flight question).

Writes <out_prefix>.json, <out_prefix>.txt and <out_prefix>_one_iteration.csv (the raw rows of one iteration, times
relative to its first dispatch) -- small enough for profiles/.
"""
import argparse
import collections
import csv
import json
import os
import re
import sys

FAMILIES = [  # first match wins
    ("ours: kernel A (gm_*)", r"gm_(fwd|bwd|finalize|pack)_kernel"),
    ("ours: kernel B/C, commit (step, tv, loss)", r"candidate_step(_vec4)?_kernel|tv_norm(_vec4)?_kernel|loss_commit_kernel|grad_sumsq|grad_norm_finalize|state_reset"),
    ("ours: kernel D (bn_sums/finalize/bwd)", r"bn_sums_kernel|bn_finalize_kernel|bn_bwd_kernel|bn_bwd_acc_kernel"),
    ("ours: kernel E (bn_eval_*)", r"bn_eval_"),
    ("ours: kernel F (ln_*)", r"ln_(fwd|bwd)"),
    ("ours: multi-tensor (mt_*)", r"mt_kernel|orthogonality_kernel|psnr_mse_kernel|psnr_finalize"),
    ("runtime: copyBuffer / fillBuffer", r"__amd_rocclr_"),
    ("MIOpen: Winograd / asm direct conv", r"miopenSp3AsmConv|miopenGcnAsm|gcnAsmConv|conv\d+x\d+u|MIOpenConv"),
    ("MIOpen: Im2Col / Col2Im", r"Im2d2Col|Col2Im|Im3d2Col|Col2Im3d"),
    ("MIOpen: layout transposes", r"batched_transpose|transpose_NCHW|transpose_NHWC"),
    ("MIOpen: implicit GEMM (igemm / CK)", r"igemm_|kernel_grouped_conv|kernel_batched_gemm|gridwise|ck::|_ZN2ck"),
    ("MIOpen: SubTensorOp / tensor ops", r"SubTensorOp|OpTensor|ScaleTensor|SetTensor"),
    ("rocBLAS/Tensile GEMM", r"Cijk_|rocblas|gemv|gemm"),
    ("ATen: fill (zeros / zero_)", r"FillFunctor"),
    ("ATen: add (accumulation, residual)", r"CUDAFunctor_add|CUDAFunctorOnSelf_add"),
    ("ATen: relu / threshold / clamp", r"clamp|threshold|relu"),
    ("ATen: other elementwise", r"elementwise_kernel|vectorized_elementwise"),
    ("ATen: reductions", r"reduce_kernel"),
    ("ATen: pooling", r"pool"),
    ("ATen: softmax / nll / scatter-gather / index", r"softmax|nll_loss|scatter|gather|index"),
    ("ATen: copy / cat / other", r"at::native|at_cuda|CatArray"),
]
COMPILED = [(name, re.compile(rx)) for name, rx in FAMILIES]
GAP_BINS = [(0, 0.5), (0.5, 1), (1, 2), (2, 3), (3, 5), (5, 10), (10, 50), (50, 1e12)]


def family(name):
    for fam, rx in COMPILED:
        if rx.search(name):
            return fam
    return "other"


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")[:100]


def find_trace(path):
    if os.path.isfile(path):
        return path
    for root, _, files in os.walk(path):
        for f in files:
            if f.endswith("kernel_trace.csv"):
                return os.path.join(root, f)
    raise SystemExit(f"no *kernel_trace.csv under {path}")


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append(dict(queue=r.get("Queue_Id", "0"), stream=r.get("Stream_Id", ""), name=r["Kernel_Name"],
                             start=int(r["Start_Timestamp"]), end=int(r["End_Timestamp"]),
                             grid=int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1),
                             wg=int(r.get("Workgroup_Size_X", 0) or 0) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1),
                             scratch=int(r.get("Scratch_Size", 0) or 0)))
    return rows


def census_queue(rows, iters, skip_tail):
    rows = sorted(rows, key=lambda r: r["start"])
    cuts = [i for i, r in enumerate(rows) if re.search(r"candidate_step(_vec4)?_kernel", r["name"])]
    # one iteration = (cut[k-1], cut[k]]; joint attacks step two tensors per iteration -> cut at the last step of a burst
    merged = []
    for c in cuts:
        if merged and c - merged[-1] <= 3:
            merged[-1] = c
        else:
            merged.append(c)
    cuts = merged
    if skip_tail:
        cuts = cuts[:-skip_tail]
    segments = [(cuts[k - 1] + 1, cuts[k] + 1) for k in range(1, len(cuts))][-iters:]
    per_iter = []
    fam_dur = collections.defaultdict(float)
    fam_calls = collections.Counter()
    fam_gap_after = collections.defaultdict(float)
    fam_gap_before = collections.defaultdict(float)
    fam_gap_after_n = collections.Counter()
    hist = collections.Counter()
    hist_us = collections.defaultdict(float)
    for lo, hi in segments:
        seg = rows[lo:hi]
        prev_end = rows[lo - 1]["end"]  # the previous iteration's candidate step
        prev_fam = family(rows[lo - 1]["name"])
        dur = gap = 0.0
        overlapped = 0
        for r in seg:
            d = (r["end"] - r["start"]) / 1e3
            g = (r["start"] - prev_end) / 1e3
            fam = family(r["name"])
            fam_dur[fam] += d
            fam_calls[fam] += 1
            dur += d
            if g < 0:
                overlapped += 1
                g = 0.0
            gap += g
            fam_gap_after[prev_fam] += g
            fam_gap_after_n[prev_fam] += 1
            fam_gap_before[fam] += g
            for a, b in GAP_BINS:
                if a <= g < b:
                    hist[(a, b)] += 1
                    hist_us[(a, b)] += g
                    break
            if r["end"] >= prev_end:
                prev_end, prev_fam = r["end"], fam
        per_iter.append(dict(dispatches=len(seg), kernel_us=dur, gap_us=gap, overlapped=overlapped,
                             wall_us=(seg[-1]["end"] - rows[lo - 1]["end"]) / 1e3))
    n = max(len(segments), 1)
    fams = []
    for fam in sorted(fam_calls, key=lambda k: -(fam_dur[k] + fam_gap_before[k])):
        c = fam_calls[fam]
        fams.append(dict(family=fam, calls_per_iter=round(c / n, 2), kernel_us_per_iter=round(fam_dur[fam] / n, 2),
                         avg_kernel_us=round(fam_dur[fam] / c, 2),
                         gap_before_us_per_iter=round(fam_gap_before[fam] / n, 2), avg_gap_before_us=round(fam_gap_before[fam] / c, 2),
                         gap_after_us_per_iter=round(fam_gap_after[fam] / n, 2),
                         avg_gap_after_us=round(fam_gap_after[fam] / max(fam_gap_after_n[fam], 1), 2)))
    mean = lambda key: sum(p[key] for p in per_iter) / n  # noqa: E731
    total_gaps = sum(hist.values())
    one = []
    if segments:
        lo, hi = segments[len(segments) // 2]
        t0 = rows[lo]["start"]
        prev_end = rows[lo - 1]["end"]
        for r in rows[lo:hi]:
            one.append([round((r["start"] - t0) / 1e3, 3), round((r["end"] - r["start"]) / 1e3, 3),
                        round((r["start"] - prev_end) / 1e3, 3), r["grid"], r["wg"], r["scratch"], family(r["name"]), short(r["name"])])
            prev_end = max(prev_end, r["end"])
    return dict(iterations=len(segments), dispatches_per_iter=round(mean("dispatches"), 1), kernel_us_per_iter=round(mean("kernel_us"), 1),
                gap_us_per_iter=round(mean("gap_us"), 1), wall_us_per_iter=round(mean("wall_us"), 1),
                avg_kernel_us=round(mean("kernel_us") / max(mean("dispatches"), 1), 3), avg_gap_us=round(mean("gap_us") / max(mean("dispatches"), 1), 3),
                overlapped_per_iter=round(mean("overlapped"), 2), families=fams,
                gap_histogram=[dict(bin_us=f"{a}-{b if b < 1e11 else 'inf'}", count_per_iter=round(hist[(a, b)] / n, 2),
                                    share_of_gaps=round(hist[(a, b)] / max(total_gaps, 1), 4), us_per_iter=round(hist_us[(a, b)] / n, 2))
                               for a, b in GAP_BINS]), one, segments and (rows[segments[0][0]]["start"], rows[segments[-1][1] - 1]["end"])


def concurrency(rows_by_queue, window):
    """Fraction of the window during which exactly k of the queues have a dispatch executing."""
    if not window:
        return None
    lo, hi = window
    events = []
    for q, rows in rows_by_queue.items():
        merged_end = None
        for r in sorted(rows, key=lambda r: r["start"]):
            if r["end"] < lo or r["start"] > hi:
                continue
            s, e = max(r["start"], lo), min(r["end"], hi)
            if merged_end is not None and s < merged_end:  # overlapping dispatches of one queue count once
                s = merged_end
            if e > s:
                events.append((s, 1)), events.append((e, -1))
                merged_end = e
    events.sort()
    busy = collections.defaultdict(int)
    level, last = 0, lo
    for t, d in events:
        busy[level] += t - last
        level, last = level + d, t
    busy[level] += hi - last
    span = max(hi - lo, 1)
    return {str(k): round(v / span, 4) for k, v in sorted(busy.items())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("out")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--skip-tail", type=int, default=0, help="drop the last N iterations of every queue (an eager tail)")
    ap.add_argument("--label", default="")
    args = ap.parse_args()
    trace = find_trace(args.src)
    rows = load(trace)
    by_queue = collections.defaultdict(list)
    for r in rows:
        by_queue[r["queue"]].append(r)
    result = dict(label=args.label, trace=os.path.basename(trace), dispatches=len(rows),
                  queues={q: dict(dispatches=len(v), streams=sorted({r["stream"] for r in v}),
                                  candidate_steps=sum(bool(re.search(r"candidate_step(_vec4)?_kernel", r["name"])) for r in v)) for q, v in by_queue.items()},
                  per_queue={})
    one_iteration = None
    windows = []
    for q, v in by_queue.items():
        if sum(bool(re.search(r"candidate_step(_vec4)?_kernel", r["name"])) for r in v) < 3:
            continue
        res, one, window = census_queue(v, args.iters, args.skip_tail)
        result["per_queue"][q] = res
        if one_iteration is None:
            one_iteration = one
        if window:
            windows.append(window)
    if windows:
        common = (max(w[0] for w in windows), min(w[1] for w in windows))
        if common[1] > common[0]:
            attack_queues = {q: by_queue[q] for q in result["per_queue"]}
            result["queues_busy_at_once"] = concurrency(attack_queues, common)
            result["common_window_ms"] = round((common[1] - common[0]) / 1e6, 3)
            its = sum(sum(1 for r in v if re.search(r"candidate_step(_vec4)?_kernel", r["name"]) and common[0] <= r["end"] <= common[1]) for v in attack_queues.values())
            result["trial_iterations_in_window"] = its
            result["trial_iterations_per_s_in_window"] = round(its / ((common[1] - common[0]) / 1e9), 1)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out + ".json", "w") as f:
        json.dump(result, f, indent=1)
    with open(args.out + ".txt", "w") as f:
        f.write(f"gap census {args.label} -- {result['dispatches']} dispatches in {result['trace']}\n")
        for q, info in result["queues"].items():
            f.write(f"  queue {q}: {info['dispatches']} dispatches, streams {info['streams']}, candidate steps {info['candidate_steps']}\n")
        if "queues_busy_at_once" in result:
            f.write(f"  queues busy at once (share of a {result['common_window_ms']} ms window): {result['queues_busy_at_once']}; "
                    f"{result['trial_iterations_per_s_in_window']} trial-iterations/s in it\n")
        for q, res in result["per_queue"].items():
            f.write(f"\nqueue {q}: last {res['iterations']} iterations -- per iteration {res['dispatches_per_iter']} dispatches, wall {res['wall_us_per_iter']} us = "
                    f"kernels {res['kernel_us_per_iter']} us (avg {res['avg_kernel_us']}) + gaps {res['gap_us_per_iter']} us (avg {res['avg_gap_us']}); "
                    f"{res['overlapped_per_iter']} dispatches started before their predecessor ended\n")
            f.write(f"  {'family':48s} {'calls':>7s} {'kern us':>9s} {'avg':>6s} | {'gap before':>10s} {'avg':>6s} | {'gap after':>10s} {'avg':>6s}\n")
            for r in res["families"]:
                f.write(f"  {r['family']:48s} {r['calls_per_iter']:7.1f} {r['kernel_us_per_iter']:9.1f} {r['avg_kernel_us']:6.2f} | "
                        f"{r['gap_before_us_per_iter']:10.1f} {r['avg_gap_before_us']:6.2f} | {r['gap_after_us_per_iter']:10.1f} {r['avg_gap_after_us']:6.2f}\n")
            f.write("  gap histogram (us: dispatches per iteration, us per iteration): " +
                    "; ".join(f"{h['bin_us']}: {h['count_per_iter']}, {h['us_per_iter']}" for h in res["gap_histogram"]) + "\n")
    if one_iteration:
        with open(args.out + "_one_iteration.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["start_us", "duration_us", "gap_before_us", "grid", "workgroup", "scratch", "family", "kernel"])
            w.writerows(one_iteration)
    print(open(args.out + ".txt").read())


if __name__ == "__main__":
    sys.exit(main())
