"""Condense rocprofv3 CSV output into small summaries that fit gpurun's copy-back limit and profiles/.

    python scripts/summarize_prof.py <rocprof_dir> <out_prefix> [pmc_counter_name]
"""
import csv, os, sys, collections

# every kernel of libbreach_hip.so: A (gm_*), B / commit (candidate_step, loss_commit, grad_sumsq, state_reset), C (tv_norm),
# D (bn_sums / bn_finalize / bn_bwd / bn_bwd_acc), E (bn_eval_*), F (ln_*), multi-tensor and metric kernels
OURS = ("gm_fwd_kernel", "gm_bwd_kernel", "gm_finalize_kernel", "tv_norm_kernel", "candidate_step_kernel", "loss_commit_kernel",
        "bn_sums_kernel", "bn_finalize_kernel", "bn_bwd_kernel", "bn_bwd_acc_kernel", "mt_kernel", "orthogonality_kernel", "psnr_mse_kernel",
        "grad_sumsq", "gm_pack_kernel", "state_reset", "bn_eval_fwd_kernel", "bn_eval_bwd_kernel", "bn_eval_bwd_bwd_kernel",
        "bn_eval_combine_kernel", "ln_fwd_kernel", "ln_bwd_", "grad_norm_finalize_kernel", "psnr_finalize_kernel", "tv_norm_vec4_kernel",
        "candidate_step_vec4_kernel", "candidate_step_list_kernel")
src, out = sys.argv[1], sys.argv[2]
counter = sys.argv[3] if len(sys.argv) > 3 else None
files = {f: os.path.join(src, f) for f in os.listdir(src)}
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")[:90]


trace = next((p for f, p in files.items() if f.endswith("kernel_trace.csv")), None)
if trace:
    per = collections.defaultdict(list)
    by_grid = collections.defaultdict(list)  # our kernels again, split by launch size: one probe run often covers several list sizes
    first, last = None, None
    with open(trace) as f:
        for row in csv.DictReader(f):
            s, e = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
            per[row["Kernel_Name"]].append(e - s)
            if "Grid_Size_X" in row and any(k in row["Kernel_Name"] for k in OURS):
                by_grid[(row["Kernel_Name"], int(row["Grid_Size_X"]) // max(int(row.get("Workgroup_Size_X") or 1), 1))].append(e - s)
            first = s if first is None else min(first, s)
            last = e if last is None else max(last, e)
    total = sum(sum(v) for v in per.values())
    rows = sorted(per.items(), key=lambda kv: -sum(kv[1]))
    with open(out + "_kernel_summary.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct_of_kernel_time", "ours"])
        for name, d in rows:
            w.writerow([short(name), len(d), round(sum(d) / 1e3, 2), round(sum(d) / len(d) / 1e3, 3), round(min(d) / 1e3, 3),
                        round(max(d) / 1e3, 3), round(100 * sum(d) / total, 2), int(any(k in name for k in OURS))])
    with open(out + "_kernel_summary.txt", "w") as f:
        f.write(f"kernels: {sum(len(v) for v in per.values())} dispatches, {len(per)} distinct, busy {total/1e6:.2f} ms over a span of {(last-first)/1e6:.2f} ms\n")
        for name, d in rows:
            if any(k in name for k in OURS):
                f.write(f"{short(name):70s} calls={len(d):5d} avg={sum(d)/len(d)/1e3:8.2f}us min={min(d)/1e3:8.2f}us max={max(d)/1e3:8.2f}us\n")
    if by_grid:
        with open(out + "_kernel_by_grid.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "workgroups", "calls", "avg_us", "median_us", "min_us", "max_us"])
            for (name, wgs), d in sorted(by_grid.items(), key=lambda kv: (kv[0][0], kv[0][1])):
                d = sorted(d)
                w.writerow([short(name), wgs, len(d), round(sum(d) / len(d) / 1e3, 3), round(d[len(d) // 2] / 1e3, 3), round(d[0] / 1e3, 3), round(d[-1] / 1e3, 3)])
    print(open(out + "_kernel_summary.txt").read())

cc = next((p for f, p in files.items() if f.endswith("counter_collection.csv")), None)
if cc:
    per = collections.defaultdict(list)
    with open(cc) as f:
        for row in csv.DictReader(f):
            if counter and row.get("Counter_Name") != counter:
                continue
            per[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
    with open(out + "_pmc_summary.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "avg_value", "min_value", "max_value"])
        for (name, cn), d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            if any(k in name for k in OURS):
                w.writerow([short(name), cn, len(d), round(sum(d) / len(d), 3), min(d), max(d)])
    print(open(out + "_pmc_summary.csv").read())
