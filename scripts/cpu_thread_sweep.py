"""How many host threads should bench.py's cpu_baseline use?  BASELINE.md section 3 says os.cpu_count(); on the 256-logical-core host
of an MI355X box torch's CPU convolutions at B = 1 stop scaling long before that.  Times oracle/restate.py (the port of the
reference loop; the reference itself does not exist on the GPU box) on the bench workload at several thread counts.

    python scripts/cpu_thread_sweep.py [--iters 16] [--threads 8,16,32,64,128,256]  ->  JSON (commit as profiles/r4_cpu_thread_sweep.json)
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=16)
ap.add_argument("--threads", default="8,16,32,64,128,256")
args = ap.parse_args()
import breaching_amd
from breaching_amd.cases import build_case, initial_candidate
from oracle import restate

case = build_case("resnet18", "ImageNet", 1)
x0 = initial_candidate(case.data_cfg, 1)
cfg = breaching_amd.get_attack_config("invertinggradients")
out = dict(workload="ResNet-18 (1000 classes) 1x3x224x224, attack=invertinggradients, oracle/restate.py", host_cpu_count=os.cpu_count(),
           torch=torch.__version__, iterations=args.iters, iterations_per_s={})
for threads in [int(t) for t in args.threads.split(",")]:
    if threads > (os.cpu_count() or 1):
        continue
    torch.set_num_threads(threads)
    restate.run_attack(case.model, case.loss_fn, cfg, case.server_payload, case.shared_data, initial_data=x0, max_iterations=2)
    timing = []
    restate.run_attack(case.model, case.loss_fn, cfg, case.server_payload, case.shared_data, initial_data=x0, max_iterations=args.iters, timing=timing)
    out["iterations_per_s"][str(threads)] = round(args.iters / timing[0], 3)
    print(f"  {threads} threads: {out['iterations_per_s'][str(threads)]} it/s", file=sys.stderr, flush=True)
    if out["iterations_per_s"][str(threads)] < 0.1 * max(out["iterations_per_s"].values()):
        out["stopped_after"] = threads  # more threads only get slower (oversubscribed tiny convolutions): not worth minutes of box time
        break
best = max(out["iterations_per_s"], key=lambda k: out["iterations_per_s"][k])
out["best_threads"] = int(best)
print(json.dumps(out, indent=1))
