"""Is the 4-restarts-in-flight rate (~474 it/s, GPU 55 % busy) bounded by the HOST launching the hipGraphs?  Measures, for W
ResNet-18 trials on W side streams: (a) round-robin replays from one thread (what _run_trial_group does), (b) the host time
of one replay call with nothing to wait for, (c) one Python thread per trial, each replaying its own graph on its own stream
(graph.replay() releases the GIL).  Prints one JSON line.

    python scripts/replay_threads_probe.py [--width 4] [--steps 150]
"""
import argparse, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import breaching_amd
from breaching_amd.attacker import FusedTrial
from breaching_amd.cases import build_case, initial_candidate

p = argparse.ArgumentParser()
p.add_argument("--width", type=int, default=4)
p.add_argument("--steps", type=int, default=150)
args = p.parse_args()
dev = torch.device("cuda:0")
case = build_case("resnet18", "ImageNet", 1, device=dev, gradient_device=dev)
cfg = breaching_amd.get_attack_config("invertinggradients")
attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, dict(device=dev, dtype=torch.float))
rec_models, labels, stats = attacker.prepare_attack(case.server_payload, case.shared_data)
attacker.objective.initialize(attacker.loss_fn, cfg.impl, None)
for reg in attacker.regularizers:
    reg.initialize(rec_models, case.shared_data, labels)
attacker.objective.prepare(rec_models, case.shared_data)
main = torch.cuda.current_stream(dev)
trials = []
for j in range(args.width):
    stream = torch.cuda.Stream(dev)
    stream.wait_stream(main)
    x = initial_candidate(case.data_cfg, 1, trial=j).to(dev).requires_grad_(True)
    with torch.cuda.stream(stream):
        trials.append((stream, FusedTrial(attacker, [x], labels, rec_models, case.shared_data)))
for _ in range(8):  # eager warm-up + capture
    for stream, run in trials:
        with torch.cuda.stream(stream):
            run.step()
torch.cuda.synchronize()
assert all(run.graph is not None for _, run in trials)
out = dict(width=args.width, steps=args.steps)

# (b) host time of one replay call: an idle GPU, a handful of calls, no synchronisation in between
stream, run = trials[0]
t0 = time.perf_counter()
with torch.cuda.stream(stream):
    for _ in range(10):
        run.step()
host = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
out["host_ms_per_replay_call"] = round(host * 1e3, 3)

# (a) one thread, round robin
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    for stream, run in trials:
        with torch.cuda.stream(stream):
            run.step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out["round_robin_it_per_s"] = round(args.width * args.steps / dt, 1)


# (c) one thread per trial
def worker(stream, run):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(stream):
        for _ in range(args.steps):
            run.step()


threads = [threading.Thread(target=worker, args=t) for t in trials]
torch.cuda.synchronize()
t0 = time.perf_counter()
for th in threads:
    th.start()
for th in threads:
    th.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out["thread_per_trial_it_per_s"] = round(args.width * args.steps / dt, 1)
out["final_totals"] = [run.read_state()["total"] for _, run in trials]
print(json.dumps(out))
