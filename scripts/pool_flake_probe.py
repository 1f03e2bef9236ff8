"""Which trials of a 4-rank worker-pool run (32 restarts, ConvNet, smooth soft-sign configuration: the scenario of
tests/test_gpu_baseline_configs.py::test_configs3_shape_32_restarts_over_four_ranks_two_groups_of_four_each) deviate from the
single-rank run, from which iteration on and by how much?  Round 6: that test failed once on a fresh box (trial 1 = the first trial
of worker rank 1, from iteration 1, 16 %), and the round-5 tree showed the same signature once in 8 runs (trial 3, 7e-5); 64 runs at
HEAD were clean (profiles/r6_pool_flake_probe.log).  Extra arguments are config overrides.

    for i in $(seq 16); do python scripts/pool_flake_probe.py | grep deviating; done
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import breaching_amd
from breaching_amd.cases import build_case

def main():
    os.environ["BREACH_HIP_GRAPH_STRICT"] = "1"
    over = ["objective.type=euclidean", "objective.scale=0.01", "optim.signed=soft", "optim.max_iterations=8",
            "restarts.num_trials=32", "restarts.scoring=euclidean", "optim.callback=4", "impl.trial_pool=required"] + sys.argv[1:]
    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    setup = dict(device=torch.device("cuda:0"), dtype=torch.float)
    results = {}
    for devices in ("[0]", "[0, 0, 0, 0]"):
        cfg = breaching_amd.get_attack_config("invertinggradients", over + [f"impl.trial_devices={devices}"])
        attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, setup)
        try:
            torch.manual_seed(5)
            shared = [dict(gradients=list(d["gradients"]), buffers=d["buffers"], metadata=dict(d["metadata"])) for d in case.shared_data]
            rec, stats = attacker.reconstruct(case.server_payload, shared, {})
            results[devices] = stats
            print(devices, "streams:", stats["execution"]["trial_streams"])
        finally:
            attacker.close()
    base = results["[0]"]
    for devices in ("[0, 0, 0, 0]",):
        world = devices.count("0")
        bad = []
        for t in range(32):
            a, b = np.asarray(results[devices][f"Trial_{t}_Val"]), np.asarray(base[f"Trial_{t}_Val"])
            dev = np.abs(a / b - 1)
            if dev.max() > 1e-6:
                bad.append((t, t % world, int(np.argmax(dev > 1e-6)), float(dev.max())))
        print(devices, "deviating trials (trial, rank, first iteration, max rel):", bad)


if __name__ == "__main__":
    main()
