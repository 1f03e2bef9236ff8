"""Attack a gradient of your own model with the MI355X hot path -- no federated-learning simulation needed.

Counterpart of the reference's ``minimal_example.py``: build the two dictionaries by hand, call ``reconstruct``.
Runs on one ROCm GPU:  python examples/minimal_example.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import breaching_amd
from breaching_amd.cases import ResNet, psnr
from breaching_amd.priors import psnr_on_device


def main():
    device = torch.device("cuda:0")
    setup = dict(device=device, dtype=torch.float)

    # your model and loss (random init here; ImageNet statistics for the box constraint)
    model = ResNet(18, num_classes=1000).to(device).eval()
    loss_fn = torch.nn.CrossEntropyLoss()
    data_cfg = breaching_amd.get_data_config("ImageNet")

    # the user's secret batch and the gradient it shares
    x_true = ((torch.rand(1, 3, 224, 224) - torch.tensor(data_cfg.mean)[None, :, None, None])
              / torch.tensor(data_cfg.std)[None, :, None, None]).to(device)
    labels = torch.tensor([412], device=device)
    gradients = torch.autograd.grad(loss_fn(model(x_true), labels), tuple(model.parameters()))

    # what server and user exchange (reference: servers.py:138-147, users.py:176-183)
    server_payload = [dict(parameters=list(model.parameters()), buffers=list(model.buffers()), metadata=data_cfg)]
    shared_data = [dict(gradients=[g.detach() for g in gradients], buffers=None,
                        metadata=dict(num_data_points=1, labels=labels, local_hyperparams=None))]

    cfg = breaching_amd.get_attack_config("invertinggradients", ["optim.max_iterations=2000", "optim.callback=500"])
    attacker = breaching_amd.prepare_attack(model, loss_fn, cfg, setup)
    reconstruction, stats = attacker.reconstruct(server_payload, shared_data, {}, dryrun=False)

    print(f"final objective {stats['Trial_0_Val'][-1]:.4f}, selected score {stats['opt_value']:.4f}, "
          f"PSNR {psnr(reconstruction['data'], x_true, data_cfg):.2f} dB "
          f"(on the device: {psnr_on_device(reconstruction['data'], x_true, data_cfg.mean, data_cfg.std)[0].item():.2f} dB)")
    attacker.close()  # stops the per-GPU trial workers, if restarts were sharded over several GPUs


if __name__ == "__main__":
    import logging

    logging.basicConfig(level=logging.INFO, format="%(message)s")
    main()
