"""breaching_amd -- MI355X-native hot path for JonasGeiping/breaching's optimisation-based gradient inversion.

Public surface (mirrors the reference for this path only):
  * ``prepare_attack(model, loss, cfg_attack, setup)``          reference: breaching/attacks/__init__.py:12-34
  * ``get_attack_config(name, overrides)``                      reference: breaching/__init__.py:24-29
  * ``install()`` -- rebind the reference's two optimisation attack classes so ``simulate_breach.py`` runs unchanged
"""

import os as _os

# Restarts in flight run on separate HIP streams (attacker._run_trial_group).  ROCclr multiplexes streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4, one of which the caller's stream holds); with two trials sharing a queue
# the four-in-flight rate drops from ~475 to ~310 iterations/s (measured in round 2).  Eight queues give every side stream
# one of its own next to the caller's; the queues in turn sit on four hardware compute pipes, which is what
# breaching_amd/streams.py picks the side streams by (round 4).  Only a default, and only effective when set before the HIP
# runtime initialises, i.e. before the first CUDA call of the process.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .config import AttrDict, get_attack_config, get_data_config  # noqa: E402,F401

__version__ = "0.1.0"


def prepare_attack(model, loss, cfg_attack, setup=None):
    from . import attacker

    if setup is None:
        setup = attacker._DEFAULT_SETUP
    return attacker.prepare_attack(model, loss, cfg_attack, setup)


def install():
    """Make an importable reference package use the HIP attackers for `optimization` / `joint-optimization`.

    ``breaching.attacks.prepare_attack`` (attacks/__init__.py:12-34) looks the classes up as module globals, so
    rebinding the two names is all that is needed; every other attack type keeps the reference implementation.
    """
    import breaching.attacks as ref_attacks

    from .attacker import HipOptimizationAttacker, HipOptimizationJointAttacker

    ref_attacks.OptimizationBasedAttacker = HipOptimizationAttacker
    ref_attacks.OptimizationJointAttacker = HipOptimizationJointAttacker
    return ref_attacks
