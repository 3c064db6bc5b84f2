"""Drop-in installation under an UNMODIFIED PKU-MARL/HARL checkout.

The reference reaches the on-policy path through two registries and a handful of module-level names
(SURVEY.md §8b): ``harl.runners.RUNNER_REGISTRY`` (examples/train.py:87-91), ``harl.algorithms.actors.ALGO_REGISTRY`` and
the names ``OnPolicyBaseRunner.__init__`` resolves in its own module (``ALGO_REGISTRY``, ``VCritic``, the three buffer
classes, ``ValueNorm``; harl/runners/on_policy_base_runner.py:7-12,99-161).  ``install()`` points those at the harl_amd
classes and registers runner classes that INHERIT the reference's ``OnPolicyBaseRunner`` -- so environment creation,
loggers, run directories, config dumps, ``run()``, ``eval()``, ``render()``, ``save()``, ``restore()`` and ``close()`` are
the reference's own code, byte for byte -- and override only the methods that touch the buffers:

    warmup / collect / insert / compute / train / after_update      (harl_amd/runner.py, device resident)

``collect()`` hands ``run()`` the actions as a NumPy array (that is what ``envs.step`` wants, on_policy_base_runner.py:207-216);
values, log-probs and hidden states stay on the device and go back into ``insert()`` untouched.

Launcher (what replaces ``python examples/train.py ...``):

    python -m harl_amd.dropin --reference /path/to/HARL -- --algo happo --env pettingzoo_mpe --exp_name x [key value ...]

Scope: one process, one GPU (the reference's constructor creates ``n_rollout_threads`` environments in this process;
the data-parallel path shards rollout threads across ranks through harl_amd.runner.OnPolicyHARunner(envs=...), see
bench.py).  ``algo_args['device']['cuda']`` must be true: there is no CPU path.
"""
from __future__ import annotations

import os
import runpy
import sys
import types
from typing import Optional

_INSTALLED = False


def stub_optional_modules() -> None:
    """The reference imports ``absl.flags`` (harl/envs/__init__.py), ``setproctitle`` (on_policy_base_runner.py:6) and
    ``tensorboardX`` (utils/configs_tools.py:86).  A real HARL environment has them; where one is missing (this build
    image) a no-op stand-in keeps the launcher usable.  Existing installations are never shadowed."""
    def missing(name):
        try:
            __import__(name)
            return False
        except ImportError:
            return True

    if missing("absl"):
        absl, flags = types.ModuleType("absl"), types.ModuleType("absl.flags")
        flags.FLAGS = lambda argv: argv
        absl.flags = flags
        sys.modules["absl"], sys.modules["absl.flags"] = absl, flags
    if missing("setproctitle"):
        sp = types.ModuleType("setproctitle")
        sp.setproctitle = lambda s: None
        sys.modules["setproctitle"] = sp
    if missing("tensorboardX"):
        tb = types.ModuleType("tensorboardX")

        class SummaryWriter:  # the three methods the loggers call (common/base_logger.py:164-174, runners' close())
            def __init__(self, *a, **k):
                self.scalars = {}

            def add_scalars(self, main_tag, tag_scalar_dict, global_step=None):
                for k, v in tag_scalar_dict.items():
                    self.scalars.setdefault(f"{main_tag}/{k}", []).append((global_step, float(v)))

            def export_scalars_to_json(self, path):
                import json
                with open(path, "w", encoding="utf-8") as f:
                    json.dump(self.scalars, f)

            def close(self):
                pass

        tb.SummaryWriter = SummaryWriter
        sys.modules["tensorboardX"] = tb


def _runner_class(base_cls, ours):
    """A subclass of the reference's OnPolicyBaseRunner with the buffer-touching methods taken from ``ours``
    (harl_amd.runner.OnPolicyHARunner / OnPolicyMARunner)."""
    import torch

    class Runner(base_cls):
        __doc__ = f"harl_amd drop-in: {base_cls.__module__}.{base_cls.__name__} + device-resident {ours.__name__} methods"

        # update-side state the reference constructor knows nothing about (communicator, column shard, scratch)
        _init_update_state = ours._init_update_state
        _ensure_update_state = ours._ensure_update_state
        _critic_first_ok = getattr(ours, "_critic_first_ok", None)
        _fp_normalised_advantages = ours._fp_normalised_advantages
        _check_comms = ours._check_comms
        warmup = ours.warmup
        insert = ours.insert
        compute = ours.compute
        train = ours.train
        after_update = ours.after_update

        def __init__(self, *a, **k):
            # the reference's constructor first; then -- still in constructor order on every rank -- everything COLLECTIVE the
            # update needs: the communicator (HARL_ALLREDUCE=oneshot / auto exchange handles while it is built) and, with
            # HARL_CRITIC_GROUP=1, the critic's own group (dist.second_group: never from a lazy path)
            super().__init__(*a, **k)
            if not self.algo_args["render"]["use_render"]:  # (render mode builds no critic and no buffers: base_runner.py:126)
                from .dist import Comm
                self.comm = Comm()
                self._critic_comm = self.comm.second_group()
                self._init_update_state()

        @torch.no_grad()
        def collect(self, step):
            """Reference contract (on_policy_base_runner.py:285-343): ``actions`` must be what ``envs.step`` accepts, a NumPy
            array [n_threads, n_agents, act_w]; the other four items only travel back into ``insert()`` and stay device
            tensors."""
            values, actions, logp, rnn, rnn_c = ours.collect(self, step)
            return values, actions.cpu().numpy(), logp, rnn, rnn_c

    Runner.__name__ = ours.__name__
    Runner.__qualname__ = ours.__name__
    return Runner


def install(reference_root: Optional[str] = None, stubs: bool = True) -> dict:
    """Patch the reference's registries in place.  Returns the runner registry entries that were installed."""
    global _INSTALLED
    if reference_root and reference_root not in sys.path:
        sys.path.insert(0, reference_root)
    if stubs:
        stub_optional_modules()
    import harl.algorithms.actors as ref_actors
    import harl.runners as ref_runners
    import harl.runners.on_policy_base_runner as base

    from . import buffers, runner, valuenorm
    from .happo import HAA2C, HAPPO
    from .hatrpo import HATRPO
    from .mappo import MAPPO
    from .v_critic import VCritic

    ours_algos = {"happo": HAPPO, "hatrpo": HATRPO, "haa2c": HAA2C, "mappo": MAPPO}
    ref_actors.ALGO_REGISTRY.update(ours_algos)
    base.ALGO_REGISTRY = ref_actors.ALGO_REGISTRY
    base.VCritic = VCritic
    base.ValueNorm = valuenorm.ValueNorm
    base.OnPolicyActorBuffer = buffers.OnPolicyActorBuffer
    base.OnPolicyCriticBufferEP = buffers.OnPolicyCriticBufferEP
    base.OnPolicyCriticBufferFP = buffers.OnPolicyCriticBufferFP
    ha = _runner_class(base.OnPolicyBaseRunner, runner.OnPolicyHARunner)
    ma = _runner_class(base.OnPolicyBaseRunner, runner.OnPolicyMARunner)
    installed = {"happo": ha, "hatrpo": ha, "haa2c": ha, "mappo": ma}
    ref_runners.RUNNER_REGISTRY.update(installed)
    _INSTALLED = True
    return installed


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    ref = os.environ.get("HARL_REFERENCE")
    if "--reference" in argv:
        i = argv.index("--reference")
        ref = argv[i + 1]
        del argv[i:i + 2]
    if "--" in argv:
        argv.remove("--")
    if not ref or not os.path.exists(os.path.join(ref, "examples", "train.py")):
        raise SystemExit("harl_amd.dropin: --reference /path/to/HARL (or HARL_REFERENCE) must contain examples/train.py")
    install(ref)
    sys.argv = [os.path.join(ref, "examples", "train.py")] + argv
    runpy.run_path(sys.argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
