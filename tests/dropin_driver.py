#!/usr/bin/env python3
"""Run the UNMODIFIED reference launcher (``<reference>/examples/train.py``) on a fake MPE environment, either as it is
(``--mode reference``: BASELINE.json configs[0], the reference's own CPU path) or with the harl_amd classes installed under
it (``--mode dropin``).  Without a GPU (``--stub-kernels``) the C-ABI calls are replaced by a recorder -- no arithmetic
happens, the run only proves the plumbing: constructor compatibility, data hand-offs between the reference's run()/eval()/
logger/save() and the device-resident buffers, return types and dictionary keys.  Prints one JSON line.

    python tests/dropin_driver.py --reference /root/reference --mode dropin --stub-kernels --log-dir /tmp/x [train.py args]
"""
import argparse
import ctypes
import json
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("HARL_REFERENCE", "/root/reference"))
    ap.add_argument("--mode", choices=["reference", "dropin"], required=True)
    ap.add_argument("--stub-kernels", action="store_true")
    ap.add_argument("--log-dir", required=True)
    a, rest = ap.parse_known_args()
    sys.path.insert(0, a.reference)
    from harl_amd import dropin
    dropin.stub_optional_modules()
    import tests.fake_mpe as fake
    sys.modules["harl.envs.pettingzoo_mpe.pettingzoo_mpe_env"] = fake  # the real module needs pettingzoo + supersuit

    calls = {}
    if a.stub_kernels:
        os.environ["HARL_DEVICE"] = "cpu"  # the reference builds the buffers without a device argument
        import torch
        from harl_amd import _lib
        real_call = _lib.call
        host_side = {"harl_randperm_replay", "harl_rng_advance"}

        def recorder(name, *args, tag=None):
            calls[name] = calls.get(name, 0) + 1
            if name in host_side:
                return real_call(name, *args, tag=tag)
            if name == "harl_masked_moments":  # (x, active, n, out3, scratch, stream): pretend every entry is active so train() proceeds
                ctypes.c_double.from_address(args[3] + 16).value = float(args[2])
            return None

        _lib.call = recorder
        _lib.require_gpu = lambda device: None
        _lib.stream = lambda: 0
        _lib.scratch = lambda kind: 0
        for mod in ("nets", "buffers", "happo", "hatrpo", "mappo", "v_critic", "valuenorm", "runner"):
            m = __import__(f"harl_amd.{mod}", fromlist=["x"])
            for nm in ("call", "stream"):
                if hasattr(m, nm):
                    setattr(m, nm, getattr(_lib, nm))

        class _Ev:  # torch.cuda.Event stand-in
            def __init__(self, *a_, **k_):
                pass

            def record(self, *a_):
                pass

            def synchronize(self):
                pass

        torch.cuda.Event = _Ev
    installed = {}
    if a.mode == "dropin":
        installed = dropin.install(a.reference)
    import harl.runners as rr
    captured = {}
    for key in ("happo", "hatrpo", "haa2c", "mappo"):
        cls = rr.RUNNER_REGISTRY[key]
        if getattr(cls, "_wrapped_for_test", False):
            continue

        class Wrapped(cls):  # remember the runner object so the test can look at what it left behind
            _wrapped_for_test = True

            def __init__(self, *aa, **kk):
                super().__init__(*aa, **kk)
                captured["runner"] = self

        Wrapped.__name__ = cls.__name__
        rr.RUNNER_REGISTRY[key] = Wrapped
    sys.argv = [os.path.join(a.reference, "examples", "train.py")] + rest + ["--log_dir", a.log_dir]
    runpy.run_path(sys.argv[0], run_name="__main__")
    r = captured["runner"]
    out = dict(mode=a.mode, runner_class=f"{type(r).__mro__[1].__module__}.{type(r).__mro__[1].__name__}",
               base_classes=[f"{c.__module__}.{c.__name__}" for c in type(r).__mro__[1:4]],
               actor_class=f"{type(r.actor[0]).__module__}.{type(r.actor[0]).__name__}",
               critic_class=f"{type(r.critic).__module__}.{type(r.critic).__name__}",
               buffer_class=f"{type(r.actor_buffer[0]).__module__}.{type(r.actor_buffer[0]).__name__}",
               save_dir=str(r.save_dir), saved=sorted(os.listdir(str(r.save_dir))),
               run_dir_files=sorted(os.listdir(str(r.run_dir))), kernel_calls=calls,
               obs_type=type(r.actor_buffer[0].obs).__name__)
    print("DROPIN_RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
