"""harl_amd -- MI355X-native on-policy sequential-update training step for HARL (HAPPO / HATRPO / HAA2C / MAPPO + V-critic).

Same Runner / Algorithm / Buffer class surface as PKU-MARL/HARL's on-policy path; arithmetic in hand-written
gfx950 HIP kernels behind the C ABI of include/harl_hip.h.  There is no CPU fallback: constructing any compute
class without an MI355X raises.
"""
__version__ = "0.1.0"

__all__ = ["OnPolicyHARunner", "OnPolicyMARunner", "HAPPO", "HATRPO", "HAA2C", "MAPPO", "VCritic", "OnPolicyActorBuffer",
           "OnPolicyCriticBufferEP", "OnPolicyCriticBufferFP", "ValueNorm"]


def __getattr__(name):  # lazy: importing harl_amd.synthetic (pure NumPy) must not need torch/HIP
    if name == "OnPolicyHARunner":
        from .runner import OnPolicyHARunner
        return OnPolicyHARunner
    if name == "OnPolicyMARunner":
        from .runner import OnPolicyMARunner
        return OnPolicyMARunner
    if name == "HAPPO":
        from .happo import HAPPO
        return HAPPO
    if name == "HAA2C":
        from .happo import HAA2C
        return HAA2C
    if name == "MAPPO":
        from .mappo import MAPPO
        return MAPPO
    if name == "HATRPO":
        from .hatrpo import HATRPO
        return HATRPO
    if name == "VCritic":
        from .v_critic import VCritic
        return VCritic
    if name in ("OnPolicyActorBuffer", "OnPolicyCriticBufferEP", "OnPolicyCriticBufferFP"):
        from . import buffers
        return getattr(buffers, name)
    if name == "ValueNorm":
        from .valuenorm import ValueNorm
        return ValueNorm
    raise AttributeError(name)
