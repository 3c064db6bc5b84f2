#!/usr/bin/env python3
"""Survey of the reference's tuned on-policy configurations -> tests/golden/tuned_configs_survey.json.

For every ``tuned_configs/**/config.json`` whose algorithm is on the on-policy path (happo, hatrpo, haa2c, mappo) the fields
that decide whether this implementation supports it are recorded (no reference source or config file is copied: one
condensed record per configuration).  Run in the container that has /root/reference:  python oracle/gen_config_survey.py
"""
import glob
import json
import os

REF = os.environ.get("HARL_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tuned_configs_survey.json")
KEEP_MODEL = ("hidden_sizes", "activation_func", "use_feature_normalization", "use_recurrent_policy",
              "use_naive_recurrent_policy", "recurrent_n", "data_chunk_length")
KEEP_ALGO = ("ppo_epoch", "a2c_epoch", "critic_epoch", "actor_num_mini_batch", "critic_num_mini_batch", "share_param",
             "fixed_order", "action_aggregation", "use_huber_loss", "use_policy_active_masks", "use_clipped_value_loss")
KEEP_TRAIN = ("n_rollout_threads", "episode_length", "use_valuenorm", "use_proper_time_limits")


def n_actions(env: str, env_args: dict):
    """Size of the action head where it follows from the configuration alone (None = small / environment defined)."""
    if env == "smac":  # 6 + n_enemies (StarCraft2_Env.py:275-277), from the reference's map registry
        import re
        src = open(os.path.join(REF, "harl", "envs", "smac", "smac_maps.py")).read()
        m = re.search(r'"%s":\s*\{\s*"n_agents":\s*\d+,\s*"n_enemies":\s*(\d+)' % re.escape(env_args["map_name"]), src)
        return 6 + int(m.group(1)) if m else None
    if env == "smacv2":  # <race>_<n>_vs_<m>: 6 + m
        name = env_args.get("map_name", "")
        return 6 + int(name.rsplit("_vs_", 1)[1]) if "_vs_" in name else None
    if env == "football":
        return 19  # gfootball default action set
    return None


def main():
    recs = []
    for p in sorted(glob.glob(os.path.join(REF, "tuned_configs", "**", "config.json"), recursive=True)):
        c = json.load(open(p))
        algo = c["main_args"]["algo"]
        if algo not in ("happo", "hatrpo", "haa2c", "mappo"):
            continue
        aa = c["algo_args"]
        recs.append(dict(path=os.path.relpath(p, os.path.join(REF, "tuned_configs")), algo=algo, env=c["main_args"]["env"],
                         state_type=c["env_args"].get("state_type", "EP"), n_actions=n_actions(c["main_args"]["env"], c["env_args"]),
                         model={k: aa["model"][k] for k in KEEP_MODEL if k in aa["model"]},
                         algo_args={k: aa["algo"][k] for k in KEEP_ALGO if k in aa["algo"]},
                         train={k: aa["train"][k] for k in KEEP_TRAIN if k in aa["train"]}))
    json.dump(recs, open(OUT, "w"), indent=1, sort_keys=True)
    print(f"{len(recs)} on-policy tuned configurations -> {OUT}")


if __name__ == "__main__":
    main()
