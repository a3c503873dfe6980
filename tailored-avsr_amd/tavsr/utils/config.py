"""YAML recipe handling with the reference's override grammar (src/utils/config.py:1-20):
``conf:key:value`` or ``key:value``, the value typed after the existing entry."""
import argparse

import yaml


def override_yaml(yaml_config, to_override):
    for item in to_override or []:
        parts = item.split(":")
        if len(parts) == 2:
            holder, key, value = yaml_config, parts[0], parts[1]
        elif len(parts) == 3:
            holder, key, value = yaml_config[parts[0]], parts[1], parts[2]
        else:
            continue
        caster = type(holder[key])
        holder[key] = (value == "true") if caster is bool else caster(value)
    return yaml_config


def load_config(path, overrides=None) -> argparse.Namespace:
    with open(path, "r") as f:
        conf = yaml.safe_load(f)
    return argparse.Namespace(**override_yaml(conf, overrides))
