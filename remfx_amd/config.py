"""The subset of Hydra / OmegaConf the reference's entry points rely on, over PyYAML
(hydra-core / omegaconf are not installed in this image; when they are, the reference's own
scripts work unchanged with the `remfx` alias package -- INTEGRATION.md).

Covers what cfg/ uses (SURVEY section 5 "Config / flags"):
  * primary config + `defaults:` list with `_self_`, `group: option`, `override /group: option`,
    `group: null`; group files are `# @package _global_` (merged at the root);
  * command line: `+exp=NAME` (adds cfg/exp/NAME.yaml with its own defaults), `group=option`,
    `a.b.c=value`, `+a.b=value`, list / null / bool / number literals;
  * interpolation: `${key}`, `${a.b}` (value AND whole-node references), `${oc.env:VAR}`,
    `${oc.env:VAR,default}`, `${now:%fmt}`;
  * `instantiate(node, **kwargs)`: recursive `_target_` construction with extra kwargs
    (`_convert_="partial"` semantics: containers become plain dict / list).
`_target_` strings of the reference resolve to this repo's classes through TARGET_ALIASES.
"""
import copy
import importlib
import os
import re
import time

import yaml

TARGET_ALIASES = {
    "remfx.": "remfx_amd.",
    "pytorch_lightning.Trainer": "remfx_amd.trainer.Trainer",
    "pytorch_lightning.loggers.CSVLogger": "remfx_amd.trainer.CSVLogger",
    "pytorch_lightning.loggers.csv_logs.CSVLogger": "remfx_amd.trainer.CSVLogger",
}
SKIP_TARGET_PREFIXES = ("pytorch_lightning.callbacks.", "remfx.callbacks.")


class _Loader(yaml.SafeLoader):
    """PyYAML (YAML 1.1) reads `1e-4` as a string; OmegaConf reads it as a float.  Follow OmegaConf."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(?:[0-9][0-9_]*\.[0-9_]*(?:[eE][-+]?[0-9]+)?|\.[0-9_]+(?:[eE][-+]?[0-9]+)?"
               r"|[0-9][0-9_]*[eE][-+]?[0-9]+|\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$"),
    list("-+0123456789."))


def _yaml(text):
    return yaml.load(text, Loader=_Loader)


def _load(path):
    with open(path) as f:
        return _yaml(f.read()) or {}


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _parse_value(s):
    try:
        return _yaml(s)
    except yaml.YAMLError:
        return s


def _set_path(cfg, dotted, value, create):
    keys = dotted.split(".")
    node = cfg
    for k in keys[:-1]:
        if k not in node or not isinstance(node[k], dict):
            if not create and k not in node:
                raise KeyError(f"override '{dotted}': key '{k}' not in config (use +{dotted}=...)")
            node[k] = {} if not isinstance(node.get(k), dict) else node[k]
        node = node[k]
    if not create and keys[-1] not in node:
        raise KeyError(f"override '{dotted}': key not in config (use +{dotted}=...)")
    node[keys[-1]] = value


def _apply_defaults(cfg_dir, body, cfg, groups):
    """Process one file's `defaults:` list (order matters; _self_ marks where the body lands)."""
    defaults = body.pop("defaults", None)
    if defaults is None:
        _merge(cfg, body)
        return
    merged_self = False
    for d in defaults:
        if d == "_self_":
            _merge(cfg, body)
            merged_self = True
            continue
        (k, v), = d.items()
        k = k.strip()
        if k.startswith("override "):
            k = k[len("override "):].strip()
        k = k.lstrip("/")
        groups[k] = v
    if not merged_self:
        _merge(cfg, body)


def compose(cfg_dir, config_name="config.yaml", overrides=()):
    cfg_dir = os.path.abspath(cfg_dir)
    cfg, groups = {}, {}
    _apply_defaults(cfg_dir, _load(os.path.join(cfg_dir, config_name)), cfg, groups)
    plain, extra_groups = [], []
    group_dirs = {d for d in os.listdir(cfg_dir) if os.path.isdir(os.path.join(cfg_dir, d))}
    for ov in overrides:
        key, _, val = ov.partition("=")
        add = key.startswith("+")
        key = key.lstrip("+")
        if key in group_dirs and "." not in key:
            if add:
                extra_groups.append((key, val))
            else:
                groups[key] = val
        else:
            plain.append((key, _parse_value(val), add))
    # `+exp=NAME`: its own defaults may override /model, /effects ...; command-line groups win
    cli_groups = dict(groups)
    for g, opt in extra_groups:
        body = _load(os.path.join(cfg_dir, g, f"{opt}.yaml"))
        sub = {}
        _apply_defaults(cfg_dir, body, sub, groups)
        cfg.setdefault("__late__", []).append(sub)
    for k, v in cli_groups.items():
        if any(ov.partition("=")[0] == k for ov in overrides):
            groups[k] = v
    for g, opt in groups.items():
        if opt in (None, "null"):
            continue
        _merge(cfg, _load(os.path.join(cfg_dir, g, f"{opt}.yaml")))
    for sub in cfg.pop("__late__", []):
        _merge(cfg, sub)
    for key, val, add in plain:
        _set_path(cfg, key, val, create=add)
    return resolve(cfg)


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _lookup(root, dotted):
    node = root
    for k in dotted.split("."):
        node = node[k]
    return node


def _resolve_str(root, s, depth=0):
    if depth > 20:
        raise ValueError(f"interpolation loop in {s!r}")
    m = _INTERP.fullmatch(s)
    if m:                                        # whole-value reference: keeps type, may be a node
        return _resolve_any(root, _eval(root, m.group(1)), depth + 1)
    return _INTERP.sub(lambda mm: str(_resolve_any(root, _eval(root, mm.group(1)), depth + 1)), s)


def _eval(root, expr):
    if expr.startswith("oc.env:"):
        name, _, default = expr[len("oc.env:"):].partition(",")
        if name in os.environ:
            return os.environ[name]
        if default != "":
            return _parse_value(default)
        return f"<unset env {name}>"             # only fails if something really consumes it
    if expr.startswith("now:"):
        return time.strftime(expr[len("now:"):])
    return copy.deepcopy(_lookup(root, expr))


def _resolve_any(root, v, depth=0):
    if isinstance(v, str):
        return _resolve_str(root, v, depth) if "${" in v else v
    if isinstance(v, dict):
        return {k: _resolve_any(root, x, depth) for k, x in v.items()}
    if isinstance(v, list):
        return [_resolve_any(root, x, depth) for x in v]
    return v


def resolve(cfg):
    return _resolve_any(cfg, cfg)


def _locate(target):
    if target in TARGET_ALIASES:
        target = TARGET_ALIASES[target]
    else:
        mod0 = target.split(".")[0]
        try:
            importlib.import_module(mod0)
        except Exception:
            for k, v in TARGET_ALIASES.items():
                if k.endswith(".") and target.startswith(k):
                    target = v + target[len(k):]
                    break
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(node, **kwargs):
    """hydra.utils.instantiate(node, _convert_="partial", **kwargs) for the cfg tree's needs."""
    kwargs.pop("_convert_", None)
    if isinstance(node, list):
        return [instantiate(x) for x in node]
    if not isinstance(node, dict):
        return node
    if "_target_" not in node:
        return {k: instantiate(v) for k, v in node.items()}
    target = node["_target_"]
    if target.startswith(SKIP_TARGET_PREFIXES):
        return None                               # observability callbacks: outside the hot path
    args = {k: instantiate(v) for k, v in node.items() if k != "_target_"}
    args.update(kwargs)
    return _locate(target)(**args)
