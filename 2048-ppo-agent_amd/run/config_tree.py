"""Compose a Hydra-style config TREE with PyYAML only (Hydra / OmegaConf are not dependencies of this engine).

The reference's CLI is ``@hydra.main(config_path="../configs", config_name="train_ppo_agent.yaml")``
(run/train_ppo_agent.py:19-21) over ``configs/train_ppo_agent.yaml:5-11``::

    defaults: [_self_, {data: default}, {model: transformer_combined}, {trainer: default}, {paths: default}, {hydra: default}]

``compose(path, overrides)`` reads exactly that: the primary file's ``defaults`` list in order (``_self_`` = the primary file's own
keys; ``{group: name}`` = ``<dir>/<group>/<name>.yaml`` placed under the key ``group``, or merged at the root when the file starts
with ``# @package _global_``), later entries overriding earlier ones key by key, then the command line:

    key.sub=value            set a value (YAML-parsed), creating the path
    group=name               re-select a config group that is a directory next to the primary file (``model=mlp``)
    +group=name              add a group that the defaults list does not name (``+experiment=resume_train_ppo_agent``)
    ~key.sub                 delete a key

and finally ``${...}`` interpolations: ``${a.b}`` (another node), ``${hydra:runtime.cwd}``, ``${now:%Y-%m-%d}``, ``${oc.env:VAR}``;
an unresolvable one is left as it stands (the engine's CLI reads none of the interpolated nodes except
``trainer.resume_from_checkpoint``).  A file WITHOUT a ``defaults`` list is returned as it is (the flattened config this repository
ships), so both layouts go through the same call.
"""
from __future__ import annotations

import copy
import datetime
import os
import re

import yaml

_GLOBAL = re.compile(r"^\s*#\s*@package\s+_global_\s*$", re.M)
# PyYAML's YAML 1.1 resolver reads `4e-4` / `1e-6` (no dot) as strings; OmegaConf reads them as floats
_FLOAT = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+$")


def _floats(node):
    if isinstance(node, dict):
        return {k: _floats(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_floats(v) for v in node]
    if isinstance(node, str) and _FLOAT.match(node):
        return float(node)
    return node


def _load(path: str):
    text = open(path).read()
    return _floats(yaml.safe_load(text) or {}), bool(_GLOBAL.search(text.split("\n\n", 1)[0]) or _GLOBAL.search(text[:200]))


def merge(dst: dict, src: dict) -> dict:
    """src over dst, dictionaries key by key, everything else replaced."""
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _group_file(root: str, group: str, name: str) -> str:
    name = str(name)
    for cand in (name, name + ".yaml", name + ".yml"):
        p = os.path.join(root, group, cand)
        if os.path.isfile(p):
            return p
    raise FileNotFoundError(f"config group '{group}' has no option '{name}' under {os.path.join(root, group)}")


def _place(cfg: dict, root: str, group: str, name) -> None:
    if name is None:  # `group: null` in a defaults list: nothing selected
        return
    node, is_global = _load(_group_file(root, group, name))
    node.pop("defaults", None)
    if is_global:
        merge(cfg, node)
    else:
        merge(cfg.setdefault(group, {}), node)


def _set(cfg: dict, dotted: str, value) -> None:
    keys = dotted.split(".")
    for k in keys[:-1]:
        nxt = cfg.get(k)
        if not isinstance(nxt, dict):
            nxt = cfg[k] = {}
        cfg = nxt
    cfg[keys[-1]] = value


def _get(cfg, dotted: str):
    for k in dotted.split("."):
        if not isinstance(cfg, dict) or k not in cfg:
            raise KeyError(dotted)
        cfg = cfg[k]
    return cfg


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _resolve(root: dict, node, depth: int = 0):
    if isinstance(node, dict):
        return {k: _resolve(root, v, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(root, v, depth) for v in node]
    if not isinstance(node, str) or "${" not in node or depth > 8:
        return node

    def one(expr: str):
        expr = expr.strip()
        if expr.startswith("hydra:"):
            if expr[6:] in ("runtime.cwd", "runtime.output_dir"):
                return os.getcwd()
            raise KeyError(expr)
        if expr.startswith("now:"):
            return datetime.datetime.now().strftime(expr[4:])
        if expr.startswith("oc.env:"):
            name, _, default = expr[7:].partition(",")
            return os.environ[name] if name in os.environ or not default else default
        return _get(root, expr)

    whole = _INTERP.fullmatch(node)
    try:
        if whole:  # the node IS one interpolation: keep the target's type
            return _resolve(root, one(whole.group(1)), depth + 1)
        out = _INTERP.sub(lambda m: str(_resolve(root, one(m.group(1)), depth + 1)), node)
    except KeyError:
        return node
    return _resolve(root, out, depth + 1) if out != node else out


def compose(path: str, overrides=()) -> dict:
    """The composed configuration of the primary file at ``path`` (see the module docstring)."""
    root = os.path.dirname(os.path.abspath(path))
    primary, _ = _load(path)
    defaults = primary.pop("defaults", None)
    cfg: dict = {}
    groups = {}  # group -> selected option, in defaults order
    if defaults is None:
        cfg = primary
    else:
        order = []
        for d in defaults:
            if d == "_self_":
                order.append("_self_")
            elif isinstance(d, dict) and len(d) == 1:
                (g, n), = d.items()
                g = str(g).replace("override ", "").replace("optional ", "").strip()
                groups[g] = n
                order.append(g)
            elif isinstance(d, str):  # a bare file next to the primary one
                order.append(("file", d))
            else:
                raise ValueError(f"unsupported defaults entry {d!r} in {path}")
        if "_self_" not in order:
            order.append("_self_")  # Hydra >= 1.1: the primary file composes last unless it says otherwise
        # group selections on the command line act on the defaults list, before anything is merged
        rest = []
        for ov in overrides:
            key, eq, val = ov.partition("=")
            plain = key.lstrip("+")
            if eq and "." not in plain and os.path.isdir(os.path.join(root, plain)) and not key.startswith("~"):
                if plain not in groups:
                    order.append(plain)
                groups[plain] = yaml.safe_load(val)
            else:
                rest.append(ov)
        overrides = rest
        for item in order:
            if item == "_self_":
                merge(cfg, primary)
            elif isinstance(item, tuple):
                node, _ = _load(_group_file(root, "", item[1]))
                merge(cfg, node)
            else:
                _place(cfg, root, item, groups[item])
    for ov in overrides:
        if ov.startswith("~"):
            keys = ov[1:].split(".")
            node = cfg
            for k in keys[:-1]:
                node = node.get(k, {})
            node.pop(keys[-1], None)
            continue
        key, eq, val = ov.partition("=")
        if not eq:
            raise ValueError(f"override {ov!r} is not key=value")
        _set(cfg, key.lstrip("+"), _floats(yaml.safe_load(val)))
    return _resolve(cfg, cfg)
