"""Minimal Hydra/OmegaConf-compatible composer for the OneProt config tree.

When hydra-core is installed use it; it is not in this environment, and the drop-in contract is that the reference's YAML
(`configs/model/*.yaml`, `configs/model/components/*.yaml`, ...) loads unchanged.  Supported subset -- exactly what those files use:
  * `defaults:` lists with `- group: option`, `- option` (same group), `- sub/option` (package = group.sub), `_self_`,
    `override /group: option`, optional `.yaml` suffix, `# @package _global_` headers;
  * interpolation `${a.b.c}`, relative `${.x}` / `${..x.y}`, `${oc.env:VAR}` / `${oc.env:VAR,default}`;
  * `instantiate`: `_target_` dotted path, recursive, `_partial_: true` -> functools.partial, `_recursive_`/`_convert_` ignored.
"""
import functools
import importlib
import os
import re

import yaml


def _load_yaml(path):
    with open(path) as f:
        data = yaml.safe_load(f)
    return data or {}


def _find(config_dir, rel):
    for cand in (rel, rel + ".yaml", rel + ".yml"):
        p = os.path.join(config_dir, cand)
        if os.path.isfile(p):
            return p
    raise FileNotFoundError(f"config '{rel}' not found under {config_dir}")


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def _place(root, package, content):
    node = root
    for part in [p for p in package.split(".") if p]:
        node = node.setdefault(part, {})
    _merge(node, content)


def _compose_file(config_dir, rel, package, root, choices):
    """Merge file `rel` (path relative to config_dir, no suffix needed) into `root` at dotted `package`."""
    path = _find(config_dir, rel)
    data = _load_yaml(path)
    with open(path) as f:
        head = f.readline()
    if "@package _global_" in head:
        package = ""
    defaults = data.pop("defaults", None) or []
    group = os.path.dirname(rel)
    own_done = False
    for item in defaults + ([] if "_self_" in defaults else ["_self_"]):
        if item == "_self_":
            _place(root, package, data)
            own_done = True
            continue
        if isinstance(item, dict):
            (k, v), = item.items()
            if v is None:
                continue
            k = k.replace("override ", "").replace("optional ", "").strip()
            v = choices.get(k.lstrip("/"), v)
            v = str(v)
            v = v[:-5] if v.endswith(".yaml") else v
            if k.startswith("/"):
                g = k.lstrip("/")
                _compose_file(config_dir, os.path.join(g, v), g.replace("/", "."), root, choices)
            else:
                sub_group = os.path.join(group, k) if group else k
                sub_pkg = (package + "." if package else "") + k.replace("/", ".") if package or True else k
                if not package and not group:
                    sub_pkg = k.replace("/", ".")
                _compose_file(config_dir, os.path.join(sub_group, v), sub_pkg, root, choices)
        else:
            name = str(item)
            name = name[:-5] if name.endswith(".yaml") else name
            sub = os.path.dirname(name)
            sub_pkg = package + ("." + sub.replace("/", ".") if sub else "") if package else sub.replace("/", ".")
            _compose_file(config_dir, os.path.join(group, name) if group else name, sub_pkg, root, choices)
    assert own_done
    return root


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _lookup(root, parts):
    node = root
    for p in parts:
        node = node[int(p)] if isinstance(node, list) else node[p]
    return node


def _resolve_str(s, root, path):
    def repl(m):
        expr = m.group(1).strip()
        if expr.startswith("oc.env:"):
            var, _, default = expr[7:].partition(",")
            val = os.environ.get(var.strip(), default.strip() if default else None)
            if val is None:
                raise KeyError(f"environment variable {var} is not set (needed by ${{{expr}}})")
            return val
        if expr.startswith("."):
            ndots = len(expr) - len(expr.lstrip("."))
            base = path[:-ndots] if ndots <= len(path) else []
            target = base + [p for p in expr.lstrip(".").split(".") if p]
        else:
            target = expr.split(".")
        val = _lookup(root, target)
        if isinstance(val, str) and "${" in val:
            val = _resolve_str(val, root, target)
        return val if m.group(0) == s else str(val)
    m = _INTERP.fullmatch(s)
    if m:
        return repl(m)
    return _INTERP.sub(lambda mm: str(repl(mm)), s)


def _resolve(node, root, path):
    if isinstance(node, dict):
        return {k: _resolve(v, root, path + [k]) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, path + [str(i)]) for i, v in enumerate(node)]
    if isinstance(node, str) and "${" in node:
        return _resolve_str(node, root, path)
    return node


def compose(config_dir, config_name, overrides=None, resolve=True):
    """overrides: {"group": "option"} choices and/or dotted "a.b.c": value assignments."""
    overrides = overrides or {}
    choices = {k: v for k, v in overrides.items() if "." not in k and os.path.isdir(os.path.join(config_dir, k))}
    root = {}
    name = config_name[:-5] if config_name.endswith(".yaml") else config_name
    group = os.path.dirname(name)
    _compose_file(config_dir, name, group.replace("/", "."), root, choices)
    for k, v in overrides.items():
        if k in choices:
            continue
        node = root
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return _resolve(root, root, []) if resolve else root


def _locate(dotted):
    mod, _, attr = dotted.rpartition(".")
    return getattr(importlib.import_module(mod), attr)


def instantiate(node, **kwargs):
    """hydra.utils.instantiate subset: recursive, `_partial_` aware."""
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    if not isinstance(node, dict):
        return node
    if "_target_" not in node:
        return {k: instantiate(v) for k, v in node.items()}
    args = {k: instantiate(v) for k, v in node.items() if k not in ("_target_", "_partial_", "_recursive_", "_convert_")}
    args.update(kwargs)
    fn = _locate(node["_target_"])
    if node.get("_partial_", False):
        return functools.partial(fn, **args)
    return fn(**args)
