"""YAML + dotted-flag loader for the dataclass config trees of both entry scripts.

The reference parses its configs with pyrallis (`@pyrallis.wrap()`, scripts/train_latent_paint.py:8,
scripts/train_latent_nerf.py:8): `--config_path file.yaml` and/or `--section.field value`.  pyrallis is
not installed here, so the same two input forms are read with argparse + yaml; the dataclasses
themselves stay pyrallis-compatible."""
import argparse
import dataclasses
import sys
from enum import Enum
from pathlib import Path
from typing import get_type_hints

import yaml


def coerce(value, typ):
    """Turn a YAML / command-line value into the annotated type of a config field."""
    if typ is bool:
        return value if isinstance(value, bool) else str(value).lower() in ("1", "true", "yes", "y")
    if isinstance(typ, type) and issubclass(typ, Enum):
        return value if isinstance(value, typ) else typ(str(value))
    if typ is Path:
        return Path(value)
    origin = getattr(typ, "__origin__", None)
    if origin is tuple:
        if isinstance(value, str):
            value = [v for v in value.replace("(", "").replace(")", "").split(",") if v.strip()]
        return tuple(float(v) for v in value)
    if typ in (int, float, str):
        return typ(value)
    args = getattr(typ, "__args__", ())
    if type(None) in args:  # Optional[X]
        if value is None or str(value).lower() in ("none", "null"):
            return None
        return coerce(value, [a for a in args if a is not type(None)][0])
    return value


def apply_overrides(cfg, flat: dict):
    """flat: {'log.exp_name': 'x', 'render.nerf_type': 'latent', ...}.  Sets the fields, tells the config
    which ones the user set (`cfg.note_explicit(key)`, if the config has it) and re-runs `__post_init__`
    ONCE, after every override is in place."""
    for key, value in flat.items():
        section, _, name = key.partition(".")
        sub = getattr(cfg, section, None)
        if sub is None or not dataclasses.is_dataclass(sub) or name not in {f.name for f in dataclasses.fields(sub)}:
            raise KeyError("unknown config field %r" % key)
        setattr(sub, name, coerce(value, get_type_hints(type(sub))[name]))
        if hasattr(cfg, "note_explicit"):
            cfg.note_explicit(key, getattr(sub, name))
    if hasattr(cfg, "__post_init__"):
        cfg.__post_init__()
    return cfg


def load_config(root_cls, argv=None):
    """`--config_path file.yaml` and/or dotted flags `--section.field value` -> root_cls instance."""
    ap = argparse.ArgumentParser(add_help=True)
    ap.add_argument("--config_path", default=None)
    args, rest = ap.parse_known_args(argv)
    flat = {}
    if args.config_path:
        doc = yaml.safe_load(open(args.config_path)) or {}
        for section, body in doc.items():
            for name, value in (body or {}).items():
                flat["%s.%s" % (section, name)] = value
    i = 0
    while i < len(rest):
        tok = rest[i]
        if not tok.startswith("--"):
            raise SystemExit("unexpected argument %r" % tok)
        if "=" in tok:
            k, v = tok[2:].split("=", 1)
            i += 1
        else:
            k = tok[2:]
            if i + 1 >= len(rest):
                raise SystemExit("flag %r needs a value" % tok)
            v = rest[i + 1]
            i += 2
        flat[k] = v
    return apply_overrides(root_cls(), flat)


def to_plain_dict(cfg):
    """dataclass tree -> JSON/YAML-friendly dict (Paths and enums as strings)."""
    def conv(v):
        if isinstance(v, dict):
            return {k: conv(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [conv(x) for x in v]
        if isinstance(v, Enum):
            return v.value
        if isinstance(v, Path):
            return str(v)
        return v
    return conv(dataclasses.asdict(cfg))


def make_section(name, spec, namespace=None, doc=None):
    """A config section as a dataclass from a table of (field, type, default, help) rows -- the table keeps a
    section's CLI contract (names, types, defaults) and its help text in one place; `section_help(cls)` returns the
    help strings for `--help` output."""
    fields = []
    for fname, ftype, default, _help in spec:
        if isinstance(default, (list, dict, set)):
            fields.append((fname, ftype, dataclasses.field(default_factory=lambda d=default: type(d)(d))))
        else:
            fields.append((fname, ftype, dataclasses.field(default=default)))
    cls = dataclasses.make_dataclass(name, fields, namespace=dict(namespace or {}))
    cls.__module__ = sys._getframe(1).f_globals.get("__name__", cls.__module__)   # picklable / honest repr
    cls.__doc__ = doc or name
    cls.__field_help__ = {row[0]: row[3] for row in spec}
    return cls


def section_help(cls):
    return dict(getattr(cls, "__field_help__", {}))
