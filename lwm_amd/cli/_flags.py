"""absl/tux-style flags on argparse: `--name=value`, `--name value`, `--flag` / `--noflag` for
booleans, and the dotted config-dict flags (`--optimizer.adamw_optimizer.lr=8e-5`,
`--train_dataset.json_dataset.seq_length=2048`, ...) which tux.define_flags_with_default creates for
every ConfigDict-valued default (lwm/train.py:31-56)."""
from __future__ import annotations

import ast
import sys


def _literal(text):
    """'8e-5' -> 8e-5, 'True' -> True, \"'adamw'\" -> 'adamw', anything else stays a string."""
    t = text.strip()
    if t in ("True", "true"):
        return True
    if t in ("False", "false"):
        return False
    try:
        return ast.literal_eval(t)
    except (ValueError, SyntaxError):
        return t


class Flags(dict):
    __getattr__ = dict.__getitem__


def parse(defaults: dict, groups: tuple, argv=None, prog="") -> Flags:
    """defaults: {flag: default}; groups: names of the ConfigDict-valued flags whose dotted children
    are collected into nested dicts.  Unknown flags are an error, as with absl."""
    argv = list(sys.argv[1:] if argv is None else argv)
    out = Flags({k: v for k, v in defaults.items()})
    for g in groups:
        out[g] = {}
    i = 0
    while i < len(argv):
        a = argv[i]
        i += 1
        if a in ("-h", "--help", "--helpfull"):
            print(f"usage: python -m {prog} " + " ".join(f"--{k}={v!r}" for k, v in defaults.items()))
            print("config groups (dotted flags): " + ", ".join(groups))
            raise SystemExit(0)
        if not a.startswith("--"):
            raise SystemExit(f"{prog}: unexpected positional argument {a!r}")
        body = a[2:]
        if "=" in body:
            name, val = body.split("=", 1)
        elif body in defaults and isinstance(defaults[body], bool):
            name, val = body, "True"
        elif body.startswith("no") and body[2:] in defaults and isinstance(defaults[body[2:]], bool):
            name, val = body[2:], "False"
        else:
            if i >= len(argv):
                raise SystemExit(f"{prog}: flag --{body} needs a value")
            name, val = body, argv[i]
            i += 1
        head = name.split(".", 1)[0]
        if head in groups:
            node = out[head]
            parts = name.split(".")[1:]
            if not parts:
                raise SystemExit(f"{prog}: --{name} is a config group; set its fields (--{name}.field=value)")
            for p in parts[:-1]:
                node = node.setdefault(p, {})
            node[parts[-1]] = _literal(val)
        elif name in defaults:
            d = defaults[name]
            v = _literal(val)
            if isinstance(d, bool):
                v = bool(v)
            elif isinstance(d, int) and not isinstance(v, bool) and isinstance(v, (int, float)):
                v = int(v)
            elif isinstance(d, float) and isinstance(v, (int, float)):
                v = float(v)
            elif isinstance(d, str):
                v = val if not isinstance(v, str) else v
            out[name] = v
        else:
            raise SystemExit(f"{prog}: unknown flag --{name}")
    return out
