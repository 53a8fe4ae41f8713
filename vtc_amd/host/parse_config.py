"""Config loading for the drop-in entry points: the subset of utils/parse_config.py the eval
path uses -- JSONC reading (comments + trailing commas, configs/pretrained_clip.jsonc:35-36;
the reference uses pyjson5, which is not installed here), ``a;b;c`` key-path CLI overrides
(utils/parse_config.py:162-187) and the name-based factory ``init_obj`` (:97-112)."""
from __future__ import annotations

import json
import re
from functools import reduce
from operator import getitem
from pathlib import Path


def read_jsonc(path) -> dict:
    text = Path(path).read_text()
    out, i, n, in_str = [], 0, len(text), False
    while i < n:                                   # strip // and /* */ comments outside strings
        c = text[i]
        if in_str:
            out.append(c)
            if c == "\\" and i + 1 < n:
                out.append(text[i + 1]); i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True; out.append(c)
        elif text.startswith("//", i):
            while i < n and text[i] != "\n":
                i += 1
            continue
        elif text.startswith("/*", i):
            i = text.index("*/", i) + 2
            continue
        else:
            out.append(c)
        i += 1
    clean = re.sub(r",(\s*[}\]])", r"\1", "".join(out))   # trailing commas
    clean = re.sub(r",(\s*[}\]])", r"\1", clean)
    return json.loads(clean)


class ConfigParser:
    def __init__(self, config: dict, resume=None, modification=None):
        self._config = dict(config)
        for k, v in (modification or {}).items():
            if v is not None:
                keys = k.split(";")
                reduce(getitem, keys[:-1], self._config)[keys[-1]] = v
        self.resume = Path(resume) if resume else None

    @classmethod
    def from_file(cls, path, resume=None, modification=None):
        return cls(read_jsonc(path), resume, modification)

    def init_obj(self, name, module, *args, **kwargs):
        """utils/parse_config.py:97-112: getattr(module, cfg[name]["type"])(*args, **cfg[name]["args"])."""
        module_name = self[name]["type"]
        module_args = dict(self[name].get("args", {}))
        assert all(k not in module_args for k in kwargs), "Overwriting kwargs given in config file is not allowed"
        module_args.update(kwargs)
        return getattr(module, module_name)(*args, **module_args)

    def __getitem__(self, name):
        return self._config[name]

    @property
    def config(self):
        return self._config
