"""Reader for the reference's ``confs/*.conf`` files (HOCON as pyhocon parses them, exp_runner_blending.py:38-44):
nested ``name { ... }`` objects, ``key = value`` / ``key: value`` fields separated by newlines or commas, ``#`` and
``//`` comments, unquoted / quoted strings, numbers, booleans (``True``/``true``…), lists, ``a.b = v`` dotted keys.
Substitutions, includes and multi-line strings are not used by any shipped conf and raise.  The result offers the
accessors the reference uses: ``conf['model.nerf']``, ``conf.get_int/float/bool/string/list(path, default=…)``."""
from __future__ import annotations

import re

_MISSING = object()


class ConfigTree(dict):
    def _walk(self, path):
        node = self
        for part in path.split("."):
            if not isinstance(node, dict) or not dict.__contains__(node, part):
                raise KeyError(path)
            node = dict.__getitem__(node, part)
        return node

    def __getitem__(self, path):
        return self._walk(path) if isinstance(path, str) and "." in path else dict.__getitem__(self, path)

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def __setitem__(self, path, value):
        if isinstance(path, str) and "." in path:
            head, _, last = path.rpartition(".")
            node = self
            for part in head.split("."):
                if not dict.__contains__(node, part) or not isinstance(dict.__getitem__(node, part), dict):
                    dict.__setitem__(node, part, ConfigTree())
                node = dict.__getitem__(node, part)
            dict.__setitem__(node, last, value)
        else:
            dict.__setitem__(self, path, value)

    def get(self, path, default=_MISSING):
        try:
            return self[path]
        except KeyError:
            if default is _MISSING:
                raise
            return default

    def _typed(self, path, default, conv):
        try:
            v = self[path]
        except KeyError:
            if default is _MISSING:
                raise
            return default
        return conv(v)

    def get_int(self, path, default=_MISSING):
        return self._typed(path, default, int)

    def get_float(self, path, default=_MISSING):
        return self._typed(path, default, float)

    def get_string(self, path, default=_MISSING):
        return self._typed(path, default, str)

    def get_list(self, path, default=_MISSING):
        return self._typed(path, default, list)

    def get_bool(self, path, default=_MISSING):
        def conv(v):
            if isinstance(v, str):
                return v.lower() in ("true", "yes", "on")
            return bool(v)
        return self._typed(path, default, conv)


_TOKEN = re.compile(r"""
    (?P<ws>[ \t\r]+) | (?P<comment>(\#|//)[^\n]*) | (?P<nl>\n) | (?P<punct>[{}\[\],=:]) |
    (?P<qstr>"(?:[^"\\]|\\.)*") | (?P<bare>[^\s{}\[\],=:"\#]+)
""", re.X)
_NUM = re.compile(r"^[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?)$")


def _scalar(words):
    text = " ".join(words)
    if len(words) == 1:
        w = words[0]
        if _NUM.match(w):
            return int(w) if re.match(r"^[+-]?\d+$", w) else float(w)
        if w.lower() in ("true", "yes", "on"):
            return True
        if w.lower() in ("false", "no", "off"):
            return False
        if w.lower() == "null":
            return None
    if "${" in text:
        raise ValueError("HOCON substitutions are not supported")
    return text


class _Parser:
    def __init__(self, text):
        self.toks = []
        pos = 0
        while pos < len(text):
            m = _TOKEN.match(text, pos)
            if not m:
                raise ValueError(f"conf: cannot tokenise at offset {pos}: {text[pos:pos + 20]!r}")
            pos = m.end()
            kind = m.lastgroup
            if kind in ("ws", "comment"):
                continue
            if kind == "bare" and m.group().startswith("//"):
                continue
            self.toks.append((kind, m.group()))
        self.i = 0

    def peek(self):
        return self.toks[self.i] if self.i < len(self.toks) else ("eof", "")

    def take(self):
        t = self.peek()
        self.i += 1
        return t

    def skip_sep(self):
        while self.peek()[0] == "nl" or self.peek() == ("punct", ","):
            self.i += 1

    def parse_object(self, closing):
        tree = ConfigTree()
        while True:
            self.skip_sep()
            kind, val = self.peek()
            if kind == "eof":
                if closing:
                    raise ValueError("conf: missing '}'")
                return tree
            if (kind, val) == ("punct", "}"):
                if not closing:
                    raise ValueError("conf: unexpected '}'")
                self.take()
                return tree
            if kind not in ("bare", "qstr"):
                raise ValueError(f"conf: expected a key, got {val!r}")
            key = self.take()[1]
            key = key[1:-1] if kind == "qstr" else key
            kind, val = self.peek()
            if (kind, val) == ("punct", "{"):
                self.take()
                value = self.parse_object(True)
            elif kind == "punct" and val in "=:":
                self.take()
                value = self.parse_value()
            else:
                raise ValueError(f"conf: expected '=', ':' or '{{' after key {key!r}")
            if isinstance(value, ConfigTree) and key in tree and isinstance(tree[key], ConfigTree):
                tree[key].update(value)            # HOCON merges repeated objects
            else:
                tree[key] = value

    def parse_value(self):
        kind, val = self.peek()
        if (kind, val) == ("punct", "{"):
            self.take()
            return self.parse_object(True)
        if (kind, val) == ("punct", "["):
            self.take()
            items = []
            while True:
                self.skip_sep()
                if self.peek() == ("punct", "]"):
                    self.take()
                    return items
                if self.peek()[0] == "eof":
                    raise ValueError("conf: missing ']'")
                items.append(self.parse_value())
        words = []
        while self.peek()[0] in ("bare", "qstr"):
            k, v = self.take()
            if k == "qstr":
                if words:
                    raise ValueError("conf: string concatenation is not supported")
                return bytes(v[1:-1], "utf-8").decode("unicode_escape")
            words.append(v)
        if not words:
            raise ValueError(f"conf: expected a value, got {val!r}")
        return _scalar(words)


def parse_string(text: str) -> ConfigTree:
    p = _Parser(text)
    return p.parse_object(False)


def parse_file(path: str, case: str = "CASE_NAME") -> ConfigTree:
    """read a conf the way the runner does (exp_runner_blending.py:38-45): CASE_NAME is replaced textually."""
    with open(path) as f:
        text = f.read()
    return parse_string(text.replace("CASE_NAME", case))
